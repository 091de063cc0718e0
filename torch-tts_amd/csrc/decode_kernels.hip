// Decode-step kernels for gfx950 (MI355X).  One decode step of the reference
// (tacotron/decoder.py:47-58 around tacotron/decoder_cell.py:180-195) is the launch sequence
//   frame (frame_kernel.hip: finish proj(t-1), PreNet) -> lstm(att) -> query -> attention+context
//   -> lstm(dec) -> proj                         [dims the frame kernel does not cover: prenet0 ->
//   prenet1 -> ... -> proj with its own epilogue]
// Kernel boundaries are the all-to-all seams of the step (every output row needs a whole
// hidden vector produced by all workgroups of the previous phase).  This file holds the row
// GEMM (PreNet / query / projection / Postnet / Encoder2 / VITS2 GEMMs), the LSTM cell, the
// attention kernel and the small bookkeeping / packing kernels.
#include "step_bodies.h"

namespace ttsdec {

// ===========================================================================
// generic row GEMM:  out[m, n] = epi( sum_k A[m, k] * W[n, k] )
// ===========================================================================
// implicit im2col for Conv1d(k, padding=(k-1)/2) on channel-last activations x [B*T, Cin]:
// with k = tap*Cin + c the im2col row of frame m is the contiguous window
// x[(m - taps/2)*Cin + k], valid while the tapped frame stays inside the utterance.
template <int EB>
struct LoaderConv {
  const void *x, *x_lo;
  int m0, M, T, Cin, taps, K;
  static constexpr bool kRange = true;
  __device__ __forceinline__ int nseg() const { return 1; }
  __device__ __forceinline__ int seglen(int i) const { return i == 0 ? K : 0; }
  __device__ __forceinline__ bool row_ok(int r) const { return m0 + r < M; }
  __device__ __forceinline__ gbyte* row_ptr(int r, int, int plane) const {
    return as_global(plane == 0 ? x : x_lo) + (((long)(m0 + r) - (taps >> 1)) * Cin) * EB;
  }
  __device__ __forceinline__ int k_lo(int r) const {
    const int t = (m0 + r) % T, half = taps >> 1;
    return (half - t > 0 ? half - t : 0) * Cin;
  }
  __device__ __forceinline__ int k_hi(int r) const {
    const int t = (m0 + r) % T, half = taps >> 1;
    const int hi = (T - t + half) * Cin;
    return hi < K ? hi : K;
  }
  __device__ __forceinline__ long col_off(int c16) const { return (long)c16 * 16; }
  __device__ __forceinline__ long tile_inc(int rowb) const { return rowb; }
  // buffer-descriptor form (gemm_tile kBufDma): based at THIS workgroup's first im2col row - so the per-lane offsets stay small
  // whatever the size of x (Postnet at 2048 x 600 frames: 2.5 GB) - which for the batch's first frames lies `taps/2` frames
  // in front of x: only lanes whose k is inside the row's window [k_lo, k_hi) are ever in range, and those address x itself
  __device__ __forceinline__ const void* seg_base(int, int plane) const {
    return static_cast<const char*>(plane == 0 ? x : x_lo) + ((long)m0 - (taps >> 1)) * Cin * EB;
  }
  __device__ __forceinline__ unsigned row_off(int r, int) const { return (unsigned)r * (unsigned)Cin * EB; }
};

// rows n0.. of a PyTorch-layout weight [N, K] cut into the same K segments as A
template <int EB>
struct LoaderW {
  Seg3 w, w_lo;
  int n0, N;
  static constexpr bool kRange = false;
  __device__ __forceinline__ int nseg() const { return seg_count(w); }
  __device__ __forceinline__ int seglen(int i) const { return seg_len(w, i); }
  __device__ __forceinline__ bool row_ok(int r) const { return n0 + r < N; }
  __device__ __forceinline__ gbyte* row_ptr(int r, int i, int plane) const {
    return seg_row_ptr<EB>(plane == 0 ? w : w_lo, n0 + r, i);
  }
  __device__ __forceinline__ long col_off(int c16) const { return (long)c16 * 16; }
  __device__ __forceinline__ long tile_inc(int rowb) const { return rowb; }
  // buffer-descriptor form (gemm_tile kBufDma; row-major weights): based at this workgroup's first weight row
  __device__ __forceinline__ const void* seg_base(int i, int plane) const {
    const Seg3& s = plane == 0 ? w : w_lo;
    const int ld = i == 0 ? s.ld0 : (i == 1 ? s.ld1 : s.ld2);
    return static_cast<const char*>(i == 0 ? s.p0 : (i == 1 ? s.p1 : s.p2)) + (long)n0 * ld * EB;
  }
  __device__ __forceinline__ unsigned row_off(int r, int i) const {
    const int ld = i == 0 ? w.ld0 : (i == 1 ? w.ld1 : w.ld2);
    return (unsigned)r * (unsigned)ld * EB;
  }
};

template <class Cfg, int AK, int EK>
__global__ __launch_bounds__(kGemmThreads, Cfg::kWavesPerSimd) void gemm_rows_kernel(GemmArgs g) {
  loop_stamp(g.ctrl, g.slot, g.node);
  bool live = true;
  if (g.ctrl != nullptr) {  // step kernel inside a decode call: "now" and the call's buffers come from *ctrl
    const Ctrl* c = g.ctrl;
    const StepNow now = step_now(c, g.slot);
    live = now.live;
    g.t = now.t;
    g.t_rel = now.t_rel;
    g.t_stride = c->t_stride;
    if (EK == EPI_RELU_DROPOUT) {
      g.dropout_mode = c->dropout_mode;
      g.seed = c->seed;
      g.masks = c->masks ? c->masks + (size_t)now.t_rel * g.mask_step_stride + g.mask_layer_off : nullptr;
      if (g.layer == 0) {
        g.teacher = c->teacher;
        g.teacher_T = c->teacher_T;
        g.teacher_flags = c->teacher_flags;
      }
    }
    if (EK == EPI_PROJ) {
      g.y_out = c->y;
      g.s_out = c->s;
      g.stop_thr = c->stop_thr;
      g.check_stop = c->check_stop;
    }
  }
  __shared__ __attribute__((aligned(16))) float smem[Cfg::kLdsFloats];
  constexpr int BM = Cfg::BM, BN = Cfg::BN, LDO = Cfg::LDO;
  constexpr int EPT = BM * BN / kGemmThreads;  // output elements per thread
  // XCD-aware tile order for the large GEMMs: workgroups are dealt round-robin over the 8 XCDs by
  // linear id, so with the natural order the column tiles that share one row tile's A operand land on
  // different XCDs and each L2 fetches that A tile again.  Remap so that all column tiles of a row tile
  // run on the same XCD: id -> (xcd = id % 8, slot = id / 8) -> row tile (slot / nx) * 8 + xcd.
  int bx = blockIdx.x, by = blockIdx.y;
  if (AK == A_CONV || Cfg::kBig) {
    const int nx = gridDim.x, ny = gridDim.y;
    const int id = by * nx + bx, full = (ny / 8) * 8;
    if (id < full * nx) {
      const int xcd = id & 7, slot = id >> 3;
      by = (slot / nx) * 8 + xcd;
      bx = slot % nx;
    }
  }
  const int m0 = by * BM;
  const int n0 = bx * BN;

  // Epilogue operands are requested BEFORE the K loop (their latency hides under it) - for the
  // small tiles, where a thread owns <= 8 outputs; the 128x128 tiles fetch them in the epilogue.
  // (not the lean tile: it runs at 128 registers so that two workgroups share a CU, and 4 x 8 pre-loaded operands across the K
  // loop made round 3's short-K row GEMM - gemm_rows_kernel<TileCfg<2,2,1,4,F16S,0,1,1,1>, A_PLAIN, EPI_GENERIC>, a quarter of a
  // VITS2 pass - spill 14 registers to scratch; its per-column operand is loaded once and the per-element ones in one batch
  // after the loop, like the 128 x 128 tiles')
  constexpr bool kPre = EPT <= 8 && Cfg::kWavesPerSimd < 4;
  constexpr int NPRE = kPre ? EPT : 1;
  auto load_epi = [&](int m, int n, bool ok, float& pb, float& pr, float& prm, uint8_t& pm) {
    pb = (ok && g.bias != nullptr) ? g.bias[n] : 0.f;
    pm = 1;
    pr = 0.f;
    prm = 1.0f;
    if (EK == EPI_RELU_DROPOUT && ok && g.dropout_mode == TTSDEC_DROPOUT_MASKS) pm = as_g(g.masks)[(size_t)m * g.N + n];
    if ((EK == EPI_BN_ISRU || EK == EPI_BN_LRELU || EK == EPI_BN_ISRLU) && ok) {
      pb = g.alpha[n];
      pr = g.beta[n];
    }
    if (EK == EPI_RESIDUAL && ok) pr = g.resid[(size_t)m * g.ldo + n];
    if (EK == EPI_GENERIC && ok) {
      if (g.resid != nullptr) pr = g.resid[(size_t)m * g.ldo + n];
      if (g.row_mask != nullptr) prm = g.row_mask[m];
    }
  };
  // 128x128 tiles: a thread's 32 outputs all sit in ONE column (512 threads = 4 x 128 columns), so the per-column operands
  // - bias, or the folded BatchNorm scale and shift - are requested once, before the K loop.  Fetched per output inside the
  // epilogue they were 64 loads with a wait each: a fifth of a Postnet layer's time (3.64 -> 3.0x ms).
  constexpr bool kColConst = !kPre && (kGemmThreads % BN == 0);
  constexpr bool kBnEpi = EK == EPI_BN_ISRU || EK == EPI_BN_LRELU || EK == EPI_BN_ISRLU;
  float col_b = 0.f, col_r = 0.f;
  if constexpr (kColConst) {
    const int n = n0 + (int)(threadIdx.x % BN);
    const bool ok = live && n < g.N;
    if constexpr (kBnEpi) {
      col_b = ok ? g.alpha[n] : 0.f;
      col_r = ok ? g.beta[n] : 0.f;
    } else {
      col_b = (ok && g.bias != nullptr) ? g.bias[n] : 0.f;
    }
  }
  float pre_bias[NPRE], pre_res[NPRE], pre_rm[NPRE];
  uint8_t pre_mask[NPRE];
  if constexpr (kPre) {
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      const int e = threadIdx.x + j * kGemmThreads;
      const int m = m0 + e / BN, n = n0 + e % BN;
      load_epi(m, n, live && m < g.M && n < g.N, pre_bias[j], pre_res[j], pre_rm[j], pre_mask[j]);
    }
  }

  constexpr int EB = Cfg::EB;
  if (AK == A_CONV) {
    const LoaderConv<EB> la{g.a.p0, g.a_lo.p0, m0, g.M, g.T, g.Cin, g.taps, g.K};
    const LoaderW<EB> lb{make_seg1(g.W, g.ldw, g.K), make_seg1(g.W_lo, g.ldw, g.K), n0, g.N};
    // (one K segment, both operands based at the workgroup's own first row: the buffer-descriptor loaders of gemm_tile.h apply
    // whatever the operand sizes)
    gemm_tile<Cfg, LoaderConv<EB>, LoaderW<EB>, NoGate, 2>(la, lb, smem, live);
  } else {
    Seg3 s = g.a, s_lo = g.a_lo;
    if (g.teacher != nullptr && g.t > 0 && as_g(g.teacher_flags)[g.t - 1] != 0) {
      // decoder.py:65-66: next input = last frame of teacher group t-1 = teacher frame t*r - 1 (fp32 path only)
      s = make_seg1(g.teacher + (size_t)(g.t * g.r - 1) * g.d_mel, g.teacher_T * g.d_mel, g.d_mel);
      s_lo = s;
    }
    int kbase = 0;
    if (EK == EPI_PLAIN && g.ksplit > 1) {  // this workgroup's slice of the K axis
      kbase = blockIdx.z * g.kchunk;
      const int kend = kbase + g.kchunk < g.K ? kbase + g.kchunk : g.K;
      s = seg_window(s, kbase, kend, EB);
      s_lo = seg_window(s_lo, kbase, kend, EB);
      g.out += (size_t)blockIdx.z * g.split_stride;
    }
    // W is one [N, K] matrix: cut it at A's segment boundaries
    const int k0 = s.e0, k1 = s.e1 - s.e0, k2 = s.e2 - s.e1;
    const char *w = static_cast<const char*>(g.W) + (size_t)kbase * EB, *wl = static_cast<const char*>(g.W_lo) + (size_t)kbase * EB;
    const Seg3 ws = make_seg3(w, g.ldw, k0, w + (size_t)k0 * EB, g.ldw, k1, w + (size_t)(k0 + k1) * EB, g.ldw, k2);
    const Seg3 wsl = make_seg3(wl, g.ldw, k0, wl + (size_t)k0 * EB, g.ldw, k1, wl + (size_t)(k0 + k1) * EB, g.ldw, k2);
    const LoaderW<EB> lb{ws, wsl, n0, g.N};
    const LoaderPlain<EB, true> la{s, s_lo, m0, g.M};
    gemm_tile<Cfg, LoaderPlain<EB, true>, LoaderW<EB>, NoGate, 2>(la, lb, smem, live);
  }
  if (!live) return;

  // 128x128 tiles: the per-ELEMENT operands (residual, row mask) of all 32 outputs are requested in one batch - inside the
  // loop below each was a load with its own wait
  constexpr int NBATCH = kColConst && (EK == EPI_RESIDUAL || EK == EPI_GENERIC) ? EPT : 1;
  float bat_res[NBATCH], bat_rm[NBATCH];
  if constexpr (NBATCH > 1) {
    const bool has_res = EK == EPI_RESIDUAL || g.resid != nullptr, has_rm = EK == EPI_GENERIC && g.row_mask != nullptr;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      const int e = threadIdx.x + j * kGemmThreads;
      const int m = m0 + e / BN, n = n0 + e % BN;
      const bool ok = m < g.M && n < g.N;
      bat_res[j] = (ok && has_res) ? g.resid[(size_t)m * g.ldo + n] : 0.f;
      bat_rm[j] = (ok && has_rm) ? g.row_mask[m] : 1.0f;
    }
  }
  // (the 128x128 tiles step the element offset from row to row instead of recomputing m * ldo + n in 64-bit arithmetic
  // for each of a thread's 32 outputs; the 16-bit output kind and the range check are resolved once)
  const size_t o_first = (size_t)(m0 + (int)(threadIdx.x / BN)) * g.ldo + n0 + (int)(threadIdx.x % BN);
  const size_t o_step = (size_t)(kGemmThreads / BN) * g.ldo;
  const int okind = g.out_kind;
  bool over = false;
  // A tile that lies wholly inside the matrix (all but the last row / column of tiles) takes a copy of the loop without the
  // per-element bounds test: with the test every output is its own basic block - LDS read, wait, arithmetic, stores, one
  // after the other, 32 times; without it the compiler batches the reads and overlaps the stores.
  auto emit = [&](auto whole_c) {
  constexpr bool kWholeTile = decltype(whole_c)::value;
  float vals[EPT];
#pragma unroll
  for (int j = 0; j < EPT; ++j) {
    const int e = threadIdx.x + j * kGemmThreads;
    const int row = e / BN, col = e % BN;
    const int m = m0 + row, n = n0 + col;
    if constexpr (!kWholeTile) {
      if (m >= g.M || n >= g.N) continue;
    }
    const size_t o = kColConst ? o_first + (size_t)j * o_step : (size_t)m * g.ldo + n;
    float v = smem[row * LDO + col];
    float pb, pr, prm;
    uint8_t pm;
    if constexpr (kPre) { pb = pre_bias[j]; pr = pre_res[j]; prm = pre_rm[j]; pm = pre_mask[j]; }
    else if constexpr (kColConst) {
      pb = col_b; pr = col_r; prm = 1.0f; pm = 1;
      if constexpr (NBATCH > 1) { pr = bat_res[j]; prm = bat_rm[j]; }
      if (EK == EPI_RELU_DROPOUT && g.dropout_mode == TTSDEC_DROPOUT_MASKS) pm = as_g(g.masks)[(size_t)m * g.N + n];
    } else load_epi(m, n, true, pb, pr, prm, pm);
    if (EK != EPI_BN_ISRU && EK != EPI_BN_LRELU && EK != EPI_BN_ISRLU && g.bias != nullptr) v = add_rn(v, pb);
    auto store16 = [&](float val) { vals[j] = val; };  // 16-bit copies for a following 16-bit GEMM: second loop, below
    if (EK == EPI_PLAIN) {
      g.out[o] = v;
      store16(v);
    } else if (EK == EPI_RELU_DROPOUT) {
      // modules.py:39-40: relu then dropout(p, always): kept units scaled by 1/(1-p)
      v = v > 0.f ? v : 0.f;
      if (g.dropout_mode == TTSDEC_DROPOUT_MASKS) {
        v = pm ? mul_rn(v, g.keep_scale) : 0.f;
      } else if (g.dropout_mode == TTSDEC_DROPOUT_PHILOX) {
        v = philox_keep(g.seed, (uint32_t)g.t, (uint32_t)g.layer, (uint32_t)m, (uint32_t)n) ? mul_rn(v, g.keep_scale) : 0.f;
      }
      g.out[o] = v;
      store16(v);
    } else if (EK == EPI_PROJ) {
      const int nm = g.r * g.d_mel;
      if (n < nm) {
        // decoder.py:53-54: leaky_relu(fc_mel(d_t), 0.01) viewed as [B, r, d_mel]
        v = v > 0.f ? v : mul_rn(v, 0.01f);
        const int jf = n / g.d_mel, c = n - jf * g.d_mel;
        as_g(g.y_out)[((size_t)m * g.t_stride * g.r + (size_t)g.t_rel * g.r + jf) * g.d_mel + c] = v;
        if (jf == g.r - 1) g.ynext[(size_t)m * g.d_mel + c] = v;  // decoder.py:48 y_t[:, -1, :]
      } else {
        // decoder.py:52 stop logit; decoder.py:68 batch-global rule
        const int jf = n - nm;
        as_g(g.s_out)[(size_t)m * g.t_stride * g.r + (size_t)g.t_rel * g.r + jf] = v;
        if (g.check_stop && v < g.stop_thr) atomicMin(&g.ctrl->stop_t, g.t);
      }
    } else if (EK == EPI_BN_ISRU) {
      // modules.py:181 isru(BatchNorm1d(conv(x))) with eval-mode BN as x*alpha + beta
      v = isru_fast(add_rn(mul_rn(v, pb), pr));
      if (g.out != nullptr) g.out[o] = v;
      store16(v);
    } else if (EK == EPI_BN_ISRLU) {
      // encoder.py:49-57 ISRLU(BatchNorm1d(conv(x))): x >= 0 ? x : x / sqrt(1 + x*x)  (activations.py:13-14)
      v = add_rn(mul_rn(v, pb), pr);
      v = v >= 0.f ? v : isru_fast(v);
      if (g.out != nullptr) g.out[o] = v;
      store16(v);
    } else if (EK == EPI_BN_LRELU) {
      // modules.py:196-198 LeakyReLU(BatchNorm1d(conv(x))), default negative_slope 0.01
      v = add_rn(mul_rn(v, pb), pr);
      v = v > 0.f ? v : mul_rn(v, 0.01f);
      if (g.out != nullptr) g.out[o] = v;
      store16(v);
    } else if (EK == EPI_GENERIC) {
      if (g.act == 1) v = v > 0.f ? v : 0.f;
      if (g.row_mask != nullptr) v = mul_rn(v, prm);
      if (g.resid != nullptr) v = add_rn(pr, v);
      g.out[o] = v;
      store16(v);
    } else if (EK == EPI_RESIDUAL) {
      // modules.py:184 x + fc_out(...)  /  modules.py:215 x + layer(x)
      v = add_rn(pr, v);
      g.out[o] = v;
      store16(v);
    }
  }
  // the 16-bit copies in a loop of their own per kind: the kind is the same for every output, and tested per output it cut
  // the loop above into one basic block per element
  if constexpr (EK != EPI_PROJ) {
    auto each = [&](auto&& f) {
#pragma unroll
      for (int j = 0; j < EPT; ++j) {
        const int e = threadIdx.x + j * kGemmThreads;
        const int m = m0 + e / BN, n = n0 + e % BN;
        if constexpr (!kWholeTile) {
          if (m >= g.M || n >= g.N) continue;
        }
        f(kColConst ? o_first + (size_t)j * o_step : (size_t)m * g.ldo + n, vals[j]);
      }
    };
    if (okind == 1) each([&](size_t o, float val) { split_f16_flag(val, g.out_h[o], g.out_l[o], over); });  // (saturating; reported once, below)
    else if (okind == 2) each([&](size_t o, float val) { reinterpret_cast<bf16*>(g.out_h)[o] = (bf16)val; });
  }
  };
  if (m0 + BM <= g.M && n0 + BN <= g.N) emit(std::true_type{});
  else emit(std::false_type{});
  report_range(over, g.ctrl);
}

template <int AK, int EK, int PREC>
static void launch_gemm_cfg(const GemmArgs& a, hipStream_t st) {
  // Pick the tile so the grid covers the 256 CUs when it can; small M uses 32x32 tiles
  // with the four MFMA waves splitting K.
  const long tiles_big = (long)((a.M + 63) / 64) * ((a.N + 63) / 64);
  if constexpr (PREC == PREC_F16S && EK == EPI_PLAIN) {
    // the step's small split-K GEMMs (query, mel/stop projection): 32x32 tiles on two MFMA waves that split a
    // 128-element K tile, so the grid stays as wide as the exact path's (these launches are latency-bound)
    if (tiles_big < 512) {
      using Cfg = TileCfg<1, 1, 2, 4, PREC_F16S>;
      dim3 grid((a.N + Cfg::BN - 1) / Cfg::BN, (a.M + Cfg::BM - 1) / Cfg::BM, a.ksplit > 1 ? a.ksplit : 1);
      hipLaunchKernelGGL((gemm_rows_kernel<Cfg, AK, EK>), grid, dim3(kGemmThreads), 0, st, a);
      return;
    }
  }
  if constexpr (PREC != PREC_F32) {
    // (256x128 tiles for bf16 - TileCfg<2, 2, 1, 4, PREC_BF16, 0, 4, 2, 1> - were measured: 1.58 ms against 1.35 ms for
    // the Postnet at 256 x 600 frames.  Their 132-KiB output tile leaves one workgroup per CU where the 128x128 tile
    // fits two, and the second workgroup hides more latency than the bigger tile saves in staged bytes.)
    // large 16-bit GEMMs (Postnet convs): 128x128 tiles
    if constexpr (PREC == PREC_F16S && AK == A_PLAIN && EK == EPI_GENERIC) {
      // short-K row GEMMs (the VITS2 1x1 convs, K = 192: six k32 tiles): the lean 64x64 tile, two workgroups per CU, so
      // that one workgroup's loads and stores run beside the other's MFMAs - on the 128x128 tile these launches were
      // 8.8 % MFMA-busy.  VITS2 pass 8.39 -> 8.25 ms (same box; TTSDEC_NO_LEAN_SKINNY=1 restores the big tile).
      static const bool lean_skinny = getenv("TTSDEC_NO_LEAN_SKINNY") == nullptr;
      if (lean_skinny && a.K <= 256 && a.M >= 2048) {
        using Cfg = TileCfg<2, 2, 1, 4, PREC_F16S, 0, 1, 1, 1>;
        dim3 grid((a.N + Cfg::BN - 1) / Cfg::BN, (a.M + Cfg::BM - 1) / Cfg::BM);
        hipLaunchKernelGGL((gemm_rows_kernel<Cfg, AK, EK>), grid, dim3(kGemmThreads), 0, st, a);
        return;
      }
    }
    if (a.N >= 128 && a.M >= 2048) {
      // 128x128 tiles, full-depth stages (64 k), double buffered: half the per-stage barriers of the four half-depth stages
      // this ran on until round 3 for the same bytes in flight (Postnet 256 x 600: split-fp16 3.44 -> 3.32 ms, bf16 1.26 ->
      // 1.19 ms, VITS2 flow 6.18 -> 6.13 ms, same box, bit-identical; profiles/r03_x_big_tile_full_depth.txt)
      using Cfg = TileCfg<2, 2, 1, 2, PREC, 0, 2, 2, 0>;
      dim3 grid((a.N + Cfg::BN - 1) / Cfg::BN, (a.M + Cfg::BM - 1) / Cfg::BM);
      hipLaunchKernelGGL((gemm_rows_kernel<Cfg, AK, EK>), grid, dim3(kGemmThreads), 0, st, a);
      return;
    }
  }
  if (PREC != PREC_F32 || (a.M >= 64 && tiles_big >= 512)) {
    // 64x64 tiles; ring: fp32 4 x 16 KiB, bf16 5 x 16 KiB, split-fp16 4 x 32 KiB
    using Cfg = TileCfg<2, 2, 1, (PREC == PREC_BF16 ? 5 : 4), PREC>;
    dim3 grid((a.N + Cfg::BN - 1) / Cfg::BN, (a.M + Cfg::BM - 1) / Cfg::BM, (EK == EPI_PLAIN && a.ksplit > 1) ? a.ksplit : 1);
    hipLaunchKernelGGL((gemm_rows_kernel<Cfg, AK, EK>), grid, dim3(kGemmThreads), 0, st, a);
  } else {
    using Cfg = TileCfg<1, 1, 4, 4>;
    dim3 grid((a.N + Cfg::BN - 1) / Cfg::BN, (a.M + Cfg::BM - 1) / Cfg::BM, (EK == EPI_PLAIN && a.ksplit > 1) ? a.ksplit : 1);
    hipLaunchKernelGGL((gemm_rows_kernel<Cfg, AK, EK>), grid, dim3(kGemmThreads), 0, st, a);
  }
}

template <int AK, int EK>
static void launch_gemm_prec(const GemmArgs& a, hipStream_t st) {
  if (a.prec == PREC_F16S) return launch_gemm_cfg<AK, EK, PREC_F16S>(a, st);
  if (a.prec == PREC_BF16) return launch_gemm_cfg<AK, EK, PREC_BF16>(a, st);
  return launch_gemm_cfg<AK, EK, PREC_F32>(a, st);
}

void launch_gemm(const GemmArgs& a, AKind ak, EpiKind ek, hipStream_t st) {
  if (a.M <= 0 || a.N <= 0) return;
  if (ak == A_CONV && ek == EPI_BN_ISRU) return launch_gemm_prec<A_CONV, EPI_BN_ISRU>(a, st);
  if (ak == A_CONV && ek == EPI_BN_LRELU) return launch_gemm_prec<A_CONV, EPI_BN_LRELU>(a, st);
  if (ak == A_CONV && ek == EPI_BN_ISRLU) return launch_gemm_prec<A_CONV, EPI_BN_ISRLU>(a, st);
  if (ak == A_CONV && ek == EPI_RESIDUAL) return launch_gemm_prec<A_CONV, EPI_RESIDUAL>(a, st);
  if (ak == A_CONV && ek == EPI_GENERIC) return launch_gemm_prec<A_CONV, EPI_GENERIC>(a, st);
  if (ek == EPI_GENERIC) return launch_gemm_prec<A_PLAIN, EPI_GENERIC>(a, st);
  switch (ek) {
    case EPI_PLAIN:
      if (a.prec == PREC_F16S) return launch_gemm_cfg<A_PLAIN, EPI_PLAIN, PREC_F16S>(a, st);
      return launch_gemm_cfg<A_PLAIN, EPI_PLAIN, PREC_F32>(a, st);
    case EPI_RELU_DROPOUT: return launch_gemm_cfg<A_PLAIN, EPI_RELU_DROPOUT, PREC_F32>(a, st);
    case EPI_PROJ: return launch_gemm_cfg<A_PLAIN, EPI_PROJ, PREC_F32>(a, st);
    case EPI_RESIDUAL: return launch_gemm_prec<A_PLAIN, EPI_RESIDUAL>(a, st);
    default: return;
  }
}

// ===========================================================================
// LSTM cell launches (body: step_bodies.h lstm_body)
// ===========================================================================
// TAG only names the instantiation (0 = attention LSTM / generic, 1 = decoder LSTM) so that profilers
// list the two cells of a decode step as separate kernels.
template <class Cfg, int TAG = 0>
__global__ __launch_bounds__(kGemmThreads) void lstm_kernel(LstmArgs g) {
  loop_stamp(g.ctrl, g.slot, g.node);
  __shared__ __attribute__((aligned(16))) float smem[Cfg::kLdsFloats];
  lstm_body<Cfg>(g, smem, blockIdx.x, blockIdx.y);
}
template <class Cfg>
static void launch_lstm_tagged(const LstmArgs& a, dim3 grid, hipStream_t st) {
  if (a.tag == 1) hipLaunchKernelGGL((lstm_kernel<Cfg, 1>), grid, dim3(kGemmThreads), 0, st, a);
  else hipLaunchKernelGGL((lstm_kernel<Cfg, 0>), grid, dim3(kGemmThreads), 0, st, a);
}
// two independent cells in one launch (blockIdx.z picks one): the two directions of a BiLSTM step
struct LstmPair {
  LstmArgs d[2];
};
template <class Cfg>
__global__ __launch_bounds__(kGemmThreads) void lstm_pair_kernel(LstmPair p) {
  __shared__ __attribute__((aligned(16))) float smem[Cfg::kLdsFloats];
  lstm_body<Cfg>(blockIdx.z ? p.d[1] : p.d[0], smem, blockIdx.x, blockIdx.y);
}

void launch_lstm_pair(const LstmArgs& a0, const LstmArgs& a1, hipStream_t st) {
  // same shape, exact fp32 (Encoder2's recurrence); anything else goes out as two launches
  if (a0.M <= 0) return;
  if (a0.prec != 0 || a1.prec != 0 || a0.M != a1.M || a0.H != a1.H || a0.K != a1.K) {
    launch_lstm(a0, st);
    launch_lstm(a1, st);
    return;
  }
  LstmPair p;
  p.d[0] = a0;
  p.d[1] = a1;
  if (a0.M >= 192) {
    using Cfg = TileCfg<2, 2, 1, 6>;
    hipLaunchKernelGGL((lstm_pair_kernel<Cfg>), dim3((a0.H + 15) / 16, (a0.M + 63) / 64, 2), dim3(kGemmThreads), 0, st, p);
  } else if (a0.M >= 96) {
    using Cfg = TileCfg<2, 1, 2, 4>;
    hipLaunchKernelGGL((lstm_pair_kernel<Cfg>), dim3((a0.H + 7) / 8, (a0.M + 63) / 64, 2), dim3(kGemmThreads), 0, st, p);
  } else {
    using Cfg = TileCfg<1, 1, 4, 4>;
    hipLaunchKernelGGL((lstm_pair_kernel<Cfg>), dim3((a0.H + 7) / 8, (a0.M + 31) / 32, 2), dim3(kGemmThreads), 0, st, p);
  }
}

void launch_lstm(const LstmArgs& a, hipStream_t st) {
  if (a.M <= 0) return;
  // (more than two 64-row blocks of 8-unit tiles would be more workgroups than CUs - a second round: 45.8 us for the decoder
  // LSTM at B = 160 against 25.8 at B = 128 - so anything above 128 rows takes the 16-unit tile)
  if (a.prec == 1) {
    if (a.M > 128) {
      using Cfg = TileCfg<2, 2, 1, 4, PREC_F16S>;  // 64 rows x 16 units, 32 KiB stages (64 k each); S=5 and nt weight loads measured slower
      dim3 grid((a.H + 15) / 16, (a.M + 63) / 64);
      launch_lstm_tagged<Cfg>(a, grid, st);
    } else if (a.M >= 96) {
      using Cfg = TileCfg<2, 1, 2, 3, PREC_F16S>;  // 64 rows x 8 units, 48 KiB stages (128 k each)
      dim3 grid((a.H + 7) / 8, (a.M + 63) / 64);
      launch_lstm_tagged<Cfg>(a, grid, st);
    } else {
      // small batches: 32 rows x 8 units on two MFMA waves, so B = 64 still launches 256 workgroups
      // (the kernel is bound by streaming its weight rows, which this halves per workgroup)
      using Cfg = TileCfg<1, 1, 2, 4, PREC_F16S>;  // 32 KiB stages (128 k each)
      dim3 grid((a.H + 7) / 8, (a.M + 31) / 32);
      launch_lstm_tagged<Cfg>(a, grid, st);
    }
    return;
  }
  if (a.M > 128) {
    using Cfg = TileCfg<2, 2, 1, 6>;  // 64 rows x 16 units, 16 KiB stages
    dim3 grid((a.H + 15) / 16, (a.M + 63) / 64);
    launch_lstm_tagged<Cfg>(a, grid, st);
  } else if (a.M >= 96) {
    using Cfg = TileCfg<2, 1, 2, 4>;  // 64 rows x 8 units, 24 KiB stages
    dim3 grid((a.H + 7) / 8, (a.M + 63) / 64);
    launch_lstm_tagged<Cfg>(a, grid, st);
  } else {
    using Cfg = TileCfg<1, 1, 4, 4>;  // 32 rows x 8 units, 32 KiB stages
    dim3 grid((a.H + 7) / 8, (a.M + 31) / 32);
    launch_lstm_tagged<Cfg>(a, grid, st);
  }
}

// ===========================================================================
// attention + context launch (body: step_bodies.h attn_body)
// ===========================================================================
template <int NJ>
__global__ __launch_bounds__(kAttnThreads) void attn_kernel(AttnArgs g) {
  loop_stamp(g.ctrl, g.slot, g.node);
  __shared__ __attribute__((aligned(16))) float part[attn_lds_floats<NJ>()];
  attn_body<NJ>(g, part, blockIdx.x);
}

void launch_attn(const AttnArgs& a, hipStream_t st) {
  if (a.B <= 0) return;
  const int d4 = a.D / 4;
  dim3 grid(a.B), block(kAttnThreads);
  if (d4 <= 64) hipLaunchKernelGGL((attn_kernel<1>), grid, block, 0, st, a);
  else if (d4 <= 128) hipLaunchKernelGGL((attn_kernel<2>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((attn_kernel<4>), grid, block, 0, st, a);
}

// ===========================================================================
// state init (decoder_cell.py:9-17,165-178; decoder.py:35) and bookkeeping
// ===========================================================================
__global__ void init_state_kernel(InitArgs g) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    g.ctrl->stop_t = kStopNever;
    g.ctrl->steps_done = 0;
    g.ctrl->range_err = 0;
  }
  const size_t nHa = (size_t)g.B * g.Ha, nHd = (size_t)g.B * g.Hd;
  if (i < nHa) {
    g.h_att[i] = g.h0_att[i % g.Ha];
    g.c_att[i] = g.c0_att[i % g.Ha];
    if (g.h_att_h != nullptr) {
      const size_t o = g.out_mpad > 0 ? chunk_idx((int)(i / g.Ha), (int)(i % g.Ha), g.out_mpad) : i;
      split_f16(g.h0_att[i % g.Ha], g.h_att_h[o], g.h_att_l[o]);
    }
  }
  if (i < nHd) {
    g.h_dec[i] = g.h0_dec[i % g.Hd];
    g.c_dec[i] = g.c0_dec[i % g.Hd];
    if (g.h_dec_h != nullptr) {
      const size_t o = g.out_mpad > 0 ? chunk_idx((int)(i / g.Hd), (int)(i % g.Hd), g.out_mpad) : i;
      split_f16(g.h0_dec[i % g.Hd], g.h_dec_h[o], g.h_dec_l[o]);
    }
  }
  if (i < (size_t)g.B * g.D) {
    // w_0 is one-hot at position 0, so bmm(w_0, memory) is memory[:, 0, :]
    const float c0 = g.memory ? g.memory[(i / g.D) * (size_t)g.L * g.D + (i % g.D)] : 0.f;
    g.ctx[i] = c0;
    if (g.ctx_h != nullptr) {
      const size_t o = g.out_mpad > 0 ? chunk_idx((int)(i / g.D), (int)(i % g.D), g.out_mpad) : i;
      split_f16(c0, g.ctx_h[o], g.ctx_l[o]);
    }
  }
  if (i < (size_t)g.B * g.L) g.w[i] = (i % g.L == 0) ? 1.0f : 0.f;
  if (i < (size_t)g.B * g.d_mel) g.ynext[i] = 0.f;
}

void launch_init(const InitArgs& a, hipStream_t st) {
  size_t n = (size_t)a.B * a.Ha;
  const size_t c[] = {(size_t)a.B * a.Hd, (size_t)a.B * a.D, (size_t)a.B * a.L, (size_t)a.B * a.d_mel};
  for (size_t v : c) n = v > n ? v : n;
  hipLaunchKernelGGL(init_state_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
}

__global__ void finish_kernel(Ctrl* ctrl, int32_t* T_out) {
  const int stop = ctrl->stop_t;
  const int t_end = ctrl->t_end;
  const int fired = (stop != kStopNever && stop < t_end) ? 1 : 0;
  const int done = fired ? stop + 1 : t_end;
  ctrl->steps_done = done;
  if (T_out) {
    T_out[0] = done;
    T_out[1] = fired | ((ctrl->range_err & 1) ? 2 : 0) | ((ctrl->range_err & 2) ? 4 : 0);
  }
}

void launch_finish(Ctrl* ctrl, int32_t* T_out, hipStream_t st) {
  hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(1), 0, st, ctrl, T_out);
}

__global__ void set_call_kernel(Ctrl* c, CallArgs a) {
  c->t_cur = a.t_begin;
  c->t_call = a.t_begin;
  c->t_end = a.t_begin + a.n_steps;
  c->t_stride = a.t_stride;
  c->check_stop = a.check_stop;
  c->dropout_mode = a.dropout_mode;
  c->stop_thr = a.stop_thr;
  c->teacher_T = a.teacher_T;
  c->seed = a.seed;
  c->memory = a.memory;
  c->masks = a.masks;
  c->teacher = a.teacher;
  c->teacher_flags = a.teacher_flags;
  c->y = a.y;
  c->s = a.s;
  c->w = a.w;
  c->stamps = a.stamps;
  c->debug_flags = a.debug_flags;
  c->spin_limit = a.spin_limit;
  c->loop_stamps = a.loop_stamps;
}
void launch_set_call(Ctrl* ctrl, const CallArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(set_call_kernel, dim3(1), dim3(1), 0, st, ctrl, a);
}

__global__ void advance_kernel(Ctrl* c, int n_slots) { c->t_cur += n_slots; }
void launch_advance(Ctrl* ctrl, int n_slots, hipStream_t st) {
  hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, st, ctrl, n_slots);
}

// ===========================================================================
// weight packing helpers
// ===========================================================================
__global__ void add_vec_kernel(const float* a, const float* b, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = add_rn(a[i], b[i]);
}
void launch_add_vec(const float* a, const float* b, float* out, int n, hipStream_t st) {
  hipLaunchKernelGGL(add_vec_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a, b, out, n);
}

__global__ void copy_kernel(const float* src, float* dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
void launch_copy(const float* src, float* dst, size_t n, hipStream_t st) {
  if (n == 0 || src == nullptr) return;
  hipLaunchKernelGGL(copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n);
}

__global__ void split_kernel(const float* src, f16* hi, f16* lo, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) split_f16(src[i], hi[i], lo[i]);
}
void launch_split(const float* src, f16* hi, f16* lo, size_t n, hipStream_t st) {
  if (n == 0 || src == nullptr) return;
  hipLaunchKernelGGL(split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, hi, lo, n);
}

// The PreNet weights in the order the frame kernel's lanes take them (frame_body.h: lane (l32, half) of wave w holds a K run of
// weight row n): piece (w, j) = the 64 lanes' 16 bytes back to back, so every weight load of the kernel is one coalesced KiB.
// Row-per-lane loads touched 32-40 cache lines per instruction; beside an LSTM workgroup on the same CU that cost the LSTM's tile
// stream ~2 us per step at B = 256 (profiles/r03_u_*).
//   layer 0 (W [PH, K0], K0H = K0 / 2):  n = 32 w + l32, k = half K0H + 8 j + e       -> ((w NW0 + j) 64 + 32 half + l32) 8 + e
//   layer 1 (W [P, PH], KQ = PH / 4):    n = 64 bx + 32 (w & 1) + l32, k = (w >> 1) KQ + half KQ/2 + 8 j + e
//                                                                         -> (((8 bx + w) NW1 + j) 64 + 32 half + l32) 8 + e
// (rows past P of the last 64-column block stay zero: the destination is cleared first)
__global__ void split_frame_order_kernel(const float* src, f16* hi, f16* lo, int N, int K, int layer) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * K) return;
  const int n = (int)(i / K), k = (int)(i % K);
  size_t o;
  if (layer == 0) {
    const int K0H = K / 2, NW0 = K0H / 8;
    const int w = n >> 5, l32 = n & 31, half = k / K0H, j = (k % K0H) >> 3, e = k & 7;
    o = ((size_t)((w * NW0 + j) * 64 + 32 * half + l32)) * 8 + e;
  } else {
    const int KQ = K / 4, K1H = KQ / 2, NW1 = K1H / 8;
    const int bx = n >> 6, wl = (n >> 5) & 1, l32 = n & 31, wh = k / KQ, half = (k % KQ) / K1H, j = (k % K1H) >> 3, e = k & 7;
    o = ((size_t)(((8 * bx + 2 * wh + wl) * NW1 + j) * 64 + 32 * half + l32)) * 8 + e;
  }
  split_f16(src[i], hi[o], lo[o]);
}
void launch_split_frame_order(const float* src, f16* hi, f16* lo, int N, int K, int layer, hipStream_t st) {
  const size_t n = (size_t)N * K;
  if (n == 0 || src == nullptr) return;
  hipLaunchKernelGGL(split_frame_order_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, hi, lo, N, K, layer);
}

__global__ void split_chunked_kernel(const float* src, f16* hi, f16* lo, int M, int K, int mpad) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * K) return;
  const int m = (int)(i / K), k = (int)(i % K);
  const size_t o = chunk_idx(m, k, mpad);
  split_f16(src[i], hi[o], lo[o]);
}
void launch_split_chunked(const float* src, f16* hi, f16* lo, int M, int K, int mpad, hipStream_t st) {
  const size_t n = (size_t)M * K;
  if (n == 0 || src == nullptr) return;
  hipLaunchKernelGGL(split_chunked_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, hi, lo, M, K, mpad);
}

__global__ void pack_lstm_chunked_kernel(const float* src, f16* hi, f16* lo, int H, int K) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)4 * H * K) return;
  const int row = (int)(i / K), k = (int)(i % K);
  const int gate = row / H, u = row % H;
  const size_t o = (((size_t)(u >> 4) * (K >> 5) + (k >> 5)) * 64 + gate * 16 + (u & 15)) * kChunkK + (k & 31);
  split_f16(src[i], hi[o], lo[o]);
}
void launch_pack_lstm_chunked(const float* src, f16* hi, f16* lo, int H, int K, hipStream_t st) {
  const size_t n = (size_t)4 * H * K;
  if (n == 0 || src == nullptr) return;
  hipLaunchKernelGGL(pack_lstm_chunked_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, hi, lo, H, K);
}

// max |x| over a tensor, folded into *out (a non-negative float's bits order like an unsigned int)
__global__ void absmax_kernel(const float* src, size_t n, unsigned int* out) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float a = fabsf(src[i]);
    m = (a > m || a != a) ? a : m;  // a NaN weight poisons the maximum on purpose
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float o = __shfl_xor(m, off, 64);
    m = (o > m || o != o) ? o : m;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}
void launch_absmax(const float* src, size_t n, float* out, hipStream_t st) {
  if (n == 0 || src == nullptr) return;
  const unsigned blocks = (unsigned)((n + 256 * 16 - 1) / (256 * 16));
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks < 1024 ? blocks : 1024), dim3(256), 0, st, src, n, reinterpret_cast<unsigned int*>(out));
}

__global__ void embed_kernel(const long long* ids, const float* table, int n_table, int n_rows, int E, float* out_a, int lda,
                             float* out_b, int ldb, int* status) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n_rows * E) return;
  const int m = (int)(i / E), c = (int)(i % E);
  long long id = ids[m];
  // an id outside the table (nn.Embedding raises IndexError there): never read outside the table - clamp - and tell the host
  // through the status word, which it reads with the results (no host-side scan of the ids, no sync before the launch)
  if (id < 0 || id >= n_table) {
    if (c == 0 && status != nullptr) atomicOr(status, 1);
    id = id < 0 ? 0 : n_table - 1;
  }
  const float v = table[(size_t)id * E + c];
  out_a[(size_t)m * lda + c] = v;
  out_b[(size_t)m * ldb + c] = v;
}
void launch_embed(const long long* ids, const float* table, int n_table, int n_rows, int E, float* out_a, int lda, float* out_b,
                  int ldb, int* status, hipStream_t st) {
  const size_t n = (size_t)n_rows * E;
  hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ids, table, n_table, n_rows, E, out_a, lda, out_b,
                     ldb, status);
}

__global__ void fill_rows_kernel(float* dst, const float* row, int n_rows, int n_cols) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)n_rows * n_cols) dst[i] = row[i % n_cols];
}
void launch_fill_rows(float* dst, const float* row, int n_rows, int n_cols, hipStream_t st) {
  const size_t n = (size_t)n_rows * n_cols;
  hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dst, row, n_rows, n_cols);
}

__global__ void to_bf16_kernel(const float* src, bf16* dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (bf16)src[i];
}
void launch_to_bf16(const float* src, void* dst, size_t n, hipStream_t st) {
  if (n == 0 || src == nullptr) return;
  hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, static_cast<bf16*>(dst), n);
}

__global__ void conv_transpose_kernel(const float* w, float* out, int Co, int Ci, int k) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)Co * Ci * k;
  if (i >= n) return;
  // out[co][tap][ci] = w[co][ci][tap]
  const int ci = (int)(i % Ci);
  const int tap = (int)((i / Ci) % k);
  const int co = (int)(i / ((size_t)Ci * k));
  out[i] = w[((size_t)co * Ci + ci) * k + tap];
}
void launch_conv_transpose(const float* w, float* out, int Co, int Ci, int k, hipStream_t st) {
  const size_t n = (size_t)Co * Ci * k;
  hipLaunchKernelGGL(conv_transpose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, out, Co, Ci, k);
}

__global__ void conv1dfix_pack_kernel(const float* w, float* out, int Co, int Ci, int k) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)Co * Ci * k;
  if (i >= n) return;
  const int ci = (int)(i % Ci);
  const int tap = (int)((i / Ci) % k);
  const int co = (int)(i / ((size_t)Ci * k));
  out[i] = w[(size_t)co * Ci * k + (size_t)(k - 1 - tap) * Ci + ci];
}
void launch_conv1dfix_pack(const float* w, float* out, int Co, int Ci, int k, hipStream_t st) {
  const size_t n = (size_t)Co * Ci * k;
  hipLaunchKernelGGL(conv1dfix_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, out, Co, Ci, k);
}

__global__ void bn_fold_kernel(const float* gamma, const float* betap, const float* mean, const float* var, float eps,
                               float* alpha, float* beta, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // eval-mode BatchNorm1d as y = x*alpha + beta, alpha = gamma/sqrt(var+eps), beta = b - mean*alpha
  const float invstd = div_rn(1.0f, sqrt_rn(add_rn(var[i], eps)));
  const float a = gamma ? mul_rn(invstd, gamma[i]) : invstd;  // affine=False: gamma = 1, beta = 0
  alpha[i] = a;
  beta[i] = sub_rn(betap ? betap[i] : 0.f, mul_rn(mean[i], a));
}
void launch_bn_fold(const float* gamma, const float* betap, const float* mean, const float* var, float eps, float* alpha,
                    float* beta, int n, hipStream_t st) {
  hipLaunchKernelGGL(bn_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, st, gamma, betap, mean, var, eps, alpha, beta, n);
}

}  // namespace ttsdec
