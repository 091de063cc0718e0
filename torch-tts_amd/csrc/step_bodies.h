// Device-side bodies of the decode step's big kernels, shared by their stand-alone launches
// (decode_kernels.hip) and by the launches that run two independent roles side by side (fused_kernels.hip).
// Each body takes its LDS as a pointer into the calling kernel's ONE __shared__ array and its tile
// coordinates as arguments, so a kernel can give different workgroups different roles.
#pragma once
#include "gemm_tile.h"
#include "kernels.h"

namespace ttsdec {

// rows m0.. of an [M, K] activation made of up to three K segments; EB-byte elements,
// plane 1 (fp16 lo) comes from a second Seg3 of identical shape
// kWgBase: the buffer-descriptor form is based at THIS workgroup's first row (per-lane offsets stay small whatever the operand's
// size: the generic row GEMMs); false: at the segment's own pointer, the row block's offset riding in the per-lane offsets (the
// LSTM kernels: library-internal operands far below 2 GiB - and six 64-bit base computations fewer in kernels that sit at the
// scalar-register limit; with them the split-fp16 two-role kernels spilled to scratch and the step went from 61.6 to 89.5 us).
template <int EB, bool kWgBase = false>
struct LoaderPlain {
  Seg3 s, s_lo;
  int m0, M;
  static constexpr bool kRange = false;
  __device__ __forceinline__ int nseg() const { return seg_count(s); }
  __device__ __forceinline__ int seglen(int i) const { return seg_len(s, i); }
  __device__ __forceinline__ bool row_ok(int r) const { return m0 + r < M; }
  __device__ __forceinline__ gbyte* row_ptr(int r, int i, int plane) const {
    return seg_row_ptr<EB>(plane == 0 ? s : s_lo, m0 + r, i);
  }
  __device__ __forceinline__ int k_lo(int) const { return 0; }
  __device__ __forceinline__ int k_hi(int) const { return 0; }
  __device__ __forceinline__ long col_off(int c16) const { return seg_col_off(s, c16); }
  __device__ __forceinline__ long tile_inc(int rowb) const { return seg_tile_inc(s, rowb); }
  // buffer-descriptor form (gemm_tile kBufDma): the segment's base (uniform) and a row's byte offset from it
  __device__ __forceinline__ const void* seg_base(int i, int plane) const {
    const Seg3& q = plane == 0 ? s : s_lo;
    const char* p = static_cast<const char*>(i == 0 ? q.p0 : (i == 1 ? q.p1 : q.p2));
    if constexpr (!kWgBase) return p;
    const int ld = i == 0 ? q.ld0 : (i == 1 ? q.ld1 : q.ld2);
    return p + (s.mpad > 0 ? (long)m0 * kChunkBytes : (long)m0 * ld * EB);
  }
  __device__ __forceinline__ unsigned row_off(int r, int i) const {
    const int ld = i == 0 ? s.ld0 : (i == 1 ? s.ld1 : s.ld2);
    const int row = kWgBase ? r : m0 + r;
    return s.mpad > 0 ? (unsigned)row * kChunkBytes : (unsigned)row * (unsigned)ld * EB;
  }
};

// ===========================================================================
// LSTMZoneoutCell (eval), tacotron/modules/rnn.py:24-39.  A workgroup owns BU hidden
// units x BM batch rows and computes all four gates of those units (B-tile rows are
// gathered from the i/f/g/o row blocks of the PyTorch-layout weights), so the cell
// update happens in the epilogue without another pass.
// ===========================================================================
template <int BU, int EB>
struct LoaderWLstm {
  Seg3 w, w_lo;
  int u0, H;
  static constexpr bool kRange = false;
  __device__ __forceinline__ int nseg() const { return seg_count(w); }
  __device__ __forceinline__ int seglen(int i) const { return seg_len(w, i); }
  __device__ __forceinline__ bool row_ok(int r) const { return u0 + (r % BU) < H; }
  __device__ __forceinline__ gbyte* row_ptr(int r, int i, int plane) const {
    const Seg3& s = plane == 0 ? w : w_lo;
    if (s.mpad != 0) {
      // chunked weights (pack_lstm_chunked_kernel): [16-unit block][chunk][gate * 16 + unit % 16][32 k]; the segment
      // pointer is chunk 0 of the segment in unit block 0, ld the matrix's chunks per unit block
      const void* p = i == 0 ? s.p0 : (i == 1 ? s.p1 : s.p2);
      const int nch = i == 0 ? s.ld0 : (i == 1 ? s.ld1 : s.ld2);
      const int u = u0 + (r % BU);
      return as_global(p) + ((long)(u >> 4) * nch * 64 + (r / BU) * 16 + (u & 15)) * kChunkBytes;
    }
    return seg_row_ptr<EB>(s, (r / BU) * H + u0 + (r % BU), i);  // PyTorch gate blocks i,f,g,o
  }
  __device__ __forceinline__ long col_off(int c16) const { return w.mpad != 0 ? (long)(c16 >> 2) * 64 * kChunkBytes + (c16 & 3) * 16 : (long)c16 * 16; }
  __device__ __forceinline__ long tile_inc(int rowb) const { return w.mpad != 0 ? (long)(rowb / kChunkBytes) * 64 * kChunkBytes : (long)rowb; }
  // buffer-descriptor form (gemm_tile kBufDma)
  __device__ __forceinline__ const void* seg_base(int i, int plane) const {
    const Seg3& s = plane == 0 ? w : w_lo;
    return i == 0 ? s.p0 : (i == 1 ? s.p1 : s.p2);
  }
  __device__ __forceinline__ unsigned row_off(int r, int i) const {
    const int ld = i == 0 ? w.ld0 : (i == 1 ? w.ld1 : w.ld2);  // row-major: leading dimension; chunked: chunks per 16-unit block
    const int u = u0 + (r % BU);
    if (w.mpad != 0) return ((unsigned)(u >> 4) * (unsigned)ld * 64u + (unsigned)((r / BU) * 16 + (u & 15))) * kChunkBytes;
    return ((unsigned)(r / BU) * (unsigned)H + (unsigned)u) * (unsigned)ld * EB;
  }
};

// smem: Cfg::kLdsFloats floats of LDS (the caller's ONE shared array); (bx, by): unit block / row block.
// SC1 (the LSTM roles of the two-role launches): every A-operand tile is taken by sc1 LDS-DMA - past this CU's vector L1, served
// by L2 / the fabric - so the gate is the poll alone: the producers store the handed-off planes write-through and drain them
// before they signal (role_signal), the polling wave issues its loads behind its poll and the other waves behind the
// barrier it then joins (gemm_tile gate_sync).  The agent-scope acquire it replaces (buffer_inv sc1 + the wait for it) took
// the critical path of both launches 1-2.5 us per step (time stamps: gate reached -> passed).
template <bool SC1>
struct RoleGateT {
  static constexpr int kAuxA = SC1 ? 16 : 0;
  int seg, seg2;
  const unsigned int *counter, *counter1;  // the arrival counters of this workgroup's (up to two) 32-row blocks
  unsigned int target, target1;
  const unsigned int *counter2, *counter21;  // the same for the second gated segment (seg2; LstmArgs::dep2_*)
  unsigned int target2, target21;
  Ctrl* ctrl;
  int kind;
  stamp_ptr st;  // (loaded by the caller, once: see common.h stamp)
  // K-loop progress (measurement only): the time at K tiles 16 and 32, stamps 6 and 7 (the second gate's slots: not both at once)
  __device__ __forceinline__ void mark(int t) const {
    if (st != nullptr && seg2 < 0) {
      // (with the shader clock beside the 100-MHz one: in-kernel clock = d s_memtime / d s_memrealtime x 100 MHz, MI355X_MICROARCH.md DVFS (6))
      if (t == 16) { stamp(st, kind, 6, now_rt()); stamp(st, kind + 3, 0, __builtin_amdgcn_s_memtime()); }
      if (t == 32) { stamp(st, kind, 7, now_rt()); stamp(st, kind + 3, 1, __builtin_amdgcn_s_memtime()); }
    }
  }
  __device__ __forceinline__ void wait(int which) const {
    if (which == 1) {
      stamp(st, kind, 6, now_rt());
      if constexpr (SC1) role_poll(counter2, target2, ctrl, counter21, target21);
      else role_wait(counter2, target2, ctrl, counter21, target21);
      stamp(st, kind, 7, now_rt());
      return;
    }
    stamp(st, kind, 3, now_rt());
    if constexpr (SC1) role_poll(counter, target, ctrl, counter1, target1);
    else role_wait(counter, target, ctrl, counter1, target1);
    stamp(st, kind, 4, now_rt());
  }
};

// kEarlyEpi: request the cell-update operands before the K loop (their latency hides under it); the lean tile
// requests them after it instead - 20 live registers fewer across the loop, to stay within 128 VGPRs.
// kWhole: the caller only ever runs whole cells (mode 0, no sequence mode): the parked partial sums and their registers
// drop out, which is what lets the lean tile of the two-role launches request the rest of the operands early.
template <class Cfg, bool kEarlyEpi = true, bool kWhole = false>
__device__ __forceinline__ void lstm_body(LstmArgs g, float* smem, int bx, int by) {
  if constexpr (kWhole) { g.mode = 0; g.seq_lens = nullptr; g.seq_out = nullptr; }
  bool live = true;
  if (g.ctrl != nullptr) {
    // live_lag: this launch also holds the frame kernel's workgroups, which may lower stop_t to t-1 while it is
    // being read here; t-1 <= stop_t cannot be changed by that, so every wave takes the same decision (a wave
    // that disagreed with its workgroup about `live` would miss the tile loop's barriers)
    const int t = g.ctrl->t_cur + g.slot;
    live = t < g.ctrl->t_end && t - (g.live_lag ? 1 : 0) <= g.ctrl->stop_t;
  }
  constexpr int BM = Cfg::BM, BN = Cfg::BN, LDO = Cfg::LDO, BU = BN / 4, EB = Cfg::EB;
  constexpr int EPT = (BM * BU + kGemmThreads - 1) / kGemmThreads;  // (row, unit) pairs per thread
  const int m0 = by * BM;
  const int u0 = bx * BU;
  const int H = g.H;

  float pb[EPT][4], pc[EPT], ph[EPT], pp[EPT][4];
  auto load_epi = [&]() {
#pragma unroll
  for (int j = 0; j < EPT; ++j) {
    const int e = threadIdx.x + j * kGemmThreads;
    const int m = m0 + e / BU, unit = u0 + e % BU;
    const bool ok = live && e < BM * BU && m < g.M && unit < H && g.mode != 1;
    const size_t idx = (size_t)m * H + unit;
    size_t pidx = (size_t)m * 4 * H + unit;
    const float* part = g.partial;
    bool pok = ok && g.mode == 2;
    if (g.seq_lens != nullptr && ok) {  // packed-sequence step: this row's input projection at its own position
      const int len = g.seq_lens[m];
      const int pos = g.seq_reverse ? len - 1 - g.seq_t : g.seq_t;
      pok = g.seq_t < len;
      part = g.gx;
      pidx = ((size_t)m * g.seq_L + (pok ? pos : 0)) * g.gx_ld + g.gx_off + unit;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pb[j][k] = (ok && g.bsum != nullptr) ? g.bsum[k * H + unit] : 0.f;
      pp[j][k] = pok ? part[pidx + (size_t)k * H] : 0.f;
    }
    pc[j] = ok ? g.c[idx] : 0.f;
    ph[j] = ok ? g.h_prev[idx] : 0.f;
  }
  };
  if constexpr (kEarlyEpi) load_epi();

  const LoaderPlain<EB> la{g.a, g.a_lo, m0, g.M};
  const LoaderWLstm<BU, EB> lb{g.w, g.w_lo, u0, g.H};
  RoleGateT<kWhole> gate;  // (kWhole = a role of a two-role launch)
  const stamp_ptr st = g.dep_n > 0 ? stamps_of(g.ctrl) : (stamp_ptr) nullptr;  // measurement only (TTSDEC_STAMPS)
  gate.seg = gate.seg2 = -1; gate.counter = gate.counter1 = gate.counter2 = gate.counter21 = nullptr;
  gate.target = gate.target1 = gate.target2 = gate.target21 = 0; gate.ctrl = g.ctrl; gate.kind = g.dep_which; gate.st = st;
  if (threadIdx.x == 0) {
    stamp(st, g.dep_which, 0, (1ull << 32) | __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)));
    stamp(st, g.dep_which, 1, __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)));
    stamp(st, g.dep_which, 2, now_rt());
  }
  if (g.dep_n > 0) {
    gate.seg = g.dep_seg;
    if (g.ctrl != nullptr) {  // (no control block = profiling: the producers' data is whatever the last launch left)
      // the 32-row blocks this tile's rows [m0, m0 + BM) lie in: one (BM = 32) or two (BM = 64); arrivals per block and step:
      // dep_n, or the block's number of batch rows (dep_rows)
      static_assert(BM == 32 || BM == 64, "gated LSTM tiles span one or two 32-row blocks");
      const unsigned int steps = (unsigned int)(g.ctrl->t_cur + g.slot - g.ctrl->t_call + 1);
      auto per_block = [&](int rb) {
        const int rows = g.M - 32 * rb;
        return rows <= 0 ? 0u : (unsigned int)(g.dep_rows ? (rows < 32 ? rows : 32) : g.dep_n);
      };
      const int rb = m0 / 32;
      gate.counter = g.dep_cnt + rb * kDepLine;
      gate.target = steps * per_block(rb);
      if (BM == 64 && per_block(rb + 1) > 0) {
        gate.counter1 = g.dep_cnt + (rb + 1) * kDepLine;
        gate.target1 = steps * per_block(rb + 1);
      }
      if (g.dep2_n > 0) {  // (one-launch step: the h_att segment, dep2_n tiles of the attention LSTM per 32-row block)
        gate.seg2 = g.dep2_seg;
        gate.counter2 = g.dep2_cnt + rb * kDepLine;
        gate.target2 = steps * (unsigned int)g.dep2_n;
        if (BM == 64 && per_block(rb + 1) > 0) {
          gate.counter21 = g.dep2_cnt + (rb + 1) * kDepLine;
          gate.target21 = steps * (unsigned int)g.dep2_n;
        }
      }
    }
  }
  gemm_tile<Cfg, LoaderPlain<EB>, LoaderWLstm<BU, EB>, RoleGateT<kWhole>, 1>(la, lb, smem, live, g.dbg, gate);
  if (!live) return;
  if constexpr (!kEarlyEpi) load_epi();

#pragma unroll
  for (int j = 0; j < EPT; ++j) {
    const int e = threadIdx.x + j * kGemmThreads;
    const int row = e / BU, u = e % BU;
    const int m = m0 + row, unit = u0 + u;
    if (e >= BM * BU || m >= g.M || unit >= H) continue;
    const float* tr = smem + row * LDO;
    float si = tr[0 * BU + u], sf = tr[1 * BU + u], sg = tr[2 * BU + u], so = tr[3 * BU + u];
    const size_t pidx = (size_t)m * 4 * H + unit;
    if (g.mode == 1) {  // early part: park the raw gate sums
      g.partial[pidx] = si;
      g.partial[pidx + H] = sf;
      g.partial[pidx + 2 * H] = sg;
      g.partial[pidx + 3 * H] = so;
      continue;
    }
    const size_t idx = (size_t)m * H + unit;
    int seq_pos = 0;
    if (g.seq_lens != nullptr) {
      const int len = g.seq_lens[m];
      if (g.seq_t >= len) {  // this utterance has ended: state is carried unchanged, nothing is emitted
        g.h_out[idx] = ph[j];
        continue;
      }
      seq_pos = g.seq_reverse ? len - 1 - g.seq_t : g.seq_t;
    }
    if (g.mode == 2) {  // finishing part: early sums + the late segments (fixed order: deterministic)
      si = add_rn(pp[j][0], si);
      sf = add_rn(pp[j][1], sf);
      sg = add_rn(pp[j][2], sg);
      so = add_rn(pp[j][3], so);
    }
    const float gi = add_rn(si, pb[j][0]);
    const float gf = add_rn(sf, pb[j][1]);
    const float gg = add_rn(sg, pb[j][2]);
    const float go = add_rn(so, pb[j][3]);
    const float c_prev = pc[j];
    const float h_prev = ph[j];
    // nn.LSTMCell: c' = sigmoid(f)*c + sigmoid(i)*tanh(g); h' = sigmoid(o)*tanh(c')
    const float c_new = add_rn(mul_rn(sigmoid_fast(gf), c_prev), mul_rn(sigmoid_fast(gi), tanh_fast(gg)));
    const float h_new = mul_rn(sigmoid_fast(go), tanh_fast(c_new));
    // rnn.py:36-38 eval-mode zoneout: p*prev + (1-p)*new
    const float q = sub_rn(1.0f, g.pz);
    const float h = add_rn(mul_rn(g.pz, h_prev), mul_rn(q, h_new));
    g.h_out[idx] = h;
    if (g.h_out_h != nullptr) {
      const size_t o = g.out_mpad > 0 ? chunk_idx(m, unit, g.out_mpad) : idx;
      if (kWhole && g.sig_cnt != nullptr) {  // consumed by other workgroups of this very launch: write-through (role_signal)
        f16 hi, lo;
        split_f16(h, hi, lo);
        store_wt(g.h_out_h + o, hi);
        store_wt(g.h_out_l + o, lo);
      } else {
        split_f16(h, g.h_out_h[o], g.h_out_l[o]);
      }
    }
    if (g.seq_out != nullptr) g.seq_out[((size_t)m * g.seq_Lout + seq_pos) * g.seq_out_ld + g.seq_out_off + unit] = h;
    g.c[idx] = add_rn(mul_rn(g.pz, c_prev), mul_rn(q, c_new));
  }
  if constexpr (kWhole) {
    if (g.sig_cnt != nullptr && !(g.ctrl != nullptr && (g.ctrl->debug_flags & 8))) {
      static_assert(BM == 32 || BM == 64, "signalling LSTM tiles span one or two 32-row blocks");
      const int rb = m0 / 32;
      role_signal2(g.sig_cnt + rb * kDepLine, (BM == 64 && g.M > 32 * (rb + 1)) ? g.sig_cnt + (rb + 1) * kDepLine : nullptr);
    }
  }
  if (threadIdx.x == 0) stamp(st, g.dep_which, 5, now_rt());
}

// ===========================================================================
// StepwiseMonotonicAttention + context (tacotron/modules/attention.py:104-126,
// tacotron/decoder_cell.py:189).  One workgroup per utterance, 8 waves; a wave
// owns a contiguous range of memory rows and makes ONE pass over them: each row
// is loaded once (float4 per lane, coalesced), dotted with q (wave butterfly),
// turned into p0 and the new weight, and accumulated into the context while it
// is still in registers.  Row l needs p0[l-1], so a wave recomputes the energy
// of the row just before its range.
// ===========================================================================
// part: kAttnThreads / 64 * NJ * 256 floats of LDS (the caller's ONE shared array); b: the utterance.
template <int NJ>
__device__ __forceinline__ void attn_body(AttnArgs g, float* part, int b) {
  // Rows that can carry weight at this step: the stepwise-monotonic update (attention.py:119-123) moves weight by at most one
  // row per step from the one-hot start, so after t steps w_prev is EXACTLY zero beyond row t, w_new beyond row t + 1 - rows
  // past that add exact zeros to the context and need neither their energies nor their memory (245 KB per utterance and step
  // otherwise, all of it Infinity-Cache traffic beside the decoder LSTM's weight stream).  Known only with a control block
  // (t counts from the utterance's first step); the wave ranges stay as they are, so every sum keeps its order.
  int row_lim = g.L;
  if (g.ctrl != nullptr) {
    const Ctrl* c = g.ctrl;
    StepNow now = step_now(c, g.slot);
    if (now.t + 2 < row_lim) row_lim = now.t + 2;
    if (g.live_lag) now.live = now.t < c->t_end && now.t - 1 <= c->stop_t;  // (a role of the one-launch step: see lstm_body)
    if (!now.live) return;
    g.memory = c->memory;
    g.w_out = c->w;
    g.t_rel = now.t_rel;
    g.t_stride = c->t_stride;
  }
  constexpr int NW = kAttnThreads / 64;
  const stamp_ptr st = g.dep_signal ? stamps_of(g.ctrl) : (stamp_ptr) nullptr;  // measurement only (TTSDEC_STAMPS)
  if (threadIdx.x == 0) {
    stamp(st, 1, 0, __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)));
    stamp(st, 1, 1, __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)));
    stamp(st, 1, 2, now_rt());
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int L = g.L, D = g.D, D4 = D >> 2;
  const auto mem = as_g(g.memory) + (size_t)b * L * D;  // (g.memory comes from the control block: see as_g)
  const float* wprev = g.w_prev + (size_t)b * L;
  float* wnew = g.w_new + (size_t)b * L;
  const auto wout = as_g(g.w_out ? g.w_out + ((size_t)b * g.t_stride + g.t_rel) * L : nullptr);

  // q of this utterance: from an earlier launch, or (query role) from workgroups of this very launch - then behind the
  // arrival counter of the utterance's 32-row block and with sc1 loads (common.h load_wt; the slabs were stored write-through
  // and drained before each signal)
  const bool q_here = g.q_tiles > 0 && g.ctrl != nullptr;
  if (q_here) {
    if (wv == 0) role_poll(g.q_cnt + (b / 32) * kDepLine, (unsigned int)(g.t_rel + 1) * (unsigned int)g.q_wait_n, g.ctrl);
    lds_barrier();
  }
  auto load_q4 = [&](const float* p) {
    if (q_here) return make_float4(load_wt(p), load_wt(p + 1), load_wt(p + 2), load_wt(p + 3));
    return *reinterpret_cast<const float4*>(p);
  };
  float4 qv[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c4 = lane + 64 * j;
    qv[j] = (c4 < D4 && !g.ctx_only) ? load_q4(g.q + (size_t)b * D + c4 * 4)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);  // ctx_only: no query, energies unused
  }
  if (!g.ctx_only) {  // split-K query: add the partial slabs in index order
    for (int z = 1; z < g.q_parts; ++z) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c4 = lane + 64 * j;
        if (c4 < D4) {
          const float4 v = load_q4(g.q + z * g.q_stride + (size_t)b * D + c4 * 4);
          qv[j].x = add_rn(qv[j].x, v.x); qv[j].y = add_rn(qv[j].y, v.y);
          qv[j].z = add_rn(qv[j].z, v.z); qv[j].w = add_rn(qv[j].w, v.w);
        }
      }
    }
  }
  auto load_row = [&](int l, float4 (&r)[NJ]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c4 = lane + 64 * j;
      if (c4 < D4) {
        const f32x4 v = *reinterpret_cast<__attribute__((address_space(1))) const f32x4*>(mem + (size_t)l * D + c4 * 4);
        r[j] = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        r[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto dot_row = [&](const float4 (&r)[NJ]) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      s = fmaf(r[j].x, qv[j].x, s);
      s = fmaf(r[j].y, qv[j].y, s);
      s = fmaf(r[j].z, qv[j].z, s);
      s = fmaf(r[j].w, qv[j].w, s);
    }
    return wave_sum(s);
  };

  const int chunk = (L + NW - 1) / NW;
  const int l0 = wv * chunk;
  const int l1_all = (l0 + chunk < L) ? l0 + chunk : L;  // this wave's rows [l0, l1_all): all get their (possibly zero) weight stored

  float4 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);

  // The previous weights of this wave's rows, l0-1 .. l1-1, in ONE coalesced load (lane i holds w_prev[l0 - 1 + i]); the new
  // ones are gathered the same way and stored after the loop: a store inside the loop sits in the same in-order queue as the
  // next group's row loads, whose wait then also waits for the store.
  const int nrows = l1_all > l0 ? l1_all - l0 : 0;
  const bool lanes_hold_w = chunk < 64;  // (rows per wave + 1 <= 64 lanes; longer memories take the per-row accesses)
  // ... of which [l0, l1) can be non-zero (the per-row store path of long memories walks all rows)
  const int l1 = (lanes_hold_w && row_lim < l1_all) ? (row_lim > l0 ? row_lim : l0) : l1_all;
  float wp_lane = 0.f, wn_lane = 0.f;
  if (lanes_hold_w && lane <= nrows && l0 - 1 + lane >= 0 && l0 - 1 + lane < L) wp_lane = wprev[l0 - 1 + lane];
  auto w_prev_of = [&](int l) {  // l in [l0 - 1, l1): uniform
    return lanes_hold_w ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wp_lane), l - l0 + 1)) : wprev[l];
  };

  if (l0 < l1) {
    float w1_prev = 0.f;  // w[l-1] * (1 - p0[l-1])   attention.py:120
    if (l0 > 0) {
      float4 r[NJ];
      load_row(l0 - 1, r);
      const float p0 = isru_sigmoid(dot_row(r));  // l0-1 < L-1, never the overridden column
      w1_prev = mul_rn(w_prev_of(l0 - 1), sub_rn(1.0f, p0));
    }
    constexpr int G = 4;  // rows in flight per wave (8 measured slower, 17.2 vs 15.2 us; all 15 rows of a wave at once 18.2 vs 15.6 us at
                          // B = 256 and no faster at B = 1: the pass runs at the Infinity-Cache rate, not at a latency chain's)
    for (int lb = l0; lb < l1; lb += G) {
      float4 r[G][NJ];
      float e[G];
#pragma unroll
      for (int i = 0; i < G; ++i)
        if (lb + i < l1) load_row(lb + i, r[i]);
#pragma unroll
      for (int i = 0; i < G; ++i)
        if (lb + i < l1) e[i] = dot_row(r[i]);
#pragma unroll
      for (int i = 0; i < G; ++i) {
        const int l = lb + i;
        if (l < l1) {
          const float en = (l == L - 1) ? 1e4f : e[i];  // attention.py:117
          const float p0 = isru_sigmoid(en);            // attention.py:118
          const float wl = w_prev_of(l);
          const float w0 = mul_rn(wl, p0);                      // :119
          float wn = (l > 0) ? add_rn(w0, w1_prev) : w0;        // :122-123
          w1_prev = mul_rn(wl, sub_rn(1.0f, p0));               // :120
          if (g.ctx_only) wn = wl;
          if (lanes_hold_w) {
            if (lane == l - l0) wn_lane = wn;
          } else if (lane == 0 && !g.ctx_only) {
            wnew[l] = wn;
            if (wout) wout[l] = wn;
          }
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            acc[j].x = fmaf(wn, r[i][j].x, acc[j].x);
            acc[j].y = fmaf(wn, r[i][j].y, acc[j].y);
            acc[j].z = fmaf(wn, r[i][j].z, acc[j].z);
            acc[j].w = fmaf(wn, r[i][j].w, acc[j].w);
          }
        }
      }
    }
  }
  if (lanes_hold_w && !g.ctx_only && lane < nrows) {  // (all of the wave's rows: the ones past row_lim get their zero)
    wnew[l0 + lane] = wn_lane;
    if (wout) wout[l0 + lane] = wn_lane;
  }
  // cross-wave sum of the context partials in a fixed order (deterministic)
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    *reinterpret_cast<float4*>(part + ((wv * NJ + j) * 64 + lane) * 4) = acc[j];
  lds_barrier();  // (not __syncthreads(): the w stores above stay in flight)
  for (int d = threadIdx.x; d < D; d += kAttnThreads) {
    const int c4 = d >> 2, comp = d & 3;
    const int j = c4 >> 6, ln = c4 & 63;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += part[((w * NJ + j) * 64 + ln) * 4 + comp];
    const size_t o = g.out_mpad > 0 ? chunk_idx(b, d, g.out_mpad) : (size_t)b * D + d;
    if (g.dep_signal) {  // handed to the decoder LSTM of this very launch: write-through (see role_signal)
      store_wt(g.ctx + (size_t)b * D + d, s);
      if (g.ctx_h != nullptr) {
        f16 hi, lo;
        split_f16_checked(s, hi, lo, g.ctrl);
        store_wt(g.ctx_h + o, hi);
        store_wt(g.ctx_l + o, lo);
      }
    } else {
      g.ctx[(size_t)b * D + d] = s;
      if (g.ctx_h != nullptr) split_f16_checked(s, g.ctx_h[o], g.ctx_l[o], g.ctrl);
    }
  }
  if (g.dep_signal && g.ctrl != nullptr && !(g.ctrl->debug_flags & 2)) {
    role_signal(g.dep_cnt + (b / 32) * kDepLine);  // the decoder LSTM workgroups of these rows wait for ctx
    if (threadIdx.x == 0) stamp(st, 1, 5, now_rt());
  }
}

template <int NJ>
constexpr int attn_lds_floats() { return kAttnThreads / 64 * NJ * 256; }

}  // namespace ttsdec
