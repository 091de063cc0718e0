// Frame kernel body: the row-local chain between two decoder LSTM launches (see frame_kernel.hip for the
// description).  A device function so that it can also run as one role of a two-role launch (fused_kernels.hip).
#pragma once
#include "kernels.h"

namespace ttsdec {

constexpr int kFrameThreads = 512;
constexpr int kFrameRows = 32;
constexpr int kFrameCols = 64;

typedef __attribute__((address_space(1))) const f32x4 gf32x4;
typedef __attribute__((address_space(1))) const f16x8 gf16x8;

constexpr int kFrameMaxNI = 12;  // projection columns per thread in phase F: r*d_mel + r <= 16 * kFrameMaxNI

// LDS layout of one frame workgroup (floats): A operands (fp32 rows padded by 4 floats, or two fp16 planes
// padded by 8 halfs), the layer-1 reduction tile, the Philox keep bits
template <int K0H, int PH, int PREC>
struct FrameLds {
  static constexpr bool F16 = PREC == PREC_F16S;
  static constexpr int K0 = 2 * K0H;
  static constexpr int RS = 33;
  static constexpr int XS = F16 ? (K0 + 8) / 2 : K0 + 4;  // row strides in floats
  static constexpr int HS = F16 ? (PH + 8) / 2 : PH + 4;
  static constexpr int NPL = F16 ? 2 : 1;
  static constexpr int G0 = (PH + 127) / 128;  // Philox groups (128 keep bits each) of a hidden row
  static constexpr int kXs = 0;
  static constexpr int kH0 = kXs + NPL * kFrameRows * XS;
  static constexpr int kRed = kH0 + NPL * kFrameRows * HS;
  static constexpr int kPm0 = kRed + 8 * 32 * RS;
  static constexpr int kPm1 = kPm0 + kFrameRows * G0 * 4;
  static constexpr int kFloats = (kPm1 + kFrameRows * 4 + 3) & ~3;
};

// K0H = d_mel / 2, PH = hidden width of the PreNet, PREC = PREC_F32 (exact fp32 MFMA) or PREC_F16S
// (split-fp16 planes, 3 f16 MFMA products per k16 step, fp32 accumulate: see gemm_tile.h).
// NI: projection columns per thread in phase F (r*d_mel + r <= 16 * NI); the two-role launch uses 6 to stay within 128 VGPRs.
// lds: FrameLds<...>::kFloats floats (16-byte aligned) of the calling kernel's ONE shared array;
// (bx, by): column block / row block of this workgroup.
// kLean: for launches whose register budget is 128 (two workgroups per CU): the layer-1 weight fragments are requested
// after layer 0's MFMAs instead of at kernel entry - 32 registers fewer across phase F and layer 0.
// kHead: the partial sums come from a projection ROLE at the head of this very launch (FrameArgs::wait_n workgroups): they are
// requested once that role's arrival counter is complete, after the control block has been read, instead of first thing.
template <int K0H, int PH, int PREC, int NI = kFrameMaxNI, bool kLean = false, bool kHead = false>
__device__ __forceinline__ void frame_body(FrameArgs g, float* lds, int bx, int by) {
  using LD = FrameLds<K0H, PH, PREC>;
  constexpr bool F16 = PREC == PREC_F16S;
  constexpr int K0 = 2 * K0H;
  constexpr int KQ = PH / 4;   // layer 1: K range of one wave
  constexpr int K1H = KQ / 2;  //          and of one lane half
  constexpr int RS = LD::RS, XS = LD::XS, HS = LD::HS, G0 = LD::G0;
  static_assert(K0H % 8 == 0 && K1H % 8 == 0 && PH % 32 == 0 && PH <= 256, "unsupported PreNet shape");
  float* const xs = lds + LD::kXs;
  float* const h0s = lds + LD::kH0;
  float* const red = lds + LD::kRed;
  uint32_t* const pm0 = reinterpret_cast<uint32_t*>(lds + LD::kPm0);  // layer-0 keep bits [row][unit >> 5]
  uint32_t* const pm1 = reinterpret_cast<uint32_t*>(lds + LD::kPm1);  // layer-1 keep bits of this workgroup's 64 columns' group
  f16* const xs_h = reinterpret_cast<f16*>(xs);
  f16* const xs_l = xs_h + kFrameRows * XS * 2;
  f16* const h0_h = reinterpret_cast<f16*>(h0s);
  f16* const h0_l = h0_h + kFrameRows * HS * 2;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
  const int m0 = by * kFrameRows, n0 = bx * kFrameCols;
  const bool writer = bx == 0;
  const unsigned long long t_entry = now_rt();  // (measurement only: stamped below)

  // ---- weight fragments (independent of the control block) ----
  // lane (n = l32, half) holds a contiguous K run of weight row n: [half*K0H, +K0H) for layer 0,
  // [wave's quarter + half*K1H, +K1H) for layer 1
  const bool p0_wave = wave * 32 < PH;
  constexpr int NW0 = F16 ? K0H / 8 : K0H / 4, NW1 = F16 ? K1H / 8 : K1H / 4;
  f32x4 w0[F16 ? 1 : NW0], w1[F16 ? 1 : NW1];
  f16x8 w0h[F16 ? NW0 : 1], w0l[F16 ? NW0 : 1], w1h[F16 ? NW1 : 1], w1l[F16 ? NW1 : 1];
  float b0v = 0.f;
  const int n1 = n0 + (wave & 1) * 32 + l32;
  if (!g.only_finalize) {
    if (p0_wave) {
      const size_t o = (size_t)(wave * 32 + l32) * K0 + half * K0H;
      if constexpr (F16) {
        // (planes in lane order, decode_kernels.hip split_frame_order_kernel: piece (wave, j) = 64 x 16 bytes, one coalesced KiB)
        gf16x8 *sh = (gf16x8*)g.W0h + (size_t)wave * NW0 * 64 + lane, *sl = (gf16x8*)g.W0l + (size_t)wave * NW0 * 64 + lane;
#pragma unroll
        for (int j = 0; j < NW0; ++j) { w0h[j] = sh[j * 64]; w0l[j] = sl[j * 64]; }
      } else {
        gf32x4* src = (gf32x4*)(g.W0 + o);
#pragma unroll
        for (int j = 0; j < NW0; ++j) w0[j] = src[j];
      }
      b0v = g.b0[wave * 32 + l32];
    }
  }
  auto load_w1 = [&]() {
    const size_t o1 = (size_t)(n1 < g.P ? n1 : 0) * PH + (wave >> 1) * KQ + half * K1H;
    if constexpr (F16) {
      const size_t op = ((size_t)(bx * 8 + wave) * NW1) * 64 + lane;  // (lane order, as layer 0; columns past P are zero there)
      gf16x8 *sh = (gf16x8*)g.W1h + op, *sl = (gf16x8*)g.W1l + op;
#pragma unroll
      for (int j = 0; j < NW1; ++j) { w1h[j] = sh[j * 64]; w1l[j] = sl[j * 64]; }
    } else {
      gf32x4* src = (gf32x4*)(g.W1 + o1);
#pragma unroll
      for (int j = 0; j < NW1; ++j) w1[j] = src[j];
    }
  };
  if constexpr (!kLean) {
    if (!g.only_finalize) load_w1();
  }

  // ---- projection partial sums of the previous step (addresses independent of the control
  // block too; harmless when it then turns out there is nothing to finish) ----
  // thread -> (row = tid / 16, columns (tid % 16) + 16 i)
  const int nm = g.r * g.d_mel, NJ = nm + g.r;
  const int frow = tid >> 4, fm = m0 + frow;
  // (every load below is unconditional - out-of-range lanes read element 0 and slabs past n_parts are
  // discarded after the fact - so the compiler issues them all back to back instead of one dependent
  // load per branch)
  constexpr int kMaxParts = 4;
  float pv[NI];
  auto load_parts = [&]() {
    if (g.parts != nullptr) {
      float pz[NI][kMaxParts], pbias[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int n = (tid & 15) + 16 * i;
        const bool ok = fm < g.M && n < NJ;
        const size_t idx = ok ? (size_t)fm * g.ldp + n : 0;
#pragma unroll
        for (int z = 0; z < kMaxParts; ++z) {
          // (head role's slabs: written by other workgroups of this very launch - sc1 loads, see the wait below)
          if constexpr (kHead) pz[i][z] = load_wt(g.parts + z * g.part_stride + idx);
          else pz[i][z] = g.parts[z * g.part_stride + idx];
        }
        pbias[i] = g.proj_bias[ok ? n : 0];
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        float v = pz[i][0];
#pragma unroll
        for (int z = 1; z < kMaxParts; ++z) v = add_rn(v, z < g.n_parts ? pz[i][z] : 0.f);
        pv[i] = add_rn(v, pbias[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NI; ++i) pv[i] = 0.f;
    }
  };
  if constexpr (!kHead) load_parts();

  // ---- "now" ----
  int t = g.t, t_rel = g.t_rel, finalize = g.finalize;
  if (g.ctrl != nullptr) {
    const Ctrl* c = g.ctrl;
    bool live;
    if (g.only_finalize) {  // after the last step of the call: the frame of step t_end - 1
      t = c->t_end;
      live = c->t_end > c->t_call && c->t_end - 1 <= c->stop_t;
    } else {
      // This launch finishes step t-1's frame, and its writer workgroups may lower stop_t to t-1
      // while others are still reading it.  Gate on t-1 <= stop_t: that atomicMin cannot change it,
      // so every wave of every workgroup takes the same decision and all rows of the firing step
      // get written.  (When the rule fires here the PreNet below runs on a dead step: harmless.)
      t = c->t_cur + g.slot;
      live = t < c->t_end && t - 1 <= c->stop_t;
    }
    if (!live) return;
    t_rel = t - c->t_call;
    finalize = t > c->t_call;  // (the frame a call starts from was finished by the call before it)
    g.t_stride = c->t_stride;
    g.dropout_mode = c->dropout_mode;
    g.seed = c->seed;
    g.masks = c->masks ? c->masks + (size_t)t_rel * g.mask_step_stride : nullptr;
    g.teacher = c->teacher;
    g.teacher_T = c->teacher_T;
    g.teacher_flags = c->teacher_flags;
    g.y_out = c->y;
    g.s_out = c->s;
    g.stop_thr = c->stop_thr;
    g.check_stop = c->check_stop;
  }

  if constexpr (kHead) {
    if (finalize && g.wait_n > 0 && g.ctrl != nullptr) {
      // step t-1's projection is complete when (t - t_call) steps x wait_n workgroups have signalled.  No acquire: the slabs
      // were stored write-through and drained before each signal (proj_body), and every load of them below is an sc1 load
      // issued behind this poll (wave 0) or behind the barrier that wave 0 then joins (common.h load_wt)
      if (wave == 0) role_poll(g.wait_cnt + by * kDepLine, (unsigned int)t_rel * (unsigned int)g.wait_n, g.ctrl);
      lds_barrier();
    }
    load_parts();
  }
  const stamp_ptr st = g.dep_signal ? stamps_of(g.ctrl) : (stamp_ptr) nullptr;  // measurement only (TTSDEC_STAMPS)
  if (tid == 0) {  // role 0 of launch kind 0
    stamp(st, 0, 0, __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)));
    stamp(st, 0, 1, __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)));
    stamp(st, 0, 2, now_rt());
    stamp(st, 0, 7, t_entry);
  }
  // ---- epilogue operands of both PreNet layers, requested early ----
  // split-fp16 consumers read x_pre's planes, never its fp32 form: 4 columns per thread, one 8-byte store per plane - when the
  // tile has rows enough to keep the threads busy (one valid row: 16 threads x 4 serial elements, 12.7 against 12.1 us at B = 1)
  const bool planes_only = g.xpre_h != nullptr && !(g.P & 3) && g.M - m0 >= 16;
  uint8_t mk0[16], mk1[4];
  float b1v[4];
#pragma unroll
  for (int r = 0; r < 16; ++r) mk0[r] = 1;
#pragma unroll
  for (int j = 0; j < 4; ++j) { mk1[j] = 1; b1v[j] = 0.f; }
  if (!g.only_finalize) {
    if (g.dropout_mode == TTSDEC_DROPOUT_MASKS && p0_wave) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < g.M) mk0[r] = as_g(g.masks)[(size_t)m * PH + wave * 32 + l32];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // layer-1 outputs of this thread: 4 consecutive columns of one row when only the planes are written (see the
      // store at the end), else elements tid + 512 j of the tile
      const int e = planes_only ? tid * 4 + j : tid + j * kFrameThreads;
      const int m = m0 + e / kFrameCols, n = n0 + e % kFrameCols;
      if (m < g.M && n < g.P) {
        b1v[j] = g.b1[n];
        if (g.dropout_mode == TTSDEC_DROPOUT_MASKS) mk1[j] = as_g(g.masks)[(size_t)g.M * PH + (size_t)m * g.P + n];
      }
    }
  }

  if (!g.only_finalize && g.dropout_mode == TTSDEC_DROPOUT_PHILOX && tid < kFrameRows * (G0 + 1)) {
    // one Philox call per (row, 128 units): 32 x (G0 + 1) calls per workgroup instead of one per unit
    const int row = tid % kFrameRows, grp = tid / kFrameRows;
    const bool l1 = grp == G0;
    const Philox4 k = philox_keep_group(g.seed, (uint32_t)t, l1 ? 1u : 0u, (uint32_t)(m0 + row), l1 ? (uint32_t)(n0 >> 7) : (uint32_t)grp);
    uint32_t* dst = l1 ? pm1 + row * 4 : pm0 + (row * G0 + grp) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = k.w[i];
  }

  // Retire every load issued so far HERE, before phase F issues its y / s stores.  vmcnt counts loads and stores in ONE
  // in-order queue: the first use of an early load after those stores (b0v in layer 0's epilogue was the one) makes the
  // compiler wait for the stores too - a write round trip of 2.5-3.9 us in the middle of the role's critical path (time
  // stamps, and `s_waitcnt vmcnt(8)` in front of that add).  The loads themselves have long arrived by now.
  {
    asm volatile("" ::"v"(b0v), "v"(b1v[0]), "v"(b1v[1]), "v"(b1v[2]), "v"(b1v[3]), "v"(pv[NI - 1]));
    if constexpr (F16) asm volatile("" ::"v"(w0h[NW0 - 1]), "v"(w0l[NW0 - 1]));
    else asm volatile("" ::"v"(w0[NW0 - 1]));
    if constexpr (!kLean) {
      if constexpr (F16) asm volatile("" ::"v"(w1h[NW1 - 1]), "v"(w1l[NW1 - 1]));
      else asm volatile("" ::"v"(w1[NW1 - 1]));
    }
  }
  // injected keep-masks of layer 0: the 16 bytes become one 16-bit mask here (15 registers fewer across layer 0)
  unsigned keep_bits = 0xffffu;
  if (g.dropout_mode == TTSDEC_DROPOUT_MASKS) {
    keep_bits = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) keep_bits |= (mk0[r] ? 1u : 0u) << r;
  }
  {
    const unsigned keep1 = mk1[0] | mk1[1] | mk1[2] | mk1[3];
    asm volatile("" ::"v"(keep1), "v"(keep_bits));
  }
  bool over = false;  // a value left the fp16 range (split-fp16 planes saturate): reported once, report_range below
  // store one element of the input frame as the layer-0 A operand
  auto put_x = [&](int row, int c, float v) {
    if constexpr (F16) split_f16_flag(v, xs_h[row * (XS * 2) + c], xs_l[row * (XS * 2) + c], over);
    else xs[row * XS + c] = v;
  };

  // ---- phase F: the input frame of step t ----
  const bool teach = g.teacher != nullptr && t > 0 && as_g(g.teacher_flags)[t - 1] != 0;
  if (finalize) {
    // (row pointers and the row / writer predicates once, the stop rule as one flag per thread: written with the tests and
    // the 64-bit index arithmetic inside the loop this phase was ~110 instructions per column)
    const bool rowok = fm < g.M, store = rowok && writer;
    const size_t fr = (size_t)fm * g.t_stride * g.r + (size_t)(t_rel - 1) * g.r;
    const int n_last = nm - g.d_mel;                       // first channel of the group's last frame (decoder.py:48 y_t[:, -1, :])
    const auto yrow = as_g(g.y_out) + fr * g.d_mel;        // [B, t_stride*r, d_mel]: frame jf, channel c = yrow[jf*d_mel + c]
    const auto ynrow = as_g(g.ynext) + (size_t)fm * g.d_mel - n_last;
    const auto srow = as_g(g.s_out) + fr - nm;
    bool fire = false;
    if (g.r == 1) {
      // one frame per step (every shipped config but one): columns [0, K0) are the frame, column K0 the stop logit - no
      // per-column tests at all
      static_assert(K0 % 16 == 0 && K0 / 16 + 1 <= NI, "phase F: columns per thread");
#pragma unroll
      for (int i = 0; i < K0 / 16; ++i) {
        const int n = (tid & 15) + 16 * i;
        const float lv = pv[i] > 0.f ? pv[i] : mul_rn(pv[i], 0.01f);  // decoder.py:53-54
        put_x(frow, n, rowok ? lv : 0.f);
        if (store) {
          yrow[n] = lv;
          ynrow[n] = lv;
        }
      }
      if (store && (tid & 15) == 0) {
        srow[K0] = pv[K0 / 16];  // decoder.py:52
        fire = pv[K0 / 16] < g.stop_thr;
      }
    } else
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int n = (tid & 15) + 16 * i;
      const float v = pv[i];
      const bool mel = n < nm, last = mel && n >= n_last;
      const float lv = v > 0.f ? v : mul_rn(v, 0.01f);  // decoder.py:53-54
      if (last) put_x(frow, n - n_last, rowok ? lv : 0.f);
      if (store) {
        if (mel) {
          yrow[n] = lv;
          if (last) ynrow[n] = lv;
        } else if (n < NJ) {
          srow[n] = v;  // decoder.py:52
          fire |= v < g.stop_thr;
        }
      }
    }
    if (fire && g.check_stop && g.ctrl != nullptr) atomicMin(&g.ctrl->stop_t, t - 1);  // decoder.py:68
  }
  if (g.only_finalize) {
    report_range(over, g.ctrl);
    return;
  }
  if (tid == 0) stamp(st, 0, 3, now_rt());  // the previous step's frame is finished
  if (!finalize || teach) {
    if (teach) __syncthreads();  // the teacher frame replaces what phase F put there
    for (int c = tid & 15; c < K0; c += 16) {
      float v = 0.f;
      if (fm < g.M)
        v = teach ? as_g(g.teacher)[((size_t)fm * g.teacher_T + (size_t)t * g.r - 1) * g.d_mel + c]  // decoder.py:65-66
                  : as_g(g.ynext)[(size_t)fm * g.d_mel + c];
      put_x(frow, c, v);
    }
  }
  lds_barrier();

  // ---- PreNet layer 0: h0 = dropout(relu(x W0^T + b0)), one 32-column tile per wave ----
  if (p0_wave) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!(g.dbg & 2)) {
      if constexpr (F16) {
        f32x16 acc2 = acc;
        const f16* ah = xs_h + l32 * (XS * 2) + half * K0H;
        const f16* al = xs_l + l32 * (XS * 2) + half * K0H;
#pragma unroll
        for (int j = 0; j < NW0; ++j) {
          const f16x8 a_h = *reinterpret_cast<const f16x8*>(ah + 8 * j), a_l = *reinterpret_cast<const f16x8*>(al + 8 * j);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, w0h[j], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, w0l[j], acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l, w0h[j], acc2, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fmaf(acc2[i], 1.0f / kSplitScale, acc[i]);
      } else {
        const float* arow = xs + l32 * XS + half * K0H;
#pragma unroll
        for (int j = 0; j < NW0; ++j) {
          const f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + 4 * j);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], w0[j][e], acc, 0, 0, 0);
        }
      }
    }
    if constexpr (kLean) load_w1();  // (layer 0's weight registers are free now; the epilogue and a barrier hide the latency)
    const int col = wave * 32 + l32;
    // Straight-line epilogue: the keep decisions of the 16 rows are one bit mask (injected masks: packed above; Philox: the
    // words read from LDS here in one batch), the range check is a running maximum (ReLU outputs are
    // >= 0).  Written naively - mode branches, a conditional atomic and an LDS read per row - this loop was ~70
    // instructions per row and 2-3 us of the role's critical path (time stamps).
    if (g.dropout_mode == TTSDEC_DROPOUT_PHILOX) {
      uint32_t kb[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) kb[r] = pm0[((r & 3) + 8 * (r >> 2) + 4 * half) * G0 * 4 + wave];
      keep_bits = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) keep_bits |= ((kb[r] >> l32) & 1u) << r;  // unit = 32*wave + l32
    }
    const float ks = g.dropout_mode == TTSDEC_DROPOUT_OFF ? 1.0f : g.keep_scale;  // (x * 1.0f is exact)
    float vmax = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      float v = add_rn(acc[r], b0v);
      v = v > 0.f ? v : 0.f;
      v = ((keep_bits >> r) & 1u) ? mul_rn(v, ks) : 0.f;
      if constexpr (F16) {
        vmax = fmaxf(vmax, v);
        split_f16_pos(v, h0_h[row * (HS * 2) + col], h0_l[row * (HS * 2) + col]);
      } else {
        h0s[row * HS + col] = v;
      }
    }
    over |= vmax > kSplitMax;
  } else {
    if constexpr (kLean) load_w1();  // (a wave without a layer-0 tile, PreNet hidden width 128, needs its layer-1 fragments too)
  }
  lds_barrier();

  if (tid == 0) stamp(st, 0, 4, now_rt());  // layer 0 done
  // ---- PreNet layer 1: 2 column tiles x 4 K quarters over the 8 waves ----
  {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!(g.dbg & 4)) {
      if constexpr (F16) {
        f32x16 acc2 = acc;
        const f16* ah = h0_h + l32 * (HS * 2) + (wave >> 1) * KQ + half * K1H;
        const f16* al = h0_l + l32 * (HS * 2) + (wave >> 1) * KQ + half * K1H;
#pragma unroll
        for (int j = 0; j < NW1; ++j) {
          const f16x8 a_h = *reinterpret_cast<const f16x8*>(ah + 8 * j), a_l = *reinterpret_cast<const f16x8*>(al + 8 * j);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, w1h[j], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, w1l[j], acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l, w1h[j], acc2, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fmaf(acc2[i], 1.0f / kSplitScale, acc[i]);
      } else {
        const float* arow = h0s + l32 * HS + (wave >> 1) * KQ + half * K1H;
#pragma unroll
        for (int j = 0; j < NW1; ++j) {
          const f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + 4 * j);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], w1[j][e], acc, 0, 0, 0);
        }
      }
    }
    float* out = red + wave * 32 * RS;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * half) * RS + l32] = acc[r];
  }
  lds_barrier();
  // element (row, col) of the layer-1 tile: sum of the K quarters, bias, relu, dropout
  // (dropout mode resolved once, no branch per element: see layer 0's epilogue)
  const int dmode1 = g.dropout_mode;
  const float ks1 = dmode1 == TTSDEC_DROPOUT_OFF ? 1.0f : g.keep_scale;
  auto finish = [&](int row, int col, float bias, uint8_t keep_byte) {
    const int n = n0 + col;
    const float* pr = red + (col >> 5) * 32 * RS + row * RS + (col & 31);
    const float q0 = pr[0], q1 = pr[2 * 32 * RS], q2 = pr[4 * 32 * RS], q3 = pr[6 * 32 * RS];
    const uint32_t kw = pm1[row * 4 + ((n >> 5) & 3)];
    float v = add_rn(add_rn(add_rn(q0, q1), q2), q3);  // K quarters in order
    v = add_rn(v, bias);
    v = v > 0.f ? v : 0.f;
    const bool keep = dmode1 == TTSDEC_DROPOUT_PHILOX ? ((kw >> (n & 31)) & 1u) != 0 : keep_byte != 0;  // (keep_byte is 1 unless injected)
    return keep ? mul_rn(v, ks1) : 0.f;
  };
  if (planes_only) {
    // 4 consecutive columns per thread, ONE 8-byte store per plane - write-through when the attention LSTM of this very
    // launch waits for them (2-byte write-through stores cost ~12x the fabric time per byte, common.h store_wt8)
    const int row = tid >> 4, col = (tid & 15) * 4;
    const int m = m0 + row, n = n0 + col;
    if (m < g.M && n < g.P) {
      union { f16 h[4]; unsigned long long u; } hi, lo;
      float vmax = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float v = finish(row, col + c, b1v[c], mk1[c]);
        vmax = fmaxf(vmax, v);
        split_f16_pos(v, hi.h[c], lo.h[c]);
      }
      over |= vmax > kSplitMax;
      const size_t oc = g.out_mpad > 0 ? chunk_idx(m, n, g.out_mpad) : (size_t)m * g.P + n;
      if (g.dep_signal) {
        store_wt8(g.xpre_h + oc, hi.u);
        store_wt8(g.xpre_l + oc, lo.u);
      } else {
        *reinterpret_cast<unsigned long long*>(g.xpre_h + oc) = hi.u;
        *reinterpret_cast<unsigned long long*>(g.xpre_l + oc) = lo.u;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = tid + j * kFrameThreads;
      const int row = e / kFrameCols, col = e % kFrameCols;
      const int m = m0 + row, n = n0 + col;
      if (m >= g.M || n >= g.P) continue;
      const float v = finish(row, col, b1v[j], mk1[j]);
      const size_t o = (size_t)m * g.P + n;
      const size_t oc = g.out_mpad > 0 ? chunk_idx(m, n, g.out_mpad) : o;
      if (g.dep_signal) {  // handed to the attention LSTM of this very launch: write-through (see role_signal)
        store_wt(g.xpre + o, v);
        if (g.xpre_h != nullptr) {
          f16 hi, lo;
          split_f16_flag(v, hi, lo, over);
          store_wt(g.xpre_h + oc, hi);
          store_wt(g.xpre_l + oc, lo);
        }
      } else {
        g.xpre[o] = v;
        if (g.xpre_h != nullptr) split_f16_flag(v, g.xpre_h[oc], g.xpre_l[oc], over);
      }
    }
  }
  report_range(over, g.ctrl);
  if (tid == 0) stamp(st, 0, 6, now_rt());  // x_pre stores issued
  if (g.dep_signal && g.ctrl != nullptr && !(g.ctrl->debug_flags & 1)) {
    role_signal(g.dep_cnt + by * kDepLine);  // the attention LSTM workgroups of these rows wait for x_pre
    if (tid == 0) stamp(st, 0, 5, now_rt());
  }
}


// ===========================================================================
// mel/stop projection [fc_mel; fc_stop] (decoder.py:52-53) as split-K partial sums for the frame kernel (kernels.h ProjArgs)
// ===========================================================================
constexpr int kProjTile = 32;  // rows and columns of a workgroup's output tile
constexpr int kProjNS = 4;     // k16 steps per wave at most

// address of element (row, k) of a segmented activation operand with EB-byte elements
// (The segment index differs from lane to lane here.  Selecting between the struct's MEMBERS with it lets the compiler turn
// "select of loads" into one indexed load from the kernel-argument struct - which it then copies to scratch: 216 bytes per lane
// and ten scratch loads per fragment in the stand-alone projection kernel, 6.5 -> 14 us per launch before this was noticed.
// The members are therefore first pinned in scalar registers and the select made between those.)
template <class T>
__device__ __forceinline__ T pinned_scalar(T v) {
  asm volatile("" : "+s"(v));
  return v;
}
// kPin = false: the plain member selects (the multi-role kernels, where the pattern does not arise and the extra scalar
// registers would spill).
template <int EB, bool kPin>
__device__ __forceinline__ gbyte* seg_elem_ptr(const Seg3& s, int row, int k) {
  if constexpr (!kPin) {
    const int i = k < s.e0 ? 0 : (k < s.e1 ? 1 : 2);
    const int kk = k - (i == 0 ? 0 : (i == 1 ? s.e0 : s.e1));
    gbyte* p = seg_row_ptr<EB>(s, row, i);
    return p + (s.mpad > 0 ? (long)(kk >> 5) * s.mpad * kChunkBytes + (long)(kk & 31) * EB : (long)kk * EB);
  }
  const int e0 = pinned_scalar(s.e0), e1 = pinned_scalar(s.e1), mpad = s.mpad;
  const unsigned long long p0 = pinned_scalar((unsigned long long)s.p0), p1 = pinned_scalar((unsigned long long)s.p1),
                           p2 = pinned_scalar((unsigned long long)s.p2);
  const int ld0 = pinned_scalar(s.ld0), ld1 = pinned_scalar(s.ld1), ld2 = pinned_scalar(s.ld2);
  const int i = k < e0 ? 0 : (k < e1 ? 1 : 2);
  const int kk = k - (i == 0 ? 0 : (i == 1 ? e0 : e1));
  const unsigned long long base = i == 0 ? p0 : (i == 1 ? p1 : p2);
  const int ld = i == 0 ? ld0 : (i == 1 ? ld1 : ld2);
  gbyte* p = (gbyte*)base + (mpad > 0 ? (long)row * kChunkBytes : (long)row * ld * EB);
  return p + (mpad > 0 ? (long)(kk >> 5) * mpad * kChunkBytes + (long)(kk & 31) * EB : (long)kk * EB);
}

constexpr int kProjLdsFloats = 8 * 32 * 33;
// red: kProjLdsFloats floats of LDS (the caller's ONE shared array); id: index of the workgroup within the kernel or role
// kPin: seg_elem_ptr above (true in the stand-alone projection kernel)
template <int PREC, bool kPin = false>
__device__ __forceinline__ void proj_body(ProjArgs g, float* red, int id) {
  constexpr bool F16 = PREC == PREC_F16S;
  constexpr int EB = F16 ? 2 : 4, RS = 33;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
  const int nx = (g.N + kProjTile - 1) / kProjTile, ny = (g.M + kProjTile - 1) / kProjTile;
  const int n0 = (id % nx) * kProjTile, m0 = ((id / nx) % ny) * kProjTile, z = id / (nx * ny);
  const int spw = g.K / (128 * g.ksplit);  // k16 steps per wave, <= kProjNS
  const unsigned long long t_entry = now_rt();  // (measurement only: TTSDEC_STAMPS)
  // this lane's K run: slice z, the wave's share of it, the lane half's half of that - 8 * spw consecutive k
  const int kbeg = ((z * 8 + wave) * 2 + half) * spw * 8;

  // ---- weights: independent of everything, requested first ----
  f16x8 wh[F16 ? kProjNS : 1], wl[F16 ? kProjNS : 1];
  f32x4 wf[F16 ? 1 : 2 * kProjNS];
  {
    const int n = n0 + l32 < g.N ? n0 + l32 : 0;
    const size_t o = ((size_t)n * g.ldw + kbeg) * EB;
    if constexpr (F16) {
      gf16x8 *sh = (gf16x8*)(as_global(g.W) + o), *sl = (gf16x8*)(as_global(g.W_lo) + o);
#pragma unroll
      for (int j = 0; j < kProjNS; ++j)
        if (j < spw) { wh[j] = sh[j]; wl[j] = sl[j]; }
    } else {
      gf32x4* sf = (gf32x4*)(as_global(g.W) + o);
#pragma unroll
      for (int j = 0; j < 2 * kProjNS; ++j)
        if (j < 2 * spw) wf[j] = sf[j];
    }
  }
  // ---- activations: row m0 + l32, the same K run (A and W only have to agree on which k a lane element means).  Requested
  // BEFORE the control block is looked at, like the weights: they are the previous launch's outputs, and a wait for the
  // control block is a wait for every load issued before it (one in-order queue) - with the activations behind that wait the
  // role paid two memory round trips in series, each several us beside the LSTM role's tile stream ----
  f16x8 ah[F16 ? kProjNS : 1], al[F16 ? kProjNS : 1];
  f32x4 af[F16 ? 1 : 2 * kProjNS];
  if (g.wait_cnt != nullptr && g.ctrl != nullptr) {
    // one-launch step: the operand comes from workgroups of this very launch (PROJ_QUERY: h_att from the attention LSTM's tiles
    // of this tile's 32-row block) - one wave polls, the others take their runs behind the barrier it then joins, all with sc1
    // loads (common.h load_wt8; one poller per workgroup: gemm_tile.h on what 1024 pollers of one counter cost)
    const Ctrl* c = g.ctrl;
    const int t = c->t_cur + g.slot;
    if (!(t < c->t_end && t - 1 <= c->stop_t)) return;  // (live_lag: see below; the same for every wave)
    if (wave == 0) role_poll(g.wait_cnt + (m0 / kProjTile) * kDepLine, (unsigned int)(t - c->t_call + 1) * (unsigned int)g.wait_n, g.ctrl, nullptr, 0, g.wait_sleep);
    __builtin_amdgcn_s_barrier();
    const int m = m0 + l32 < g.M ? m0 + l32 : g.M - 1;
    // (16-byte sc1 loads: buffer loads with the sc1 policy bit - the atomic-load builtins stop at 8 bytes, and every load
    // instruction of this role queues behind the LSTM roles' tile stream; the operand is ONE segment - the host's business)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    gbyte *base_h = as_global(g.a.p0), *base_l = as_global(g.a_lo.p0);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void*)g.a.p0, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)g.a_lo.p0, 0, 0x7fffffff, 0x00020000);
    if constexpr (F16) {
#pragma unroll
      for (int j = 0; j < kProjNS; ++j)
        if (j < spw) {
          const int oh = (int)(seg_elem_ptr<2, kPin>(g.a, m, kbeg + 8 * j) - base_h), ol = (int)(seg_elem_ptr<2, kPin>(g.a_lo, m, kbeg + 8 * j) - base_l);
          ah[j] = __builtin_bit_cast(f16x8, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rh, oh, 0, 16));
          al[j] = __builtin_bit_cast(f16x8, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rl, ol, 0, 16));
        }
    } else {
#pragma unroll
      for (int j = 0; j < 2 * kProjNS; ++j)
        if (j < 2 * spw) {
          const int oa = (int)(seg_elem_ptr<4, kPin>(g.a, m, kbeg + 4 * j) - base_h);
          af[j] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rh, oa, 0, 16));
        }
    }
  } else {
    const int m = m0 + l32 < g.M ? m0 + l32 : g.M - 1;
    if constexpr (F16) {
#pragma unroll
      for (int j = 0; j < kProjNS; ++j)
        if (j < spw) {
          ah[j] = *(gf16x8*)seg_elem_ptr<2, kPin>(g.a, m, kbeg + 8 * j);
          al[j] = *(gf16x8*)seg_elem_ptr<2, kPin>(g.a_lo, m, kbeg + 8 * j);
        }
    } else {
#pragma unroll
      for (int j = 0; j < 2 * kProjNS; ++j)
        if (j < 2 * spw) af[j] = *(gf32x4*)seg_elem_ptr<4, kPin>(g.a, m, kbeg + 4 * j);
    }
  }
  bool signal = false;
  if (g.ctrl != nullptr) {
    const Ctrl* c = g.ctrl;
    const int t = c->t_cur + g.slot;
    bool live;
    // (PROJ_HEAD: the frame role of the same launch may lower stop_t to t-1 while this is read; t-1 <= stop_t cannot be
    // changed by that - see lstm_body's live_lag)
    if (g.mode == PROJ_HEAD) live = t < c->t_end && t - 1 <= c->stop_t && t > c->t_call;
    else if (g.mode == PROJ_FINAL) live = c->t_end > c->t_call && c->t_end - 1 <= c->stop_t;
    else live = t < c->t_end && t - (g.live_lag ? 1 : 0) <= c->stop_t;  // (live_lag: a role of the one-launch step, see lstm_body)
    if (!live) return;
    signal = g.mode == PROJ_HEAD || g.mode == PROJ_QUERY;
  }
  const stamp_ptr st = g.mode == PROJ_HEAD && signal ? stamps_of(g.ctrl) : (stamp_ptr) nullptr;
  const unsigned long long t_ctrl = now_rt();

  f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  {
    if constexpr (F16) {
      f32x16 acc2 = acc;
#pragma unroll
      for (int j = 0; j < kProjNS; ++j)
        if (j < spw) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], wh[j], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], wl[j], acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[j], wh[j], acc2, 0, 0, 0);
        }
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = fmaf(acc2[i], 1.0f / kSplitScale, acc[i]);
    } else {
#pragma unroll
      for (int j = 0; j < 2 * kProjNS; ++j)
        if (j < 2 * spw) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][e], wf[j][e], acc, 0, 0, 0);
        }
    }
  }

  // ---- the 8 waves' partial tiles, added in wave order (deterministic) ----
  {
    float* out = red + wave * 32 * RS;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * half) * RS + l32] = acc[r];
  }
  __syncthreads();
  float* slab = g.out + (size_t)z * g.split_stride;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int e = tid + j * kFrameThreads;
    const int row = e >> 5, col = e & 31;
    const int m = m0 + row, n = n0 + col;
    if (m >= g.M || n >= g.N) continue;
    const float* pr = red + row * RS + col;
    float v = pr[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) v = add_rn(v, pr[w * 32 * RS]);
    if (signal) store_wt(slab + (size_t)m * g.ldo + n, v);  // read by the frame role of this very launch
    else slab[(size_t)m * g.ldo + n] = v;
  }
  const unsigned long long t_red = now_rt();
  if (signal && !(g.ctrl->debug_flags & 4)) role_signal(g.dep_cnt + (m0 / kProjTile) * kDepLine);  // (32-row blocks: the frame role's)
  if (tid == 0 && st != nullptr) {
    stamp(st, 2, 0, __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)));
    stamp(st, 2, 1, __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)));
    stamp(st, 2, 2, t_entry);
    stamp(st, 2, 3, t_ctrl);
    stamp(st, 2, 4, t_red);
    stamp(st, 2, 5, now_rt());
  }
}


}  // namespace ttsdec
