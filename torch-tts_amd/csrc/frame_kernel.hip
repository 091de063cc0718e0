// Frame kernel: the row-local chain between two decoder LSTM launches, as ONE launch.
//
//   step t-1:  ... lstm_dec -> proj GEMM (split-K partial sums)            reference:
//   step t  :  [ sum partials + bias -> leaky_relu -> y, s, stop rule ]    decoder.py:52-54,68
//              [ next input frame (last frame of the group / teacher) ]    decoder.py:48,61-66
//              [ PreNet layer 0 -> PreNet layer 1 (relu + always-on dropout) ]  modules/modules.py:37-41
//              -> lstm_att ...
//
// A workgroup owns 32 batch rows x 64 PreNet output columns.  The frame (d_mel values per row) and
// the hidden PreNet layer are recomputed by every column block of a row block (80 x 256 MACs per
// row: cheaper than another launch); only column block 0 writes y / s / the stop flag.
// Weights never touch LDS: with fp32 32x32x2 MFMAs the K index may be permuted freely as long as
// A and B agree, so each lane takes a CONTIGUOUS run of K for its weight row (lane half 0 the
// first half of the K range, half 1 the second) and loads it with global_load_dwordx4 at kernel
// entry, before the control block has even arrived.
#include "kernels.h"

namespace ttsdec {

constexpr int kFrameThreads = 512;
constexpr int kFrameRows = 32;
constexpr int kFrameCols = 64;

typedef __attribute__((address_space(1))) const f32x4 gf32x4;

// K0H = d_mel / 2, PH = hidden width of the PreNet
template <int K0H, int PH>
__global__ __launch_bounds__(kFrameThreads) void frame_kernel(FrameArgs g) {
  constexpr int K0 = 2 * K0H;
  constexpr int XS = K0 + 4, HS = PH + 4;  // padded LDS row strides (floats)
  constexpr int KQ = PH / 4;               // layer 1: K range of one wave
  constexpr int K1H = KQ / 2;              //          and of one lane half
  constexpr int RS = 33;
  static_assert(K0H % 4 == 0 && K1H % 4 == 0 && PH % 32 == 0 && PH <= 256, "unsupported PreNet shape");
  __shared__ __attribute__((aligned(16))) float xs[kFrameRows * XS];
  __shared__ __attribute__((aligned(16))) float h0s[kFrameRows * HS];
  __shared__ __attribute__((aligned(16))) float red[8 * 32 * RS];
  constexpr int G0 = (PH + 127) / 128;               // Philox groups (128 keep bits each) of a hidden row
  __shared__ uint32_t pm0[kFrameRows * G0 * 4];      // layer-0 keep bits [row][unit >> 5]
  __shared__ uint32_t pm1[kFrameRows * 4];           // layer-1 keep bits of this workgroup's 64 columns' group

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
  const int m0 = blockIdx.y * kFrameRows, n0 = blockIdx.x * kFrameCols;
  const bool writer = blockIdx.x == 0;

  // ---- weight fragments (independent of the control block) ----
  const bool p0_wave = wave * 32 < PH;
  f32x4 w0[K0H / 4], w1[K1H / 4];
  float b0v = 0.f;
  if (!g.only_finalize) {
    if (p0_wave) {
      gf32x4* src = (gf32x4*)(g.W0 + (size_t)(wave * 32 + l32) * K0 + half * K0H);
#pragma unroll
      for (int j = 0; j < K0H / 4; ++j) w0[j] = src[j];
      b0v = g.b0[wave * 32 + l32];
    }
    const int n1 = n0 + (wave & 1) * 32 + l32;
    if (n1 < g.P) {
      gf32x4* src = (gf32x4*)(g.W1 + (size_t)n1 * PH + (wave >> 1) * KQ + half * K1H);
#pragma unroll
      for (int j = 0; j < K1H / 4; ++j) w1[j] = src[j];
    } else {
#pragma unroll
      for (int j = 0; j < K1H / 4; ++j) w1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }

  // ---- "now" ----
  int t = g.t, t_rel = g.t_rel, finalize = g.finalize;
  if (g.ctrl != nullptr) {
    const Ctrl* c = g.ctrl;
    bool live;
    if (g.only_finalize) {  // after the last step of the call: the frame of step t_end - 1
      t = c->t_end;
      live = c->t_end > c->t_call && c->t_end - 1 <= c->stop_t;
    } else {
      const StepNow now = step_now(c, g.slot);
      t = now.t;
      live = now.live;
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

  // ---- epilogue operands of both PreNet layers, requested early ----
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
        if (m < g.M) mk0[r] = g.masks[(size_t)m * PH + wave * 32 + l32];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = tid + j * kFrameThreads;
      const int m = m0 + e / kFrameCols, n = n0 + e % kFrameCols;
      if (m < g.M && n < g.P) {
        b1v[j] = g.b1[n];
        if (g.dropout_mode == TTSDEC_DROPOUT_MASKS) mk1[j] = g.masks[(size_t)g.M * PH + (size_t)m * g.P + n];
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

  // ---- phase F: the input frame of step t ----
  const int nm = g.r * g.d_mel, NJ = nm + g.r;
  const bool teach = g.teacher != nullptr && t > 0 && g.teacher_flags[t - 1] != 0;
  if (finalize) {
    for (int e = tid; e < kFrameRows * NJ; e += kFrameThreads) {
      const int row = e / NJ, n = e - row * NJ;
      const int m = m0 + row;
      if (m >= g.M) {
        if (n >= nm - g.d_mel && n < nm) xs[row * XS + n - (nm - g.d_mel)] = 0.f;
        continue;
      }
      float v = g.parts[(size_t)m * g.ldp + n];
      for (int z = 1; z < g.n_parts; ++z) v = add_rn(v, g.parts[z * g.part_stride + (size_t)m * g.ldp + n]);
      v = add_rn(v, g.proj_bias[n]);
      const size_t fr = (size_t)m * g.t_stride * g.r + (size_t)(t_rel - 1) * g.r;
      if (n < nm) {
        v = v > 0.f ? v : mul_rn(v, 0.01f);  // decoder.py:53-54
        const int jf = n / g.d_mel, c = n - jf * g.d_mel;
        if (writer) g.y_out[(fr + jf) * g.d_mel + c] = v;
        if (jf == g.r - 1) {  // decoder.py:48 y_t[:, -1, :]
          xs[row * XS + c] = v;
          if (writer) g.ynext[(size_t)m * g.d_mel + c] = v;
        }
      } else if (writer) {
        g.s_out[fr + (n - nm)] = v;  // decoder.py:52
        if (g.check_stop && g.ctrl != nullptr && v < g.stop_thr) atomicMin(&g.ctrl->stop_t, t - 1);  // decoder.py:68
      }
    }
  }
  if (g.only_finalize) return;
  if (!finalize || teach) {
    if (teach) __syncthreads();  // the teacher frame replaces what phase F put there
    for (int e = tid; e < kFrameRows * K0; e += kFrameThreads) {
      const int row = e / K0, c = e - row * K0;
      const int m = m0 + row;
      float v = 0.f;
      if (m < g.M)
        v = teach ? g.teacher[((size_t)m * g.teacher_T + (size_t)t * g.r - 1) * g.d_mel + c]  // decoder.py:65-66
                  : g.ynext[(size_t)m * g.d_mel + c];
      xs[row * XS + c] = v;
    }
  }
  __syncthreads();

  // ---- PreNet layer 0: h0 = dropout(relu(x W0^T + b0)), one 32-column tile per wave ----
  if (p0_wave) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* arow = xs + l32 * XS + half * K0H;
    if (!(g.dbg & 2))
#pragma unroll
    for (int j = 0; j < K0H / 4; ++j) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + 4 * j);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], w0[j][e], acc, 0, 0, 0);
    }
    const int col = wave * 32 + l32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      float v = add_rn(acc[r], b0v);
      v = v > 0.f ? v : 0.f;
      if (g.dropout_mode == TTSDEC_DROPOUT_MASKS) v = mk0[r] ? mul_rn(v, g.keep_scale) : 0.f;
      else if (g.dropout_mode == TTSDEC_DROPOUT_PHILOX)
        v = ((pm0[row * G0 * 4 + wave] >> l32) & 1u) ? mul_rn(v, g.keep_scale) : 0.f;  // unit = 32*wave + l32
      h0s[row * HS + col] = v;
    }
  }
  __syncthreads();

  // ---- PreNet layer 1: 2 column tiles x 4 K quarters over the 8 waves ----
  {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* arow = h0s + l32 * HS + (wave >> 1) * KQ + half * K1H;
    if (!(g.dbg & 4))
#pragma unroll
    for (int j = 0; j < K1H / 4; ++j) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + 4 * j);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], w1[j][e], acc, 0, 0, 0);
    }
    float* out = red + wave * 32 * RS;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * half) * RS + l32] = acc[r];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = tid + j * kFrameThreads;
    const int row = e / kFrameCols, col = e % kFrameCols;
    const int m = m0 + row, n = n0 + col;
    if (m >= g.M || n >= g.P) continue;
    const float* pr = red + (col >> 5) * 32 * RS + row * RS + (col & 31);
    float v = pr[0];
#pragma unroll
    for (int q = 1; q < 4; ++q) v = add_rn(v, pr[q * 2 * 32 * RS]);  // K quarters in order
    v = add_rn(v, b1v[j]);
    v = v > 0.f ? v : 0.f;
    if (g.dropout_mode == TTSDEC_DROPOUT_MASKS) v = mk1[j] ? mul_rn(v, g.keep_scale) : 0.f;
    else if (g.dropout_mode == TTSDEC_DROPOUT_PHILOX)
      v = ((pm1[row * 4 + ((n >> 5) & 3)] >> (n & 31)) & 1u) ? mul_rn(v, g.keep_scale) : 0.f;
    const size_t o = (size_t)m * g.P + n;
    g.xpre[o] = v;
    if (g.xpre_h != nullptr) split_f16(v, g.xpre_h[o], g.xpre_l[o]);
  }
}

bool frame_supported(int d_mel, int Ph, int P) { return d_mel == 80 && (Ph == 256 || Ph == 128) && P % 4 == 0; }

void launch_frame(const FrameArgs& a, hipStream_t st) {
  if (a.M <= 0) return;
  const int cols = a.only_finalize ? 1 : (a.P + kFrameCols - 1) / kFrameCols;
  dim3 grid(cols, (a.M + kFrameRows - 1) / kFrameRows);
  if (a.Ph == 256) hipLaunchKernelGGL((frame_kernel<40, 256>), grid, dim3(kFrameThreads), 0, st, a);
  else hipLaunchKernelGGL((frame_kernel<40, 128>), grid, dim3(kFrameThreads), 0, st, a);
}

}  // namespace ttsdec
