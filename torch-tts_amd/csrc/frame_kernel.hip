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
#include "frame_body.h"

namespace ttsdec {

// NI: projection columns per thread (16 * NI >= r*d_mel + r); 6 covers one frame per step with half the partial-sum loads
template <int K0H, int PH, int PREC, int NI>
__global__ __launch_bounds__(kFrameThreads) void frame_kernel(FrameArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[FrameLds<K0H, PH, PREC>::kFloats];
  frame_body<K0H, PH, PREC, NI>(g, lds, blockIdx.x, blockIdx.y);
}

// ===========================================================================
// mel/stop projection [fc_mel; fc_stop] (decoder.py:52-53) as split-K partial sums for the frame kernel (kernels.h ProjArgs)
// ===========================================================================
constexpr int kProjTile = 32;  // rows and columns of a workgroup's output tile
constexpr int kProjNS = 4;     // k16 steps per wave at most

// address of element (row, k) of a segmented activation operand with EB-byte elements
template <int EB>
__device__ __forceinline__ gbyte* seg_elem_ptr(const Seg3& s, int row, int k) {
  const int i = k < s.e0 ? 0 : (k < s.e1 ? 1 : 2);
  const int kk = k - (i == 0 ? 0 : (i == 1 ? s.e0 : s.e1));
  gbyte* p = seg_row_ptr<EB>(s, row, i);
  return p + (s.mpad > 0 ? (long)(kk >> 5) * s.mpad * kChunkBytes + (long)(kk & 31) * EB : (long)kk * EB);
}

template <int PREC>
__global__ __launch_bounds__(kFrameThreads) void proj_kernel(ProjArgs g) {
  constexpr bool F16 = PREC == PREC_F16S;
  constexpr int EB = F16 ? 2 : 4, RS = 33;
  __shared__ __attribute__((aligned(16))) float red[8 * 32 * RS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
  const int n0 = blockIdx.x * kProjTile, m0 = blockIdx.y * kProjTile, z = blockIdx.z;
  const int spw = g.K / (128 * g.ksplit);  // k16 steps per wave, <= kProjNS
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
  if (g.ctrl != nullptr && !step_now(g.ctrl, g.slot).live) return;

  // ---- activations: row m0 + l32, the same K run (A and W only have to agree on which k a lane element means) ----
  f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  {
    const int m = m0 + l32 < g.M ? m0 + l32 : g.M - 1;
    if constexpr (F16) {
      f16x8 ah[kProjNS], al[kProjNS];
#pragma unroll
      for (int j = 0; j < kProjNS; ++j)
        if (j < spw) {
          ah[j] = *(gf16x8*)seg_elem_ptr<2>(g.a, m, kbeg + 8 * j);
          al[j] = *(gf16x8*)seg_elem_ptr<2>(g.a_lo, m, kbeg + 8 * j);
        }
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
      f32x4 af[2 * kProjNS];
#pragma unroll
      for (int j = 0; j < 2 * kProjNS; ++j)
        if (j < 2 * spw) af[j] = *(gf32x4*)seg_elem_ptr<4>(g.a, m, kbeg + 4 * j);
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
    slab[(size_t)m * g.ldo + n] = v;
  }
}

int proj_split(int K) {
  for (int s = 1; s <= 4; s *= 2)
    if (K % (128 * s) == 0 && K / (128 * s) <= kProjNS) return s;
  return 0;
}
void launch_proj(const ProjArgs& a, hipStream_t st) {
  if (a.M <= 0) return;
  dim3 grid((a.N + kProjTile - 1) / kProjTile, (a.M + kProjTile - 1) / kProjTile, a.ksplit), block(kFrameThreads);
  if (a.prec == PREC_F16S) hipLaunchKernelGGL(proj_kernel<PREC_F16S>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(proj_kernel<PREC_F32>, grid, block, 0, st, a);
}

bool frame_supported(int d_mel, int r, int Ph, int P) {
  return d_mel == 80 && (Ph == 256 || Ph == 128) && P % 4 == 0 && r * d_mel + r <= 16 * kFrameMaxNI;
}

void launch_frame(const FrameArgs& a, hipStream_t st) {
  if (a.M <= 0) return;
  const int cols = a.only_finalize ? 1 : (a.P + kFrameCols - 1) / kFrameCols;
  dim3 grid(cols, (a.M + kFrameRows - 1) / kFrameRows), block(kFrameThreads);
  const bool few = a.r * a.d_mel + a.r <= 16 * 6;
  auto go = [&](auto ni) {
    constexpr int NI = decltype(ni)::value;
    if (a.prec == PREC_F16S) {
      if (a.Ph == 256) hipLaunchKernelGGL((frame_kernel<40, 256, PREC_F16S, NI>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((frame_kernel<40, 128, PREC_F16S, NI>), grid, block, 0, st, a);
    } else {
      if (a.Ph == 256) hipLaunchKernelGGL((frame_kernel<40, 256, PREC_F32, NI>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((frame_kernel<40, 128, PREC_F32, NI>), grid, block, 0, st, a);
    }
  };
  if (few) go(std::integral_constant<int, 6>{});
  else go(std::integral_constant<int, kFrameMaxNI>{});
}

}  // namespace ttsdec
