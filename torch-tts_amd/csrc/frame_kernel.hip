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
  loop_stamp(g.ctrl, g.slot, g.node);
  __shared__ __attribute__((aligned(16))) float lds[FrameLds<K0H, PH, PREC>::kFloats];
  frame_body<K0H, PH, PREC, NI>(g, lds, blockIdx.x, blockIdx.y);
}

template <int PREC>
__global__ __launch_bounds__(kFrameThreads) void proj_kernel(ProjArgs g) {
  loop_stamp(g.ctrl, g.slot, g.node);
  __shared__ __attribute__((aligned(16))) float red[kProjLdsFloats];
  proj_body<PREC, true>(g, red, blockIdx.x);
}

int proj_split(int K) {
  for (int s = 1; s <= 4; s *= 2)
    if (K % (128 * s) == 0 && K / (128 * s) <= kProjNS) return s;
  return 0;
}
int proj_grid_size(int M, int N, int ksplit) { return ((N + kProjTile - 1) / kProjTile) * ((M + kProjTile - 1) / kProjTile) * ksplit; }
void launch_proj(const ProjArgs& a, hipStream_t st) {
  if (a.M <= 0) return;
  dim3 grid(proj_grid_size(a.M, a.N, a.ksplit)), block(kFrameThreads);
  if (a.prec == PREC_F16S) hipLaunchKernelGGL(proj_kernel<PREC_F16S>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(proj_kernel<PREC_F32>, grid, block, 0, st, a);
}

// [proj(t-1) | frame(t)]: the projection as a head role of the FRAME kernel's own launch - the form of fused_kernels.hip's
// three-role launch for configurations that run one role per launch otherwise (exact fp32 in the middle batch range, two frames
// per step): the step still loses the projection's launch.  Projection workgroups have the lowest block ids and wait for
// nothing; a frame workgroup waits for the projection workgroups of its 32-row block (frame_body kHead).
template <int K0H, int PH, int PREC, int NI>
__global__ __launch_bounds__(kFrameThreads) void proj_frame_kernel(ProjArgs pj, FrameArgs g, int n_proj, int frame_cols) {
  loop_stamp(g.ctrl, g.slot, g.node);
  constexpr int kF = FrameLds<K0H, PH, PREC>::kFloats;
  __shared__ __attribute__((aligned(16))) float lds[kF > kProjLdsFloats ? kF : kProjLdsFloats];
  int id = blockIdx.x;
  if (id < n_proj) {
    proj_body<PREC>(pj, lds, id);
    return;
  }
  id -= n_proj;
  frame_body<K0H, PH, PREC, NI, false, true>(g, lds, id % frame_cols, id / frame_cols);
}

void launch_proj_frame(const ProjArgs& pj, const FrameArgs& a, hipStream_t st) {
  if (a.M <= 0) return;
  const int cols = (a.P + kFrameCols - 1) / kFrameCols, n_frame = cols * ((a.M + kFrameRows - 1) / kFrameRows);
  const int n_proj = proj_grid_size(pj.M, pj.N, pj.ksplit);
  dim3 grid(n_proj + n_frame), block(kFrameThreads);
  const bool few = a.r * a.d_mel + a.r <= 16 * 6;
  auto go = [&](auto ni) {
    constexpr int NI = decltype(ni)::value;
    if (a.prec == PREC_F16S) {
      if (a.Ph == 256) hipLaunchKernelGGL((proj_frame_kernel<40, 256, PREC_F16S, NI>), grid, block, 0, st, pj, a, n_proj, cols);
      else hipLaunchKernelGGL((proj_frame_kernel<40, 128, PREC_F16S, NI>), grid, block, 0, st, pj, a, n_proj, cols);
    } else {
      if (a.Ph == 256) hipLaunchKernelGGL((proj_frame_kernel<40, 256, PREC_F32, NI>), grid, block, 0, st, pj, a, n_proj, cols);
      else hipLaunchKernelGGL((proj_frame_kernel<40, 128, PREC_F32, NI>), grid, block, 0, st, pj, a, n_proj, cols);
    }
  };
  if (few) go(std::integral_constant<int, 6>{});
  else go(std::integral_constant<int, kFrameMaxNI>{});
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
