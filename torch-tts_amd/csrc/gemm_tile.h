// LDS-tiled fp32 GEMM main loop on the CDNA4 fp32-input matrix instruction
// v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fmaf chain per output).
//
//   C[BM x BN] = sum_k A[row, k] * B[col, k]        (both operands K-contiguous,
//                                                   i.e. activations [M, K] and
//                                                   PyTorch-layout weights [N, K])
//
// One workgroup = 4 waves (256 threads) arranged WM x WN x WK; every wave owns one
// 32x32 accumulator (16 VGPRs).  BM = 32*WM, BN = 32*WN; each main-loop iteration
// consumes KT = 32*WK of K, wave (.,.,wk) taking the wk-th 32-wide slice (intra-
// workgroup split-K, summed through LDS in the epilogue).  Tiles are staged
// global -> registers -> LDS with the next tile's global loads issued before the
// current tile's MFMAs (one barrier per iteration, two LDS buffers).
//
// LDS rows are padded by one 16-byte access (LDK = KT + 4 floats) which makes the
// ds_read_b128 fragment reads conflict-free (row stride 36/68/132 dwords: 16 distinct
// rows of a lane group land on 16 distinct 4-bank slots).
//
// K permutation: within a 32-wide slice lane half h = lane>>5 supplies k = 16h + 4q + e
// for MFMA (q, e); A and B use the same map, so each MFMA's two k-slots pair up
// correctly and every k is consumed exactly once.
#pragma once
#include "common.h"

namespace ttsdec {

template <int WM, int WN, int WK>
struct TileCfg {
  static_assert(WM * WN * WK == 4, "4 waves per workgroup");
  static constexpr int BM = 32 * WM;
  static constexpr int BN = 32 * WN;
  static constexpr int KT = 32 * WK;
  static constexpr int LDK = KT + 4;
  static constexpr int NA = BM * KT / 4 / kGemmThreads;  // float4 per thread per A tile
  static constexpr int NB = BN * KT / 4 / kGemmThreads;
  static constexpr int LDO = BN + 1;
  static constexpr int kStageFloats = 2 * (BM + BN) * LDK;
  static constexpr int kOutFloats = WK * BM * LDO;
  static constexpr int kLdsFloats = kStageFloats > kOutFloats ? kStageFloats : kOutFloats;
};

// LoaderA / LoaderB: `float4 load(int r, int k) const` returns 4 consecutive-k values
// of tile row r (0 <= r < BM or BN) at virtual k (multiple of 4), zero-filled outside
// the operand.  After the call `smem` holds the BM x BN result, row-major with leading
// dimension Cfg::LDO, summed over the WK slices, visible to all threads.
template <class Cfg, class LoaderA, class LoaderB>
__device__ __forceinline__ void gemm_tile_f32(const LoaderA& la, const LoaderB& lb, int K, float* smem) {
  constexpr int BM = Cfg::BM, BN = Cfg::BN, KT = Cfg::KT, LDK = Cfg::LDK;
  constexpr int NA = Cfg::NA, NB = Cfg::NB, LDO = Cfg::LDO;
  constexpr int WN_ = BN / 32, WK_ = KT / 32;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wk = wave % WK_;
  const int wn = (wave / WK_) % WN_;
  const int wm = wave / (WK_ * WN_);
  const int half = lane >> 5;
  const int l32 = lane & 31;

  float* As = smem;                  // [2][BM][LDK]
  float* Bs = smem + 2 * BM * LDK;   // [2][BN][LDK]

  float4 ra[NA], rb[NB];
  constexpr int C4 = KT / 4;  // float4 per tile row

  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int f = tid + i * kGemmThreads;
      ra[i] = la.load(f / C4, k0 + (f % C4) * 4);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int f = tid + i * kGemmThreads;
      rb[i] = lb.load(f / C4, k0 + (f % C4) * 4);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int f = tid + i * kGemmThreads;
      *reinterpret_cast<float4*>(As + (buf * BM + f / C4) * LDK + (f % C4) * 4) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int f = tid + i * kGemmThreads;
      *reinterpret_cast<float4*>(Bs + (buf * BN + f / C4) * LDK + (f % C4) * 4) = rb[i];
    }
  };

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  const int nk = (K + KT - 1) / KT;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * KT);
    const float* ap = As + (buf * BM + wm * 32 + l32) * LDK + wk * 32 + half * 16;
    const float* bp = Bs + (buf * BN + wn * 32 + l32) * LDK + wk * 32 + half * 16;
    float4 a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a[q] = *reinterpret_cast<const float4*>(ap + 4 * q);
      b[q] = *reinterpret_cast<const float4*>(bp + 4 * q);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, b[q].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, b[q].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, b[q].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, b[q].w, acc, 0, 0, 0);
    }
    if (kt + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

  // accumulators -> LDS out tile (aliases the staging buffers; the loop ended on a barrier).
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
  float* out = smem + wk * BM * LDO;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    out[row * LDO + wn * 32 + l32] = acc[r];
  }
  __syncthreads();
  if (WK_ > 1) {
    for (int e = tid; e < BM * BN; e += kGemmThreads) {
      const int row = e / BN, col = e % BN;
      float v = smem[row * LDO + col];
#pragma unroll
      for (int s = 1; s < WK_; ++s) v = add_rn(v, smem[s * BM * LDO + row * LDO + col]);
      smem[row * LDO + col] = v;
    }
    __syncthreads();
  }
}

}  // namespace ttsdec
