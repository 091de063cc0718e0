// Micro-benchmark: the shipped GEMM core (csrc/gemm_tile.h) on the decoder LSTM's operand shape, tile by tile.
// Question it answers: would a 128x128 output tile with the K axis split over 4 workgroups (half the operand
// bytes per CU, same MFMA work, + a cross-workgroup reduction) beat the shipped 64x64 tile over the whole K?
// Results and what followed from them: profiles/r02_g_ubench_tiles*.txt, DESIGN.md section 5.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I torch-tts_amd/csrc -o gpurun_out/ubench_tiles tools/ubench_tiles.hip
//
// Shape: lstm_dec at B = 256: C[256, 4096] = A[256, 2560] * W[4096, 2560]^T, split-fp16 planes in the chunked layout
// (common.h Seg3), 256 workgroups either way.  The result tile is reduced to one float per thread and written out,
// so the kernels differ from the real ones only by the missing cell update.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "step_bodies.h"

using namespace ttsdec;

constexpr int kM = 256, kN = 4096, kK = 2560;

template <class Cfg, int KSPLIT, int XCD_LOCAL>
__global__ __launch_bounds__(kGemmThreads) void tile_kernel(const f16* ah, const f16* al, const f16* wh, const f16* wl, int K, float* sink, int dbg) {
  __shared__ __attribute__((aligned(16))) float smem[Cfg::kLdsFloats];
  constexpr int NX = kN / Cfg::BN, NY = kM / Cfg::BM;
  int id = blockIdx.x, bx, by, z;
  if (XCD_LOCAL) {
    // the KSPLIT workgroups of a tile, and both row tiles of a column tile, on one XCD (ids go round-robin over 8 XCDs)
    const int xcd = id & 7, j = id >> 3;           // j in [0, NX*NY*KSPLIT/8)
    const int per = NX * NY * KSPLIT / 8;          // workgroups per XCD
    (void)per;
    z = j % KSPLIT;
    const int tile = xcd * (NX * NY / 8) + j / KSPLIT;
    by = tile % NY;
    bx = tile / NY;
  } else {
    bx = id % NX; by = (id / NX) % NY; z = id / (NX * NY);
  }
  const int kc = K / KSPLIT;  // this workgroup's K range [z*kc, z*kc + kc)
  const size_t a_off = (size_t)(z * kc / kChunkK) * kM * kChunkK, w_off = (size_t)(z * kc / kChunkK) * kN * kChunkK;
  const LoaderPlain<2> la{chunked(make_seg1(ah + a_off, kK, kc), kM), chunked(make_seg1(al + a_off, kK, kc), kM), by * Cfg::BM, kM};
  const LoaderPlain<2> lb{chunked(make_seg1(wh + w_off, kK, kc), kN), chunked(make_seg1(wl + w_off, kK, kc), kN), bx * Cfg::BN, kN};
  gemm_tile<Cfg>(la, lb, smem, true, dbg);
  float s = 0.f;
  for (int e = threadIdx.x; e < Cfg::BM * Cfg::BN; e += kGemmThreads) s += smem[(e / Cfg::BN) * Cfg::LDO + e % Cfg::BN];
  sink[(size_t)blockIdx.x * kGemmThreads + threadIdx.x] = s;
}

template <class Cfg, int KSPLIT, int XCD_LOCAL>
static float run(const char* name, const f16* ah, const f16* al, const f16* wh, const f16* wl, int K, float* sink, int dbg = 0) {
  const int grid = (kN / Cfg::BN) * (kM / Cfg::BM) * KSPLIT;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((tile_kernel<Cfg, KSPLIT, XCD_LOCAL>), dim3(grid), dim3(kGemmThreads), 0, 0, ah, al, wh, wl, K, sink, dbg);
  hipEventRecord(e0, 0);
  const int iters = 300;
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((tile_kernel<Cfg, KSPLIT, XCD_LOCAL>), dim3(grid), dim3(kGemmThreads), 0, 0, ah, al, wh, wl, K, sink, dbg);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const float us = ms * 1e3f / iters;
  printf("%-58s grid %4d  K/wg %4d  LDS %3d KiB : %7.2f us per launch\n", name, grid, K / KSPLIT, Cfg::kLdsBytes / 1024, us);
  return us;
}

int main() {
  f16 *ah, *al, *wh, *wl;
  float* sink;
  const size_t na = (size_t)kM * kK, nw = (size_t)kN * kK;
  hipMalloc(&ah, na * 2); hipMalloc(&al, na * 2); hipMalloc(&wh, nw * 2); hipMalloc(&wl, nw * 2);
  hipMalloc(&sink, (size_t)1024 * kGemmThreads * 4);
  std::vector<unsigned short> h(nw);
  for (size_t i = 0; i < nw; ++i) h[i] = 0x2c00 + (unsigned short)(i * 2654435761u >> 22 & 0x3ff);  // finite small halfs
  hipMemcpy(wh, h.data(), nw * 2, hipMemcpyHostToDevice); hipMemcpy(wl, h.data(), nw * 2, hipMemcpyHostToDevice);
  hipMemcpy(ah, h.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(al, h.data(), na * 2, hipMemcpyHostToDevice);

  using Fat64 = TileCfg<2, 2, 1, 4, PREC_F16S>;                 // shipped lstm_dec tile: 64 x 64, 64 k per stage
  using Big128 = TileCfg<2, 2, 1, 4, PREC_F16S, 0, 2, 2, 1>;    // 128 x 128, 32 k per stage, 4 stages
  using Big128s5 = TileCfg<2, 2, 1, 5, PREC_F16S, 0, 2, 2, 1>;  // ... 5 stages (the Postnet's tile)
  using Big128x64 = TileCfg<2, 2, 1, 5, PREC_F16S, 0, 2, 1, 1>;  // 128 x 64
  printf("C[256, 4096] = A[256, K] W[4096, K]^T, split-fp16, chunked planes; 300 back-to-back launches each\n");
  run<Fat64, 1, 0>("64x64 tile, whole K (shipped)", ah, al, wh, wl, kK, sink);
  run<Fat64, 1, 0>("64x64 tile, K = 64 (launch + prologue + epilogue only)", ah, al, wh, wl, 64, sink);
  run<Big128, 4, 1>("128x128 tile, K split over 4 workgroups, XCD-local", ah, al, wh, wl, kK, sink);
  run<Big128, 4, 0>("128x128 tile, K split over 4 workgroups, natural order", ah, al, wh, wl, kK, sink);
  run<Big128s5, 4, 1>("128x128 tile, 5 stages, K split 4, XCD-local", ah, al, wh, wl, kK, sink);
  run<Big128s5, 4, 1>("128x128 tile, 5 stages, K = 128 per wg (fixed costs only)", ah, al, wh, wl, 512, sink);
  run<Big128x64, 2, 1>("128x64 tile, K split over 2 workgroups, XCD-local", ah, al, wh, wl, kK, sink);
  // ablations (gemm_tile's dbg): 1 = every load reads one cached 16-byte block, 3 = no loads inside the K loop
  run<Fat64, 1, 0>("64x64 tile, whole K, all loads from one cached block", ah, al, wh, wl, kK, sink, 1);
  run<Fat64, 1, 0>("64x64 tile, whole K, no in-loop loads", ah, al, wh, wl, kK, sink, 3);
  run<Big128s5, 4, 1>("128x128 tile, 5 stages, split 4, loads from one cached block", ah, al, wh, wl, kK, sink, 1);
  run<Big128s5, 4, 1>("128x128 tile, 5 stages, split 4, no in-loop loads", ah, al, wh, wl, kK, sink, 3);
  // the attention LSTM's K
  run<Fat64, 1, 0>("64x64 tile, K = 1792 (lstm_att)", ah, al, wh, wl, 1792, sink);
  run<Big128s5, 4, 1>("128x128 tile, 5 stages, K = 1792 split 4 (lstm_att)", ah, al, wh, wl, 1792, sink);
  hipDeviceSynchronize();
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
