// Micro-benchmark: how fast can ONE CU take in GEMM operand tiles, by LDS-DMA, by plain
// register loads, and by both at once?  (Decides whether the LSTM K loop is bound by the
// L2->CU fabric or by bytes-in-flight x latency.)
//
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/ubench_ingest tools/ubench_ingest.hip
//
// Geometry mirrors lstm_dec at B=256: 256 workgroups of 512 threads, workgroup (x, y) streams
// "A" rows of row block y (4 blocks, shared by 64 workgroups) and "B" rows of column block x
// (64 blocks, shared by 4 workgroups); K = 2560 in 40 tiles of 64; operands are two fp16
// planes, i.e. 16 KiB of A and 16 KiB of B per tile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void global_void;
typedef __attribute__((address_space(1))) const char gbyte;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const u32x4 gu32x4;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int kK = 2560;            // elements per operand row
constexpr int kRowBytes = kK * 2;   // one fp16 plane
constexpr int kTiles = kK / 64;

// DNA / DNB: LDS-DMA wave-instructions per loader wave per tile for A / B (each 1 KiB; 4 loader
// waves; 4+4 = the shipped kernel).  DS: ring stages.  RNB: register-load wave-instructions per
// MFMA-role wave per tile for B (1 KiB each, 4 waves).  RD: tiles of register loads in flight.
template <int DNA, int DNB, int DS, int RNB, int RD, int BT = 0, int AUXB = 0>
__global__ __launch_bounds__(512) void ingest_kernel(const char* a_hi, const char* a_lo, const char* b_hi,
                                                     const char* b_lo, const char* b_packed, int tiles,
                                                     unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int x = blockIdx.x & 63, y = blockIdx.x >> 6;
  constexpr int kStage = (DNA + DNB) * 4 * 1024;
  if (wave >= 4) {
    if constexpr (DNA + DNB > 0) {
      const int w = wave - 4;
      // a wave-instruction covers 8 rows x 128 B (8 lanes x 16 B per row), like the shipped loaders
      const int r8 = lane >> 3, c = lane & 7;
      gbyte* pa[DNA > 0 ? DNA : 1];
      gbyte* pb[DNB > 0 ? DNB : 1];
#pragma unroll
      for (int i = 0; i < DNA; ++i) {
        const int j = w * DNA + i;              // 0 .. 4*DNA-1 ; DNA = 4: j>>3 = plane, (j&7)*8 = row base
        constexpr int D2 = DNA > 0 ? 2 * DNA : 1; const int plane = j / D2, row = (j % D2) * 8 + r8;
        pa[i] = (gbyte*)(plane ? a_lo : a_hi) + (long)(y * 64 + row) * kRowBytes + ((c ^ ((row >> 1) & 7)) * 16);
      }
#pragma unroll
      for (int i = 0; i < DNB; ++i) {
        const int j = w * DNB + i;
        constexpr int D2 = DNB > 0 ? 2 * DNB : 1; const int plane = j / D2, row = (j % D2) * 8 + r8;
        pb[i] = (gbyte*)(plane ? b_lo : b_hi) + (long)(x * 64 + row) * kRowBytes + ((c ^ ((row >> 1) & 7)) * 16);
        if constexpr (BT) pb[i] = (gbyte*)b_packed + (long)x * (kTiles + 16) * (DNB * 4096) + j * 1024 + lane * 16;  // pre-tiled image
      }
      int stage = 0;
      auto issue = [&]() {
        char* st = smem + stage * kStage;
#pragma unroll
        for (int i = 0; i < DNA; ++i) {
          __builtin_amdgcn_global_load_lds((global_void*)pa[i], (lds_void*)(st + (w * DNA + i) * 1024), 16, 0, 0);
          pa[i] += 128;
        }
#pragma unroll
        for (int i = 0; i < DNB; ++i) {
          if constexpr (AUXB == 2) __builtin_amdgcn_global_load_lds((global_void*)pb[i], (lds_void*)(st + DNA * 4096 + (w * DNB + i) * 1024), 16, 0, 2);
          else if constexpr (AUXB == 1) __builtin_amdgcn_global_load_lds((global_void*)pb[i], (lds_void*)(st + DNA * 4096 + (w * DNB + i) * 1024), 16, 0, 1);
          else __builtin_amdgcn_global_load_lds((global_void*)pb[i], (lds_void*)(st + DNA * 4096 + (w * DNB + i) * 1024), 16, 0, 0);
          pb[i] += BT ? DNB * 4096 : 128;
        }
        stage = stage + 1 == DS ? 0 : stage + 1;
      };
      for (int t = 0; t < DS - 1; ++t) issue();
      for (int t = 0; t < tiles; ++t) {
        wait_vmcnt<(DS - 2) * (DNA + DNB)>();
        issue();  // (reads past the end of a row stay inside the allocation: rows are padded by DS tiles)
      }
      wait_vmcnt<0>();
    }
  } else {
    if constexpr (RNB > 0) {
      // pre-tiled B: tile t of column block x is 4*RNB KiB contiguous
      gu32x4* p = (gu32x4*)((gbyte*)b_packed + ((long)x * (kTiles + RD) + 0) * (RNB * 4096) + (wave * RNB) * 1024 + lane * 16);
      u32x4 buf[RD][RNB];
      u32x4 acc = {0, 0, 0, 0};
#pragma unroll
      for (int d = 0; d < RD; ++d) {
#pragma unroll
        for (int i = 0; i < RNB; ++i) buf[d][i] = p[i * 64];
        p += RNB * 4096 / 16;
      }
      for (int t = 0; t < tiles; t += RD) {
#pragma unroll
        for (int d = 0; d < RD; ++d) {
#pragma unroll
          for (int i = 0; i < RNB; ++i) {
            acc ^= buf[d][i];
            buf[d][i] = p[i * 64];
          }
          p += RNB * 4096 / 16;
        }
      }
#pragma unroll
      for (int d = 0; d < RD; ++d)
#pragma unroll
        for (int i = 0; i < RNB; ++i) acc ^= buf[d][i];
      if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
    }
  }
}


// Lean-tile geometry (csrc/gemm_tile.h HK = 1): 16 KiB stages of 32 k (64-byte row pieces), 5 stages, 2 x kTiles tiles.
// MODE 0: row-major operands (a wave-instruction covers 16 rows x 64 B = 16 half cache lines);
// MODE 1: both operands pre-tiled (a wave-instruction reads 1 KiB contiguous); MODE 2: only B pre-tiled.
template <int MODE, int DS>
__global__ __launch_bounds__(512) void ingest_lean_kernel(const char* a_hi, const char* a_lo, const char* b_hi, const char* b_lo,
                                                          const char* a_packed, const char* b_packed, int tiles, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int x = blockIdx.x & 63, y = blockIdx.x >> 6;
  constexpr int kStage = 16 * 1024;
  if (wave < 4) return;
  const int w = wave - 4;
  const int r16 = lane >> 2, c = lane & 3;
  const int row = w * 16 + r16;
  gbyte* p[4];  // A hi, A lo, B hi, B lo
  const long aoff = (long)(y * 64 + row) * kRowBytes + ((c ^ ((row >> 2) & 3)) * 16);
  const long boff = (long)(x * 64 + row) * kRowBytes + ((c ^ ((row >> 2) & 3)) * 16);
  p[0] = (gbyte*)a_hi + aoff; p[1] = (gbyte*)a_lo + aoff; p[2] = (gbyte*)b_hi + boff; p[3] = (gbyte*)b_lo + boff;
  int inc[4] = {64, 64, 64, 64};
  if (MODE == 1) {
    for (int i = 0; i < 2; ++i) { p[i] = (gbyte*)a_packed + (long)y * (2 * kTiles + 16) * 8192 + i * 4096 + w * 1024 + lane * 16; inc[i] = 8192; }
  }
  if (MODE >= 1) {
    for (int i = 2; i < 4; ++i) { p[i] = (gbyte*)b_packed + (long)x * (2 * kTiles + 16) * 8192 + (i - 2) * 4096 + w * 1024 + lane * 16; inc[i] = 8192; }
  }
  int stage = 0;
  auto issue = [&]() {
    char* st = smem + stage * kStage;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((global_void*)p[i], (lds_void*)(st + i * 4096 + w * 1024), 16, 0, 0);
      p[i] += inc[i];
    }
    stage = stage + 1 == DS ? 0 : stage + 1;
  };
  for (int t = 0; t < DS - 1; ++t) issue();
  for (int t = 0; t < tiles; ++t) {
    wait_vmcnt<(DS - 2) * 4>();
    issue();
  }
  wait_vmcnt<0>();
}

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

template <int DNA, int DNB, int DS, int RNB, int RD, int BT = 0, int AUXB = 0>
void run(const char* name, char* a_hi, char* a_lo, char* b_hi, char* b_lo, char* b_packed, unsigned* sink) {
  auto kern = ingest_kernel<DNA, DNB, DS, RNB, RD, BT, AUXB>;
  const int lds = (DNA + DNB) * 4096 * DS;
  CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds > 0 ? lds : 16));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float ms[2];
  const int tl[2] = {kTiles, 0};
  for (int v = 0; v < 2; ++v) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, a_hi, a_lo, b_hi, b_lo, b_packed, tl[v], sink);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, a_hi, a_lo, b_hi, b_lo, b_packed, tl[v], sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms[v], e0, e1));
  }
  const double us = ms[0] * 1000.0 / 200, us0 = ms[1] * 1000.0 / 200;
  const double bytes = (double)kTiles * ((DNA + DNB) * 4096.0 + RNB * 4096.0);
  printf("%-44s lds %3d KiB  %6.2f us/launch (empty %5.2f)  per tile %5.3f us  %6.1f GB/s per CU (loop only %6.1f)\n", name,
         lds / 1024, us, us0, (us - us0) / kTiles, bytes / us * 1e-3, bytes / (us - us0) * 1e-3);
  fflush(stdout);
}

template <int MODE, int DS>
void run_lean(const char* name, char* a_hi, char* a_lo, char* b_hi, char* b_lo, char* a_packed, char* b_packed, unsigned* sink) {
  auto kern = ingest_lean_kernel<MODE, DS>;
  const int lds = 16 * 1024 * DS;
  CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float ms[2];
  const int tl[2] = {2 * kTiles, 0};
  for (int v = 0; v < 2; ++v) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, a_hi, a_lo, b_hi, b_lo, a_packed, b_packed, tl[v], sink);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, a_hi, a_lo, b_hi, b_lo, a_packed, b_packed, tl[v], sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms[v], e0, e1));
  }
  const double us = ms[0] * 1000.0 / 200, us0 = ms[1] * 1000.0 / 200;
  const double bytes = (double)kTiles * 32768.0;
  printf("%-44s lds %3d KiB  %6.2f us/launch (empty %5.2f)  per 64-k %5.3f us  %6.1f GB/s per CU (loop only %6.1f)\n", name,
         lds / 1024, us, us0, (us - us0) / kTiles, bytes / us * 1e-3, bytes / (us - us0) * 1e-3);
  fflush(stdout);
}

int main() {
  const size_t rowpad = 16 * 128;
  const size_t a_bytes = 256 * (size_t)kRowBytes + rowpad, b_bytes = 4096 * (size_t)kRowBytes + rowpad;
  const size_t bp_bytes = 64 * (size_t)(kTiles + 16) * 8 * 4096;
  char *a_hi, *a_lo, *b_hi, *b_lo, *b_packed;
  unsigned* sink;
  CHECK(hipMalloc(&a_hi, a_bytes)); CHECK(hipMalloc(&a_lo, a_bytes));
  CHECK(hipMalloc(&b_hi, b_bytes)); CHECK(hipMalloc(&b_lo, b_bytes));
  CHECK(hipMalloc(&b_packed, bp_bytes)); CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(a_hi, 1, a_bytes)); CHECK(hipMemset(a_lo, 2, a_bytes));
  CHECK(hipMemset(b_hi, 3, b_bytes)); CHECK(hipMemset(b_lo, 4, b_bytes));
  CHECK(hipMemset(b_packed, 5, bp_bytes)); CHECK(hipMemset(sink, 0, 4));
  //            DNA DNB DS RNB RD
  run<4, 4, 4, 0, 1>("dma A+B, 4 stages (shipped)", a_hi, a_lo, b_hi, b_lo, b_packed, sink);
  run<4, 4, 3, 0, 1>("dma A+B, 3 stages", a_hi, a_lo, b_hi, b_lo, b_packed, sink);
  run<4, 0, 4, 0, 1>("dma A only, 4 stages", a_hi, a_lo, b_hi, b_lo, b_packed, sink);
  run<4, 0, 8, 0, 1>("dma A only, 8 stages", a_hi, a_lo, b_hi, b_lo, b_packed, sink);
  run<0, 4, 8, 0, 1>("dma B only, 8 stages", a_hi, a_lo, b_hi, b_lo, b_packed, sink);
  run<0, 0, 1, 4, 2>("reg B only (16 KiB/tile), 2 tiles in flight", a_hi, a_lo, b_hi, b_lo, b_packed, sink);
  run<0, 0, 1, 4, 4>("reg B only, 4 tiles in flight", a_hi, a_lo, b_hi, b_lo, b_packed, sink);
  run<0, 0, 1, 4, 8>("reg B only, 8 tiles in flight", a_hi, a_lo, b_hi, b_lo, b_packed, sink);
  run<0, 0, 1, 8, 4>("reg 32 KiB/tile, 4 tiles in flight", a_hi, a_lo, b_hi, b_lo, b_packed, sink);
  run<4, 0, 4, 4, 4>("dma A (4 st) + reg B (4 in flight)", a_hi, a_lo, b_hi, b_lo, b_packed, sink);
  run<4, 0, 8, 4, 4>("dma A (8 st) + reg B (4 in flight)", a_hi, a_lo, b_hi, b_lo, b_packed, sink);
  run<4, 0, 8, 4, 8>("dma A (8 st) + reg B (8 in flight)", a_hi, a_lo, b_hi, b_lo, b_packed, sink);
  run<4, 0, 6, 4, 8>("dma A (6 st) + reg B (8 in flight)", a_hi, a_lo, b_hi, b_lo, b_packed, sink);
  run<4, 4, 5, 0, 1>("dma A+B, 5 stages", a_hi, a_lo, b_hi, b_lo, b_packed, sink);
  run<0, 4, 4, 0, 1, 1>("dma B only, pre-tiled B, 4 stages", a_hi, a_lo, b_hi, b_lo, b_packed, sink);
  run<0, 4, 8, 0, 1, 1>("dma B only, pre-tiled B, 8 stages", a_hi, a_lo, b_hi, b_lo, b_packed, sink);
  run<4, 4, 4, 0, 1, 1>("dma A+B, pre-tiled B, 4 stages", a_hi, a_lo, b_hi, b_lo, b_packed, sink);
  run<4, 4, 3, 0, 1, 1>("dma A+B, pre-tiled B, 3 stages", a_hi, a_lo, b_hi, b_lo, b_packed, sink);
  run<4, 4, 4, 0, 1, 1, 2>("dma A+B, pre-tiled B nt, 4 stages", a_hi, a_lo, b_hi, b_lo, b_packed, sink);
  run<4, 4, 4, 0, 1, 1, 1>("dma A+B, pre-tiled B sc0, 4 stages", a_hi, a_lo, b_hi, b_lo, b_packed, sink);
  run<4, 4, 4, 0, 1, 0, 2>("dma A+B, row-major B nt, 4 stages", a_hi, a_lo, b_hi, b_lo, b_packed, sink);
  char* a_packed;
  CHECK(hipMalloc(&a_packed, 4 * (size_t)(2 * kTiles + 16) * 8192));
  CHECK(hipMemset(a_packed, 6, 4 * (size_t)(2 * kTiles + 16) * 8192));
  run_lean<0, 5>("lean 16-KiB stages x5, row-major A+B", a_hi, a_lo, b_hi, b_lo, a_packed, b_packed, sink);
  run_lean<0, 4>("lean 16-KiB stages x4, row-major A+B", a_hi, a_lo, b_hi, b_lo, a_packed, b_packed, sink);
  run_lean<2, 5>("lean x5, pre-tiled B", a_hi, a_lo, b_hi, b_lo, a_packed, b_packed, sink);
  run_lean<1, 5>("lean x5, pre-tiled A+B", a_hi, a_lo, b_hi, b_lo, a_packed, b_packed, sink);
  run_lean<1, 4>("lean x4, pre-tiled A+B", a_hi, a_lo, b_hi, b_lo, a_packed, b_packed, sink);
  return 0;
}
