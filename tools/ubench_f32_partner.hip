// Micro-benchmark: what does a wave that shares a SIMD with an exact-fp32 MFMA chain take from it?
//
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/ubench_f32_partner tools/ubench_f32_partner.hip && gpurun_out/ubench_f32_partner
//
// The fp32 LSTM tile of the decode step (csrc/gemm_tile.h, PREC_F32) runs 16 dependent v_mfma_f32_32x32x2_f32 per K tile on
// each of a workgroup's four MFMA waves (1024 cycles at the instruction's 64-cycle issue interval) and measures ~1700 cycles
// per tile in the kernel: ~1350 with no LDS-DMA in the loop, ~640 for the loader waves alone (tools/ablate_lean.py, round 4).
// This program takes the loop apart: geometry as in the library (256 workgroups x 512 threads = two waves per SIMD; waves 0-3
// run the MFMA chain, waves 4-7 a partner stream), cycles per tile of the MFMA waves by s_memtime, median over workgroups.
//   reads:   0 = operands stay in registers; 1 = the next tile's 8 ds_read_b128 in front of the chain (the library's fp32 form);
//            2 = the same reads placed one per two MFMAs
//   partner: 0 = none; 1 = 64 independent v_add_u32 per tile; 2 = 64 v_fma_f32; 3 = 4 LDS-DMA (1 KiB each, one cached line) with
//            constant addresses; 4 = 4 LDS-DMA + the library's per-instruction address arithmetic (two 64-bit adds and a select
//            per DMA, ~40 VALU per tile); 5 = 4 LDS-DMA through a buffer descriptor with a scalar offset (no per-lane arithmetic);
//            6 = 64 s_add_u32 (scalar only)
//   barrier: 1 = one workgroup barrier per tile (as in the library), 0 = free-running
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void global_void;
typedef __attribute__((address_space(1))) const char gbyte;
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kTiles = 80;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

template <int READS, int PARTNER, int BARRIER, int PRIO = 0, int NV = 16>
__global__ __launch_bounds__(512, 2) void k(const float* src, unsigned long long* out, float* sink) {
  __shared__ __attribute__((aligned(16))) char lds[5 * 16384 + 4096];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  // fill the ring with something (values do not matter for timing as long as they are not all zero)
  for (int i = threadIdx.x; i < (5 * 16384) / 4; i += 512) reinterpret_cast<float*>(lds)[i] = src[i & 4095];
  __syncthreads();
  if (wave < 4) {
    if (PRIO == 1) __builtin_amdgcn_s_setprio(3);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    f32x4 fa[2][4], fb[2][4];
    const int row = lane & 31, half = lane >> 5;
    int offq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) offq[q] = row * 128 + (((half * 4 + q) ^ ((row >> 1) & 7)) << 4);  // the library's swizzled image
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      fa[0][q] = *reinterpret_cast<const f32x4*>(lds + offq[q]);
      fb[0][q] = *reinterpret_cast<const f32x4*>(lds + 8192 + offq[q]);
      fa[1][q] = fa[0][q];
      fb[1][q] = fb[0][q];
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if (BARRIER) __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int stage = 0;
    auto step = [&](auto cur_c) {
      constexpr int cur = decltype(cur_c)::value;
      if (BARRIER) __builtin_amdgcn_s_barrier();
      if constexpr (READS > 0) {
        __builtin_amdgcn_s_waitcnt(0xC07F);
        const char* st = lds + stage * 16384;
        stage = stage == 4 ? 0 : stage + 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          fa[cur ^ 1][q] = *reinterpret_cast<const f32x4*>(st + offq[q]);
          fb[cur ^ 1][q] = *reinterpret_cast<const f32x4*>(st + 8192 + offq[q]);
        }
        if constexpr (READS == 1) {
          __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][q][e], fb[cur][q][e], acc, 0, 0, 0);
    };
    for (int t = 0; t < kTiles; t += 2) {
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 123.456f) sink[threadIdx.x] = s;
    if (lane == 0) out[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
  } else {
    const int w = wave - 4;
    if (PRIO == 2) __builtin_amdgcn_s_setprio(3);
    gbyte* p = (gbyte*)src + lane * 16;
    gbyte* p2 = (gbyte*)src + 4096 + lane * 16;
    long inc = (lane & 1) ? 0 : 0;  // (kept in a register: the compiler cannot fold the adds away)
    asm volatile("" : "+v"(inc));
    unsigned a0 = lane, a1 = lane * 3, a2 = lane * 5, a3 = lane * 7;
    float f0 = lane, f1 = 1.0f, f2 = 0.5f, f3 = 2.f;
    unsigned s0 = 1;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(src, 1u << 20);
    const int voff = lane * 16;
    if (BARRIER) __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    constexpr int kPT = BARRIER ? kTiles : 6 * kTiles;  // free-running: the partner outlasts the MFMA waves
    for (int t = 0; t < kPT; ++t) {
      if (BARRIER) __builtin_amdgcn_s_barrier();
      char* dst = lds + 5 * 16384 + 0 * w;  // (a dummy 4-KiB area: the ring itself is being read by the MFMA waves)
      if constexpr (PARTNER == 7) {
        __builtin_amdgcn_s_sleep(8);  // 8 x 64 cycles: a partner that arrives ~512 cycles after the barrier's release, having issued nothing
      } else if constexpr (PARTNER == 8) {
#pragma unroll
        for (int i = 0; i < NV * 4; ++i) asm volatile("s_nop 3");
      } else if constexpr (PARTNER == 1) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4"
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(lane));
        }
      } else if constexpr (PARTNER == 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          asm volatile("v_fma_f32 %0, %0, %4, %4\n\tv_fma_f32 %1, %1, %4, %4\n\tv_fma_f32 %2, %2, %4, %4\n\tv_fma_f32 %3, %3, %4, %4"
                       : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(0.999f));
        }
      } else if constexpr (PARTNER == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((global_void*)p, (lds_void*)(dst + i * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else if constexpr (PARTNER == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          gbyte* q = (a0 & 0x80000000u) ? p2 : p;  // the library's "partial tile ? zero block : row pointer" select
          __builtin_amdgcn_global_load_lds((global_void*)q, (lds_void*)(dst + i * 1024), 16, 0, 0);
          p += inc;
          p2 += inc;
          a0 += (unsigned)inc;
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else if constexpr (PARTNER == 5) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + i * 1024), 16, voff, (int)(s0 & 3u) * 1024, 0, 0);
          s0 += 1;  // (a scalar offset that moves: four distinct instructions per tile)
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else if constexpr (PARTNER == 6) {
#pragma unroll
        for (int i = 0; i < 64; ++i) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s0));
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345u && f0 + f1 + f2 + f3 == 1.25f && s0 == 77) sink[threadIdx.x] = 1.f;
    if (lane == 0) out[(blockIdx.x * 8 + wave) * 2] = (t1 - t0) / (BARRIER ? 1 : 6);
  }
}

template <int READS, int PARTNER, int BARRIER, int PRIO = 0, int NV = 16>
static void run(const char* name, const float* src, unsigned long long* out, float* sink) {
  std::vector<unsigned long long> h(256 * 8 * 2);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((k<READS, PARTNER, BARRIER, PRIO, NV>), dim3(256), dim3(512), 0, 0, src, out, sink);
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
  hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> m, p;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < 8; ++w) (w < 4 ? m : p).push_back((double)h[(b * 8 + w) * 2] / kTiles);
  std::sort(m.begin(), m.end());
  std::sort(p.begin(), p.end());
  printf("reads %d partner %d barrier %d prio %d nv %2d  %-44s MFMA waves %7.1f cycles/tile (min %7.1f max %7.1f)   partner waves %7.1f\n", READS, PARTNER,
         BARRIER, PRIO, NV, name, m[m.size() / 2], m.front(), m.back(), p[p.size() / 2]);
}

int main() {
  float* src;
  unsigned long long* out;
  float* sink;
  hipMalloc(&src, 1 << 20);
  hipMalloc(&out, 256 * 8 * 2 * 8);
  hipMalloc(&sink, 4096);
  std::vector<float> h((1 << 20) / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
  hipMemcpy(src, h.data(), 1 << 20, hipMemcpyHostToDevice);
  printf("16 dependent v_mfma_f32_32x32x2_f32 per tile = 1024 cycles at the 64-cycle issue interval; %d tiles\n", kTiles);
#define RUN(R, P, B, NAME) run<R, P, B>(NAME, src, out, sink)
  RUN(0, 0, 0, "bare chain");
  RUN(1, 0, 0, "reads in front");
  RUN(2, 0, 0, "reads interleaved");
  RUN(1, 0, 1, "reads in front, barrier");
  RUN(2, 0, 1, "reads interleaved, barrier");
  RUN(0, 1, 1, "partner 64 v_add_u32, barrier");
  RUN(0, 2, 1, "partner 64 v_fma_f32, barrier");
  RUN(0, 3, 1, "partner 4 LDS-DMA const addr, barrier");
  RUN(0, 4, 1, "partner 4 LDS-DMA + address VALU, barrier");
  RUN(0, 5, 1, "partner 4 LDS-DMA buffer, barrier");
  RUN(0, 1, 0, "partner 64 v_add_u32");
  RUN(0, 2, 0, "partner 64 v_fma_f32");
  RUN(0, 6, 0, "partner 64 s_add_u32");
  RUN(0, 3, 0, "partner 4 LDS-DMA const addr");
  RUN(0, 4, 0, "partner 4 LDS-DMA + address VALU");
  RUN(0, 5, 0, "partner 4 LDS-DMA buffer/scalar offset");
  RUN(1, 4, 1, "library form: reads in front, DMA+VALU, barrier");
  RUN(2, 4, 1, "reads interleaved, DMA+VALU, barrier");
  RUN(2, 3, 1, "reads interleaved, DMA const, barrier");
  RUN(2, 5, 1, "reads interleaved, DMA buffer, barrier");
  run<0, 0, 1>("barrier only, partner idle", src, out, sink);
  run<0, 7, 1>("partner sleeps 512 cycles, barrier", src, out, sink);
  run<0, 8, 1>("partner 64 s_nop 3, barrier", src, out, sink);
  run<0, 1, 1, 0, 4>("partner 16 v_add_u32, barrier", src, out, sink);
  run<0, 1, 1, 0, 8>("partner 32 v_add_u32, barrier", src, out, sink);
  run<0, 1, 1, 0, 32>("partner 128 v_add_u32, barrier", src, out, sink);
  run<0, 1, 1, 1, 16>("partner 64 v_add_u32, barrier, MFMA waves prio 3", src, out, sink);
  run<0, 1, 1, 2, 16>("partner 64 v_add_u32, barrier, partner prio 3", src, out, sink);
  run<0, 3, 1, 1>("partner 4 LDS-DMA const, barrier, MFMA prio 3", src, out, sink);
  run<0, 3, 1, 2>("partner 4 LDS-DMA const, barrier, partner prio 3", src, out, sink);
  run<2, 4, 1, 1>("reads interleaved, DMA+VALU, barrier, MFMA prio 3", src, out, sink);
  run<2, 4, 1, 2>("reads interleaved, DMA+VALU, barrier, partner prio 3", src, out, sink);
  return 0;
}
