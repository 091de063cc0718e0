#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// Where inside a K tile do the two waves of a SIMD spend their time when a workgroup barrier couples them?  (round 4)
// per-tile stamps: MFMA wave 0: [after barrier, before next barrier]; partner wave 4: same
template <int PARTNER, int CHAIN = 0>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* out, float* sink) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  unsigned long long st[32];
  if (wave < 4) {
    f32x16 acc, acc2;
    f32x4 c4a, c4b;
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 0.f; }
    for (int i = 0; i < 4; ++i) { c4a[i] = 0.f; c4b[i] = 0.f; }
    float a = lane * 0.001f, b = 1.0f - lane * 0.002f;
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0); st[2 * t] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
      if constexpr (CHAIN == 0) {  // one accumulator: 16 dependent 32x32x2
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      } else if constexpr (CHAIN == 1) {  // two accumulators alternating
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc2, 0, 0, 0);
        }
      } else {  // the same 1024 cycles as 32 v_mfma_f32_16x16x4_f32 on two accumulators
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          c4a = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c4a, 0, 0, 0);
          c4b = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c4b, 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0); st[2 * t + 1] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i] + acc2[i];
    for (int i = 0; i < 4; ++i) s += c4a[i] + c4b[i];
    if (s == 123.456f) sink[threadIdx.x] = s;
  } else {
    unsigned a0 = lane, a1 = lane * 3, a2 = lane * 5, a3 = lane * 7;
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0); st[2 * t] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
      if (PARTNER == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(lane));
      } else if (PARTNER == 2) {
        __builtin_amdgcn_s_sleep(8);
      }
      __builtin_amdgcn_sched_barrier(0); st[2 * t + 1] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
    }
    if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345u) sink[threadIdx.x] = 1.f;
  }
  if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4))
    for (int i = 0; i < 32; ++i) out[(wave / 4) * 32 + i] = st[i];
}
template <int P, int C = 0> void run(unsigned long long* out, float* sink) {
  hipLaunchKernelGGL((k<P, C>), dim3(256), dim3(512), 0, 0, out, sink);
  hipLaunchKernelGGL((k<P, C>), dim3(256), dim3(512), 0, 0, out, sink);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(64);
  hipMemcpy(h.data(), out, 64 * 8, hipMemcpyDeviceToHost);
  unsigned long long t0 = h[0];
  printf("partner %d chain %d\n tile: MFMA wave [after barrier .. chain issued]   partner [after barrier .. work done]\n", P, C);
  for (int t = 4; t < 10; ++t)
    printf("  %2d: %6lld .. %6lld     %6lld .. %6lld\n", t, (long long)(h[2 * t] - t0), (long long)(h[2 * t + 1] - t0), (long long)(h[32 + 2 * t] - t0), (long long)(h[32 + 2 * t + 1] - t0));
}
int main() {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 64 * 8); hipMalloc(&sink, 4096);
  run<0>(out, sink); run<1>(out, sink); run<2>(out, sink);
  run<1, 1>(out, sink); run<1, 2>(out, sink); run<0, 2>(out, sink);
  return 0;
}
