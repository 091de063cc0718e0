// Shared device/host helpers for libttsdec (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ttsdec.h"

namespace ttsdec {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGemmThreads = 256;
constexpr int kAttnThreads = 512;
constexpr int kStopNever = 0x7fffffff;

// Control words living at the head of the decoder workspace.  Written by kernels
// only (initialised by init_state_kernel), read by every step kernel.
struct Ctrl {
  int stop_t;      // first step at which the batch-global stop rule fired, else kStopNever
  int steps_done;  // total steps produced so far (maintained by finish_kernel)
  int pad[62];
};

// ---------------------------------------------------------------------------
// Elementwise math that mirrors the reference's separate ATen ops: every
// multiply/add is individually rounded (no FMA contraction), sqrt and divide are
// correctly rounded.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float div_rn(float a, float b) { return __fdiv_rn(a, b); }
__device__ __forceinline__ float sqrt_rn(float a) { return __fsqrt_rn(a); }

// isru(x) = x / sqrt(1 + 1.0*(x*x))            tacotron/modules/activations.py:9-10
__device__ __forceinline__ float isru(float x) { return div_rn(x, sqrt_rn(add_rn(1.0f, mul_rn(x, x)))); }
// isru_sigmoid(x) = (1 + isru(x/2)) / 2         tacotron/modules/activations.py:5-6
__device__ __forceinline__ float isru_sigmoid(float x) { return mul_rn(add_rn(1.0f, isru(mul_rn(x, 0.5f))), 0.5f); }

__device__ __forceinline__ float sigmoid_f(float x) { return div_rn(1.0f, add_rn(1.0f, expf(-x))); }

// ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011), used for the on-device Prenet dropout.
// keep(step, layer, b, unit) = bit 0 of word 0 of philox(counter = {unit, b, step*2+layer, 0},
// key = {seed_lo, seed_hi}).  The oracle carries the same function.
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t philox_keep(uint64_t seed, uint32_t step, uint32_t layer, uint32_t b,
                                                         uint32_t unit) {
  uint32_t c0 = unit, c1 = b, c2 = step * 2u + layer, c3 = 0u;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c0 & 1u;
}

// ---------------------------------------------------------------------------
// A K-dimension made of up to three row-major segments (the torch.cat inputs of
// the reference: decoder_cell.py:187,191,192).  kend[] are cumulative ends.
// ---------------------------------------------------------------------------
struct Seg3 {
  const float *p0, *p1, *p2;
  int ld0, ld1, ld2;
  int e0, e1, e2;  // cumulative segment ends in the virtual K axis
};

__host__ __device__ inline Seg3 make_seg3(const float* p0, int ld0, int k0, const float* p1, int ld1, int k1,
                                          const float* p2, int ld2, int k2) {
  Seg3 s;
  s.p0 = p0; s.ld0 = ld0; s.e0 = k0;
  s.p1 = p1; s.ld1 = ld1; s.e1 = k0 + k1;
  s.p2 = p2; s.ld2 = ld2; s.e2 = k0 + k1 + k2;
  return s;
}
__host__ __device__ inline Seg3 make_seg2(const float* p0, int ld0, int k0, const float* p1, int ld1, int k1) {
  return make_seg3(p0, ld0, k0, p1, ld1, k1, p1, ld1, 0);
}
__host__ __device__ inline Seg3 make_seg1(const float* p0, int ld0, int k0) {
  return make_seg3(p0, ld0, k0, p0, ld0, 0, p0, ld0, 0);
}

// 4 consecutive k of row `row`; k is a multiple of 4 and segments are multiples of 4
// long, so a float4 never straddles segments.  Out-of-range -> zeros.
__device__ __forceinline__ float4 seg_load4(const Seg3& s, int row, int k) {
  if (k >= s.e2) return make_float4(0.f, 0.f, 0.f, 0.f);
  const float* p;
  int ld, kb;
  if (k < s.e0) { p = s.p0; ld = s.ld0; kb = 0; }
  else if (k < s.e1) { p = s.p1; ld = s.ld1; kb = s.e0; }
  else { p = s.p2; ld = s.ld2; kb = s.e1; }
  return *reinterpret_cast<const float4*>(p + (size_t)row * ld + (k - kb));
}

}  // namespace ttsdec
