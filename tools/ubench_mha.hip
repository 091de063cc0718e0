// Micro-benchmark: the VITS2 attention kernel (csrc/vits2.hip mha_mfma_kernel) on the reverse flow's and the text encoder's
// shapes, with its passes switched off one at a time, to see where a workgroup's time goes.  Two copies of the kernel:
// the one that shipped until r02_i ("full" lines) and the one that ships now ("candidate" lines); the program also checks
// that the two produce the same bits.  Results: profiles/r02_j_ubench_mha.txt, DESIGN.md section 4.9.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I torch-tts_amd/csrc -o gpurun_out/ubench_mha tools/ubench_mha.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <type_traits>
#include <vector>

#include "common.h"

using namespace ttsdec;

struct MhaArgs {
  const float* qkv;  // [B*T, 3C]: q | k | v, head h = channels [h*dk, (h+1)*dk)
  const float* mask;  // [B*T]
  const float *ek, *ev;  // [2w+1, dk] or nullptr
  float* out;            // [B*T, C]
  f16* out_p;            // optional split-fp16 planes of out (hi, then lo n_out halfs later)
  size_t n_out;          // B*T*C
  int T, C, dk, window;
  float qscale;  // sqrt(dk): q is DIVIDED by it, as the reference does
};

constexpr int kMhaMRows = 32, kMhaMThreads = 256;
template <int DKH, int ABL>
__global__ __launch_bounds__(kMhaMThreads) void mha_mfma_kernel(MhaArgs g) {
  constexpr int DK = 2 * DKH, NDT = (DK + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T = g.T, C = g.C, w = g.window;
  const int ST = T | 1;                    // odd row stride: the 32 rows of a column hit 32 banks
  float* S = sm;                           // [32][ST]   (aliased by the pass-2 reduction buffer)
  float* R = S + kMhaMRows * ST;           // [32][33] relative-key logits (windowed attention only)
  const int nrel = w >= 0 ? 2 * w + 1 : 0;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
  const int b = blockIdx.z, hd = blockIdx.y, i0 = blockIdx.x * kMhaMRows;
  const size_t rowb = (size_t)b * T;
  const float* base = g.qkv + rowb * 3 * C + hd * DK;
  typedef __attribute__((address_space(1))) const f32x4 gf32x4;

  // A fragments of Q (scaled): row i0 + l32, k in [half*DKH, +DKH)
  f32x4 qa[DKH / 4];
  {
    const int i = i0 + l32 < T ? i0 + l32 : T - 1;
    gf32x4* src = (gf32x4*)(base + (size_t)i * 3 * C + half * DKH);
#pragma unroll
    for (int j = 0; j < DKH / 4; ++j) {
      f32x4 v = src[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = div_rn(v[e], g.qscale);
      qa[j] = v;
    }
  }
  // relative-key logits: R[i][r] = q_i . E_k[r]   (wave 0; rows r >= nrel of the B operand are zero)
  if (w >= 0 && wave == 0) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < DKH / 4; ++j) {
      f32x4 e4 = {0.f, 0.f, 0.f, 0.f};
      if (l32 < nrel) e4 = *reinterpret_cast<const f32x4*>(g.ek + (size_t)l32 * DK + half * DKH + 4 * j);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[j][e], e4[e], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) R[((r & 3) + 8 * (r >> 2) + 4 * half) * 33 + l32] = acc[r];
  }
  __syncthreads();
  // ---- pass 1: scores, 32 keys per tile, tiles round-robin over the waves; the next tile's K
  // fragments are requested before the current tile's MFMAs ----
  const int ntile = (T + 31) / 32;
  auto load_k = [&](int kt, f32x4 (&kb)[DKH / 4]) {
    const int j = kt * 32 + l32;
    gf32x4* src = (gf32x4*)(base + (size_t)(j < T ? j : T - 1) * 3 * C + C + half * DKH);
#pragma unroll
    for (int q = 0; q < DKH / 4; ++q) kb[q] = src[q];
  };
  float mrow[16];  // frame mask of the 16 query rows this lane's accumulator registers belong to
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * half;
    mrow[r] = i < T ? g.mask[rowb + i] : 0.f;
  }
  auto score_tile = [&](int kt, const f32x4 (&kb)[DKH / 4]) {
    const int j = kt * 32 + l32;
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < DKH / 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[q][e], kb[q][e], acc, 0, 0, 0);
    if (j < T) {
      const float mj = g.mask[rowb + j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int i = i0 + row;
        float s = acc[r];
        const int rr = j - i + w;
        if (w >= 0 && rr >= 0 && rr <= 2 * w) s = add_rn(s, R[row * 33 + rr]);
        if (mrow[r] * mj == 0.f) s = -1e4f;
        S[row * ST + j] = s;
      }
    }
  };
  {
    f32x4 kb0[DKH / 4], kb1[DKH / 4];
    if (wave < ntile) load_k(wave, kb0);
    if (ABL < 3) for (int kt = wave; kt < ntile; kt += 8) {
      if (kt + 4 < ntile) load_k(kt + 4, kb1);
      score_tile(kt, kb0);
      if (kt + 4 < ntile) {
        if (kt + 8 < ntile) load_k(kt + 8, kb0);
        score_tile(kt + 4, kb1);
      }
    }
  }
  __syncthreads();
  // ---- softmax numerators: wave v owns rows 8v .. 8v+7.  S keeps e = exp(s - max); the division by the
  // row sum is applied once to the 32 x dk output instead of to the 32 x T scores ----
  float* rinv = S + kMhaMRows * ST + (w >= 0 ? 32 * 33 : 0);  // [32] 1 / row sum
  if (ABL < 2) for (int rr = 0; rr < 8; ++rr) {
    float* row = S + (wave * 8 + rr) * ST;
    float mx = -3.4e38f;
    for (int j = lane; j < T; j += 64) mx = fmaxf(mx, row[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < T; j += 64) {
      const float e = __expf(row[j] - mx);
      row[j] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) rinv[wave * 8 + rr] = 1.0f / sum;
  }
  __syncthreads();
  // ---- pass 2: out = P V; wave v takes the key pairs {v, v+4, v+8, ...} ----
  f32x16 acc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) acc[dt] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  {
    const float* vb = base + 2 * C;
    const float* prow = S + l32 * ST;
    const int npair = (T + 1) / 2;
    constexpr int G = 8;  // key pairs whose V values are requested together (latency paid once per group)
    if (ABL < 1) for (int p0 = wave * G; p0 < npair; p0 += 4 * G) {
      float pv[G], vv[G][NDT];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int key = 2 * (p0 + u) + half;
        const bool ok = key < T;
        pv[u] = ok ? prow[key] : 0.f;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const int d = dt * 32 + l32;
          vv[u][dt] = (ok && d < DK) ? vb[(size_t)key * 3 * C + d] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < G; ++u)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(pv[u], vv[u][dt], acc[dt], 0, 0, 0);
    }
  }
  // relative values need p[i, i + r - w]: read them (and the row's 1/sum) before S is reused as the reduction buffer
  const int oi = tid >> 3;  // output row of this thread in the final loop (32 rows x 8 threads)
  const float ri = rinv[oi];
  float prel[32];
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    const int j = i0 + oi + r - w;
    prel[r] = (w >= 0 && r < nrel && j >= 0 && j < T) ? S[oi * ST + j] : 0.f;
  }
  __syncthreads();
  float* red = S;  // [4 waves][32][DK + 1]
  constexpr int RS = DK + 1;
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) {
    const int d = dt * 32 + l32;
    if (d < DK) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * RS + d] = acc[dt][r];
    }
  }
  __syncthreads();
  if (i0 + oi < T) {
    for (int d = tid & 7; d < DK; d += 8) {
      float v = red[oi * RS + d];
#pragma unroll
      for (int q = 1; q < 4; ++q) v = add_rn(v, red[(q * 32 + oi) * RS + d]);
      if (w >= 0) {
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r)
          if (r < nrel) a = fmaf(prel[r], g.ev[(size_t)r * DK + d], a);
        v = add_rn(v, a);
      }
      const size_t o = (rowb + i0 + oi) * C + hd * DK + d;
      g.out[o] = v * ri;
      if (g.out_p) split_f16(v * ri, g.out_p[o], g.out_p[g.n_out + o]);
    }
  }
}

// ---- the kernel as it ships now (csrc/vits2.hip mha_mfma_kernel, pasted with the same ablation switches) ----
template <int DKH, int ABL = 0>
__global__ __launch_bounds__(kMhaMThreads, 2) void mha_v2(MhaArgs g) {
  constexpr int DK = 2 * DKH, NDT = (DK + 31) / 32, NQ = DKH / 4;
  constexpr int DK1 = DKH > 24 ? 2 : 4, DV = DKH > 24 ? 2 : 4, G = 8;  // tiles / groups in flight per wave
  constexpr int NEV = (31 * DK + kMhaMThreads - 1) / kMhaMThreads;  // E_v values per thread (window <= 15)
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T = g.T, C = g.C, w = g.window;
  const int ST = T | 1;
  float* S = sm;
  float* R = S + kMhaMRows * ST;
  const int nrel = w >= 0 ? 2 * w + 1 : 0;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l32 = lane & 31, half = lane >> 5;
  const int b = blockIdx.z, hd = blockIdx.y, i0 = blockIdx.x * kMhaMRows;
  const size_t rowb = (size_t)b * T;
  const float* base = g.qkv + rowb * 3 * C + hd * DK;
  typedef __attribute__((address_space(1))) const f32x4 gf32x4;
  typedef __attribute__((address_space(1))) const float gf32;
  gf32* maskg = (gf32*)(g.mask + rowb);
  const int ntile = (T + 31) / 32;

  f32x4 qa[NQ];
  {
    const int i = i0 + l32 < T ? i0 + l32 : T - 1;
    gf32x4* src = (gf32x4*)(base + (size_t)i * 3 * C + half * DKH);
#pragma unroll
    for (int j = 0; j < NQ; ++j) qa[j] = src[j];
  }
  f32x4 e4[NQ];
  if (w >= 0 && wave == 0) {
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      e4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (l32 < nrel) e4[j] = *(gf32x4*)(g.ek + (size_t)l32 * DK + half * DKH + 4 * j);
    }
  }
  float mrow[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * half;
    mrow[r] = i < T ? maskg[i] : 0.f;
  }
  f32x4 kb[DK1][NQ];
  float mj[DK1];
  auto load_k = [&](int kt, int u) {
    const int j = kt * 32 + l32, jc = j < T ? j : T - 1;
    gf32x4* src = (gf32x4*)(base + (size_t)jc * 3 * C + C + half * DKH);
#pragma unroll
    for (int q = 0; q < NQ; ++q) kb[u][q] = src[q];
    mj[u] = maskg[jc];
  };
#pragma unroll
  for (int u = 0; u < DK1; ++u) load_k(wave + 4 * u, u);  // past the last tile: the clamped row again (no branch, so the
                                                           // compiler can count the loads in flight exactly)
  float evr[NEV];
#pragma unroll
  for (int q = 0; q < NEV; ++q) {
    const int e = tid + q * kMhaMThreads;
    evr[q] = (w >= 0 && e < nrel * DK) ? ((gf32*)g.ev)[e] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < NQ; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) qa[j][e] = div_rn(qa[j][e], g.qscale);
  if (w >= 0 && wave == 0) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NQ; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[j][e], e4[j][e], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) R[((r & 3) + 8 * (r >> 2) + 4 * half) * 33 + l32] = acc[r];
  }
  lds_barrier();
  // ---- pass 1 ----
  auto score_tile = [&](int kt, int u) {
    const int j = kt * 32 + l32;
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[q][e], kb[u][q][e], acc, 0, 0, 0);
    if (j < T) {
      const float mjv = mj[u];
      float* sj = S + j;
      // does any (query, key) pair of this tile lie inside the relative window?  (the same answer in every lane)
      const bool near = w >= 0 && kt * 32 + 31 + w >= i0 && kt * 32 <= i0 + 31 + w;
      if (near) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
          float s = acc[r];
          const int rr = j - (i0 + row) + w;
          if (rr >= 0 && rr <= 2 * w) s = add_rn(s, R[row * 33 + rr]);
          if (mrow[r] * mjv == 0.f) s = -1e4f;
          sj[row * ST] = s;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
          sj[row * ST] = mrow[r] * mjv == 0.f ? -1e4f : acc[r];
        }
      }
    }
  };
  if (ABL < 3) for (int kt0 = wave; kt0 < ntile; kt0 += 4 * DK1) {
#pragma unroll
    for (int u = 0; u < DK1; ++u) {
      const int kt = kt0 + 4 * u;
      if (kt < ntile) score_tile(kt, u);
      load_k(kt + 4 * DK1, u);
    }
  }
  lds_barrier();
  // ---- softmax numerators, the wave's 8 rows side by side ----
  float* rinv = S + kMhaMRows * ST + (w >= 0 ? 32 * 33 : 0);
  if (ABL < 2) {
    float* row0 = S + (wave * 8) * ST;
    float mx[8], sum[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) mx[q] = -3.4e38f, sum[q] = 0.f;
    for (int j = lane; j < T; j += 64)
#pragma unroll
      for (int q = 0; q < 8; ++q) mx[q] = fmaxf(mx[q], row0[q * ST + j]);
#pragma unroll
    for (int q = 0; q < 8; ++q) mx[q] = wave_max(mx[q]);
    for (int j = lane; j < T; j += 64)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float e = __expf(row0[q * ST + j] - mx[q]);
        row0[q * ST + j] = e;
        sum[q] += e;
      }
#pragma unroll
    for (int q = 0; q < 8; ++q) sum[q] = wave_sum(sum[q]);
    if (lane == 0)
#pragma unroll
      for (int q = 0; q < 8; ++q) rinv[wave * 8 + q] = 1.0f / sum[q];
  }
  lds_barrier();
  // ---- pass 2 ----
  f32x16 acc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) acc[dt] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  {
    gf32* vb = (gf32*)(base + 2 * C);
    const float* prow = S + l32 * ST;
    const int npair = (T + 1) / 2;
    struct VN { float v[NDT]; };  // DK % NDT == 0 for the built sizes: a lane's columns are all inside or all outside
    static_assert(DK % NDT == 0, "");
    float pv[DV][G], vv[DV][G][NDT];
    // one key pair of a group: its P value (LDS) and the lane's NDT adjacent V values (one global load)
    // lanes whose columns lie past DK read column 0 instead: their accumulators are never stored
    gf32* vlane = vb + (NDT * l32 < DK ? NDT * l32 : 0) + (size_t)half * 3 * C;
    const float* plane = prow + half;
    auto load_1 = [&](int p0, int s, int u) {
      if (2 * (p0 + G) <= T) {  // every key of the group exists (the same answer in all lanes): no predicates
        pv[s][u] = plane[2 * (p0 + u)];
        gf32* vk = vlane + (size_t)(2 * (p0 + u)) * 3 * C;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) vv[s][u][dt] = vk[dt];
      } else {
        const int key = 2 * (p0 + u) + half;
        const bool ok = key < T;
        const int kc = ok ? key : T - 1;
        const float pl = prow[kc];
        gf32* vk = vlane + (size_t)(kc - half) * 3 * C;
        pv[s][u] = ok ? pl : 0.f;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const float vl = vk[dt];
          vv[s][u][dt] = ok ? vl : 0.f;
        }
      }
    };
#pragma unroll
    for (int s = 0; s < DV; ++s)
      if (wave * G + s * 4 * G < npair) {
#pragma unroll
        for (int u = 0; u < G; ++u) load_1(wave * G + s * 4 * G, s, u);
      }
    if (ABL < 1) for (int pb = wave * G; pb < npair; pb += DV * 4 * G) {
#pragma unroll
      for (int s = 0; s < DV; ++s) {
        const int p0 = pb + s * 4 * G;
        if (p0 < npair) {
          const bool more = p0 + DV * 4 * G < npair;
          // the slot's next key pair is requested right behind the MFMAs that consumed the old one, so the address
          // arithmetic runs while the matrix pipe is busy instead of after the whole group
#pragma unroll
          for (int u = 0; u < G; ++u) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(pv[s][u], vv[s][u][dt], acc[dt], 0, 0, 0);
            if (more) load_1(p0 + DV * 4 * G, s, u);
          }
        }
      }
    }
  }
  // relative values need p[i, i + r - w]: read them (and the row's 1/sum) before S is reused as the reduction buffer.
  // NR: compile-time bound on the window rows (9 covers the reference's window_size = 4)
  auto finish = [&](auto nr_c) {
    constexpr int NR = decltype(nr_c)::value;
    const int oi = tid >> 3;  // output row of this thread (32 rows x 8 threads)
    const float ri = rinv[oi];
    float prel[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int j = i0 + oi + r - w;
      prel[r] = (r < nrel && j >= 0 && j < T) ? S[oi * ST + j] : 0.f;
    }
    lds_barrier();
    float* red = S;  // [4 waves][32][DK + 1], then E_v [nrel][DK]
    constexpr int RS = DK + 1;
    float* evs = red + 4 * 32 * RS;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const int d = NDT * l32 + dt;
      if (d < DK) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * RS + d] = acc[dt][r];
      }
    }
#pragma unroll
    for (int q = 0; q < NEV; ++q) {
      const int e = tid + q * kMhaMThreads;
      if (e < nrel * DK) evs[e] = evr[q];
    }
    lds_barrier();
    if (i0 + oi < T) {
      const float* rp = red + oi * RS + (tid & 7);
      const float* ep = evs + (tid & 7);
      float* op = g.out + (rowb + i0 + oi) * C + hd * DK + (tid & 7);
#pragma unroll
      for (int d0 = 0; d0 < DK; d0 += 8) {
        float v = rp[d0];
#pragma unroll
        for (int q = 1; q < 4; ++q) v = add_rn(v, rp[q * 32 * RS + d0]);
        if (NR > 0 && w >= 0) {
          float a = 0.f;
#pragma unroll
          for (int r = 0; r < NR; ++r)
            if (r < nrel) a = fmaf(prel[r], ep[r * DK + d0], a);
          v = add_rn(v, a);
        }
        op[d0] = v * ri;
        if (g.out_p) {
          const size_t o = (size_t)(op - g.out) + d0;
          split_f16(v * ri, g.out_p[o], g.out_p[g.n_out + o]);
        }
      }
    }
  };
  if (nrel <= 9) finish(std::integral_constant<int, 9>{});
  else finish(std::integral_constant<int, 31>{});
}


// ---- experiment: the same kernel with NW waves per workgroup (8: twice the waves per CU at the same LDS) ----
template <int DKH, int NW>
__global__ __launch_bounds__(64 * NW, NW / 2) void mha_nw(MhaArgs g) {
  constexpr int DK = 2 * DKH, NDT = (DK + 31) / 32, NQ = DKH / 4;
  constexpr int DK1 = 1, DV = 2, G = 8, NT = 64 * NW, RW = 32 / NW;  // tiles / groups in flight per wave
  constexpr int NEV = (31 * DK + NT - 1) / NT;  // E_v values per thread (window <= 15)
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T = g.T, C = g.C, w = g.window;
  const int ST = T | 1;
  float* S = sm;
  float* R = S + kMhaMRows * ST;
  const int nrel = w >= 0 ? 2 * w + 1 : 0;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l32 = lane & 31, half = lane >> 5;
  const int b = blockIdx.z, hd = blockIdx.y, i0 = blockIdx.x * kMhaMRows;
  const size_t rowb = (size_t)b * T;
  const float* base = g.qkv + rowb * 3 * C + hd * DK;
  typedef __attribute__((address_space(1))) const f32x4 gf32x4;
  typedef __attribute__((address_space(1))) const float gf32;
  gf32* maskg = (gf32*)(g.mask + rowb);
  const int ntile = (T + 31) / 32;

  f32x4 qa[NQ];
  {
    const int i = i0 + l32 < T ? i0 + l32 : T - 1;
    gf32x4* src = (gf32x4*)(base + (size_t)i * 3 * C + half * DKH);
#pragma unroll
    for (int j = 0; j < NQ; ++j) qa[j] = src[j];
  }
  f32x4 e4[NQ];
  if (w >= 0 && wave == 0) {
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      e4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (l32 < nrel) e4[j] = *(gf32x4*)(g.ek + (size_t)l32 * DK + half * DKH + 4 * j);
    }
  }
  float mrow[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * half;
    mrow[r] = i < T ? maskg[i] : 0.f;
  }
  f32x4 kb[DK1][NQ];
  float mj[DK1];
  auto load_k = [&](int kt, int u) {
    const int j = kt * 32 + l32, jc = j < T ? j : T - 1;
    gf32x4* src = (gf32x4*)(base + (size_t)jc * 3 * C + C + half * DKH);
#pragma unroll
    for (int q = 0; q < NQ; ++q) kb[u][q] = src[q];
    mj[u] = maskg[jc];
  };
#pragma unroll
  for (int u = 0; u < DK1; ++u) load_k(wave + NW * u, u);  // past the last tile: the clamped row again (no branch, so the
                                                           // compiler can count the loads in flight exactly)
  float evr[NEV];
#pragma unroll
  for (int q = 0; q < NEV; ++q) {
    const int e = tid + q * NT;
    evr[q] = (w >= 0 && e < nrel * DK) ? ((gf32*)g.ev)[e] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < NQ; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) qa[j][e] = div_rn(qa[j][e], g.qscale);
  if (w >= 0 && wave == 0) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NQ; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[j][e], e4[j][e], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) R[((r & 3) + 8 * (r >> 2) + 4 * half) * 33 + l32] = acc[r];
  }
  lds_barrier();
  // ---- pass 1 ----
  auto score_tile = [&](int kt, int u) {
    const int j = kt * 32 + l32;
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[q][e], kb[u][q][e], acc, 0, 0, 0);
    if (j < T) {
      const float mjv = mj[u];
      float* sj = S + j;
      // does any (query, key) pair of this tile lie inside the relative window?  (the same answer in every lane)
      const bool near = w >= 0 && kt * 32 + 31 + w >= i0 && kt * 32 <= i0 + 31 + w;
      if (near) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
          float s = acc[r];
          const int rr = j - (i0 + row) + w;
          if (rr >= 0 && rr <= 2 * w) s = add_rn(s, R[row * 33 + rr]);
          if (mrow[r] * mjv == 0.f) s = -1e4f;
          sj[row * ST] = s;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
          sj[row * ST] = mrow[r] * mjv == 0.f ? -1e4f : acc[r];
        }
      }
    }
  };
  for (int kt0 = wave; kt0 < ntile; kt0 += NW * DK1) {
#pragma unroll
    for (int u = 0; u < DK1; ++u) {
      const int kt = kt0 + NW * u;
      if (kt < ntile) score_tile(kt, u);
      load_k(kt + NW * DK1, u);
    }
  }
  lds_barrier();
  // ---- softmax numerators, the wave's 8 rows side by side ----
  float* rinv = S + kMhaMRows * ST + (w >= 0 ? 32 * 33 : 0);
  {
    float* row0 = S + (wave * RW) * ST;
    float mx[RW], sum[RW];
#pragma unroll
    for (int q = 0; q < RW; ++q) mx[q] = -3.4e38f, sum[q] = 0.f;
    for (int j = lane; j < T; j += 64)
#pragma unroll
      for (int q = 0; q < RW; ++q) mx[q] = fmaxf(mx[q], row0[q * ST + j]);
#pragma unroll
    for (int q = 0; q < RW; ++q) mx[q] = wave_max(mx[q]);
    for (int j = lane; j < T; j += 64)
#pragma unroll
      for (int q = 0; q < RW; ++q) {
        const float e = __expf(row0[q * ST + j] - mx[q]);
        row0[q * ST + j] = e;
        sum[q] += e;
      }
#pragma unroll
    for (int q = 0; q < RW; ++q) sum[q] = wave_sum(sum[q]);
    if (lane == 0)
#pragma unroll
      for (int q = 0; q < RW; ++q) rinv[wave * RW + q] = 1.0f / sum[q];
  }
  lds_barrier();
  // ---- pass 2 ----
  f32x16 acc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) acc[dt] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  {
    gf32* vb = (gf32*)(base + 2 * C);
    const float* prow = S + l32 * ST;
    const int npair = (T + 1) / 2;
    struct VN { float v[NDT]; };  // DK % NDT == 0 for the built sizes: a lane's columns are all inside or all outside
    static_assert(DK % NDT == 0, "");
    float pv[DV][G], vv[DV][G][NDT];
    // one key pair of a group: its P value (LDS) and the lane's NDT adjacent V values (one global load)
    // lanes whose columns lie past DK read column 0 instead: their accumulators are never stored
    gf32* vlane = vb + (NDT * l32 < DK ? NDT * l32 : 0) + (size_t)half * 3 * C;
    const float* plane = prow + half;
    auto load_1 = [&](int p0, int s, int u) {
      if (2 * (p0 + G) <= T) {  // every key of the group exists (the same answer in all lanes): no predicates
        pv[s][u] = plane[2 * (p0 + u)];
        gf32* vk = vlane + (size_t)(2 * (p0 + u)) * 3 * C;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) vv[s][u][dt] = vk[dt];
      } else {
        const int key = 2 * (p0 + u) + half;
        const bool ok = key < T;
        const int kc = ok ? key : T - 1;
        const float pl = prow[kc];
        gf32* vk = vlane + (size_t)(kc - half) * 3 * C;
        pv[s][u] = ok ? pl : 0.f;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const float vl = vk[dt];
          vv[s][u][dt] = ok ? vl : 0.f;
        }
      }
    };
#pragma unroll
    for (int s = 0; s < DV; ++s)
      if (wave * G + s * NW * G < npair) {
#pragma unroll
        for (int u = 0; u < G; ++u) load_1(wave * G + s * NW * G, s, u);
      }
    for (int pb = wave * G; pb < npair; pb += DV * NW * G) {
#pragma unroll
      for (int s = 0; s < DV; ++s) {
        const int p0 = pb + s * NW * G;
        if (p0 < npair) {
          const bool more = p0 + DV * NW * G < npair;
          // the slot's next key pair is requested right behind the MFMAs that consumed the old one, so the address
          // arithmetic runs while the matrix pipe is busy instead of after the whole group
#pragma unroll
          for (int u = 0; u < G; ++u) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(pv[s][u], vv[s][u][dt], acc[dt], 0, 0, 0);
            if (more) load_1(p0 + DV * NW * G, s, u);
          }
        }
      }
    }
  }
  // relative values need p[i, i + r - w]: read them (and the row's 1/sum) before S is reused as the reduction buffer.
  // NR: compile-time bound on the window rows (9 covers the reference's window_size = 4)
  auto finish = [&](auto nr_c) {
    constexpr int NR = decltype(nr_c)::value;
    constexpr int TPR = NT / 32; const int oi = tid / TPR;  // output row of this thread (32 rows x 8 threads)
    const float ri = rinv[oi];
    float prel[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int j = i0 + oi + r - w;
      prel[r] = (r < nrel && j >= 0 && j < T) ? S[oi * ST + j] : 0.f;
    }
    lds_barrier();
    float* red = S;  // [4 waves][32][DK + 1], then E_v [nrel][DK]
    constexpr int RS = DK + 1;
    float* evs = red + NW * 32 * RS;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const int d = NDT * l32 + dt;
      if (d < DK) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * RS + d] = acc[dt][r];
      }
    }
#pragma unroll
    for (int q = 0; q < NEV; ++q) {
      const int e = tid + q * NT;
      if (e < nrel * DK) evs[e] = evr[q];
    }
    lds_barrier();
    if (i0 + oi < T) {
      const float* rp = red + oi * RS + (tid % TPR);
      const float* ep = evs + (tid % TPR);
      float* op = g.out + (rowb + i0 + oi) * C + hd * DK + (tid % TPR);
#pragma unroll
      for (int d0 = 0; d0 < DK; d0 += TPR) {
        float v = rp[d0];
#pragma unroll
        for (int q = 1; q < NW; ++q) v = add_rn(v, rp[q * 32 * RS + d0]);
        if (NR > 0 && w >= 0) {
          float a = 0.f;
#pragma unroll
          for (int r = 0; r < NR; ++r)
            if (r < nrel) a = fmaf(prel[r], ep[r * DK + d0], a);
          v = add_rn(v, a);
        }
        op[d0] = v * ri;
        if (g.out_p) {
          const size_t o = (size_t)(op - g.out) + d0;
          split_f16(v * ri, g.out_p[o], g.out_p[g.n_out + o]);
        }
      }
    }
  };
  if (nrel <= 9) finish(std::integral_constant<int, 9>{});
  else finish(std::integral_constant<int, 31>{});
}


template <int DKH, int ABL>
static void run(const char* name, MhaArgs a, int B, int heads, size_t lds) {
  hipFuncSetAttribute((const void*)mha_mfma_kernel<DKH, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const dim3 grid((a.T + 31) / 32, heads, B), block(256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((mha_mfma_kernel<DKH, ABL>), grid, block, lds, 0, a);
  hipEventRecord(e0, 0);
  const int iters = 20;
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((mha_mfma_kernel<DKH, ABL>), grid, block, lds, 0, a);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-50s grid %5d  LDS %3zu KiB : %8.2f us per launch  (%s)\n", name, grid.x * grid.y * grid.z, lds / 1024, ms * 1e3f / iters,
         hipGetErrorString(hipGetLastError()));
}

template <int DKH>
static void shape(int B, int T, int heads, int window) {
  const int dk = 2 * DKH, C = heads * dk;
  const size_t M = (size_t)B * T;
  std::vector<float> h(M * 3 * C), m(M, 1.f), e((size_t)33 * dk);
  unsigned s = 12345u;
  for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
  for (auto& x : e) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xffff) / 65536.f * 0.1f; }
  float *qkv, *mask, *ek, *ev, *out;
  hipMalloc(&qkv, h.size() * 4); hipMalloc(&mask, M * 4); hipMalloc(&ek, e.size() * 4); hipMalloc(&ev, e.size() * 4); hipMalloc(&out, M * C * 4);
  hipMemcpy(qkv, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(mask, m.data(), M * 4, hipMemcpyHostToDevice);
  hipMemcpy(ek, e.data(), e.size() * 4, hipMemcpyHostToDevice); hipMemcpy(ev, e.data(), e.size() * 4, hipMemcpyHostToDevice);
  MhaArgs a;
  a.qkv = qkv; a.mask = mask; a.ek = window >= 0 ? ek : nullptr; a.ev = window >= 0 ? ev : nullptr; a.out = out; a.out_p = nullptr; a.n_out = M * C;
  a.T = T; a.C = C; a.dk = dk; a.window = window; a.qscale = sqrtf((float)dk);
  const size_t rel = (window >= 0 ? 32 * 33 : 0) + 32;
  const size_t sz = (size_t)32 * (T | 1) + rel, red = (size_t)4 * 32 * (dk + 1);
  const size_t lds = (sz > red ? sz : red) * 4;
  printf("B %d  T %d  heads %d  dk %d  window %d\n", B, T, heads, dk, window);
  run<DKH, 0>("full", a, B, heads, lds);
  run<DKH, 1>("without the P V loop", a, B, heads, lds);
  run<DKH, 2>("... and without the softmax", a, B, heads, lds);
  run<DKH, 3>("... and without the score tiles (fixed costs)", a, B, heads, lds);
  {
    std::vector<float> o1(M * C), o2(M * C);
    hipLaunchKernelGGL((mha_mfma_kernel<DKH, 0>), dim3((T + 31) / 32, heads, B), dim3(256), lds, 0, a);
    hipMemcpy(o1.data(), out, o1.size() * 4, hipMemcpyDeviceToHost);
    hipMemset(out, 0, o1.size() * 4);
    const size_t lds2 = std::max(lds, ((size_t)4 * 32 * (dk + 1) + (size_t)(window >= 0 ? 2 * window + 1 : 0) * dk) * 4);
    hipFuncSetAttribute((const void*)mha_v2<DKH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    const dim3 grid((T + 31) / 32, heads, B), block(256);
    hipLaunchKernelGGL((mha_v2<DKH>), grid, block, lds2, 0, a);
    hipMemcpy(o2.data(), out, o2.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < o1.size(); ++i) bad += memcmp(&o1[i], &o2[i], 4) != 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((mha_v2<DKH>), grid, block, lds2, 0, a);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-50s grid %5d  LDS %3zu KiB : %8.2f us per launch  (%s; %zu of %zu outputs differ from the shipped kernel's)\n", "candidate", grid.x * grid.y * grid.z,
           lds2 / 1024, ms * 1e3f / 20, hipGetErrorString(hipGetLastError()), bad, o1.size());
    {
      int nb1 = 0, nb2 = 0;
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb1, (const void*)mha_mfma_kernel<DKH, 0>, 256, lds);
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb2, (const void*)mha_v2<DKH, 0>, 256, lds2);
      printf("workgroups per CU by the occupancy query: shipped %d, candidate %d\n", nb1, nb2);
    }
    {
      const size_t lds8 = std::max(lds, ((size_t)8 * 32 * (dk + 1) + (size_t)(window >= 0 ? 2 * window + 1 : 0) * dk) * 4);
      hipFuncSetAttribute((const void*)mha_nw<DKH, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8);
      hipMemset(out, 0, o1.size() * 4);
      hipLaunchKernelGGL((mha_nw<DKH, 8>), grid, dim3(512), lds8, 0, a);
      hipMemcpy(o2.data(), out, o2.size() * 4, hipMemcpyDeviceToHost);
      double md = 0, mv = 0;
      for (size_t i = 0; i < o1.size(); ++i) { md = std::max(md, (double)fabsf(o1[i] - o2[i])); mv = std::max(mv, (double)fabsf(o1[i])); }
      hipEventRecord(e0, 0);
      for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((mha_nw<DKH, 8>), grid, dim3(512), lds8, 0, a);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      int nb = 0;
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)mha_nw<DKH, 8>, 512, lds8);
      printf("%-50s %8.2f us  (%s; max |diff| %.3g of max |out| %.3g; %d workgroups per CU)\n", "8 waves per workgroup", ms * 1e3f / 20,
             hipGetErrorString(hipGetLastError()), md, mv, nb);
    }
    auto abl = [&](const char* name, auto kern) {
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
      hipLaunchKernelGGL(kern, grid, block, lds2, 0, a);
      hipEventRecord(e0, 0);
      for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, grid, block, lds2, 0, a);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      printf("%-50s %8.2f us\n", name, ms * 1e3f / 20);
    };
    {
      const size_t big = lds2 + 8192 > 84 * 1024 ? lds2 + 8192 : 84 * 1024;  // more than half a CU's LDS: one workgroup per CU
      hipFuncSetAttribute((const void*)mha_v2<DKH, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)big);
      hipLaunchKernelGGL((mha_v2<DKH, 0>), grid, block, big, 0, a);
      hipEventRecord(e0, 0);
      for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((mha_v2<DKH, 0>), grid, block, big, 0, a);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      printf("%-50s %8.2f us\n", "candidate, one workgroup per CU (LDS padded)", ms * 1e3f / 20);
    }
    abl("candidate without the P V loop", mha_v2<DKH, 1>);
    abl("... and without the softmax", mha_v2<DKH, 2>);
    abl("... and without the score tiles", mha_v2<DKH, 3>);
  }
  hipFree(qkv); hipFree(mask); hipFree(ek); hipFree(ev); hipFree(out);
}

int main() {
  shape<24>(64, 600, 2, 4);   // reverse flow's pre-transformer: C = 96, 2 heads
  shape<24>(64, 576, 2, 4);   // ... a little shorter: 3 KiB less LDS per workgroup
  shape<24>(64, 384, 2, 4);   // ... 3 workgroups per CU by LDS
  shape<48>(64, 120, 2, 4);   // text encoder: C = 192, 2 heads
  hipDeviceSynchronize();
  return 0;
}
