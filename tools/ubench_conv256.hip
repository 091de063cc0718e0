// ubench_conv256.hip - csrc/conv256.hip alone: correctness against a naive fp32-accumulating kernel on the same bf16 operands,
// then its time on the Postnet's shape.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I torch-tts_amd/csrc -o /tmp/ub256 tools/ubench_conv256.hip && /tmp/ub256
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../torch-tts_amd/csrc/conv256.hip"

using namespace ttsdec;

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));            \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__global__ void fill_bf16(bf16* p, size_t n, unsigned seed, float scale) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned h = (unsigned)i * 2654435761u ^ seed;
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  p[i] = (bf16)(((float)(h & 0xffff) / 32768.0f - 1.0f) * scale);
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float lo, float hi) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned h = (unsigned)i * 2654435761u ^ seed;
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  p[i] = lo + (hi - lo) * (float)(h & 0xffff) / 65536.0f;
}
// one thread per output, rows sampled: out_ref[m_idx, n]
__global__ void ref_kernel(const bf16* x, const bf16* w, const float* alpha, const float* beta, const int* rows, int n_rows, float* ref, int M, int T,
                           int Cin, int taps, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows * N) return;
  const int m = rows[i / N], n = i % N, t = m % T, pad = taps / 2;
  float acc = 0.f;
  for (int j = 0; j < taps; ++j) {
    const int tt = t + j - pad;
    if (tt < 0 || tt >= T) continue;
    const bf16* xr = x + (size_t)(m + j - pad) * Cin;
    const bf16* wr = w + (size_t)n * taps * Cin + (size_t)j * Cin;
    for (int c = 0; c < Cin; ++c) acc = fmaf((float)xr[c], (float)wr[c], acc);
  }
  const float v = acc * alpha[n] + beta[n];
  ref[i] = v / sqrtf(1.0f + v * v);
}

static int run(int B, int T, int Cin, int taps, int N, int reps) {
  const int M = B * T;
  const size_t nx = (size_t)M * Cin, nw = (size_t)N * taps * Cin, no = (size_t)M * N;
  bf16 *x, *w, *out;
  float *alpha, *beta;
  CK(hipMalloc(&x, nx * 2)); CK(hipMalloc(&w, nw * 2)); CK(hipMalloc(&out, no * 2));
  CK(hipMalloc(&alpha, N * 4)); CK(hipMalloc(&beta, N * 4));
  fill_bf16<<<(unsigned)((nx + 255) / 256), 256>>>(x, nx, 1u, 1.0f);
  fill_bf16<<<(unsigned)((nw + 255) / 256), 256>>>(w, nw, 2u, 0.05f);
  fill_f32<<<(N + 255) / 256, 256>>>(alpha, N, 3u, 0.5f, 1.5f);
  fill_f32<<<(N + 255) / 256, 256>>>(beta, N, 4u, -0.3f, 0.3f);
  CK(hipMemset(out, 0xff, no * 2));
  CK(hipDeviceSynchronize());
  if (!launch_conv256_abl<5, false>(x, w, alpha, beta, out, M, T, Cin, taps, N, nullptr)) { printf("shape refused\n"); return 1; }
  CK(hipDeviceSynchronize());
  // rows to check: utterance edges, tile edges, a spread of others
  std::vector<int> rows;
  for (int r : {0, 1, 2, 3, T - 3, T - 2, T - 1, T, T + 1, 255, 256, 257, M - 1, M - 2, M - 3, M - T, M - T - 1, 131071, 131072, 131073, 131072 + 95, 131072 + 96, 131072 + 175, 131072 + 176, 131072 + 177, 131072 + 351, 131072 + 352}) if (r >= 0 && r < M) rows.push_back(r);
  for (int k = 0; k < 200; ++k) rows.push_back((int)(((long long)k * 7919 * 131 + 17) % M));
  int* drows; float* dref;
  CK(hipMalloc(&drows, rows.size() * 4)); CK(hipMalloc(&dref, rows.size() * N * 4));
  CK(hipMemcpy(drows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
  ref_kernel<<<(unsigned)((rows.size() * N + 255) / 256), 256>>>(x, w, alpha, beta, drows, (int)rows.size(), dref, M, T, Cin, taps, N);
  CK(hipDeviceSynchronize());
  std::vector<float> ref(rows.size() * N);
  std::vector<unsigned short> got(no);
  CK(hipMemcpy(ref.data(), dref, ref.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(got.data(), out, no * 2, hipMemcpyDeviceToHost));
  double max_err = 0;
  long bad = 0;
  for (size_t i = 0; i < rows.size(); ++i)
    for (int n = 0; n < N; ++n) {
      const unsigned u = (unsigned)got[(size_t)rows[i] * N + n] << 16;
      float v;
      std::memcpy(&v, &u, 4);
      const double e = fabs((double)v - (double)ref[i * N + n]);
      if (!(e <= 0.006)) { if (bad < 5) printf("  row %d col %d got %f ref %f\n", rows[i], n, v, ref[i * N + n]); ++bad; }
      if (e > max_err) max_err = e;
    }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch_conv256_abl<5, false>(x, w, alpha, beta, out, M, T, Cin, taps, N, nullptr);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch_conv256_abl<5, false>(x, w, alpha, beta, out, M, T, Cin, taps, N, nullptr);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double flop = 2.0 * M * (double)taps * Cin * N;
  printf("B %d T %d Cin %d taps %d N %d: %zu rows checked, max |err| %.5f, bad %ld;  %.4f ms  %.1f TFLOP/s (%.3f of 2.5 PFLOP/s)\n", B, T, Cin, taps, N,
         rows.size(), max_err, bad, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500.0);
  if (reps >= 10) {  // ablations (wrong outputs, only the time matters)
    auto timeit = [&](auto fn, const char* what) {
      for (int i = 0; i < 2; ++i) fn();
      CK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) fn();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float t;
      CK(hipEventElapsedTime(&t, e0, e1));
      printf("    %-28s %.4f ms\n", what, t / reps);
    };
    timeit([&] { launch_conv256_abl<1, false>(x, w, alpha, beta, out, M, T, Cin, taps, N, nullptr); }, "no DMA inside the loop");
    timeit([&] { launch_conv256_abl<2, false>(x, w, alpha, beta, out, M, T, Cin, taps, N, nullptr); }, "no MFMAs");
    timeit([&] { launch_conv256_abl<3, false>(x, w, alpha, beta, out, M, T, Cin, taps, N, nullptr); }, "DMA only");
    timeit([&] { launch_conv256_abl<4, false>(x, w, alpha, beta, out, M, T, Cin, taps, N, nullptr); }, "no vmcnt waits in the loop");
  }
  for (void* p : {(void*)x, (void*)w, (void*)out, (void*)alpha, (void*)beta, (void*)drows, (void*)dref}) CK(hipFree(p));
  return bad != 0;
}

__global__ void ref_kernel_f32(const float* x, const float* w, const float* alpha, const float* beta, const int* rows, int n_rows, float* ref, int M,
                               int T, int Cin, int taps, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows * N) return;
  const int m = rows[i / N], n = i % N, t = m % T, pad = taps / 2;
  double acc = 0.0;
  for (int j = 0; j < taps; ++j) {
    const int tt = t + j - pad;
    if (tt < 0 || tt >= T) continue;
    const float* xr = x + (size_t)(m + j - pad) * Cin;
    const float* wr = w + (size_t)n * taps * Cin + (size_t)j * Cin;
    for (int c = 0; c < Cin; ++c) acc += (double)xr[c] * (double)wr[c];
  }
  const float v = (float)acc * alpha[n] + beta[n];
  ref[i] = v / sqrtf(1.0f + v * v);
}

static int run_f32(int B, int T, int Cin, int taps, int N, int reps) {
  const int M = B * T;
  const size_t nx = (size_t)M * Cin, nw = (size_t)N * taps * Cin, no = (size_t)M * N;
  float *x, *w, *out, *alpha, *beta;
  CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&out, no * 4));
  CK(hipMalloc(&alpha, N * 4)); CK(hipMalloc(&beta, N * 4));
  fill_f32<<<(unsigned)((nx + 255) / 256), 256>>>(x, nx, 1u, -1.0f, 1.0f);
  fill_f32<<<(unsigned)((nw + 255) / 256), 256>>>(w, nw, 2u, -0.05f, 0.05f);
  fill_f32<<<(N + 255) / 256, 256>>>(alpha, N, 3u, 0.5f, 1.5f);
  fill_f32<<<(N + 255) / 256, 256>>>(beta, N, 4u, -0.3f, 0.3f);
  CK(hipMemset(out, 0xff, no * 4));
  CK(hipDeviceSynchronize());
  if (!launch_conv256_abl<5, true>(x, w, alpha, beta, out, M, T, Cin, taps, N, nullptr)) { printf("shape refused\n"); return 1; }
  CK(hipDeviceSynchronize());
  std::vector<int> rows;
  for (int r : {0, 1, 2, 3, T - 3, T - 2, T - 1, T, T + 1, 255, 256, 257, M - 1, M - 2, M - 3, M - T, M - T - 1, 131071, 131072, 131073, 131072 + 95, 131072 + 96, 131072 + 175, 131072 + 176, 131072 + 177, 131072 + 351, 131072 + 352}) if (r >= 0 && r < M) rows.push_back(r);
  for (int k = 0; k < 200; ++k) rows.push_back((int)(((long long)k * 7919 * 131 + 17) % M));
  int* drows; float* dref;
  CK(hipMalloc(&drows, rows.size() * 4)); CK(hipMalloc(&dref, rows.size() * N * 4));
  CK(hipMemcpy(drows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
  ref_kernel_f32<<<(unsigned)((rows.size() * N + 255) / 256), 256>>>(x, w, alpha, beta, drows, (int)rows.size(), dref, M, T, Cin, taps, N);
  CK(hipDeviceSynchronize());
  std::vector<float> ref(rows.size() * N), got(no);
  CK(hipMemcpy(ref.data(), dref, ref.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(got.data(), out, no * 4, hipMemcpyDeviceToHost));
  double max_err = 0;
  long bad = 0;
  for (size_t i = 0; i < rows.size(); ++i)
    for (int n = 0; n < N; ++n) {
      const double e = fabs((double)got[(size_t)rows[i] * N + n] - (double)ref[i * N + n]);
      if (!(e <= 1e-5)) { if (bad < 5) printf("  row %d col %d got %f ref %f\n", rows[i], n, got[(size_t)rows[i] * N + n], ref[i * N + n]); ++bad; }
      if (e > max_err) max_err = e;
    }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) launch_conv256_abl<5, true>(x, w, alpha, beta, out, M, T, Cin, taps, N, nullptr);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch_conv256_abl<5, true>(x, w, alpha, beta, out, M, T, Cin, taps, N, nullptr);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double flop = 2.0 * M * (double)taps * Cin * N;
  printf("fp32 B %d T %d Cin %d taps %d N %d: %zu rows checked, max |err| %.2e, bad %ld;  %.4f ms  %.1f TFLOP/s (%.3f of 157.3)\n", B, T, Cin, taps, N,
         rows.size(), max_err, bad, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
  for (void* p : {(void*)x, (void*)w, (void*)out, (void*)alpha, (void*)beta, (void*)drows, (void*)dref}) CK(hipFree(p));
  return bad != 0;
}

int main() {
  int rc = 0;
  rc |= run(3, 77, 64, 3, 256, 2);       // ragged: one partial row tile, utterances shorter than a tile
  rc |= run(5, 600, 512, 5, 512, 20);    // ragged rows (3000 = 11.7 tiles): every workgroup alone on its CU
  rc |= run(256, 600, 512, 5, 512, 20);  // the Postnet's hidden layers at the bench shape
  rc |= run_f32(3, 77, 32, 3, 256, 2);
  rc |= run_f32(5, 600, 512, 5, 512, 3);
  rc |= run_f32(256, 600, 512, 5, 512, 5);
  return rc;
}
