// What does an out-of-range lane of `buffer_load_dwordx4 ... lds` leave in LDS, and is the scalar offset part of the range check?
// (round 4: the GEMM core's loaders moved from global_load_lds to buffer loads with a scalar K offset - gemm_tile.h)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const float* src, float* out, unsigned num_records, unsigned oob_voff, int soff) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 4];
  const int lane = threadIdx.x;
  for (int i = 0; i < 4; ++i) lds[lane * 4 + i] = -7.0f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)num_records, 0x00020000);
  const unsigned voff = (lane & 1) ? oob_voff : (unsigned)lane * 16u;  // odd lanes out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds, 16, (int)voff, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = lds[lane * 4 + i];
}
int main() {
  float *src, *out;
  hipMalloc(&src, 1 << 16); hipMalloc(&out, 1024);
  std::vector<float> h((1 << 16) / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
  hipMemcpy(src, h.data(), 1 << 16, hipMemcpyHostToDevice);
  struct { unsigned nr, oob; int soff; const char* what; } cases[] = {
      {1u << 16, 0x7FFFF000u, 0, "records 64 KiB, odd lanes voffset 0x7FFFF000, soffset 0"},
      {1u << 16, 0x7FFFF000u, 4096, "the same, soffset 4096 (even lanes should read floats 1024 + 4*lane ..)"},
      {2048u, 0x7FFFF000u, 4096, "records 2048 B, soffset 4096: in range only if the scalar offset is NOT range-checked"},
      {0x40000000u, 0x7FFFF000u, 0, "records 1 GiB, odd lanes 0x7FFFF000"},
  };
  for (auto& c : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out, c.nr, c.oob, c.soff);
    std::vector<float> o(256);
    hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost);
    printf("%s\n  lane0: %g %g %g %g | lane1 (oob): %g %g %g %g | lane2: %g %g %g %g | lane3 (oob): %g %g | lane 62: %g\n", c.what, o[0], o[1], o[2], o[3], o[4],
           o[5], o[6], o[7], o[8], o[9], o[10], o[11], o[12], o[13], o[62 * 4]);
  }
  return 0;
}
