// conv256.hip - the Postnet's wide conv layers (bf16 and exact fp32) as a 256 x 256-tile GEMM (tacotron/modules/modules.py:170-184 MelPostnet:
// Conv1d(k, pad=(k-1)/2, no bias) -> BatchNorm1d(eval) -> isru, the hidden -> hidden layers).
//
// Why a second GEMM schedule: the shared tile (gemm_tile.h, 128 x 128, loader waves + 4 MFMA waves, two workgroups per CU)
// stages 64 KiB per 128 x 128 x 64 products; chip-wide that is 6.3 GB per layer through an L2 -> LDS path that delivers
// 12-16 TB/s, i.e. the layer's whole 0.47 ms (tools/ubench_conv256.hip: this kernel's own DMA-only / MFMA-only ablations).
// A 256 x 256 tile stages half the bytes per product.  Here: ONE workgroup per CU, 8 waves as 2 (rows) x 4 (columns), each
// wave a 128 x 64 block of the tile on v_mfma_f32_32x32x16_bf16 (4 x 2 accumulators = 128 VGPRs).  EVERY wave issues both
// LDS-DMA and MFMAs: two waves per SIMD fill each other's gaps, a wave's DMA instructions sit between its MFMA groups, and the
// fragment reads of a k16 step are issued one step ahead of its MFMAs.  Operand ring: five 32-KiB slots (160 KiB of LDS), a
// slot = one operand's 256 rows x 64 k; the order B(s) A(s) B(s+1) A(s+1) B(s+2) keeps the weights two steps and the
// activations one step ahead (the weights are the lines EVERY workgroup asks its L2 for at the same moment: with the leads the
// other way round the layer took 5 % longer), behind a counted s_waitcnt vmcnt and ONE raw s_barrier per step (gemm_tile.h on why
// not __syncthreads()).
//
// The same tile, ring and schedule in exact fp32 (kF32: 32 fp32 of k per 128-byte row piece, v_mfma_f32_32x32x2_f32, fp32 output
// stored straight from the accumulators) serves the fp32 Postnet: there the matrix pipe binds and the gain is the loop's (0.91 of
// the instruction's rate against ~0.84 for the shared 64 x 64 tile).
//
// Rounds: one workgroup per CU means whole rounds of tiles; what the whole rounds leave is one more launch of SHORT tiles (NMT < 4
// 32-row MFMA blocks per wave), and the library takes the kernel only where that pays (launch_conv256_abl).
//
// Operands: x [M = B * T, Cin] bf16, channel-last, so the im2col row of frame m is the contiguous window of `taps` frames
// around it; a K step is 64 channels of one tap: tile row r reads frame m0 + r + tap - taps/2 of the SAME utterance or zeros
// (buffer loads with an out-of-range offset write zeros to LDS - gemm_tile.h make_rsrc).  w [N, taps * Cin] bf16 (k = tap * Cin
// + c, as the blob holds it).  out [M, N] bf16 = isru(acc * alpha[n] + beta[n]).
//
// LDS image of a slot: [256 rows][64 k] bf16 = 128-byte rows (whole cache lines: 64-byte pieces read at 0.56 of the rate,
// common.h Seg3) of eight 16-byte chunks; chunk c of row r sits at chunk position c ^ ((r >> 1) & 7) (source-side swizzle: an
// LDS-DMA instruction writes the wave's 64 x 16 bytes lane-linear, so the lane that fills position p fetches chunk
// p ^ swizzle): the 16 rows of a ds_read_b128 lane group then cover all 64 banks (MI355X_MICROARCH.md, LDS).
#include "common.h"
#include "kernels.h"

namespace ttsdec {
namespace {

constexpr int kT256 = 256;                     // tile rows = tile columns
constexpr int kRowB = 128;                     // bytes of a row piece per step: whole cache lines (64 bf16 or 32 fp32 of k)
constexpr int kSlot = kT256 * kRowB;           // one operand's bytes per step (32 KiB)
constexpr int kSlots = 5;                      // ring of operand slots: B(s), A(s), B(s + 1), A(s + 1), B(s + 2) = 160 KiB
constexpr int kThreads256 = 512;
constexpr unsigned kOob = 0x7FFFF000u;         // (gemm_tile.h kBufRange: an offset no buffer reaches -> the lane's 16 bytes are zeros)

typedef __attribute__((address_space(3))) void lds_void256;

struct Conv256Args {
  const void* x;
  const void* w;
  const float* alpha;
  const float* beta;
  void* out;
  int M, T, Cin, taps, N;
  int n_row_tiles, n_col_tiles;
  int m_base, rows_t;  // this launch's first frame and the rows of its tiles (256, or fewer for the tiles of a last, partial round)
};

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ABL (tools/ubench_conv256.hip only): 1 = no DMA inside the loop, 2 = no MFMAs, 3 = DMA only, 4 = DMA without the vmcnt waits,
// 5 = the kernel as shipped, launched whatever the tile count
// kF32: exact fp32 operands on v_mfma_f32_32x32x2_f32 (32 k per step, fp32 output) instead of bf16 on v_mfma_f32_32x32x16_bf16
// (64 k per step, bf16 output); same tile, ring and schedule
// NMT: 32-row MFMA blocks per wave (4 = whole 256-row tiles; the tiles of a last partial round have 1 ... 3: launch_conv256_abl)
template <int ABL, bool kF32, int NMT>
__global__ __launch_bounds__(kThreads256, 2) void conv256_kernel(Conv256Args g) {
  constexpr int EB = kF32 ? 4 : 2;          // operand element bytes
  constexpr int kBK = kRowB / EB;            // k per step
  __shared__ __attribute__((aligned(16))) char smem[kSlots * kSlot];
  // XCD-aware order: the column tiles of a row tile run back to back on one XCD (blockIdx % 8 labels the workgroups that share
  // an XCD under round-robin placement - speed only), so its L2 fetches the activation rows once
  const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
  const int ct = j % g.n_col_tiles, rt = (j / g.n_col_tiles) * 8 + xcd;
  if (rt >= g.n_row_tiles) return;
  const int m0 = g.m_base + rt * g.rows_t, n0 = ct * kT256;
  const int m_end = (m0 + g.rows_t < g.M) ? m0 + g.rows_t : g.M;  // one past the tile's last frame
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int K = g.taps * g.Cin, pad = g.taps / 2, steps_per_tap = g.Cin / kBK, nk = g.taps * steps_per_tap;

  // ---- per-column epilogue operands first: they are older than every DMA in the in-order vmcnt queue ----
  float al[2], be[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = n0 + wc * 64 + nt * 32 + (lane & 31);
    al[nt] = g.alpha[n];
    be[nt] = g.beta[n];
  }

  // ---- loader side: this wave's four 8-row groups of each operand (one DMA instruction = 8 rows x 128 bytes) ----
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)g.x, 0, 0x7FFFF000, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)g.w, 0, 0x7FFFF000, 0x00020000);
  int a_m[4], a_t[4];          // frame index of this lane's row and its position inside the utterance
  unsigned a_cb[4], b_off[4];  // byte offset of this lane's source chunk inside a 64-k run; weight row offset
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    a_m[i] = m0 + row;
    a_t[i] = a_m[i] % g.T;
    a_cb[i] = (unsigned)c * 16u;
    b_off[i] = (unsigned)(n0 + row) * (unsigned)K * (unsigned)EB + (unsigned)c * 16u;
  }
  unsigned a_off[4] = {kOob, kOob, kOob, kOob};
  auto set_tap = [&](int tap) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int tt = a_t[i] + tap - pad;
      const bool ok = a_m[i] < m_end && tt >= 0 && tt < g.T;
      a_off[i] = ok ? (unsigned)(a_m[i] + tap - pad) * (unsigned)g.Cin * (unsigned)EB + a_cb[i] : kOob;
    }
  };
  // the activation stream and the weight stream advance on their own (the ring holds B one step further ahead than A)
  int la_step = 0, la_tap = 0, la_in_tap = 0, la_slot = 1;  // next A unit to request and the slot it goes to
  int lb_step = 0, lb_slot = 0;
  set_tap(0);
  bool in_loop = false;
  auto issue_a = [&](int i) {  // (past the last step: zeros, so that the vmcnt arithmetic stays uniform)
    if (ABL == 1 && in_loop) return;
    char* d = smem + la_slot * kSlot + (wave * 4 + i) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void256*)d, 16, (int)(la_step < nk ? a_off[i] : kOob), la_in_tap * kRowB, 0, 0);
  };
  auto next_a = [&]() {
    ++la_step;
    la_slot += 2; if (la_slot >= kSlots) la_slot -= kSlots;
    if (++la_in_tap == steps_per_tap) {
      la_in_tap = 0;
      ++la_tap;
      set_tap(la_tap);
    }
  };
  auto issue_b = [&](int i) {
    if (ABL == 1 && in_loop) return;
    char* d = smem + lb_slot * kSlot + (wave * 4 + i) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void256*)d, 16, (int)(lb_step < nk ? b_off[i] : kOob), lb_step * kRowB, 0, 0);
  };
  auto next_b = [&]() {
    ++lb_step;
    lb_slot += 2; if (lb_slot >= kSlots) lb_slot -= kSlots;
  };

  // ---- MFMA side: fragment addresses (bytes inside a slot) ----
  // A fragment of 32 rows x 16 k: lane -> row lane % 32, k group lane / 32 (8 bf16 = one chunk); chunk = 2 * k16 + group, so the
  // address of k16 step ks is the one of step 0 with bits 5-6 flipped by ks (the swizzle only XORs the chunk index)
  unsigned fa0, fb0;
  {
    const int ra = wr * (NMT * 32) + (lane & 31), rb = wc * 64 + (lane & 31);
    fa0 = (unsigned)ra * 128u + (unsigned)(((lane >> 5) ^ ((ra >> 1) & 7)) * 16);
    fb0 = (unsigned)rb * 128u + (unsigned)(((lane >> 5) ^ ((rb >> 1) & 7)) * 16);
  }
  f32x16 acc[NMT][2];
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;

  // Fragment registers: two sets, for the even and the odd k16 steps of a stage.  The reads of a k16 step are issued one step
  // AHEAD of its MFMAs - a stage's first step right behind the stage's barrier, under the previous stage's last 8 MFMAs - so that
  // the 8 waves' LDS reads run beside the matrix pipe instead of in front of it.
  // A DMA instruction holds its wave's issue slot for ~100 cycles (gemm_tile.h: why the shared tile has loader waves); here every
  // wave is both, so its eight DMA instructions per step sit between its MFMA groups - each issues under the matrix pipe time of
  // the four MFMAs in front of it, the SIMD's other wave fills what is left (sched_barrier pins the order).  The activations of the
  // NEXT step go first (they are needed one step from now), then the weights of the step after it.
  // (fp32: a fragment register set is 4 consecutive k per lane group, i.e. MFMA e of a read takes k pair (e, 4 + e) of its 8 - a
  // permutation of k that both operands share)
  f32x4 a0[NMT] = {}, b0[2] = {}, a1[NMT] = {}, b1[2] = {};
  // prologue: B(0), A(0), B(1)
  for (int i = 0; i < 4; ++i) issue_b(i);
  next_b();
  for (int i = 0; i < 4; ++i) issue_a(i);
  next_a();
  for (int i = 0; i < 4; ++i) issue_b(i);
  next_b();
#define TTSDEC_MFMA4(A, B, mt0)                                                                                                      \
  _Pragma("unroll") for (int mt = mt0; mt < mt0 + 2 && mt < NMT; ++mt) _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {               \
    if (ABL == 2 || ABL == 3) asm volatile("" ::"v"(A[mt]), "v"(B[nt]));                                                                        \
    else if constexpr (kF32) {                                                                                                        \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[mt][e], B[nt][e], acc[mt][nt], 0, 0, 0); \
    } else {                                                                                                                          \
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[mt]), __builtin_bit_cast(bf16x8, B[nt]), acc[mt][nt], 0, 0, 0); \
    }                                                                                                                                 \
  }
#define TTSDEC_READS(A, B, ks)                                                                                                       \
  _Pragma("unroll") for (int mt = 0; mt < NMT; ++mt) if (ABL != 3) A[mt] = *reinterpret_cast<const f32x4*>(sa + (fa0 ^ ((ks) << 5)) + mt * 4096); \
  _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) if (ABL != 3) B[nt] = *reinterpret_cast<const f32x4*>(sb + (fb0 ^ ((ks) << 5)) + nt * 4096)
#define TTSDEC_FENCE() __builtin_amdgcn_sched_barrier(0)
  in_loop = true;
  int ra_slot = 1, rb_slot = 0;
  for (int kk = 0; kk < nk; ++kk) {
    if (ABL != 1 && ABL != 4) wait_vm<4>();  // this wave's loads of B(kk), A(kk) have landed (B(kk + 1) may still be in flight)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ... and its reads of step kk - 1's slots are done (issued 8 MFMAs ago)
    __builtin_amdgcn_s_barrier();   // everybody's: A(kk + 1) and B(kk + 2) may go into those slots
    const char* sa = smem + ra_slot * kSlot;
    const char* sb = smem + rb_slot * kSlot;
    ra_slot += 2; if (ra_slot >= kSlots) ra_slot -= kSlots;
    rb_slot += 2; if (rb_slot >= kSlots) rb_slot -= kSlots;
    TTSDEC_READS(a0, b0, 0);
    TTSDEC_FENCE();
    TTSDEC_MFMA4(a1, b1, 0);  // the previous step's last k16 (zeros at kk = 0)
    TTSDEC_FENCE();
    issue_a(0); issue_a(1);
    TTSDEC_FENCE();
    TTSDEC_MFMA4(a1, b1, 2);
    TTSDEC_FENCE();
    issue_a(2); issue_a(3);
    next_a();
    TTSDEC_READS(a1, b1, 1);
    TTSDEC_FENCE();
    TTSDEC_MFMA4(a0, b0, 0);
    TTSDEC_FENCE();
    issue_b(0);
    TTSDEC_FENCE();
    TTSDEC_MFMA4(a0, b0, 2);
    TTSDEC_FENCE();
    issue_b(1);
    TTSDEC_READS(a0, b0, 2);
    TTSDEC_FENCE();
    TTSDEC_MFMA4(a1, b1, 0);
    TTSDEC_FENCE();
    issue_b(2);
    TTSDEC_FENCE();
    TTSDEC_MFMA4(a1, b1, 2);
    TTSDEC_FENCE();
    issue_b(3);
    next_b();
    TTSDEC_READS(a1, b1, 3);
    TTSDEC_FENCE();
    TTSDEC_MFMA4(a0, b0, 0);
    TTSDEC_FENCE();
    TTSDEC_MFMA4(a0, b0, 2);
  }
  TTSDEC_MFMA4(a1, b1, 0);
  TTSDEC_MFMA4(a1, b1, 2);
  wait_vm<0>();                  // the trailing zero loads
  __builtin_amdgcn_s_barrier();  // nobody reads a stage any more: the ring becomes the output staging area

  if constexpr (kF32) {
    // ---- epilogue, fp32: BN (folded) + isru, stored straight from the accumulators (a store instruction = two rows x 128 bytes) ----
    float* out = static_cast<float*>(g.out);
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + (wr * NMT + mt) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          const int n = n0 + wc * 64 + nt * 32 + (lane & 31);
          const float v = isru_fast(add_rn(mul_rn(acc[mt][nt][e], al[nt]), be[nt]));  // (as decode_kernels.hip EPI_BN_ISRU)
          if (m < m_end) out[(size_t)m * g.N + n] = v;
        }
  } else {
  // ---- epilogue: BN (folded) + isru, to bf16, through this wave's 16 KiB of LDS so that the stores are whole 16-byte pieces ----
  char* ow = smem + wave * 16384;  // [NMT * 32 rows][64 columns] bf16
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const int col = nt * 32 + (lane & 31);
        // modules.py:181 isru(BatchNorm1d(conv(x))), eval-mode BN as x * alpha + beta (as decode_kernels.hip EPI_BN_ISRU)
        const float v = isru_fast(add_rn(mul_rn(acc[mt][nt][e], al[nt]), be[nt]));
        *reinterpret_cast<bf16*>(ow + row * 128 + col * 2) = (bf16)v;
      }
  // (each wave reads back only what it wrote itself: no barrier, the compiler's lgkmcnt wait orders the LDS accesses)
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int it = 0; it < 4 * NMT; ++it) {
    const int q = it * 64 + lane, row = q >> 3, cc = q & 7;
    const int m = m0 + wr * (NMT * 32) + row;
    const u32x4 v = *reinterpret_cast<const u32x4*>(ow + row * 128 + cc * 16);
    if (m < m_end) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(g.out) + ((size_t)m * g.N + n0 + wc * 64) * 2 + cc * 16) = v;
  }
  }
}

#undef TTSDEC_MFMA4
#undef TTSDEC_READS
#undef TTSDEC_FENCE

}  // namespace

// one launch: `n_rt` row tiles of `rows_t` rows (NMT 32-row blocks per wave) from frame `m_base`
template <int ABL, bool kF32, int NMT>
static void launch_conv256_tiles(Conv256Args g, int m_base, int rows_t, int n_rt, hipStream_t st) {
  g.m_base = m_base; g.rows_t = rows_t; g.n_row_tiles = n_rt;
  const int groups = (n_rt + 7) / 8;
  hipLaunchKernelGGL((conv256_kernel<ABL, kF32, NMT>), dim3((unsigned)(groups * 8 * g.n_col_tiles)), dim3(kThreads256), 0, st, g);
}

// false: the shape is not this kernel's (the caller keeps the shared GEMM)
template <int ABL, bool kF32>
static bool launch_conv256_abl(const void* x, const void* w, const float* alpha, const float* beta, void* out, int M, int T, int Cin, int taps, int N,
                               hipStream_t st) {
  constexpr int EB = kF32 ? 4 : 2;
  if (M <= 0 || T <= 0 || ((Cin * EB) % kRowB) || (N % kT256) || !(taps & 1) || taps > 15) return false;
  if ((size_t)M * Cin * EB >= kOob || (size_t)N * taps * Cin * EB >= kOob) return false;  // 32-bit buffer offsets
  Conv256Args g;
  g.x = x; g.w = w; g.alpha = alpha; g.beta = beta; g.out = out;
  g.M = M; g.T = T; g.Cin = Cin; g.taps = taps; g.N = N;
  g.n_col_tiles = N / kT256;
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    return n;
  }();
  if (cus <= 0) return false;
  // One workgroup per CU and whole ROUNDS of tiles.  The row tiles of the whole rounds are one launch; what is left is cut into one
  // more round of SHORT tiles - h rows each, ceil(h / 64) of the four 32-row MFMA blocks per wave, their own instantiation - so
  // that 4.7 rounds cost 4.75, not 5 (the same kernel with the block count as a run-time predicate lost what this wins).
  const int n_rt = (M + kT256 - 1) / kT256, slots = cus / g.n_col_tiles > 0 ? cus / g.n_col_tiles : 1;
  const int full = (n_rt / slots) * slots;        // row tiles of the whole rounds
  const int rem = M - full * kT256;               // rows left for the last round (<= 0: none)
  int h = 0, nmt = 0, n_short = 0;
  if (rem > 0) {
    h = (rem + slots - 1) / slots;
    h = (h + 7) & ~7;                             // (whole 8-row DMA groups)
    nmt = (h + 63) >> 6;
    n_short = (rem + h - 1) / h;
  }
  // The schedule pays where its rounds are full: its advantage over the shared tile is 9-12 % per layer; below ~0.86 of the ideal
  // the shared tile - hundreds of small tiles, no rounds to speak of - is the faster one (one utterance: 6 tiles here, 0.6 ms a
  // layer against 0.04).
  {
    const double ideal = (double)M / kT256 / slots, cost = (double)(full / slots) + (rem > 0 ? nmt / 4.0 : 0.0);
    const char* force = getenv("TTSDEC_CONV256_FORCE");  // (tests: the kernel's edge cases at shapes a CPU oracle finishes in seconds)
    if (ABL == 0 && !(force && force[0] == '1') && ideal < 0.86 * cost) return false;
  }
  if (full > 0) launch_conv256_tiles<ABL, kF32, 4>(g, 0, kT256, full, st);
  if (rem > 0) {
    const int mb = full * kT256;
    if (nmt >= 4) launch_conv256_tiles<ABL, kF32, 4>(g, mb, h, n_short, st);
    else if (nmt == 3) launch_conv256_tiles<ABL, kF32, 3>(g, mb, h, n_short, st);
    else if (nmt == 2) launch_conv256_tiles<ABL, kF32, 2>(g, mb, h, n_short, st);
    else launch_conv256_tiles<ABL, kF32, 1>(g, mb, h, n_short, st);
  }
  return true;
}
bool launch_conv256_bf16(const void* x, const void* w, const float* alpha, const float* beta, void* out, int M, int T, int Cin, int taps, int N,
                         hipStream_t st) {
  return launch_conv256_abl<0, false>(x, w, alpha, beta, out, M, T, Cin, taps, N, st);
}
bool launch_conv256_f32(const float* x, const float* w, const float* alpha, const float* beta, float* out, int M, int T, int Cin, int taps, int N,
                        hipStream_t st) {
  return launch_conv256_abl<0, true>(x, w, alpha, beta, out, M, T, Cin, taps, N, st);
}

}  // namespace ttsdec
