// Shared device/host helpers for libttsdec (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/ttsdec.h"

namespace ttsdec {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGemmThreads = 512;  // 4 MFMA waves + 4 loader waves
constexpr int kAttnThreads = 512;
constexpr int kStopNever = 0x7fffffff;

// arithmetic modes of the GEMM core (gemm_tile.h)
enum Prec { PREC_F32 = 0, PREC_F16S = 1, PREC_BF16 = 2 };

// Control block at the head of the decoder workspace.  Everything that changes from one
// ttsdec_decode call (or one graph replay) to the next lives here, so the step kernels of
// a captured hipGraph take only a baked slot index: step t = t_cur + slot.
struct Ctrl {
  int stop_t;      // first step at which the batch-global stop rule fired, else kStopNever
  int steps_done;  // total steps produced so far (maintained by finish_kernel)
  int t_cur;       // global step index of slot 0 of the current launch batch / graph replay
  int t_call;      // t_begin of the current ttsdec_decode call (outputs are indexed t - t_call)
  int t_end;       // t_begin + n_steps of the call: slots at or past it do nothing
  int t_stride;    // capacity (steps) of this call's y / s / w
  int check_stop;
  int dropout_mode;
  float stop_thr;
  int teacher_T;
  unsigned long long seed;
  const float* memory;
  const uint8_t* masks;  // this call's [n_steps, 2, B, d_pre]
  const float* teacher;
  const uint8_t* teacher_flags;
  float *y, *s, *w;
  int range_err;  // bit 0: split-fp16 mode, an activation entering a 16-bit GEMM was outside the fp16 range (saturated);
                  // bit 1: a two-role launch gave up waiting for its producer role (results of the call are invalid)
  unsigned int unused_dep[3];  // (the arrival counters of the two-role launches live in the workspace: kernels.h DepCounters)
  // measurement only (TTSDEC_STAMPS=1): per-workgroup wall-clock stamps of the two-role launches, else nullptr
  unsigned long long* stamps;
  // test hooks (include/ttsdec.h TTSDEC_OPT_DEBUG_FLAGS / _SPIN_LIMIT): bit 0 = the frame role does not signal, bit 1 = the
  // attention role does not, bit 2 = the projection head role does not, bit 3 = the attention LSTM's tiles (one-launch step) do
  // not; polls before role_wait gives up
  int debug_flags, spin_limit;
  // measurement only (ttsdec_profile_loop): when set, workgroup 0 of every step kernel stores its entry time at
  // loop_stamps[slot * kLoopStampNodes + position of the launch in the step order] - the launches' start times inside the
  // replayed graph, at the price of one 8-byte store per launch
  unsigned long long* loop_stamps;
  int pad[22];
};
constexpr int kLoopStampNodes = 8;   // >= launches per step
constexpr int kLoopStampSlots = 32;  // >= steps per captured graph
__device__ __forceinline__ void loop_stamp(const Ctrl* c, int slot, int node) {
  if (c != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
    unsigned long long* p = c->loop_stamps;
    if (p != nullptr && slot < kLoopStampSlots) p[slot * kLoopStampNodes + node] = __builtin_amdgcn_s_memrealtime();
  }
}
constexpr int kStampKinds = 5;  // 3, 4: shader-clock readings (s_memtime) of kinds 0 / 1 at K tiles 16 and 32, for the in-kernel clock
// stamps[(kind * 1024 + block) * 8 + k]; kind 0 = frame || lstm_att, 1 = attention || lstm_dec, 2 = the projection role at the head of
// kind 0's launch (k: 2 = entry, 3 = control block and operands arrived, 4 = partial tile reduced, 5 = signalled);
// k: 0 = role << 32 | HW_ID, 1 = XCC_ID, 2 = start, 3 = gate reached, 4 = gate passed, 5 = end (s_memrealtime, 10 ns units)
// `st` = Ctrl::stamps, loaded ONCE by the caller (stamps_of): a load in front of every stamp is a vector-memory wait in the
// middle of the code being measured - and was, in the gate of the two-role launches, a microsecond on their critical path.
typedef __attribute__((address_space(1))) unsigned long long* stamp_ptr;
__device__ __forceinline__ stamp_ptr stamps_of(const Ctrl* c) { return c != nullptr ? (stamp_ptr)c->stamps : (stamp_ptr) nullptr; }
__device__ __forceinline__ void stamp(stamp_ptr st, int kind, int k, unsigned long long v) {
  if (st != nullptr && (threadIdx.x & 63) == 0 && blockIdx.x < 1024) st[((size_t)kind * 1024 + blockIdx.x) * 8 + k] = v;
}
__device__ __forceinline__ unsigned long long now_rt() { return __builtin_amdgcn_s_memrealtime(); }

// what a step kernel needs to know about "now"
struct StepNow {
  int t, t_rel;
  bool live;
};
__device__ __forceinline__ StepNow step_now(const Ctrl* c, int slot) {
  StepNow n;
  n.t = c->t_cur + slot;
  n.t_rel = n.t - c->t_call;
  n.live = n.t < c->t_end && n.t <= c->stop_t;
  return n;
}

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
// The same on the hardware reciprocal square root (v_rsq_f32, ~1 ulp) for the conv epilogues of the Postnet and Encoder2, where
// a thread finishes 32 outputs per tile: relative error ~2e-7 against a 1e-4 bar.  The attention weights keep the exact form.
__device__ __forceinline__ float isru_fast(float x) { return x * __builtin_amdgcn_rsqf(fmaf(x, x, 1.0f)); }
// isru_sigmoid(x) = (1 + isru(x/2)) / 2         tacotron/modules/activations.py:5-6
__device__ __forceinline__ float isru_sigmoid(float x) { return mul_rn(add_rn(1.0f, isru(mul_rn(x, 0.5f))), 0.5f); }

__device__ __forceinline__ float sigmoid_f(float x) { return div_rn(1.0f, add_rn(1.0f, expf(-x))); }
// The LSTM cell's gate functions on the hardware exponential and reciprocal (v_exp_f32, v_rcp_f32, ~1 ulp each): absolute error <= ~2e-7, two orders below
// the 1e-4 bar and below what split-fp16 operands already carry (2^-22 relative).  The library forms (range reduction,
// branches per interval) made the cell update ~350 instructions per element - 1.3 us at the end of each LSTM launch.
__device__ __forceinline__ float exp_hw(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }  // v_exp_f32
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + exp_hw(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + exp_hw(2.0f * x)); }

// ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011), used for the on-device PreNet dropout.  One call yields
// 128 keep bits: keep(step, layer, b, unit) = bit (unit & 31) of word ((unit >> 5) & 3) of
// philox(counter = {unit >> 7, b, step*2+layer, 0}, key = {seed_lo, seed_hi}).  (The 32-bit
// multiplies are quarter rate on CDNA: one call per unit cost more than the PreNet GEMMs.)
// The oracle carries the same function.
// ---------------------------------------------------------------------------
struct Philox4 {
  uint32_t w[4];
};
__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
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
  Philox4 r;
  r.w[0] = c0; r.w[1] = c1; r.w[2] = c2; r.w[3] = c3;
  return r;
}
// the 128 keep bits of units [128*group, 128*group + 128) of row b
__host__ __device__ __forceinline__ Philox4 philox_keep_group(uint64_t seed, uint32_t step, uint32_t layer, uint32_t b,
                                                              uint32_t group) {
  return philox4x32_10(seed, group, b, step * 2u + layer, 0u);
}
__host__ __device__ __forceinline__ uint32_t philox_keep(uint64_t seed, uint32_t step, uint32_t layer, uint32_t b,
                                                         uint32_t unit) {
  const Philox4 r = philox_keep_group(seed, step, layer, b, unit >> 7);
  const uint32_t w = (unit >> 5) & 3u;
  const uint32_t word = w == 0 ? r.w[0] : (w == 1 ? r.w[1] : (w == 2 ? r.w[2] : r.w[3]));
  return (word >> (unit & 31u)) & 1u;
}

// ---------------------------------------------------------------------------
// A K-dimension made of up to three row-major segments (the torch.cat inputs of
// the reference: decoder_cell.py:187,191,192).  kend[] are cumulative ends.
// ---------------------------------------------------------------------------
struct Seg3 {
  const void *p0, *p1, *p2;  // element type (fp32 or fp16 plane) is the kernel's business
  int ld0, ld1, ld2;         // leading dimensions in elements
  int e0, e1, e2;            // cumulative segment ends in the virtual K axis (elements)
  // Chunked layout (fp16 planes only), mpad > 0: the operand is stored as [K / 32][mpad rows][32 elements], i.e.
  // element (row, k) sits at ((k / 32) * mpad + row) * 32 + k % 32 - the 64-byte pieces that a tile takes from
  // consecutive rows are adjacent, so an LDS-DMA instruction (16 rows x 64 B, or 8 rows x 2 chunks) reads whole
  // cache lines back to back.  Measured (profiles/r02_b_ubench_ingest.txt): 0.32 us per 32 KiB tile with both
  // operands chunked against 0.38 row-major, and 0.56 for row-major 64-byte pieces (half lines).  ld is unused.
  // For the LSTM weights (LoaderWLstm) mpad != 0 is just the flag and ld = chunks per 16-unit block of the matrix.
  int mpad;
};
constexpr int kChunkK = 32;      // elements per chunk row
constexpr int kChunkBytes = 64;  // = kChunkK fp16
__host__ __device__ inline size_t chunk_idx(int row, int k, int mpad) { return ((size_t)(k >> 5) * mpad + row) * kChunkK + (k & 31); }

__host__ __device__ inline Seg3 make_seg3(const void* p0, int ld0, int k0, const void* p1, int ld1, int k1,
                                          const void* p2, int ld2, int k2) {
  Seg3 s;
  s.p0 = p0; s.ld0 = ld0; s.e0 = k0;
  s.p1 = p1; s.ld1 = ld1; s.e1 = k0 + k1;
  s.p2 = p2; s.ld2 = ld2; s.e2 = k0 + k1 + k2;
  s.mpad = 0;
  return s;
}
__host__ __device__ inline Seg3 chunked(Seg3 s, int mpad) {
  s.mpad = mpad;
  return s;
}
__host__ __device__ inline Seg3 make_seg2(const void* p0, int ld0, int k0, const void* p1, int ld1, int k1) {
  return make_seg3(p0, ld0, k0, p1, ld1, k1, p1, ld1, 0);
}
__host__ __device__ inline Seg3 make_seg1(const void* p0, int ld0, int k0) {
  return make_seg3(p0, ld0, k0, p0, ld0, 0, p0, ld0, 0);
}

// Global-address-space pointer types.  Pointers that reach a kernel inside a by-value
// struct are generic to the compiler, which then emits flat_load (uncountable in vmcnt,
// so every wait becomes vmcnt(0)); casting to address_space(1) gives global_load.
typedef __attribute__((address_space(1))) const char gbyte;
__device__ __forceinline__ gbyte* as_global(const void* p) { return (gbyte*)p; }
// The same for typed accesses through pointers that were LOADED from memory (the per-call pointers of the control block: y,
// s, w, masks, teacher, memory).  A flat store or load also counts in lgkmcnt, so every later wait for LDS data - the next
// ds_read, a barrier - waits for it to complete as well: the frame kernel's layer-0 epilogue (it reads the Philox bits from
// LDS) sat 2.5-3.9 us behind the flat stores of y / s (time stamps), the attention kernel's final LDS sum behind its w stores.
template <class T>
__device__ __forceinline__ __attribute__((address_space(1))) T* as_g(T* p) {
  return (__attribute__((address_space(1))) T*)p;
}

// 16 zero bytes in device memory: out-of-range tile elements are loaded from here, so
// a loader only ever selects an ADDRESS (every lane always issues its load).
__device__ __attribute__((aligned(16))) float g_zero4[4] = {0.f, 0.f, 0.f, 0.f};
__device__ __forceinline__ gbyte* zero_addr() { return (gbyte*)g_zero4; }

// segment s of a Seg3 with EB-byte elements: pointer to (row, k = 0) and length in elements
template <int EB>
__device__ __forceinline__ gbyte* seg_row_ptr(const Seg3& s, int row, int seg) {
  const void* p = seg == 0 ? s.p0 : (seg == 1 ? s.p1 : s.p2);
  const int ld = seg == 0 ? s.ld0 : (seg == 1 ? s.ld1 : s.ld2);
  if (s.mpad > 0) return as_global(p) + (long)row * kChunkBytes;
  return as_global(p) + (long)row * ld * EB;
}
// byte offset of a tile row's 16-byte column c16, and the per-tile pointer advance (rowb = tile row bytes per plane)
__device__ __forceinline__ long seg_col_off(const Seg3& s, int c16) {
  return s.mpad > 0 ? (long)(c16 >> 2) * s.mpad * kChunkBytes + (c16 & 3) * 16 : (long)c16 * 16;
}
__device__ __forceinline__ long seg_tile_inc(const Seg3& s, int rowb) {
  return s.mpad > 0 ? (long)(rowb / kChunkBytes) * s.mpad * kChunkBytes : (long)rowb;
}
__device__ __forceinline__ int seg_len(const Seg3& s, int seg) {
  return seg == 0 ? s.e0 : (seg == 1 ? s.e1 - s.e0 : s.e2 - s.e1);
}
__device__ __forceinline__ int seg_count(const Seg3& s) { return s.e2 > s.e1 ? 3 : (s.e1 > s.e0 ? 2 : 1); }

// The part [lo, hi) of the virtual K axis of s (EB-byte elements), empty segments squeezed out:
// what one workgroup of a split-K launch contracts over.
__device__ __forceinline__ Seg3 seg_window(const Seg3& s, int lo, int hi, int eb) {
  auto cut = [&](const void* p, int b, int e, const char*& q, int& n) {
    const int a = lo > b ? lo : b, z = hi < e ? hi : e;
    n = z > a ? z - a : 0;
    q = static_cast<const char*>(p) + (s.mpad > 0 ? (size_t)((a - b) / kChunkK) * s.mpad * kChunkBytes : (size_t)(a - b) * eb);
  };
  const char *q0, *q1, *q2;
  int n0, n1, n2, l0 = s.ld0, l1 = s.ld1, l2 = s.ld2;
  cut(s.p0, 0, s.e0, q0, n0);
  cut(s.p1, s.e0, s.e1, q1, n1);
  cut(s.p2, s.e1, s.e2, q2, n2);
  if (n0 == 0) { q0 = q1; n0 = n1; l0 = l1; q1 = q2; n1 = n2; l1 = l2; n2 = 0; }
  if (n0 == 0) { q0 = q1; n0 = n1; l0 = l1; n1 = 0; }
  if (n1 == 0) { q1 = q2; n1 = n2; l1 = l2; n2 = 0; }
  return chunked(make_seg3(q0, l0, n0, q1, l1, n1, q2, l2, n2), s.mpad);
}

// ---------------------------------------------------------------------------
// Split-fp16 representation of an fp32 value: x ~= hi + lo * 2^-11 with hi = fp16(x)
// (forced to 0 when it would be subnormal) and lo = fp16((x - hi) * 2^11).  22 significand
// bits; exact products of such halves accumulate in fp32 on the f16 MFMA:
//   a*b ~= ah*bh + (ah*bl + al*bh) * 2^-11        (dropped al*bl term: 2^-22 relative)
// ---------------------------------------------------------------------------
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr float kSplitScale = 2048.0f;
constexpr float kSplitMax = 65504.0f;  // largest finite fp16: |x| beyond it has no split-fp16 form
// Saturating: |x| > kSplitMax becomes +-kSplitMax (never inf / NaN out of a finite input); NaN stays NaN.
// Callers on the decode path report the saturation through Ctrl::range_err (split_f16_checked).
__device__ __forceinline__ void split_f16(float x, f16& hi, f16& lo) {
  if (fabsf(x) > kSplitMax) x = copysignf(kSplitMax, x);
  f16 h = (f16)x;
  if (fabsf(x) < 6.103515625e-05f) h = (f16)0.0f;
  hi = h;
  lo = (f16)((x - (float)h) * kSplitScale);
}
// The same for loops: the saturation goes into a local flag, reported once (report_range).  A conditional atomic inside an
// unrolled epilogue loop keeps the compiler from overlapping the iterations' LDS reads and writes: 16 elements of the frame
// kernel's layer-0 epilogue took 2.5-3.9 us that way (time stamps), most of the role's critical path.
__device__ __forceinline__ void split_f16_flag(float x, f16& hi, f16& lo, bool& over) {
  over |= fabsf(x) > kSplitMax;
  split_f16(x, hi, lo);
}
// split_f16 of a value known to be >= 0 and not NaN (a ReLU output): min instead of the sign-preserving clamp
__device__ __forceinline__ void split_f16_pos(float x, f16& hi, f16& lo) {
  x = fminf(x, kSplitMax);
  f16 h = (f16)x;
  if (x < 6.103515625e-05f) h = (f16)0.0f;
  hi = h;
  lo = (f16)((x - (float)h) * kSplitScale);
}
__device__ __forceinline__ void report_range(bool over, Ctrl* ctrl) {
  if (over && ctrl != nullptr) atomicOr(&ctrl->range_err, 1);
}
__device__ __forceinline__ void split_f16_checked(float x, f16& hi, f16& lo, Ctrl* ctrl) {
  if (ctrl != nullptr && fabsf(x) > kSplitMax) atomicOr(&ctrl->range_err, 1);
  split_f16(x, hi, lo);
}

// ---------------------------------------------------------------------------
// Hand-off between the two roles of one launch (fused_kernels.hip), after MI355X_MICROARCH.md "Workgroup
// dispatch ... inter-workgroup visibility" / cdna_hip_programming.md Guideline 16.
//   producer workgroup, after its last store_wt() of the handed-off data:  role_signal()  (whole workgroup calls it)
//   consumer wave, before its first load of that data:                 role_wait()
// Workgroups of the producer role have the LOWEST block ids of the launch and never wait for anything, so they
// are resident (or done) before any consumer spins; the spin is bounded all the same - on a timeout the call
// is flagged (Ctrl::range_err bit 1) and the consumer proceeds, so the grid always drains.
// ---------------------------------------------------------------------------
// The handed-off data is stored WRITE-THROUGH (store_wt: relaxed agent-scope stores = global_store ... sc1), so the
// producer needs no release fence: an agent-scope release is a write-back of the whole XCD L2 (buffer_wbl2), and
// 32 attention workgroups per XCD each paying one cost the step 20 us (98.9 vs 76 us of kernel time).
__device__ __forceinline__ void store_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_wt(f16* p, f16 v) {
  __hip_atomic_store(reinterpret_cast<unsigned short*>(p), __builtin_bit_cast(unsigned short, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 8 bytes per lane: the widest store the atomic builtins give, and wide enough - 2-byte write-through stores cost ~12x the
// fabric time per byte of 16-byte ones, 8-byte ones ~2.7x (MI355X_MICROARCH.md, stores of each flavour)
__device__ __forceinline__ void store_wt8(void* p, unsigned long long v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void role_signal(unsigned int* counter) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's (write-through) stores have left
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a producer whose rows span two of the consumers' 32-row blocks (c1 may be nullptr)
__device__ __forceinline__ void role_signal2(unsigned int* c0, unsigned int* c1) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(c0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 64 && c1 != nullptr) __hip_atomic_fetch_add(c1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Polls before a consumer role gives up: x (~0.25 us of s_sleep + the poll's round trip) ~ 1-4 ms - three orders of magnitude
// beyond any producer role's run time (they end within tens of microseconds of the launch's start), and short enough that a
// call that does time out (an over-subscribed or partitioned GPU on which the producers are not resident) costs ONE such wait:
// once the flag is set every later gate of the call returns at once (role_poll), and Decoder.forward repeats the call on the
// one-role-per-launch schedule.  (Round 3 had 2^17 polls and no short-circuit: hundreds of gates x > 30 ms each.)
constexpr int kRoleSpinLimit = 1 << 12;
// Arrival counters of the two-role launches (in the workspace, zeroed by every ttsdec_decode call): one per 32-ROW BLOCK of the
// batch and hand-off kind, each on a 128-byte line of its own.  A consumer waits for the producers of ITS rows only - 8 frame
// workgroups instead of all 32, 12 projection workgroups instead of 96, 64 attention workgroups instead of 256 - so no
// workgroup waits for the slowest producer of the whole chip, and no counter takes more than 64 adds per step (96 adds to
// ONE word took the projection role's signal ~1.5 us; MI355X_MICROARCH.md "fanin").
constexpr int kDepLine = 32;  // unsigned ints per counter line
// (DEP_HATT: the attention LSTM's tiles -> the query role and the decoder LSTM, in the one-launch step - fused_kernels.hip step_kernel)
enum DepKind { DEP_FRAME = 0, DEP_ATTN = 1, DEP_PROJ = 2, DEP_QUERY = 3, DEP_HATT = 4, DEP_KINDS = 5 };
// the poll alone: for a consumer that takes every handed-off byte with sc1 loads (load_wt), or that only wants to know.
// Two counters (c1 may be nullptr): a consumer whose rows span two of the producers' 32-row blocks.
__device__ __forceinline__ void role_poll(const unsigned int* c0, unsigned int target0, Ctrl* ctrl, const unsigned int* c1 = nullptr,
                                          unsigned int target1 = 0, int long_sleep = 0) {
  if (target0 == 0 && (c1 == nullptr || target1 == 0)) return;
  int spins = 0;
  // Has a hand-off of this call already timed out (Ctrl::range_err bit 1)?  Then its arrival targets - cumulative over the
  // call's steps - can never be met: do not wait again.  The load travels with the first poll (one round trip, not two).
  const int gave_up = ctrl != nullptr ? (__hip_atomic_load(&ctrl->range_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2) : 0;
  for (;;) {
    const unsigned int v0 = __hip_atomic_load(c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned int v1 = c1 != nullptr ? __hip_atomic_load(c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target1;
    if (v0 >= target0 && v1 >= target1) break;
    if (gave_up) break;
    __builtin_amdgcn_s_sleep(8);
    for (int i = 0; i < long_sleep; ++i) __builtin_amdgcn_s_sleep(8);  // (waits known to be long: fewer polls beside the tile streams)
    if (++spins > (ctrl != nullptr ? ctrl->spin_limit : kRoleSpinLimit)) {
      if (ctrl != nullptr) atomicOr(&ctrl->range_err, 2);
      break;
    }
  }
}
__device__ __forceinline__ void role_wait(const unsigned int* counter, unsigned int target, Ctrl* ctrl, const unsigned int* c1 = nullptr,
                                          unsigned int target1 = 0) {
  if (target == 0 && (c1 == nullptr || target1 == 0)) return;
  role_poll(counter, target, ctrl, c1, target1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// A handed-off fp32 value read past this CU's vector L1 (global_load_dword sc1): with EVERY load of the handed-off bytes of
// this form, every store of them write-through and drained before the counter add (role_signal), and the loads issued by the
// polling wave after its poll matched and by the other waves after a workgroup barrier behind it, the consumer needs no
// agent-scope acquire - 1.7 us less per hop (MI355X_MICROARCH.md, "Hand-offs measured with sc1 loads in place of the acquire").
__device__ __forceinline__ float load_wt(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long load_wt8(gbyte* p) {  // 8 bytes of the same kind (global_load_dwordx2 sc1)
  return __hip_atomic_load((__attribute__((address_space(1))) const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Workgroup barrier for data exchanged through LDS: waits for this wave's LDS traffic only.  __syncthreads() also drains
// the wave's outstanding GLOBAL loads and stores (s_waitcnt vmcnt(0)) - behind freshly issued stores that is a whole write
// round trip, 1-2 us (time stamps: the frame kernel's barrier after its y / s stores, the attention kernel's after its
// weight stores).  Not for data handed over through global memory.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Sum / maximum over the 64 lanes, the same value returned to every lane.  DPP row operations (register-to-register), not
// __shfl_xor: that is ds_bpermute_b32, an LDS-path instruction with a wait behind each of its six steps (the attention
// kernel's row energies: 16.0 -> 14.2 us per launch at B = 256, 9.8 -> 7.8 at B = 1).
template <class Op>
__device__ __forceinline__ float wave_reduce(float v, Op op) {
  auto dpp = [](float x, auto ctrl, auto row_mask, float fill) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, x), decltype(ctrl)::value,
                                                                 decltype(row_mask)::value, 0xf, false));
  };
  using std::integral_constant;
  v = op(v, dpp(v, integral_constant<int, 0xB1>{}, integral_constant<int, 0xf>{}, v));   // quad_perm [1,0,3,2]
  v = op(v, dpp(v, integral_constant<int, 0x4E>{}, integral_constant<int, 0xf>{}, v));   // quad_perm [2,3,0,1]
  v = op(v, dpp(v, integral_constant<int, 0x141>{}, integral_constant<int, 0xf>{}, v));  // row_half_mirror
  v = op(v, dpp(v, integral_constant<int, 0x140>{}, integral_constant<int, 0xf>{}, v));  // row_mirror: each lane holds its 16-lane row's result
  // (lanes outside the row mask get `fill` = their own value: op(v, v) must leave v unchanged only for max - so the sum adds
  // through a second form below)
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = wave_reduce(v, [](float a, float b) { return a + b; });
  auto dpp0 = [](float x, auto ctrl, auto row_mask) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, decltype(row_mask)::value, 0xf, true));
  };
  using std::integral_constant;
  v += dpp0(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});  // row_bcast:15 into rows 1 and 3 (others add 0)
  v += dpp0(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});  // row_bcast:31 into rows 2 and 3: lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = wave_reduce(v, [](float a, float b) { return fmaxf(a, b); });
  // the four row maxima: lanes 15, 31, 47, 63
  const float m0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 15));
  const float m1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
  const float m2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 47));
  const float m3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
  return fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
}

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

}  // namespace ttsdec
