// Two-role launches of the decode step.
//
// The step is a serial chain  frame -> lstm_att -> query -> attention -> lstm_dec -> proj  in which the
// two LSTMs are bound by what a CU can take in per microsecond and the other four launches by latency:
// 25-30 us per step in which most of the chip idles.  But most of an LSTM's K axis does not depend on the
// launch just before it (tacotron/decoder_cell.py:187,191):
//     lstm_att(t) = [ x_pre(t) | ctx(t-1) | h_att(t-1) ]      only x_pre waits for the frame kernel
//     lstm_dec(t) = [ h_att(t) | ctx(t)   | h_dec(t-1) ]      only ctx waits for the attention kernel
// So each LSTM shares a launch with the latency-bound kernel in front of it:
//     [frame || lstm_att] -> query -> [attention || lstm_dec] -> proj            (4 launches instead of 6)
// Workgroups [0, n_a) of such a launch run role A (the producer), the rest the LSTM, which walks its K axis
// with the dependent segment LAST and, right before the first tile of that segment, waits for role A's
// arrival counter (common.h role_signal / role_poll: every handed-off byte stored write-through and drained, one lane's
// agent-scope counter add; the consumer's sc1 poll, then sc1 loads - MI355X_MICROARCH.md "Hand-offs measured with sc1 loads
// in place of the acquire", row 1).  Role A has the lowest block ids and waits for nothing, so under the
// observed in-order dispatch it is resident or done before any LSTM workgroup spins; nothing DEPENDS on that order: the spin
// is bounded (~1-4 ms, once per call), a timed-out call is flagged and repeated with one role per launch.  Normally nobody
// waits long: the early segments take about as long as role A.
// What makes it pay is co-residency: the LSTM runs on the "lean" GEMM tile (gemm_tile.h: 64-80 KiB of LDS,
// <= 128 VGPRs), so a CU holds one workgroup of each role and the LSTM's tile stream fills the issue slots
// and memory queues the small kernel leaves empty.
// Measured alternatives: the same overlap as two STREAMS (round 1: slower - cross-stream graph
// edges, one 128-KiB workgroup per CU); the early segments as a separate partial-sum launch beside role A plus
// a "late" launch (round 2: 86.4 vs 88.3 us per step - the extra launch costs what the overlap wins).
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <utility>

#include "frame_body.h"
#include "step_bodies.h"

namespace ttsdec {

// The LSTM role's tiles, per arithmetic mode (PREC_F16S / PREC_F32):
//   Lean64x16: 64 rows x 16 units, 16-KiB stages x 5 = 80 KiB (split-fp16: 32 k per stage on chunked planes; exact fp32:
//              32 k per stage, 128-byte rows).  Two workgroups per CU.
//   Lean64x8:  batches of <= 64 utterances.  Split-fp16: 64 rows x 8 units on two MFMA waves (12-KiB stages x 5), so the weight
//              stream is spread over 128 workgroups instead of 64; rows past the batch read nothing (zeros).  Exact fp32
//              (round 4): 32 rows x 8 units, the tile's K split over all FOUR MFMA waves (64-byte slices, 16-KiB stages of
//              64 k x 5) - 2 x 128 = 256 workgroups at 64 utterances, every matrix pipe of the chip busy.  The 64-row fp32 form
//              (two MFMA waves on 128 workgroups: a quarter of the chip's fp32 matrix rate) is why round 3 had to switch the
//              two-role launches off for exact fp32 between 32 and 192 utterances.
//   SmallFat:  batches of <= 32 utterances: both roles together are fewer workgroups than the chip has CUs, so nothing
//              has to share a CU and the LSTM keeps the stand-alone small-batch tile (128 KiB of LDS, three tiles in
//              flight - a batch-1 LSTM is a pure weight stream and lives on bytes in flight; the lean tiles' 48 KiB
//              took 16.4 us for the attention LSTM at B = 1 against 10.9 us).
template <int PREC>
struct LeanTiles;
template <>
struct LeanTiles<PREC_F16S> {
  using Lean64x16 = TileCfg<2, 2, 1, 5, PREC_F16S, 0, 1, 1, 1>;
  using Lean64x8 = TileCfg<2, 1, 1, 5, PREC_F16S, 0, 1, 1, 1>;
  using SmallFat = TileCfg<1, 1, 2, 4, PREC_F16S>;
};
template <>
struct LeanTiles<PREC_F32> {
  using Lean64x16 = TileCfg<2, 2, 1, 5, PREC_F32>;
  using Lean64x8 = TileCfg<1, 1, 4, 5, PREC_F32, 0, 1, 1, 1>;
  using SmallFat = TileCfg<1, 1, 4, 4, PREC_F32>;
};
// exact fp32: the 32-row tile (every matrix pipe busy from 33 utterances up) while its workgroups - 128 per 32 rows - still fit the
// chip beside the producer role's: us per step at levels 0 / 1 / 2 with it, B = 80 96.3 / 79.4 / 65.4, B = 96 84.8 / 74.4 / 66.5, B = 128
// 86.2 / 84.7 / 88.2 - against the 64-row tile's B = 80 96.3 / 86.8 / 90.7, 96 85.5 / 81.3 / 90.7, 112 86.0 / 81.5 / 91.2, 128 84.8 / 79.4 / 88.0
// (gpurun sessions r4k / r4l)
constexpr int kLean8MaxRowsF32 = 96;
constexpr int kLean8MaxRows = 64;  // (measured with 128: B = 128 56.6 against 49.3 us per step, B = 96 53.6 / 48.1 - profiles/r03_w)
constexpr int kSmallFatMaxRows = 32;
static_assert(LeanTiles<PREC_F16S>::Lean64x16::kLdsBytes <= 80 * 1024 && LeanTiles<PREC_F32>::Lean64x16::kLdsBytes <= 80 * 1024 &&
                  LeanTiles<PREC_F16S>::Lean64x8::kLdsBytes <= 80 * 1024 && LeanTiles<PREC_F32>::Lean64x8::kLdsBytes <= 80 * 1024,
              "lean tiles: two workgroups per CU");

template <int A, int B>
constexpr int cmax() { return A > B ? A : B; }

// ---- role A = frame kernel (finish proj(t-1), PreNet), role B = early part of the attention LSTM ----
// WPE: waves per SIMD the register budget must allow (4 = two 512-thread workgroups per CU, 2 = one)
// kHead: the launch starts with the previous step's mel/stop projection as a third role, [proj(t-1) | frame | lstm_att] by
// block id.  Every role waits only for roles with LOWER ids and the grid dispatches in id order, so whatever a workgroup
// waits for is resident or done.  The frame role has slack for it at large batches (it ends at 10.8 us where the LSTM
// reaches its gate at 14.2, B = 256): the projection's 5.8-us launch disappears from the step.
template <int K0H, int PH, class Cfg, int WPE, bool kHead>
__global__ __launch_bounds__(kGemmThreads, WPE) void frame_lstm_kernel(FrameArgs f, LstmArgs l, ProjArgs pj, int n_proj, int n_frame, int frame_cols,
                                                                       int lstm_cols) {
  constexpr int PREC = Cfg::kPrec;
  __shared__ __attribute__((aligned(16))) float smem[cmax<cmax<Cfg::kLdsFloats, FrameLds<K0H, PH, PREC>::kFloats>(), kHead ? kProjLdsFloats : 1>()];
  loop_stamp(f.ctrl, f.slot, f.node);
  int id = blockIdx.x;
  if constexpr (kHead) {
    if (id < n_proj) {
      __builtin_amdgcn_s_setprio(3);
      proj_body<PREC>(pj, smem, id);
      return;
    }
    id -= n_proj;
  }
  // (Measured and dropped, profiles/r03_y: the frame role LAST in the grid, so that its workgroups draw the first-dispatched
  // LSTM workgroups as CU partners, and the frame role at normal wave priority - the 32 LSTM workgroups that share a CU with a
  // frame workgroup stay the launch's tail either way: they reach their gate ~6 us after the others, 2.5 us after x_pre is there.)
  if (id < n_frame) {
    __builtin_amdgcn_s_setprio(3);  // the producer role is the launch's critical path: it wins issue arbitration
    frame_body<K0H, PH, PREC, 6, WPE == 4, kHead>(f, smem, id % frame_cols, id / frame_cols);
  } else {
    const int j = id - n_frame;
    lstm_body<Cfg, true, true>(l, smem, j % lstm_cols, j / lstm_cols);
  }
}

// ---- role A = attention + context, role B = early part of the decoder LSTM ----
// Query role: the attention role's workgroups first compute the attention query GEMM between them (AttnArgs::q_tiles tiles
// of proj_body, workgroup b the tiles b, b + B, ...), each signalling its tile's 32-row block; an attention pass starts once
// the tiles of its utterance's row block are in.  The step loses the query launch (~6.5 us) and gains a hop inside this one,
// on a role that has slack: the decoder LSTM reaches its gate at ~22 us, the attention pass used to end at ~15.
// All attention-role workgroups must be resident together for this (they have the lowest block ids; B <= the chip's two
// workgroups per CU): the host only asks for it then.
template <int NJ, class Cfg, int WPE>
__global__ __launch_bounds__(kGemmThreads, WPE) void attn_lstm_kernel(AttnArgs a, LstmArgs l, ProjArgs pq, int n_attn, int lstm_cols) {
  __shared__ __attribute__((aligned(16))) float smem[cmax<cmax<Cfg::kLdsFloats, attn_lds_floats<NJ>()>(), kProjLdsFloats>()];
  loop_stamp(a.ctrl, a.slot, a.node);
  const int id = blockIdx.x;
  if (id < n_attn) {
    __builtin_amdgcn_s_setprio(3);  // (as above: the decoder LSTM's last segment waits for every one of these)
    if (a.q_tiles > 0) {
      for (int tile = id; tile < a.q_tiles; tile += n_attn) {
        proj_body<Cfg::kPrec>(pq, smem, tile);
        __syncthreads();  // (the reduction tile in LDS is reused by the next tile / the attention pass)
      }
    }
    attn_body<NJ>(a, smem, id);
  } else {
    const int j = id - n_attn;
    lstm_body<Cfg, true, true>(l, smem, j % lstm_cols, j / lstm_cols);
  }
}

// ---- the whole step as ONE launch ----
// [proj(t-1) | frame(t) | lstm_att(t) | query(t) -> attention(t) | lstm_dec(t)] by block id: the two launches above back to back
// inside one grid, so the step loses the boundary between them (the first launch's drain, the release / acquire pair and the
// second's dispatch: ~4.5-5.4 us of the step at B = 256, time stamps) and the decoder LSTM's workgroups take over a CU's slot the
// moment an attention-LSTM workgroup leaves it instead of after the slowest one of the chip.
// What the boundary ordered is ordered by one more hand-off: every attention-LSTM tile stores its split planes write-through
// and signals the DEP_HATT counter(s) of the 32-row block(s) it covers (lstm_body, LstmArgs::sig_cnt); the query tiles of a
// row block (proj_body, ProjArgs::wait_cnt) and the decoder LSTM's h_att segment (a second gate, LstmArgs::dep2_*) wait for
// all of that block's tiles and read them with sc1 loads.
// Progress: every role waits only for roles with lower block ids (the attention workgroups also for each other's query tiles),
// the grid dispatches in id order, and the roles in front of the attention role end without it - so every attention workgroup
// becomes resident (n_attn <= the chip's 512 slots, checked by the host) and everything a workgroup waits for is resident or
// done.  Every poll is bounded regardless (common.h role_poll).
// Liveness: all roles of the launch go by t - 1 <= stop_t (live_lag), because the frame role may lower stop_t to t - 1 while
// they read it: the step after the one at which the stop rule fires is still computed (and never read).
// Small batches (the stand-alone small-batch LSTM tile, one workgroup per CU, WPE = 2): fewer attention workgroups than query
// tiles, so the query tiles get workgroups of their own (n_q of them, between the attention LSTM and the attention role);
// nothing shares a CU there, and the decoder LSTM's h_dec segment streams on the CUs the first half of the step leaves idle.
struct StepGrid {
  int n_proj, n_frame, frame_cols, n_la, la_cols, n_q, n_attn, ld_cols;
  int tune;  // measurement (option merged_tune): bit 1 = wave priorities 2 / 0 for the attention / decoder LSTM roles
};
template <int K0H, int PH, class Cfg, int NJ, int WPE>
__global__ __launch_bounds__(kGemmThreads, WPE) void step_kernel(FrameArgs f, LstmArgs la, ProjArgs pj, AttnArgs a, LstmArgs ld, ProjArgs pq,
                                                                  StepGrid n) {
  constexpr int PREC = Cfg::kPrec;
  __shared__ __attribute__((aligned(16)))
  float smem[cmax<cmax<Cfg::kLdsFloats, FrameLds<K0H, PH, PREC>::kFloats>(), cmax<kProjLdsFloats, attn_lds_floats<NJ>()>()>()];
  loop_stamp(f.ctrl, f.slot, f.node);
  int id = blockIdx.x;
  if (id < n.n_proj) {
    __builtin_amdgcn_s_setprio(3);
    proj_body<PREC>(pj, smem, id);
    return;
  }
  id -= n.n_proj;
  if (id < n.n_frame) {
    __builtin_amdgcn_s_setprio(3);
    frame_body<K0H, PH, PREC, 6, WPE == 4, true>(f, smem, id % n.frame_cols, id / n.frame_cols);
    return;
  }
  id -= n.n_frame;
  if (id < n.n_la) {
    if (n.tune & 2) __builtin_amdgcn_s_setprio(2);
    lstm_body<Cfg, true, true>(la, smem, id % n.la_cols, id / n.la_cols);
    return;
  }
  id -= n.n_la;
  if (id < n.n_q + n.n_attn) {
    __builtin_amdgcn_s_setprio(3);
    const bool is_q = id < n.n_q;  // a query-tile workgroup (small batches), else the attention workgroup of utterance id - n_q
    const int aid = id - n.n_q;
    const stamp_ptr st = stamps_of(a.ctrl);  // measurement only (TTSDEC_STAMPS)
    if (threadIdx.x == 0 && !is_q) stamp(st, 1, 6, now_rt());
    // query tiles: this workgroup's one (query workgroups), none (attention workgroups beside them), or tiles aid, aid + n_attn, ...
    const int t0 = n.n_q > 0 ? (is_q ? id : 0) : aid, t1 = n.n_q > 0 ? (is_q ? id + 1 : 0) : a.q_tiles, ts = n.n_q > 0 ? 1 : n.n_attn;
    for (int tile = t0; tile < t1; tile += ts) {
      proj_body<PREC>(pq, smem, tile);
      __syncthreads();  // (the reduction tile in LDS is reused by the next tile / the attention pass)
    }
    if (is_q) return;
    attn_body<NJ>(a, smem, aid);
    return;
  }
  id -= n.n_q + n.n_attn;
  lstm_body<Cfg, true, true>(ld, smem, id % n.ld_cols, id / n.ld_cols);
}

// the early part on its own (profiling / ablation: what the role costs without a partner)
template <class Cfg>
__global__ __launch_bounds__(kGemmThreads, 4) void lstm_lean_kernel(LstmArgs l) {
  __shared__ __attribute__((aligned(16))) float smem[Cfg::kLdsFloats];
  lstm_body<Cfg, false>(l, smem, blockIdx.x, blockIdx.y);
}

static_assert(kFrameThreads == kGemmThreads && kAttnThreads == kGemmThreads, "roles share one block size");

bool fused_supported(int d_mel, int r, int Ph, int P, int D) {
  // (frame role: 6 projection columns per thread; attention role: NJ <= 2 - both to stay within 128 VGPRs)
  return frame_supported(d_mel, r, Ph, P) && r * d_mel + r <= 96 && D / 4 <= 128;
}

int frame_grid_size(int M, int P) { return ((P + kFrameCols - 1) / kFrameCols) * ((M + kFrameRows - 1) / kFrameRows); }

// which tile the LSTM role runs on, by batch size (see the three configurations above)
enum LeanKind { LEAN_64x16, LEAN_64x8, SMALL_FAT };
template <int PREC>
static int lean_bm(LeanKind kind) {  // batch rows of the kind's tile
  using TL = LeanTiles<PREC>;
  return kind == SMALL_FAT ? TL::SmallFat::BM : (kind == LEAN_64x8 ? TL::Lean64x8::BM : TL::Lean64x16::BM);
}
static int lean8_max_rows(bool f16) {
  if (f16) return kLean8MaxRows;
  static const int f32_max = [] { const char* e = getenv("TTSDEC_LEAN8_F32_MAX"); return e ? atoi(e) : kLean8MaxRowsF32; }();  // (measurement)
  return f32_max;
}
static LeanKind lean_kind(int M, int n_producer, int H, bool f16 = true) {
  if (M <= kSmallFatMaxRows && n_producer + (H + 7) / 8 <= 224) return SMALL_FAT;
  return M <= lean8_max_rows(f16) ? LEAN_64x8 : LEAN_64x16;
}

template <int K0H, int PH, int PREC, bool kHead>
static void launch_frame_lstm_ph(const FrameArgs& f, const LstmArgs& l, const ProjArgs& pj, hipStream_t st) {
  using TL = LeanTiles<PREC>;
  const int fcols = (f.P + kFrameCols - 1) / kFrameCols, frows = (f.M + kFrameRows - 1) / kFrameRows;
  const int n_frame = fcols * frows, n_proj = kHead ? proj_grid_size(pj.M, pj.N, pj.ksplit) : 0;
  const LeanKind kind = lean_kind(l.M, n_proj + n_frame, l.H, PREC == PREC_F16S);
  const int lcols = kind == LEAN_64x16 ? (l.H + 15) / 16 : (l.H + 7) / 8;
  const int bm = lean_bm<PREC>(kind);
  const int lrows = (l.M + bm - 1) / bm;
  dim3 grid(n_proj + n_frame + lcols * lrows), block(kGemmThreads);
  if (kind == SMALL_FAT)
    hipLaunchKernelGGL((frame_lstm_kernel<K0H, PH, typename TL::SmallFat, 2, kHead>), grid, block, 0, st, f, l, pj, n_proj, n_frame, fcols, lcols);
  else if (kind == LEAN_64x8)
    hipLaunchKernelGGL((frame_lstm_kernel<K0H, PH, typename TL::Lean64x8, 4, kHead>), grid, block, 0, st, f, l, pj, n_proj, n_frame, fcols, lcols);
  else
    hipLaunchKernelGGL((frame_lstm_kernel<K0H, PH, typename TL::Lean64x16, 4, kHead>), grid, block, 0, st, f, l, pj, n_proj, n_frame, fcols, lcols);
}
template <bool kHead>
static void launch_frame_lstm_any(const FrameArgs& f, const LstmArgs& l, const ProjArgs& pj, hipStream_t st) {
  if (f.M <= 0) return;
  const bool f16 = l.prec == 1;
  if (f.Ph == 256) { if (f16) launch_frame_lstm_ph<40, 256, PREC_F16S, kHead>(f, l, pj, st); else launch_frame_lstm_ph<40, 256, PREC_F32, kHead>(f, l, pj, st); }
  else { if (f16) launch_frame_lstm_ph<40, 128, PREC_F16S, kHead>(f, l, pj, st); else launch_frame_lstm_ph<40, 128, PREC_F32, kHead>(f, l, pj, st); }
}
void launch_frame_lstm(const FrameArgs& f, const LstmArgs& l, hipStream_t st) {
  ProjArgs none;
  memset(&none, 0, sizeof(none));
  launch_frame_lstm_any<false>(f, l, none, st);
}
void launch_proj_frame_lstm(const ProjArgs& pj, const FrameArgs& f, const LstmArgs& l, hipStream_t st) { launch_frame_lstm_any<true>(f, l, pj, st); }

// Workgroups of a multi-role kernel the device holds AT ONCE, as its runtime says: CUs x hipOccupancyMaxActiveBlocksPerMultiprocessor
// (registers, LDS and the wave limit of THIS kernel on THIS device - a partitioned or smaller part answers for itself).  The
// schedules in which workgroups of one role wait for EACH OTHER (the query role, the one-launch step) are only taken when all
// of them fit that number; they further assume the GPU is not shared with another process's kernels (what the number cannot
// know - the bounded spin and Decoder.forward's fallback cover that case).  0 when the query fails: those schedules stay off.
static int resident_slots(const void* fn) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_pair(dev, fn);
  const auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int cus = 0, per_cu = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kGemmThreads, 0) != hipSuccess) {
    (void)hipGetLastError();
    cus = per_cu = 0;
  }
  return cache[key] = cus * per_cu;
}

typedef void (*attn_lstm_fn)(AttnArgs, LstmArgs, ProjArgs, int, int);
template <int NJ, int PREC>
static attn_lstm_fn attn_lstm_kernel_of(LeanKind kind) {
  using TL = LeanTiles<PREC>;
  if (kind == SMALL_FAT) return attn_lstm_kernel<NJ, typename TL::SmallFat, 2>;
  if (kind == LEAN_64x8) return attn_lstm_kernel<NJ, typename TL::Lean64x8, 4>;
  return attn_lstm_kernel<NJ, typename TL::Lean64x16, 4>;
}
static attn_lstm_fn attn_lstm_kernel_for(int D, bool f16, LeanKind kind) {
  if (D / 4 <= 64) return f16 ? attn_lstm_kernel_of<1, PREC_F16S>(kind) : attn_lstm_kernel_of<1, PREC_F32>(kind);
  return f16 ? attn_lstm_kernel_of<2, PREC_F16S>(kind) : attn_lstm_kernel_of<2, PREC_F32>(kind);
}
int attn_lstm_resident_slots(int B, int H, int D, bool f16) {
  return resident_slots(reinterpret_cast<const void*>(attn_lstm_kernel_for(D, f16, lean_kind(B, B, H, f16))));
}
void launch_attn_lstm(const AttnArgs& a, const LstmArgs& l, const ProjArgs* q, hipStream_t st) {
  if (a.B <= 0) return;
  ProjArgs pq;
  if (q != nullptr) pq = *q;
  else memset(&pq, 0, sizeof(pq));
  const LeanKind kind = lean_kind(l.M, a.B, l.H, l.prec == 1);
  const int lcols = kind == LEAN_64x16 ? (l.H + 15) / 16 : (l.H + 7) / 8;
  const int bm = l.prec == 1 ? lean_bm<PREC_F16S>(kind) : lean_bm<PREC_F32>(kind);
  const int lrows = (l.M + bm - 1) / bm;
  dim3 grid(a.B + lcols * lrows), block(kGemmThreads);
  hipLaunchKernelGGL(attn_lstm_kernel_for(a.D, l.prec == 1, kind), grid, block, 0, st, a, l, pq, a.B, lcols);
}

// ---- one-launch step ----
typedef void (*step_fn)(FrameArgs, LstmArgs, ProjArgs, AttnArgs, LstmArgs, ProjArgs, StepGrid);
struct StepKernel {
  step_fn fn;
  int bm, bu;  // batch rows / hidden units of an LSTM tile
};
template <int PH, int NJ>
static StepKernel step_kernel_of(LeanKind kind) {
  using TL = LeanTiles<PREC_F16S>;
  if (kind == SMALL_FAT) return {step_kernel<40, PH, TL::SmallFat, NJ, 2>, TL::SmallFat::BM, TL::SmallFat::BN / 4};
  if (kind == LEAN_64x8) return {step_kernel<40, PH, TL::Lean64x8, NJ, 4>, TL::Lean64x8::BM, TL::Lean64x8::BN / 4};
  return {step_kernel<40, PH, TL::Lean64x16, NJ, 4>, TL::Lean64x16::BM, TL::Lean64x16::BN / 4};
}
static StepKernel step_kernel_for(int Ph, int D, LeanKind kind) {
  if (Ph == 256) return D / 4 <= 64 ? step_kernel_of<256, 1>(kind) : step_kernel_of<256, 2>(kind);
  return D / 4 <= 64 ? step_kernel_of<128, 1>(kind) : step_kernel_of<128, 2>(kind);
}
static int step_roles_in_front(int B, int P, int n_out, int ksplit) {  // the workgroups in front of the attention LSTM's
  return proj_grid_size(B, n_out, ksplit) + frame_grid_size(B, P);
}
bool step_merged_supported(int B, int Ha, int Hd, int Ph, int P, int D, int n_out, int ksplit) {
  // The attention role's workgroups - one per utterance - wait for each other's query tiles: all of them resident at once,
  // beside workgroups of the roles in front of them that have not ended yet - so at most HALF of what the device holds of this
  // kernel (resident_slots: 2 x 256 on an MI355X).
  if (B < 1) return false;
  const LeanKind kind = lean_kind(B, step_roles_in_front(B, P, n_out, ksplit), Ha > Hd ? Ha : Hd);
  return B <= resident_slots(reinterpret_cast<const void*>(step_kernel_for(Ph, D, kind).fn)) / 2;
}
void launch_step_merged(const ProjArgs& pj, const FrameArgs& f, const LstmArgs& la, const ProjArgs& pq, const AttnArgs& a, const LstmArgs& ld,
                        hipStream_t st) {
  if (f.M <= 0 || la.prec != 1) return;  // (split-fp16 only: the host never asks for it otherwise)
  const LeanKind kind = lean_kind(la.M, step_roles_in_front(f.M, f.P, pj.N, pj.ksplit), la.H > ld.H ? la.H : ld.H);
  const StepKernel k = step_kernel_for(f.Ph, a.D, kind);
  StepGrid n;
  n.tune = f.dbg >> 8;
  n.frame_cols = (f.P + kFrameCols - 1) / kFrameCols;
  n.n_frame = n.frame_cols * ((f.M + kFrameRows - 1) / kFrameRows);
  n.n_proj = proj_grid_size(pj.M, pj.N, pj.ksplit);
  const int lrows = (la.M + k.bm - 1) / k.bm;
  n.la_cols = (la.H + k.bu - 1) / k.bu; n.n_la = n.la_cols * lrows;
  n.ld_cols = (ld.H + k.bu - 1) / k.bu;
  n.n_attn = a.B;
  n.n_q = a.q_tiles > a.B ? a.q_tiles : 0;  // (fewer attention workgroups than query tiles: the tiles get workgroups of their own)
  dim3 grid(n.n_proj + n.n_frame + n.n_la + n.n_q + n.n_attn + n.ld_cols * lrows), block(kGemmThreads);
  hipLaunchKernelGGL(k.fn, grid, block, 0, st, f, la, pj, a, ld, pq, n);
}

void launch_lstm_lean(const LstmArgs& l, hipStream_t st) {
  if (l.M <= 0) return;
  const bool small = l.M <= lean8_max_rows(l.prec == 1);
  const LeanKind kind = small ? LEAN_64x8 : LEAN_64x16;
  const int bm = l.prec == 1 ? lean_bm<PREC_F16S>(kind) : lean_bm<PREC_F32>(kind);
  dim3 grid(small ? (l.H + 7) / 8 : (l.H + 15) / 16, (l.M + bm - 1) / bm), block(kGemmThreads);
  if (l.prec == 1) {
    if (small) hipLaunchKernelGGL((lstm_lean_kernel<LeanTiles<PREC_F16S>::Lean64x8>), grid, block, 0, st, l);
    else hipLaunchKernelGGL((lstm_lean_kernel<LeanTiles<PREC_F16S>::Lean64x16>), grid, block, 0, st, l);
  } else {
    if (small) hipLaunchKernelGGL((lstm_lean_kernel<LeanTiles<PREC_F32>::Lean64x8>), grid, block, 0, st, l);
    else hipLaunchKernelGGL((lstm_lean_kernel<LeanTiles<PREC_F32>::Lean64x16>), grid, block, 0, st, l);
  }
}

}  // namespace ttsdec
