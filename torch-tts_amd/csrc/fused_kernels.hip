// Two-role launches of the decode step (B >= 192, split-fp16).
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
// arrival counter (common.h role_signal / role_wait: agent-scope release -> counter -> relaxed poll ->
// acquire, Guideline 16).  Role A has the lowest block ids and waits for nothing, so it is resident or done
// before any LSTM workgroup spins; the spin is bounded regardless.  Normally nobody waits: the early
// segments take longer than role A.
// What makes it pay is co-residency: the LSTM runs on the "lean" GEMM tile (gemm_tile.h: 64-80 KiB of LDS,
// <= 128 VGPRs), so a CU holds one workgroup of each role and the LSTM's tile stream fills the issue slots
// and memory queues the small kernel leaves empty.
// Measured alternatives: the same overlap as two STREAMS (round 1: slower - cross-stream graph
// edges, one 128-KiB workgroup per CU); the early segments as a separate partial-sum launch beside role A plus
// a "late" launch (round 2: 86.4 vs 88.3 us per step - the extra launch costs what the overlap wins).
#include "frame_body.h"
#include "step_bodies.h"

namespace ttsdec {

// 64 rows x 16 units, 16 KiB stages (32 k each) x 5 = 80 KiB
using LeanCfg = TileCfg<2, 2, 1, 5, PREC_F16S, 0, 1, 1, 1>;
constexpr int kLeanLds = LeanCfg::kLdsFloats;

template <int A, int B>
constexpr int cmax() { return A > B ? A : B; }

// ---- role A = frame kernel (finish proj(t-1), PreNet), role B = early part of the attention LSTM ----
template <int K0H, int PH>
__global__ __launch_bounds__(kGemmThreads, 4) void frame_lstm_kernel(FrameArgs f, LstmArgs l, int n_frame, int frame_cols, int lstm_cols) {
  __shared__ __attribute__((aligned(16))) float smem[cmax<kLeanLds, FrameLds<K0H, PH, PREC_F16S>::kFloats>()];
  const int id = blockIdx.x;
  if (id < n_frame) {
    __builtin_amdgcn_s_setprio(3);  // the producer role is the launch's critical path: it wins issue arbitration
    frame_body<K0H, PH, PREC_F16S, 6>(f, smem, id % frame_cols, id / frame_cols);
  } else {
    const int j = id - n_frame;
    lstm_body<LeanCfg, false>(l, smem, j % lstm_cols, j / lstm_cols);
  }
}

// ---- role A = attention + context, role B = early part of the decoder LSTM ----
template <int NJ>
__global__ __launch_bounds__(kGemmThreads, 4) void attn_lstm_kernel(AttnArgs a, LstmArgs l, int n_attn, int lstm_cols) {
  __shared__ __attribute__((aligned(16))) float smem[cmax<kLeanLds, attn_lds_floats<NJ>()>()];
  const int id = blockIdx.x;
  if (id < n_attn) {
    __builtin_amdgcn_s_setprio(3);  // (as above: the decoder LSTM's last segment waits for every one of these)
    attn_body<NJ>(a, smem, id);
  } else {
    const int j = id - n_attn;
    lstm_body<LeanCfg, false>(l, smem, j % lstm_cols, j / lstm_cols);
  }
}

// the early part on its own (profiling / ablation: what the role costs without a partner)
__global__ __launch_bounds__(kGemmThreads, 4) void lstm_lean_kernel(LstmArgs l) {
  __shared__ __attribute__((aligned(16))) float smem[kLeanLds];
  lstm_body<LeanCfg, false>(l, smem, blockIdx.x, blockIdx.y);
}

static_assert(kFrameThreads == kGemmThreads && kAttnThreads == kGemmThreads, "roles share one block size");

bool fused_supported(int d_mel, int r, int Ph, int P, int D) {
  // (frame role: 6 projection columns per thread; attention role: NJ <= 2 - both to stay within 128 VGPRs)
  return frame_supported(d_mel, r, Ph, P) && r * d_mel + r <= 96 && D / 4 <= 128;
}

int frame_grid_size(int M, int P) { return ((P + kFrameCols - 1) / kFrameCols) * ((M + kFrameRows - 1) / kFrameRows); }

void launch_frame_lstm(const FrameArgs& f, const LstmArgs& l, hipStream_t st) {
  if (f.M <= 0) return;
  const int fcols = (f.P + kFrameCols - 1) / kFrameCols, frows = (f.M + kFrameRows - 1) / kFrameRows;
  const int lcols = (l.H + 15) / 16, lrows = (l.M + 63) / 64;
  const int n_frame = fcols * frows;
  dim3 grid(n_frame + lcols * lrows), block(kGemmThreads);
  if (f.Ph == 256) hipLaunchKernelGGL((frame_lstm_kernel<40, 256>), grid, block, 0, st, f, l, n_frame, fcols, lcols);
  else hipLaunchKernelGGL((frame_lstm_kernel<40, 128>), grid, block, 0, st, f, l, n_frame, fcols, lcols);
}

void launch_attn_lstm(const AttnArgs& a, const LstmArgs& l, hipStream_t st) {
  if (a.B <= 0) return;
  const int lcols = (l.H + 15) / 16, lrows = (l.M + 63) / 64;
  dim3 grid(a.B + lcols * lrows), block(kGemmThreads);
  if (a.D / 4 <= 64) hipLaunchKernelGGL((attn_lstm_kernel<1>), grid, block, 0, st, a, l, a.B, lcols);
  else hipLaunchKernelGGL((attn_lstm_kernel<2>), grid, block, 0, st, a, l, a.B, lcols);
}

void launch_lstm_lean(const LstmArgs& l, hipStream_t st) {
  if (l.M <= 0) return;
  hipLaunchKernelGGL(lstm_lean_kernel, dim3((l.H + 15) / 16, (l.M + 63) / 64), dim3(kGemmThreads), 0, st, l);
}

}  // namespace ttsdec
