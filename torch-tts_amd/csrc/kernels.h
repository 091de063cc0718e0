// Internal launch interface between the C ABI (api.hip) and the kernels.
#pragma once
#include "common.h"

namespace ttsdec {

// ---- generic row-GEMM (PreNet layers, query projection, mel/stop projection, postnet) ----
enum AKind { A_PLAIN = 0, A_CONV = 1 };
enum EpiKind { EPI_PLAIN = 0, EPI_RELU_DROPOUT = 1, EPI_PROJ = 2, EPI_BN_ISRU = 3, EPI_RESIDUAL = 4, EPI_BN_LRELU = 5, EPI_BN_ISRLU = 6, EPI_GENERIC = 7 };

struct GemmArgs {
  // A operand.  A_PLAIN: up to three K segments of an [M, K] activation.
  // A_CONV: implicit im2col of x [Bc*T, Cin] with `taps` taps centred on the row's frame:
  //         A[m][tap*Cin + c] = x[b, t + tap - taps/2, c] (zero outside [0, T)).
  Seg3 a;
  Seg3 a_lo;            // second plane of A (split-fp16 lo), same shape; 16-bit modes only
  int T, Cin, taps;
  // B operand: W [N, K] row-major with leading dimension ldw (element type per `prec`).
  const void* W;
  const void* W_lo;     // split-fp16 lo plane of W
  int ldw;
  int prec;             // PREC_F32 / PREC_F16S / PREC_BF16 (gemm_tile.h): element type of A and W
  int M, N, K;
  // epilogue operands
  const float* bias;   // [N] or nullptr
  const float* alpha;  // EPI_BN_ISRU: out = isru(acc * alpha[n] + beta[n])
  const float* beta;
  const float* resid;  // EPI_RESIDUAL: out = resid[m, n] + acc
  // EPI_GENERIC: out = [resid[m, n] +] row_mask[m] * act(acc + bias[n]); act 0 = identity, 1 = relu;
  // row_mask / resid / bias may each be nullptr (resid has leading dimension ldo)
  int act;
  const float* row_mask;
  float* out;          // [M, ldo]
  int ldo;
  // Split-K over workgroups (A_PLAIN + EPI_PLAIN only): gridDim.z = ksplit, slice z contracts
  // k in [z*kchunk, (z+1)*kchunk) and stores its raw partial sums to out + z*split_stride.
  // The consumer adds the slabs in index order (deterministic); bias must be nullptr.
  int ksplit, kchunk;
  size_t split_stride;
  // optional 16-bit copies of the result for a following 16-bit GEMM (same shape as out):
  // out_kind 1: split-fp16 planes out_h / out_l; out_kind 2: one bf16 plane in out_h.
  // `out` itself may be nullptr when only the 16-bit form is consumed.
  f16* out_h;
  f16* out_l;
  int out_kind;
  // EPI_RELU_DROPOUT
  int dropout_mode;
  const uint8_t* masks;  // [M, N] keep-mask of this (step, layer)
  size_t mask_step_stride, mask_layer_off;  // ctrl mode: masks = ctrl->masks + t_rel*stride + off
  uint64_t seed;
  int layer;
  float keep_scale;  // 1 / (1 - p)
  // EPI_PROJ: columns [0, r*d_mel) -> leaky_relu -> y, columns [r*d_mel, r*d_mel + r) -> stop logits
  float* y_out;  // [B, t_stride*r, d_mel]
  float* s_out;  // [B, t_stride*r]
  float* ynext;  // [B, d_mel] last frame of the group (next step's PreNet input)
  int r, d_mel, t_rel, t_stride;
  float stop_thr;
  int check_stop;
  // teacher forcing (PreNet layer 0 only): input row source switches to the teacher frame
  const float* teacher;  // [B, teacher_T, d_mel] or nullptr
  int teacher_T;
  const uint8_t* teacher_flags;
  // step control.  ctrl != nullptr: the step index is ctrl->t_cur + slot and the per-call
  // pointers (masks, teacher, y, s, ...) come from *ctrl; the fields above are ignored for them.
  // ctrl == nullptr (postnet, cell_step, profiling): everything comes from this struct.
  Ctrl* ctrl;
  int slot;
  int node;  // position of the launch in the step order (measurement: common.h loop_stamp)
  int t;  // absolute step index when ctrl == nullptr
};

void launch_gemm(const GemmArgs& a, AKind ak, EpiKind ek, hipStream_t st);

// ---- LSTM zoneout cell: gates GEMM + cell update fused ----
struct LstmArgs {
  Seg3 a;  // [x | ctx | h_prev] segments, M rows (fp32; or the fp16 hi planes when prec = 1)
  Seg3 w;  // matching segments of [W_ih | W_hh], rows = 4*H in PyTorch gate order i,f,g,o
  Seg3 a_lo, w_lo;  // fp16 lo planes (prec = 1 only)
  int prec;         // 0 = exact fp32 MFMA, 1 = split-fp16 MFMA
  f16 *h_out_h, *h_out_l;  // optional split planes of h_out
  int out_mpad;            // > 0: those planes are in the chunked layout with this many rows per chunk (common.h Seg3)
  const float* bsum;    // [4H] = b_ih + b_hh
  const float* h_prev;  // [M, H]
  float* c;             // [M, H] updated in place
  float* h_out;         // [M, H]
  int M, H, K;
  float pz;  // zoneout probability (eval-mode blend)
  // mode 0: whole cell.  mode 1: gates GEMM over the given segments only, raw sums stored to
  // partial [M, 4H] (PyTorch gate order) - the part of the cell that does not wait for the kernel
  // just before it.  mode 2: the remaining segments, + partial, then the cell update.
  int mode;
  float* partial;
  // Sequence mode (packed LSTM over a padded batch, Encoder2's BiLSTM): seq_lens != nullptr.
  // Row b is active while seq_t < seq_lens[b]; it then reads its input projection from
  // gx[(b*seq_L + pos)*gx_ld + gx_off + gate*H + unit] (pos = seq_t, or len-1-seq_t when reversed)
  // instead of `partial`, and writes h to seq_out[(b*seq_Lout + pos)*seq_out_ld + seq_out_off + unit].
  // Inactive rows keep their state.  bsum may be nullptr (no bias).
  const int* seq_lens;
  int seq_t, seq_L, seq_Lout, seq_reverse;
  const float* gx;
  int gx_ld, gx_off;
  float* seq_out;
  int seq_out_ld, seq_out_off;
  Ctrl* ctrl;
  int slot;
  int node;  // position of the launch in the step order (measurement: common.h loop_stamp)
  int dbg;  // measurement ablations (ttsdec_profile_step only): 1 = every load reads the zero block
  int tag;  // 1 = the decoder LSTM of a decode step (separate kernel symbol for profilers), else 0
  int live_lag;  // 1: this cell runs in the same launch as the frame kernel of its step (see lstm_body)
  // Two-role launch: K segment dep_seg of the A operand is produced by the other role of this very launch (dep_which 0: frame
  // kernel, 1: attention), which signals the arrival counters dep_cnt[32-row block * kDepLine] (common.h): per step dep_n
  // arrivals per block, or - dep_rows = 1 - one per batch row of the block.  dep_n = 0: no gate.
  int dep_n, dep_seg, dep_which, dep_rows;
  unsigned int* dep_cnt;
  // One-launch step (fused_kernels.hip step_kernel): a SECOND gated K segment, dep2_seg < dep_seg, waiting for dep2_n arrivals per
  // 32-row block and step at dep2_cnt (the decoder LSTM's h_att segment, produced by the attention LSTM's tiles of the launch);
  // and sig_cnt != nullptr: this cell is such a producer - its split planes are stored write-through and every tile signals
  // sig_cnt[32-row block * kDepLine] for the row blocks it covers once its outputs have left.
  int dep2_n, dep2_seg;
  unsigned int* dep2_cnt;
  unsigned int* sig_cnt;
};
void launch_lstm(const LstmArgs& a, hipStream_t st);
void launch_lstm_pair(const LstmArgs& a0, const LstmArgs& a1, hipStream_t st);  // two same-shape fp32 cells, one launch

// ---- stepwise monotonic attention + context ----
struct AttnArgs {
  const float* memory;  // [B, L, D]
  const float* q;       // [B, D]; q_parts > 1: q_parts partial slabs q_stride floats apart, summed in order
  int q_parts;
  size_t q_stride;
  const float* w_prev;  // [B, L]
  float* w_new;         // [B, L]
  float* w_out;         // [B, t_stride, L] (row t_rel) or nullptr
  float* ctx;           // [B, D]
  f16 *ctx_h, *ctx_l;   // optional split-fp16 planes of ctx
  int out_mpad;         // > 0: chunked layout of those planes (common.h Seg3)
  int ctx_only;         // 1: no weight update, just ctx = sum_l w_prev[l] * memory[l] (decoder_cell.py:118)
  int dep_signal;       // 1: two-role launch - every workgroup signals dep_cnt[(b / 32) * kDepLine] after its last store
  unsigned int* dep_cnt;
  // Query role (fused_kernels.hip): BEFORE its attention pass workgroup b computes tiles b, b + B, ... of the q_tiles 32 x 32 x
  // K-slice tiles of the query GEMM (the launch's ProjArgs, frame_body.h proj_body) and signals q_cnt[32-row block of the
  // tile]; the attention pass then waits until the q_wait_n tiles of ITS row block have arrived (per step) and takes q with
  // sc1 loads.  q_tiles = 0: q comes from an earlier launch.
  int q_tiles, q_wait_n;
  unsigned int* q_cnt;
  int live_lag;  // 1: the role shares its launch with its step's frame role (one-launch step): live while t - 1 <= stop_t, see lstm_body
  int B, L, D, t_rel, t_stride;
  Ctrl* ctrl;  // != nullptr: memory, w_out, t_rel, t_stride come from *ctrl
  int slot;
  int node;  // position of the launch in the step order (measurement: common.h loop_stamp)
};
void launch_attn(const AttnArgs& a, hipStream_t st);

// ---- frame kernel: finish the previous step's mel/stop projection, then the PreNet ----
// The projection GEMM of step t-1 leaves split-K partial sums; this kernel (first launch of step t)
// adds them up, applies bias / leaky-ReLU (decoder.py:52-54), writes y, s and the stop rule
// (decoder.py:68) for step t-1, picks the next input frame (decoder.py:48, teacher forcing :61-66)
// and runs both PreNet layers on it (modules/modules.py:37-41).  Everything here is local to a
// block of batch rows, so it is one launch instead of three.
struct FrameArgs {
  const float* parts;  // [n_parts][M][ldp] partial sums of [fc_mel; fc_stop]
  int n_parts, ldp;
  size_t part_stride;
  const float* proj_bias;  // [r*d_mel + r]
  float* y_out;            // [B, t_stride*r, d_mel]
  float* s_out;            // [B, t_stride*r]
  float* ynext;            // [B, d_mel] input frame of the next step (kept for the call boundary / cell_step)
  int r, d_mel, t_rel, t_stride;
  float stop_thr;
  int check_stop;
  // ctrl == nullptr: 1 = the partial sums are step t-1's frame; 0 = the input frame is ynext as it stands
  int finalize;
  int only_finalize;  // end-of-call launch: finish the last step's frame, no PreNet
  const float* teacher;
  int teacher_T;
  const uint8_t* teacher_flags;
  const float *W0, *b0, *W1, *b1;  // PreNet: [Ph, d_mel], [Ph], [P, Ph], [P]
  const f16 *W0h, *W0l, *W1h, *W1l;  // split-fp16 planes of W0 / W1 (prec = PREC_F16S)
  int prec;                          // PREC_F32: exact fp32 MFMAs; PREC_F16S: split-fp16 (same accuracy class)
  int Ph, P;
  int dropout_mode;
  const uint8_t* masks;  // this step's keep-masks: layer 0 [M, Ph] then layer 1 [M, P]
  size_t mask_step_stride;
  uint64_t seed;
  float keep_scale;
  float* xpre;           // [M, P]
  f16 *xpre_h, *xpre_l;  // optional split-fp16 planes
  int out_mpad;          // > 0: chunked layout of those planes (common.h Seg3)
  int M;
  Ctrl* ctrl;
  int slot;
  int node;  // position of the launch in the step order (measurement: common.h loop_stamp)
  int t;    // ctrl == nullptr
  int dbg;  // measurement ablations (ttsdec_profile_step only): bit 1 = no layer-0 MFMAs, bit 2 = no layer-1 MFMAs
  int dep_signal;  // 1: two-role launch - every workgroup signals dep_cnt[row block * kDepLine] after its last store
  unsigned int* dep_cnt;
  int wait_n;      // > 0: the partial sums of a row block come from wait_n projection-role workgroups at the head of this very
                   // launch, which signal wait_cnt[row block * kDepLine]
  unsigned int* wait_cnt;
};
bool frame_supported(int d_mel, int r, int Ph, int P);
void launch_frame(const FrameArgs& a, hipStream_t st);

// ---- the mel/stop projection GEMM with its weights in registers (frame_kernel.hip) ----
// out[z][M, ldo] (+ z * split_stride) = A[M, K slice z] * W[N, K slice z]^T, raw partial sums for the frame kernel.
// 32 x 32 outputs per workgroup; the K slice is cut over its 8 waves and every lane HALF holds one contiguous K run of
// its weight row in registers (the frame kernel's permuted-K trick), requested before the control block is read.
// For N = r*d_mel + r <= 96 this beats the LDS-staged tile (5.5 against 6.7 us per step at B = 256, 4.7 / 5.6 at B = 1).
struct ProjArgs {
  Seg3 a, a_lo;          // [M, K] in up to three K segments (multiples of 8): fp32, or the fp16 hi / lo planes (prec = PREC_F16S)
  const void *W, *W_lo;  // [N, ldw] row-major: fp32, or fp16 hi / lo planes
  int ldw, prec;
  int M, N, K;
  int ksplit;            // K slices: proj_split(K); K % (128 * ksplit) == 0 and K / (128 * ksplit) <= 4
  size_t split_stride;
  float* out;
  int ldo;
  Ctrl* ctrl;            // != nullptr: the launch does nothing unless its step is live (see mode)
  int slot;
  int node;              // position of the launch in the step order (measurement: common.h loop_stamp)
  // mode 0: a launch of its own at the end of step t = ctrl->t_cur + slot.  mode 1 (PROJ_HEAD): a ROLE at the head of step t's
  // frame launch, computing step t-1's projection - live like that launch's finalize phase (t-1 <= stop_t, t > t_call) - whose
  // slabs are stored write-through and signalled through dep_cnt[32-row block * kDepLine].  mode 2 (PROJ_FINAL): the projection of the call's
  // last step, a launch of its own in front of the end-of-call frame launch.
  int mode;
  unsigned int* dep_cnt;
  // One-launch step: the A operand is produced inside the launch (PROJ_QUERY: h_att by the attention LSTM's tiles) - wait for
  // wait_n arrivals per step at wait_cnt[32-row block of the tile * kDepLine], then take it with sc1 loads.  live_lag as AttnArgs.
  int wait_n, live_lag;
  int wait_sleep;  // extra s_sleep(8)s between the polls of that wait (a wait of many microseconds beside the LSTM roles' streams)
  unsigned int* wait_cnt;
};
// PROJ_QUERY: live like PROJ_STEP, signals like PROJ_HEAD - the attention query as a job of the attention role's workgroups
enum ProjMode { PROJ_STEP = 0, PROJ_HEAD = 1, PROJ_FINAL = 2, PROJ_QUERY = 3 };
int proj_split(int K);  // the ksplit of this kernel for K, or 0 when it does not cover K
void launch_proj(const ProjArgs& a, hipStream_t st);
void launch_proj_frame(const ProjArgs& pj, const FrameArgs& f, hipStream_t st);  // [proj(t-1) | frame(t)] as one launch (frame_kernel.hip)

// ---- two-role launches (fused_kernels.hip): a latency-bound kernel and the early part of the next LSTM ----
// The LstmArgs must be a split-fp16 cell with M >= 192 whose gated K segment (dep_seg) comes last; its gate GEMM
// runs on the lean tile.
bool fused_supported(int d_mel, int r, int Ph, int P, int D);
int frame_grid_size(int M, int P);                                              // workgroups of the frame role
void launch_frame_lstm(const FrameArgs& f, const LstmArgs& l, hipStream_t st);  // frame kernel || attention LSTM
// ... with the previous step's mel/stop projection as a role at its head: proj(t-1) -> frame(t) || lstm_att(t)
void launch_proj_frame_lstm(const ProjArgs& pj, const FrameArgs& f, const LstmArgs& l, hipStream_t st);
int proj_grid_size(int M, int N, int ksplit);  // workgroups of the projection kernel / role
void launch_attn_lstm(const AttnArgs& a, const LstmArgs& l, const ProjArgs* q, hipStream_t st);  // [query ->] attention || decoder LSTM
void launch_lstm_lean(const LstmArgs& l, hipStream_t st);                       // an LSTM on the lean tile alone (profiling)
// The whole step as ONE launch: [proj(t-1) | frame | lstm_att | query -> attention | lstm_dec] by block id, every role waiting
// only for roles with lower ids (fused_kernels.hip step_kernel).  False: the configuration is not covered (nothing launched).
bool step_merged_supported(int B, int Ha, int Hd, int Ph, int P, int D, int n_out, int ksplit);
// workgroups of the [query ->] attention || decoder-LSTM kernel the current device holds at once (0 = unknown): fused_kernels.hip
int attn_lstm_resident_slots(int B, int H, int D, bool f16);
void launch_step_merged(const ProjArgs& pj, const FrameArgs& f, const LstmArgs& la, const ProjArgs& pq, const AttnArgs& a, const LstmArgs& ld,
                        hipStream_t st);

// ---- state init / bookkeeping ----
struct InitArgs {
  Ctrl* ctrl;
  const float *h0_att, *c0_att, *h0_dec, *c0_dec;  // [H] initial_decoder_{h,c}
  float *h_att, *c_att, *h_dec, *c_dec;            // [B, H]
  float* ctx;                                      // [B, D]
  float* w;                                        // [B, L]
  float* ynext;                                    // [B, d_mel]
  f16 *h_att_h, *h_att_l, *h_dec_h, *h_dec_l, *ctx_h, *ctx_l;  // split planes of the initial state
  int out_mpad;  // > 0: chunked layout of those planes (common.h Seg3)
  const float* memory;  // != nullptr: ctx_0 = bmm(w_0, memory) = memory[:, 0, :] (Taco2DecoderCell, decoder_cell.py:118)
  int B, L, D, Ha, Hd, d_mel;
};
void launch_init(const InitArgs& a, hipStream_t st);
void launch_finish(Ctrl* ctrl, int32_t* T_out, hipStream_t st);
struct CallArgs {  // copied into *ctrl by launch_set_call at the start of every ttsdec_decode
  int t_begin, n_steps, t_stride, check_stop, dropout_mode, teacher_T;
  float stop_thr;
  unsigned long long seed;
  const float* memory;
  const uint8_t* masks;
  const float* teacher;
  const uint8_t* teacher_flags;
  float *y, *s, *w;
  unsigned long long* stamps;  // measurement only
  int debug_flags, spin_limit;  // test hooks (Ctrl)
  unsigned long long* loop_stamps;  // measurement only (Ctrl)
};
void launch_set_call(Ctrl* ctrl, const CallArgs& a, hipStream_t st);
void launch_advance(Ctrl* ctrl, int n_slots, hipStream_t st);

// ---- weight packing ----
void launch_add_vec(const float* a, const float* b, float* out, int n, hipStream_t st);
void launch_copy(const float* src, float* dst, size_t n, hipStream_t st);
void launch_split(const float* src, f16* hi, f16* lo, size_t n, hipStream_t st);
// src [M, K] row-major -> split-fp16 planes in the chunked layout [K / 32][mpad][32] (common.h Seg3)
void launch_split_frame_order(const float* src, f16* hi, f16* lo, int N, int K, int layer, hipStream_t st);  // PreNet planes, frame kernel's lane order
void launch_split_chunked(const float* src, f16* hi, f16* lo, int M, int K, int mpad, hipStream_t st);
// LSTM weight matrix src [4H, K] (PyTorch gate blocks i,f,g,o) -> chunked split-fp16 planes
// [H / 16][K / 32][gate * 16 + unit % 16][32]  (step_bodies.h LoaderWLstm); needs H % 16 == 0, K % 32 == 0
void launch_pack_lstm_chunked(const float* src, f16* hi, f16* lo, int H, int K, hipStream_t st);
void launch_to_bf16(const float* src, void* dst, size_t n, hipStream_t st);
// conv256.hip: out[M, N] bf16 = isru(conv_k(x [M = B * T, Cin] bf16, w [N, taps * Cin] bf16) * alpha[n] + beta[n]) on 256 x 256 tiles;
// false = not that kernel's shape (Cin % 64, N % 256, odd taps, 32-bit offsets): the caller keeps the shared GEMM
bool launch_conv256_bf16(const void* x, const void* w, const float* alpha, const float* beta, void* out, int M, int T, int Cin, int taps, int N,
                         hipStream_t st);
// ... and the same in exact fp32 (x, w, out fp32; Cin % 32)
bool launch_conv256_f32(const float* x, const float* w, const float* alpha, const float* beta, float* out, int M, int T, int Cin, int taps, int N,
                        hipStream_t st);
// *out = max(*out, max |src[i]|)  (*out must hold a non-negative float, e.g. 0)
void launch_absmax(const float* src, size_t n, float* out, hipStream_t st);
// out_a[m, 0:E] (ld lda) and out_b[m, 0:E] (ld ldb) = table[ids[m], :]   (nn.Embedding lookup)
// (ids outside [0, n_table) are clamped and reported: bit 0 of *status, a device word that may be nullptr)
void launch_embed(const long long* ids, const float* table, int n_table, int n_rows, int E, float* out_a, int lda, float* out_b,
                  int ldb, int* status, hipStream_t st);
// device bookkeeping shared by the three handle types
int current_device_or_minus1();
bool device_is_current(int device);
void launch_fill_rows(float* dst, const float* row, int n_rows, int n_cols, hipStream_t st);  // dst[m, :] = row[:]
void launch_conv_transpose(const float* w /*[Co,Ci,k]*/, float* out /*[Co,k,Ci]*/, int Co, int Ci, int k, hipStream_t st);
// Conv1dFix (mps_fixes.py:22-29) pairs flat-weight column n*Ci + c with x[c, t + pad - n]:
// out[co][tap][ci] = wflat[co][(k-1-tap)*Ci + ci]
void launch_conv1dfix_pack(const float* w, float* out, int Co, int Ci, int k, hipStream_t st);
void launch_bn_fold(const float* gamma, const float* betap, const float* mean, const float* var, float eps, float* alpha,
                    float* beta, int n, hipStream_t st);

}  // namespace ttsdec
