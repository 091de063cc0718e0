// C ABI of libttsdec.so (see include/ttsdec.h).  Host-side orchestration only:
// argument checks, blob / workspace carving, and the per-step launch sequence.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>

#include "kernels.h"

using namespace ttsdec;

namespace ttsdec {
int current_device_or_minus1() {
  int ndev = 0, dev = -1;
  if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0 && hipGetDevice(&dev) == hipSuccess) return dev;
  (void)hipGetLastError();
  return -1;
}
bool device_is_current(int device) {
  if (device < 0) return false;
  int cur = -1;
  return hipGetDevice(&cur) == hipSuccess && cur == device;
}
}  // namespace ttsdec

namespace {

constexpr int kMaxPostnetLayers = 8;
constexpr size_t kAlignFloats = 64;  // 256 B

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct BlobLayout {  // offsets in floats
  size_t pre0_w, pre0_b, pre1_w, pre1_b, wq;
  size_t pre0_h, pre0_l, pre1_h, pre1_l;  // split-fp16 planes of the PreNet weights
  size_t pre0_ph, pre0_pl, pre1_ph, pre1_pl;  // the same in the frame kernel's lane order (launch_split_frame_order; layer 1 padded to 64 rows)
  size_t wq_h, wq_l, proj_h, proj_l;      // ... of the query and mel/stop projection weights
  size_t att_ih, att_hh, att_b, dec_ih, dec_hh, dec_b;
  size_t att_ih_h, att_ih_l, att_hh_h, att_hh_l, dec_ih_h, dec_ih_l, dec_hh_h, dec_hh_l;  // split-fp16 planes
  // the same planes in the chunked layout (common.h Seg3 / launch_pack_lstm_chunked), used when chunk_ok(dims)
  size_t att_ih_ch, att_ih_cl, att_hh_ch, att_hh_cl, dec_ih_ch, dec_ih_cl, dec_hh_ch, dec_hh_cl;
  size_t h0a, c0a, h0d, c0d;
  size_t proj_w, proj_b;
  size_t conv_w[kMaxPostnetLayers], conv_alpha[kMaxPostnetLayers], conv_beta[kMaxPostnetLayers];
  size_t conv_wb[kMaxPostnetLayers], conv_wh[kMaxPostnetLayers], conv_wl[kMaxPostnetLayers];  // bf16 / split-fp16 planes
  size_t fc_w, fc_wb, fc_wh, fc_wl;
  // MelPostnet2: per layer three Conv1dFix weights (+ 16-bit planes) and two folded BatchNorms
  size_t p2_w[kMaxPostnetLayers][3], p2_wb[kMaxPostnetLayers][3], p2_wh[kMaxPostnetLayers][3], p2_wl[kMaxPostnetLayers][3];
  size_t p2_alpha[kMaxPostnetLayers][2], p2_beta[kMaxPostnetLayers][2];
  size_t wmax;  // [0] max |w| over the decoder matrices that have split-fp16 planes, [1] the same over the Postnet's
  size_t total;
};

struct WsLayout {  // offsets in bytes
  size_t ctrl, xpre0, xpre, ctx, h_att[2], c_att, h_dec[2], c_dec, q, ynext, w[2];
  size_t xpre_h, xpre_l, ctx_h, ctx_l, h_att_h[2], h_att_l[2], h_dec_h[2], h_dec_l[2];  // split-fp16 planes
  size_t jparts;  // split-K partial sums of the mel/stop projection [kProjSplit][B, proj_ldp]
  size_t pa, pd;  // fp32 partial gate sums [B, 4H] of the two LSTMs' early parts (two-role step)
  size_t loop_stamps;     // measurement only: launch start times of one graph replay [kLoopStampSlots][kLoopStampNodes] (common.h)
  size_t dep, dep_bytes;  // arrival counters of the two-role launches: [DEP_KINDS][32-row blocks][kDepLine] (common.h)
  size_t total;
};

}  // namespace

constexpr int kGraphSlots = 30;  // decode steps per captured graph (even: buffer parity is baked per slot)
// Split-K factors of the two small GEMMs of a step (their consumers add the slabs in order):
// at B=256 the query projection then launches 256 workgroups instead of 128 and the mel/stop
// projection 96 instead of 24, each with a proportionally shorter K loop.
constexpr int kQuerySplit = 4;  // (upper bound: 2 slices for K = 1024, 4 for the Taco2 cell's K = 2048)
constexpr int kProjSplit = 4;

struct ttsdec_handle {
  ttsdec_dims d;
  int precision;  // TTSDEC_PREC_*
  int device;  // -1: no HIP device was available at create (host-only queries still work)
  BlobLayout bl;
  const float* blob;
  std::string hip_err;
  bool use_graph;   // replay a captured hipGraph instead of launching every kernel (option "graph")
  // Two-role launches (fused_kernels.hip) where they apply: 1 = frame || lstm_att; 2 = also attention || lstm_dec;
  // 0 = off; -1 (default) = by batch size: level 2 up to 320 utterances, level 1 above.  Measured us per step for levels
  // 1 / 2 (end of round 2): B = 128 69.0 / 53.7, B = 192 66.2 / 59.5, B = 256 77.2 / 70.2.  Until the attention role's row sums
  // moved from ds_bpermute to DPP (common.h wave_sum) level 2 LOST from 192 utterances on (B = 256: 80.8 / 86.1): the role's
  // LDS-path shuffles queued behind the co-resident LSTM workgroup's LDS traffic, the attention workgroups finished late and
  // every LSTM workgroup waited for the slowest of them.  (Split-fp16; exact fp32: see overlap_level.)
  // Option "overlap" = 0 / 1 / 2.
  int overlap;
  bool chunk_a, chunk_b;  // chunked layout of the activation planes / LSTM weight planes (options "chunk_a" / "chunk_b")
  bool proj_regw;         // mel/stop projection on the register-weight kernel where it applies (option "proj_regw")
  int opt_graph, opt_chunk_a, opt_chunk_b, opt_proj_regw;  // the options behind those four: -1 = default (on), 0, 1
  int head_proj;          // that projection as a role at the head of the NEXT step's frame launch: 1 / 0, -1 = by batch size; TTSDEC_HEAD_PROJ
  int merged_tune;        // measurement: knobs of the one-launch step (include/ttsdec.h)
  int query_role;         // attention query as a job of the attention role's workgroups (step_order): 1 / 0, -1 = default
  int profile_ablation;   // ttsdec_profile_step only: the kernels' dbg switches (measurement ablations)
  int debug_flags;        // test hooks, copied into Ctrl::debug_flags: bit 0 = the frame role does not signal, bit 1 = the attention
                          // role does not, bit 2 = the projection head role does not (drives the bounded-spin time-out path)
  int spin_limit;         // polls before a role gives up waiting (common.h role_wait); 0 = kRoleSpinLimit
  hipStream_t cap_stream;
  bool streams_ready;
  // one cached graph: valid for exactly this (workspace, blob, B, L, precision)
  hipGraphExec_t gexec;
  hipGraph_t graph;
  const void* g_ws;
  const void* g_blob;
  int g_B, g_L, g_prec;
  // max |w| read back from the blob header (pack / bind): a weight at or beyond the fp16 range has no
  // split-fp16 form, so the affected GEMMs stay on exact fp32 (ttsdec_get_precision reports it)
  float wmax_dec, wmax_post;
  unsigned long long* stamps_buf;  // measurement only (TTSDEC_STAMPS): device buffer of this handle's device
};

namespace {

// derived sizes that depend on the cell type
inline int pre_hidden(const ttsdec_dims& d) { return d.d_pre_hidden > 0 ? d.d_pre_hidden : d.d_pre; }
inline bool is_taco2(const ttsdec_dims& d) { return d.cell_type == TTSDEC_CELL_TACO2; }
inline int query_ld(const ttsdec_dims& d) { return is_taco2(d) ? d.h_att + d.h_dec + d.d_ctx : d.h_att; }
inline int query_k(const ttsdec_dims& d) { return is_taco2(d) ? d.h_att + d.h_dec : d.h_att; }
inline int proj_ld(const ttsdec_dims& d) { return is_taco2(d) ? d.h_att + d.h_dec + d.d_ctx : d.h_dec + d.d_ctx; }
inline int proj_k(const ttsdec_dims& d) { return is_taco2(d) ? d.h_att + d.h_dec : d.h_dec + d.d_ctx; }
inline int proj_n(const ttsdec_dims& d) { return d.r * d.d_mel + d.r; }
inline int proj_ldp(const ttsdec_dims& d) { return (proj_n(d) + 3) & ~3; }
// The fused frame kernel (frame_kernel.hip) covers the shipped PreNet shapes; other dims keep the
// three-launch form (proj with its own epilogue, prenet0, prenet1).
inline bool use_frame(const ttsdec_dims& d) { return frame_supported(d.d_mel, d.r, pre_hidden(d), d.d_pre); }
// The chunked operand layout of the LSTM GEMMs (common.h Seg3): whole 32-element chunks in every K segment,
// whole 16-unit blocks, and the frame kernel as the producer of the x_pre planes.
inline bool chunk_ok(const ttsdec_dims& d) {
  return use_frame(d) && !((d.d_pre | d.d_ctx | d.h_att | d.h_dec) & 31);
}
inline int rows_pad(int B) { return (B + 63) / 64 * 64; }  // rows per chunk of the activation planes
inline int split_of(int K, int want);
inline int query_split(const ttsdec_dims& d);
inline int split_of(int K, int want) {  // split-K factor: slices must be whole 128-element K tiles
  while (want > 1 && (K % want || (K / want) % 128)) --want;
  return want;
}

inline int query_split(const ttsdec_dims& d) { return split_of(query_k(d), query_k(d) >= 2048 ? kQuerySplit : 2); }
// The mel/stop projection on the register-weight kernel (frame_kernel.hip proj_kernel) in split-fp16 mode where it
// covers the shape: whole 8-element groups in every K segment, K slices of whole 128-element blocks; else the LDS-staged
// split-K GEMM.  Measured, proj launch / step in the loop: B = 256 5.0 / 83.6 us against 6.7 / 84.1, B = 1 4.3 / 46.9
// against 5.5 / 47.9; exact fp32 (64-cycle MFMAs, 8 per k16 step) 7.0 against 6.4 - so split-fp16 only.
// (Exact fp32 takes it where the projection then becomes the head role of the next step's first launch - batches with the
// two-role launches: the role's 7 us hide under the fp32 attention LSTM's 48-us stream and the step loses the projection's own
// launch: 128.6 -> 121.4 us per step at B = 256, 56.2 -> 54.4 at B = 32, 54.0 -> 50.2 at B = 1; profiles/r03_w_fp32_head_proj.txt.)
inline bool proj_regw_shapes(const ttsdec_handle* h) {
  const ttsdec_dims& d = h->d;
  return h->proj_regw && use_frame(d) && !((d.h_att | d.h_dec | d.d_ctx) & 7) && proj_split(proj_k(d)) > 0 && proj_n(d) <= 192;
}
inline bool proj_regw(const ttsdec_handle* h, int prec, int B) {
  (void)B;
  return proj_regw_shapes(h) && (prec || h->head_proj != 0);  // (exact fp32: as a head role - head_proj() - only)
}
inline int proj_parts(const ttsdec_handle* h, int prec, int B) {
  return proj_regw(h, prec, B) ? proj_split(proj_k(h->d)) : split_of(proj_k(h->d), kProjSplit);
}

BlobLayout make_blob_layout(const ttsdec_dims& d) {
  BlobLayout L;
  memset(&L, 0, sizeof(L));
  size_t off = 0;
  auto take = [&](size_t n) {
    const size_t o = off;
    off = align_up(off + n, kAlignFloats);
    return o;
  };
  L.wmax = take(2);
  const size_t Ha = d.h_att, Hd = d.h_dec, D = d.d_ctx, P = d.d_pre, Mel = d.d_mel, R = d.r, Ph = pre_hidden(d);
  L.pre0_w = take(Ph * Mel);
  L.pre0_b = take(Ph);
  L.pre1_w = take(P * Ph);
  L.pre1_b = take(P);
  L.wq = take(D * (size_t)query_ld(d));
  L.att_ih = take(4 * Ha * (P + D));
  L.att_hh = take(4 * Ha * Ha);
  L.att_b = take(4 * Ha);
  L.dec_ih = take(4 * Hd * (Ha + D));
  L.dec_hh = take(4 * Hd * Hd);
  L.dec_b = take(4 * Hd);
  // fp16 planes occupy half a float per element
  L.pre0_h = take((Ph * Mel + 1) / 2); L.pre0_l = take((Ph * Mel + 1) / 2);
  L.pre1_h = take((P * Ph + 1) / 2); L.pre1_l = take((P * Ph + 1) / 2);
  L.pre0_ph = take((Ph * Mel + 1) / 2); L.pre0_pl = take((Ph * Mel + 1) / 2);
  const size_t p64 = (P + 63) / 64 * 64;
  L.pre1_ph = take((p64 * Ph + 1) / 2); L.pre1_pl = take((p64 * Ph + 1) / 2);
  L.att_ih_h = take(2 * Ha * (P + D)); L.att_ih_l = take(2 * Ha * (P + D));
  L.att_hh_h = take(2 * Ha * Ha);      L.att_hh_l = take(2 * Ha * Ha);
  L.dec_ih_h = take(2 * Hd * (Ha + D)); L.dec_ih_l = take(2 * Hd * (Ha + D));
  L.dec_hh_h = take(2 * Hd * Hd);      L.dec_hh_l = take(2 * Hd * Hd);
  if (chunk_ok(d)) {
    L.att_ih_ch = take(2 * Ha * (P + D)); L.att_ih_cl = take(2 * Ha * (P + D));
    L.att_hh_ch = take(2 * Ha * Ha);      L.att_hh_cl = take(2 * Ha * Ha);
    L.dec_ih_ch = take(2 * Hd * (Ha + D)); L.dec_ih_cl = take(2 * Hd * (Ha + D));
    L.dec_hh_ch = take(2 * Hd * Hd);      L.dec_hh_cl = take(2 * Hd * Hd);
  }
  L.h0a = take(Ha);
  L.c0a = take(Ha);
  L.h0d = take(Hd);
  L.c0d = take(Hd);
  L.proj_w = take((R * Mel + R) * (size_t)proj_ld(d));
  L.proj_b = take(R * Mel + R);
  L.wq_h = take((D * (size_t)query_ld(d) + 1) / 2); L.wq_l = take((D * (size_t)query_ld(d) + 1) / 2);
  L.proj_h = take(((R * Mel + R) * (size_t)proj_ld(d) + 1) / 2); L.proj_l = take(((R * Mel + R) * (size_t)proj_ld(d) + 1) / 2);
  size_t cin = Mel;
  if (d.postnet_type == TTSDEC_POSTNET_TYPE_MEL2) {
    const size_t Hh = d.postnet_hidden, k = d.postnet_kernel;
    const size_t shapes[3][2] = {{Hh, Mel}, {Hh, Hh}, {Mel, Hh}};  // [C_out, C_in] of the three convs
    for (int i = 0; i < d.postnet_layers; ++i) {
      for (int c = 0; c < 3; ++c) {
        const size_t n = shapes[c][0] * shapes[c][1] * k;
        L.p2_w[i][c] = take(n);
        L.p2_wb[i][c] = take((n + 1) / 2); L.p2_wh[i][c] = take((n + 1) / 2); L.p2_wl[i][c] = take((n + 1) / 2);
      }
      for (int c = 0; c < 2; ++c) { L.p2_alpha[i][c] = take(Hh); L.p2_beta[i][c] = take(Hh); }
    }
    L.total = off;
    return L;
  }
  for (int i = 0; i < d.postnet_layers; ++i) {
    L.conv_w[i] = take((size_t)d.postnet_hidden * d.postnet_kernel * cin);
    L.conv_alpha[i] = take(d.postnet_hidden);
    L.conv_beta[i] = take(d.postnet_hidden);
    const size_t nh = ((size_t)d.postnet_hidden * d.postnet_kernel * cin + 1) / 2;  // 16-bit plane, in floats
    L.conv_wb[i] = take(nh); L.conv_wh[i] = take(nh); L.conv_wl[i] = take(nh);
    cin = d.postnet_hidden;
  }
  if (d.postnet_layers > 0) {
    L.fc_w = take(Mel * (size_t)d.postnet_hidden);
    const size_t nh = (Mel * (size_t)d.postnet_hidden + 1) / 2;
    L.fc_wb = take(nh); L.fc_wh = take(nh); L.fc_wl = take(nh);
  }
  L.total = off;
  return L;
}

WsLayout make_ws_layout(const ttsdec_dims& d, int B, int Lm) {
  WsLayout W;
  size_t off = 0;
  auto take = [&](size_t nfloats) {
    const size_t o = off;
    off = align_up(off + nfloats * sizeof(float), 256);
    return o;
  };
  W.ctrl = off;
  off += align_up(sizeof(Ctrl), 256);
  const size_t b = (size_t)B;
  W.xpre0 = take(b * pre_hidden(d));
  W.xpre = take(b * d.d_pre);
  W.ctx = take(b * d.d_ctx);
  W.h_att[0] = take(b * d.h_att);
  W.h_att[1] = take(b * d.h_att);
  W.c_att = take(b * d.h_att);
  W.h_dec[0] = take(b * d.h_dec);
  W.h_dec[1] = take(b * d.h_dec);
  W.c_dec = take(b * d.h_dec);
  W.q = take((size_t)kQuerySplit * b * d.d_ctx);
  W.ynext = take(b * d.d_mel);
  W.w[0] = take(b * Lm);
  W.w[1] = take(b * Lm);
  auto takeh = [&](size_t nhalfs) { return take((nhalfs + 1) / 2); };
  const size_t bp = (size_t)rows_pad(B);  // (the chunked layout pads the rows of a plane to whole 64-row blocks)
  W.xpre_h = takeh(bp * d.d_pre); W.xpre_l = takeh(bp * d.d_pre);
  W.ctx_h = takeh(bp * d.d_ctx); W.ctx_l = takeh(bp * d.d_ctx);
  for (int i = 0; i < 2; ++i) {
    W.h_att_h[i] = takeh(bp * d.h_att); W.h_att_l[i] = takeh(bp * d.h_att);
    W.h_dec_h[i] = takeh(bp * d.h_dec); W.h_dec_l[i] = takeh(bp * d.h_dec);
  }
  W.jparts = take((size_t)kProjSplit * b * proj_ldp(d));
  W.pa = take(b * 4 * d.h_att);
  W.pd = take(b * 4 * d.h_dec);
  W.loop_stamps = take(kLoopStampSlots * kLoopStampNodes * 2);
  W.dep_bytes = (size_t)DEP_KINDS * ((B + 31) / 32) * kDepLine * sizeof(unsigned int);
  W.dep = take(W.dep_bytes / sizeof(float));
  W.total = off;
  return W;
}

int check_dims(const ttsdec_dims& d) {
  const int v[] = {d.d_mel, d.d_pre, d.d_ctx, d.h_att, d.h_dec};
  for (int x : v)
    if (x <= 0 || (x & 3)) return TTSDEC_ERR_DIMS;
  if (d.r < 1 || d.d_ctx > 1024) return TTSDEC_ERR_DIMS;  // attn_kernel<4> covers d_ctx/4 <= 256 float4 columns
  if (d.d_pre_hidden < 0 || (d.d_pre_hidden & 3)) return TTSDEC_ERR_DIMS;
  if (d.cell_type != TTSDEC_CELL_TACO2PROD && d.cell_type != TTSDEC_CELL_TACO2) return TTSDEC_ERR_DIMS;
  if (d.postnet_type != TTSDEC_POSTNET_TYPE_MEL && d.postnet_type != TTSDEC_POSTNET_TYPE_MEL2) return TTSDEC_ERR_DIMS;
  if (d.postnet_layers < 0 || d.postnet_layers > kMaxPostnetLayers) return TTSDEC_ERR_DIMS;
  if (d.postnet_layers > 0) {
    if (d.postnet_hidden <= 0 || (d.postnet_hidden & 3)) return TTSDEC_ERR_DIMS;
    if (d.postnet_kernel < 1 || !(d.postnet_kernel & 1)) return TTSDEC_ERR_DIMS;
  }
  if (!(d.p_zoneout >= 0.f && d.p_zoneout < 1.f)) return TTSDEC_ERR_DIMS;
  if (!(d.p_dropout >= 0.f && d.p_dropout < 1.f)) return TTSDEC_ERR_DIMS;
  return TTSDEC_OK;
}

int hip_fail(ttsdec_handle* h, hipError_t e, const char* where) {
  if (h) h->hip_err = std::string(where) + ": " + hipGetErrorString(e);
  return TTSDEC_ERR_HIP;
}

#define HIP_TRY(h, expr)                                  \
  do {                                                    \
    hipError_t _e = (expr);                               \
    if (_e != hipSuccess) return hip_fail(h, _e, #expr); \
  } while (0)

int check_device(ttsdec_handle* h) { return device_is_current(h->device) ? TTSDEC_OK : TTSDEC_ERR_DEVICE; }

int check_launch(ttsdec_handle* h, const char* where) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(h, e, where);
  return TTSDEC_OK;
}

struct StepBufs {
  Ctrl* ctrl;
  float *xpre0, *xpre, *ctx, *h_att[2], *c_att, *h_dec[2], *c_dec, *q, *ynext, *w[2];
  f16 *xpre_h, *xpre_l, *ctx_h, *ctx_l, *h_att_h[2], *h_att_l[2], *h_dec_h[2], *h_dec_l[2];
  float* jparts;
  float *pa, *pd;
  unsigned long long* loop_stamps;
  unsigned int* dep;
  int dep_blocks;  // 32-row blocks of the batch: dep + kind * dep_blocks * kDepLine is a hand-off kind's counter array
};

StepBufs carve(const WsLayout& W, void* ws) {
  char* p = static_cast<char*>(ws);
  StepBufs s;
  s.ctrl = reinterpret_cast<Ctrl*>(p + W.ctrl);
  auto f = [&](size_t off) { return reinterpret_cast<float*>(p + off); };
  s.xpre0 = f(W.xpre0); s.xpre = f(W.xpre); s.ctx = f(W.ctx);
  s.h_att[0] = f(W.h_att[0]); s.h_att[1] = f(W.h_att[1]); s.c_att = f(W.c_att);
  s.h_dec[0] = f(W.h_dec[0]); s.h_dec[1] = f(W.h_dec[1]); s.c_dec = f(W.c_dec);
  s.q = f(W.q); s.ynext = f(W.ynext); s.w[0] = f(W.w[0]); s.w[1] = f(W.w[1]);
  auto hf = [&](size_t off) { return reinterpret_cast<f16*>(p + off); };
  s.xpre_h = hf(W.xpre_h); s.xpre_l = hf(W.xpre_l); s.ctx_h = hf(W.ctx_h); s.ctx_l = hf(W.ctx_l);
  for (int i = 0; i < 2; ++i) {
    s.h_att_h[i] = hf(W.h_att_h[i]); s.h_att_l[i] = hf(W.h_att_l[i]);
    s.h_dec_h[i] = hf(W.h_dec_h[i]); s.h_dec_l[i] = hf(W.h_dec_l[i]);
  }
  s.jparts = f(W.jparts);
  s.pa = f(W.pa); s.pd = f(W.pd);
  s.loop_stamps = reinterpret_cast<unsigned long long*>(p + W.loop_stamps);
  s.dep = reinterpret_cast<unsigned int*>(p + W.dep);
  s.dep_blocks = (int)(W.dep_bytes / (DEP_KINDS * kDepLine * sizeof(unsigned int)));
  return s;
}

// How one step's kernels find "now": inside ttsdec_decode everything per-call lives in the
// device control block and a kernel only gets its slot (use_ctrl); cell_step / profiling pass
// explicit values.
struct StepIo {
  const float* memory;
  int B, L;
  int slot;               // use_ctrl: step = ctrl->t_cur + slot; buffer parity = slot & 1 (t_cur is even)
  int t, t_rel, t_stride;  // !use_ctrl
  int dropout_mode;
  const uint8_t* masks;  // !use_ctrl: [2, B, d_pre] of this step
  uint64_t seed;
  float *y, *s, *w;
  bool use_ctrl;
  int finalize;  // !use_ctrl, frame kernel: the projection partial sums are the previous step's frame
  int dbg;
  int node_pos;  // position of the launch in the step order (measurement: common.h loop_stamp)
};

// The kernels of one decode step.  N_F is the fused frame kernel (finish the previous step's
// projection + both PreNet layers), N_FIN its end-of-call form; N_P0 / N_P1 are the separate
// PreNet layers used when the dims are outside what the frame kernel covers.
// Two-role step (fused_kernels.hip): N_FA = frame || attention LSTM, N_TD = attention || decoder LSTM;
// N_AG / N_DG are those LSTMs alone on the lean tile (profiling).  N_JFA = N_FA with the PREVIOUS step's mel/stop projection as a
// role at its head (then no N_J in the step), N_JFIN = the projection of a call's last step.
// N_QTD = N_TD whose attention-role workgroups first compute the query GEMM between them (then no N_Q in the step).
// N_STEP = N_JFA + N_QTD as one launch.
// N_JF = [proj(t-1) | frame] as one launch (no LSTM role).
enum Node { N_F, N_FIN, N_P0, N_P1, N_A, N_Q, N_T, N_D, N_J, N_FA, N_TD, N_AG, N_DG, N_JFA, N_JFIN, N_QTD, N_STEP, N_JF };
// PART_GATED: the whole cell with the segment that waits for the other role of the launch LAST
enum LstmPart { PART_WHOLE = 0, PART_EARLY = 1, PART_LATE = 2, PART_GATED = 3 };

constexpr int kMaxKernelsPerStep = 7;
struct StepOrder {
  int n;
  Node nodes[kMaxKernelsPerStep];
  const char* names[kMaxKernelsPerStep];
};

// split-fp16 needs every K segment to be whole 16-byte columns of fp16 (multiples of 8)
bool split_ok(const ttsdec_dims& d) { return !((d.d_pre | d.d_ctx | d.h_att | d.h_dec) & 7); }
// launch order of one step: the Prod cell attends between its two LSTMs, the Taco2 cell after both
const StepOrder& step_order(const ttsdec_handle* h, int B);
bool head_proj(const ttsdec_handle* h, int B);
bool query_role(const ttsdec_handle* h, int B);
bool step_merged(const ttsdec_handle* h, int B);
int& option_ref(ttsdec_handle* h, int o);
void apply_env_options(ttsdec_handle* h);
void drop_graph(ttsdec_handle* h);
int lstm_prec(const ttsdec_handle* h) {
  return (h->precision == TTSDEC_PREC_SPLIT_F16 && split_ok(h->d) && h->wmax_dec < kSplitMax) ? 1 : 0;
}

// reads the two weight maxima back from the blob header (one small synchronous copy)
int read_wmax(ttsdec_handle* h, hipStream_t st) {
  float v[2] = {0.f, 0.f};
  if (hipStreamSynchronize(st) != hipSuccess) return hip_fail(h, hipGetLastError(), "wmax sync");
  hipError_t e = hipMemcpy(v, h->blob + h->bl.wmax, sizeof(v), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return hip_fail(h, e, "wmax readback");
  // (a NaN maximum compares false below, i.e. also keeps the GEMMs on fp32)
  h->wmax_dec = v[0] == v[0] ? v[0] : INFINITY;
  h->wmax_post = v[1] == v[1] ? v[1] : INFINITY;
  return TTSDEC_OK;
}

void launch_node(const ttsdec_handle* h, const StepBufs& sb, const StepIo& io, Node node, hipStream_t st) {
  const ttsdec_dims& d = h->d;
  const BlobLayout& bl = h->bl;
  const float* blob = h->blob;
  const int p = (io.use_ctrl ? io.slot : io.t) & 1;
  Ctrl* ctrl = io.use_ctrl ? sb.ctrl : nullptr;
  const int B = io.B, P = d.d_pre, D = d.d_ctx, Ha = d.h_att, Hd = d.h_dec;
  const float keep_scale = 1.0f / (1.0f - d.p_dropout);
  const int prec = lstm_prec(h);
  auto plane = [&](size_t float_off) { return reinterpret_cast<const f16*>(blob + float_off); };
  // chunked operand layout of the split-fp16 planes (0 = row-major)
  const int mpad = (prec && chunk_ok(d) && h->chunk_a) ? rows_pad(B) : 0;
  auto act = [&](Seg3 s) { return chunked(s, mpad); };  // an activation-plane segment list in the layout in use
  auto dep = [&](int kind) { return sb.dep + (size_t)kind * sb.dep_blocks * kDepLine; };  // a hand-off kind's arrival counters

  auto frame_args = [&](bool fin_only) {
    FrameArgs f;
    memset(&f, 0, sizeof(f));
    const int Ph = pre_hidden(d);
    f.parts = sb.jparts; f.n_parts = proj_parts(h, prec, B); f.ldp = proj_ldp(d);
    f.part_stride = (size_t)B * proj_ldp(d);
    f.proj_bias = blob + bl.proj_b;
    f.y_out = io.y; f.s_out = io.s; f.ynext = sb.ynext;
    f.r = d.r; f.d_mel = d.d_mel; f.t_rel = io.t_rel; f.t_stride = io.t_stride;
    f.finalize = io.finalize; f.only_finalize = fin_only ? 1 : 0;
    f.dbg = io.dbg;
    if (io.dbg & 1) f.finalize = 0;       // measurement ablations (profile_step only)
    if (io.dbg & 8) f.only_finalize = 1;
    f.W0 = blob + bl.pre0_w; f.b0 = blob + bl.pre0_b; f.W1 = blob + bl.pre1_w; f.b1 = blob + bl.pre1_b;
    f.W0h = plane(bl.pre0_ph); f.W0l = plane(bl.pre0_pl); f.W1h = plane(bl.pre1_ph); f.W1l = plane(bl.pre1_pl);  // (lane order)
    f.prec = prec ? PREC_F16S : PREC_F32;
    f.Ph = Ph; f.P = P;
    f.dropout_mode = io.dropout_mode; f.masks = io.masks; f.mask_step_stride = (size_t)B * (Ph + P);
    f.seed = io.seed; f.keep_scale = keep_scale;
    f.xpre = sb.xpre;
    if (prec) { f.xpre_h = sb.xpre_h; f.xpre_l = sb.xpre_l; f.out_mpad = mpad; }
    f.M = B; f.ctrl = ctrl; f.slot = io.slot; f.node = io.node_pos; f.t = io.t;
    return f;
  };
  // One LSTM cell as a whole, or cut along its K axis (fused_kernels.hip): EARLY = the segments that do not wait
  // for the launch just before the cell (raw gate sums parked in `partial`), LATE = the segment that does, plus
  // the parked sums, then the cell update.  which = 0: attention LSTM, cat[x_pre, ctx_prev | h_att]
  // (decoder_cell.py:187), early = [ctx_prev | h_att]; which = 1: decoder LSTM, cat[h_att, ctx | h_dec] (:191),
  // early = [h_att | h_dec].
  auto lstm_args = [&](int which, LstmPart part) {
    LstmArgs a;
    memset(&a, 0, sizeof(a));
    a.prec = prec;
    const int H = which ? Hd : Ha;
    const int k0 = which ? Ha : P;  // W_ih = [seg0 | ctx] columns
    const void *x0, *x0l, *x1, *x1l, *x2, *x2l;
    if (which == 0) {
      x0 = prec ? (const void*)sb.xpre_h : sb.xpre; x0l = prec ? (const void*)sb.xpre_l : sb.xpre;
      x2 = prec ? (const void*)sb.h_att_h[p] : sb.h_att[p]; x2l = prec ? (const void*)sb.h_att_l[p] : sb.h_att[p];
    } else {
      x0 = prec ? (const void*)sb.h_att_h[1 - p] : sb.h_att[1 - p]; x0l = prec ? (const void*)sb.h_att_l[1 - p] : sb.h_att[1 - p];
      x2 = prec ? (const void*)sb.h_dec_h[p] : sb.h_dec[p]; x2l = prec ? (const void*)sb.h_dec_l[p] : sb.h_dec[p];
    }
    x1 = prec ? (const void*)sb.ctx_h : sb.ctx; x1l = prec ? (const void*)sb.ctx_l : sb.ctx;
    const bool ck = prec && chunk_ok(d) && h->chunk_b;
    const size_t o_ih_h = which ? (ck ? bl.dec_ih_ch : bl.dec_ih_h) : (ck ? bl.att_ih_ch : bl.att_ih_h);
    const size_t o_ih_l = which ? (ck ? bl.dec_ih_cl : bl.dec_ih_l) : (ck ? bl.att_ih_cl : bl.att_ih_l);
    const size_t o_hh_h = which ? (ck ? bl.dec_hh_ch : bl.dec_hh_h) : (ck ? bl.att_hh_ch : bl.att_hh_h);
    const size_t o_hh_l = which ? (ck ? bl.dec_hh_cl : bl.dec_hh_l) : (ck ? bl.att_hh_cl : bl.att_hh_l);
    const f16 *wih_h = plane(o_ih_h), *wih_l = plane(o_ih_l), *whh_h = plane(o_hh_h), *whh_l = plane(o_hh_l);
    const float *wih = blob + (which ? bl.dec_ih : bl.att_ih), *whh = blob + (which ? bl.dec_hh : bl.att_hh);
    // column k0 of W_ih: row-major + k0 elements; chunked: chunk k0 / 32 of unit block 0 = k0 / 32 * (64 rows * 32) elements
    const size_t w1_off = ck ? (size_t)k0 * 64 : (size_t)k0;
    auto W0 = [&](bool lo) { return prec ? (const void*)(lo ? wih_l : wih_h) : (const void*)wih; };
    auto W1 = [&](bool lo) { return prec ? (const void*)((lo ? wih_l : wih_h) + w1_off) : (const void*)(wih + k0); };
    auto W2 = [&](bool lo) { return prec ? (const void*)(lo ? whh_l : whh_h) : (const void*)whh; };
    // leading "dimension" of a weight segment: row-major elements per row, or (chunked) the matrix's chunks per unit block
    const int wld_ih = ck ? (k0 + D) / kChunkK : k0 + D, wld_hh = ck ? H / kChunkK : H;
    if (part == PART_WHOLE) {
      a.a = act(make_seg3(x0, k0, k0, x1, D, D, x2, H, H)); a.a_lo = act(make_seg3(x0l, k0, k0, x1l, D, D, x2l, H, H));
      a.w = make_seg3(W0(false), wld_ih, k0, W1(false), wld_ih, D, W2(false), wld_hh, H);
      a.w_lo = make_seg3(W0(true), wld_ih, k0, W1(true), wld_ih, D, W2(true), wld_hh, H);
      a.K = k0 + D + H;
    } else if (part == PART_GATED && which == 0) {
      // [ctx_prev | h_att | x_pre]: x_pre is written by the frame role of the same launch
      a.a = act(make_seg3(x1, D, D, x2, H, H, x0, k0, k0)); a.a_lo = act(make_seg3(x1l, D, D, x2l, H, H, x0l, k0, k0));
      a.w = make_seg3(W1(false), wld_ih, D, W2(false), wld_hh, H, W0(false), wld_ih, k0);
      a.w_lo = make_seg3(W1(true), wld_ih, D, W2(true), wld_hh, H, W0(true), wld_ih, k0);
      a.K = k0 + D + H;
      a.dep_n = frame_grid_size(32, P); a.dep_seg = 2; a.dep_which = 0;  // (the frame workgroups of one 32-row block)
      a.dep_cnt = dep(DEP_FRAME);
      a.live_lag = 1;  // (same launch as the frame kernel: see lstm_body)
    } else if (part == PART_GATED) {
      // [h_dec | h_att | ctx]: ctx is written by the attention role of the same launch; h_att second, so that the one-launch
      // step (N_STEP), where it comes from the attention LSTM's tiles of the same launch, starts on the segment nobody is waited for
      a.a = act(make_seg3(x2, H, H, x0, k0, k0, x1, D, D)); a.a_lo = act(make_seg3(x2l, H, H, x0l, k0, k0, x1l, D, D));
      a.w = make_seg3(W2(false), wld_hh, H, W0(false), wld_ih, k0, W1(false), wld_ih, D);
      a.w_lo = make_seg3(W2(true), wld_hh, H, W0(true), wld_ih, k0, W1(true), wld_ih, D);
      a.K = k0 + D + H;
      a.dep_n = 32; a.dep_rows = 1; a.dep_seg = 2; a.dep_which = 1;  // (one attention workgroup per batch row)
      a.dep_cnt = dep(DEP_ATTN);
    } else if (which == 0 ? part == PART_EARLY : part == PART_LATE) {
      // [ctx (| h_att)]: the attention LSTM's early part, or the decoder LSTM's late part (ctx alone)
      const int kh = which == 0 ? H : 0;
      a.a = act(make_seg2(x1, D, D, x2, H, kh)); a.a_lo = act(make_seg2(x1l, D, D, x2l, H, kh));
      a.w = make_seg2(W1(false), wld_ih, D, W2(false), wld_hh, kh); a.w_lo = make_seg2(W1(true), wld_ih, D, W2(true), wld_hh, kh);
      a.K = D + kh;
    } else {
      // [seg0 (| h_dec)]: the attention LSTM's late part (x_pre alone), or the decoder LSTM's early part
      const int kh = which == 1 ? H : 0;
      a.a = act(make_seg2(x0, k0, k0, x2, H, kh)); a.a_lo = act(make_seg2(x0l, k0, k0, x2l, H, kh));
      a.w = make_seg2(W0(false), wld_ih, k0, W2(false), wld_hh, kh); a.w_lo = make_seg2(W0(true), wld_ih, k0, W2(true), wld_hh, kh);
      a.K = k0 + kh;
    }
    if (ck) { a.w.mpad = 1; a.w_lo.mpad = 1; }  // (flag: LoaderWLstm reads the chunked weight image)
    a.mode = part == PART_GATED ? 0 : (int)part;
    a.partial = which ? sb.pd : sb.pa;
    if (prec) { a.h_out_h = which ? sb.h_dec_h[1 - p] : sb.h_att_h[1 - p]; a.h_out_l = which ? sb.h_dec_l[1 - p] : sb.h_att_l[1 - p]; a.out_mpad = mpad; }
    a.bsum = blob + (which ? bl.dec_b : bl.att_b);
    a.h_prev = which ? sb.h_dec[p] : sb.h_att[p];
    a.c = which ? sb.c_dec : sb.c_att;
    a.h_out = which ? sb.h_dec[1 - p] : sb.h_att[1 - p];
    a.M = B; a.H = H; a.pz = d.p_zoneout; a.ctrl = ctrl; a.slot = io.slot; a.node = io.node_pos; a.dbg = io.dbg;
    a.tag = which;
    return a;
  };
  auto attn_args = [&]() {
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    if (prec) { a.ctx_h = sb.ctx_h; a.ctx_l = sb.ctx_l; a.out_mpad = mpad; }
    a.memory = io.memory; a.q = sb.q; a.q_parts = query_split(d); a.q_stride = (size_t)B * D;
    a.w_prev = sb.w[p]; a.w_new = sb.w[1 - p]; a.w_out = io.w; a.ctx = sb.ctx;
    a.B = B; a.L = io.L; a.D = D; a.t_rel = io.t_rel; a.t_stride = io.t_stride; a.ctrl = ctrl; a.slot = io.slot; a.node = io.node_pos;
    return a;
  };

  // the attention query GEMM (decoder_cell.py:188 -> attention.py:105) as a job of the attention role's workgroups
  auto query_role_args = [&]() {
    ProjArgs pq;
    memset(&pq, 0, sizeof(pq));
    pq.prec = prec ? PREC_F16S : PREC_F32;
    if (prec) {
      pq.a = act(make_seg1(sb.h_att_h[1 - p], Ha, Ha)); pq.a_lo = act(make_seg1(sb.h_att_l[1 - p], Ha, Ha));
      pq.W = plane(bl.wq_h); pq.W_lo = plane(bl.wq_l);
    } else {
      pq.a = make_seg1(sb.h_att[1 - p], Ha, Ha);
      pq.W = blob + bl.wq;
    }
    pq.ldw = query_ld(d); pq.M = B; pq.N = D; pq.K = query_k(d); pq.ksplit = proj_split(pq.K);
    pq.split_stride = (size_t)B * D; pq.out = sb.q; pq.ldo = D; pq.ctrl = ctrl; pq.slot = io.slot; pq.mode = PROJ_QUERY;
    pq.dep_cnt = dep(DEP_QUERY);
    return pq;
  };
  auto query_attn_args = [&](const ProjArgs& pq) {
    AttnArgs a = attn_args();
    a.dep_signal = 1; a.dep_cnt = dep(DEP_ATTN);
    a.q_parts = pq.ksplit;
    a.q_tiles = proj_grid_size(B, D, pq.ksplit); a.q_wait_n = proj_grid_size(32, D, pq.ksplit); a.q_cnt = pq.dep_cnt;
    return a;
  };
  // the mel/stop projection on the register-weight kernel, split-fp16 Prod cell: cat[h_dec, ctx] of buffer parity 1 - pp
  auto head_proj_args = [&](int pp, int mode) {
    ProjArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.a = act(make_seg2(sb.h_dec_h[1 - pp], Hd, Hd, sb.ctx_h, D, D));
    pa.a_lo = act(make_seg2(sb.h_dec_l[1 - pp], Hd, Hd, sb.ctx_l, D, D));
    pa.W = plane(bl.proj_h); pa.W_lo = plane(bl.proj_l); pa.ldw = proj_ld(d); pa.prec = PREC_F16S;
    pa.M = B; pa.N = proj_n(d); pa.K = proj_k(d); pa.ksplit = proj_parts(h, prec, B); pa.split_stride = (size_t)B * proj_ldp(d);
    pa.out = sb.jparts; pa.ldo = proj_ldp(d); pa.ctrl = ctrl; pa.slot = io.slot; pa.node = io.node_pos;
    pa.mode = mode;
    return pa;
  };

  switch (node) {
    case N_F:
    case N_FIN:
      launch_frame(frame_args(node == N_FIN), st);
      break;
    case N_FA: {
      FrameArgs f = frame_args(false);
      f.dep_signal = 1; f.dep_cnt = dep(DEP_FRAME);
      launch_frame_lstm(f, lstm_args(0, PART_GATED), st);
      break;
    }
    case N_TD: {
      AttnArgs a = attn_args();
      a.dep_signal = 1; a.dep_cnt = dep(DEP_ATTN);
      launch_attn_lstm(a, lstm_args(1, PART_GATED), nullptr, st);
      break;
    }
    case N_QTD: {
      const ProjArgs pq = query_role_args();
      launch_attn_lstm(query_attn_args(pq), lstm_args(1, PART_GATED), &pq, st);
      break;
    }
    case N_STEP: {
      // the whole step as one launch (fused_kernels.hip step_kernel): N_JFA's and N_QTD's roles, the boundary between the two
      // launches replaced by the DEP_HATT hand-off; every role live by t - 1 <= stop_t
      ProjArgs pa = head_proj_args(1 - p, PROJ_HEAD);
      FrameArgs f = frame_args(false);
      f.dep_signal = 1; f.dep_cnt = dep(DEP_FRAME);
      pa.dep_cnt = f.wait_cnt = dep(DEP_PROJ);
      f.wait_n = proj_grid_size(32, pa.N, pa.ksplit);
      LstmArgs la = lstm_args(0, PART_GATED);
      la.sig_cnt = dep(DEP_HATT);
      const int bu = B <= 64 ? 8 : 16;  // (units per LSTM tile: fused_kernels.hip Lean64x8 / Lean64x16)
      const int la_tiles = (Ha + bu - 1) / bu;  // attention-LSTM tiles per 32-row block and step
      ProjArgs pq = query_role_args();
      pq.wait_cnt = dep(DEP_HATT); pq.wait_n = la_tiles; pq.live_lag = 1;
      AttnArgs a = query_attn_args(pq);
      a.live_lag = 1;
      LstmArgs ld = lstm_args(1, PART_GATED);
      ld.live_lag = 1;
      ld.dep2_seg = 1; ld.dep2_n = la_tiles; ld.dep2_cnt = dep(DEP_HATT);
      const int tune = h->merged_tune > 0 ? h->merged_tune : 0;  // (measurement option)
      if (tune & 1) ld.dep2_seg = 0;  // the decoder LSTM starts nothing before its rows' h_att is complete
      f.dbg |= (tune & 0xff) << 8;
      pq.wait_sleep = (tune >> 8) & 0xff;
      launch_step_merged(pa, f, la, pq, a, ld, st);
      break;
    }
    case N_AG:
    case N_DG: {
      LstmArgs l = lstm_args(node == N_DG ? 1 : 0, PART_GATED);
      l.dep_n = 0;  // (alone: nothing to wait for)
      launch_lstm_lean(l, st);
      break;
    }
    case N_A: launch_lstm(lstm_args(0, PART_WHOLE), st); break;
    case N_D: launch_lstm(lstm_args(1, PART_WHOLE), st); break;
    case N_T: launch_attn(attn_args(), st); break;
    case N_P0:
    case N_P1: {
      const int layer = node == N_P0 ? 0 : 1;
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      const int Ph = pre_hidden(d);
      if (layer == 0) {
        g.a = make_seg1(sb.ynext, d.d_mel, d.d_mel);
        g.W = blob + bl.pre0_w; g.ldw = d.d_mel; g.K = d.d_mel; g.bias = blob + bl.pre0_b;
        g.out = sb.xpre0; g.N = Ph; g.ldo = Ph;
      } else {
        g.a = make_seg1(sb.xpre0, Ph, Ph);
        g.W = blob + bl.pre1_w; g.ldw = Ph; g.K = Ph; g.bias = blob + bl.pre1_b;
        g.out = sb.xpre; g.N = P; g.ldo = P;
        if (prec) { g.out_h = sb.xpre_h; g.out_l = sb.xpre_l; g.out_kind = 1; }
      }
      g.M = B;
      g.dropout_mode = io.dropout_mode;
      // masks of one step: layer 0 [B, Ph] then layer 1 [B, P]
      g.mask_step_stride = (size_t)B * (Ph + P);
      g.mask_layer_off = layer ? (size_t)B * Ph : 0;
      g.masks = io.masks ? io.masks + g.mask_layer_off : nullptr;
      g.seed = io.seed; g.layer = layer; g.keep_scale = keep_scale;
      g.r = d.r; g.d_mel = d.d_mel;
      g.ctrl = ctrl; g.slot = io.slot; g.node = io.node_pos; g.t = io.t;
      launch_gemm(g, A_PLAIN, EPI_RELU_DROPOUT, st);
      break;
    }
    case N_Q: {
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      // query_layer input: h_att (Prod, decoder_cell.py:188) or cat[h0, h1, zeros] (Taco2, :126)
      if (is_taco2(d)) g.a = make_seg2(sb.h_att[1 - p], Ha, Ha, sb.h_dec[1 - p], Hd, Hd);
      else g.a = make_seg1(sb.h_att[1 - p], Ha, Ha);
      g.W = blob + bl.wq; g.ldw = query_ld(d); g.K = query_k(d); g.M = B; g.N = D; g.out = sb.q; g.ldo = D;
      if (prec) {  // split-fp16 planes: the LSTM epilogues emitted them for h, the pack step for the weights
        g.prec = PREC_F16S;
        g.W = plane(bl.wq_h); g.W_lo = plane(bl.wq_l);
        if (is_taco2(d)) {
          g.a = act(make_seg2(sb.h_att_h[1 - p], Ha, Ha, sb.h_dec_h[1 - p], Hd, Hd));
          g.a_lo = act(make_seg2(sb.h_att_l[1 - p], Ha, Ha, sb.h_dec_l[1 - p], Hd, Hd));
        } else {
          g.a = act(make_seg1(sb.h_att_h[1 - p], Ha, Ha));
          g.a_lo = act(make_seg1(sb.h_att_l[1 - p], Ha, Ha));
        }
      }
      g.ksplit = query_split(d); g.kchunk = g.K / g.ksplit; g.split_stride = (size_t)B * D;
      g.ctrl = ctrl; g.slot = io.slot; g.node = io.node_pos; g.t = io.t;
      launch_gemm(g, A_PLAIN, EPI_PLAIN, st);
      break;
    }
    case N_JFA:
    case N_JF:
    case N_JFIN:
    case N_J: {
      // N_J: at the end of step t (buffer parity p).  N_JFA: the same GEMM for step t-1, at the head of step t's launch -
      // step t-1's outputs are this step's "previous" buffers.  N_JFIN: io.slot carries the last step's parity.
      const bool head = node == N_JFA || node == N_JF;
      const int p = head ? 1 - ((io.use_ctrl ? io.slot : io.t) & 1) : ((io.use_ctrl ? io.slot : io.t) & 1);
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      // projection input: cat[h_dec, ctx] (Prod, decoder_cell.py:192) or cat[h0, h1, zeros] (Taco2, :136)
      if (is_taco2(d)) g.a = make_seg2(sb.h_att[1 - p], Ha, Ha, sb.h_dec[1 - p], Hd, Hd);
      else g.a = make_seg2(sb.h_dec[1 - p], Hd, Hd, sb.ctx, D, D);
      g.W = blob + bl.proj_w; g.ldw = proj_ld(d); g.K = proj_k(d); g.M = B; g.N = proj_n(d);
      g.ctrl = ctrl; g.slot = io.slot; g.node = io.node_pos; g.t = io.t;
      if (prec && use_frame(d)) {
        g.prec = PREC_F16S;
        g.W = plane(bl.proj_h); g.W_lo = plane(bl.proj_l);
        if (is_taco2(d)) {
          g.a = act(make_seg2(sb.h_att_h[1 - p], Ha, Ha, sb.h_dec_h[1 - p], Hd, Hd));
          g.a_lo = act(make_seg2(sb.h_att_l[1 - p], Ha, Ha, sb.h_dec_l[1 - p], Hd, Hd));
        } else {
          g.a = act(make_seg2(sb.h_dec_h[1 - p], Hd, Hd, sb.ctx_h, D, D));
          g.a_lo = act(make_seg2(sb.h_dec_l[1 - p], Hd, Hd, sb.ctx_l, D, D));
        }
      }
      if (proj_regw(h, prec, B)) {  // (same slabs, from the kernel that keeps its weight fragments in registers)
        ProjArgs pa;
        memset(&pa, 0, sizeof(pa));
        pa.a = g.a; pa.a_lo = g.a_lo; pa.W = g.W; pa.W_lo = g.W_lo; pa.ldw = g.ldw; pa.prec = g.prec;
        pa.M = B; pa.N = g.N; pa.K = g.K; pa.ksplit = proj_parts(h, prec, B); pa.split_stride = (size_t)B * proj_ldp(d);
        pa.out = sb.jparts; pa.ldo = proj_ldp(d); pa.ctrl = ctrl; pa.slot = io.slot; pa.node = io.node_pos;
        pa.mode = head ? PROJ_HEAD : (node == N_JFIN ? PROJ_FINAL : PROJ_STEP);
        if (head) {
          FrameArgs f = frame_args(false);
          f.dep_signal = 1; f.dep_cnt = dep(DEP_FRAME);
          pa.dep_cnt = f.wait_cnt = dep(DEP_PROJ);
          f.wait_n = proj_grid_size(32, pa.N, pa.ksplit);  // (the projection workgroups of one 32-row block)
          if (node == N_JF) {  // no LSTM role behind the frame role: nobody to signal
            f.dep_signal = 0;
            launch_proj_frame(pa, f, st);
            break;
          }
          launch_proj_frame_lstm(pa, f, lstm_args(0, PART_GATED), st);
        } else {
          launch_proj(pa, st);
        }
        break;
      }
      if (use_frame(d)) {
        // raw split-K partial sums; bias, leaky-ReLU, y / s / stop rule happen in the next frame kernel
        g.out = sb.jparts; g.ldo = proj_ldp(d);
        g.ksplit = split_of(g.K, kProjSplit); g.kchunk = g.K / g.ksplit; g.split_stride = (size_t)B * proj_ldp(d);
        launch_gemm(g, A_PLAIN, EPI_PLAIN, st);
        break;
      }
      g.bias = blob + bl.proj_b;
      g.y_out = io.y; g.s_out = io.s; g.ynext = sb.ynext; g.r = d.r; g.d_mel = d.d_mel;
      g.t_rel = io.t_rel; g.t_stride = io.t_stride; g.stop_thr = 0.f; g.check_stop = 0;
      launch_gemm(g, A_PLAIN, EPI_PROJ, st);
      break;
    }
  }
}

const StepOrder kOrderProd = {7, {N_P0, N_P1, N_A, N_Q, N_T, N_D, N_J},   // decoder_cell.py:185-192
                              {"prenet0", "prenet1", "lstm_att", "query", "attention", "lstm_dec", "proj"}};
const StepOrder kOrderTaco2 = {7, {N_P0, N_P1, N_A, N_D, N_Q, N_T, N_J},  // decoder_cell.py:116-136
                               {"prenet0", "prenet1", "lstm_att", "lstm_dec", "query", "attention", "proj"}};
const StepOrder kOrderProdF = {6, {N_F, N_A, N_Q, N_T, N_D, N_J}, {"prenet", "lstm_att", "query", "attention", "lstm_dec", "proj"}};
const StepOrder kOrderTaco2F = {6, {N_F, N_A, N_D, N_Q, N_T, N_J}, {"prenet", "lstm_att", "lstm_dec", "query", "attention", "proj"}};
const StepOrder kOrderProdO = {5, {N_FA, N_Q, N_T, N_D, N_J}, {"prenet+lstm_att", "query", "attention", "lstm_dec", "proj"}};
// ... and where one role per launch stays the rule: the projection at the head of the frame kernel's launch
const StepOrder kOrderProdHF = {5, {N_JF, N_A, N_Q, N_T, N_D}, {"proj+prenet", "lstm_att", "query", "attention", "lstm_dec"}};
const StepOrder kOrderTaco2HF = {5, {N_JF, N_A, N_D, N_Q, N_T}, {"proj+prenet", "lstm_att", "lstm_dec", "query", "attention"}};
const StepOrder kOrderTaco2O = {5, {N_FA, N_D, N_Q, N_T, N_J}, {"prenet+lstm_att", "lstm_dec", "query", "attention", "proj"}};
const StepOrder kOrderTaco2H = {4, {N_JFA, N_D, N_Q, N_T}, {"proj+prenet+lstm_att", "lstm_dec", "query", "attention"}};
const StepOrder kOrderProdO2 = {4, {N_FA, N_Q, N_TD, N_J}, {"prenet+lstm_att", "query", "attention+lstm_dec", "proj"}};
// ... with step t-1's projection at the head of step t's first launch
const StepOrder kOrderProdH = {4, {N_JFA, N_Q, N_T, N_D}, {"proj+prenet+lstm_att", "query", "attention", "lstm_dec"}};
const StepOrder kOrderProdH2 = {3, {N_JFA, N_Q, N_TD}, {"proj+prenet+lstm_att", "query", "attention+lstm_dec"}};
// ... and with the query as a job of the attention role's workgroups: two launches per step
const StepOrder kOrderProdO2Q = {3, {N_FA, N_QTD, N_J}, {"prenet+lstm_att", "query+attention+lstm_dec", "proj"}};
const StepOrder kOrderProdH2Q = {2, {N_JFA, N_QTD}, {"proj+prenet+lstm_att", "query+attention+lstm_dec"}};
// ... and the boundary between those two replaced by a hand-off: one launch per step
const StepOrder kOrderProdS = {1, {N_STEP}, {"step"}};
// the two-role step: LJSpeech-type cell, either arithmetic mode
int overlap_level(const ttsdec_handle* h, int B) {
  const ttsdec_dims& d = h->d;
  if (!fused_supported(d.d_mel, d.r, pre_hidden(d), d.d_pre, d.d_ctx)) return 0;
  if (is_taco2(d)) {
    // Taco2DecoderCell (decoder_cell.py:116-136): its attention pass FOLLOWS both LSTMs, so only the first two-role launch
    // applies - frame || lstm_att, with the projection (which reads the two LSTM states there, not the context) as its head
    // role.  rdh config, us per step without / with (profiles/r03_w_taco2_level1.txt): split-fp16 B = 256 92.3 / 80.7, B = 64
    // 60.5 / 52.8, B = 1 54.6 / 46.9; exact fp32 B = 256 147.2 / 137.1, B = 64 73.0 / 80.6, B = 1 69.3 / 59.3 - the fp32 rule of the
    // other cell: not in the middle batch range.
    if (h->overlap >= 0) return h->overlap > 1 ? 1 : h->overlap;
    // (round 4, with the 32-row fp32 lean tile: exact fp32 B = 64 69.9 / 62.9, B = 128 86.9 / 82.7 - level 1 at every batch size now)
    return 1;
  }
  if (h->overlap >= 0) return h->overlap;
  // Round 3 (after the sc1 hand-offs and the per-row-block counters; profiles/r03_t_levels_sweep.txt, us per step, levels 0 / 1 / 2):
  // split-fp16 B = 96 78.3 / 62.4 / 48.8, 192 80.1 / 64.8 / 59.3, 320 128.5 / 102.5 / 88.8, 384 131.1 / 110.0 / 98.3, 512 147.7 / 131.6 /
  // 117.8, 1024 - / 262.6 / 235.0, 2048 - / 504.1 / 455.8: level 2 at every batch size (round 2 had level 1 above 320, +-1.5 % then).
  if (lstm_prec(h)) return 2;
  // exact fp32 (round 4, after the loaders' buffer-descriptor DMA, the interleaved fragment reads and the 32-row lean tile whose
  // four MFMA waves share a block's K axis - fused_kernels.hip): us per step for levels 0 / 1 / 2: B = 32 68.7 / 60.4 / 50.5,
  // 40 68.6 / 63.4 / 60.0, 64 68.5 / 62.8 / 61.2, 80 96.3 / 79.4 / 65.4, 96 84.8 / 74.4 / 66.5, 112 86.0 / 81.5 / 91.2, 128 84.8 / 79.4 /
  // 88.0, 144 121.3 / 105.7 / 97.5, 176 121.6 / 106.7 / 98.3, 192 122.4 / 107.6 / 99.1, 256 - / - / 102.7 (round 3: level 0 between 32
  // and 192 utterances - its 64-row lean tile kept two of a CU's four matrix pipes busy there: B = 64 77.1 / 87.2 / 111.4).
  // Between 97 and 128 utterances the 64 x 16-unit tile has only two row blocks (128 workgroups): one two-role launch.
  return B <= 96 ? 2 : (B <= 128 ? 1 : 2);
}
// The projection as the head role of the next step's frame launch, wherever the register-weight kernel applies (split-fp16).
// us per step without / with it, same box: B = 1 43.0 / 39.8, B = 64 51.8 / 50.1, B = 128 54.2 / 53.5, B = 256 73.4 / 73.3 (there
// the frame role's chain is what the launch waits for, and it grows by what the projection's own launch cost).
// Round 3: also where the step runs one role per launch (N_JF: [proj | frame]) - us per step without / with it
// (profiles/r03_w_head_proj_level0.txt): exact fp32 B = 64 76.5 / 74.7, B = 128 103.0 / 100.9; sandra config (two frames per
// step) 63.4 / 63.3, B = 1 44.3 / 44.1 - the hop costs there what the launch did.
bool head_proj(const ttsdec_handle* h, int B) {
  (void)B;
  return proj_regw_shapes(h) && h->head_proj != 0;  // (-1 = default = on)
}
// The query as a job of the attention role (fused_kernels.hip attn_lstm_kernel): needs every attention-role workgroup resident
// at once - one per utterance, and they wait for each other's query tiles; they have the launch's lowest block ids, so that is
// at most what the device holds of that kernel at once (attn_lstm_resident_slots: CUs x the runtime's occupancy answer, 512 on
// an MI355X; a partitioned or smaller part answers for itself; a GPU shared with another process is what the bounded spin and
// Decoder.forward's fallback are for) - and whole K slices for the register-weight GEMM body.
// Measured, us per step with / without it (same box, profiles/r03_f_*, r03_t_query_role_large_batches.txt): split-fp16 B = 256 60.4 /
// 61.6, 128 49.4 / 50.0, 64 46.6 / 47.0, 320 83.9 / 89.7, 384 98.5 / 99.0, 448 108.5 / 106.9, 512 121.5 / 118.7; B = 32 44.1 / 42.5 (the
// roles do not share CUs there: the hop costs more than the launch), B = 1 175 / 38 (one workgroup would run all 32 tiles); exact
// fp32 B = 256 125.0 / 125.0, B = 32 58.6 / 55.3.  So: split-fp16, 64..384 utterances, at most one tile per workgroup (option
// query_role = 1 forces it wherever it is possible at all).
bool query_role(const ttsdec_handle* h, int B) {
  const ttsdec_dims& d = h->d;
  if (overlap_level(h, B) < 2 || h->query_role == 0) return false;
  if (B > attn_lstm_resident_slots(B, d.h_dec, d.d_ctx, lstm_prec(h) != 0)) return false;
  const int ps = proj_split(query_k(d));
  if (!(ps > 0 && ps <= kQuerySplit && !(d.h_att & 7))) return false;
  if (h->query_role > 0) return true;
  return lstm_prec(h) && B >= 64 && B <= 384 && proj_grid_size(B, d.d_ctx, ps) <= B;
}
// The one-launch step (option overlap = 3): split-fp16 Prod cell with the projection as a head role and the query on the
// register-weight body, at batches whose attention workgroups fit the chip at once.
bool step_merged(const ttsdec_handle* h, int B) {
  const ttsdec_dims& d = h->d;
  if (overlap_level(h, B) < 3 || !lstm_prec(h) || !head_proj(h, B)) return false;
  const int ps = proj_split(query_k(d));  // (the query on the register-weight body: whole K slices)
  if (!(ps > 0 && ps <= kQuerySplit && !(d.h_att & 7))) return false;
  return step_merged_supported(B, d.h_att, d.h_dec, pre_hidden(d), d.d_pre, d.d_ctx, proj_n(d), proj_parts(h, lstm_prec(h), B));
}
const StepOrder& step_order(const ttsdec_handle* h, int B) {
  const ttsdec_dims& d = h->d;
  if (const int lv = overlap_level(h, B)) {
    if (is_taco2(d)) return head_proj(h, B) ? kOrderTaco2H : kOrderTaco2O;
    if (step_merged(h, B)) return kOrderProdS;
    if (lv >= 2 && query_role(h, B)) return head_proj(h, B) ? kOrderProdH2Q : kOrderProdO2Q;
    if (head_proj(h, B)) return lv >= 2 ? kOrderProdH2 : kOrderProdH;
    return lv >= 2 ? kOrderProdO2 : kOrderProdO;
  }
  if (use_frame(d) && head_proj(h, B)) return is_taco2(d) ? kOrderTaco2HF : kOrderProdHF;
  if (use_frame(d)) return is_taco2(d) ? kOrderTaco2F : kOrderProdF;
  return is_taco2(d) ? kOrderTaco2 : kOrderProd;
}

// One step on a single stream, in the reference's order.  (For the Taco2 cell the attention
// kernel at the end of step t also produces the context bmm(w_t, memory) that step t+1 starts
// from, decoder_cell.py:118.)
void launch_step_serial(const ttsdec_handle* h, const StepBufs& sb, const StepIo& io0, hipStream_t st) {
  const StepOrder& order = step_order(h, io0.B);
  StepIo io = io0;
  for (int i = 0; i < order.n; ++i) {
    io.node_pos = i;
    launch_node(h, sb, io, order.nodes[i], st);
  }
}

int ensure_streams(ttsdec_handle* h) {
  if (h->streams_ready) return TTSDEC_OK;
  if (hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking) != hipSuccess) return TTSDEC_ERR_HIP;
  h->streams_ready = true;
  return TTSDEC_OK;
}

void drop_graph(ttsdec_handle* h) {
  if (h->gexec) { (void)hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
  if (h->graph) { (void)hipGraphDestroy(h->graph); h->graph = nullptr; }
}

// Captures kGraphSlots steps (+ the t_cur advance) once per (workspace, blob, B, L, mode).
int ensure_graph(ttsdec_handle* h, const StepBufs& sb, const void* ws, int B, int L) {
  const int prec = lstm_prec(h);
  if (h->gexec && h->g_ws == ws && h->g_blob == h->blob && h->g_B == B && h->g_L == L && h->g_prec == prec) return TTSDEC_OK;
  drop_graph(h);
  StepIo io;
  memset(&io, 0, sizeof(io));
  io.B = B; io.L = L; io.use_ctrl = true;
  if (hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeRelaxed) != hipSuccess) return TTSDEC_ERR_HIP;
  int rc = TTSDEC_OK;
  for (int i = 0; i < kGraphSlots; ++i) {
    io.slot = i;
    launch_step_serial(h, sb, io, h->cap_stream);
  }
  launch_advance(sb.ctrl, kGraphSlots, h->cap_stream);
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(h->cap_stream, &graph);
  if (e != hipSuccess || rc != TTSDEC_OK || graph == nullptr) {
    if (graph) (void)hipGraphDestroy(graph);
    h->hip_err = std::string("graph capture: ") + hipGetErrorString(e);
    (void)hipGetLastError();
    return TTSDEC_ERR_HIP;
  }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(graph);
    h->hip_err = std::string("hipGraphInstantiate: ") + hipGetErrorString(e);
    return TTSDEC_ERR_HIP;
  }
  h->graph = graph; h->gexec = exec;
  h->g_ws = ws; h->g_blob = h->blob; h->g_B = B; h->g_L = L; h->g_prec = prec;
  return TTSDEC_OK;
}


// ---- options (include/ttsdec.h TTSDEC_OPT_*) ----
const char* const kOptionNames[TTSDEC_OPT_COUNT] = {"graph",     "overlap",    "chunk_a",          "chunk_b",     "proj_regw",
                                                    "head_proj", "query_role", "merged_tune", "profile_ablation", "debug_flags", "spin_limit"};
int& option_ref(ttsdec_handle* h, int o) {
  switch (o) {
    case TTSDEC_OPT_OVERLAP: return h->overlap;
    case TTSDEC_OPT_CHUNK_A: return h->opt_chunk_a;
    case TTSDEC_OPT_CHUNK_B: return h->opt_chunk_b;
    case TTSDEC_OPT_PROJ_REGW: return h->opt_proj_regw;
    case TTSDEC_OPT_HEAD_PROJ: return h->head_proj;
    case TTSDEC_OPT_QUERY_ROLE: return h->query_role;
    case TTSDEC_OPT_MERGED_TUNE: return h->merged_tune;
    case TTSDEC_OPT_PROFILE_ABLATION: return h->profile_ablation;
    case TTSDEC_OPT_DEBUG_FLAGS: return h->debug_flags;
    case TTSDEC_OPT_SPIN_LIMIT: return h->spin_limit;
    default: return h->opt_graph;
  }
}
// the derived switches the launch code reads (-1 = library default)
void refresh_options(ttsdec_handle* h) {
  h->use_graph = h->opt_graph != 0;
  h->chunk_a = h->opt_chunk_a != 0;
  h->chunk_b = h->opt_chunk_b != 0;
  h->proj_regw = h->opt_proj_regw != 0;
  drop_graph(h);  // a captured graph bakes the launch sequence in
}
void apply_env_options(ttsdec_handle* h) {
  if (const char* e = getenv("TTSDEC_OPTIONS")) {
    std::string str(e);
    size_t pos = 0;
    while (pos < str.size()) {
      size_t end = str.find(',', pos);
      if (end == std::string::npos) end = str.size();
      const std::string item = str.substr(pos, end - pos);
      const size_t eq = item.find('=');
      if (eq != std::string::npos) {
        const std::string name = item.substr(0, eq);
        for (int o = 0; o < TTSDEC_OPT_COUNT; ++o)
          if (name == kOptionNames[o]) {
            const long v = strtol(item.c_str() + eq + 1, nullptr, 10);  // (saturates; the string comes from the environment)
            option_ref(h, o) = v > 0x7fffffffL ? 0x7fffffff : (v < -0x7fffffffL ? -0x7fffffff : (int)v);
          }
      }
      pos = end + 1;
    }
  }
  refresh_options(h);
}
}  // namespace

extern "C" {

int ttsdec_version(void) { return TTSDEC_VERSION; }

const char* ttsdec_strerror(int code) {
  switch (code) {
    case TTSDEC_OK: return "ok";
    case TTSDEC_ERR_INVALID_ARG: return "invalid argument";
    case TTSDEC_ERR_DIMS: return "unsupported dimensions (feature sizes must be positive multiples of 4)";
    case TTSDEC_ERR_HIP: return "HIP runtime error";
    case TTSDEC_ERR_NOT_BOUND: return "no packed weights bound to the handle";
    case TTSDEC_ERR_WORKSPACE: return "workspace too small or not 256-byte aligned";
    case TTSDEC_ERR_DEVICE: return "no usable HIP device, or the current device is not the handle's";
    default: return "unknown error";
  }
}

const char* ttsdec_last_hip_error(const ttsdec_handle* h) { return h ? h->hip_err.c_str() : ""; }

int ttsdec_create(const ttsdec_dims* dims, ttsdec_handle** out) {
  if (!dims || !out) return TTSDEC_ERR_INVALID_ARG;
  *out = nullptr;
  const int rc = check_dims(*dims);
  if (rc != TTSDEC_OK) return rc;
  ttsdec_handle* h = new (std::nothrow) ttsdec_handle();
  if (!h) return TTSDEC_ERR_INVALID_ARG;
  h->d = *dims;
  h->precision = TTSDEC_PREC_F32;
  h->bl = make_blob_layout(*dims);
  h->blob = nullptr;
  h->streams_ready = false;
  h->gexec = nullptr;
  h->graph = nullptr;
  h->g_ws = h->g_blob = nullptr;
  h->wmax_dec = h->wmax_post = 0.f;
  h->stamps_buf = nullptr;
  // Tuning / measurement options (include/ttsdec.h TTSDEC_OPT_*): library defaults, then the process-wide
  // TTSDEC_OPTIONS="name=value,..." (read here, once per handle), then ttsdec_set_option.
  // (A two-stream schedule that ran the LSTMs' early K segments beside the small critical-path kernels was built and
  // measured SLOWER on MI355X - 126 vs 96 us per step at B=256: the early GEMM's 256 workgroups hold every CU's LDS, so
  // the small kernels queue behind them - and was removed; see DESIGN.md.)
  for (int o = 0; o < TTSDEC_OPT_COUNT; ++o) option_ref(h, o) = -1;
  h->profile_ablation = 0; h->debug_flags = 0; h->spin_limit = 0; h->merged_tune = 0;
  apply_env_options(h);
  h->device = current_device_or_minus1();
  *out = h;
  return TTSDEC_OK;
}

int ttsdec_destroy(ttsdec_handle* h) {
  if (!h) return TTSDEC_OK;
  drop_graph(h);
  if (h->streams_ready) (void)hipStreamDestroy(h->cap_stream);
  if (h->stamps_buf) (void)hipFree(h->stamps_buf);
  delete h;
  return TTSDEC_OK;
}

int ttsdec_num_weight_tensors(const ttsdec_handle* h) {
  if (!h) return TTSDEC_ERR_INVALID_ARG;
  if (h->d.postnet_layers <= 0) return TTSDEC_W_DECODER_COUNT;
  if (h->d.postnet_type == TTSDEC_POSTNET_TYPE_MEL2) return TTSDEC_W_DECODER_COUNT + TTSDEC_W_POSTNET2_PER_LAYER * h->d.postnet_layers;
  return TTSDEC_W_DECODER_COUNT + TTSDEC_W_POSTNET_PER_LAYER * h->d.postnet_layers + 1;
}

int ttsdec_set_precision(ttsdec_handle* h, int precision) {
  if (!h || (precision != TTSDEC_PREC_F32 && precision != TTSDEC_PREC_SPLIT_F16)) return TTSDEC_ERR_INVALID_ARG;
  h->precision = precision;
  return TTSDEC_OK;
}

int ttsdec_get_precision(const ttsdec_handle* h) {
  if (!h) return TTSDEC_ERR_INVALID_ARG;
  return lstm_prec(h) ? TTSDEC_PREC_SPLIT_F16 : TTSDEC_PREC_F32;
}

int ttsdec_set_option(ttsdec_handle* h, int option, int value) {
  if (!h || option < 0 || option >= TTSDEC_OPT_COUNT) return TTSDEC_ERR_INVALID_ARG;
  option_ref(h, option) = value;
  refresh_options(h);
  return TTSDEC_OK;
}

int ttsdec_get_option(const ttsdec_handle* h, int option, int* value) {
  if (!h || !value || option < 0 || option >= TTSDEC_OPT_COUNT) return TTSDEC_ERR_INVALID_ARG;
  *value = option_ref(const_cast<ttsdec_handle*>(h), option);
  return TTSDEC_OK;
}

const char* ttsdec_option_name(int option) { return (option >= 0 && option < TTSDEC_OPT_COUNT) ? kOptionNames[option] : nullptr; }

size_t ttsdec_packed_bytes(const ttsdec_handle* h) { return h ? h->bl.total * sizeof(float) : 0; }

int ttsdec_pack_weights(ttsdec_handle* h, const float* const* src, int n_src, void* blob, void* stream) {
  if (!h || !src || !blob) return TTSDEC_ERR_INVALID_ARG;
  if (n_src != ttsdec_num_weight_tensors(h)) return TTSDEC_ERR_INVALID_ARG;
  // a NULL entry leaves that tensor's region zero (a module that owns only part of the
  // parameters - a bare decoder cell, a postnet - packs what it has)
  if (reinterpret_cast<uintptr_t>(blob) & 255) return TTSDEC_ERR_WORKSPACE;
  int rc = check_device(h);
  if (rc != TTSDEC_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const ttsdec_dims& d = h->d;
  const BlobLayout& L = h->bl;
  float* b = static_cast<float*>(blob);
  const size_t Ha = d.h_att, Hd = d.h_dec, D = d.d_ctx, P = d.d_pre, Mel = d.d_mel, R = d.r, Ph = pre_hidden(d);
  const size_t Dp = proj_ld(d);
  HIP_TRY(h, hipMemsetAsync(blob, 0, L.total * sizeof(float), st));
  launch_copy(src[TTSDEC_W_PRE0_W], b + L.pre0_w, Ph * Mel, st);
  launch_copy(src[TTSDEC_W_PRE0_B], b + L.pre0_b, Ph, st);
  launch_copy(src[TTSDEC_W_PRE1_W], b + L.pre1_w, P * Ph, st);
  launch_copy(src[TTSDEC_W_PRE1_B], b + L.pre1_b, P, st);
  launch_copy(src[TTSDEC_W_QUERY_W], b + L.wq, D * (size_t)query_ld(d), st);
  launch_copy(src[TTSDEC_W_ATT_IH], b + L.att_ih, 4 * Ha * (P + D), st);
  launch_copy(src[TTSDEC_W_ATT_HH], b + L.att_hh, 4 * Ha * Ha, st);
  if (src[TTSDEC_W_ATT_BIH] && src[TTSDEC_W_ATT_BHH])
    launch_add_vec(src[TTSDEC_W_ATT_BIH], src[TTSDEC_W_ATT_BHH], b + L.att_b, (int)(4 * Ha), st);
  launch_copy(src[TTSDEC_W_DEC_IH], b + L.dec_ih, 4 * Hd * (Ha + D), st);
  launch_copy(src[TTSDEC_W_DEC_HH], b + L.dec_hh, 4 * Hd * Hd, st);
  if (src[TTSDEC_W_DEC_BIH] && src[TTSDEC_W_DEC_BHH])
    launch_add_vec(src[TTSDEC_W_DEC_BIH], src[TTSDEC_W_DEC_BHH], b + L.dec_b, (int)(4 * Hd), st);
  auto hp = [&](size_t float_off) { return reinterpret_cast<f16*>(b + float_off); };
  launch_split(src[TTSDEC_W_PRE0_W], hp(L.pre0_h), hp(L.pre0_l), Ph * Mel, st);
  launch_split(src[TTSDEC_W_PRE1_W], hp(L.pre1_h), hp(L.pre1_l), P * Ph, st);
  if (use_frame(d)) {  // (the frame kernel's shapes: frame_supported - d_mel = 80, Ph = 128 or 256)
    launch_split_frame_order(src[TTSDEC_W_PRE0_W], hp(L.pre0_ph), hp(L.pre0_pl), (int)Ph, (int)Mel, 0, st);
    launch_split_frame_order(src[TTSDEC_W_PRE1_W], hp(L.pre1_ph), hp(L.pre1_pl), (int)P, (int)Ph, 1, st);
  }
  launch_split(src[TTSDEC_W_ATT_IH], hp(L.att_ih_h), hp(L.att_ih_l), 4 * Ha * (P + D), st);
  launch_split(src[TTSDEC_W_ATT_HH], hp(L.att_hh_h), hp(L.att_hh_l), 4 * Ha * Ha, st);
  launch_split(src[TTSDEC_W_DEC_IH], hp(L.dec_ih_h), hp(L.dec_ih_l), 4 * Hd * (Ha + D), st);
  launch_split(src[TTSDEC_W_DEC_HH], hp(L.dec_hh_h), hp(L.dec_hh_l), 4 * Hd * Hd, st);
  if (chunk_ok(d)) {
    launch_pack_lstm_chunked(src[TTSDEC_W_ATT_IH], hp(L.att_ih_ch), hp(L.att_ih_cl), (int)Ha, (int)(P + D), st);
    launch_pack_lstm_chunked(src[TTSDEC_W_ATT_HH], hp(L.att_hh_ch), hp(L.att_hh_cl), (int)Ha, (int)Ha, st);
    launch_pack_lstm_chunked(src[TTSDEC_W_DEC_IH], hp(L.dec_ih_ch), hp(L.dec_ih_cl), (int)Hd, (int)(Ha + D), st);
    launch_pack_lstm_chunked(src[TTSDEC_W_DEC_HH], hp(L.dec_hh_ch), hp(L.dec_hh_cl), (int)Hd, (int)Hd, st);
  }
  launch_copy(src[TTSDEC_W_INIT_H0], b + L.h0a, Ha, st);
  launch_copy(src[TTSDEC_W_INIT_C0], b + L.c0a, Ha, st);
  launch_copy(src[TTSDEC_W_INIT_H1], b + L.h0d, Hd, st);
  launch_copy(src[TTSDEC_W_INIT_C1], b + L.c0d, Hd, st);
  // [fc_mel ; fc_stop] stacked on the output axis (decoder.py:13-14)
  launch_copy(src[TTSDEC_W_MEL_W], b + L.proj_w, R * Mel * Dp, st);
  launch_copy(src[TTSDEC_W_STOP_W], b + L.proj_w + R * Mel * Dp, R * Dp, st);
  launch_copy(src[TTSDEC_W_MEL_B], b + L.proj_b, R * Mel, st);
  launch_copy(src[TTSDEC_W_STOP_B], b + L.proj_b + R * Mel, R, st);
  launch_split(src[TTSDEC_W_QUERY_W], hp(L.wq_h), hp(L.wq_l), D * (size_t)query_ld(d), st);
  if (src[TTSDEC_W_MEL_W] && src[TTSDEC_W_STOP_W])
    launch_split(b + L.proj_w, hp(L.proj_h), hp(L.proj_l), (R * Mel + R) * Dp, st);
  int cin = d.d_mel;
  if (d.postnet_type == TTSDEC_POSTNET_TYPE_MEL2) {
    const int Hh = d.postnet_hidden, k = d.postnet_kernel;
    const int shapes[3][2] = {{Hh, (int)Mel}, {Hh, Hh}, {(int)Mel, Hh}};
    for (int i = 0; i < d.postnet_layers; ++i) {
      // per layer: conv1.w, bn1.{w,b,mean,var}, conv2.w, bn2.{w,b,mean,var}, conv3.w
      const float* const* ps = src + TTSDEC_W_DECODER_COUNT + TTSDEC_W_POSTNET2_PER_LAYER * i;
      const int conv_idx[3] = {0, 5, 10};
      for (int c = 0; c < 3; ++c) {
        if (!ps[conv_idx[c]]) return TTSDEC_ERR_INVALID_ARG;
        const size_t n = (size_t)shapes[c][0] * shapes[c][1] * k;
        launch_conv1dfix_pack(ps[conv_idx[c]], b + L.p2_w[i][c], shapes[c][0], shapes[c][1], k, st);
        launch_split(b + L.p2_w[i][c], hp(L.p2_wh[i][c]), hp(L.p2_wl[i][c]), n, st);
        launch_to_bf16(b + L.p2_w[i][c], b + L.p2_wb[i][c], n, st);
      }
      for (int c = 0; c < 2; ++c) {
        const float* const* bn = ps + 1 + 5 * c;
        if (!bn[0] || !bn[1] || !bn[2] || !bn[3]) return TTSDEC_ERR_INVALID_ARG;
        launch_bn_fold(bn[0], bn[1], bn[2], bn[3], d.bn_eps, b + L.p2_alpha[i][c], b + L.p2_beta[i][c], Hh, st);
      }
    }
  } else
  for (int i = 0; i < d.postnet_layers; ++i) {
    const float* const* ps = src + TTSDEC_W_DECODER_COUNT + TTSDEC_W_POSTNET_PER_LAYER * i;
    if (!ps[0] || !ps[1] || !ps[2] || !ps[3] || !ps[4]) return TTSDEC_ERR_INVALID_ARG;
    launch_conv_transpose(ps[0], b + L.conv_w[i], d.postnet_hidden, cin, d.postnet_kernel, st);
    {
      const size_t n = (size_t)d.postnet_hidden * d.postnet_kernel * cin;
      launch_split(b + L.conv_w[i], hp(L.conv_wh[i]), hp(L.conv_wl[i]), n, st);
      launch_to_bf16(b + L.conv_w[i], b + L.conv_wb[i], n, st);
    }
    launch_bn_fold(ps[1], ps[2], ps[3], ps[4], d.bn_eps, b + L.conv_alpha[i], b + L.conv_beta[i], d.postnet_hidden, st);
    cin = d.postnet_hidden;
  }
  if (d.postnet_layers > 0 && d.postnet_type == TTSDEC_POSTNET_TYPE_MEL) {
    const float* fc = src[TTSDEC_W_DECODER_COUNT + TTSDEC_W_POSTNET_PER_LAYER * d.postnet_layers];
    const size_t n = Mel * (size_t)d.postnet_hidden;
    launch_copy(fc, b + L.fc_w, n, st);
    launch_split(fc, hp(L.fc_wh), hp(L.fc_wl), n, st);
    launch_to_bf16(fc, b + L.fc_wb, n, st);
  }
  // range guard of the split-fp16 planes: max |w| of every matrix that has them
  {
    float* wm = b + L.wmax;
    launch_absmax(src[TTSDEC_W_PRE0_W], Ph * Mel, wm, st);
    launch_absmax(src[TTSDEC_W_PRE1_W], P * Ph, wm, st);
    launch_absmax(src[TTSDEC_W_ATT_IH], 4 * Ha * (P + D), wm, st);
    launch_absmax(src[TTSDEC_W_ATT_HH], 4 * Ha * Ha, wm, st);
    launch_absmax(src[TTSDEC_W_DEC_IH], 4 * Hd * (Ha + D), wm, st);
    launch_absmax(src[TTSDEC_W_DEC_HH], 4 * Hd * Hd, wm, st);
    launch_absmax(src[TTSDEC_W_QUERY_W], D * (size_t)query_ld(d), wm, st);
    launch_absmax(b + L.proj_w, (R * Mel + R) * Dp, wm, st);
    if (d.postnet_type == TTSDEC_POSTNET_TYPE_MEL2) {
      const size_t Hh = d.postnet_hidden, k = d.postnet_kernel;
      const size_t n3[3] = {Hh * Mel * k, Hh * Hh * k, Mel * Hh * k};
      for (int i = 0; i < d.postnet_layers; ++i)
        for (int c = 0; c < 3; ++c) launch_absmax(b + L.p2_w[i][c], n3[c], wm + 1, st);
    } else {
      size_t ci = Mel;
      for (int i = 0; i < d.postnet_layers; ++i) {
        launch_absmax(b + L.conv_w[i], (size_t)d.postnet_hidden * d.postnet_kernel * ci, wm + 1, st);
        ci = d.postnet_hidden;
      }
      if (d.postnet_layers > 0) launch_absmax(b + L.fc_w, Mel * (size_t)d.postnet_hidden, wm + 1, st);
    }
  }
  rc = check_launch(h, "pack_weights");
  if (rc != TTSDEC_OK) return rc;
  h->blob = b;
  return read_wmax(h, st);
}

int ttsdec_bind_weights(ttsdec_handle* h, const void* blob) {
  if (!h || !blob) return TTSDEC_ERR_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(blob) & 255) return TTSDEC_ERR_WORKSPACE;
  const int rc = check_device(h);
  if (rc != TTSDEC_OK) return rc;
  h->blob = static_cast<const float*>(blob);
  return read_wmax(h, nullptr);  // (synchronises the null stream: the blob must be complete, e.g. the broadcast done)
}

size_t ttsdec_workspace_bytes(const ttsdec_handle* h, int B, int L) {
  if (!h || B <= 0 || L <= 0) return 0;
  return make_ws_layout(h->d, B, L).total;
}

// stamp_loop (ttsdec_profile_loop): every step kernel records its start time (common.h loop_stamp) in the workspace
static int decode_impl(ttsdec_handle* h, const float* memory, int B, int L, int t_begin, int n_steps, int t_stride,
                       float stop_threshold, int check_stop, int dropout_mode, const uint8_t* masks, uint64_t seed,
                       const float* teacher, int teacher_T, const uint8_t* teacher_flags, float* y, float* s, float* w,
                       int32_t* T_out, void* workspace, size_t workspace_bytes, void* stream, bool stamp_loop) {
  if (!h || !memory || !y || !s || !w || !workspace) return TTSDEC_ERR_INVALID_ARG;
  if (B <= 0 || L <= 0 || t_begin < 0 || n_steps < 0 || t_stride < n_steps) return TTSDEC_ERR_INVALID_ARG;
  if (dropout_mode < TTSDEC_DROPOUT_OFF || dropout_mode > TTSDEC_DROPOUT_PHILOX) return TTSDEC_ERR_INVALID_ARG;
  if (dropout_mode == TTSDEC_DROPOUT_MASKS && !masks) return TTSDEC_ERR_INVALID_ARG;
  if (dropout_mode == TTSDEC_DROPOUT_PHILOX && h->d.p_dropout != 0.5f) return TTSDEC_ERR_INVALID_ARG;  // one keep BIT per unit
  if (teacher && (!teacher_flags || teacher_T < (t_begin + n_steps - 1) * h->d.r)) return TTSDEC_ERR_INVALID_ARG;
  if (!h->blob) return TTSDEC_ERR_NOT_BOUND;
  int rc = check_device(h);
  if (rc != TTSDEC_OK) return rc;
  const WsLayout W = make_ws_layout(h->d, B, L);
  if (workspace_bytes < W.total || (reinterpret_cast<uintptr_t>(workspace) & 255)) return TTSDEC_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const StepBufs sb = carve(W, workspace);
  const ttsdec_dims& d = h->d;

  if (t_begin & 1) return TTSDEC_ERR_INVALID_ARG;  // buffer parity is tied to the step index
  rc = ensure_streams(h);
  if (rc != TTSDEC_OK) return rc;

  if (t_begin == 0) {
    InitArgs ia;
    ia.ctrl = sb.ctrl;
    ia.h0_att = h->blob + h->bl.h0a; ia.c0_att = h->blob + h->bl.c0a;
    ia.h0_dec = h->blob + h->bl.h0d; ia.c0_dec = h->blob + h->bl.c0d;
    ia.h_att = sb.h_att[0]; ia.c_att = sb.c_att; ia.h_dec = sb.h_dec[0]; ia.c_dec = sb.c_dec;
    ia.ctx = sb.ctx; ia.w = sb.w[0]; ia.ynext = sb.ynext;
    ia.h_att_h = sb.h_att_h[0]; ia.h_att_l = sb.h_att_l[0]; ia.h_dec_h = sb.h_dec_h[0]; ia.h_dec_l = sb.h_dec_l[0];
    ia.ctx_h = sb.ctx_h; ia.ctx_l = sb.ctx_l;
    ia.out_mpad = (lstm_prec(h) && chunk_ok(d) && h->chunk_a) ? rows_pad(B) : 0;
    ia.memory = is_taco2(d) ? memory : nullptr;
    ia.B = B; ia.L = L; ia.D = d.d_ctx; ia.Ha = d.h_att; ia.Hd = d.h_dec; ia.d_mel = d.d_mel;
    launch_init(ia, st);
  }
  CallArgs ca;
  ca.t_begin = t_begin; ca.n_steps = n_steps; ca.t_stride = t_stride; ca.check_stop = check_stop;
  ca.dropout_mode = dropout_mode; ca.teacher_T = teacher_T; ca.stop_thr = stop_threshold; ca.seed = seed;
  ca.memory = memory; ca.masks = masks; ca.teacher = teacher; ca.teacher_flags = teacher_flags;
  ca.y = y; ca.s = s; ca.w = w;
  ca.debug_flags = h->debug_flags > 0 ? h->debug_flags : 0;
  ca.spin_limit = h->spin_limit > 0 ? h->spin_limit : kRoleSpinLimit;
  ca.loop_stamps = stamp_loop ? sb.loop_stamps : nullptr;
  // measurement only: TTSDEC_STAMPS=<file> collects per-workgroup time stamps of the two-role launches of the
  // call's last step and writes them to <file> (synchronises; never set in production)
  const char* stamp_file = getenv("TTSDEC_STAMPS");
  ca.stamps = nullptr;
  if (stamp_file && *stamp_file) {  // (the buffer belongs to the handle, hence to its device; freed by ttsdec_destroy)
    if (!h->stamps_buf) { HIP_TRY(h, hipMalloc(&h->stamps_buf, kStampKinds * 1024 * 8 * sizeof(unsigned long long))); }
    HIP_TRY(h, hipMemsetAsync(h->stamps_buf, 0, kStampKinds * 1024 * 8 * sizeof(unsigned long long), st));
    ca.stamps = h->stamps_buf;
  }
  launch_set_call(sb.ctrl, ca, st);
  HIP_TRY(h, hipMemsetAsync(sb.dep, 0, W.dep_bytes, st));  // arrival counters count from the call's first step

  StepIo io;
  memset(&io, 0, sizeof(io));
  io.B = B; io.L = L; io.use_ctrl = true;
  if (h->use_graph && n_steps >= kGraphSlots / 2) {
    rc = ensure_graph(h, sb, workspace, B, L);
    if (rc != TTSDEC_OK) return rc;
    for (int done = 0; done < n_steps; done += kGraphSlots) HIP_TRY(h, hipGraphLaunch(h->gexec, st));
  } else {
    for (int i = 0; i < n_steps; ++i) {
      io.slot = i;
      launch_step_serial(h, sb, io, st);
    }
  }
  io.node_pos = kLoopStampNodes - 1;     // (the end-of-call launches below stamp a node index no step launch uses)
  if (head_proj(h, B) && n_steps > 0) {  // (that step order leaves each step's projection to the NEXT step's launch)
    io.slot = (n_steps - 1) & 1;        // buffer parity of the call's last step (t_begin is even)
    launch_node(h, sb, io, N_JFIN, st);
  }
  if (use_frame(d)) launch_node(h, sb, io, N_FIN, st);  // the last step's frame: y, s, stop rule, next input
  launch_finish(sb.ctrl, T_out, st);
  if (ca.stamps) {
    std::string buf(kStampKinds * 1024 * 8 * sizeof(unsigned long long), '\0');
    HIP_TRY(h, hipStreamSynchronize(st));
    HIP_TRY(h, hipMemcpy(&buf[0], ca.stamps, buf.size(), hipMemcpyDeviceToHost));
    if (FILE* f = fopen(stamp_file, "wb")) { fwrite(buf.data(), 1, buf.size(), f); fclose(f); }
  }
  return check_launch(h, "decode");
}

int ttsdec_decode(ttsdec_handle* h, const float* memory, int B, int L, int t_begin, int n_steps, int t_stride,
                  float stop_threshold, int check_stop, int dropout_mode, const uint8_t* masks, uint64_t seed,
                  const float* teacher, int teacher_T, const uint8_t* teacher_flags, float* y, float* s, float* w,
                  int32_t* T_out, void* workspace, size_t workspace_bytes, void* stream) {
  return decode_impl(h, memory, B, L, t_begin, n_steps, t_stride, stop_threshold, check_stop, dropout_mode, masks, seed, teacher, teacher_T,
                     teacher_flags, y, s, w, T_out, workspace, workspace_bytes, stream, false);
}

int ttsdec_profile_loop(ttsdec_handle* h, const float* memory, int B, int L, int n_steps, int dropout_mode, const uint8_t* masks,
                        uint64_t seed, float* y, float* s, float* w, int32_t* T_out, void* workspace, size_t workspace_bytes, void* stream,
                        float* ms_out, const char** names_out, int n_out, int* n_kernels, float* step_ms) {
  if (!h || !ms_out || !workspace || n_steps < 2 * kGraphSlots || n_steps % kGraphSlots) return TTSDEC_ERR_INVALID_ARG;
  if (!h->use_graph) return TTSDEC_ERR_INVALID_ARG;  // (it is the replayed graph this function looks into)
  const StepOrder& order = step_order(h, B);
  if (n_kernels) *n_kernels = order.n;
  if (n_out < order.n || order.n >= kLoopStampNodes) return TTSDEC_ERR_INVALID_ARG;
  const WsLayout W = make_ws_layout(h->d, B, L);
  if (workspace_bytes < W.total) return TTSDEC_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc = decode_impl(h, memory, B, L, 0, n_steps, n_steps, -1e30f, 0, dropout_mode, masks, seed, nullptr, 0, nullptr, y, s, w, T_out, workspace,
                       workspace_bytes, stream, true);
  if (rc != TTSDEC_OK) return rc;
  unsigned long long t[kLoopStampSlots * kLoopStampNodes];
  HIP_TRY(h, hipStreamSynchronize(st));
  HIP_TRY(h, hipMemcpy(t, static_cast<char*>(workspace) + W.loop_stamps, sizeof(t), hipMemcpyDeviceToHost));
  // the LAST replay's start times: slots 0 .. kGraphSlots-1; a launch's span = the next launch's start - its own
  // (s_memrealtime ticks of 10 ns); the last launch of the last slot has no successor inside the replay and is left out
  double total = 0.0;
  for (int k = 0; k < order.n; ++k) {
    double sum = 0.0;
    int cnt = 0;
    for (int i = 0; i < kGraphSlots; ++i) {
      const bool last = k + 1 == order.n;
      if (last && i + 1 == kGraphSlots) continue;
      const unsigned long long a = t[i * kLoopStampNodes + k], b = last ? t[(i + 1) * kLoopStampNodes] : t[i * kLoopStampNodes + k + 1];
      if (a == 0 || b <= a) return TTSDEC_ERR_INVALID_ARG;  // (a launch that did not stamp: not a step kernel this function knows)
      sum += (double)(b - a) * 1e-5;  // ms
      ++cnt;
    }
    ms_out[k] = (float)(sum / cnt);
    if (names_out) names_out[k] = order.names[k];
    total += ms_out[k];
  }
  if (step_ms) *step_ms = (float)total;
  return TTSDEC_OK;
}

size_t ttsdec_postnet_workspace_bytes(const ttsdec_handle* h, int B, int T) {
  if (!h || B <= 0 || T <= 0 || h->d.postnet_layers <= 0) return 0;
  // two activation buffers (fp32, or hi+lo fp16 planes, or one bf16 plane) + 16-bit planes of the input
  // (+ for MelPostnet2: two fp32 residual-stream buffers and a second plane buffer)
  const size_t act = align_up((size_t)B * T * h->d.postnet_hidden * sizeof(float), 256);
  const size_t xin = align_up((size_t)B * T * h->d.d_mel * sizeof(float), 256);
  return 2 * act + (h->d.postnet_type == TTSDEC_POSTNET_TYPE_MEL2 ? 4 * xin : xin);
}

int ttsdec_postnet(ttsdec_handle* h, const float* y, int B, int T, int precision, float* y_post, void* workspace,
                   size_t workspace_bytes, void* stream) {
  if (!h || !y || !y_post || !workspace || B <= 0 || T <= 0) return TTSDEC_ERR_INVALID_ARG;
  if (h->d.postnet_layers <= 0) return TTSDEC_ERR_DIMS;
  if (precision != TTSDEC_POSTNET_F32 && precision != TTSDEC_POSTNET_BF16 && precision != TTSDEC_POSTNET_SPLIT_F16)
    return TTSDEC_ERR_INVALID_ARG;
  if (!h->blob) return TTSDEC_ERR_NOT_BOUND;
  int rc = check_device(h);
  if (rc != TTSDEC_OK) return rc;
  if (workspace_bytes < ttsdec_postnet_workspace_bytes(h, B, T) || (reinterpret_cast<uintptr_t>(workspace) & 255))
    return TTSDEC_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const ttsdec_dims& d = h->d;
  const BlobLayout& L = h->bl;
  // 16-bit modes need whole 16-byte columns of 16-bit elements in every K segment
  int prec = PREC_F32;
  if (!((d.d_mel | d.postnet_hidden) & 7)) {
    if (precision == TTSDEC_POSTNET_BF16) prec = PREC_BF16;
    if (precision == TTSDEC_POSTNET_SPLIT_F16 && h->wmax_post < kSplitMax) prec = PREC_F16S;
  }
  const size_t M = (size_t)B * T;
  const size_t act_bytes = align_up(M * d.postnet_hidden * sizeof(float), 256);
  char* wsb = static_cast<char*>(workspace);
  char* act[2] = {wsb, wsb + act_bytes};
  char* yin = wsb + 2 * act_bytes;
  auto plane = [&](size_t float_off) { return reinterpret_cast<const void*>(h->blob + float_off); };

  if (d.postnet_type == TTSDEC_POSTNET_TYPE_MEL2) {
    // MelPostnet2.forward (modules.py:213-216): x = x + conv3(lrelu(BN(conv2(lrelu(BN(conv1(x))))))) per layer
    const size_t xin_bytes = align_up(M * d.d_mel * sizeof(float), 256);
    char* xbuf[2] = {yin, yin + xin_bytes};             // fp32 residual stream, ping-pong
    char* xpl[2] = {yin + 2 * xin_bytes, yin + 3 * xin_bytes};  // its 16-bit planes, ping-pong
    const int Hh = d.postnet_hidden, kk = d.postnet_kernel;
    const float* xcur = y;
    const void *x0 = y, *x1 = y;
    if (prec == PREC_F16S) {
      f16* xh = reinterpret_cast<f16*>(xpl[1]);
      launch_split(y, xh, xh + M * d.d_mel, M * d.d_mel, st);
      x0 = xh; x1 = xh + M * d.d_mel;
    } else if (prec == PREC_BF16) {
      launch_to_bf16(y, xpl[1], M * d.d_mel, st);
      x0 = x1 = xpl[1];
    }
    auto wsel = [&](int i, int c, bool lo) -> const void* {
      if (prec == PREC_F32) return h->blob + L.p2_w[i][c];
      if (prec == PREC_BF16) return plane(L.p2_wb[i][c]);
      return plane(lo ? L.p2_wl[i][c] : L.p2_wh[i][c]);
    };
    for (int i = 0; i < d.postnet_layers; ++i) {
      const void *a0 = x0, *a1 = x1;
      int cin2 = d.d_mel;
      for (int c = 0; c < 2; ++c) {  // conv1 / conv2 + BN + LeakyReLU
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.prec = prec;
        g.a = make_seg1(a0, cin2, cin2); g.a_lo = make_seg1(a1, cin2, cin2);
        g.T = T; g.Cin = cin2; g.taps = kk; g.ldw = kk * cin2; g.K = kk * cin2;
        g.W = wsel(i, c, false); g.W_lo = wsel(i, c, true);
        g.M = (int)M; g.N = Hh; g.ldo = Hh;
        g.alpha = h->blob + L.p2_alpha[i][c]; g.beta = h->blob + L.p2_beta[i][c];
        char* o = act[c];
        if (prec == PREC_F32) { g.out = reinterpret_cast<float*>(o); a0 = a1 = o; }
        else if (prec == PREC_F16S) {
          g.out_kind = 1; g.out_h = reinterpret_cast<f16*>(o); g.out_l = g.out_h + M * Hh; a0 = g.out_h; a1 = g.out_l;
        } else { g.out_kind = 2; g.out_h = reinterpret_cast<f16*>(o); a0 = a1 = o; }
        launch_gemm(g, A_CONV, EPI_BN_LRELU, st);
        cin2 = Hh;
      }
      GemmArgs g;  // conv3 + residual
      memset(&g, 0, sizeof(g));
      g.prec = prec;
      g.a = make_seg1(a0, Hh, Hh); g.a_lo = make_seg1(a1, Hh, Hh);
      g.T = T; g.Cin = Hh; g.taps = kk; g.ldw = kk * Hh; g.K = kk * Hh;
      g.W = wsel(i, 2, false); g.W_lo = wsel(i, 2, true);
      g.M = (int)M; g.N = d.d_mel; g.ldo = d.d_mel;
      g.resid = xcur;
      const bool last = (i == d.postnet_layers - 1);
      float* xnew = last ? y_post : reinterpret_cast<float*>(xbuf[i & 1]);
      g.out = xnew;
      if (!last && prec == PREC_F16S) {
        g.out_kind = 1; g.out_h = reinterpret_cast<f16*>(xpl[i & 1]); g.out_l = g.out_h + M * d.d_mel; x0 = g.out_h; x1 = g.out_l;
      } else if (!last && prec == PREC_BF16) {
        g.out_kind = 2; g.out_h = reinterpret_cast<f16*>(xpl[i & 1]); x0 = x1 = xpl[i & 1];
      } else if (!last) {
        x0 = x1 = xnew;
      }
      launch_gemm(g, A_CONV, EPI_RESIDUAL, st);
      xcur = xnew;
    }
    return check_launch(h, "postnet2");
  }

  // current layer input as (plane 0, plane 1)
  const void *in0 = y, *in1 = y;
  if (prec == PREC_F16S) {
    f16* yh = reinterpret_cast<f16*>(yin);
    f16* yl = yh + M * d.d_mel;
    launch_split(y, yh, yl, M * d.d_mel, st);
    in0 = yh; in1 = yl;
  } else if (prec == PREC_BF16) {
    launch_to_bf16(y, yin, M * d.d_mel, st);
    in0 = in1 = yin;
  }
  int cin = d.d_mel;
  for (int i = 0; i < d.postnet_layers; ++i) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.prec = prec;
    g.a = make_seg1(in0, cin, cin);
    g.a_lo = make_seg1(in1, cin, cin);
    g.T = T; g.Cin = cin; g.taps = d.postnet_kernel;
    g.ldw = d.postnet_kernel * cin; g.K = d.postnet_kernel * cin;
    g.M = (int)M; g.N = d.postnet_hidden;
    g.alpha = h->blob + L.conv_alpha[i]; g.beta = h->blob + L.conv_beta[i];
    g.ldo = d.postnet_hidden;
    char* o = act[i & 1];
    // the wide hidden -> hidden layers on the 256 x 256 schedule (conv256.hip) where the shape is that kernel's - exact fp32 and
    // bf16; TTSDEC_CONV256=0 keeps the shared tile (measurement)
    static const bool use256 = [] { const char* e = getenv("TTSDEC_CONV256"); return !(e && e[0] == '0'); }();
    if (prec == PREC_F32) {
      g.W = g.W_lo = h->blob + L.conv_w[i];
      g.out = reinterpret_cast<float*>(o);
      in0 = in1 = o;
      if (use256 && launch_conv256_f32(static_cast<const float*>(g.a.p0), static_cast<const float*>(g.W), g.alpha, g.beta, g.out, (int)M, T, cin,
                                       d.postnet_kernel, d.postnet_hidden, st)) {
        cin = d.postnet_hidden;
        continue;
      }
    } else if (prec == PREC_F16S) {
      g.W = plane(L.conv_wh[i]); g.W_lo = plane(L.conv_wl[i]);
      g.out_kind = 1;
      g.out_h = reinterpret_cast<f16*>(o);
      g.out_l = g.out_h + M * d.postnet_hidden;
      in0 = g.out_h; in1 = g.out_l;
    } else {
      g.W = g.W_lo = plane(L.conv_wb[i]);
      g.out_kind = 2;
      g.out_h = reinterpret_cast<f16*>(o);
      in0 = in1 = o;
      if (use256 && launch_conv256_bf16(g.a.p0, g.W, g.alpha, g.beta, o, (int)M, T, cin, d.postnet_kernel, d.postnet_hidden, st)) {
        cin = d.postnet_hidden;
        continue;
      }
    }
    launch_gemm(g, A_CONV, EPI_BN_ISRU, st);
    cin = d.postnet_hidden;
  }
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.prec = prec;
  g.a = make_seg1(in0, cin, cin);
  g.a_lo = make_seg1(in1, cin, cin);
  if (prec == PREC_F32) g.W = g.W_lo = h->blob + L.fc_w;
  else if (prec == PREC_F16S) { g.W = plane(L.fc_wh); g.W_lo = plane(L.fc_wl); }
  else g.W = g.W_lo = plane(L.fc_wb);
  g.ldw = cin; g.K = cin; g.M = (int)M; g.N = d.d_mel;
  g.resid = y; g.out = y_post; g.ldo = d.d_mel;
  launch_gemm(g, A_PLAIN, EPI_RESIDUAL, st);
  return check_launch(h, "postnet");
}

int ttsdec_cell_step(ttsdec_handle* h, const float* x, const float* memory, int B, int L, float* w, float* ctx,
                     float* h_att, float* c_att, float* h_dec, float* c_dec, int dropout_mode, const uint8_t* masks,
                     uint64_t seed, int step, float* x_dec, void* workspace, size_t workspace_bytes, void* stream) {
  if (!h || !x || !memory || !w || !ctx || !h_att || !c_att || !h_dec || !c_dec || !x_dec || !workspace)
    return TTSDEC_ERR_INVALID_ARG;
  if (B <= 0 || L <= 0 || step < 0) return TTSDEC_ERR_INVALID_ARG;
  if (dropout_mode == TTSDEC_DROPOUT_MASKS && !masks) return TTSDEC_ERR_INVALID_ARG;
  if (dropout_mode == TTSDEC_DROPOUT_PHILOX && h->d.p_dropout != 0.5f) return TTSDEC_ERR_INVALID_ARG;
  if (!h->blob) return TTSDEC_ERR_NOT_BOUND;
  int rc = check_device(h);
  if (rc != TTSDEC_OK) return rc;
  const WsLayout W = make_ws_layout(h->d, B, L);
  if (workspace_bytes < W.total || (reinterpret_cast<uintptr_t>(workspace) & 255)) return TTSDEC_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const StepBufs sb = carve(W, workspace);
  const ttsdec_dims& d = h->d;
  const size_t b = (size_t)B;
  const int p = step & 1;
  const bool t2 = is_taco2(d);
  // caller state -> workspace (the step kernels ping-pong h and w)
  launch_copy(x, sb.ynext, b * d.d_mel, st);
  if (!t2) launch_copy(ctx, sb.ctx, b * d.d_ctx, st);
  launch_copy(w, sb.w[p], b * L, st);
  launch_copy(h_att, sb.h_att[p], b * d.h_att, st);
  launch_copy(c_att, sb.c_att, b * d.h_att, st);
  launch_copy(h_dec, sb.h_dec[p], b * d.h_dec, st);
  launch_copy(c_dec, sb.c_dec, b * d.h_dec, st);
  const int prec = lstm_prec(h);
  if (prec && chunk_ok(d) && h->chunk_a) {
    const int mp = rows_pad(B);
    if (!t2) launch_split_chunked(ctx, sb.ctx_h, sb.ctx_l, B, d.d_ctx, mp, st);
    launch_split_chunked(h_att, sb.h_att_h[p], sb.h_att_l[p], B, d.h_att, mp, st);
    launch_split_chunked(h_dec, sb.h_dec_h[p], sb.h_dec_l[p], B, d.h_dec, mp, st);
  } else if (prec) {
    if (!t2) launch_split(ctx, sb.ctx_h, sb.ctx_l, b * d.d_ctx, st);
    launch_split(h_att, sb.h_att_h[p], sb.h_att_l[p], b * d.h_att, st);
    launch_split(h_dec, sb.h_dec_h[p], sb.h_dec_l[p], b * d.h_dec, st);
  }
  StepIo io;
  memset(&io, 0, sizeof(io));
  io.memory = memory; io.B = B; io.L = L; io.t = step; io.t_rel = 0; io.t_stride = 1;
  io.dropout_mode = dropout_mode; io.masks = masks; io.seed = seed;
  io.use_ctrl = false;
  if (t2) {
    // Taco2DecoderCell: ctx = bmm(w_prev, memory) first (decoder_cell.py:118), then both LSTMs,
    // then the weight update; the ctx handed back is the one the LSTMs consumed
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.memory = memory; a.w_prev = sb.w[p]; a.w_new = sb.w[1 - p]; a.ctx = sb.ctx; a.ctx_only = 1;
    if (prec) { a.ctx_h = sb.ctx_h; a.ctx_l = sb.ctx_l; a.out_mpad = (chunk_ok(d) && h->chunk_a) ? rows_pad(B) : 0; }
    a.B = B; a.L = L; a.D = d.d_ctx; a.t_stride = 1;
    launch_attn(a, st);
    if (use_frame(d)) launch_node(h, sb, io, N_F, st);  // (io.finalize = 0: the input frame is x itself)
    else { launch_node(h, sb, io, N_P0, st); launch_node(h, sb, io, N_P1, st); }
    launch_node(h, sb, io, N_A, st);
    launch_node(h, sb, io, N_D, st);
    launch_copy(sb.ctx, ctx, b * d.d_ctx, st);
    launch_node(h, sb, io, N_Q, st);
    launch_node(h, sb, io, N_T, st);
  } else {
    // the projection (fc_mel / fc_stop) belongs to Decoder, not to the cell
    if (use_frame(d)) launch_node(h, sb, io, N_F, st);
    else { launch_node(h, sb, io, N_P0, st); launch_node(h, sb, io, N_P1, st); }
    const Node cell_nodes[4] = {N_A, N_Q, N_T, N_D};
    for (Node n : cell_nodes) launch_node(h, sb, io, n, st);
    launch_copy(sb.ctx, ctx, b * d.d_ctx, st);
  }
  launch_copy(sb.w[1 - p], w, b * L, st);
  launch_copy(sb.h_att[1 - p], h_att, b * d.h_att, st);
  launch_copy(sb.c_att, c_att, b * d.h_att, st);
  launch_copy(sb.h_dec[1 - p], h_dec, b * d.h_dec, st);
  launch_copy(sb.c_dec, c_dec, b * d.h_dec, st);
  if (t2) {
    // x_dec = cat[h0, h1, zeros_like(ctx)] (decoder_cell.py:136)
    const size_t ld = (size_t)(d.h_att + d.h_dec + d.d_ctx) * sizeof(float);
    HIP_TRY(h, hipMemsetAsync(x_dec, 0, ld * b, st));
    HIP_TRY(h, hipMemcpy2DAsync(x_dec, ld, sb.h_att[1 - p], (size_t)d.h_att * sizeof(float), (size_t)d.h_att * sizeof(float), b,
                                hipMemcpyDeviceToDevice, st));
    HIP_TRY(h, hipMemcpy2DAsync(x_dec + d.h_att, ld, sb.h_dec[1 - p], (size_t)d.h_dec * sizeof(float),
                                (size_t)d.h_dec * sizeof(float), b, hipMemcpyDeviceToDevice, st));
  } else {
    // x_dec = cat[h_dec, ctx] (decoder_cell.py:192)
    const size_t ld = (size_t)(d.h_dec + d.d_ctx) * sizeof(float);
    HIP_TRY(h, hipMemcpy2DAsync(x_dec, ld, sb.h_dec[1 - p], (size_t)d.h_dec * sizeof(float), (size_t)d.h_dec * sizeof(float), b,
                                hipMemcpyDeviceToDevice, st));
    HIP_TRY(h, hipMemcpy2DAsync(x_dec + d.h_dec, ld, sb.ctx, (size_t)d.d_ctx * sizeof(float), (size_t)d.d_ctx * sizeof(float), b,
                                hipMemcpyDeviceToDevice, st));
  }
  return check_launch(h, "cell_step");
}

int ttsdec_profile_step(ttsdec_handle* h, const float* memory, int B, int L, int iters, int dropout_mode,
                        const uint8_t* masks, uint64_t seed, float* y, float* s, float* w, void* workspace,
                        size_t workspace_bytes, void* stream, float* ms_out, const char** names_out, int n_out,
                        int* n_kernels) {
  if (!h || !memory || !y || !s || !w || !workspace || !ms_out || iters <= 0) return TTSDEC_ERR_INVALID_ARG;
  const StepOrder& order = step_order(h, B);
  if (n_kernels) *n_kernels = order.n;
  if (n_out < order.n) return TTSDEC_ERR_INVALID_ARG;
  if (dropout_mode == TTSDEC_DROPOUT_MASKS && !masks) return TTSDEC_ERR_INVALID_ARG;
  if (dropout_mode == TTSDEC_DROPOUT_PHILOX && h->d.p_dropout != 0.5f) return TTSDEC_ERR_INVALID_ARG;
  if (!h->blob) return TTSDEC_ERR_NOT_BOUND;
  int rc = check_device(h);
  if (rc != TTSDEC_OK) return rc;
  const WsLayout W = make_ws_layout(h->d, B, L);
  if (workspace_bytes < W.total || (reinterpret_cast<uintptr_t>(workspace) & 255)) return TTSDEC_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const StepBufs sb = carve(W, workspace);
  StepIo io;
  memset(&io, 0, sizeof(io));
  io.memory = memory; io.B = B; io.L = L; io.t = 0; io.t_rel = 0; io.t_stride = 1;
  io.dropout_mode = dropout_mode; io.masks = masks; io.seed = seed;
  io.y = y; io.s = s; io.w = w; io.use_ctrl = false;
  io.t = 1; io.t_rel = 1; io.t_stride = 2; io.finalize = 1;  // a mid-sequence step: the frame kernel also finishes step 0's frame
  io.dbg = h->profile_ablation > 0 ? h->profile_ablation : 0;  // measurement only (option "profile_ablation")
  hipEvent_t e0, e1;
  HIP_TRY(h, hipEventCreate(&e0));
  HIP_TRY(h, hipEventCreate(&e1));
  // the step's own launches, then (two-role step only, when the caller left room) each role on its own:
  // what the partner of a two-role launch costs alone
  Node nodes[kMaxKernelsPerStep + 4];
  const char* names[kMaxKernelsPerStep + 4];
  int nn = 0;
  for (int k = 0; k < order.n; ++k) { nodes[nn] = order.nodes[k]; names[nn++] = order.names[k]; }
  if ((&order == &kOrderProdO || &order == &kOrderProdO2 || &order == &kOrderProdH || &order == &kOrderProdH2 || &order == &kOrderProdO2Q ||
       &order == &kOrderProdH2Q) && n_out >= order.n + 4) {
    const Node extra[4] = {N_F, N_AG, N_T, N_DG};
    const char* extra_names[4] = {"prenet(alone)", "lstm_att_lean(alone)", "attention(alone)", "lstm_dec_lean(alone)"};
    for (int k = 0; k < 4; ++k) { nodes[nn] = extra[k]; names[nn++] = extra_names[k]; }
  }
  if (n_kernels) *n_kernels = nn;
  for (int k = 0; k < nn; ++k) {
    for (int i = 0; i < 3; ++i) launch_node(h, sb, io, nodes[k], st);  // warm
    HIP_TRY(h, hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) launch_node(h, sb, io, nodes[k], st);
    HIP_TRY(h, hipEventRecord(e1, st));
    HIP_TRY(h, hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&ms, e0, e1));
    ms_out[k] = ms / iters;
    if (names_out) names_out[k] = names[k];
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return check_launch(h, "profile_step");
}

}  // extern "C"
