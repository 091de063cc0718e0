// C ABI of the text encoder: Encoder2.forward in eval mode,
// tacotron/encoder.py:67-82 with BiDiLSTM (tacotron/modules/rnn.py:112-127).  Runs once per batch,
// before the decode loop; built from the same kernels as the hot path:
//   embedding gather -> 3 x implicit-GEMM conv (k=5) + BN + ISRLU -> cat[conv, emb]
//   -> one GEMM for the input projections of both LSTM directions over all time steps
//   -> L sequential steps of the LSTM kernel in packed-sequence mode (forward and reverse).
// Convs and the input projection: exact fp32 GEMMs (default) or split-fp16 ones (ttsenc_set_precision); the recurrence: exact fp32.
#include <string.h>

#include <new>
#include <string>

#include "kernels.h"

using namespace ttsdec;

namespace {
constexpr size_t kAlign = 64;  // floats
inline size_t up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct EncBlob {  // offsets in floats
  size_t emb, conv_w[3], alpha[3], beta[3], w_ih, w_hh[2], h0, c0, total;
};
struct EncWs {  // offsets in bytes
  size_t x, act[2], cat, gx, h[2][2], c[2], planes, total;
};
}  // namespace

struct ttsenc_handle {
  ttsenc_dims d;
  EncBlob bl;
  const float* blob;
  int device;  // HIP device current at create (-1: none); must be current for every later call
  int precision;  // TTSDEC_PREC_F32 (default: the reference's arithmetic) or TTSDEC_PREC_SPLIT_F16 for the convs and the input projection
  std::string hip_err;
};

namespace {
EncBlob make_layout(const ttsenc_dims& d) {
  EncBlob L;
  memset(&L, 0, sizeof(L));
  size_t off = 0;
  auto take = [&](size_t n) { const size_t o = off; off = up(off + n, kAlign); return o; };
  const size_t E = d.d_emb, H = d.d_out / 2, k = d.conv_kernel;
  L.emb = take((size_t)d.alphabet_size * E);
  // (GEMM weights of n elements take 2n floats: fp32 | fp16 hi plane | fp16 lo plane)
  for (int i = 0; i < 3; ++i) { L.conv_w[i] = take(2 * E * k * E); L.alpha[i] = take(E); L.beta[i] = take(E); }
  L.w_ih = take(2 * 8 * H * 2 * E);  // [fwd i,f,g,o ; rev i,f,g,o] x [conv | emb]
  L.w_hh[0] = take(4 * H * H);
  L.w_hh[1] = take(4 * H * H);
  L.h0 = take(2 * H);
  L.c0 = take(2 * H);
  L.total = off;
  return L;
}
EncWs make_ws(const ttsenc_dims& d, int B, int Lm) {
  EncWs W;
  size_t off = 0;
  auto take = [&](size_t nfloats) { const size_t o = off; off = up(off + nfloats * sizeof(float), 256); return o; };
  const size_t M = (size_t)B * Lm, E = d.d_emb, H = d.d_out / 2;
  W.x = take(M * E);
  W.act[0] = take(M * E);
  W.act[1] = take(M * E);
  W.cat = take(M * 2 * E);
  W.gx = take(M * 8 * H);
  for (int dir = 0; dir < 2; ++dir) {
    W.h[dir][0] = take((size_t)B * H);
    W.h[dir][1] = take((size_t)B * H);
    W.c[dir] = take((size_t)B * H);
  }
  W.planes = take(M * 2 * E);  // hi + lo fp16 planes of one GEMM's A operand
  W.total = off;
  return W;
}
int enc_fail(ttsenc_handle* h, const char* where) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return TTSDEC_OK;
  if (h) h->hip_err = std::string(where) + ": " + hipGetErrorString(e);
  return TTSDEC_ERR_HIP;
}
}  // namespace

extern "C" {

int ttsenc_create(const ttsenc_dims* dims, ttsenc_handle** out) {
  if (!dims || !out) return TTSDEC_ERR_INVALID_ARG;
  *out = nullptr;
  const ttsenc_dims& d = *dims;
  if (d.alphabet_size <= 0 || d.d_emb <= 0 || (d.d_emb & 3) || d.d_out <= 0 || (d.d_out & 7)) return TTSDEC_ERR_DIMS;
  if (d.conv_kernel < 1 || !(d.conv_kernel & 1)) return TTSDEC_ERR_DIMS;
  ttsenc_handle* h = new (std::nothrow) ttsenc_handle();
  if (!h) return TTSDEC_ERR_INVALID_ARG;
  h->d = d;
  h->bl = make_layout(d);
  h->blob = nullptr;
  h->device = current_device_or_minus1();
  h->precision = TTSDEC_PREC_F32;
  *out = h;
  return TTSDEC_OK;
}

int ttsenc_destroy(ttsenc_handle* h) {
  delete h;
  return TTSDEC_OK;
}

const char* ttsenc_last_hip_error(const ttsenc_handle* h) { return h ? h->hip_err.c_str() : ""; }
int ttsenc_set_precision(ttsenc_handle* h, int precision) {
  if (!h || (precision != TTSDEC_PREC_F32 && precision != TTSDEC_PREC_SPLIT_F16)) return TTSDEC_ERR_INVALID_ARG;
  h->precision = precision;
  return TTSDEC_OK;
}
int ttsenc_get_precision(const ttsenc_handle* h) { return h ? h->precision : TTSDEC_ERR_INVALID_ARG; }
int ttsenc_num_weight_tensors(const ttsenc_handle* h) { return h ? TTSENC_W_COUNT : TTSDEC_ERR_INVALID_ARG; }
size_t ttsenc_packed_bytes(const ttsenc_handle* h) { return h ? h->bl.total * sizeof(float) : 0; }
size_t ttsenc_workspace_bytes(const ttsenc_handle* h, int B, int L) {
  if (!h || B <= 0 || L <= 0) return 0;
  return make_ws(h->d, B, L).total;
}

int ttsenc_pack_weights(ttsenc_handle* h, const float* const* src, int n_src, void* blob, void* stream) {
  if (!h || !src || !blob || n_src != TTSENC_W_COUNT) return TTSDEC_ERR_INVALID_ARG;
  for (int i = 0; i < n_src; ++i)
    if (!src[i]) return TTSDEC_ERR_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(blob) & 255) return TTSDEC_ERR_WORKSPACE;
  if (!device_is_current(h->device)) return TTSDEC_ERR_DEVICE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const ttsenc_dims& d = h->d;
  const EncBlob& L = h->bl;
  float* b = static_cast<float*>(blob);
  const size_t E = d.d_emb, H = d.d_out / 2;
  if (hipMemsetAsync(blob, 0, L.total * sizeof(float), st) != hipSuccess) return enc_fail(h, "memset");
  launch_copy(src[TTSENC_W_EMB], b + L.emb, (size_t)d.alphabet_size * E, st);
  // conv.{0,3,6}.weight -> [C_out][tap][C_in]; BatchNorm1d (the third has no affine) -> (alpha, beta)
  launch_conv_transpose(src[TTSENC_W_CONV0], b + L.conv_w[0], (int)E, (int)E, d.conv_kernel, st);
  launch_bn_fold(src[TTSENC_W_BN0_W], src[TTSENC_W_BN0_B], src[TTSENC_W_BN0_MEAN], src[TTSENC_W_BN0_VAR], d.bn_eps, b + L.alpha[0],
                 b + L.beta[0], (int)E, st);
  launch_conv_transpose(src[TTSENC_W_CONV1], b + L.conv_w[1], (int)E, (int)E, d.conv_kernel, st);
  launch_bn_fold(src[TTSENC_W_BN1_W], src[TTSENC_W_BN1_B], src[TTSENC_W_BN1_MEAN], src[TTSENC_W_BN1_VAR], d.bn_eps, b + L.alpha[1],
                 b + L.beta[1], (int)E, st);
  launch_conv_transpose(src[TTSENC_W_CONV2], b + L.conv_w[2], (int)E, (int)E, d.conv_kernel, st);
  launch_bn_fold(nullptr, nullptr, src[TTSENC_W_BN2_MEAN], src[TTSENC_W_BN2_VAR], d.bn_eps, b + L.alpha[2], b + L.beta[2], (int)E, st);
  // both directions' input weights stacked on the output axis: one GEMM gives every gate pre-activation
  launch_copy(src[TTSENC_W_IH_FWD], b + L.w_ih, 4 * H * 2 * E, st);
  launch_copy(src[TTSENC_W_IH_REV], b + L.w_ih + 4 * H * 2 * E, 4 * H * 2 * E, st);
  {
    // split-fp16 planes of the conv and input-projection weights (used when d_emb is a multiple of 8)
    auto planes = [&](size_t off, size_t n) {
      f16* hi = reinterpret_cast<f16*>(b + off + n);
      launch_split(b + off, hi, hi + n, n, st);
    };
    for (int i = 0; i < 3; ++i) planes(L.conv_w[i], E * d.conv_kernel * E);
    planes(L.w_ih, 8 * H * 2 * E);
  }
  launch_copy(src[TTSENC_W_HH_FWD], b + L.w_hh[0], 4 * H * H, st);
  launch_copy(src[TTSENC_W_HH_REV], b + L.w_hh[1], 4 * H * H, st);
  launch_copy(src[TTSENC_W_H0], b + L.h0, 2 * H, st);
  launch_copy(src[TTSENC_W_C0], b + L.c0, 2 * H, st);
  const int rc = enc_fail(h, "pack_weights");
  if (rc == TTSDEC_OK) h->blob = b;
  return rc;
}

int ttsenc_bind_weights(ttsenc_handle* h, const void* blob) {
  if (!h || !blob || (reinterpret_cast<uintptr_t>(blob) & 255)) return TTSDEC_ERR_INVALID_ARG;
  h->blob = static_cast<const float*>(blob);
  return TTSDEC_OK;
}

int ttsenc_forward(ttsenc_handle* h, const int64_t* ids, const int32_t* lengths, int B, int L, int L_out, float* memory,
                   void* workspace, size_t workspace_bytes, void* stream, int32_t* status) {
  if (!h || !ids || !lengths || !memory || !workspace || B <= 0 || L <= 0 || L_out <= 0 || L_out > L) return TTSDEC_ERR_INVALID_ARG;
  if (!h->blob) return TTSDEC_ERR_NOT_BOUND;
  if (!device_is_current(h->device)) return TTSDEC_ERR_DEVICE;
  const ttsenc_dims& d = h->d;
  const EncWs W = make_ws(d, B, L);
  if (workspace_bytes < W.total || (reinterpret_cast<uintptr_t>(workspace) & 255)) return TTSDEC_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const EncBlob& bl = h->bl;
  const float* blob = h->blob;
  char* ws = static_cast<char*>(workspace);
  auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
  const int E = d.d_emb, H = d.d_out / 2;
  const int M = B * L;
  float *x = F(W.x), *cat = F(W.cat), *gx = F(W.gx);

  // encoder.py:69: embedding (row 0 of the table is the zero padding vector); also the right half of the cat
  launch_embed(reinterpret_cast<const long long*>(ids), blob + bl.emb, d.alphabet_size, M, E, x, E, cat + E, 2 * E, status, st);
  // encoder.py:70: three conv blocks over the padded sequence; the last writes the left half of the cat
  // In split-fp16 mode the convs and the input projection run on two fp16 planes per operand (gemm_tile.h): one elementwise
  // pass makes the hi / lo planes of each A operand.
  const bool split = h->precision == TTSDEC_PREC_SPLIT_F16 && !(E & 7);  // (ttsenc_set_precision; exact fp32 MFMAs otherwise)
  f16* ph = reinterpret_cast<f16*>(ws + W.planes);
  const float* in = x;
  for (int i = 0; i < 3; ++i) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.a = make_seg1(in, E, E); g.a_lo = g.a;
    g.T = L; g.Cin = E; g.taps = d.conv_kernel;
    g.W = g.W_lo = blob + bl.conv_w[i]; g.ldw = d.conv_kernel * E; g.K = d.conv_kernel * E;
    if (split) {
      const size_t n = (size_t)M * E, nw = (size_t)E * d.conv_kernel * E;
      launch_split(in, ph, ph + n, n, st);
      g.a = make_seg1(ph, E, E); g.a_lo = make_seg1(ph + n, E, E);
      g.prec = PREC_F16S;
      g.W = reinterpret_cast<const f16*>(blob + bl.conv_w[i] + nw);
      g.W_lo = reinterpret_cast<const f16*>(blob + bl.conv_w[i] + nw) + nw;
    }
    g.M = M; g.N = E;
    g.alpha = blob + bl.alpha[i]; g.beta = blob + bl.beta[i];
    if (i < 2) { g.out = F(W.act[i]); g.ldo = E; in = g.out; }
    else { g.out = cat; g.ldo = 2 * E; }
    launch_gemm(g, A_CONV, EPI_BN_ISRLU, st);
  }
  // input projections of both directions for every (b, t): gx [M, 8H] = cat [M, 2E] . W_ih^T
  {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.a = make_seg1(cat, 2 * E, 2 * E); g.a_lo = g.a;
    g.W = g.W_lo = blob + bl.w_ih; g.ldw = 2 * E; g.K = 2 * E; g.M = M; g.N = 8 * H; g.out = gx; g.ldo = 8 * H;
    if (split) {
      const size_t n = (size_t)M * 2 * E, nw = (size_t)8 * H * 2 * E;
      launch_split(cat, ph, ph + n, n, st);
      g.a = make_seg1(ph, 2 * E, 2 * E); g.a_lo = make_seg1(ph + n, 2 * E, 2 * E);
      g.prec = PREC_F16S;
      g.W = reinterpret_cast<const f16*>(blob + bl.w_ih + nw);
      g.W_lo = reinterpret_cast<const f16*>(blob + bl.w_ih + nw) + nw;
      launch_gemm(g, A_PLAIN, EPI_GENERIC, st);  // (the precision-aware plain epilogue: no bias / mask / residual here)
    } else {
      launch_gemm(g, A_PLAIN, EPI_PLAIN, st);
    }
  }
  // initial state (rnn.py:117-118: rnn_h0 / rnn_c0 chunked over the two directions), zero-padded output
  for (int dir = 0; dir < 2; ++dir) {
    launch_fill_rows(F(W.h[dir][0]), blob + bl.h0 + dir * H, B, H, st);
    launch_fill_rows(F(W.c[dir]), blob + bl.c0 + dir * H, B, H, st);
  }
  if (hipMemsetAsync(memory, 0, (size_t)B * L_out * 2 * H * sizeof(float), st) != hipSuccess) return enc_fail(h, "memset");
  // the recurrence: step t advances every utterance that is still running; the reverse direction
  // walks each utterance from its own last token (packed-sequence semantics, rnn.py:113-126)
  for (int t = 0; t < L_out; ++t) {
    const int p = t & 1;
    LstmArgs dirs[2];
    for (int dir = 0; dir < 2; ++dir) {
      LstmArgs& a = dirs[dir];
      memset(&a, 0, sizeof(a));
      a.a = make_seg1(F(W.h[dir][p]), H, H); a.a_lo = a.a;
      a.w = make_seg1(blob + bl.w_hh[dir], H, H); a.w_lo = a.w;
      a.bsum = nullptr; a.h_prev = F(W.h[dir][p]); a.c = F(W.c[dir]); a.h_out = F(W.h[dir][1 - p]);
      a.M = B; a.H = H; a.K = H; a.pz = 0.f; a.mode = 2;
      a.seq_lens = lengths; a.seq_t = t; a.seq_L = L; a.seq_Lout = L_out; a.seq_reverse = dir;
      a.gx = gx; a.gx_ld = 8 * H; a.gx_off = dir * 4 * H;
      a.seq_out = memory; a.seq_out_ld = 2 * H; a.seq_out_off = dir * H;
    }
    launch_lstm_pair(dirs[0], dirs[1], st);  // both directions of step t in one launch
  }
  return enc_fail(h, "encoder forward");
}

}  // extern "C"
