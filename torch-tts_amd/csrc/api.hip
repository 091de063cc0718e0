// C ABI of libttsdec.so (see include/ttsdec.h).  Host-side orchestration only:
// argument checks, blob / workspace carving, and the per-step launch sequence.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>

#include "kernels.h"

using namespace ttsdec;

namespace {

constexpr int kMaxPostnetLayers = 8;
constexpr size_t kAlignFloats = 64;  // 256 B

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct BlobLayout {  // offsets in floats
  size_t pre0_w, pre0_b, pre1_w, pre1_b, wq;
  size_t att_ih, att_hh, att_b, dec_ih, dec_hh, dec_b;
  size_t att_ih_h, att_ih_l, att_hh_h, att_hh_l, dec_ih_h, dec_ih_l, dec_hh_h, dec_hh_l;  // split-fp16 planes
  size_t h0a, c0a, h0d, c0d;
  size_t proj_w, proj_b;
  size_t conv_w[kMaxPostnetLayers], conv_alpha[kMaxPostnetLayers], conv_beta[kMaxPostnetLayers];
  size_t fc_w;
  size_t total;
};

struct WsLayout {  // offsets in bytes
  size_t ctrl, xpre0, xpre, ctx, h_att[2], c_att, h_dec[2], c_dec, q, ynext, w[2];
  size_t xpre_h, xpre_l, ctx_h, ctx_l, h_att_h[2], h_att_l[2], h_dec_h[2], h_dec_l[2];  // split-fp16 planes
  size_t total;
};

}  // namespace

struct ttsdec_handle {
  ttsdec_dims d;
  int precision;  // TTSDEC_PREC_*
  int device;  // -1: no HIP device was available at create (host-only queries still work)
  BlobLayout bl;
  const float* blob;
  std::string hip_err;
};

namespace {

BlobLayout make_blob_layout(const ttsdec_dims& d) {
  BlobLayout L;
  memset(&L, 0, sizeof(L));
  size_t off = 0;
  auto take = [&](size_t n) {
    const size_t o = off;
    off = align_up(off + n, kAlignFloats);
    return o;
  };
  const size_t Ha = d.h_att, Hd = d.h_dec, D = d.d_ctx, P = d.d_pre, Mel = d.d_mel, R = d.r;
  L.pre0_w = take(P * Mel);
  L.pre0_b = take(P);
  L.pre1_w = take(P * P);
  L.pre1_b = take(P);
  L.wq = take(D * Ha);
  L.att_ih = take(4 * Ha * (P + D));
  L.att_hh = take(4 * Ha * Ha);
  L.att_b = take(4 * Ha);
  L.dec_ih = take(4 * Hd * (Ha + D));
  L.dec_hh = take(4 * Hd * Hd);
  L.dec_b = take(4 * Hd);
  // fp16 planes occupy half a float per element
  L.att_ih_h = take(2 * Ha * (P + D)); L.att_ih_l = take(2 * Ha * (P + D));
  L.att_hh_h = take(2 * Ha * Ha);      L.att_hh_l = take(2 * Ha * Ha);
  L.dec_ih_h = take(2 * Hd * (Ha + D)); L.dec_ih_l = take(2 * Hd * (Ha + D));
  L.dec_hh_h = take(2 * Hd * Hd);      L.dec_hh_l = take(2 * Hd * Hd);
  L.h0a = take(Ha);
  L.c0a = take(Ha);
  L.h0d = take(Hd);
  L.c0d = take(Hd);
  L.proj_w = take((R * Mel + R) * (Hd + D));
  L.proj_b = take(R * Mel + R);
  size_t cin = Mel;
  for (int i = 0; i < d.postnet_layers; ++i) {
    L.conv_w[i] = take((size_t)d.postnet_hidden * d.postnet_kernel * cin);
    L.conv_alpha[i] = take(d.postnet_hidden);
    L.conv_beta[i] = take(d.postnet_hidden);
    cin = d.postnet_hidden;
  }
  if (d.postnet_layers > 0) L.fc_w = take(Mel * (size_t)d.postnet_hidden);
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
  W.xpre0 = take(b * d.d_pre);
  W.xpre = take(b * d.d_pre);
  W.ctx = take(b * d.d_ctx);
  W.h_att[0] = take(b * d.h_att);
  W.h_att[1] = take(b * d.h_att);
  W.c_att = take(b * d.h_att);
  W.h_dec[0] = take(b * d.h_dec);
  W.h_dec[1] = take(b * d.h_dec);
  W.c_dec = take(b * d.h_dec);
  W.q = take(b * d.d_ctx);
  W.ynext = take(b * d.d_mel);
  W.w[0] = take(b * Lm);
  W.w[1] = take(b * Lm);
  auto takeh = [&](size_t nhalfs) { return take((nhalfs + 1) / 2); };
  W.xpre_h = takeh(b * d.d_pre); W.xpre_l = takeh(b * d.d_pre);
  W.ctx_h = takeh(b * d.d_ctx); W.ctx_l = takeh(b * d.d_ctx);
  for (int i = 0; i < 2; ++i) {
    W.h_att_h[i] = takeh(b * d.h_att); W.h_att_l[i] = takeh(b * d.h_att);
    W.h_dec_h[i] = takeh(b * d.h_dec); W.h_dec_l[i] = takeh(b * d.h_dec);
  }
  W.total = off;
  return W;
}

int check_dims(const ttsdec_dims& d) {
  const int v[] = {d.d_mel, d.d_pre, d.d_ctx, d.h_att, d.h_dec};
  for (int x : v)
    if (x <= 0 || (x & 3)) return TTSDEC_ERR_DIMS;
  if (d.r < 1 || d.d_ctx > 4096) return TTSDEC_ERR_DIMS;
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

int check_device(ttsdec_handle* h) {
  if (h->device < 0) return TTSDEC_ERR_DEVICE;
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != h->device) return TTSDEC_ERR_DEVICE;
  return TTSDEC_OK;
}

int check_launch(ttsdec_handle* h, const char* where) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(h, e, where);
  return TTSDEC_OK;
}

struct StepBufs {
  Ctrl* ctrl;
  float *xpre0, *xpre, *ctx, *h_att[2], *c_att, *h_dec[2], *c_dec, *q, *ynext, *w[2];
  f16 *xpre_h, *xpre_l, *ctx_h, *ctx_l, *h_att_h[2], *h_att_l[2], *h_dec_h[2], *h_dec_l[2];
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
  return s;
}

struct StepIo {
  const float* memory;
  int B, L;
  int t, t_rel, t_stride;
  float stop_thr;
  int check_stop;
  int dropout_mode;
  const uint8_t* masks;  // base of this call's [n_steps, 2, B, d_pre]
  uint64_t seed;
  const float* teacher;
  int teacher_T;
  const uint8_t* teacher_flags;
  float *y, *s, *w;
  bool use_ctrl;
  int dbg;
};

// split-fp16 needs every K segment to be whole 16-byte columns of fp16 (multiples of 8)
bool split_ok(const ttsdec_dims& d) { return !((d.d_pre | d.d_ctx | d.h_att | d.h_dec) & 7); }
int lstm_prec(const ttsdec_handle* h) { return (h->precision == TTSDEC_PREC_SPLIT_F16 && split_ok(h->d)) ? 1 : 0; }

constexpr int kKernelsPerStep = 7;
const char* const kKernelNames[kKernelsPerStep] = {"prenet0", "prenet1", "lstm_att", "query", "attention", "lstm_dec", "proj"};

// Launches kernel `which` (0..6) of decode step io.t; which < 0 launches the whole step.
void launch_step(const ttsdec_handle* h, const StepBufs& sb, const StepIo& io, int which, hipStream_t st) {
  const ttsdec_dims& d = h->d;
  const BlobLayout& bl = h->bl;
  const float* blob = h->blob;
  const int p = io.t & 1;
  Ctrl* ctrl = io.use_ctrl ? sb.ctrl : nullptr;
  const int B = io.B, P = d.d_pre, D = d.d_ctx, Ha = d.h_att, Hd = d.h_dec;
  const float keep_scale = 1.0f / (1.0f - d.p_dropout);
  const int prec = lstm_prec(h);
  const f16* bh = reinterpret_cast<const f16*>(blob);  // fp16 planes live at float offsets of the same blob
  auto plane = [&](size_t float_off) { return reinterpret_cast<const f16*>(blob + float_off); };
  (void)bh;

  if (which < 0 || which == 0 || which == 1) {
    for (int layer = 0; layer < 2; ++layer) {
      if (which >= 0 && which != layer) continue;
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      if (layer == 0) {
        g.a = make_seg1(sb.ynext, d.d_mel, d.d_mel);
        g.W = blob + bl.pre0_w; g.ldw = d.d_mel; g.K = d.d_mel; g.bias = blob + bl.pre0_b;
        g.out = sb.xpre0;
        g.teacher = io.teacher; g.teacher_T = io.teacher_T; g.teacher_flags = io.teacher_flags;
      } else {
        g.a = make_seg1(sb.xpre0, P, P);
        g.W = blob + bl.pre1_w; g.ldw = P; g.K = P; g.bias = blob + bl.pre1_b;
        g.out = sb.xpre;
        if (prec) { g.out_h = sb.xpre_h; g.out_l = sb.xpre_l; }
      }
      g.M = B; g.N = P; g.ldo = P;
      g.dropout_mode = io.dropout_mode;
      g.masks = io.masks ? io.masks + ((size_t)io.t_rel * 2 + layer) * B * P : nullptr;
      g.seed = io.seed; g.layer = layer; g.keep_scale = keep_scale;
      g.r = d.r; g.d_mel = d.d_mel;
      g.ctrl = ctrl; g.t = io.t;
      launch_gemm(g, A_PLAIN, EPI_RELU_DROPOUT, st);
    }
  }
  if (which < 0 || which == 2) {
    LstmArgs a;
    memset(&a, 0, sizeof(a));
    a.prec = prec;
    if (prec) {
      a.a = make_seg3(sb.xpre_h, P, P, sb.ctx_h, D, D, sb.h_att_h[p], Ha, Ha);
      a.a_lo = make_seg3(sb.xpre_l, P, P, sb.ctx_l, D, D, sb.h_att_l[p], Ha, Ha);
      a.w = make_seg3(plane(bl.att_ih_h), P + D, P, plane(bl.att_ih_h) + P, P + D, D, plane(bl.att_hh_h), Ha, Ha);
      a.w_lo = make_seg3(plane(bl.att_ih_l), P + D, P, plane(bl.att_ih_l) + P, P + D, D, plane(bl.att_hh_l), Ha, Ha);
      a.h_out_h = sb.h_att_h[1 - p]; a.h_out_l = sb.h_att_l[1 - p];
    } else {
      a.a = make_seg3(sb.xpre, P, P, sb.ctx, D, D, sb.h_att[p], Ha, Ha);
      a.w = make_seg3(blob + bl.att_ih, P + D, P, blob + bl.att_ih + P, P + D, D, blob + bl.att_hh, Ha, Ha);
      a.a_lo = a.a; a.w_lo = a.w;
    }
    a.bsum = blob + bl.att_b; a.h_prev = sb.h_att[p]; a.c = sb.c_att; a.h_out = sb.h_att[1 - p];
    a.M = B; a.H = Ha; a.K = P + D + Ha; a.pz = d.p_zoneout; a.ctrl = ctrl; a.t = io.t; a.dbg = io.dbg;
    launch_lstm(a, st);
  }
  if (which < 0 || which == 3) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.a = make_seg1(sb.h_att[1 - p], Ha, Ha);
    g.W = blob + bl.wq; g.ldw = Ha; g.K = Ha; g.M = B; g.N = D; g.out = sb.q; g.ldo = D;
    g.ctrl = ctrl; g.t = io.t;
    launch_gemm(g, A_PLAIN, EPI_PLAIN, st);
  }
  if (which < 0 || which == 4) {
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    if (prec) { a.ctx_h = sb.ctx_h; a.ctx_l = sb.ctx_l; }
    a.memory = io.memory; a.q = sb.q; a.w_prev = sb.w[p]; a.w_new = sb.w[1 - p]; a.w_out = io.w; a.ctx = sb.ctx;
    a.B = B; a.L = io.L; a.D = D; a.t_rel = io.t_rel; a.t_stride = io.t_stride; a.ctrl = ctrl; a.t = io.t;
    launch_attn(a, st);
  }
  if (which < 0 || which == 5) {
    LstmArgs a;
    memset(&a, 0, sizeof(a));
    a.prec = prec;
    if (prec) {
      a.a = make_seg3(sb.h_att_h[1 - p], Ha, Ha, sb.ctx_h, D, D, sb.h_dec_h[p], Hd, Hd);
      a.a_lo = make_seg3(sb.h_att_l[1 - p], Ha, Ha, sb.ctx_l, D, D, sb.h_dec_l[p], Hd, Hd);
      a.w = make_seg3(plane(bl.dec_ih_h), Ha + D, Ha, plane(bl.dec_ih_h) + Ha, Ha + D, D, plane(bl.dec_hh_h), Hd, Hd);
      a.w_lo = make_seg3(plane(bl.dec_ih_l), Ha + D, Ha, plane(bl.dec_ih_l) + Ha, Ha + D, D, plane(bl.dec_hh_l), Hd, Hd);
      a.h_out_h = sb.h_dec_h[1 - p]; a.h_out_l = sb.h_dec_l[1 - p];
    } else {
      a.a = make_seg3(sb.h_att[1 - p], Ha, Ha, sb.ctx, D, D, sb.h_dec[p], Hd, Hd);
      a.w = make_seg3(blob + bl.dec_ih, Ha + D, Ha, blob + bl.dec_ih + Ha, Ha + D, D, blob + bl.dec_hh, Hd, Hd);
      a.a_lo = a.a; a.w_lo = a.w;
    }
    a.bsum = blob + bl.dec_b; a.h_prev = sb.h_dec[p]; a.c = sb.c_dec; a.h_out = sb.h_dec[1 - p];
    a.M = B; a.H = Hd; a.K = Ha + D + Hd; a.pz = d.p_zoneout; a.ctrl = ctrl; a.t = io.t; a.dbg = io.dbg;
    launch_lstm(a, st);
  }
  if (which < 0 || which == 6) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.a = make_seg2(sb.h_dec[1 - p], Hd, Hd, sb.ctx, D, D);
    g.W = blob + bl.proj_w; g.ldw = Hd + D; g.K = Hd + D; g.M = B; g.N = d.r * d.d_mel + d.r;
    g.bias = blob + bl.proj_b;
    g.y_out = io.y; g.s_out = io.s; g.ynext = sb.ynext; g.r = d.r; g.d_mel = d.d_mel;
    g.t_rel = io.t_rel; g.t_stride = io.t_stride; g.stop_thr = io.stop_thr; g.check_stop = io.check_stop && io.use_ctrl;
    g.ctrl = ctrl; g.t = io.t;
    launch_gemm(g, A_PLAIN, EPI_PROJ, st);
  }
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
  int ndev = 0, dev = -1;
  if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0 && hipGetDevice(&dev) == hipSuccess) {
    h->device = dev;
  } else {
    h->device = -1;
    (void)hipGetLastError();
  }
  *out = h;
  return TTSDEC_OK;
}

int ttsdec_destroy(ttsdec_handle* h) {
  delete h;
  return TTSDEC_OK;
}

int ttsdec_num_weight_tensors(const ttsdec_handle* h) {
  if (!h) return TTSDEC_ERR_INVALID_ARG;
  return TTSDEC_W_DECODER_COUNT + (h->d.postnet_layers > 0 ? TTSDEC_W_POSTNET_PER_LAYER * h->d.postnet_layers + 1 : 0);
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
  const size_t Ha = d.h_att, Hd = d.h_dec, D = d.d_ctx, P = d.d_pre, Mel = d.d_mel, R = d.r;
  HIP_TRY(h, hipMemsetAsync(blob, 0, L.total * sizeof(float), st));
  launch_copy(src[TTSDEC_W_PRE0_W], b + L.pre0_w, P * Mel, st);
  launch_copy(src[TTSDEC_W_PRE0_B], b + L.pre0_b, P, st);
  launch_copy(src[TTSDEC_W_PRE1_W], b + L.pre1_w, P * P, st);
  launch_copy(src[TTSDEC_W_PRE1_B], b + L.pre1_b, P, st);
  launch_copy(src[TTSDEC_W_QUERY_W], b + L.wq, D * Ha, st);
  launch_copy(src[TTSDEC_W_ATT_IH], b + L.att_ih, 4 * Ha * (P + D), st);
  launch_copy(src[TTSDEC_W_ATT_HH], b + L.att_hh, 4 * Ha * Ha, st);
  if (src[TTSDEC_W_ATT_BIH] && src[TTSDEC_W_ATT_BHH])
    launch_add_vec(src[TTSDEC_W_ATT_BIH], src[TTSDEC_W_ATT_BHH], b + L.att_b, (int)(4 * Ha), st);
  launch_copy(src[TTSDEC_W_DEC_IH], b + L.dec_ih, 4 * Hd * (Ha + D), st);
  launch_copy(src[TTSDEC_W_DEC_HH], b + L.dec_hh, 4 * Hd * Hd, st);
  if (src[TTSDEC_W_DEC_BIH] && src[TTSDEC_W_DEC_BHH])
    launch_add_vec(src[TTSDEC_W_DEC_BIH], src[TTSDEC_W_DEC_BHH], b + L.dec_b, (int)(4 * Hd), st);
  auto hp = [&](size_t float_off) { return reinterpret_cast<f16*>(b + float_off); };
  launch_split(src[TTSDEC_W_ATT_IH], hp(L.att_ih_h), hp(L.att_ih_l), 4 * Ha * (P + D), st);
  launch_split(src[TTSDEC_W_ATT_HH], hp(L.att_hh_h), hp(L.att_hh_l), 4 * Ha * Ha, st);
  launch_split(src[TTSDEC_W_DEC_IH], hp(L.dec_ih_h), hp(L.dec_ih_l), 4 * Hd * (Ha + D), st);
  launch_split(src[TTSDEC_W_DEC_HH], hp(L.dec_hh_h), hp(L.dec_hh_l), 4 * Hd * Hd, st);
  launch_copy(src[TTSDEC_W_INIT_H0], b + L.h0a, Ha, st);
  launch_copy(src[TTSDEC_W_INIT_C0], b + L.c0a, Ha, st);
  launch_copy(src[TTSDEC_W_INIT_H1], b + L.h0d, Hd, st);
  launch_copy(src[TTSDEC_W_INIT_C1], b + L.c0d, Hd, st);
  // [fc_mel ; fc_stop] stacked on the output axis (decoder.py:13-14)
  launch_copy(src[TTSDEC_W_MEL_W], b + L.proj_w, R * Mel * (Hd + D), st);
  launch_copy(src[TTSDEC_W_STOP_W], b + L.proj_w + R * Mel * (Hd + D), R * (Hd + D), st);
  launch_copy(src[TTSDEC_W_MEL_B], b + L.proj_b, R * Mel, st);
  launch_copy(src[TTSDEC_W_STOP_B], b + L.proj_b + R * Mel, R, st);
  int cin = d.d_mel;
  for (int i = 0; i < d.postnet_layers; ++i) {
    const float* const* ps = src + TTSDEC_W_DECODER_COUNT + TTSDEC_W_POSTNET_PER_LAYER * i;
    if (!ps[0] || !ps[1] || !ps[2] || !ps[3] || !ps[4]) return TTSDEC_ERR_INVALID_ARG;
    launch_conv_transpose(ps[0], b + L.conv_w[i], d.postnet_hidden, cin, d.postnet_kernel, st);
    launch_bn_fold(ps[1], ps[2], ps[3], ps[4], d.bn_eps, b + L.conv_alpha[i], b + L.conv_beta[i], d.postnet_hidden, st);
    cin = d.postnet_hidden;
  }
  if (d.postnet_layers > 0)
    launch_copy(src[TTSDEC_W_DECODER_COUNT + TTSDEC_W_POSTNET_PER_LAYER * d.postnet_layers], b + L.fc_w,
                Mel * (size_t)d.postnet_hidden, st);
  rc = check_launch(h, "pack_weights");
  if (rc != TTSDEC_OK) return rc;
  h->blob = b;
  return TTSDEC_OK;
}

int ttsdec_bind_weights(ttsdec_handle* h, const void* blob) {
  if (!h || !blob) return TTSDEC_ERR_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(blob) & 255) return TTSDEC_ERR_WORKSPACE;
  h->blob = static_cast<const float*>(blob);
  return TTSDEC_OK;
}

size_t ttsdec_workspace_bytes(const ttsdec_handle* h, int B, int L) {
  if (!h || B <= 0 || L <= 0) return 0;
  return make_ws_layout(h->d, B, L).total;
}

int ttsdec_decode(ttsdec_handle* h, const float* memory, int B, int L, int t_begin, int n_steps, int t_stride,
                  float stop_threshold, int check_stop, int dropout_mode, const uint8_t* masks, uint64_t seed,
                  const float* teacher, int teacher_T, const uint8_t* teacher_flags, float* y, float* s, float* w,
                  int32_t* T_out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!h || !memory || !y || !s || !w || !workspace) return TTSDEC_ERR_INVALID_ARG;
  if (B <= 0 || L <= 0 || t_begin < 0 || n_steps < 0 || t_stride < n_steps) return TTSDEC_ERR_INVALID_ARG;
  if (dropout_mode < TTSDEC_DROPOUT_OFF || dropout_mode > TTSDEC_DROPOUT_PHILOX) return TTSDEC_ERR_INVALID_ARG;
  if (dropout_mode == TTSDEC_DROPOUT_MASKS && !masks) return TTSDEC_ERR_INVALID_ARG;
  if (teacher && (!teacher_flags || teacher_T < (t_begin + n_steps - 1) * h->d.r)) return TTSDEC_ERR_INVALID_ARG;
  if (!h->blob) return TTSDEC_ERR_NOT_BOUND;
  int rc = check_device(h);
  if (rc != TTSDEC_OK) return rc;
  const WsLayout W = make_ws_layout(h->d, B, L);
  if (workspace_bytes < W.total || (reinterpret_cast<uintptr_t>(workspace) & 255)) return TTSDEC_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const StepBufs sb = carve(W, workspace);
  const ttsdec_dims& d = h->d;

  if (t_begin == 0) {
    InitArgs ia;
    ia.ctrl = sb.ctrl;
    ia.h0_att = h->blob + h->bl.h0a; ia.c0_att = h->blob + h->bl.c0a;
    ia.h0_dec = h->blob + h->bl.h0d; ia.c0_dec = h->blob + h->bl.c0d;
    ia.h_att = sb.h_att[0]; ia.c_att = sb.c_att; ia.h_dec = sb.h_dec[0]; ia.c_dec = sb.c_dec;
    ia.ctx = sb.ctx; ia.w = sb.w[0]; ia.ynext = sb.ynext;
    ia.h_att_h = sb.h_att_h[0]; ia.h_att_l = sb.h_att_l[0]; ia.h_dec_h = sb.h_dec_h[0]; ia.h_dec_l = sb.h_dec_l[0];
    ia.ctx_h = sb.ctx_h; ia.ctx_l = sb.ctx_l;
    ia.B = B; ia.L = L; ia.D = d.d_ctx; ia.Ha = d.h_att; ia.Hd = d.h_dec; ia.d_mel = d.d_mel;
    launch_init(ia, st);
  }
  StepIo io;
  io.memory = memory; io.B = B; io.L = L; io.t_stride = t_stride;
  io.stop_thr = stop_threshold; io.check_stop = check_stop;
  io.dropout_mode = dropout_mode; io.masks = masks; io.seed = seed;
  io.teacher = teacher; io.teacher_T = teacher_T; io.teacher_flags = teacher_flags;
  io.y = y; io.s = s; io.w = w; io.use_ctrl = true; io.dbg = 0;
  for (int i = 0; i < n_steps; ++i) {
    io.t = t_begin + i;
    io.t_rel = i;
    launch_step(h, sb, io, -1, st);
  }
  launch_finish(sb.ctrl, t_begin + n_steps, T_out, st);
  return check_launch(h, "decode");
}

size_t ttsdec_postnet_workspace_bytes(const ttsdec_handle* h, int B, int T) {
  if (!h || B <= 0 || T <= 0 || h->d.postnet_layers <= 0) return 0;
  return 2 * align_up((size_t)B * T * h->d.postnet_hidden * sizeof(float), 256);
}

int ttsdec_postnet(ttsdec_handle* h, const float* y, int B, int T, int precision, float* y_post, void* workspace,
                   size_t workspace_bytes, void* stream) {
  if (!h || !y || !y_post || !workspace || B <= 0 || T <= 0) return TTSDEC_ERR_INVALID_ARG;
  if (h->d.postnet_layers <= 0) return TTSDEC_ERR_DIMS;
  if (precision != TTSDEC_POSTNET_F32) return TTSDEC_ERR_INVALID_ARG;  // bf16 path: not built yet
  if (!h->blob) return TTSDEC_ERR_NOT_BOUND;
  int rc = check_device(h);
  if (rc != TTSDEC_OK) return rc;
  if (workspace_bytes < ttsdec_postnet_workspace_bytes(h, B, T) || (reinterpret_cast<uintptr_t>(workspace) & 255))
    return TTSDEC_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const ttsdec_dims& d = h->d;
  const size_t half = align_up((size_t)B * T * d.postnet_hidden * sizeof(float), 256);
  float* act[2] = {static_cast<float*>(workspace), reinterpret_cast<float*>(static_cast<char*>(workspace) + half)};
  const float* in = y;
  int cin = d.d_mel;
  for (int i = 0; i < d.postnet_layers; ++i) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.a = make_seg1(in, cin, cin);
    g.T = T; g.Cin = cin; g.taps = d.postnet_kernel;
    g.W = h->blob + h->bl.conv_w[i]; g.ldw = d.postnet_kernel * cin; g.K = d.postnet_kernel * cin;
    g.M = B * T; g.N = d.postnet_hidden;
    g.alpha = h->blob + h->bl.conv_alpha[i]; g.beta = h->blob + h->bl.conv_beta[i];
    g.out = act[i & 1]; g.ldo = d.postnet_hidden;
    launch_gemm(g, A_CONV, EPI_BN_ISRU, st);
    in = act[i & 1];
    cin = d.postnet_hidden;
  }
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.a = make_seg1(in, cin, cin);
  g.W = h->blob + h->bl.fc_w; g.ldw = cin; g.K = cin; g.M = B * T; g.N = d.d_mel;
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
  // caller state -> workspace (the step kernels ping-pong h and w)
  launch_copy(x, sb.ynext, b * d.d_mel, st);
  launch_copy(ctx, sb.ctx, b * d.d_ctx, st);
  launch_copy(w, sb.w[p], b * L, st);
  launch_copy(h_att, sb.h_att[p], b * d.h_att, st);
  launch_copy(c_att, sb.c_att, b * d.h_att, st);
  launch_copy(h_dec, sb.h_dec[p], b * d.h_dec, st);
  launch_copy(c_dec, sb.c_dec, b * d.h_dec, st);
  if (lstm_prec(h)) {
    launch_split(ctx, sb.ctx_h, sb.ctx_l, b * d.d_ctx, st);
    launch_split(h_att, sb.h_att_h[p], sb.h_att_l[p], b * d.h_att, st);
    launch_split(h_dec, sb.h_dec_h[p], sb.h_dec_l[p], b * d.h_dec, st);
  }
  StepIo io;
  memset(&io, 0, sizeof(io));
  io.memory = memory; io.B = B; io.L = L; io.t = step; io.t_rel = 0; io.t_stride = 1;
  io.dropout_mode = dropout_mode; io.masks = masks; io.seed = seed;
  // y/s of the projection are not part of the cell: park them in x_dec's tail? no - run steps 0..5 only
  io.use_ctrl = false;
  for (int k = 0; k < 6; ++k) launch_step(h, sb, io, k, st);
  launch_copy(sb.ctx, ctx, b * d.d_ctx, st);
  launch_copy(sb.w[1 - p], w, b * L, st);
  launch_copy(sb.h_att[1 - p], h_att, b * d.h_att, st);
  launch_copy(sb.c_att, c_att, b * d.h_att, st);
  launch_copy(sb.h_dec[1 - p], h_dec, b * d.h_dec, st);
  launch_copy(sb.c_dec, c_dec, b * d.h_dec, st);
  // x_dec = cat[h_dec, ctx] (decoder_cell.py:192)
  HIP_TRY(h, hipMemcpy2DAsync(x_dec, (size_t)(d.h_dec + d.d_ctx) * sizeof(float), sb.h_dec[1 - p], (size_t)d.h_dec * sizeof(float),
                              (size_t)d.h_dec * sizeof(float), b, hipMemcpyDeviceToDevice, st));
  HIP_TRY(h, hipMemcpy2DAsync(x_dec + d.h_dec, (size_t)(d.h_dec + d.d_ctx) * sizeof(float), sb.ctx, (size_t)d.d_ctx * sizeof(float),
                              (size_t)d.d_ctx * sizeof(float), b, hipMemcpyDeviceToDevice, st));
  return check_launch(h, "cell_step");
}

int ttsdec_profile_step(ttsdec_handle* h, const float* memory, int B, int L, int iters, int dropout_mode,
                        const uint8_t* masks, uint64_t seed, float* y, float* s, float* w, void* workspace,
                        size_t workspace_bytes, void* stream, float* ms_out, const char** names_out, int n_out,
                        int* n_kernels) {
  if (!h || !memory || !y || !s || !w || !workspace || !ms_out || iters <= 0) return TTSDEC_ERR_INVALID_ARG;
  if (n_kernels) *n_kernels = kKernelsPerStep;
  if (n_out < kKernelsPerStep) return TTSDEC_ERR_INVALID_ARG;
  if (dropout_mode == TTSDEC_DROPOUT_MASKS && !masks) return TTSDEC_ERR_INVALID_ARG;
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
  if (const char* e = getenv("TTSDEC_PROFILE_ABLATION")) io.dbg = atoi(e);  // measurement only
  hipEvent_t e0, e1;
  HIP_TRY(h, hipEventCreate(&e0));
  HIP_TRY(h, hipEventCreate(&e1));
  for (int k = 0; k < kKernelsPerStep; ++k) {
    for (int i = 0; i < 3; ++i) launch_step(h, sb, io, k, st);  // warm
    HIP_TRY(h, hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) launch_step(h, sb, io, k, st);
    HIP_TRY(h, hipEventRecord(e1, st));
    HIP_TRY(h, hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&ms, e0, e1));
    ms_out[k] = ms / iters;
    if (names_out) names_out[k] = kKernelNames[k];
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return check_launch(h, "profile_step");
}

}  // extern "C"
