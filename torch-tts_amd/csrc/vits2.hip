// C ABI of the VITS2 second hot path (BASELINE.json configs[4]):
//   ttsvits_text_encoder   TextEncoder.forward                      vits2/models.py:369-380
//   ttsvits_flow_reverse   ResidualCouplingTransformersBlock.forward(reverse=True)   models.py:506-531, 803-810
// built on the decoder path's GEMM core (1x1 convs are row GEMMs, k-tap convs implicit GEMMs over
// channel-last activations, bias / ReLU / frame mask / residual in the epilogue) plus four small
// kernels of its own: channel LayerNorm, relative-position multi-head attention, the WN gate and the
// coupling update.  Activations are channel-last [B*T, C] (one row per frame) throughout - the
// reference's [B, C, T] is transposed once at the boundary by the host module.  GEMMs run in the
// split-fp16 mode of the GEMM core (fp32-class accuracy, gemm_tile.h); everything else is fp32.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>
#include <type_traits>

#include "kernels.h"

using namespace ttsdec;

namespace {
constexpr size_t kAlign = 64;  // floats
constexpr int kMaxLayers = 12, kMaxFlows = 8, kMaxWn = 8;
inline size_t up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// one attentions.Encoder stack (attentions.py:14-93)
struct StackBlob {
  size_t wqkv[kMaxLayers], bqkv[kMaxLayers], wo[kMaxLayers], bo[kMaxLayers], ek[kMaxLayers], ev[kMaxLayers];
  size_t g1[kMaxLayers], b1[kMaxLayers], w1[kMaxLayers], c1[kMaxLayers], w2[kMaxLayers], c2[kMaxLayers], g2[kMaxLayers], b2[kMaxLayers];
};
struct StackDims {
  int C, F, heads, layers, kernel, window;  // window < 0: no relative-position terms
};
struct FlowBlob {
  StackBlob tf;
  size_t pre_w, pre_b, in_w[kMaxWn], in_b[kMaxWn], rs_w[kMaxWn], rs_b[kMaxWn], post_w, post_b;
  size_t cond_w, cond_b;  // WN.cond_layer (gin_channels > 0): [2 Fh n_layers, gin], [2 Fh n_layers]
};
struct VitsBlob {  // offsets in floats
  size_t emb;
  size_t spk_w, spk_b;  // encoder.spk_emb_linear (gin_channels > 0): [hidden, gin], [hidden]
  StackBlob enc;
  size_t proj_w, proj_b;
  FlowBlob flow[kMaxFlows];
  size_t total;
};
}  // namespace

struct ttsvits_handle {
  ttsvits_dims d;
  VitsBlob bl;
  const float* blob;
  int device;  // HIP device current at create (-1: none); must be current for every later call
  int precision;  // TTSDEC_PREC_F32 (default) or TTSDEC_PREC_SPLIT_F16: arithmetic of every GEMM (ttsvits_set_precision)
  std::string hip_err;
};

namespace {

StackDims enc_dims(const ttsvits_dims& d) { return {d.hidden_channels, d.filter_channels, d.n_heads, d.n_layers, d.kernel_size, d.window_size}; }
StackDims tf_dims(const ttsvits_dims& d) {
  const int half = d.inter_channels / 2;
  return {half, half, d.flow_tf_heads, d.flow_tf_layers, d.flow_tf_kernel, -1};
}

VitsBlob make_layout(const ttsvits_dims& d) {
  VitsBlob L;
  memset(&L, 0, sizeof(L));
  size_t off = 0;
  auto take = [&](size_t n) { const size_t o = off; off = up(off + n, kAlign); return o; };
  // a GEMM weight of n elements occupies 2n floats: fp32 | fp16 hi plane | fp16 lo plane (w_hi / w_lo below)
  auto take_w = [&](size_t n) { return take(2 * n); };
  auto stack = [&](StackBlob& s, const StackDims& sd) {
    const size_t C = sd.C, F = sd.F, k = sd.kernel, dk = sd.C / sd.heads;
    for (int i = 0; i < sd.layers; ++i) {
      s.wqkv[i] = take_w(3 * C * C); s.bqkv[i] = take(3 * C); s.wo[i] = take_w(C * C); s.bo[i] = take(C);
      if (sd.window >= 0) { s.ek[i] = take((2 * sd.window + 1) * dk); s.ev[i] = take((2 * sd.window + 1) * dk); }
      s.g1[i] = take(C); s.b1[i] = take(C);
      s.w1[i] = take_w(F * k * C); s.c1[i] = take(F); s.w2[i] = take_w(C * k * F); s.c2[i] = take(C);
      s.g2[i] = take(C); s.b2[i] = take(C);
    }
  };
  const size_t H = d.hidden_channels, I = d.inter_channels, half = I / 2, Fh = d.flow_hidden;
  L.emb = take((size_t)d.n_vocab * H);
  if (d.gin_channels > 0) { L.spk_w = take_w(H * d.gin_channels); L.spk_b = take(H); }
  stack(L.enc, enc_dims(d));
  L.proj_w = take_w(2 * I * H); L.proj_b = take(2 * I);
  for (int f = 0; f < d.n_flows; ++f) {
    FlowBlob& fb = L.flow[f];
    stack(fb.tf, tf_dims(d));
    fb.pre_w = take_w(Fh * half); fb.pre_b = take(Fh);
    if (d.gin_channels > 0) { fb.cond_w = take_w(2 * Fh * d.flow_wn_layers * d.gin_channels); fb.cond_b = take(2 * Fh * d.flow_wn_layers); }
    for (int j = 0; j < d.flow_wn_layers; ++j) {
      const size_t cr = j < d.flow_wn_layers - 1 ? 2 * Fh : Fh;
      fb.in_w[j] = take_w(2 * Fh * d.flow_kernel * Fh); fb.in_b[j] = take(2 * Fh);
      fb.rs_w[j] = take_w(cr * Fh); fb.rs_b[j] = take(cr);
    }
    fb.post_w = take_w(half * Fh); fb.post_b = take(half);
  }
  L.total = off;
  return L;
}

int n_stack_tensors(const StackDims& sd) { return sd.layers * (sd.window >= 0 ? 18 : 16); }
int n_cond_tensors(const ttsvits_dims& d) { return d.gin_channels > 0 ? 2 : 0; }
int n_text_tensors(const ttsvits_dims& d) { return 1 + n_cond_tensors(d) + n_stack_tensors(enc_dims(d)) + 2; }
int n_flow_tensors(const ttsvits_dims& d) {
  return d.n_flows * (n_stack_tensors(tf_dims(d)) + 2 + n_cond_tensors(d) + 4 * d.flow_wn_layers + 2);
}

int vits_fail(ttsvits_handle* h, const char* where) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return TTSDEC_OK;
  if (h) h->hip_err = std::string(where) + ": " + hipGetErrorString(e);
  return TTSDEC_ERR_HIP;
}

// ===========================================================================
// kernels
// ===========================================================================
// TextEncoder.forward, models.py:370-376: x = emb(ids) * sqrt(H), masked; also the per-frame mask
// (every producer below can also emit the split-fp16 planes of its output - hi at p, lo at p + M*C - so
// the GEMM that consumes it needs no separate conversion pass)
__global__ void embed_scale_kernel(const long long* ids, const int* lengths, const float* table, int n_vocab, int T, int H, float scale,
                                   float* x, f16* xp, float* mask, int M, int* status) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * H) return;
  const int m = (int)(i / H), c = (int)(i % H);
  const int b = m / T, t = m - b * T;
  const float mk = t < lengths[b] ? 1.0f : 0.0f;
  long long id = ids[m];
  if (id < 0 || id >= n_vocab) {  // (nn.Embedding raises there: clamp - never read outside the table - and report, see embed_kernel)
    if (c == 0 && status != nullptr) atomicOr(status, 1);
    id = id < 0 ? 0 : n_vocab - 1;
  }
  const float v = mul_rn(mul_rn(table[(size_t)id * H + c], scale), mk);
  x[i] = v;
  if (xp) split_f16(v, xp[i], xp[(size_t)M * H + i]);
  if (c == 0) mask[m] = mk;
}
__global__ void frame_mask_kernel(const int* lengths, int T, float* mask, int M) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m < M) mask[m] = (m % T) < lengths[m / T] ? 1.0f : 0.0f;
}

// modules.LayerNorm (modules.py:24-27) over the channels of one frame; one wave per row.
// out = LN(in) ; out_m = LN(in) * mask  (either may be nullptr)
// NJ: 64-channel groups a lane walks (C <= 64 NJ; the 96- and 192-channel stacks of the VITS2 path take 2 and 4 instead of 16
// predicated rounds per pass)
template <int NJ>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float* in, const float* gamma, const float* beta, const float* mask,
                                                             float* out, float* out_m, f16* out_p, f16* outm_p, int M, int C, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* x = in + (size_t)row * C;
  float v[NJ];  // C <= 64 NJ
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    v[j] = c < C ? x[c] : 0.f;
    s += v[j];
  }
  s = wave_sum(s);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    const float dlt = c < C ? v[j] - mean : 0.f;
    q += dlt * dlt;
  }
  q = wave_sum(q);
  const float rstd = 1.0f / sqrt_rn(add_rn(q / (float)C, eps));
  const float mk = mask ? mask[row] : 1.0f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    if (c < C) {
      const float y = add_rn(mul_rn(mul_rn(v[j] - mean, rstd), gamma[c]), beta[c]);
      const size_t o = (size_t)row * C + c, n = (size_t)M * C;
      if (out) out[o] = y;
      if (out_m) out_m[o] = mul_rn(y, mk);
      if (out_p) split_f16(y, out_p[o], out_p[n + o]);
      if (outm_p) split_f16(mul_rn(y, mk), outm_p[o], outm_p[n + o]);
    }
  }
}

// MultiHeadAttention.attention (attentions.py:246-295), self-attention with the frame mask and the
// optional relative-position window.  One workgroup = 16 query frames of one (utterance, head).
//   scores[i, j] = (q_i / sqrt(dk)) . k_j  [+ (q_i / sqrt(dk)) . E_k[j - i + w] if |j - i| <= w]
//   scores[i, j] = -1e4 where mask_i * mask_j == 0 ; p = softmax_j
//   out[i] = sum_j p[i, j] v_j  [+ sum_{|j-i|<=w} p[i, j] E_v[j - i + w]]
constexpr int kMhaRows = 16, kMhaChunk = 64, kMhaThreads = 256;
struct MhaArgs {
  const float* qkv;  // [B*T, 3C]: q | k | v, head h = channels [h*dk, (h+1)*dk)
  const float* mask;  // [B*T]
  const float *ek, *ev;  // [2w+1, dk] or nullptr
  float* out;            // [B*T, C]
  f16* out_p;            // optional split-fp16 planes of out (hi, then lo n_out halfs later)
  size_t n_out;          // B*T*C
  int planes_only;       // flash kernel: 1 = write the planes only (their consumer, the output GEMM in split-fp16 mode, reads nothing else)
  int T, C, dk, window;
  float qscale;  // sqrt(dk): q is DIVIDED by it, as the reference does
};
__global__ __launch_bounds__(kMhaThreads) void mha_kernel(MhaArgs g) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T = g.T, dk = g.dk, C = g.C, w = g.window;
  const int KS = dk + 1;
  float* qs = sm;                           // [16][dk]
  float* kv = qs + kMhaRows * dk;           // [64][dk+1]
  float* rel = kv + kMhaChunk * KS;         // [2][2w+1][dk]  (E_k, E_v)
  const int nrel = w >= 0 ? 2 * w + 1 : 0;
  float* S = rel + 2 * nrel * dk;           // [16][T]
  const int tid = threadIdx.x, b = blockIdx.z, hd = blockIdx.y, i0 = blockIdx.x * kMhaRows;
  const size_t rowb = (size_t)b * T;
  const float* base = g.qkv + rowb * 3 * C + hd * dk;
  for (int e = tid; e < kMhaRows * dk; e += kMhaThreads) {
    const int i = e / dk, d = e - i * dk;
    qs[e] = (i0 + i < T) ? div_rn(base[(size_t)(i0 + i) * 3 * C + d], g.qscale) : 0.f;
  }
  for (int e = tid; e < nrel * dk; e += kMhaThreads) {
    rel[e] = g.ek[e];
    rel[nrel * dk + e] = g.ev[e];
  }
  const int ri = tid >> 4, rj = tid & 15;  // this thread's query row / key lane
  const float mi = (i0 + ri < T) ? g.mask[rowb + i0 + ri] : 0.f;
  // ---- pass 1: scores ----
  for (int c0 = 0; c0 < T; c0 += kMhaChunk) {
    __syncthreads();
    for (int e = tid; e < kMhaChunk * dk; e += kMhaThreads) {
      const int j = e / dk, d = e - j * dk;
      kv[j * KS + d] = (c0 + j < T) ? base[(size_t)(c0 + j) * 3 * C + C + d] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < kMhaChunk / 16; ++jj) {
      const int jl = rj + 16 * jj, j = c0 + jl;
      if (j >= T) continue;
      const float* qr = qs + ri * dk;
      const float* kr = kv + jl * KS;
      float s = 0.f;
      for (int d = 0; d < dk; ++d) s = fmaf(qr[d], kr[d], s);
      const int r = j - (i0 + ri) + w;
      if (w >= 0 && r >= 0 && r <= 2 * w) {
        const float* er = rel + r * dk;
        float s2 = 0.f;
        for (int d = 0; d < dk; ++d) s2 = fmaf(qr[d], er[d], s2);
        s = add_rn(s, s2);
      }
      if (mi * g.mask[rowb + j] == 0.f) s = -1e4f;
      S[ri * T + j] = s;
    }
  }
  __syncthreads();
  // ---- softmax: wave v owns rows 4v .. 4v+3 ----
  {
    const int wave = tid >> 6, lane = tid & 63;
    for (int rr = 0; rr < 4; ++rr) {
      float* row = S + (wave * 4 + rr) * T;
      float mx = -3.4e38f;
      for (int j = lane; j < T; j += 64) mx = fmaxf(mx, row[j]);
      mx = wave_max(mx);
      float sum = 0.f;
      for (int j = lane; j < T; j += 64) {
        const float e = expf(row[j] - mx);
        row[j] = e;
        sum += e;
      }
      sum = wave_sum(sum);
      for (int j = lane; j < T; j += 64) row[j] = div_rn(row[j], sum);
    }
  }
  // ---- pass 2: out = p v (+ relative values) ----
  constexpr int ND = 16;  // dk <= 256
  float acc[ND];
#pragma unroll
  for (int dd = 0; dd < ND; ++dd) acc[dd] = 0.f;
  for (int c0 = 0; c0 < T; c0 += kMhaChunk) {
    __syncthreads();
    for (int e = tid; e < kMhaChunk * dk; e += kMhaThreads) {
      const int j = e / dk, d = e - j * dk;
      kv[j * KS + d] = (c0 + j < T) ? base[(size_t)(c0 + j) * 3 * C + 2 * C + d] : 0.f;
    }
    __syncthreads();
    const int jn = (T - c0) < kMhaChunk ? (T - c0) : kMhaChunk;
    const float* pr = S + ri * T + c0;
    for (int j = 0; j < jn; ++j) {
      const float p = pr[j];
#pragma unroll
      for (int dd = 0; dd < ND; ++dd) {
        const int d = rj + 16 * dd;
        if (d < dk) acc[dd] = fmaf(p, kv[j * KS + d], acc[dd]);
      }
    }
  }
  if (w >= 0) {
    for (int r = 0; r <= 2 * w; ++r) {
      const int j = i0 + ri + r - w;
      if (j < 0 || j >= T) continue;
      const float p = S[ri * T + j];
#pragma unroll
      for (int dd = 0; dd < ND; ++dd) {
        const int d = rj + 16 * dd;
        if (d < dk) acc[dd] = fmaf(p, rel[(nrel + r) * dk + d], acc[dd]);
      }
    }
  }
  if (i0 + ri < T) {
#pragma unroll
    for (int dd = 0; dd < ND; ++dd) {
      const int d = rj + 16 * dd;
      if (d < dk) {
        const size_t o = (rowb + i0 + ri) * C + hd * dk + d;
        g.out[o] = acc[dd];
        if (g.out_p) split_f16(acc[dd], g.out_p[o], g.out_p[g.n_out + o]);
      }
    }
  }
}

// The same attention on the matrix cores (exact fp32 32x32x2 MFMAs).  One workgroup = 32 query
// frames of one (utterance, head); the 4 waves split the keys.
//   pass 1: S[32, T] = Q K^T, 32 keys per MFMA tile; both operands go global -> registers directly
//           (a lane holds a contiguous half of its row's dk values: the K index of an fp32 MFMA may
//           be permuted freely as long as A and B agree); relative-key logits R = Q E_k^T the same way
//   softmax over the rows of S in LDS
//   pass 2: out[32, dk] = P V: P from LDS, V from global (a lane owns NDT ADJACENT output columns, so its
//           values of one key are one load), keys split over the waves, partial tiles reduced through LDS
//           in wave order
// How the loads and the instruction stream are arranged (tools/ubench_mha.hip has the shipped-before
// kernel next to this one, pass by pass; profiles/r02_j_ubench_mha.txt the numbers):
//  * every global load a wave needs before its first use is requested up front (Q, the row masks, the first K tiles with
//    their key masks, E_k, E_v); barriers that only order LDS do not drain the vector-memory queue
//  * pass 1: DK1 key tiles in flight per wave, the key's mask value travels with its tile (a global load inside the tile's
//    epilogue would wait for every prefetched tile behind it: one in-order queue)
//  * softmax: the 8 rows of a wave side by side
//  * pass 2: DV groups of V values in flight per wave
//  * E_v staged in LDS for the output loop
constexpr int kMhaMRows = 32, kMhaMThreads = 256;
template <int DKH>  // dk / 2 (a multiple of 4)
__global__ __launch_bounds__(kMhaMThreads, 2) void mha_mfma_kernel(MhaArgs g) {
  constexpr int DK = 2 * DKH, NDT = (DK + 31) / 32, NQ = DKH / 4;
  constexpr int DK1 = DKH > 24 ? 2 : 4, DV = DKH > 24 ? 2 : 4, G = 8;  // tiles / groups in flight per wave
  constexpr int NEV = (31 * DK + kMhaMThreads - 1) / kMhaMThreads;  // E_v values per thread (window <= 15)
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T = g.T, C = g.C, w = g.window;
  const int ST = T | 1;
  float* S = sm;
  float* R = S + kMhaMRows * ST;
  const int nrel = w >= 0 ? 2 * w + 1 : 0;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l32 = lane & 31, half = lane >> 5;
  const int b = blockIdx.z, hd = blockIdx.y, i0 = blockIdx.x * kMhaMRows;
  const size_t rowb = (size_t)b * T;
  const float* base = g.qkv + rowb * 3 * C + hd * DK;
  typedef __attribute__((address_space(1))) const f32x4 gf32x4;
  typedef __attribute__((address_space(1))) const float gf32;
  gf32* maskg = (gf32*)(g.mask + rowb);
  const int ntile = (T + 31) / 32;

  f32x4 qa[NQ];
  {
    const int i = i0 + l32 < T ? i0 + l32 : T - 1;
    gf32x4* src = (gf32x4*)(base + (size_t)i * 3 * C + half * DKH);
#pragma unroll
    for (int j = 0; j < NQ; ++j) qa[j] = src[j];
  }
  f32x4 e4[NQ];
  if (w >= 0 && wave == 0) {
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      e4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (l32 < nrel) e4[j] = *(gf32x4*)(g.ek + (size_t)l32 * DK + half * DKH + 4 * j);
    }
  }
  float mrow[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * half;
    mrow[r] = i < T ? maskg[i] : 0.f;
  }
  f32x4 kb[DK1][NQ];
  float mj[DK1];
  auto load_k = [&](int kt, int u) {
    const int j = kt * 32 + l32, jc = j < T ? j : T - 1;
    gf32x4* src = (gf32x4*)(base + (size_t)jc * 3 * C + C + half * DKH);
#pragma unroll
    for (int q = 0; q < NQ; ++q) kb[u][q] = src[q];
    mj[u] = maskg[jc];
  };
#pragma unroll
  for (int u = 0; u < DK1; ++u) load_k(wave + 4 * u, u);  // past the last tile: the clamped row again (no branch, so the
                                                           // compiler can count the loads in flight exactly)
  float evr[NEV];
#pragma unroll
  for (int q = 0; q < NEV; ++q) {
    const int e = tid + q * kMhaMThreads;
    evr[q] = (w >= 0 && e < nrel * DK) ? ((gf32*)g.ev)[e] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < NQ; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) qa[j][e] = div_rn(qa[j][e], g.qscale);
  if (w >= 0 && wave == 0) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NQ; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[j][e], e4[j][e], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) R[((r & 3) + 8 * (r >> 2) + 4 * half) * 33 + l32] = acc[r];
  }
  lds_barrier();
  // ---- pass 1 ----
  auto score_tile = [&](int kt, int u) {
    const int j = kt * 32 + l32;
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[q][e], kb[u][q][e], acc, 0, 0, 0);
    if (j < T) {
      const float mjv = mj[u];
      float* sj = S + j;
      // does any (query, key) pair of this tile lie inside the relative window?  (the same answer in every lane)
      const bool near = w >= 0 && kt * 32 + 31 + w >= i0 && kt * 32 <= i0 + 31 + w;
      if (near) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
          float s = acc[r];
          const int rr = j - (i0 + row) + w;
          if (rr >= 0 && rr <= 2 * w) s = add_rn(s, R[row * 33 + rr]);
          if (mrow[r] * mjv == 0.f) s = -1e4f;
          sj[row * ST] = s;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
          sj[row * ST] = mrow[r] * mjv == 0.f ? -1e4f : acc[r];
        }
      }
    }
  };
  for (int kt0 = wave; kt0 < ntile; kt0 += 4 * DK1) {
#pragma unroll
    for (int u = 0; u < DK1; ++u) {
      const int kt = kt0 + 4 * u;
      if (kt < ntile) score_tile(kt, u);
      load_k(kt + 4 * DK1, u);
    }
  }
  lds_barrier();
  // ---- softmax numerators, the wave's 8 rows side by side ----
  float* rinv = S + kMhaMRows * ST + (w >= 0 ? 32 * 33 : 0);
  {
    float* row0 = S + (wave * 8) * ST;
    float mx[8], sum[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) mx[q] = -3.4e38f, sum[q] = 0.f;
    for (int j = lane; j < T; j += 64)
#pragma unroll
      for (int q = 0; q < 8; ++q) mx[q] = fmaxf(mx[q], row0[q * ST + j]);
#pragma unroll
    for (int q = 0; q < 8; ++q) mx[q] = wave_max(mx[q]);
    for (int j = lane; j < T; j += 64)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float e = __expf(row0[q * ST + j] - mx[q]);
        row0[q * ST + j] = e;
        sum[q] += e;
      }
#pragma unroll
    for (int q = 0; q < 8; ++q) sum[q] = wave_sum(sum[q]);
    if (lane == 0)
#pragma unroll
      for (int q = 0; q < 8; ++q) rinv[wave * 8 + q] = 1.0f / sum[q];
  }
  lds_barrier();
  // ---- pass 2 ----
  f32x16 acc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) acc[dt] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  {
    gf32* vb = (gf32*)(base + 2 * C);
    const float* prow = S + l32 * ST;
    const int npair = (T + 1) / 2;
    struct VN { float v[NDT]; };  // DK % NDT == 0 for the built sizes: a lane's columns are all inside or all outside
    static_assert(DK % NDT == 0, "");
    float pv[DV][G], vv[DV][G][NDT];
    // one key pair of a group: its P value (LDS) and the lane's NDT adjacent V values (one global load)
    // lanes whose columns lie past DK read column 0 instead: their accumulators are never stored
    gf32* vlane = vb + (NDT * l32 < DK ? NDT * l32 : 0) + (size_t)half * 3 * C;
    const float* plane = prow + half;
    auto load_1 = [&](int p0, int s, int u) {
      if (2 * (p0 + G) <= T) {  // every key of the group exists (the same answer in all lanes): no predicates
        pv[s][u] = plane[2 * (p0 + u)];
        gf32* vk = vlane + (size_t)(2 * (p0 + u)) * 3 * C;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) vv[s][u][dt] = vk[dt];
      } else {
        const int key = 2 * (p0 + u) + half;
        const bool ok = key < T;
        const int kc = ok ? key : T - 1;
        const float pl = prow[kc];
        gf32* vk = vlane + (size_t)(kc - half) * 3 * C;
        pv[s][u] = ok ? pl : 0.f;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const float vl = vk[dt];
          vv[s][u][dt] = ok ? vl : 0.f;
        }
      }
    };
#pragma unroll
    for (int s = 0; s < DV; ++s)
      if (wave * G + s * 4 * G < npair) {
#pragma unroll
        for (int u = 0; u < G; ++u) load_1(wave * G + s * 4 * G, s, u);
      }
    for (int pb = wave * G; pb < npair; pb += DV * 4 * G) {
#pragma unroll
      for (int s = 0; s < DV; ++s) {
        const int p0 = pb + s * 4 * G;
        if (p0 < npair) {
          const bool more = p0 + DV * 4 * G < npair;
          // the slot's next key pair is requested right behind the MFMAs that consumed the old one, so the address
          // arithmetic runs while the matrix pipe is busy instead of after the whole group
#pragma unroll
          for (int u = 0; u < G; ++u) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(pv[s][u], vv[s][u][dt], acc[dt], 0, 0, 0);
            if (more) load_1(p0 + DV * 4 * G, s, u);
          }
        }
      }
    }
  }
  // relative values need p[i, i + r - w]: read them (and the row's 1/sum) before S is reused as the reduction buffer.
  // NR: compile-time bound on the window rows (9 covers the reference's window_size = 4)
  auto finish = [&](auto nr_c) {
    constexpr int NR = decltype(nr_c)::value;
    const int oi = tid >> 3;  // output row of this thread (32 rows x 8 threads)
    const float ri = rinv[oi];
    float prel[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int j = i0 + oi + r - w;
      prel[r] = (r < nrel && j >= 0 && j < T) ? S[oi * ST + j] : 0.f;
    }
    lds_barrier();
    float* red = S;  // [4 waves][32][DK + 1], then E_v [nrel][DK]
    constexpr int RS = DK + 1;
    float* evs = red + 4 * 32 * RS;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const int d = NDT * l32 + dt;
      if (d < DK) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * RS + d] = acc[dt][r];
      }
    }
#pragma unroll
    for (int q = 0; q < NEV; ++q) {
      const int e = tid + q * kMhaMThreads;
      if (e < nrel * DK) evs[e] = evr[q];
    }
    lds_barrier();
    if (i0 + oi < T) {
      const float* rp = red + oi * RS + (tid & 7);
      const float* ep = evs + (tid & 7);
      float* op = g.out + (rowb + i0 + oi) * C + hd * DK + (tid & 7);
#pragma unroll
      for (int d0 = 0; d0 < DK; d0 += 8) {
        float v = rp[d0];
#pragma unroll
        for (int q = 1; q < 4; ++q) v = add_rn(v, rp[q * 32 * RS + d0]);
        if (NR > 0 && w >= 0) {
          float a = 0.f;
#pragma unroll
          for (int r = 0; r < NR; ++r)
            if (r < nrel) a = fmaf(prel[r], ep[r * DK + d0], a);
          v = add_rn(v, a);
        }
        op[d0] = v * ri;
        if (g.out_p) {
          const size_t o = (size_t)(op - g.out) + d0;
          split_f16(v * ri, g.out_p[o], g.out_p[g.n_out + o]);
        }
      }
    }
  };
  if (nrel <= 9) finish(std::integral_constant<int, 9>{});
  else finish(std::integral_constant<int, 31>{});
}

size_t mha_mfma_lds_bytes(int T, int dk, int window) {
  const size_t rel = (window >= 0 ? 32 * 33 : 0) + 32;  // relative-key logits, 1 / row sums
  const size_t s = (size_t)kMhaMRows * (T | 1) + rel;
  const size_t red = (size_t)4 * 32 * (dk + 1) + (size_t)(window >= 0 ? 2 * window + 1 : 0) * dk;  // partial tiles, E_v
  return (s > red ? s : red) * sizeof(float);
}

// ===========================================================================
// The same attention, WITHOUT the relative-position window (the flow's pre_transformer, models.py:508), as ONE pass over the
// keys with an online softmax, on the f16 matrix instruction with split-fp16 operands (hi + lo fp16 planes, fp32 accumulate:
// the arithmetic of every GEMM of this file).  One workgroup = 4 waves x 32 query frames of one (utterance, head); the key /
// value frames go through LDS in tiles of 32, ONCE per workgroup - the kernel above, 32 queries per workgroup and its
// operands straight from global memory, re-read K and V 19 times per (utterance, head) at 600 frames and ran on exact-fp32
// MFMAs (1/5.3 of this rate): 205 us per launch on the flow's shape against a matrix-pipe floor of ~15 us here.
//
// Orientation (cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand"): the score tile is computed
// TRANSPOSED, X = K Q^T (rows = keys, columns = queries), so a lane holds ONE query's scores of 16 keys in its 16 accumulator
// registers (the other 16 keys sit in lane + 32): the softmax statistics are lane-local plus one half-wave exchange, and the
// probabilities, converted to fp16 planes in registers, ARE the B operand of the second product Y = V^T P^T (rows = head
// channels, columns = queries) - no LDS round trip, no transpose of P.  Y's columns are queries again, so the online
// rescaling multiplies a lane's own registers.  The k index of that second product is permuted (element j of lane half h
// of k-step s is key 16 s + 8 (j >> 2) + 4 h + (j & 3)); V stays row-major in LDS and the A fragment of V^T comes out of the
// transposing LDS read (ds_read_b64_tr_b16), two per plane and k-step.
//   scores: s = (q / sqrt(dk)) . k, -1e4 where mask_q * mask_k == 0 (attentions.py:270-271), keys past T excluded
//   p = exp(s - running max);  out = sum_k p v / sum_k p      (softmax and P V of attentions.py:283-286, one pass)
// LDS per buffer: K [32 keys][dk] hi / lo with 16-byte row padding (row stride 2 dk + 16 bytes: the 16 rows of a ds_read_b128
// group land on 16 distinct 16-byte slots), V [32 keys][dk] hi / lo with 192-byte rows, the 32 key masks.  (A first version
// staged V transposed with 2-byte LDS stores: 12-way bank conflicts, ~6 000 LDS cycles per tile and CU.)  Two buffers: the next tile's global loads are in flight during a tile's MFMAs and written to LDS
// behind them (one barrier per tile).
constexpr int kFaWaves = 4, kFaThreads = 64 * kFaWaves, kFaQ = 32 * kFaWaves;
template <int DK>
struct FaLds {
  static constexpr int NDT = (DK + 31) / 32, KROW = 2 * DK + 16, VROW = 192;
  static constexpr int KPL = 32 * KROW, VPL = 32 * VROW;
  static constexpr int MK = 2 * KPL + 2 * VPL;  // byte offset of the key masks (32 floats + the "plain tile" word)
  static constexpr int BUF = MK + 144;
};
template <int DK>  // head width: a multiple of 16, at most 64
__global__ __launch_bounds__(kFaThreads, 2) void mha_flash_kernel(MhaArgs g) {
  using L = FaLds<DK>;
  constexpr int NKS = DK / 16, NDT = L::NDT, C4 = DK / 4, NI = NKS;  // NI: float4 items per thread per tile (2 * 32 * C4 / 256)
  static_assert(DK % 16 == 0 && DK <= 64 && 2 * 32 * C4 == NI * kFaThreads, "head width");
  extern __shared__ __attribute__((aligned(16))) char fsm[];
  const int T = g.T, C = g.C;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l32 = lane & 31, half = lane >> 5;
  // workgroup -> (utterance, head, query block).  Workgroups go round-robin over the 8 XCDs (b and b + 8 share one, speed
  // only): the query blocks of one (utterance, head) pair take ids 8 apart, so one XCD's L2 serves the pair's K and V to
  // all of them (the plain order spread them over five XCDs, each fetching the pair's 230 KB from the Infinity Cache).
  const int nqb = (T + kFaQ - 1) / kFaQ, heads = C / DK, npair = gridDim.x / nqb;  // grid = npair * nqb
  int pair, qb;
  if (npair % 8 == 0) {
    pair = (blockIdx.x & 7) + 8 * ((blockIdx.x >> 3) / nqb);
    qb = (blockIdx.x >> 3) % nqb;
  } else {
    pair = blockIdx.x / nqb;
    qb = blockIdx.x % nqb;
  }
  const int b = pair / heads, hd = pair % heads, q0 = qb * kFaQ + wave * 32;
  const size_t rowb = (size_t)b * T;
  typedef __attribute__((address_space(1))) const f32x4 gf32x4;
  typedef __attribute__((address_space(1))) const float gf32;
  gf32* base = (gf32*)(g.qkv + rowb * 3 * C + hd * DK);
  gf32* maskg = (gf32*)(g.mask + rowb);
  const int nt = (T + 31) / 32;
  const bool wave_active = q0 < T;  // (the last workgroup of a sequence: its idle waves only help with the staging)

  // ---- staging: item i = tid + 256 n: i < 32 C4: K value group (key i / C4, channels 4 (i % C4) ..); else the same of V ----
  // Two register sets: the loads of tiles t + 1 and t + 2 are in flight while tile t is computed (one tile ahead left the
  // workgroup's two tiles of LDS traffic waiting on a single 12-KB round trip to L2 / the Infinity Cache per tile: 110 us per
  // launch, latency-bound at ~3 TB/s chip-wide)
  struct Stage { f32x4 v[NI]; float mk; };
  Stage stA, stB;
  auto stage_load = [&](int t, Stage& sg) {
    f32x4 (&stg)[NI] = sg.v;
    float& stg_mk = sg.mk;
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      int i = tid + kFaThreads * n;
      const bool isv = i >= 32 * C4;
      if (isv) i -= 32 * C4;
      const int key = t * 32 + i / C4, c4 = i % C4;
      const int kc = key < T ? key : T - 1;
      const f32x4 v = *(gf32x4*)(base + (size_t)kc * 3 * C + (isv ? 2 * C : C) + 4 * c4);
      stg[n] = key < T ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (tid < 32) {
      const int key = t * 32 + tid;
      stg_mk = key < T ? maskg[key] : -1.0f;  // (-1: no such key)
    }
  };
  auto stage_store = [&](int p, const Stage& sg) {
    const f32x4 (&stg)[NI] = sg.v;
    const float mkv = sg.mk;
    char* buf = fsm + p * L::BUF;
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      int i = tid + kFaThreads * n;
      const bool isv = i >= 32 * C4;
      if (isv) i -= 32 * C4;
      const int kk = i / C4, c4 = i % C4;
      // hi = the fp16 below |x| in magnitude (v_cvt_pkrtz: two values per instruction), lo = fp16((x - hi) * 2^11): 21
      // significand bits for 5 instructions per value where the saturating round-to-nearest split_f16 takes 12
      typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
      union { f16x2 h[2]; uint2 u; } hi, lo;
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        const float a = __builtin_amdgcn_fmed3f(stg[n][e], -kSplitMax, kSplitMax), c = __builtin_amdgcn_fmed3f(stg[n][e + 1], -kSplitMax, kSplitMax);
        hi.h[e >> 1] = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, c));
        lo.h[e >> 1] = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz((a - (float)hi.h[e >> 1][0]) * kSplitScale, (c - (float)hi.h[e >> 1][1]) * kSplitScale));
      }
      if (!isv) {
        *reinterpret_cast<uint2*>(buf + kk * L::KROW + c4 * 8) = hi.u;
        *reinterpret_cast<uint2*>(buf + L::KPL + kk * L::KROW + c4 * 8) = lo.u;
      } else {
        *reinterpret_cast<uint2*>(buf + 2 * L::KPL + kk * L::VROW + c4 * 8) = hi.u;
        *reinterpret_cast<uint2*>(buf + 2 * L::KPL + L::VPL + kk * L::VROW + c4 * 8) = lo.u;
      }
    }
    if (tid < 32) *reinterpret_cast<float*>(buf + L::MK + tid * 4) = mkv;
    // one more word behind the 32 masks: 1 when every key of the tile exists and is unmasked (the common tile: no selects)
    if (tid < 64) {
      const bool plain = __all(tid >= 32 || mkv == 1.0f);
      if (tid == 0) *reinterpret_cast<float*>(buf + L::MK + 128) = plain ? 1.0f : 0.0f;
    }
  };

  stage_load(0, stA);
  if (nt > 1) stage_load(1, stB);
  // zero LDS once: the V columns past DK (never staged; read by the second channel tile when DK < 64) must hold finite values
  for (int i = tid; i < 2 * L::BUF / 4; i += kFaThreads) reinterpret_cast<float*>(fsm)[i] = 0.f;
  // ---- this lane's query: column l32 of the wave's 32 queries; B operand of X = K Q^T: Q[query][16 s + 8 half + j] ----
  const int qi = q0 + l32;
  const bool qok = qi < T;
  f16x8 qh[NKS], ql[NKS];
  float mq = 0.f;
  {
    const int qc = qok ? qi : T - 1;
    gf32* src = base + (size_t)qc * 3 * C + 8 * half;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      const f32x4 a = *(gf32x4*)(src + 16 * s), c = *(gf32x4*)(src + 16 * s + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // attentions.py:261: query / sqrt(k_channels); times log2(e): the scores come out in the base-2 domain of v_exp_f32
        f16 h0, l0, h1, l1;
        split_f16(div_rn(a[e], g.qscale) * 1.4426950408889634f, h0, l0);
        split_f16(div_rn(c[e], g.qscale) * 1.4426950408889634f, h1, l1);
        qh[s][e] = h0; ql[s][e] = l0; qh[s][4 + e] = h1; ql[s][4 + e] = l1;
      }
    }
    mq = qok ? maskg[qi] : 0.f;
  }
  __syncthreads();
  stage_store(0, stA);
  __syncthreads();

  f32x16 y[NDT], y2[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { y[dt][r] = 0.f; y2[dt][r] = 0.f; }
  float m_run = -3.0e38f, lsum = 0.f;
  auto pair_with_other_half = [&](float v, auto op) {  // op(v of this lane, v of lane +- 32)
    return op(v, __shfl_xor(v, 32));
  };

  // one tile: request tile t + 2 into `ld` (free: its tile went to LDS an iteration ago), compute tile t, store tile t + 1
  // (requested an iteration ago into `sv`) into the other LDS buffer
  auto tile_iter = [&](int t, Stage& ld, const Stage& sv) {
    const int p = t & 1;
    if (t + 2 < nt) stage_load(t + 2, ld);
    if (wave_active) {
      const char* buf = fsm + p * L::BUF;
      // ---- X = K Q^T: rows = the tile's 32 keys (A operand from LDS), columns = the wave's 32 queries ----
      f32x16 x, x2;
#pragma unroll
      for (int r = 0; r < 16; ++r) { x[r] = 0.f; x2[r] = 0.f; }
      const char* ka = buf + l32 * L::KROW + half * 16;
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        const f16x8 kh = *reinterpret_cast<const f16x8*>(ka + 32 * s), kl = *reinterpret_cast<const f16x8*>(ka + L::KPL + 32 * s);
        x = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s], x, 0, 0, 0);
        x2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s], x2, 0, 0, 0);
        x2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s], x2, 0, 0, 0);
      }
      // register r of this lane: key (r & 3) + 8 (r >> 2) + 4 half of the tile
      float sc[16];
      float tmax = -3.0e38f;
      const bool plain = *reinterpret_cast<const float*>(buf + L::MK + 128) != 0.f && __all(mq != 0.f);  // (uniform)
      if (plain) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          sc[r] = fmaf(x2[r], 1.0f / kSplitScale, x[r]);
          tmax = fmaxf(tmax, sc[r]);
        }
      } else {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const f32x4 mk4 = *reinterpret_cast<const f32x4*>(buf + L::MK + (8 * gq + 4 * half) * 4);  // the masks of 4 of its keys
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * gq + e;
            float v = fmaf(x2[r], 1.0f / kSplitScale, x[r]);
            v = mk4[e] < 0.f ? -3.0e38f : (mq * mk4[e] == 0.f ? -1e4f * 1.4426950408889634f : v);  // attentions.py:270-271 masked_fill(mask == 0, -1e4)
            sc[r] = v;
            tmax = fmaxf(tmax, v);
          }
        }
      }
      tmax = pair_with_other_half(tmax, [](float a, float c) { return fmaxf(a, c); });
      // Online softmax with a LAZY running maximum (base-2 domain): m_run moves only when a tile's maximum exceeds it by more
      // than 8, so p = 2^(s - m_run) <= 256 (exact in the split planes) and the 64 accumulator registers are rescaled a few
      // times per query instead of at almost every tile (with 32 queries per wave SOME maximum moves in 85 % of the tiles).
      // Every quantity that is on the old scale - the sums and both accumulator sets - is rescaled together, before this
      // tile's p are formed.
      const bool move = tmax > m_run + 8.0f;  // (first tile: m_run = -3e38)
      if (__any(move)) {
        const float m_new = move ? tmax : m_run;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // (1 where nothing moved; 0 on the first tile, where the sums are 0)
        m_run = m_new;
        lsum *= alpha;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) { y[dt][r] *= alpha; y2[dt][r] *= alpha; }
      }
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sc[r] = __builtin_amdgcn_exp2f(sc[r] - m_run);
        ps += sc[r];
      }
      lsum += ps;
      // ---- Y += V^T P^T: P's registers 8 s .. 8 s + 7 are k-step s of the B operand; A = V^T rows (channels) from LDS ----
      f16x8 ph[2], pl[2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          // p in [0, 256]: hi = the fp16 below it (v_cvt_pkrtz, two values per instruction), lo = fp16((p - hi) * 2^11): 21
          // significand bits, no range or subnormal cases to handle
          typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
          const float a = sc[8 * s + j], c = sc[8 * s + j + 1];
          const f16x2 h2 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, c));
          const f16x2 l2 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz((a - (float)h2[0]) * kSplitScale, (c - (float)h2[1]) * kSplitScale));
          ph[s][j] = h2[0]; ph[s][j + 1] = h2[1];
          pl[s][j] = l2[0]; pl[s][j + 1] = l2[1];
        }
      // A fragment of V^T (row = channel 32 dt + l32): element j = V[key 16 s + 8 (j >> 2) + 4 half + (j & 3)][channel] - four
      // consecutive KEYS of one channel, twice.  V sits row-major in LDS ([key][channel], 192-byte rows); the transposing read
      // ds_read_b64_tr_b16 hands every lane of a 16-lane group one COLUMN (channel) of a 4-row x 16-column block: lane 4 q + p
      // of the group supplies the address of row q, columns 4 p .. 4 p + 3, lane i receives column i, row q in element q
      // (cdna_hip_programming.md T10).  Rows 192 bytes apart put the block's four rows on four disjoint quarters of the banks.
      const unsigned va0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(buf + 2 * L::KPL) +
                           (4 * half + ((lane & 15) >> 2)) * L::VROW + (16 * ((lane & 31) >> 4) + 4 * (lane & 3)) * 2;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          union { uint2 u[2]; f16x8 v; } vh, vl;
          const unsigned a = va0 + 16 * s * L::VROW + 64 * dt;
          asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:%c5\n\t"
                       "ds_read_b64_tr_b16 %2, %4 offset:%c6\n\tds_read_b64_tr_b16 %3, %4 offset:%c7\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(vh.u[0]), "=&v"(vh.u[1]), "=&v"(vl.u[0]), "=&v"(vl.u[1])
                       : "v"(a), "i"(8 * L::VROW), "i"(L::VPL), "i"(L::VPL + 8 * L::VROW)
                       : "memory");
          y[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh.v, ph[s], y[dt], 0, 0, 0);
          y2[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh.v, pl[s], y2[dt], 0, 0, 0);
          y2[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl.v, ph[s], y2[dt], 0, 0, 0);
        }
      }
    }
    if (t + 1 < nt) stage_store(p ^ 1, sv);  // (buffer p ^ 1 was last read in iteration t - 1: every wave is past that barrier)
    __syncthreads();
  };
  for (int t = 0; t < nt; t += 2) {
    tile_iter(t, stA, stB);
    if (t + 1 < nt) tile_iter(t + 1, stB, stA);
  }
  if (!wave_active) return;
  const float ltot = pair_with_other_half(lsum, [](float a, float c) { return a + c; });
  const float inv = 1.0f / ltot;
  if (qok) {
    float* orow = g.out + (rowb + qi) * C + hd * DK;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int d0 = 32 * dt + 8 * gq + 4 * half;  // this lane's registers 4 gq .. 4 gq + 3 of tile dt: channels d0 .. d0 + 3
        if (d0 < DK) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = fmaf(y2[dt][4 * gq + e], 1.0f / kSplitScale, y[dt][4 * gq + e]) * inv;
          if (!g.planes_only) *reinterpret_cast<f32x4*>(orow + d0) = o;  // (the output GEMM of split-fp16 mode reads the planes)
          if (g.out_p) {
            union { f16 h[4]; uint2 u; } hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) split_f16(o[e], hi.h[e], lo.h[e]);
            const size_t oo = (size_t)(orow - g.out) + d0;
            *reinterpret_cast<uint2*>(g.out_p + oo) = hi.u;
            *reinterpret_cast<uint2*>(g.out_p + g.n_out + oo) = lo.u;
          }
        }
      }
  }
}

// attentions.Encoder.forward:80-84: at layer cond_layer_idx, x = (x + spk_emb_linear(g)) * x_mask with g [B, gin, 1] broadcast over
// the frames; gvec [B, C] is the projected embedding.  In place on the layer's input (and its split planes).
__global__ void add_spk_kernel(float* x, f16* x_p, const float* gvec, const float* mask, int M, int C, int T) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * C) return;
  const size_t m = i / C, c = i % C;
  const float v = mul_rn(add_rn(x[i], gvec[(m / T) * C + c]), mask[m]);
  x[i] = v;
  if (x_p) split_f16(v, x_p[i], x_p[(size_t)M * C + i]);
}

// commons.fused_add_tanh_sigmoid_multiply with g = None (commons.py:102-109): [M, 2H] -> [M, H]
__global__ void wn_gate_kernel(const float* xin, float* acts, f16* acts_p, int M, int H, const float* gl, int T, int ldg) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * H) return;
  const size_t m = i / H, c = i % H;
  float a = xin[m * 2 * H + c], s = xin[m * 2 * H + H + c];
  if (gl != nullptr) { a = add_rn(a, gl[(m / T) * ldg + c]); s = add_rn(s, gl[(m / T) * ldg + H + c]); }
  const float v = mul_rn(tanhf(a), sigmoid_f(s));
  if (acts != nullptr) acts[i] = v;  // (nullptr: the consumer - the res/skip GEMM in split-fp16 mode - reads the planes only)
  if (acts_p) split_f16(v, acts_p[i], acts_p[(size_t)M * H + i]);
}
// modules.WN.forward:201-208: not last: x = (x + rs[:, :H]) * mask; output += rs[:, H:]
//                             last:     output = (output + rs) * mask   (the final `output * x_mask` folded in)
__global__ void wn_update_kernel(float* x, float* output, const float* rs, const float* mask, f16* x_p, f16* out_p, int M, int H, int last,
                                 int first) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * H) return;
  const size_t m = i / H, c = i % H;
  const float o = first ? 0.f : output[i];
  const size_t n = (size_t)M * H;
  if (last) {
    const float v = mul_rn(add_rn(o, rs[m * H + c]), mask[m]);
    output[i] = v;
    if (out_p) split_f16(v, out_p[i], out_p[n + i]);
  } else {
    const float v = mul_rn(add_rn(x[i], rs[m * 2 * H + c]), mask[m]);
    x[i] = v;
    if (x_p) split_f16(v, x_p[i], x_p[n + i]);
    output[i] = add_rn(o, rs[m * 2 * H + H + c]);
  }
}
// The same two kernels, four channels per thread (H % 4 == 0, fewer than 2^32 elements): 16-byte loads and stores, 8-byte
// plane stores, one 32-bit division per four elements instead of a 64-bit one per element.  As one-element kernels they
// were bound by their instruction streams (the division; tanhf + expf + an exact division per gate: ~100 instructions per
// element), not by the 59 / 103 MB they stream.  kFast (the split-fp16 mode, whose GEMMs round these values to 22 bits
// anyway): the gate on the hardware exp / rcp like the decoder's LSTM cell (common.h tanh_fast / sigmoid_fast); the
// exact-fp32 mode keeps the library functions.
typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split4_store(const f32x4& v, f16* planes, size_t n, size_t i4) {
  f16x4v hi, lo;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f16 h, l;
    split_f16(v[e], h, l);
    hi[e] = h;
    lo[e] = l;
  }
  *reinterpret_cast<f16x4v*>(planes + 4 * i4) = hi;
  *reinterpret_cast<f16x4v*>(planes + n + 4 * i4) = lo;
}
// (gl != nullptr: g_l of modules.py:193-199, the speaker conditioning of this WN layer - constant over an utterance's frames, so a
// [B, ldg] matrix whose row is the frame's utterance; the reference adds it to x_in before the two activations, commons.py:102-109)
template <bool kFast>
__global__ void wn_gate4_kernel(const float* xin, float* acts, f16* acts_p, uint32_t M, uint32_t H4, const float* gl, uint32_t T, uint32_t ldg) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * H4) return;
  const uint32_t m = i / H4, c4 = i - m * H4;
  const f32x4* row = reinterpret_cast<const f32x4*>(xin) + (size_t)m * 2 * H4;
  f32x4 a = row[c4], s = row[H4 + c4];
  if (gl != nullptr) {
    const f32x4* grow = reinterpret_cast<const f32x4*>(gl + (size_t)(m / T) * ldg);
    const f32x4 ga = grow[c4], gs = grow[H4 + c4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] = add_rn(a[e], ga[e]); s[e] = add_rn(s[e], gs[e]); }
  }
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = kFast ? mul_rn(tanh_fast(a[e]), sigmoid_fast(s[e])) : mul_rn(tanhf(a[e]), sigmoid_f(s[e]));
  if (acts != nullptr) reinterpret_cast<f32x4*>(acts)[i] = v;  // (nullptr: see wn_gate_kernel)
  if (acts_p) split4_store(v, acts_p, (size_t)M * H4 * 4, i);
}
__global__ void wn_update4_kernel(float* x, float* output, const float* rs, const float* mask, f16* x_p, f16* out_p, uint32_t M, uint32_t H4,
                                  int last, int first) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * H4) return;
  const uint32_t m = i / H4, c4 = i - m * H4;
  const size_t n = (size_t)M * H4 * 4;
  const float mk = mask[m];
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  if (!first) o = reinterpret_cast<const f32x4*>(output)[i];
  f32x4 v;
  if (last) {
    const f32x4 r = reinterpret_cast<const f32x4*>(rs)[(size_t)m * H4 + c4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = mul_rn(add_rn(o[e], r[e]), mk);
    reinterpret_cast<f32x4*>(output)[i] = v;
    if (out_p) split4_store(v, out_p, n, i);
  } else {
    const f32x4* row = reinterpret_cast<const f32x4*>(rs) + (size_t)m * 2 * H4;
    const f32x4 r0 = row[c4], r1 = row[H4 + c4], xv = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 on;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = mul_rn(add_rn(xv[e], r0[e]), mk);
      on[e] = add_rn(o[e], r1[e]);
    }
    reinterpret_cast<f32x4*>(x)[i] = v;
    if (x_p) split4_store(v, x_p, n, i);
    reinterpret_cast<f32x4*>(output)[i] = on;
  }
}
// modules.Flip (modules.py:374-381) + split: xf = flip(x); x0m = xf[:, :half] * mask
__global__ void flip_split_kernel(const float* x, const float* mask, float* xf, float* x0m, f16* x0m_p, int M, int I) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * I) return;
  const size_t m = i / I, c = i % I;
  const float v = x[m * I + (I - 1 - c)];
  xf[i] = v;
  const int half = I / 2;
  if ((int)c < half) {
    const float vm = mul_rn(v, mask[m]);
    const size_t o = m * half + c;
    x0m[o] = vm;
    if (x0m_p) split_f16(vm, x0m_p[o], x0m_p[(size_t)M * half + o]);
  }
}
// x0_ = pre_transformer(...) + x0 (models.py:509): enc [M, half] += xf[:, :half]
__global__ void add_x0_kernel(float* enc, f16* enc_p, const float* xf, int M, int I) {
  const int half = I / 2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * half) return;
  const size_t m = i / half, c = i % half;
  const float v = add_rn(enc[i], xf[m * I + c]);
  enc[i] = v;
  if (enc_p) split_f16(v, enc_p[i], enc_p[(size_t)M * half + i]);
}
// x1 = (x1 - m) * exp(-0) * mask (models.py:529): xf[:, half:] updated in place; mm is post(h) * mask
__global__ void couple_kernel(float* xf, const float* mm, const float* mask, int M, int I) {
  const int half = I / 2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * half) return;
  const size_t m = i / half, c = i % half;
  float* p = xf + m * I + half + c;
  *p = mul_rn(mul_rn(sub_rn(*p, mm[i]), 1.0f), mask[m]);
}
// flip+split, x0 residual and the coupling update, four channels per thread (I % 8 == 0, fewer than 2^32 elements)
__global__ void flip_split4_kernel(const float* x, const float* mask, float* xf, float* x0m, f16* x0m_p, uint32_t M, uint32_t I4) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * I4) return;
  const uint32_t m = i / I4, c4 = i - m * I4, half4 = I4 / 2;
  const f32x4 s = reinterpret_cast<const f32x4*>(x)[(size_t)m * I4 + (I4 - 1 - c4)];
  const f32x4 v = {s[3], s[2], s[1], s[0]};
  reinterpret_cast<f32x4*>(xf)[i] = v;
  if (c4 < half4) {
    const float mk = mask[m];
    f32x4 vm;
#pragma unroll
    for (int e = 0; e < 4; ++e) vm[e] = mul_rn(v[e], mk);
    const size_t o = (size_t)m * half4 + c4;
    reinterpret_cast<f32x4*>(x0m)[o] = vm;
    if (x0m_p) split4_store(vm, x0m_p, (size_t)M * half4 * 4, o);
  }
}
__global__ void add_x04_kernel(float* enc, f16* enc_p, const float* xf, uint32_t M, uint32_t I4) {
  const uint32_t half4 = I4 / 2;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * half4) return;
  const uint32_t m = i / half4, c4 = i - m * half4;
  const f32x4 a = reinterpret_cast<const f32x4*>(enc)[i], b = reinterpret_cast<const f32x4*>(xf)[(size_t)m * I4 + c4];
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = add_rn(a[e], b[e]);
  reinterpret_cast<f32x4*>(enc)[i] = v;
  if (enc_p) split4_store(v, enc_p, (size_t)M * half4 * 4, i);
}
__global__ void couple4_kernel(float* xf, const float* mm, const float* mask, uint32_t M, uint32_t I4) {
  const uint32_t half4 = I4 / 2;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * half4) return;
  const uint32_t m = i / half4, c4 = i - m * half4;
  f32x4* p = reinterpret_cast<f32x4*>(xf) + (size_t)m * I4 + half4 + c4;
  const f32x4 a = *p, b = reinterpret_cast<const f32x4*>(mm)[i];
  const float mk = mask[m];
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = mul_rn(mul_rn(sub_rn(a[e], b[e]), 1.0f), mk);
  *p = v;
}
__global__ void copy_f_kernel(const float* a, float* b, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] = a[i];
}
// pack: rows of three [C, C] matrices stacked -> [3C, C]; conv weights via launch_conv_transpose

inline dim3 grid1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

// ---------------------------------------------------------------------------
// host-side building blocks
// ---------------------------------------------------------------------------
// One GEMM of the path.  W points at the packed weight (fp32 | hi | lo, see make_layout) of n_w elements.
// Split-fp16 arithmetic (3 f16 MFMA products, fp32 accumulate - the same accuracy class as fp32, see
// gemm_tile.h): the A operand's planes are produced by one elementwise pass into `planes`.
struct GemmCtx {
  f16* planes;  // scratch for the A operand's hi / lo planes: 2 * max(M * K) halfs
  bool split;   // false: exact fp32 MFMAs
};
// a_pl: the A operand's planes if its producer already wrote them (else they are made here);
// out_pl: where to put the planes of the result for the next GEMM (needs ldo == N), or nullptr.
void gemm_generic(const GemmCtx& cx, const float* a, const f16* a_pl, int lda, int K, const float* W, size_t n_w, const float* bias, int M,
                  int N, float* out, f16* out_pl, int ldo, int act, const float* row_mask, const float* resid, int taps, int T,
                  hipStream_t st) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.M = M; g.N = N; g.bias = bias; g.out = out; g.ldo = ldo; g.act = act; g.row_mask = row_mask; g.resid = resid;
  const void *a0 = a, *a1 = a;
  g.W = g.W_lo = W;
  const bool split = cx.split && lda == K && !(K & 7);
  if (out_pl != nullptr && ldo == N) { g.out_kind = 1; g.out_h = out_pl; g.out_l = out_pl + (size_t)M * N; }
  if (split) {
    const f16* ph = a_pl;
    if (ph == nullptr) {
      launch_split(a, cx.planes, cx.planes + (size_t)M * K, (size_t)M * K, st);
      ph = cx.planes;
    }
    a0 = ph; a1 = ph + (size_t)M * K;
    g.prec = PREC_F16S;
    g.W = reinterpret_cast<const f16*>(W + n_w);
    g.W_lo = reinterpret_cast<const f16*>(W + n_w) + n_w;
  }
  if (taps > 1) {
    g.a = make_seg1(a0, K, K); g.a_lo = make_seg1(a1, K, K);
    g.T = T; g.Cin = K; g.taps = taps; g.ldw = taps * K; g.K = taps * K;
    launch_gemm(g, A_CONV, EPI_GENERIC, st);
  } else {
    g.a = make_seg1(a0, lda, K); g.a_lo = make_seg1(a1, lda, K);
    g.ldw = K; g.K = K;
    launch_gemm(g, A_PLAIN, EPI_GENERIC, st);
  }
}

struct StackWs {
  float *x, *xm, *qkv, *att, *t, *f;  // [M,C] [M,C] [M,3C] [M,C] [M,C] [M,F]
  f16 *x_p, *xm_p, *att_p, *f_p;      // split-fp16 planes of x, xm, att, f (written by their producers)
  GemmCtx cx;
};
size_t mha_lds_bytes(int T, int dk, int window) {
  const int nrel = window >= 0 ? 2 * window + 1 : 0;
  return ((size_t)kMhaRows * dk + (size_t)kMhaChunk * (dk + 1) + 2 * (size_t)nrel * dk + (size_t)kMhaRows * T) * sizeof(float);
}
constexpr size_t kMhaMaxLds = 150 * 1024;

void launch_layernorm(const float* in, const float* gamma, const float* beta, const float* mask, float* out, float* out_m, f16* out_p,
                      f16* outm_p, int M, int C, hipStream_t st) {
  const dim3 grid((M + 3) / 4), block(256);
  if (C <= 128) hipLaunchKernelGGL(layernorm_rows_kernel<2>, grid, block, 0, st, in, gamma, beta, mask, out, out_m, out_p, outm_p, M, C, 1e-5f);
  else if (C <= 256) hipLaunchKernelGGL(layernorm_rows_kernel<4>, grid, block, 0, st, in, gamma, beta, mask, out, out_m, out_p, outm_p, M, C, 1e-5f);
  else hipLaunchKernelGGL(layernorm_rows_kernel<16>, grid, block, 0, st, in, gamma, beta, mask, out, out_m, out_p, outm_p, M, C, 1e-5f);
}

// attentions.Encoder.forward (attentions.py:76-93), eval mode.  Input: sw.x = sw.xm = x * mask.
// Result: sw.xm (= x * mask of the last layer).
// gvec / cond_idx: the text encoder's speaker conditioning (add_spk_kernel), nullptr / -1 without.
int run_stack(ttsvits_handle* h, const StackBlob& sb, const StackDims& sd, const StackWs& sw, const float* mask, int B, int T,
              hipStream_t st, const float* gvec = nullptr, int cond_idx = -1) {
  const float* blob = h->blob;
  const int M = B * T, C = sd.C, dk = C / sd.heads;
  // matrix-core attention when the score tile fits LDS (T <= ~1150) and dk is one of the built sizes;
  // the scalar kernel covers everything else
  const size_t lds_m = mha_mfma_lds_bytes(T, dk, sd.window);
  const bool use_mfma = lds_m <= kMhaMaxLds && sd.window <= 15 && (dk == 96 || dk == 48 || dk == 16 || dk == 8);
  const size_t lds = use_mfma ? lds_m : mha_lds_bytes(T, dk, sd.window);
  if (lds > kMhaMaxLds || dk > 256) return TTSDEC_ERR_DIMS;
  // no relative window, split-fp16 arithmetic, a head width the flash kernel is built for: one pass over the keys on the f16
  // matrix instruction (the exact-fp32 mode keeps the exact-fp32 kernels)
  const bool use_flash = sd.window < 0 && h->precision == TTSDEC_PREC_SPLIT_F16 && (dk == 48 || dk == 32 || dk == 64 || dk == 16) && !(C & 3);
  const void* kfn = !use_mfma ? (const void*)mha_kernel
                   : dk == 96 ? (const void*)mha_mfma_kernel<48>
                   : dk == 48 ? (const void*)mha_mfma_kernel<24>
                   : dk == 16 ? (const void*)mha_mfma_kernel<8> : (const void*)mha_mfma_kernel<4>;
  if (hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return vits_fail(h, "hipFuncSetAttribute(mha kernel)");
  const float* xin = sw.xm;  // layer 0 attends over x * mask; later layers over the unmasked LayerNorm output
  for (int i = 0; i < sd.layers; ++i) {
    float* xa = i == 0 ? sw.xm : sw.x;
    f16* xa_p = i == 0 ? sw.xm_p : sw.x_p;
    if (gvec != nullptr && i == cond_idx)
      hipLaunchKernelGGL(add_spk_kernel, grid1((size_t)M * C), dim3(256), 0, st, xa, sw.cx.split ? xa_p : nullptr, gvec, mask, M, C, T);
    gemm_generic(sw.cx, xa, xa_p, C, C, blob + sb.wqkv[i], (size_t)3 * C * C, blob + sb.bqkv[i], M, 3 * C, sw.qkv, nullptr, 3 * C, 0, nullptr,
                 nullptr, 1, T, st);
    MhaArgs a;
    a.qkv = sw.qkv; a.mask = mask; a.out = sw.att; a.out_p = sw.att_p; a.n_out = (size_t)M * C;
    a.planes_only = (use_flash && sw.cx.split && sw.att_p != nullptr && !(C & 7)) ? 1 : 0;
    a.T = T; a.C = C; a.dk = dk; a.window = sd.window;
    a.ek = sd.window >= 0 ? blob + sb.ek[i] : nullptr; a.ev = sd.window >= 0 ? blob + sb.ev[i] : nullptr;
    a.qscale = sqrtf((float)dk);
    if (use_flash) {
      const dim3 grid(((T + kFaQ - 1) / kFaQ) * sd.heads * B), block(kFaThreads);
      if (dk == 48) hipLaunchKernelGGL(mha_flash_kernel<48>, grid, block, 2 * FaLds<48>::BUF, st, a);
      else if (dk == 32) hipLaunchKernelGGL(mha_flash_kernel<32>, grid, block, 2 * FaLds<32>::BUF, st, a);
      else if (dk == 64) hipLaunchKernelGGL(mha_flash_kernel<64>, grid, block, 2 * FaLds<64>::BUF, st, a);
      else hipLaunchKernelGGL(mha_flash_kernel<16>, grid, block, 2 * FaLds<16>::BUF, st, a);
    } else if (use_mfma) {
      const dim3 grid((T + kMhaMRows - 1) / kMhaMRows, sd.heads, B), block(kMhaMThreads);
      if (dk == 96) hipLaunchKernelGGL(mha_mfma_kernel<48>, grid, block, lds, st, a);
      else if (dk == 48) hipLaunchKernelGGL(mha_mfma_kernel<24>, grid, block, lds, st, a);
      else if (dk == 16) hipLaunchKernelGGL(mha_mfma_kernel<8>, grid, block, lds, st, a);
      else hipLaunchKernelGGL(mha_mfma_kernel<4>, grid, block, lds, st, a);
    } else {
      hipLaunchKernelGGL(mha_kernel, dim3((T + kMhaRows - 1) / kMhaRows, sd.heads, B), dim3(kMhaThreads), lds, st, a);
    }
    // x = LayerNorm(x + conv_o(att))
    gemm_generic(sw.cx, sw.att, sw.att_p, C, C, blob + sb.wo[i], (size_t)C * C, blob + sb.bo[i], M, C, sw.t, nullptr, C, 0, nullptr, xa, 1, T, st);
    // (only the outputs somebody reads: x for the FFN's residual; x * mask as the FFN's input - its planes where conv_1 runs on
    // planes (gemm_generic: split mode and C % 8 == 0), the fp32 form otherwise)
    const bool pl = sw.cx.split && !(C & 7) && sw.xm_p != nullptr;
    launch_layernorm(sw.t, blob + sb.g1[i], blob + sb.b1[i], mask, sw.x, pl ? (float*)nullptr : sw.xm, (f16*)nullptr,
                     pl ? sw.xm_p : (f16*)nullptr, M, C, st);
    // FFN (attentions.py:411-419): conv_2(relu(conv_1(x * mask)) * mask) * mask, then x = LayerNorm(x + y)
    gemm_generic(sw.cx, sw.xm, sw.xm_p, C, C, blob + sb.w1[i], (size_t)sd.F * sd.kernel * C, blob + sb.c1[i], M, sd.F, sw.f, sw.f_p, sd.F, 1, mask,
                 nullptr, sd.kernel, T, st);
    gemm_generic(sw.cx, sw.f, sw.f_p, sd.F, sd.F, blob + sb.w2[i], (size_t)C * sd.kernel * sd.F, blob + sb.c2[i], M, C, sw.t, nullptr, C, 0, mask,
                 sw.x, sd.kernel, T, st);
    // (a layer in the middle hands the next one x and, where its QKV GEMM runs on planes, x's planes; the last layer hands the
    // caller x * mask in both forms)
    const bool last_layer = i == sd.layers - 1;
    launch_layernorm(sw.t, blob + sb.g2[i], blob + sb.b2[i], mask, last_layer ? (float*)nullptr : sw.x, last_layer ? sw.xm : (float*)nullptr,
                     (!last_layer && pl) ? sw.x_p : (f16*)nullptr, last_layer ? sw.xm_p : (f16*)nullptr, M, C, st);
  }
  (void)xin;
  return TTSDEC_OK;
}

size_t stack_ws_floats(const StackDims& sd, size_t M) {
  return M * (size_t)(4 * sd.C + 3 * sd.C + sd.F + (sd.C > sd.F ? sd.C : sd.F) + 3 * sd.C + sd.F) + 13 * kAlign;
}
StackWs carve_stack(float*& p, const StackDims& sd, size_t M, bool split) {
  auto take = [&](size_t n) { float* r = p; p += up(n, kAlign); return r; };
  StackWs w;
  w.x = take(M * sd.C); w.xm = take(M * sd.C); w.qkv = take(M * 3 * sd.C); w.att = take(M * sd.C); w.t = take(M * sd.C);
  w.f = take(M * sd.F);
  w.cx.planes = reinterpret_cast<f16*>(take(M * (sd.C > sd.F ? sd.C : sd.F)));  // hi + lo planes of one A operand (fallback)
  w.x_p = reinterpret_cast<f16*>(take(M * sd.C)); w.xm_p = reinterpret_cast<f16*>(take(M * sd.C));
  w.att_p = reinterpret_cast<f16*>(take(M * sd.C)); w.f_p = reinterpret_cast<f16*>(take(M * sd.F));
  w.cx.split = split;  // (ttsvits_set_precision: split-fp16 planes, or exact fp32 MFMAs for every GEMM)
  return w;
}

// fp32 weight at float offset `off` (n elements) -> its fp16 hi / lo planes right behind it
void pack_planes(float* b, size_t off, size_t n, hipStream_t st) {
  f16* hi = reinterpret_cast<f16*>(b + off + n);
  launch_split(b + off, hi, hi + n, n, st);
}

int pack_stack(const float* const* src, int& k, float* b, const StackBlob& s, const StackDims& sd, hipStream_t st) {
  const size_t C = sd.C, F = sd.F, dk = sd.C / sd.heads;
  for (int i = 0; i < sd.layers; ++i) {
    for (int j = 0; j < 3; ++j) {  // conv_q / conv_k / conv_v stacked on the output axis
      launch_copy(src[k + 2 * j], b + s.wqkv[i] + j * C * C, C * C, st);
      launch_copy(src[k + 2 * j + 1], b + s.bqkv[i] + j * C, C, st);
    }
    launch_copy(src[k + 6], b + s.wo[i], C * C, st);
    launch_copy(src[k + 7], b + s.bo[i], C, st);
    pack_planes(b, s.wqkv[i], 3 * C * C, st);
    pack_planes(b, s.wo[i], C * C, st);
    k += 8;
    if (sd.window >= 0) {
      launch_copy(src[k], b + s.ek[i], (2 * sd.window + 1) * dk, st);
      launch_copy(src[k + 1], b + s.ev[i], (2 * sd.window + 1) * dk, st);
      k += 2;
    }
    launch_copy(src[k], b + s.g1[i], C, st);
    launch_copy(src[k + 1], b + s.b1[i], C, st);
    if (src[k + 2]) launch_conv_transpose(src[k + 2], b + s.w1[i], (int)F, (int)C, sd.kernel, st);
    launch_copy(src[k + 3], b + s.c1[i], F, st);
    if (src[k + 4]) launch_conv_transpose(src[k + 4], b + s.w2[i], (int)C, (int)F, sd.kernel, st);
    launch_copy(src[k + 5], b + s.c2[i], C, st);
    launch_copy(src[k + 6], b + s.g2[i], C, st);
    launch_copy(src[k + 7], b + s.b2[i], C, st);
    pack_planes(b, s.w1[i], F * sd.kernel * C, st);
    pack_planes(b, s.w2[i], C * sd.kernel * F, st);
    k += 8;
  }
  return TTSDEC_OK;
}

bool dims_ok(const ttsvits_dims& d) {
  const int v[] = {d.inter_channels, d.hidden_channels, d.filter_channels, d.flow_hidden};
  for (int x : v)
    if (x <= 0 || (x & 3)) return false;
  if ((d.inter_channels / 2) & 3) return false;
  if (d.n_vocab <= 0 || d.n_heads <= 0 || d.hidden_channels % d.n_heads) return false;
  if (d.flow_tf_heads <= 0 || (d.inter_channels / 2) % d.flow_tf_heads) return false;
  if (d.n_layers < 0 || d.n_layers > kMaxLayers || d.flow_tf_layers < 0 || d.flow_tf_layers > kMaxLayers) return false;
  if (d.n_flows < 0 || d.n_flows > kMaxFlows || d.flow_wn_layers < 1 || d.flow_wn_layers > kMaxWn) return false;
  if (!(d.kernel_size & 1) || !(d.flow_kernel & 1) || !(d.flow_tf_kernel & 1) || d.kernel_size < 1 || d.flow_kernel < 1 || d.flow_tf_kernel < 1)
    return false;
  if (d.window_size > 64 || d.hidden_channels > 1024 || d.inter_channels > 2048) return false;
  if (d.gin_channels < 0 || (d.gin_channels & 3) || d.gin_channels > 4096) return false;
  // attentions.py:50-52 asserts cond_layer_idx < n_layers when the encoder is speaker-conditioned
  if (d.gin_channels > 0 && d.n_layers > 0 && (d.cond_layer_idx < 0 || d.cond_layer_idx >= d.n_layers)) return false;
  return true;
}

}  // namespace

extern "C" {

int ttsvits_create(const ttsvits_dims* dims, ttsvits_handle** out) {
  if (!dims || !out) return TTSDEC_ERR_INVALID_ARG;
  *out = nullptr;
  if (!dims_ok(*dims)) return TTSDEC_ERR_DIMS;
  ttsvits_handle* h = new (std::nothrow) ttsvits_handle();
  if (!h) return TTSDEC_ERR_INVALID_ARG;
  h->d = *dims;
  h->bl = make_layout(*dims);
  h->blob = nullptr;
  h->device = current_device_or_minus1();
  h->precision = TTSDEC_PREC_F32;  // (the reference's arithmetic; ttsvits_set_precision opts into split-fp16)
  *out = h;
  return TTSDEC_OK;
}
int ttsvits_set_precision(ttsvits_handle* h, int precision) {
  if (!h || (precision != TTSDEC_PREC_F32 && precision != TTSDEC_PREC_SPLIT_F16)) return TTSDEC_ERR_INVALID_ARG;
  h->precision = precision;
  return TTSDEC_OK;
}
int ttsvits_get_precision(const ttsvits_handle* h) { return h ? h->precision : TTSDEC_ERR_INVALID_ARG; }
int ttsvits_destroy(ttsvits_handle* h) {
  delete h;
  return TTSDEC_OK;
}
const char* ttsvits_last_hip_error(const ttsvits_handle* h) { return h ? h->hip_err.c_str() : ""; }
int ttsvits_num_weight_tensors(const ttsvits_handle* h) { return h ? n_text_tensors(h->d) + n_flow_tensors(h->d) : TTSDEC_ERR_INVALID_ARG; }
size_t ttsvits_packed_bytes(const ttsvits_handle* h) { return h ? h->bl.total * sizeof(float) : 0; }

int ttsvits_pack_weights(ttsvits_handle* h, const float* const* src, int n_src, void* blob, void* stream) {
  if (!h || !src || !blob || n_src != ttsvits_num_weight_tensors(h)) return TTSDEC_ERR_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(blob) & 255) return TTSDEC_ERR_WORKSPACE;
  if (!device_is_current(h->device)) return TTSDEC_ERR_DEVICE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const ttsvits_dims& d = h->d;
  const VitsBlob& L = h->bl;
  float* b = static_cast<float*>(blob);
  if (hipMemsetAsync(blob, 0, L.total * sizeof(float), st) != hipSuccess) return vits_fail(h, "memset");
  const size_t H = d.hidden_channels, I = d.inter_channels, half = I / 2, Fh = d.flow_hidden;
  int k = 0;
  // (launch_copy / launch_conv_transpose skip NULL sources: a module that owns only the text encoder
  // or only the flow packs what it has)
  launch_copy(src[k++], b + L.emb, (size_t)d.n_vocab * H, st);
  if (d.gin_channels > 0) {
    launch_copy(src[k++], b + L.spk_w, H * d.gin_channels, st);
    launch_copy(src[k++], b + L.spk_b, H, st);
    pack_planes(b, L.spk_w, H * d.gin_channels, st);
  }
  pack_stack(src, k, b, L.enc, enc_dims(d), st);
  launch_copy(src[k++], b + L.proj_w, 2 * I * H, st);
  launch_copy(src[k++], b + L.proj_b, 2 * I, st);
  pack_planes(b, L.proj_w, 2 * I * H, st);
  for (int f = 0; f < d.n_flows; ++f) {
    const FlowBlob& fb = L.flow[f];
    pack_stack(src, k, b, fb.tf, tf_dims(d), st);
    launch_copy(src[k++], b + fb.pre_w, Fh * half, st);
    launch_copy(src[k++], b + fb.pre_b, Fh, st);
    pack_planes(b, fb.pre_w, Fh * half, st);
    if (d.gin_channels > 0) {
      const size_t nc = 2 * Fh * d.flow_wn_layers;
      launch_copy(src[k++], b + fb.cond_w, nc * d.gin_channels, st);
      launch_copy(src[k++], b + fb.cond_b, nc, st);
      pack_planes(b, fb.cond_w, nc * d.gin_channels, st);
    }
    for (int j = 0; j < d.flow_wn_layers; ++j) {
      const size_t cr = j < d.flow_wn_layers - 1 ? 2 * Fh : Fh;
      if (src[k]) launch_conv_transpose(src[k], b + fb.in_w[j], (int)(2 * Fh), (int)Fh, d.flow_kernel, st);
      launch_copy(src[k + 1], b + fb.in_b[j], 2 * Fh, st);
      launch_copy(src[k + 2], b + fb.rs_w[j], cr * Fh, st);
      launch_copy(src[k + 3], b + fb.rs_b[j], cr, st);
      pack_planes(b, fb.in_w[j], 2 * Fh * d.flow_kernel * Fh, st);
      pack_planes(b, fb.rs_w[j], cr * Fh, st);
      k += 4;
    }
    launch_copy(src[k++], b + fb.post_w, half * Fh, st);
    launch_copy(src[k++], b + fb.post_b, half, st);
    pack_planes(b, fb.post_w, half * Fh, st);
  }
  const int rc = vits_fail(h, "pack_weights");
  if (rc == TTSDEC_OK) h->blob = b;
  return rc;
}

int ttsvits_bind_weights(ttsvits_handle* h, const void* blob) {
  if (!h || !blob || (reinterpret_cast<uintptr_t>(blob) & 255)) return TTSDEC_ERR_INVALID_ARG;
  h->blob = static_cast<const float*>(blob);
  return TTSDEC_OK;
}

size_t ttsvits_text_encoder_workspace_bytes(const ttsvits_handle* h, int B, int T) {
  if (!h || B <= 0 || T <= 0) return 0;
  const size_t M = (size_t)B * T;
  return (stack_ws_floats(enc_dims(h->d), M) + up(M, kAlign) + up(M * 2 * h->d.inter_channels, kAlign) +
          up((size_t)B * h->d.hidden_channels, kAlign)) * sizeof(float);
}

int ttsvits_text_encoder(ttsvits_handle* h, const int64_t* ids, const int32_t* lengths, const float* g, int B, int T, float* x, float* m,
                         float* logs, void* workspace, size_t workspace_bytes, void* stream, int32_t* status) {
  if (!h || !ids || !lengths || !x || !m || !logs || !workspace || B <= 0 || T <= 0) return TTSDEC_ERR_INVALID_ARG;
  if (g != nullptr && h->d.gin_channels <= 0) return TTSDEC_ERR_INVALID_ARG;
  if (!h->blob) return TTSDEC_ERR_NOT_BOUND;
  if (workspace_bytes < ttsvits_text_encoder_workspace_bytes(h, B, T) || (reinterpret_cast<uintptr_t>(workspace) & 255))
    return TTSDEC_ERR_WORKSPACE;
  if (!device_is_current(h->device)) return TTSDEC_ERR_DEVICE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const ttsvits_dims& d = h->d;
  const VitsBlob& L = h->bl;
  const int M = B * T, H = d.hidden_channels, I = d.inter_channels;
  float* p = static_cast<float*>(workspace);
  const StackDims sd = enc_dims(d);
  StackWs sw = carve_stack(p, sd, (size_t)M, h->precision == TTSDEC_PREC_SPLIT_F16);
  float* mask = p; p += up((size_t)M, kAlign);
  float* stats = p; p += up((size_t)M * 2 * I, kAlign);
  float* gvec = p;
  GemmCtx exact;  // (the [B, gin] projections of g are a few KFLOP: always the exact fp32 instruction)
  exact.planes = nullptr; exact.split = false;
  if (g != nullptr)  // attentions.py:81: spk_emb_linear(g)
    gemm_generic(exact, g, nullptr, d.gin_channels, d.gin_channels, h->blob + L.spk_w, (size_t)H * d.gin_channels, h->blob + L.spk_b, B, H, gvec,
                 nullptr, H, 0, nullptr, nullptr, 1, 1, st);
  // models.py:370-376
  hipLaunchKernelGGL(embed_scale_kernel, grid1((size_t)M * H), dim3(256), 0, st, reinterpret_cast<const long long*>(ids), lengths,
                     h->blob + L.emb, d.n_vocab, T, H, sqrtf((float)H), sw.xm, sw.xm_p, mask, M, status);
  int rc = run_stack(h, L.enc, sd, sw, mask, B, T, st, g != nullptr ? gvec : nullptr, d.cond_layer_idx);
  if (rc != TTSDEC_OK) return rc;
  // models.py:377-379: stats = proj(x) * x_mask; m, logs = split(stats)
  gemm_generic(sw.cx, sw.xm, sw.xm_p, H, H, h->blob + L.proj_w, (size_t)2 * I * H, h->blob + L.proj_b, M, 2 * I, stats, nullptr, 2 * I, 0, mask,
               nullptr, 1, T, st);
  if (hipMemcpyAsync(x, sw.xm, (size_t)M * H * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return vits_fail(h, "copy x");
  if (hipMemcpy2DAsync(m, (size_t)I * sizeof(float), stats, (size_t)2 * I * sizeof(float), (size_t)I * sizeof(float), M,
                       hipMemcpyDeviceToDevice, st) != hipSuccess)
    return vits_fail(h, "copy m");
  if (hipMemcpy2DAsync(logs, (size_t)I * sizeof(float), stats + I, (size_t)2 * I * sizeof(float), (size_t)I * sizeof(float), M,
                       hipMemcpyDeviceToDevice, st) != hipSuccess)
    return vits_fail(h, "copy logs");
  return vits_fail(h, "text_encoder");
}

size_t ttsvits_flow_workspace_bytes(const ttsvits_handle* h, int B, int T) {
  if (!h || B <= 0 || T <= 0) return 0;
  const size_t M = (size_t)B * T, I = h->d.inter_channels, Fh = h->d.flow_hidden;
  const size_t fl = up(M, kAlign) + 2 * up(M * I, kAlign) + up(M * (I / 2), kAlign) + 2 * up(M * Fh, kAlign) + up(M * Fh, kAlign) +
                    2 * up(M * 2 * Fh, kAlign) + up(M * (Fh > I / 2 ? Fh : I / 2), kAlign) + 3 * up(M * Fh, kAlign) +
                    up((size_t)B * 2 * Fh * h->d.flow_wn_layers, kAlign);
  return (stack_ws_floats(tf_dims(h->d), M) + fl) * sizeof(float);
}

int ttsvits_flow_reverse(ttsvits_handle* h, const float* z, const int32_t* lengths, const float* g, int B, int T, float* out,
                         void* workspace, size_t workspace_bytes, void* stream) {
  if (!h || !z || !lengths || !out || !workspace || B <= 0 || T <= 0) return TTSDEC_ERR_INVALID_ARG;
  if (g != nullptr && h->d.gin_channels <= 0) return TTSDEC_ERR_INVALID_ARG;
  if (!h->blob) return TTSDEC_ERR_NOT_BOUND;
  if (workspace_bytes < ttsvits_flow_workspace_bytes(h, B, T) || (reinterpret_cast<uintptr_t>(workspace) & 255)) return TTSDEC_ERR_WORKSPACE;
  if (!device_is_current(h->device)) return TTSDEC_ERR_DEVICE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const ttsvits_dims& d = h->d;
  const VitsBlob& L = h->bl;
  const float* blob = h->blob;
  const int M = B * T, I = d.inter_channels, half = I / 2, Fh = d.flow_hidden;
  float* p = static_cast<float*>(workspace);
  const StackDims sd = tf_dims(d);
  StackWs sw = carve_stack(p, sd, (size_t)M, h->precision == TTSDEC_PREC_SPLIT_F16);
  auto take = [&](size_t n) { float* r = p; p += up(n, kAlign); return r; };
  float* mask = take(M);
  float* xa = take((size_t)M * I);   // current x (ping)
  float* xb = take((size_t)M * I);   // flipped x (pong)
  float* mm = take((size_t)M * half);
  float* hx = take((size_t)M * Fh);   // WN running x
  float* ho = take((size_t)M * Fh);   // WN output accumulator
  float* acts = take((size_t)M * Fh);
  float* xin = take((size_t)M * 2 * Fh);
  float* rs = take((size_t)M * 2 * Fh);
  GemmCtx fcx;
  fcx.planes = reinterpret_cast<f16*>(take((size_t)M * (Fh > half ? Fh : half)));
  fcx.split = h->precision == TTSDEC_PREC_SPLIT_F16;
  f16* hx_p = reinterpret_cast<f16*>(take((size_t)M * Fh));  // planes of hx / acts / ho, written by their producers
  f16* acts_p = reinterpret_cast<f16*>(take((size_t)M * Fh));
  f16* ho_p = reinterpret_cast<f16*>(take((size_t)M * Fh));
  const int ncond = 2 * Fh * d.flow_wn_layers;
  float* cond = take((size_t)B * ncond);  // WN.cond_layer(g) of the current coupling layer, [B, 2 Fh n_layers]
  GemmCtx exact;
  exact.planes = nullptr; exact.split = false;
  hipLaunchKernelGGL(frame_mask_kernel, grid1(M), dim3(256), 0, st, lengths, T, mask, M);
  const float* cur = z;
  for (int f = d.n_flows - 1; f >= 0; --f) {  // models.py:807-809: reversed(flows) = Flip, layer_f, ...
    const FlowBlob& fb = L.flow[f];
    const bool i4 = I % 8 == 0 && (size_t)M * I < ((size_t)1 << 32) && reinterpret_cast<uintptr_t>(cur) % 16 == 0;  // (cur may be the caller's z)
    if (i4) hipLaunchKernelGGL(flip_split4_kernel, grid1((size_t)M * I / 4), dim3(256), 0, st, cur, mask, xb, sw.xm, sw.xm_p, (uint32_t)M, (uint32_t)I / 4);
    else hipLaunchKernelGGL(flip_split_kernel, grid1((size_t)M * I), dim3(256), 0, st, cur, mask, xb, sw.xm, sw.xm_p, M, I);
    // x0_ = pre_transformer(x0 * mask, mask) + x0                                   models.py:508-509
    int rc = run_stack(h, fb.tf, sd, sw, mask, B, T, st);
    if (rc != TTSDEC_OK) return rc;
    if (i4) hipLaunchKernelGGL(add_x04_kernel, grid1((size_t)M * half / 4), dim3(256), 0, st, sw.xm, sw.xm_p, xb, (uint32_t)M, (uint32_t)I / 4);
    else hipLaunchKernelGGL(add_x0_kernel, grid1((size_t)M * half), dim3(256), 0, st, sw.xm, sw.xm_p, xb, M, I);
    // h = pre(x0_) * mask                                                           :510
    gemm_generic(fcx, sw.xm, sw.xm_p, half, half, blob + fb.pre_w, (size_t)Fh * half, blob + fb.pre_b, M, Fh, hx, hx_p, Fh, 0, mask, nullptr, 1, T, st);
    // h = WN(h, mask, g)                                                            :511, modules.py:185-210
    if (g != nullptr)  // g = cond_layer(g)  (modules.py:189-190; a 1x1 conv of [B, gin, 1]: one small GEMM per coupling layer)
      gemm_generic(exact, g, nullptr, d.gin_channels, d.gin_channels, blob + fb.cond_w, (size_t)ncond * d.gin_channels, blob + fb.cond_b, B, ncond,
                   cond, nullptr, ncond, 0, nullptr, nullptr, 1, 1, st);
    for (int j = 0; j < d.flow_wn_layers; ++j) {
      const float* gl = g != nullptr ? cond + (size_t)j * 2 * Fh : nullptr;  // g_l = g[:, 2 Fh j : 2 Fh (j + 1)]      modules.py:194-196
      const bool last = j == d.flow_wn_layers - 1;
      gemm_generic(fcx, hx, hx_p, Fh, Fh, blob + fb.in_w[j], (size_t)2 * Fh * d.flow_kernel * Fh, blob + fb.in_b[j], M, 2 * Fh, xin, nullptr, 2 * Fh, 0,
                   nullptr, nullptr, d.flow_kernel, T, st);
      const bool vec4 = Fh % 4 == 0 && (size_t)M * Fh < ((size_t)1 << 32);
      if (vec4 && fcx.split) hipLaunchKernelGGL(wn_gate4_kernel<true>, grid1((size_t)M * Fh / 4), dim3(256), 0, st, xin, (Fh & 7) ? acts : (float*)nullptr, acts_p, (uint32_t)M, (uint32_t)Fh / 4, gl, (uint32_t)T, (uint32_t)ncond);
      else if (vec4) hipLaunchKernelGGL(wn_gate4_kernel<false>, grid1((size_t)M * Fh / 4), dim3(256), 0, st, xin, acts, acts_p, (uint32_t)M, (uint32_t)Fh / 4, gl, (uint32_t)T, (uint32_t)ncond);
      else hipLaunchKernelGGL(wn_gate_kernel, grid1((size_t)M * Fh), dim3(256), 0, st, xin, acts, acts_p, M, Fh, gl, T, ncond);
      const int cr = last ? Fh : 2 * Fh;
      gemm_generic(fcx, acts, acts_p, Fh, Fh, blob + fb.rs_w[j], (size_t)cr * Fh, blob + fb.rs_b[j], M, cr, rs, nullptr, cr, 0, nullptr, nullptr, 1, T, st);
      if (vec4) hipLaunchKernelGGL(wn_update4_kernel, grid1((size_t)M * Fh / 4), dim3(256), 0, st, hx, ho, rs, mask, hx_p, ho_p, (uint32_t)M, (uint32_t)Fh / 4, last ? 1 : 0, j == 0 ? 1 : 0);
      else hipLaunchKernelGGL(wn_update_kernel, grid1((size_t)M * Fh), dim3(256), 0, st, hx, ho, rs, mask, hx_p, ho_p, M, Fh, last ? 1 : 0, j == 0 ? 1 : 0);
    }
    // m = post(h) * mask ; x1 = (x1 - m) * mask                                     :517, 529
    gemm_generic(fcx, ho, ho_p, Fh, Fh, blob + fb.post_w, (size_t)half * Fh, blob + fb.post_b, M, half, mm, nullptr, half, 0, mask, nullptr, 1, T, st);
    if (i4) hipLaunchKernelGGL(couple4_kernel, grid1((size_t)M * half / 4), dim3(256), 0, st, xb, mm, mask, (uint32_t)M, (uint32_t)I / 4);
    else hipLaunchKernelGGL(couple_kernel, grid1((size_t)M * half), dim3(256), 0, st, xb, mm, mask, M, I);
    float* t = xa; xa = xb; xb = t;  // the coupled tensor becomes the next layer's input
    cur = xa;
  }
  hipLaunchKernelGGL(copy_f_kernel, grid1((size_t)M * I), dim3(256), 0, st, cur, out, (size_t)M * I);
  return vits_fail(h, "flow_reverse");
}

}  // extern "C"
