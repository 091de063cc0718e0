/*
 * ttsdec.h - C ABI of libttsdec.so: the MI355X (gfx950) implementation of the
 * Tacotron autoregressive mel-decoder hot path of kgoba/torch-tts.
 *
 * The reference has no FFI / plugin interface for this path: its boundary is the
 * Python module API (tacotron/decoder.py:6-18 Decoder, tacotron/decoder_cell.py:143-195
 * Taco2ProdDecoderCell, tacotron/modules/modules.py:155-184 MelPostnet).  This C ABI is
 * what the host-side mirror of those modules (the Python files of torch-tts_amd/, ctypes) binds; each
 * entry point below names the reference code it replaces.  INTEGRATION.md shows the
 * reference-side stub.
 *
 * Conventions
 *   - plain pointers and sizes only; every data pointer is a DEVICE pointer (fp32,
 *     row-major contiguous) unless it says "host".
 *   - the caller owns every buffer (weights blob, workspace, inputs, outputs); the
 *     library allocates nothing on the device and keeps only the pointers bound with
 *     ttsdec_bind_weights().
 *   - all work is enqueued on the hipStream_t passed in (as void*); nothing on the decode / postnet / encoder / VITS2 paths
 *     synchronises.  The two exceptions are set-up calls and say so below: ttsdec_pack_weights (one stream synchronisation +
 *     an 8-byte read-back for the range guard) and ttsdec_bind_weights (a synchronous 8-byte copy of the same header).
 *     Calls on one handle must be serialised by the caller.
 *   - return value: 0 = TTSDEC_OK, negative = error (ttsdec_strerror()).  Nothing
 *     throws or aborts across this boundary.
 *   - the handle is bound to the HIP device that was current at ttsdec_create();
 *     that device must be current for every later call.
 */
#ifndef TTSDEC_H
#define TTSDEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version: bumped whenever an entry point's argument list or a struct layout changes.  2 (round 4): ttsenc_forward /
 * ttsvits_text_encoder / ttsvits_flow_reverse carry `g` and a status word, ttsvits_dims two more fields (round 3, unversioned
 * then); ttsenc_set_precision / ttsenc_get_precision added, and exact fp32 became the default arithmetic of the ttsenc_ / ttsvits_
 * handles as it always was of the ttsdec_ ones (split-fp16 is opt-in everywhere).  The Python binding refuses a library whose ttsdec_version() differs from the version it was written for. */
#define TTSDEC_VERSION 2

enum {
  TTSDEC_OK = 0,
  TTSDEC_ERR_INVALID_ARG = -1, /* null pointer, negative size, bad enum */
  TTSDEC_ERR_DIMS = -2,        /* a feature dimension is not a multiple of 4 / out of range */
  TTSDEC_ERR_HIP = -3,         /* a HIP runtime call failed (hipGetLastError text via ttsdec_last_hip_error) */
  TTSDEC_ERR_NOT_BOUND = -4,   /* decode/postnet before ttsdec_bind_weights */
  TTSDEC_ERR_WORKSPACE = -5,   /* workspace too small or misaligned */
  TTSDEC_ERR_DEVICE = -6       /* current device differs from the handle's, or is not gfx950 */
};

/* Model dimensions.  Source of truth: configs/config-ljspeech.yaml:47-69 via
 * build_tacotron (tacotron/tacotron.py:165-214). */
typedef struct ttsdec_dims {
  int32_t d_mel;           /* audio.num_mels                     (80)   */
  int32_t r;               /* decoder.r, frames per step          (1)   */
  int32_t d_pre;           /* decoder.dim_pre, both PreNet layers (256) */
  int32_t d_ctx;           /* encoder.dim_out                     (512) */
  int32_t h_att;           /* decoder.dim_rnn[0]                  (1024)*/
  int32_t h_dec;           /* decoder.dim_rnn[1]                  (1024)*/
  float p_zoneout;         /* decoder_cell.py:145                 (0.1) */
  float p_dropout;         /* modules.py:25 PreNet p_dropout      (0.5) */
  int32_t postnet_layers;  /* model.postnet.num_layers            (3); 0 = no postnet */
  int32_t postnet_hidden;  /* model.postnet.dim_hidden            (512) */
  int32_t postnet_kernel;  /* MelPostnet kernel_size              (5)   */
  float bn_eps;            /* nn.BatchNorm1d eps                  (1e-5)*/
  int32_t cell_type;       /* TTSDEC_CELL_*: which decoder cell of decoder_cell.py                  */
  int32_t d_pre_hidden;    /* width of PreNet layer 0; 0 = d_pre (Taco2ProdDecoderCell, decoder_cell.py:152);
                            * Taco2DecoderCell uses 128 (decoder_cell.py:74-76)                      */
  int32_t postnet_type;    /* TTSDEC_POSTNET_TYPE_*                                                  */
} ttsdec_dims;

/* Decoder cells (tacotron/tacotron.py:171-176 picks by model.decoder.type). */
enum {
  TTSDEC_CELL_TACO2PROD = 0, /* Taco2ProdDecoderCell (decoder_cell.py:143-195): LJSpeech config.  h_att / h_dec are the
                              * attention-rnn and decoder-rnn widths.                                               */
  TTSDEC_CELL_TACO2 = 1      /* Taco2DecoderCell (decoder_cell.py:66-140): rdh / sandra / template configs.  Context from
                              * the previous weights feeds both stacked LSTMs (h_att = dim_rnn[0], h_dec = dim_rnn[1]);
                              * query and projection read cat[h0, h1, zeros]; no ctx in the state.                   */
};
/* Postnets (tacotron/tacotron.py:199-214 picks by model.postnet.type). */
enum {
  TTSDEC_POSTNET_TYPE_MEL = 0,  /* MelPostnet  (modules/modules.py:155-184)                                            */
  TTSDEC_POSTNET_TYPE_MEL2 = 1  /* MelPostnet2 (modules/modules.py:187-216): postnet_layers residual blocks of
                                 * Conv1dFix-BN-LeakyReLU x2 + Conv1dFix (mps_fixes.py:6-29)                           */
};

typedef struct ttsdec_handle ttsdec_handle;

/* Order of the source tensors handed to ttsdec_pack_weights: the reference's
 * state-dict order.  Postnet entries repeat per layer. */
enum {
  TTSDEC_W_PRE0_W = 0, /* decoder.decoder_cell.pre_net.layers.0.weight [d_pre_hidden, d_mel] */
  TTSDEC_W_PRE0_B,     /* ...layers.0.bias   [d_pre_hidden]                                   */
  TTSDEC_W_PRE1_W,     /* ...layers.1.weight [d_pre, d_pre_hidden]                            */
  TTSDEC_W_PRE1_B,     /* ...layers.1.bias   [d_pre]                                          */
  TTSDEC_W_QUERY_W,    /* decoder_cell.attention_module.query_layer.weight [d_ctx, h_att]     */
  TTSDEC_W_ATT_IH,     /* decoder_cell.attention_rnn.weight_ih [4*h_att, d_pre+d_ctx]         */
  TTSDEC_W_ATT_HH,     /* ...weight_hh [4*h_att, h_att]                                       */
  TTSDEC_W_ATT_BIH,    /* ...bias_ih   [4*h_att]                                              */
  TTSDEC_W_ATT_BHH,    /* ...bias_hh   [4*h_att]                                              */
  TTSDEC_W_DEC_IH,     /* decoder_cell.decoder_rnn.weight_ih [4*h_dec, h_att+d_ctx]           */
  TTSDEC_W_DEC_HH,     /* ...weight_hh [4*h_dec, h_dec]                                       */
  TTSDEC_W_DEC_BIH,    /* ...bias_ih   [4*h_dec]                                              */
  TTSDEC_W_DEC_BHH,    /* ...bias_hh   [4*h_dec]                                              */
  TTSDEC_W_INIT_H0,    /* decoder_cell.initial_decoder_h.0 [1, h_att]                         */
  TTSDEC_W_INIT_H1,    /* decoder_cell.initial_decoder_h.1 [1, h_dec]                         */
  TTSDEC_W_INIT_C0,    /* decoder_cell.initial_decoder_c.0 [1, h_att]                         */
  TTSDEC_W_INIT_C1,    /* decoder_cell.initial_decoder_c.1 [1, h_dec]                         */
  TTSDEC_W_MEL_W,      /* decoder.fc_mel.weight  [r*d_mel, h_dec+d_ctx]                       */
  TTSDEC_W_MEL_B,      /* decoder.fc_mel.bias    [r*d_mel]                                    */
  TTSDEC_W_STOP_W,     /* decoder.fc_stop.weight [r, h_dec+d_ctx]                             */
  TTSDEC_W_STOP_B,     /* decoder.fc_stop.bias   [r]                                          */
  /* TTSDEC_CELL_TACO2 uses the same slots: QUERY_W is [d_ctx, h_att+h_dec+d_ctx]; ATT_* / DEC_* are
   * decoder_rnn_list.0 / .1 ([4*h_att, d_pre+d_ctx], [4*h_dec, h_att+d_ctx]); MEL_W / STOP_W have
   * h_att+h_dec+d_ctx columns. */
  TTSDEC_W_DECODER_COUNT, /* = 21; postnet tensors follow:                                   */
  /* per layer i (5 each): postnet.conv.i.0.weight [C_out, C_in, k], conv.i.1.weight [C_out],
   * conv.i.1.bias, conv.i.1.running_mean, conv.i.1.running_var; then postnet.fc_out.weight
   * [d_mel, hidden].  Total = 21 + 5*postnet_layers + 1 (or 21 when postnet_layers == 0). */
  TTSDEC_W_POSTNET_PER_LAYER = 5,
  /* TTSDEC_POSTNET_TYPE_MEL2, per layer i (11 each, no fc_out): postnet.layers.i.1.weight [hidden, d_mel, k],
   * layers.i.2.{weight,bias,running_mean,running_var}, layers.i.5.weight [hidden, hidden, k],
   * layers.i.6.{weight,bias,running_mean,running_var}, layers.i.9.weight [d_mel, hidden, k]. */
  TTSDEC_W_POSTNET2_PER_LAYER = 11
};

/* Prenet dropout modes (the reference's dropout is always on, modules.py:40). */
enum {
  TTSDEC_DROPOUT_OFF = 0,   /* no dropout (not reference behaviour; for analysis)              */
  TTSDEC_DROPOUT_MASKS = 1, /* keep-masks injected: uint8 [steps, 2, B, d_pre], 1 = keep       */
  TTSDEC_DROPOUT_PHILOX = 2 /* on-device Philox4x32-10 keyed by (seed; step, layer, b, unit): one keep BIT per
                             * unit, i.e. keep probability exactly 1/2 - valid only for p_dropout == 0.5 (the
                             * reference's value, modules.py:25); other p: TTSDEC_ERR_INVALID_ARG, use MASKS   */
};

/* Arithmetic of the step's GEMMs: the two LSTM gate GEMMs (95 % of the step's FLOPs), the PreNet, the query and the
 * mel/stop projection. */
enum {
  TTSDEC_PREC_F32 = 0,       /* exact fp32 on the fp32-input matrix instruction (default)             */
  TTSDEC_PREC_SPLIT_F16 = 1  /* split-fp16: x = hi + lo*2^-11 (two fp16 planes, 22 significand bits),
                              * a*b = ah*bh + (ah*bl + al*bh)*2^-11 on the f16 matrix instruction with
                              * fp32 accumulation; ~2^-22 relative per product.  Needs d_pre, d_ctx,
                              * h_att, h_dec to be multiples of 8, else the handle stays on F32.      */
};

/* Postnet arithmetic. */
enum {
  TTSDEC_POSTNET_F32 = 0,       /* exact fp32 (fp32-input MFMA)                                              */
  TTSDEC_POSTNET_BF16 = 1,      /* bf16 operands on the bf16 MFMA, fp32 accumulate (BASELINE.json configs[2]):
                                 * ~3 significant digits; the tolerance is reported by the tests/bench       */
  TTSDEC_POSTNET_SPLIT_F16 = 2  /* split-fp16 operands (see TTSDEC_PREC_SPLIT_F16): fp32-grade               */
};                              /* 16-bit modes need d_mel and postnet_hidden to be multiples of 8, else F32 is used */

int ttsdec_version(void);
const char* ttsdec_strerror(int code);
/* Text of the last HIP error seen on this handle ("" if none).  Host string, owned by the library. */
const char* ttsdec_last_hip_error(const ttsdec_handle* h);

/* Replaces: module construction in build_tacotron (tacotron/tacotron.py:178-206). */
int ttsdec_create(const ttsdec_dims* dims, ttsdec_handle** out);
int ttsdec_destroy(ttsdec_handle* h);

/* Selects TTSDEC_PREC_* for later decode / cell_step calls (weights of both forms live in
 * the packed blob, so this can be switched at any time).  ttsdec_get_precision returns the
 * mode actually in effect for the handle's dims. */
int ttsdec_set_precision(ttsdec_handle* h, int precision);
int ttsdec_get_precision(const ttsdec_handle* h);

/* Tuning and measurement options of a decoder handle.  The defaults (value -1) are the measured-best settings per batch size
 * and arithmetic mode; nothing here changes results beyond the summation order of a GEMM's K segments (every setting is
 * held to the same parity bar by tests/test_hip_parity.py::test_all_step_orders_and_layouts_vs_oracle).  Setting an
 * option drops the handle's captured graph.  MEASUREMENT ONLY - not part of the drop-in surface.
 * The environment variable TTSDEC_OPTIONS="name=value,name=value" (names below, lower case without the prefix) presets
 * the options of every handle created afterwards; it is read once per ttsdec_create.  The only other environment
 * variables the library reads, all measurement-only as well: TTSDEC_STAMPS=<file> (per-workgroup time stamps of the
 * two-role launches of a decode call's last step, written to <file>; synchronises the stream),
 * TTSDEC_NO_LEAN_SKINNY=1 (short-K row GEMMs of the VITS2 path back on the 128 x 128 tile), TTSDEC_LEAN8_F32_MAX=<rows>
 * (largest batch on the 32-row exact-fp32 LSTM tile), TTSDEC_CONV256=0 (the Postnet's hidden conv layers back on the
 * shared GEMM tile) and TTSDEC_CONV256_FORCE=1 (those layers on the 256 x 256 kernel even where its tiles do not fill
 * the chip - the tests' switch). */
enum {
  TTSDEC_OPT_GRAPH = 0,        /* "graph": 0 = launch every step kernel from the host instead of replaying a captured hipGraph  */
  TTSDEC_OPT_OVERLAP,          /* "overlap": two-role launches 0 = none, 1 = frame || lstm_att, 2 = also attention || lstm_dec;
                                * 3 = the whole step as ONE launch where it applies (split-fp16, projection head role on, at most
                                * 256 utterances; level 2 elsewhere).  Never the default: measured slower at every batch size,
                                * DESIGN.md section 4.5                                                                        */
  TTSDEC_OPT_CHUNK_A,          /* "chunk_a": 0 = row-major fp16 activation planes instead of the chunked layout                */
  TTSDEC_OPT_CHUNK_B,          /* "chunk_b": 0 = row-major LSTM weight planes                                                  */
  TTSDEC_OPT_PROJ_REGW,        /* "proj_regw": 0 = mel/stop projection on the LDS-staged split-K GEMM                          */
  TTSDEC_OPT_HEAD_PROJ,        /* "head_proj": 1 / 0 = that projection as a role at the head of the next step's first launch   */
  TTSDEC_OPT_QUERY_ROLE,       /* "query_role": 1 / 0 = the attention query GEMM as a job of the attention role's workgroups
                                * (overlap 2, at most 512 utterances) instead of a launch of its own                          */
  TTSDEC_OPT_MERGED_TUNE,      /* "merged_tune": measurement knobs of the one-launch step (overlap = 3): bit 0 = the decoder LSTM waits for
                                * h_att before its first tile, bit 1 = wave priority 2 / 0 for the attention / decoder LSTM roles,
                                * bits 8-15 = extra sleeps between the query role's polls for h_att                              */
  TTSDEC_OPT_PROFILE_ABLATION, /* "profile_ablation": ttsdec_profile_step only, kernel-internal ablation switches              */
  TTSDEC_OPT_DEBUG_FLAGS,      /* "debug_flags": TEST HOOK. bit 0 / 1 / 2 / 3: the frame / attention / projection-head role of a
                                * two-role launch / the attention LSTM's tiles of the one-launch step do not signal their
                                * consumers, which then run into the bounded-spin time-out (T_out[1] bit 2)                   */
  TTSDEC_OPT_SPIN_LIMIT,       /* "spin_limit": TEST HOOK. polls before a consumer role gives up (default 4096, ~1-4 ms)         */
  TTSDEC_OPT_COUNT
};
int ttsdec_set_option(ttsdec_handle* h, int option, int value);
int ttsdec_get_option(const ttsdec_handle* h, int option, int* value);
const char* ttsdec_option_name(int option); /* NULL for an unknown option */

/* Number of source tensors ttsdec_pack_weights expects for these dims. */
int ttsdec_num_weight_tensors(const ttsdec_handle* h);
/* Size of the packed weight blob (bytes, multiple of 256). */
size_t ttsdec_packed_bytes(const ttsdec_handle* h);

/* Packs the reference-layout parameters into the kernel layout inside `blob`
 * (caller-allocated, ttsdec_packed_bytes(), 256-B aligned): sums the LSTM bias
 * pairs, stacks fc_mel/fc_stop, transposes conv weights to [C_out][tap][C_in],
 * turns BatchNorm running stats into per-channel (alpha, beta).  The blob is
 * position-independent: it can be broadcast to other GPUs (RCCL) and bound there.
 * Range guard of the split-fp16 planes: the call also reduces max |w| over the matrices that have
 * planes into the blob header and reads it back - the ONE place this library synchronises `stream`.
 * A weight with |w| >= 65504 (or NaN) has no split-fp16 form: the handle then stays on exact fp32
 * for the affected GEMMs whatever ttsdec_set_precision / the postnet precision ask
 * (ttsdec_get_precision reports the mode in effect).
 * Replaces: nn.Module parameter storage / load_state_dict (train_util.py:23-45). */
int ttsdec_pack_weights(ttsdec_handle* h, const float* const* src /*host array of device ptrs; NULL entry = skip*/, int n_src,
                        void* blob, void* stream);
/* Binds a packed blob (from ttsdec_pack_weights here or on another rank).  The blob must be complete
 * (e.g. the broadcast that fills it finished): its header (the range guard's maxima) is read back here
 * with a synchronous copy. */
int ttsdec_bind_weights(ttsdec_handle* h, const void* blob);

/* Decoder workspace (recurrent state + scratch) for a batch of B utterances with
 * memory length L.  The state persists in the workspace between ttsdec_decode calls. */
size_t ttsdec_workspace_bytes(const ttsdec_handle* h, int B, int L);

/*
 * Runs decode steps t_begin .. t_begin+n_steps-1 of Decoder.forward
 * (tacotron/decoder.py:47-71) for the whole batch; one step = PreNet ->
 * attention LSTM -> stepwise-monotonic attention -> context -> decoder LSTM ->
 * mel/stop projection (tacotron/decoder_cell.py:180-195).
 *
 *   memory        [B, L, d_ctx] encoder outputs (zero on padded rows)
 *   t_begin       global index of the first step of this call (must be even); 0 (re)initialises
 *                 the recurrent state from the initial_decoder_{h,c} parameters
 *                 (decoder_cell.py:165-178) and the GO frame (decoder.py:35)
 *   n_steps       steps to run at most in this call
 *   t_stride      capacity (in steps) of the output buffers of this call, >= n_steps
 *   stop_threshold / check_stop
 *                 inference stop rule (decoder.py:68): the first step at which ANY
 *                 utterance's stop logit < threshold is the last one produced
 *                 (inclusive, batch-global); later steps of this and following
 *                 calls are skipped.  check_stop = 0 disables it (teacher mode).
 *   dropout_mode  TTSDEC_DROPOUT_*; masks for MASKS: per step (relative to t_begin) the keep-mask
 *                 of PreNet layer 0 [B, d_pre_hidden] followed by layer 1 [B, d_pre], uint8
 *                 (= [n_steps, 2, B, d_pre] when the two widths are equal); seed for PHILOX
 *   teacher       optional [B, teacher_T, d_mel] ground-truth frames (decoder.py:38-42);
 *   teacher_flags optional uint8 [>= t_begin+n_steps]: flags[t-1] != 0 => the input of
 *                 step t (t >= 1) is teacher frame t*r-1 instead of the model's own
 *                 last frame (decoder.py:65-66).  NULL teacher = free-running.
 *   y [B, t_stride*r, d_mel], s [B, t_stride*r], w [B, t_stride, L]
 *                 outputs of this call, step t stored at row (t - t_begin)
 *   T_out         device int32[2]: [0] = total number of steps produced so far
 *                 (= stop step + 1 if the rule fired, else t_begin+n_steps),
 *                 [1] = bit 0: the stop rule has fired; bit 2: a two-role launch of the step gave up waiting
 *                 for its producer role (bounded spin of ~1-4 ms, once per call: later gates of the same call
 *                 return at once; only possible when the producer role's workgroups are not resident - a GPU
 *                 shared with another process's kernels, a stalled queue): the outputs of the call are INVALID.
 *                 What a direct caller of this ABI must do then (torch-tts_amd/decoder.py does exactly this):
 *                 ttsdec_set_option(h, TTSDEC_OPT_OVERLAP, 0) and ttsdec_set_option(h, TTSDEC_OPT_HEAD_PROJ, 0) -
 *                 one role per launch, no hand-off inside a launch - and decode the utterance batch again from
 *                 t_begin = 0 (the recurrent state left in the workspace is not a state of the sequence any more);
 *                 bit 1 (split-fp16 mode only): an activation
 *                 entering a 16-bit GEMM (input / teacher frame, PreNet output, context) had
 *                 |x| > 65504, the fp16 range - it was SATURATED, never inf/NaN, so outputs stay
 *                 finite but those rows are not fp32-accurate: rerun with TTSDEC_PREC_F32.
 *                 (Activation bound of the split mode: |x| <= 65504; h is bounded by 1, the context
 *                 by max |memory|.)
 *   State after a fired stop: the outputs up to and including the stop step are final, but the recurrent state left in the
 *   workspace is NOT the state at the stop step (launches that overlap the next step's early work have already advanced
 *   parts of it: the attention LSTM's cell state is one step past the stop frame).  A decode that stopped cannot be
 *   continued; start the next utterance batch with t_begin = 0.
 */
int ttsdec_decode(ttsdec_handle* h, const float* memory, int B, int L, int t_begin, int n_steps, int t_stride,
                  float stop_threshold, int check_stop, int dropout_mode, const uint8_t* masks, uint64_t seed,
                  const float* teacher, int teacher_T, const uint8_t* teacher_flags, float* y, float* s, float* w,
                  int32_t* T_out, void* workspace, size_t workspace_bytes, void* stream);

/* Postnet scratch for B*T frames. */
size_t ttsdec_postnet_workspace_bytes(const ttsdec_handle* h, int B, int T);

/* MelPostnet.forward in eval mode (tacotron/modules/modules.py:178-184):
 * y [B, T, d_mel] -> y_post [B, T, d_mel] = y + fc_out(isru(BN(conv(...)))); or, for
 * TTSDEC_POSTNET_TYPE_MEL2, MelPostnet2.forward (modules.py:213-216). */
int ttsdec_postnet(ttsdec_handle* h, const float* y, int B, int T, int precision, float* y_post, void* workspace,
                   size_t workspace_bytes, void* stream);

/* One decoder-cell step on caller-held state (Taco2ProdDecoderCell.forward,
 * tacotron/decoder_cell.py:180-195), for callers that drive the cell directly.
 * Not needed by Decoder.forward; provided for API completeness.  State tensors are
 * updated in place: w [B,L], ctx [B,d_ctx], h_att/c_att [B,h_att], h_dec/c_dec [B,h_dec];
 * x [B, d_mel] is the input frame; x_dec [B, h_dec+d_ctx] receives cat[h_dec, ctx].
 * TTSDEC_CELL_TACO2 (decoder_cell.py:110-140): ctx is output only (the context the LSTMs consumed,
 * = bmm(w_in, memory)); x_dec is [B, h_att+h_dec+d_ctx] = cat[h0, h1, zeros].
 * masks: [2, B, d_pre] uint8 or NULL per dropout_mode; step only keys the Philox stream. */
int ttsdec_cell_step(ttsdec_handle* h, const float* x, const float* memory, int B, int L, float* w, float* ctx,
                     float* h_att, float* c_att, float* h_dec, float* c_dec, int dropout_mode, const uint8_t* masks,
                     uint64_t seed, int step, float* x_dec, void* workspace, size_t workspace_bytes, void* stream);

/* Measurement aid for bench.py: runs `iters` decode steps on the current
 * workspace state and reports the mean duration (ms, HIP events on `stream`) of each
 * kernel of the step, in launch order, into ms_out[0..n_out) (host array); returns the
 * number of kernels per step in *n_kernels and their names (static strings) in names_out.
 * The profiled step is step 1 of a two-step call: y [B, 2r, d_mel], s [B, 2r], w [B, 2, L]. */
int ttsdec_profile_step(ttsdec_handle* h, const float* memory, int B, int L, int iters, int dropout_mode,
                        const uint8_t* masks, uint64_t seed, float* y, float* s, float* w, void* workspace,
                        size_t workspace_bytes, void* stream, float* ms_out, const char** names_out, int n_out,
                        int* n_kernels);

/* Measurement aid for bench.py: the step's launches timed INSIDE the replayed graph of the step loop.  Runs one ordinary
 * decode call of n_steps steps from step 0 (a multiple of the graph's 30 steps, >= 60; stop rule off) in which workgroup 0 of
 * every step kernel stores its start time (one 8-byte store per launch: the only difference to ttsdec_decode); ms_out[k] is
 * the mean time from launch k's start to the next launch's start - the launch's duration in the loop including the gap
 * behind it - over the last graph replay; the entries add up to *step_ms, the loop's time per step.  Synchronises the
 * stream.  Names and counts as for ttsdec_profile_step; y [B, n_steps*r, d_mel], s [B, n_steps*r], w [B, n_steps, L]. */
int ttsdec_profile_loop(ttsdec_handle* h, const float* memory, int B, int L, int n_steps, int dropout_mode, const uint8_t* masks,
                        uint64_t seed, float* y, float* s, float* w, int32_t* T_out, void* workspace, size_t workspace_bytes, void* stream,
                        float* ms_out, const char** names_out, int n_out, int* n_kernels, float* step_ms);

/* ---------------------------------------------------------------------------------------
 * Text encoder: Encoder2.forward in eval mode, tacotron/encoder.py:27-82
 * with the packed bidirectional LSTM of tacotron/modules/rnn.py:112-127.  Produces the `memory`
 * the decoder consumes.  Runs once per batch; split-fp16 GEMMs for the convs and the input projection (fp32-class accuracy), exact fp32 recurrence.
 * ------------------------------------------------------------------------------------- */
typedef struct ttsenc_dims {
  int32_t alphabet_size; /* 1 + len(text.alphabet) (+ phonemes), tacotron.py:189-191 */
  int32_t d_emb;         /* model.encoder.dim_emb (512): embedding and conv channels  */
  int32_t d_out;         /* model.encoder.dim_out (512): 2 x LSTM hidden              */
  int32_t conv_kernel;   /* 5                                                          */
  float bn_eps;          /* 1e-5                                                       */
} ttsenc_dims;
typedef struct ttsenc_handle ttsenc_handle;

/* Source tensors for ttsenc_pack_weights, the reference's state-dict order below ``encoder.``. */
enum {
  TTSENC_W_EMB = 0,   /* emb.weight [alphabet, d_emb]                                   */
  TTSENC_W_CONV0,     /* conv.0.weight [d_emb, d_emb, k]                                */
  TTSENC_W_BN0_W, TTSENC_W_BN0_B, TTSENC_W_BN0_MEAN, TTSENC_W_BN0_VAR, /* conv.1.*      */
  TTSENC_W_CONV1,     /* conv.3.weight                                                  */
  TTSENC_W_BN1_W, TTSENC_W_BN1_B, TTSENC_W_BN1_MEAN, TTSENC_W_BN1_VAR, /* conv.4.*      */
  TTSENC_W_CONV2,     /* conv.6.weight                                                  */
  TTSENC_W_BN2_MEAN, TTSENC_W_BN2_VAR, /* conv.7.running_{mean,var} (affine=False)      */
  TTSENC_W_IH_FWD,    /* rnn.rnn.weight_ih_l0 [4H, 2*d_emb], H = d_out/2                */
  TTSENC_W_HH_FWD,    /* rnn.rnn.weight_hh_l0 [4H, H]                                   */
  TTSENC_W_IH_REV,    /* rnn.rnn.weight_ih_l0_reverse                                   */
  TTSENC_W_HH_REV,    /* rnn.rnn.weight_hh_l0_reverse                                   */
  TTSENC_W_H0,        /* rnn_h0 [1, 1, d_out]                                           */
  TTSENC_W_C0,        /* rnn_c0 [1, 1, d_out]                                           */
  TTSENC_W_COUNT
};

int ttsenc_create(const ttsenc_dims* dims, ttsenc_handle** out);
int ttsenc_destroy(ttsenc_handle* h);
const char* ttsenc_last_hip_error(const ttsenc_handle* h);
int ttsenc_num_weight_tensors(const ttsenc_handle* h);
size_t ttsenc_packed_bytes(const ttsenc_handle* h);
int ttsenc_pack_weights(ttsenc_handle* h, const float* const* src, int n_src, void* blob, void* stream);
int ttsenc_bind_weights(ttsenc_handle* h, const void* blob);
size_t ttsenc_workspace_bytes(const ttsenc_handle* h, int B, int L);
/* ids [B, L] int64 (0 = padding), lengths [B] int32 on the device; L_out = max(lengths) (host-known:
 * the reference pads its output to the longest utterance, rnn.py:126); memory [B, L_out, d_out].
 * status: optional device int32 (caller zeroes it): bit 0 is set when an id lay outside [0, alphabet_size) - nn.Embedding raises
 * IndexError there (encoder.py:69); the kernel reads the nearest table row instead of memory outside the table and the caller
 * reads the word with the results (no scan of the ids on the host, no synchronisation before the launch). */
int ttsenc_forward(ttsenc_handle* h, const int64_t* ids, const int32_t* lengths, int B, int L, int L_out, float* memory,
                   void* workspace, size_t workspace_bytes, void* stream, int32_t* status);
/* Arithmetic of the encoder's conv and input-projection GEMMs: TTSDEC_PREC_F32 (default: exact fp32, the reference's own) or
 * TTSDEC_PREC_SPLIT_F16 (two fp16 planes per operand; needs d_emb % 8 == 0, else exact fp32 stays).  The recurrence is always
 * exact fp32.  Both weight forms live in the packed blob: switchable per call. */
int ttsenc_set_precision(ttsenc_handle* h, int precision);
int ttsenc_get_precision(const ttsenc_handle* h);

/* ---------------------------------------------------------------------------------------
 * VITS2 second hot path (BASELINE.json configs[4]):
 *   ttsvits_text_encoder  = TextEncoder.forward, vits2/models.py:369-380
 *       (attentions.Encoder :76-93, MultiHeadAttention.attention :246-295 with the relative-position
 *        window :297-368, FFN :411-419, modules.LayerNorm modules.py:24-27)
 *   ttsvits_flow_reverse  = ResidualCouplingTransformersBlock.forward(reverse=True), models.py:803-810
 *       over ResidualCouplingTransformersLayer.forward, models.py:506-531 (mean-only), with
 *       modules.WN.forward modules.py:185-210, commons.fused_add_tanh_sigmoid_multiply commons.py:102-109
 *       and modules.Flip modules.py:374-381.
 * Eval mode, no speaker conditioning (g = None), split-fp16 GEMMs (fp32-class accuracy: hi+lo fp16 planes, fp32 accumulate), fp32 elsewhere.  Activations at this boundary are
 * CHANNEL-LAST: [B, T, C] (the reference's [B, C, T] transposed).
 * ------------------------------------------------------------------------------------- */
typedef struct ttsvits_dims {
  int32_t n_vocab;          /* len(symbols)                                              */
  int32_t inter_channels;   /* 192: flow channels; TextEncoder emits m, logs of this width */
  int32_t hidden_channels;  /* 192                                                       */
  int32_t filter_channels;  /* 768                                                       */
  int32_t n_heads;          /* 2                                                         */
  int32_t n_layers;         /* 6                                                         */
  int32_t kernel_size;      /* 3 (FFN convs of the text encoder)                         */
  int32_t window_size;      /* 4 (relative-position window; heads share the tables)      */
  int32_t flow_hidden;      /* 192: WN width                                             */
  int32_t flow_kernel;      /* 5:   WN conv taps (dilation_rate 1)                       */
  int32_t flow_wn_layers;   /* 4                                                         */
  int32_t n_flows;          /* 4 coupling layers (each followed by a Flip)               */
  int32_t flow_tf_layers;   /* 2: pre_transformer = Encoder(half, half, 2 heads, 2 layers, k=3, no window) */
  int32_t flow_tf_heads;    /* 2                                                         */
  int32_t flow_tf_kernel;   /* 3                                                         */
  int32_t gin_channels;     /* 0, or the speaker-embedding width (a multiple of 4): each coupling layer's WN gets cond_layer
                             * (modules.py:149-153) and the text encoder spk_emb_linear (attentions.py:42-46)                */
  int32_t cond_layer_idx;   /* text-encoder layer at whose input the projected embedding is added (attentions.py:47-52: 2
                             * unless configured); read only when gin_channels > 0 and n_layers > 0                           */
} ttsvits_dims;
typedef struct ttsvits_handle ttsvits_handle;

/* Source tensors for ttsvits_pack_weights (device fp32, the reference's parameter shapes), in this order:
 *   enc_p.emb.weight;
 *   when gin_channels > 0: enc_p.encoder.spk_emb_linear.{weight,bias};
 *   per text-encoder layer i: attn_layers.i.conv_{q,k,v,o}.{weight,bias} (8), emb_rel_k, emb_rel_v,
 *       norm_layers_1.i.{gamma,beta}, ffn_layers.i.conv_1.{weight,bias}, conv_2.{weight,bias},
 *       norm_layers_2.i.{gamma,beta}                                            (18 per layer);
 *   enc_p.proj.{weight,bias};
 *   per coupling layer f (flow.flows.{2f}): per pre_transformer layer the same 16 tensors without the
 *       emb_rel pair; pre.{weight,bias}; when gin_channels > 0: enc.cond_layer EFFECTIVE weight [2*flow_hidden*
 *       flow_wn_layers, gin(, 1)], bias; per WN layer j: in_layers.j EFFECTIVE weight (g*v/||v||), bias,
 *       res_skip_layers.j effective weight, bias; post.{weight,bias}.
 * A NULL entry leaves that tensor zero (a module that owns only the text encoder or only the flow). */
int ttsvits_create(const ttsvits_dims* dims, ttsvits_handle** out);
int ttsvits_destroy(ttsvits_handle* h);
/* Arithmetic of every GEMM of the two entry points below: TTSDEC_PREC_F32 (default: exact fp32 matrix instruction, the
 * reference's own arithmetic) or TTSDEC_PREC_SPLIT_F16 (hi + lo fp16 planes, fp32 accumulate: opt-in, ~1.8x faster); both forms
 * of the weights live in the packed blob, so this can be switched at any time.  The elementwise math is fp32 in both modes;
 * the flow's attention runs on the f16 flash kernel in split mode and on exact-fp32 MFMAs otherwise. */
int ttsvits_set_precision(ttsvits_handle* h, int precision);
int ttsvits_get_precision(const ttsvits_handle* h);
const char* ttsvits_last_hip_error(const ttsvits_handle* h);
int ttsvits_num_weight_tensors(const ttsvits_handle* h);
size_t ttsvits_packed_bytes(const ttsvits_handle* h);
int ttsvits_pack_weights(ttsvits_handle* h, const float* const* src, int n_src, void* blob, void* stream);
int ttsvits_bind_weights(ttsvits_handle* h, const void* blob);
size_t ttsvits_text_encoder_workspace_bytes(const ttsvits_handle* h, int B, int T);
/* ids [B, T] int64, lengths [B] int32 (device).  x [B, T, hidden], m and logs [B, T, inter]; padded frames are zero.
 * g: NULL, or the speaker embedding [B, gin_channels] fp32 (device) - the reference's g [B, gin, 1] (models.py:369, 376;
 * attentions.py:80-84); TTSDEC_ERR_INVALID_ARG when g is given and the handle has gin_channels == 0.
 * status: as for ttsenc_forward (bit 0: an id outside [0, n_vocab), models.py:370). */
int ttsvits_text_encoder(ttsvits_handle* h, const int64_t* ids, const int32_t* lengths, const float* g, int B, int T, float* x,
                         float* m, float* logs, void* workspace, size_t workspace_bytes, void* stream, int32_t* status);
size_t ttsvits_flow_workspace_bytes(const ttsvits_handle* h, int B, int T);
/* z [B, T, inter] -> out [B, T, inter]; lengths [B] int32 (device) give y_mask.  g: NULL or the speaker embedding
 * [B, gin_channels] (the reference's g [B, gin, 1], models.py:506, 511; modules.py:185-199), as above. */
int ttsvits_flow_reverse(ttsvits_handle* h, const float* z, const int32_t* lengths, const float* g, int B, int T, float* out,
                         void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TTSDEC_H */
