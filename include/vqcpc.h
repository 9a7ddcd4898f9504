/*
 * vqcpc.h -- C ABI of libvqcpc_hip.so: the MI355X (gfx950) implementation of the
 * VQ-CPC inference hot path.
 *
 * The reference (tarepan/VectorQuantizedCPC) has no FFI for this path; its seam is the
 * nn.Module method surface.  Each entry point below names the reference interface it
 * replaces (file:line in /root/reference).  The Python drop-in classes in
 * vectorquantizedcpc_amd/{model,network_vocoder}.py bind these with ctypes
 * (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *  - Every pointer marked DEVICE is fp32 / int64 memory on the current HIP device,
 *    contiguous, borrowed for the duration of the enqueued work.  HOST pointers are read
 *    before the call returns.
 *  - All work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default
 *    stream).  No call synchronises the device except where stated: the *_create calls do
 *    (once); vqcpc_melfront_run and vqcpc_loudness_* synchronise `stream` once while they upload
 *    host-built length tables BEFORE enqueuing their kernels, and return with the work still in
 *    flight; vqcpc_vocoder_kernel_times is a measurement call and returns after its launches have
 *    finished.  vqcpc_vocoder_generate / _logits do NOT synchronise the stream: their host-built
 *    tables (lengths, decode-slot schedule, call records) go through a pinned staging arena owned by
 *    the handle, and a call waits only for the event behind the PREVIOUS call's uploads before it
 *    reuses that arena.
 *  - Return 0 on success, a negative vqcpc_status otherwise; never throws.
 *    vqcpc_last_error() returns a thread-local message for the last failure.
 *  - One handle per device; a handle is not thread-safe; distinct handles are independent.
 *  - No CPU fallback exists: without a gfx950 device every compute call fails.
 */
#ifndef VQCPC_H
#define VQCPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQCPC_ABI_VERSION 1

typedef enum {
    VQCPC_OK = 0,
    VQCPC_ERR_INVALID = -1,     /* bad argument / unsupported shape        */
    VQCPC_ERR_HIP = -2,         /* a HIP runtime call failed               */
    VQCPC_ERR_NO_DEVICE = -3,   /* no gfx950 device visible                */
    VQCPC_ERR_ALLOC = -4
} vqcpc_status;

int vqcpc_abi_version(void);
const char *vqcpc_last_error(void);

/* Number of visible HIP devices whose arch is gfx950 (0 = none; never faults). */
int vqcpc_device_count(void);

/* ------------------------------------------------------------------ Encoder ---------- */

/* Encoder.state_dict() (model.py:43-57).  DEVICE pointers, shapes as in the reference:
 * conv_weight (channels, in_channels, 4); ln_*[i] (channels) for encoder.{0,3,6,9,12};
 * fc_weight[i] (channels, channels) for encoder.{2,5,8,11}; out_weight (z_dim, channels),
 * out_bias (z_dim); codebook (n_embeddings, z_dim) = codebook.embedding;
 * rnn_* = nn.LSTM(z_dim, c_dim) parameters (4*c_dim rows, gate order i,f,g,o).
 * Supported: in_channels % 16 == 0 (<= 256), channels == 512, z_dim == 64,
 * n_embeddings % 64 == 0 (<= 4096), c_dim 64, 128, 256 (the reference's) or 512. */
typedef struct {
    const float *conv_weight;
    const float *ln_weight[5];
    const float *ln_bias[5];
    const float *fc_weight[4];
    const float *out_weight;
    const float *out_bias;
    const float *codebook;
    const float *rnn_w_ih, *rnn_w_hh, *rnn_b_ih, *rnn_b_hh;
    int in_channels, channels, n_embeddings, z_dim, c_dim;
} vqcpc_encoder_weights;

typedef struct vqcpc_encoder vqcpc_encoder;

/* Replaces Encoder.__init__ + load_state_dict (model.py:33-57; encode.py:24-30).
 * Copies and re-lays the weights on the current device (synchronises once). */
int vqcpc_encoder_create(const vqcpc_encoder_weights *w, vqcpc_encoder **out);
void vqcpc_encoder_destroy(vqcpc_encoder *enc);

/* conv_mode: which of the reference's two CPU conv back-ends to reproduce bit for bit
 * (ATen chooses by call shape; they sum in different orders):
 *   0 = as the reference would for this (B, T): 2 if B > 1 or B*in_channels*T > 20480 else 1
 *   1 = im2col + sgemm order (a single short utterance, encode.py:44-46's batch-1 call)
 *   2 = oneDNN direct-conv order (batched calls, e.g. vocoder.py:59) */
#define VQCPC_CONV_AUTO 0
#define VQCPC_CONV_IM2COL 1
#define VQCPC_CONV_DIRECT 2

/* Replaces Encoder.encode (model.py:59-70; callers encode.py:46, convert.py:76,
 * vocoder.py:59).  mel DEVICE (B, in_channels, T), T >= 2; To = (T - 2) / 2 + 1 output frames
 * (= T/2 for even T; an odd T keeps its last frame, as Conv1d(k=4, s=2, p=1) does).  Outputs
 * DEVICE: z_q (B, To, z_dim) quantised vectors; idx (B, To) int64 code indices;
 * c (B, To, c_dim) LSTM context or NULL to skip the LSTM (convert.py:76 discards it);
 * z_pre (B, To, z_dim) pre-VQ activations (the encode.py:34-40 hook) or NULL.
 * Code indices are bit-identical to the reference's PyTorch-CPU path for calls with
 * >= 16 output frames ("Bit-exactness contract"). */
int vqcpc_encoder_encode(vqcpc_encoder *enc, const float *mel, int B, int T, int conv_mode,
                         float *z_q, float *c, int64_t *idx, float *z_pre, void *stream);

/* Encoder options.  fused: which schedule of the front end runs (all produce the same bits): 1 = ONE launch, conv +
 * LayerNorms + FC stack + VQ search for 16 whole rows per workgroup with the activations resident in LDS; 2 = six
 * column-split launches (one per Linear, LayerNorm applied on load, VQ in the last) for calls too small to fill the chip
 * with whole-row workgroups; 0 = the layered kernels, one launch per module of model.py:43-55; -1 (default) = 2 for calls
 * of up to `split_max_tiles` (default 80) 16-row tiles, else 1; 0 when 4 * in_channels > 512.
 * persistent_context (default 1): the context LSTM (model.py:57) of a ONE-utterance call -- encode.py:42-46's batch 1 --
 * runs as one resident kernel with in-kernel exchanges of h_t instead of one launch per time step; 0 = always launches;
 * 2 = the resident kernel with agent-scope stores forced (its fallback when the workers do not share an XCD; for tests).
 * Same bits either way. */
int vqcpc_encoder_set_option(vqcpc_encoder *enc, const char *name, int value);

/* Activations after one stage of the front end for the same inputs as encode() -- the analogue
 * of a forward hook on the reference's modules (encode.py:34-40 hooks encoder.encoder[-1]):
 * stage 0 = conv output transposed to rows (model.py:65-67), 1 = encoder.0+1 (LN, ReLU),
 * 2+2l = encoder.{2,5,8,11}[l] (Linear), 3+2l = the LN+ReLU after it, 10 = encoder.14 (z_pre).
 * out DEVICE (B*To, channels) fp32, or (B*To, z_dim) for stage 10. */
int vqcpc_encoder_stage(vqcpc_encoder *enc, const float *mel, int B, int T, int conv_mode,
                        int stage, float *out, void *stream);

/* Replaces VQEmbeddingEMA.encode called on its own (model.py:103-115): x DEVICE (n_rows, z_dim) fp32 rows,
 * 16-byte aligned -> z_q DEVICE (n_rows, z_dim) = codebook rows, idx DEVICE (n_rows) int64, first index wins
 * ties.  The same kernel vqcpc_encoder_encode runs after the front end. */
int vqcpc_encoder_vq_encode(vqcpc_encoder *enc, const float *x, int n_rows, float *z_q, int64_t *idx,
                            void *stream);

/* Replaces the eval branch of VQEmbeddingEMA.forward as used by Encoder.forward
 * (model.py:72-86, :117-155): from encode()'s z_pre / z_q / idx (n_rows = B*T/2) computes
 * z_st = x + (q - x) (DEVICE, n_rows*z_dim, or NULL), loss = 0.25*mse (DEVICE scalar) and
 * perplexity (DEVICE scalar).  The context for forward() is the LSTM over z_st:
 * vqcpc_encoder_context. */
int vqcpc_encoder_forward_stats(vqcpc_encoder *enc, const float *z_pre, const float *z_q,
                                const int64_t *idx, int n_rows, float *z_st, float *loss,
                                float *perplexity, void *stream);

/* After the caller has synchronised the stream that carried vqcpc_encoder_encode / _context: did the resident context
 * scan of that call (persistent_context) give up on an in-kernel exchange?  VQCPC_OK, or VQCPC_ERR_HIP: the context `c`
 * of that call is incomplete, the handle now runs one launch per time step, and the call should be repeated.  Reads a
 * host-mapped word: no HIP call, no synchronisation.  The reference has no counterpart (its operators are synchronous,
 * encode.py:45-46): consumers that read results back -- driver.encode_utterances, cli.encode_dataset -- call it
 * before anything is written.  Options for tests of this path: context_debug_drop_step (one worker skips its publish at
 * that time step), context_timeout_ms. */
int vqcpc_encoder_check(vqcpc_encoder *enc);

/* nn.LSTM over z (B, Tz, z_dim) -> c (B, Tz, c_dim)  (model.py:69 / :85). */
int vqcpc_encoder_context(vqcpc_encoder *enc, const float *z, int B, int Tz, float *c,
                          void *stream);

/* ------------------------------------------------------------------ Vocoder ---------- */

/* Vocoder.state_dict(): own tables (network_vocoder.py:37-38) plus the RNN_MS core the
 * reference imports from the third-party `rnnms` package (network_vocoder.py:8, :39;
 * shapes from config.py:58-77, :198-199).  DEVICE pointers.
 * prenet: 2-layer bidirectional nn.GRU(dz+ds -> Hp), index [layer][direction];
 * ar: embedding (n_cls, de); nn.GRU(de + 2*Hp -> Hr) single layer, gate order r,z,n;
 * fc1 (Hf, Hr) + ReLU; fc2 (n_cls, Hf).
 * Supported: (dz + ds) % 32 == 0; Hp 64, 128 (the reference's), 256 or 512; Hr 512, 896 (the reference's) or 1024;
 * de % 32 == 0; Hf 256 (the reference's), 512, 768 or 1024; bits_mu_law 8 (the reference's), 9 or 10 with n_cls = 2^bits.
 * The resident decoders (`xcd`, `xcm`) exist for the reference's sizes (size_h_rnn 896 / size_h_fc 256 / 8-bit mu-law,
 * config.py:69,76-77); other sizes run on the launch-per-step kernels at about half the speed -- vqcpc_vocoder_last_path says which
 * loop a call ran. */
typedef struct {
    const float *code_embedding;      /* (n_codes, dz)           */
    const float *speaker_embedding;   /* (n_speakers, ds)        */
    const float *prenet_w_ih[2][2], *prenet_w_hh[2][2], *prenet_b_ih[2][2], *prenet_b_hh[2][2];
    const float *ar_embedding;        /* (n_cls, de)             */
    const float *ar_w_ih, *ar_w_hh, *ar_b_ih, *ar_b_hh;   /* (3Hr, de+2Hp), (3Hr, Hr), (3Hr) x2 */
    const float *fc1_weight, *fc1_bias, *fc2_weight, *fc2_bias;
    int n_codes, dz, n_speakers, ds, Hp, de, Hr, Hf, n_cls;
    int upsample_t;                   /* samples per conditioning frame (config.py:70 = hop 160) */
    int bits_mu_law;                  /* config.py:69 */
} vqcpc_vocoder_weights;

typedef struct vqcpc_vocoder vqcpc_vocoder;

/* Replaces Vocoder.__init__ + load_state_dict (network_vocoder.py:31-39; convert.py:33,45). */
int vqcpc_vocoder_create(const vqcpc_vocoder_weights *w, vqcpc_vocoder **out);
void vqcpc_vocoder_destroy(vqcpc_vocoder *voc);

/* Replaces Vocoder.generate (network_vocoder.py:69-78; callers convert.py:77,
 * vocoder.py:74-76).  idx DEVICE (B, Tc) int64 code indices; speaker DEVICE (B) int64;
 * n_codes HOST (B) per-utterance valid code counts, or NULL = all Tc (ragged batches:
 * utterance b produces 2*upsample_t*n_codes[b] samples, the rest of its row is zero).
 * Sampling protocol (project spec).  The reference draws x_t ~ Categorical(softmax(l_t)) from
 * torch's global CPU RNG, which ATen implements as an exponential race (argmax_k p_k / q_k,
 * q_k ~ Exp(1)); no device kernel can share that RNG stream, so the algorithm is kept and the
 * stream fixed: class k of sample t of utterance u (= utt_ids[b], HOST (B), or utt_base + b when
 * utt_ids is NULL -- an utterance's stream does not depend on how utterances are batched) uses
 * w = Philox4x32-10(counter = (t, u, k >> 2, 0), key = seed)[k & 3],
 * g_k = -log(-log(((w >> 9) + 0.5) * 2^-23)) (uniform exact in fp32, strictly inside (0, 1)), and x_t = first argmax_k (l_k + g_k).  Outputs DEVICE: wav (B, L) fp32 in [-1, 1] with
 * L = 2*upsample_t*Tc, mu-law decoded (preprocess.py:30-35); mulaw (B, L) int64 class
 * indices or NULL.  max_steps > 0 stops after that many samples (tests). */
int vqcpc_vocoder_generate(vqcpc_vocoder *voc, const int64_t *idx, const int64_t *speaker,
                           int B, int Tc, const int *n_codes, uint64_t seed, uint32_t utt_base,
                           const uint32_t *utt_ids, float *wav, int64_t *mulaw, int max_steps,
                           void *stream);

/* Replaces Vocoder.forward (network_vocoder.py:41-67; caller vocoder.py:62): teacher-forced
 * energies.  x DEVICE (B, Ts) int64 mu-law input samples, Ts <= 2*upsample_t*Tc;
 * logits DEVICE (B, Ts, n_cls) fp32.  Runs as a fused scan: with x given, only the GRU step is serial (one launch per
 * sample that also stores h_t); fc1 + ReLU and fc2 run as two batched GEMMs per chunk of steps. */
int vqcpc_vocoder_logits(vqcpc_vocoder *voc, const int64_t *x, const int64_t *idx,
                         const int64_t *speaker, int B, int Tc, int Ts, float *logits,
                         void *stream);

/* The wrapper's own glue alone (network_vocoder.py:73-77 / :56-66): series DEVICE (B, 2*Tc, dz + ds) = what the reference's
 * Vocoder.generate / Vocoder.forward hand to rnnms -- [:, :, :dz] the code embedding of code t/2, [:, :, dz:] the speaker
 * embedding.  The same kernel vqcpc_vocoder_generate / _logits / _condition run first; exposed so that it can be checked
 * against the fixture captured from the reference (tests/golden/vocoder_glue.npz). */
int vqcpc_vocoder_glue(vqcpc_vocoder *voc, const int64_t *idx, const int64_t *speaker, int B, int Tc, float *series,
                       void *stream);

/* Conditioning series after the prenet, (B, 2*Tc, 2*Hp) DEVICE -- stage-level tests. */
int vqcpc_vocoder_condition(vqcpc_vocoder *voc, const int64_t *idx, const int64_t *speaker,
                            int B, int Tc, float *cond, void *stream);

/* Decode-loop options.
 * fuse_fc2 (default 1): in calls of up to 4 utterance tiles (64 decode slots per tile group) the fc2 + draw of sample t-1
 * and the GRU step of sample t share ONE launch -- W_hh h does not depend on the drawn sample, so it runs while the fc2
 * workgroups of the same launch produce the candidates, which the GRU's gate waves then pick up through 8-byte granules.
 * Two launches per sample instead of three; same bits.  A wait that ever times out (0.25 s) sets the handle's status word
 * (vqcpc_vocoder_check).
 * xcd (default -1 = auto: up to xcm_min utterances in flight; 0 never, also turns xcm off; 1 whenever the dimensions are
 * the reference's): generate() runs as EIGHT resident,
 * weight-stationary decoders, one per XCD (ar_xcd.hip): decode slot s lives on XCD s % 8; each XCD keeps a full copy of the
 * recurrent weights on its 32 CUs (W_hh in VGPRs, fc1 / fc2 / the sample-embedding table in LDS) and exchanges h_t, a_t and
 * the draw candidates through its own L2; no launches per sample.  Same samples as every other path.  xcd_slots: decode
 * slots it may use (default and maximum 32); more utterances than slots run back to back in them (longest first).
 * xcm (default -1 = auto: more than xcm_min (68) and fewer than xcm_max (512) utterances in flight; 0 never; 1 whenever the
 * dimensions are the reference's): the same resident decoders with 16 decode slots per XCD on the matrix cores
 * (ar_xcm.hip): [W_hh; W_fc1] h_t as six v_mfma_f32_16x16x4_f32 tiles per workgroup, A fragments pinned in registers.
 * xcm_slots: decode slots it may use (default and maximum 128).  Same samples as every other path; shares xcd's timeout,
 * debug-drop and agent-store options and its status word.
 * The resident decoders (xcd, xcm) assume the GPU is theirs for the call: ONE launch of 256 workgroups (768 threads, ~150 KB of
 * LDS each: one per CU) that must all be resident, 32 on each XCD, within xcd_timeout_ms -- checked in-kernel, not assumed.
 * On a GPU shared with another process or stream the placement can miss: the launch then writes nothing and reports it
 * (vqcpc_vocoder_check: "not dealt 32"), the handle keeps its options and the caller repeats the call; only a second miss in a
 * row switches the resident decoders off for the handle.  A hand-off TIMEOUT inside a launch switches them off at once; the
 * handle re-arms itself after 16 clean calls, or when xcd / fuse_fc2 is set again.
 * xcd_agent_stores / xcd_timeout_ms / xcd_debug_drop_step, handoff_timeout_ms / handoff_debug_drop_step: A-B and tests
 * of the abort path (one worker skips a publish at that step; the waits give up after the timeout; vqcpc_vocoder_check).
 * xcd_debug_misplace (one shot): workgroup 0 of the next resident launch reports the XCD next to its own, so that the placement
 * check fails (tests of the "not dealt 32" path).
 * tf_chunk_replays: graph replays per chunk of the teacher-forced scan (vqcpc_vocoder_logits; default 4).
 * use_graph: replay the per-sample kernels from a captured hipGraph
 * (default 1) instead of launching them one by one.  steps_per_graph: samples per replay (even).
 * slots: continuous batching -- decode with this many slots, each running utterances back to back
 * (0 = one slot per utterance).  big_min_tiles: utterance tiles (of 16) from which the LDS-staged
 * large-batch GRU kernel is used (default 5, 0 = never).  two_groups: run calls of 3..big_min_tiles-1
 * tiles, or of >= 2*big_min_tiles tiles, as two independent tile groups on two streams (default 1). */
int vqcpc_vocoder_set_option(vqcpc_vocoder *voc, const char *name, int value);

/* After the caller has synchronised the stream that carried vqcpc_vocoder_generate: did an in-kernel hand-off of a call
 * since the last check (per-XCD decoders, fused fc2 || GRU launch) time out, or were the per-XCD decoders' workgroups not
 * dealt 32 to each XCD?  VQCPC_OK, or VQCPC_ERR_HIP: the waveform of that call is incomplete (or was not written at all) and
 * the call should be repeated -- the repeat gives the same samples the fast path would have (the sampling stream does not
 * depend on the path); the message names the call (the resident decoders tag the status word with the handle's call count)
 * and says whether the handle has switched its in-kernel hand-offs off (see the options).  Reads a host-mapped word: no HIP
 * call.  Without the synchronisation the word of a call still running reads zero: the next vqcpc_vocoder_generate on the
 * handle looks at it again (and reports what it finds) but clears nothing.  convert.py:75-83 writes the wav right after
 * generate(): the Python Vocoder.generate checks (and repeats once) by default; driver.convert_utterances, cli.convert and
 * shard.convert_sharded go through it before anything is written or gathered. */
int vqcpc_vocoder_check(vqcpc_vocoder *voc);

/* Which decode loop the last generate()/logits() call on the handle ran (measurement and tests; the samples do not depend
 * on it): 0 = launch-per-step kernels (more than 511 utterances in flight, dimensions other than the reference's 896 / 256 /
 * 8 bits, teacher-forced calls, the fallback), 2 = the per-XCD resident decoders, 3 = their matrix-core form (16 slots per XCD);
 * 1 = no decode loop has run yet; -1 = null handle. */
int vqcpc_vocoder_last_path(vqcpc_vocoder *voc);

/* Which decode loop a call of B utterances producing n_samples[b] samples each would take under these options (the arguments
 * mirror vqcpc_vocoder_set_option: xcd, xcm -1 / 0 / 1; xcm_min, xcm_max, xcd_slots, xcm_slots <= 0 = the defaults 68, 512, 32,
 * 128; slots 0 = one per utterance), for the reference's dimensions: *path as vqcpc_vocoder_last_path, *slots_used, and *longest =
 * the longest back-to-back schedule of a resident decode slot in samples.  Pure host arithmetic (no GPU): the rule
 * vqcpc_vocoder_generate applies.  A slot's schedule must stay below 2^24 - 1 samples for the resident decoders; in auto mode
 * such a call takes the launch path, with xcd / xcm = 1 it is VQCPC_ERR_INVALID. */
int vqcpc_vocoder_plan(int xcd, int xcm, int xcm_min, int xcm_max, int xcd_slots, int xcm_slots, int slots,
                       const int *n_samples, int B, int *path, int *slots_used, int64_t *longest);

/* Decode slots the last call's loop ran through (per-XCD decoders: at most 32; matrix-core form: at most 128; launch path:
 * the `slots` option or one per utterance); -1 = null handle. */
int vqcpc_vocoder_last_slots(vqcpc_vocoder *voc);

/* Device memory the handle's grow-only work buffers hold at the moment (conditioning rows, schedules, exchange areas, ...:
 * the peak of the calls so far; the weights are not counted). */
int vqcpc_vocoder_workspace_bytes(vqcpc_vocoder *voc, uint64_t *bytes);

/* Device time, in milliseconds, of the whole decode loop of the last generate()/logits() call
 * (HIP events on the launch stream) and the number of samples per utterance it covers.
 * Valid after the stream has been synchronised. */
int vqcpc_vocoder_last_timing(vqcpc_vocoder *voc, float *loop_ms, int *n_steps);

/* ------------------------------------------------------------------ Mel front-end ---- */

/* Replaces wave_to_mel (preprocess.py:53-75; convert.py:54-70 is the same sequence): peak-normalise
 * to 0.999, pre-emphasis (preprocess.py:16-17), centred STFT magnitude (periodic Hann `win`
 * zero-padded to n_fft, reflect padding: librosa ^0.8), Slaney mel filterbank from fmin to sr/2,
 * amplitude_to_db with top_db against the utterance maximum, / top_db + 1.  Defaults of the
 * reference: config.py:103-112 (16000, 2048, 80, 160, 400, 50, 0.97, 80). */
typedef struct vqcpc_melfront vqcpc_melfront;
int vqcpc_melfront_create(int sr, int n_fft, int n_mels, int hop, int win, float fmin, float preemph,
                          float top_db, vqcpc_melfront **out);
void vqcpc_melfront_destroy(vqcpc_melfront *f);
/* Frames of an n_samples utterance: 1 + n_samples / hop (centred STFT). */
int vqcpc_melfront_frames(const vqcpc_melfront *f, int n_samples);
/* wav DEVICE (B, Lmax) fp32, lens HOST (B) valid samples; mel DEVICE (B, n_mels, 1 + Lmax/hop) --
 * the encoder's input layout; frames past an utterance's own length are zero. */
int vqcpc_melfront_run(vqcpc_melfront *f, const float *wav, const int *lens, int B, int Lmax, float *mel,
                       void *stream);

/* ------------------------------------------------------------------ Resampling ---- */

/* Replaces the resampling inside librosa.load(path, sr=cfg.preprocessing.sr) (convert.py:54-56): librosa ^0.8's
 * res_type "kaiser_best" = resampy's band-limited sinc interpolation (64 zero crossings, 512 table entries per crossing,
 * Kaiser taper, linear interpolation between entries), fp64 arithmetic, fp32 result.  resampy is absent offline: parity
 * unpinned (oracle/resample_ref.py restates the algorithm and its published constants). */
typedef struct vqcpc_resampler vqcpc_resampler;
int vqcpc_resampler_create(int sr_in, int sr_out, vqcpc_resampler **out);
void vqcpc_resampler_destroy(vqcpc_resampler *r);
/* Output samples of an n_in-sample signal: ceil(n_in * sr_out / sr_in), as librosa.resample(fix=True). */
int vqcpc_resampler_out_len(const vqcpc_resampler *r, int n_in);
/* wav_in DEVICE (B, Lin_max) fp32, lens_in HOST (B) valid samples; wav_out DEVICE (B, Lout_max) fp32 with
 * Lout_max >= vqcpc_resampler_out_len(max lens_in): row b holds its utterance's resampled samples, zeros behind them.
 * Does not synchronise (the lengths travel as kernel arguments), except when a call needs a longer output-clock table than
 * any before it on this handle. */
int vqcpc_resampler_run(vqcpc_resampler *r, const float *wav_in, const int *lens_in, int B, int Lin_max,
                        float *wav_out, int Lout_max, void *stream);

/* ------------------------------------------------------------------ Loudness ---- */

/* Replaces pyloudnorm.Meter(sr) (convert.py:50) for mono audio: ITU-R BS.1770-4 gated integrated
 * loudness as pyloudnorm ^0.1.0 computes it (K-weighting biquads derived at `rate`, 400 ms blocks with
 * 75 % overlap, gates at -70 LUFS absolute and -10 LU relative).  fp64 on the device. */
typedef struct vqcpc_loudness vqcpc_loudness;
int vqcpc_loudness_create(int rate, vqcpc_loudness **out);
void vqcpc_loudness_destroy(vqcpc_loudness *m);
/* Gating blocks of an n_samples signal; 0 when it is shorter than one block (the meter refuses it). */
int vqcpc_loudness_blocks(const vqcpc_loudness *m, int n_samples);
/* meter.integrated_loudness (convert.py:57, :79) of B padded mono signals.  wav DEVICE (B, Lmax) fp32,
 * lens HOST (B); lufs DEVICE (B) fp64 (-inf when every block is gated out); block_energy DEVICE
 * (sum of vqcpc_loudness_blocks(lens[b])) fp64 or NULL.  VQCPC_ERR_INVALID when an utterance is shorter
 * than one block (pyloudnorm raises ValueError).  Synchronises `stream` once (host length tables). */
int vqcpc_loudness_integrated(vqcpc_loudness *m, const float *wav, const int *lens, int B, int Lmax, double *lufs,
                              double *block_energy, void *stream);
/* pyloudnorm.normalize.loudness (convert.py:80), in place: wav[b] *= 10^((target[b] - measured[b]) / 20),
 * product in fp64 rounded to fp32.  measured / target DEVICE (B) fp64. */
int vqcpc_loudness_normalize(vqcpc_loudness *m, float *wav, const int *lens, int B, int Lmax, const double *measured,
                             const double *target, void *stream);

/* Average wall time, in microseconds, of `reps` back-to-back launches of each per-sample kernel
 * of the decode loop on the state the last generate()/logits() call left (HIP events on
 * `stream`; synchronises it).  out_us[5] = {GRU step, fc1, fc2 + draw, decode slots one launch
 * covers (a call of 33..80 or >= 192 utterances runs as two independent tile groups), which GRU-step
 * kernel that is: 0 = one tile, 1 = two tiles per workgroup, 2 = LDS-staged large-batch kernel; 4 / 5 = the fused launch
 * (fc2 + draw of the previous sample in front of the GRU step) on the small / the large-batch kernel -- out_us[2] is then
 * fc2 as a launch of its own, for reference}.
 * Each time includes this chip's ~1.5 us dependent-launch boundary. */
int vqcpc_vocoder_kernel_times(vqcpc_vocoder *voc, int reps, float *out_us, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VQCPC_H */
