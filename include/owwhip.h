/*
 * owwhip.h -- C ABI of libowwhip.so: the MI355X (gfx950) streaming wake-word path.
 *
 * This is the drop-in boundary for the ONE hot path of dscripka/openWakeWord:
 *   int16 PCM 16 kHz -> log-mel -> speech-embedding CNN -> wake-word heads -> post-processing.
 * The reference crosses this boundary through three onnxruntime / LiteRT closures
 *   melspec_model_predict        openwakeword/utils.py:87  (122-136 for tflite)
 *   embedding_model_predict      openwakeword/utils.py:93  (147-161)
 *   model_prediction_function[]  openwakeword/model.py:137-138, 158-159 (116-119, 174-175)
 * and keeps all per-stream state in Python objects (utils.py:163-170, model.py:198).  Here the state
 * of S concurrent streams lives in HBM and one call advances every stream by one 80 ms step.
 *
 * Conventions: plain C, opaque handle, every function returns 0 on success and a negative OWW_E*
 * code on failure (message via oww_last_error()); nothing throws across the ABI.  One handle = one
 * GPU; calls on one handle must be serialised by the caller (the reference object is single-threaded
 * too).  "dev" pointers are HIP device pointers valid on the handle's device; "host" pointers are
 * ordinary host memory.  All float data is IEEE fp32, row-major, C order.
 */
#ifndef OWWHIP_H
#define OWWHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OWW_ABI_VERSION 6

#define OWW_OK            0
#define OWW_EINVAL       -1   /* bad argument */
#define OWW_EHIP         -2   /* HIP runtime error */
#define OWW_ESTATE       -3   /* call out of order (e.g. step before weights committed) */
#define OWW_ENOMEM       -4
#define OWW_ERANGE       -5   /* fp16-split kernels (use_mfma = 3): an activation left the f16 range; sticky, see oww_range_status */

#define OWW_CHUNK        1280 /* samples per 80 ms step          (utils.py:417-434) */
#define OWW_MEL_BINS       32 /* melspectrogram.onnx output bins (utils.py:271)     */
#define OWW_MEL_PER_CHUNK   8 /* new mel rows per step                 */
#define OWW_WINDOW         76 /* mel rows per embedding window   (utils.py:225,440) */
#define OWW_EMB_DIM        96 /* embedding_model.onnx output     (utils.py:323)     */
#define OWW_SCORE_RING     30 /* prediction_buffer maxlen        (model.py:198)     */
#define OWW_MAX_HEADS      16
#define OWW_MAX_HEAD_BLOCKS 8    /* hidden blocks of one head network (train.py:73); the released models have 1 */
#define OWW_MAX_LABELS     32
#define OWW_MAX_CALL_CHUNKS 4096 /* longest oww_step call: 4096 x 80 ms = 5.5 minutes of audio per stream */

typedef struct oww_ctx oww_ctx;

typedef struct oww_config {
    int32_t device;        /* HIP device ordinal */
    int32_t n_streams;     /* S: concurrent audio streams owned by this handle */
    int32_t max_chunks;    /* largest n_chunks a single oww_step may carry (>=1) */
    int32_t feature_ring;  /* rows of the per-stream feature ring; 0 = max head T (>=16) */
    int32_t use_mfma;      /* kernel family of the embedding CNN and the heads:
                              3 = register-resident kernels, fp32 products as three f16 MFMAs (hi/lo split, fp32 accumulate;
                                  default path, agrees with the fp32 families to fp32 round-off),
                              1 = register-resident kernels on the exact fp32 MFMA,
                              2 = LDS-tiled fp32-MFMA kernels, 0 = plain-VALU kernels of the LDS-tiled dataflow */
    int32_t debug_layers;  /* 1 = keep per-layer CNN outputs of the last step for oww_debug_read */
    void*   stream;        /* hipStream_t to launch on; NULL = a stream owned by the handle */
} oww_config;

/* ---- lifetime -------------------------------------------------------------------------------- */
int  oww_abi_version(void);
/* "src=<16 hex digits> arch=gfx950": the hash of the sources this binary was built from (openwakeword_amd/_build.py: source_hash).
 * Profiles committed under profiles/ carry the same hash, so a measurement can be tied to the kernels it was taken on. */
const char* oww_build_info(void);
const char* oww_last_error(void);
int  oww_create(const oww_config* cfg, oww_ctx** out);
int  oww_destroy(oww_ctx* h);

/* ---- weights (replace the .onnx/.tflite files of openwakeword/__init__.py:8-51) ----------------
 * Host blobs, copied.  Layouts:
 *  mel   : float hann[400]; int32 start[32]; float taps[32][16]          (window + sparse mel filterbank)
 *  embed : for each of the 20 conv layers in graph order: float w[kh][kw][cin][cout] (Keras HWIO),
 *          then for layers 0..18: float scale[cout]; float shift[cout]   (inference BatchNorm folded)
 *  head  : int32 hdr[8] = {kind(0 binary,1 gated,2 multiclass), T, hidden, n_out, has_layernorm, extra_blocks,0,0}
 *          then per net (1 net; 2 for gated): w1[T*96][hidden] b1[hidden] (ln1_g ln1_b [hidden] if
 *          has_layernorm), then 1 + extra_blocks times { w2[hidden][hidden] b2 (ln2_g ln2_b) }, then w3[hidden][n_out] b3[n_out].
 *          extra_blocks = 0 is the network of every released model (train.py:27,67: n_blocks = 1); train.py's Net takes any
 *          n_blocks (train.py:73): extra_blocks = n_blocks - 1, from -1 (no hidden block) to OWW_MAX_HEAD_BLOCKS - 1.  Nets with
 *          other than one block are evaluated by the generic heads kernel (any kernel family), not by the MFMA head kernels.
 *          In the default family (use_mfma = 3) one-block nets of up to 64 hidden units and one sigmoid output run on the f16-split
 *          heads kernel, and ungated one-block nets of up to 128 hidden units / 8 outputs (the multiclass `timer` model,
 *          docs/models/timers.md:9-27) on its wide form; everything else on the generic kernel.
 *  head, recurrent (train.py:85-98, model_type "rnn"): hdr = {3, T (<= 64), 64, n_out, 0, 0, 0, 0}; then for LSTM layer 0 (input 96) and 1
 *          (input 128), for direction forward and reverse: w[in + 64][256] (rows x ; h, columns i | f | g | o -- torch's gate order),
 *          b[256] (= b_ih + b_hh); then w_out[128][n_out], b_out[n_out].  Output = Sigmoid (n_out = 1) or ReLU + softmax of the
 *          linear layer on the last time step (out[:, -1]).
 */
int  oww_load_mel(oww_ctx* h, const void* blob, size_t nbytes);
int  oww_load_embedding(oww_ctx* h, const void* blob, size_t nbytes);
int  oww_add_head(oww_ctx* h, const void* blob, size_t nbytes);     /* returns head index >= 0 */
/* oww_commit: pack + upload, allocate state, derive reset state, then reset all.  For the default kernel family (use_mfma = 3) it also
 * (1) calibrates: a scratch handle of the exact-fp32 family runs the all-ones mel history and 32 probe streams x 16 frames (silence,
 *     noise RMS 30..32767, full-scale square waves) on the same weights; from every layer's largest |activation| each layer gets a
 *     power-of-two scale under which its f16-split operands keep full precision, and BatchNorm is folded into the packed weights;
 * (2) self-tests: the probes are replayed on the f16-split kernels and embeddings / raw head outputs compared with the fp32 run.
 * Weights the split cannot carry within the north-star tolerance (1e-3) -- the reference's fp32 graphs have no such limit,
 * utils.py:84-93 -- are REFUSED here with OWW_ERANGE (message names the deviation; use use_mfma = 1 for them) instead of scoring
 * differently later.  Adds ~0.05-0.3 s to the call.  (OWW_NO_COMMIT_SELFTEST=1 in the environment skips step (2): development only.) */
int  oww_commit(oww_ctx* h);
/* Optional, before oww_commit, use_mfma = 3 only (ignored by the fp32 families): calibration audio of the caller's domain -- int16
 * [n_streams][n_frames * 1280], host pointer, copied.  oww_commit derives the fp16-split kernels' scale ladder from, and runs its
 * f16x3-vs-fp32 self-test on, the built-in synthetic probes (silence, noise at five levels, square waves) PLUS this audio, cut into
 * 16-frame segments (at most 224 of them are used).  The reference's fp32 graphs have no range to calibrate (utils.py:84-93); the
 * Python layer passes speech (the reference's three fixture clips at several gains) by default.  pcm = NULL clears the set. */
int  oww_set_calibration(oww_ctx* h, const int16_t* pcm, int32_t n_streams, int32_t n_frames);
/* What oww_commit measured and decided (use_mfma = 3; any pointer may be NULL): absmax[l] = largest |activation| of CNN layer l over
 * the probe set on the exact-fp32 kernels; exps[l] = the layer's power-of-two scale exponent, exps[20] = the exponent the heads' GEMM
 * applies to the features; n_probe_streams = 32 built-in + the caller's 16-frame segments; selftest = {max |embedding - fp32|,
 * max |embedding|, max |raw score - fp32|} of the fp16-split replay of the probes. */
int  oww_calibration_info(oww_ctx* h, float absmax[20], int32_t exps[21], int32_t* n_probe_streams, float selftest[3]);
int  oww_n_labels(const oww_ctx* h);  /* total score columns = sum of n_out over heads */

/* ---- per-stream state (AudioFeatures.reset utils.py:172-178 + Model.reset model.py:226-230) -----
 * stream_ids == NULL resets every stream.  init_features (host, [feature_ring][96], oldest row first)
 * seeds the feature ring the way the reference seeds it with embeddings of random audio
 * (utils.py:169); NULL = zeros. */
int  oww_reset(oww_ctx* h, const int32_t* stream_ids, int32_t n, const float* init_features);

/* ---- post-processing options of Model.predict (model.py:340-359); arrays of oww_n_labels() -------
 * patience[i] <= 0 and debounce_frames <= 0 disable the respective rule for label i. */
int  oww_set_postproc(oww_ctx* h, const int32_t* patience, const float* threshold, int32_t debounce_frames);

/* ---- custom verifier models (model.py:320-328; trained by custom_verifier_model.py:95-113) on the device ----------------------
 * The reference re-scores a label whose base score reaches `threshold` (custom_verifier_threshold, default 0.1) with a pickled
 * scikit-learn pipeline: flatten -> StandardScaler -> LogisticRegression over the last T feature rows of the label's model
 * (T x 96 values, oldest row first).  Scaler and regression fold into one affine map, which is what this entry takes:
 * w[T*96] = coef / scale, bias = intercept - sum(coef * mean / scale); the label then reads sigmoid(w . features + bias)
 * (= predict_proba(...)[0][-1]) whenever its head output is >= threshold, before the post-processing rules.  w == NULL removes
 * the label's verifier.  Host pointers; call after oww_commit. */
int  oww_set_verifier(oww_ctx* h, int32_t label, const float* w, int32_t n_w, float bias, float threshold);

/* ---- VAD gate of Model.predict (model.py:366-381) for the batched path -------------------------------------------------
 * Two ways to feed the gate.  (1) External network (the reference's silero_vad.onnx is a release asset whose graph is not in the
 * reference checkout, so it cannot be built into this library): the caller supplies one VAD score per stream and step -- the mean
 * over the step's 640-sample sub-frames that VAD.__call__ appends (vad.py:98-130) -- with oww_push_vad.  (2) A network on the device:
 * oww_load_vad below (a structural stand-in with the same interface and state).  Either way the library keeps the per-stream
 * score ring and applies the gate: when the largest
 * of the scores pushed 5..7 steps ago (ring[-7:-4]; nothing during the first four steps) is below `threshold`, every label of
 * that stream reads 0 for the step (the 30-deep score ring keeps the ungated value, as in the reference).  threshold <= 0
 * switches the gate off (default).  Call oww_push_vad BEFORE the oww_step / oww_submit of the same frame; vad_scores is
 * fp32 [S], host or device.  Like Model.reset(), oww_reset leaves the VAD ring alone. */
int  oww_set_vad_threshold(oww_ctx* h, float threshold);
int  oww_push_vad(oww_ctx* h, const float* vad_scores, int on_device);
/* Voice-activity NETWORK on the device (BASELINE configs[4]: "VAD gate fused into the per-frame HIP pipeline").  Silero's graph
 * is unavailable (see above), so this loads a structural STAND-IN with the interface and state the reference fixes around the
 * network (vad.py:98-130: 640-sample sub-frames / 32767, recurrent state h, c [2, 1, 64] carried from call to call, one score
 * per sub-frame, the mean per predict() call appended to the ring): |STFT| (256-sample Hann frames, hop 64, bins 1..128) ->
 * log(1 + gain |X|) -> 4 x [Conv1d(k=3, pad 1) + ReLU] 128->16 (stride 1), 16->32 (2), 32->32 (2), 32->64 (1) -> 2-layer LSTM(64)
 * -> ReLU -> Linear(64->1) -> sigmoid, mean over time.  Blob (call before oww_commit): int32 hdr[8] = {1, 256, 64, 128, 64, 0, 0, 0};
 * float gain; float hann[256]; per conv: w[3][cin][cout], b[cout]; per LSTM layer: w[128][256] (rows x ; h, columns i | f | g | o),
 * b[256] (= b_ih + b_hh); float wd[64]; float bd.  Once loaded, every oww_step / oww_submit (n_chunks must be 1) runs the network
 * on the frame's two sub-frames of every stream and pushes the mean score into the stream's ring; oww_push_vad is then refused.
 * oww_get_vad copies the scores of the last step (host fp32 [S]); oww_reset_vad zeroes recurrent state and score history of the
 * listed streams (NULL = all) -- oww_reset does not, exactly like Model.reset() leaves Model.vad alone (model.py:226-230). */
int  oww_load_vad(oww_ctx* h, const void* blob, size_t nbytes);
int  oww_get_vad(oww_ctx* h, float* out);
int  oww_reset_vad(oww_ctx* h, const int32_t* stream_ids, int32_t n);

/* ---- the hot loop: one Model.predict per stream (model.py:232-386) --------------------------------
 * pcm: int16 [S][n_chunks*1280], stream-major.  scores: fp32 [S][n_labels] or NULL.
 * *_on_device: 0 = host pointer (copied on the handle's stream), 1 = device pointer.
 * n_chunks > 1 reproduces the reference's multi-chunk call: one mel pass over the whole span
 * (single top_db clamp), one embedding + head evaluation per chunk, max over chunks (model.py:287-298).
 * n_chunks may exceed oww_config.max_chunks (up to OWW_MAX_CALL_CHUNKS): the call is then evaluated in slices of max_chunks chunks
 * behind one extra pass of the mel kernel that finds the CALL's maximum, so that the clamp floor is still the reference's single
 * "max - 80 dB" over the whole call (utils.py:387-401: one run of melspectrogram.onnx per call); such a call is synchronous when pcm
 * is a host pointer, and is refused with the on-device VAD network (one chunk per step).
 * Asynchronous on the handle's stream unless scores is a host pointer.
 * A DEVICE pcm pointer should be 16-byte aligned: the fused front end loads eight samples at a time; any other alignment (a slice of
 * a larger int16 buffer) is accepted and takes the separate mel launch with scalar sample loads -- same scores to fp32 round-off
 * (1e-6), about 10 % slower. */
int  oww_step(oww_ctx* h, const int16_t* pcm, int pcm_on_device, int32_t n_chunks,
              float* scores, int scores_on_device);
int  oww_sync(oww_ctx* h);
/* The same one-chunk step for the streams a serving edge actually has a full 80 ms chunk for (clients of
 * examples/web/streaming_server.py:49-66 arrive, stall and leave independently; in the reference each of them simply does not
 * call predict() while it has no audio).  stream_on: uint8 [S], 1 = the stream takes part: exactly oww_step for it; 0 = the
 * stream sits the step out: none of its state moves (sample tail, conv histories, feature and score rings, frame counters,
 * VAD state) and its row of `scores` repeats its previous step's values; its 1280 PCM samples are not read.  A stream
 * stepped k times through any interleaving of masked steps is in the state k plain steps leave it in, bit for bit.
 * Cost: with a HOST mask of which at most half the streams take part, only the stage groups (1 / 2 / 4 / 8 streams of a wave) that
 * hold a participating stream are launched and the heads run on the participants alone, so the step costs roughly in proportion to
 * the participation (streams sharing a group with a participant are computed and not stored); a device-resident mask, or more than
 * half of the streams, runs the full launches.  Both register-resident kernel families take masked steps (use_mfma = 3 and the exact
 * fp32 family 1, so weights the fp16 split refuses at commit can still be served; heads of any form); the proportional cost is the
 * default family's (fused front end, heads of the [T,96] -> 64 -> 64 -> 1 form) -- every other combination runs full launches whose
 * stores skip the streams that sit out.  The LDS-tiled families (use_mfma = 2, 0) return OWW_EINVAL. */
int  oww_step_masked(oww_ctx* h, const int16_t* pcm, int pcm_on_device, const uint8_t* stream_on, int stream_on_on_device,
                     float* scores, int scores_on_device);
/* Range guard of the default kernel family (use_mfma = 3 evaluates every fp32 product as three f16 MFMAs on hi/lo-split
 * operands, so an activation with |x| >= 65520 cannot be represented).  The reference's fp32 graphs have no such limit
 * (onnxruntime CPU kernels, utils.py:84-93), so instead of scoring silently differently the kernels test one accumulator
 * per position tile and layer for the NaN such an operand produces and raise a sticky per-handle flag.  Once raised,
 * oww_step / oww_submit / oww_collect / oww_sync / oww_embed* return OWW_ERANGE (a step issued with device-resident
 * scores is asynchronous: the error surfaces on the next call that looks).  oww_range_status waits for the handle's
 * stream and returns OWW_OK or OWW_ERANGE; clear != 0 lowers the flag (after e.g. resetting the offending streams).
 * The exact-fp32 family (use_mfma = 1) never raises it. */
int  oww_range_status(oww_ctx* h, int clear);
/* Which streams: after OWW_ERANGE, first_stream / n_streams name the contiguous streams of (one of) the wave(s) that saw the
 * out-of-range value -- the offender is among them -- so that a server can oww_reset just those and clear the flag instead of
 * restarting every stream; (-1, 0) when the flag is down or the position is not known (a step over a participant list). */
int  oww_range_where(oww_ctx* h, int32_t* first_stream, int32_t* n_streams);

/* ---- host-fed pipeline: the same step with PCM arriving in host memory every 80 ms (the serving edge of
 *      examples/web/streaming_server.py:32-70 and detect_from_microphone.py: audio is produced on the host) ----------
 * oww_submit enqueues  H2D(pcm) -> step -> D2H(scores)  and returns at once: the upload runs on its own stream, so the
 * PCM of step t+1 crosses PCIe while the kernels of step t execute; oww_collect blocks until the OLDEST submitted step
 * has delivered and copies its fp32 [S][n_labels] scores to `scores` (host).  At most two steps may be in flight
 * (submit, submit, collect, submit, collect, ...); a third submit returns OWW_ESTATE.  `pcm` must stay valid and
 * unmodified until the matching oww_collect returns; use page-locked memory (oww_host_alloc, hipHostMalloc,
 * torch pin_memory) -- pageable memory works but makes the upload synchronous.  Results are identical to oww_step. */
int  oww_submit(oww_ctx* h, const int16_t* pcm, int32_t n_chunks);
/* oww_submit for the streams with stream_on[s] != 0 only (one chunk; see oww_step_masked): the serving edge's pipelined form.
 * stream_on: host uint8 [S], read before the call returns. */
int  oww_submit_masked(oww_ctx* h, const int16_t* pcm, const uint8_t* stream_on);
int  oww_collect(oww_ctx* h, float* scores);
int  oww_host_alloc(void** out, size_t nbytes);   /* page-locked host memory for PCM / score buffers */
int  oww_host_free(void* p);
const float* oww_scores_dev(const oww_ctx* h);     /* device [S][n_labels], valid until the next step */
/* raw head outputs of the last step, BEFORE post-processing (what model_prediction_function returned,
 * model.py:313-317; max over chunks for n_chunks > 1): host fp32 [S][n_labels].  Blocking.  Lets a host
 * shim run the reference's own post-processing (Model.predict with short / unaligned calls). */
int  oww_get_raw(oww_ctx* h, float* out);
/* Rate conversion of every stream's message on the device, the batched form of the serving example's per-message
 * resampy.resample(data, sample_rate, 16000) (examples/web/streaming_server.py:57-58): a polyphase FIR,
 *   out[s][j] = sat_int16(rint(sum_k taps[(j p) % q][k] * in[s][(j p) / q + k - n_taps / 2 + 1])),   samples outside the message = 0,
 * with p / q = input rate / 16000 in lowest terms, taps float [q][n_taps] on the host (n_taps even; the Kaiser-windowed sinc bank of
 * openwakeword_amd/resample.py::design, or any other) and n_out = n_in * q / p.  in: int16 [S][n_in], out: int16 [S][n_out], host or
 * device.  Stateless per call, like the example; asynchronous when out is a device pointer. */
int  oww_resample(oww_ctx* h, const int16_t* in, int in_on_device, int32_t n_in, int32_t p, int32_t q, const float* taps, int32_t n_taps,
                  int16_t* out, int out_on_device, int32_t n_out);

/* ---- multi-GPU: delivery of the scores to one rank over RCCL (xGMI), for binders without torch.distributed ----------------------
 * Streams are independent, so N GPUs = N handles in N processes, each owning a contiguous range of the global streams (the
 * reference's only scale-out, utils.py:502-536 bulk_predict, splits FILES over processes the same way); nothing in the data path
 * crosses GPUs.  The one exchange is handing the per-stream scores to whoever consumes them:
 *   oww_comm_id       rank 0: fills id[OWW_COMM_ID_BYTES] (ncclGetUniqueId); the binder ships these bytes to the other ranks by
 *                     whatever side channel it has (a file, a socket, an environment variable, MPI)
 *   oww_comm_init     every rank, collectively: joins the communicator (ncclCommInitRank) on the handle's device
 *   oww_gather_scores enqueued on the handle's stream after a step: every rank sends its fp32 [S_r][n_labels] scores to rank 0,
 *                     which receives them at out + sum(counts[0..r)) * n_labels (device pointer; counts[r] = streams of rank r, the
 *                     same array on every rank).  One grouped ncclSend / ncclRecv exchange; asynchronous (oww_sync to wait).
 *   oww_comm_count    *ranks = what the communicator itself reports (ncclCommCount): the number of RCCL ranks actually joined
 *   oww_comm_destroy  leaves the communicator (oww_destroy does it too)
 * librccl.so is bound at run time by the first of these calls; a process that already holds a copy (e.g. torch's) shares it.
 * Python: openwakeword_amd.shard.ScoreGather is the torch.distributed form of the same exchange. */
#define OWW_COMM_ID_BYTES 128
int  oww_comm_id(void* id);
int  oww_comm_init(oww_ctx* h, const void* id, int32_t rank, int32_t world);
int  oww_gather_scores(oww_ctx* h, float* out, const int32_t* counts);
int  oww_comm_count(oww_ctx* h, int32_t* ranks);
int  oww_comm_destroy(oww_ctx* h);

/* ---- stage-level entry points (the reference's per-stage closures; used for parity tests and for
 *      AudioFeatures._get_melspectrogram / embed_clips style callers).  Host pointers. ---------------
 * oww_mel: int16 [B][n] -> dB values [B][F][32], F=(n-512)/160+1, clamp floor shared over the call
 *          exactly like melspec_model_predict (utils.py:87,202).  Requires B <= n_streams.
 * oww_embed: mel rows [B][rows][32] (rows = 76 + 8*(n_out-1)) -> embeddings [B][n_out][96]; the
 *          windows are rows[8i : 8i+76] (utils.py:229-236).  Runs the same incremental kernels; the
 *          streaming state of the streams it borrows is parked and put back, so live streams are unaffected.
 * oww_head: features [B][T][96] -> raw outputs [B][n_out] of head `head` (model.py:137-138). */
int  oww_mel(oww_ctx* h, const int16_t* pcm, int32_t B, int32_t n, float* out_db);
/* the same with one clamp floor PER CLIP: what the reference's CPU path computes when it maps _get_melspectrogram over
 * the clips of a batch (utils.py:243-290, embed_clips / feature extraction for training, utils.py:542-601) */
int  oww_mel_clips(oww_ctx* h, const int16_t* pcm, int32_t B, int32_t n, float* out_db);
int  oww_embed(oww_ctx* h, const float* mel_rows, int32_t B, int32_t rows, float* out);
/* AudioFeatures.embed_clips (utils.py:354-385; the feature extractor behind compute_features_from_generator,
 * utils.py:542-601): int16 clips [B][n] -> embeddings [B][n_out][96], n_out = (F-76)/8 + 1 with F = (n-512)/160 + 1
 * mel frames per clip.  Everything stays on the device: per-clip mel (one clamp floor per clip, x/10+2), then the
 * incremental CNN walks each clip in steps of 8 mel rows (4 lead-in rows + 9 warm-up steps, then one embedding per
 * step) -- 7.5x fewer flops than evaluating every 76-row window on its own, same results.  `pcm` / `out` are host
 * pointers unless the matching *_on_device flag is set.  Requires B <= n_streams, F >= 76; like oww_embed it borrows the
 * streaming machinery of streams [0,B) and restores their state before returning. */
int  oww_embed_clips(oww_ctx* h, const int16_t* pcm, int32_t pcm_on_device, int32_t B, int32_t n,
                     float* out, int32_t out_on_device);
int  oww_head(oww_ctx* h, int32_t head, const float* features, int32_t B, float* out);

/* ---- introspection ------------------------------------------------------------------------------ */
/* AudioFeatures.get_features (utils.py:454-460): last T rows of stream sid's feature ring, oldest first */
int  oww_get_features(oww_ctx* h, int32_t sid, int32_t T, float* out);
/* newest n_rows (<= 8*n_chunks of the last step) transformed mel rows (x/10+2) of stream sid from the last step */
int  oww_get_mel(oww_ctx* h, int32_t sid, float* out, int32_t n_rows);
/* per-layer CNN outputs (new rows only, dense [rows][F][C]) of stream sid from the last chunk processed;
 * layer 0..19; needs cfg.debug_layers.  Returns the number of floats written. */
int  oww_debug_read(oww_ctx* h, int32_t sid, int32_t layer, float* out, int32_t cap);
/* in-kernel phase stamps (shader clock) of one workgroup of each of the four generic CNN stage kernels
 * from the last step: out[stage 0..3][wave 0..15][mark 0..15]; only when the environment variable
 * OWW_PROF_BLOCK=<workgroup index> was set at oww_commit time.  Development aid. */
int  oww_debug_profile(oww_ctx* h, int64_t* out, int32_t cap);
/* kernel timing: records hipEvents around every kernel of subsequent oww_step calls when enabled;
 * oww_kernel_times fills ms[i] with the accumulated time and n[i] with the launch count of kernel
 * class i (0 mel,1 stageA,2 stageB,3 stageC,4 stageD,5 stageE,6 heads,7 postproc,8 vad_front,9 vad_lstm) and clears them. */
#define OWW_N_KERNEL_CLASSES 10
int  oww_enable_timing(oww_ctx* h, int on);
int  oww_kernel_times(oww_ctx* h, double ms[OWW_N_KERNEL_CLASSES], int64_t n[OWW_N_KERNEL_CLASSES]);
/* capture the n_chunks==1 device-to-device step into a hipGraph and replay it on later steps */
int  oww_use_graph(oww_ctx* h, int on);

#ifdef __cplusplus
}
#endif
#endif /* OWWHIP_H */
