/* nhans_hip.h -- C ABI of libnhans_hip.so, the MI355X (gfx950) implementation of the N-HANS
 * per-frame inference hot path.
 *
 * The reference has no FFI/plugin interface: its hot path is a chain of TensorFlow graph calls
 * inside apply_snc / apply_separator.  Each entry point below replaces one link of that chain
 * (citations relative to /root/reference; SN = N_HANS___Selective_Noise, SS = N_HANS___Source_Separation):
 *
 *   nhans_stft_features  <- tf.signal.stft + log(abs+1e-5) + angle       SN/apply.py:368-375  (SS/apply.py:316-322)
 *   nhans_embed          <- model(): 'embedding' scopes                   SN/main.py:190-216   (SS/main.py:206-229)
 *   nhans_mask_net       <- strided_crop + minibatch loop + model() stack SN/apply.py:378,398-450, SN/main.py:219-242
 *                           (tensor contract: feed mixedph/noise*contextph, fetch add_72:0, SN/apply.py:434-446)
 *   nhans_istft          <- recover_samples_from_spectrum                 SN/apply.py:189-204  (SS/apply.py:158-171)
 *   nhans_enhance_clips  <- apply_snc / apply_separator after handle_signals  SN/apply.py:368-458 (SS/apply.py:316-393)
 *
 * Conventions: plain C, no exceptions cross the boundary.  Every int-returning function gives 0
 * on success or a negative NHANS_E* code, with a message in nhans_last_error() (thread-local).
 * `*_dev` pointers are device (HBM) addresses owned by the caller; `*_host` pointers are host
 * arrays.  `stream` is a hipStream_t (0 = default stream); all work is enqueued on it and the
 * functions do not synchronise.  The library owns only its context: folded weights and a
 * workspace that grows on demand.  One context per (device, model); a context is not
 * thread-safe, distinct contexts are independent (also on different devices of one process).
 * Consecutive calls on one context share its workspace: the library orders them itself -- a call
 * on another stream than the previous one first makes its stream wait (hipStreamWaitEvent) for the
 * previous call's work -- so results never depend on the caller's choice of streams, but two calls
 * on one context never overlap either; use one context per concurrent stream for that.
 * A kernel launch the runtime rejects (invalid grid, LDS request, ...) is reported by the entry
 * point that issued it as NHANS_EHIP, naming the kernel; nothing runs on unlaunched results.
 *
 * Ragged batches: clip c owns samples [sample_offsets[c], sample_offsets[c+1]) of the wav
 * buffer and frames [frame_offsets[c], frame_offsets[c+1]) of every [T_total, 201] tensor, with
 * T_c = nhans_num_frames(n_c) = 1 + (n_c - 400) / 160 for n_c >= 400 (the caller has already
 * applied the reference's normalise + tail-trim, SN/apply.py:150-161).  A batch may hold any number of
 * frames (64-bit offsets between clips); ONE clip is limited to 5,000,000 frames (13.9 hours: offsets inside a
 * clip are 32-bit) and a longer one is refused with NHANS_EINVAL.
 */
#ifndef NHANS_HIP_H
#define NHANS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NHANS_ABI_VERSION 5

#define NHANS_DENOISER 0   /* SN model: emb_a = positive context (--pos), emb_b = negative (--neg) */
#define NHANS_SEPARATOR 1  /* SS model: emb_a = interferer  (--neg),      emb_b = target   (--pos) */

#define NHANS_OK 0
#define NHANS_EINVAL (-1)   /* bad argument / malformed blob */
#define NHANS_EHIP (-2)     /* HIP runtime error */
#define NHANS_ENOMEM (-3)   /* workspace allocation failed */
#define NHANS_ESHORT (-4)   /* a conditioning recording yields fewer than 200 frames */

/* sticky device-side status bits, see nhans_take_status() */
#define NHANS_STATUS_SATURATED 1   /* precision 1: an activation did not fit the f16 range (|v| >= 65504 or NaN)
                                      and was clamped -- the outputs of the calls since the last
                                      nhans_take_status() are not trustworthy; rerun them with precision 0,
                                      best inside a "calibrate" bracket so that the exponents follow */

/* tensors of the network that carry an activation exponent (see "calibrate" below): tower block b conv1 / conv2
 * outputs at 2b / 2b+1 (b < 4), stack block b conv1 / conv2 outputs at 8+2b / 8+2b+1 (b < 8), last_conv at 24 */
#define NHANS_NUM_ACTIVATIONS 25

#define NHANS_WIN 400
#define NHANS_HOP 160
#define NHANS_BINS 201
#define NHANS_MIX_WIN 35
#define NHANS_CTX_FRAMES 200
#define NHANS_EMB 512

typedef struct nhans_ctx nhans_ctx;

int nhans_abi_version(void);
const char* nhans_last_error(void);

/* frames of an n-sample (already trimmed or not) signal: 0 if n < 400, else 1 + (n-400)/160 */
int64_t nhans_num_frames(int64_t nsamples);

/* Build a context from the folded-weights blob produced by nhans_amd.fold.fold_weights()
 * (format documented in n-hans_amd/fold.py).  The blob is copied to the device.  If the blob carries split-f16
 * weights, the activation exponents are calibrated here on a built-in two-second signal (one small pass of the whole
 * path in f32 mode; see "calibrate").  The blob's header carries the version of the packing conventions
 * (fold.BLOB_VERSION, 2 since ABI 4); a blob of another version is refused with NHANS_EINVAL -- it would load and compute
 * wrong results -- and the message says to fold the weights again. */
int nhans_create(int model_kind, const void* folded_blob, size_t nbytes, int device_id, nhans_ctx** out);

/* (ABI 5) The same with the activation exponents handed in -- the NHANS_NUM_ACTIVATIONS values a previous context
 * reported for THIS blob (nhans_get_activation_exponents after nhans_create): the calibration pass is skipped, which
 * with a cached folded blob (nhans_amd/blobcache.py) is what brings a cold one-file `nhans_denoiser` call -- the
 * reference's normal use, SN/apply.py:478-527: one process per file -- under a second.  act_exp == NULL: as nhans_create.
 * Exponents outside [-60, 60] or n_exp != NHANS_NUM_ACTIVATIONS: NHANS_EINVAL.  Wrong exponents cannot corrupt results
 * silently: too small raises NHANS_STATUS_SATURATED (Engine redoes the batch in f32), too large costs accuracy bits. */
int nhans_create_ex(int model_kind, const void* folded_blob, size_t nbytes, int device_id, const int* act_exp, int n_exp,
                    nhans_ctx** out);
void nhans_destroy(nhans_ctx* ctx);

/* Options: "frames_per_chunk" (mask-net frame windows per pass, 1..4769 -- the conv kernels address a pass's
 *           largest tensor, frames x 35 x 201 x 64 elements, with 32-bit offsets; default 3776: 24 GB of workspace
 *           for batches that large, 6.4 MB per frame, chosen so that the launches fill whole waves of 256 workgroups),
 *          "contexts_per_chunk" (embedding-tower images per pass, default 64),
 *          "profile" (1: time every kernel launch with hipEvents on the launch stream),
 *          "precision" (0: exact f32 matrix-core path, default; 1: split-f16 x3 -- every operand is
 *           carried as hi+lo f16 and multiplied with three f16 MFMAs into an f32 accumulator;
 *           FP32-class accuracy, activations must stay below the f16 range 65504),
 *          "conv_variant" (-1: automatic, default; 0: register-staged 128-pixel kernel; 1: LDS-DMA
 *           256-pixel kernel; 2: halo-reuse kernel with producer/consumer waves where the conv
 *           allows it (512-pixel tiles for the 64-channel convs), the same pipeline with one staged image per tap
 *           for the strided / VALID convs with >= 128 output channels, else 1 -- same results
 *           within rounding, different speed).
 *          "epilogue_wide" (1, default: split-f16 epilogues move 8 channels = 16-byte pieces per thread;
 *           0: 4 channels -- identical bits, kept for A/B),
 *          "consumer_interleave" (1, default: the MFMA waves of the halo kernel issue their LDS operand
 *           reads between their MFMAs, one behind each of the first MFMAs of a half-tap; 2: spread evenly over the
 *           half-tap; 0: read block then MFMA block -- identical bits, kept for A/B),
 *          "calibrate" (activation exponents of the split-f16 mode: every stored tensor is kept as x * 2^-e with one
 *           integer e per tensor, so that models whose activations are far from O(1) -- no BatchNorm statistics can
 *           promise that -- stay inside the f16 range instead of tripping NHANS_STATUS_SATURATED; scaling by a power
 *           of two is exact, the arithmetic is otherwise bit for bit that of e = 0.  1: start recording the largest
 *           |x| of every tensor in the calls that follow (any precision; precision 0 cannot saturate and is what a
 *           calibration on own data should use); 0: stop and set every e so that the recorded maximum is stored as
 *           at most 2^8; 2: stop and only RAISE exponents (what a caller does after a saturated batch: rerun it at
 *           precision 0 inside the bracket -- that is the correct result for it -- and go on at precision 1);
 *           3: stop and discard (the recorded pass failed).  Stopping synchronises the device.  A tensor the pass never
 *           wrote keeps its exponent.  A maximum that is not finite is refused with NHANS_EINVAL by 0 and ignored by 2
 *           (a NaN / Inf INPUT raises the flag as well and says nothing about the range).  Exponents raised after a
 *           saturated batch persist: the bits of later batches depend on that history -- set them explicitly
 *           (nhans_set_activation_exponents) where ranks or runs must agree bit for bit.),
 *          "winograd" (1, default: in split-f16 mode the stride-1 4x4 convs of the residual stack run as 1-D
 *           Winograd convolutions F(5,4) along the image width, 2.5 x fewer matrix-core MACs -- conv_wino.hip;
 *           0: the direct kernels for every conv -- results agree to ~1e-5 on the logits),
 *          "winograd_f32_tensors" (1, default: with the Winograd form, the stack tensors that only Winograd launches read
 *           are stored f32 NHWC instead of split NHWC -- same size, same scaled values, less work in the transform;
 *           0: every tensor split -- results agree to ~1e-6 on the logits; 2: a TEST value -- f32 storage whatever the
 *           readers are: a reader that is not a Winograd launch then refuses, the call returns NHANS_EHIP and nothing is
 *           computed on a wrong layout; 3: a TEST value -- only the output of resblock1_2 is f32: its conv2 then has a
 *           split residual and an f32 output, the one layout pair the Winograd epilogue does not implement, and the launch
 *           is refused the same way.  After a refused or failed launch nothing further of that pass is launched.),
 *          "stream_1x1" (1, default: the stand-alone 1x1 strided `_transform` conv of resblock2_1 -- an HBM stream, 1.75 GB in and
 *           3.5 GB out per pass -- runs on its own streaming kernel, conv_1x1_stream.hip; 0: on the generic implicit-GEMM kernel.
 *           Identical bits),
 *          "split_k" (1, default: the launches too small to fill the chip -- the head's dense layer, the embedding tower
 *           at a few clips -- run one workgroup per (tile, K group) through a scratch buffer; 0: every workgroup walks
 *           its K groups itself.  The groups and the order of the additions depend on the layer only: identical bits),
 * (A `make DEV=1` build adds "debug_cycles_ptr" and the NHANS_ABLATE environment switch used by tools/; the default build has no developer hooks and reads no environment.)
 * Besides the workspace a context holds split-K scratch for the few launches that are too small to fill the chip (the
 * head's dense layer, the embedding tower at a few clips): allocated on the first such launch, sized by what the launches
 * need (32 MB ... 384 MB; 120 MB for the head at the default 3,776 frame windows per pass).  If that allocation fails
 * the launches run unsplit -- same bits, slower -- and nothing is reported. */
int nhans_set_option(nhans_ctx* ctx, const char* key, int64_t value);

/* Activation exponents (see "calibrate"): n must be NHANS_NUM_ACTIVATIONS.  Tensors that share one accumulator -- the
 * input of a channel-changing block and its conv1 output -- are kept on one exponent (the larger); `get` returns what
 * is in effect.  nhans_get_activation_amax: the maxima the last finished calibration recorded. */
int nhans_set_activation_exponents(nhans_ctx* ctx, const int* e, int n);
int nhans_get_activation_exponents(nhans_ctx* ctx, int* e_out, int n);
int nhans_get_activation_amax(nhans_ctx* ctx, float* amax_out, int n);

/* Bytes of device workspace the context would hold for a batch of this shape. */
size_t nhans_workspace_bytes(nhans_ctx* ctx, int64_t total_frames, int nclips);

/* wav -> log-magnitude and phase.  max_frames_per_clip > 0 truncates every clip to its first
 * frames (used for the 200-frame conditioning contexts, SN/apply.py:381).  Outputs are
 * [sum_c min(T_c, max), 201] float32; phase_dev may be NULL. */
int nhans_stft_features(nhans_ctx* ctx, const float* wav_dev, const int64_t* sample_offsets_host,
                        int nclips, int max_frames_per_clip, float* logmag_dev, float* phase_dev,
                        void* stream);

/* Embedding tower: n context images [n,200,201] -> [n,512]. */
int nhans_embed(nhans_ctx* ctx, const float* ctx_logmag_dev, int n, float* emb_out_dev, void* stream);

/* Conditioned residual stack + head over every frame of every clip.  logmag [T_total,201];
 * emb_a/emb_b [nclips,512] in resnet_block argument order (see NHANS_DENOISER/SEPARATOR).
 * logits_out_dev (nullable) receives `out` = last_dense output; denoised_out_dev receives
 * logmag + out (the reference's add_72:0).  Sliding 35-frame windows are gathered on the fly
 * with rows outside the clip equal to 0.0 (SN/apply.py:170-186). */
int nhans_mask_net(nhans_ctx* ctx, const float* logmag_dev, const int64_t* frame_offsets_host,
                   int nclips, const float* emb_a_dev, const float* emb_b_dev,
                   float* logits_out_dev, float* denoised_out_dev, void* stream);

/* exp/polar -> 400-point inverse real FFT -> synthesis window -> overlap-add.  Clip c writes
 * (T_c-1)*160+400 samples at wav_out_dev + out_offsets_host[c]. */
int nhans_istft(nhans_ctx* ctx, const float* logmag_dev, const float* phase_dev,
                const int64_t* frame_offsets_host, int nclips, const int64_t* out_offsets_host,
                float* wav_out_dev, void* stream);

/* Whole hot path for a ragged batch.  Mixture clips must be trimmed so (n-400)%160 == 0;
 * context clips need >= 32,240 samples (only their first 200 frames are used).  ctx_a/ctx_b
 * follow the emb_a/emb_b convention.  denoised_wav_dev and mixed_wav_dev (nullable: the
 * reference's *mixed_processed.wav round trip) use the mixture's sample offsets.  Optional
 * taps (nullable): logmag_out_dev/phase_out_dev/logits_out_dev [T_total,201], emb_out_dev
 * [2*nclips,512] (a-rows then b-rows). */
int nhans_enhance_clips(nhans_ctx* ctx, const float* mix_wav_dev, const int64_t* mix_offsets_host,
                        int nclips, const float* ctx_a_wav_dev, const int64_t* ctx_a_offsets_host,
                        const float* ctx_b_wav_dev, const int64_t* ctx_b_offsets_host,
                        float* denoised_wav_dev, float* mixed_wav_dev, float* logmag_out_dev,
                        float* phase_out_dev, float* logits_out_dev, float* emb_out_dev, void* stream);

/* Debug tap: run the stack on `nframes` frame windows starting at global frame `frame0` and copy
 * the NHWC output of main block `block` (0..7; 8 = last_conv) to out_dev. */
int nhans_debug_block_output(nhans_ctx* ctx, const float* logmag_dev, const int64_t* frame_offsets_host,
                             int nclips, const float* emb_a_dev, const float* emb_b_dev,
                             int64_t frame0, int nframes, int block, float* out_dev, void* stream);

/* Waits for `stream`, then returns the context's sticky status bits (NHANS_STATUS_*) in
 * *flags_out and clears them.  The hot-path calls are asynchronous, so conditions detected on the
 * device (split-f16 activation overflow) cannot be part of their return code; a caller that uses
 * precision 1 on weights it has not validated calls this once per batch. */
int nhans_take_status(nhans_ctx* ctx, int* flags_out, void* stream);

/* Self-test of the launch-error path (needs no context): launches a trivial kernel on the current
 * device with `dynamic_lds_bytes` of dynamic LDS.  Returns NHANS_OK if the launch was accepted,
 * NHANS_EHIP (with the runtime's message in nhans_last_error()) if not -- e.g. for a request above
 * the 160 KB a gfx950 CU has, or when no HIP device is present. */
int nhans_debug_launch_probe(size_t dynamic_lds_bytes, void* stream);

/* Measures what the f16 matrix pipes of the current device sustain at its socket power cap (needs no
 * context): back-to-back independent v_mfma_f32_32x32x16_f16 -- the instruction the split-f16 conv kernels
 * issue -- on pseudo-random register operands, no LDS, no memory, launch after launch for `seconds`
 * (0 < seconds <= 60; blocks the calling thread).  *sustained_tflops = mean rate over the second half of the
 * launches, *first_tflops (nullable) = the first launch (boost clock), *launches (nullable) = launches run.
 * bench.py reports it as roofline.peak_at_power_cap: the data-sheet 2.5 PFLOP/s is reached only on operands
 * that do not toggle the multipliers.  Replaces nothing in the reference: measurement. */
int nhans_debug_mfma_ceiling(double seconds, void* stream, double* sustained_tflops, double* first_tflops,
                             int* launches);

/* Host helper (no device involved): CRC-32C (Castagnoli) of a host buffer continued from `crc`
 * (0 to start).  TensorFlow checkpoint bundles store crc32c::Mask() of it per tensor; tfbundle.py
 * verifies the 116 MB data shard with it (BundleEntryProto field 6 of the reference's shipped trained_model .index files). */
uint32_t nhans_crc32c(uint32_t crc, const void* data_host, size_t nbytes);

/* Profiling (option "profile" = 1): per-kernel launch counts, summed milliseconds, summed
 * algorithmic FLOPs / bytes ("flops": 2*M*K*N of the DIRECT convolution whichever form runs it) and the
 * FLOPs the matrix cores executed for them ("mfma_flops": three products per MAC in split-f16 mode, fewer
 * MACs for the Winograd form) since the last reset, as a JSON object written to buf.  Synchronises
 * the recorded events.  Returns the number of bytes needed (excluding the NUL). */
int nhans_profile_json(nhans_ctx* ctx, char* buf, size_t buflen);
int nhans_profile_reset(nhans_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* NHANS_HIP_H */
