/*
 * sir_hip.h -- C ABI of the MI355X-native speech-intent hot path (libsir_hip.so).
 *
 * The reference (avi2924/Speech-Intent-Recognizer) is pure Python on torch/torchaudio and has
 * no FFI of its own; its boundary for this path is the Python surface listed below.  Each entry
 * point here names the reference call it replaces (file:line under /root/reference).  The Python
 * host layer (speech-intent-recognizer_amd/) binds these with ctypes and keeps the reference's
 * signatures; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions: plain C, no torch types.  Unless stated, every pointer is a DEVICE pointer owned by
 * the caller; `stream` is a hipStream_t passed as void*.  All calls are asynchronous on `stream`,
 * allocate nothing, and return 0 on success or a negative SIR_E* code (never throw);
 * sir_last_error() gives the message of the last failure on the calling thread.
 */
#ifndef SIR_HIP_H
#define SIR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIR_ABI_VERSION 1

#define SIR_OK 0
#define SIR_EINVAL (-1)   /* bad argument (shape, alignment, NULL) */
#define SIR_ENOMEM (-2)   /* workspace too small / hipMalloc failed */
#define SIR_EHIP (-3)     /* HIP runtime error */
#define SIR_EUNSUPPORTED (-4)
#define SIR_ETIMEOUT (-5) /* a GRU recurrence kernel gave up waiting for a peer workgroup: results invalid */

#define SIR_WAVE_F32 0
#define SIR_WAVE_I16 1    /* PCM16; dequantised as s / 32768 (torchaudio.load convention) */

typedef struct sir_handle sir_handle;

/* Feature-extractor configuration.
 * Replaces AudioFeatureExtractor.__init__ (scripts/precompute_features.py:21-36), i.e.
 * torchaudio MelSpectrogram(sample_rate, n_fft, hop_length, n_mels) + AmplitudeToDB() with their
 * defaults: win_length = n_fft, periodic Hann, power 2, center + reflect pad, HTK mel, norm None,
 * f_min 0, f_max sample_rate/2, 10*log10(max(x,1e-10)).
 * `window` / `mel_fb` are optional HOST arrays (n_fft floats; [n_fft/2+1][n_mels] dense, row-major)
 * so that the host can hand over torch's own float32 tables bit for bit; NULL = computed in double
 * here and rounded to float. */
typedef struct sir_feature_config {
    int sample_rate;   /* 16000 */
    int n_fft;         /* 1024 (the only size built) */
    int hop_length;    /* 512  (= n_fft/2, the only hop built) */
    int n_mels;        /* 64 (<= 64) */
    float f_min;       /* 0 */
    float f_max;       /* sample_rate/2 */
    const float* window;
    const float* mel_fb;
} sir_feature_config;

/* Optional fused augmentation (scripts/augment.py:6-28 time_shift, :82-96 add_noise on the
 * waveform; scripts/dataset.py:160-176 SpecAugment masks on the features).  Any pointer may be
 * NULL (= that augmentation off).  All arrays are DEVICE arrays of length batch. */
typedef struct sir_augment {
    const int32_t* shift;        /* samples; >0 delays (zero fill on the left), <0 advances */
    const float* noise_sigma;    /* N(0, sigma^2) added per sample, counter-based RNG */
    uint64_t noise_seed;
    const int32_t* time_mask;    /* [batch][2] = {start frame, width}, width 0 = none */
    const int32_t* freq_mask;    /* [batch][2] = {start mel,   width} */
} sir_augment;

int sir_abi_version(void);
const char* sir_last_error(void);

/* Create / destroy a handle.  Uploads window, twiddles and the sparse mel filterbank to the
 * current HIP device (the only allocating calls).  Host-synchronous. */
int sir_create(const sir_feature_config* cfg, sir_handle** out);
int sir_destroy(sir_handle* h);

/* ---- waveform front-end --------------------------------------------
 * sir_mix_to_mono replaces `waveform = torch.mean(waveform, dim=0, keepdim=True)` after
 *   torchaudio.load (scripts/precompute_features.py:47-51, scripts/dataset.py:126-130,
 *   scripts/test_model.py:62-66) for a batch of decoded clips.
 *   pcm   : [batch][clip_stride] INTERLEAVED samples (frame-major, `channels` per frame), i16 or f32
 *           (dtype = SIR_WAVE_*); i16 is dequantised as s / 32768 first, as torchaudio.load does
 *   frames: device int32[batch] (NULL = all max_frames); out rows are zero beyond frames[b]
 *   out   : [batch][out_stride] f32 mono
 * sir_resample replaces `torchaudio.transforms.Resample(sr, 16000)(waveform)`
 *   (precompute_features.py:54-56, dataset.py:132-135, test_model.py:68-72): sinc_interp_hann,
 *   lowpass_filter_width 6, rolloff 0.99, output length ceil(new * length / orig) per clip.
 *   wave/lengths as in sir_features_fwd; out: [batch][out_stride] f32, zero beyond the clip's output
 *   length (clamped to max_out_len); out_lengths: optional device int32[batch].
 *   The first call for a rate pair builds the filter table on the host and uploads it (host-synchronous,
 *   two small hipMallocs owned by the handle); later calls only launch.
 * sir_resample_out_len: host helper, ceil(new * length / orig) with gcd-reduced rates (-1 on bad input). */
int sir_mix_to_mono(sir_handle* h, const void* pcm, int dtype, int channels, int64_t clip_stride,
                    const int32_t* frames, int batch, int max_frames, float* out, int64_t out_stride,
                    void* stream);
int sir_resample_out_len(int length, int orig_freq, int new_freq);
int sir_resample(sir_handle* h, const void* wave, int wave_dtype, int64_t wave_stride,
                 const int32_t* lengths, int batch, int max_len, int orig_freq, int new_freq, float* out,
                 int64_t out_stride, int max_out_len, int32_t* out_lengths, void* stream);

/* ---- feature path --------------------------------------------------------------------------
 * sir_features_fwd replaces, for a whole batch in one launch pair,
 *   AudioFeatureExtractor.extract_features  scripts/precompute_features.py:59-73
 *     (truncate is done by the caller through `lengths`; mel power, dB, whole-utterance z-norm)
 *   FSCIntentDataset.extract_features       scripts/dataset.py:137-152  (same arithmetic)
 *   pad / trim to t_pad frames              scripts/dataset.py:109-113, scripts/train.py:58-62
 * wave   : [batch][wave_stride] samples, f32 or i16 (wave_dtype), only [0, lengths[b]) is read
 * lengths: device int32[batch]; a clip with length <= n_fft/2 yields an all-zero row (the reference
 *          fails in torch.stft's reflect pad and substitutes zeros, dataset.py:121-123,156-158)
 * out    : [batch][n_mels][t_pad] f32; frames >= 1 + length/hop are zero
 * db_out : optional (NULL = off) [batch][n_mels][t_pad] f32 copy of the un-normalised dB values
 *          (10*log10(max(mel,1e-10)), zero in the padding) -- the AmplitudeToDB output of
 *          precompute_features.py:67, exposed so that the mel/dB stage can be checked on its own
 * workspace: sir_features_workspace_bytes(batch, max_len) bytes, 16-byte aligned */
size_t sir_features_workspace_bytes(const sir_handle* h, int batch, int max_len);
int sir_features_fwd(sir_handle* h, const void* wave, int wave_dtype, int64_t wave_stride,
                     const int32_t* lengths, int batch, int max_len, float* out, int t_pad,
                     float* db_out, void* workspace, size_t workspace_bytes, const sir_augment* aug,
                     void* stream);

/* ---- batch assembly from an HBM-resident feature store -----------------------------------------
 * sir_gather_features replaces, for a whole batch in one launch, what the reference does per item in DataLoader worker
 * processes and then copies over PCIe: FSCIntentDataset.__getitem__ (scripts/dataset.py:78-115: cache lookup, SpecAugment,
 * pad / trim -- the store rows are already padded to t_pad) + collate_fn's torch.stack (scripts/train.py:49-70) +
 * mel.to(device) (scripts/train.py:86).  The split's cached features live in HBM once (sir_amd/feature_store.py: a few
 * hundred MB of the 288 GB); a step's batch is a gather by index.
 *   store     : [n_store][n_mels][t_pad] f32 (device), rows zero beyond each clip's frames
 *   index     : device int64[batch], each in [0, n_store) (an index outside yields a zero row and SIR_EINVAL at the next
 *               sir_check_status)
 *   time_mask / freq_mask : optional device int32[batch][2] = {start, width} bands to zero (scripts/dataset.py:160-176,
 *               drawn by the host as torchaudio's mask_along_axis does); NULL = none
 *   out       : [batch][n_mels][t_pad] f32 */
int sir_gather_features(sir_handle* h, const float* store, int64_t n_store, const int64_t* index, int batch,
                        int n_mels, int t_pad, const int32_t* time_mask, const int32_t* freq_mask, float* out,
                        void* stream);

/* ---- model path ----------------------------------------------------------------------------
 * Device pointers to the reference's parameters/buffers under their state_dict names
 * (models/models.py:10-39): index 0..2 = conv1..3 / bn1..3; GRU index = 2*layer + reverse. */
typedef struct sir_model_weights {
    const float* conv_w[3];      /* [32,1,3,3] [64,32,3,3] [128,64,3,3], no bias */
    const float* bn_w[3];
    const float* bn_b[3];
    const float* bn_mean[3];     /* running_mean */
    const float* bn_var[3];      /* running_var  */
    const float* gru_w_ih[4];    /* [768,1024] x2, [768,512] x2 ; gate order r,z,n */
    const float* gru_w_hh[4];    /* [768,256] */
    const float* gru_b_ih[4];    /* [768] */
    const float* gru_b_hh[4];    /* [768] */
    const float* attn_w;         /* [1,512] */
    const float* attn_b;         /* [1] */
    const float* fc_w;           /* [num_classes,512] */
    const float* fc_b;           /* [num_classes] */
    int num_classes;             /* <= 64 */
} sir_model_weights;

/* sir_model_infer replaces CNNAudioGRU.forward in eval() (models/models.py:41-68) followed by
 * torch.argmax(outputs, dim=1) (scripts/evaluate.py:82-83) / torch.max (scripts/train.py:149).
 * feats  : [batch][64][t_frames] f32 (t_frames >= 8; 200 on the training path)
 * logits : [batch][num_classes] f32
 * argmax : int64[batch] or NULL
 * workspace: sir_model_workspace_bytes(batch, t_frames, 0) bytes, 256-byte aligned */
size_t sir_model_workspace_bytes(const sir_handle* h, int batch, int t_frames, int train);
/* Byte offsets of the intermediate buffers inside the workspace, in the order
 *   0 conv1 out NHWC [B][32][T/2][32]   1 conv2 out NHWC [B][16][T/4][64]
 *   2 GRU input [B][S][1024] (feature = c*8+h, models.py:55-57)   3 input projections [B*S][1536]
 *   4 GRU layer-0 out [B][S][512]   5 GRU layer-1 out [B][S][512]   6 context [B][512]  ...
 * so that tests can check every stage against the oracle.  Returns the number of buffers. */
int sir_model_workspace_offsets(const sir_handle* h, int batch, int t_frames, int train,
                                size_t* offsets, int n);
/* Optional: tell the library which version of the weights the next sir_model_infer calls will see.  The
 * derived weight layouts (bf16x3 planes, folded BatchNorm, ...) live in the caller's workspace; when the
 * version is non-zero and unchanged since the previous call with the same workspace and shape, they are
 * reused instead of rebuilt (~40 us per call).  0 (the default) = always rebuild. */
int sir_model_set_weights_version(sir_handle* h, uint64_t version);
int sir_model_infer(sir_handle* h, const sir_model_weights* w, const float* feats, int batch,
                    int t_frames, float* logits, int64_t* argmax, void* workspace,
                    size_t workspace_bytes, void* stream);
/* The GRU recurrence kernels (forward and backward) exchange hidden-state slices between the workgroups of a
 * cluster through tagged granules in global memory and rely on the cluster being co-resident.  A workgroup that
 * spins past its limit (a partitioned / oversubscribed GPU, a stalled peer) sets a device status word owned by the
 * handle and carries on with invalid values.  sir_check_status waits for `stream`, returns SIR_ETIMEOUT if any
 * recurrence launched on this handle since the last check timed out (and clears the word), SIR_OK otherwise.
 * Call it wherever the host synchronises anyway -- once per batch of predictions (scripts/evaluate.py:85-86's
 * .cpu()) or per epoch (scripts/train.py:116's loss.item()); sir_profile_collect performs the same check.
 * The same word carries sir_ce_loss's "label outside [0, num_classes)" flag (nn.CrossEntropyLoss raises on such a
 * target, train.py:242/:105; the kernel makes that step's loss NaN): reported here as SIR_EINVAL. */
int sir_check_status(sir_handle* h, void* stream);

/* ---- cross-batch pipelining (owned by the library) ------------------------------------------------
 * One batch's kernels run back to back on one stream, and some of them cannot fill the chip on their own: the GRU
 * recurrence is a chain of S dependent steps that occupies half of the CUs at a fraction of their matrix throughput
 * (all utterances of the batch already advance in parallel inside it, so splitting a batch does not shorten the chain --
 * DESIGN.md section 4).  What hides it is the NEXT batch's convolutions.  A sir_pipeline owns n_slots HIP streams and the
 * events that order them against the caller's stream, so that a caller living on ONE stream gets the overlap:
 *     sir_pipeline_begin(p, caller_stream, &slot, &slot_stream)   -- slot = submission count mod n_slots; its stream now
 *                                                                    waits for everything queued on caller_stream
 *     sir_features_fwd(..., slot_stream); sir_model_infer(..., slot_stream);   (buffers / workspace of THAT slot)
 *     sir_pipeline_end(p, slot)                                   -- marks the slot's work complete-able
 *     ... more batches ...
 *     sir_pipeline_join(p, caller_stream)                         -- caller_stream waits for every slot (no host sync)
 * A slot's buffers are reused n_slots submissions later (same stream: ordered); the caller reads results on
 * caller_stream after the join.  n_slots = 1 degenerates to the caller's own stream.  Results are bit-identical to the
 * single-stream order.  Replaces nothing in the reference (its evaluate loop is serial, scripts/evaluate.py:79-86). */
typedef struct sir_pipeline sir_pipeline;
int sir_pipeline_create(sir_handle* h, int n_slots, sir_pipeline** out);     /* 1 <= n_slots <= 4 */
int sir_pipeline_destroy(sir_pipeline* p);
int sir_pipeline_begin(sir_pipeline* p, void* caller_stream, int* slot, void** slot_stream);
int sir_pipeline_end(sir_pipeline* p, int slot);
int sir_pipeline_join(sir_pipeline* p, void* caller_stream);

/* ---- training step ----------------------------------------------------------------------------
 * Replaces the body of train_epoch (scripts/train.py:90-107): model(mel) in train() mode,
 * criterion(output, label), loss.backward(), optimizer.step().
 * Gradients are written (not accumulated) to the device pointers of sir_model_grads, which mirror
 * the parameter pointers of sir_model_weights (the BN running statistics have no gradient). */
typedef struct sir_model_grads {
    float* conv_w[3];
    float* bn_w[3];
    float* bn_b[3];
    float* gru_w_ih[4];
    float* gru_w_hh[4];
    float* gru_b_ih[4];
    float* gru_b_hh[4];
    float* attn_w;
    float* attn_b;
    float* fc_w;
    float* fc_b;
} sir_model_grads;

/* Training-mode forward (models/models.py:41-68 under model.train()): BatchNorm uses batch
 * statistics and updates bn_running_mean/var IN PLACE (momentum, unbiased variance); the inter-layer
 * GRU dropout (models.py:32) uses a counter-based mask keyed by dropout_seed (dropout_p = 0 turns it
 * off); activations needed by the backward pass stay in `workspace`
 * (sir_model_workspace_bytes(h, batch, t_frames, 1) bytes), which must be handed unchanged to
 * sir_model_train_bwd. */
int sir_model_train_fwd(sir_handle* h, const sir_model_weights* w, float* const bn_running_mean[3],
                        float* const bn_running_var[3], const float* feats, int batch, int t_frames,
                        float bn_momentum, float dropout_p, uint64_t dropout_seed, float* logits,
                        void* workspace, size_t workspace_bytes, void* stream);

/* nn.CrossEntropyLoss() (mean) of train.py:242/105: loss[0] = -mean log softmax(logits)[label];
 * dlogits (optional) = d loss / d logits * grad_scale.  As torch's default ignore_index, a label of -100 takes its row out of
 * the loss, the gradient and the mean's divisor; any other label outside [0, num_classes) -- torch raises on it -- makes the
 * loss NaN and is reported by sir_check_status (SIR_EINVAL). */
int sir_ce_loss(sir_handle* h, const float* logits, const int64_t* labels, int batch, int num_classes,
                float* loss, float* dlogits, float grad_scale, void* stream);

/* loss.backward() (train.py:106): all 29 parameter gradients from dlogits and the saved workspace. */
int sir_model_train_bwd(sir_handle* h, const sir_model_weights* w, const float* feats, const float* dlogits,
                        int batch, int t_frames, float dropout_p, uint64_t dropout_seed,
                        const sir_model_grads* grads, void* workspace, size_t workspace_bytes, void* stream);
/* The same backward in two halves, for data-parallel training that starts the gradient exchange early (SURVEY.md
 * section 8(e)): SIR_BWD_HEAD_GRU writes the fc / attention / GRU gradients (96 % of the bytes; they are final after this
 * call) and leaves d(loss)/d(GRU input) in the workspace; SIR_BWD_CNN, called next on the same workspace, writes the
 * conv / BatchNorm gradients.  SIR_BWD_ALL = sir_model_train_bwd. */
#define SIR_BWD_ALL 0
#define SIR_BWD_HEAD_GRU 1
#define SIR_BWD_CNN 2
int sir_model_train_bwd_part(sir_handle* h, const sir_model_weights* w, const float* feats, const float* dlogits,
                             int batch, int t_frames, float dropout_p, uint64_t dropout_seed,
                             const sir_model_grads* grads, void* workspace, size_t workspace_bytes, int part,
                             void* stream);
int sir_model_train_workspace_offsets(const sir_handle* h, int batch, int t_frames, size_t* offsets, int n);

/* optimizer.step() for torch.optim.Adam(lr, betas, eps, weight_decay) with coupled L2
 * (train.py:246-250, :107): one multi-tensor launch.  The pointer arrays are HOST arrays of device
 * pointers (n_tensors <= 32); `step` is the 1-based step count used for bias correction. */
int sir_adam_step(sir_handle* h, int n_tensors, float* const* params, const float* const* grads,
                  float* const* exp_avg, float* const* exp_avg_sq, const int64_t* sizes, int step,
                  float lr, float beta1, float beta2, float eps, float weight_decay, void* stream);

/* ---- measurement -----------------------------------------------------------------------------
 * HIP-event timing of the kernels of the path, recorded on the stream they are launched on
 * (bench.py's roofline figures come from here).  mode 0 = off, 1 = every kernel, 2 = only
 * `kernel_id`.  sir_profile_collect waits for the recorded events, returns per-kernel total
 * milliseconds and launch counts since the last collect (arrays of length n), and resets. */
int sir_profile_kernel_count(void);
const char* sir_profile_kernel_name(int kernel_id);
int sir_profile_enable(sir_handle* h, int mode, int kernel_id);
int sir_profile_collect(sir_handle* h, double* total_ms, int64_t* launches, int n);

#ifdef __cplusplus
}
#endif
#endif /* SIR_HIP_H */
