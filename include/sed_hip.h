/*
 * sed_hip.h -- C ABI of libsedhip.so: hand-written HIP (gfx950) kernels for the batched log-mel
 * front-end and the frame-level event decoders of a sound-event-detection pipeline.
 *
 * Plain pointers and sizes only.  Unless stated otherwise every data pointer is a DEVICE pointer on
 * the device that was current when the plan was created, every call is asynchronous on `stream`
 * (a hipStream_t passed as void*), performs no allocation and no host synchronisation, and returns
 * 0 on success or a negative sed_status.  Plans are immutable after creation/set_* and may be shared
 * across host threads; one in-flight call per (plan, stream).
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference root).
 */
#ifndef SED_HIP_H_
#define SED_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sed_status {
  SED_OK = 0,
  SED_ERR_BAD_ARG = -1,          /* null pointer, negative size, inconsistent configuration      */
  SED_ERR_UNSUPPORTED_NFFT = -2, /* n_fft not in {512, 1024, 2048}                                */
  SED_ERR_CAPACITY = -3,         /* a fixed-capacity output would overflow                        */
  SED_ERR_SHORT_INPUT = -4,      /* reflect padding needs L > n_fft/2                             */
  SED_ERR_WRONG_DEVICE = -5,     /* current device differs from the plan's device                 */
  SED_ERR_HIP = -1000            /* -(1000 + hipError_t) for HIP runtime failures                 */
} sed_status;

enum { SED_PAD_NONE = 0, SED_PAD_CONSTANT = 1, SED_PAD_REFLECT = 2 };     /* centre padding mode   */
enum { SED_SPEC_MAGNITUDE = 1, SED_SPEC_POWER = 2 };
enum { SED_MEL_SLANEY = 0, SED_MEL_HTK_TF = 1, SED_MEL_KALDI = 2 };
enum { SED_LOG_NONE = 0, SED_LOG_DB10 = 1, SED_LOG_LN = 2, SED_LOG_LN_FLOOR = 3 };
enum { SED_PCM_F32 = 0, SED_PCM_I16 = 1 };
enum { SED_WIN_CENTER = 0, SED_WIN_LEFT = 1 };
enum { SED_WINDOW_HANN_PERIODIC = 0, SED_WINDOW_HANN_SYMMETRIC = 1 };
enum { SED_SMOOTH_NONE = 0, SED_SMOOTH_MEDIAN3 = 1, SED_SMOOTH_GAUSS25 = 2 };

/*
 * One front-end flavour.  The four the reference uses:
 *   HTS-AT  (external/hts_audio_transformer/model/htsat.py:443-459): 32000/1024/1024/320, center, REFLECT,
 *           POWER, SLANEY 64 mel 50..14000, DB10(amin 1e-10, ref 1, top_db none), WIN_CENTER
 *   YAMNet  (src/models/yamnet_train.py:46-47): 16000/512/400/160, no centring, MAGNITUDE, HTK_TF 64 mel
 *           125..7500, LN(0.001), WIN_LEFT, tail_min_samples 15600, tail_hop_samples 7680
 *   CRNN    (src/models/crnn/process_audio.py:29-34): 44100/2048/2048/1024, center, CONSTANT, MAGNITUDE,
 *           SLANEY 40 mel 0..22050, LN(1e-9)
 *   dB mel  (src/utils/audio_to_spectrograms.py:30-44): 44100/2048/2048/512, center, CONSTANT, POWER,
 *           SLANEY 64 mel 0..22050, DB10 then sed_db_normalize(per clip, top_db 80)
 * and the Kaldi filterbank features of the Audio-Mamba classifier
 *   AuM     (external/audio_mamba/src/dataloader.py:133-134, torchaudio.compliance.kaldi.fbank(htk_compat=True,
 *           use_energy=False, window_type='hanning', num_mel_bins=128, dither=0, frame_shift=10)): sr/512/400/160 at
 *           16 kHz, no centring, WIN_LEFT, HANN_SYMMETRIC, remove_dc 1, preemphasis 0.97, POWER, KALDI 128 mel
 *           20..sr/2, LN_FLOOR(amin = FLT_EPSILON)
 */
typedef struct sed_frontend_cfg {
  int32_t sample_rate;
  int32_t n_fft;            /* 512, 1024 or 2048                                                   */
  int32_t win_length;       /* <= n_fft; periodic Hann                                             */
  int32_t hop;              /* any; multiples of 4 samples take the aligned (faster) sample reads                  */
  int32_t center;           /* 1: pad n_fft/2 both sides with pad_mode; T = 1 + L/hop              */
  int32_t pad_mode;         /* SED_PAD_* (used when center)                                        */
  int32_t win_align;        /* SED_WIN_CENTER (librosa pad_center) or SED_WIN_LEFT (tf.signal.stft) */
  int32_t tail_min_samples; /* >0: zero-pad the tail to tail_min + tail_hop*ceil((L-tail_min)/tail_hop) */
  int32_t tail_hop_samples;
  int32_t spectrum;         /* SED_SPEC_*                                                          */
  int32_t mel_kind;         /* SED_MEL_*                                                           */
  int32_t n_mels;           /* 1..128                                                              */
  double fmin, fmax;        /* band edges in Hz, fp64 as librosa / torchaudio take them: the triangle feet depend on every bit */
  int32_t log_kind;         /* SED_LOG_*                                                           */
  float amin, ref;          /* DB10: 10*log10(max(x, amin)) - 10*log10(max(amin, ref))             */
  float ln_offset;          /* LN: ln(x + ln_offset)                                               */
  int32_t pcm_kind;         /* SED_PCM_*                                                           */
  float pcm_divisor;        /* I16: sample / pcm_divisor (32767 or 32768)                          */
  int32_t window_kind;      /* SED_WINDOW_*                                                        */
  int32_t remove_dc;        /* 1: subtract the mean of the win_length samples of each frame        */
  float preemphasis;        /* c > 0: y[n] = x[n] - c*x[n-1] inside each frame, x[-1] := x[0]; 0: off */
} sed_frontend_cfg;

typedef struct sed_plan sed_plan;

const char* sed_version(void);

/* Human-readable text for a status returned by any entry point (static storage). */
const char* sed_status_string(int status);

/* ---- host-only table builders (no device needed) ------------------------------------------------- */

/* Mel filterbank as the reference modules hold it: melW[k*n_mels + m], k = 0..n_fft/2.
 * SED_MEL_SLANEY = librosa.filters.mel(htk=False, norm='slaney').T, the buffer logmel_extractor.melW built at
 * external/hts_audio_transformer/model/htsat.py:457; SED_MEL_HTK_TF = tf.signal.linear_to_mel_weight_matrix
 * (the YAMNet graph behind src/models/yamnet_train.py:47); SED_MEL_KALDI = torchaudio.compliance.kaldi.get_mel_banks
 * (low_freq fmin, high_freq fmax, no VTLN) with a zero row for the Nyquist bin, as kaldi.fbank applies it
 * (external/audio_mamba/src/dataloader.py:133).  fp64 arithmetic, rounded once to fp32; SED_MEL_SLANEY follows librosa
 * operation for operation instead (rfftfreq bin frequencies, triangles rounded to fp32, then scaled by the fp64 Slaney
 * factor and rounded again, as its in-place `weights *= enorm` on an fp32 array does). */
int sed_mel_filterbank(int32_t mel_kind, int32_t sample_rate, int32_t n_fft, int32_t n_mels, double fmin, double fmax,
                       float* melW_host);

/* Periodic Hann of win_length inside n_fft samples (centred or left aligned), fp32 [n_fft]. */
int sed_hann_window(int32_t n_fft, int32_t win_length, int32_t win_align, float* window_host);
/* Same with the window kind chosen: SED_WINDOW_HANN_SYMMETRIC = torch.hann_window(win_length, periodic=False),
 * the 'hanning' window of torchaudio.compliance.kaldi (external/audio_mamba/src/dataloader.py:133). */
int sed_window(int32_t window_kind, int32_t n_fft, int32_t win_length, int32_t win_align, float* window_host);

/* ---- plans (host side; tables are built in fp64 on the host and uploaded once) ---------------------- */

/* Replaces the constructors torchlibrosa.stft.Spectrogram(...) + LogmelFilterBank(...) as called at
 * external/hts_audio_transformer/model/htsat.py:453-459 (and the equivalent constant set-up inside
 * librosa.feature.melspectrogram / the YAMNet graph for the other flavours). */
int sed_plan_create(const sed_frontend_cfg* cfg, sed_plan** out);
void sed_plan_destroy(sed_plan* plan);

/* Number of STFT frames for L samples per clip (T of the output), or a negative sed_status. */
int64_t sed_plan_num_frames(const sed_plan* plan, int64_t n_samples);

/* Pure host helpers: the tables a state_dict of the reference module holds
 * (logmel_extractor.melW (F, n_mels); the full-length window).  Host pointers. */
int sed_plan_get_mel_weights(const sed_plan* plan, float* melW_host /* [n_fft/2+1][n_mels] */);
int sed_plan_get_window(const sed_plan* plan, float* window_host /* [n_fft] */);
/* Replace the mel matrix (e.g. the one found in a loaded checkpoint).  Host pointer. */
int sed_plan_set_mel_weights(sed_plan* plan, const float* melW_host /* [n_fft/2+1][n_mels] */);
/* Replace the analysis window by the caller's fp32 values (all n_fft of them, zero padding included).  The plan's own
 * window is the closed form evaluated in fp64 and rounded once; torchaudio.compliance.kaldi builds its 'hanning' window
 * with torch.hann_window(win_length, periodic=False) in fp32 (external/audio_mamba/src/dataloader.py:133), whose bits
 * differ from that in the last place, and this call installs exactly those.  Host pointer. */
int sed_plan_set_window(sed_plan* plan, const float* window_host /* [n_fft] */);
/* Per-mel affine epilogue out = x*scale[m] + shift[m]: eval-mode bn0 (htsat.py:463,767-769) or the
 * CRNN StandardScaler (src/models/crnn/process_audio.py:271-279).  Host pointers; NULL,NULL disables. */
int sed_plan_set_affine(sed_plan* plan, const float* scale_host, const float* shift_host);

/* ---- front-end kernels ----------------------------------------------------------------------------- */

/* Fused framing + window + real FFT + power/magnitude + mel + log (+ affine).
 * Replaces logmel_extractor(spectrogram_extractor(x)) at htsat.py:763-764, the features inside
 * yamnet_model(audio) (yamnet_train.py:47), extract_mbe (crnn/process_audio.py:29-34).
 * pcm: [B] clips of L samples (fp32 or int16 per the plan), clip b at pcm + b*pcm_stride elements.
 * out: fp32, frame t of clip b at out + b*out_stride_b + t*n_mels. */
int sed_logmel_forward(const sed_plan* plan, const void* pcm, int64_t B, int64_t L, int64_t pcm_stride,
                       float* out, int64_t out_stride_b, void* stream);

/* The fused log-mel of SEGMENTS of a resident PCM buffer: the detector -> classifier hand-off of
 * src/utils/cut_audio.py:13-53 + main.py:123-127 (events cut out of the recordings and written as files for stage 2)
 * followed by the classifier's front-end (external/hts_audio_transformer/model/htsat.py:763-764), without the copy in
 * between.  Clip e is the seg_len[e] samples from element seg_off[e] of pcm on (any sample alignment; for a batch
 * [streams][stride]: stream * stride + first sample) followed by zeros up to L samples, i.e. exactly row e of the
 * zero-padded batch sed_cut_events would write with cap = L, and the result equals sed_logmel_forward on that batch bit for
 * bit.  seg_off int64 [n_segments] and seg_len int32 [n_segments] are DEVICE arrays; 0 <= seg_len[e] <= L, and every
 * segment must lie inside the buffer (not checked: the arrays are not read on the host).  out fp32 [n_segments][T][n_mels],
 * T = sed_plan_num_frames(plan, L).  Plans with per-frame conditioning (the Kaldi filterbank) are served; plans with a
 * YAMNet tail return SED_ERR_BAD_ARG. */
int sed_logmel_forward_segments(const sed_plan* plan, const void* pcm, const int64_t* seg_off, const int32_t* seg_len,
                                int64_t n_segments, int64_t L, float* out, int64_t out_stride_b, void* stream);

/* The same hand-off with every segment framed as a RECORDING OF ITS OWN, as stage 2 of the reference sees it: each
 * event file is loaded at its natural length and fed alone (src/models/htsat/data.py:138-146, batch_size = 1 at
 * src/models/htsat/classification.py:55-59), so the padding rule of the STFT applies at the event's own end
 * (htsat.py:763, center=True, pad_mode='reflect') and the event has its own number of frames.  Segment e, n = seg_len[e]
 * samples (clamped to L), gets frames_e = 1 + n / hop frames (center) or 1 + (n - frame) / hop (not centred), and rows
 * [0, frames_e) of out[e] equal sed_logmel_forward on the n samples alone, bit for bit.  Rows from frames_e on are
 * written as zeros.  A segment too short for the plan (n <= n_fft/2 under reflect padding, n < frame when not
 * centred, n = 0) has frames_e = 0.  L is only the row capacity: out is
 * [n_segments][sed_plan_num_frames(plan, L)][n_mels].  Arguments and errors otherwise as sed_logmel_forward_segments. */
int sed_logmel_forward_segments_own(const sed_plan* plan, const void* pcm, const int64_t* seg_off, const int32_t* seg_len,
                                    int64_t n_segments, int64_t L, float* out, int64_t out_stride_b, void* stream);

/* Unfused parity path: |STFT|^2 (or |STFT|), replaces spectrogram_extractor(x) at htsat.py:763.
 * out: fp32 [B][T][n_fft/2+1], clip b at out + b*out_stride_b. */
int sed_spectrogram_forward(const sed_plan* plan, const void* pcm, int64_t B, int64_t L, int64_t pcm_stride,
                            float* out, int64_t out_stride_b, void* stream);

/* mel projection + log of an already materialised spectrogram: replaces logmel_extractor(x) at htsat.py:764.
 * spec: fp32 [rows][n_fft/2+1] contiguous; out: fp32 [rows][n_mels]. */
int sed_mel_from_spectrogram(const sed_plan* plan, const float* spec, int64_t rows, float* out, void* stream);

/* In-place dB post-pass over `groups` contiguous groups of `group_elems` floats: x -= max(group) when
 * subtract_max, then x = max(x, max(group) - top_db) when top_db >= 0.  librosa.power_to_db(ref=np.max,
 * top_db=80) per clip (audio_to_spectrograms.py:43) and torchlibrosa's top_db over the whole tensor. */
int sed_db_normalize(float* x, int64_t groups, int64_t group_elems, int32_t subtract_max, float top_db,
                     void* stream);
/* (With fewer than 64 groups of more than 65536 floats each, sed_db_normalize cuts the groups into pieces and keeps their
 * partial maxima in a stream-ordered allocation, hipMallocAsync / hipFreeAsync on `stream`, for the two launches: the one
 * case in which an entry point of this header allocates at call time.)
 *
 * The same post-pass for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc): no counterpart in the reference.  It always runs the two-pass form and keeps the partial maxima in a
 * 64 KB array that the library holds on every GPU it runs on; it calls no allocating routine.  groups * pieces must
 * not exceed 16384, pieces = ceil(group_elems / 16384) (pieces of 32768, 65536, .. floats when that is more than 1024;
 * e.g. one group of 268 M floats, or 16 groups of 16 M), else SED_ERR_CAPACITY; groups <= 65535.  All calls on one GPU
 * share the array: calls that overlap in time on one GPU (different streams) race; calls of one stream do not.
 * The first call on a device looks the array's address up (hipGetSymbolAddress, a synchronous runtime call that may
 * initialise the code object): make one call per device BEFORE capturing a HIP graph that contains this entry point;
 * every later call is stream operations only.  x must live on the current device, else SED_ERR_WRONG_DEVICE. */
int sed_db_normalize_static(float* x, int64_t groups, int64_t group_elems, int32_t subtract_max, float top_db,
                            void* stream);

/* ---- decoders -------------------------------------------------------------------------------------- */

/* Smoothing along time + strict threshold + run extraction for every (clip, class) trace.
 * Replaces gaussian_filter1d + `> posterior_thresh` + find_contiguous_regions
 * (src/models/crnn/crnn.py:418-425, src/models/crnn/utils.py:77-89) and median_filter + `> threshold` + diff
 * (src/models/yamnet_train.py:57-66).
 * probs fp32 [B][T][C]; runs int32 [B][C][cap][2] = (first active frame, one past last active frame) in
 * ascending order; counts int32 [B][C] = number of runs found (may exceed cap: only the first cap are
 * stored, the caller checks).  GAUSS25 is evaluated in fp64 in scipy's summation order and rounded to
 * fp32 before the compare, so the activity equals scipy's bit for bit. */
int sed_threshold_runs(const float* probs, int64_t B, int64_t T, int64_t C, float threshold, int32_t smoothing,
                       int32_t* runs, int32_t* counts, int32_t cap, void* stream);

/* Same for an integer activity tensor (decode_predictions' input): a run starts where a[t]-a[t-1] == 1 and
 * ends where it == -1 (zero padded both ends), src/models/crnn/utils.py:83-87.  starts/ends are separate
 * lists with separate counts (the reference zips them, which truncates to the shorter).
 * act int32 [B][T][C]; starts/ends int32 [B][C][cap]; n_starts/n_ends int32 [B][C]. */
int sed_activity_edges(const int32_t* act, int64_t B, int64_t T, int64_t C, int32_t* starts, int32_t* ends,
                       int32_t* n_starts, int32_t* n_ends, int32_t cap, void* stream);

/* Smoothing + per-frame argmax over classes + run-length over labels.
 * Replaces get_events_from_batch (src/models/htsat/combined.py:26-71) and predict_class/predict_events
 * (src/models/single_stage_yamnet_frame.py:174-208).  probs fp32 [B][T][C];
 * runs int32 [B][cap][3] = (label, first frame, one past last frame), all labels including background;
 * counts int32 [B].  drop_last_classes > 0 ignores that many trailing classes in the argmax. */
int sed_argmax_runs(const float* probs, int64_t B, int64_t T, int64_t C, int32_t smoothing,
                    int32_t drop_last_classes, int32_t* runs, int32_t* counts, int32_t cap, void* stream);

/* Per-clip histogram of the per-frame argmax (trailing classes dropped): the majority vote of
 * get_classification_events (src/models/htsat/classification.py:20-25).  hist int32 [B][C - drop_last]. */
int sed_argmax_histogram(const float* probs, int64_t B, int64_t T, int64_t C, int32_t drop_last_classes,
                         int32_t* hist, void* stream);

/* Gap merge of onset-sorted event groups (one group = one (file, label) pair):
 * next.onset - current.offset <= gap folds next into current, offset = max.
 * Replaces the per-label loop of merge_close_events (src/utils/merge_events.py:36-55).
 * onsets/offsets fp64 [n_events]; group_start int32 [n_groups+1] (prefix offsets);
 * out_onsets/out_offsets fp64 [n_events] (merged events of group g at group_start[g] + i);
 * out_counts int32 [n_groups]. */
int sed_merge_events(const double* onsets, const double* offsets, const int32_t* group_start, int64_t n_groups,
                     double gap, double* out_onsets, double* out_offsets, int32_t* out_counts, void* stream);

/* Same merge on the frame-index runs written by sed_threshold_runs, with times looked up in a table
 * (time_table[k] = the host-formatted time of frame k, fp64 [table_len]); n_groups = B*C.
 * out_runs int32 [n_groups][cap][2]; out_counts int32 [n_groups]. */
int sed_merge_runs(const int32_t* runs, const int32_t* counts, int64_t n_groups, int32_t cap,
                   const double* time_table, int32_t table_len, double gap, int32_t* out_runs,
                   int32_t* out_counts, void* stream);

/* sed_threshold_runs followed by sed_merge_runs in one launch (the detector decode of yamnet_train.py:55-71, then
 * the gap merge of merge_events.py:36-55, per stream and class): packed int32 [B*C][1 + 2*cap] holds, per trace,
 * the merged count and then its (start, end) frame pairs.  cap >= T/2 + 1 (a trace of T frames has at most that many
 * runs) and T <= 8190 (the raw runs of a trace stay in LDS, sized per launch for T), else SED_ERR_CAPACITY. */
int sed_threshold_merge_runs(const float* probs, int64_t B, int64_t T, int64_t C, float threshold, int32_t smoothing,
                             const double* time_table, int32_t table_len, double gap, int32_t* packed, int32_t cap,
                             void* stream);

/* Event records of a stream batch for the exchange between GPUs (the per-file event lists the reference builds in
 * src/models/yamnet_train.py:274-283 and src/models/crnn/utils.py:99-134, as frame indices): for every stream, its
 * runs of all classes in class-major, onset-ascending order as int32 records (stream + first_stream, class, first
 * frame, end frame).  records int32 [n_streams][cap_out + 1][4]: row 0 of a stream = (records kept, records dropped
 * because they did not fit cap_out, 0, 0), rows 1.. = the records, unused rows zero.  The runs of trace
 * (s, c) start at runs + (s*C + c)*runs_stride ints (pairs, at most cap_in of them), its count is
 * counts[(s*C + c)*counts_stride]: (2*cap, 1) for the buffers of sed_threshold_runs / sed_merge_runs, and
 * runs = packed + 1, counts = packed, both strides 1 + 2*cap for the buffer of sed_threshold_merge_runs. */
int sed_pack_events(const int32_t* runs, int64_t runs_stride, const int32_t* counts, int64_t counts_stride,
                    int64_t n_streams, int64_t C, int32_t cap_in, int32_t first_stream, int32_t* records,
                    int32_t cap_out, void* stream);

/* sed_threshold_merge_runs followed by sed_pack_events in one launch, with the same records as the two calls give
 * (detector decode of src/models/yamnet_train.py:55-71, gap merge of src/utils/merge_events.py:36-55, record order of
 * src/models/crnn/utils.py:99-134): probs fp32 [B][T][C] -> records int32 [B][cap_out + 1][4] as sed_pack_events
 * lays them out.  T <= 8190 (the raw and merged runs of a trace stay in LDS, sized per launch for T: 862-frame CRNN
 * traces and 4096-frame HTS-AT framewise maps fit), else SED_ERR_CAPACITY. */
int sed_threshold_merge_records(const float* probs, int64_t B, int64_t T, int64_t C, float threshold, int32_t smoothing,
                                const double* time_table, int32_t table_len, double gap, int32_t first_stream,
                                int32_t* records, int32_t cap_out, void* stream);

/* ---- detector -> classifier hand-off ----------------------------------------------------------------- */

/* Cuts event e = samples [start, end) of stream ev_stream[e] out of the resident PCM batch into row e of a
 * zero-padded fp32 batch, without going through files.  Replaces cut_events_from_audio (src/utils/cut_audio.py:13-53:
 * sf.read, audio_array[int(start_sec*sr):int(end_sec*sr)], sf.write per event) as called at main.py:123-127,136-138.
 * ev_start/ev_end: the caller's int(sec*sr) values; they follow numpy's slice rules (negative counts from the end,
 * then both are clipped to [0, L]).  pcm: [n_streams] clips of L samples, fp32 or int16 (divided by pcm_divisor,
 * 32768 for soundfile's int16 decoding).  out: fp32 [n_events][cap]; out_len int32 [n_events] = slice length, of
 * which the first min(len, cap) samples are copied.  All pointers are device pointers. */
int sed_cut_events(const void* pcm, int32_t pcm_kind, float pcm_divisor, int64_t n_streams, int64_t L,
                   int64_t pcm_stride, const int32_t* ev_stream, const int64_t* ev_start, const int64_t* ev_end,
                   int64_t n_events, float* out, int64_t cap, int32_t* out_len, void* stream);

/* ---- HTS-AT glue behind the front-end ------------------------------------------------------------------ */

/* reshape_wav2img (external/hts_audio_transformer/model/htsat.py:729-744) for mel_bins == spec_size / freq_ratio:
 * bicubic interpolation (align_corners=True) of logmel fp32 [B][T][M] along time to target_T = spec_size*freq_ratio
 * frames (copied when T == target_T), then img fp32 [B][1][freq_ratio*M][target_T/freq_ratio] with
 * img[b][0][r*M + f][u] = y[b][r*(target_T/freq_ratio) + u][f].  T <= target_T, target_T % freq_ratio == 0. */
int sed_htsat_wav2img(const float* logmel, int64_t B, int64_t T, int64_t M, int64_t target_T, int32_t freq_ratio,
                      float* img, void* stream);

/* The same for clips with frame counts of their own (events of different lengths, each stretched to the image as the file
 * of that event would be, htsat.py:729-744 with x.shape[2] = frames[b]): logmel fp32 [B][T][M] holds frames[b] <= T rows
 * for clip b (frames: int32 DEVICE array [B]; values are clamped to [0, T]), and the interpolation of clip b runs over
 * those rows only.  A clip with no frame gives an image of zeros. */
int sed_htsat_wav2img_ragged(const float* logmel, const int32_t* frames, int64_t B, int64_t T, int64_t M, int64_t target_T,
                             int32_t freq_ratio, float* img, void* stream);

/* repeat_wat2img (external/hts_audio_transformer/model/htsat.py:747-760; the enable_repeat_mode branch :783-805):
 * the same interpolation to target_T frames, then the spec_size frames from cur_pos on, transposed and stacked
 * `repeats` times (4 in the reference): img fp32 [B][1][repeats*M][spec_size], img[b][0][r*M + f][u] = y[b][cur_pos + u][f].
 * 0 <= cur_pos, cur_pos + spec_size <= target_T. */
int sed_htsat_repeat2img(const float* logmel, int64_t B, int64_t T, int64_t M, int64_t target_T, int64_t spec_size,
                         int64_t cur_pos, int32_t repeats, float* img, void* stream);

/* interpolate(x, ratio) (external/hts_audio_transformer/utils.py:158-172; call at model/htsat.py:707) followed by
 * pad_framewise_output(., frames_out) (utils.py:175-192): x fp32 [B][T][C] -> out fp32 [B][frames_out][C] with
 * out[b][t][c] = x[b][min(t / ratio, T - 1)][c].  frames_out = T*ratio is interpolate alone; ratio = 1 is the padding
 * alone. */
int sed_framewise_expand(const float* x, int64_t B, int64_t T, int64_t C, int32_t ratio, int64_t frames_out,
                         float* out, void* stream);

/* ---- sample-rate conversion ahead of the front-end ------------------------------------------------------ */

/* Polyphase resampling with the arithmetic of scipy.signal.resample_poly(x, target_sr/g, orig_sr/g), which is what
 * librosa.resample(..., res_type='polyphase') runs: same Kaiser(5.0) filter bits, same fp32 accumulation order, so the
 * output equals scipy's bit for bit on fp32 input.  The reference resamples with librosa's default res_type
 * (librosa.load(file_path, sr=16000), src/models/yamnet_train.py:46 and src/models/single_stage_yamnet_frame.py:25-27;
 * librosa.resample(track, ofs, rfs), external/hts_audio_transformer/utils.py:211), i.e. soxr_hq, whose filter lives in
 * libsoxr: this entry point replaces those calls only when they are given res_type='polyphase'.
 * Any pair of rates with max(up, down) <= 2^20 (up = target_sr/g, down = orig_sr/g): pairs with up <= 512 and at most 128
 * filter coefficients per phase (orig_sr/target_sr up to about 6), or up <= 1024 and at most 64 of them, run the tiled
 * LDS kernels (every pair of the usual audio rates from 8 to 48 kHz but 11.025 -> 32 kHz), the others a general kernel with the same bits at a fraction of the
 * speed. */
typedef struct sed_resampler sed_resampler;
int sed_resampler_create(int32_t orig_sr, int32_t target_sr, sed_resampler** out);
void sed_resampler_destroy(sed_resampler* r);
/* ceil(n_in * up / down), the length scipy returns. */
int64_t sed_resampler_out_len(const sed_resampler* r, int64_t n_in);
/* up, down, coefficients per phase, the number of leading outputs scipy drops, and (if table_host != NULL) the fp32
 * coefficient table [up][taps] with table[phase][j] = zero-padded filter[phase + up*j].  Any pointer may be NULL. */
int sed_resampler_get_filter(const sed_resampler* r, int32_t* up, int32_t* down, int32_t* taps, int32_t* n_pre_remove,
                             float* table_host);
/* pcm [B][pcm_stride] fp32 or int16 (SED_PCM_*; int16 samples are divided by pcm_divisor first) -> out fp32
 * [B][out_stride], sed_resampler_out_len(L) samples per row.  Asynchronous on `stream`. */
int sed_resampler_forward(const sed_resampler* r, const void* pcm, int32_t pcm_kind, float pcm_divisor, int64_t B,
                          int64_t L, int64_t pcm_stride, float* out, int64_t out_stride, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SED_HIP_H_ */
