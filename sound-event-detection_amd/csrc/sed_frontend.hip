// Fused log-mel front-end for gfx950 (MI355X): framing + Hann window + real FFT + power/magnitude +
// mel projection + log, one launch per batch, the (T, F) spectrogram never leaves the CU.
//
// Work decomposition
//   workgroup  : WAVES wavefronts; owns a contiguous range of "tiles" (FPI consecutive frames of one clip),
//                loops over them (persistent), tables live in LDS for the whole launch
//   per tile   : the PCM span of the tile is staged once in LDS (every sample read once from HBM per tile,
//                padding modes and int16 conversion resolved here), then each wavefront runs FPW frames
//   per frame  : LPF lanes; see fft_core.h for the lane decomposition
//   per wave   : exchange, power rows and mel rows use a wavefront-private LDS region, so the only
//                workgroup barriers are the two around staging
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>

#include "sed_hip.h"
#include "tables.h"

namespace sed {

using Geo512 = Geo<16, 16, 8, 4>;
using Geo1024 = Geo<32, 16, 16, 4>;
using Geo2048 = Geo<32, 32, 16, 2>;

struct FrontParams {
  const void* pcm;
  float* out;
  int64_t L, pcm_stride, out_stride_b;
  int32_t T, hop, pad_left, pad_mode, pcm_kind, vec_ok;
  float pcm_div;
  int32_t n_mels, log_kind, affine;
  float amin, db_offset, ln_offset;
  int32_t tiles_per_clip, total_tiles, tiles_per_block;
  // LDS carve, in floats
  int32_t off_xw, xwf, off_win, off_tw1, off_tw2, off_meta, off_melw;
  int32_t mel_slots, mel_wcount;
  const float* window;
  const cplx* tw1;
  const cplx* tw2;
  const MelEntry* meta;
  const float* melw;
  const float* aff_scale;
  const float* aff_shift;
};

// Orders LDS accesses of different lanes of ONE wavefront (the hardware executes a wave's LDS operations in
// order; this only stops the compiler from moving them across the phase boundary).
__device__ __forceinline__ void wave_phase_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float log_epilogue(float acc, int log_kind, float amin, float db_offset, float ln_offset) {
  if (log_kind == SED_LOG_DB10) return 10.0f * log10f(fmaxf(acc, amin)) - db_offset;
  if (log_kind == SED_LOG_LN) return logf(acc + ln_offset);
  return acc;
}

// Sample `s` of the virtual padded clip: reflect (index -j -> j, L-1+j -> L-1-j) or zeros outside [0, L).
template <typename T>
__device__ __forceinline__ float load_padded(const T* src, int64_t s, int64_t L, int pad_mode, float div) {
  if (s < 0) s = (pad_mode == SED_PAD_REFLECT) ? -s : -1;
  else if (s >= L) s = (pad_mode == SED_PAD_REFLECT) ? 2 * (L - 1) - s : -1;
  if (s < 0 || s >= L) return 0.0f;
  if constexpr (std::is_same<T, float>::value) return src[s];
  else return (float)src[s] / div;
}

template <class G>
__device__ __forceinline__ void stage_tile(const FrontParams& p, float* stage, int b, int t0, int nf, int tid) {
  constexpr int NT = G::WAVES * 64;
  const int span = (nf - 1) * p.hop + G::NFFT;
  const int64_t s0 = (int64_t)t0 * p.hop - p.pad_left;
  const bool interior = s0 >= 0 && s0 + span <= p.L;
  if (p.pcm_kind == SED_PCM_F32) {
    const float* src = (const float*)p.pcm + (int64_t)b * p.pcm_stride;
    if (interior && p.vec_ok && (s0 & 3) == 0) {
      const float4* s4 = (const float4*)(src + s0);
      float4* d4 = (float4*)stage;
      for (int i = tid; i < span / 4; i += NT) d4[i] = s4[i];
    } else {
      for (int i = tid; i < span; i += NT) stage[i] = load_padded<float>(src, s0 + i, p.L, p.pad_mode, 1.0f);
    }
  } else {
    const int16_t* src = (const int16_t*)p.pcm + (int64_t)b * p.pcm_stride;
    if (interior && p.vec_ok && (s0 & 3) == 0) {
      const short4* s4 = (const short4*)(src + s0);
      float4* d4 = (float4*)stage;
      for (int i = tid; i < span / 4; i += NT) {
        const short4 v = s4[i];
        d4[i] = make_float4((float)v.x / p.pcm_div, (float)v.y / p.pcm_div, (float)v.z / p.pcm_div,
                            (float)v.w / p.pcm_div);
      }
    } else {
      for (int i = tid; i < span; i += NT) stage[i] = load_padded<int16_t>(src, s0 + i, p.L, p.pad_mode, p.pcm_div);
    }
  }
}

// Mel filters of one frame for lane l of its LPF-lane group: banded dot products from the power row.
__device__ __forceinline__ void mel_project_lane(const float* pw, const MelEntry* meta, const float* melw, int slots,
                                                 int l, float* melrow, const FrontParams& p) {
  for (int j = 0; j < slots; ++j) {
    const MelEntry e = meta[l * slots + j];
    if (e.mel < 0) continue;
    const float* x = pw + e.start;
    const float* w = melw + e.woff;
    float acc = 0.0f;
    for (int i = 0; i < e.len; ++i) acc = __builtin_fmaf(x[i], w[i], acc);
    float v = log_epilogue(acc, p.log_kind, p.amin, p.db_offset, p.ln_offset);
    if (p.affine) v = __builtin_fmaf(v, p.aff_scale[e.mel], p.aff_shift[e.mel]);
    melrow[e.mel] = v;
  }
}

template <class G, bool MAG, bool SPEC_OUT>
__global__ void __launch_bounds__(G::WAVES * 64, G::WAVES == 4 ? 2 : 1) stft_mel_kernel(const FrontParams p) {
  extern __shared__ __align__(16) float smem[];
  constexpr int NT = G::WAVES * 64;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int fi = lane / G::LPF, l = lane % G::LPF;

  float* stage = smem;
  float* xw = smem + p.off_xw + wave * p.xwf;
  float* winL = smem + p.off_win;
  cplx* tw1L = (cplx*)(smem + p.off_tw1);
  cplx* tw2L = (cplx*)(smem + p.off_tw2);
  MelEntry* metaL = (MelEntry*)(smem + p.off_meta);
  float* melwL = smem + p.off_melw;

  for (int i = tid; i < G::NFFT; i += NT) winL[i] = p.window[i];
  for (int i = tid; i < G::R1 * G::R2; i += NT) tw1L[i] = p.tw1[i];
  for (int i = tid; i < G::SLOTS * G::LPF; i += NT) tw2L[i] = p.tw2[i];
  if constexpr (!SPEC_OUT) {
    for (int i = tid; i < G::LPF * p.mel_slots; i += NT) metaL[i] = p.meta[i];
    for (int i = tid; i < p.mel_wcount; i += NT) melwL[i] = p.melw[i];
  }
  __syncthreads();

  const int first = blockIdx.x * p.tiles_per_block;
  const int last = min(first + p.tiles_per_block, p.total_tiles);
  for (int tile = first; tile < last; ++tile) {
    const int b = tile / p.tiles_per_clip;
    const int t0 = (tile - b * p.tiles_per_clip) * G::FPI;
    const int nf = min(G::FPI, p.T - t0);
    stage_tile<G>(p, stage, b, t0, nf, tid);
    __syncthreads();
    const int wf0 = wave * G::FPW;  // first frame of this wave inside the tile
    if (wf0 < nf) {
      LaneState<G> st;
      lane_stage1<G>(st, stage + (wf0 + fi) * p.hop, winL, tw1L, l);
      float* xfr = xw + fi * G::XFR;
      lane_xwrite<G, 0>(st, xfr, l);
      wave_phase_fence();
      lane_xread<G, 0>(st, xfr, l);
      wave_phase_fence();
      lane_xwrite<G, 1>(st, xfr, l);
      wave_phase_fence();
      lane_xread<G, 1>(st, xfr, l);
      wave_phase_fence();
      lane_stage2_power<G, MAG>(st, xw + fi * G::PP, tw2L, l);
      wave_phase_fence();
      const int nvalid = min(G::FPW, nf - wf0);
      if constexpr (SPEC_OUT) {
        float* dst = p.out + (int64_t)b * p.out_stride_b + (int64_t)(t0 + wf0) * G::NBINS;
        for (int f = 0; f < nvalid; ++f)
          for (int k = lane; k < G::NBINS; k += 64) dst[(int64_t)f * G::NBINS + k] = xw[f * G::PP + k];
      } else {
        float* melbuf = xw + G::FPW * G::PP;
        mel_project_lane(xw + fi * G::PP, metaL, melwL, p.mel_slots, l, melbuf + fi * p.n_mels, p);
        wave_phase_fence();
        float* dst = p.out + (int64_t)b * p.out_stride_b + (int64_t)(t0 + wf0) * p.n_mels;
        const int n = nvalid * p.n_mels;
        for (int i = lane; i < n; i += 64) dst[i] = melbuf[i];
      }
    }
    __syncthreads();
  }
}

// Mel projection + log of a materialised spectrogram: 16 lanes per row, 16 rows per workgroup pass.
__global__ void __launch_bounds__(256) mel_rows_kernel(const FrontParams p, const float* spec, int64_t rows, int nbins) {
  extern __shared__ __align__(16) float smem[];
  const int tid = threadIdx.x;
  const int pitch = nbins + 3;
  float* rowsL = smem;                                    // 16 rows
  float* melbuf = smem + 16 * pitch;                      // 16 x n_mels
  MelEntry* metaL = (MelEntry*)(melbuf + 16 * p.n_mels);
  float* melwL = (float*)(metaL + 16 * p.mel_slots);
  for (int i = tid; i < 16 * p.mel_slots; i += 256) metaL[i] = p.meta[i];
  for (int i = tid; i < p.mel_wcount; i += 256) melwL[i] = p.melw[i];
  for (int64_t r0 = (int64_t)blockIdx.x * 16; r0 < rows; r0 += (int64_t)gridDim.x * 16) {
    const int nr = (int)min((int64_t)16, rows - r0);
    __syncthreads();
    for (int i = tid; i < nr * nbins; i += 256) {
      const int r = i / nbins, k = i - r * nbins;
      rowsL[r * pitch + k] = spec[(r0 + r) * nbins + k];
    }
    __syncthreads();
    const int r = tid >> 4, l = tid & 15;
    if (r < nr) mel_project_lane(rowsL + r * pitch, metaL, melwL, p.mel_slots, l, melbuf + r * p.n_mels, p);
    __syncthreads();
    for (int i = tid; i < nr * p.n_mels; i += 256) p.out[r0 * p.n_mels + i] = melbuf[i];
  }
}

// dB post-pass: one workgroup per group; max reduction then subtract / floor.
__global__ void __launch_bounds__(256) db_normalize_kernel(float* x, int64_t group_elems, int subtract_max, float top_db) {
  __shared__ float red[4];
  float* g = x + (int64_t)blockIdx.x * group_elems;
  float m = -INFINITY;
  for (int64_t i = threadIdx.x; i < group_elems; i += 256) m = fmaxf(m, g[i]);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sub = subtract_max ? m : 0.0f;
  const float floor_v = (top_db >= 0.0f) ? (m - sub) - top_db : -INFINITY;
  for (int64_t i = threadIdx.x; i < group_elems; i += 256) g[i] = fmaxf(g[i] - sub, floor_v);
}

}  // namespace sed

// ======================================================================================================
// host side
// ======================================================================================================

using namespace sed;

struct sed_plan {
  sed_frontend_cfg cfg;
  int device = 0;
  int num_cus = 256;
  int geo = 1;  // 0: 512, 1: 1024, 2: 2048
  std::vector<double> window_host;
  std::vector<float> melW_host;  // [F][M]
  float* d_window = nullptr;     // 0.5 * window, fp32
  cplx* d_tw1 = nullptr;
  cplx* d_tw2 = nullptr;
  MelEntry* d_meta = nullptr;    // schedule for the fused kernel (LPF lanes)
  float* d_melw = nullptr;
  MelEntry* d_meta16 = nullptr;  // schedule for mel_rows_kernel (16 lanes)
  float* d_melw16 = nullptr;
  float* d_scale = nullptr;
  float* d_shift = nullptr;
  int mel_slots = 0, mel_wcount = 0, mel_slots16 = 0, mel_wcount16 = 0;
  bool affine = false;
  float db_offset = 0.0f;
};

static inline int hip_status(hipError_t e) { return e == hipSuccess ? SED_OK : SED_ERR_HIP - (int)e; }
#define SED_HIP_TRY(expr)                         \
  do {                                            \
    hipError_t e_ = (expr);                       \
    if (e_ != hipSuccess) return hip_status(e_);  \
  } while (0)

template <typename T>
static int upload(T** dst, const std::vector<T>& src) {
  if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
  SED_HIP_TRY(hipMalloc((void**)dst, std::max<size_t>(1, src.size()) * sizeof(T)));
  if (!src.empty()) SED_HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return SED_OK;
}

static int lpf_of(int geo) { return geo == 0 ? Geo512::LPF : Geo1024::LPF; }

static int upload_mel(sed_plan* pl) {
  const int F = pl->cfg.n_fft / 2 + 1, M = pl->cfg.n_mels;
  MelSchedule s = schedule_mel(pl->melW_host, F, M, lpf_of(pl->geo));
  pl->mel_slots = s.slots;
  pl->mel_wcount = (int)s.weights.size();
  int rc = upload(&pl->d_meta, s.entries);
  if (rc) return rc;
  rc = upload(&pl->d_melw, s.weights);
  if (rc) return rc;
  MelSchedule s16 = schedule_mel(pl->melW_host, F, M, 16);
  pl->mel_slots16 = s16.slots;
  pl->mel_wcount16 = (int)s16.weights.size();
  rc = upload(&pl->d_meta16, s16.entries);
  if (rc) return rc;
  return upload(&pl->d_melw16, s16.weights);
}

extern "C" const char* sed_version(void) { return "sedhip 0.1.0 (gfx950)"; }

extern "C" const char* sed_status_string(int status) {
  switch (status) {
    case SED_OK: return "ok";
    case SED_ERR_BAD_ARG: return "bad argument";
    case SED_ERR_UNSUPPORTED_NFFT: return "unsupported n_fft (512, 1024 and 2048 are built)";
    case SED_ERR_CAPACITY: return "output capacity exceeded";
    case SED_ERR_SHORT_INPUT: return "input too short for reflect padding (needs n_samples > n_fft/2)";
    case SED_ERR_WRONG_DEVICE: return "current device is not the plan's device";
    default: break;
  }
  if (status <= SED_ERR_HIP) return hipGetErrorString((hipError_t)(SED_ERR_HIP - status));
  return "unknown status";
}

extern "C" int sed_mel_filterbank(int32_t mel_kind, int32_t sample_rate, int32_t n_fft, int32_t n_mels, float fmin,
                                  float fmax, float* melW) {
  if (!melW || sample_rate < 1 || n_fft < 2 || n_mels < 1 || !(fmax > fmin) || fmin < 0.0f) return SED_ERR_BAD_ARG;
  const std::vector<float> w = (mel_kind == SED_MEL_HTK_TF) ? mel_htk_tf(sample_rate, n_fft, n_mels, fmin, fmax)
                                                            : mel_slaney(sample_rate, n_fft, n_mels, fmin, fmax);
  std::memcpy(melW, w.data(), w.size() * sizeof(float));
  return SED_OK;
}

extern "C" int sed_hann_window(int32_t n_fft, int32_t win_length, int32_t win_align, float* window) {
  if (!window || win_length < 1 || win_length > n_fft) return SED_ERR_BAD_ARG;
  const std::vector<double> w = build_window(n_fft, win_length, win_align == SED_WIN_LEFT);
  for (int i = 0; i < n_fft; ++i) window[i] = (float)w[i];
  return SED_OK;
}

extern "C" int sed_plan_create(const sed_frontend_cfg* cfg, sed_plan** out) {
  if (!cfg || !out) return SED_ERR_BAD_ARG;
  *out = nullptr;
  int geo;
  if (cfg->n_fft == 512) geo = 0;
  else if (cfg->n_fft == 1024) geo = 1;
  else if (cfg->n_fft == 2048) geo = 2;
  else return SED_ERR_UNSUPPORTED_NFFT;
  if (cfg->win_length < 1 || cfg->win_length > cfg->n_fft || cfg->hop < 4 || cfg->hop % 4 != 0) return SED_ERR_BAD_ARG;
  if (cfg->n_mels < 1 || cfg->n_mels > 128 || cfg->sample_rate < 1) return SED_ERR_BAD_ARG;
  if (cfg->spectrum != SED_SPEC_MAGNITUDE && cfg->spectrum != SED_SPEC_POWER) return SED_ERR_BAD_ARG;
  if (cfg->center && cfg->pad_mode != SED_PAD_CONSTANT && cfg->pad_mode != SED_PAD_REFLECT) return SED_ERR_BAD_ARG;
  if (cfg->pcm_kind == SED_PCM_I16 && !(cfg->pcm_divisor > 0.0f)) return SED_ERR_BAD_ARG;
  if (cfg->tail_min_samples > 0 && cfg->tail_hop_samples < 1) return SED_ERR_BAD_ARG;
  if (!(cfg->fmax > cfg->fmin) || cfg->fmin < 0.0f) return SED_ERR_BAD_ARG;

  sed_plan* pl = new (std::nothrow) sed_plan();
  if (!pl) return SED_ERR_BAD_ARG;
  pl->cfg = *cfg;
  pl->geo = geo;
  int rc = hip_status(hipGetDevice(&pl->device));
  hipDeviceProp_t prop;
  if (!rc) rc = hip_status(hipGetDeviceProperties(&prop, pl->device));
  if (rc) { delete pl; return rc; }
  pl->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

  pl->window_host = build_window(cfg->n_fft, cfg->win_length, cfg->win_align == SED_WIN_LEFT);
  std::vector<float> wh(cfg->n_fft);
  for (int i = 0; i < cfg->n_fft; ++i) wh[i] = (float)(0.5 * pl->window_host[i]);
  pl->melW_host = (cfg->mel_kind == SED_MEL_HTK_TF)
                      ? mel_htk_tf(cfg->sample_rate, cfg->n_fft, cfg->n_mels, cfg->fmin, cfg->fmax)
                      : mel_slaney(cfg->sample_rate, cfg->n_fft, cfg->n_mels, cfg->fmin, cfg->fmax);
  pl->db_offset = (float)(10.0 * std::log10(std::max((double)cfg->amin, (double)cfg->ref)));

  rc = upload(&pl->d_window, wh);
  if (!rc) rc = geo == 0 ? upload(&pl->d_tw1, build_tw1<Geo512>())
                : geo == 1 ? upload(&pl->d_tw1, build_tw1<Geo1024>()) : upload(&pl->d_tw1, build_tw1<Geo2048>());
  if (!rc) rc = geo == 0 ? upload(&pl->d_tw2, build_tw2<Geo512>())
                : geo == 1 ? upload(&pl->d_tw2, build_tw2<Geo1024>()) : upload(&pl->d_tw2, build_tw2<Geo2048>());
  if (!rc) rc = upload_mel(pl);
  if (rc) { sed_plan_destroy(pl); return rc; }
  *out = pl;
  return SED_OK;
}

extern "C" void sed_plan_destroy(sed_plan* pl) {
  if (!pl) return;
  void* bufs[] = {pl->d_window, pl->d_tw1, pl->d_tw2, pl->d_meta, pl->d_melw, pl->d_meta16, pl->d_melw16,
                  pl->d_scale, pl->d_shift};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  delete pl;
}

extern "C" int64_t sed_plan_num_frames(const sed_plan* pl, int64_t n) {
  if (!pl || n < 0) return SED_ERR_BAD_ARG;
  const sed_frontend_cfg& c = pl->cfg;
  if (c.tail_min_samples > 0) {
    const int64_t m = std::max<int64_t>(n, c.tail_min_samples);
    const int64_t hops = (m - c.tail_min_samples + c.tail_hop_samples - 1) / c.tail_hop_samples;
    n = c.tail_min_samples + hops * c.tail_hop_samples;
  }
  if (c.center) {
    if (c.pad_mode == SED_PAD_REFLECT && n <= c.n_fft / 2) return SED_ERR_SHORT_INPUT;
    return 1 + n / c.hop;
  }
  const int64_t frame = (c.win_align == SED_WIN_LEFT) ? c.win_length : c.n_fft;
  if (n < frame) return 0;
  return 1 + (n - frame) / c.hop;
}

extern "C" int sed_plan_get_mel_weights(const sed_plan* pl, float* melW) {
  if (!pl || !melW) return SED_ERR_BAD_ARG;
  std::memcpy(melW, pl->melW_host.data(), pl->melW_host.size() * sizeof(float));
  return SED_OK;
}

extern "C" int sed_plan_get_window(const sed_plan* pl, float* w) {
  if (!pl || !w) return SED_ERR_BAD_ARG;
  for (int i = 0; i < pl->cfg.n_fft; ++i) w[i] = (float)pl->window_host[i];
  return SED_OK;
}

static int check_device(const sed_plan* pl) {
  int dev = -1;
  SED_HIP_TRY(hipGetDevice(&dev));
  return dev == pl->device ? SED_OK : SED_ERR_WRONG_DEVICE;
}

extern "C" int sed_plan_set_mel_weights(sed_plan* pl, const float* melW) {
  if (!pl || !melW) return SED_ERR_BAD_ARG;
  int rc = check_device(pl);
  if (rc) return rc;
  SED_HIP_TRY(hipDeviceSynchronize());
  std::memcpy(pl->melW_host.data(), melW, pl->melW_host.size() * sizeof(float));
  return upload_mel(pl);
}

extern "C" int sed_plan_set_affine(sed_plan* pl, const float* scale, const float* shift) {
  if (!pl || ((scale == nullptr) != (shift == nullptr))) return SED_ERR_BAD_ARG;
  int rc = check_device(pl);
  if (rc) return rc;
  SED_HIP_TRY(hipDeviceSynchronize());
  pl->affine = scale != nullptr;
  if (!scale) return SED_OK;
  const int M = pl->cfg.n_mels;
  rc = upload(&pl->d_scale, std::vector<float>(scale, scale + M));
  if (rc) return rc;
  return upload(&pl->d_shift, std::vector<float>(shift, shift + M));
}

static void fill_common(const sed_plan* pl, FrontParams& p) {
  const sed_frontend_cfg& c = pl->cfg;
  p.hop = c.hop;
  p.pad_left = c.center ? c.n_fft / 2 : 0;
  p.pad_mode = c.center ? c.pad_mode : SED_PAD_NONE;
  p.pcm_kind = c.pcm_kind;
  p.pcm_div = c.pcm_kind == SED_PCM_I16 ? c.pcm_divisor : 1.0f;
  p.n_mels = c.n_mels;
  p.log_kind = c.log_kind;
  p.affine = pl->affine ? 1 : 0;
  p.amin = c.amin;
  p.db_offset = pl->db_offset;
  p.ln_offset = c.ln_offset;
  p.window = pl->d_window;
  p.tw1 = pl->d_tw1;
  p.tw2 = pl->d_tw2;
  p.meta = pl->d_meta;
  p.melw = pl->d_melw;
  p.mel_slots = pl->mel_slots;
  p.mel_wcount = pl->mel_wcount;
  p.aff_scale = pl->d_scale;
  p.aff_shift = pl->d_shift;
}

static inline int round4(int v) { return (v + 3) & ~3; }

template <class G, bool MAG, bool SPEC_OUT>
static int launch_front(const sed_plan* pl, FrontParams p, hipStream_t stream) {
  const int span = round4((G::FPI - 1) * p.hop + G::NFFT);
  int off = span;
  p.off_xw = off;
  p.xwf = round4(std::max(G::FPW * G::XFR, G::FPW * G::PP + (SPEC_OUT ? 0 : G::FPW * p.n_mels)));
  off += G::WAVES * p.xwf;
  p.off_win = off; off += G::NFFT;
  p.off_tw1 = off; off += 2 * G::R1 * G::R2;
  p.off_tw2 = off; off += round4(2 * G::SLOTS * G::LPF);
  p.off_meta = off; off += SPEC_OUT ? 0 : 4 * G::LPF * p.mel_slots;
  p.off_melw = off; off += SPEC_OUT ? 0 : round4(p.mel_wcount);
  const size_t lds_bytes = (size_t)off * sizeof(float);
  if (lds_bytes > 160 * 1024) return SED_ERR_BAD_ARG;
  auto kern = stft_mel_kernel<G, MAG, SPEC_OUT>;
  SED_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds_bytes));
  const int per_cu = std::max(1, std::min(8, (int)((160 * 1024) / lds_bytes)));
  const int grid = std::min(p.total_tiles, pl->num_cus * per_cu);
  p.tiles_per_block = (p.total_tiles + grid - 1) / grid;
  const int blocks = (p.total_tiles + p.tiles_per_block - 1) / p.tiles_per_block;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(G::WAVES * 64), lds_bytes, stream, p);
  return hip_status(hipGetLastError());
}

template <bool SPEC_OUT>
static int front_forward(const sed_plan* pl, const void* pcm, int64_t B, int64_t L, int64_t pcm_stride, float* out,
                         int64_t out_stride_b, void* stream) {
  if (!pl || B < 0 || L < 0) return SED_ERR_BAD_ARG;
  const int64_t T = sed_plan_num_frames(pl, L);
  if (T < 0) return (int)T;
  if (B == 0 || T == 0) return SED_OK;
  if (!pcm || !out || pcm_stride < L) return SED_ERR_BAD_ARG;
  const int64_t row = SPEC_OUT ? pl->cfg.n_fft / 2 + 1 : pl->cfg.n_mels;
  if (out_stride_b < T * row) return SED_ERR_BAD_ARG;
  int rc = check_device(pl);
  if (rc) return rc;
  FrontParams p{};
  fill_common(pl, p);
  p.pcm = pcm;
  p.out = out;
  p.L = L;
  p.pcm_stride = pcm_stride;
  p.out_stride_b = out_stride_b;
  p.T = (int32_t)T;
  const int fpi = pl->geo == 0 ? Geo512::FPI : pl->geo == 1 ? Geo1024::FPI : Geo2048::FPI;
  p.tiles_per_clip = (int32_t)((T + fpi - 1) / fpi);
  const int64_t total = (int64_t)p.tiles_per_clip * B;
  if (T > INT32_MAX / 2 || total > INT32_MAX / 2) return SED_ERR_BAD_ARG;
  p.total_tiles = (int32_t)total;
  const size_t esz = pl->cfg.pcm_kind == SED_PCM_I16 ? 2 : 4;
  const size_t need = pl->cfg.pcm_kind == SED_PCM_I16 ? 8 : 16;
  p.vec_ok = ((uintptr_t)pcm % need == 0) && ((pcm_stride * esz) % need == 0) ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  const bool mag = pl->cfg.spectrum == SED_SPEC_MAGNITUDE;
#define SED_DISPATCH(GEO)                                                   \
  return mag ? launch_front<GEO, true, SPEC_OUT>(pl, p, s) : launch_front<GEO, false, SPEC_OUT>(pl, p, s)
  if (pl->geo == 0) { SED_DISPATCH(Geo512); }
  if (pl->geo == 1) { SED_DISPATCH(Geo1024); }
  SED_DISPATCH(Geo2048);
#undef SED_DISPATCH
}

extern "C" int sed_logmel_forward(const sed_plan* pl, const void* pcm, int64_t B, int64_t L, int64_t pcm_stride,
                                  float* out, int64_t out_stride_b, void* stream) {
  return front_forward<false>(pl, pcm, B, L, pcm_stride, out, out_stride_b, stream);
}

extern "C" int sed_spectrogram_forward(const sed_plan* pl, const void* pcm, int64_t B, int64_t L, int64_t pcm_stride,
                                       float* out, int64_t out_stride_b, void* stream) {
  return front_forward<true>(pl, pcm, B, L, pcm_stride, out, out_stride_b, stream);
}

extern "C" int sed_mel_from_spectrogram(const sed_plan* pl, const float* spec, int64_t rows, float* out, void* stream) {
  if (!pl || rows < 0) return SED_ERR_BAD_ARG;
  if (rows == 0) return SED_OK;
  if (!spec || !out) return SED_ERR_BAD_ARG;
  int rc = check_device(pl);
  if (rc) return rc;
  FrontParams p{};
  fill_common(pl, p);
  p.out = out;
  p.meta = pl->d_meta16;
  p.melw = pl->d_melw16;
  p.mel_slots = pl->mel_slots16;
  p.mel_wcount = pl->mel_wcount16;
  const int nbins = pl->cfg.n_fft / 2 + 1;
  const size_t lds = sizeof(float) * (16 * (nbins + 3) + 16 * p.n_mels + 4 * 16 * p.mel_slots + p.mel_wcount);
  if (lds > 160 * 1024) return SED_ERR_BAD_ARG;
  SED_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mel_rows_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t groups = (rows + 15) / 16;
  const int grid = (int)std::min<int64_t>(groups, (int64_t)pl->num_cus * 2);
  hipLaunchKernelGGL(mel_rows_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, p, spec, rows, nbins);
  return hip_status(hipGetLastError());
}

extern "C" int sed_db_normalize(float* x, int64_t groups, int64_t group_elems, int32_t subtract_max, float top_db,
                                void* stream) {
  if (groups < 0 || group_elems < 0 || groups > INT32_MAX) return SED_ERR_BAD_ARG;
  if (groups == 0 || group_elems == 0) return SED_OK;
  if (!x) return SED_ERR_BAD_ARG;
  hipLaunchKernelGGL(db_normalize_kernel, dim3((unsigned)groups), dim3(256), 0, (hipStream_t)stream, x, group_elems,
                     (int)subtract_max, top_db);
  return hip_status(hipGetLastError());
}
