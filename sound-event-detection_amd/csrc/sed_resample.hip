// Polyphase sample-rate conversion for gfx950 with the arithmetic of scipy.signal.resample_poly (the implementation
// behind librosa.resample(res_type='polyphase')): Kaiser(5.0) low-pass of 20*max(up, down)+1 taps designed in fp64,
// cast to fp32 and scaled by `up` in fp32; every output sample accumulated in fp32 from the oldest contributing input
// sample to the newest with separately rounded multiplies and adds.  The results equal scipy's bit for bit.
//
// Work decomposition: a workgroup of up*P threads produces up*P consecutive output samples per tile; since the tile
// length is a multiple of `up`, a thread meets the same filter phase in every tile and keeps that phase's
// coefficients in registers for the whole launch.  The input samples a tile needs (P*down + taps, contiguous) are
// staged once in LDS.  Bound: HBM (read L, write L*up/down samples per clip) once the LDS reads are hidden.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <numeric>
#include <vector>

#pragma clang fp contract(off)

#include "sed_hip.h"

namespace sed {

struct ResampleParams {
  const void* pcm;
  float* out;
  const float* table;  // [up][TP]: table[phase][j] = padded filter[phase + up*j], zero beyond the filter
  int64_t B, L, pcm_stride, out_stride, n_out, tiles;
  int32_t up, down, n_pre_remove, qstep, stage_count;
  float div;
};

template <int TP, typename T>
__global__ void __launch_bounds__(TP > 64 ? 512 : 1024) resample_kernel(const ResampleParams p) {
  extern __shared__ float xs[];
  const int t = threadIdx.x, NT = blockDim.x;
  const int64_t tt = ((int64_t)t + p.n_pre_remove) * p.down;
  const int phase = (int)(tt % p.up);
  const int64_t q_t = tt / p.up;                                          // newest input sample of output t of tile 0
  const int64_t q_0 = ((int64_t)p.n_pre_remove * p.down) / p.up;          // the same for thread 0
  float c[TP];
#pragma unroll
  for (int j = 0; j < TP; ++j) c[j] = p.table[(int64_t)phase * TP + j];
  const int base = (int)(q_t - q_0) + TP - 1;                             // position of q_t in the staged span
  for (int64_t work = blockIdx.x; work < p.B * p.tiles; work += gridDim.x) {
    const int64_t b = work / p.tiles, tile = work - b * p.tiles;
    const int64_t k_lo = q_0 + tile * p.qstep - (TP - 1);
    const T* src = (const T*)p.pcm + b * p.pcm_stride;
    for (int i = t; i < p.stage_count; i += NT) {
      const int64_t k = k_lo + i;
      float v = 0.0f;
      if (k >= 0 && k < p.L) {
        if constexpr (std::is_same<T, float>::value) v = src[k];
        else v = (float)src[k] / p.div;
      }
      xs[i] = v;
    }
    __syncthreads();
    const int64_t m = tile * NT + t;
    if (m < p.n_out) {
      float acc = 0.0f;
#pragma unroll
      for (int j = TP - 1; j >= 0; --j) acc = __fadd_rn(acc, __fmul_rn(xs[base - j], c[j]));
      p.out[b * p.out_stride + m] = acc;
    }
    __syncthreads();
  }
}

static double bessel_i0(double x) {
  const double q = 0.25 * x * x;
  double term = 1.0, total = 1.0;
  for (int k = 1; k < 60; ++k) {
    term = term * q / ((double)k * (double)k);
    total += term;
  }
  return total;
}

static double pairwise_sum(const double* a, size_t n) {  // numpy's summation order matters only in the last fp64 bits
  if (n <= 8) {
    double s = 0.0;
    for (size_t i = 0; i < n; ++i) s += a[i];
    return s;
  }
  const size_t h = n / 2;
  return pairwise_sum(a, h) + pairwise_sum(a + h, n - h);
}

}  // namespace sed

using namespace sed;

struct sed_resampler {
  int32_t up = 1, down = 1, taps = 1, tp = 32, n_pre_remove = 0;
  int device = 0, num_cus = 256;
  std::vector<float> table_host;  // [up][taps]
  float* d_table = nullptr;       // [up][tp]
};

static inline int rs_status(hipError_t e) { return e == hipSuccess ? SED_OK : SED_ERR_HIP - (int)e; }

extern "C" int sed_resampler_create(int32_t orig_sr, int32_t target_sr, sed_resampler** out) {
  if (!out || orig_sr < 1 || target_sr < 1) return SED_ERR_BAD_ARG;
  *out = nullptr;
  const int32_t g = std::gcd(orig_sr, target_sr);
  const int32_t up = target_sr / g, down = orig_sr / g;
  sed_resampler* r = new (std::nothrow) sed_resampler();
  if (!r) return SED_ERR_BAD_ARG;
  r->up = up;
  r->down = down;
  if (up == 1 && down == 1) {
    r->taps = 1;
    r->n_pre_remove = 0;
    r->table_host = {1.0f};
  } else {
    const int max_rate = std::max(up, down);
    const double f_c = 1.0 / max_rate;
    const int half_len = 10 * max_rate, n = 2 * half_len + 1;
    std::vector<double> h(n);
    const double i0b = bessel_i0(5.0);
    for (int i = 0; i < n; ++i) {
      const double m = (double)i - half_len;
      const double xs = f_c * m;
      const double y = M_PI * (xs == 0.0 ? 1.0e-20 : xs);
      const double r2 = m / half_len;
      h[i] = f_c * (std::sin(y) / y) * (bessel_i0(5.0 * std::sqrt(std::max(0.0, 1.0 - r2 * r2))) / i0b);
    }
    const double s = pairwise_sum(h.data(), h.size());
    const int n_pre_pad = down - half_len % down;
    r->n_pre_remove = (half_len + n_pre_pad) / down;
    const int len = n + n_pre_pad;
    r->taps = (len + up - 1) / up;
    r->table_host.assign((size_t)up * r->taps, 0.0f);
    for (int i = 0; i < n; ++i) {
      const float hv = (float)(h[i] / s) * (float)up;  // cast to fp32, then scaled in fp32
      const int idx = i + n_pre_pad;
      r->table_host[(size_t)(idx % up) * r->taps + idx / up] = hv;
    }
  }
  r->tp = r->taps <= 32 ? 32 : r->taps <= 64 ? 64 : 128;
  if (up > 1024 || r->taps > 128) {  // a thread per phase, a register per coefficient
    delete r;
    return SED_ERR_BAD_ARG;
  }
  int rc = rs_status(hipGetDevice(&r->device));
  hipDeviceProp_t prop;
  if (!rc) rc = rs_status(hipGetDeviceProperties(&prop, r->device));
  if (!rc) r->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  std::vector<float> padded((size_t)up * r->tp, 0.0f);
  for (int ph = 0; ph < up; ++ph)
    for (int j = 0; j < r->taps; ++j) padded[(size_t)ph * r->tp + j] = r->table_host[(size_t)ph * r->taps + j];
  if (!rc) rc = rs_status(hipMalloc((void**)&r->d_table, padded.size() * sizeof(float)));
  if (!rc) rc = rs_status(hipMemcpy(r->d_table, padded.data(), padded.size() * sizeof(float), hipMemcpyHostToDevice));
  if (rc) {
    if (r->d_table) (void)hipFree(r->d_table);
    delete r;
    return rc;
  }
  *out = r;
  return SED_OK;
}

extern "C" void sed_resampler_destroy(sed_resampler* r) {
  if (!r) return;
  if (r->d_table) (void)hipFree(r->d_table);
  delete r;
}

extern "C" int64_t sed_resampler_out_len(const sed_resampler* r, int64_t n_in) {
  if (!r || n_in < 0) return SED_ERR_BAD_ARG;
  return (n_in * r->up + r->down - 1) / r->down;
}

extern "C" int sed_resampler_get_filter(const sed_resampler* r, int32_t* up, int32_t* down, int32_t* taps,
                                        int32_t* n_pre_remove, float* table_host) {
  if (!r) return SED_ERR_BAD_ARG;
  if (up) *up = r->up;
  if (down) *down = r->down;
  if (taps) *taps = r->taps;
  if (n_pre_remove) *n_pre_remove = r->n_pre_remove;
  if (table_host) std::memcpy(table_host, r->table_host.data(), r->table_host.size() * sizeof(float));
  return SED_OK;
}

template <int TP>
static int launch_resample(const sed_resampler* r, ResampleParams p, int32_t pcm_kind, int threads, size_t lds,
                           int blocks, hipStream_t stream) {
  if (pcm_kind == SED_PCM_F32) {
    auto kern = resample_kernel<TP, float>;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, stream, p);
  } else {
    auto kern = resample_kernel<TP, int16_t>;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, stream, p);
  }
  return rs_status(hipGetLastError());
}

extern "C" int sed_resampler_forward(const sed_resampler* r, const void* pcm, int32_t pcm_kind, float pcm_divisor,
                                     int64_t B, int64_t L, int64_t pcm_stride, float* out, int64_t out_stride,
                                     void* stream) {
  if (!r || B < 0 || L < 0 || pcm_stride < L) return SED_ERR_BAD_ARG;
  if (pcm_kind != SED_PCM_F32 && pcm_kind != SED_PCM_I16) return SED_ERR_BAD_ARG;
  if (pcm_kind == SED_PCM_I16 && !(pcm_divisor > 0.0f)) return SED_ERR_BAD_ARG;
  const int64_t n_out = sed_resampler_out_len(r, L);
  if (out_stride < n_out) return SED_ERR_BAD_ARG;
  if (B == 0 || n_out == 0) return SED_OK;
  if (!pcm || !out) return SED_ERR_BAD_ARG;
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return rs_status(e);
  if (dev != r->device) return SED_ERR_WRONG_DEVICE;
  // P periods of `up` outputs per tile: about 256 threads, at most 1024 (512 when a thread holds 128 coefficients)
  const int max_threads = r->tp > 64 ? 512 : 1024;
  if (r->up > max_threads) return SED_ERR_BAD_ARG;
  const int P = std::max(1, std::min(max_threads / r->up, (256 + r->up - 1) / r->up));
  const int threads = r->up * P;
  ResampleParams p;
  p.pcm = pcm;
  p.out = out;
  p.table = r->d_table;
  p.B = B;
  p.L = L;
  p.pcm_stride = pcm_stride;
  p.out_stride = out_stride;
  p.n_out = n_out;
  p.tiles = (n_out + threads - 1) / threads;
  p.up = r->up;
  p.down = r->down;
  p.n_pre_remove = r->n_pre_remove;
  p.qstep = P * r->down;
  p.div = pcm_kind == SED_PCM_I16 ? pcm_divisor : 1.0f;
  // inputs of a tile: from (oldest tap of thread 0) to (newest sample of the last thread)
  const int64_t q_0 = ((int64_t)r->n_pre_remove * r->down) / r->up;
  const int64_t q_last = (((int64_t)threads - 1 + r->n_pre_remove) * r->down) / r->up;
  p.stage_count = (int32_t)(q_last - q_0) + r->tp;
  const size_t lds = (size_t)p.stage_count * sizeof(float);
  if (lds > 64 * 1024) return SED_ERR_BAD_ARG;
  const int64_t total = B * p.tiles;
  const int blocks = (int)std::min<int64_t>(total, (int64_t)r->num_cus * std::max(1, 2048 / threads) * 2);
  hipStream_t s = (hipStream_t)stream;
  if (r->tp == 32) return launch_resample<32>(r, p, pcm_kind, threads, lds, blocks, s);
  if (r->tp == 64) return launch_resample<64>(r, p, pcm_kind, threads, lds, blocks, s);
  return launch_resample<128>(r, p, pcm_kind, threads, lds, blocks, s);
}
