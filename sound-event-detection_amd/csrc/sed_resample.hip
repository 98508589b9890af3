// Polyphase sample-rate conversion for gfx950 with the arithmetic of scipy.signal.resample_poly (the implementation
// behind librosa.resample(res_type='polyphase')): Kaiser(5.0) low-pass of 20*max(up, down)+1 taps designed in fp64,
// cast to fp32 and scaled by `up` in fp32; every output sample accumulated in fp32 from the oldest contributing input
// sample to the newest with separately rounded multiplies and adds.  The results equal scipy's bit for bit.
//
// Work decomposition: a workgroup of up*P threads (about 256) produces R groups of up*P consecutive output samples per
// tile; a group is a multiple of `up` long, so a thread meets the same filter phase in every group and keeps that
// phase's coefficients in registers for the whole launch.  The input samples a tile needs (R*P*down + taps, contiguous)
// are staged once in LDS, either straight from global memory by LDS-DMA (resample_dma_kernel, double buffered) or
// through registers (resample_kernel).  Bound: HBM (read L, write L*up/down samples per clip).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
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
  int64_t B, pcm_stride, out_stride;
  const int32_t* assign;  // [blockDim.x] output of a group owned by each thread
  int32_t L, n_out, tiles, up, down, n_pre_remove, q0, qstep, stage_count, n_iter, span_stride;
  float div;
  int32_t div3;  // 1: x * (1/div) corrected by two fused steps equals x / div for every int16 x (checked on the host)
};

constexpr int stage_regs(int groups) { return groups >= 8 ? 24 : 16; }  // staging registers per thread

// Tile = R groups of blockDim.x consecutive outputs.  Thread t owns output o = assign[t] of every group (a permutation
// made on the host so that the lanes of a wavefront read from different LDS banks): the group length is a multiple of
// `up`, so the thread meets one filter phase only and keeps its TP coefficients in registers.  The input span of a
// tile is staged in LDS once; the span of the next tile is fetched into registers while the current one is computed.
template <int TP, int R, typename T>
__global__ void __launch_bounds__(512) resample_kernel(const ResampleParams p) {
  extern __shared__ __align__(16) float lds[];
  const int t = threadIdx.x, NT = blockDim.x;
  float* xs = lds;                   // [stage_count] input span of the tile
  float* ys = lds + p.stage_count;   // [R][NT] outputs of the tile in natural order
  const int o = p.assign[t];
  const int64_t tt = ((int64_t)o + p.n_pre_remove) * p.down;
  const int phase = (int)(tt % p.up);
  const int lo = (int)(tt / p.up) - p.q0;  // oldest input of output o (group 0) within the span
  float c[TP];
#pragma unroll
  for (int j = 0; j < TP; ++j) c[j] = p.table[phase * TP + j];
  constexpr int kStageRegs = stage_regs(R);
  T pre[kStageRegs];
  const int total = (int)p.B * p.tiles;
  auto fetch = [&](int work) {
    const int b = work / p.tiles, tile = work - b * p.tiles;
    const int k_lo = p.q0 + tile * (R * p.qstep) - (TP - 1);
    const T* src = (const T*)p.pcm + (int64_t)b * p.pcm_stride;
#pragma unroll
    for (int n = 0; n < kStageRegs; ++n)
      if (n < p.n_iter) {
        // always a load from inside the clip (clamped), never a select on the loaded value: nothing here waits for memory
        const int k = k_lo + t + n * NT;
        pre[n] = src[min(max(k, 0), p.L - 1)];
      }
  };
  if ((int)blockIdx.x < total) fetch(blockIdx.x);
  for (int work = blockIdx.x; work < total; work += gridDim.x) {
    const int b = work / p.tiles, tile = work - b * p.tiles;
    const int k_lo = p.q0 + tile * (R * p.qstep) - (TP - 1);
#pragma unroll
    for (int n = 0; n < kStageRegs; ++n)
      if (n < p.n_iter) {
        const int i = t + n * NT, k = k_lo + i;
        if (i < p.stage_count) {
          float v = 0.0f;
          if (k >= 0 && k < p.L) {
            if constexpr (std::is_same<T, float>::value) v = pre[n];
            else v = (float)pre[n] / p.div;
          }
          xs[i] = v;
        }
      }
    __syncthreads();
    if (work + (int)gridDim.x < total) fetch(work + gridDim.x);
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0f;
    int lo_here = lo;
    asm volatile("" : "+v"(lo_here));  // addresses derived from it are formed per tile, not kept in registers across tiles
    const float* mine = xs + lo_here;
    // four taps of all R groups per step; the reads of the next step are issued before the arithmetic of this one
    // (pinned: left alone, the compiler issues every read of the tile up front and runs out of registers)
    constexpr int S = R >= 8 ? 2 : 4;  // taps per step: 16 reads in flight per step whatever R is
    float xv[2][S][R];
    auto load = [&](int buf, int i0) {
#pragma unroll
      for (int e = 0; e < S; ++e)
#pragma unroll
        for (int r = 0; r < R; ++r) xv[buf][e][r] = mine[i0 + e + r * p.qstep];
    };
    load(0, 0);
#pragma unroll
    for (int i0 = 0; i0 < TP; i0 += S) {  // oldest input first, as scipy accumulates
      if (i0 + S < TP) load(((i0 / S) + 1) & 1, i0 + S);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < S; ++e)
#pragma unroll
        for (int r = 0; r < R; ++r)
          acc[r] = __fadd_rn(acc[r], __fmul_rn(xv[(i0 / S) & 1][e][r], c[TP - 1 - i0 - e]));
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) ys[r * NT + o] = acc[r];
    __syncthreads();
    float* dst = p.out + (int64_t)b * p.out_stride;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int m = (tile * R + r) * NT + t;
      if (m < p.n_out) dst[m] = ys[r * NT + t];
    }
  }
}

// The same computation for fp32 input with the span moved from global memory straight into LDS (LDS-DMA, 4 bytes per
// lane so that any sample alignment works), double buffered: the span of tile i+1 is in flight while tile i is computed,
// no staging registers, one workgroup barrier per tile.  blockDim.x is a multiple of 32 (a last wavefront of 32 threads
// moves 32 samples per round).
//   buffers: xs[2][span_stride] (span_stride = DMA rounds x blockDim.x >= stage_count), ys[2][R][blockDim.x]
// B64: every lane starts on an even sample (an odd first sample is skipped: its coefficient is the table's zero padding,
// and the lane's coefficients move up by one), so the tap loop reads its sample pairs with 8-byte LDS reads, which the
// LDS serves at twice the rate of two 4-byte reads per lane.  Needs an even number of input samples per group and at
// least one zero coefficient of padding per phase.
// MAXT: the largest workgroup the kernel is built for (512: up to 256 registers; 1024: up to 128, phases of at most 64
// coefficients).
template <int TP, int R, typename T, bool B64, int MAXT>
__global__ void __launch_bounds__(MAXT) resample_dma_kernel(const ResampleParams p) {
  extern __shared__ __align__(16) float lds[];
  const int t = threadIdx.x, NT = blockDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  float* xs = lds;
  float* ys = lds + 2 * p.span_stride;
  const int o = p.assign[t];
  const int64_t tt = ((int64_t)o + p.n_pre_remove) * p.down;
  const int phase = (int)(tt % p.up);
  int lo = (int)(tt / p.up) - p.q0;
  float c[TP];
  if constexpr (B64) {
    const int odd = lo & 1;
    lo += odd;
#pragma unroll
    for (int j = 0; j < TP; ++j) c[j] = j >= odd ? p.table[phase * TP + j - odd] : 0.0f;
  } else {
#pragma unroll
    for (int j = 0; j < TP; ++j) c[j] = p.table[phase * TP + j];
  }
  const int total = (int)p.B * p.tiles;
  // issue: wave w moves samples [64 (w + n NT/64), +64) of the span in round n; addresses are clamped into the clip
  auto issue = [&](int work, float* dstbuf) {
    const int b = work / p.tiles, tile = work - b * p.tiles;
    const int k_lo = p.q0 + tile * (R * p.qstep) - (TP - 1);
    const T* src = (const T*)p.pcm + (int64_t)b * p.pcm_stride;
    for (int n = 0; n < p.n_iter; ++n) {
      const int i0 = n * NT + 64 * wave;
      const int k = min(max(k_lo + i0 + lane, 0), p.L - 1);
      // (an int16 sample lands zero-extended in the 4-byte LDS slot of its lane)
      if constexpr (std::is_same<T, float>::value)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + k),
                                         (__attribute__((address_space(3))) void*)(dstbuf + i0), 4, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + k),
                                         (__attribute__((address_space(3))) void*)(dstbuf + i0), 2, 0, 0);
    }
  };
  // after the wave's own transfers have landed: zeros where the span reaches outside the clip (first / last tiles)
  auto fix_edges = [&](int work, float* dstbuf) {
    const int b = work / p.tiles, tile = work - b * p.tiles;
    const int k_lo = p.q0 + tile * (R * p.qstep) - (TP - 1);
    if constexpr (std::is_same<T, float>::value) {
      if (k_lo >= 0 && k_lo + p.n_iter * NT <= p.L) return;  // wave-uniform
      for (int n = 0; n < p.n_iter; ++n) {
        const int i = n * NT + 64 * wave + lane, k = k_lo + i;
        if (k < 0 || k >= p.L) dstbuf[i] = 0.0f;
      }
    } else {  // int16: every landed sample is converted in place by the lane that fetched it
      for (int n = 0; n < p.n_iter; ++n) {
        const int i = n * NT + 64 * wave + lane, k = k_lo + i;
        const int raw = (int)(short)(reinterpret_cast<const unsigned int*>(dstbuf)[i] & 0xffffu);
        float v;
        if (p.div3) {  // wave-uniform
          const float rcp = 1.0f / p.div, xf = (float)raw, q = xf * rcp;
          v = __builtin_fmaf(__builtin_fmaf(-q, p.div, xf), rcp, q);
        } else {
          v = (float)raw / p.div;
        }
        dstbuf[i] = (k < 0 || k >= p.L) ? 0.0f : v;
      }
    }
  };
  if ((int)blockIdx.x < total) {
    issue(blockIdx.x, xs);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fix_edges(blockIdx.x, xs);
  }
  __syncthreads();
  int parity = 0;
  for (int work = blockIdx.x; work < total; work += gridDim.x, parity ^= 1) {
    const int b = work / p.tiles, tile = work - b * p.tiles;
    const float* cur = xs + parity * p.span_stride;
    float* nxt = xs + (parity ^ 1) * p.span_stride;
    float* yb = ys + parity * (R * NT);
    const bool more = work + (int)gridDim.x < total;
    if (more) issue(work + gridDim.x, nxt);
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0f;
    int lo_here = lo;
    asm volatile("" : "+v"(lo_here));  // addresses derived from it are formed per tile, not kept in registers across tiles
    const float* mine = cur + lo_here;
    constexpr int S = R >= 8 ? 2 : 4;
#ifndef SED_RS_NO_ASM_PIPE
    // Hand-placed LDS reads and waits: with an LDS-DMA transfer in flight the compiler waits for ALL outstanding LDS
    // reads before every use (it cannot order them against the transfer), which would serialise fetch and arithmetic.
    // The reads return in order, so after the reads of the next step are issued, waiting until only those are
    // outstanding makes this step's samples valid.  Nothing else of the wavefront uses the LDS counter in this loop.
    typedef float f2 __attribute__((ext_vector_type(2)));
    static_assert(S % 2 == 0, "taps are read two at a time");
    f2 xp[2][S / 2][R];
    unsigned base[R];
#pragma unroll
    for (int r = 0; r < R; ++r) base[r] = (unsigned)(size_t)(mine + r * p.qstep);
    auto load = [&](int buf, int i0) {
#pragma unroll
      for (int e = 0; e < S / 2; ++e)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if constexpr (B64)
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xp[buf][e][r]) : "v"(base[r]), "n"(4 * (i0 + 2 * e)) : "memory");
          else
            asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3"
                         : "=v"(xp[buf][e][r]) : "v"(base[r]), "n"(i0 + 2 * e), "n"(i0 + 2 * e + 1) : "memory");
        }
    };
    constexpr int kPerStep = (S / 2) * R;  // read instructions of one step
    static_assert(kPerStep <= 15, "the wait counter holds 15");
    load(0, 0);
#pragma unroll
    for (int i0 = 0; i0 < TP; i0 += S) {  // oldest input first, as scipy accumulates
      const int cur_buf = (i0 / S) & 1;
      if (i0 + S < TP) {
        load(cur_buf ^ 1, i0 + S);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(kPerStep) : "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      // (the samples pass through an empty statement so that no use of them can be placed above the wait)
#pragma unroll
      for (int e = 0; e < S / 2; ++e)
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("" : "+v"(xp[cur_buf][e][r]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < S; ++e)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const f2 pr = xp[cur_buf][e / 2][r];
          acc[r] = __fadd_rn(acc[r], __fmul_rn((e & 1) ? pr.y : pr.x, c[TP - 1 - i0 - e]));
        }
      __builtin_amdgcn_sched_barrier(0);
    }
#else
    float xv[2][S][R];
    auto load = [&](int buf, int i0) {
#pragma unroll
      for (int e = 0; e < S; ++e)
#pragma unroll
        for (int r = 0; r < R; ++r) xv[buf][e][r] = mine[i0 + e + r * p.qstep];
    };
    load(0, 0);
#pragma unroll
    for (int i0 = 0; i0 < TP; i0 += S) {  // oldest input first, as scipy accumulates
      if (i0 + S < TP) load(((i0 / S) + 1) & 1, i0 + S);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < S; ++e)
#pragma unroll
        for (int r = 0; r < R; ++r)
          acc[r] = __fadd_rn(acc[r], __fmul_rn(xv[(i0 / S) & 1][e][r], c[TP - 1 - i0 - e]));
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
#pragma unroll
    for (int r = 0; r < R; ++r) yb[r * NT + o] = acc[r];
    // the next span has had the whole tile to land; the stores below are issued after this wait, so it never waits for them
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (more) fix_edges(work + gridDim.x, nxt);
    __syncthreads();
    float* dst = p.out + (int64_t)b * p.out_stride;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int m = (tile * R + r) * NT + t;
      if (m < p.n_out) dst[m] = yb[r * NT + t];
    }
  }
}

// Any rate pair: one thread per output sample, coefficients ([up][taps], `taps` the row length) and samples read from
// global memory; neighbouring threads re-read both, so they come from the caches.  For the pairs the kernels above do not
// hold (more than 512 phases, more than 128 coefficients per phase) and for PCM pointers the LDS-DMA kernel cannot take.
// The same order of operations, hence the same bits; several times slower per sample.
template <typename T>
__global__ void __launch_bounds__(256) resample_any_kernel(const ResampleParams p, int taps) {
  const int64_t total = p.B * (int64_t)p.n_out;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / p.n_out;
    const int m = (int)(i - b * p.n_out);
    const int64_t tt = ((int64_t)m + p.n_pre_remove) * p.down;
    const int phase = (int)(tt % p.up);
    const int64_t q = tt / p.up;  // newest input sample of output m
    const T* src = (const T*)p.pcm + b * p.pcm_stride;
    const float* c = p.table + (int64_t)phase * taps;
    float acc = 0.0f;
    for (int j = taps - 1; j >= 0; --j) {  // oldest input first, as scipy accumulates
      const int64_t k = q - j;
      float v = 0.0f;
      if (k >= 0 && k < p.L) {
        if constexpr (std::is_same<T, float>::value) v = src[k];
        else v = (float)src[k] / p.div;
      }
      acc = __fadd_rn(acc, __fmul_rn(v, c[j]));
    }
    p.out[b * p.out_stride + m] = acc;
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
  int32_t threads = 256, periods = 1, groups = 1, q0 = 0, qstep = 0, stage_count = 0;
  int32_t dma_groups = 0, dma_stage_count = 0, dma_rounds = 0;  // LDS-DMA variant (fp32 input), 0 groups: not available
  std::vector<int32_t> assign;    // [threads] output of a group owned by each thread
  int32_t* d_assign = nullptr;
  bool any = false;               // served by resample_any_kernel alone (d_table is [up][taps] then)
  bool b64 = false;               // the LDS-DMA variant reads sample pairs from even offsets (8-byte LDS reads)
  std::vector<int32_t> assign64;  // its thread -> output table
  int32_t* d_assign64 = nullptr;
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
  if (std::max(up, down) > (1 << 20)) {  // a filter of 20 * max(up, down) + 1 coefficients: rates without a common factor to speak of
    delete r;
    return SED_ERR_BAD_ARG;
  }
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
  static const int kSizes[] = {16, 32, 48, 56, 60, 64, 96, 128};
  r->tp = 0;
  for (int sz : kSizes)
    if (r->tp == 0 && r->taps <= sz) r->tp = sz;
  const int max_threads = 512;
  // the tiled kernels: a thread per phase, a register per coefficient; more than 512 phases in the LDS-DMA kernel only,
  // whose workgroups of up to 1024 threads are built for phases of at most 64 coefficients
  r->any = r->tp == 0 || up > 1024 || (up > max_threads && r->tp > 64);
  int span1 = 0;
  if (!r->any) {
    // P periods of `up` outputs per group: about 256 threads; R groups per tile, as many of {4, 2, 1} as the staging
    // registers and 64 KB of LDS allow
    r->periods = std::max(1, std::min(max_threads / up, (256 + up - 1) / up));
    {
      // Workgroups whose wavefront count is not a multiple of four leave the CU half empty: with 320 threads (5 wavefronts) of
      // 110 registers a CU held 7 wavefronts on average instead of the 15 its registers and LDS allow, with 480 threads (8
      // wavefronts) it holds 14 (SQ_WAVE_CYCLES against GRBM_GUI_ACTIVE) -- as if the wavefronts of every workgroup went to the
      // four SIMDs in turn starting from the first, so that two 5-wavefront workgroups fill the first SIMD's registers.  Prefer, among
      // the period counts of at least 256 threads, the one whose wavefronts spread most evenly over the SIMDs, and one
      // whose group advances by an even number of input samples (8-byte LDS reads, see resample_dma_kernel).  Workgroups of
      // more than 512 threads exist for the LDS-DMA kernel only (phases of at most 64 coefficients, whole half wavefronts,
      // its buffers within 64 KB).
      auto dma_fits = [&](int P) {
        const int64_t threads = (int64_t)up * P, qstep = (int64_t)P * down;
        if (threads % 32 != 0) return false;
        const int64_t count = ((threads - 1 + r->n_pre_remove) * down) / up - ((int64_t)r->n_pre_remove * down) / up + r->tp + 1;
        const int64_t rounds = (count + threads - 1) / threads;
        (void)qstep;
        return (2 * rounds * threads + 2 * threads) * 4 <= 64 * 1024;  // one group per tile at least
      };
      double best = 1e30;
      const int first = r->periods;
      for (int P = first; P >= 1 && (int64_t)up * P <= 1024; ++P) {
        if (up * P > max_threads && !(r->tp <= 64 && dma_fits(P))) continue;
        const int waves = (up * P + 63) / 64;
        const double score = (double)((waves + 3) / 4) / (waves / 4.0)             // most loaded SIMD / average
                             - (((int64_t)P * down) % 2 == 0 ? 0.2 : 0.0) + 1e-3 * P;
        if (score < best - 1e-9) { best = score; r->periods = P; }
      }
    }
    r->threads = up * r->periods;
    r->q0 = (int32_t)(((int64_t)r->n_pre_remove * down) / up);
    r->qstep = r->periods * down;
    span1 = (int)((((int64_t)r->threads - 1 + r->n_pre_remove) * down) / up) - r->q0 + r->tp;
    r->groups = 0;
    for (int R : {8, 4, 2, 1}) {
      const int64_t count = (int64_t)(R - 1) * r->qstep + span1;
      if (r->groups == 0 && count <= (int64_t)stage_regs(R) * r->threads && (count + (int64_t)R * r->threads) * 4 <= 64 * 1024) {
        r->groups = R;
        r->stage_count = (int32_t)count;
      }
    }
    if (r->groups == 0 && r->threads <= max_threads) r->any = true;
  }
  if (r->any) {
    r->tp = r->taps;
    r->groups = r->dma_groups = 0;
    r->threads = 256;
  }
  if (!r->any && r->threads % 32 == 0)  // (a last wavefront with 32 threads moves 32 samples per round: the rounds stay contiguous)
    for (int R : {4, 2, 1}) {  // two span buffers and two output buffers in at most 64 KB: two workgroups per CU
      const int64_t count = (int64_t)(R - 1) * r->qstep + span1;
      const int64_t rounds = (count + r->threads - 1) / r->threads;
      if (r->dma_groups == 0 && (2 * rounds * r->threads + 2 * (int64_t)R * r->threads) * 4 <= 64 * 1024) {
        r->dma_groups = R;
        r->dma_stage_count = (int32_t)count;
        r->dma_rounds = (int32_t)rounds;
      }
    }
  if (!r->any && r->threads > max_threads && r->dma_groups == 0) {  // more than 512 phases, and no LDS-DMA shape for them either
    r->any = true;
    r->tp = r->taps;
    r->groups = 0;
    r->threads = 256;
  }
  r->b64 = r->dma_groups > 0 && r->qstep % 2 == 0 && r->taps < r->tp;
  if (r->b64) {
    // a lane that skips an odd first sample reads one sample past its window (coefficient 0): one more staged sample
    const int count = r->dma_stage_count + 1, rounds = (count + r->threads - 1) / r->threads;
    if ((2 * (int64_t)rounds * r->threads + 2 * (int64_t)r->dma_groups * r->threads) * 4 <= 64 * 1024) {
      r->dma_stage_count = count;
      r->dma_rounds = rounds;
    } else {
      r->b64 = false;
    }
  }
  if (r->b64) {
    // thread -> output table for 8-byte reads: the 32 lanes the LDS serves together start on 32 different bank pairs,
    // as far as the residues allow
    const int NT = r->threads;
    std::vector<std::vector<int>> bucket(32);
    for (int o = NT - 1; o >= 0; --o) {
      int pos = (int)((((int64_t)o + r->n_pre_remove) * down) / up) - r->q0;
      pos += pos & 1;
      bucket[(pos / 2) % 32].push_back(o);
    }
    r->assign64.assign(NT, -1);
    int nb = 0;
    for (int t = 0; t < NT; ++t) {
      if (t % 32 == 0) nb = 0;
      for (int tries = 0; tries < 32 && bucket[nb].empty(); ++tries) nb = (nb + 1) % 32;
      r->assign64[t] = bucket[nb].back();
      bucket[nb].pop_back();
      nb = (nb + 1) % 32;
    }
  }
  if (!r->any) {
    // thread -> output assignment of a group: every wavefront gets outputs whose first LDS read falls into 64 different
    // banks (lane = bank; lanes i and i + 32 then also differ modulo 32), as far as the residues allow; later taps and
    // groups shift every lane by the same amount, so the property holds for all reads.
    const int NT = r->threads;
    std::vector<std::vector<int>> bucket(64);
    for (int o = NT - 1; o >= 0; --o) {
      const int pos = (int)((((int64_t)o + r->n_pre_remove) * down) / up) - r->q0;
      bucket[pos % 64].push_back(o);
    }
    // wavefront by wavefront, one output from every residue that still has some, round and round until the 64 lanes
    // are filled: a residue that must repeat within a wavefront (hops of an even number of samples reach only half of the
    // banks) repeats as rarely as possible
    r->assign.assign(NT, -1);
    int next_bucket = 0;
    for (int t = 0; t < NT; ++t) {
      if (t % 64 == 0) next_bucket = 0;
      for (int tries = 0; tries < 64 && bucket[next_bucket].empty(); ++tries) next_bucket = (next_bucket + 1) % 64;
      auto& bk = bucket[next_bucket];
      r->assign[t] = bk.back();
      bk.pop_back();
      next_bucket = (next_bucket + 1) % 64;
    }
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
  if (!rc && !r->any) rc = rs_status(hipMalloc((void**)&r->d_assign, r->assign.size() * sizeof(int32_t)));
  if (!rc && !r->any) rc = rs_status(hipMemcpy(r->d_assign, r->assign.data(), r->assign.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  if (!rc && r->b64) {
    rc = rs_status(hipMalloc((void**)&r->d_assign64, r->assign64.size() * sizeof(int32_t)));
    if (!rc) rc = rs_status(hipMemcpy(r->d_assign64, r->assign64.data(), r->assign64.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  if (rc) {
    if (r->d_table) (void)hipFree(r->d_table);
    if (r->d_assign) (void)hipFree(r->d_assign);
    if (r->d_assign64) (void)hipFree(r->d_assign64);
    delete r;
    return rc;
  }
  *out = r;
  return SED_OK;
}

extern "C" void sed_resampler_destroy(sed_resampler* r) {
  if (!r) return;
  if (r->d_table) (void)hipFree(r->d_table);
  if (r->d_assign) (void)hipFree(r->d_assign);
  if (r->d_assign64) (void)hipFree(r->d_assign64);
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

template <int TP, int R>
static int launch_resample_r(const ResampleParams& p, int32_t pcm_kind, int threads, size_t lds, int blocks,
                             hipStream_t stream) {
  if (pcm_kind == SED_PCM_F32) {
    auto kern = resample_kernel<TP, R, float>;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, stream, p);
  } else {
    auto kern = resample_kernel<TP, R, int16_t>;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, stream, p);
  }
  return rs_status(hipGetLastError());
}

template <int TP, typename T, bool B64, int MAXT>
static int launch_resample_dma_t(int groups, const ResampleParams& p, int threads, size_t lds, int blocks, hipStream_t stream) {
  if (groups == 4) hipLaunchKernelGGL((resample_dma_kernel<TP, 4, T, B64, MAXT>), dim3(blocks), dim3(threads), lds, stream, p);
  else if (groups == 2) hipLaunchKernelGGL((resample_dma_kernel<TP, 2, T, B64, MAXT>), dim3(blocks), dim3(threads), lds, stream, p);
  else hipLaunchKernelGGL((resample_dma_kernel<TP, 1, T, B64, MAXT>), dim3(blocks), dim3(threads), lds, stream, p);
  return rs_status(hipGetLastError());
}

template <int TP, int MAXT>
static int launch_resample_dma_m(int groups, bool b64, const ResampleParams& p, int threads, size_t lds, int blocks, hipStream_t stream) {
  if (p.div == 0.0f)
    return b64 ? launch_resample_dma_t<TP, float, true, MAXT>(groups, p, threads, lds, blocks, stream)
               : launch_resample_dma_t<TP, float, false, MAXT>(groups, p, threads, lds, blocks, stream);
  return b64 ? launch_resample_dma_t<TP, int16_t, true, MAXT>(groups, p, threads, lds, blocks, stream)
             : launch_resample_dma_t<TP, int16_t, false, MAXT>(groups, p, threads, lds, blocks, stream);
}

template <int TP>
static int launch_resample_dma(int groups, bool b64, const ResampleParams& p, int threads, size_t lds, int blocks, hipStream_t stream) {
  if (threads <= 512) return launch_resample_dma_m<TP, 512>(groups, b64, p, threads, lds, blocks, stream);
  if constexpr (TP <= 64) return launch_resample_dma_m<TP, 1024>(groups, b64, p, threads, lds, blocks, stream);
  else return SED_ERR_BAD_ARG;
}

template <int TP>
static int launch_resample(int groups, const ResampleParams& p, int32_t pcm_kind, int threads, size_t lds, int blocks,
                           hipStream_t stream) {
  if (groups == 8) return launch_resample_r<TP, 8>(p, pcm_kind, threads, lds, blocks, stream);
  if (groups == 4) return launch_resample_r<TP, 4>(p, pcm_kind, threads, lds, blocks, stream);
  if (groups == 2) return launch_resample_r<TP, 2>(p, pcm_kind, threads, lds, blocks, stream);
  return launch_resample_r<TP, 1>(p, pcm_kind, threads, lds, blocks, stream);
}

// Whether q = x * rcp, q + fma(-q, d, x) * rcp reproduces the IEEE quotient x / d for all 65536 int16 values; remembered
// for the few divisors a process ever uses.
static bool three_step_division_is_exact(float d) {
  static std::mutex mu;
  static std::vector<std::pair<float, bool>> seen;
  std::lock_guard<std::mutex> lock(mu);
  for (const auto& e : seen)
    if (std::memcmp(&e.first, &d, sizeof(float)) == 0) return e.second;
  const float r = 1.0f / d;
  bool same = std::isfinite(r) && r > 0.0f;
  for (int x = -32768; same && x <= 32767; ++x) {
    const float xf = (float)x, q = xf * r;
    const float q1 = std::fmaf(std::fmaf(-q, d, xf), r, q);
    const float want = xf / d;
    same = std::memcmp(&q1, &want, sizeof(float)) == 0 || (q1 == 0.0f && want == 0.0f);
  }
  if (seen.size() < 64) seen.emplace_back(d, same);
  return same;
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
  // the kernel indexes samples within a clip and tiles within the batch with 32-bit integers
  if (L > INT32_MAX / 2 || n_out > INT32_MAX / 2) return SED_ERR_BAD_ARG;
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return rs_status(e);
  if (dev != r->device) return SED_ERR_WRONG_DEVICE;
  bool dma = r->dma_groups > 0 && (uintptr_t)pcm % (pcm_kind == SED_PCM_F32 ? 4 : 2) == 0;
#ifdef SED_DIAGNOSTIC_BUILD  // timing experiments only: force the register-staged path
  static const bool no_dma = getenv("SED_RS_NO_DMA") != nullptr, no_dma16 = getenv("SED_RS_NO_DMA16") != nullptr;
  dma = dma && !no_dma && (pcm_kind == SED_PCM_F32 || !no_dma16);
#endif
  // the general kernel: rate pairs the tiled kernels do not hold, and a workgroup of more than 512 threads whose PCM
  // pointer is not a multiple of its sample size (such a workgroup exists for the LDS-DMA kernel only)
  const bool any = r->any || (!dma && r->threads > 512);
  const int threads = r->threads, R = any ? 1 : dma ? r->dma_groups : r->groups;
  ResampleParams p;
  p.pcm = pcm;
  p.out = out;
  p.table = r->d_table;
  p.B = B;
  p.L = (int32_t)L;
  p.pcm_stride = pcm_stride;
  p.out_stride = out_stride;
  p.n_out = (int32_t)n_out;
  if (any) {
    p.up = r->up;
    p.down = r->down;
    p.n_pre_remove = r->n_pre_remove;
    p.div = pcm_kind == SED_PCM_I16 ? pcm_divisor : 1.0f;
    const int64_t want = (B * n_out + 255) / 256;
    const int blocks = (int)std::min<int64_t>(want, (int64_t)r->num_cus * 64);
    if (pcm_kind == SED_PCM_F32) hipLaunchKernelGGL(resample_any_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, r->tp);
    else hipLaunchKernelGGL(resample_any_kernel<int16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, r->tp);
    return rs_status(hipGetLastError());
  }
  p.tiles = (int32_t)((n_out + (int64_t)R * threads - 1) / ((int64_t)R * threads));
  if (B * p.tiles > INT32_MAX / 2) return SED_ERR_BAD_ARG;
  p.up = r->up;
  p.down = r->down;
  p.n_pre_remove = r->n_pre_remove;
  p.q0 = r->q0;
  p.qstep = r->qstep;
  p.stage_count = dma ? r->dma_stage_count : r->stage_count;
  const bool b64 = dma && r->b64;
  p.assign = b64 ? r->d_assign64 : r->d_assign;
  p.n_iter = dma ? r->dma_rounds : (r->stage_count + threads - 1) / threads;
  p.span_stride = r->dma_rounds * threads;
  p.div3 = (pcm_kind == SED_PCM_I16 && dma && three_step_division_is_exact(pcm_divisor)) ? 1 : 0;
  p.div = pcm_kind == SED_PCM_I16 ? pcm_divisor : (dma ? 0.0f : 1.0f);  // (0: fp32 input, for the DMA launcher)
  const size_t lds = dma ? (size_t)(2 * p.span_stride + 2 * R * threads) * sizeof(float)
                         : (size_t)(r->stage_count + R * threads) * sizeof(float);
  const int64_t total = B * p.tiles;
  const int blocks = (int)std::min<int64_t>(total, (int64_t)r->num_cus * std::max(1, 2048 / threads));
  hipStream_t s = (hipStream_t)stream;
  if (dma) switch (r->tp) {
      case 16: return launch_resample_dma<16>(R, b64, p, threads, lds, blocks, s);
      case 32: return launch_resample_dma<32>(R, b64, p, threads, lds, blocks, s);
      case 48: return launch_resample_dma<48>(R, b64, p, threads, lds, blocks, s);
      case 56: return launch_resample_dma<56>(R, b64, p, threads, lds, blocks, s);
      case 60: return launch_resample_dma<60>(R, b64, p, threads, lds, blocks, s);
      case 64: return launch_resample_dma<64>(R, b64, p, threads, lds, blocks, s);
      case 96: return launch_resample_dma<96>(R, b64, p, threads, lds, blocks, s);
      default: return launch_resample_dma<128>(R, b64, p, threads, lds, blocks, s);
    }
  switch (r->tp) {
    case 16: return launch_resample<16>(R, p, pcm_kind, threads, lds, blocks, s);
    case 32: return launch_resample<32>(R, p, pcm_kind, threads, lds, blocks, s);
    case 48: return launch_resample<48>(R, p, pcm_kind, threads, lds, blocks, s);
    case 56: return launch_resample<56>(R, p, pcm_kind, threads, lds, blocks, s);
    case 60: return launch_resample<60>(R, p, pcm_kind, threads, lds, blocks, s);
    case 64: return launch_resample<64>(R, p, pcm_kind, threads, lds, blocks, s);
    case 96: return launch_resample<96>(R, p, pcm_kind, threads, lds, blocks, s);
    default: return launch_resample<128>(R, p, pcm_kind, threads, lds, blocks, s);
  }
}
