// Frame-level event decoders for gfx950: smoothing + threshold / argmax, run extraction and gap merging.
// One 64-lane wavefront owns one trace (clip, class), one clip, or one (file, label) event group and walks
// it in 64-element chunks; boundaries are found with ballots, output slots with popcounts of the ballot
// below the lane, running maxima with shuffle scans.  No workgroup barriers, no atomics, results do not
// depend on scheduling.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

// The fp64 smoothing and the gap compare must round exactly like the CPU code they mirror: no fused
// multiply-add may be formed from a separate multiply and add anywhere in this file.
#pragma clang fp contract(off)

#include "sed_hip.h"

namespace sed {

// scipy.ndimage.gaussian_filter1d(sigma=3): radius 12, weights centre -> edge (fp64).
__constant__ double kGauss3[13] = {
    0x1.105a329f98197p-3, 0x1.01a25f86eb137p-3,  0x1.b42a57d56c0bep-4,  0x1.4a614d1afd337p-4, 0x1.bfde9c12bec92p-5,
    0x1.0fa58939b528fp-5, 0x1.26defcaeb0202p-6,  0x1.1e6bccad344bap-7,  0x1.f1e9915139406p-9, 0x1.8345966f69518p-10,
    0x1.0d8a5ad43c165p-11, 0x1.4fbe39149e277p-13, 0x1.763a210dfb306p-15};

__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// Smoothed probability of frame t of a trace with element stride `stride`.  Frames are clamped to [0, T-1]
// (scipy 'nearest' for the gaussian; 'reflect' of a 3-tap median duplicates the edge sample, same thing).
// GAUSS25: fp64, acc = x0*w0, then j = 12..1: acc += (x[-j] + x[+j]) * w[j], separate multiply and add,
// rounded once to fp32 -- scipy's correlate1d order, so the result has scipy's bits.
__device__ __forceinline__ float smoothed(const float* x, int64_t stride, int t, int T, int smoothing) {
  if (smoothing == SED_SMOOTH_NONE) return x[(int64_t)t * stride];
  if (smoothing == SED_SMOOTH_MEDIAN3) {
    const float a = x[(int64_t)max(t - 1, 0) * stride], b = x[(int64_t)t * stride];
    const float c = x[(int64_t)min(t + 1, T - 1) * stride];
    return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c));
  }
  double acc = __dmul_rn((double)x[(int64_t)t * stride], kGauss3[0]);
#pragma unroll
  for (int j = 12; j >= 1; --j) {
    const double lo = (double)x[(int64_t)max(t - j, 0) * stride];
    const double hi = (double)x[(int64_t)min(t + j, T - 1) * stride];
    acc = __dadd_rn(acc, __dmul_rn(__dadd_rn(lo, hi), kGauss3[j]));
  }
  return (float)acc;
}

// ---- threshold runs -------------------------------------------------------------------------------
// One wavefront, one trace: writes (start, end) frame pairs to out[2 r], out[2 r + 1] and returns the number of runs.
__device__ __forceinline__ int threshold_runs_wave(const float* x, int T, int C, float thr, int smoothing, int32_t* out,
                                                   int cap, int lane) {
  int n_runs = 0, n_ends = 0, carry = 0;
  for (int t0 = 0; t0 <= T; t0 += 64) {  // frame T is a virtual inactive frame that closes a trailing run
    const int t = t0 + lane;
    const int act = (t < T) ? (smoothed(x, C, t, T, smoothing) > thr) : 0;
    const uint64_t word = __ballot(act);
    const int prev = lane ? (int)((word >> (lane - 1)) & 1ull) : carry;
    const int is_start = act & !prev, is_end = (!act) & prev & (t <= T);
    const uint64_t sm = __ballot(is_start), em = __ballot(is_end);
    if (is_start) {
      const int r = n_runs + __popcll(sm & lanemask_lt(lane));
      if (r < cap) out[2 * r] = t;
    }
    if (is_end) {
      const int r = n_ends + __popcll(em & lanemask_lt(lane));
      if (r < cap) out[2 * r + 1] = t;
    }
    n_runs += __popcll(sm);
    n_ends += __popcll(em);
    carry = (int)(word >> 63);
  }
  return n_runs;
}

__global__ void __launch_bounds__(256) threshold_runs_kernel(const float* probs, int64_t n_traces, int T, int C,
                                                             float thr, int smoothing, int32_t* runs, int32_t* counts,
                                                             int cap) {
  const int lane = threadIdx.x & 63;
  const int64_t trace = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (trace >= n_traces) return;
  const int64_t b = trace / C;
  const int c = (int)(trace - b * C);
  const int n_runs = threshold_runs_wave(probs + b * (int64_t)T * C + c, T, C, thr, smoothing,
                                         runs + trace * (int64_t)cap * 2, cap, lane);
  if (lane == 0) counts[trace] = n_runs;
}

// ---- integer activity edges -----------------------------------------------------------------------
__global__ void __launch_bounds__(256) activity_edges_kernel(const int32_t* act, int64_t n_traces, int T, int C,
                                                             int32_t* starts, int32_t* ends, int32_t* n_starts,
                                                             int32_t* n_ends, int cap) {
  const int lane = threadIdx.x & 63;
  const int64_t trace = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (trace >= n_traces) return;
  const int64_t b = trace / C;
  const int c = (int)(trace - b * C);
  const int32_t* x = act + b * (int64_t)T * C + c;
  int32_t* so = starts + trace * (int64_t)cap;
  int32_t* eo = ends + trace * (int64_t)cap;
  int ns = 0, ne = 0;
  for (int t0 = 0; t0 <= T; t0 += 64) {
    const int t = t0 + lane;
    const int cur = (t < T) ? x[(int64_t)t * C] : 0;
    const int prv = (t >= 1 && t <= T) ? x[(int64_t)(t - 1) * C] : 0;
    const int64_t d = (int64_t)cur - (int64_t)prv;
    const int is_s = (t <= T) && d == 1, is_e = (t <= T) && d == -1;
    const uint64_t sm = __ballot(is_s), em = __ballot(is_e);
    if (is_s) {
      const int r = ns + __popcll(sm & lanemask_lt(lane));
      if (r < cap) so[r] = t;
    }
    if (is_e) {
      const int r = ne + __popcll(em & lanemask_lt(lane));
      if (r < cap) eo[r] = t;
    }
    ns += __popcll(sm);
    ne += __popcll(em);
  }
  if (lane == 0) { n_starts[trace] = ns; n_ends[trace] = ne; }
}

// ---- argmax runs ----------------------------------------------------------------------------------
__device__ __forceinline__ int argmax_label(const float* row0, int t, int T, int C, int n_cls, int smoothing) {
  // first maximum wins (numpy / torch CPU argmax)
  int best = 0;
  float bv = smoothed(row0, C, t, T, smoothing);
  for (int c = 1; c < n_cls; ++c) {
    const float v = smoothed(row0 + c, C, t, T, smoothing);
    if (v > bv) { bv = v; best = c; }
  }
  return best;
}

__global__ void __launch_bounds__(256) argmax_runs_kernel(const float* probs, int64_t B, int T, int C, int smoothing,
                                                          int drop_last, int32_t* runs, int32_t* counts, int cap) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float* x = probs + b * (int64_t)T * C;
  int32_t* out = runs + b * (int64_t)cap * 3;
  const int n_cls = C - drop_last;
  int n_runs = 0, carry_label = -1;
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int t = t0 + lane;
    const int lab = (t < T) ? argmax_label(x, t, T, C, n_cls, smoothing) : -2;
    int prev = __shfl_up(lab, 1);
    if (lane == 0) prev = carry_label;
    const int boundary = (t < T) && (lab != prev);
    const uint64_t bm = __ballot(boundary);
    if (boundary) {
      const int r = n_runs + __popcll(bm & lanemask_lt(lane));
      if (r < cap) { out[3 * r] = lab; out[3 * r + 1] = t; }
      if (r >= 1 && r - 1 < cap) out[3 * (r - 1) + 2] = t;
    }
    n_runs += __popcll(bm);
    carry_label = __shfl(lab, 63);
  }
  if (lane == 0) {
    if (n_runs >= 1 && n_runs - 1 < cap) out[3 * (n_runs - 1) + 2] = T;
    counts[b] = n_runs;
  }
}

__global__ void __launch_bounds__(256) argmax_hist_kernel(const float* probs, int64_t B, int T, int C, int drop_last,
                                                          int32_t* hist) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float* x = probs + b * (int64_t)T * C;
  const int n_cls = C - drop_last;
  int mine = 0;  // lane c accumulates the count of class c (n_cls <= 64)
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int t = t0 + lane;
    const int lab = (t < T) ? argmax_label(x, t, T, C, n_cls, SED_SMOOTH_NONE) : -1;
    for (int c = 0; c < n_cls; ++c) {
      const int n = __popcll(__ballot(lab == c));
      if (lane == c) mine += n;
    }
  }
  if (lane < n_cls) hist[b * n_cls + lane] = mine;
}

// ---- gap merge ------------------------------------------------------------------------------------
// Sequential rule per group (events sorted by onset): fold next into current when
// next.onset - current.offset <= gap, current.offset = max(current.offset, next.offset).
// Parallel form: with P_i = max of ALL earlier offsets in the group, "fl(onset_i - P_i) > gap" marks exactly
// the events that open a new merged event (onsets ascend and fl(a - b) is monotone in b, so an offset from
// an already closed merged event can only be the maximum when the current one would break anyway);
// a merged event's offset is the maximum offset between two consecutive marks.
// One wavefront, one group of n onset-sorted events starting at element `base` of the input arrays (runs: frame
// pairs looked up in `table`; otherwise fp64 onsets/offsets); merged events go to element out_base + r of the output
// arrays.  Returns the number of merged events.
template <bool FROM_RUNS>
__device__ __forceinline__ int merge_wave(const double* onsets, const double* offsets, const int32_t* runs, int64_t base,
                                          int n, const double* table, int table_len, double gap, double* out_on,
                                          double* out_off, int32_t* out_runs, int64_t out_base, int lane) {
  int n_out = 0;
  double carry_pmax = -INFINITY;   // max of all offsets seen so far in the group
  double carry_smax = -INFINITY;   // max offset of the merged event that is still open
  int carry_sidx = 0;              // frame index behind carry_smax (FROM_RUNS)
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int i = i0 + lane;
    const bool valid = i < n;
    double on = INFINITY, off = -INFINITY;
    int on_k = 0, off_k = 0;
    if (valid) {
      if (FROM_RUNS) {
        on_k = runs[(base + i) * 2];
        off_k = runs[(base + i) * 2 + 1];
        on = table[min(max(on_k, 0), table_len - 1)];
        off = table[min(max(off_k, 0), table_len - 1)];
      } else {
        on = onsets[base + i];
        off = offsets[base + i];
      }
    }
    // exclusive prefix max of offsets over the whole group
    double inc = off;
    for (int d = 1; d < 64; d <<= 1) {
      const double o = __shfl_up(inc, d);
      if (lane >= d) inc = fmax(inc, o);
    }
    double exc = __shfl_up(inc, 1);
    if (lane == 0) exc = -INFINITY;
    exc = fmax(exc, carry_pmax);
    const bool head = valid && ((i == 0) || (__dsub_rn(on, exc) > gap));
    const uint64_t hm = __ballot(head);
    // forward max over the lanes of the same merged event inside this chunk
    const uint64_t below = hm & (lanemask_lt(lane) | (1ull << lane));
    const int seg_lo = below ? 63 - __clzll(below) : 0;
    double smax = off;
    int sidx = off_k;
    for (int d = 1; d < 64; d <<= 1) {
      const double o = __shfl_up(smax, d);
      const int oi = __shfl_up(sidx, d);
      if (lane - d >= seg_lo && o > smax) { smax = o; sidx = oi; }
    }
    if (!below && carry_smax > smax) { smax = carry_smax; sidx = carry_sidx; }  // continues the open event
    // a head closes the previous merged event with the running max of the lane before it
    double prev_smax = __shfl_up(smax, 1);
    int prev_sidx = __shfl_up(sidx, 1);
    if (lane == 0) { prev_smax = carry_smax; prev_sidx = carry_sidx; }
    if (head) {
      const int r = n_out + __popcll(hm & lanemask_lt(lane));
      if (FROM_RUNS) {
        out_runs[(out_base + r) * 2] = on_k;
        if (r >= 1) out_runs[(out_base + r - 1) * 2 + 1] = prev_sidx;
      } else {
        out_on[out_base + r] = on;
        if (r >= 1) out_off[out_base + r - 1] = prev_smax;
      }
    }
    n_out += __popcll(hm);
    const int last = min(63, n - 1 - i0);
    carry_pmax = fmax(carry_pmax, __shfl(inc, last));
    carry_smax = __shfl(smax, last);
    carry_sidx = __shfl(sidx, last);
  }
  if (lane == 0 && n_out >= 1) {
    if (FROM_RUNS) out_runs[(out_base + n_out - 1) * 2 + 1] = carry_sidx;
    else out_off[out_base + n_out - 1] = carry_smax;
  }
  return n_out;
}

template <bool FROM_RUNS>
__global__ void __launch_bounds__(256) merge_kernel(const double* onsets, const double* offsets,
                                                    const int32_t* group_start, const int32_t* runs,
                                                    const int32_t* counts, int cap, const double* table, int table_len,
                                                    int64_t n_groups, double gap, double* out_on, double* out_off,
                                                    int32_t* out_runs, int32_t* out_counts) {
  const int lane = threadIdx.x & 63;
  const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= n_groups) return;
  int64_t base;
  int n;
  if (FROM_RUNS) { base = g * (int64_t)cap; n = min(counts[g], cap); }
  else { base = group_start[g]; n = group_start[g + 1] - group_start[g]; }
  const int n_out = merge_wave<FROM_RUNS>(onsets, offsets, runs, base, n, table, table_len, gap, out_on, out_off,
                                          out_runs, base, lane);
  if (lane == 0) out_counts[g] = n_out;
}

// Threshold -> runs -> gap merge -> packed record in one launch: the raw runs of a trace stay in LDS.
// packed[trace] = (count, start_0, end_0, start_1, end_1, ...), 1 + 2*cap ints per trace.
// The LDS buffers are sized by the host for the trace length: kcap = T/2 + 1 run slots per wavefront (a trace of T
// frames has at most that many runs), as many wavefronts per workgroup (1..4) as 64 KB of LDS hold.
constexpr int kFusedMaxCap = 4096;     // run slots of the longest trace the fused kernels take: T <= 8190 frames
constexpr int kFusedLdsBytes = 65536;  // LDS a workgroup of the fused kernels may use
__global__ void __launch_bounds__(256) threshold_merge_kernel(const float* probs, int64_t n_traces, int T, int C,
                                                              float thr, int smoothing, const double* table,
                                                              int table_len, double gap, int32_t* packed, int cap,
                                                              int kcap) {
  extern __shared__ int32_t fused_lds[];  // [wavefront][2 * kcap] raw runs
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t trace = (int64_t)blockIdx.x * (blockDim.x >> 6) + w;
  if (trace >= n_traces) return;
  int32_t* raw = fused_lds + (size_t)w * 2 * kcap;
  const int64_t b = trace / C;
  const int c = (int)(trace - b * C);
  const int n = threshold_runs_wave(probs + b * (int64_t)T * C + c, T, C, thr, smoothing, raw, kcap, lane);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  int32_t* rec = packed + trace * (int64_t)(1 + 2 * cap);
  const int n_out = merge_wave<true>(nullptr, nullptr, raw, 0, min(n, kcap), table, table_len, gap, nullptr, nullptr,
                                     rec + 1, 0, lane);
  if (lane == 0) rec[0] = n_out;
}

// Event records of one stream for the exchange between GPUs: all classes' runs of a stream, class-major and
// onset-ascending (the order decode_predictions emits), as (stream, class, first frame, end frame) records behind a
// header row (records kept, records dropped, 0, 0).  One wavefront per stream; the class offsets are a running sum over
// the (few) classes, the copies are 64 events wide.  Every row of the output is written, so the buffer needs no clearing.
__global__ void __launch_bounds__(256) pack_events_kernel(const int32_t* runs, int64_t runs_stride, const int32_t* counts,
                                                          int64_t counts_stride, int64_t n_streams, int C, int cap_in,
                                                          int first_stream, int32_t* records, int cap_out) {
  const int lane = threadIdx.x & 63;
  const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= n_streams) return;
  int4* rec = reinterpret_cast<int4*>(records) + s * (int64_t)(cap_out + 1);
  int base = 0;
  for (int c = 0; c < C; ++c) {
    const int64_t trace = s * C + c;
    const int n = min(max(counts[trace * counts_stride], 0), cap_in);
    const int32_t* r = runs + trace * runs_stride;
    for (int k = lane; k < n && base + k < cap_out; k += 64)
      rec[1 + base + k] = make_int4((int)s + first_stream, c, r[2 * k], r[2 * k + 1]);
    base += n;
  }
  const int kept = min(base, cap_out);
  for (int k = kept + lane; k < cap_out; k += 64) rec[1 + k] = make_int4(0, 0, 0, 0);
  if (lane == 0) rec[0] = make_int4(kept, base - kept, 0, 0);
}

// threshold_merge_kernel and pack_events_kernel in one launch: one wavefront per stream walks the stream's classes,
// raw and merged runs of the class in hand stay in LDS, records go out 64 at a time.
__global__ void __launch_bounds__(256) threshold_merge_records_kernel(const float* probs, int64_t n_streams, int T, int C,
                                                                      float thr, int smoothing, const double* table,
                                                                      int table_len, double gap, int first_stream,
                                                                      int32_t* records, int cap_out, int kcap) {
  extern __shared__ int32_t fused_lds[];  // [wavefront][raw runs 2 * kcap | merged runs 2 * kcap]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t s = (int64_t)blockIdx.x * (blockDim.x >> 6) + w;
  if (s >= n_streams) return;
  int32_t* raw = fused_lds + (size_t)w * 4 * kcap;
  int32_t* merged = raw + 2 * kcap;
  int4* rec = reinterpret_cast<int4*>(records) + s * (int64_t)(cap_out + 1);
  int base = 0;
  for (int c = 0; c < C; ++c) {
    const int n = threshold_runs_wave(probs + s * (int64_t)T * C + c, T, C, thr, smoothing, raw, kcap, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int n_out = merge_wave<true>(nullptr, nullptr, raw, 0, min(n, kcap), table, table_len, gap, nullptr,
                                       nullptr, merged, 0, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int k = lane; k < n_out && base + k < cap_out; k += 64)
      rec[1 + base + k] = make_int4((int)s + first_stream, c, merged[2 * k], merged[2 * k + 1]);
    base += n_out;
    __builtin_amdgcn_wave_barrier();  // the next class overwrites both buffers
  }
  const int kept = min(base, cap_out);
  for (int k = kept + lane; k < cap_out; k += 64) rec[1 + k] = make_int4(0, 0, 0, 0);
  if (lane == 0) rec[0] = make_int4(kept, base - kept, 0, 0);
}

}  // namespace sed

using namespace sed;

static inline int hip_status(hipError_t e) { return e == hipSuccess ? SED_OK : SED_ERR_HIP - (int)e; }
static inline unsigned blocks_for(int64_t waves) { return (unsigned)((waves + 3) / 4); }
// Launch shape of the fused decoders: run slots per wavefront for traces of T frames, wavefronts per workgroup that
// `bytes_per_slot` bytes of LDS per slot allow (0: the trace is too long for the LDS buffers).
static inline int fused_waves(int64_t T, int bytes_per_slot, int* kcap) {
  *kcap = (int)std::max<int64_t>(256, T / 2 + 1);
  if (*kcap > kFusedMaxCap) return 0;
  return std::max(1, std::min(4, kFusedLdsBytes / (bytes_per_slot * *kcap)));
}

extern "C" int sed_threshold_runs(const float* probs, int64_t B, int64_t T, int64_t C, float threshold,
                                  int32_t smoothing, int32_t* runs, int32_t* counts, int32_t cap, void* stream) {
  if (B < 0 || T < 0 || C < 0 || cap < 1 || smoothing < 0 || smoothing > 2) return SED_ERR_BAD_ARG;
  if (T >= INT32_MAX - 64 || C > INT32_MAX || B * C > (int64_t)INT32_MAX * 4) return SED_ERR_BAD_ARG;
  if (B * C == 0) return SED_OK;
  if (!counts || !runs || (T > 0 && !probs)) return SED_ERR_BAD_ARG;
  hipLaunchKernelGGL(threshold_runs_kernel, dim3(blocks_for(B * C)), dim3(256), 0, (hipStream_t)stream, probs, B * C,
                     (int)T, (int)C, threshold, (int)smoothing, runs, counts, (int)cap);
  return hip_status(hipGetLastError());
}

extern "C" int sed_activity_edges(const int32_t* act, int64_t B, int64_t T, int64_t C, int32_t* starts, int32_t* ends,
                                  int32_t* n_starts, int32_t* n_ends, int32_t cap, void* stream) {
  if (B < 0 || T < 0 || C < 0 || cap < 1) return SED_ERR_BAD_ARG;
  if (T >= INT32_MAX - 64 || C > INT32_MAX || B * C > (int64_t)INT32_MAX * 4) return SED_ERR_BAD_ARG;
  if (B * C == 0) return SED_OK;
  if (!starts || !ends || !n_starts || !n_ends || (T > 0 && !act)) return SED_ERR_BAD_ARG;
  hipLaunchKernelGGL(activity_edges_kernel, dim3(blocks_for(B * C)), dim3(256), 0, (hipStream_t)stream, act, B * C,
                     (int)T, (int)C, starts, ends, n_starts, n_ends, (int)cap);
  return hip_status(hipGetLastError());
}

extern "C" int sed_argmax_runs(const float* probs, int64_t B, int64_t T, int64_t C, int32_t smoothing,
                               int32_t drop_last_classes, int32_t* runs, int32_t* counts, int32_t cap, void* stream) {
  if (B < 0 || T < 0 || C < 1 || cap < 1 || smoothing < 0 || smoothing > 2) return SED_ERR_BAD_ARG;
  if (drop_last_classes < 0 || drop_last_classes >= C || T >= INT32_MAX - 64 || C > INT32_MAX) return SED_ERR_BAD_ARG;
  if (B == 0) return SED_OK;
  if (!runs || !counts || (T > 0 && !probs)) return SED_ERR_BAD_ARG;
  hipLaunchKernelGGL(argmax_runs_kernel, dim3(blocks_for(B)), dim3(256), 0, (hipStream_t)stream, probs, B, (int)T,
                     (int)C, (int)smoothing, (int)drop_last_classes, runs, counts, (int)cap);
  return hip_status(hipGetLastError());
}

extern "C" int sed_argmax_histogram(const float* probs, int64_t B, int64_t T, int64_t C, int32_t drop_last_classes,
                                    int32_t* hist, void* stream) {
  if (B < 0 || T < 0 || C < 1 || drop_last_classes < 0 || drop_last_classes >= C) return SED_ERR_BAD_ARG;
  if (C - drop_last_classes > 64 || T >= INT32_MAX - 64) return SED_ERR_BAD_ARG;
  if (B == 0) return SED_OK;
  if (!hist || (T > 0 && !probs)) return SED_ERR_BAD_ARG;
  hipLaunchKernelGGL(argmax_hist_kernel, dim3(blocks_for(B)), dim3(256), 0, (hipStream_t)stream, probs, B, (int)T,
                     (int)C, (int)drop_last_classes, hist);
  return hip_status(hipGetLastError());
}

extern "C" int sed_merge_events(const double* onsets, const double* offsets, const int32_t* group_start,
                                int64_t n_groups, double gap, double* out_onsets, double* out_offsets,
                                int32_t* out_counts, void* stream) {
  if (n_groups < 0) return SED_ERR_BAD_ARG;
  if (n_groups == 0) return SED_OK;
  if (!onsets || !offsets || !group_start || !out_onsets || !out_offsets || !out_counts) return SED_ERR_BAD_ARG;
  hipLaunchKernelGGL(merge_kernel<false>, dim3(blocks_for(n_groups)), dim3(256), 0, (hipStream_t)stream, onsets,
                     offsets, group_start, nullptr, nullptr, 0, nullptr, 0, n_groups, gap, out_onsets, out_offsets,
                     nullptr, out_counts);
  return hip_status(hipGetLastError());
}

extern "C" int sed_merge_runs(const int32_t* runs, const int32_t* counts, int64_t n_groups, int32_t cap,
                              const double* time_table, int32_t table_len, double gap, int32_t* out_runs,
                              int32_t* out_counts, void* stream) {
  if (n_groups < 0 || cap < 1 || table_len < 1) return SED_ERR_BAD_ARG;
  if (n_groups == 0) return SED_OK;
  if (!runs || !counts || !time_table || !out_runs || !out_counts) return SED_ERR_BAD_ARG;
  hipLaunchKernelGGL(merge_kernel<true>, dim3(blocks_for(n_groups)), dim3(256), 0, (hipStream_t)stream, nullptr,
                     nullptr, nullptr, runs, counts, (int)cap, time_table, (int)table_len, n_groups, gap, nullptr,
                     nullptr, out_runs, out_counts);
  return hip_status(hipGetLastError());
}

extern "C" int sed_threshold_merge_runs(const float* probs, int64_t B, int64_t T, int64_t C, float threshold,
                                        int32_t smoothing, const double* time_table, int32_t table_len, double gap,
                                        int32_t* packed, int32_t cap, void* stream) {
  if (B < 0 || T < 0 || C < 0 || cap < 1 || smoothing < 0 || smoothing > 2 || table_len < 1) return SED_ERR_BAD_ARG;
  if (T >= INT32_MAX - 64 || C > INT32_MAX || B * C > (int64_t)INT32_MAX * 4) return SED_ERR_BAD_ARG;
  int kcap;
  const int waves = fused_waves(T, 8, &kcap);
  if (!waves || T / 2 + 1 > cap) return SED_ERR_CAPACITY;  // every raw run must fit the LDS buffer and the record
  if (B * C == 0) return SED_OK;
  if (!packed || !time_table || (T > 0 && !probs)) return SED_ERR_BAD_ARG;
  hipLaunchKernelGGL(threshold_merge_kernel, dim3((unsigned)((B * C + waves - 1) / waves)), dim3(64 * waves),
                     (size_t)waves * 8 * kcap, (hipStream_t)stream, probs, B * C, (int)T, (int)C, threshold, (int)smoothing,
                     time_table, (int)table_len, gap, packed, (int)cap, kcap);
  return hip_status(hipGetLastError());
}

extern "C" int sed_threshold_merge_records(const float* probs, int64_t B, int64_t T, int64_t C, float threshold,
                                           int32_t smoothing, const double* time_table, int32_t table_len, double gap,
                                           int32_t first_stream, int32_t* records, int32_t cap_out, void* stream) {
  if (B < 0 || T < 0 || C < 0 || cap_out < 1 || smoothing < 0 || smoothing > 2 || table_len < 1) return SED_ERR_BAD_ARG;
  if (T >= INT32_MAX - 64 || C > INT32_MAX || B + (int64_t)first_stream > INT32_MAX) return SED_ERR_BAD_ARG;
  int kcap;
  const int waves = fused_waves(T, 16, &kcap);
  if (!waves) return SED_ERR_CAPACITY;  // every raw run of a trace must fit the LDS buffer
  if (B == 0) return SED_OK;
  if (!records || !time_table || (T > 0 && C > 0 && !probs)) return SED_ERR_BAD_ARG;
  hipLaunchKernelGGL(threshold_merge_records_kernel, dim3((unsigned)((B + waves - 1) / waves)), dim3(64 * waves),
                     (size_t)waves * 16 * kcap, (hipStream_t)stream, probs, B, (int)T, (int)C, threshold, (int)smoothing,
                     time_table, (int)table_len, gap, (int)first_stream, records, (int)cap_out, kcap);
  return hip_status(hipGetLastError());
}

extern "C" int sed_pack_events(const int32_t* runs, int64_t runs_stride, const int32_t* counts, int64_t counts_stride,
                               int64_t n_streams, int64_t C, int32_t cap_in, int32_t first_stream, int32_t* records,
                               int32_t cap_out, void* stream) {
  if (n_streams < 0 || C < 0 || cap_in < 0 || cap_out < 1 || runs_stride < 0 || counts_stride < 0) return SED_ERR_BAD_ARG;
  if (C > INT32_MAX || n_streams + (int64_t)first_stream > INT32_MAX) return SED_ERR_BAD_ARG;
  if (n_streams == 0) return SED_OK;
  if (!records || (C > 0 && (!runs || !counts))) return SED_ERR_BAD_ARG;
  hipLaunchKernelGGL(pack_events_kernel, dim3(blocks_for(n_streams)), dim3(256), 0, (hipStream_t)stream, runs, runs_stride,
                     counts, counts_stride, n_streams, (int)C, (int)cap_in, (int)first_stream, records, (int)cap_out);
  return hip_status(hipGetLastError());
}

// ---- detector -> classifier hand-off: event ranges of the resident PCM batch into a zero-padded batch ----------
// grid.y = event, grid.x x 256 threads x 4 samples cover the row; every output sample is written exactly once
// (16-byte stores; the source is only sample-aligned, so it is read element by element through L1/L2).
template <typename T>
__global__ void __launch_bounds__(256) cut_events_kernel(const T* pcm, float div, int64_t n_streams, int64_t L,
                                                         int64_t pcm_stride, const int32_t* ev_stream,
                                                         const int64_t* ev_start, const int64_t* ev_end, float* out,
                                                         int64_t cap, int32_t* out_len) {
  const int64_t e = blockIdx.y;
  int64_t a = ev_start[e], b = ev_end[e];
  if (a < 0) a += L;
  if (b < 0) b += L;
  a = a < 0 ? 0 : (a > L ? L : a);
  b = b < 0 ? 0 : (b > L ? L : b);
  const int sidx = ev_stream[e];
  const bool ok = sidx >= 0 && sidx < n_streams;
  const int64_t len = (ok && b > a) ? b - a : 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) out_len[e] = (int32_t)(len > INT32_MAX ? INT32_MAX : len);
  const T* src = pcm + (ok ? (int64_t)sidx * pcm_stride + a : 0);
  float* dst = out + e * cap;
  const bool vec = (cap % 4 == 0) && (((uintptr_t)out) % 16 == 0);
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < cap; i += (int64_t)gridDim.x * 1024) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t j = i + k;
      float x = 0.0f;
      if (j < len) x = std::is_same<T, float>::value ? (float)src[j] : (float)src[j] / div;
      v[k] = x;
    }
    if (vec) {
      *reinterpret_cast<float4*>(dst + i) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (i + k < cap) dst[i + k] = v[k];
    }
  }
}

extern "C" int sed_cut_events(const void* pcm, int32_t pcm_kind, float pcm_divisor, int64_t n_streams, int64_t L,
                              int64_t pcm_stride, const int32_t* ev_stream, const int64_t* ev_start,
                              const int64_t* ev_end, int64_t n_events, float* out, int64_t cap, int32_t* out_len,
                              void* stream) {
  if (n_streams < 0 || L < 0 || pcm_stride < L || n_events < 0 || cap < 0) return SED_ERR_BAD_ARG;
  if (pcm_kind != SED_PCM_F32 && pcm_kind != SED_PCM_I16) return SED_ERR_BAD_ARG;
  if (pcm_kind == SED_PCM_I16 && !(pcm_divisor > 0.0f)) return SED_ERR_BAD_ARG;
  if (n_events == 0) return SED_OK;
  if (!ev_stream || !ev_start || !ev_end || !out_len || (cap > 0 && !out) || (n_streams > 0 && L > 0 && !pcm))
    return SED_ERR_BAD_ARG;
  const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>((cap + 1023) / 1024, 64));
  for (int64_t e0 = 0; e0 < n_events; e0 += 65535) {  // grid.y limit
    const dim3 grid(gx, (unsigned)std::min<int64_t>(65535, n_events - e0));
    if (pcm_kind == SED_PCM_F32)
      hipLaunchKernelGGL(cut_events_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)pcm, 1.0f,
                         n_streams, L, pcm_stride, ev_stream + e0, ev_start + e0, ev_end + e0, out + e0 * cap, cap,
                         out_len + e0);
    else
      hipLaunchKernelGGL(cut_events_kernel<int16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const int16_t*)pcm,
                         pcm_divisor, n_streams, L, pcm_stride, ev_stream + e0, ev_start + e0, ev_end + e0,
                         out + e0 * cap, cap, out_len + e0);
  }
  return hip_status(hipGetLastError());
}

// ---- HTS-AT glue: log-mel (B, T, M) -> square image (B, 1, freq_ratio*M, target_T/freq_ratio) -----------------
// reshape_wav2img: bicubic interpolation along time to target_T frames (torch's upsample_bicubic2d with
// align_corners=True, A = -0.75, border indices clamped; the mel axis keeps its size, so its taps are (0,1,0,0)),
// then the time axis is folded freq_ratio times under the mel axis: img[b][r*M + f][u] = y[b][r*(target_T/ratio) + u][f].
// repeat_wat2img is the same kernel with every band r showing the same window: img[b][r*M + f][u] = y[b][cur_pos + u][f].
// In general band r shows the interpolated frames col0 + r*col_step + [0, seg).
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A; }

// frames (optional): clip b holds frames[b] of the T rows of its slot, and its interpolation runs over those alone.
__global__ void __launch_bounds__(256) wav2img_kernel(const float* x, int T, int M, int target_T, int ratio, int seg,
                                                      int col0, int col_step, float scale, float* img,
                                                      const int32_t* frames) {
  extern __shared__ float rows[];  // input frames [n_rows][M + 1]
  const int tiles = (seg + 63) / 64;
  const int tile = blockIdx.x % tiles, r = blockIdx.x / tiles, b = blockIdx.y;
  const int u0 = tile * 64, t0 = col0 + r * col_step + u0;  // first interpolated frame of this tile
  const int n_out = min(64, seg - u0);
  const int slot = T;  // rows between clips
  if (frames) {
    T = min(max(frames[b], 0), slot);
    if (T == 0) {  // a clip without frames: zeros
      for (int i = threadIdx.x; i < n_out * M; i += 256)
        img[(((int64_t)b * ratio + r) * M + i / n_out) * seg + u0 + i % n_out] = 0.0f;
      return;
    }
    scale = target_T > 1 ? (float)(T - 1) / (float)(target_T - 1) : 0.0f;
  }
  const int lo = max(0, (int)floorf(scale * (float)t0) - 1);
  const int hi = min(T - 1, (int)floorf(scale * (float)(t0 + n_out - 1)) + 2);
  const int n_rows = hi - lo + 1, pitch = M + 1;
  const float* src = x + ((int64_t)b * slot + lo) * M;
  // staging: 16 bytes per load when the rows allow it (the tile's rows are contiguous in memory)
  if (M % 4 == 0 && ((uintptr_t)src & 15) == 0) {
    for (int i4 = threadIdx.x; i4 < n_rows * (M / 4); i4 += 256) {
      const float4 v = reinterpret_cast<const float4*>(src)[i4];
      const int row = i4 / (M / 4), col = 4 * (i4 % (M / 4));
      float* d = rows + row * pitch + col;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  } else {
    for (int i = threadIdx.x; i < n_rows * M; i += 256) rows[(i / M) * pitch + i % M] = src[i];
  }
  __syncthreads();
  const float A = -0.75f;
  if (n_out == 64) {
    // full tile: a thread keeps its image column (u = thread % 64), so its source frames and the four weights are formed
    // once and the loop over the mel rows is four LDS reads, the weighted sum and one store (64 consecutive columns =
    // 256 bytes per wavefront)
    const int u = threadIdx.x & 63, f0 = threadIdx.x >> 6;
    const int t = t0 + u;
    float* o = img + (((int64_t)b * ratio + r) * M) * seg + u0 + u;
    if (T == target_T) {
      const float* p0 = rows + (t - lo) * pitch;
      for (int f = f0; f < M; f += 4) o[(int64_t)f * seg] = p0[f];
    } else {
      const float real = scale * (float)t;
      const float fl = floorf(real);
      const int k = (int)fl;
      const float tt = real - fl;
      const float c0 = cubic2(tt + 1.0f, A), c1 = cubic1(tt, A), c2 = cubic1(1.0f - tt, A), c3 = cubic2(2.0f - tt, A);
      const float* p0 = rows + (min(max(k - 1, 0), T - 1) - lo) * pitch;
      const float* p1 = rows + (min(max(k, 0), T - 1) - lo) * pitch;
      const float* p2 = rows + (min(max(k + 1, 0), T - 1) - lo) * pitch;
      const float* p3 = rows + (min(max(k + 2, 0), T - 1) - lo) * pitch;
      for (int f = f0; f < M; f += 4) o[(int64_t)f * seg] = p0[f] * c0 + p1[f] * c1 + p2[f] * c2 + p3[f] * c3;
    }
    return;
  }
  for (int i = threadIdx.x; i < n_out * M; i += 256) {
    const int u = i % n_out, f = i / n_out;     // consecutive threads -> consecutive image columns
    const int t = t0 + u;
    float v;
    if (T == target_T) {
      v = rows[(t - lo) * pitch + f];
    } else {
      const float real = scale * (float)t;
      const float fl = floorf(real);
      const int k = (int)fl;
      const float tt = real - fl;
      const float c0 = cubic2(tt + 1.0f, A), c1 = cubic1(tt, A), c2 = cubic1(1.0f - tt, A), c3 = cubic2(2.0f - tt, A);
      const float v0 = rows[(min(max(k - 1, 0), T - 1) - lo) * pitch + f];
      const float v1 = rows[(min(max(k, 0), T - 1) - lo) * pitch + f];
      const float v2 = rows[(min(max(k + 1, 0), T - 1) - lo) * pitch + f];
      const float v3 = rows[(min(max(k + 2, 0), T - 1) - lo) * pitch + f];
      v = v0 * c0 + v1 * c1 + v2 * c2 + v3 * c3;
    }
    img[(((int64_t)b * ratio + r) * M + f) * seg + u0 + u] = v;
  }
}

// One launch per 65535 clips (the grid's second dimension): clip b of a chunk is blockIdx.y.
static void launch_wav2img(const float* logmel, const int32_t* frames, int64_t B, int64_t T, int64_t M, int64_t target_T,
                           int bands, int seg, int col0, int col_step, float scale, float* img, hipStream_t stream) {
  const int tiles = (seg + 63) / 64;
  const size_t lds = (size_t)(64 + 6) * (M + 1) * sizeof(float);
  for (int64_t b0 = 0; b0 < B; b0 += 65535) {
    const int64_t nb = std::min<int64_t>(65535, B - b0);
    hipLaunchKernelGGL(wav2img_kernel, dim3((unsigned)(tiles * bands), (unsigned)nb), dim3(256), lds, stream,
                       logmel + b0 * T * M, (int)T, (int)M, (int)target_T, bands, seg, col0, col_step, scale,
                       img + b0 * bands * M * seg, frames ? frames + b0 : (const int32_t*)nullptr);
  }
}

extern "C" int sed_htsat_wav2img(const float* logmel, int64_t B, int64_t T, int64_t M, int64_t target_T,
                                 int32_t freq_ratio, float* img, void* stream) {
  if (B < 0 || T < 1 || M < 1 || M > 512 || freq_ratio < 1 || target_T < T || target_T % freq_ratio != 0)
    return SED_ERR_BAD_ARG;
  if (target_T > INT32_MAX / 2) return SED_ERR_BAD_ARG;
  if (B == 0) return SED_OK;
  if (!logmel || !img) return SED_ERR_BAD_ARG;
  const float scale = target_T > 1 ? (float)(T - 1) / (float)(target_T - 1) : 0.0f;
  const int seg = (int)(target_T / freq_ratio);
  launch_wav2img(logmel, nullptr, B, T, M, target_T, (int)freq_ratio, seg, 0, seg, scale, img, (hipStream_t)stream);
  return hip_status(hipGetLastError());
}

extern "C" int sed_htsat_wav2img_ragged(const float* logmel, const int32_t* frames, int64_t B, int64_t T, int64_t M,
                                        int64_t target_T, int32_t freq_ratio, float* img, void* stream) {
  if (B < 0 || T < 1 || M < 1 || M > 512 || freq_ratio < 1 || target_T < T || target_T % freq_ratio != 0)
    return SED_ERR_BAD_ARG;
  if (target_T > INT32_MAX / 2) return SED_ERR_BAD_ARG;
  if (B == 0) return SED_OK;
  if (!logmel || !img || !frames) return SED_ERR_BAD_ARG;
  const int seg = (int)(target_T / freq_ratio);
  launch_wav2img(logmel, frames, B, T, M, target_T, (int)freq_ratio, seg, 0, seg, 0.0f, img, (hipStream_t)stream);
  return hip_status(hipGetLastError());
}

extern "C" int sed_htsat_repeat2img(const float* logmel, int64_t B, int64_t T, int64_t M, int64_t target_T,
                                    int64_t spec_size, int64_t cur_pos, int32_t repeats, float* img, void* stream) {
  if (B < 0 || T < 1 || M < 1 || M > 512 || repeats < 1 || target_T < T || spec_size < 1) return SED_ERR_BAD_ARG;
  if (target_T > INT32_MAX / 2) return SED_ERR_BAD_ARG;
  if (cur_pos < 0 || cur_pos + spec_size > target_T) return SED_ERR_BAD_ARG;
  if (B == 0) return SED_OK;
  if (!logmel || !img) return SED_ERR_BAD_ARG;
  const float scale = target_T > 1 ? (float)(T - 1) / (float)(target_T - 1) : 0.0f;
  launch_wav2img(logmel, nullptr, B, T, M, target_T, (int)repeats, (int)spec_size, (int)cur_pos, 0, scale, img,
                 (hipStream_t)stream);
  return hip_status(hipGetLastError());
}

// ---- framewise output of the detector back on the input's frame grid ------------------------------------------
// interpolate(x, ratio) followed by pad_framewise_output(., frames_out): out[b][t][c] = x[b][min(t / ratio, T-1)][c].
__global__ void __launch_bounds__(256) framewise_expand_any_kernel(const float* x, int64_t T, int64_t C, int ratio,
                                                               int64_t frames_out, int64_t total, float* out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t c = i % C, bt = i / C;
    const int64_t t = bt % frames_out, b = bt / frames_out;
    const int64_t ts = min(t / ratio, T - 1);
    out[i] = x[(b * T + ts) * C + c];
  }
}

// One workgroup per (source frame ts, clip b): the frame's C values go to LDS once and are written out `reps` times in a
// row (ratio copies, and for the last frame also the padding up to frames_out), a contiguous run of reps * C floats.
// The class index of a thread's element advances by 256 mod C from one store to the next, so there is no division in
// the loop (the element-per-thread kernel above, four 64-bit divisions per element, reaches 1.4 .. 1.8 TB/s).
__global__ void __launch_bounds__(256) framewise_expand_kernel(const float* x, int T, int C, int ratio, int64_t frames_out,
                                                               int G, float* out) {
  // G > 1 (few classes): the workgroup takes G consecutive source frames, so that it has a few thousand floats to write
  extern __shared__ float row[];
  const int ts0 = blockIdx.x * G, b = blockIdx.y;
  const int nts = min(G, T - ts0);
  const float* src = x + ((int64_t)b * T + ts0) * C;
  for (int c = threadIdx.x; c < nts * C; c += 256) row[c] = src[c];
  __syncthreads();
  const int64_t first = (int64_t)ts0 * ratio;                                            // first output frame of the chunk
  const int64_t last = ts0 + nts == T ? frames_out : min(first + (int64_t)nts * ratio, frames_out);  // one past its last
  if (last <= first) return;
  const int64_t n = (last - first) * C;
  float* o = out + ((int64_t)b * frames_out + first) * C;
  if (G == 1) {
    const int step = 256 % C;
    int c = threadIdx.x % C;
    for (int64_t j = threadIdx.x; j < n; j += 256) {
      o[j] = row[c];
      c += step;
      if (c >= C) c -= C;
    }
  } else {  // n is small here (the host picks G for about 4096 floats, the padding of the last chunk aside)
    for (int64_t j = threadIdx.x; j < n; j += 256) {
      const int q = (int)(j / C), c = (int)(j - (int64_t)q * C);  // output frame of the chunk, class
      o[j] = row[min(q / ratio, nts - 1) * C + c];
    }
  }
}

extern "C" int sed_framewise_expand(const float* x, int64_t B, int64_t T, int64_t C, int32_t ratio, int64_t frames_out,
                                    float* out, void* stream) {
  if (B < 0 || T < 1 || C < 1 || ratio < 1 || frames_out < 0) return SED_ERR_BAD_ARG;
  const int64_t total = B * frames_out * C;
  if (total == 0) return SED_OK;
  if (!x || !out) return SED_ERR_BAD_ARG;
  if (B <= 65535 && T <= INT32_MAX / 2 && C <= 8192) {
    const int64_t per_frame = (int64_t)ratio * C;
    const int G = per_frame >= 2048 ? 1 : (int)std::min<int64_t>(T, std::max<int64_t>(1, 4096 / per_frame));
    hipLaunchKernelGGL(framewise_expand_kernel, dim3((unsigned)((T + G - 1) / G), (unsigned)B), dim3(256),
                       (size_t)G * C * sizeof(float), (hipStream_t)stream, x, (int)T, (int)C, (int)ratio, frames_out, G, out);
  } else {
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(framewise_expand_any_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, T, C, (int)ratio,
                       frames_out, total, out);
  }
  return hip_status(hipGetLastError());
}
