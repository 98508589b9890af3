// Fused log-mel front-end for gfx950 (MI355X): framing + Hann window + real FFT + power/magnitude +
// mel projection + log, one launch per batch, the (T, F) spectrogram never leaves the CU.
//
// Work decomposition
//   workgroup  : WAVES wavefronts, one workgroup per CU, tables in LDS for the whole launch; after the
//                table load there is no workgroup barrier: every wavefront streams on its own
//   wavefront  : owns "chunks" = runs of consecutive frames of one clip and walks them FPW frames at a
//                time.  The PCM span of the FPW frames is copied into the wavefront's LDS region (LDS-DMA for
//                aligned fp32 clips, issued one group ahead), which then serves as exchange buffer, power rows and
//                partial-sum area of the same group
//   per frame  : LPF lanes; see fft_core.h for the lane decomposition
//   mel        : power rows are multiplied by the banded filterbank on the matrix pipe
//                (v_mfma_f32_4x4x1_16b_f32: 16 blocks x [4 filters x 4 frames] per instruction, exact f32)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <new>
#include <utility>
#include <vector>

#include "sed_hip.h"
#include "tables.h"

namespace sed {

using Geo512 = Geo<16, 16, 8, 12>;
using Geo1024 = Geo<32, 16, 16, 12>;
using Geo2048 = Geo<32, 32, 16, 8>;

#if defined(SED_TIMELINE) || defined(SED_X_MASK) || defined(SED_STAMPS)
#define SED_DIAGNOSTIC_BUILD 1  // diagnostic builds read SED_DEBUG_WAVES from the environment
#endif

#ifndef SED_WPE
#define SED_WPE(x) (x)   // experiments: -DSED_WPE(x)=4 asks the register allocator for 4 wavefronts per SIMD
#endif

#ifndef SED_BURST2048
#define SED_BURST2048 24
#endif
constexpr int kMaxMelSets = 4;
constexpr int kMelBurst = 8;    // steps of 4 bins whose MFMA operands are fetched from LDS in one burst
constexpr int kPartStride = 20;  // floats between the partial sums of consecutive positions: [frame][filter] + pad (4*odd: bank spread)
constexpr int kMaxSpanBlocks = 32;  // 1 KB blocks of a staged span: (FPW - 1) * hop + n_fft <= 4 * 2048 samples

// The tables of a plan as one image in global memory, laid out exactly as the kernel keeps them in LDS (so the
// prologue is a straight 16-byte copy): [window | inter-stage twiddles | split twiddles] (lane-major, see Geo), then
// for the fused kernel [mel weights | per-lane B-operand offsets | per-lane gather words | affine | 4 zeros].
struct ImageLayout {
  int32_t off_tw1, off_tw2, off_mela, off_boff, off_gw, off_aff, off_zero, floats_spec, floats_fused;
  int32_t aff_stride;  // floats between the scale and the shift table: 64 or 128 (n_mels rounded up)
};

struct FrontParams {
  const void* pcm;
  float* out;
  const float* image;
  int64_t L, pcm_stride, out_stride_b;
  int32_t T, hop, pad_left, pad_mode, pcm_kind;
  float pcm_div;
  int32_t n_mels, log_kind, affine;
  float amin, db_offset, ln_offset;
  // per-frame conditioning of the Kaldi flavour: DC removal over win_length samples and pre-emphasis
  int32_t remove_dc, win_length, pre_off;  // pre_off: floats from the start of the wave region to its pooling scratch
  float preemph, inv_win_length;
  // work units: a group = FPW consecutive frames of one clip
  int32_t groups_per_clip, total_groups;
  // per-wave LDS region, in floats: [staged span, then exchange rows, then power rows | mel partial sums at part_off]
  int32_t span, span_blocks, dma, early, wave_floats, part_off;
  int32_t image_floats, off_queue, off_wave;
  ImageLayout im;
  int32_t mel_sets, mel_u, mel_ulo, mel_nhi, mel_maxpos, n_quads, out_vec, aff_stride;
  // segment mode (SEG kernels): clip b is seg_len[b] samples starting at element seg_off[b] of pcm, followed by zeros up to L
  const int64_t* seg_off;
  const int32_t* seg_len;
  // seg_own: every segment is framed as a recording of its own length (padding rule at its own end, zero rows behind its
  // own frames): a segment of n samples has 0 frames if n < own_min, else 1 + (n - own_sub) / hop
  int32_t seg_own, own_min, own_sub;
  unsigned long long* stamps;  // SED_TIMELINE builds only
};

// In-kernel phase timing (diagnostic build, -DSED_STAMPS): cycles per phase summed per wavefront.
#ifdef SED_STAMPS
#define SED_STAMP_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_t = sed_stamp();
#define SED_STAMP(k) { const unsigned long long t_ = sed_stamp(); st_acc[k] += t_ - st_t; st_t = t_; }
__device__ __forceinline__ unsigned long long sed_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#else
#define SED_STAMP_DECL
#define SED_STAMP(k)
#endif

// Orders LDS accesses of different lanes of ONE wavefront.  The hardware executes a wavefront's LDS operations in
// the order they were issued, so no wait is needed: this only stops the compiler from moving memory operations
// across the phase boundary (a fence instruction would make it wait for every outstanding LDS access).
__device__ __forceinline__ void wave_phase_fence() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

// log2 on the transcendental unit (v_log_f32, about 1 ulp of the log2 value), scaled to dB or nepers: the
// arguments are clamped or offset away from zero, so no denormal handling is needed.
__device__ __forceinline__ float log_epilogue(float acc, int log_kind, float amin, float db_offset, float ln_offset) {
  if (log_kind == SED_LOG_DB10) return 3.0102999566398120f * __log2f(fmaxf(acc, amin)) - db_offset;
  if (log_kind == SED_LOG_LN) return 0.6931471805599453f * __log2f(acc + ln_offset);
  if (log_kind == SED_LOG_LN_FLOOR) return 0.6931471805599453f * __log2f(fmaxf(acc, amin));
  return acc;
}

// Flag in FrontParams::pcm_kind: multiply + two fused steps give the bits of the division for every int16 value
// (checked on the host when the plan is made).
constexpr int kPcmDiv3 = 0x100;

// int16 sample -> fp32 as the reference does it (x / 32767. in fp32).  For divisors the host has checked over all
// 65536 sample values, q = x * rcp, then q + (x - q * div) * rcp with fused steps is the correctly rounded quotient:
// three instructions instead of the ten of an IEEE division.
__device__ __forceinline__ float pcm16_to_f32(float x, float div, float rcp) {
  const float q = x * rcp;
  return __builtin_fmaf(__builtin_fmaf(-q, div, x), rcp, q);
}

// Staging copy of a span that touches a clip edge (reflect / zero padding), runs past the stored samples of a segment, or
// belongs to an int16 plan whose divisor needs the IEEE division: every sample is mapped to its source index on its own.
// Kept out of line so that its address arithmetic stays out of the streaming loop, but batched like the interior path: a
// lane forms the source indices of 8 x 4 samples, issues all 32 loads with clamped (never predicated) addresses, and only
// then selects and stores, so the memory latency is paid once per batch and not once per sample.
template <typename T>
__device__ __attribute__((noinline)) void stage_span_edges(const T* src, float div, int64_t L, int64_t valid, int pad_mode,
                                                           float* dst, int64_t s0, int count, int lane, int hop, int skew) {
  // sample i of the span goes to float i + skew * (i / hop) (skew 0: the plain layout); groups of 4 never straddle a hop
  if (valid <= 0) {  // a clip without stored samples: zeros, and no load at all
    for (int i = 4 * lane; i < count; i += 256) st4(dst + i + (skew ? skew * (i / hop) : 0), v4{0.0f, 0.0f, 0.0f, 0.0f});
    return;
  }
  for (int base = 0; base < count; base += 8 * 256) {
    float v[8][4];
    static_for<0, 8>([&](auto Kf) {
      constexpr int k = decltype(Kf)::value;
      const int i = min(base + 256 * k + 4 * lane, count - 4);
      static_for<0, 4>([&](auto Ef) {
        constexpr int e = decltype(Ef)::value;
        int64_t sidx = s0 + i + e;
        if (sidx < 0) sidx = (pad_mode == SED_PAD_REFLECT) ? -sidx : -1;
        else if (sidx >= L) sidx = (pad_mode == SED_PAD_REFLECT) ? 2 * (L - 1) - sidx : -1;
        const bool ok = sidx >= 0 && sidx < valid;
        const T raw = src[ok ? sidx : 0];
        float x;
        if constexpr (std::is_same<T, float>::value) x = raw;
        else x = (float)raw / div;
        v[k][e] = ok ? x : 0.0f;
      });
    });
    static_for<0, 8>([&](auto Kf) {
      constexpr int k = decltype(Kf)::value;
      const int i = min(base + 256 * k + 4 * lane, count - 4);
      st4(dst + i + (skew ? skew * (i / hop) : 0), v4{v[k][0], v[k][1], v[k][2], v[k][3]});
    });
  }
}

// 16- and 8-byte global loads from addresses that are only aligned to the sample size (global memory accepts them).
struct __attribute__((packed, aligned(4))) f32x4_u { float x, y, z, w; };
struct __attribute__((packed, aligned(2))) i16x4_u { short x, y, z, w; };

// One wavefront copies `count` samples (multiple of 4) of clip b, starting at padded-clip sample s0, to LDS.
// The fast path issues its loads in batches of 8 x 4 samples per lane before the first LDS store, with clamped
// (never predicated) addresses: a load under a divergent branch would be waited for on its own.
// HOP / SKEW: the skewed layout (groups of 4 samples never straddle a hop: both are multiples of 4).
// The clip starts at element clip_off of the PCM buffer and holds `valid` stored samples (then zeros up to clipL, the
// length the padding rule is applied to).
template <bool I16, int HOP = 0, int SKEW = 0>
__device__ __forceinline__ void stage_span(const FrontParams& p, float* dst, int64_t clip_off, int64_t valid, int64_t clipL,
                                           int64_t s0, int count, int lane) {
  auto at = [](int i) { if constexpr (SKEW > 0) return i + SKEW * (i / HOP); else return i; };
  // (int16 plans whose divisor failed the three-instruction check take the out-of-line path with the IEEE division)
  bool fast = s0 >= 0 && s0 + count <= valid;
  if constexpr (I16) fast = fast && (p.pcm_kind & kPcmDiv3);
  if (!fast) {
    if constexpr (I16) stage_span_edges<int16_t>((const int16_t*)p.pcm + clip_off, p.pcm_div, clipL, valid, p.pad_mode, dst, s0, count,
                                                 lane, SKEW > 0 ? HOP : 1, SKEW);
    else stage_span_edges<float>((const float*)p.pcm + clip_off, 1.0f, clipL, valid, p.pad_mode, dst, s0, count, lane,
                                 SKEW > 0 ? HOP : 1, SKEW);
  } else if constexpr (!I16) {
    const float* src = (const float*)p.pcm + clip_off + s0;
    for (int base = 0; base < count; base += 8 * 256) {
      f32x4_u v[8];
      static_for<0, 8>([&](auto Kf) {
        v[decltype(Kf)::value] = *reinterpret_cast<const f32x4_u*>(src + min(base + 256 * decltype(Kf)::value + 4 * lane, count - 4));
      });
      static_for<0, 8>([&](auto Kf) {
        const f32x4_u w = v[decltype(Kf)::value];
        st4(dst + at(min(base + 256 * decltype(Kf)::value + 4 * lane, count - 4)), v4{w.x, w.y, w.z, w.w});
      });
    }
  } else {
    const int16_t* src = (const int16_t*)p.pcm + clip_off + s0;
    const float rcp = 1.0f / p.pcm_div;  // correctly rounded, as the host's check assumed
    for (int base = 0; base < count; base += 8 * 256) {
      i16x4_u v[8];
      static_for<0, 8>([&](auto Kf) {
        v[decltype(Kf)::value] = *reinterpret_cast<const i16x4_u*>(src + min(base + 256 * decltype(Kf)::value + 4 * lane, count - 4));
      });
      static_for<0, 8>([&](auto Kf) {
        const i16x4_u w = v[decltype(Kf)::value];
        st4(dst + at(min(base + 256 * decltype(Kf)::value + 4 * lane, count - 4)),
            v4{pcm16_to_f32((float)w.x, p.pcm_div, rcp), pcm16_to_f32((float)w.y, p.pcm_div, rcp),
               pcm16_to_f32((float)w.z, p.pcm_div, rcp), pcm16_to_f32((float)w.w, p.pcm_div, rcp)});
      });
    }
  }
}

// Interior spans of 16-byte-aligned fp32 clips go from global memory straight to LDS (no registers in between),
// 16 bytes per lane and 1 KB per instruction, every sample fetched once per group of frames.  The caller has checked
// that span_blocks * 256 samples from s0 on lie inside the clip, so the last block simply lands past the span.
// One per-lane address serves four instructions (the 13-bit instruction offset moves the source and the LDS
// destination together).  Completion: vmcnt.
__device__ __forceinline__ void stage_span_dma(const FrontParams& p, float* dst, int64_t clip_off, int64_t s0, int lane) {
  const float* src = (const float*)p.pcm + clip_off + s0 + 4 * lane;
  // a rolled loop over groups of four blocks keeps the scalar state to a pointer pair and a counter
  int left = p.span_blocks;
#pragma nounroll
  for (; left >= 4; left -= 4, src += 1024, dst += 1024) {
    const auto* gs = (const __attribute__((address_space(1))) void*)src;
    auto* ld = (__attribute__((address_space(3))) void*)dst;
    __builtin_amdgcn_global_load_lds(gs, ld, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(gs, ld, 16, 1024, 0);
    __builtin_amdgcn_global_load_lds(gs, ld, 16, 2048, 0);
    __builtin_amdgcn_global_load_lds(gs, ld, 16, 3072, 0);
  }
  if (left > 0) {
    const auto* gs = (const __attribute__((address_space(1))) void*)src;
    auto* ld = (__attribute__((address_space(3))) void*)dst;
    __builtin_amdgcn_global_load_lds(gs, ld, 16, 0, 0);
    if (left > 1) __builtin_amdgcn_global_load_lds(gs, ld, 16, 1024, 0);
    if (left > 2) __builtin_amdgcn_global_load_lds(gs, ld, 16, 2048, 0);
  }
}

// The same for the skewed layout: hop h of the span (HOP samples; the last one NFFT - ... samples short) lands at float
// h * (HOP + SKEW), as 16-byte pieces of 256 samples and 4-byte pieces of 64.  The instruction offset moves source and
// destination together and the LDS base (M0) takes the rest, so two per-lane addresses serve up to four hops.
template <class G, int HOP, int SKEW>
__device__ __forceinline__ void stage_span_dma_skewed(const FrontParams& p, float* dst, int64_t clip_off, int64_t s0, int lane) {
  constexpr int SPAN = (G::FPW - 1) * HOP + G::NFFT;
#ifdef SED_SPAN_HOPS  // timing experiments: fetch only the first hops of the span (results are then garbage)
  constexpr int HOPS = SED_SPAN_HOPS;
#else
  constexpr int HOPS = (SPAN + HOP - 1) / HOP;
#endif
  static_assert(HOP % 64 == 0 && SPAN % 64 == 0 && HOP * 4 * 3 + 4096 > 0, "whole 64-sample pieces");
  const float* src = (const float*)p.pcm + clip_off + s0;
  static_for<0, (HOPS + 2) / 3>([&](auto Tf) {
    constexpr int h0 = 3 * decltype(Tf)::value;  // three hops per pair of per-lane addresses: offsets stay below 4096 bytes
    const float* s16 = src + h0 * HOP + 4 * lane;
    const float* s4 = src + h0 * HOP + lane;
    static_for<0, 3>([&](auto Hf) {
      constexpr int h = h0 + decltype(Hf)::value;
      if constexpr (h < HOPS) {
        constexpr int n = (h + 1) * HOP <= SPAN ? HOP : SPAN - h * HOP;  // samples of this hop that the span holds
        constexpr int N16 = n / 256, N4 = (n % 256) / 64;
        constexpr int goff = (h - h0) * HOP;  // source offset from the per-lane address, in floats
        float* d = dst + h * (HOP + SKEW) - goff;  // so that d + goff is where the hop goes
        static_for<0, N16>([&](auto Jf) {
          constexpr int off = 4 * (goff + 256 * decltype(Jf)::value);
          static_assert(off < 4096, "13-bit instruction offset");
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s16,
                                           (__attribute__((address_space(3))) void*)d, 16, off, 0);
        });
        static_for<0, N4>([&](auto Jf) {
          constexpr int off = 4 * (goff + 256 * N16 + 64 * decltype(Jf)::value);
          static_assert(off < 4096, "13-bit instruction offset");
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s4,
                                           (__attribute__((address_space(3))) void*)d, 4, off, 0);
        });
      }
    });
  });
}

// Shape of a plan's mel schedule.  The shapes of the reference's own flavours are compiled in, which turns the
// loops over accumulator sets, steps and positions into straight-line code; any other filterbank runs the same code with
// the shape read from the launch parameters.
// Accumulator sets [0, NHI) run U steps of 4 bins, sets [NHI, SETS) run ULO <= U steps (see MelMfmaSchedule).
template <int SETS_, int U_, int MAXPOS_, int ROUNDS_, int HOP_ = 0, int ULO_ = U_, int NHI_ = SETS_>
struct MelStatic {
  static constexpr bool kStatic = true;
  static constexpr int SETS = SETS_, U = U_, MAXPOS = MAXPOS_, ROUNDS = ROUNDS_, ULO = ULO_, NHI = NHI_;
  static constexpr int HOP = HOP_;  // the flavour's hop when it is compiled in too (0: read from the launch parameters)
  static constexpr int STEPS = NHI * U + (SETS - NHI) * ULO;  // steps of all sets together
  static constexpr int steps(int s) { return s < NHI ? U : ULO; }
  static constexpr int a_offset(int s) { return 256 * ((s < NHI ? s : NHI) * U + (s > NHI ? s - NHI : 0) * ULO); }  // floats
};
struct MelDynamic {
  static constexpr bool kStatic = false;
  static constexpr int SETS = 0, U = 0, MAXPOS = 0, ROUNDS = 0, HOP = 0, ULO = 0, NHI = 0, STEPS = 1 << 20;
  static constexpr int steps(int) { return 0; }
  static constexpr int a_offset(int) { return 0; }
};

// Skew of the staged span, in floats per hop.  With 16 lanes per frame the LDS serves the sample reads of two frames
// together, each frame's lanes covering 32 consecutive banks: when the hop is a multiple of 64 floats both frames sit
// on the same 32 banks and every read takes two passes (measured 1.71 against 0.94 ns per instruction and CU).  The
// span is then staged hop by hop, each hop 32 floats further than the samples before it, which puts the second frame
// on the other 32 banks.  Needs the hop at compile time (the reads keep immediate offsets) and whole 64-float DMA
// pieces per hop.
template <class G, class Mel, bool PRE, bool SPEC_OUT>
constexpr int span_skew() {
  return (Mel::HOP > 0 && !PRE && !SPEC_OUT && G::A == 1 && G::LPF == 16 && Mel::HOP % 64 == 0) ? 32 : 0;
}
// floats of LDS a skewed span of `span` samples takes
constexpr int skewed_floats(int span, int hop, int skew) { return (span + skew * ((span - 1) / hop) + 3) & ~3; }
template <class Mel> __device__ __forceinline__ int mel_sets_of(const FrontParams& p) { if constexpr (Mel::kStatic) return Mel::SETS; else return p.mel_sets; }
// steps of accumulator set s and where its weights start in the A-operand table (floats)
template <class Mel> __device__ __forceinline__ int mel_steps_of(const FrontParams& p, int s) {
  if constexpr (Mel::kStatic) return Mel::steps(s); else return s < p.mel_nhi ? p.mel_u : p.mel_ulo;
}
template <class Mel> __device__ __forceinline__ int mel_aoff_of(const FrontParams& p, int s) {
  if constexpr (Mel::kStatic) return Mel::a_offset(s);
  else return 256 * (min(s, p.mel_nhi) * p.mel_u + max(0, s - p.mel_nhi) * p.mel_ulo);
}
template <class Mel> __device__ __forceinline__ int mel_maxpos_of(const FrontParams& p) { if constexpr (Mel::kStatic) return Mel::MAXPOS; else return p.mel_maxpos; }
template <class Mel> __device__ __forceinline__ bool mel_two_rounds(const FrontParams& p) { if constexpr (Mel::kStatic) return Mel::ROUNDS > 1; else return p.n_quads > 16; }

// Small compiled-in shapes: the operands of all accumulator sets (and of NG groups of 4 frames, `gpitch` floats apart)
// are fetched in ONE burst of 2 * NG * STEPS 16-byte reads (the transform registers are free by now), so the stage
// waits for the LDS once instead of once per set and group, and the instructions of the chains alternate, which also
// leaves no dependent pair back to back.  acc[g][s]: partial sums of set s for group g, as mel_mfma_4frames stores them.
template <class Mel, int NG>
__device__ __forceinline__ void mel_mfma_one_burst(const int (&boff)[kMaxMelSets], const float* prow0, int gpitch,
                                                   const float* mela, v4 (&acc)[NG][Mel::SETS], int lane) {
  v4 a[Mel::SETS][Mel::U], bb[NG][Mel::SETS][Mel::U];
  static_for<0, Mel::SETS>([&](auto Sf) {
    constexpr int s = decltype(Sf)::value;
    const float* ap = mela + Mel::a_offset(s) + lane * 4;
    static_for<0, Mel::steps(s)>([&](auto Uf) {
      constexpr int u = decltype(Uf)::value;
      a[s][u] = ld4(ap + 256 * u);
      static_for<0, NG>([&](auto Gf) {
        constexpr int g = decltype(Gf)::value;
        bb[g][s][u] = ld4(prow0 + g * gpitch + (boff[s] & 0xffff) + 4 * u);
      });
    });
    static_for<0, NG>([&](auto Gf) { acc[decltype(Gf)::value][s] = v4{0.0f, 0.0f, 0.0f, 0.0f}; });
  });
  SED_SCHED_FENCE();
  static_for<0, Mel::U>([&](auto Uf) {
    constexpr int u = decltype(Uf)::value;
    static_for<0, 4>([&](auto Ef) {
      constexpr int e = decltype(Ef)::value;
      static_for<0, NG>([&](auto Gf) {
        constexpr int g = decltype(Gf)::value;
        static_for<0, Mel::SETS>([&](auto Sf) {
          constexpr int s = decltype(Sf)::value;
          if constexpr (u < Mel::steps(s)) acc[g][s] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[s][u][e], bb[g][s][u][e], acc[g][s], 0, 0, 0);
        });
      });
    });
  });
}

// Mel projection of 4 frames (power rows prow0 + j*PITCH, j = 0..3) on the matrix pipe.
// Lane 4*block + j feeds the power of frame j as the B operand and the weight of filter 4*quad + j as the
// A operand of block `block`; after the last instruction it holds, in acc[i], the partial sum of filter
// 4*quad + i for frame j over the bins of position (set, block).  The partial sums go to LDS, kPartStride floats per slot; which
// slot a block fills comes from the lane's table word (blocks are placed for bank-conflict-free operand reads, slots
// follow the quads).
template <class Mel, int BURST_STEPS = 12>
__device__ __forceinline__ void mel_mfma_4frames(const FrontParams& p, const int32_t* boffL, const float* prow0,
                                                 const float* mela, float* part, int lane) {
  asm volatile("" : "+v"(lane));  // keeps the per-lane table reads inside the streaming loop instead of pinning registers
  const int sets = SED_SKIP(32) ? 0 : mel_sets_of<Mel>(p);
  int boff[kMaxMelSets];
  static_for<0, kMaxMelSets>([&](auto Sf) {
    constexpr int s = decltype(Sf)::value;
    boff[s] = (Mel::kStatic ? s < Mel::SETS : s < p.mel_sets) ? boffL[s * 64 + lane] : 0;
  });
#ifndef SED_NO_MEL_ONE_BURST
  if constexpr (Mel::kStatic && Mel::STEPS <= BURST_STEPS) {
    if (!SED_SKIP(32)) {
      v4 acc[1][Mel::SETS];
      mel_mfma_one_burst<Mel, 1>(boff, prow0, 0, mela, acc, lane);
      static_for<0, Mel::SETS>([&](auto Sf) { constexpr int s = decltype(Sf)::value; st4(part + (boff[s] >> 16), acc[0][s]); });
    }
    wave_phase_fence();
    return;
  }
#endif
  static_for<0, kMaxMelSets>([&](auto Sf) {
    constexpr int s = decltype(Sf)::value;
    if (s < sets) {
      const float* bp = prow0 + (boff[s] & 0xffff);
      const int U = mel_steps_of<Mel>(p, s);
      const float* ap = mela + mel_aoff_of<Mel>(p, s) + lane * 4;
      v4 acc = v4{0.0f, 0.0f, 0.0f, 0.0f};
      for (int u0 = 0; u0 < U; u0 += kMelBurst) {
        // one burst of LDS reads for up to kMelBurst steps of 4 bins, then the instructions, two steps per
        // (wave-uniform) test
        v4 a[kMelBurst], bb[kMelBurst];
        static_for<0, kMelBurst / 2>([&](auto Pf) {
          constexpr int k = 2 * decltype(Pf)::value;
          if (u0 + k < U) {  // wave-uniform: pairs of steps past the end are not fetched at all
            a[k] = ld4(ap + 256 * (u0 + k));
            bb[k] = ld4(bp + 4 * (u0 + k));
            a[k + 1] = ld4(ap + 256 * min(u0 + k + 1, U - 1));
            if (u0 + k + 1 >= U) a[k + 1] = v4{0.0f, 0.0f, 0.0f, 0.0f};  // wave-uniform: an odd step count ends on zero weights
            bb[k + 1] = ld4(bp + 4 * min(u0 + k + 1, U - 1));
          }
        });
        SED_SCHED_FENCE();  // the whole burst is in flight before the first instruction waits for its operands
        static_for<0, kMelBurst / 2>([&](auto Pf) {
          constexpr int k = 2 * decltype(Pf)::value;
          if (u0 + k < U) {
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k].x, bb[k].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k].y, bb[k].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k].z, bb[k].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k].w, bb[k].w, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k + 1].x, bb[k + 1].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k + 1].y, bb[k + 1].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k + 1].z, bb[k + 1].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k + 1].w, bb[k + 1].w, acc, 0, 0, 0);
          }
        });
      }
      st4(part + (boff[s] >> 16), acc);
    }
  });
  wave_phase_fence();
}

// Round r of the gather: lane 4*q + j sums the partial sums of quad 16 r + q for frame j, takes the log and applies
// the optional affine.  It then holds 4 consecutive filters of one frame.  The gather word of a lane holds the float
// offset of its quad's first partial sum (bits 0-15) and the quad's number of positions (bits 16-20); lanes whose quad
// has fewer positions than the widest one read a block of zeros instead.
template <class Mel>
__device__ __forceinline__ v4 mel_gather_log(const FrontParams& p, const int32_t* gwL, const float* affL,
                                             const float* zero4, const float* part, int r, int lane) {
  asm volatile("" : "+v"(lane));
  const int g = gwL[r * 64 + lane];
  const int npos = g >> 16;
  const float* first = part + (g & 0xffff);
  v4 sum = ld4(first);  // every quad has at least one position
  const int maxpos = mel_maxpos_of<Mel>(p);
#pragma unroll
  for (int i0 = 1; i0 < maxpos; i0 += 2) {  // wave-uniform trip count
    const v4 x = ld4(i0 < npos ? first + i0 * kPartStride : zero4);
    const v4 y = ld4(i0 + 1 < npos ? first + (i0 + 1) * kPartStride : zero4);
    sum += x + y;
  }
  v4 val = sum;
  if (p.log_kind == SED_LOG_DB10) {
    val.x = 3.0102999566398120f * __log2f(fmaxf(sum.x, p.amin)) - p.db_offset;
    val.y = 3.0102999566398120f * __log2f(fmaxf(sum.y, p.amin)) - p.db_offset;
    val.z = 3.0102999566398120f * __log2f(fmaxf(sum.z, p.amin)) - p.db_offset;
    val.w = 3.0102999566398120f * __log2f(fmaxf(sum.w, p.amin)) - p.db_offset;
  } else if (p.log_kind == SED_LOG_LN) {
    val.x = 0.6931471805599453f * __log2f(sum.x + p.ln_offset);
    val.y = 0.6931471805599453f * __log2f(sum.y + p.ln_offset);
    val.z = 0.6931471805599453f * __log2f(sum.z + p.ln_offset);
    val.w = 0.6931471805599453f * __log2f(sum.w + p.ln_offset);
  } else if (p.log_kind == SED_LOG_LN_FLOOR) {
    val.x = 0.6931471805599453f * __log2f(fmaxf(sum.x, p.amin));
    val.y = 0.6931471805599453f * __log2f(fmaxf(sum.y, p.amin));
    val.z = 0.6931471805599453f * __log2f(fmaxf(sum.z, p.amin));
    val.w = 0.6931471805599453f * __log2f(fmaxf(sum.w, p.amin));
  }
  const int m0 = 4 * (16 * r + (lane >> 2));
  if (p.affine) val = __builtin_elementwise_fma(val, ld4(affL + m0), ld4(affL + p.aff_stride + m0));
  return val;
}

// The gather in two halves, so that the reads of several gathers can be in flight before the first sum is formed.
// Loads of one gather: the quad's first partial sum, then pairs (i0, i0 + 1) for i0 = 1, 3, .. below maxpos.
template <class Mel> constexpr int gather_loads() { return Mel::kStatic ? 1 + 2 * (Mel::MAXPOS / 2) : 1 + 2 * 8; }
template <class Mel>
__device__ __forceinline__ void mel_gather_issue(const FrontParams& p, const int32_t* gwL, const float* zero4,
                                                 const float* part, int r, int lane, v4 (&x)[gather_loads<Mel>()]) {
  asm volatile("" : "+v"(lane));
  const int g = gwL[r * 64 + lane];
  const int npos = g >> 16;
  const float* first = part + (g & 0xffff);
  x[0] = ld4(first);  // every quad has at least one position
  const int maxpos = mel_maxpos_of<Mel>(p);
  static_for<1, gather_loads<Mel>()>([&](auto If) {
    constexpr int i = decltype(If)::value;
    constexpr int i0 = (i & 1) ? i : i - 1;  // the pair (i0, i0 + 1) is fetched when i0 < maxpos (wave-uniform)
    if (i0 < maxpos) x[i] = ld4(i < npos ? first + i * kPartStride : zero4);
    else x[i] = v4{0.0f, 0.0f, 0.0f, 0.0f};
  });
}
template <class Mel>
__device__ __forceinline__ v4 mel_gather_finish(const FrontParams& p, const float* affL, int r, int lane,
                                                const v4 (&x)[gather_loads<Mel>()]) {
  asm volatile("" : "+v"(lane));
  v4 sum = x[0];
  const int maxpos = mel_maxpos_of<Mel>(p);
  static_for<0, (gather_loads<Mel>() - 1) / 2>([&](auto Tf) {
    constexpr int i0 = 1 + 2 * decltype(Tf)::value;
    if (i0 < maxpos) sum += x[i0] + x[i0 + 1];
  });
  v4 val = sum;
  if (p.log_kind == SED_LOG_DB10) {
    val.x = 3.0102999566398120f * __log2f(fmaxf(sum.x, p.amin)) - p.db_offset;
    val.y = 3.0102999566398120f * __log2f(fmaxf(sum.y, p.amin)) - p.db_offset;
    val.z = 3.0102999566398120f * __log2f(fmaxf(sum.z, p.amin)) - p.db_offset;
    val.w = 3.0102999566398120f * __log2f(fmaxf(sum.w, p.amin)) - p.db_offset;
  } else if (p.log_kind == SED_LOG_LN) {
    val.x = 0.6931471805599453f * __log2f(sum.x + p.ln_offset);
    val.y = 0.6931471805599453f * __log2f(sum.y + p.ln_offset);
    val.z = 0.6931471805599453f * __log2f(sum.z + p.ln_offset);
    val.w = 0.6931471805599453f * __log2f(sum.w + p.ln_offset);
  } else if (p.log_kind == SED_LOG_LN_FLOOR) {
    val.x = 0.6931471805599453f * __log2f(fmaxf(sum.x, p.amin));
    val.y = 0.6931471805599453f * __log2f(fmaxf(sum.y, p.amin));
    val.z = 0.6931471805599453f * __log2f(fmaxf(sum.z, p.amin));
    val.w = 0.6931471805599453f * __log2f(fmaxf(sum.w, p.amin));
  }
  const int m0 = 4 * (16 * r + (lane >> 2));
  if (p.affine) val = __builtin_elementwise_fma(val, ld4(affL + m0), ld4(affL + p.aff_stride + m0));
  return val;
}

// Store of one gather round: 16 bytes per lane, 1 KB contiguous per wavefront (frames are n_mels floats apart).
// ZERO_TAIL (segment kernels): rows [nvalid, nrows) are written as zeros (the rows behind a segment's own frames).
template <bool ZERO_TAIL = false>
__device__ __forceinline__ void mel_store(const FrontParams& p, float* dst, v4 val, int r, int nvalid, int lane, int nrows = 0) {
  asm volatile("" : "+v"(lane));  // the per-lane output offset is recomputed here instead of living across the loop
  const int j = lane & 3, m0 = 4 * (16 * r + (lane >> 2));
  if constexpr (ZERO_TAIL) {
    if (j >= nvalid) val = v4{0.0f, 0.0f, 0.0f, 0.0f};
    nvalid = nrows;
  }
  if (j < nvalid && m0 < p.n_mels) {
    float* o = dst + j * p.n_mels + m0;
    if (p.out_vec) {
      st4(o, val);
    } else {
      o[0] = val.x;
      if (m0 + 1 < p.n_mels) o[1] = val.y;
      if (m0 + 2 < p.n_mels) o[2] = val.z;
      if (m0 + 3 < p.n_mels) o[3] = val.w;
    }
  }
}

// Wavefronts per workgroup (= per CU).  The spectrogram-output kernels (the unfused parity path) keep a few more values
// live and get 8 wavefronts with 256 registers each instead of 12 with 168: no spills, and their rate is set by the
// (T, F) writes anyway.
template <class G, bool SPEC_OUT>
constexpr int wg_waves_of() { return SPEC_OUT && G::WAVES > 8 ? 8 : G::WAVES; }

// Work distribution.  A group is FPW consecutive frames of one clip.  Every workgroup owns a contiguous range of
// groups (so a CU streams through contiguous PCM); its wavefronts take them one at a time from a counter in LDS.
// The hardware issues instructions oldest-wavefront-first, so wavefronts with equal static shares finish far
// apart (measured: the first after 58 % of the kernel time) and the last ones run on a nearly empty CU; with the
// queue they all stay busy until the range is drained.  The span of the next group is ordered as soon as the current
// one's power rows are consumed, whichever clip it belongs to.
// SEG: the clips are segments of a resident PCM buffer (the detector -> classifier hand-off without a copy): clip b holds
// seg_len[b] samples from element seg_off[b] on and zeros behind them, exactly the row a cut of the segment into a
// zero-padded batch would hold.
template <class G, bool MAG, bool SPEC_OUT, bool PRE, bool I16, class Mel, bool SEG = false>
__global__ void __launch_bounds__((wg_waves_of<G, SPEC_OUT>()) * 64)
__attribute__((amdgpu_waves_per_eu(SED_WPE((wg_waves_of<G, SPEC_OUT>()) / 4), SED_WPE((wg_waves_of<G, SPEC_OUT>()) / 4))))
stft_mel_kernel(const FrontParams p) {
  extern __shared__ __align__(16) float smem[];
#ifdef SED_TIMELINE
  const unsigned long long tl_start = __builtin_amdgcn_s_memrealtime();
#endif
  const int NT = blockDim.x;  // up to G::WAVES wavefronts; the host launches fewer when LDS is short
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: group bookkeeping stays scalar
  const int lane = tid & 63;
  const int fi = lane / G::LPF, l = lane % G::LPF;
  const int wg_waves = NT >> 6;
  constexpr int SKEW = span_skew<G, Mel, PRE, SPEC_OUT>();
  // samples from the start of the span that an LDS-DMA transfer of it reads
  const int dma_reach = SKEW > 0 ? (G::FPW - 1) * Mel::HOP + G::NFFT : 256 * p.span_blocks;

  float* winL = smem;
  float* tw1L = smem + p.im.off_tw1;
  float* tw2L = smem + p.im.off_tw2;
  float* melaL = smem + p.im.off_mela;
  int32_t* boffL = (int32_t*)(smem + p.im.off_boff);
  int32_t* gwL = (int32_t*)(smem + p.im.off_gw);
  float* affL = smem + p.im.off_aff;
  float* zeroL = smem + p.im.off_zero;
  int* qhead = (int*)(smem + p.off_queue);
  float* X = smem + p.off_wave + wave * p.wave_floats;

  // tables: the plan's image is already in LDS order
  for (int i = 4 * tid; i < p.image_floats; i += 4 * NT) st4(smem + i, ld4(p.image + i));
  if (tid == 0) *qhead = wg_waves;  // wavefront w starts with group w of the range
  __syncthreads();

  // Groups of a workgroup.  Clips of equal length: a contiguous range (a CU streams through contiguous PCM).  Segments:
  // their work differs (groups behind a short segment are skipped), so chunks of kSegChunk groups are dealt out to the
  // workgroups round robin: local group v is group (blockIdx + (v / kSegChunk) * gridDim) * kSegChunk + v % kSegChunk,
  // ascending in v, so the first one past the end ends the workgroup's work.
  constexpr int kSegChunk = 4;
  const int g_begin = SEG ? 0 : (int)((int64_t)p.total_groups * blockIdx.x / gridDim.x);
  const int g_end = SEG ? p.total_groups : (int)((int64_t)p.total_groups * (blockIdx.x + 1) / gridDim.x);
  auto group_of = [&](int v) {
    if constexpr (SEG) return (int)((blockIdx.x + (unsigned)(v / kSegChunk) * gridDim.x) * kSegChunk + v % kSegChunk);
    else return g_begin + v;
  };
#ifdef SED_TIMELINE
  const unsigned long long tl_loop = __builtin_amdgcn_s_memrealtime();
  int tl_iters = 0;
#endif
  constexpr int kGroups = G::FPW / 4;
  int gi = group_of(wave);
  if (gi < g_end) {
    int b = gi / p.groups_per_clip;
    int f0 = (gi - b * p.groups_per_clip) * G::FPW;
    // staged : the span of the group about to be processed is already on its way into X
    // held   : log-mel values of the previous group, stored one iteration late: vmcnt counts loads and stores in
    //          order, and a store issued just before the wait for the span would put its own latency on the path
    bool staged = false, held = false;
    v4 hval[kGroups][2];
    float* hdst = nullptr;
    int hvalid = 0, hrows = 0;  // rows of the held group with values / rows written (segments: zeros behind the values)
    // Segments: the log-mel values of a frame made of zeros only (the padding behind a short segment), through the very
    // expressions of the gather (sum of zero partial sums, log, affine), so that groups of such frames can skip the transform.
    v4 zc[2] = {v4{0.0f, 0.0f, 0.0f, 0.0f}, v4{0.0f, 0.0f, 0.0f, 0.0f}};
    if constexpr (SEG && !SPEC_OUT) {
      v4 zeros[gather_loads<Mel>()];
      static_for<0, gather_loads<Mel>()>([&](auto If) { zeros[decltype(If)::value] = v4{0.0f, 0.0f, 0.0f, 0.0f}; });
      zc[0] = mel_gather_finish<Mel>(p, affL, 0, lane, zeros);
      if (mel_two_rounds<Mel>(p)) zc[1] = mel_gather_finish<Mel>(p, affL, 1, lane, zeros);
    }
    SED_STAMP_DECL
    while (true) {
#ifdef SED_TIMELINE
      ++tl_iters;
#endif
      // take the next group now; the answer is needed when its span is ordered, half an iteration from here
      SED_P(0);
      int taken = 0;
      if (lane == 0) taken = atomicAdd(qhead, 1);
#ifdef SED_EST_FAST_EDGES  // timing experiments: clip-edge groups take the interior path (their results are then garbage)
      const int64_t s0 = min(max((int64_t)f0 * p.hop - p.pad_left, (int64_t)0), p.L - dma_reach);
#else
      const int64_t s0 = (int64_t)f0 * p.hop - p.pad_left;
#endif
      int Tb = p.T;  // frames of clip b that are written
      if constexpr (SEG && !SPEC_OUT) {
        // A group whose whole span lies in the zeros behind the segment (and, at the clip's end, whose reflected samples do
        // too): nothing to fetch and nothing to transform.  (A span ordered early always lies inside the stored samples.)
        // Segments framed as recordings of their own: the groups behind the segment's own frames.
        const int64_t valid = min((int64_t)p.seg_len[b], p.L);
        const int64_t last = s0 + p.span - 1;  // last sample of the span in the virtual clip
        bool all_zero;
        if (p.seg_own) {
          Tb = valid < p.own_min ? 0 : 1 + (int)((unsigned)(valid - p.own_sub) / (unsigned)p.hop);
          all_zero = !staged && f0 >= Tb;
        } else {
          all_zero = !staged && (valid <= 0 || (s0 >= valid && (last < p.L || p.pad_mode != SED_PAD_REFLECT ||
                                                                2 * (p.L - 1) - last >= valid)));
        }
        if (all_zero) {
          if (held) {
            static_for<0, kGroups>([&](auto Gf) {
              constexpr int g = decltype(Gf)::value;
              mel_store<SEG>(p, hdst + 4 * g * p.n_mels, hval[g][0], 0, hvalid - 4 * g, lane, hrows - 4 * g);
              if (mel_two_rounds<Mel>(p)) mel_store<SEG>(p, hdst + 4 * g * p.n_mels, hval[g][1], 1, hvalid - 4 * g, lane, hrows - 4 * g);
            });
          }
          static_for<0, kGroups>([&](auto Gf) { hval[decltype(Gf)::value][0] = zc[0]; hval[decltype(Gf)::value][1] = zc[1]; });
          held = true;
          hdst = p.out + (int64_t)b * p.out_stride_b + (int64_t)f0 * p.n_mels;
          hrows = min(G::FPW, p.T - f0);
          hvalid = p.seg_own ? 0 : hrows;  // (the rows behind a segment's own frames are zero)
          gi = group_of(__builtin_amdgcn_readfirstlane(taken));
          if (gi >= g_end) break;
          b = gi / p.groups_per_clip;
          f0 = (gi - b * p.groups_per_clip) * G::FPW;
          continue;
        }
      }
      LaneState<G> st;
      if (!staged) {
        // where the clip starts in the PCM buffer and how many stored samples it has (segments: any sample alignment,
        // so the LDS-DMA path is taken only by segments that start on a 16-byte boundary)
        int64_t coff, valid;
        if constexpr (SEG) { coff = p.seg_off[b]; valid = min((int64_t)p.seg_len[b], p.L); }  // never past the row, whatever the table says
        else { coff = (int64_t)b * p.pcm_stride; valid = p.L; }
        if (!I16 && p.dma && s0 >= 0 && s0 + dma_reach <= valid && (!SEG || (coff & 3) == 0)) {
          if constexpr (SKEW > 0) stage_span_dma_skewed<G, Mel::HOP, SKEW>(p, X, coff, s0, lane);
          else if (!SED_SKIP(1)) stage_span_dma(p, X, coff, s0, lane);
        } else {
          int64_t clipL = p.L;
          if constexpr (SEG) { if (p.seg_own) clipL = valid; }
          stage_span<I16, Mel::HOP, SKEW>(p, X, coff, valid, clipL, s0, p.span, lane);
        }
      }
      SED_STAMP(0)
      // The wait for the span is the compiler's own instruction (vmcnt 0, the other counters left alone), not inline
      // assembly: the compiler keeps a transfer into LDS on its books until it sees it waited for, and while one is
      // outstanding it turns every wait for an LDS read into a wait for all of them, which undoes the ordering of the table
      // reads below (issued well ahead of their use) for the whole loop.
      if (!SED_SKIP(1024)) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("" ::: "memory");
      }
      wave_phase_fence();
      SED_STAMP(1)
      if constexpr (!SPEC_OUT) {
        if (held) {
          static_for<0, kGroups>([&](auto Gf) {
            constexpr int g = decltype(Gf)::value;
            mel_store<SEG>(p, hdst + 4 * g * p.n_mels, hval[g][0], 0, hvalid - 4 * g, lane, hrows - 4 * g);
            if (mel_two_rounds<Mel>(p)) mel_store<SEG>(p, hdst + 4 * g * p.n_mels, hval[g][1], 1, hvalid - 4 * g, lane, hrows - 4 * g);
          });
        }
      }
      SED_P(1);
      if constexpr (PRE) {
        // Kaldi framing: the lanes of a frame pool their partial sums through LDS (fixed summation order), then
        // every lane conditions its own samples with the frame mean.
        float* pool = X + p.pre_off + fi * G::LPF;
        // wave-uniform: frames start on whole 8- / 16-byte units (n_fft 2048 keeps the one path its registers hold)
        const bool vec = G::NFFT < 2048 && p.hop % (2 * G::A) == 0;
        int wl = p.win_length;
        asm volatile("" : "+s"(wl));  // the row tests stay scalar compares inside the loop instead of hoisted, spilled masks
        pool[l] = vec ? lane_frame_sum<G, true>(X + fi * p.hop, l, wl) : lane_frame_sum<G, false>(X + fi * p.hop, l, wl);
        wave_phase_fence();
        float mean = 0.0f;
        if (p.remove_dc) {
          for (int j = 0; j < G::LPF; ++j) mean += pool[j];
          mean *= p.inv_win_length;
        }
        if (vec) lane_stage1_conditioned<G, true>(st, X + fi * p.hop, winL, tw1L, l, mean, p.preemph);
        else lane_stage1_conditioned<G, false>(st, X + fi * p.hop, winL, tw1L, l, mean, p.preemph);
      } else {
        if constexpr (SKEW > 0) {
          lane_load_samples_skewed<G, Mel::HOP, SKEW>(st, X + fi * (Mel::HOP + SKEW), l);
          lane_stage1<G, SrcF32, true>(st, SrcF32{nullptr}, winL, tw1L, l);
        } else {
          // frames of a hop that is not a multiple of 4 samples start off the 8- / 16-byte units of the aligned reads:
          // built for the run-time mel shapes only (every compiled-in flavour has such a hop)
          bool aligned = true;
          if constexpr (!Mel::kStatic) aligned = (p.hop & 3) == 0;  // wave-uniform
          if (aligned) lane_stage1<G>(st, SrcF32{X + fi * p.hop}, winL, tw1L, l);
          else lane_stage1<G, SrcF32U>(st, SrcF32U{X + fi * p.hop}, winL, tw1L, l);
        }
      }
      wave_phase_fence();
      SED_STAMP(2)
      float* xfr = X + fi * G::XFR;
      if (!SED_SKIP(8)) {
        SED_P(3);
        lane_xwrite<G, 0>(st, xfr, l);
        wave_phase_fence();
        lane_xread<G, 0>(st, xfr, l);
        wave_phase_fence();
        lane_xwrite<G, 1>(st, xfr, l);
        wave_phase_fence();
        lane_xread<G, 1>(st, xfr, l);
        lane_xwait<G>(st);
        wave_phase_fence();
      } else {
        static_for<0, G::R2>([&](auto Nf) { st.p[decltype(Nf)::value] = st.y[0][decltype(Nf)::value]; st.q[decltype(Nf)::value] = st.y[0][G::R1 - 1 - decltype(Nf)::value]; });
      }
      SED_STAMP(3)
      float* pw = X + fi * G::PP;
      SED_P(4);
      lane_stage2_power<G, MAG>(st, pw, tw2L, l);
      for (int k = G::NBINS + l; k < G::PP; k += G::LPF) pw[k] = 0.0f;  // row padding read (with zero weight) by the mel stage
      wave_phase_fence();
      SED_STAMP(4)
      const int nvalid = min(G::FPW, Tb - f0);
      // the group after this one: the one just taken from the queue
      gi = group_of(__builtin_amdgcn_readfirstlane(taken));
      const bool more = gi < g_end;
      const int nb = gi / p.groups_per_clip;
      const int nf0 = (gi - nb * p.groups_per_clip) * G::FPW;
      staged = false;
      if constexpr (SPEC_OUT) {
        // the group's rows are one contiguous run of nvalid * NBINS floats in the (T, F) output: a few scalar stores up to
        // the first 16-byte boundary, then 16 bytes per lane (the four values of a vector may sit in two power rows)
        float* dst = p.out + (int64_t)b * p.out_stride_b + (int64_t)f0 * G::NBINS;
        const int n = nvalid * G::NBINS;
        const int head = min(n, (int)((4 - (((uintptr_t)dst >> 2) & 3)) & 3));
        const int nvec = (n - head) >> 2;
        auto at = [&](int e) { const int f = e / G::NBINS; return X[f * G::PP + (e - f * G::NBINS)]; };
        if (lane < head) dst[lane] = at(lane);
        for (int j = lane; j < nvec; j += 64) {
          const int e = head + 4 * j;
          st4(dst + e, v4{at(e), at(e + 1), at(e + 2), at(e + 3)});
        }
        const int tail0 = head + 4 * nvec;
        if (tail0 + lane < n) dst[tail0 + lane] = at(tail0 + lane);
      } else {
        float* part = X + p.part_off;
#ifndef SED_NO_MEL_ONE_BURST
        if constexpr (kGroups == 2 && Mel::kStatic && 2 * Mel::STEPS <= 16) {
          // Both groups of 4 frames in one pass: one burst of operand reads, then the partial sums of the first group,
          // its gather reads, the partial sums of the second group (the LDS executes a wavefront's accesses in order, so
          // they may reuse the area) and its gather reads, and one wait for all of them.
          SED_P(5);
          int lv = lane;
          asm volatile("" : "+v"(lv));
          int boff[kMaxMelSets];
          static_for<0, kMaxMelSets>([&](auto Sf) {
            constexpr int s = decltype(Sf)::value;
            boff[s] = s < Mel::SETS ? boffL[s * 64 + lv] : 0;
          });
          v4 acc[2][Mel::SETS];
          if (!SED_SKIP(32)) mel_mfma_one_burst<Mel, 2>(boff, X, 4 * G::PP, melaL, acc, lane);
          wave_phase_fence();
          SED_STAMP(5)
          SED_P(6);
          const int64_t s1 = (int64_t)nf0 * p.hop - p.pad_left;
          if (!I16 && p.early && more && s1 >= 0 && !SED_SKIP(1)) {
            int64_t ncoff, nvalid;
            if constexpr (SEG) { ncoff = p.seg_off[nb]; nvalid = min((int64_t)p.seg_len[nb], p.L); }
            else { ncoff = (int64_t)nb * p.pcm_stride; nvalid = p.L; }
            if (s1 + dma_reach <= nvalid && (!SEG || (ncoff & 3) == 0)) {
              if constexpr (SKEW > 0) stage_span_dma_skewed<G, Mel::HOP, SKEW>(p, X, ncoff, s1, lane);
              else stage_span_dma(p, X, ncoff, s1, lane);
              staged = true;
            }
          }
          if (!SED_SKIP(64)) {
            v4 x[2][Mel::ROUNDS][gather_loads<Mel>()];
            static_for<0, 2>([&](auto Gf) {
              constexpr int g = decltype(Gf)::value;
              static_for<0, Mel::SETS>([&](auto Sf) { constexpr int s = decltype(Sf)::value; st4(part + (boff[s] >> 16), acc[g][s]); });
              wave_phase_fence();
              static_for<0, Mel::ROUNDS>([&](auto Rf) { constexpr int r = decltype(Rf)::value; mel_gather_issue<Mel>(p, gwL, zeroL, part, r, lane, x[g][r]); });
              wave_phase_fence();
            });
            static_for<0, 2>([&](auto Gf) {
              constexpr int g = decltype(Gf)::value;
              static_for<0, Mel::ROUNDS>([&](auto Rf) { constexpr int r = decltype(Rf)::value; hval[g][r] = mel_gather_finish<Mel>(p, affL, r, lane, x[g][r]); });
            });
          }
          wave_phase_fence();
        } else
#endif
        static_for<0, kGroups>([&](auto Gf) {
          constexpr int g = decltype(Gf)::value;
          SED_P(5);
          // (the n_fft 2048 kernels may use 256 registers: their larger shapes fit one burst too)
          mel_mfma_4frames<Mel, (G::NFFT == 2048 ? SED_BURST2048 : 12)>(p, boffL, X + 4 * g * G::PP, melaL, part, lane);
          SED_STAMP(5)
          SED_P(6);
          if constexpr (g == kGroups - 1) {
            // the power rows are consumed: the next group's span may land on them while the gather runs
            const int64_t s1 = (int64_t)nf0 * p.hop - p.pad_left;
            if (!I16 && p.early && more && s1 >= 0 && !SED_SKIP(1)) {
              int64_t ncoff, nvalid;
              if constexpr (SEG) { ncoff = p.seg_off[nb]; nvalid = min((int64_t)p.seg_len[nb], p.L); }
              else { ncoff = (int64_t)nb * p.pcm_stride; nvalid = p.L; }
              if (s1 + dma_reach <= nvalid && (!SEG || (ncoff & 3) == 0)) {
                if constexpr (SKEW > 0) stage_span_dma_skewed<G, Mel::HOP, SKEW>(p, X, ncoff, s1, lane);
                else stage_span_dma(p, X, ncoff, s1, lane);
                staged = true;
              }
            }
          }
          if (!SED_SKIP(64)) {
            hval[g][0] = mel_gather_log<Mel>(p, gwL, affL, zeroL, part, 0, lane);
            if (mel_two_rounds<Mel>(p)) hval[g][1] = mel_gather_log<Mel>(p, gwL, affL, zeroL, part, 1, lane);
          }
          wave_phase_fence();
        });
        held = !SED_SKIP(64);
        hdst = p.out + (int64_t)b * p.out_stride_b + (int64_t)f0 * p.n_mels;
        hvalid = nvalid;
        hrows = min(G::FPW, p.T - f0);
      }
      wave_phase_fence();
      SED_STAMP(6)
      if (!more) break;
      b = nb;
      f0 = nf0;
    }
    if constexpr (!SPEC_OUT) {
      if (held) {
        static_for<0, kGroups>([&](auto Gf) {
          constexpr int g = decltype(Gf)::value;
          mel_store<SEG>(p, hdst + 4 * g * p.n_mels, hval[g][0], 0, hvalid - 4 * g, lane, hrows - 4 * g);
          if (mel_two_rounds<Mel>(p)) mel_store<SEG>(p, hdst + 4 * g * p.n_mels, hval[g][1], 1, hvalid - 4 * g, lane, hrows - 4 * g);
        });
      }
    }
#ifdef SED_STAMPS
    if (lane == 0 && p.stamps)
      for (int k = 0; k < 8; ++k) p.stamps[(size_t)(blockIdx.x * wg_waves + wave) * 8 + k] = st_acc[k];
#endif
  }
#ifdef SED_TIMELINE
  if (lane == 0 && p.stamps) {
    unsigned long long* o = p.stamps + (size_t)(blockIdx.x * wg_waves + wave) * 4;
    o[0] = tl_start; o[1] = tl_loop; o[2] = __builtin_amdgcn_s_memrealtime();
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    o[3] = ((unsigned long long)xcc << 32) | (unsigned)(tl_iters & 0xffff);
  }
#endif
}

// ---- mel projection + log of a materialised spectrogram (LogmelFilterBank.forward) -------------------------
struct MelRowsParams {
  const float* spec;
  float* out;
  int64_t rows;
  int32_t nbins, n_mels, log_kind, affine, slots, wcount;
  float amin, db_offset, ln_offset;
  const MelEntry* meta;
  const float* melw;
  const float* aff_scale;
  const float* aff_shift;
};

// 16 lanes per row, 16 rows per workgroup pass; every lane evaluates its share of the banded filters.
// NV = values per thread of a tile of 16 rows (ceil(16 * nbins / 256)).  The next tile's values are loaded into registers
// while the current tile is summed, so a workgroup never waits for global memory with nothing else to do.
template <int NV>
__global__ void __launch_bounds__(256) mel_rows_kernel(const MelRowsParams p) {
  extern __shared__ __align__(16) float smem[];
  const int tid = threadIdx.x;
  const int pitch = p.nbins + 3;
  float* rowsL = smem;                                    // 16 rows
  float* melbuf = smem + 16 * pitch;                      // 16 x n_mels
  MelEntry* metaL = (MelEntry*)(smem + ((16 * pitch + 16 * p.n_mels + 3) & ~3));
  float* melwL = (float*)(metaL + 16 * p.slots);
  for (int i = tid; i < 16 * p.slots; i += 256) metaL[i] = p.meta[i];
  for (int i = tid; i < p.wcount; i += 256) melwL[i] = p.melw[i];
  // a tile's rows are contiguous in memory: element tid + 256 j of the tile for j = 0..NV-1 (clamped addresses, never
  // predicated loads)
  float v[NV];
  auto fetch = [&](int64_t r0) {
    const int n = (int)min((int64_t)16, p.rows - r0) * p.nbins;
    const float* src = p.spec + r0 * p.nbins;
    static_for<0, NV>([&](auto Kf) {
      constexpr int j = decltype(Kf)::value;
      v[j] = src[min(tid + 256 * j, n - 1)];
    });
  };
  const int64_t stride = (int64_t)gridDim.x * 16;
  int64_t r0 = (int64_t)blockIdx.x * 16;
  if (r0 < p.rows) fetch(r0);
  for (; r0 < p.rows; r0 += stride) {
    const int nr = (int)min((int64_t)16, p.rows - r0);
    const int n = nr * p.nbins;
    __syncthreads();  // the previous tile's sums and stores are done with the LDS
    {
      // row and bin of a thread's element advance by 256 from one to the next (a row has more than 256 bins, so at most
      // one wrap per step)
      int r = tid / p.nbins, k = tid - r * p.nbins;
      static_for<0, NV>([&](auto Kf) {
        constexpr int j = decltype(Kf)::value;
        if (tid + 256 * j < n) rowsL[r * pitch + k] = v[j];
        k += 256;
        if (k >= p.nbins) { k -= p.nbins; ++r; }
      });
    }
    __syncthreads();
    if (r0 + stride < p.rows) fetch(r0 + stride);  // in flight during the sums
    const int r = tid >> 4, l = tid & 15;
    if (r < nr) {
      for (int j = 0; j < p.slots; ++j) {
        const MelEntry e = metaL[l * p.slots + j];
        if (e.mel < 0) continue;
        const float* x = rowsL + r * pitch + e.start;
        const float* w = melwL + e.woff;
        float acc = 0.0f;
        for (int i = 0; i < e.len; ++i) acc = __builtin_fmaf(x[i], w[i], acc);
        float val = log_epilogue(acc, p.log_kind, p.amin, p.db_offset, p.ln_offset);
        if (p.affine) val = __builtin_fmaf(val, p.aff_scale[e.mel], p.aff_shift[e.mel]);
        melbuf[r * p.n_mels + e.mel] = val;
      }
    }
    __syncthreads();
    for (int i = tid; i < nr * p.n_mels; i += 256) p.out[r0 * p.n_mels + i] = melbuf[i];
  }
}

// dB post-pass: one workgroup per group; max reduction then subtract / floor.
__global__ void __launch_bounds__(1024) db_normalize_kernel(float* x, int64_t group_elems, int subtract_max, float top_db) {
  // 16 waves per group and 16 bytes per access where the group allows it: a clip's values (a few hundred KB) are read
  // twice, the second time from L2
  __shared__ float red[16];
  float* g = x + (int64_t)blockIdx.x * group_elems;
  const int tid = threadIdx.x;
  const bool vec = (group_elems % 4 == 0) && (((uintptr_t)g & 15) == 0);
  const int64_t n4 = vec ? group_elems / 4 : 0;
  float m = -INFINITY;
  if (vec) {
    for (int64_t i = tid; i < n4; i += 1024) {
      const v4 v = reinterpret_cast<const v4*>(g)[i];
      m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
  } else {
    for (int64_t i = tid; i < group_elems; i += 1024) m = fmaxf(m, g[i]);
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = red[0];
  for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
  const float sub = subtract_max ? m : 0.0f;
  const float floor_v = (top_db >= 0.0f) ? (m - sub) - top_db : -INFINITY;
  if (vec) {
    for (int64_t i = tid; i < n4; i += 1024) {
      v4 v = reinterpret_cast<const v4*>(g)[i];
      v.x = fmaxf(v.x - sub, floor_v); v.y = fmaxf(v.y - sub, floor_v);
      v.z = fmaxf(v.z - sub, floor_v); v.w = fmaxf(v.w - sub, floor_v);
      reinterpret_cast<v4*>(g)[i] = v;
    }
  } else {
    for (int64_t i = tid; i < group_elems; i += 1024) g[i] = fmaxf(g[i] - sub, floor_v);
  }
}

// The same post-pass for few, large groups (torchlibrosa's top_db clamps against the maximum of the WHOLE tensor): a group
// is cut into `chunks` pieces of `chunk` floats, one workgroup each.  Pass 1 writes the maximum of every piece to
// partial[group * chunks + piece]; pass 2 reduces its group's row of partial maxima (they are few) and rewrites its own
// piece.  max() is exact and order-free, so the result has the bits of the one-workgroup kernel.
__global__ void __launch_bounds__(256) db_partial_max_kernel(const float* x, int64_t group_elems, int64_t chunk, float* partial) {
  __shared__ float red[4];
  const int64_t lo = (int64_t)blockIdx.x * chunk, hi = min(group_elems, lo + chunk);
  const float* g = x + (int64_t)blockIdx.y * group_elems;
  float m = -INFINITY;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) m = fmaxf(m, g[i]);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__global__ void __launch_bounds__(256) db_apply_kernel(float* x, int64_t group_elems, int64_t chunk, const float* partial,
                                                       int subtract_max, float top_db) {
  __shared__ float red[4];
  const float* row = partial + (int64_t)blockIdx.y * gridDim.x;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) m = fmaxf(m, row[i]);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sub = subtract_max ? m : 0.0f;
  const float floor_v = (top_db >= 0.0f) ? (m - sub) - top_db : -INFINITY;
  const int64_t lo = (int64_t)blockIdx.x * chunk, hi = min(group_elems, lo + chunk);
  float* g = x + (int64_t)blockIdx.y * group_elems;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) g[i] = fmaxf(g[i] - sub, floor_v);
}

// Scratch of sed_db_normalize_static: partial maxima, one float per (group, piece).  Every GPU holds its own copy; the
// host reaches it through the address hipGetSymbolAddress returns for the current device, never through this name.
constexpr int kDbStaticFloats = 16384;
static __device__ __align__(256) float db_static_scratch[kDbStaticFloats];

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
  // every table of the fused / spectrogram kernels in LDS order (see ImageLayout)
  std::vector<float> image_host;
  float* d_image = nullptr;
  ImageLayout im{};
  int mel_sets = 0, mel_u = 0, mel_ulo = 0, mel_nhi = 0, mel_maxpos = 0, n_quads = 0;
  // banded mel matrix scheduled over 16 lanes (mel_rows_kernel)
  MelEntry* d_meta16 = nullptr;
  float* d_melw16 = nullptr;
  int mel_slots16 = 0, mel_wcount16 = 0;
  std::vector<float> scale_host, shift_host;
  float* d_scale = nullptr;
  float* d_shift = nullptr;
  bool affine = false;
  float db_offset = 0.0f;
  bool pcm_div3 = false;  // see pcm16_to_f32
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

static inline int round4(int v) { return (v + 3) & ~3; }
static inline int round64(int v) { return (v + 63) & ~63; }

// Builds the LDS image of a plan: window (with the factor 1/2 of the real-FFT split), twiddles, and the mel
// projection scheduled for the matrix pipe with its per-lane constants.
template <class G>
static void build_image(sed_plan* pl) {
  const int F = pl->cfg.n_fft / 2 + 1, M = pl->cfg.n_mels;
  const MelMfmaSchedule sc = schedule_mel_mfma(pl->melW_host, F, M, G::PP, kMaxMelSets);
  pl->mel_sets = sc.sets;
  pl->mel_u = sc.u_hi;
  pl->mel_ulo = sc.u_lo;
  pl->mel_nhi = sc.n_hi;
  pl->mel_maxpos = sc.max_pos;
  pl->n_quads = sc.n_quads;
  ImageLayout im{};
  int off = G::WIN_FLOATS;
  im.off_tw1 = off; off += G::TW1_FLOATS;
  im.off_tw2 = off; off += G::TW2_FLOATS;
  im.floats_spec = round4(off);
  off = im.floats_spec;
  // per-lane words and affine tables are sized for the plan: sets actually used, one or two gather rounds, n_mels
  const int rounds = sc.n_quads > 16 ? 2 : 1;
  im.aff_stride = M <= 64 ? 64 : 128;
  im.off_mela = off; off += (int)sc.a.size();
  im.off_boff = off; off += sc.sets * 64;
  im.off_gw = off; off += rounds * 64;
  im.off_aff = off; off += 2 * im.aff_stride;
  im.off_zero = off; off += 4;
  im.floats_fused = round4(off);
  std::vector<float> img(im.floats_fused, 0.0f);
  for (int i = 0; i < G::NFFT; ++i) img[win_slot<G>(i)] = (float)(0.5 * pl->window_host[i]);
  const std::vector<cplx> t1 = build_tw1<G>();
  for (int k1 = 0; k1 < G::R1; ++k1)
    for (int n2 = 0; n2 < G::R2; ++n2) {
      img[im.off_tw1 + tw1_slot<G>(k1, n2)] = t1[k1 * G::R2 + n2].x;
      img[im.off_tw1 + tw1_slot<G>(k1, n2) + 1] = t1[k1 * G::R2 + n2].y;
    }
  const std::vector<cplx> t2 = build_tw2<G>();
  for (int t = 0; t < G::SLOTS; ++t)  // (the odd slot count is padded with zeros: slots are read in pairs)
    for (int l = 0; l < G::LPF; ++l) {
      img[im.off_tw2 + tw2_slot<G>(t, l)] = t2[t * G::LPF + l].x;
      img[im.off_tw2 + tw2_slot<G>(t, l) + 1] = t2[t * G::LPF + l].y;
    }
  std::copy(sc.a.begin(), sc.a.end(), img.begin() + im.off_mela);
  int32_t* boff = reinterpret_cast<int32_t*>(img.data() + im.off_boff);
  for (int s = 0; s < sc.sets; ++s)
    for (int ln = 0; ln < 64; ++ln)  // low half: the lane's first power value; high half: where its partial sums go
      boff[s * 64 + ln] = ((ln & 3) * G::PP + sc.base[s * 16 + (ln >> 2)]) |
                          ((sc.slot[s * 16 + (ln >> 2)] * kPartStride + (ln & 3) * 4) << 16);
  int32_t* gw = reinterpret_cast<int32_t*>(img.data() + im.off_gw);
  for (int ol = 0; ol < 64 * rounds; ++ol) {  // gather word of output lane ol: lane 4*quad + frame sums the partial sums of its quad
    const int q = ol >> 2, jj = ol & 3;
    // lanes without a quad read the zero block: offset of a position that holds finite values, zero positions beyond the first
    gw[ol] = q < sc.n_quads ? (((sc.quadpos[q] & 0xffff) * kPartStride + jj * 4) | ((sc.quadpos[q] >> 16) << 16)) : (1 << 16);
  }
  for (int m = 0; m < im.aff_stride; ++m) {
    img[im.off_aff + m] = (pl->affine && m < M) ? pl->scale_host[m] : 0.0f;
    img[im.off_aff + im.aff_stride + m] = (pl->affine && m < M) ? pl->shift_host[m] : 0.0f;
  }
  pl->im = im;
  pl->image_host.swap(img);
}

static int upload_tables(sed_plan* pl) {
  if (pl->geo == 0) build_image<Geo512>(pl);
  else if (pl->geo == 1) build_image<Geo1024>(pl);
  else build_image<Geo2048>(pl);
  int rc = upload(&pl->d_image, pl->image_host);
  if (rc) return rc;
  const int F = pl->cfg.n_fft / 2 + 1, M = pl->cfg.n_mels;
  MelSchedule s16 = schedule_mel(pl->melW_host, F, M, 16);
  pl->mel_slots16 = s16.slots;
  pl->mel_wcount16 = (int)s16.weights.size();
  rc = upload(&pl->d_meta16, s16.entries);
  if (rc) return rc;
  return upload(&pl->d_melw16, s16.weights);
}

static std::vector<float> build_mel(int mel_kind, int sr, int n_fft, int n_mels, double fmin, double fmax) {
  if (mel_kind == SED_MEL_HTK_TF) return mel_htk_tf(sr, n_fft, n_mels, fmin, fmax);
  if (mel_kind == SED_MEL_KALDI) return mel_kaldi(sr, n_fft, n_mels, fmin, fmax);
  return mel_slaney(sr, n_fft, n_mels, fmin, fmax);
}

extern "C" const char* sed_version(void) { return "sedhip 0.3.0 (gfx950)"; }

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

extern "C" int sed_mel_filterbank(int32_t mel_kind, int32_t sample_rate, int32_t n_fft, int32_t n_mels, double fmin,
                                  double fmax, float* melW) {
  if (!melW || sample_rate < 1 || n_fft < 2 || n_mels < 1 || !(fmin >= 0.0)) return SED_ERR_BAD_ARG;
  if (mel_kind == SED_MEL_KALDI ? !((fmax <= 0.0 ? fmax + 0.5 * sample_rate : fmax) > fmin) : !(fmax > fmin)) return SED_ERR_BAD_ARG;
  const std::vector<float> w = build_mel(mel_kind, sample_rate, n_fft, n_mels, fmin, fmax);
  std::memcpy(melW, w.data(), w.size() * sizeof(float));
  return SED_OK;
}

extern "C" int sed_window(int32_t window_kind, int32_t n_fft, int32_t win_length, int32_t win_align, float* window) {
  if (!window || win_length < 1 || win_length > n_fft) return SED_ERR_BAD_ARG;
  if (window_kind != SED_WINDOW_HANN_PERIODIC && window_kind != SED_WINDOW_HANN_SYMMETRIC) return SED_ERR_BAD_ARG;
  const std::vector<double> w = build_window(n_fft, win_length, win_align == SED_WIN_LEFT, window_kind == SED_WINDOW_HANN_SYMMETRIC);
  for (int i = 0; i < n_fft; ++i) window[i] = (float)w[i];
  return SED_OK;
}

extern "C" int sed_hann_window(int32_t n_fft, int32_t win_length, int32_t win_align, float* window) {
  return sed_window(SED_WINDOW_HANN_PERIODIC, n_fft, win_length, win_align, window);
}

extern "C" int sed_plan_create(const sed_frontend_cfg* cfg, sed_plan** out) {
  if (!cfg || !out) return SED_ERR_BAD_ARG;
  *out = nullptr;
  int geo;
  if (cfg->n_fft == 512) geo = 0;
  else if (cfg->n_fft == 1024) geo = 1;
  else if (cfg->n_fft == 2048) geo = 2;
  else return SED_ERR_UNSUPPORTED_NFFT;
  if (cfg->win_length < 1 || cfg->win_length > cfg->n_fft) return SED_ERR_BAD_ARG;
  if (cfg->hop < 1 || cfg->hop > cfg->n_fft) return SED_ERR_BAD_ARG;
  // frames are read from LDS 8 or 16 bytes at a time: the hop must keep them aligned, except with per-frame
  // conditioning, which reads sample by sample (Kaldi's 10 ms shift at 44.1 kHz is 441 samples)
  if (cfg->n_mels < 1 || cfg->n_mels > 128 || cfg->sample_rate < 1) return SED_ERR_BAD_ARG;
  if (cfg->spectrum != SED_SPEC_MAGNITUDE && cfg->spectrum != SED_SPEC_POWER) return SED_ERR_BAD_ARG;
  if (cfg->center && cfg->pad_mode != SED_PAD_CONSTANT && cfg->pad_mode != SED_PAD_REFLECT) return SED_ERR_BAD_ARG;
  if (cfg->pcm_kind == SED_PCM_I16 && !(cfg->pcm_divisor > 0.0f)) return SED_ERR_BAD_ARG;
  if (cfg->tail_min_samples > 0 && cfg->tail_hop_samples < 1) return SED_ERR_BAD_ARG;
  if (!(cfg->fmin >= 0.0)) return SED_ERR_BAD_ARG;
  if (cfg->mel_kind == SED_MEL_KALDI ? !((cfg->fmax <= 0.0 ? cfg->fmax + 0.5 * cfg->sample_rate : cfg->fmax) > cfg->fmin)
                                     : !(cfg->fmax > cfg->fmin)) return SED_ERR_BAD_ARG;
  if (cfg->window_kind != SED_WINDOW_HANN_PERIODIC && cfg->window_kind != SED_WINDOW_HANN_SYMMETRIC) return SED_ERR_BAD_ARG;
  if (cfg->preemphasis < 0.0f || cfg->preemphasis >= 1.0f || (cfg->remove_dc != 0 && cfg->remove_dc != 1)) return SED_ERR_BAD_ARG;
  // per-frame conditioning (Kaldi framing) is built for the power spectrum of uncentred, left-aligned frames
  if ((cfg->remove_dc || cfg->preemphasis > 0.0f) &&
      (cfg->center || cfg->win_align != SED_WIN_LEFT || cfg->spectrum != SED_SPEC_POWER)) return SED_ERR_BAD_ARG;

  sed_plan* pl = new (std::nothrow) sed_plan();
  if (!pl) return SED_ERR_BAD_ARG;
  pl->cfg = *cfg;
  pl->geo = geo;
  int rc = hip_status(hipGetDevice(&pl->device));
  hipDeviceProp_t prop;
  if (!rc) rc = hip_status(hipGetDeviceProperties(&prop, pl->device));
  if (rc) { delete pl; return rc; }
  pl->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

  pl->window_host = build_window(cfg->n_fft, cfg->win_length, cfg->win_align == SED_WIN_LEFT,
                                 cfg->window_kind == SED_WINDOW_HANN_SYMMETRIC);
  pl->melW_host = build_mel(cfg->mel_kind, cfg->sample_rate, cfg->n_fft, cfg->n_mels, cfg->fmin, cfg->fmax);
  pl->db_offset = (float)(10.0 * std::log10(std::max((double)cfg->amin, (double)cfg->ref)));

  if (cfg->pcm_kind == SED_PCM_I16) {
    const float d = cfg->pcm_divisor, r = 1.0f / d;
    bool same = std::isfinite(r) && r > 0.0f;
    for (int x = -32768; same && x <= 32767; ++x) {
      const float xf = (float)x, q = xf * r;
      const float q1 = std::fmaf(std::fmaf(-q, d, xf), r, q);
      const float want = xf / d;
      same = std::memcmp(&q1, &want, sizeof(float)) == 0 || (q1 == 0.0f && want == 0.0f);
    }
    pl->pcm_div3 = same;
  }

  rc = upload_tables(pl);
  if (rc) { sed_plan_destroy(pl); return rc; }
  *out = pl;
  return SED_OK;
}

extern "C" void sed_plan_destroy(sed_plan* pl) {
  if (!pl) return;
  void* bufs[] = {pl->d_image, pl->d_meta16, pl->d_melw16, pl->d_scale, pl->d_shift};
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
  return upload_tables(pl);
}

extern "C" int sed_plan_set_window(sed_plan* pl, const float* window) {
  if (!pl || !window) return SED_ERR_BAD_ARG;
  for (int i = 0; i < pl->cfg.n_fft; ++i)
    if (!std::isfinite(window[i])) return SED_ERR_BAD_ARG;
  int rc = check_device(pl);
  if (rc) return rc;
  SED_HIP_TRY(hipDeviceSynchronize());
  for (int i = 0; i < pl->cfg.n_fft; ++i) pl->window_host[i] = (double)window[i];
  return upload_tables(pl);
}

extern "C" int sed_plan_set_affine(sed_plan* pl, const float* scale, const float* shift) {
  if (!pl || ((scale == nullptr) != (shift == nullptr))) return SED_ERR_BAD_ARG;
  int rc = check_device(pl);
  if (rc) return rc;
  SED_HIP_TRY(hipDeviceSynchronize());
  pl->affine = scale != nullptr;
  const int M = pl->cfg.n_mels;
  if (scale) {
    pl->scale_host.assign(scale, scale + M);
    pl->shift_host.assign(shift, shift + M);
    rc = upload(&pl->d_scale, pl->scale_host);
    if (!rc) rc = upload(&pl->d_shift, pl->shift_host);
    if (rc) return rc;
  }
  // Only the 256 affine floats of the LDS image change: the mel schedule and the other tables stay as they are.
  const int stride = pl->im.aff_stride;
  if (!pl->d_image || stride < 1 || pl->image_host.size() < (size_t)pl->im.off_aff + 2 * stride) return upload_tables(pl);
  float* aff = pl->image_host.data() + pl->im.off_aff;
  for (int m = 0; m < stride; ++m) {
    aff[m] = (pl->affine && m < M) ? pl->scale_host[m] : 0.0f;
    aff[stride + m] = (pl->affine && m < M) ? pl->shift_host[m] : 0.0f;
  }
  return hip_status(hipMemcpy(pl->d_image + pl->im.off_aff, aff, 2 * stride * sizeof(float), hipMemcpyHostToDevice));
}

template <class G, bool MAG, bool SPEC_OUT, bool PRE, class Mel>
static int launch_front_shape(const sed_plan* pl, FrontParams p, int64_t B, hipStream_t stream) {
  // per-wave LDS region
  p.span = round4((G::FPW - 1) * p.hop + G::NFFT);
  p.span_blocks = (p.span + 255) / 256;
  if (p.span_blocks > kMaxSpanBlocks) return SED_ERR_BAD_ARG;
  constexpr int SKEW = span_skew<G, Mel, PRE, SPEC_OUT>();
  // floats of the region an early span transfer writes: whole 1 KB blocks, or the skewed span exactly
  const int span_floats = SKEW > 0 ? skewed_floats(p.span, Mel::HOP, SKEW) : 256 * p.span_blocks;
  // The partial sums of the mel stage normally sit behind the power rows.  When the operands of ALL matrix instructions
  // of a group are fetched in one burst (the compiled-in shapes of 4-frame geometries), the power rows are dead by the time
  // the partial sums are stored (the LDS serves a wavefront's accesses in order), so the sums may lie on top of them:
  // right behind the area the next span's early transfer writes.
  constexpr bool kOverlay = !SPEC_OUT && G::FPW == 4 && Mel::kStatic && Mel::STEPS <= (G::NFFT == 2048 ? SED_BURST2048 : 12);
  p.part_off = std::max(G::FPW * G::PP, SKEW > 0 ? span_floats : 0);  // (a skewed span may reach past the power rows)
#ifndef SED_NO_MEL_ONE_BURST
  if (kOverlay && span_floats < G::FPW * G::PP) p.part_off = round64(span_floats);
#endif
  p.early = p.dma && !SPEC_OUT && span_floats <= p.part_off;  // the span may land while the partial sums are in use
  const int part_floats = SPEC_OUT ? 0 : p.mel_sets * 16 * kPartStride;
  p.pre_off = 256 * p.span_blocks;  // behind the staged span: FPW frames x LPF partial sums (Kaldi DC removal)
  p.wave_floats = round64(std::max(std::max(span_floats + (PRE ? 64 : 0), G::FPW * G::XFR), p.part_off + part_floats));
  // shared tables (the plan's image), then the work-queue head, then the wave regions
  p.im = pl->im;
  p.image = pl->d_image;
  p.image_floats = SPEC_OUT ? pl->im.floats_spec : pl->im.floats_fused;
  p.off_queue = p.image_floats;
  p.off_wave = round64(p.off_queue + 4);
  // as many wavefronts per workgroup as the geometry allows and LDS holds
  int waves = wg_waves_of<G, SPEC_OUT>();
#ifdef SED_DIAGNOSTIC_BUILD
  static const int debug_waves = getenv("SED_DEBUG_WAVES") ? atoi(getenv("SED_DEBUG_WAVES")) : 0;
  if (debug_waves > 0) waves = std::min(waves, debug_waves);
#endif
  while (waves > 1 && (size_t)(p.off_wave + waves * p.wave_floats) * sizeof(float) > 160 * 1024) --waves;
  // groups of FPW frames; one workgroup per CU, each with a contiguous share of the groups
  p.groups_per_clip = (p.T + G::FPW - 1) / G::FPW;
  const int64_t total_groups = (int64_t)p.groups_per_clip * B;
  if (total_groups > INT32_MAX / 2) return SED_ERR_BAD_ARG;
  p.total_groups = (int32_t)total_groups;
  // few groups (a handful of streams): rather every CU with a few wavefronts than a few CUs with all of theirs (measured
  // 11.1 -> 9.4 us for 1 .. 4 streams of 10 s).  A large plan image is copied into LDS by at least two wavefronts.
  if (total_groups < (int64_t)waves * pl->num_cus) {
    const int64_t floor_waves = p.image_floats > 8192 ? 2 : 1;
    waves = (int)std::min<int64_t>(waves, std::max<int64_t>(floor_waves, (total_groups + pl->num_cus - 1) / pl->num_cus));
  }
  const size_t lds_bytes = (size_t)(p.off_wave + waves * p.wave_floats) * sizeof(float);
#ifdef SED_DIAGNOSTIC_BUILD
  static const bool debug_shape = getenv("SED_DEBUG_SHAPE") != nullptr;
  if (debug_shape)
    fprintf(stderr, "shape: image %d floats, region %d (span %d, exchange %d, power rows %d, partial sums at %d + %d), %d wavefronts, %zu bytes\n",
            p.image_floats, p.wave_floats, span_floats, G::FPW * G::XFR, G::FPW * G::PP, p.part_off, part_floats, waves, lds_bytes);
#endif
  if (lds_bytes > 160 * 1024) return SED_ERR_BAD_ARG;
#ifdef SED_ONLY_BENCH  // diagnostic builds: one instantiation only, seconds to compile
  if ((p.pcm_kind & ~kPcmDiv3) == SED_PCM_I16 || p.seg_off) return SED_ERR_BAD_ARG;
  auto kern = stft_mel_kernel<G, MAG, SPEC_OUT, PRE, false, Mel>;
#else
  auto kern = (p.pcm_kind & ~kPcmDiv3) == SED_PCM_I16 ? stft_mel_kernel<G, MAG, SPEC_OUT, PRE, true, Mel>
                                                       : stft_mel_kernel<G, MAG, SPEC_OUT, PRE, false, Mel>;
  if (p.seg_off) {  // clips are segments of a resident buffer: built for the fused log-mel kernels
    if constexpr (SPEC_OUT) return SED_ERR_BAD_ARG;
    else kern = (p.pcm_kind & ~kPcmDiv3) == SED_PCM_I16 ? stft_mel_kernel<G, MAG, false, PRE, true, Mel, true>
                                                         : stft_mel_kernel<G, MAG, false, PRE, false, Mel, true>;
  }
#endif
  // the dynamic LDS limit is a property of the kernel function, set once per instantiation (per device: the attribute
  // lives in the device's code object) to the hardware maximum, so that concurrent launches with different sizes
  // never race on it
  {
    static std::mutex mu;
    static std::vector<std::pair<const void*, int>> done;
    std::lock_guard<std::mutex> lock(mu);
    const std::pair<const void*, int> key{reinterpret_cast<const void*>(kern), pl->device};
    if (std::find(done.begin(), done.end(), key) == done.end()) {
      SED_HIP_TRY(hipFuncSetAttribute(key.first, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      done.push_back(key);
    }
  }
  const int blocks = (int)std::min<int64_t>((total_groups + waves - 1) / waves, (int64_t)pl->num_cus);
#ifdef SED_TIMELINE
  static unsigned long long* d_tl = nullptr;
  const size_t n_tl = (size_t)blocks * waves * 4;
  if (!d_tl) (void)hipMalloc((void**)&d_tl, 8192 * 4 * sizeof(unsigned long long));
  p.stamps = d_tl;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), lds_bytes, stream, p);
  static int tl_calls = 0;
  if (++tl_calls == 100) {
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(n_tl);
    (void)hipMemcpy(h.data(), d_tl, n_tl * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t1 = 0;
    for (size_t w = 0; w < n_tl / 4; ++w) { t0 = std::min(t0, h[w * 4]); t1 = std::max(t1, h[w * 4 + 2]); }
    // s_memrealtime ticks at 100 MHz
    double sum_res = 0, sum_start = 0, sum_end = 0, sum_setup = 0, max_start = 0, min_end = 1e30;
    double xe[8] = {0}, xmax[8] = {0}; int xn[8] = {0};
    for (size_t w = 0; w < n_tl / 4; ++w) {
      const double a = (h[w * 4] - t0) * 0.01, l = (h[w * 4 + 1] - t0) * 0.01, e = (h[w * 4 + 2] - t0) * 0.01;
      sum_res += e - a; sum_start += a; sum_end += e; sum_setup += l - a;
      max_start = std::max(max_start, a); min_end = std::min(min_end, e);
      const int x = (int)((h[w * 4 + 3] >> 32) & 7);
      xe[x] += e; xmax[x] = std::max(xmax[x], e); ++xn[x];
    }
    const double nw = (double)(n_tl / 4);
    fprintf(stderr, "TIMELINE waves %zu  span %.2f us  mean start %.2f (max %.2f)  mean setup %.2f  mean end %.2f (min %.2f)  mean residency %.2f us\n",
            n_tl / 4, (t1 - t0) * 0.01, sum_start / nw, max_start, sum_setup / nw, sum_end / nw, min_end, sum_res / nw);
    {
      double iters = 0; int max_iters = 0, min_iters = 1 << 30;
      for (size_t w = 0; w < n_tl / 4; ++w) {
        const int it = (int)(h[w * 4 + 3] & 0xffff);
        iters += it; max_iters = std::max(max_iters, it); min_iters = std::min(min_iters, it);
      }
      fprintf(stderr, "TIMELINE groups per wave %.2f (min %d, max %d)\n", iters / nw, min_iters, max_iters);
    }
    for (int x = 0; x < 8; ++x) if (xn[x]) fprintf(stderr, "TIMELINE xcc %d: waves %d mean end %.2f max end %.2f\n", x, xn[x], xe[x] / xn[x], xmax[x]);
    int hist[20] = {0};
    for (size_t w = 0; w < n_tl / 4; ++w) { int bb = (int)((h[w * 4 + 2] - t0) * 0.01 / ((t1 - t0) * 0.01) * 20); hist[std::min(bb, 19)]++; }
    fprintf(stderr, "TIMELINE end-time histogram (20 bins over the span):");
    for (int bb = 0; bb < 20; ++bb) fprintf(stderr, " %d", hist[bb]);
    fprintf(stderr, "\n");
  }
  return hip_status(hipGetLastError());
#elif defined(SED_STAMPS)
  static unsigned long long* d_stamps = nullptr;
  const size_t n_st = (size_t)blocks * waves * 8;
  if (!d_stamps) { (void)hipMalloc((void**)&d_stamps, 4096 * 8 * 8 * sizeof(unsigned long long)); (void)hipMemset(d_stamps, 0, 4096 * 8 * 8 * sizeof(unsigned long long)); }
  p.stamps = d_stamps;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), lds_bytes, stream, p);
  static int calls = 0;
  if (++calls == 100) {
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(n_st);
    (void)hipMemcpy(h.data(), d_stamps, n_st * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sum[8] = {0};
    for (size_t w = 0; w < n_st / 8; ++w) for (int k = 0; k < 8; ++k) sum[k] += (double)h[w * 8 + k];
    const char* names[8] = {"queue pop + span issue", "wait for the span", "stores + stage 1", "exchange", "stage 2 + power", "mel mfma", "span order + gather + log", "-"};
    double tot = 0; for (int k = 0; k < 8; ++k) tot += sum[k];
    for (int k = 0; k < 7; ++k) fprintf(stderr, "STAMP %-28s %10.0f ticks/wave  %5.1f%%\n", names[k], sum[k] / (n_st / 8), 100.0 * sum[k] / tot);
    fprintf(stderr, "STAMP total %10.0f ticks/wave over %zu waves\n", tot / (n_st / 8), n_st / 8);
  }
  return hip_status(hipGetLastError());
#else
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), lds_bytes, stream, p);
  return hip_status(hipGetLastError());
#endif
}

// Picks the compiled-in mel shape when the plan's schedule has it (the flavour's default filterbank does).
template <class G, bool MAG, bool SPEC_OUT, bool PRE_ = false>
static int launch_front(const sed_plan* pl, const FrontParams& p, int64_t B, hipStream_t stream) {
  constexpr bool PRE = PRE_;
  if constexpr (!SPEC_OUT) {
    const bool two = p.n_quads > 16;
    // (sets, steps of the first n_hi sets, steps of the others, n_hi, most positions per quad, two gather rounds)
    auto is = [&](int sets, int u, int ulo, int nhi, int maxpos, bool rounds2) {
      // (hops off the 16-byte grid run the general kernel, except with frame conditioning, which reads sample by sample)
      return (PRE || p.hop % 4 == 0) && p.mel_sets == sets && p.mel_u == u && p.mel_ulo == ulo && p.mel_nhi == nhi &&
             p.mel_maxpos == maxpos && two == rounds2;
    };
    if constexpr (G::NFFT == 1024 && !MAG && !PRE) {  // HTS-AT: Slaney 64 mel, 50..14000 Hz at 32 kHz
#ifdef SED_EST_MEL_U  // timing experiments: a shorter chain per accumulator set than the plan's schedule has (results are then garbage)
      if (is(2, 6, 4, 1, 5, false) && p.hop == 320) return launch_front_shape<G, MAG, SPEC_OUT, PRE, MelStatic<2, SED_EST_MEL_U, 5, 1, 320>>(pl, p, B, stream);
#endif
      if (is(2, 6, 4, 1, 5, false) && p.hop == 320) return launch_front_shape<G, MAG, SPEC_OUT, PRE, MelStatic<2, 6, 5, 1, 320, 4, 1>>(pl, p, B, stream);
      if (is(2, 6, 4, 1, 5, false)) return launch_front_shape<G, MAG, SPEC_OUT, PRE, MelStatic<2, 6, 5, 1, 0, 4, 1>>(pl, p, B, stream);
      // the same model at 44.1 kHz, the rate the reference itself runs it at (src/models/htsat/model.py:171-186)
      if (is(4, 4, 1, 1, 6, false) && p.hop == 320) return launch_front_shape<G, MAG, SPEC_OUT, PRE, MelStatic<4, 4, 6, 1, 320, 1, 1>>(pl, p, B, stream);
    }
    if constexpr (G::NFFT == 512 && MAG && !PRE) {  // YAMNet: HTK 64 mel, 125..7500 Hz at 16 kHz
      if (is(2, 4, 2, 1, 3, false)) return launch_front_shape<G, MAG, SPEC_OUT, PRE, MelStatic<2, 4, 3, 1, 0, 2, 1>>(pl, p, B, stream);
    }
    if constexpr (G::NFFT == 2048 && MAG && !PRE) {  // CRNN: Slaney 40 mel, 0..22050 Hz at 44.1 kHz
      if (is(3, 14, 3, 1, 9, false)) return launch_front_shape<G, MAG, SPEC_OUT, PRE, MelStatic<3, 14, 9, 1, 0, 3, 1>>(pl, p, B, stream);
    }
    if constexpr (G::NFFT == 2048 && !MAG && !PRE) {  // dB mel: Slaney 64 mel, 0..22050 Hz at 44.1 kHz
      if (is(3, 11, 5, 1, 7, false)) return launch_front_shape<G, MAG, SPEC_OUT, PRE, MelStatic<3, 11, 7, 1, 0, 5, 1>>(pl, p, B, stream);
    }
    if constexpr (G::NFFT == 512 && !MAG && PRE) {  // Kaldi fbank: 128 mel, 20..8000 Hz at 16 kHz
      if (is(3, 3, 2, 1, 3, true)) return launch_front_shape<G, MAG, SPEC_OUT, PRE, MelStatic<3, 3, 3, 2, 0, 2, 1>>(pl, p, B, stream);
    }
    if constexpr (G::NFFT == 2048 && !MAG && PRE) {  // the same at 44.1 kHz, the rate of the reference's files (25 ms = 1102 samples)
      if (is(4, 11, 4, 1, 6, true)) return launch_front_shape<G, MAG, SPEC_OUT, PRE, MelStatic<4, 11, 6, 2, 0, 4, 1>>(pl, p, B, stream);
    }
  }
  return launch_front_shape<G, MAG, SPEC_OUT, PRE, MelDynamic>(pl, p, B, stream);
}

template <bool SPEC_OUT>
static int front_forward(const sed_plan* pl, const void* pcm, int64_t B, int64_t L, int64_t pcm_stride, float* out,
                         int64_t out_stride_b, void* stream, const int64_t* seg_off = nullptr, const int32_t* seg_len = nullptr,
                         bool seg_own = false) {
  if (!pl || B < 0 || L < 0) return SED_ERR_BAD_ARG;
  const int64_t T = sed_plan_num_frames(pl, L);
  if (T < 0) return (int)T;
  if (B == 0 || T == 0) return SED_OK;
  if (!pcm || !out || (!seg_off && pcm_stride < L) || ((seg_off == nullptr) != (seg_len == nullptr))) return SED_ERR_BAD_ARG;
  const int64_t row = SPEC_OUT ? pl->cfg.n_fft / 2 + 1 : pl->cfg.n_mels;
  if (out_stride_b < T * row || T > INT32_MAX / 4) return SED_ERR_BAD_ARG;
  int rc = check_device(pl);
  if (rc) return rc;
  const sed_frontend_cfg& c = pl->cfg;
  FrontParams p{};
  p.pcm = pcm;
  p.out = out;
  p.L = L;
  p.pcm_stride = pcm_stride;
  p.out_stride_b = out_stride_b;
  p.T = (int32_t)T;
  p.hop = c.hop;
  p.pad_left = c.center ? c.n_fft / 2 : 0;
  p.pad_mode = c.center ? c.pad_mode : SED_PAD_NONE;
  p.pcm_kind = c.pcm_kind;
  p.pcm_div = c.pcm_kind == SED_PCM_I16 ? c.pcm_divisor : 1.0f;
  if (pl->pcm_div3) p.pcm_kind |= kPcmDiv3;
  p.n_mels = c.n_mels;
  p.log_kind = c.log_kind;
  p.affine = pl->affine ? 1 : 0;
  p.amin = c.amin;
  p.db_offset = pl->db_offset;
  p.ln_offset = c.ln_offset;
  p.remove_dc = c.remove_dc;
  p.preemph = c.preemphasis;
  p.win_length = c.win_length;
  p.inv_win_length = 1.0f / (float)c.win_length;
  p.mel_sets = pl->mel_sets;
  p.mel_u = pl->mel_u;
  p.mel_ulo = pl->mel_ulo;
  p.mel_nhi = pl->mel_nhi;
  p.mel_maxpos = pl->mel_maxpos;
  p.n_quads = pl->n_quads;
  p.aff_stride = pl->im.aff_stride;
  p.seg_off = seg_off;
  p.seg_len = seg_len;
  p.seg_own = seg_own ? 1 : 0;
  p.own_sub = c.center ? 0 : (c.win_align == SED_WIN_LEFT ? c.win_length : c.n_fft);
  p.own_min = c.center ? (c.pad_mode == SED_PAD_REFLECT ? c.n_fft / 2 + 1 : 1) : p.own_sub;
  // (segments start anywhere: the kernel takes the LDS-DMA path for those that start on a 16-byte boundary)
  p.dma = (c.pcm_kind == SED_PCM_F32) && ((uintptr_t)pcm % 16 == 0) && (seg_off || pcm_stride % 4 == 0) && (c.hop % 4 == 0) ? 1 : 0;  // n_fft/2: a multiple of 4
  if (SED_SKIP(2048)) p.dma = 0;  // experiment: register-staged spans for fp32 too
  p.out_vec = (c.n_mels % 4 == 0) && ((uintptr_t)out % 16 == 0) && (out_stride_b % 4 == 0) ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  const bool mag = c.spectrum == SED_SPEC_MAGNITUDE;
#ifdef SED_ONLY_BENCH
  if (SPEC_OUT || mag || pl->geo != 1 || c.remove_dc || c.preemphasis > 0.0f) return SED_ERR_BAD_ARG;
  return launch_front<Geo1024, false, false>(pl, p, B, s);
#endif
  if (c.remove_dc || c.preemphasis > 0.0f) {  // Kaldi framing: power spectrum, fused log-mel output only
    if constexpr (SPEC_OUT) {
      return SED_ERR_BAD_ARG;
    } else {
      if (pl->geo == 0) return launch_front<Geo512, false, false, true>(pl, p, B, s);
      if (pl->geo == 1) return launch_front<Geo1024, false, false, true>(pl, p, B, s);
      return launch_front<Geo2048, false, false, true>(pl, p, B, s);
    }
  }
#define SED_DISPATCH(GEO)                                                         \
  return mag ? launch_front<GEO, true, SPEC_OUT>(pl, p, B, s) : launch_front<GEO, false, SPEC_OUT>(pl, p, B, s)
  if (pl->geo == 0) { SED_DISPATCH(Geo512); }
  if (pl->geo == 1) { SED_DISPATCH(Geo1024); }
  SED_DISPATCH(Geo2048);
#undef SED_DISPATCH
}

extern "C" int sed_logmel_forward(const sed_plan* pl, const void* pcm, int64_t B, int64_t L, int64_t pcm_stride,
                                  float* out, int64_t out_stride_b, void* stream) {
  return front_forward<false>(pl, pcm, B, L, pcm_stride, out, out_stride_b, stream);
}

extern "C" int sed_logmel_forward_segments(const sed_plan* pl, const void* pcm, const int64_t* seg_off, const int32_t* seg_len,
                                           int64_t n_segments, int64_t L, float* out, int64_t out_stride_b, void* stream) {
  if (!seg_off || !seg_len) return SED_ERR_BAD_ARG;
  if (pl && pl->cfg.tail_min_samples > 0) return SED_ERR_BAD_ARG;
  return front_forward<false>(pl, pcm, n_segments, L, L, out, out_stride_b, stream, seg_off, seg_len);
}

extern "C" int sed_logmel_forward_segments_own(const sed_plan* pl, const void* pcm, const int64_t* seg_off,
                                               const int32_t* seg_len, int64_t n_segments, int64_t L, float* out,
                                               int64_t out_stride_b, void* stream) {
  if (!seg_off || !seg_len) return SED_ERR_BAD_ARG;
  if (pl && pl->cfg.tail_min_samples > 0) return SED_ERR_BAD_ARG;
  return front_forward<false>(pl, pcm, n_segments, L, L, out, out_stride_b, stream, seg_off, seg_len, true);
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
  const sed_frontend_cfg& c = pl->cfg;
  MelRowsParams p{};
  p.spec = spec;
  p.out = out;
  p.rows = rows;
  p.nbins = c.n_fft / 2 + 1;
  p.n_mels = c.n_mels;
  p.log_kind = c.log_kind;
  p.affine = pl->affine ? 1 : 0;
  p.slots = pl->mel_slots16;
  p.wcount = pl->mel_wcount16;
  p.amin = c.amin;
  p.db_offset = pl->db_offset;
  p.ln_offset = c.ln_offset;
  p.meta = pl->d_meta16;
  p.melw = pl->d_melw16;
  p.aff_scale = pl->d_scale;
  p.aff_shift = pl->d_shift;
  const size_t lds = sizeof(float) * (((16 * (p.nbins + 3) + 16 * p.n_mels + 3) & ~3) + 4 * 16 * p.slots + p.wcount);
  if (lds > 160 * 1024) return SED_ERR_BAD_ARG;
  auto kfn = c.n_fft == 512 ? mel_rows_kernel<17> : c.n_fft == 1024 ? mel_rows_kernel<33> : mel_rows_kernel<65>;
  const void* kern = reinterpret_cast<const void*>(kfn);
  {
    // the dynamic LDS limit is a property of the kernel function: set once per kernel and device to the hardware maximum,
    // so that concurrent launches with different sizes never race on it
    static std::mutex mu;
    static std::vector<std::pair<const void*, int>> done;
    std::lock_guard<std::mutex> lock(mu);
    const std::pair<const void*, int> key{kern, pl->device};
    if (std::find(done.begin(), done.end(), key) == done.end()) {
      SED_HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      done.push_back(key);
    }
  }
  const int64_t groups = (rows + 15) / 16;
  // as many workgroups per CU as the LDS holds (the tiles are latency-bound: load, barrier, short sums, barrier, store)
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds));
  const int grid = (int)std::min<int64_t>(groups, (int64_t)pl->num_cus * per_cu);
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
  return hip_status(hipGetLastError());
}

// Pieces per group for the two-pass form: about 16 K floats each, at most 1024 per group.
static inline void db_pieces(int64_t group_elems, int64_t* chunk, int64_t* chunks) {
  int64_t c = 16384;
  while ((group_elems + c - 1) / c > 1024) c *= 2;
  *chunk = c;
  *chunks = (group_elems + c - 1) / c;
}

static int db_two_pass(float* x, int64_t groups, int64_t group_elems, int32_t subtract_max, float top_db, float* partial,
                       hipStream_t stream) {
  int64_t chunk, chunks;
  db_pieces(group_elems, &chunk, &chunks);
  const dim3 grid((unsigned)chunks, (unsigned)groups);
  hipLaunchKernelGGL(db_partial_max_kernel, grid, dim3(256), 0, stream, x, group_elems, chunk, partial);
  hipLaunchKernelGGL(db_apply_kernel, grid, dim3(256), 0, stream, x, group_elems, chunk, partial, (int)subtract_max, top_db);
  return hip_status(hipGetLastError());
}

extern "C" int sed_db_normalize(float* x, int64_t groups, int64_t group_elems, int32_t subtract_max, float top_db,
                                void* stream) {
  if (groups < 0 || group_elems < 0 || groups > INT32_MAX) return SED_ERR_BAD_ARG;
  if (groups == 0 || group_elems == 0) return SED_OK;
  if (!x) return SED_ERR_BAD_ARG;
  // many groups (a clip each) fill the GPU with one workgroup per group; few large ones are cut into pieces, whose
  // partial maxima live in a stream-ordered allocation for the duration of the two launches
  if (groups >= 64 || group_elems <= 65536 || groups > 65535) {
    hipLaunchKernelGGL(db_normalize_kernel, dim3((unsigned)groups), dim3(1024), 0, (hipStream_t)stream, x, group_elems,
                       (int)subtract_max, top_db);
    return hip_status(hipGetLastError());
  }
  int64_t chunk, chunks;
  db_pieces(group_elems, &chunk, &chunks);
  float* partial = nullptr;
  SED_HIP_TRY(hipMallocAsync((void**)&partial, (size_t)(groups * chunks) * sizeof(float), (hipStream_t)stream));
  const int rc = db_two_pass(x, groups, group_elems, subtract_max, top_db, partial, (hipStream_t)stream);
  const hipError_t fe = hipFreeAsync(partial, (hipStream_t)stream);
  return rc ? rc : hip_status(fe);
}

extern "C" int sed_db_normalize_static(float* x, int64_t groups, int64_t group_elems, int32_t subtract_max, float top_db,
                                       void* stream) {
  if (groups < 0 || group_elems < 0) return SED_ERR_BAD_ARG;
  if (groups == 0 || group_elems == 0) return SED_OK;
  if (!x) return SED_ERR_BAD_ARG;
  int64_t chunk, chunks;
  db_pieces(group_elems, &chunk, &chunks);
  if (groups > 65535 || groups * chunks > kDbStaticFloats) return SED_ERR_CAPACITY;
  // the scratch array's address on the current device, looked up once per device
  static std::mutex mu;
  static float* ptr_of[16] = {};
  int dev = -1;
  SED_HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return SED_ERR_BAD_ARG;
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing((hipStream_t)stream, &capturing) != hipSuccess) { (void)hipGetLastError(); capturing = hipStreamCaptureStatusNone; }
  if (capturing == hipStreamCaptureStatusNone) {
    // the scratch array is the current device's: data that lives on another GPU is refused (a query, no allocation;
    // left out under stream capture, where the call is stream operations only)
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, x) == hipSuccess) {
      if (attr.type == hipMemoryTypeDevice && attr.device != dev) return SED_ERR_WRONG_DEVICE;
    } else {
      (void)hipGetLastError();  // an unregistered pointer: let the launch report it
    }
  }
  float* partial = nullptr;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (!ptr_of[dev]) {
      float* p = nullptr;
      SED_HIP_TRY(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(db_static_scratch)));
      ptr_of[dev] = p;
    }
    partial = ptr_of[dev];
  }
  // every piece's slot is written by pass 1 before pass 2 reads it: nothing has to start at zero
  return db_two_pass(x, groups, group_elems, subtract_max, top_db, partial, (hipStream_t)stream);
}
