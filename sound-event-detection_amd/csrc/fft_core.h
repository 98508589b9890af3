// Per-lane pieces of the fused STFT: in-register radix-2 FFTs, the two-phase LDS exchange and the
// real-FFT split + power step.  Everything here is __host__ __device__ so that the lane decomposition can
// be executed lane by lane on the CPU (host_sim.cc) against a naive DFT before it runs on a GPU.
//
// Decomposition of one frame of NFFT = 2*NC real samples, NC = R1*R2 complex points z[n] = x[2n] + i x[2n+1]:
//   n = R2*n1 + n2, k = k1 + R1*k2
//   stage 1: lane owning n2 runs an R1-point FFT over n1, multiplies by W_NC^(n2*k1)
//   exchange through LDS in two phases (k1 < R1/2, k1 >= R1/2)
//   stage 2: lane l (LPF = R1/2 lanes per frame) runs two R2-point FFTs over n2, for k1 = l and k1 = R1-l
//            (lane 0: k1 = 0 and k1 = R1/2), so Z[k] and Z[NC-k] always sit in the same lane
//   split:   X[k] = E + W_NFFT^k O, X[NC-k] = conj(E - W_NFFT^k O) with E = Z[k] + conj Z[NC-k],
//            O = -i (Z[k] - conj Z[NC-k]); the factor 1/2 is folded into the window table.
#pragma once

#include <type_traits>

#if !defined(__clang__)
#error "fft_core.h uses clang vector extensions: build with hipcc, or ROCm's clang++ for host-only programs"
#endif

#if defined(__HIPCC__)
#define SED_HD __host__ __device__ __forceinline__
#else
#define SED_HD inline
#endif

// Keeps the compiler's instruction scheduler from moving anything across this point (device code only).  Used to pin
// batches of LDS reads ahead of the arithmetic that hides their latency: left alone, the scheduler sinks every table
// read to just before its use and then waits for it, one LDS round trip per value.
#if defined(__HIP_DEVICE_COMPILE__)
#define SED_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define SED_SCHED_FENCE() ((void)0)
#endif

// Compile-time ablations for timing experiments (results are then garbage): -DSED_X_MASK=<bits>.
#if defined(SED_X_MASK)
#define SED_SKIP(bit) (((SED_X_MASK) & (bit)) != 0)
#else
#define SED_SKIP(bit) false
#endif

// Issue priority of a wavefront per phase of the streaming loop (s_setprio), one hex digit 0..3 per phase:
// 0 loop top and span order, 1 stage-1 loads, 2 stage-1 transform, 3 exchange, 4 stage 2 and power, 5 mel on the matrix
// pipe, 6 gather and log.  With equal priorities the wavefronts of a SIMD advance through their arithmetic in step and
// reach their LDS-only phases together; a wavefront inside the exchange (LDS traffic only) yielding to the others
// measured 2 % less kernel time on every variant tried (0x3330333, 0x1130311, 0x2220222: 158-159 against 162-164 us).
#ifndef SED_PRIOS
#define SED_PRIOS 0x3330333
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define SED_P(ph) __builtin_amdgcn_s_setprio(((SED_PRIOS) >> (4 * (ph))) & 3)
#else
#define SED_P(ph) ((void)0)
#endif

namespace sed {

struct alignas(8) cplx {
  float x, y;
};

// A complex value (re, im) in one aligned register pair.  gfx950 executes add/mul/fma on such pairs in one
// instruction (v_pk_*_f32) and can swap or broadcast the halves of each operand for free, so every step below
// is written on pairs; on the host the same expressions evaluate element by element with the same roundings.
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));

// 8- and 16-byte accesses: the pointer handed to ld2/ld4/st4 must be aligned to the access size.
SED_HD v2 ld2(const float* p) { return *reinterpret_cast<const v2*>(p); }
SED_HD v2 ld2(const cplx* p) { return *reinterpret_cast<const v2*>(p); }
SED_HD v4 ld4(const float* p) { return *reinterpret_cast<const v4*>(p); }
SED_HD void st4(float* p, v4 v) { *reinterpret_cast<v4*>(p) = v; }

// Square root of a power value for the magnitude spectra.  On the device: the v_sqrt_f32 instruction itself (1 ulp;
// inputs below 2^-126 count as zero), one instruction where the IEEE-exact sqrtf expands to about ten.  The magnitudes
// only feed a mel sum and a logarithm with an additive offset (1e-3 or 1e-9), or a spectrogram compared to 2e-6.
SED_HD float fast_sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_sqrtf(x);
#else
  return std::sqrt(x);
#endif
}

SED_HD v2 pk_fma(v2 a, v2 b, v2 c) { return __builtin_elementwise_fma(a, b, c); }
SED_HD v2 swp(v2 a) { return __builtin_shufflevector(a, a, 1, 0); }
SED_HD v2 lo2(v2 a) { return __builtin_shufflevector(a, a, 0, 0); }
SED_HD v2 hi2(v2 a) { return __builtin_shufflevector(a, a, 1, 1); }
// By-value select: a conditional between two lvalues would select between their ADDRESSES, which keeps
// register arrays in scratch memory.
SED_HD v2 select(bool c, v2 a, v2 b) { return c ? a : b; }

// Complex product y * w: (yx wx - yy wy, yx wy + yy wx) as one packed multiply and one packed fma whose second
// operand is w with its halves swapped and the low half negated (operand modifiers of the instruction).
SED_HD v2 cmul(v2 y, v2 w) {
  const v2 t = lo2(y) * w;
#if defined(__HIP_DEVICE_COMPILE__)
  v2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(y), "v"(w), "v"(t));
  return d;
#else
  return v2{__builtin_fmaf(-y.y, w.y, t.x), __builtin_fmaf(y.y, w.x, t.y)};
#endif
}

template <int I, int N, class F>
SED_HD void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n >> 1); }
constexpr int bitrev(int v, int bits) {
  int r = 0;
  for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1) << (bits - 1 - i);
  return r;
}

// W_32^k = cos(2 pi k/32) - i sin(2 pi k/32), k = 0..15
constexpr float kCos32[16] = {1.0f, 0.9807852804032304f, 0.9238795325112867f, 0.8314696123025452f,
                              0.7071067811865476f, 0.5555702330196022f, 0.3826834323650898f, 0.19509032201612828f,
                              0.0f, -0.19509032201612828f, -0.3826834323650898f, -0.5555702330196022f,
                              -0.7071067811865476f, -0.8314696123025452f, -0.9238795325112867f, -0.9807852804032304f};
constexpr float kSin32[16] = {0.0f, 0.19509032201612828f, 0.3826834323650898f, 0.5555702330196022f,
                              0.7071067811865476f, 0.8314696123025452f, 0.9238795325112867f, 0.9807852804032304f,
                              1.0f, 0.9807852804032304f, 0.9238795325112867f, 0.8314696123025452f,
                              0.7071067811865476f, 0.5555702330196022f, 0.3826834323650898f, 0.19509032201612828f};

// One decimation-in-time butterfly (a, b) -> (a + w b, a - w b) with w = W_32^TW = c - i s.
// General case: a + w b = (a + c b) + (s, -s) * swap(b) in two packed fmas, a - w b = 2a - (a + w b) in a third.
template <int TW>
SED_HD void butterfly(v2& a, v2& b) {
  if constexpr (TW == 0) {
    const v2 t = b;
    b = a - t;
    a = a + t;
  } else if constexpr (TW == 8) {  // w = -i : w b = (bi, -br)
    const v2 t = swp(b), k = v2{1.0f, -1.0f};
    b = pk_fma(t, -k, a);
    a = pk_fma(t, k, a);
  } else {
    constexpr float c = kCos32[TW], s = kSin32[TW];
    const v2 n = pk_fma(swp(b), v2{s, -s}, pk_fma(b, v2{c, c}, a));
    b = pk_fma(a, v2{2.0f, 2.0f}, -n);
    a = n;
  }
}

// In-register N-point FFT (N = 2..32, power of two).  Input in bit-reversed order, output in natural order.
template <int N>
SED_HD void fft_dit(v2 (&x)[N]) {
  constexpr int LOG = ilog2(N);
  if (SED_SKIP(N == 32 ? 256 : 512)) return;
  static_for<0, LOG>([&](auto S) {
    constexpr int h = 1 << decltype(S)::value;  // half span of this stage
    static_for<0, N / 2>([&](auto Bf) {
      constexpr int b = decltype(Bf)::value;
      constexpr int g = b / h, j = b % h;
      constexpr int i0 = g * 2 * h + j, i1 = i0 + h;
      constexpr int tw = j * (16 / h);  // W_{2h}^j = W_32^{j*16/h}
      butterfly<tw>(x[i0], x[i1]);
    });
  });
}

template <int R1_, int R2_, int LPF_, int WAVES_>
struct Geo {
  static constexpr int R1 = R1_, R2 = R2_, LPF = LPF_, WAVES = WAVES_;
  static constexpr int NC = R1 * R2, NFFT = 2 * NC, NBINS = NC + 1;
  static constexpr int A = R2 / LPF;        // stage-1 FFTs per lane
  static constexpr int FPW = 64 / LPF;      // frames per wave
  static constexpr int FPI = WAVES * FPW;   // frames per workgroup iteration
  // Exchange rows: pitch RP and floats per frame XFR.  The LDS serves an 8-byte access 32 lanes at a time, i.e. 32 / LPF
  // frames together; the pads are picked (by enumerating every write and read instruction of the exchange) so that
  // those lanes fall on distinct banks: no access of any of the three geometries conflicts.  (With pitch R1 + 4 and a
  // 32-float frame pad the writes of two frames met on the same banks: 64 % of the kernel's bank-conflict cycles.)
  // A lane with two stage-1 FFTs (n2 = 2l and 2l + 1) keeps them LPF rows apart, see xrow(): with its two rows
  // adjacent, the lanes of a frame would be 2 * RP floats apart, a multiple of 4, and either the writes or the reads
  // of the frames served together would share banks whatever the pitch.
  static constexpr int RP = R1 + (LPF == 16 ? (R2 == 16 ? 6 : 2) : 2);
  static constexpr int XFR = R2 * RP + (LPF == 16 ? (R2 == 16 ? 0 : 32) : 16);
  static_assert(RP % 2 == 0 && XFR % 2 == 0 && RP >= R1, "8-byte accesses, R1/2 complex values per row and phase");
  // row of the frame's exchange region that holds the stage-1 results of n2
  static constexpr int xrow(int n2) { return A == 1 ? n2 : n2 / A + LPF * (n2 % A); }
  static constexpr int PP = NC + 16;        // power-row pitch in floats (16 mod 64: the 4 frames of a wave tile the banks)
  static constexpr int SLOTS = R2 + 1;      // split slots per lane (one of them stores from lane 0 only)
  // Lane-major table layouts: the values one lane reads are contiguous, so that it fetches two of them per 16-byte
  // LDS read; row pitches are 4 * odd floats, which spreads the 16 lanes of a frame over all banks.
  static constexpr int WROW = 2 * R1 + 4;                       // window row of lane l: (w[2(R2 n1 + l)], w[.. + 1]) for n1 = 0..R1-1
  static constexpr int WIN_FLOATS = A == 1 ? LPF * WROW : NFFT; // (natural order when a lane already reads 16 bytes per n1)
  static constexpr int TROW = 2 * R1 + 4;                       // inter-stage twiddle row of n2: W_NC^(n2 k1), k1 = 0..R1-1
  static constexpr int TW1_FLOATS = A == 1 ? R2 * TROW : 2 * R1 * R2;
  static constexpr int SROW = (2 * (SLOTS + 1) + 3) / 4 * 4;    // split twiddle row of lane l: slots 0..SLOTS-1 (+ padding)
  static constexpr int TW2_FLOATS = LPF * SROW;
  static_assert((WROW / 4) % 2 == 1 && (SROW / 4) % 2 == 1, "row pitches must be 4 * odd floats");
  static_assert(R1 == 2 * LPF, "two stage-2 FFTs per lane");
  static_assert(A == 1 || A == 2, "one or two stage-1 FFTs per lane");
  static_assert(R1 <= 32 && R2 <= 32, "twiddle literals cover N <= 32");
};

// Position (in floats) of natural-order table entries in the lane-major LDS tables.
template <class G>
SED_HD int win_slot(int i) {  // window sample i
  if constexpr (G::A != 1) return i;
  const int n1 = i / (2 * G::R2), rem = i % (2 * G::R2);
  return (rem >> 1) * G::WROW + 2 * n1 + (rem & 1);
}
template <class G>
SED_HD int tw1_slot(int k1, int n2) {  // W_NC^(n2 k1)
  if constexpr (G::A != 1) return 2 * (k1 * G::R2 + n2);
  return n2 * G::TROW + 2 * k1;
}
template <class G>
SED_HD int tw2_slot(int t, int l) {  // split twiddle of slot t in lane l
  return l * G::SROW + 2 * t;
}

template <class G>
struct LaneState {
  v2 y[G::A][G::R1];  // stage-1 data, natural k1 order after the FFT
  v2 p[G::R2];        // stage-2 FFT for k1 = l        (lane 0: k1 = 0)
  v2 q[G::R2];        // stage-2 FFT for k1 = R1 - l   (lane 0: k1 = R1/2)
};

#if defined(__HIP_DEVICE_COMPILE__)
SED_HD unsigned lds_addr(const float* p) { return (unsigned)(size_t)p; }  // low half of a generic LDS address = LDS offset
#endif

// Sample sources for stage 1: fp32 PCM (LDS or global memory) and int16 PCM (divided by `div`, as the
// int16 -> fp32 conversion of the reference does).  Offsets are in samples from the frame's first sample and are
// multiples of the access width.
struct SrcF32 {
  const float* p;
  SED_HD v2 load2(int o) const { return ld2(p + o); }
  SED_HD v4 load4(int o) const { return ld4(p + o); }
};
// fp32 samples from a frame that starts on any sample (hops that are not a multiple of 4): 4-byte reads
struct SrcF32U {
  const float* p;
  SED_HD v2 load2(int o) const { return v2{p[o], p[o + 1]}; }
  SED_HD v4 load4(int o) const { return v4{p[o], p[o + 1], p[o + 2], p[o + 3]}; }
};
struct alignas(4) i16x2 {
  short x, y;
};
struct alignas(8) i16x4 {
  short x, y, z, w;
};
struct SrcI16 {
  const short* p;
  float div;
  SED_HD v2 load2(int o) const {
    const i16x2 v = *reinterpret_cast<const i16x2*>(p + o);
    return v2{(float)v.x / div, (float)v.y / div};
  }
  SED_HD v4 load4(int o) const {
    const i16x4 v = *reinterpret_cast<const i16x4*>(p + o);
    return v4{(float)v.x / div, (float)v.y / div, (float)v.z / div, (float)v.w / div};
  }
};

// Stage 1 for lane l of a frame: load + window, R1-point FFT(s), inter-stage twiddle.
// src: the frame's NFFT samples; win: window (already holding the factor 1/2) at win_slot(); tw1: W_NC^(n2*k1) at
// tw1_slot().
// Issue order: all samples and the first half of the window, then (as the products of the first half are formed)
// the second half of the window and the twiddles, so that every table value is in flight long before it is used.
// Sample fetch of stage 1 from a span staged with a skew: the samples of hop h (HOP floats each) start at float
// h * (HOP + SKEW) of the span, so that the two frames whose lanes the LDS serves together start 32 banks apart when the
// hop itself is a multiple of 64 floats.  HOP is a multiple of 2 * R2, so a lane's pair never straddles two hops and
// every offset is an immediate.  Hand-written 8-byte reads (the compiler would fuse pairs of them into the slower
// ds_read2_b64); it does not count them, so the first use of a sample must come after a wait for a younger LDS read
// (lane_stage1 issues the window reads after them and multiplies every sample by a window value first).
template <class G, int HOP, int SKEW>
SED_HD void lane_load_samples_skewed(LaneState<G>& st, const float* frame, int l) {
  static_assert(G::A == 1 && HOP % (2 * G::R2) == 0, "one stage-1 FFT per lane, hops of whole lane rows");
  constexpr int LOG1 = ilog2(G::R1);
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned fa = lds_addr(frame + 2 * l);
#endif
  static_for<0, G::R1>([&](auto N1) {
    constexpr int n1 = decltype(N1)::value;
    constexpr int off = 2 * G::R2 * n1 + SKEW * ((2 * G::R2 * n1) / HOP);
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned addr = fa;
    v2 val;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(val) : "v"(addr), "n"(4 * off) : "memory");
    st.y[0][bitrev(n1, LOG1)] = val;
#else
    st.y[0][bitrev(n1, LOG1)] = ld2(frame + 2 * l + off);
#endif
  });
}

// LOADED: st.y[0] already holds the samples in bit-reversed order (lane_load_samples_skewed), src is not read.
template <class G, class Src, bool LOADED = false>
SED_HD void lane_stage1(LaneState<G>& st, const Src src, const float* win, const float* tw1, int l) {
  constexpr int LOG1 = ilog2(G::R1), H = G::R1 / 2;
  if constexpr (G::A == 1) {
    constexpr int Q = G::R1 / 4;  // window values per quarter, fetched two at a time
    const float* wrow = win + l * G::WROW;
    const float* trow = tw1 + l * G::TROW;
    v4 w[4][Q / 2], ta[H / 2], tb[H / 2];
    auto issue_window = [&](auto Qf) {
      constexpr int q = decltype(Qf)::value;
      static_for<0, Q / 2>([&](auto Pf) { w[q][decltype(Pf)::value] = SED_SKIP(2) ? v4{0.5f, 0.25f, 0.125f, (float)l} : ld4(wrow + 2 * Q * q + 4 * decltype(Pf)::value); });
    };
    auto apply_window = [&](auto Qf) {
      constexpr int q = decltype(Qf)::value;
      static_for<0, Q / 2>([&](auto Pf) {
        constexpr int i = decltype(Pf)::value, n1 = Q * q + 2 * i;
        st.y[0][bitrev(n1, LOG1)] *= w[q][i].xy;
        st.y[0][bitrev(n1 + 1, LOG1)] *= w[q][i].zw;
      });
    };
    if constexpr (!LOADED) {
      static_for<0, G::R1>([&](auto N1) {
        constexpr int n1 = decltype(N1)::value;
        st.y[0][bitrev(n1, LOG1)] = SED_SKIP(4) ? v2{(float)(l + n1), 1.0f} : src.load2(2 * G::R2 * n1 + 2 * l);
      });
    }
    issue_window(std::integral_constant<int, 0>{});
    issue_window(std::integral_constant<int, 1>{});
    SED_SCHED_FENCE();
    apply_window(std::integral_constant<int, 0>{});
    issue_window(std::integral_constant<int, 2>{});
    SED_SCHED_FENCE();
    apply_window(std::integral_constant<int, 1>{});
    issue_window(std::integral_constant<int, 3>{});
    SED_SCHED_FENCE();
    apply_window(std::integral_constant<int, 2>{});
    static_for<0, H / 2>([&](auto Pf) { ta[decltype(Pf)::value] = SED_SKIP(2) ? v4{0.5f, 0.25f, 0.125f, (float)l} : ld4(trow + 4 * decltype(Pf)::value); });
    SED_SCHED_FENCE();
    apply_window(std::integral_constant<int, 3>{});
    SED_SCHED_FENCE();
    SED_P(2);
    fft_dit<G::R1>(st.y[0]);
    SED_SCHED_FENCE();
    static_for<0, H / 2>([&](auto Pf) { tb[decltype(Pf)::value] = SED_SKIP(2) ? v4{0.5f, 0.25f, 0.125f, (float)l} : ld4(trow + 2 * H + 4 * decltype(Pf)::value); });
    static_for<0, H / 2>([&](auto Pf) {
      constexpr int i = decltype(Pf)::value;
      if constexpr (i > 0) st.y[0][2 * i] = cmul(st.y[0][2 * i], ta[i].xy);  // k1 = 0: W^0
      st.y[0][2 * i + 1] = cmul(st.y[0][2 * i + 1], ta[i].zw);
    });
    SED_SCHED_FENCE();
    static_for<0, H / 2>([&](auto Pf) {
      constexpr int i = decltype(Pf)::value;
      st.y[0][H + 2 * i] = cmul(st.y[0][H + 2 * i], tb[i].xy);
      st.y[0][H + 2 * i + 1] = cmul(st.y[0][H + 2 * i + 1], tb[i].zw);
    });
  } else {
    static_for<0, G::R1>([&](auto N1) {
      constexpr int n1 = decltype(N1)::value;
      constexpr int pos = bitrev(n1, LOG1);
      const v4 pw = src.load4(2 * G::R2 * n1 + 4 * l) * ld4(win + 2 * G::R2 * n1 + 4 * l);
      st.y[0][pos] = pw.xy;
      st.y[1][pos] = pw.zw;
    });
    fft_dit<G::R1>(st.y[0]);
    fft_dit<G::R1>(st.y[1]);
    static_for<1, G::R1>([&](auto K1) {  // n2 = 2l and 2l + 1 are adjacent in the table
      constexpr int k1 = decltype(K1)::value;
      const v4 w = ld4(tw1 + 2 * (k1 * G::R2 + 2 * l));
      st.y[0][k1] = cmul(st.y[0][k1], w.xy);
      st.y[1][k1] = cmul(st.y[1][k1], w.zw);
    });
  }
}

// Kaldi framing (torchaudio.compliance.kaldi._get_window): sum of the lane's samples among the first win_length of
// the frame (the mean over win_length samples is removed before pre-emphasis and windowing).
// VEC: the frame starts on a multiple of 2 * A floats, so a lane's 2 * A consecutive samples come in one 8- or 16-byte
// read (same values, same summation order).  Besides the instruction count this removes the bank conflicts of the
// 4-byte reads, whose 64 lanes are served together: with 8 frames of 8 lanes in a wavefront and a hop of 160 floats,
// every other frame sits on the same banks.
template <class G, bool VEC = false>
SED_HD float lane_frame_sum(const float* frame, int l, int win_length) {
  float sum = 0.0f;
  static_for<0, G::R1>([&](auto N1) {
    constexpr int n1 = decltype(N1)::value;
    const int base = 2 * G::R2 * n1 + 2 * G::A * l;
    float x[2 * G::A];
    if constexpr (VEC && G::A == 2) {
      const v4 v = ld4(frame + base);
      x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    } else if constexpr (VEC) {
      const v2 v = ld2(frame + base);
      x[0] = v.x; x[1] = v.y;
    } else {
      static_for<0, 2 * G::A>([&](auto Ef) { x[decltype(Ef)::value] = frame[base + decltype(Ef)::value]; });
    }
    // rows of 2 * R2 samples wholly inside the window need no test, rows wholly outside add nothing (adding +0 to a
    // sum that started from +0 leaves its bits alone): only the row the window ends in is predicated, and the tests on
    // the row are wave-uniform
    if (2 * G::R2 * (n1 + 1) <= win_length) {
      static_for<0, 2 * G::A>([&](auto Ef) { sum += x[decltype(Ef)::value]; });
    } else if (2 * G::R2 * n1 < win_length) {
      static_for<0, 2 * G::A>([&](auto Ef) {
        constexpr int e = decltype(Ef)::value;
        sum += (base + e < win_length) ? x[e] : 0.0f;
      });
    }
  });
  return sum;
}

// Stage 1 with per-frame conditioning: y[n] = (x[n] - mean) - c (x[n-1] - mean), x[-1] := x[0], then window, FFT(s)
// and inter-stage twiddle as in lane_stage1.  frame: the frame's samples in natural order (frame[-1] must be
// readable; its value is not used).
template <class G, bool VEC = false>
SED_HD void lane_stage1_conditioned(LaneState<G>& st, const float* frame, const float* win, const float* tw1, int l,
                                    float mean, float c) {
  constexpr int LOG1 = ilog2(G::R1);
  static_for<0, G::R1>([&](auto N1) {
    constexpr int n1 = decltype(N1)::value;
    constexpr int pos = bitrev(n1, LOG1);
    const int base = 2 * G::R2 * n1 + 2 * G::A * l;
    float x[2 * G::A + 1];
    x[0] = frame[base == 0 ? 0 : base - 1];
    if constexpr (VEC && G::A == 2) {
      const v4 v = ld4(frame + base);
      x[1] = v.x; x[2] = v.y; x[3] = v.z; x[4] = v.w;
    } else if constexpr (VEC) {
      const v2 v = ld2(frame + base);
      x[1] = v.x; x[2] = v.y;
    } else {
      static_for<0, 2 * G::A>([&](auto Ef) { x[decltype(Ef)::value + 1] = frame[base + decltype(Ef)::value]; });
    }
    float y[2 * G::A];
    static_for<0, 2 * G::A>([&](auto Ef) {
      constexpr int e = decltype(Ef)::value;
      y[e] = (x[e + 1] - mean) - c * (x[e] - mean);
    });
    if constexpr (G::A == 1) {
      st.y[0][pos] = v2{y[0], y[1]} * ld2(win + l * G::WROW + 2 * n1);  // lane-major window table
    } else {
      const v4 w = ld4(win + 2 * G::R2 * n1 + 4 * l);
      st.y[0][pos] = v2{y[0], y[1]} * w.xy;
      st.y[1][pos] = v2{y[2], y[3]} * w.zw;
    }
  });
  if constexpr (G::A == 1) {
    fft_dit<G::R1>(st.y[0]);
    const float* trow = tw1 + l * G::TROW;
    static_for<1, G::R1>([&](auto K1) {
      constexpr int k1 = decltype(K1)::value;
      st.y[0][k1] = cmul(st.y[0][k1], ld2(trow + 2 * k1));
    });
  } else {
    fft_dit<G::R1>(st.y[0]);
    fft_dit<G::R1>(st.y[1]);
    static_for<1, G::R1>([&](auto K1) {
      constexpr int k1 = decltype(K1)::value;
      const v4 w = ld4(tw1 + 2 * (k1 * G::R2 + 2 * l));
      st.y[0][k1] = cmul(st.y[0][k1], w.xy);
      st.y[1][k1] = cmul(st.y[1][k1], w.zw);
    });
  }
}

// Exchange phase PH (0: k1 in [0, R1/2), 1: k1 in [R1/2, R1)): lane l writes row n2 of the frame's region.
// On the device the accesses are written as 8-byte LDS instructions by hand: measured on gfx950 per instruction and
// CU, ds_read_b64 0.95 ns against 3.4 ns for the ds_read2_b64 the compiler would fuse two of them into, and
// ds_write_b64 2.6 ns against 5.8 ns for a ds_write_b128.  The compiler does not count hand-written LDS reads, so
// lane_xwait() must come between the last read and the first use of st.p / st.q.
template <class G, int PH>
SED_HD void lane_xwrite(const LaneState<G>& st, float* xfr, int l) {
  static_for<0, G::A>([&](auto Af) {
    constexpr int a = decltype(Af)::value;
    float* row = xfr + (l + G::LPF * a) * G::RP;  // = xrow(A * l + a)
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned ra = lds_addr(row);
    static_for<0, G::R1 / 2>([&](auto Q) {
      constexpr int q = decltype(Q)::value;
      const unsigned addr = ra;  // (asm operands must be locals of the innermost lambda)
      const v2 val = st.y[a][PH * (G::R1 / 2) + q];
      asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(8 * q) : "memory");
    });
#else
    static_for<0, G::R1 / 2>([&](auto Q) {
      constexpr int q = decltype(Q)::value;
      *reinterpret_cast<v2*>(row + 2 * q) = st.y[a][PH * (G::R1 / 2) + q];
    });
#endif
  });
}

// Column of the exchange rows that lane l consumes in phase PH.
template <class G, int PH>
SED_HD int lane_xcol(int l) {
  if constexpr (PH == 0) return l;
  return l == 0 ? 0 : G::R1 / 2 - l;
}

template <class G, int PH>
SED_HD void lane_xread(LaneState<G>& st, const float* xfr, int l) {
  constexpr int LOG2 = ilog2(G::R2);
  const float* col = xfr + 2 * lane_xcol<G, PH>(l);
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned ca = lds_addr(col);
#endif
  static_for<0, G::R2>([&](auto N2) {
    constexpr int n2 = decltype(N2)::value;
    constexpr int pos = bitrev(n2, LOG2);
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned addr = ca;
    v2 val;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(val) : "v"(addr), "n"(4 * G::xrow(n2) * G::RP) : "memory");
    if constexpr (PH == 0) st.p[pos] = val;
    else st.q[pos] = val;
#else
    if constexpr (PH == 0) st.p[pos] = ld2(col + G::xrow(n2) * G::RP);
    else st.q[pos] = ld2(col + G::xrow(n2) * G::RP);
#endif
  });
}

// Completion of the hand-written exchange reads: every value is tied to the wait, so no use can move above it.
template <class G>
SED_HD void lane_xwait(LaneState<G>& st) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(G::R2 == 16 || G::R2 == 32, "operand lists below");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  static_for<0, G::R2>([&](auto Bf) {
    constexpr int b = decltype(Bf)::value;
    v2 x = st.p[b], y = st.q[b];
    asm volatile("" : "+v"(x), "+v"(y));
    st.p[b] = x;
    st.q[b] = y;
  });
#endif
}

// Split slots.  After stage 2 a lane l > 0 holds P[s] = Z[l + R1 s] and Q[s] = Z[(R1 - l) + R1 s]; bin k = l + R1 s
// pairs with NC - k = Q[R2-1-s].  Lane 0 holds P[s] = Z[R1 s] and Q[s] = Z[R1/2 + R1 s], whose partners are in the
// same array: P[s] with P[R2-s], Q[s] with Q[R2-1-s].  Slot t of SLOTS = R2 + 1:
//   t <= R2/2 ("P slot", s = t)    : U = P[s], V = Q[R2-1-s]            (lane 0: V = P[(R2-s) % R2]),  k = l + R1 s
//   t >  R2/2 ("Q slot", s = t - 1): U = P[s] (lane 0: Q[s]), V = Q[R2-1-s],  k = l + R1 s  (lane 0: + R1/2)
// so only one operand per slot depends on the lane, every store address is a per-lane base plus a constant, and
// lane 0 alone stores in slot t = R2/2 + 1 (for the other lanes it repeats slot R2/2).
template <class G>
SED_HD int lane_slot_bin(int l, int t) {
  const int s = t <= G::R2 / 2 ? t : t - 1;
  return l + G::R1 * s + ((t > G::R2 / 2 && l == 0) ? G::R1 / 2 : 0);
}

// Stage-2 FFTs, real-FFT split and |X|^2 (or |X|) into the frame's power row pw[0..NC].
// tw2: W_NFFT^(lane_slot_bin(l, t)) at tw2_slot(t, l).
// With E = U + conj V, O = -i (U - conj V), T = W O:  X[k] = E + T,  conj X[NC-k] = E - T.
template <class G, bool MAGNITUDE>
SED_HD void lane_stage2_power(LaneState<G>& st, float* pw, const float* tw2, int l) {
  // the split twiddles of all slots are in flight while the transforms run
  const float* srow = tw2 + l * G::SROW;
  v4 wtab[(G::SLOTS + 1) / 2];
  static_for<0, (G::SLOTS + 1) / 2>([&](auto Tf) {
    constexpr int t2 = decltype(Tf)::value;
    wtab[t2] = SED_SKIP(2) ? v4{0.5f, 0.25f, 0.125f, (float)l} : ld4(srow + 4 * t2);
  });
  SED_SCHED_FENCE();
  fft_dit<G::R2>(st.p);
  fft_dit<G::R2>(st.q);
  SED_SCHED_FENCE();
  const bool l0 = (l == 0);
  const int off = l0 ? G::R1 / 2 : 0;
  float* lo_p = pw + l;                 // pw[k] of P slots at lo_p[R1 s]
  float* lo_q = lo_p + off;             // ... of Q slots
  float* hi_p = pw + G::R1 - l;         // pw[NC - k] of P slots at hi_p[R1 (R2-1-s)]
  float* hi_q = hi_p - off;
  const v2 k = v2{1.0f, -1.0f};
  static_for<0, G::SLOTS>([&](auto Tf) {
    constexpr int t = decltype(Tf)::value;
    const v4 wpair = wtab[t / 2];  // twiddles of slots t and t + 1
    constexpr bool qslot = t > G::R2 / 2;
    constexpr int s = qslot ? t - 1 : t;
    v2 u, v;
    if constexpr (!qslot) {
      u = st.p[s];
      v = select(l0, st.p[(G::R2 - s) % G::R2], st.q[G::R2 - 1 - s]);
    } else {
      u = select(l0, st.q[s], st.p[s]);
      v = st.q[G::R2 - 1 - s];
    }
    const v2 e = pk_fma(v, k, u);             // (ur + vr, ui - vi)
    const v2 o = pk_fma(swp(u), k, swp(v));   // (ui + vi, vr - ur)
    const v2 tt = cmul(o, t % 2 == 0 ? wpair.xy : wpair.zw);
    const v2 re = pk_fma(lo2(tt), k, lo2(e)); // (Re X[k], Re X[NC-k])
    const v2 im = pk_fma(hi2(tt), k, hi2(e)); // (Im X[k], -Im X[NC-k])
    v2 pp = pk_fma(im, im, re * re);
    if constexpr (MAGNITUDE) pp = v2{fast_sqrt(pp.x), fast_sqrt(pp.y)};
    float* lo = qslot ? lo_q : lo_p;
    float* hi = qslot ? hi_q : hi_p;
    if ((t != G::R2 / 2 + 1 || l0) && !(SED_SKIP(16) && pp.x != 12345.0f)) {
      lo[G::R1 * s] = pp.x;
      hi[G::R1 * (G::R2 - 1 - s)] = pp.y;
    }
  });
}

}  // namespace sed
