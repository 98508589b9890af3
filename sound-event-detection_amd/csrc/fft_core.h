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

#if defined(__HIPCC__)
#define SED_HD __host__ __device__ __forceinline__
#else
#define SED_HD inline
#endif

namespace sed {

struct alignas(8) cplx {
  float x, y;
};

// 8- and 16-byte accesses: the pointer handed to ld2/ld4/st4 must be aligned to the access size.
struct alignas(8) f2 {
  float x, y;
};
struct alignas(16) f4 {
  float x, y, z, w;
};
SED_HD f2 ld2(const float* p) { return *reinterpret_cast<const f2*>(p); }
SED_HD f4 ld4(const float* p) { return *reinterpret_cast<const f4*>(p); }
SED_HD void st4(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }

template <int I, int N, class F>
SED_HD void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// By-value select: a conditional between two lvalues would select between their ADDRESSES, which keeps
// register arrays in scratch memory.
SED_HD float select(bool c, float a, float b) { return c ? a : b; }

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

// One decimation-in-time butterfly (a, b) -> (a + w b, a - w b) with w = W_32^TW.
// The general case forms a + w b with four fused multiply-adds and a - w b as 2a - (a + w b).
template <int TW>
SED_HD void butterfly(float& ar, float& ai, float& br, float& bi) {
  if constexpr (TW == 0) {
    const float tr = br, ti = bi;
    br = ar - tr; bi = ai - ti;
    ar = ar + tr; ai = ai + ti;
  } else if constexpr (TW == 8) {  // w = -i : w b = (bi, -br)
    const float tr = bi, ti = br;
    br = ar - tr; bi = ai + ti;
    ar = ar + tr; ai = ai - ti;
  } else if constexpr (TW == 4) {  // w = (1 - i)/sqrt2 : w b = s (br + bi, bi - br)
    constexpr float s = 0.7071067811865476f;
    const float p = br + bi, q = bi - br;
    br = __builtin_fmaf(-s, p, ar); bi = __builtin_fmaf(-s, q, ai);
    ar = __builtin_fmaf(s, p, ar);  ai = __builtin_fmaf(s, q, ai);
  } else if constexpr (TW == 12) {  // w = (-1 - i)/sqrt2 : w b = s (bi - br, -(br + bi))
    constexpr float s = 0.7071067811865476f;
    const float p = bi - br, q = br + bi;
    br = __builtin_fmaf(-s, p, ar); bi = __builtin_fmaf(s, q, ai);
    ar = __builtin_fmaf(s, p, ar);  ai = __builtin_fmaf(-s, q, ai);
  } else {
    constexpr float c = kCos32[TW], s = kSin32[TW];  // w = c - i s : w b = (c br + s bi, c bi - s br)
    const float nr = __builtin_fmaf(s, bi, __builtin_fmaf(c, br, ar));
    const float ni = __builtin_fmaf(-s, br, __builtin_fmaf(c, bi, ai));
    br = __builtin_fmaf(2.0f, ar, -nr);
    bi = __builtin_fmaf(2.0f, ai, -ni);
    ar = nr; ai = ni;
  }
}

// In-register N-point FFT (N = 2..32, power of two).  Input in bit-reversed order, output in natural order.
template <int N>
SED_HD void fft_dit(float (&xr)[N], float (&xi)[N]) {
  constexpr int LOG = ilog2(N);
  static_for<0, LOG>([&](auto S) {
    constexpr int h = 1 << decltype(S)::value;  // half span of this stage
    static_for<0, N / 2>([&](auto Bf) {
      constexpr int b = decltype(Bf)::value;
      constexpr int g = b / h, j = b % h;
      constexpr int i0 = g * 2 * h + j, i1 = i0 + h;
      constexpr int tw = j * (16 / h);  // W_{2h}^j = W_32^{j*16/h}
      butterfly<tw>(xr[i0], xi[i0], xr[i1], xi[i1]);
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
  static constexpr int RP = R1 + 4;         // exchange row pitch in floats (R1/2 complex + pad)
  static constexpr int XFR = R2 * RP + 32;  // exchange floats per frame
  static constexpr int PP = NC + 16;        // power-row pitch in floats (16 mod 64: the 4 frames of a wave tile the banks)
  static constexpr int SLOTS = R2 + 1;      // split slots per lane (the last one is used by lane 0 only)
  static_assert(R1 == 2 * LPF, "two stage-2 FFTs per lane");
  static_assert(A == 1 || A == 2, "one or two stage-1 FFTs per lane");
  static_assert(R1 <= 32 && R2 <= 32, "twiddle literals cover N <= 32");
};

template <class G>
struct LaneState {
  float yr[G::A][G::R1], yi[G::A][G::R1];  // stage-1 data, natural k1 order after the FFT
  float pr[G::R2], pi[G::R2];              // stage-2 FFT for k1 = l        (lane 0: k1 = 0)
  float qr[G::R2], qi[G::R2];              // stage-2 FFT for k1 = R1 - l   (lane 0: k1 = R1/2)
};

// Sample sources for stage 1: fp32 PCM (LDS or global memory) and int16 PCM (divided by `div`, as the
// int16 -> fp32 conversion of the reference does).  Offsets are in samples from the frame's first sample and are
// multiples of the access width.
struct SrcF32 {
  const float* p;
  SED_HD f2 load2(int o) const { return ld2(p + o); }
  SED_HD f4 load4(int o) const { return ld4(p + o); }
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
  SED_HD f2 load2(int o) const {
    const i16x2 v = *reinterpret_cast<const i16x2*>(p + o);
    return f2{(float)v.x / div, (float)v.y / div};
  }
  SED_HD f4 load4(int o) const {
    const i16x4 v = *reinterpret_cast<const i16x4*>(p + o);
    return f4{(float)v.x / div, (float)v.y / div, (float)v.z / div, (float)v.w / div};
  }
};

// Stage 1 for lane l of a frame: load + window, R1-point FFT(s), inter-stage twiddle.
// src: the frame's NFFT samples; win: NFFT window values (already holding the factor 1/2);
// tw1[k1*R2 + n2] = W_NC^(n2*k1).
template <class G, class Src>
SED_HD void lane_stage1(LaneState<G>& st, const Src src, const float* win, const cplx* tw1, int l) {
  constexpr int LOG1 = ilog2(G::R1);
  static_for<0, G::R1>([&](auto N1) {
    constexpr int n1 = decltype(N1)::value;
    constexpr int pos = bitrev(n1, LOG1);
    if constexpr (G::A == 1) {
      const f2 p = src.load2(2 * G::R2 * n1 + 2 * l);
      const f2 w = ld2(win + 2 * G::R2 * n1 + 2 * l);
      st.yr[0][pos] = p.x * w.x;
      st.yi[0][pos] = p.y * w.y;
    } else {
      const f4 p = src.load4(2 * G::R2 * n1 + 4 * l);
      const f4 w = ld4(win + 2 * G::R2 * n1 + 4 * l);
      st.yr[0][pos] = p.x * w.x;
      st.yi[0][pos] = p.y * w.y;
      st.yr[1][pos] = p.z * w.z;
      st.yi[1][pos] = p.w * w.w;
    }
  });
  static_for<0, G::A>([&](auto Af) {
    constexpr int a = decltype(Af)::value;
    fft_dit<G::R1>(st.yr[a], st.yi[a]);
    const int n2 = G::A * l + a;
    static_for<1, G::R1>([&](auto K1) {
      constexpr int k1 = decltype(K1)::value;
      const cplx w = tw1[k1 * G::R2 + n2];
      const float r = st.yr[a][k1], i = st.yi[a][k1];
      st.yr[a][k1] = __builtin_fmaf(-i, w.y, r * w.x);
      st.yi[a][k1] = __builtin_fmaf(i, w.x, r * w.y);
    });
  });
}

// Exchange phase PH (0: k1 in [0, R1/2), 1: k1 in [R1/2, R1)): lane l writes row n2 of the frame's region.
template <class G, int PH>
SED_HD void lane_xwrite(const LaneState<G>& st, float* xfr, int l) {
  static_for<0, G::A>([&](auto Af) {
    constexpr int a = decltype(Af)::value;
    float* row = xfr + (G::A * l + a) * G::RP;
    static_for<0, G::R1 / 4>([&](auto Q) {
      constexpr int q = decltype(Q)::value;
      constexpr int k1 = PH * (G::R1 / 2) + 2 * q;
      st4(row + 4 * q, f4{st.yr[a][k1], st.yi[a][k1], st.yr[a][k1 + 1], st.yi[a][k1 + 1]});
    });
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
  static_for<0, G::R2>([&](auto N2) {
    constexpr int n2 = decltype(N2)::value;
    constexpr int pos = bitrev(n2, LOG2);
    const f2 p = ld2(col + n2 * G::RP);
    if constexpr (PH == 0) {
      st.pr[pos] = p.x;
      st.pi[pos] = p.y;
    } else {
      st.qr[pos] = p.x;
      st.qi[pos] = p.y;
    }
  });
}

// Bin index held in split slot s of lane l (the partner bin is NC - bin): l + s * lane_slot_stride(l).
template <class G>
SED_HD int lane_slot_stride(int l) {
  return l == 0 ? G::R1 / 2 : G::R1;
}
template <class G>
SED_HD int lane_slot_bin(int l, int s) {
  return l + s * lane_slot_stride<G>(l);
}

// Stage-2 FFTs, real-FFT split and |X|^2 (or |X|) into the frame's power row pw[0..NC].
// tw2[s*LPF + l] = W_NFFT^(lane_slot_bin(l, s)); slot_stride = lane_slot_stride(l), passed in so that a
// caller inside a loop can keep the per-slot addresses from being precomputed into registers.
template <class G, bool MAGNITUDE>
SED_HD void lane_stage2_power(LaneState<G>& st, float* pw, const cplx* tw2, int l, int slot_stride) {
  fft_dit<G::R2>(st.pr, st.pi);
  fft_dit<G::R2>(st.qr, st.qi);
  const bool l0 = (l == 0);
  static_for<0, G::SLOTS>([&](auto Sf) {
    constexpr int s = decltype(Sf)::value;
    // lane 0 holds Z[(R1/2) m], m = 0..2*R2-1: even m in P[m/2], odd m in Q[(m-1)/2]
    constexpr int mu = s, mv = (2 * G::R2 - s) % (2 * G::R2);
    float u0r, u0i, v0r, v0i;
    if constexpr (mu % 2 == 0) { u0r = st.pr[mu / 2]; u0i = st.pi[mu / 2]; }
    else { u0r = st.qr[mu / 2]; u0i = st.qi[mu / 2]; }
    if constexpr (mv % 2 == 0) { v0r = st.pr[mv / 2]; v0i = st.pi[mv / 2]; }
    else { v0r = st.qr[mv / 2]; v0i = st.qi[mv / 2]; }
    float ur, ui, vr, vi;
    if constexpr (s < G::R2) {
      ur = select(l0, u0r, st.pr[s]);
      ui = select(l0, u0i, st.pi[s]);
      vr = select(l0, v0r, st.qr[G::R2 - 1 - s]);
      vi = select(l0, v0i, st.qi[G::R2 - 1 - s]);
    } else {
      ur = u0r; ui = u0i; vr = v0r; vi = v0i;
    }
    const cplx w = tw2[s * G::LPF + l];
    const float er = ur + vr, ei = ui - vi;     // E = U + conj V
    const float orr = ui + vi, oi = vr - ur;    // O = -i (U - conj V)
    const float tr = __builtin_fmaf(-oi, w.y, orr * w.x);  // T = W O
    const float ti = __builtin_fmaf(oi, w.x, orr * w.y);
    const float ar = er + tr, ai = ei + ti;     // X[k]
    const float br = er - tr, bi = ei - ti;     // conj X[NC-k]
    float pa = __builtin_fmaf(ai, ai, ar * ar);
    float pb = __builtin_fmaf(bi, bi, br * br);
    if constexpr (MAGNITUDE) {
      pa = __builtin_sqrtf(pa);
      pb = __builtin_sqrtf(pb);
    }
    const int k = l + s * slot_stride;
    if (s < G::R2 || l0) {
      pw[k] = pa;
      pw[G::NC - k] = pb;
    }
  });
}

}  // namespace sed
