// Host-side table builders (fp64 arithmetic, rounded once to fp32): window, FFT twiddles for the lane
// decomposition in fft_core.h, mel filterbanks, and the banded/lane-scheduled form of a mel matrix.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>

#include "fft_core.h"

namespace sed {

constexpr double kPi = 3.14159265358979323846264338327950288;

// Periodic Hann of win_length placed in n_fft samples: centred (librosa pad_center) or left aligned
// (tf.signal.stft zero-extends the frame at its end).
// symmetric: torch.hann_window(win_length, periodic=False), denominator win_length - 1.
inline std::vector<double> build_window(int n_fft, int win_length, bool left_aligned, bool symmetric = false) {
  std::vector<double> w(n_fft, 0.0);
  const int off = left_aligned ? 0 : (n_fft - win_length) / 2;
  const double den = symmetric ? std::max(1, win_length - 1) : win_length;
  for (int n = 0; n < win_length; ++n) w[off + n] = 0.5 - 0.5 * std::cos(2.0 * kPi * n / den);
  if (symmetric && win_length == 1) w[off] = 1.0;
  return w;
}

inline cplx unit_root(int64_t num, int64_t den) {  // exp(-2 pi i num/den), argument reduced exactly
  num %= den;
  const double a = -2.0 * kPi * (double)num / (double)den;
  return cplx{(float)std::cos(a), (float)std::sin(a)};
}

template <class G>
std::vector<cplx> build_tw1() {  // [k1*R2 + n2] = W_NC^(n2 k1)
  std::vector<cplx> t(G::R1 * G::R2);
  for (int k1 = 0; k1 < G::R1; ++k1)
    for (int n2 = 0; n2 < G::R2; ++n2) t[k1 * G::R2 + n2] = unit_root((int64_t)n2 * k1, G::NC);
  return t;
}

template <class G>
std::vector<cplx> build_tw2() {  // [t*LPF + l] = W_NFFT^(bin of slot t in lane l)
  std::vector<cplx> t(G::SLOTS * G::LPF);
  for (int s = 0; s < G::SLOTS; ++s)
    for (int l = 0; l < G::LPF; ++l) t[s * G::LPF + l] = unit_root(lane_slot_bin<G>(l, s), G::NFFT);
  return t;
}

// ---- mel filterbanks -------------------------------------------------------------------------------

// numpy.linspace(start, stop, num): arange(num) * ((stop-start)/(num-1)) + start, last point = stop.
inline std::vector<double> linspace(double start, double stop, int num) {
  std::vector<double> y(num);
  const double step = (stop - start) / (num - 1);
  for (int i = 0; i < num; ++i) y[i] = i * step + start;
  y[num - 1] = stop;
  return y;
}

inline double hz_to_mel_slaney(double f) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
  const double logstep = std::log(6.4) / 27.0;
  return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
inline double mel_to_hz_slaney(double m) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
  const double logstep = std::log(6.4) / 27.0;
  return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

// librosa.filters.mel(htk=False, norm='slaney') transposed: melW[k*n_mels + m], k = 0..n_fft/2.  Operation for operation
// as librosa forms it: bin frequencies k * (1 / (n_fft * (1 / sr))) (np.fft.rfftfreq), the triangle rounded to fp32 (it is
// stored into an fp32 array), then multiplied by the fp64 Slaney factor and rounded to fp32 again (`weights *= enorm`).
inline std::vector<float> mel_slaney(int sr, int n_fft, int n_mels, double fmin, double fmax) {
  const int F = n_fft / 2 + 1;
  std::vector<double> pts = linspace(hz_to_mel_slaney(fmin), hz_to_mel_slaney(fmax), n_mels + 2);
  for (double& p : pts) p = mel_to_hz_slaney(p);
  const double d = 1.0 / (double)sr;
  const double val = 1.0 / ((double)n_fft * d);
  std::vector<float> w((size_t)F * n_mels, 0.0f);
  for (int m = 0; m < n_mels; ++m) {
    const double enorm = 2.0 / (pts[m + 2] - pts[m]);
    for (int k = 0; k < F; ++k) {
      const double f = (double)k * val;
      const double lower = -(pts[m] - f) / (pts[m + 1] - pts[m]);
      const double upper = (pts[m + 2] - f) / (pts[m + 2] - pts[m + 1]);
      const float tri = (float)std::max(0.0, std::min(lower, upper));
      w[(size_t)k * n_mels + m] = (float)((double)tri * enorm);
    }
  }
  return w;
}

// tf.signal.linear_to_mel_weight_matrix: HTK mel, triangles linear in mel, peak 1, DC row zero.
inline std::vector<float> mel_htk_tf(int sr, int n_fft, int n_mels, double fmin, double fmax) {
  const int F = n_fft / 2 + 1;
  auto h2m = [](double f) { return 1127.0 * std::log1p(f / 700.0); };
  const std::vector<double> e = linspace(h2m(fmin), h2m(fmax), n_mels + 2);
  const std::vector<double> freqs = linspace(0.0, (double)sr / 2, F);
  std::vector<float> w((size_t)F * n_mels, 0.0f);
  for (int k = 1; k < F; ++k) {
    const double mel = h2m(freqs[k]);
    for (int m = 0; m < n_mels; ++m) {
      const double lower = (mel - e[m]) / (e[m + 1] - e[m]);
      const double upper = (e[m + 2] - mel) / (e[m + 2] - e[m + 1]);
      w[(size_t)k * n_mels + m] = (float)std::max(0.0, std::min(lower, upper));
    }
  }
  return w;
}

// torchaudio.compliance.kaldi.get_mel_banks (vtln_warp 1): triangles linear in mel = 1127 ln(1 + f/700) between
// low_freq and high_freq (<= 0: offset from Nyquist), evaluated at the n_fft/2 bins below Nyquist; the Nyquist row is
// zero (kaldi.fbank pads the matrix with a zero column).
inline std::vector<float> mel_kaldi(int sr, int n_fft, int n_mels, double low_freq, double high_freq) {
  const int F = n_fft / 2 + 1;
  auto mel = [](double f) { return 1127.0 * std::log(1.0 + f / 700.0); };
  if (high_freq <= 0.0) high_freq += 0.5 * sr;
  const double bin_width = (double)sr / n_fft;
  const double mel_low = mel(low_freq), mel_high = mel(high_freq);
  const double delta = (mel_high - mel_low) / (n_mels + 1);
  std::vector<float> w((size_t)F * n_mels, 0.0f);
  for (int k = 0; k < n_fft / 2; ++k) {
    const double m = mel(bin_width * k);
    for (int b = 0; b < n_mels; ++b) {
      const double left = mel_low + b * delta, center = left + delta, right = left + 2.0 * delta;
      const double up = (m - left) / (center - left), down = (right - m) / (right - center);
      w[(size_t)k * n_mels + b] = (float)std::max(0.0, std::min(up, down));
    }
  }
  return w;
}

// ---- banded form + per-lane schedule ---------------------------------------------------------------

struct alignas(16) MelEntry {  // one mel filter as seen by the lane that evaluates it
  int32_t mel;     // output index, -1 = padding entry
  int32_t start;   // first bin of the band
  int32_t len;     // number of consecutive bins (0 = empty filter)
  int32_t woff;    // offset of its weights in the packed weight array
};

struct MelSchedule {
  int lanes = 0, slots = 0;            // entries[lane*slots + j]
  std::vector<MelEntry> entries;
  std::vector<float> weights;          // bands back to back
};

// Compress a dense [F][M] matrix to one contiguous band per filter and spread the filters over `lanes`
// lanes so that the per-lane sums of band lengths are balanced (longest first, to the least loaded lane).
inline MelSchedule schedule_mel(const std::vector<float>& melW, int F, int M, int lanes) {
  MelSchedule s;
  s.lanes = lanes;
  s.slots = (M + lanes - 1) / lanes;
  std::vector<MelEntry> bands(M);
  for (int m = 0; m < M; ++m) {
    int lo = F, hi = -1;
    for (int k = 0; k < F; ++k)
      if (melW[(size_t)k * M + m] != 0.0f) { lo = std::min(lo, k); hi = std::max(hi, k); }
    MelEntry e{m, 0, 0, (int32_t)s.weights.size()};
    if (hi >= lo) {
      e.start = lo; e.len = hi - lo + 1;
      for (int k = lo; k <= hi; ++k) s.weights.push_back(melW[(size_t)k * M + m]);
    }
    bands[m] = e;
  }
  std::vector<int> order(M);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return bands[a].len > bands[b].len; });
  std::vector<std::vector<MelEntry>> per(lanes);
  std::vector<int> load(lanes, 0);
  for (int m : order) {
    int best = -1;
    for (int l = 0; l < lanes; ++l)
      if ((int)per[l].size() < s.slots && (best < 0 || load[l] < load[best])) best = l;
    per[best].push_back(bands[m]);
    load[best] += bands[m].len;
  }
  s.entries.assign((size_t)lanes * s.slots, MelEntry{-1, 0, 0, 0});
  for (int l = 0; l < lanes; ++l)
    for (size_t j = 0; j < per[l].size(); ++j) s.entries[(size_t)l * s.slots + j] = per[l][j];
  if (s.weights.empty()) s.weights.push_back(0.0f);
  return s;
}

// ---- mel projection on the 4x4x1 16-block f32 MFMA -------------------------------------------------
// One v_mfma_f32_4x4x1_16b_f32 multiplies, in each of its 16 blocks, 4 filter weights (one bin, 4
// consecutive mel filters = a "quad") by the powers of that bin in 4 frames.  A "position" is one block
// of one accumulator set; it owns a segment of consecutive bins of one quad.  Long quads are cut into
// several positions.  All positions of a set run the same number of steps (4 bins each); the first n_hi
// sets run u_hi steps, the others u_lo <= u_hi, so that short quads do not cost as many instructions as
// the pieces of long ones (HTS-AT: 6 + 4 steps instead of 6 + 6).
struct MelMfmaSchedule {
  int sets = 0;        // accumulator sets (4 VGPRs each); positions = 16 * sets
  int tau = 0;         // bins per position of the long sets (= 4 * u_hi)
  int u_hi = 0, u_lo = 0, n_hi = 0;  // steps of sets [0, n_hi) and of sets [n_hi, sets)
  int n_quads = 0;
  int max_pos = 0;     // most positions any quad was cut into
  std::vector<float> a;         // [set][steps of the set][lane = 4*block + i][4]: weight of bin base+4u+e for mel 4*quad+i
  std::vector<int32_t> base;    // [set][block]: first bin the position reads (multiple of 4, base + 4 * steps <= row_pitch)
  std::vector<int32_t> slot;    // [set][block]: where the block's partial sums go (positions in quad order; unused blocks behind them)
  std::vector<int32_t> quadpos; // [quad]: first slot | (number of positions << 16); a quad's slots are consecutive
  int steps_of(int set) const { return set < n_hi ? u_hi : u_lo; }
  int a_offset(int set) const { return 256 * (std::min(set, n_hi) * u_hi + std::max(0, set - n_hi) * u_lo); }  // floats
  int total_steps() const { return n_hi * u_hi + (sets - n_hi) * u_lo; }
};

// Cuts quads of n[q] steps into pieces of at most u_hi steps (at most cnt_hi of them) and of at most u_lo steps (at most
// cnt_lo), no quad into more than max_pieces: a_out[q] long and b_out[q] short pieces, or false.  Exact (dynamic
// programme over the quads: fewest short pieces for every number of long pieces used so far).
inline bool mel_cut_quads(const std::vector<int>& n, int u_hi, int cnt_hi, int u_lo, int cnt_lo, int max_pieces,
                          std::vector<int>* a_out, std::vector<int>* b_out) {
  const int Q = (int)n.size(), INF = 1 << 28;
  std::vector<std::vector<int>> best(Q + 1, std::vector<int>(cnt_hi + 1, INF)), choice(Q + 1, std::vector<int>(cnt_hi + 1, -1));
  best[0][0] = 0;
  for (int q = 0; q < Q; ++q)
    for (int h = 0; h <= cnt_hi; ++h) {
      if (best[q][h] >= INF) continue;
      const int need = std::max(1, n[q]);
      for (int a = 0; a <= std::min(max_pieces, cnt_hi - h); ++a) {
        const int rem = need - a * u_hi;
        int b = rem > 0 ? (u_lo > 0 ? (rem + u_lo - 1) / u_lo : INF) : 0;
        if (a + b == 0) b = 1;
        if (b >= INF || a + b > max_pieces) { if (rem <= 0) break; continue; }
        if (best[q][h] + b < best[q + 1][h + a]) { best[q + 1][h + a] = best[q][h] + b; choice[q + 1][h + a] = a; }
        if (rem <= 0) break;
      }
    }
  int h = -1;
  for (int k = 0; k <= cnt_hi; ++k)
    if (best[Q][k] <= cnt_lo && (h < 0 || best[Q][k] < best[Q][h])) h = k;
  if (h < 0) return false;
  a_out->assign(Q, 0);
  b_out->assign(Q, 0);
  for (int q = Q; q > 0; --q) {
    const int a = choice[q][h];
    (*a_out)[q - 1] = a;
    (*b_out)[q - 1] = best[q][h] - best[q - 1][h - a];
    h -= a;
  }
  return true;
}

inline MelMfmaSchedule schedule_mel_mfma(const std::vector<float>& melW, int F, int M, int row_pitch, int max_sets = 4) {
  MelMfmaSchedule sc;
  const int NQ = (M + 3) / 4;
  sc.n_quads = NQ;
  std::vector<int> lo(NQ, 0), len(NQ, 0), steps(NQ, 0);
  for (int q = 0; q < NQ; ++q) {
    int a = F, b = -1;
    for (int k = 0; k < F; ++k)
      for (int i = 0; i < 4 && 4 * q + i < M; ++i)
        if (melW[(size_t)k * M + 4 * q + i] != 0.0f) { a = std::min(a, k); b = std::max(b, k); }
    if (b >= a) { lo[q] = a / 4 * 4; len[q] = (b + 1 - lo[q] + 3) / 4 * 4; }
    steps[q] = len[q] / 4;
  }
  // Fewest instructions first (4 per step and set), then fewest positions per quad (gather reads), then fewest sets.
  // Total steps are tried in ascending order from the bound 16 * total >= sum of steps; for one total every split into
  // n_hi sets of u_hi and the rest of u_lo < u_hi steps (or all sets alike) is tested with the exact cutter.
  const int max_u = std::max(1, row_pitch / 4);
  int sum_steps = 0, longest = 1;
  for (int q = 0; q < NQ; ++q) { sum_steps += std::max(1, steps[q]); longest = std::max(longest, steps[q]); }
  std::vector<int> cut_a, cut_b;
  bool found = false;
  for (int total = std::max(1, (sum_steps + 15) / 16); !found && total <= max_sets * std::min(longest, max_u); ++total) {
    int best_pos = 1 << 28, best_sets = 0;
    for (int s = 1; s <= max_sets; ++s) {
      if (16 * s < NQ) continue;
      for (int n_hi = 1; n_hi <= s; ++n_hi)
        for (int u_hi = 1; u_hi <= std::min(longest, max_u); ++u_hi) {
          const int rest = total - n_hi * u_hi;
          int u_lo;
          if (n_hi == s) { if (rest != 0) continue; u_lo = u_hi; }
          else { if (rest <= 0 || rest % (s - n_hi) != 0) continue; u_lo = rest / (s - n_hi); if (u_lo >= u_hi) continue; }
          for (int pieces = 1; pieces <= 16 && pieces <= best_pos; ++pieces) {
            std::vector<int> a, b;
            if (!mel_cut_quads(steps, u_hi, 16 * n_hi, u_lo, n_hi == s ? 0 : 16 * (s - n_hi), pieces, &a, &b)) continue;
            if (pieces < best_pos || (pieces == best_pos && s < best_sets)) {
              best_pos = pieces; best_sets = s;
              sc.sets = s; sc.n_hi = n_hi; sc.u_hi = u_hi; sc.u_lo = u_lo;
              cut_a = a; cut_b = b;
              found = true;
            }
            break;
          }
        }
    }
  }
  if (!found) {  // (cannot happen for max_sets * 16 >= NQ: one long piece per quad always fits)
    sc.sets = max_sets; sc.n_hi = max_sets; sc.u_hi = sc.u_lo = std::min(longest, max_u);
    mel_cut_quads(steps, sc.u_hi, 16 * max_sets, sc.u_lo, 0, 16, &cut_a, &cut_b);
  }
  sc.tau = 4 * sc.u_hi;
  const int P = 16 * sc.sets;
  sc.a.assign((size_t)256 * sc.total_steps(), 0.0f);
  sc.base.assign(P, 0);
  sc.slot.assign(P, 0);
  // Positions in quad order (= the order of their partial-sum slots), then their places (set, block).  The 16 lanes the
  // LDS serves together when the power values are fetched are 4 consecutive blocks x 4 frames, frame rows 16 banks
  // apart, 4 banks per lane: they fall on distinct banks when the first bins of the 4 blocks differ mod 16 in their
  // multiple of 4.  So the positions are dealt out to the groups of 4 blocks like cards, sorted by that residue: every
  // group gets its even share of each residue (all four when the filterbank offers them).  Long pieces go to the long
  // sets, short ones to the short sets.
  struct Pos { int quad, seg_lo, seg_hi, base, is_hi; };
  std::vector<Pos> pos;
  std::vector<std::vector<int>> of_quad(NQ);
  for (int q = 0; q < NQ; ++q) {
    int seg_lo = lo[q];
    const int q_end = lo[q] + len[q];
    const int pieces = std::max(1, cut_a[q] + cut_b[q]);
    for (int g = 0; g < pieces; ++g) {
      const int is_hi = g < cut_a[q] ? 1 : 0;
      const int tau = 4 * (is_hi ? sc.u_hi : sc.u_lo);
      const int seg_hi = std::min(q_end, seg_lo + tau);                                 // bins owned
      const int base = std::max(0, std::min(seg_lo, row_pitch / 4 * 4 - tau));           // bins read
      of_quad[q].push_back((int)pos.size());
      pos.push_back(Pos{q, seg_lo, seg_hi, base, is_hi});
      seg_lo = seg_hi;
    }
  }
  std::vector<int> place_of(pos.size(), 0);
  std::vector<char> used(P, 0);
  for (int kind = 1; kind >= 0; --kind) {
    const int first_group = kind ? 0 : 4 * sc.n_hi, n_groups = kind ? 4 * sc.n_hi : 4 * (sc.sets - sc.n_hi);
    std::vector<int> order;
    for (size_t i = 0; i < pos.size(); ++i)
      if (pos[i].is_hi == kind) order.push_back((int)i);
    if (order.empty() || n_groups == 0) continue;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return (pos[x].base / 4) % 4 < (pos[y].base / 4) % 4; });
    std::vector<int> filled(n_groups, 0);
    for (size_t i = 0; i < order.size(); ++i) {
      const int grp = (int)(i % n_groups);
      place_of[order[i]] = 4 * (first_group + grp) + filled[grp]++;
      used[place_of[order[i]]] = 1;
    }
  }
  int spare = (int)pos.size();  // unused places (all-zero weights) keep slots of their own behind the real ones
  for (int pl = 0; pl < P; ++pl)
    if (!used[pl]) sc.slot[pl] = spare++;
  for (size_t p = 0; p < pos.size(); ++p) {
    const int q = pos[p].quad, seg_lo = pos[p].seg_lo, seg_hi = pos[p].seg_hi, base = pos[p].base;
    const int place = place_of[p];
    sc.base[place] = base;
    sc.slot[place] = (int)p;
    const int set = place / 16, block = place % 16;
    const int tau = 4 * sc.steps_of(set);
    for (int t = 0; t < tau; ++t) {
      const int k = base + t;
      if (k < seg_lo || k >= seg_hi || k >= F) continue;
      for (int i = 0; i < 4 && 4 * q + i < M; ++i)
        sc.a[(size_t)sc.a_offset(set) + ((size_t)(t / 4) * 64 + 4 * block + i) * 4 + t % 4] = melW[(size_t)k * M + 4 * q + i];
    }
  }
  for (const auto& v : of_quad) sc.max_pos = std::max(sc.max_pos, (int)v.size());
  sc.quadpos.assign(NQ, 0);
  for (int q = 0; q < NQ; ++q) sc.quadpos[q] = of_quad[q].front() | ((int32_t)of_quad[q].size() << 16);
  return sc;
}

}  // namespace sed
