// CPU check of the lane decomposition in fft_core.h: every lane of one frame is executed phase by phase
// (the order a wavefront imposes) and the resulting power spectrum is compared with a naive fp64 DFT.
// Prints one line per geometry: "<n_fft> max_rel_err <value>"; exit status 1 if any exceeds 2e-6.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "tables.h"

using namespace sed;

template <class G, bool MAG>
double run_one(unsigned seed, int win_length, bool left) {
  std::mt19937 rng(seed);
  std::normal_distribution<float> nd(0.0f, 0.3f);
  std::vector<float> frame(G::NFFT);
  for (auto& v : frame) v = nd(rng);
  const auto wd = build_window(G::NFFT, win_length, left);
  // tables in the layout the kernel keeps in LDS
  std::vector<float> win(G::WIN_FLOATS, 0.0f), tw1(G::TW1_FLOATS, 0.0f), tw2(G::TW2_FLOATS, 0.0f);
  for (int i = 0; i < G::NFFT; ++i) win[win_slot<G>(i)] = (float)(0.5 * wd[i]);
  const auto t1 = build_tw1<G>();
  const auto t2 = build_tw2<G>();
  for (int k1 = 0; k1 < G::R1; ++k1)
    for (int n2 = 0; n2 < G::R2; ++n2) {
      tw1[tw1_slot<G>(k1, n2)] = t1[k1 * G::R2 + n2].x;
      tw1[tw1_slot<G>(k1, n2) + 1] = t1[k1 * G::R2 + n2].y;
    }
  for (int t = 0; t < G::SLOTS; ++t)
    for (int l = 0; l < G::LPF; ++l) {
      tw2[tw2_slot<G>(t, l)] = t2[t * G::LPF + l].x;
      tw2[tw2_slot<G>(t, l) + 1] = t2[t * G::LPF + l].y;
    }
  std::vector<LaneState<G>> st(G::LPF);
  std::vector<float> xfr(G::XFR, 0.0f), pw(G::PP, -1.0f);
  for (int l = 0; l < G::LPF; ++l) lane_stage1<G>(st[l], SrcF32{frame.data()}, win.data(), tw1.data(), l);
  for (int l = 0; l < G::LPF; ++l) lane_xwrite<G, 0>(st[l], xfr.data(), l);
  for (int l = 0; l < G::LPF; ++l) lane_xread<G, 0>(st[l], xfr.data(), l);
  for (int l = 0; l < G::LPF; ++l) lane_xwrite<G, 1>(st[l], xfr.data(), l);
  for (int l = 0; l < G::LPF; ++l) lane_xread<G, 1>(st[l], xfr.data(), l);
  for (int l = 0; l < G::LPF; ++l) lane_stage2_power<G, MAG>(st[l], pw.data(), tw2.data(), l);
  double worst = 0.0, scale = 0.0;
  std::vector<double> ref(G::NBINS);
  for (int k = 0; k <= G::NC; ++k) {
    double re = 0, im = 0;
    for (int n = 0; n < G::NFFT; ++n) {
      const double a = -2.0 * kPi * (double)(((int64_t)n * k) % G::NFFT) / G::NFFT;
      const double v = (double)frame[n] * wd[n];
      re += v * std::cos(a);
      im += v * std::sin(a);
    }
    ref[k] = MAG ? std::sqrt(re * re + im * im) : re * re + im * im;
    scale = std::max(scale, ref[k]);
  }
  for (int k = 0; k <= G::NC; ++k) worst = std::max(worst, std::fabs((double)pw[k] - ref[k]) / scale);
  return worst;
}

// Kaldi framing (DC removal over win_length samples, pre-emphasis, symmetric Hann, zero padding to n_fft) through the
// conditioned stage 1, against the same steps in fp64 followed by a naive DFT.
template <class G>
double run_conditioned(unsigned seed, int win_length, float coef) {
  std::mt19937 rng(seed);
  std::normal_distribution<float> nd(0.05f, 0.3f);
  std::vector<float> buf(G::NFFT + 1);
  for (auto& v : buf) v = nd(rng);
  const float* frame = buf.data() + 1;  // frame[-1] is readable
  const auto wd = build_window(G::NFFT, win_length, true, true);
  std::vector<float> win(G::WIN_FLOATS, 0.0f), tw1(G::TW1_FLOATS, 0.0f), tw2(G::TW2_FLOATS, 0.0f);
  for (int i = 0; i < G::NFFT; ++i) win[win_slot<G>(i)] = (float)(0.5 * wd[i]);
  const auto t1 = build_tw1<G>();
  const auto t2 = build_tw2<G>();
  for (int k1 = 0; k1 < G::R1; ++k1)
    for (int n2 = 0; n2 < G::R2; ++n2) {
      tw1[tw1_slot<G>(k1, n2)] = t1[k1 * G::R2 + n2].x;
      tw1[tw1_slot<G>(k1, n2) + 1] = t1[k1 * G::R2 + n2].y;
    }
  for (int t = 0; t < G::SLOTS; ++t)
    for (int l = 0; l < G::LPF; ++l) {
      tw2[tw2_slot<G>(t, l)] = t2[t * G::LPF + l].x;
      tw2[tw2_slot<G>(t, l) + 1] = t2[t * G::LPF + l].y;
    }
  float mean = 0.0f;
  for (int l = 0; l < G::LPF; ++l) mean += lane_frame_sum<G>(frame, l, win_length);
  mean *= 1.0f / (float)win_length;
  std::vector<LaneState<G>> st(G::LPF);
  std::vector<float> xfr(G::XFR, 0.0f), pw(G::PP, -1.0f);
  for (int l = 0; l < G::LPF; ++l) lane_stage1_conditioned<G>(st[l], frame, win.data(), tw1.data(), l, mean, coef);
  for (int l = 0; l < G::LPF; ++l) lane_xwrite<G, 0>(st[l], xfr.data(), l);
  for (int l = 0; l < G::LPF; ++l) lane_xread<G, 0>(st[l], xfr.data(), l);
  for (int l = 0; l < G::LPF; ++l) lane_xwrite<G, 1>(st[l], xfr.data(), l);
  for (int l = 0; l < G::LPF; ++l) lane_xread<G, 1>(st[l], xfr.data(), l);
  for (int l = 0; l < G::LPF; ++l) lane_stage2_power<G, false>(st[l], pw.data(), tw2.data(), l);
  double mu = 0.0;
  for (int n = 0; n < win_length; ++n) mu += frame[n];
  mu /= win_length;
  std::vector<double> y(G::NFFT, 0.0);
  for (int n = 0; n < win_length; ++n) {
    const double a = frame[n] - mu, b = frame[n == 0 ? 0 : n - 1] - mu;
    y[n] = (a - (double)coef * b) * wd[n];
  }
  double worst = 0.0, scale = 0.0;
  std::vector<double> ref(G::NBINS);
  for (int k = 0; k <= G::NC; ++k) {
    double re = 0, im = 0;
    for (int n = 0; n < G::NFFT; ++n) {
      const double a = -2.0 * kPi * (double)(((int64_t)n * k) % G::NFFT) / G::NFFT;
      re += y[n] * std::cos(a);
      im += y[n] * std::sin(a);
    }
    ref[k] = re * re + im * im;
    scale = std::max(scale, ref[k]);
  }
  for (int k = 0; k <= G::NC; ++k) worst = std::max(worst, std::fabs((double)pw[k] - ref[k]) / scale);
  return worst;
}

// Emulates the MFMA mel projection from its schedule tables and compares with the dense product.
static double check_mel_schedule(int sr, int n_fft, int M, double fmin, double fmax, int kind, int pitch) {
  const int F = n_fft / 2 + 1;  // kind 0: Slaney (librosa), 1: HTK (TensorFlow), 2: Kaldi
  const std::vector<float> W = kind == 2 ? mel_kaldi(sr, n_fft, M, fmin, fmax)
                               : kind == 1 ? mel_htk_tf(sr, n_fft, M, fmin, fmax) : mel_slaney(sr, n_fft, M, fmin, fmax);
  const MelMfmaSchedule sc = schedule_mel_mfma(W, F, M, pitch);
  std::mt19937 rng(5);
  std::uniform_real_distribution<float> ud(0.0f, 1.0f);
  std::vector<float> pw(4 * pitch, 0.0f);
  for (int j = 0; j < 4; ++j)
    for (int k = 0; k < F; ++k) pw[j * pitch + k] = ud(rng) * (k < 40 ? 100.0f : 1.0f);
  std::vector<double> part((size_t)16 * sc.sets * 16, 0.0);  // [position][i][j]
  for (int set = 0; set < sc.sets; ++set)
    for (int u = 0; u < sc.steps_of(set); ++u)
      for (int e = 0; e < 4; ++e)
        for (int block = 0; block < 16; ++block)
          for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
              const float a = sc.a[(size_t)sc.a_offset(set) + ((size_t)u * 64 + 4 * block + i) * 4 + e];
              if (sc.base[set * 16 + block] + 4 * u + e >= pitch) { std::printf("position reads past the row\n"); return 1.0; }
              const float b = pw[j * pitch + sc.base[set * 16 + block] + 4 * u + e];
              part[((size_t)sc.slot[set * 16 + block] * 4 + i) * 4 + j] += (double)a * b;
            }
  double worst = 0.0;
  for (int j = 0; j < 4; ++j)
    for (int m = 0; m < M; ++m) {
      double got = 0.0, want = 0.0;
      const int first = sc.quadpos[m / 4] & 0xffff, npos = sc.quadpos[m / 4] >> 16;
      for (int pi = 0; pi < npos; ++pi) got += part[((size_t)(first + pi) * 4 + m % 4) * 4 + j];
      for (int k = 0; k < F; ++k) want += (double)W[(size_t)k * M + m] * pw[j * pitch + k];
      worst = std::max(worst, std::fabs(got - want) / std::max(1e-30, std::fabs(want)));
    }
  std::printf("mel schedule n_fft=%d M=%d sets=%d steps %d x %d + %d x %d instr=%d max_pos=%d rel_err %.2e\n", n_fft, M, sc.sets,
              sc.n_hi, sc.u_hi, sc.sets - sc.n_hi, sc.u_lo, 4 * sc.total_steps(), sc.max_pos, worst);
  return worst;
}

int main() {
  int bad = 0;
  if (check_mel_schedule(32000, 1024, 64, 50, 14000, 0, 516) > 1e-12) bad = 1;
  if (check_mel_schedule(16000, 512, 64, 125, 7500, 1, 260) > 1e-12) bad = 1;
  if (check_mel_schedule(44100, 2048, 40, 0, 22050, 0, 1028) > 1e-12) bad = 1;
  if (check_mel_schedule(44100, 2048, 64, 0, 22050, 0, 1028) > 1e-12) bad = 1;
  if (check_mel_schedule(44100, 2048, 128, 0, 22050, 0, 1028) > 1e-12) bad = 1;
  if (check_mel_schedule(8000, 512, 3, 100, 3000, 0, 260) > 1e-12) bad = 1;
  if (check_mel_schedule(16000, 512, 128, 20, 0, 2, 260) > 1e-12) bad = 1;
  if (check_mel_schedule(32000, 1024, 128, 0, 16000, 0, 516) > 1e-12) bad = 1;
  if (check_mel_schedule(22050, 2048, 17, 300, 9000, 1, 1028) > 1e-12) bad = 1;
  if (check_mel_schedule(44100, 1024, 64, 50, 14000, 0, 516) > 1e-12) bad = 1;  // HTS-AT at the rate the reference runs it
  if (check_mel_schedule(44100, 2048, 128, 20, 0, 2, 1028) > 1e-12) bad = 1;    // Kaldi fbank at the rate of the reference's files
  auto report = [&](int n_fft, const char* what, double e) {
    std::printf("%d %s max_rel_err %.3e\n", n_fft, what, e);
    if (!(e < 2e-6)) bad = 1;
  };
  for (unsigned seed = 1; seed <= 3; ++seed) {
    report(512, "magnitude/left400", run_one<Geo<16, 16, 8, 12>, true>(seed, 400, true));
    report(512, "power", run_one<Geo<16, 16, 8, 12>, false>(seed, 512, false));
    report(1024, "power", run_one<Geo<32, 16, 16, 12>, false>(seed, 1024, false));
    report(2048, "magnitude", run_one<Geo<32, 32, 16, 8>, true>(seed, 2048, false));
    report(2048, "power", run_one<Geo<32, 32, 16, 8>, false>(seed, 2048, false));
  }
  report(512, "kaldi framing 400", run_conditioned<Geo<16, 16, 8, 12>>(7, 400, 0.97f));
  report(1024, "kaldi framing 800", run_conditioned<Geo<32, 16, 16, 12>>(8, 800, 0.97f));
  report(2048, "kaldi framing 1102", run_conditioned<Geo<32, 32, 16, 8>>(9, 1102, 0.97f));
  return bad;
}
