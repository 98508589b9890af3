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
  std::vector<float> win(G::NFFT);
  for (int i = 0; i < G::NFFT; ++i) win[i] = (float)(0.5 * wd[i]);
  const auto tw1 = build_tw1<G>();
  const auto tw2 = build_tw2<G>();
  std::vector<LaneState<G>> st(G::LPF);
  std::vector<float> xfr(G::XFR, 0.0f), pw(G::PP, -1.0f);
  for (int l = 0; l < G::LPF; ++l) lane_stage1<G>(st[l], frame.data(), win.data(), tw1.data(), l);
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

int main() {
  int bad = 0;
  auto report = [&](int n_fft, const char* what, double e) {
    std::printf("%d %s max_rel_err %.3e\n", n_fft, what, e);
    if (!(e < 2e-6)) bad = 1;
  };
  for (unsigned seed = 1; seed <= 3; ++seed) {
    report(512, "magnitude/left400", run_one<Geo<16, 16, 8, 4>, true>(seed, 400, true));
    report(512, "power", run_one<Geo<16, 16, 8, 4>, false>(seed, 512, false));
    report(1024, "power", run_one<Geo<32, 16, 16, 4>, false>(seed, 1024, false));
    report(2048, "magnitude", run_one<Geo<32, 32, 16, 2>, true>(seed, 2048, false));
    report(2048, "power", run_one<Geo<32, 32, 16, 2>, false>(seed, 2048, false));
  }
  return bad;
}
