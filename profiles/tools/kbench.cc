// Times sed_logmel_forward of one or more builds of libsedhip.so on the headline workload (or another flavour) and
// compares their outputs with the first build's.  Diagnostic tool:
//   hipcc -O2 -o kbench kbench.cc -ldl ;  ./kbench [-c htsat|htsat_i16|htsat44|yamnet|crnn|dbmel|kaldi|kaldi44|htsat1h] [-n steps] [-b clips] lib1.so [lib2.so ...]
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sed_hip.h"

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(2);                                                                  \
    }                                                                           \
  } while (0)

// Libraries up to version 0.2 take the band edges of the configuration as fp32 (they are fp64 since 0.3): builds of
// both kinds can be timed in one call, each given the layout it was compiled for.
struct sed_frontend_cfg_v02 {
  int32_t sample_rate, n_fft, win_length, hop, center, pad_mode, win_align, tail_min_samples, tail_hop_samples, spectrum,
      mel_kind, n_mels;
  float fmin, fmax;
  int32_t log_kind;
  float amin, ref, ln_offset;
  int32_t pcm_kind;
  float pcm_divisor;
  int32_t window_kind, remove_dc;
  float preemphasis;
};

static sed_frontend_cfg_v02 to_v02(const sed_frontend_cfg& c) {
  return sed_frontend_cfg_v02{c.sample_rate, c.n_fft, c.win_length, c.hop, c.center, c.pad_mode, c.win_align,
                              c.tail_min_samples, c.tail_hop_samples, c.spectrum, c.mel_kind, c.n_mels, (float)c.fmin,
                              (float)c.fmax, c.log_kind, c.amin, c.ref, c.ln_offset, c.pcm_kind, c.pcm_divisor,
                              c.window_kind, c.remove_dc, c.preemphasis};
}

struct Lib {
  void* h;
  bool v02;
  int (*create)(const void*, sed_plan**);
  void (*destroy)(sed_plan*);
  int64_t (*frames)(const sed_plan*, int64_t);
  int (*forward)(const sed_plan*, const void*, int64_t, int64_t, int64_t, float*, int64_t, void*);
};

static Lib open_lib(const char* path) {
  Lib l{};
  l.h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!l.h) { fprintf(stderr, "dlopen %s: %s\n", path, dlerror()); exit(2); }
  l.create = (decltype(l.create))dlsym(l.h, "sed_plan_create");
  auto version = (const char* (*)())dlsym(l.h, "sed_version");
  l.v02 = version && (!strncmp(version(), "sedhip 0.1", 10) || !strncmp(version(), "sedhip 0.2", 10));
  l.destroy = (decltype(l.destroy))dlsym(l.h, "sed_plan_destroy");
  l.frames = (decltype(l.frames))dlsym(l.h, "sed_plan_num_frames");
  l.forward = (decltype(l.forward))dlsym(l.h, "sed_logmel_forward");
  if (!l.create || !l.destroy || !l.frames || !l.forward) { fprintf(stderr, "%s: missing symbols\n", path); exit(2); }
  return l;
}

int main(int argc, char** argv) {
  std::string cfgname = "htsat";
  int steps = 200, warm = 30;
  long long batch = 0;  // 0: the flavour's own batch
  int mels = 0, pre = -1, hop = 0;
  std::vector<const char*> libs;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "-c") && i + 1 < argc) cfgname = argv[++i];
    else if (!strcmp(argv[i], "-n") && i + 1 < argc) steps = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-w") && i + 1 < argc) warm = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-b") && i + 1 < argc) batch = atoll(argv[++i]);
    else if (!strcmp(argv[i], "-H") && i + 1 < argc) hop = atoi(argv[++i]);         // experiments: another hop
    else if (!strcmp(argv[i], "-M") && i + 1 < argc) mels = atoi(argv[++i]);        // experiments: another mel count
    else if (!strcmp(argv[i], "-P") && i + 1 < argc) pre = atoi(argv[++i]);         // bit 0: DC removal, bit 1: pre-emphasis
    else libs.push_back(argv[i]);
  }
  if (libs.empty()) { fprintf(stderr, "usage: kbench [-c cfg] [-n steps] lib.so ...\n"); return 2; }

  sed_frontend_cfg c{};
  int64_t B = 256, L = 320000;
  bool i16 = false;
  c.window_kind = SED_WINDOW_HANN_PERIODIC;
  c.pcm_kind = SED_PCM_F32;
  c.pcm_divisor = 32767.0f;
  if (cfgname == "htsat" || cfgname == "htsat_i16" || cfgname == "htsat1h" || cfgname == "htsat44") {
    c.sample_rate = 32000; c.n_fft = 1024; c.win_length = 1024; c.hop = 320; c.center = 1; c.pad_mode = SED_PAD_REFLECT;
    c.win_align = SED_WIN_CENTER; c.spectrum = SED_SPEC_POWER; c.mel_kind = SED_MEL_SLANEY; c.n_mels = 64;
    c.fmin = 50; c.fmax = 14000; c.log_kind = SED_LOG_DB10; c.amin = 1e-10f; c.ref = 1.0f;
    if (cfgname == "htsat_i16") { i16 = true; c.pcm_kind = SED_PCM_I16; }
    if (cfgname == "htsat1h") { B = 4; L = 3600LL * 32000; }
    if (cfgname == "htsat44") { c.sample_rate = 44100; B = 128; L = 882000; }  // the rate and clip length the reference runs (src/models/htsat/model.py:171-186)
  } else if (cfgname == "yamnet") {
    c.sample_rate = 16000; c.n_fft = 512; c.win_length = 400; c.hop = 160; c.center = 0; c.win_align = SED_WIN_LEFT;
    c.tail_min_samples = 15600; c.tail_hop_samples = 7680; c.spectrum = SED_SPEC_MAGNITUDE; c.mel_kind = SED_MEL_HTK_TF;
    c.n_mels = 64; c.fmin = 125; c.fmax = 7500; c.log_kind = SED_LOG_LN; c.ln_offset = 0.001f;
    B = 512; L = 160000;
  } else if (cfgname == "crnn" || cfgname == "dbmel") {
    c.sample_rate = 44100; c.n_fft = 2048; c.win_length = 2048; c.center = 1; c.pad_mode = SED_PAD_CONSTANT;
    c.win_align = SED_WIN_CENTER; c.mel_kind = SED_MEL_SLANEY; c.fmin = 0; c.fmax = 22050;
    if (cfgname == "crnn") { c.hop = 1024; c.spectrum = SED_SPEC_MAGNITUDE; c.n_mels = 40; c.log_kind = SED_LOG_LN; c.ln_offset = 1e-9f; }
    else { c.hop = 512; c.spectrum = SED_SPEC_POWER; c.n_mels = 64; c.log_kind = SED_LOG_DB10; c.amin = 1e-10f; c.ref = 1.0f; }
    B = 128; L = 441000;
  } else if (cfgname == "kaldi44") {  // the Audio-Mamba input at the rate of the reference's files (torchaudio.load gives 44.1 kHz)
    c.sample_rate = 44100; c.n_fft = 2048; c.win_length = 1102; c.hop = 441; c.center = 0; c.win_align = SED_WIN_LEFT;
    c.window_kind = SED_WINDOW_HANN_SYMMETRIC; c.remove_dc = 1; c.preemphasis = 0.97f; c.spectrum = SED_SPEC_POWER;
    c.mel_kind = SED_MEL_KALDI; c.n_mels = 128; c.fmin = 20; c.fmax = 0; c.log_kind = SED_LOG_LN_FLOOR; c.amin = 1.1920929e-07f;
    B = 256; L = 441000;
  } else if (cfgname == "kaldi") {
    c.sample_rate = 16000; c.n_fft = 512; c.win_length = 400; c.hop = 160; c.center = 0; c.win_align = SED_WIN_LEFT;
    c.window_kind = SED_WINDOW_HANN_SYMMETRIC; c.remove_dc = 1; c.preemphasis = 0.97f; c.spectrum = SED_SPEC_POWER;
    c.mel_kind = SED_MEL_KALDI; c.n_mels = 128; c.fmin = 20; c.fmax = 0; c.log_kind = SED_LOG_LN_FLOOR; c.amin = 1.1920929e-07f;
    B = 512; L = 160000;
  } else { fprintf(stderr, "unknown cfg %s\n", cfgname.c_str()); return 2; }

  if (batch > 0) B = batch;
  if (mels > 0) c.n_mels = mels;
  if (hop > 0) c.hop = hop;
  if (pre >= 0) { c.remove_dc = pre & 1; c.preemphasis = (pre & 2) ? 0.97f : 0.0f; }
  // deterministic noise + a tone + a stretch of silence
  const size_t n = (size_t)B * L;
  const size_t esz = i16 ? 2 : 4;
  std::vector<float> h(i16 ? 0 : n);
  std::vector<int16_t> h16(i16 ? n : 0);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  for (size_t i = 0; i < n; ++i) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    float v = ((int32_t)(s >> 33) / 1073741824.0f - 1.0f) * 0.1f + 0.2f * sinf(0.05f * (float)(i % L));
    if ((i % L) > (size_t)L / 2 && (i % L) < (size_t)L / 2 + 8000) v = 0.0f;
    if (i16) h16[i] = (int16_t)lrintf(v * 32767.0f); else h[i] = v;
  }
  void* d_pcm;
  CK(hipMalloc(&d_pcm, n * esz));
  CK(hipMemcpy(d_pcm, i16 ? (void*)h16.data() : (void*)h.data(), n * esz, hipMemcpyHostToDevice));

  std::vector<float> ref;
  for (size_t li = 0; li < libs.size(); ++li) {
    Lib lib = open_lib(libs[li]);
    sed_plan* pl = nullptr;
    const sed_frontend_cfg_v02 c02 = to_v02(c);
    int rc = lib.create(lib.v02 ? (const void*)&c02 : (const void*)&c, &pl);
    if (rc) { printf("%-40s plan_create failed %d\n", libs[li], rc); continue; }
    const int64_t T = lib.frames(pl, L);
    const size_t on = (size_t)B * T * c.n_mels;
    float* d_out;
    CK(hipMalloc((void**)&d_out, on * 4));
    CK(hipMemset(d_out, 0xff, on * 4));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int i = 0; i < warm; ++i) {
      rc = lib.forward(pl, d_pcm, B, L, L, d_out, T * c.n_mels, st);
      if (rc) { printf("forward failed %d\n", rc); return 1; }
    }
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // three timed blocks: report the best and the mean
    double best = 1e30, sum = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < steps; ++i) lib.forward(pl, d_pcm, B, L, L, d_out, T * c.n_mels, st);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / steps;
      best = std::min(best, us);
      sum += us;
    }
    std::vector<float> out(on);
    CK(hipMemcpy(out.data(), d_out, on * 4, hipMemcpyDeviceToHost));
    double asum = 0, maxd = 0;
    size_t nbad = 0, worst = 0;
    for (size_t i = 0; i < on; ++i) {
      if (!std::isfinite(out[i])) ++nbad;
      asum += fabs((double)out[i]);
      if (!ref.empty()) {
        const double d = fabs((double)out[i] - (double)ref[i]);
        if (d > maxd) { maxd = d; worst = i; }
      }
    }
    const double bytes = (double)n * esz + (double)on * 4;
    printf("%-44s %s B=%lld T=%lld  best %8.1f us  mean %8.1f us  %6.2f TB/s (%.1f%% of 8 TB/s)  mean|y| %.6f  nonfinite %zu", libs[li],
           cfgname.c_str(), (long long)B, (long long)T, best, sum / 3, bytes / best / 1e6, bytes / best / 1e6 / 8 * 100, asum / on, nbad);
    if (!ref.empty()) printf("  max|d vs first| %.3g at %zu", maxd, worst);
    printf("\n");
    fflush(stdout);
    if (ref.empty()) ref = out;
    CK(hipFree(d_out));
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    CK(hipStreamDestroy(st));
    lib.destroy(pl);
  }
  CK(hipFree(d_pcm));
  return 0;
}
