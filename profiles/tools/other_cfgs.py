"""Times the front-end flavours other than the headline one (diagnostic)."""
import sys, time, torch, numpy as np
sys.path.insert(0, "sound-event-detection_amd")
import sed_amd
dev = torch.device("cuda", 0)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(0)
cases = [
  ("htsat f32 256x320000", sed_amd.FrontendConfig.htsat(), torch.randn(256, 320000, generator=g) * 0.1),
  ("htsat i16 256x320000", sed_amd.FrontendConfig.htsat(pcm_kind=1), (torch.randn(256, 320000, generator=g) * 3000).to(torch.int16)),
  ("htsat f32 unaligned rows", sed_amd.FrontendConfig.htsat(), None),
  ("yamnet f32 512x160000", sed_amd.FrontendConfig.yamnet(), torch.randn(512, 160000, generator=g) * 0.1),
  ("crnn f32 128x441000", sed_amd.FrontendConfig.crnn(), torch.randn(128, 441000, generator=g) * 0.1),
  ("db_mel f32 128x441000", sed_amd.FrontendConfig.db_mel(), torch.randn(128, 441000, generator=g) * 0.1),
  ("kaldi fbank 16k 128 mel 512x160000", sed_amd.FrontendConfig.kaldi_fbank(16000.0, 128), torch.randn(512, 160000, generator=g) * 0.1),
]
for name, cfg, x in cases:
    if x is None:
        base = (torch.randn(256, 320001, generator=g) * 0.1).to(dev)
        x = base[:, 1:]
    else:
        x = x.to(dev)
    fe = sed_amd.LogMelFrontend(cfg, layout="btm")
    y = fe(x)
    us = timeit(lambda: fe(x))
    esz = 2 if x.dtype == torch.int16 else 4
    byts = x.shape[0] * x.shape[1] * esz + y.numel() * 4
    hours = x.shape[0] * x.shape[1] / cfg.sample_rate / 3600
    print(f"{name:28s} {us:8.1f} us  {byts/us/1e6:7.2f} TB/s algorithmic  {hours/(us*1e-6):9.0f} audio-h/s  frames {y.shape[1]}")

# detector -> classifier hand-off: 2048 events of 1..3 s cut out of 256 x 10 s @ 32 kHz
rng = np.random.default_rng(0)
pcm = (torch.randn(256, 320000, generator=g) * 0.1).to(dev)
ev = [(int(rng.integers(256)), float(a), float(a + d)) for a, d in zip(rng.uniform(0, 7, 2048), rng.uniform(1, 3, 2048))]
clips, lens = sed_amd.cut_events(pcm, 32000, ev, cap=96000)
us = timeit(lambda: sed_amd.cut_events(pcm, 32000, ev, cap=96000))
moved = int(lens.clamp(max=96000).sum().item()) * 4 + clips.numel() * 4
print(f"{'cut_events 2048 x <=3 s':28s} {us:8.1f} us  {moved/us/1e6:7.2f} TB/s (slice reads + padded batch writes; includes the host-side index upload)")

# HTS-AT glue: (256, 1, 1001, 64) log-mel -> (256, 1, 256, 256) image
lm = torch.randn(256, 1, 1001, 64, device=dev)
img = sed_amd.reshape_wav2img(lm)
us = timeit(lambda: sed_amd.reshape_wav2img(lm))
print(f"{'reshape_wav2img 256 clips':28s} {us:8.1f} us  {(lm.numel() + img.numel()) * 4 / us / 1e6:7.2f} TB/s (read log-mel + write image)")

# BASELINE.json configs[4]: 1-hour recordings (4 x 3600 s @ 32 kHz fp32 = 1.84 GB)
xl = torch.randn(4, 3600 * 32000, device=dev) * 0.1
fe = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat(), layout="btm")
y = fe(xl)
us = timeit(lambda: fe(xl), n=10)
print(f"{'htsat 4 x 1 h':28s} {us:8.1f} us  {(xl.numel() + y.numel()) * 4 / us / 1e6:7.2f} TB/s algorithmic  {4 / (us * 1e-6):9.0f} audio-h/s  frames {y.shape[1]}")

# sample-rate conversion ahead of the YAMNet front-end: 512 x 10 s, 44.1 kHz -> 16 kHz
for name, xin in (("resample 44.1k->16k f32 512x10s", torch.randn(512, 441000, device=dev) * 0.1),
                  ("resample 44.1k->16k i16 512x10s", (torch.randn(512, 441000, device=dev) * 3000).to(torch.int16)),
                  ("resample 48k->16k f32 512x10s", torch.randn(512, 480000, device=dev) * 0.1)):
    rs = sed_amd.Resampler(48000 if "48k" in name else 44100, 16000)
    y = rs(xin)
    us = timeit(lambda: rs(xin), n=20)
    moved = xin.numel() * xin.element_size() + y.numel() * 4
    sr_in = 48000 if "48k" in name else 44100
    print(f"{name:34s} {us:8.1f} us  {moved / us / 1e6:7.2f} TB/s (read + write)  {xin.shape[0] * xin.shape[1] / sr_in / 3600 / (us * 1e-6):9.0f} audio-h/s")
