"""Times the HTS-AT glue kernels of the library named by SED_HIP_LIB (default: the in-tree one) and prints a digest of
their outputs, so that two builds can be compared: python profiles/tools/glue_bench.py"""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "sound-event-detection_amd"))
import sed_amd
dev = torch.device("cuda", 0)
def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def dig(t): return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:12]
g = torch.Generator(device=dev).manual_seed(0)
for name, shape, kw in (("reshape_wav2img 256 x (1001, 64) -> 256 x 256", (256, 1, 1001, 64), {}),
                        ("reshape_wav2img 64 x (2757, 64) -> 512 x 512", (64, 1, 2757, 64), dict(spec_size=512, freq_ratio=8)),
                        ("reshape_wav2img 256 x (1024, 64) (copy)", (256, 1, 1024, 64), {})):
    lm = torch.randn(shape, generator=g, device=dev)
    img = sed_amd.reshape_wav2img(lm, **kw)
    us = timeit(lambda: sed_amd.reshape_wav2img(lm, **kw))
    print(f"{name:52s} {us:7.1f} us  {(lm.numel() + img.numel()) * 4 / us / 1e6:5.2f} TB/s  digest {dig(img)}")
from sed_amd import decode as D
for name, shape, ratio, frames in (("interpolate_and_pad 256 x (128, 6) x 32 -> 4096", (256, 128, 6), 32, 4096),
                                   ("interpolate_and_pad 64 x (128, 527) x 32 -> 4096", (64, 128, 527), 32, 4096)):
    x = torch.rand(shape, generator=g, device=dev)
    y = D.interpolate_and_pad(x, ratio, frames)
    us = timeit(lambda: D.interpolate_and_pad(x, ratio, frames))
    print(f"{name:52s} {us:7.1f} us  {(x.numel() + y.numel()) * 4 / us / 1e6:5.2f} TB/s  digest {dig(y)}")
# unfused parity path: Spectrogram (writes (B, T, 513)) and LogmelFilterBank on it
x = 0.1 * torch.randn((64, 320000), generator=g, device=dev)
sp = sed_amd.Spectrogram(n_fft=1024, hop_length=320)
lf = sed_amd.LogmelFilterBank(sr=32000, n_fft=1024, n_mels=64, fmin=50, fmax=14000, top_db=None)
s = sp(x)
us = timeit(lambda: sp(x), n=50)
print(f"{'Spectrogram 64 x 10 s (unfused)':52s} {us:7.1f} us  {(x.numel() + s.numel()) * 4 / us / 1e6:5.2f} TB/s  digest {dig(s)}")
m = lf(s)
us = timeit(lambda: lf(s), n=50)
print(f"{'LogmelFilterBank on it (unfused)':52s} {us:7.1f} us  {(s.numel() + m.numel()) * 4 / us / 1e6:5.2f} TB/s  digest {dig(m)}")
# dB post-pass of the librosa flavour: per-clip max, subtract, floor (128 clips x (862, 64))
db = -60.0 * torch.rand((128, 862, 64), generator=g, device=dev)
from sed_amd import frontend as FE
import sed_amd._lib as L
def dbn():
    t = db.clone()
    with torch.cuda.device(dev):
        L.check(L.lib.sed_db_normalize(t.data_ptr(), 128, 862 * 64, 1, 80.0, L.stream_ptr(dev)), "db")
    return t
t = dbn()
us_clone = timeit(lambda: db.clone())
us = timeit(dbn)
print(f"{'sed_db_normalize 128 x (862, 64)':52s} {us - us_clone:7.1f} us (clone {us_clone:.1f} subtracted)  {2 * db.numel() * 4 / max(us - us_clone, 1e-3) / 1e6:5.2f} TB/s  digest {dig(t)}")
# detector -> classifier hand-off: 2048 events of 1..3 s cut out of 256 x 10 s @ 32 kHz into a (2048, 96000) batch
import numpy as np
rng = np.random.default_rng(0)
pcm = 0.1 * torch.randn((256, 320000), generator=g, device=dev)
ev = [(int(rng.integers(256)), float(a), float(a + d)) for a, d in zip(rng.uniform(0, 7, 2048), rng.uniform(1, 3, 2048))]
clips, lens = sed_amd.cut_events(pcm, 32000, ev, cap=96000)
us = timeit(lambda: sed_amd.cut_events(pcm, 32000, ev, cap=96000), n=30)
moved = int(lens.clamp(max=96000).sum().item()) * 4 + clips.numel() * 4
print(f"{'cut_events 2048 x <= 3 s -> (2048, 96000)':52s} {us:7.1f} us  {moved / us / 1e6:5.2f} TB/s  digest {dig(clips[:64])}")
# torchlibrosa's top_db: one group = the whole (64, 1, 1001, 64) tensor
big = -60.0 * torch.rand((64, 1001, 64), generator=g, device=dev)
def dbn1():
    t = big.clone()
    with torch.cuda.device(dev):
        L.check(L.lib.sed_db_normalize(t.data_ptr(), 1, t.numel(), 0, 80.0, L.stream_ptr(dev)), "db")
    return t
t = dbn1()
us_clone = timeit(lambda: big.clone())
us = timeit(dbn1)
print(f"{'sed_db_normalize 1 x (64 * 1001 * 64) (top_db)':52s} {us - us_clone:7.1f} us (clone {us_clone:.1f} subtracted)  {2 * big.numel() * 4 / max(us - us_clone, 1e-3) / 1e6:5.2f} TB/s  digest {dig(t)}")
