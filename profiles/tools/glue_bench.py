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
