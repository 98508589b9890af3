"""Timing of the polyphase resampler on the rate pairs of the reference's loaders (librosa.load(..., sr=16000) of 44.1 / 48 kHz
files, librosa.resample to 32 kHz), 512 x 10 s per launch: python profiles/tools/resample_bench.py  (SED_HIP_LIB selects a build)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "sound-event-detection_amd"))
import sed_amd  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps * 1e3)
    return best


for orig, target, kind in ((44100, 16000, "f32"), (44100, 16000, "i16"), (48000, 16000, "f32"), (44100, 32000, "f32"),
                           (32000, 16000, "f32"), (22050, 16000, "f32"), (16000, 32000, "f32")):
    n = orig * 10
    x = torch.randn(512, n, device=dev) * 0.1
    if kind == "i16":
        x = (x * 30000).clamp(-32768, 32767).to(torch.int16)
    rs = sed_amd.Resampler(orig, target)
    y = rs(x)
    us = timed(lambda: rs(x))
    moved = x.numel() * x.element_size() + y.numel() * 4
    print(f"{orig:6d} -> {target:6d} {kind}  512 x 10 s  {us:8.1f} us  {moved / us / 1e6:6.2f} TB/s ({moved / us / 1e6 / 8 * 100:4.1f} % of 8 TB/s)  "
          f"{512 * 10 / 3600 / (us * 1e-6):8.0f} audio-h/s  checksum {float(y.double().abs().mean()):.9f}", flush=True)
