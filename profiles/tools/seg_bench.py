"""Timing of the fused hand-off (sed_logmel_forward_segments through sed_amd.logmel_of_events) against the dense batch, by
segment alignment and length: python profiles/tools/seg_bench.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "sound-event-detection_amd"))
import sed_amd  # noqa: E402

dev = torch.device("cuda", 0)
sr, S, n, cap, E = 32000, 512, 320000, 96000, 1024
x = (0.1 * torch.randn((S, n), device=dev)).contiguous()
fe = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat(), layout="btm")
rng = np.random.default_rng(0)


def timed(fn, reps=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def events(first_sample_mod4, seconds):
    out = []
    for e in range(E):
        start = int(rng.integers(0, (n - int(seconds * sr) - 8) // 4)) * 4 + first_sample_mod4
        out.append((e % S, (start + 0.25) / sr, (start + 0.25) / sr + seconds))
    return out


dense = torch.randn((E, cap), device=dev) * 0.1
print(f"dense batch ({E} x {cap}): {timed(lambda: fe(dense)):.1f} us")
for mod in (0, 1, 2):
    for secs in (3.0, 1.5, 1.0, 0.5):
        ev = events(mod, secs)
        t = timed(lambda: sed_amd.logmel_of_events(fe, x, sr, ev, cap))
        idx, a, length = sed_amd.handoff._event_ranges(ev, sr, n)
        d_off = torch.from_numpy(idx * n + a).to(dev)
        d_len = torch.from_numpy(np.minimum(length, cap).astype(np.int32)).to(dev)
        tk = timed(lambda: fe.forward_segments(x, d_off, d_len, cap))
        clips, _ = sed_amd.cut_events(x, sr, ev, cap=cap)
        same = torch.equal(sed_amd.logmel_of_events(fe, x, sr, ev, cap)[0], fe(clips))
        print(f"segments start = {mod} mod 4 samples, {secs} s of {cap / sr} s rows: {t:.1f} us with host prep and upload, {tk:.1f} us kernel alone, same bits {same}")

# every event framed as a file of its own (padding rule at its own end, its own frames, zeros behind them; segment tables on the device, so the frame counts are a few small launches), and the
# images stretched from each event's own rows
for secs in (3.0, 1.5, 0.5):
    ev = events(0, secs)
    idx, a, length = sed_amd.handoff._event_ranges(ev, sr, n)
    d_off = torch.from_numpy(idx * n + a).to(dev)
    d_len = torch.from_numpy(np.minimum(length, cap).astype(np.int32)).to(dev)
    tk = timed(lambda: fe.forward_segments(x, d_off, d_len, cap, own_edges=True))
    fe4 = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat())
    y, _, frames = fe4.forward_segments(x, d_off, d_len, cap, own_edges=True)
    ti = timed(lambda: sed_amd.reshape_wav2img(y, 256, 4, frames=frames))
    td = timed(lambda: sed_amd.reshape_wav2img(y, 256, 4))
    print(f"events as files of their own, {secs} s: {tk:.1f} us (kernel + frame counts), image from own rows {ti:.1f} us (dense image {td:.1f} us)")

# the Audio-Mamba hand-off at the rate of the reference's files: Kaldi filterbank of 1024 events (0.5 .. 3 s) out of 128
# recordings of 10 s at 44.1 kHz, the filterbank kernel reading the events in place against cut + wav2fbank
sr44, S44 = 44100, 128
x44 = (0.1 * torch.randn((S44, 10 * sr44), device=dev)).contiguous()
ev44 = []
for e in range(1024):
    secs = float(rng.uniform(0.5, 3.0))
    a = float(rng.uniform(0.0, 10.0 - secs))
    ev44.append((e % S44, a, a + secs))
for fused in (True, False):
    t = timed(lambda: sed_amd.fbank_of_events(x44, sr44, ev44, melbins=128, target_length=1024, norm_mean=-4.27, norm_std=4.57,
                                              fused=fused), reps=20)
    print(f"Audio-Mamba input of 1024 events (0.5 .. 3 s) at 44.1 kHz, {'filterbank kernel on the events in place' if fused else 'cut + wav2fbank'}: {t:.0f} us")
