"""Differential check of two builds of libsedhip.so: the same seeded front-end cases through the library named by
SED_HIP_LIB, one sha256 per case.  Run once per build and diff the outputs:
  SED_HIP_LIB=old.so python profiles/tools/fuzz_ab.py > a.txt ; python profiles/tools/fuzz_ab.py > b.txt ; diff a.txt b.txt
Cases stress the staging paths: odd lengths, rows that are only sample-aligned, strided batches, int16, clips made of
edge groups only, every flavour."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "sound-event-detection_amd"))
import sed_amd  # noqa: E402

dev = torch.device("cuda", 0)
rng = np.random.default_rng(2024)
F = sed_amd.FrontendConfig


def digest(t):
    return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:16]


cases = []
for name, cfg, sr in (("htsat", F.htsat(), 32000), ("yamnet", F.yamnet(), 16000), ("crnn", F.crnn(), 44100),
                      ("dbmel", F.db_mel(), 44100), ("kaldi", F.kaldi_fbank(16000.0, 128), 16000)):
    for B, L in ((1, 1025), (3, 4099), (2, 33333), (5, 64001), (16, 2 * sr + 7), (64, sr)):
        cases.append((name, cfg, B, L, "f32", 0, 0))
    cases.append((name, cfg, 4, 50001, "f32", 1, 0))      # rows start one sample off a 16-byte boundary
    cases.append((name, cfg, 4, 50000, "f32", 0, 13))     # row stride 13 samples longer than the clip
    cases.append((name, cfg, 4, 50002, "f32", 3, 5))
for B, L in ((2, 20000), (7, 64001), (32, 32000)):
    cases.append(("htsat_i16", F.htsat(pcm_kind=1), B, L, "i16", 0, 0))
    cases.append(("htsat_i16", F.htsat(pcm_kind=1), B, L, "i16", 1, 3))

fes = {}
for name, cfg, B, L, kind, shift, extra in cases:
    fe = fes.get(name)
    if fe is None:
        fe = fes[name] = sed_amd.LogMelFrontend(cfg, layout="btm")
    if kind == "f32":
        base = torch.from_numpy((0.2 * rng.standard_normal((B, L + extra + 8))).astype(np.float32)).to(dev)
    else:
        base = torch.from_numpy(rng.integers(-20000, 20000, (B, L + extra + 8), dtype=np.int16)).to(dev)
    x = base[:, shift:shift + L]                           # a view: stride L + extra + 8, first sample `shift` in
    try:
        y = fe(x)
        print(name, B, L, kind, shift, extra, tuple(y.shape), digest(y))
    except Exception as e:                                 # noqa: BLE001
        print(name, B, L, kind, shift, extra, "raises", type(e).__name__)
