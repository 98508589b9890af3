#!/usr/bin/env python3
"""Writes tests/golden/frontend_golden.npz: for every front-end flavour, the fp64 ("gold") and fp32-reference-formulation
("ref32") outputs of oracle/frontend.py on seeded synthetic clips (oracle.frontend.synth_clip: the inputs are
regenerated from the seeds, only the outputs are stored, as float32).

The reference holds no front-end arithmetic of its own (librosa / torchlibrosa / TensorFlow / torchaudio do it and none
of them is installed here), so these vectors do not pin the oracle against the reference: they freeze the oracle, so
that the oracle and the HIP path cannot drift together unnoticed.

  python tests/golden/gen_frontend_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import frontend as of  # noqa: E402

CASES = {
    # name: (first clip seed, clips, samples, sample rate)
    "htsat": (900, 2, 32000, 32000),
    "yamnet": (910, 2, 31200, 16000),
    "crnn": (920, 2, 44100, 44100),
    "db_mel": (930, 2, 44100, 44100),
    "kaldi16k": (940, 2, 16000, 16000),
    "kaldi441": (950, 1, 22050, 44100),
}


def inputs(name):
    first, n, L, sr = CASES[name]
    x = of.synth_batch(n, L, sr, first=first)
    if name.startswith("kaldi"):
        x = x + np.float32(0.02)          # a DC offset for remove_dc_offset to remove
    return x, sr


def outputs(name, mode):
    x, sr = inputs(name)
    if name == "htsat":
        return of.htsat_logmel(x, mode=mode)[:, 0]
    if name == "yamnet":
        return of.yamnet_logmel(x, mode=mode)[0]
    if name == "crnn":
        return of.crnn_extract_mbe(x, mode=mode)
    if name == "db_mel":
        return of.librosa_db_mel(x, mode=mode)
    return of.kaldi_fbank(x, float(sr), 128, mode=mode)


def main():
    out = {}
    for name in CASES:
        for mode in ("gold", "ref32"):
            out[f"{name}_{mode}"] = np.asarray(outputs(name, mode)).astype(np.float32)
            print(name, mode, out[f"{name}_{mode}"].shape)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "frontend_golden.npz"), **out)


if __name__ == "__main__":
    main()
