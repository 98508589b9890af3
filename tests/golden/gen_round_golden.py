#!/usr/bin/env python3
"""Golden vectors for the time rounding of decode_predictions at frame rates other than the two the result files use.

Runs ONLY in the build container (imports src/models/crnn/utils.py of the reference); writes d3_rounding.npz next to
this script.  The reference multiplies an np.int64 frame index by 1/frames_per_sec and rounds the np.float64 with
round(x, 3), which is numpy's rint(x * 1000) / 1000, not Python's exact-decimal rounding: the two differ for some
frames (frame 57 at 2000 frames/s: 0.028 against 0.029)."""
import os
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

from src.models.crnn.utils import decode_predictions as ref_decode  # noqa: E402

FPS = [2000.0, 1000.0 / 3.0, 250.0, 86.1328125, 43.06640625, 43.0, 31.25, 7.0]
T = 1024


def main():
    act = np.zeros((1, T, 1), dtype=int)
    act[0, 0::2, 0] = 1                                   # events [k, k + 1) for every even k: every frame index is a boundary
    pack = {"fps": np.asarray(FPS, np.float64), "T": T}
    n_diff = 0
    for i, fps in enumerate(FPS):
        ev = ref_decode(act, ["f.wav"], {0: "e"}, fps)["f.wav"]
        assert len(ev) == T // 2
        on = np.asarray([e["event_onset"] for e in ev], np.float64)
        off = np.asarray([e["event_offset"] for e in ev], np.float64)
        pack[f"on_{i}"], pack[f"off_{i}"] = on, off
        res = 1.0 / fps
        n_diff += sum(round(int(k) * res, 3) != float(v) for k, v in zip(range(0, T, 2), on))
    np.savez_compressed(os.path.join(OUT, "d3_rounding.npz"), **pack)
    print("written; frames where Python's round of a Python float differs from the reference:", n_diff)


if __name__ == "__main__":
    main()
