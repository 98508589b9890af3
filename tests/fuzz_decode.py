#!/usr/bin/env python3
"""Random sweep of the HIP decoders against the CPU oracle (run by name on a GPU box; not collected by pytest):
    python3 tests/fuzz_decode.py [cases] [seed]

Every case draws a batch shape (files, frames from 1 up to beyond the one-launch limit, classes), a probability field with
runs of random length (values placed on and next to the threshold), and checks, bit for bit / event for event:
  * the CRNN chain: gaussian smoothing + threshold + region decode (+ gap merge at a random gap) against the oracle's
    smooth_and_threshold -> decode_predictions -> merge_close_events,
  * the detector decode (median filter, threshold, onset / offset rules) per stream,
  * the single-stage YAMNet and HTS-AT argmax decoders and the clip label,
  * the multi-GPU tail (sharding.decode_and_exchange on one rank): records with caps that overflow, traces on either side
    of the one-launch limit.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sound-event-detection_amd"))

from oracle import decode as od  # noqa: E402
import sed_amd  # noqa: E402

CLASSES = ["cough", "dog_bark", "gun_shot", "siren", "car_horn"]
NAMES6 = CLASSES + ["non_event"]


def field(rng, N, T, C, thr=0.5):
    """Piecewise-constant probabilities with noise; a share of the values exactly on the threshold or one ulp off it."""
    p = np.empty((N, T, C), np.float32)
    for n in range(N):
        for c in range(C):
            t = 0
            while t < T:
                run = int(rng.integers(1, max(2, min(T, 60))))
                level = float(rng.choice([0.05, 0.3, 0.49, 0.51, 0.7, 0.95]))
                p[n, t:t + run, c] = level
                t += run
    p += (rng.standard_normal(p.shape) * rng.choice([0.0, 0.01, 0.1])).astype(np.float32)
    p = np.clip(p, 0.0, 1.0).astype(np.float32)
    edge = rng.random(p.shape) < 0.02
    p[edge] = rng.choice(np.array([thr, np.nextafter(np.float32(thr), np.float32(1)), np.nextafter(np.float32(thr), np.float32(0))],
                                  np.float32), size=int(edge.sum()))
    return p


def crnn_case(rng, case):
    N, C = int(rng.integers(1, 9)), int(rng.integers(1, 9))
    T = int(rng.choice([1, 2, 3, 12, 13, 25, 26, int(rng.integers(4, 900)), 862, int(rng.integers(900, 9000))]))
    fps = float(rng.choice([43.0, 43.06640625, 50.0, 100.0, 31.25]))
    gap = float(rng.choice([0.0, 0.1, 0.3, 0.5, 0.6, 1.0]))
    desc = dict(case=case, kind="crnn", N=N, T=T, C=C, fps=fps, gap=gap)
    probs = field(rng, N, T, C)
    names = [f"f{n}.wav" for n in range(N)]
    labels = {k: f"class{k}" for k in range(C)}
    want = od.decode_predictions(od.smooth_and_threshold(probs), names, labels, fps)
    got = sed_amd.decode_probabilities(probs, names, labels, fps)
    assert got == want, "decode_probabilities"
    act = od.smooth_and_threshold(probs)
    assert sed_amd.decode_predictions(act, names, labels, fps) == want, "decode_predictions"
    wantm = {f: od.merge_close_events(ev, gap) for f, ev in want.items()}
    assert sed_amd.merge_close_events_batch(want, gap) == wantm, "merge_close_events_batch"
    f0 = names[int(rng.integers(N))]
    assert sed_amd.merge_close_events(want[f0], gap) == wantm[f0], "merge_close_events"
    return desc, sum(len(v) for v in want.values())


def detector_case(rng, case):
    S = int(rng.integers(1, 40))
    N = int(rng.choice([1, 2, 3, 4, 20, 41, int(rng.integers(5, 300))]))
    med = bool(rng.random() < 0.7)
    thr = float(rng.choice([0.5, 0.3, 0.8]))
    desc = dict(case=case, kind="detector", S=S, N=N, med=med, thr=thr)
    preds = field(rng, S, N, 1, thr)[..., 0]
    got = sed_amd.detect_onsets_offsets_batch(preds, thr, 0.48, med)
    n_ev = 0
    for s in range(S):
        on, off = od.detect_onsets_offsets(preds[s], thr, 0.48, med)
        g_on, g_off = got[s]
        assert np.array_equal(np.asarray(g_on), on) and np.array_equal(np.asarray(g_off), off), f"stream {s}"
        n_ev += len(on)
    on1, off1 = sed_amd.detect_onsets_offsets(preds[0], thr, 0.48, med)
    w_on, w_off = od.detect_onsets_offsets(preds[0], thr, 0.48, med)
    assert np.array_equal(np.asarray(on1), w_on) and np.array_equal(np.asarray(off1), w_off), "single stream"
    return desc, n_ev


def argmax_case(rng, case):
    B = int(rng.integers(1, 6))
    T = int(rng.choice([1, 2, 41, 20, 4096, int(rng.integers(3, 6000))]))
    desc = dict(case=case, kind="argmax", B=B, T=T)
    probs = field(rng, B, T, 6)
    files = [f"g{b}.wav" for b in range(B)]
    want_y = [od.yamnet_frame_events(probs[b], files[b], CLASSES, 0.48, med_filter=True) for b in range(B)]
    got_y = sed_amd.predict_events_batch(probs, files, CLASSES, 0.48, med_filter=True)
    assert got_y == want_y, "predict_events_batch (median)"
    want_n = [od.yamnet_frame_events(probs[b], files[b], CLASSES, 0.48, med_filter=False) for b in range(B)]
    assert sed_amd.predict_events_batch(probs, files, CLASSES, 0.48, med_filter=False) == want_n, "predict_events_batch"
    for b in range(B):
        assert sed_amd.get_classification_label(torch.from_numpy(probs[b])) == NAMES6[od.htsat_clip_label(probs[b])], "clip label"
    want_h = od.htsat_events_from_batch(files, probs, NAMES6, 5, 20.0)
    got_h = sed_amd.get_events_from_batch({"file": files, "pred_map": torch.from_numpy(probs)},
                                          type("Cfg", (), {"clip_duration": 20.0})())
    assert got_h == want_h, "get_events_from_batch"
    return desc, sum(len(v) for v in want_y)


def records_case(rng, case):
    """The multi-GPU tail in one call: smooth, threshold, run merge on a time table, records of (stream, class, onset frame,
    offset frame) with a cap that may overflow, against the oracle's chain of the same steps."""
    S, C = int(rng.integers(1, 40)), int(rng.integers(1, 7))
    T = int(rng.choice([1, 2, 3, 20, 41, 510, 511, 862, int(rng.integers(4, 3000)), int(rng.integers(8000, 9500))]))
    cap = int(rng.choice([1, 4, 16, 64, T // 2 + 1]))
    smoothing = [None, "median3", "gauss"][int(rng.integers(3))]
    hop = float(rng.choice([0.48, 0.02, 1.0 / 43.06640625]))
    gap = float(rng.choice([0.0, 0.3, 0.5, 0.6, 1.0])) * (1.0 if hop > 0.1 else 0.1)
    first = int(rng.integers(0, 5000))
    desc = dict(case=case, kind="records", S=S, T=T, C=C, cap=cap, smoothing=smoothing, hop=round(hop, 4), gap=gap)
    probs = field(rng, S, T, C)
    table_host = [k * hop for k in range(T + 2)]
    dev = torch.device("cuda", 0)
    table = torch.tensor(table_host, dtype=torch.float64, device=dev)
    got = sed_amd.sharding.decode_and_exchange(torch.from_numpy(probs).to(dev), table, gap, cap=cap, first_stream=first,
                                               threshold=0.5, smoothing=smoothing).cpu().numpy()
    if smoothing == "gauss":
        act = od.smooth_and_threshold(probs)
    else:
        sm = od.median3_time(probs, 1) if smoothing == "median3" else probs
        act = (sm > np.float32(0.5)).astype(int)
    index = {t: k for k, t in enumerate(table_host)}
    cap_in = T // 2 + 1
    runs = np.zeros((S, C, cap_in, 2), np.int32)
    counts = np.zeros((S, C), np.int32)
    for s_ in range(S):
        for c in range(C):
            ev = [{"file": "f", "event_onset": table_host[a], "event_offset": table_host[b], "event_label": "e"}
                  for a, b in od.contiguous_regions(act[s_, :, c])]
            merged = od.merge_close_events(ev, gap)
            counts[s_, c] = len(merged)
            for k, m in enumerate(merged):
                runs[s_, c, k] = (index[m["event_onset"]], index[m["event_offset"]])
    want = od.pack_events(runs, counts, cap, first_stream=first)
    assert got.shape == want.shape and np.array_equal(got, want), "records"
    return desc, int(counts.sum())


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    assert torch.cuda.is_available()
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(cases):
        fn = (crnn_case, detector_case, argmax_case, records_case)[case % 4]
        try:
            desc, n_ev = fn(rng, case)
            print(f"ok   {desc} events {n_ev}", flush=True)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print(f"FAIL case {case} {fn.__name__}: {type(e).__name__}: {e}", flush=True)
    print(f"{cases - bad} of {cases} cases passed")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
