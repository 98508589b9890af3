"""CPU oracle for the frame-level event decoders.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package never does.

Each function restates one reference decoder (file:line in its docstring) in
plain numpy / Python.  Pinned by tests/golden/ fixtures generated from the
reference's own code and committed result files (tests/golden/gen_golden.py).
"""
from __future__ import annotations

import numpy as np

# scipy.ndimage.gaussian_filter1d(sigma=3): radius int(4*3+0.5)=12; centre -> edge, fp64.
GAUSS_SIGMA3_HALF = [float.fromhex(h) for h in (
    "0x1.105a329f98197p-3", "0x1.01a25f86eb137p-3", "0x1.b42a57d56c0bep-4", "0x1.4a614d1afd337p-4",
    "0x1.bfde9c12bec92p-5", "0x1.0fa58939b528fp-5", "0x1.26defcaeb0202p-6", "0x1.1e6bccad344bap-7",
    "0x1.f1e9915139406p-9", "0x1.8345966f69518p-10", "0x1.0d8a5ad43c165p-11", "0x1.4fbe39149e277p-13",
    "0x1.763a210dfb306p-15")]


def gaussian_half_kernel(sigma: float = 3.0, truncate: float = 4.0):
    """scipy _gaussian_kernel1d: exp(-0.5/sigma^2 * x^2) / sum, returned centre -> edge."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    phi = phi / phi.sum()
    return phi[radius:].tolist()


def gaussian_smooth_time(probs: np.ndarray, half=None) -> np.ndarray:
    """gaussian_filter1d(probs, sigma=3, axis=1, mode='nearest') as used by
    src/models/crnn/crnn.py:418-423 and src/models/htsat/combined.py:26-31.

    scipy's symmetric correlate1d: per output sample, fp64:
      acc = x[0]*w[0]; for j = R..1: acc += (x[-j] + x[+j]) * w[j]; edge-replicated; cast to input dtype.
    probs: (N, T, C); returns same dtype.
    """
    half = GAUSS_SIGMA3_HALF if half is None else half
    R = len(half) - 1
    x = np.asarray(probs)
    xd = x.astype(np.float64)
    T = x.shape[1]
    idx = np.clip(np.arange(-R, T + R), 0, T - 1)
    xp = xd[:, idx, :]                       # (N, T+2R, C) edge replicated
    acc = xp[:, R:R + T, :] * half[0]
    for j in range(R, 0, -1):
        acc = acc + (xp[:, R - j:R - j + T, :] + xp[:, R + j:R + j + T, :]) * half[j]
    return acc.astype(x.dtype)


def median3_time(x: np.ndarray, axis: int = 0) -> np.ndarray:
    """scipy.ndimage.median_filter(size=3 along `axis`, mode='reflect'): edge sample duplicated."""
    x = np.asarray(x)
    xm = np.moveaxis(x, axis, 0)
    prev = np.concatenate([xm[:1], xm[:-1]], axis=0)
    nxt = np.concatenate([xm[1:], xm[-1:]], axis=0)
    med = np.maximum(np.minimum(prev, xm), np.minimum(np.maximum(prev, xm), nxt))
    return np.moveaxis(med, 0, axis)


# ---------------------------------------------------------------------------- D1


def detect_onsets_offsets(preds, threshold=0.5, hop_time=0.48, med_filter=True):
    """src/models/yamnet_train.py:55-71.

    Edges are taken on the UNPADDED activity trace, so a clip that starts active has no onset and a
    clip that ends active has no offset; the length fix-up then appends (N+1)*hop or prepends 0 --
    and does nothing when both happen (reversed pair) or when the whole clip is active (no events).
    """
    p = np.asarray(preds)
    if med_filter:
        p = median3_time(p, 0)
    act = (p > threshold).astype(np.int64)
    d = act[1:] - act[:-1]
    on = (np.flatnonzero(d == 1) + 1) * hop_time
    off = (np.flatnonzero(d == -1) + 1) * hop_time
    if off.size < on.size:
        off = np.concatenate([off, [(p.shape[0] + 1) * hop_time]])
    elif off.size > on.size:
        on = np.concatenate([[0], on])
    return on, off


# ---------------------------------------------------------------------------- D2


def smooth_and_threshold(probs, posterior_thresh=0.5):
    """src/models/crnn/crnn.py:418-425: gaussian sigma=3 along time then strict > threshold, as int."""
    sm = gaussian_smooth_time(np.asarray(probs))
    return (sm > posterior_thresh).astype(int)


# ---------------------------------------------------------------------------- D3


def contiguous_regions(activity):
    """src/models/crnn/utils.py:77-89: (start, end_exclusive) of runs of ones, zero-padded both sides."""
    a = np.concatenate([[0], np.asarray(activity), [0]])
    d = a[1:] - a[:-1]
    return list(zip(np.flatnonzero(d == 1), np.flatnonzero(d == -1)))   # np.int64 indices, as the reference's np.where


def decode_predictions(preds, filenames, class_labels, frames_per_sec):
    """src/models/crnn/utils.py:91-134: class-major, onset-ascending; times round(k * (1/fps), 3) with k an np.int64,
    so the product is an np.float64 and round() is numpy's rint(x * 1000) / 1000 (it differs from Python's
    exact-decimal rounding of a Python float for some frames, e.g. frame 57 at 2000 frames/s)."""
    out = {}
    res = 1.0 / frames_per_sec
    for i in range(min(len(preds), len(filenames))):
        name = filenames[i]
        if isinstance(name, np.ndarray):
            name = name.item()
        name = str(name)
        ev = []
        for cidx, label in class_labels.items():
            for s, e in contiguous_regions(np.asarray(preds[i])[:, cidx]):
                ev.append({"file": name, "event_onset": round(s * res, 3),
                           "event_offset": round(e * res, 3), "event_label": label})
        out[name] = ev
    return out


# ---------------------------------------------------------------------------- D4


def merge_close_events(events, threshold=0.5):
    """src/utils/merge_events.py:4-60.

    Stable sort by onset; group by label in first-appearance order; within a label fold the next
    event into the current one when next.onset - cur.offset <= threshold (offset = max of both);
    concatenate groups; stable sort by onset again.
    """
    if not events:
        return []
    order = sorted(range(len(events)), key=lambda i: events[i]["event_onset"])
    groups = {}
    for i in order:
        groups.setdefault(events[i]["event_label"], []).append(i)
    merged = []
    for label, idxs in groups.items():
        cur = dict(events[idxs[0]])
        for i in idxs[1:]:
            nxt = events[i]
            if nxt["event_onset"] - cur["event_offset"] <= threshold:
                if nxt["event_offset"] > cur["event_offset"]:
                    cur["event_offset"] = nxt["event_offset"]
            else:
                merged.append(cur)
                cur = dict(nxt)
        merged.append(cur)
    merged.sort(key=lambda e: e["event_onset"])
    return merged


# ---------------------------------------------------------------------------- D5


def htsat_events_from_batch(files, pred_map, class_names, non_event_index, clip_duration):
    """src/models/htsat/combined.py:17-73.

    Gaussian smoothing along time, per-frame argmax, run-length over labels; the loop index is 1-based so
    every interior boundary is reported one frame late; the last offset is num_frame/num_frame*duration;
    only element 0 of the batch is decoded (the reference returns inside the batch loop).
    """
    pm = np.asarray(pred_map)
    n_frame = pm.shape[1]
    sm = gaussian_smooth_time(pm)
    labels = np.argmax(sm, axis=2)
    rec = labels[0]
    name = files[0]
    evs = []
    onset = 0
    cur = rec[0]
    i = 0
    for i, lab in enumerate(rec, start=1):
        if lab != cur:
            if cur != non_event_index:
                evs.append({"file": name, "event_onset": onset / n_frame * clip_duration,
                            "event_offset": i / n_frame * clip_duration, "event_label": class_names[int(cur)]})
            onset = i
            cur = lab
    if cur != non_event_index:
        evs.append({"file": name, "event_onset": onset / n_frame * clip_duration,
                    "event_offset": i / n_frame * clip_duration, "event_label": class_names[int(cur)]})
    return {name: evs}


# ---------------------------------------------------------------------------- D6


def yamnet_frame_events(frame_probs, filename, classes, frame_hop_sec=0.48, med_filter=False):
    """src/models/single_stage_yamnet_frame.py:161-209: median (3,1) over time, argmax, run-length;
    class 0 is background; label = classes[c-1]; times start*hop, i*hop, final n_frames*hop."""
    fp = np.asarray(frame_probs)
    if med_filter:
        fp = median3_time(fp, 0)
    cls = np.argmax(fp, axis=-1)
    n = cls.shape[0]
    evs = []
    cur = cls[0]
    start = 0
    for i in range(1, n):
        if cls[i] != cur:
            if cur != 0:
                evs.append({"file": filename, "event_onset": start * frame_hop_sec,
                            "event_offset": i * frame_hop_sec, "event_label": classes[int(cur) - 1]})
            cur = cls[i]
            start = i
    if cur != 0:
        evs.append({"file": filename, "event_onset": start * frame_hop_sec,
                    "event_offset": n * frame_hop_sec, "event_label": classes[int(cur) - 1]})
    return evs


# ---------------------------------------------------------------------------- D7


def htsat_clip_label(pred_map):
    """src/models/htsat/classification.py:20-26: drop the last class, per-frame argmax, mode over frames
    (torch.unique sorts ascending, argmax returns the first maximum -> smallest class id wins ties)."""
    pm = np.asarray(pred_map)[:, :-1]
    lab = np.argmax(pm, axis=1)
    counts = np.bincount(lab, minlength=pm.shape[1])
    return int(np.argmax(counts))


def htsat_classification_events(frame_predictions, events_list, class_names):
    """src/models/htsat/classification.py:15-36: per clip of every batch the clip label (htsat_clip_label), written as
    "event_label" into the first event of the clip's file whose "extracted_audio_filename" equals the base name of the clip's
    path.  Works on a deep copy."""
    import copy
    import os
    out = copy.deepcopy(events_list)
    for batch in frame_predictions:
        for i in range(len(batch["file"])):
            filename = batch["file"][i]
            extracted = os.path.basename(batch["file_path"][i])
            label = class_names[htsat_clip_label(np.asarray(batch["pred_map"][i]))]
            for ev in out[filename]:
                if ev["extracted_audio_filename"] == extracted:
                    ev["event_label"] = label
                    break
    return out


def pack_events(runs, counts, cap, first_stream=0):
    """Event records of a stream batch as the GPUs exchange them: runs (S, C, cap_in, 2) int, counts (S, C) ->
    int32 (S, cap + 1, 4).  Row 0 of a stream = (records kept, records dropped, 0, 0); rows 1.. = (stream +
    first_stream, class, onset frame, offset frame) in class-major, onset-ascending order, the order in which
    decode_predictions (src/models/crnn/utils.py:99-134) walks a file's classes and regions; unused rows are zero."""
    runs = np.asarray(runs)
    counts = np.asarray(counts)
    S, C, cap_in, _ = runs.shape
    out = np.zeros((S, cap + 1, 4), np.int32)
    for s in range(S):
        recs = [(s + first_stream, c, int(runs[s, c, k, 0]), int(runs[s, c, k, 1]))
                for c in range(C) for k in range(min(max(int(counts[s, c]), 0), cap_in))]
        out[s, 0, 0] = min(len(recs), cap)
        out[s, 0, 1] = max(0, len(recs) - cap)
        for i, r in enumerate(recs[:cap]):
            out[s, 1 + i] = r
    return out
