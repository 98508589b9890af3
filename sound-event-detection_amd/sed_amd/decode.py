"""Frame-level event decoders running on the HIP kernels of libsedhip.so.

Same names, argument meaning and output format as the reference functions they replace:

  detect_onsets_offsets      src/models/yamnet_train.py:55-71
  decode_predictions         src/models/crnn/utils.py:91-134   (binary activity -> events)
  decode_probabilities       src/models/crnn/crnn.py:418-425 + :454-459 fused (smooth, threshold, decode)
  merge_close_events         src/utils/merge_events.py:4-60
  process_json_file          src/utils/merge_events.py:63-87
  get_events_from_batch      src/models/htsat/combined.py:17-73
  predict_events             src/models/single_stage_yamnet_frame.py:185-209
  get_classification_label   src/models/htsat/classification.py:20-26

The kernels return integer frame indices (and, for the gap merge, select among the caller's fp64 times);
every time value is formed on the host with the reference's own Python expression, so event lists are
bit-identical given identical frame probabilities.  There is no CPU path.
"""
from __future__ import annotations

import itertools
import json
import os
from typing import Any, Dict, List, Sequence

import numpy as np
import torch

from . import _lib as L

CLASSES = ["cough", "dog_bark", "gun_shot", "siren", "car_horn"]           # src/config.py:4
NON_EVENT_LABEL = "non_event"                                               # src/models/htsat/constants.py:8
_SMOOTH = {None: L.SMOOTH_NONE, "none": L.SMOOTH_NONE, "median3": L.SMOOTH_MEDIAN3, "gauss": L.SMOOTH_GAUSS25}


def _smooth_code(smoothing) -> int:
    try:
        return _SMOOTH[smoothing]
    except (KeyError, TypeError):
        raise ValueError(f"smoothing must be one of None, 'none', 'median3', 'gauss', not {smoothing!r}") from None


def _check_table(time_table, device) -> None:
    """The kernels read the time table as a contiguous device array of fp64 on the device of the data."""
    if not torch.is_tensor(time_table) or time_table.dtype != torch.float64:
        raise TypeError("time_table must be a torch.float64 tensor")
    if time_table.device != device:
        raise ValueError(f"time_table is on {time_table.device}, the data on {device}")
    if time_table.dim() != 1 or not time_table.is_contiguous() or time_table.numel() < 1:
        raise ValueError("time_table must be a contiguous 1-D tensor with at least one entry")


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("no ROCm device is visible; the decoders have no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def _to_dev(x, dtype) -> torch.Tensor:
    t = torch.as_tensor(x)
    if not t.is_cuda:
        t = t.to(_device())
    return t.to(dtype).contiguous()


# ---- kernel wrappers (device tensors in, device tensors out, no host sync) ---------------------------------


def threshold_runs(probs: torch.Tensor, threshold: float, smoothing=None):
    """probs (B, T, C) fp32 on device -> (runs int32 (B, C, cap, 2), counts int32 (B, C)); cap = T//2 + 1."""
    B, T, Cn = probs.shape
    cap = T // 2 + 1
    runs = torch.empty((B, Cn, cap, 2), dtype=torch.int32, device=probs.device)
    counts = torch.empty((B, Cn), dtype=torch.int32, device=probs.device)   # the kernel writes every count
    with torch.cuda.device(probs.device):
        L.check(L.lib.sed_threshold_runs(probs.data_ptr(), B, T, Cn, float(np.float32(threshold)), _smooth_code(smoothing),
                                         runs.data_ptr(), counts.data_ptr(), cap, L.stream_ptr(probs.device)),
                "sed_threshold_runs")
    return runs, counts


def activity_edges(act: torch.Tensor):
    """act (B, T, C) int32 on device -> starts, ends (B, C, cap), n_starts, n_ends (B, C); cap = T + 1."""
    B, T, Cn = act.shape
    cap = T + 1
    starts = torch.empty((B, Cn, cap), dtype=torch.int32, device=act.device)
    ends = torch.empty_like(starts)
    ns = torch.zeros((B, Cn), dtype=torch.int32, device=act.device)
    ne = torch.zeros_like(ns)
    with torch.cuda.device(act.device):
        L.check(L.lib.sed_activity_edges(act.data_ptr(), B, T, Cn, starts.data_ptr(), ends.data_ptr(), ns.data_ptr(),
                                         ne.data_ptr(), cap, L.stream_ptr(act.device)), "sed_activity_edges")
    return starts, ends, ns, ne


def argmax_runs(probs: torch.Tensor, smoothing=None, drop_last_classes: int = 0):
    """probs (B, T, C) fp32 on device -> (runs int32 (B, T, 3) = (label, start, end), counts int32 (B,))."""
    B, T, Cn = probs.shape
    cap = max(T, 1)
    runs = torch.empty((B, cap, 3), dtype=torch.int32, device=probs.device)
    counts = torch.zeros((B,), dtype=torch.int32, device=probs.device)
    with torch.cuda.device(probs.device):
        L.check(L.lib.sed_argmax_runs(probs.data_ptr(), B, T, Cn, _smooth_code(smoothing), drop_last_classes,
                                      runs.data_ptr(), counts.data_ptr(), cap, L.stream_ptr(probs.device)),
                "sed_argmax_runs")
    return runs, counts


def argmax_histogram(probs: torch.Tensor, drop_last_classes: int = 0) -> torch.Tensor:
    B, T, Cn = probs.shape
    hist = torch.zeros((B, Cn - drop_last_classes), dtype=torch.int32, device=probs.device)
    with torch.cuda.device(probs.device):
        L.check(L.lib.sed_argmax_histogram(probs.data_ptr(), B, T, Cn, drop_last_classes, hist.data_ptr(),
                                           L.stream_ptr(probs.device)), "sed_argmax_histogram")
    return hist


def merge_runs(runs: torch.Tensor, counts: torch.Tensor, time_table: torch.Tensor, gap: float):
    """Gap-merge frame-index runs (.., cap, 2) group by group using fp64 times time_table[frame]."""
    _check_table(time_table, runs.device)
    if runs.dtype != torch.int32 or counts.dtype != torch.int32 or counts.device != runs.device:
        raise TypeError("runs and counts must be int32 tensors on one device")
    runs, counts = runs.contiguous(), counts.contiguous()
    cap = runs.shape[-2]
    n_groups = counts.numel()
    out_runs = torch.empty_like(runs)
    out_counts = torch.empty_like(counts)                                   # the kernel writes every count
    with torch.cuda.device(runs.device):
        L.check(L.lib.sed_merge_runs(runs.data_ptr(), counts.data_ptr(), n_groups, cap, time_table.data_ptr(),
                                     time_table.numel(), float(gap), out_runs.data_ptr(), out_counts.data_ptr(),
                                     L.stream_ptr(runs.device)), "sed_merge_runs")
    return out_runs, out_counts


def threshold_merge_runs(probs: torch.Tensor, threshold: float, time_table: torch.Tensor, gap: float, smoothing=None):
    """threshold_runs + merge_runs in one launch: probs (B, T, C) -> packed int32 (B, C, 1 + 2*cap), cap = T//2 + 1;
    packed[b, c, 0] = merged count, then (start, end) frame pairs.  T <= 8190."""
    if not probs.is_cuda or probs.dim() != 3 or probs.dtype != torch.float32:
        raise ValueError("probs must be an fp32 (B, T, C) tensor on a ROCm device")
    _check_table(time_table, probs.device)
    probs = probs.contiguous()
    B, T, Cn = probs.shape
    cap = T // 2 + 1
    packed = torch.empty((B, Cn, 1 + 2 * cap), dtype=torch.int32, device=probs.device)
    with torch.cuda.device(probs.device):
        L.check(L.lib.sed_threshold_merge_runs(probs.data_ptr(), B, T, Cn, float(np.float32(threshold)),
                                               _smooth_code(smoothing), time_table.data_ptr(), time_table.numel(),
                                               float(gap), packed.data_ptr(), cap, L.stream_ptr(probs.device)),
                "sed_threshold_merge_runs")
    return packed


def threshold_merge_records(probs: torch.Tensor, threshold: float, time_table: torch.Tensor, gap: float, smoothing=None,
                            cap: int = 64, first_stream: int = 0):
    """threshold_merge_runs and the record packer of the multi-GPU exchange in one launch: probs (S, T, C) fp32 on the
    device -> int32 (S, cap + 1, 4) records (row 0 = kept, dropped; rows 1.. = stream + first_stream, class, onset
    frame, offset frame), the same values as sharding.pack_events(threshold_merge_runs(...), None, cap, first_stream)."""
    if not probs.is_cuda:
        raise RuntimeError("threshold_merge_records needs device tensors (there is no CPU path)")
    if probs.dim() != 3 or probs.dtype != torch.float32:
        raise ValueError("probs must be fp32 (S, T, C)")
    _check_table(time_table, probs.device)
    probs = probs.contiguous()
    S, T, Cn = probs.shape
    records = torch.empty((S, cap + 1, 4), dtype=torch.int32, device=probs.device)
    with torch.cuda.device(probs.device):
        L.check(L.lib.sed_threshold_merge_records(probs.data_ptr(), S, T, Cn, float(np.float32(threshold)),
                                                  _smooth_code(smoothing), time_table.data_ptr(), time_table.numel(),
                                                  float(gap), int(first_stream), records.data_ptr(), int(cap),
                                                  L.stream_ptr(probs.device)), "sed_threshold_merge_records")
    return records


def merge_event_groups(onsets: torch.Tensor, offsets: torch.Tensor, group_start: torch.Tensor, gap: float):
    """fp64 onsets/offsets (n,), group_start int32 (G+1,) -> merged onsets, offsets (n,), counts (G,)."""
    G = group_start.numel() - 1
    out_on = torch.empty_like(onsets)
    out_off = torch.empty_like(offsets)
    out_counts = torch.zeros((max(G, 0),), dtype=torch.int32, device=onsets.device)
    with torch.cuda.device(onsets.device):
        L.check(L.lib.sed_merge_events(onsets.data_ptr(), offsets.data_ptr(), group_start.data_ptr(), G, float(gap),
                                       out_on.data_ptr(), out_off.data_ptr(), out_counts.data_ptr(),
                                       L.stream_ptr(onsets.device)), "sed_merge_events")
    return out_on, out_off, out_counts


# ---- D1 --------------------------------------------------------------------------------------------------


def detect_onsets_offsets(preds, threshold=0.5, hop_time=0.48, med_filter=True):
    """Framewise probabilities (N,) -> (onsets, offsets) in seconds (fp64 arrays)."""
    on, off = detect_onsets_offsets_batch(torch.as_tensor(preds)[None], threshold, hop_time, med_filter)[0]
    return on, off


def detect_onsets_offsets_batch(preds, threshold=0.5, hop_time=0.48, med_filter=True):
    """(B, N) probabilities -> list of (onsets, offsets) per stream, each exactly what the reference returns."""
    p = _to_dev(preds, torch.float32)
    B, N = p.shape
    runs, counts = threshold_runs(p[:, :, None], threshold, "median3" if med_filter else None)
    runs, counts = runs[:, 0].cpu().numpy(), counts[:, 0].cpu().numpy()
    # All streams at once: the reference diffs the UNPADDED trace, so a run starting at frame 0 has no onset and one
    # ending at N no offset; times are frame * hop_time in fp64 (np.int64 * float, as the reference's array product).
    cap = runs.shape[1]
    live = np.arange(cap)[None, :] < counts[:, None]
    on_ok = live & (runs[:, :, 0] >= 1)
    off_ok = live & (runs[:, :, 1] <= N - 1)
    on_t = runs[:, :, 0][on_ok].astype(np.int64) * hop_time           # row-major: stream by stream, in run order
    off_t = runs[:, :, 1][off_ok].astype(np.int64) * hop_time
    n_on, n_off = on_ok.sum(axis=1), off_ok.sum(axis=1)
    on_parts = np.split(on_t, np.cumsum(n_on)[:-1]) if B else []
    off_parts = np.split(off_t, np.cumsum(n_off)[:-1]) if B else []
    out = []
    for b in range(B):
        onsets, offsets = on_parts[b], off_parts[b]
        if n_off[b] < n_on[b]:
            offsets = np.append(offsets, (N + 1) * hop_time)
        elif n_off[b] > n_on[b]:
            onsets = np.append(0, onsets)
        out.append((onsets, offsets))
    return out


# ---- D2 + D3 ---------------------------------------------------------------------------------------------


def _format_events(results, filenames, class_labels, frames_per_sec, starts, ends, counts):
    """Host formatting of crnn/utils.py:99-134 for all files in a few numpy calls: starts / ends (N, C, cap) frame
    indices and counts (N, C) -> {file: [event dicts]}, class-major in the order of class_labels, onset-ascending."""
    time_resolution = 1.0 / frames_per_sec
    n_files = min(starts.shape[0], len(filenames))
    names = []
    for i in range(n_files):
        filename = filenames[i]
        if isinstance(filename, np.ndarray):
            filename = filename.item()
        names.append(str(filename))
    order = list(class_labels.keys())
    labels = list(class_labels.values())
    if n_files == 0 or not order:
        for name in names:
            results[name] = []
        return results
    cnt = counts[:n_files][:, order]                                             # (n, K) in dict order
    live = np.arange(starts.shape[2])[None, None, :] < cnt[:, :, None]
    # np.int64 * float -> np.float64, rounded by numpy (rint(x * 1000) / 1000), as round() of the reference's np.float64
    # products does (crnn/utils.py:122-129)
    on = np.round(starts[:n_files][:, order][live].astype(np.int64) * time_resolution, 3)
    off = np.round(ends[:n_files][:, order][live].astype(np.int64) * time_resolution, 3)
    lab = np.broadcast_to(np.arange(len(order))[None, :, None], live.shape)[live]
    per_file = cnt.sum(axis=1)
    bounds = np.concatenate([[0], np.cumsum(per_file)])
    file_of = np.repeat(np.arange(n_files), per_file)
    events = [{"file": names[f], "event_onset": a, "event_offset": b, "event_label": labels[c]}
              for f, a, b, c in zip(file_of.tolist(), on, off, lab.tolist())]
    for i, name in enumerate(names):
        results[name] = events[bounds[i]:bounds[i + 1]]
    return results


def decode_predictions(preds, filenames, class_labels, frames_per_sec):
    """Binary activity (N, T, C) -> {file: [{'file','event_onset','event_offset','event_label'}]}."""
    a = _to_dev(preds, torch.int32)
    starts, ends, ns, ne = (t.cpu().numpy() for t in activity_edges(a))
    return _format_events({}, filenames, class_labels, frames_per_sec, starts, ends, np.minimum(ns, ne))


def decode_probabilities(probs, filenames, class_labels, frames_per_sec, posterior_thresh=0.5, smoothing="gauss"):
    """Sigmoid outputs (N, T, C) -> events: gaussian smoothing (sigma 3, 'nearest'), strict threshold, run decode."""
    p = _to_dev(probs, torch.float32)
    runs, counts = threshold_runs(p, posterior_thresh, smoothing)
    if p.shape[0] and list(class_labels.keys()) == list(range(p.shape[2])):
        # The usual label dict (every class, in index order) is the record packer's own order: the runs are compacted on
        # the device into (file, class, onset, offset) records, sized for the busiest file, and only those come to the host.
        from . import sharding
        most = int(counts.sum(dim=1).max().item())
        rec = sharding.pack_events(runs, counts, cap=max(most, 1)).cpu().numpy()
        return _format_records({}, rec, filenames, list(class_labels.values()), frames_per_sec)
    runs, counts = runs.cpu().numpy(), counts.cpu().numpy()
    return _format_events({}, filenames, class_labels, frames_per_sec, runs[..., 0], runs[..., 1], counts)


def _format_records(results, rec, filenames, labels, frames_per_sec):
    """_format_events for the packed records of sharding.pack_events: rec (N, cap + 1, 4), row 0 = (kept, dropped, 0, 0),
    rows 1.. = (file, class, onset frame, offset frame), class-major and onset-ascending."""
    time_resolution = 1.0 / frames_per_sec
    n_files = min(rec.shape[0], len(filenames))
    names = []
    for i in range(n_files):
        filename = filenames[i]
        if isinstance(filename, np.ndarray):
            filename = filename.item()
        names.append(str(filename))
    kept = rec[:n_files, 0, 0].astype(np.int64)
    live = np.arange(rec.shape[1] - 1)[None, :] < kept[:, None]
    body = rec[:n_files, 1:]
    on = np.round(body[..., 2][live].astype(np.int64) * time_resolution, 3)   # numpy's rounding, see _format_events
    off = np.round(body[..., 3][live].astype(np.int64) * time_resolution, 3)
    bounds = np.concatenate([[0], np.cumsum(kept)]).tolist()
    events = [{"file": names[f], "event_onset": a, "event_offset": b, "event_label": labels[c]}
              for f, a, b, c in zip(np.repeat(np.arange(n_files), kept).tolist(), on, off, body[..., 1][live].tolist())]
    for i, name in enumerate(names):
        results[name] = events[bounds[i]:bounds[i + 1]]
    return results


# ---- D4 --------------------------------------------------------------------------------------------------


def merge_close_events_batch(events_by_file: Dict[str, List[Dict[str, Any]]], threshold=0.5):
    """merge_close_events for every file of a results dict with ONE kernel launch.  The grouping of the reference
    (stable sort by onset, groups by label in order of first appearance, final stable sort by onset:
    src/utils/merge_events.py:19-33, 57-58) is done for all files at once with stable numpy sorts; the only per-event
    Python work is reading the input dicts and copying the dicts of the merged events.

    `threshold` may be a sequence of gaps (the reference's result files use 0.3, 0.6 and 1.0 s): the events are read
    and grouped once, every gap is one launch, and a list of result dicts comes back."""
    many = isinstance(threshold, (list, tuple))
    gaps = list(threshold) if many else [threshold]
    keys = list(events_by_file.keys())
    lens = [len(events_by_file[k]) for k in keys]
    flat = list(itertools.chain.from_iterable(events_by_file.values()))
    n = len(flat)
    if n == 0:
        outs = [{k: [] for k in keys} for _ in gaps]
        return outs if many else outs[0]
    on = np.array([ev["event_onset"] for ev in flat], dtype=np.float64)
    off = np.array([ev["event_offset"] for ev in flat], dtype=np.float64)
    labels = [ev["event_label"] for ev in flat]
    ids = {lbl: i for i, lbl in enumerate(dict.fromkeys(labels))}
    lab = np.array(list(map(ids.__getitem__, labels)), dtype=np.int64)
    file_idx = np.repeat(np.arange(len(keys)), lens)
    # events of a file in onset order (stable), then the (file, label) groups in the order their first event appears
    by_onset = np.lexsort((on, file_idx))
    gkey = file_idx[by_onset] * len(ids) + lab[by_onset]
    _, first_pos, inv = np.unique(gkey, return_index=True, return_inverse=True)
    first_of_event = first_pos[inv]
    grouped = np.lexsort((np.arange(n), first_of_event))          # positions in by_onset, group after group
    src = by_onset[grouped]                                       # index into `flat` of the event at each grouped slot
    g_first = first_of_event[grouped]
    starts = np.flatnonzero(np.concatenate([[True], g_first[1:] != g_first[:-1]]))
    group_start = np.concatenate([starts, [n]]).astype(np.int32)
    sizes = np.diff(group_start)
    n_groups = len(sizes)
    g_on, g_off = on[src], off[src]
    file_of_group = file_idx[by_onset][g_first[group_start[:-1]]]
    # Onsets ascend inside a group, so with the onsets replaced by their ranks among all values, (group, rank) ascends
    # over the grouped array and one binary search finds the event that opens each merged event (below).
    uniq = np.unique(g_on)
    key_ev = np.repeat(np.arange(n_groups), sizes) * (len(uniq) + 1) + np.searchsorted(uniq, g_on)
    dev = _device()
    d_on, d_off, d_start = torch.from_numpy(g_on).to(dev), torch.from_numpy(g_off).to(dev), torch.from_numpy(group_start).to(dev)
    launched = [merge_event_groups(d_on, d_off, d_start, gap) for gap in gaps]
    outs = []
    for m_on, m_off, m_cnt in launched:
        m_on, m_off, m_cnt = m_on.cpu().numpy(), m_off.cpu().numpy(), m_cnt.cpu().numpy().astype(np.int64)
        total = int(m_cnt.sum())
        gid = np.repeat(np.arange(n_groups), m_cnt)               # group of every merged event
        slot = np.repeat(group_start[:-1].astype(np.int64), m_cnt) + (np.arange(total) - np.repeat(np.cumsum(m_cnt) - m_cnt, m_cnt))
        h_on, h_off = m_on[slot], m_off[slot]
        # A merged event is a copy of the first event folded into it: the first event of its group, at or behind the
        # previous merged event's, whose onset is the merged onset.
        key_h = gid * (len(uniq) + 1) + np.searchsorted(uniq, h_on)
        head = np.searchsorted(key_ev, key_h, side="left")
        if total > 1:                                             # merged events of one group that share an onset: one after another
            same = np.concatenate([[False], key_h[1:] == key_h[:-1]])
            run_start = np.maximum.accumulate(np.where(same, 0, np.arange(total)))
            head = head + (np.arange(total) - run_start)
        # merged events of a file: its groups one after the other, then a stable sort by onset (merge_events.py:57-58)
        file_of = file_of_group[gid]
        final = np.lexsort((np.arange(total), h_on, file_of))
        merged = [{**flat[j], "event_offset": o} for j, o in zip(src[head[final]].tolist(), h_off[final].tolist())]
        bounds = np.searchsorted(file_of[final], np.arange(len(keys) + 1), side="left").tolist()
        outs.append({k: merged[bounds[i]:bounds[i + 1]] for i, k in enumerate(keys)})
    return outs if many else outs[0]


def merge_close_events(events: List[Dict[str, Any]], threshold: float = 0.5) -> List[Dict[str, Any]]:
    """Merge events of one file that share a label when the gap to the previous offset is <= threshold."""
    if not events:
        return []
    return merge_close_events_batch({"_": events}, threshold)["_"]


def process_json_file(input_file: str, output_file: str, threshold: float = 0.5):
    with open(input_file, "r") as f:
        data = json.load(f)
    merged_data = merge_close_events_batch(data, threshold)
    for filename, events in data.items():
        print(f"Processed {filename}: {len(events)} -> {len(merged_data[filename])} events")
    with open(output_file, "w") as f:
        json.dump(merged_data, f, indent=4)
    print(f"\nMerged results saved to {output_file}")


# ---- D5 --------------------------------------------------------------------------------------------------


def get_events_from_batch(batch, config, class_names: Sequence[str] = tuple(CLASSES + [NON_EVENT_LABEL]),
                          non_event_label: str = NON_EVENT_LABEL):
    """batch = {'file': [...], 'pred_map': (B, T, C)}; config.clip_duration in seconds.

    As in the reference only element 0 of the batch is decoded, interior boundaries are reported one frame
    late (its loop index is 1-based) and the last offset is exactly clip_duration.
    """
    non_event = list(class_names).index(non_event_label)
    file = batch["file"]
    pred_map = _to_dev(batch["pred_map"], torch.float32)
    batch_size, num_frame, num_classes = pred_map.shape
    runs, counts = argmax_runs(pred_map[:1], "gauss")
    runs, n = runs[0].cpu().numpy(), int(counts[0].item())
    file_record = file[0]
    file_events = []
    for label, s, e in runs[:n]:
        if label == non_event:
            continue
        onset_frame = 0 if s == 0 else int(s) + 1
        i = num_frame if e == num_frame else int(e) + 1
        file_events.append({"file": file_record, "event_onset": onset_frame / num_frame * config.clip_duration,
                            "event_offset": i / num_frame * config.clip_duration,
                            "event_label": class_names[int(label)]})
    return {file_record: file_events}


def get_events(frame_predictions, config, **kw):
    """src/models/htsat/combined.py:75-80."""
    pred_event_dict = {}
    for batch in frame_predictions:
        pred_event_dict = pred_event_dict | get_events_from_batch(batch, config, **kw)
    return pred_event_dict


# ---- D6 --------------------------------------------------------------------------------------------------


def predict_events(frame_probs, file, classes: Sequence[str] = tuple(CLASSES), frame_hop_sec=0.48, med_filter=False):
    """frame_probs (n_frames, 1 + len(classes)) with class 0 = background -> events of one file."""
    return predict_events_batch(torch.as_tensor(frame_probs)[None], [file], classes, frame_hop_sec, med_filter)[0]


def predict_events_batch(frame_probs, files, classes: Sequence[str] = tuple(CLASSES), frame_hop_sec=0.48,
                         med_filter=False):
    p = _to_dev(frame_probs, torch.float32)
    runs, counts = argmax_runs(p, "median3" if med_filter else None)
    runs, counts = runs.cpu().numpy(), counts.cpu().numpy()
    out = []
    for b, file in enumerate(files):
        filename = os.path.basename(file)
        events = []
        for label, s, e in runs[b, :counts[b]]:
            if label == 0:
                continue
            events.append({"file": filename, "event_onset": int(s) * frame_hop_sec, "event_offset": int(e) * frame_hop_sec,
                           "event_label": classes[int(label) - 1]})
        out.append(events)
    return out


# ---- D7 --------------------------------------------------------------------------------------------------


def get_classification_label(pred_map, class_names: Sequence[str] = tuple(CLASSES + [NON_EVENT_LABEL])):
    """pred_map (T, C) or (B, T, C): drop the last class, per-frame argmax, majority vote (smallest id wins ties)."""
    p = _to_dev(pred_map, torch.float32)
    single = p.dim() == 2
    if single:
        p = p[None]
    hist = argmax_histogram(p, drop_last_classes=1).cpu().numpy()
    labels = [class_names[int(np.argmax(h))] for h in hist]
    return labels[0] if single else labels


def get_classification_events(frame_predictions, events_list, CLASS_LABELS=tuple(CLASSES + [NON_EVENT_LABEL])):
    """src/models/htsat/classification.py:15-36 with its signature: every clip of every test batch
    ({"file": [...], "file_path": [...], "pred_map": (B, T, C)}) gets its clip label (get_classification_label), and the
    first event of events_list[file] whose "extracted_audio_filename" is the base name of the clip's path takes it as
    "event_label".  events_list is updated in place and returned.  CLASS_LABELS: a sequence of names or an object with
    int2str (datasets.ClassLabel).  One histogram launch per batch; a file that is not in events_list raises KeyError as
    the reference does."""
    import os
    name_of = CLASS_LABELS.int2str if hasattr(CLASS_LABELS, "int2str") else (lambda k: CLASS_LABELS[k])
    for batch in frame_predictions:
        files = batch["file"]
        if len(files) == 0:
            continue
        pm = batch["pred_map"]
        if isinstance(pm, (list, tuple)):
            pm = torch.stack([torch.as_tensor(m) for m in pm])
        hist = argmax_histogram(_to_dev(pm, torch.float32)[:len(files)], drop_last_classes=1).cpu().numpy()
        for i, filename in enumerate(files):
            extracted = os.path.basename(batch["file_path"][i])
            label = name_of(int(np.argmax(hist[i])))
            events = events_list[filename]
            for ev in events:
                if ev["extracted_audio_filename"] == extracted:
                    ev["event_label"] = label
                    break
            events_list[filename] = events
    return events_list


def _framewise_expand(x: torch.Tensor, ratio: int, frames_out: int, what: str) -> torch.Tensor:
    if not (torch.is_tensor(x) and x.is_cuda):
        raise RuntimeError(f"{what}: expected a tensor on a ROCm device (this path has no CPU implementation)")
    if x.dim() != 3:
        raise ValueError(f"{what}: expected (batch_size, time_steps, classes_num), got shape {tuple(x.shape)}")
    B, T, C = x.shape
    if T < 1 or C < 1:
        raise ValueError(f"{what}: empty time or class axis")
    x = x.to(torch.float32).contiguous()
    out = torch.empty((B, frames_out, C), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        L.check(L.lib.sed_framewise_expand(x.data_ptr(), B, T, C, int(ratio), int(frames_out), out.data_ptr(),
                                           L.stream_ptr(x.device)), "sed_framewise_expand")
    return out


def interpolate(x: torch.Tensor, ratio: int) -> torch.Tensor:
    """external/hts_audio_transformer/utils.py:158-172: (batch_size, time_steps, classes_num) ->
    (batch_size, time_steps * ratio, classes_num), every frame repeated `ratio` times."""
    return _framewise_expand(x, ratio, x.shape[1] * int(ratio), "interpolate")


def pad_framewise_output(framewise_output: torch.Tensor, frames_num: int) -> torch.Tensor:
    """external/hts_audio_transformer/utils.py:175-192: pad along time to frames_num frames with the last frame."""
    if frames_num < framewise_output.shape[1]:
        raise RuntimeError("pad_framewise_output: frames_num is smaller than the number of frames")  # torch: negative repeat
    return _framewise_expand(framewise_output, 1, frames_num, "pad_framewise_output")


def interpolate_and_pad(x: torch.Tensor, ratio: int, frames_num: int) -> torch.Tensor:
    """interpolate then pad_framewise_output in one launch."""
    if frames_num < x.shape[1] * int(ratio):
        raise RuntimeError("interpolate_and_pad: frames_num is smaller than time_steps * ratio")
    return _framewise_expand(x, ratio, frames_num, "interpolate_and_pad")
