"""Detector -> classifier hand-off on the device: event ranges of the resident PCM batch become the classifier batch.

Replaces cut_events_from_audio (src/utils/cut_audio.py:13-53), which for every event re-reads the wav file, slices
audio_array[int(start_sec * sr):int(end_sec * sr)] and writes a new wav file that the classifier stage loads again
(main.py:123-127, 136-138).  The sample indices are formed here with the reference's own Python expression, so the
cut samples are exactly the ones the reference would have written.  There is no CPU path.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L


def cut_events(pcm: torch.Tensor, sr, events: Sequence[Tuple[int, float, float]], cap: int = None,
               pcm_divisor: float = 32768.0):
    """pcm (S, L) fp32 or int16 on the device; events: (stream index, start_sec, end_sec).

    Returns (clips (E, cap) fp32 zero-padded, lengths (E,) int32 = slice lengths).  cap defaults to the longest slice
    rounded up to a multiple of 4; slices longer than cap are truncated (lengths still report the full slice).
    """
    if not pcm.is_cuda:
        raise RuntimeError(f"cut_events: pcm is on {pcm.device}; the hand-off runs only on a ROCm device")
    if pcm.dim() != 2 or pcm.dtype not in (torch.float32, torch.int16):
        raise RuntimeError("cut_events: expected a (streams, samples) fp32 or int16 tensor")
    S, n = pcm.shape
    # rows must be sample-contiguous and must not overlap: an expanded clip or an as_strided / unfold view has
    # stride(0) < n, which the kernel would index as if rows were n apart
    if pcm.stride(1) != 1 or (S > 1 and pcm.stride(0) < n):
        pcm = pcm.contiguous()
    E = len(events)
    # int(start_sec * sr), int(end_sec * sr) of cut_audio.py:37-38 for all events at once: the fp64 product of Python's
    # float arithmetic, truncated toward zero as int() does
    ev = np.asarray(events, dtype=np.float64).reshape(E, 3)
    idx = ev[:, 0].astype(np.int32)
    start = np.trunc(ev[:, 1] * sr).astype(np.int64)
    end = np.trunc(ev[:, 2] * sr).astype(np.int64)
    if cap is None:
        a = np.clip(np.where(start < 0, start + n, start), 0, n)
        b = np.clip(np.where(end < 0, end + n, end), 0, n)
        cap = int(max(4, (int(np.max(b - a, initial=0)) + 3) // 4 * 4))
    dev = pcm.device
    clips = torch.empty((E, cap), dtype=torch.float32, device=dev)
    lengths = torch.empty((E,), dtype=torch.int32, device=dev)
    if E == 0:
        return clips, lengths
    # one upload for the three index arrays: [start int64 | end int64 | stream int32]
    packed = np.empty(20 * E, dtype=np.uint8)
    packed[:8 * E] = start.view(np.uint8)
    packed[8 * E:16 * E] = end.view(np.uint8)
    packed[16 * E:] = idx.view(np.uint8)
    d_packed = torch.from_numpy(packed).to(dev)
    d_start, d_end = d_packed[:8 * E].view(torch.int64), d_packed[8 * E:16 * E].view(torch.int64)
    d_idx = d_packed[16 * E:].view(torch.int32)
    with torch.cuda.device(dev):
        L.check(L.lib.sed_cut_events(pcm.data_ptr(), L.PCM_I16 if pcm.dtype == torch.int16 else L.PCM_F32,
                                     float(pcm_divisor), S, n, max(pcm.stride(0), n), d_idx.data_ptr(), d_start.data_ptr(),
                                     d_end.data_ptr(), E, clips.data_ptr(), cap, lengths.data_ptr(),
                                     L.stream_ptr(dev)), "sed_cut_events")
    return clips, lengths


def cut_events_from_audio(pcm: torch.Tensor, sr, filenames: Sequence[str], events_list: Dict[str, List[dict]],
                          cap: int = None, pcm_divisor: float = 32768.0):
    """events_list as the detector stage produces it ({filename: [{"event_onset", "event_offset", ...}]});
    filenames[i] names stream i of pcm.  Returns (clips, lengths, updated events_list): every event gains
    "extracted_audio_filename" (the name the reference would have written, cut_audio.py:44) and "clip_index"
    (its row in clips)."""
    where = {name: i for i, name in enumerate(filenames)}
    flat, out = [], {}
    for filename, events in events_list.items():
        stem = Path(filename).stem
        new = []
        for ev in events:
            s, e = ev["event_onset"], ev["event_offset"]
            new.append(ev | {"extracted_audio_filename": f"{stem}_{s:.2f}_{e:.2f}.wav", "clip_index": len(flat)})
            flat.append((where[filename], s, e))
        out[filename] = new
    clips, lengths = cut_events(pcm, sr, flat, cap, pcm_divisor)
    return clips, lengths, out
