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
    # one asynchronous upload from pinned memory for the three index arrays: [start int64 | end int64 | stream int32]
    stage = torch.empty(20 * E, dtype=torch.uint8, pin_memory=True)
    packed = stage.numpy()
    packed[:8 * E] = start.view(np.uint8)
    packed[8 * E:16 * E] = end.view(np.uint8)
    packed[16 * E:] = idx.view(np.uint8)
    d_packed = stage.to(dev, non_blocking=True)
    d_start, d_end = d_packed[:8 * E].view(torch.int64), d_packed[8 * E:16 * E].view(torch.int64)
    d_idx = d_packed[16 * E:].view(torch.int32)
    with torch.cuda.device(dev):
        L.check(L.lib.sed_cut_events(pcm.data_ptr(), L.PCM_I16 if pcm.dtype == torch.int16 else L.PCM_F32,
                                     float(pcm_divisor), S, n, max(pcm.stride(0), n), d_idx.data_ptr(), d_start.data_ptr(),
                                     d_end.data_ptr(), E, clips.data_ptr(), cap, lengths.data_ptr(),
                                     L.stream_ptr(dev)), "sed_cut_events")
    return clips, lengths


def _event_ranges(events, sr, n):
    """(stream, first sample, sample count) of every event with the index arithmetic of cut_audio.py:37-38 and numpy slicing:
    int(sec * sr) truncated toward zero, negative indices counted from the end, both ends clipped to the recording."""
    E = len(events)
    ev = np.asarray(events, dtype=np.float64).reshape(E, 3)
    idx = ev[:, 0].astype(np.int64)
    start = np.trunc(ev[:, 1] * sr).astype(np.int64)
    end = np.trunc(ev[:, 2] * sr).astype(np.int64)
    a = np.clip(np.where(start < 0, start + n, start), 0, n)
    b = np.clip(np.where(end < 0, end + n, end), 0, n)
    return idx, a, np.maximum(b - a, 0)


def logmel_of_events(frontend, pcm: torch.Tensor, sr, events: Sequence[Tuple[int, float, float]], cap: int,
                     per_file: bool = False):
    """The classifier front-end on the detected events WITHOUT the cut in between: the fused log-mel kernel reads every
    event's samples straight out of the resident recordings (src/utils/cut_audio.py:13-53 + main.py:123-127 write the events
    as files for stage 2; external/hts_audio_transformer/model/htsat.py:763-764 is the front-end they feed).

    pcm (S, L) fp32 or int16 on the device; events (stream index, start_sec, end_sec); cap = samples per classifier clip
    (longer events are truncated, shorter ones zero-padded, as cut_events(..., cap=cap) does).  frontend: a LogMelFrontend
    whose configuration takes pcm's sample type (for int16 recordings: pcm_kind=1 and the divisor of the hand-off).
    Returns (log-mel of the E events in the frontend's layout, lengths (E,) int32 on the device): the same values, bit for
    bit, as frontend(cut_events(pcm, sr, events, cap)[0]), with neither the (E, cap) batch nor its kernel.

    per_file=True gives every event the log-mel of a file of its own, which is what the reference's second stage computes:
    it loads each written event at its natural length and feeds it alone (src/models/htsat/data.py:138-146, batch_size = 1
    at src/models/htsat/classification.py:55-59), so the reflect padding of the STFT mirrors the event's own last samples
    and the event has 1 + len // hop frames of its own.  Returns (log-mel, lengths, frames): rows [0, frames[e]) of event e
    equal frontend(event's samples alone) bit for bit, the rows behind them are zero; frames (E,) int32 on the device goes to
    reshape_wav2img(..., frames=frames), which stretches every event to the image from its own rows.  Events too short for
    the front-end (the reference's STFT raises on them) have frames[e] = 0."""
    from . import frontend as F
    if not isinstance(frontend, F.LogMelFrontend):
        raise TypeError("logmel_of_events needs a sed_amd.LogMelFrontend")
    if not pcm.is_cuda:
        raise RuntimeError(f"logmel_of_events: pcm is on {pcm.device}; the hand-off runs only on a ROCm device")
    if pcm.dim() != 2 or pcm.dtype not in (torch.float32, torch.int16):
        raise RuntimeError("logmel_of_events: expected a (streams, samples) fp32 or int16 tensor")
    S, n = pcm.shape
    if pcm.stride(1) != 1 or (S > 1 and pcm.stride(0) < n):
        pcm = pcm.contiguous()
    cap = int(cap)
    if cap < 1:
        raise ValueError("cap must be at least one sample")
    idx, a, length = _event_ranges(events, sr, n)
    if len(idx) and (idx.min() < 0 or idx.max() >= S):
        raise ValueError("event names a stream outside the batch")
    return frontend.forward_segments(pcm, idx * max(pcm.stride(0), n) + a, length, cap, own_edges=bool(per_file))


def fbank_of_events(pcm: torch.Tensor, sr, events: Sequence[Tuple[int, float, float]], melbins=128, target_length=1024,
                    fshift=10, norm_mean=None, norm_std=None, pcm_divisor: float = 32768.0, fused: bool = True):
    """The Audio-Mamba input of the detected events, every event as the file of its own the reference writes and reads back
    (src/utils/cut_audio.py:13-53 + main.py:129-152; external/audio_mamba/src/dataloader.py:94-152, 219-221).
    pcm (S, L) fp32 or int16 on the device; events (stream index, start_sec, end_sec).  Returns (fbank (E, target_length,
    melbins), lengths (E,) int32 on the device).  An event shorter than one window gives the value of zero frames in every
    row.

    fused=True: the Kaldi filterbank kernel reads the events straight out of the resident recordings (segment mode, every
    event with its own 1 + (n - window) // shift frames, zeros behind them), so the (events, longest event) batch is neither
    written nor read and the frames behind short events are not transformed.  The reference's `waveform - waveform.mean()`
    is left to the per-frame DC removal that follows it inside the filterbank: (x - m) - mean(x - m) = x - mean(x) exactly,
    in fp32 up to one rounding of the samples (measured: 2.3e-5 nepers at most between the two paths on bins within
    60 dB of a frame's peak, both within 1e-3 of the fp64 restatement).  fused=False: cut_events followed by wav2fbank(..., lengths=...), with the subtraction as the
    reference writes it."""
    from . import frontend as F
    win = int(int(round(sr)) * 25.0 * 0.001)
    hop = int(int(round(sr)) * fshift * 0.001)
    E = len(events)
    length = _event_ranges(events, sr, pcm.shape[1])[2] if E else np.zeros(0, np.int64)
    if not fused:
        # rows as long as the longest event (its mean is that of all its samples, as the file's would be), never shorter than a window
        cap = (max(int(length.max(initial=0)), win) + 3) // 4 * 4
        clips, lengths = cut_events(pcm, sr, events, cap=cap, pcm_divisor=pcm_divisor)
        if E == 0:
            return clips.new_empty((0, int(target_length), int(melbins))), lengths
        fb = F.wav2fbank(clips, sr, melbins=melbins, target_length=target_length, fshift=fshift, norm_mean=norm_mean,
                         norm_std=norm_std, lengths=lengths)
        return fb, lengths
    if not pcm.is_cuda:
        raise RuntimeError(f"fbank_of_events: pcm is on {pcm.device}; the hand-off runs only on a ROCm device")
    if pcm.dim() != 2 or pcm.dtype not in (torch.float32, torch.int16):
        raise RuntimeError("fbank_of_events: expected a (streams, samples) fp32 or int16 tensor")
    S, n = pcm.shape
    if pcm.stride(1) != 1 or (S > 1 and pcm.stride(0) < n):
        pcm = pcm.contiguous()
    kw = dict(pcm_kind=L.PCM_I16, pcm_divisor=float(pcm_divisor)) if pcm.dtype == torch.int16 else {}
    fe = F._functional(F.FrontendConfig.kaldi_fbank(sr, melbins, 25.0, fshift, **kw), "btm")
    if E == 0:
        return (torch.empty((0, int(target_length), int(melbins)), dtype=torch.float32, device=pcm.device),
                torch.empty((0,), dtype=torch.int32, device=pcm.device))
    idx, a, length = _event_ranges(events, sr, n)
    if idx.min() < 0 or idx.max() >= S:
        raise ValueError("event names a stream outside the batch")
    # rows: the samples target_length frames read, or the longest event if that is shorter; never shorter than a window
    cap = max(win, min(win + (int(target_length) - 1) * hop, int(length.max(initial=0))))
    # the kernel writes into the (E, target_length, melbins) result; only the rows it wrote are normalised, the zero frames
    # behind them are filled with the value the same arithmetic gives a zero
    fb = torch.empty((E, int(target_length), int(melbins)), dtype=torch.float32, device=pcm.device)
    head, lengths, _ = fe.forward_segments(pcm, idx * max(pcm.stride(0), n) + a, length, cap, own_edges=True, out=fb)
    fill = torch.zeros((), dtype=torch.float32, device=pcm.device)
    if norm_mean is not None and norm_std is not None:
        head.sub_(norm_mean).div_(norm_std * 2)
        fill = (fill - norm_mean) / (norm_std * 2)
    if head.shape[1] < fb.shape[1]:
        fb[:, head.shape[1]:] = fill
    return fb, lengths


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
