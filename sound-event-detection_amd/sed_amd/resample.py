"""Sample-rate conversion on the device, ahead of the front-end.

Mirrors librosa.resample(y, orig_sr=, target_sr=, res_type='polyphase') = scipy.signal.resample_poly with its default
Kaiser(5.0) filter (same bits on fp32 input).  The reference resamples with librosa's default res_type, soxr_hq
(librosa.load(file_path, sr=16000) at src/models/yamnet_train.py:46, librosa.resample(track, ofs, rfs) at
external/hts_audio_transformer/utils.py:211); that filter lives in libsoxr and is not offered here, so any res_type
other than 'polyphase' raises.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Tuple

import numpy as np
import torch

from . import _lib as L


class Resampler:
    """One sample-rate pair on one device.  `forward(x)`: (B, L) or (L,) fp32 / int16 device tensor -> fp32
    (B, ceil(L * up / down)); int16 samples are divided by `pcm_divisor` first (32768.0 = soundfile's convention)."""

    def __init__(self, orig_sr: int, target_sr: int, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("no ROCm device is visible; the resampler has no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.orig_sr, self.target_sr = int(orig_sr), int(target_sr)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(L.lib.sed_resampler_create(self.orig_sr, self.target_sr, C.byref(self._h)), "sed_resampler_create")
        up, down, taps, npr = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        L.check(L.lib.sed_resampler_get_filter(self._h, C.byref(up), C.byref(down), C.byref(taps), C.byref(npr), None))
        self.up, self.down, self.taps, self.n_pre_remove = up.value, down.value, taps.value, npr.value

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                L.lib.sed_resampler_destroy(h)
            except Exception:
                pass

    def filter_table(self) -> np.ndarray:
        """fp32 [up][taps]: table[phase][j] = zero-padded filter[phase + up * j]."""
        t = np.empty((self.up, self.taps), np.float32)
        L.check(L.lib.sed_resampler_get_filter(self._h, None, None, None, None, t.ctypes.data_as(C.c_void_p)))
        return t

    def out_len(self, n_in: int) -> int:
        return int(L.lib.sed_resampler_out_len(self._h, int(n_in)))

    def forward(self, x: torch.Tensor, pcm_divisor: float = 32768.0) -> torch.Tensor:
        if not (torch.is_tensor(x) and x.is_cuda):
            raise RuntimeError("Resampler: expected a tensor on a ROCm device (no CPU path)")
        if x.device != self.device:
            raise RuntimeError(f"Resampler was made for {self.device}, input is on {x.device}")
        squeeze = x.dim() == 1
        if squeeze:
            x = x[None]
        if x.dim() != 2:
            raise RuntimeError(f"expected (batch, samples) or (samples,), got shape {tuple(x.shape)}")
        if x.dtype == torch.int16:
            kind = L.PCM_I16
        elif x.dtype == torch.float32:
            kind = L.PCM_F32
        else:
            raise RuntimeError(f"expected dtype torch.float32 or torch.int16, got {x.dtype}")
        if x.stride(1) != 1 or x.stride(0) < x.shape[1]:
            x = x.contiguous()
        B, n = x.shape
        n_out = self.out_len(n)
        out = torch.empty((B, n_out), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            L.check(L.lib.sed_resampler_forward(self._h, x.data_ptr(), kind, float(pcm_divisor), B, n, max(x.stride(0), n),
                                                out.data_ptr(), n_out, L.stream_ptr(x.device)), "sed_resampler_forward")
        return out[0] if squeeze else out

    __call__ = forward


_CACHE: Dict[Tuple[int, int, int], Resampler] = {}


def _resampler(orig_sr: int, target_sr: int, device: torch.device) -> Resampler:
    key = (int(orig_sr), int(target_sr), device.index if device.index is not None else torch.cuda.current_device())
    if key not in _CACHE:
        _CACHE[key] = Resampler(orig_sr, target_sr, torch.device("cuda", key[2]))
    return _CACHE[key]


def resample(y, *, orig_sr: float, target_sr: float, res_type: str = "polyphase", fix: bool = True, scale: bool = False,
             pcm_divisor: float = 32768.0):
    """librosa.resample for res_type='polyphase' along the last axis; numpy in -> numpy out, tensor in -> tensor out.
    int16 input (the samples as stored in a wav file) is converted to fp32 by `/ pcm_divisor` on the way in."""
    if res_type != "polyphase":
        raise ValueError(f"res_type={res_type!r} is not available on the device; only 'polyphase' "
                         "(scipy.signal.resample_poly) is")
    if int(orig_sr) != orig_sr or int(target_sr) != target_sr:
        raise ValueError("polyphase resampling needs integer sample rates")
    is_np = not torch.is_tensor(y)
    t = torch.as_tensor(y)
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError("no ROCm device is visible; the resampler has no CPU path")
        t = t.to("cuda")
    if t.dtype not in (torch.float32, torch.int16):
        t = t.to(torch.float32)
    lead = t.shape[:-1]
    flat = t.reshape(-1, t.shape[-1])
    if orig_sr == target_sr and t.dtype == torch.float32:
        out = flat
    else:
        out = _resampler(int(orig_sr), int(target_sr), flat.device)(flat, pcm_divisor)
    ratio = float(target_sr) / float(orig_sr)
    if fix:
        n = int(math.ceil(t.shape[-1] * ratio))
        if out.shape[-1] > n:
            out = out[:, :n]
        elif out.shape[-1] < n:
            out = torch.nn.functional.pad(out, (0, n - out.shape[-1]))
    if scale:
        out = out / math.sqrt(ratio)
    out = out.reshape(lead + (out.shape[-1],)).contiguous()
    return out.cpu().numpy() if is_np else out
