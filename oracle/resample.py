"""CPU restatement of polyphase resampling as librosa.resample(res_type='polyphase') does it, i.e.
scipy.signal.resample_poly(y, target_sr // g, orig_sr // g) with its default Kaiser(5.0) filter
(call sites in the reference: librosa.load(file_path, sr=16000) at src/models/yamnet_train.py:46 and
librosa.resample(track, ofs, rfs) at external/hts_audio_transformer/utils.py:211; both use librosa's default res_type,
soxr_hq, whose filter design lives in libsoxr and is not restated here).
TEST INFRASTRUCTURE ONLY (never imported by the product package).

The arithmetic follows scipy's fp32 path operation by operation (filter designed in fp64, cast to fp32, scaled by `up`
in fp32; every output accumulated in fp32 from the oldest input sample to the newest, multiply and add rounded
separately), so the restatement reproduces scipy.signal.resample_poly bit for bit on fp32 input.
"""
from __future__ import annotations

import math

import numpy as np


def _i0(x: np.ndarray) -> np.ndarray:
    """Modified Bessel function of the first kind, order 0, by its power series in fp64."""
    x = np.asarray(x, np.float64)
    q = (x * 0.5) ** 2
    term = np.ones_like(x)
    total = np.ones_like(x)
    for k in range(1, 60):
        term = term * q / (k * k)
        total = total + term
    return total


def design_filter(up: int, down: int):
    """(h fp32 already scaled by `up`, half_len): firwin(2*half_len+1, 1/max(up, down), window=('kaiser', 5.0))."""
    max_rate = max(up, down)
    f_c = 1.0 / max_rate
    half_len = 10 * max_rate
    n = 2 * half_len + 1
    m = np.arange(n, dtype=np.float64) - half_len
    h = f_c * np.sinc(f_c * m)
    win = _i0(5.0 * np.sqrt(np.maximum(0.0, 1.0 - (m / half_len) ** 2))) / _i0(np.float64(5.0))
    h = h * win
    h = h / np.sum(h)
    h32 = h.astype(np.float32)
    h32 = h32 * np.float32(up)
    return h32, half_len


def plan(orig_sr: int, target_sr: int):
    g = math.gcd(int(orig_sr), int(target_sr))
    up, down = int(target_sr) // g, int(orig_sr) // g
    if up == down == 1:
        return dict(up=1, down=1)
    h, half_len = design_filter(up, down)
    n_pre_pad = down - half_len % down
    n_pre_remove = (half_len + n_pre_pad) // down
    hp = np.concatenate([np.zeros(n_pre_pad, np.float32), h])
    taps = -(-len(hp) // up)                                    # coefficients per phase
    table = np.zeros((up, taps), np.float32)                    # table[phase][j] = hp[phase + up*j]
    full = np.zeros(up * taps, np.float32)
    full[:len(hp)] = hp
    table[:, :] = full.reshape(taps, up).T
    return dict(up=up, down=down, n_pre_remove=n_pre_remove, taps=taps, table=table)


def out_len(n_in: int, up: int, down: int) -> int:
    return -(-(n_in * up) // down)


def resample_poly_f32(x, orig_sr: int, target_sr: int) -> np.ndarray:
    """(..., L) fp32 -> (..., ceil(L*up/down)) fp32 with scipy's bits (vectorised over outputs, taps in scipy's order)."""
    x = np.asarray(x, np.float32)
    p = plan(orig_sr, target_sr)
    if p["up"] == p["down"] == 1:
        return x.copy()
    up, down, taps, table = p["up"], p["down"], p["taps"], p["table"]
    L = x.shape[-1]
    n_out = out_len(L, up, down)
    t = (np.arange(n_out, dtype=np.int64) + p["n_pre_remove"]) * down
    phase, q = t % up, t // up
    acc = np.zeros(x.shape[:-1] + (n_out,), np.float32)
    for j in range(taps - 1, -1, -1):                           # oldest input first
        k = q - j
        ok = (k >= 0) & (k < L)
        xv = np.where(ok, x[..., np.clip(k, 0, L - 1)], np.float32(0))
        acc = acc + xv * table[phase, j]                        # fp32 multiply, fp32 add
    return acc


def librosa_resample_polyphase(y, orig_sr: int, target_sr: int) -> np.ndarray:
    """librosa.resample(y, orig_sr=, target_sr=, res_type='polyphase', fix=True, scale=False)."""
    y = np.asarray(y, np.float32)
    if orig_sr == target_sr:
        return y
    out = resample_poly_f32(y, orig_sr, target_sr)
    n = int(np.ceil(y.shape[-1] * (float(target_sr) / orig_sr)))
    if out.shape[-1] > n:
        out = out[..., :n]
    elif out.shape[-1] < n:
        out = np.concatenate([out, np.zeros(out.shape[:-1] + (n - out.shape[-1],), np.float32)], axis=-1)
    return np.ascontiguousarray(out)
