"""CPU baselines for bench.py: the reference's own formulations of the hot path on host cores.
TEST / MEASUREMENT INFRASTRUCTURE ONLY (never imported by the product package).

torchlibrosa computes the STFT as two Conv1d with fp32 windowed-DFT weights over the reflect-padded waveform,
then power -> matmul(melW) -> 10*log10(clamp) (call site external/hts_audio_transformer/model/htsat.py:763-764).
torchlibrosa itself is not installable here, so this is the oracle's restatement ("kind": "port") run with
torch's CPU kernels (oneDNN conv, MKL/OpenBLAS matmul) on all the threads torch is given.
"""
from __future__ import annotations

import time

import numpy as np
import torch
import torch.nn.functional as F

from . import frontend as of


class HtsatCpuFrontend:
    def __init__(self, sr=32000, n_fft=1024, hop=320, n_mels=64, fmin=50, fmax=14000, amin=1e-10):
        wr, wi = of.windowed_dft_matrix(n_fft, of.hann_periodic(n_fft))
        self.conv_real = torch.from_numpy(wr.T.astype(np.float32))[:, None, :].contiguous()
        self.conv_imag = torch.from_numpy(wi.T.astype(np.float32))[:, None, :].contiguous()
        self.melW = torch.from_numpy(of.librosa_mel(sr, n_fft, n_mels, fmin, fmax).T.copy())
        self.n_fft, self.hop, self.amin = n_fft, hop, amin

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        x = F.pad(x[:, None, :], (self.n_fft // 2, self.n_fft // 2), mode="reflect")
        real = F.conv1d(x, self.conv_real, stride=self.hop)
        imag = F.conv1d(x, self.conv_imag, stride=self.hop)
        power = (real ** 2 + imag ** 2).transpose(1, 2)[:, None]
        mel = torch.matmul(power, self.melW)
        return 10.0 * torch.log10(torch.clamp(mel, min=self.amin))


def time_htsat_cpu(clips: torch.Tensor, budget_s: float = 15.0):
    """Audio-hours/s of the CPU formulation on `clips` (B, L) fp32 @ 32 kHz; repeats until ~budget_s of work."""
    fe = HtsatCpuFrontend()
    fe(clips[:1])  # warm-up (thread pool, oneDNN primitive cache)
    t0 = time.perf_counter()
    n = 0
    while True:
        fe(clips)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 2000:
            break
    hours = n * clips.shape[0] * clips.shape[1] / 32000.0 / 3600.0
    return hours / el, el, n


class YamnetCpuFrontend:
    """YAMNet's feature graph as TensorFlow forms it (tensorflow/models research/audioset/yamnet/features.py, called
    inside the hub model at src/models/yamnet_train.py:46-47): pad_waveform, tf.signal.stft(400, 160, fft_length 512) =
    frame + periodic Hann + rFFT, magnitude, matmul with linear_to_mel_weight_matrix, log(mel + 0.001)."""

    def __init__(self):
        self.window = torch.from_numpy(of.hann_periodic(400).astype(np.float32))
        self.melW = torch.from_numpy(of.tf_linear_to_mel_weight_matrix().astype(np.float32))

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        pad = of.yamnet_padded_length(x.shape[-1]) - x.shape[-1]
        x = F.pad(x, (0, pad))
        frames = x.unfold(-1, 400, 160) * self.window
        mag = torch.fft.rfft(frames, n=512).abs()
        return torch.log(torch.matmul(mag, self.melW) + 0.001)


def time_cpu_frontend(fe, clips: torch.Tensor, sr: int, budget_s: float = 10.0):
    """Audio-hours/s of a CPU front-end on `clips` (B, L) fp32; repeats until ~budget_s of work."""
    fe(clips[:1])
    t0 = time.perf_counter()
    n = 0
    while True:
        fe(clips)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 2000:
            break
    return n * clips.shape[0] * clips.shape[1] / float(sr) / 3600.0 / el, el, n


def time_crnn_chain_cpu(probs, files, labels, fps, gaps, want_raw=None, want_merged=None, budget_s: float = 10.0):
    """The reference's decode chain behind the CRNN on the host, as it runs it: scipy's gaussian_filter1d and the
    threshold (src/models/crnn/crnn.py:418-425), the per-file, per-class Python region decode
    (src/models/crnn/utils.py:91-134) and the pure-Python gap merge per file (src/utils/merge_events.py:4-60) at every
    gap.  Returns (ms per chain, passes, seconds, whether the events equal want_raw / want_merged)."""
    from scipy.ndimage import gaussian_filter1d
    from . import decode as od
    t0 = time.perf_counter()
    n = 0
    while True:
        sm = gaussian_filter1d(probs, sigma=3, axis=1, mode="nearest")
        act = (sm > 0.5).astype(int)
        raw = od.decode_predictions(act, files, labels, fps)
        merged = [{k: od.merge_close_events(v, g) for k, v in raw.items()} for g in gaps]
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    same = None
    if want_raw is not None:
        same = bool(raw == want_raw and all(a == b for a, b in zip(merged, want_merged)))
    return el / n * 1e3, n, el, same
