"""CPU baseline for bench.py: the reference's own formulation of the HTS-AT front-end on host cores.
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
