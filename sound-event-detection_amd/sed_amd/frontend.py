"""Drop-in log-mel front-ends running on the HIP kernels of libsedhip.so.

Mirrors, for inference, the interfaces the reference calls:

  Spectrogram / LogmelFilterBank   torchlibrosa modules as constructed at
                                   external/hts_audio_transformer/model/htsat.py:453-459 and called at :763-764
  LogMelFrontend                   the two fused into one launch (the (T, F) spectrogram is never written)
  yamnet_log_mel                   output #3 of yamnet_model(audio), src/models/yamnet_train.py:47
  extract_mbe                      src/models/crnn/process_audio.py:29-34
  get_mel_spec_db                  src/utils/audio_to_spectrograms.py:30-44 (on samples instead of a file path)

All outputs are fresh contiguous fp32 tensors on the input's device; inputs are never modified.  The kernels
are inference-only: outputs do not carry autograd history.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import math
from typing import Optional

import numpy as np
import torch
from torch import nn

from . import _lib as L


@dataclasses.dataclass(frozen=True)
class FrontendConfig:
    """One front-end flavour (field-for-field sed_frontend_cfg)."""
    sample_rate: int
    n_fft: int
    win_length: int
    hop: int
    center: bool = True
    pad_mode: int = L.PAD_REFLECT
    win_align: int = L.WIN_CENTER
    tail_min_samples: int = 0
    tail_hop_samples: int = 0
    spectrum: int = L.SPEC_POWER
    mel_kind: int = L.MEL_SLANEY
    n_mels: int = 64
    fmin: float = 0.0
    fmax: float = 0.0
    log_kind: int = L.LOG_DB10
    amin: float = 1e-10
    ref: float = 1.0
    ln_offset: float = 0.0
    pcm_kind: int = L.PCM_F32
    pcm_divisor: float = 32767.0
    window_kind: int = L.WINDOW_HANN_PERIODIC
    remove_dc: int = 0
    preemphasis: float = 0.0

    def to_c(self) -> L.FrontendCfg:
        return L.FrontendCfg(self.sample_rate, self.n_fft, self.win_length, self.hop, int(self.center), self.pad_mode,
                             self.win_align, self.tail_min_samples, self.tail_hop_samples, self.spectrum,
                             self.mel_kind, self.n_mels, self.fmin, self.fmax, self.log_kind, self.amin, self.ref,
                             self.ln_offset, self.pcm_kind, self.pcm_divisor, self.window_kind, self.remove_dc,
                             self.preemphasis)

    @property
    def n_bins(self) -> int:
        return self.n_fft // 2 + 1

    # ---- the reference's four flavours ----------------------------------------------------------------
    @staticmethod
    def htsat(sample_rate=32000, window_size=1024, hop_size=320, mel_bins=64, fmin=50, fmax=14000, **kw):
        """external/hts_audio_transformer/config.py:53-59 + model/htsat.py:445-459."""
        return FrontendConfig(sample_rate, window_size, window_size, hop_size, True, L.PAD_REFLECT, n_mels=mel_bins,
                              fmin=float(fmin), fmax=float(fmax), log_kind=L.LOG_DB10, amin=1e-10, ref=1.0, **kw)

    @staticmethod
    def yamnet(**kw):
        """google/yamnet/1 feature parameters (16 kHz, 25/10 ms, 64 mel 125..7500 Hz, log offset 0.001)."""
        return FrontendConfig(16000, 512, 400, 160, False, L.PAD_NONE, L.WIN_LEFT, 15600, 7680, L.SPEC_MAGNITUDE,
                              L.MEL_HTK_TF, 64, 125.0, 7500.0, L.LOG_LN, ln_offset=0.001, **kw)

    @staticmethod
    def crnn(sr=44100, nfft=2048, nb_mel=40, hop_len=1024, **kw):
        """src/models/crnn/process_audio.py:29-34 as called at :94 (librosa >= 0.10 pads with zeros)."""
        return FrontendConfig(sr, nfft, nfft, hop_len, True, L.PAD_CONSTANT, spectrum=L.SPEC_MAGNITUDE,
                              n_mels=nb_mel, fmin=0.0, fmax=sr / 2.0, log_kind=L.LOG_LN, ln_offset=1e-9, **kw)

    @staticmethod
    def db_mel(sr=44100, n_fft=2048, hop_length=512, n_mels=64, **kw):
        """src/utils/audio_to_spectrograms.py:30-44 (the per-clip max/top_db pass is applied by the caller)."""
        return FrontendConfig(sr, n_fft, n_fft, hop_length, True, L.PAD_CONSTANT, n_mels=n_mels, fmin=0.0,
                              fmax=float(sr // 2), log_kind=L.LOG_DB10, amin=1e-10, ref=1.0, **kw)


    @staticmethod
    def kaldi_fbank(sample_frequency=16000.0, num_mel_bins=128, frame_length=25.0, frame_shift=10.0, low_freq=20.0,
                    high_freq=0.0, preemphasis_coefficient=0.97, remove_dc_offset=True, **kw):
        """torchaudio.compliance.kaldi.fbank as called at external/audio_mamba/src/dataloader.py:133-134
        (htk_compat=True, use_energy=False, window_type='hanning', dither=0, snip_edges, power spectrum, log)."""
        sr = int(round(sample_frequency))
        win = int(sr * frame_length * 0.001)
        hop = int(sr * frame_shift * 0.001)
        n_fft = 1 << max(1, (win - 1).bit_length())
        fmax = float(high_freq) if high_freq > 0 else float(high_freq) + 0.5 * sr
        return FrontendConfig(sr, n_fft, win, hop, False, L.PAD_NONE, L.WIN_LEFT, spectrum=L.SPEC_POWER,
                              mel_kind=L.MEL_KALDI, n_mels=num_mel_bins, fmin=float(low_freq), fmax=fmax,
                              log_kind=L.LOG_LN_FLOOR, amin=float(np.finfo(np.float32).eps),
                              window_kind=L.WINDOW_HANN_SYMMETRIC, remove_dc=int(bool(remove_dc_offset)),
                              preemphasis=float(preemphasis_coefficient), **kw)


def mel_filterbank(cfg: FrontendConfig) -> torch.Tensor:
    """(F, n_mels) fp32 CPU tensor: the matrix logmel_extractor.melW holds (host-side table builder)."""
    out = np.empty((cfg.n_bins, cfg.n_mels), dtype=np.float32)
    L.check(L.lib.sed_mel_filterbank(cfg.mel_kind, cfg.sample_rate, cfg.n_fft, cfg.n_mels, cfg.fmin, cfg.fmax,
                                     out.ctypes.data), "sed_mel_filterbank")
    return torch.from_numpy(out)


def hann_window(n_fft: int, win_length: int, win_align: int = L.WIN_CENTER,
                window_kind: int = L.WINDOW_HANN_PERIODIC) -> torch.Tensor:
    out = np.empty(n_fft, dtype=np.float32)
    L.check(L.lib.sed_window(window_kind, n_fft, win_length, win_align, out.ctypes.data), "sed_window")
    return torch.from_numpy(out)


class _Plan:
    """Owns one sed_plan on one device."""

    def __init__(self, cfg: FrontendConfig, device: torch.device):
        self.cfg, self.device = cfg, device
        self.ptr = C.c_void_p()
        c = cfg.to_c()
        with torch.cuda.device(device):
            L.check(L.lib.sed_plan_create(C.byref(c), C.byref(self.ptr)), "sed_plan_create")
            if cfg.window_kind == L.WINDOW_HANN_SYMMETRIC and cfg.win_align == L.WIN_LEFT:
                # the Kaldi flavour: the very window torchaudio builds (torch.hann_window in fp32), bit for bit
                w = torch.zeros(cfg.n_fft, dtype=torch.float32)
                w[:cfg.win_length] = torch.hann_window(cfg.win_length, periodic=False, dtype=torch.float32)
                L.check(L.lib.sed_plan_set_window(self.ptr, w.data_ptr()), "sed_plan_set_window")

    def num_frames(self, n_samples: int) -> int:
        t = int(L.lib.sed_plan_num_frames(self.ptr, int(n_samples)))
        L.check(t, "sed_plan_num_frames")
        return t

    def set_window(self, window_full) -> None:
        """Replace the plan's window by n_fft fp32 values (already padded to n_fft)."""
        w = torch.as_tensor(np.asarray(window_full, dtype=np.float32)).contiguous()
        if w.numel() != self.cfg.n_fft:
            raise ValueError(f"window must have n_fft = {self.cfg.n_fft} values")
        with torch.cuda.device(self.device):
            L.check(L.lib.sed_plan_set_window(self.ptr, w.data_ptr()), "sed_plan_set_window")

    def set_mel_weights(self, melW: torch.Tensor) -> None:
        w = melW.detach().to("cpu", torch.float32).contiguous()
        if tuple(w.shape) != (self.cfg.n_bins, self.cfg.n_mels):
            raise ValueError(f"melW must be {(self.cfg.n_bins, self.cfg.n_mels)}, got {tuple(w.shape)}")
        with torch.cuda.device(self.device):
            L.check(L.lib.sed_plan_set_mel_weights(self.ptr, w.data_ptr()), "sed_plan_set_mel_weights")

    def set_affine(self, scale: Optional[torch.Tensor], shift: Optional[torch.Tensor]) -> None:
        with torch.cuda.device(self.device):
            if scale is None:
                L.check(L.lib.sed_plan_set_affine(self.ptr, None, None), "sed_plan_set_affine")
                return
            s = scale.detach().to("cpu", torch.float32).contiguous()
            b = shift.detach().to("cpu", torch.float32).contiguous()
            if s.numel() != self.cfg.n_mels or b.numel() != self.cfg.n_mels:
                raise ValueError("affine scale/shift must have n_mels elements")
            L.check(L.lib.sed_plan_set_affine(self.ptr, s.data_ptr(), b.data_ptr()), "sed_plan_set_affine")

    def __deepcopy__(self, memo):
        return None  # plans are per-process device objects; a copied module builds its own on first use

    def __del__(self):
        try:
            if self.ptr:
                L.lib.sed_plan_destroy(self.ptr)
                self.ptr = C.c_void_p()
        except Exception:
            pass


def _require_cuda(x: torch.Tensor, what: str) -> None:
    if not x.is_cuda:
        raise RuntimeError(f"{what}: input is on {x.device}; the front-end runs only on a ROCm device (no CPU path)")


def _prep_pcm(x: torch.Tensor, cfg: FrontendConfig) -> torch.Tensor:
    want = torch.int16 if cfg.pcm_kind == L.PCM_I16 else torch.float32
    if x.dim() != 2:
        raise RuntimeError(f"expected a (batch, samples) tensor, got shape {tuple(x.shape)}")
    if x.dtype != want:
        raise RuntimeError(f"expected dtype {want}, got {x.dtype}")
    if x.stride(1) != 1 or x.stride(0) < x.shape[1]:
        x = x.contiguous()
    return x


class _PlanHolder(nn.Module):
    """Lazily creates one plan per device and keeps it in step with the module's tables."""

    def __init__(self):
        super().__init__()
        self._plans = {}

    def _cfg(self) -> FrontendConfig:
        raise NotImplementedError

    def _sync_plan(self, plan: _Plan) -> None:
        """Hook: push module tables (mel matrix, affine) into a fresh or stale plan."""

    def _tables_version(self):
        return None

    def _plan_for(self, device: torch.device) -> _Plan:
        key = device.index if device.index is not None else torch.cuda.current_device()
        ent = self._plans.get(key)
        ver = self._tables_version()
        if ent is None or ent[0] is None or ent[1] != ver:
            plan = ent[0] if ent is not None and ent[0] is not None else _Plan(self._cfg(), torch.device("cuda", key))
            self._sync_plan(plan)
            self._plans[key] = (plan, ver)
            return plan
        return ent[0]


def _run_front(plan: _Plan, x: torch.Tensor, spectrogram: bool, out: torch.Tensor = None) -> torch.Tensor:
    """out: a contiguous fp32 (B, rows >= T, row) tensor whose clip slots the kernel writes in place; the view out[:, :T] is
    returned."""
    cfg = plan.cfg
    B, n = x.shape
    T = plan.num_frames(n)
    row = cfg.n_bins if spectrogram else cfg.n_mels
    if out is None:
        out = torch.empty((B, T, row), dtype=torch.float32, device=x.device)
    elif not (out.device == x.device and out.dtype == torch.float32 and out.dim() == 3 and out.shape[0] == B and
              out.shape[1] >= T and out.shape[2] == row and out.is_contiguous()):
        raise ValueError(f"out: a contiguous fp32 ({B}, >= {T}, {row}) tensor on the input's device")
    if B == 0 or T == 0:
        return out[:, :T]
    fn = L.lib.sed_spectrogram_forward if spectrogram else L.lib.sed_logmel_forward
    with torch.cuda.device(x.device):
        L.check(fn(plan.ptr, x.data_ptr(), B, n, max(x.stride(0), n), out.data_ptr(), out.shape[1] * row,
                   L.stream_ptr(x.device)), "sed_spectrogram_forward" if spectrogram else "sed_logmel_forward")
    return out[:, :T]


class _ConvWeight(nn.Module):
    """Holder for a frozen `weight` so that state_dict keys match torchlibrosa's conv_real / conv_imag."""

    def __init__(self, w: torch.Tensor):
        super().__init__()
        self.weight = nn.Parameter(w, requires_grad=False)


class _STFTTables(nn.Module):
    """stft.conv_real.weight / stft.conv_imag.weight as torchlibrosa's STFT module holds them (F, 1, n_fft): the Hann-windowed
    DFT matrix, formed with the numpy expressions of torchlibrosa's DFTBase.dft_matrix / STFT.__init__ (omega ** (x * y) in
    complex128, times the window, real and imaginary parts cast to fp32), so that a state_dict saved from this module
    carries the values a torchlibrosa checkpoint carries (htsat.py:453-455, src/models/htsat/model.py:214-215).  About a
    quarter of the entries differ by one fp32 ulp from the exactly reduced angle cos/sin(2 pi (n k mod N) / N).  The kernels
    do not read these tensors."""

    _cache: dict = {}   # (n_fft, win_length, window key) -> (real, imag) fp32 arrays (F, 1, n_fft), shared by construction and load checks

    @classmethod
    def weights(cls, n_fft: int, win_length: int, window="hann"):
        key = (int(n_fft), int(win_length), _window_key(window))
        hit = cls._cache.get(key)
        if hit is None:
            window_full = fft_window(window, win_length, n_fft)
            # columns 0 .. n_fft/2 of torchlibrosa's n_fft x n_fft matrix, the only ones it keeps (same elementwise expressions)
            (x, y) = np.meshgrid(np.arange(n_fft // 2 + 1), np.arange(n_fft))
            omega = np.exp(-2 * np.pi * 1j / n_fft)
            W = np.power(omega, x * y) * window_full[:, None]
            hit = (np.ascontiguousarray(np.real(W).T.astype(np.float32)[:, None, :]),
                   np.ascontiguousarray(np.imag(W).T.astype(np.float32)[:, None, :]))
            for a in hit:
                a.setflags(write=False)
            if key[2] is not None:
                cls._cache[key] = hit
        return hit

    def __init__(self, n_fft: int, win_length: int, window="hann"):
        super().__init__()
        real, imag = self.weights(n_fft, win_length, window)
        self.conv_real = _ConvWeight(torch.from_numpy(real.copy()))
        self.conv_imag = _ConvWeight(torch.from_numpy(imag.copy()))


def _window_key(window):
    """Hashable name of a window specification, None for arrays and callables (not cached)."""
    if isinstance(window, str):
        return window
    if isinstance(window, (int, float)):
        return float(window)
    if isinstance(window, tuple) and all(isinstance(a, (str, int, float)) for a in window):
        return window
    return None


def fft_window(window, win_length: int, n_fft: int) -> np.ndarray:
    """The fp64 window torchlibrosa's STFT multiplies into its DFT matrix (stft.py: librosa.filters.get_window(window,
    win_length, fftbins=True), then librosa.util.pad_center to n_fft): a name / tuple / number goes to
    scipy.signal.get_window as librosa passes it on, a callable is called with win_length, an array is taken as it is."""
    win_length, n_fft = int(win_length), int(n_fft)
    if isinstance(window, str) and window == "hann":
        # scipy's general_cosine for 'hann' written out: 0.5 + 0.5 cos(fac), fac = linspace(-pi, pi, win_length + 1)
        # without its last point
        fac = np.linspace(-np.pi, np.pi, win_length + 1)[:-1]
        win = np.zeros(win_length)
        for k, a in enumerate((0.5, 0.5)):
            win += a * np.cos(k * fac)
    elif callable(window):
        win = np.asarray(window(win_length), dtype=np.float64)
    elif isinstance(window, (str, tuple, int, float)):
        from scipy.signal import get_window
        win = np.asarray(get_window(window, win_length, fftbins=True), dtype=np.float64)
    else:
        win = np.asarray(window, dtype=np.float64)
    if win.shape != (win_length,):
        raise ValueError(f"window must have win_length = {win_length} samples, got shape {win.shape}")
    if not np.all(np.isfinite(win)):
        raise ValueError("window must be finite")
    lpad = (n_fft - win_length) // 2
    return np.pad(win, (lpad, n_fft - win_length - lpad))


class Spectrogram(_PlanHolder):
    """torchlibrosa.stft.Spectrogram: (B, L) -> (B, 1, T, n_fft//2+1), power spectrogram by default.

    The state_dict carries stft.conv_real.weight / stft.conv_imag.weight (F, 1, n_fft) like the reference
    module, so checkpoints saved from it load with strict=True; the kernels use their own twiddle tables and
    the loaded tensors are only checked for agreement with the analytic windowed DFT matrix.
    """

    def __init__(self, n_fft=2048, hop_length=None, win_length=None, window="hann", center=True, pad_mode="reflect",
                 power=2.0, freeze_parameters=True):
        super().__init__()
        assert pad_mode in ["constant", "reflect"]
        self.n_fft = n_fft
        self.win_length = n_fft if win_length is None else win_length
        self.hop_length = int(self.win_length // 4) if hop_length is None else hop_length
        self.center, self.pad_mode, self.power = center, pad_mode, power
        self.window = window
        self.stft = _STFTTables(n_fft, self.win_length, window)
        self.register_load_state_dict_post_hook(_check_stft_tables)

    def _sync_plan(self, plan: _Plan) -> None:
        if not (isinstance(self.window, str) and self.window == "hann"):  # the plan's own window is the periodic Hann
            plan.set_window(fft_window(self.window, self.win_length, self.n_fft))

    def _cfg(self) -> FrontendConfig:
        return FrontendConfig(1, self.n_fft, self.win_length, self.hop_length, bool(self.center),
                              L.PAD_REFLECT if self.pad_mode == "reflect" else L.PAD_CONSTANT,
                              spectrum=L.SPEC_MAGNITUDE if self.power == 1.0 else L.SPEC_POWER, n_mels=1,
                              fmin=0.0, fmax=0.5, log_kind=L.LOG_NONE)

    @torch.no_grad()
    def forward(self, input: torch.Tensor) -> torch.Tensor:
        _require_cuda(input, "Spectrogram")
        cfg = self._cfg()
        x = _prep_pcm(input, cfg)
        out = _run_front(self._plan_for(x.device), x, spectrogram=True)
        if self.power not in (1.0, 2.0):
            out = out ** (self.power / 2.0)
        return out[:, None, :, :]


def _check_stft_tables(module, incompatible_keys):
    window = getattr(module, "window", "hann")
    ref = dict(zip(("conv_real", "conv_imag"), _STFTTables.weights(module.n_fft, module.win_length, window)))
    for name in ("conv_real", "conv_imag"):
        got = getattr(module.stft, name).weight.detach().cpu()
        if got.shape != ref[name].shape or not torch.allclose(got, torch.from_numpy(ref[name].copy()), atol=1e-5, rtol=0):
            raise RuntimeError(f"loaded stft.{name}.weight is not the windowed DFT matrix this module computes")


class LogmelFilterBank(_PlanHolder):
    """torchlibrosa.stft.LogmelFilterBank: (..., T, F) power spectrogram -> (..., T, n_mels) log-mel."""

    def __init__(self, sr=22050, n_fft=2048, n_mels=64, fmin=0.0, fmax=None, is_log=True, ref=1.0, amin=1e-10,
                 top_db=80.0, freeze_parameters=True):
        super().__init__()
        self.sr, self.n_fft, self.n_mels = sr, n_fft, n_mels
        self.fmin = fmin
        self.fmax = sr // 2 if fmax is None else fmax
        self.is_log, self.ref, self.amin, self.top_db = is_log, ref, amin, top_db
        if top_db is not None and top_db < 0:
            raise ValueError("top_db must be non-negative")
        self.melW = nn.Parameter(mel_filterbank(self._cfg()), requires_grad=False)

    def _cfg(self) -> FrontendConfig:
        return FrontendConfig(int(self.sr), self.n_fft, self.n_fft, self.n_fft // 4, n_mels=self.n_mels,
                              fmin=float(self.fmin), fmax=float(self.fmax),
                              log_kind=L.LOG_DB10 if self.is_log else L.LOG_NONE, amin=float(self.amin),
                              ref=float(self.ref))

    def _tables_version(self):
        return self.melW._version

    def _sync_plan(self, plan: _Plan) -> None:
        plan.set_mel_weights(self.melW)

    @torch.no_grad()
    def forward(self, input: torch.Tensor) -> torch.Tensor:
        _require_cuda(input, "LogmelFilterBank")
        if input.shape[-1] != self.n_fft // 2 + 1:
            raise RuntimeError(f"last dimension must be {self.n_fft // 2 + 1}, got {input.shape[-1]}")
        x = input.to(torch.float32).contiguous()
        rows = x.numel() // x.shape[-1]
        out = torch.empty(x.shape[:-1] + (self.n_mels,), dtype=torch.float32, device=x.device)
        plan = self._plan_for(x.device)
        with torch.cuda.device(x.device):
            s = L.stream_ptr(x.device)
            L.check(L.lib.sed_mel_from_spectrogram(plan.ptr, x.data_ptr(), rows, out.data_ptr(), s),
                    "sed_mel_from_spectrogram")
            if self.is_log and self.top_db is not None:
                L.check(L.lib.sed_db_normalize(out.data_ptr(), 1, out.numel(), 0, float(self.top_db), s),
                        "sed_db_normalize")
        return out


class LogMelFrontend(_PlanHolder):
    """Fused front-end: waveform (B, L) -> log-mel, one kernel launch.

    layout: 'b1tm' (B, 1, T, M) as HTS-AT consumes it, 'btm' (B, T, M), 'bmt' (B, M, T) as librosa returns it.
    An optional per-mel affine (eval-mode BatchNorm over mel bins, or a fitted StandardScaler) is applied in the
    kernel's epilogue.
    """

    def __init__(self, cfg: FrontendConfig, layout: str = "b1tm", melW: Optional[torch.Tensor] = None,
                 mel_owner: Optional[nn.Module] = None, stft: Optional[nn.Module] = None, window="hann"):
        super().__init__()
        if layout not in ("b1tm", "btm", "bmt"):
            raise ValueError(layout)
        self.cfg, self.layout = cfg, layout
        # any other window than the configuration's own Hann: torchlibrosa's window argument (see fft_window)
        self.window = window
        if not (isinstance(window, str) and window == "hann") and cfg.win_align != L.WIN_CENTER:
            raise ValueError("a window argument goes with the centred (torchlibrosa / librosa) window placement")
        if mel_owner is None:
            self.melW = nn.Parameter(mel_filterbank(cfg) if melW is None else melW.detach().clone().float().cpu(),
                                     requires_grad=False)
        else:
            # the mel matrix stays a parameter of another module (so that its state_dict key does not move); kept
            # out of this module's registry, or the key would appear twice
            object.__setattr__(self, "_mel_owner", mel_owner)
        if stft is not None:
            # the windowed-DFT weights a torchlibrosa checkpoint holds under spectrogram_extractor.stft.*: carried
            # for the state_dict keys and checked on load; the kernel has its own twiddle tables
            self.stft = stft
            self.n_fft, self.win_length = cfg.n_fft, cfg.win_length
            self.register_load_state_dict_post_hook(_check_stft_tables)
        # eval-mode bn0 folded into the epilogue: derived from bn0's own entries, so not part of the state_dict
        self.register_buffer("affine_scale", None, persistent=False)
        self.register_buffer("affine_shift", None, persistent=False)
        self._affine_version = 0

    def _mel(self) -> torch.Tensor:
        owner = self.__dict__.get("_mel_owner")
        return self.melW if owner is None else owner.melW

    @classmethod
    def from_torchlibrosa(cls, spectrogram_extractor: Spectrogram, logmel_extractor: LogmelFilterBank,
                          layout: str = "b1tm", mel_owner: Optional[nn.Module] = None,
                          keep_stft: bool = False) -> "LogMelFrontend":
        """Fuse a Spectrogram + LogmelFilterBank pair (as held by HTSAT_Swin_Transformer) into one module."""
        s, m = spectrogram_extractor, logmel_extractor
        if s.power != 2.0 or not m.is_log or m.top_db is not None or s.n_fft != m.n_fft:
            raise ValueError("only power=2.0, is_log=True, top_db=None pairs with equal n_fft can be fused")
        cfg = FrontendConfig(int(m.sr), s.n_fft, s.win_length, s.hop_length, bool(s.center),
                             L.PAD_REFLECT if s.pad_mode == "reflect" else L.PAD_CONSTANT, n_mels=m.n_mels,
                             fmin=float(m.fmin), fmax=float(m.fmax), log_kind=L.LOG_DB10, amin=float(m.amin),
                             ref=float(m.ref))
        return cls(cfg, layout, melW=None if mel_owner is not None else m.melW, mel_owner=mel_owner,
                   stft=s.stft if keep_stft else None, window=getattr(s, "window", "hann"))

    def set_affine(self, scale: Optional[torch.Tensor], shift: Optional[torch.Tensor]) -> None:
        self.affine_scale = None if scale is None else scale.detach().clone().float()
        self.affine_shift = None if shift is None else shift.detach().clone().float()
        self._affine_version += 1

    def set_batchnorm(self, bn: nn.BatchNorm2d) -> None:
        """Fold an eval-mode BatchNorm2d over mel bins (htsat.py:463,767-769) into the epilogue."""
        inv = torch.rsqrt(bn.running_var.detach().float() + bn.eps)
        g = bn.weight.detach().float() if bn.affine else torch.ones_like(inv)
        b = bn.bias.detach().float() if bn.affine else torch.zeros_like(inv)
        self.set_affine(g * inv, b - bn.running_mean.detach().float() * g * inv)

    def _cfg(self) -> FrontendConfig:
        return self.cfg

    def _tables_version(self):
        mel = self._mel()
        return (id(mel), mel._version, self._affine_version)

    def _sync_plan(self, plan: _Plan) -> None:
        if not (isinstance(self.window, str) and self.window == "hann"):
            plan.set_window(fft_window(self.window, self.cfg.win_length, self.cfg.n_fft))
        plan.set_mel_weights(self._mel())
        plan.set_affine(self.affine_scale, self.affine_shift)

    def num_frames(self, n_samples: int, device=None) -> int:
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        return self._plan_for(dev).num_frames(n_samples)

    @torch.no_grad()
    def forward(self, input: torch.Tensor) -> torch.Tensor:
        _require_cuda(input, "LogMelFrontend")
        x = _prep_pcm(input, self.cfg)
        out = _run_front(self._plan_for(x.device), x, spectrogram=False)
        if self.layout == "b1tm":
            return out[:, None, :, :]
        if self.layout == "bmt":
            return out.transpose(1, 2)
        return out

    @torch.no_grad()
    def forward_segments(self, pcm: torch.Tensor, seg_off, seg_len, n_samples: int, own_edges: bool = False,
                         out: torch.Tensor = None):
        """Log-mel of segments of a resident PCM buffer (the hand-off of handoff.logmel_of_events): clip e is the seg_len[e]
        samples from ELEMENT seg_off[e] of pcm's storage on, zero-padded to n_samples.  seg_off / seg_len: integer sequences
        or arrays on the host (checked against the buffer, clamped to n_samples and uploaded in one asynchronous copy), or
        int64 / int32 tensors already on the device (taken as they are).  Returns (log-mel in the module's layout, lengths (E,)
        int32 on the device = the sample counts before clamping, for host arrays).

        own_edges=True frames every segment as a recording of its own length, as the reference's second stage sees an event
        file loaded alone (src/models/htsat/data.py:138-146, batch_size = 1 at src/models/htsat/classification.py:55-59): the
        STFT's padding rule applies at the segment's own end and the segment has its own number of frames.  Returns
        (log-mel, lengths, frames): frames (E,) int32 on the device, rows [0, frames[e]) of clip e hold its log-mel (the
        values the front-end gives for those samples alone) and the rows behind them are zero.  A segment too short for one
        frame has frames[e] = 0.

        out (layout "btm" only): a contiguous fp32 (E, rows >= frames of n_samples, n_mels) tensor on pcm's device whose
        clip slots the kernel writes in place; the returned log-mel is the view out[:, :frames of n_samples]."""
        _require_cuda(pcm, "LogMelFrontend.forward_segments")
        want = torch.int16 if self.cfg.pcm_kind == L.PCM_I16 else torch.float32
        if pcm.dtype != want:
            raise RuntimeError(f"expected dtype {want}, got {pcm.dtype}")
        if not pcm.is_contiguous() and not (pcm.dim() == 2 and pcm.stride(1) == 1):
            raise RuntimeError("forward_segments: pcm must be contiguous along the samples")
        n_samples = int(n_samples)
        plan = self._plan_for(pcm.device)
        T = plan.num_frames(n_samples)
        if torch.is_tensor(seg_off) or torch.is_tensor(seg_len):
            # arrays already on the device (prepared once for several calls): seg_len must already be clamped to n_samples
            if not (torch.is_tensor(seg_off) and torch.is_tensor(seg_len) and seg_off.device == pcm.device and
                    seg_len.device == pcm.device and seg_off.dtype == torch.int64 and seg_len.dtype == torch.int32 and
                    seg_off.dim() == 1 and seg_off.shape == seg_len.shape and seg_off.is_contiguous() and seg_len.is_contiguous()):
                raise TypeError("device-side segments: contiguous int64 offsets and int32 lengths of one shape on pcm's device")
            d_off, d_len, lengths, E = seg_off, seg_len, seg_len, seg_off.shape[0]
            frames = self._own_frames(d_len.clamp(max=n_samples)) if own_edges else None
        else:
            off = np.ascontiguousarray(seg_off, dtype=np.int64).reshape(-1)
            full = np.ascontiguousarray(seg_len, dtype=np.int64).reshape(-1)
            if off.shape != full.shape:
                raise ValueError("seg_off and seg_len must have one entry per segment")
            E = off.shape[0]
            # the storage the segments may address: from pcm's first element to the end of its last row
            extent = (pcm.shape[0] - 1) * pcm.stride(0) + pcm.shape[1] if pcm.dim() == 2 and pcm.shape[0] > 0 else pcm.numel()
            if E and (off.min() < 0 or full.min() < 0 or (off + full).max() > extent):
                raise ValueError("a segment lies outside the PCM buffer")
            # one asynchronous upload from pinned memory: [offsets int64 | clamped lengths int32 | full lengths int32 | own frames int32]
            stage = torch.empty(20 * max(E, 1), dtype=torch.uint8, pin_memory=True)
            host = stage.numpy()
            clamped = np.minimum(full, n_samples)
            host[:8 * E] = off.view(np.uint8)
            host[8 * E:12 * E] = clamped.astype(np.int32).view(np.uint8)
            host[12 * E:16 * E] = np.minimum(full, 2 ** 31 - 1).astype(np.int32).view(np.uint8)
            host[16 * E:20 * E] = self._own_frames(clamped).astype(np.int32).view(np.uint8)
            d_stage = stage.to(pcm.device, non_blocking=True)
            d_off, d_len = d_stage[:8 * E].view(torch.int64), d_stage[8 * E:12 * E].view(torch.int32)
            lengths = d_stage[12 * E:16 * E].view(torch.int32)
            frames = d_stage[16 * E:20 * E].view(torch.int32)
        if out is None:
            out = torch.empty((E, T, self.cfg.n_mels), dtype=torch.float32, device=pcm.device)
        else:
            if self.layout != "btm":
                raise ValueError("forward_segments(out=...) is for the (clips, frames, mel) layout")
            if not (torch.is_tensor(out) and out.device == pcm.device and out.dtype == torch.float32 and out.dim() == 3 and
                    out.shape[0] == E and out.shape[1] >= T and out.shape[2] == self.cfg.n_mels and out.is_contiguous()):
                raise ValueError(f"out: a contiguous fp32 ({E}, >= {T}, {self.cfg.n_mels}) tensor on pcm's device")
        rows = out.shape[1]
        if own_edges:
            entry, name = L.lib.sed_logmel_forward_segments_own, "sed_logmel_forward_segments_own"
        else:
            entry, name = L.lib.sed_logmel_forward_segments, "sed_logmel_forward_segments"
        if E and T:
            with torch.cuda.device(pcm.device):
                L.check(entry(plan.ptr, pcm.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), E, n_samples, out.data_ptr(),
                              rows * self.cfg.n_mels, L.stream_ptr(pcm.device)), name)
        out = out[:, :T]
        if self.layout == "b1tm":
            out = out[:, None, :, :]
        elif self.layout == "bmt":
            out = out.transpose(1, 2)
        if own_edges:
            return out, lengths, frames
        return out, lengths

    def _own_frames(self, n):
        """Frames of segments of n samples framed as recordings of their own (the kernel's rule); n: a numpy array or a
        tensor of sample counts already clamped to the row, the result is of the same kind."""
        c = self.cfg
        if c.center:
            sub, least = 0, (c.n_fft // 2 + 1 if c.pad_mode == L.PAD_REFLECT else 1)
        else:
            sub = c.win_length if c.win_align == L.WIN_LEFT else c.n_fft
            least = sub
        if torch.is_tensor(n):
            return torch.where(n >= least, 1 + (n - sub) // c.hop, torch.zeros_like(n)).to(torch.int32)
        return np.where(n >= least, 1 + (n - sub) // c.hop, 0)


class _LogmelPassthrough(nn.Module):
    """Stands where logmel_extractor stood once the fused kernel has taken over its work: forward is the identity,
    and it still owns `melW`, so a checkpoint's logmel_extractor.melW keeps loading into the matrix the kernel uses."""

    def __init__(self, melW: torch.Tensor):
        super().__init__()
        self.melW = nn.Parameter(melW.detach().clone(), requires_grad=False)

    def forward(self, x):
        return x


def install_fused_frontend(model: nn.Module) -> nn.Module:
    """Swap model.spectrogram_extractor / model.logmel_extractor (the attributes HTSAT_Swin_Transformer.forward
    calls at htsat.py:763-764) for the fused kernel + an identity, leaving forward() untouched.

    The state_dict keys do not change (spectrogram_extractor.stft.conv_real.weight, ...conv_imag.weight,
    logmel_extractor.melW), so this may be called before or after the strict load_state_dict of a reference
    checkpoint (src/models/htsat/model.py:159-162): a mel matrix loaded afterwards is the one the kernel uses."""
    s, m = model.spectrogram_extractor, model.logmel_extractor
    holder = _LogmelPassthrough(m.melW)
    fused = LogMelFrontend.from_torchlibrosa(s, m, mel_owner=holder, keep_stft=True)
    model.spectrogram_extractor = fused
    model.logmel_extractor = holder
    return model


# ---- functional forms of the other flavours --------------------------------------------------------------

_FUNCTIONAL = {}


def _functional(cfg: FrontendConfig, layout: str) -> LogMelFrontend:
    key = (cfg, layout)
    if key not in _FUNCTIONAL:
        _FUNCTIONAL[key] = LogMelFrontend(cfg, layout)
    return _FUNCTIONAL[key]


def _as_cuda_batch(y, dtype=torch.float32):
    t = torch.as_tensor(y)
    squeeze = t.dim() == 1
    if squeeze:
        t = t[None]
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError("no ROCm device is visible; the front-end has no CPU path")
        t = t.to("cuda")
    return t.to(dtype), squeeze


def yamnet_log_mel(waveform):
    """(L,) or (B, L) fp32 16 kHz mono in [-1, 1] -> (log_mel (B, T, 64), patches (B, N, 96, 64)).

    patches are overlapping views (0.96 s window, 0.48 s hop) of log_mel, not copies.
    """
    x, squeeze = _as_cuda_batch(waveform)
    lm = _functional(FrontendConfig.yamnet(), "btm")(x)
    patches = lm.unfold(1, 96, 48).permute(0, 1, 3, 2) if lm.shape[1] >= 96 else lm.new_empty((lm.shape[0], 0, 96, 64))
    if squeeze:
        return lm[0], patches[0]
    return lm, patches


def kaldi_fbank(waveform, sample_frequency=16000.0, num_mel_bins=23, frame_length=25.0, frame_shift=10.0,
                low_freq=20.0, high_freq=0.0, preemphasis_coefficient=0.97, remove_dc_offset=True, dither=0.0,
                use_energy=False, htk_compat=False, window_type="hanning", **unsupported):
    """torchaudio.compliance.kaldi.fbank for the options the reference uses (external/audio_mamba/src/
    dataloader.py:133-134): waveform (1, L) -> (T, num_mel_bins); a batch (B, L) with B > 1 -> (B, T, num_mel_bins).

    T = 1 + (L - window) // shift (snip_edges); every other option of torchaudio's function must keep its default.
    """
    if unsupported or dither != 0.0 or use_energy or window_type != "hanning":
        raise ValueError("kaldi_fbank supports dither=0, use_energy=False, window_type='hanning' and default "
                         f"values elsewhere (got {sorted(unsupported)})")
    x, _ = _as_cuda_batch(waveform)
    cfg = FrontendConfig.kaldi_fbank(sample_frequency, num_mel_bins, frame_length, frame_shift, low_freq, high_freq,
                                     preemphasis_coefficient, remove_dc_offset)
    if x.shape[1] < cfg.win_length:
        return x.new_empty((0, num_mel_bins)) if x.shape[0] == 1 else x.new_empty((x.shape[0], 0, num_mel_bins))
    out = _functional(cfg, "btm")(x)
    return out[0] if x.shape[0] == 1 else out


def wav2fbank(waveform, sr, melbins=128, target_length=1024, fshift=10, norm_mean=None, norm_std=None, lengths=None):
    """AudiosetDataset._wav2fbank + the normalisation of __getitem__ (external/audio_mamba/src/dataloader.py:94-152,
    219-221) for one clip (1, L) or a batch (B, L): mean removal, Kaldi fbank, zero-pad / cut to target_length frames,
    then (x - norm_mean) / (2 norm_std) when both are given.  Returns (B, target_length, melbins).

    lengths (B,): the clips are events of different lengths in a zero-padded batch (what cut_events returns).  The
    reference reads every event from a file of its own, so a clip of n samples has 1 + (n - window) // shift frames and
    zeros behind them; the same here: the mean is that of the clip's own samples, and frames that reach past them are
    zero before the normalisation (frames in front of them only ever read the clip's own samples)."""
    x, _ = _as_cuda_batch(waveform)
    if lengths is None:
        x = x - x.mean(dim=1, keepdim=True)
    else:
        n_valid = torch.as_tensor(lengths, device=x.device).to(torch.int64).clamp(min=0, max=x.shape[1])
        if n_valid.shape != (x.shape[0],):
            raise ValueError("lengths must hold one sample count per clip")
        inside = torch.arange(x.shape[1], device=x.device)[None, :] < n_valid[:, None]
        x = torch.where(inside, x, torch.zeros((), dtype=x.dtype, device=x.device))
        x = x - x.sum(dim=1, keepdim=True) / n_valid.clamp(min=1)[:, None].to(x.dtype)
    cfg = FrontendConfig.kaldi_fbank(sr, melbins, 25.0, fshift)
    plan = _functional(cfg, "btm")._plan_for(x.device)
    B, target_length = x.shape[0], int(target_length)
    T = plan.num_frames(x.shape[1]) if x.shape[1] >= cfg.win_length else 0
    if T <= target_length:
        # the kernel writes into the result; only the rows it wrote are masked and normalised, the zero frames behind them
        # are filled with the value the same arithmetic gives a zero
        fb = torch.empty((B, target_length, int(melbins)), dtype=torch.float32, device=x.device)
        head = _run_front(plan, x, spectrogram=False, out=fb) if T else fb[:, :0]
    else:
        fb = head = _run_front(plan, x, spectrogram=False)[:, :target_length]
    if lengths is not None and head.shape[1]:
        frames = torch.where(n_valid >= cfg.win_length, 1 + (n_valid - cfg.win_length) // cfg.hop, torch.zeros_like(n_valid))
        gone = torch.arange(head.shape[1], device=x.device)[None, :] >= frames[:, None]
        head.masked_fill_(gone[:, :, None], 0.0)
    fill = torch.zeros((), dtype=torch.float32, device=x.device)
    if norm_mean is not None and norm_std is not None:
        head.sub_(norm_mean).div_(norm_std * 2)
        fill = (fill - norm_mean) / (norm_std * 2)
    if head.shape[1] < fb.shape[1]:
        fb[:, head.shape[1]:] = fill
    return fb


def reshape_wav2img(x: torch.Tensor, spec_size: int = 256, freq_ratio: int = None, frames: torch.Tensor = None) -> torch.Tensor:
    """HTSAT_Swin_Transformer.reshape_wav2img (external/hts_audio_transformer/model/htsat.py:729-744):
    x (B, 1, T, F) log-mel -> (B, 1, spec_size, spec_size) for F == spec_size // freq_ratio (freq_ratio defaults to
    spec_size // F as at htsat.py:444).  T < spec_size*freq_ratio is interpolated (bicubic, align_corners=True).

    frames (B,) int32 on the device: clip b holds frames[b] <= T rows of its own (events of different lengths in one batch,
    LogMelFrontend.forward_segments(..., own_edges=True)), and each clip is stretched to the image from its own rows, as the
    reference does for an event file fed alone.  A clip with no frame gives zeros."""
    _require_cuda(x, "reshape_wav2img")
    if x.dim() != 4 or x.shape[1] != 1:
        raise RuntimeError(f"expected a (batch, 1, time, mel) tensor, got shape {tuple(x.shape)}")
    B, _, T, F = x.shape
    if freq_ratio is None:
        freq_ratio = spec_size // F
    target_T, target_F = spec_size * freq_ratio, spec_size // freq_ratio
    assert T <= target_T and F <= target_F, "the wav size should less than or equal to the swin input size"
    if F != target_F:
        raise ValueError("reshape_wav2img: the mel axis must already have spec_size // freq_ratio bins")
    x = x.to(torch.float32).contiguous()
    img = torch.empty((B, 1, freq_ratio * F, target_T // freq_ratio), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        if frames is None:
            L.check(L.lib.sed_htsat_wav2img(x.data_ptr(), B, T, F, target_T, freq_ratio, img.data_ptr(),
                                            L.stream_ptr(x.device)), "sed_htsat_wav2img")
        else:
            if not (torch.is_tensor(frames) and frames.device == x.device and frames.dtype == torch.int32 and
                    frames.shape == (B,) and frames.is_contiguous()):
                raise TypeError("frames: a contiguous int32 tensor with one entry per clip on x's device")
            L.check(L.lib.sed_htsat_wav2img_ragged(x.data_ptr(), frames.data_ptr(), B, T, F, target_T, freq_ratio,
                                                   img.data_ptr(), L.stream_ptr(x.device)), "sed_htsat_wav2img_ragged")
    return img


def repeat_wat2img(x: torch.Tensor, cur_pos: int, spec_size: int = 256, freq_ratio: int = None) -> torch.Tensor:
    """HTSAT_Swin_Transformer.repeat_wat2img (external/hts_audio_transformer/model/htsat.py:747-760): the interpolated
    log-mel's frames [cur_pos, cur_pos + spec_size), mel axis first, stacked 4 times: (B, 1, T, F) -> (B, 1, 4F, spec_size)."""
    _require_cuda(x, "repeat_wat2img")
    if x.dim() != 4 or x.shape[1] != 1:
        raise RuntimeError(f"expected a (batch, 1, time, mel) tensor, got shape {tuple(x.shape)}")
    B, _, T, F = x.shape
    if freq_ratio is None:
        freq_ratio = spec_size // F
    target_T, target_F = spec_size * freq_ratio, spec_size // freq_ratio
    assert T <= target_T and F <= target_F, "the wav size should less than or equal to the swin input size"
    if F != target_F:
        raise ValueError("repeat_wat2img: the mel axis must already have spec_size // freq_ratio bins")
    cur_pos = int(cur_pos)
    width = max(0, min(spec_size, target_T - cur_pos))               # a slice past the end is short, as in torch
    if cur_pos < 0 or width < spec_size:
        raise ValueError(f"repeat_wat2img: cur_pos {cur_pos} leaves fewer than {spec_size} frames of {target_T}")
    x = x.to(torch.float32).contiguous()
    img = torch.empty((B, 1, 4 * F, spec_size), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        L.check(L.lib.sed_htsat_repeat2img(x.data_ptr(), B, T, F, target_T, spec_size, cur_pos, 4, img.data_ptr(),
                                           L.stream_ptr(x.device)), "sed_htsat_repeat2img")
    return img


def repeat_mode_images(x: torch.Tensor, spec_size: int = 256, freq_ratio: int = None):
    """The evaluation branch of enable_repeat_mode (external/hts_audio_transformer/model/htsat.py:789-793): one image
    per cur_pos in range(0, (freq_ratio - 1) * spec_size + 1, spec_size); the caller averages the model outputs."""
    if freq_ratio is None:
        freq_ratio = spec_size // x.shape[3]
    return [repeat_wat2img(x, pos, spec_size, freq_ratio) for pos in range(0, (freq_ratio - 1) * spec_size + 1, spec_size)]


def infer_mode_image(x: torch.Tensor, spec_size: int = 256, freq_ratio: int = None) -> torch.Tensor:
    """The infer_mode branch (external/hts_audio_transformer/model/htsat.py:775-781): the log-mel is tiled along time
    floor(target_T / T) times, then reshape_wav2img."""
    if freq_ratio is None:
        freq_ratio = spec_size // x.shape[3]
    repeat_ratio = (spec_size * freq_ratio) // x.shape[2]
    return reshape_wav2img(x.repeat(1, 1, repeat_ratio, 1), spec_size, freq_ratio)


def long_input_images(x: torch.Tensor, spec_size: int = 256, freq_ratio: int = None):
    """The evaluation branch of HTSAT_Swin_Transformer.forward for inputs longer than spec_size*freq_ratio frames
    (external/hts_audio_transformer/model/htsat.py:816-823): crops of (T - 1) // 2 frames every (T - 1) // 4 frames
    (crop_wav with spe_pos), each through reshape_wav2img.  x (B, 1, T, F) -> list of (B, 1, spec_size, spec_size);
    the caller averages the model outputs over the list as the reference does (:824-835)."""
    T = x.shape[2]
    overlap_size = (T - 1) // 4
    crop_size = (T - 1) // 2
    return [reshape_wav2img(x[:, :, pos:pos + crop_size, :], spec_size, freq_ratio)
            for pos in range(0, T - crop_size - 1, overlap_size)]


def extract_mbe(_y, _sr, _nfft, _nb_mel, _hop_len):
    """Log mel-band energies, (n_mels, T) (or (B, n_mels, T) for a batch); numpy in -> numpy out."""
    x, squeeze = _as_cuda_batch(_y)
    out = _functional(FrontendConfig.crnn(int(_sr), int(_nfft), int(_nb_mel), int(_hop_len)), "bmt")(x)
    if squeeze:
        out = out[0]
    return out.cpu().numpy() if not torch.is_tensor(_y) else out


def get_mel_spec_db(y, sr, n_fft=2048, hop_length=512, n_mels=64, top_db=80.0):
    """Power mel in dB relative to the clip's maximum, floored at -top_db: (n_mels, T) / (B, n_mels, T)."""
    x, squeeze = _as_cuda_batch(y)
    fe = _functional(FrontendConfig.db_mel(int(sr), n_fft, hop_length, n_mels), "btm")
    out = fe(x)
    with torch.cuda.device(out.device):
        L.check(L.lib.sed_db_normalize(out.data_ptr(), out.shape[0], out.shape[1] * out.shape[2], 1, float(top_db),
                                       L.stream_ptr(out.device)), "sed_db_normalize")
    out = out.transpose(1, 2)
    if squeeze:
        out = out[0]
    return out.cpu().numpy() if not torch.is_tensor(y) else out
