"""CPU oracle for the log-mel front-ends.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package never does.

PARITY UNPINNED for every function in this file: the reference holds no
front-end arithmetic and no tests of its own.  Each flavour below calls into a
third-party package that is neither vendored in the reference nor installed
here, so this file restates those packages' published algorithms and anchors
them on the reference's call sites:

  F1/F2  torchlibrosa==0.0.9  Spectrogram / LogmelFilterBank
         call site external/hts_audio_transformer/model/htsat.py:443-459,763-764
         (mel matrix: librosa.filters.mel, librosa<=0.11.0)
  F3     TF-Hub google/yamnet/1 (tensorflow/models research/audioset/yamnet
         params.py / features.py / yamnet.py:pad_waveform)
         call sites src/models/yamnet_train.py:46-47,
         src/models/single_stage_yamnet_frame.py:64-65
  F4     librosa.feature.melspectrogram(power=1.0) + np.log(S + 1e-9)
         call site src/models/crnn/process_audio.py:29-34
  F5     librosa.feature.melspectrogram(power=2.0) + power_to_db(ref=np.max)
         call site src/utils/audio_to_spectrograms.py:30-44

A second opinion, short of a pin: tests/test_oracle_vs_transformers.py holds the
mel scales and filterbanks, the framing options, the dB conversion and every
flavour's whole chain (the Kaldi filterbank of F6 below included) to
transformers.audio_utils, an installed numpy restatement of the same libraries
by other authors, to a few fp32 ulps.

Two evaluation modes per flavour:
  * gold  : fp64 arithmetic on the fp32 tables the reference would hold
            (window, mel matrix) -- the yardstick for the 1e-3 tolerance.
  * ref32 : the reference's own fp32 formulation (fp32 windowed-DFT matrix as
            conv weights, fp32 matmul, fp32 log) -- used for the deep-bin
            criterion and as the CPU baseline.
"""
from __future__ import annotations

import math

import numpy as np

# ----------------------------------------------------------------------------
# tables
# ----------------------------------------------------------------------------


def hann_periodic(win_length: int) -> np.ndarray:
    """The window torchlibrosa and librosa build: scipy.signal.get_window('hann', N, fftbins=True), fp64 (the call
    itself, so that the table is the reference's bit for bit; it is 0.5 - 0.5 cos(2 pi n / N) up to a few fp64 ulp)."""
    from scipy.signal import get_window
    return get_window("hann", int(win_length), fftbins=True).astype(np.float64)


def pad_center(w: np.ndarray, size: int) -> np.ndarray:
    """librosa.util.pad_center: zero-pad symmetrically (left = (size-n)//2)."""
    n = w.shape[0]
    lpad = (size - n) // 2
    out = np.zeros(size, dtype=w.dtype)
    out[lpad:lpad + n] = w
    return out


def hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    with np.errstate(divide="ignore", invalid="ignore"):
        logpart = min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep
    return np.where(f >= min_log_hz, logpart, mels)


def mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def librosa_mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None) -> np.ndarray:
    """librosa.filters.mel(htk=False, norm='slaney') -> (n_mels, 1+n_fft//2) fp32, operation for operation as
    librosa 0.10 / 0.11 forms it (the reference ran 0.11.0): bin frequencies from np.fft.rfftfreq (k * (1 / (n_fft * (1 / sr)))),
    the triangles stored into an fp32 array, then the Slaney area normalisation applied in place (`weights *= enorm`:
    the fp32 values times the fp64 factor, rounded to fp32 again), i.e. two roundings, not one."""
    if fmax is None:
        fmax = float(sr) / 2
    n_bins = 1 + n_fft // 2
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_pts = np.linspace(hz_to_mel_slaney(fmin), hz_to_mel_slaney(fmax), n_mels + 2)
    mel_f = mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, n_bins), dtype=np.float32)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


def tf_linear_to_mel_weight_matrix(num_mel_bins=64, num_spectrogram_bins=257, sample_rate=16000,
                                   lower_edge_hertz=125.0, upper_edge_hertz=7500.0) -> np.ndarray:
    """tf.signal.linear_to_mel_weight_matrix -> (num_spectrogram_bins, num_mel_bins) fp32.

    HTK mel 1127*ln(1+f/700); triangles linear in mel, peak 1, no area norm, DC row zero.
    Evaluated in fp64 and rounded once to fp32 (TF evaluates the same expressions in fp32).
    """
    def hz_to_mel(f):
        return 1127.0 * np.log1p(np.asarray(f, dtype=np.float64) / 700.0)

    nyquist = sample_rate / 2.0
    lin = np.linspace(0.0, nyquist, num_spectrogram_bins)[1:]
    spec_mel = hz_to_mel(lin)[:, None]
    edges = np.linspace(hz_to_mel(lower_edge_hertz), hz_to_mel(upper_edge_hertz), num_mel_bins + 2)
    lower, center, upper = edges[:-2][None, :], edges[1:-1][None, :], edges[2:][None, :]
    lower_slopes = (spec_mel - lower) / (center - lower)
    upper_slopes = (upper - spec_mel) / (upper - center)
    w = np.maximum(0.0, np.minimum(lower_slopes, upper_slopes))
    w = np.pad(w, [[1, 0], [0, 0]])
    return w.astype(np.float32)


# ----------------------------------------------------------------------------
# framing
# ----------------------------------------------------------------------------


def _frames(y: np.ndarray, n_fft: int, hop: int, n_frames: int) -> np.ndarray:
    """(..., L) -> (..., n_frames, n_fft) strided view copy."""
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    return y[..., idx]


def _pad_signal(x: np.ndarray, pad: int, mode: str) -> np.ndarray:
    if pad == 0:
        return x
    widths = [(0, 0)] * (x.ndim - 1) + [(pad, pad)]
    if mode == "reflect":
        return np.pad(x, widths, mode="reflect")
    if mode == "constant":
        return np.pad(x, widths, mode="constant")
    raise ValueError(mode)


def stft_frames(x: np.ndarray, n_fft: int, hop: int, center: bool, pad_mode: str):
    """Padded+framed signal in the dtype of x. T = 1 + (L_padded - n_fft)//hop."""
    xp = _pad_signal(x, n_fft // 2, pad_mode) if center else x
    n_frames = 1 + (xp.shape[-1] - n_fft) // hop
    return _frames(xp, n_fft, hop, n_frames)


def windowed_dft_matrix(n_fft: int, window_full: np.ndarray):
    """torchlibrosa STFT conv weights: Re/Im(exp(-2*pi*i*n*k/n_fft)) * window[n], (n_fft, F) fp64.

    torchlibrosa builds W = exp(-2 pi i x y / n) via np.power(omega, x*y); evaluated here through the
    exact angle reduction (n*k mod n_fft) which is what that expression equals mathematically.
    """
    n = np.arange(n_fft)[:, None]
    k = np.arange(n_fft // 2 + 1)[None, :]
    ang = -2.0 * np.pi * ((n * k) % n_fft) / n_fft
    return np.cos(ang) * window_full[:, None], np.sin(ang) * window_full[:, None]


def torchlibrosa_conv_weights(n_fft: int, window_full: np.ndarray):
    """The fp32 tensors torchlibrosa 0.0.9 / 0.1.0 stores in stft.conv_real.weight / stft.conv_imag.weight, (F, 1, n_fft),
    formed with the numpy expressions of its published source (torchlibrosa/stft.py, DFTBase.dft_matrix and
    STFT.__init__; constructed at external/hts_audio_transformer/model/htsat.py:453-455):

        (x, y) = np.meshgrid(np.arange(n), np.arange(n)); omega = np.exp(-2 * np.pi * 1j / n); W = np.power(omega, x * y)
        conv_real.weight = torch.Tensor(np.real(W[:, 0:F] * fft_window[:, None]).T)[:, None, :]

    np.power of a complex base goes through exp(y log x) for large exponents, so a few entries differ from the exactly
    reduced angle of windowed_dft_matrix() by an fp32 ulp; this function is the recipe itself, run by the numpy installed
    here (the library is not).
    """
    (x, y) = np.meshgrid(np.arange(n_fft), np.arange(n_fft))
    omega = np.exp(-2 * np.pi * 1j / n_fft)
    W = np.power(omega, x * y)
    F = n_fft // 2 + 1
    Ww = W[:, 0:F] * np.asarray(window_full, dtype=np.float64)[:, None]
    return (np.real(Ww).T.astype(np.float32)[:, None, :], np.imag(Ww).T.astype(np.float32)[:, None, :])


def power_spectrogram_gold(x, n_fft, hop, win_length=None, center=True, pad_mode="reflect",
                           window_full=None):
    """fp64 |STFT|^2, (..., T, F), with the exact fp64 window."""
    if window_full is None:
        win_length = n_fft if win_length is None else win_length
        window_full = pad_center(hann_periodic(win_length), n_fft)
    fr = stft_frames(np.asarray(x, dtype=np.float64), n_fft, hop, center, pad_mode)
    spec = np.fft.rfft(fr * window_full.astype(np.float64), n=n_fft, axis=-1)
    return spec.real ** 2 + spec.imag ** 2


def power_spectrogram_ref32(x, n_fft, hop, win_length=None, center=True, pad_mode="reflect",
                            window_full=None):
    """fp32 conv-DFT formulation (two real matmuls with fp32 weights), (..., T, F) fp32."""
    if window_full is None:
        win_length = n_fft if win_length is None else win_length
        window_full = pad_center(hann_periodic(win_length), n_fft)
    wr, wi = windowed_dft_matrix(n_fft, window_full)
    wr = wr.astype(np.float32)
    wi = wi.astype(np.float32)
    fr = stft_frames(np.asarray(x, dtype=np.float32), n_fft, hop, center, pad_mode)
    re = fr @ wr
    im = fr @ wi
    return re * re + im * im


# ----------------------------------------------------------------------------
# F1 + F2 : torchlibrosa Spectrogram + LogmelFilterBank (HTS-AT)
# ----------------------------------------------------------------------------


def power_to_db(x, ref=1.0, amin=1e-10, top_db=None):
    """torchlibrosa LogmelFilterBank.power_to_db: top_db clamps against the GLOBAL max of the tensor."""
    dt = x.dtype
    log_spec = 10.0 * np.log10(np.maximum(x, dt.type(amin)))
    log_spec = log_spec - dt.type(10.0 * np.log10(np.maximum(amin, ref)))
    if top_db is not None:
        log_spec = np.maximum(log_spec, log_spec.max() - dt.type(top_db))
    return log_spec.astype(dt)


def htsat_logmel(x, sr=32000, n_fft=1024, hop=320, n_mels=64, fmin=50, fmax=14000,
                 ref=1.0, amin=1e-10, top_db=None, mode="gold"):
    """(B, L) -> (B, 1, T, n_mels). mode 'gold' = fp64, 'ref32' = fp32 reference formulation."""
    melW = librosa_mel(sr, n_fft, n_mels, fmin, fmax).T  # (F, M) fp32
    if mode == "gold":
        p = power_spectrogram_gold(x, n_fft, hop, center=True, pad_mode="reflect")
        mel = p @ melW.astype(np.float64)
    else:
        p = power_spectrogram_ref32(x, n_fft, hop, center=True, pad_mode="reflect")
        mel = p @ melW
    out = power_to_db(mel, ref, amin, top_db)
    return out[:, None, :, :]


# ----------------------------------------------------------------------------
# F3 : YAMNet features
# ----------------------------------------------------------------------------

YAMNET = dict(sample_rate=16000, win=400, hop=160, n_fft=512, n_mels=64, fmin=125.0, fmax=7500.0,
              log_offset=0.001, patch_frames=96, patch_hop=48,
              min_samples=15600, hop_samples=7680)


def yamnet_padded_length(L: int) -> int:
    p = YAMNET
    n = max(L, p["min_samples"])
    after = n - p["min_samples"]
    hops = int(math.ceil(np.float32(after) / np.float32(p["hop_samples"])))
    return p["min_samples"] + p["hop_samples"] * hops


def yamnet_logmel(x, mode="gold"):
    """(B, L) or (L,) fp32 in [-1,1] -> (log_mel (B,T,64), patches (B,N,96,64))."""
    p = YAMNET
    x = np.atleast_2d(np.asarray(x))
    L = x.shape[-1]
    Lp = yamnet_padded_length(L)
    dt = np.float64 if mode == "gold" else np.float32
    xp = np.zeros(x.shape[:-1] + (Lp,), dtype=dt)
    xp[..., :L] = x
    T = 1 + (Lp - p["win"]) // p["hop"]
    idx = np.arange(p["win"])[None, :] + p["hop"] * np.arange(T)[:, None]
    w = hann_periodic(p["win"]).astype(np.float32)
    melW = tf_linear_to_mel_weight_matrix(p["n_mels"], p["n_fft"] // 2 + 1, p["sample_rate"], p["fmin"], p["fmax"])
    if mode == "gold":
        fr = xp[..., idx] * w.astype(np.float64)
        mag = np.abs(np.fft.rfft(fr, n=p["n_fft"], axis=-1))
        mel = mag @ melW.astype(np.float64)
        log_mel = np.log(mel + p["log_offset"])
    else:
        wfull = np.zeros(p["n_fft"]); wfull[:p["win"]] = w
        wr, wi = windowed_dft_matrix(p["n_fft"], wfull)
        fr = np.zeros(xp.shape[:-1] + (T, p["n_fft"]), dtype=np.float32)
        fr[..., :p["win"]] = xp[..., idx]
        re = fr @ wr.astype(np.float32); im = fr @ wi.astype(np.float32)
        mag = np.sqrt(re * re + im * im)
        mel = mag @ melW
        log_mel = np.log(mel + np.float32(p["log_offset"]))
    N = 1 + (T - p["patch_frames"]) // p["patch_hop"]
    pidx = np.arange(p["patch_frames"])[None, :] + p["patch_hop"] * np.arange(N)[:, None]
    patches = log_mel[..., pidx, :]
    return log_mel, patches


# ----------------------------------------------------------------------------
# F4 : CRNN log mel-band energy ; F5 : librosa dB mel
# ----------------------------------------------------------------------------


def crnn_extract_mbe(y, sr=44100, n_fft=2048, n_mels=40, hop=1024, mode="gold", pad_mode="constant"):
    """(L,) or (B, L) -> (n_mels, T) / (B, n_mels, T): ln(mel(|STFT|) + 1e-9). librosa>=0.10 pads with zeros."""
    melW = librosa_mel(sr, n_fft, n_mels, 0.0, sr / 2.0)  # (M, F)
    if mode == "gold":
        p = power_spectrogram_gold(y, n_fft, hop, center=True, pad_mode=pad_mode)
        mel = np.sqrt(p) @ melW.T.astype(np.float64)
        out = np.log(mel + 1e-9)
    else:
        p = power_spectrogram_ref32(y, n_fft, hop, center=True, pad_mode=pad_mode)
        mel = np.sqrt(p) @ melW.T
        out = np.log(mel + np.float32(1e-9))
    return np.swapaxes(out, -1, -2)


def librosa_db_mel(y, sr=44100, n_fft=2048, hop=512, n_mels=64, mode="gold", pad_mode="constant",
                   amin=1e-10, top_db=80.0):
    """get_mel_spec: power mel -> power_to_db(ref=np.max) with the per-clip max and an 80 dB floor. (n_mels, T)."""
    melW = librosa_mel(sr, n_fft, n_mels, 0.0, float(sr // 2))
    if mode == "gold":
        p = power_spectrogram_gold(y, n_fft, hop, center=True, pad_mode=pad_mode)
        mel = p @ melW.T.astype(np.float64)
    else:
        p = power_spectrogram_ref32(y, n_fft, hop, center=True, pad_mode=pad_mode)
        mel = p @ melW.T
    dt = mel.dtype
    ref = mel.max(axis=(-1, -2), keepdims=True)
    log_spec = 10.0 * np.log10(np.maximum(dt.type(amin), mel))
    log_spec = log_spec - 10.0 * np.log10(np.maximum(dt.type(amin), ref))
    log_spec = np.maximum(log_spec, log_spec.max(axis=(-1, -2), keepdims=True) - dt.type(top_db))
    return np.swapaxes(log_spec.astype(dt), -1, -2)


# ----------------------------------------------------------------------------
# synthetic audio (seeded; the value distribution the parity tests and the bench use)
# ----------------------------------------------------------------------------


# ----------------------------------------------------------------------------
# F6 : Kaldi filterbank features of the Audio-Mamba classifier
#      (external/audio_mamba/src/dataloader.py:94-152: torchaudio.compliance.kaldi.fbank; torchaudio is not
#      installed here, so this restates its published algorithm: PARITY UNPINNED)
# ----------------------------------------------------------------------------

def kaldi_mel_banks(num_bins, window_length_padded, sample_freq, low_freq=20.0, high_freq=0.0, dtype=np.float64):
    """torchaudio.compliance.kaldi.get_mel_banks with vtln_warp 1.0, plus the zero Nyquist column kaldi.fbank pads:
    (num_bins, window_length_padded // 2 + 1)."""
    dt = dtype
    num_fft_bins = window_length_padded // 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    fft_bin_width = dt(sample_freq) / dt(window_length_padded)

    def mel_scale(f):
        return dt(1127.0) * np.log(dt(1.0) + np.asarray(f, dtype=dt) / dt(700.0))

    mel_low, mel_high = mel_scale(low_freq), mel_scale(high_freq)
    delta = (mel_high - mel_low) / dt(num_bins + 1)
    b = np.arange(num_bins, dtype=dt)[:, None]
    left, center, right = mel_low + b * delta, mel_low + (b + dt(1.0)) * delta, mel_low + (b + dt(2.0)) * delta
    mel = mel_scale(fft_bin_width * np.arange(num_fft_bins, dtype=dt))[None, :]
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    bins = np.maximum(dt(0.0), np.minimum(up, down)).astype(dt)
    return np.pad(bins, ((0, 0), (0, 1)))


def kaldi_fbank(x, sample_frequency=16000.0, num_mel_bins=128, frame_length=25.0, frame_shift=10.0, low_freq=20.0,
                high_freq=0.0, preemphasis_coefficient=0.97, remove_dc_offset=True, mode="gold"):
    """(B, L) -> (B, T, num_mel_bins), T = 1 + (L - window) // shift.  dither 0, use_energy False, hanning window,
    power spectrum, log with the fp32 epsilon floor.  mode 'gold' = fp64 throughout, 'ref32' = fp32 arithmetic in
    torchaudio's order of operations, fp32 FFT included."""
    x = np.atleast_2d(np.asarray(x))
    dt = np.float64 if mode == "gold" else np.float32
    win = int(sample_frequency * frame_length * 0.001)
    hop = int(sample_frequency * frame_shift * 0.001)
    n_fft = 1 << max(1, (win - 1).bit_length())
    B, L = x.shape
    if L < win:
        return np.zeros((B, 0, num_mel_bins), dtype=dt)
    T = 1 + (L - win) // hop
    idx = hop * np.arange(T)[:, None] + np.arange(win)[None, :]
    fr = x.astype(dt)[:, idx]                                              # (B, T, win)
    if remove_dc_offset:
        fr = fr - fr.mean(axis=-1, keepdims=True, dtype=dt)
    if preemphasis_coefficient != 0.0:
        prev = np.concatenate([fr[..., :1], fr[..., :-1]], axis=-1)         # replicate padding on the left
        fr = fr - dt(preemphasis_coefficient) * prev
    if mode == "gold":
        n = np.arange(win, dtype=np.float64)
        window = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / (win - 1))
    else:  # the fp32 window torchaudio itself builds
        import torch
        window = torch.hann_window(win, periodic=False, dtype=torch.float32).numpy()
    fr = fr * window
    fr = np.pad(fr, ((0, 0), (0, 0), (0, n_fft - win)))
    spec = np.fft.rfft(fr, axis=-1)          # numpy >= 2 transforms fp32 input in fp32, as torch.fft.rfft does
    assert spec.dtype == (np.complex128 if mode == "gold" else np.complex64)
    power = (np.abs(spec).astype(dt)) ** 2
    banks = kaldi_mel_banks(num_mel_bins, n_fft, sample_frequency, low_freq, high_freq, dtype=dt)
    mel = power @ banks.T.astype(dt)
    eps = dt(np.finfo(np.float32).eps)
    return np.log(np.maximum(mel, eps))


def aum_wav2fbank(x, sr, melbins=128, target_length=1024, fshift=10, norm_mean=None, norm_std=None, mode="gold"):
    """AudiosetDataset._wav2fbank without mixup + the normalisation of __getitem__ (dataloader.py:94-152, 219-221)."""
    x = np.atleast_2d(np.asarray(x))
    dt = np.float64 if mode == "gold" else np.float32
    x = x.astype(dt) - x.astype(dt).mean(axis=1, keepdims=True)
    fb = kaldi_fbank(x, sample_frequency=sr, num_mel_bins=melbins, frame_shift=fshift, mode=mode)
    n = fb.shape[1]
    if n < target_length:
        fb = np.pad(fb, ((0, 0), (0, target_length - n), (0, 0)))
    else:
        fb = fb[:, :target_length]
    if norm_mean is not None and norm_std is not None:
        fb = (fb - dt(norm_mean)) / (dt(norm_std) * 2)
    return fb


# ----------------------------------------------------------------------------
# HTS-AT glue: reshape_wav2img (external/hts_audio_transformer/model/htsat.py:729-744)
# pinned against torch.nn.functional.interpolate (the function the reference calls) in tests/test_oracle_frontend.py
# ----------------------------------------------------------------------------

def _htsat_time_interp(x, target_T):
    """Bicubic interpolation of (B, 1, T, F) fp32 along T to target_T frames, fp32 arithmetic in the order of torch's
    upsample_bicubic2d (A = -0.75, align_corners=True, clamped border taps); T == target_T is returned unchanged."""
    x = np.asarray(x, dtype=np.float32)
    T = x.shape[2]
    if T == target_T:
        return x
    f32 = np.float32
    A = f32(-0.75)
    scale = f32(T - 1) / f32(target_T - 1)
    real = scale * np.arange(target_T, dtype=np.float32)
    fl = np.floor(real)
    k = fl.astype(np.int64)
    t = (real - fl).astype(np.float32)

    def c1(v):
        return ((A + f32(2)) * v - (A + f32(3))) * v * v + f32(1)

    def c2(v):
        return ((A * v - f32(5) * A) * v + f32(8) * A) * v - f32(4) * A

    w = [c2(t + f32(1)), c1(t), c1(f32(1) - t), c2(f32(2) - t)]
    taps = [x[:, :, np.clip(k + d, 0, T - 1), :] for d in (-1, 0, 1, 2)]
    y = taps[0] * w[0][None, None, :, None]
    for i in (1, 2, 3):
        y = y + taps[i] * w[i][None, None, :, None]
    return y.astype(np.float32)


def htsat_reshape_wav2img(x, spec_size=256, freq_ratio=4):
    """external/hts_audio_transformer/model/htsat.py:729-744.  x (B, 1, T, F) fp32, F == spec_size // freq_ratio ->
    (B, 1, spec_size, spec_size)."""
    x = np.asarray(x, dtype=np.float32)
    B, C, T, F = x.shape
    target_T = spec_size * freq_ratio
    assert C == 1 and T <= target_T and F == spec_size // freq_ratio
    x = _htsat_time_interp(x, target_T)
    x = x.transpose(0, 1, 3, 2)
    x = x.reshape(B, C, F, freq_ratio, target_T // freq_ratio).transpose(0, 1, 3, 2, 4)
    return np.ascontiguousarray(x.reshape(B, C, F * freq_ratio, target_T // freq_ratio))


def htsat_repeat_wat2img(x, cur_pos, spec_size=256, freq_ratio=4):
    """external/hts_audio_transformer/model/htsat.py:747-760.  x (B, 1, T, F) -> (B, 1, 4 F, spec_size)."""
    x = np.asarray(x, dtype=np.float32)
    B, C, T, F = x.shape
    target_T = spec_size * freq_ratio
    assert C == 1 and T <= target_T and F == spec_size // freq_ratio
    x = _htsat_time_interp(x, target_T).transpose(0, 1, 3, 2)
    x = x[:, :, :, cur_pos:cur_pos + spec_size]
    return np.ascontiguousarray(np.tile(x, (1, 1, 4, 1)))


def htsat_interpolate(x, ratio):
    """external/hts_audio_transformer/utils.py:158-172: (B, T, C) -> (B, T * ratio, C)."""
    x = np.asarray(x)
    B, T, C = x.shape
    return np.ascontiguousarray(np.broadcast_to(x[:, :, None, :], (B, T, ratio, C)).reshape(B, T * ratio, C))


def htsat_pad_framewise_output(x, frames_num):
    """external/hts_audio_transformer/utils.py:175-192: pad along time with the last frame."""
    x = np.asarray(x)
    pad = np.repeat(x[:, -1:, :], frames_num - x.shape[1], axis=1)
    return np.concatenate([x, pad], axis=1)


def synth_clip(clip_idx: int, n_samples: int, sr: int) -> np.ndarray:
    """Pink-ish noise + two sine bursts + 0.5 s of exact digital silence. fp32 in [-1, 1]."""
    rng = np.random.default_rng(1234 + clip_idx)
    white = rng.standard_normal(n_samples)
    pink = np.cumsum(white)
    pink -= np.linspace(pink[0], pink[-1], n_samples)
    pink = pink / (np.abs(pink).max() + 1e-12) * 0.5
    x = pink + 0.1 * 0.1 * white
    t = np.arange(n_samples) / sr
    for _ in range(2):
        f = rng.uniform(200.0, 6000.0)
        hi = min(3.0, n_samples / sr * 0.5)
        dur = rng.uniform(min(0.3, 0.5 * hi), hi)
        start = rng.uniform(0.0, max(1e-3, n_samples / sr - dur))
        amp = rng.uniform(0.05, 0.5)
        m = (t >= start) & (t < start + dur)
        x[m] += amp * np.sin(2 * np.pi * f * t[m])
    x = np.clip(x, -1.0, 1.0)
    sil = int(0.5 * sr)
    if n_samples > 2 * sil:
        s0 = int(rng.integers(0, n_samples - sil))
        x[s0:s0 + sil] = 0.0
    return x.astype(np.float32)


def synth_batch(n_clips: int, n_samples: int, sr: int, first: int = 0) -> np.ndarray:
    return np.stack([synth_clip(first + i, n_samples, sr) for i in range(n_clips)])


def pcm16_to_f32(x: np.ndarray) -> np.ndarray:
    """int16 PCM -> fp32 by the HTS-AT convention (divide by 32767 in fp64, round once to fp32),
    external/hts_audio_transformer/utils.py:96-97."""
    return np.asarray(np.asarray(x, dtype=np.float64) / 32767.0, dtype=np.float32)


def f32_to_pcm16(x: np.ndarray) -> np.ndarray:
    """fp32 in [-1,1] -> int16 (clip, scale by 32767, truncate), external/hts_audio_transformer/utils.py:99-101."""
    return np.asarray(np.clip(x, -1.0, 1.0) * 32767.0, dtype=np.int16)
