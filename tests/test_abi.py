"""CPU-only checks of the C-ABI library: it loads, exports every symbol the header declares, and its host-only
table builders agree with the oracle.  No compute entry point is called here."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sed_hip.h")
CSRC = os.path.join(ROOT, "sound-event-detection_amd", "csrc")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sed_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from sed_amd import _lib
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(_lib.lib, n), n
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names
    assert _lib.lib.sed_version().decode().startswith("sedhip")
    assert _lib.status_string(-2).startswith("unsupported n_fft")


def test_cfg_struct_matches_header():
    from sed_amd import _lib
    text = open(HEADER).read()
    body = re.search(r"typedef struct sed_frontend_cfg \{(.*?)\} sed_frontend_cfg;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ty, names = decl.split(None, 1)
        fields += [(n.strip(), ty) for n in names.split(",")]
    want = [(n, {"int32_t": ctypes.c_int32, "float": ctypes.c_float, "double": ctypes.c_double}[t]) for n, t in fields]
    assert [(n, t) for n, t in _lib.FrontendCfg._fields_] == want


def test_host_tables_match_oracle():
    import sed_amd
    from oracle import frontend as of
    for cfg, ref in ((sed_amd.FrontendConfig.htsat(), of.librosa_mel(32000, 1024, 64, 50, 14000).T),
                     (sed_amd.FrontendConfig.crnn(), of.librosa_mel(44100, 2048, 40, 0, 22050).T),
                     (sed_amd.FrontendConfig.db_mel(), of.librosa_mel(44100, 2048, 64, 0, 22050).T),
                     (sed_amd.FrontendConfig.yamnet(), of.tf_linear_to_mel_weight_matrix())):
        got = sed_amd.mel_filterbank(cfg).numpy()
        assert got.shape == ref.shape
        assert np.array_equal(got, ref)
    # Band edges cross the C ABI as fp64: the feet of the triangles depend on every bit of them (an edge rounded to fp32
    # moves a weight next to a foot by 5e-4 of its value), and callers pass whatever fp64 the model was configured with.
    import dataclasses
    rng = np.random.default_rng(0)
    for case in range(60):
        sr = int(rng.choice([8000, 16000, 22050, 32000, 44100, 48000]))
        n_fft = int(rng.choice([512, 1024, 2048]))
        M = int(rng.integers(1, 129))
        fmin = float(rng.uniform(0, 400))
        fmax = float(rng.uniform(max(fmin + 500, 0.2 * sr), 0.5 * sr))
        cfg = sed_amd.FrontendConfig.htsat(sample_rate=sr, window_size=n_fft, hop_size=n_fft // 4, mel_bins=M, fmin=fmin, fmax=fmax)
        assert np.array_equal(sed_amd.mel_filterbank(cfg).numpy(), of.librosa_mel(sr, n_fft, M, fmin, fmax).T), (sr, n_fft, M, fmin, fmax)
        if case % 3 == 0:
            M2, lo, hi = int(rng.integers(8, 129)), float(rng.uniform(20, 300)), float(rng.uniform(3000, 8000))
            cfg = dataclasses.replace(sed_amd.FrontendConfig.yamnet(), n_mels=M2, fmin=lo, fmax=hi)
            assert np.array_equal(sed_amd.mel_filterbank(cfg).numpy(), of.tf_linear_to_mel_weight_matrix(M2, 257, 16000, lo, hi))
            kc = sed_amd.FrontendConfig.kaldi_fbank(16000.0, M2, low_freq=lo)
            assert np.array_equal(sed_amd.mel_filterbank(kc).numpy(), of.kaldi_mel_banks(M2, 512, 16000.0, lo, 0.0).T.astype(np.float32))
    kcfg = sed_amd.FrontendConfig.kaldi_fbank(16000.0, 128)
    assert (kcfg.n_fft, kcfg.win_length, kcfg.hop) == (512, 400, 160)
    got = sed_amd.mel_filterbank(kcfg).numpy()
    ref = of.kaldi_mel_banks(128, 512, 16000.0).T
    assert got.shape == ref.shape == (257, 128) and np.max(np.abs(got - ref)) <= 1e-7 and np.all(got[256] == 0)
    ws = sed_amd.hann_window(512, 400, win_align=1, window_kind=1).numpy()
    n = np.arange(400)
    assert np.max(np.abs(ws[:400] - (0.5 - 0.5 * np.cos(2 * np.pi * n / 399)))) < 1e-7 and np.all(ws[400:] == 0)
    w = sed_amd.hann_window(1024, 1024).numpy()
    assert np.max(np.abs(w - of.hann_periodic(1024))) < 1e-7
    wl = sed_amd.hann_window(512, 400, 1).numpy()
    assert np.all(wl[400:] == 0) and np.max(np.abs(wl[:400] - of.hann_periodic(400))) < 1e-7
    wc = sed_amd.hann_window(2048, 1024).numpy()
    assert np.max(np.abs(wc - of.pad_center(of.hann_periodic(1024), 2048))) < 1e-7
    # bit for bit against the call torchlibrosa / librosa make (scipy get_window, fp64, cast to fp32 with the weights)
    from scipy.signal import get_window
    for n in (400, 512, 800, 1024, 1102, 1764, 2048):
        n_fft = 1 << (n - 1).bit_length()
        sw = get_window("hann", n, fftbins=True)
        assert np.array_equal(sed_amd.hann_window(n_fft, n, 1).numpy()[:n], sw.astype(np.float32)), n
        assert np.array_equal(of.hann_periodic(n), sw)
        ss = get_window("hann", n, fftbins=False)
        assert np.array_equal(sed_amd.hann_window(n_fft, n, 1, 1).numpy()[:n], ss.astype(np.float32)), n
    # the window argument of the torchlibrosa surface: what scipy returns for the specification, centred in n_fft samples
    from sed_amd.frontend import _STFTTables, fft_window
    assert np.array_equal(fft_window("hann", 800, 1024), of.pad_center(get_window("hann", 800, fftbins=True), 1024))
    for spec in ("hamming", ("kaiser", 8.0), ("tukey", 0.25), "boxcar"):
        want = of.pad_center(get_window(spec, 400, fftbins=True).astype(np.float64), 512)
        assert np.array_equal(fft_window(spec, 400, 512), want)
        real, imag = _STFTTables.weights(512, 400, spec)
        wr, wi = of.torchlibrosa_conv_weights(512, want)
        assert np.array_equal(real, wr) and np.array_equal(imag, wi)
    ham = get_window("hamming", 400, fftbins=True)
    assert np.array_equal(fft_window(ham, 400, 512), fft_window("hamming", 400, 512))
    assert np.array_equal(fft_window(lambda n: get_window("hamming", n), 400, 512), fft_window("hamming", 400, 512))
    with pytest.raises(ValueError):
        fft_window(ham[:-1], 400, 512)
    with pytest.raises(ValueError):
        fft_window(np.full(400, np.nan), 400, 512)


def test_modules_construct_and_refuse_cpu_input():
    import torch
    import sed_amd
    s = sed_amd.Spectrogram(n_fft=1024, hop_length=320, win_length=1024, window="hann", center=True,
                            pad_mode="reflect", freeze_parameters=True)
    m = sed_amd.LogmelFilterBank(sr=32000, n_fft=1024, n_mels=64, fmin=50, fmax=14000, ref=1.0, amin=1e-10,
                                 top_db=None, freeze_parameters=True)
    assert sorted(s.state_dict()) == ["stft.conv_imag.weight", "stft.conv_real.weight"]
    assert s.state_dict()["stft.conv_real.weight"].shape == (513, 1, 1024)
    assert list(m.state_dict()) == ["melW"] and m.melW.shape == (513, 64)
    from oracle import frontend as of
    wr, wi = of.windowed_dft_matrix(1024, of.hann_periodic(1024))
    assert np.max(np.abs(s.stft.conv_real.weight[:, 0, :].numpy() - wr.T)) < 1e-6
    assert np.max(np.abs(s.stft.conv_imag.weight[:, 0, :].numpy() - wi.T)) < 1e-6
    # the tensors of the state_dict are the ones torchlibrosa's own numpy recipe gives (omega ** (x * y) in complex128),
    # bit for bit, at every size the library builds; they sit within one fp32 ulp (6e-8) of the exactly reduced angle, far
    # inside the 1e-5 the load check allows
    for n_fft in (512, 1024, 2048):
        sp = s if n_fft == 1024 else sed_amd.Spectrogram(n_fft=n_fft, hop_length=n_fft // 4)
        tr, ti = of.torchlibrosa_conv_weights(n_fft, of.pad_center(of.hann_periodic(n_fft), n_fft))
        assert np.array_equal(sp.stft.conv_real.weight.numpy(), tr) and np.array_equal(sp.stft.conv_imag.weight.numpy(), ti)
        er, ei = of.windowed_dft_matrix(n_fft, of.pad_center(of.hann_periodic(n_fft), n_fft))
        assert np.max(np.abs(tr[:, 0, :] - er.T)) <= 6e-8 and np.max(np.abs(ti[:, 0, :] - ei.T)) <= 6e-8
    sp = sed_amd.Spectrogram(n_fft=512, hop_length=160, win_length=400)  # a window shorter than the transform, centred
    tr, ti = of.torchlibrosa_conv_weights(512, of.pad_center(of.hann_periodic(400), 512))
    assert np.array_equal(sp.stft.conv_real.weight.numpy(), tr) and np.array_equal(sp.stft.conv_imag.weight.numpy(), ti)
    with pytest.raises(RuntimeError, match="no CPU path"):
        s(torch.zeros(1, 4096))
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 1, 3, 513))
    with pytest.raises(AssertionError):
        sed_amd.Spectrogram(pad_mode="edge")
    bad = {k: v.clone() for k, v in s.state_dict().items()}
    bad["stft.conv_real.weight"] += 0.01
    with pytest.raises(RuntimeError, match="windowed DFT"):
        s.load_state_dict(bad)


def test_lane_decomposition_on_cpu():
    """The per-lane FFT code shared with the kernels, executed lane by lane on the CPU against a naive DFT."""
    exe = os.path.join(CSRC, "host_sim")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", CSRC, "host_sim"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert out.stdout.count("max_rel_err") == 18
