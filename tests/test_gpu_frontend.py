"""GPU parity: the HIP front-end (through the C ABI) against the CPU oracle on the same seeded audio.

Tolerance (fp32 data path, fp64 gold): |hip - gold| <= 1e-3 in the log-mel domain on every mel bin within
60 dB of its frame's peak; on deeper bins the HIP error must not exceed max(1e-3, 1.5 x the error of the
reference's own fp32 conv-DFT formulation against the same gold).  Exact-silence frames equal the floor.
"""
import os

import numpy as np
import pytest
import torch

from oracle import frontend as of

pytestmark = pytest.mark.gpu

NEAR_DB = 60.0


def _dev():
    import sed_amd  # noqa: F401  (fails loudly if the extension is missing)
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def _check(hip, gold, ref32, near_span, tol=1e-3, deep_factor=1.5):
    hip, gold, ref32 = np.asarray(hip, np.float64), np.asarray(gold, np.float64), np.asarray(ref32, np.float64)
    assert hip.shape == gold.shape
    peak = gold.max(axis=-1, keepdims=True)
    near = gold >= peak - near_span
    err = np.abs(hip - gold)
    rerr = np.abs(ref32 - gold)
    near_max = float(err[near].max())
    deep_max = float(err[~near].max()) if (~near).any() else 0.0
    deep_ref = float(rerr[~near].max()) if (~near).any() else 0.0
    print(f"near-bin max err {near_max:.3e} (ref32 {float(rerr[near].max()):.3e}); deep-bin {deep_max:.3e} (ref32 {deep_ref:.3e})")
    assert near_max <= tol
    assert deep_max <= max(tol, deep_factor * deep_ref)


def test_htsat_fused_vs_oracle():
    import sed_amd
    dev = _dev()
    x = of.synth_batch(5, 64000, 32000)
    x[4, :] = 0.0                                                     # one clip of digital silence
    fe = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat())
    y = fe(torch.from_numpy(x).to(dev))
    assert y.shape == (5, 1, 201, 64) and y.dtype == torch.float32 and y.is_contiguous()
    y = y.cpu().numpy()
    gold = of.htsat_logmel(x, mode="gold")
    ref32 = of.htsat_logmel(x, mode="ref32")
    _check(y[:4], gold[:4], ref32[:4], NEAR_DB)
    assert np.all(y[4] == y[4].flat[0]) and abs(float(y[4].flat[0]) + 100.0) < 2e-5   # floor, exactly uniform


def test_htsat_full_clip_10s():
    import sed_amd
    dev = _dev()
    x = of.synth_batch(2, 320000, 32000, first=100)
    y = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat())(torch.from_numpy(x).to(dev)).cpu().numpy()
    assert y.shape == (2, 1, 1001, 64)
    _check(y, of.htsat_logmel(x, mode="gold"), of.htsat_logmel(x, mode="ref32"), NEAR_DB)


def test_unfused_modules_match_reference_call_pattern():
    """Constructed and called exactly as htsat.py:453-459 / :763-764."""
    import sed_amd
    dev = _dev()
    spectrogram_extractor = sed_amd.Spectrogram(n_fft=1024, hop_length=320, win_length=1024, window="hann",
                                                center=True, pad_mode="reflect", freeze_parameters=True).to(dev)
    logmel_extractor = sed_amd.LogmelFilterBank(sr=32000, n_fft=1024, n_mels=64, fmin=50, fmax=14000, ref=1.0,
                                                amin=1e-10, top_db=None, freeze_parameters=True).to(dev)
    x = of.synth_batch(3, 48000, 32000, first=7)
    xs = spectrogram_extractor(torch.from_numpy(x).to(dev))
    assert xs.shape == (3, 1, 151, 513)
    gold_p = of.power_spectrogram_gold(x, 1024, 320)
    p = xs[:, 0].cpu().numpy().astype(np.float64)
    assert np.max(np.abs(p - gold_p) / np.maximum(gold_p.max(axis=-1, keepdims=True), 1e-30)) < 2e-6
    xm = logmel_extractor(xs)
    assert xm.shape == (3, 1, 151, 64)
    gold = of.htsat_logmel(x, mode="gold")
    _check(xm.cpu().numpy(), gold, of.htsat_logmel(x, mode="ref32"), NEAR_DB)
    fused = sed_amd.LogMelFrontend.from_torchlibrosa(spectrogram_extractor, logmel_extractor)(torch.from_numpy(x).to(dev))
    assert torch.max(torch.abs(fused - xm)).item() < 1e-4


def test_logmel_top_db_and_power_variants():
    import sed_amd
    dev = _dev()
    x = of.synth_batch(2, 32000, 32000, first=3)
    x[0, 9000:20000] = 0.0                                               # silence: more than 80 dB below the peak
    xt = torch.from_numpy(x).to(dev)
    spec = sed_amd.Spectrogram(n_fft=1024, hop_length=320)(xt)
    lm = sed_amd.LogmelFilterBank(sr=32000, n_fft=1024, n_mels=64, fmin=50, fmax=14000, top_db=80.0)(spec)
    mel = of.power_spectrogram_ref32(x, 1024, 320) @ of.librosa_mel(32000, 1024, 64, 50, 14000).T
    want = of.power_to_db(mel, 1.0, 1e-10, 80.0)[:, None]
    assert np.max(np.abs(lm.cpu().numpy() - want)) < 2e-3
    assert abs(lm.min().item() - (lm.max().item() - 80.0)) < 1e-4
    mag = sed_amd.Spectrogram(n_fft=1024, hop_length=320, power=1.0)(xt)
    assert torch.max(torch.abs(mag * mag - spec) / spec.amax(-1, keepdim=True).clamp_min(1e-30)).item() < 1e-5
    nolog = sed_amd.LogmelFilterBank(sr=32000, n_fft=1024, n_mels=64, fmin=50, fmax=14000, is_log=False)(spec)
    assert np.max(np.abs(nolog.cpu().numpy()[:, 0] - mel) / mel.max()) < 1e-5


def test_other_windows_of_the_torchlibrosa_surface():
    """torchlibrosa's Spectrogram takes any librosa window specification (stft.py: get_window(window, win_length,
    fftbins=True) + pad_center): names, (name, parameter) tuples, arrays and callables, with win_length <= n_fft.
    Checked against the fp64 transform with the same window, unfused and fused.  Parity unpinned: the reference itself only
    ever asks for 'hann' (htsat.py:453-455)."""
    import sed_amd
    from scipy.signal import get_window
    dev = _dev()
    x = of.synth_batch(3, 40000, 32000, first=11)
    xt = torch.from_numpy(x).to(dev)
    cases = [("hamming", 1024, 1024, 320), ("blackman", 1024, 800, 256), (("kaiser", 8.0), 512, 400, 160),
             (("tukey", 0.25), 2048, 2048, 512), ("boxcar", 1024, 1024, 320)]
    for window, n_fft, win_length, hop in cases:
        wfull = of.pad_center(get_window(window, win_length, fftbins=True).astype(np.float64), n_fft)
        sp = sed_amd.Spectrogram(n_fft=n_fft, hop_length=hop, win_length=win_length, window=window).to(dev)
        p = sp(xt)[:, 0].cpu().numpy().astype(np.float64)
        gold_p = of.power_spectrogram_gold(x, n_fft, hop, win_length, window_full=wfull)
        assert p.shape == gold_p.shape
        # relative to the frame's largest bin; a window without a taper leaks more and leaves less dynamic range
        assert np.max(np.abs(p - gold_p) / np.maximum(gold_p.max(axis=-1, keepdims=True), 1e-30)) < 2e-6, window
        # the tables a checkpoint would carry are those of torchlibrosa's recipe for that window
        wr, wi = of.torchlibrosa_conv_weights(n_fft, wfull)
        assert np.array_equal(sp.stft.conv_real.weight.cpu().numpy(), wr)
        assert np.array_equal(sp.stft.conv_imag.weight.cpu().numpy(), wi)
        lm = sed_amd.LogmelFilterBank(sr=32000, n_fft=n_fft, n_mels=64, fmin=50, fmax=14000, top_db=None).to(dev)
        fused = sed_amd.LogMelFrontend.from_torchlibrosa(sp, lm)(xt).cpu().numpy()
        melW = of.librosa_mel(32000, n_fft, 64, 50, 14000).T
        gold = of.power_to_db(gold_p @ melW.astype(np.float64))[:, None]
        ref32 = of.power_to_db(of.power_spectrogram_ref32(x, n_fft, hop, win_length, window_full=wfull) @ melW)[:, None]
        _check(fused, gold, ref32, NEAR_DB)
        assert torch.max(torch.abs(torch.from_numpy(fused).to(dev) - lm(sp(xt)))).item() < 1e-4
    # the other framing options with a window that is not Hann: no centring, zero padding, hops that are no multiple of 64
    # samples, int16 samples
    for center, pad_mode, hop in ((False, "reflect", 320), (True, "constant", 252), (False, "constant", 332)):
        wfull = get_window("hamming", 1024, fftbins=True).astype(np.float64)
        sp = sed_amd.Spectrogram(n_fft=1024, hop_length=hop, window="hamming", center=center, pad_mode=pad_mode).to(dev)
        p = sp(xt)[:, 0].cpu().numpy().astype(np.float64)
        gold_p = of.power_spectrogram_gold(x, 1024, hop, 1024, center=center, pad_mode=pad_mode, window_full=wfull)
        assert p.shape == gold_p.shape, (center, pad_mode, hop)
        assert np.max(np.abs(p - gold_p) / np.maximum(gold_p.max(axis=-1, keepdims=True), 1e-30)) < 2e-6, (center, pad_mode, hop)
    cfg16 = sed_amd.FrontendConfig.htsat(pcm_kind=1, pcm_divisor=32767.0)
    s16 = of.f32_to_pcm16(x)
    fe16 = sed_amd.LogMelFrontend(cfg16, layout="btm", window=("kaiser", 6.0))
    fe32 = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat(), layout="btm", window=("kaiser", 6.0))
    got16 = fe16(torch.from_numpy(s16).to(dev))
    want16 = fe32(torch.from_numpy(of.pcm16_to_f32(s16)).to(dev))
    assert torch.equal(got16, want16)                                   # the int16 path converts as the reference does, then the same kernel
    # an array and a callable give what the name gives
    ham = get_window("hamming", 1024, fftbins=True)
    want = sed_amd.Spectrogram(n_fft=1024, hop_length=320, window="hamming")(xt)
    assert torch.equal(sed_amd.Spectrogram(n_fft=1024, hop_length=320, window=ham)(xt), want)
    assert torch.equal(sed_amd.Spectrogram(n_fft=1024, hop_length=320, window=lambda n: get_window("hamming", n))(xt), want)
    with pytest.raises(ValueError):
        sed_amd.Spectrogram(n_fft=1024, hop_length=320, window=ham[:-1])
    # hops that are no multiple of 4 samples (10 ms at 44.1 kHz = 441): frames start off the 16-byte grid of the staged span
    for n_fft, hop in ((1024, 441), (1024, 250), (512, 333), (2048, 441), (512, 161)):
        sp = sed_amd.Spectrogram(n_fft=n_fft, hop_length=hop).to(dev)
        p = sp(xt)[:, 0].cpu().numpy().astype(np.float64)
        gold_p = of.power_spectrogram_gold(x, n_fft, hop)
        assert p.shape == gold_p.shape, (n_fft, hop)
        assert np.max(np.abs(p - gold_p) / np.maximum(gold_p.max(axis=-1, keepdims=True), 1e-30)) < 2e-6, (n_fft, hop)
        lm = sed_amd.LogmelFilterBank(sr=32000, n_fft=n_fft, n_mels=64, fmin=50, fmax=14000, top_db=None).to(dev)
        fused = sed_amd.LogMelFrontend.from_torchlibrosa(sp, lm)(xt)
        assert torch.max(torch.abs(fused - lm(sp(xt)))).item() < 1e-4, (n_fft, hop)
        melW = of.librosa_mel(32000, n_fft, 64, 50, 14000).T
        gold = of.power_to_db(gold_p @ melW.astype(np.float64))[:, None]
        ref32 = of.power_to_db(of.power_spectrogram_ref32(x, n_fft, hop) @ melW)[:, None]
        _check(fused.cpu().numpy(), gold, ref32, NEAR_DB)
    # the same audio as int16 through such a hop
    cfgo = sed_amd.FrontendConfig.htsat(hop_size=441)
    cfgo16 = sed_amd.FrontendConfig.htsat(hop_size=441, pcm_kind=1, pcm_divisor=32767.0)
    a = sed_amd.LogMelFrontend(cfgo16, layout="btm")(torch.from_numpy(s16).to(dev))
    b = sed_amd.LogMelFrontend(cfgo, layout="btm")(torch.from_numpy(of.pcm16_to_f32(s16)).to(dev))
    assert torch.equal(a, b)
    # a state_dict saved with one window does not load into a module built with another
    a = sed_amd.Spectrogram(n_fft=1024, hop_length=320, window="hamming")
    b = sed_amd.Spectrogram(n_fft=1024, hop_length=320, window="hann")
    with pytest.raises(RuntimeError):
        b.load_state_dict(a.state_dict())
    b2 = sed_amd.Spectrogram(n_fft=1024, hop_length=320, window="hamming")
    b2.load_state_dict(a.state_dict())


def test_yamnet_features():
    import sed_amd
    _dev()
    x = of.synth_batch(3, 160000, 16000, first=20)
    lm, patches = sed_amd.yamnet_log_mel(torch.from_numpy(x))
    assert lm.shape == (3, 1008, 64) and patches.shape == (3, 20, 96, 64)
    g_lm, g_p = of.yamnet_logmel(x, mode="gold")
    r_lm, _ = of.yamnet_logmel(x, mode="ref32")
    _check(lm.cpu().numpy(), g_lm, r_lm, NEAR_DB / 20.0 * np.log(10.0))
    assert np.array_equal(patches.cpu().numpy(), lm.cpu().numpy()[:, np.arange(96)[None] + 48 * np.arange(20)[:, None]])
    lm1, p1 = sed_amd.yamnet_log_mel(x[0, :50000])                       # 1-D input, tail padded to 54000
    g1, gp1 = of.yamnet_logmel(x[0, :50000], mode="gold")
    assert lm1.shape == g1.shape[1:] == (336, 64) and p1.shape == gp1.shape[1:] == (6, 96, 64)
    assert np.max(np.abs(lm1.cpu().numpy() - g1[0])) < 1e-3
    z, _ = sed_amd.yamnet_log_mel(np.zeros(16000, np.float32))
    assert z.shape == (144, 64) and torch.all(z == z[0, 0]) and abs(z[0, 0].item() - np.log(0.001)) < 1e-6


def test_crnn_extract_mbe():
    import sed_amd
    _dev()
    y = of.synth_clip(5, 220500, 44100)
    mbe = sed_amd.extract_mbe(y, 44100, 2048, 40, 1024)
    assert isinstance(mbe, np.ndarray) and mbe.shape == (40, 216) and mbe.dtype == np.float32
    _check(mbe.T, of.crnn_extract_mbe(y, mode="gold").T, of.crnn_extract_mbe(y, mode="ref32").T,
           NEAR_DB / 20.0 * np.log(10.0))
    z = sed_amd.extract_mbe(np.zeros(44100, np.float32), 44100, 2048, 40, 1024)
    assert np.all(z == z[0, 0]) and abs(float(z[0, 0]) - np.log(1e-9)) < 1e-5


def test_db_mel_per_clip_reference():
    import sed_amd
    _dev()
    y = of.synth_batch(2, 88200, 44100, first=30)
    got = sed_amd.get_mel_spec_db(y, 44100)
    want = of.librosa_db_mel(y, mode="gold")
    assert got.shape == (2, 64, 173) and got.max() == 0.0 and got.min() >= -80.0
    assert np.max(np.abs(got - want)) < 1e-3


def test_int16_input_and_exhaustive_conversion():
    import sed_amd
    dev = _dev()
    cfg16 = sed_amd.FrontendConfig.htsat(pcm_kind=1, pcm_divisor=32767.0)
    s = of.f32_to_pcm16(of.synth_batch(2, 65536, 32000, first=40))
    s[1] = np.arange(-32768, 32768, dtype=np.int16)                      # every int16 value once
    y16 = sed_amd.LogMelFrontend(cfg16)(torch.from_numpy(s).to(dev))
    yf = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat())(torch.from_numpy(of.pcm16_to_f32(s)).to(dev))
    assert torch.equal(y16, yf)                                          # identical samples -> identical bits
    for div in (32768.0, 12345.678, 3.0, 1e-3):                          # soundfile's convention and arbitrary scales
        cfg = sed_amd.FrontendConfig.htsat(pcm_kind=1, pcm_divisor=div)
        f = (s.astype(np.float32) / np.float32(div)).astype(np.float32)
        assert torch.equal(sed_amd.LogMelFrontend(cfg)(torch.from_numpy(s).to(dev)),
                           sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat())(torch.from_numpy(f).to(dev))), div
    with pytest.raises(RuntimeError, match="dtype"):
        sed_amd.LogMelFrontend(cfg16)(torch.zeros(1, 4096, device=dev))


def test_edges_ragged_short_empty_strided():
    import sed_amd
    dev = _dev()
    fe = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat(), layout="btm")
    for n in (513, 640, 1023, 1024, 5000, 5441, 10239, 33333):          # T = 2 .. 105, none a multiple of 16
        x = of.synth_batch(2, n, 32000, first=n)
        y = fe(torch.from_numpy(x).to(dev)).cpu().numpy()
        g = of.htsat_logmel(x, mode="gold")[:, 0]
        assert y.shape == g.shape == (2, 1 + n // 320, 64)
        near = g >= g.max(-1, keepdims=True) - NEAR_DB
        assert np.max(np.abs(y - g)[near]) < 1e-3, n
    with pytest.raises(ValueError, match="too short"):
        fe(torch.zeros(1, 512, device=dev))
    assert fe(torch.zeros(0, 4096, device=dev)).shape == (0, 13, 64)
    one = of.synth_clip(7, 5000, 32000)[None]                             # a (1, n) view whose batch stride numpy leaves at 0
    assert torch.equal(fe(torch.from_numpy(one).to(dev)), fe(torch.from_numpy(one.copy()).to(dev)))
    big = torch.from_numpy(of.synth_batch(3, 20001, 32000, first=9)).to(dev)
    a = fe(big[:, 1:])                                                    # misaligned rows, row stride != length
    b = fe(big[:, 1:].contiguous())
    assert torch.equal(a, b)
    assert torch.equal(fe(big[1:2, 1:]), a[1:2])                          # batch position does not matter


def test_mel_counts_hops_and_affine_variants():
    """Mel counts that are not multiples of 4 (scalar stores, partial last quad), more than 64 filters (two gather
    rounds), hops that change the staging path (span wider than 2048 samples), and the affine epilogue on them."""
    import sed_amd
    dev = _dev()
    x = of.synth_batch(3, 40000, 32000, first=60)
    xt = torch.from_numpy(x).to(dev)
    for n_mels, hop, n_fft in ((3, 320, 1024), (30, 320, 1024), (65, 320, 1024), (128, 320, 1024), (64, 1024, 1024),
                               (64, 512, 1024), (50, 160, 512), (128, 256, 2048)):
        cfg = sed_amd.FrontendConfig.htsat(window_size=n_fft, hop_size=hop, mel_bins=n_mels)
        fe = sed_amd.LogMelFrontend(cfg, layout="btm")
        y = fe(xt)
        gold = of.htsat_logmel(x, n_fft=n_fft, hop=hop, n_mels=n_mels, mode="gold")[:, 0]
        ref32 = of.htsat_logmel(x, n_fft=n_fft, hop=hop, n_mels=n_mels, mode="ref32")[:, 0]
        assert y.shape == gold.shape, (n_mels, hop, n_fft)
        _check(y.cpu().numpy(), gold, ref32, NEAR_DB)
        scale = torch.linspace(0.5, 2.0, n_mels)
        shift = torch.linspace(-3.0, 3.0, n_mels)
        fe.set_affine(scale, shift)
        ya = fe(xt)
        want = y * scale.to(dev) + shift.to(dev)
        assert torch.max(torch.abs(ya - want)).item() < 1e-4, (n_mels, hop, n_fft)
        out = torch.empty(3 * y.shape[1] * n_mels + 1, device=dev)[1:]      # output base that is only 4-byte aligned
        fe.set_affine(None, None)
        plan = fe._plan_for(xt.device)
        from sed_amd import _lib as L
        L.check(L.lib.sed_logmel_forward(plan.ptr, xt.data_ptr(), 3, xt.shape[1], xt.stride(0), out.data_ptr(),
                                         y.shape[1] * n_mels, L.stream_ptr(xt.device)))
        assert torch.equal(out.view(3, y.shape[1], n_mels), y), (n_mels, hop, n_fft)


def test_kaldi_fbank_and_aum_wav2fbank():
    """Audio-Mamba front-end (external/audio_mamba/src/dataloader.py:94-152, 219-221).  Tolerance: 1e-3 in the natural-log
    domain on bins within 60 dB of the frame's peak; deeper bins no worse than 2.5x the fp32 restatement, whose
    transform is pocketfft's fp32 real FFT (fewer roundings than the kernel's radix-2 stages + real-FFT split)."""
    import sed_amd
    _dev()
    span = NEAR_DB / 10.0 * np.log(10.0)                                 # 60 dB of power in nepers
    # (44.1 kHz with 128 bins is the rate of the reference's own files, torchaudio.load at dataloader.py:98: 1102-sample
    # frames every 441 samples in 2048-point transforms)
    for sr, n, mel in ((16000, 160000, 128), (32000, 100000, 128), (44100, 120000, 64), (44100, 100000, 128)):
        x = of.synth_batch(3, n, sr, first=70) + 0.02                    # a DC offset for remove_dc_offset to remove
        got = sed_amd.kaldi_fbank(torch.from_numpy(x), sample_frequency=sr, num_mel_bins=mel, frame_shift=10,
                                  htk_compat=True, use_energy=False, window_type="hanning", dither=0.0)
        gold = of.kaldi_fbank(x, sr, mel, mode="gold")
        ref32 = of.kaldi_fbank(x, sr, mel, mode="ref32")
        assert got.shape == gold.shape == (3, 1 + (n - int(sr * 0.025)) // int(sr * 0.01), mel)
        _check(got.cpu().numpy(), gold, ref32, span, deep_factor=2.5)
    x = of.synth_batch(2, 160000, 16000, first=80)
    one = sed_amd.kaldi_fbank(torch.from_numpy(x[:1]), sample_frequency=16000, num_mel_bins=128, frame_shift=10)
    assert one.shape == (998, 128)                                       # (1, L) -> (T, mel) like torchaudio
    fb = sed_amd.wav2fbank(torch.from_numpy(x), 16000, melbins=128, target_length=1024, norm_mean=-0.8207, norm_std=2.15)
    want = of.aum_wav2fbank(x, 16000, 128, 1024, 10, -0.8207, 2.15, mode="gold")
    assert fb.shape == want.shape == (2, 1024, 128)
    assert np.allclose(fb[:, 998:].cpu().numpy(), 0.8207 / 4.3)          # zero-padded frames after normalisation
    g = of.aum_wav2fbank(x, 16000, 128, 1024, 10, mode="gold")[:, :998]
    near = g >= g.max(-1, keepdims=True) - span
    assert np.max(np.abs(fb.cpu().numpy()[:, :998] - want[:, :998])[near]) < 1e-3 / 4.3 + 1e-6
    cut = sed_amd.wav2fbank(torch.from_numpy(x), 16000, melbins=128, target_length=512)
    assert cut.shape == (2, 512, 128)
    z = sed_amd.kaldi_fbank(torch.zeros(1, 16000), sample_frequency=16000, num_mel_bins=128)
    assert z.shape == (98, 128) and torch.all(z == z[0, 0]) and abs(z[0, 0].item() - np.log(np.finfo(np.float32).eps)) < 1e-5
    assert sed_amd.kaldi_fbank(torch.zeros(1, 300), sample_frequency=16000, num_mel_bins=128).shape == (0, 128)
    with pytest.raises(ValueError):
        sed_amd.kaldi_fbank(torch.zeros(1, 16000), sample_frequency=16000, use_energy=True)


def test_reshape_wav2img_against_restatement():
    """HTS-AT glue (htsat.py:729-744): HIP vs the fp32 restatement (itself pinned against torch on the CPU)."""
    import sed_amd
    dev = _dev()
    rng = np.random.default_rng(4)
    for (B, T, F, spec, ratio) in ((3, 1001, 64, 256, 4), (2, 1024, 64, 256, 4), (3, 37, 4, 16, 4), (1, 5, 8, 8, 1),
                                   (2, 400, 128, 256, 2)):
        x = (rng.standard_normal((B, 1, T, F)) * 20 - 40).astype(np.float32)
        got = sed_amd.reshape_wav2img(torch.from_numpy(x).to(dev), spec, ratio)
        want = of.htsat_reshape_wav2img(x, spec, ratio)
        assert got.shape == want.shape == (B, 1, spec, spec)
        assert np.max(np.abs(got.cpu().numpy() - want)) <= 1e-6 * np.max(np.abs(x)) * 2, (T, F)
        if T == spec * ratio:
            assert np.array_equal(got.cpu().numpy(), want)
    fe = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat())
    y = fe(torch.from_numpy(of.synth_batch(2, 320000, 32000, first=5)).to(dev))     # (2, 1, 1001, 64)
    assert sed_amd.reshape_wav2img(y).shape == (2, 1, 256, 256)
    with pytest.raises(AssertionError):
        sed_amd.reshape_wav2img(torch.zeros(1, 1, 1025, 64, device=dev))
    xl = (rng.standard_normal((2, 1, 2001, 64)) * 20 - 40).astype(np.float32)        # longer than 1024 frames: htsat.py:816-823
    imgs = sed_amd.long_input_images(torch.from_numpy(xl).to(dev))
    pos = list(range(0, 2001 - 1000 - 1, 500))
    assert len(imgs) == len(pos) == 2
    for im, p0 in zip(imgs, pos):
        want = of.htsat_reshape_wav2img(xl[:, :, p0:p0 + 1000], 256, 4)
        assert np.max(np.abs(im.cpu().numpy() - want)) <= 1e-6 * np.max(np.abs(xl)) * 2


def test_reference_htsat_operating_point_44k1_20s_spec512():
    """The configuration the reference itself runs (src/models/htsat/model.py:171-186 on top of
    external/hts_audio_transformer/config.py:53-59): 44.1 kHz, 20 s clips, htsat_spec_size 512 -> L = 882000 samples,
    T = 2757 frames of 64 mel, then reshape_wav2img with freq_ratio 512 // 64 = 8 interpolates to 4096 frames and folds
    them into a 512 x 512 image (external/hts_audio_transformer/model/htsat.py:729-744)."""
    import sed_amd
    dev = _dev()
    sr, L = 44100, 882000
    x = of.synth_batch(2, L, sr, first=40)
    fe = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat(sample_rate=sr))
    y = fe(torch.from_numpy(x).to(dev))
    assert y.shape == (2, 1, 2757, 64)
    yn = y.cpu().numpy()
    _check(yn, of.htsat_logmel(x, sr=sr, mode="gold"), of.htsat_logmel(x, sr=sr, mode="ref32"), NEAR_DB)
    img = sed_amd.reshape_wav2img(y, 512, 8)
    assert img.shape == (2, 1, 512, 512)
    want = of.htsat_reshape_wav2img(yn, 512, 8)
    assert np.max(np.abs(img.cpu().numpy() - want)) <= 1e-6 * np.max(np.abs(yn)) * 2
    assert torch.equal(sed_amd.reshape_wav2img(y, 512), img)           # freq_ratio defaults to spec_size // mel bins
    with pytest.raises(AssertionError):                                # 2757 frames do not fit the 256-size model
        sed_amd.reshape_wav2img(y)
    # the detector-side decode of the same operating point: (1, 4096, 6) framewise map, 20 s clip (combined.py:17-73)
    rng = np.random.default_rng(8)
    pm = rng.random((1, 4096, 6), dtype=np.float32)
    from oracle import decode as od
    names = ["cough", "dog_bark", "gun_shot", "siren", "car_horn", "non_event"]
    lab = np.repeat(rng.integers(0, 6, 4096 // 300 + 1), 300)[:4096]
    pm[0, np.arange(4096), lab] += 0.6

    class Cfg:
        clip_duration = 20.0
    want_ev = od.htsat_events_from_batch(["a.wav"], pm, names, 5, 20.0)
    got_ev = sed_amd.get_events_from_batch({"file": ["a.wav"], "pred_map": torch.from_numpy(pm).to(dev)}, Cfg)
    assert got_ev == want_ev and len(want_ev["a.wav"]) >= 5


def test_repeat_mode_and_infer_mode_images():
    """enable_repeat_mode (htsat.py:747-760, 789-793) and infer_mode (:775-781) image builders vs the restatement."""
    import sed_amd
    dev = _dev()
    rng = np.random.default_rng(8)
    for (B, T, F, spec, ratio) in ((2, 1001, 64, 256, 4), (1, 1024, 64, 256, 4), (2, 37, 4, 16, 4)):
        x = (rng.standard_normal((B, 1, T, F)) * 20 - 40).astype(np.float32)
        xd = torch.from_numpy(x).to(dev)
        imgs = sed_amd.repeat_mode_images(xd, spec, ratio)
        positions = list(range(0, (ratio - 1) * spec + 1, spec))
        assert len(imgs) == len(positions) == ratio
        for im, pos in zip(imgs, positions):
            want = of.htsat_repeat_wat2img(x, pos, spec, ratio)
            assert im.shape == want.shape == (B, 1, 4 * F, spec)
            assert np.max(np.abs(im.cpu().numpy() - want)) <= 2e-6 * np.max(np.abs(x)), (T, pos)
        odd = sed_amd.repeat_wat2img(xd, 3, spec, ratio)                               # any cur_pos, as in training (:785)
        assert np.max(np.abs(odd.cpu().numpy() - of.htsat_repeat_wat2img(x, 3, spec, ratio))) <= 2e-6 * np.max(np.abs(x))
    with pytest.raises(ValueError):
        sed_amd.repeat_wat2img(xd, 16 * 4 - 15, 16, 4)                                 # fewer than spec_size frames left
    x = (rng.standard_normal((2, 1, 300, 64)) * 20 - 40).astype(np.float32)           # 1024 // 300 = 3 tiles -> 900 frames
    got = sed_amd.infer_mode_image(torch.from_numpy(x).to(dev))
    want = of.htsat_reshape_wav2img(np.tile(x, (1, 1, 3, 1)), 256, 4)
    assert np.max(np.abs(got.cpu().numpy() - want)) <= 2e-6 * np.max(np.abs(x))


def test_state_dict_and_install_on_reference_shaped_model():
    import sed_amd
    from torch import nn
    dev = _dev()

    class Host(nn.Module):                                               # attribute names of HTSAT_Swin_Transformer
        def __init__(self):
            super().__init__()
            self.spectrogram_extractor = sed_amd.Spectrogram(n_fft=1024, hop_length=320, win_length=1024, window="hann",
                                                             center=True, pad_mode="reflect", freeze_parameters=True)
            self.logmel_extractor = sed_amd.LogmelFilterBank(sr=32000, n_fft=1024, n_mels=64, fmin=50, fmax=14000,
                                                             ref=1.0, amin=1e-10, top_db=None, freeze_parameters=True)
            self.bn0 = nn.BatchNorm2d(64)

        def forward(self, x):
            x = self.spectrogram_extractor(x)
            x = self.logmel_extractor(x)
            x = x.transpose(1, 3)
            x = self.bn0(x)
            return x.transpose(1, 3)

    m = Host().to(dev).eval()
    keys = set(m.state_dict())
    assert {"spectrogram_extractor.stft.conv_real.weight", "spectrogram_extractor.stft.conv_imag.weight",
            "logmel_extractor.melW"} <= keys
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["bn0.running_mean"] = torch.linspace(-40, -10, 64, device=dev)
    sd["bn0.running_var"] = torch.linspace(50, 200, 64, device=dev)
    sd["bn0.weight"] = torch.linspace(0.5, 1.5, 64, device=dev)
    sd["bn0.bias"] = torch.linspace(-1, 1, 64, device=dev)
    m.load_state_dict(sd, strict=True)
    x = torch.from_numpy(of.synth_batch(2, 32000, 32000, first=50)).to(dev)
    want = m(x)
    sed_amd.install_fused_frontend(m)
    got = m(x)
    assert got.shape == want.shape and torch.max(torch.abs(got - want)).item() < 1e-4
    fused = m.spectrogram_extractor
    fused.set_batchnorm(m.bn0)                                           # bn0 folded into the kernel epilogue
    assert torch.max(torch.abs(fused(x) - want)).item() < 1e-4
    fused.set_affine(None, None)
    # the keys did not move: a checkpoint saved from the unfused pair loads strictly AFTER the install too, which is the
    # order init_model uses (construct, then load_state_dict: src/models/htsat/model.py:151-162), and the mel matrix it
    # carries is the one the fused kernel then applies
    assert set(m.state_dict()) == keys
    fresh = Host().to(dev).eval()
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["logmel_extractor.melW"] = sd["logmel_extractor.melW"] * 0.5     # a checkpoint with its own mel matrix
    fresh.load_state_dict(sd2, strict=True)
    want2 = fresh(x)
    m.load_state_dict(sd2, strict=True)
    got2 = m(x)
    assert torch.max(torch.abs(got2 - want2)).item() < 1e-4
    assert torch.max(torch.abs(got2 - got)).item() > 0.1                  # (halving the matrix moves every value by 3 dB / sigma)
    late = sed_amd.install_fused_frontend(Host().to(dev).eval())         # install first, load second
    late.load_state_dict(sd2, strict=True)
    assert torch.equal(late(x), got2)
    bad = {k: v.clone() for k, v in sd2.items()}
    bad["spectrogram_extractor.stft.conv_real.weight"] = bad["spectrogram_extractor.stft.conv_real.weight"] * 1.01
    with pytest.raises(RuntimeError):
        late.load_state_dict(bad, strict=True)


def test_full_size_properties_config2():
    """BASELINE.json configs[1] (256 x 10 s @ 32 kHz): size-independent properties instead of a full oracle pass."""
    import sed_amd
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(1234)
    x = (0.1 * torch.randn(256, 320000, generator=g)).to(dev)
    x[:, 100000:116000] = 0.0
    fe = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat())
    y = fe(x)
    assert y.shape == (256, 1, 1001, 64) and torch.isfinite(y).all()
    for b in (0, 17, 255):                                               # batch invariance, bit exact
        assert torch.equal(fe(x[b:b + 1]), y[b:b + 1])
    shifted = fe(x[:4, 320 * 7:])                                        # time-shift by 7 hops: interior frames equal
    assert torch.equal(shifted[:, :, 4:900], y[:4, :, 11:907])
    sil = y[:, 0, 316:360]                                               # frames fully inside the silent span
    assert torch.all(sil == sil[0, 0, 0])
    spec = sed_amd.Spectrogram(n_fft=1024, hop_length=320)(x[:8])        # Parseval on the windowed frames
    w = sed_amd.hann_window(1024, 1024).to(dev)
    fr = torch.nn.functional.pad(x[:8, None], (512, 512), mode="reflect")[:, 0].unfold(1, 1024, 320) * w
    energy = (fr.double() ** 2).sum(-1)
    s = spec[:, 0].double()
    pars = (s[..., 0] + s[..., -1] + 2 * s[..., 1:-1].sum(-1)) / 1024
    assert torch.max(torch.abs(pars - energy) / energy.clamp_min(1e-12)).item() < 1e-5
    ref = of.htsat_logmel(x[250:252].cpu().numpy(), mode="gold")
    near = ref >= ref.max(-1, keepdims=True) - NEAR_DB
    assert np.max(np.abs(y[250:252].cpu().numpy() - ref)[near]) < 1e-3


def test_long_form_recording_equals_overlapped_chunks():
    """BASELINE.json configs[4]: a 1-hour recording processed whole equals the same recording processed in 10-minute chunks
    that overlap by the frames' reach (n_fft/2 samples of context on either side of a chunk's frames): bit exact."""
    import sed_amd
    dev = _dev()
    sr, hop, n_fft = 32000, 320, 1024
    g = torch.Generator(device=dev).manual_seed(5)
    n = 3600 * sr
    x = torch.randn(1, n, generator=g, device=dev) * 0.1
    fe = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat(), layout="btm")
    whole = fe(x)
    assert whole.shape == (1, 1 + n // hop, 64) and torch.isfinite(whole).all()
    frames_per_chunk = 600 * sr // hop
    for c in (0, 1, 3, 5):                                              # chunks incl. the first (reflect edge) and the last
        f0 = c * frames_per_chunk
        f1 = min(whole.shape[1], f0 + frames_per_chunk)
        lo = max(0, (f0 - 10) * hop)                                     # ten hops of context (> n_fft/2) on either side,
        hi = min(n, (f1 - 1 + 10) * hop)                                 # chunk start on the whole recording's frame grid
        part = fe(x[:, lo:hi].contiguous())
        first = f0 - lo // hop                                           # frame of `part` that is frame f0 of `whole`
        got, want = part[:, first:first + (f1 - f0)], whole[:, f0:f1]
        if hi == n:                                                      # the recording's own last frames reflect the same samples
            assert torch.equal(got, want), c
        else:
            assert torch.equal(got, want), (c, (got != want).nonzero()[:4].tolist())


def test_all_flavours_against_committed_vectors():
    """Every front-end flavour against tests/golden/frontend_golden.npz (oracle outputs frozen on seeded clips)."""
    import importlib.util
    import sed_amd
    dev = _dev()
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("gen_frontend_golden", os.path.join(here, "gen_frontend_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    vec = np.load(os.path.join(here, "frontend_golden.npz"))
    db, neper_mag, neper_pow = NEAR_DB, NEAR_DB / 20.0 * np.log(10.0), NEAR_DB / 10.0 * np.log(10.0)
    for name in gen.CASES:
        x, sr = gen.inputs(name)
        xd = torch.from_numpy(x).to(dev)
        gold, ref32 = vec[f"{name}_gold"], vec[f"{name}_ref32"]
        if name == "htsat":
            got, span, axis_last = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat())(xd)[:, 0].cpu().numpy(), db, True
        elif name == "yamnet":
            got, span, axis_last = sed_amd.yamnet_log_mel(xd)[0].cpu().numpy(), neper_mag, True
        elif name == "crnn":
            got, span, axis_last = sed_amd.extract_mbe(x, 44100, 2048, 40, 1024), neper_mag, False
        elif name == "db_mel":
            got = sed_amd.get_mel_spec_db(x, 44100)
            assert got.shape == gold.shape and np.max(np.abs(got - gold)) < 1e-3, name
            continue
        else:
            got = sed_amd.kaldi_fbank(xd, sample_frequency=sr, num_mel_bins=128, frame_shift=10).cpu().numpy()
            got, span, axis_last = got.reshape(gold.shape), neper_pow, True
        if not axis_last:                                                # (B, mel, T) -> mel last for the per-frame peak
            got, gold, ref32 = got.swapaxes(-1, -2), gold.swapaxes(-1, -2), ref32.swapaxes(-1, -2)
        _check(got, gold, ref32, span, deep_factor=2.5 if name.startswith("kaldi") else 1.5)


def test_random_htsat_style_configurations():
    """Seeded sweep over sample rate, n_fft, hop (any multiple of 4), mel count and band edges in the torchlibrosa
    flavour (centred, reflect padding, power, Slaney mel, dB) against the oracle at the usual tolerance."""
    import sed_amd
    dev = _dev()
    rng = np.random.default_rng(2024)
    for case in range(14):
        n_fft = int(rng.choice([512, 1024, 2048]))
        sr = int(rng.choice([8000, 16000, 22050, 32000, 44100, 48000]))
        hop = 4 * int(rng.integers(8, n_fft // 4 + 1))
        n_mels = int(rng.integers(1, 129))
        fmin = float(rng.choice([0.0, 20.0, 50.0, 125.0, 300.0]))
        fmax = float(rng.uniform(max(fmin + 500.0, 0.2 * sr), 0.5 * sr))
        n = int(rng.integers(n_fft // 2 + 1, 3 * sr // 4))
        B = int(rng.integers(1, 4))
        x = of.synth_batch(B, n, sr, first=300 + case)
        cfg = sed_amd.FrontendConfig.htsat(sample_rate=sr, window_size=n_fft, hop_size=hop, mel_bins=n_mels, fmin=fmin, fmax=fmax)
        got = sed_amd.LogMelFrontend(cfg)(torch.from_numpy(x).to(dev))[:, 0].cpu().numpy()
        gold = of.htsat_logmel(x, sr, n_fft, hop, n_mels, fmin, fmax, mode="gold")[:, 0]
        ref32 = of.htsat_logmel(x, sr, n_fft, hop, n_mels, fmin, fmax, mode="ref32")[:, 0]
        assert got.shape == gold.shape == (B, 1 + n // hop, n_mels), (case, sr, n_fft, hop, n_mels)
        print(case, sr, n_fft, hop, n_mels, fmin, round(fmax, 1), n, end=": ")
        _check(got, gold, ref32, NEAR_DB)


def test_image_builders_beyond_one_launch_of_clips():
    """More clips than one launch holds in its grid (65535): the image builders go on in further launches; clips on either
    side of the boundary against the same call on that clip alone."""
    import sed_amd
    dev = _dev()
    B, T, F, spec, ratio = 70000, 9, 16, 16, 1
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn((B, 1, T, F), generator=g, device=dev) * 20 - 40
    frames = torch.randint(0, T + 1, (B,), generator=g, device=dev, dtype=torch.int32)
    dense = sed_amd.reshape_wav2img(x, spec, ratio)
    ragged = sed_amd.reshape_wav2img(x, spec, ratio, frames=frames)
    assert dense.shape == ragged.shape == (B, 1, spec, spec)
    for b in (0, 65534, 65535, 65536, B - 1):
        assert torch.equal(dense[b:b + 1], sed_amd.reshape_wav2img(x[b:b + 1].contiguous(), spec, ratio)), b
        n = int(frames[b])
        if n:
            assert torch.equal(ragged[b:b + 1], sed_amd.reshape_wav2img(x[b:b + 1, :, :n].contiguous(), spec, ratio)), b
        else:
            assert not ragged[b].any()
    x4 = x[:, :, :, :4].contiguous()                                  # 4 mel bins, spec 16, ratio 4: 64 frames, windows of 16
    rep = sed_amd.repeat_wat2img(x4, 5, 16, 4)
    for b in (65534, 65535, B - 1):
        assert torch.equal(rep[b:b + 1], sed_amd.repeat_wat2img(x4[b:b + 1].contiguous(), 5, 16, 4)), b
