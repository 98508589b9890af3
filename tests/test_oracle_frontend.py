"""The front-end oracle: shapes, floors, and a cross-check against torch.stft (independent implementation)."""
import os

import numpy as np
import pytest
import torch

from oracle import frontend as of


def test_mel_matrix_htsat_structure():
    w = of.librosa_mel(32000, 1024, 64, 50, 14000)
    assert w.shape == (64, 513) and w.dtype == np.float32
    assert int((w != 0).sum()) == 866
    nz = [np.flatnonzero(r) for r in w]
    assert all(np.all(np.diff(i) == 1) for i in nz)            # every filter is one contiguous band
    assert all(nz[i][0] <= nz[i + 1][0] for i in range(63))


def test_mel_matrix_tf_structure():
    w = of.tf_linear_to_mel_weight_matrix()
    assert w.shape == (257, 64) and np.all(w[0] == 0) and w.max() <= 1.0
    assert np.all(w.sum(axis=0) > 0)


@pytest.mark.parametrize("n_fft,hop,center,mode", [(1024, 320, True, "reflect"), (2048, 1024, True, "constant"),
                                                    (512, 160, False, "reflect")])
def test_power_spectrogram_vs_torch_stft(n_fft, hop, center, mode):
    x = of.synth_batch(2, 16000, 16000)
    got = of.power_spectrogram_gold(x, n_fft, hop, center=center, pad_mode=mode)
    win = torch.hann_window(n_fft, periodic=True, dtype=torch.float64)
    s = torch.stft(torch.from_numpy(x).double(), n_fft, hop, n_fft, win, center=center, pad_mode=mode,
                   return_complex=True)
    want = (s.abs() ** 2).transpose(1, 2).numpy()
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) <= 1e-9 * max(1.0, want.max())
    r32 = of.power_spectrogram_ref32(x, n_fft, hop, center=center, pad_mode=mode)
    assert np.max(np.abs(r32 - want)) <= 2e-5 * want.max()


def test_shapes_and_floors():
    x = np.zeros((2, 320000), np.float32)
    y = of.htsat_logmel(x, mode="ref32")
    assert y.shape == (2, 1, 1001, 64) and np.all(y == y.flat[0]) and abs(float(y.flat[0]) + 100.0) < 2e-5
    lm, patches = of.yamnet_logmel(np.zeros((1, 160000), np.float32), mode="ref32")
    assert lm.shape == (1, 1008, 64) and patches.shape == (1, 20, 96, 64)
    assert np.all(lm == np.log(np.float32(0.001)))
    assert of.yamnet_padded_length(320000) == 322800
    lm20, p20 = of.yamnet_logmel(np.zeros((1, 320000), np.float32), mode="ref32")
    assert lm20.shape[1] == 2016 and p20.shape[1] == 41
    m = of.crnn_extract_mbe(np.zeros(882000, np.float32), mode="ref32")
    assert m.shape == (40, 862) and np.all(m == np.log(np.float32(1e-9)))
    d = of.librosa_db_mel(of.synth_clip(0, 44100, 44100), mode="ref32")
    assert d.shape == (64, 1 + 44100 // 512) and d.max() == 0.0 and d.min() >= -80.0


def test_gold_vs_ref32_tolerance_broadband():
    x = of.synth_batch(2, 32000, 32000)
    g = of.htsat_logmel(x, mode="gold")
    r = of.htsat_logmel(x, mode="ref32")
    peak = g.max(axis=-1, keepdims=True)
    near = g >= peak - 60.0
    assert np.max(np.abs(g - r)[near]) < 1e-3


def test_pcm16_convention():
    s = np.arange(-32768, 32768, dtype=np.int16)
    f = of.pcm16_to_f32(s)
    assert f.dtype == np.float32
    assert np.array_equal(f, (s.astype(np.float32) / np.float32(32767.0)))   # fp32 division gives the same bits


def test_reshape_wav2img_restatement_matches_torch_interpolate():
    """The reference calls torch.nn.functional.interpolate(mode='bicubic', align_corners=True) and folds the time axis
    (htsat.py:729-744); torch is importable here, so the restatement is pinned against the very function."""
    import torch
    rng = np.random.default_rng(3)
    for (B, T, F, spec, ratio) in ((2, 1001, 64, 256, 4), (1, 1024, 64, 256, 4), (3, 37, 4, 16, 4), (1, 5, 8, 8, 1)):
        x = (rng.standard_normal((B, 1, T, F)) * 20 - 40).astype(np.float32)
        got = of.htsat_reshape_wav2img(x, spec, ratio)
        y = torch.from_numpy(x)
        if T < spec * ratio:
            y = torch.nn.functional.interpolate(y, (spec * ratio, F), mode="bicubic", align_corners=True)
        y = y.permute(0, 1, 3, 2).contiguous()
        y = y.reshape(B, 1, F, ratio, spec).permute(0, 1, 3, 2, 4).contiguous().reshape(B, 1, F * ratio, spec)
        assert got.shape == tuple(y.shape)
        assert np.max(np.abs(got - y.numpy())) <= 1e-6 * np.max(np.abs(x)) * 2      # a few ulp of the largest value
        if T == spec * ratio:
            assert np.array_equal(got, y.numpy())


def test_repeat_mode_and_framewise_restatements_match_torch():
    """repeat_wat2img (htsat.py:747-760), interpolate and pad_framewise_output (utils.py:158-192) written with the
    torch calls the reference makes, against the numpy restatements."""
    import torch
    rng = np.random.default_rng(6)
    for (B, T, F, spec, ratio, pos) in ((2, 1001, 64, 256, 4, 256), (1, 1024, 64, 256, 4, 768), (2, 37, 4, 16, 4, 5)):
        x = (rng.standard_normal((B, 1, T, F)) * 20 - 40).astype(np.float32)
        y = torch.from_numpy(x)
        if T < spec * ratio:
            y = torch.nn.functional.interpolate(y, (spec * ratio, F), mode="bicubic", align_corners=True)
        y = y.permute(0, 1, 3, 2).contiguous()[:, :, :, pos:pos + spec].repeat(repeats=(1, 1, 4, 1))
        got = of.htsat_repeat_wat2img(x, pos, spec, ratio)
        assert got.shape == tuple(y.shape) == (B, 1, 4 * F, spec)
        assert np.max(np.abs(got - y.numpy())) <= 2e-6 * np.max(np.abs(x))
    p = rng.random((3, 7, 6)).astype(np.float32)
    t = torch.from_numpy(p)
    up = t[:, :, None, :].repeat(1, 1, 32, 1).reshape(3, 7 * 32, 6)
    assert np.array_equal(of.htsat_interpolate(p, 32), up.numpy())
    padded = torch.cat((up, up[:, -1:, :].repeat(1, 250 - up.shape[1], 1)), dim=1)
    assert np.array_equal(of.htsat_pad_framewise_output(up.numpy(), 250), padded.numpy())


def _frontend_golden():
    import importlib.util
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("gen_frontend_golden", os.path.join(here, "gen_frontend_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    return gen, np.load(os.path.join(here, "frontend_golden.npz"))


def test_oracle_reproduces_committed_frontend_vectors():
    """The fp64 outputs of every front-end flavour on the seeded clips equal the committed vectors (stored as float32):
    the oracle cannot change without this test noticing."""
    gen, gold = _frontend_golden()
    for name in gen.CASES:
        got = np.asarray(gen.outputs(name, "gold"), np.float64)
        want = gold[f"{name}_gold"].astype(np.float64)
        assert got.shape == want.shape, name
        assert np.max(np.abs(got - want)) <= 1e-5 + 1e-6 * np.max(np.abs(want)), name     # float32 storage of dB / neper values


def test_cpu_baselines_of_the_bench_compute_the_oracles_values():
    """The CPU formulations bench.py times (oracle/cpu_baseline.py) are the oracle's features / events, not something cheaper:
    the TensorFlow-style YAMNet features against yamnet_logmel, and the reference-style decode chain against the oracle chain."""
    import torch
    from oracle import cpu_baseline as cb
    from oracle import decode as od
    x = of.synth_batch(2, 48000, 16000)
    got = cb.YamnetCpuFrontend()(torch.from_numpy(x)).numpy()
    want = of.yamnet_logmel(x, mode="gold")
    want = want[0] if isinstance(want, tuple) else want
    assert got.shape == np.asarray(want).shape and np.abs(got - np.asarray(want)).max() < 2e-4
    rng = np.random.default_rng(3)
    from scipy.ndimage import gaussian_filter1d
    probs = (1.0 / (1.0 + np.exp(-8.0 * gaussian_filter1d(rng.standard_normal((6, 300, 3)), 5.0, axis=1)))).astype(np.float32)
    files, labels, gaps = [f"f{i}" for i in range(6)], {0: "a", 1: "b", 2: "c"}, (0.3, 1.0)
    raw = od.decode_predictions(od.smooth_and_threshold(probs), files, labels, 43.0)
    merged = [{k: od.merge_close_events(v, g) for k, v in raw.items()} for g in gaps]
    ms, n, el, same = cb.time_crnn_chain_cpu(probs, files, labels, 43.0, gaps, raw, merged, budget_s=0.05)
    assert same is True and n >= 1 and ms > 0
