"""GPU parity of the polyphase resampler (through the C ABI) against the CPU restatement, which tests/
test_oracle_resample.py holds to scipy.signal.resample_poly bit for bit.  Bar: identical values (np.array_equal)."""
import numpy as np
import pytest
import torch

from oracle import frontend as of
from oracle import resample as R

pytestmark = pytest.mark.gpu


def _dev():
    import sed_amd  # noqa: F401
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("orig,target", ((44100, 16000), (44100, 32000), (48000, 16000), (32000, 16000), (16000, 44100),
                                         (22050, 16000), (8000, 16000), (96000, 16000), (16000, 16000)))
def test_bits_of_the_restatement(orig, target):
    import sed_amd
    dev = _dev()
    rng = np.random.default_rng(orig * 3 + target)
    rs = sed_amd.Resampler(orig, target)
    plan = R.plan(orig, target)
    if plan["up"] != plan["down"]:
        assert (rs.up, rs.down, rs.taps, rs.n_pre_remove) == (plan["up"], plan["down"], plan["taps"], plan["n_pre_remove"])
        assert np.array_equal(rs.filter_table(), plan["table"])               # the filter designed in C++ has numpy's bits
    for n in (1, 5, 1000, 44100, 70001):
        x = (rng.standard_normal((3, n)) * 0.3).astype(np.float32)
        got = rs(torch.from_numpy(x).to(dev)).cpu().numpy()
        want = R.resample_poly_f32(x, orig, target)
        assert got.shape == want.shape and np.array_equal(got, want), n


@pytest.mark.parametrize("orig,target", ((192000, 8000), (96000, 12000), (22050, 32000), (11025, 48000), (37000, 6000), (7, 4200), (11025, 32000)))
def test_pairs_outside_the_tiled_kernels(orig, target):
    """More than 512 phases (22.05 -> 32 kHz is 640 : 441) or more than 128 coefficients per phase (a ratio above about
    6): the general kernel, one thread per output sample, same bits."""
    import sed_amd
    dev = _dev()
    rng = np.random.default_rng(orig + target)
    rs = sed_amd.Resampler(orig, target)
    plan = R.plan(orig, target)
    assert (rs.up, rs.down, rs.taps, rs.n_pre_remove) == (plan["up"], plan["down"], plan["taps"], plan["n_pre_remove"])
    assert np.array_equal(rs.filter_table(), plan["table"])
    for n in (1, 3, 999, 50001 if target <= 64 * orig else 2001):
        x = (rng.standard_normal((2, n)) * 0.3).astype(np.float32)
        got = rs(torch.from_numpy(x).to(dev)).cpu().numpy()
        want = R.resample_poly_f32(x, orig, target)
        assert got.shape == want.shape and np.array_equal(got, want), n
    s = rng.integers(-32768, 32768, size=(2, 20011 if target <= 64 * orig else 1501), dtype=np.int16)
    got = sed_amd.resample(torch.from_numpy(s).to(dev), orig_sr=orig, target_sr=target, pcm_divisor=32768.0).cpu().numpy()
    assert np.array_equal(got, R.librosa_resample_polyphase(s.astype(np.float32) / np.float32(32768.0), orig, target))


def test_librosa_signature_int16_and_errors():
    import sed_amd
    dev = _dev()
    x = of.synth_batch(2, 88200, 44100, first=11)
    y = sed_amd.resample(x, orig_sr=44100, target_sr=16000, res_type="polyphase")
    assert isinstance(y, np.ndarray) and np.array_equal(y, R.librosa_resample_polyphase(x, 44100, 16000))
    one = sed_amd.resample(torch.from_numpy(x[0]).to(dev), orig_sr=44100, target_sr=16000)
    assert torch.is_tensor(one) and one.shape == (32000,) and np.array_equal(one.cpu().numpy(), y[0])
    s = (x * 32767.0).round().astype(np.int16)                               # samples as stored in a wav file
    got = sed_amd.resample(torch.from_numpy(s).to(dev), orig_sr=44100, target_sr=16000, pcm_divisor=32768.0)
    want = R.resample_poly_f32(s.astype(np.float32) / np.float32(32768.0), 44100, 16000)
    assert np.array_equal(got.cpu().numpy(), want)
    s[0, :65536] = np.arange(-32768, 32768, dtype=np.int16)                  # every int16 value once
    for div in (32767.0, 32768.0, 1234.5, 3.0):                              # exact whatever the divisor
        got = sed_amd.resample(torch.from_numpy(s).to(dev), orig_sr=44100, target_sr=16000, pcm_divisor=div)
        want = R.resample_poly_f32(s.astype(np.float32) / np.float32(div), 44100, 16000)
        assert np.array_equal(got.cpu().numpy(), want), div
        odd = sed_amd.resample(torch.from_numpy(s).to(dev), orig_sr=44100, target_sr=48000, pcm_divisor=div)  # register path
        assert np.array_equal(odd.cpu().numpy(), R.librosa_resample_polyphase(s.astype(np.float32) / np.float32(div), 44100, 48000)), div
    stereo = torch.from_numpy(np.stack([x, -x], axis=1)).to(dev)             # (B, channels, L): last axis is resampled
    z = sed_amd.resample(stereo, orig_sr=44100, target_sr=16000)
    assert z.shape == (2, 2, 32000) and torch.equal(z[:, 1], -z[:, 0])
    with pytest.raises(ValueError, match="polyphase"):
        sed_amd.resample(x, orig_sr=44100, target_sr=16000, res_type="soxr_hq")
    with pytest.raises(ValueError):
        sed_amd.Resampler(3000001, 2)                                         # a filter of 60 million coefficients
    assert sed_amd.Resampler(44100, 16000)(torch.zeros(2, 0, device=dev)).shape == (2, 0)


def test_resampled_audio_feeds_the_yamnet_front_end():
    """librosa.load(file_path, sr=16000) then the YAMNet features (src/models/yamnet_train.py:46-47) without leaving the
    device: equals the same two steps done by the restatements."""
    import sed_amd
    dev = _dev()
    x = of.synth_batch(4, 441000, 44100, first=21)                           # 10 s @ 44.1 kHz
    y16 = sed_amd.resample(torch.from_numpy(x).to(dev), orig_sr=44100, target_sr=16000)
    assert y16.shape == (4, 160000)
    lm, patches = sed_amd.yamnet_log_mel(y16)
    want16 = R.librosa_resample_polyphase(x, 44100, 16000)
    assert np.array_equal(y16.cpu().numpy(), want16)
    g_lm, _ = of.yamnet_logmel(want16, mode="gold")
    near = g_lm >= g_lm.max(-1, keepdims=True) - 60.0 / 20.0 * np.log(10.0)
    assert lm.shape == g_lm.shape == (4, 1008, 64) and patches.shape == (4, 20, 96, 64)
    assert np.max(np.abs(lm.cpu().numpy() - g_lm)[near]) < 1e-3
