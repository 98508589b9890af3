"""The polyphase-resampling restatement against scipy.signal.resample_poly itself (scipy is installed here; it is what
librosa.resample(res_type='polyphase') calls): identical bits for fp32 input, filter included."""
import math

import numpy as np
import pytest

from oracle import resample as R

PAIRS = ((44100, 16000, 44100), (44100, 16000, 10007), (44100, 32000, 30001), (48000, 16000, 4801), (32000, 16000, 999),
         (16000, 44100, 5000), (22050, 16000, 22051), (8000, 16000, 100), (44100, 16000, 5), (16000, 16000, 77))


@pytest.mark.parametrize("orig,target,n", PAIRS)
def test_restatement_has_scipys_bits(orig, target, n):
    ss = pytest.importorskip("scipy.signal")
    rng = np.random.default_rng(orig + target + n)
    x = (rng.standard_normal((2, n)) * 0.3).astype(np.float32)
    g = math.gcd(orig, target)
    up, down = target // g, orig // g
    want = ss.resample_poly(x, up, down, axis=-1)
    got = R.resample_poly_f32(x, orig, target)
    assert got.dtype == np.float32 and got.shape == want.shape == (2, R.out_len(n, up, down))
    assert np.array_equal(got, want)
    if up != down:
        h, half_len = R.design_filter(up, down)
        hs = ss.firwin(2 * half_len + 1, 1.0 / max(up, down), window=("kaiser", 5.0)).astype(np.float32) * np.float32(up)
        assert np.array_equal(h, hs)


def test_librosa_wrapper_length_rule():
    x = np.ones((3, 1001), np.float32)
    y = R.librosa_resample_polyphase(x, 44100, 16000)
    assert y.shape == (3, int(np.ceil(1001 * 16000 / 44100)))
    assert R.librosa_resample_polyphase(x, 16000, 16000) is x
