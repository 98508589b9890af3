"""GPU parity against a third implementation: the HIP front-end (through the C ABI) against `transformers.audio_utils`,
without the oracle in between.

`transformers` ships numpy restatements of librosa's and torchaudio's feature code for its own feature extractors
(CLAP's, whose audio encoder is HTS-AT; AST's, whose input is the Kaldi filterbank the Audio-Mamba classifier takes).
The calls below are the ones those extractors make.  Bar: 1e-3 in the log domain on every mel bin within 60 dB of its
frame's peak, as in test_gpu_frontend.py (deeper bins are compared there against the fp32 formulation of the reference;
`audio_utils` keeps its spectrum in complex64 and is no better placed to judge them than the kernel is).
"""
import numpy as np
import pytest
import torch

from oracle import frontend as of

pytestmark = pytest.mark.gpu

au = pytest.importorskip("transformers.audio_utils")

NEAR_DB = 60.0


def _dev():
    import sed_amd  # noqa: F401  (fails loudly if the extension is missing)
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def _near_err(got, want, span):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape
    near = want >= want.max(axis=-1, keepdims=True) - span
    e = float(np.abs(got - want)[near].max())
    print(f"near-bin max err {e:.3e} over {int(near.sum())} of {near.size} bins")
    return e


def test_htsat_logmel_against_the_clap_extractor_call():
    import sed_amd
    dev = _dev()
    sr, n_fft, hop = 32000, 1024, 320
    x = of.synth_batch(3, 3 * sr, sr, first=200)
    got = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat())(torch.from_numpy(x).to(dev)).cpu().numpy()[:, 0]
    filt = au.mel_filter_bank(n_fft // 2 + 1, 64, 50.0, 14000.0, sr, norm="slaney", mel_scale="slaney")
    win = au.window_function(n_fft, "hann")
    for b in range(len(x)):
        want = au.spectrogram(x[b], win, frame_length=n_fft, hop_length=hop, power=2.0, center=True, pad_mode="reflect",
                              mel_filters=filt, log_mel="dB", reference=1.0, min_value=1e-10, db_range=None).T
        assert _near_err(got[b], want, NEAR_DB) <= 1e-3


def test_audio_mamba_input_against_the_ast_extractor():
    """external/audio_mamba/src/dataloader.py:94-152, 219-221 with the AudioSet statistics both sides hold."""
    import sed_amd
    from transformers import ASTFeatureExtractor
    _dev()
    fe = ASTFeatureExtractor()
    x = of.synth_batch(2, 5 * 16000 + 77, 16000, first=210).astype(np.float64)
    x = (x - x.mean(axis=1, keepdims=True)).astype(np.float32)       # the data loader centres the clip, the extractor does not
    got = sed_amd.wav2fbank(torch.from_numpy(x), 16000, melbins=128, target_length=1024, norm_mean=fe.mean,
                            norm_std=fe.std).cpu().numpy()
    want = fe([c for c in x], sampling_rate=16000, return_tensors="np")["input_values"]
    assert got.shape == want.shape == (2, 1024, 128)
    T = 1 + (x.shape[1] - 400) // 160
    scale = 2.0 * fe.std
    assert _near_err(got[:, :T], want[:, :T], NEAR_DB / 10.0 * np.log(10.0) / scale) <= 1e-3 / scale + 1e-6
    assert np.allclose(got[:, T:], want[:, T:], rtol=0, atol=1e-6)   # zero-padded frames after normalisation


def test_yamnet_features_against_audio_utils():
    import sed_amd
    _dev()
    L = of.yamnet_padded_length(3 * 16000)
    x = of.synth_batch(2, L, 16000, first=220)
    got, _ = sed_amd.yamnet_log_mel(torch.from_numpy(x))
    filt = au.mel_filter_bank(257, 64, 125.0, 7500.0, 16000, norm=None, mel_scale="htk", triangularize_in_mel_space=True)
    for b in range(len(x)):
        mel = au.spectrogram(x[b], au.window_function(400, "hann"), frame_length=400, hop_length=160, fft_length=512,
                             power=1.0, center=False, mel_filters=filt, mel_floor=0.0, dtype=np.float64).T
        assert _near_err(got[b].cpu().numpy(), np.log(mel + 0.001), NEAR_DB / 20.0 * np.log(10.0)) <= 1e-3


def test_crnn_and_db_mel_against_audio_utils():
    import sed_amd
    _dev()
    sr, n_fft = 44100, 2048
    y = of.synth_clip(230, 2 * sr, sr)
    win = au.window_function(n_fft, "hann")
    filt40 = au.mel_filter_bank(n_fft // 2 + 1, 40, 0.0, sr / 2.0, sr, norm="slaney", mel_scale="slaney")
    mag = au.spectrogram(y, win, frame_length=n_fft, hop_length=1024, power=1.0, center=True, pad_mode="constant",
                         mel_filters=filt40, mel_floor=0.0, dtype=np.float64)
    mbe = sed_amd.extract_mbe(y, sr, n_fft, 40, 1024)
    assert _near_err(mbe.T, np.log(mag + 1e-9).T, NEAR_DB / 20.0 * np.log(10.0)) <= 1e-3
    filt64 = au.mel_filter_bank(n_fft // 2 + 1, 64, 0.0, float(sr // 2), sr, norm="slaney", mel_scale="slaney")
    power = au.spectrogram(y, win, frame_length=n_fft, hop_length=512, power=2.0, center=True, pad_mode="constant",
                           mel_filters=filt64, mel_floor=0.0, dtype=np.float64)
    want = au.power_to_db(power, reference=power.max(), min_value=1e-10, db_range=80.0)
    got = sed_amd.get_mel_spec_db(y[None], sr)[0]
    assert got.shape == want.shape and got.max() == 0.0 and got.min() >= -80.0
    assert _near_err(got.T, want.T, NEAR_DB) <= 1e-3
