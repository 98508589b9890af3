"""The front-end oracle against an implementation nobody here wrote: `transformers.audio_utils`.

librosa, torchlibrosa, torchaudio and tensorflow are not installed, and the reference ships no feature dump, so
nothing the reference holds pins `oracle/frontend.py`.  `transformers` is installed, and its `audio_utils` module is a
numpy implementation written to reproduce exactly those libraries: `mel_filter_bank(norm="slaney", mel_scale="slaney")`
is its `librosa.filters.mel`, `mel_scale="kaldi", triangularize_in_mel_space=True` its
`torchaudio.compliance.kaldi.get_mel_banks`, `spectrogram(..., log_mel="dB")` its `librosa.power_to_db` /
`librosa.feature.melspectrogram`, and the numpy branch of its `ASTFeatureExtractor` its
`torchaudio.compliance.kaldi.fbank` (the very call of external/audio_mamba/src/dataloader.py:127-129).  The calls below
are the ones its own feature extractors make (CLAP's for the HTS-AT encoder, AST's for the Kaldi filterbank).

This does not turn the oracle into one pinned by the reference; it does mean that every mel scale, the filterbank
normalisation, the framing options, the dB conversion and the Kaldi frame conditioning have met a second, independent
implementation.  Tolerances: the oracle computes in fp64; `audio_utils` forms its frames and filterbanks in fp64 but keeps
the complex spectrum in complex64 and (by default) returns fp32, so the bars are a few fp32 ulps of the values compared.
"""
import numpy as np
import pytest

au = pytest.importorskip("transformers.audio_utils")

from oracle import frontend as ofe  # noqa: E402


def _clip(seed, n, sr):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    x = 0.3 * np.sin(2 * np.pi * (220.0 + 40 * seed) * t) + 0.1 * np.sin(2 * np.pi * 3100.0 * t + 1.0)
    x += 0.05 * rng.standard_normal(n)
    x *= np.linspace(0.2, 1.0, n)
    return x.astype(np.float32)


@pytest.mark.parametrize("sr,n_fft,n_mels,fmin,fmax", [
    (32000, 1024, 64, 50.0, 14000.0),     # HTS-AT / PANNs (external/hts_audio_transformer/model/htsat.py:763-768)
    (44100, 2048, 40, 0.0, 22050.0),      # CRNN (src/models/crnn/process_audio.py:29-34)
    (44100, 2048, 64, 0.0, 22050.0),      # dB-mel (src/utils/audio_to_spectrograms.py:30-44)
    (16000, 512, 128, 0.0, 8000.0),
])
def test_slaney_filterbank(sr, n_fft, n_mels, fmin, fmax):
    mine = ofe.librosa_mel(sr, n_fft, n_mels, fmin, fmax)                                   # (M, F) fp32
    theirs = au.mel_filter_bank(n_fft // 2 + 1, n_mels, fmin, fmax, sr, norm="slaney", mel_scale="slaney").T
    assert mine.shape == theirs.shape
    # the same triangles; librosa (and the oracle) round to fp32 before the area normalisation, audio_utils after it
    np.testing.assert_allclose(mine, theirs, rtol=3e-7, atol=1e-12)
    assert np.array_equal(mine != 0, theirs.astype(np.float32) != 0)


def test_mel_scales():
    f = np.array([0.0, 20.0, 125.0, 999.0, 1000.0, 1001.0, 4000.0, 7500.0, 14000.0, 22050.0])
    np.testing.assert_allclose(ofe.hz_to_mel_slaney(f), au.hertz_to_mel(f, "slaney"), rtol=1e-14, atol=1e-14)
    m = np.linspace(0.0, 45.0, 91)
    np.testing.assert_allclose(ofe.mel_to_hz_slaney(m), au.mel_to_hertz(m, "slaney"), rtol=1e-14, atol=1e-12)


@pytest.mark.parametrize("sr,n_bins", [(16000, 128), (32000, 128), (44100, 64)])
def test_kaldi_mel_banks(sr, n_bins):
    mine = ofe.kaldi_mel_banks(n_bins, 512 if sr == 16000 else 1024 if sr == 32000 else 2048, float(sr))  # (M, F) with zero Nyquist
    n_fft = 2 * (mine.shape[1] - 1)
    theirs = au.mel_filter_bank(n_fft // 2 + 1, n_bins, 20.0, sr / 2, sr, norm=None, mel_scale="kaldi",
                                triangularize_in_mel_space=True).T
    # audio_utils evaluates the Nyquist bin as well (torchaudio leaves it out and pads a zero column); with
    # high_freq = Nyquist that bin lies on the last triangle's foot, i.e. it is zero either way
    np.testing.assert_allclose(mine, theirs, rtol=1e-12, atol=1e-12)


def test_htk_filterbank_of_yamnet():
    """tf.signal.linear_to_mel_weight_matrix (external/yamnet/features.py:52-58): HTK mel, triangles linear in mel, peak 1,
    DC bin zero.  audio_utils' HTK scale is 2595 * log10(1 + f / 700), TF's 1127 * ln(1 + f / 700): the same function up to
    rounding, and the triangles only use ratios of mel differences."""
    mine = ofe.tf_linear_to_mel_weight_matrix(64, 257, 16000, 125.0, 7500.0)                 # (F, M) fp32
    theirs = au.mel_filter_bank(257, 64, 125.0, 7500.0, 16000, norm=None, mel_scale="htk", triangularize_in_mel_space=True)
    np.testing.assert_allclose(mine, theirs, rtol=0, atol=1e-7)
    assert np.all(mine[0] == 0) and np.all(theirs[0] == 0)


def test_yamnet_log_mel():
    """external/yamnet/features.py:26-63: frames of 400 samples every 160 (no centring), periodic Hann, 512-point
    magnitude spectrum, the HTK bank above, ln(mel + 0.001)."""
    L = ofe.yamnet_padded_length(16000)
    x = _clip(5, L, 16000)
    mine, _ = ofe.yamnet_logmel(x, mode="gold")                                              # (1, T, 64)
    filt = au.mel_filter_bank(257, 64, 125.0, 7500.0, 16000, norm=None, mel_scale="htk", triangularize_in_mel_space=True)
    mel = au.spectrogram(x, au.window_function(400, "hann"), frame_length=400, hop_length=160, fft_length=512, power=1.0,
                         center=False, mel_filters=filt, mel_floor=0.0, dtype=np.float64).T
    assert mine[0].shape == mel.shape == (1 + (L - 400) // 160, 64)
    np.testing.assert_allclose(mine[0], np.log(mel + 0.001), rtol=0, atol=2e-6)


def test_htsat_logmel_through_the_clap_extractor_call():
    """torchlibrosa Spectrogram + LogmelFilterBank of the HTS-AT front-end (htsat.py:763-768: hann, center, reflect,
    power 2, ref 1, amin 1e-10, no top_db) as CLAP's feature extractor (whose audio encoder is HTS-AT) calls audio_utils."""
    sr, n_fft, hop = 32000, 1024, 320
    x = _clip(1, 2 * sr, sr)
    mine = ofe.htsat_logmel(x[None], sr, n_fft, hop, 64, 50, 14000, mode="gold")[0, 0]      # (T, M) fp64
    filt = au.mel_filter_bank(n_fft // 2 + 1, 64, 50.0, 14000.0, sr, norm="slaney", mel_scale="slaney")
    theirs = au.spectrogram(x, au.window_function(n_fft, "hann"), frame_length=n_fft, hop_length=hop, power=2.0,
                            center=True, pad_mode="reflect", mel_filters=filt, log_mel="dB", reference=1.0,
                            min_value=1e-10, db_range=None).T
    assert mine.shape == theirs.shape == (1 + len(x) // hop, 64)
    np.testing.assert_allclose(mine, theirs, rtol=0, atol=8e-6)                              # dB down to -100: one fp32 ulp of audio_utils' result


def test_db_mel_with_clip_maximum_and_floor():
    """librosa.feature.melspectrogram + power_to_db(ref=np.max) of src/utils/audio_to_spectrograms.py:30-44."""
    sr, n_fft, hop = 44100, 2048, 512
    x = _clip(2, sr, sr)
    mine = ofe.librosa_db_mel(x, sr, n_fft, hop, 64, mode="gold", pad_mode="constant")       # (M, T)
    filt = au.mel_filter_bank(n_fft // 2 + 1, 64, 0.0, float(sr // 2), sr, norm="slaney", mel_scale="slaney")
    power = au.spectrogram(x, au.window_function(n_fft, "hann"), frame_length=n_fft, hop_length=hop, power=2.0,
                           center=True, pad_mode="constant", mel_filters=filt, mel_floor=0.0, dtype=np.float64)
    theirs = au.power_to_db(power, reference=power.max(), min_value=1e-10, db_range=80.0)
    assert mine.shape == theirs.shape
    np.testing.assert_allclose(mine, theirs, rtol=0, atol=2e-6)
    assert mine.min() >= -80.0 and mine.max() == 0.0


def test_crnn_mel_band_energies():
    """librosa.stft magnitude -> mel -> ln(. + 1e-9) of src/models/crnn/process_audio.py:29-34."""
    sr, n_fft, hop = 44100, 2048, 1024
    x = _clip(3, sr, sr)
    mine = ofe.crnn_extract_mbe(x, sr, n_fft, 40, hop, mode="gold", pad_mode="constant")      # (M, T)
    filt = au.mel_filter_bank(n_fft // 2 + 1, 40, 0.0, sr / 2.0, sr, norm="slaney", mel_scale="slaney")
    mag = au.spectrogram(x, au.window_function(n_fft, "hann"), frame_length=n_fft, hop_length=hop, power=1.0,
                         center=True, pad_mode="constant", mel_filters=filt, mel_floor=0.0, dtype=np.float64)
    np.testing.assert_allclose(mine, np.log(mag + 1e-9), rtol=0, atol=2e-7)


@pytest.mark.parametrize("sr", [16000, 32000])
def test_kaldi_fbank_through_the_ast_extractor_call(sr):
    """torchaudio.compliance.kaldi.fbank(htk_compat=True, use_energy=False, window_type='hanning', num_mel_bins=128,
    dither=0.0, frame_shift=10) of external/audio_mamba/src/dataloader.py:127-129, as the numpy branch of transformers'
    ASTFeatureExtractor restates it."""
    win, hop = int(sr * 0.025), int(sr * 0.010)
    n_fft = 1 << (win - 1).bit_length()
    x = _clip(4, sr + 123, sr)
    x = x - x.mean()
    mine = ofe.kaldi_fbank(x[None], sample_frequency=float(sr), num_mel_bins=128, mode="gold")[0]   # (T, M)
    filt = au.mel_filter_bank(n_fft // 2 + 1, 128, 20.0, sr // 2, sr, norm=None, mel_scale="kaldi",
                              triangularize_in_mel_space=True)
    theirs = au.spectrogram(x, au.window_function(win, "hann", periodic=False), frame_length=win, hop_length=hop,
                            fft_length=n_fft, power=2.0, center=False, preemphasis=0.97, mel_filters=filt, log_mel="log",
                            mel_floor=1.192092955078125e-07, remove_dc_offset=True).T
    assert mine.shape == theirs.shape == (1 + (len(x) - win) // hop, 128)
    np.testing.assert_allclose(mine, theirs, rtol=0, atol=2e-6)                              # nepers down to -16: two fp32 ulps of audio_utils' result


def test_audio_mamba_input_through_the_ast_extractor():
    """AudiosetDataset._wav2fbank + the normalisation of __getitem__ (external/audio_mamba/src/dataloader.py:94-152,
    219-221: 128 bins, 1024 frames, (x + 4.2677393) / (2 * 4.5689974)) against transformers' ASTFeatureExtractor with its
    default configuration, which holds the same AudioSet statistics.  The extractor does not centre the waveform, the data
    loader does, so the clip is centred beforehand."""
    from transformers import ASTFeatureExtractor
    x = _clip(6, 3 * 16000 + 57, 16000).astype(np.float64)
    x = (x - x.mean()).astype(np.float32)
    fe = ASTFeatureExtractor()
    assert (fe.num_mel_bins, fe.max_length, fe.mean, fe.std) == (128, 1024, -4.2677393, 4.5689974)
    theirs = fe(x, sampling_rate=16000, return_tensors="np")["input_values"][0]              # (1024, 128)
    mine = ofe.aum_wav2fbank(x[None], 16000, 128, 1024, 10, -4.2677393, 4.5689974, mode="gold")[0]
    assert mine.shape == theirs.shape == (1024, 128)
    np.testing.assert_allclose(mine, theirs, rtol=0, atol=1e-6)


def test_windows():
    np.testing.assert_allclose(ofe.hann_periodic(1024), au.window_function(1024, "hann"), rtol=0, atol=1e-15)
    np.testing.assert_allclose(ofe.pad_center(ofe.hann_periodic(400), 512),
                               au.window_function(400, "hann", frame_length=512), rtol=0, atol=1e-15)


def test_htsat_image_fold_and_framewise_upsampling_against_clap():
    """reshape_wav2img (external/hts_audio_transformer/model/htsat.py:729-744) and interpolate (utils.py:158-172) against
    transformers' CLAP audio encoder, a port of HTS-AT that states the same fold with another order of reshapes."""
    import types
    import torch
    clap = pytest.importorskip("transformers.models.clap.modeling_clap")
    rng = np.random.default_rng(7)
    for T, spec, ratio in ((1001, 256, 4), (1024, 256, 4), (700, 512, 8)):
        x = rng.standard_normal((2, 1, T, spec // ratio)).astype(np.float32)
        enc = types.SimpleNamespace(spec_size=spec, freq_ratio=ratio)
        theirs = clap.ClapAudioEncoder.reshape_mel2img(enc, torch.from_numpy(x)).numpy()
        mine = ofe.htsat_reshape_wav2img(x, spec, ratio)
        assert mine.shape == theirs.shape == (2, 1, spec, spec)
        np.testing.assert_allclose(mine, theirs, rtol=0, atol=0 if T == spec * ratio else 4e-6)
    h = rng.standard_normal((3, 32, 6)).astype(np.float32)
    assert np.array_equal(ofe.htsat_interpolate(h, 32), clap.interpolate(torch.from_numpy(h), 32).numpy())
