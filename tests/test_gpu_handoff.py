"""GPU parity of the detector -> classifier hand-off against numpy slicing (what cut_events_from_audio writes,
src/utils/cut_audio.py:33-41): bit exact."""
import numpy as np
import pytest
import torch

from oracle import frontend as of

pytestmark = pytest.mark.gpu


def _expected(x, sr, events, cap, div=None):
    out = np.zeros((len(events), cap), np.float32)
    lens = np.zeros(len(events), np.int32)
    for i, (s, a, b) in enumerate(events):
        sl = x[s][int(a * sr):int(b * sr)]
        if div is not None:
            sl = (sl.astype(np.float32) / np.float32(div)).astype(np.float32)
        lens[i] = len(sl)
        out[i, :min(len(sl), cap)] = sl[:cap]
    return out, lens


def test_cut_events_matches_numpy_slices():
    import sed_amd
    dev = torch.device("cuda", 0)
    sr = 44100
    x = of.synth_batch(4, 5 * sr + 17, sr, first=90)
    events = [(0, 0.0, 0.48), (1, 0.96, 2.4), (3, 4.8, 6.0), (2, 1.234, 1.2345), (2, 3.0, 2.0), (0, 0.333, 4.999),
              (1, 7.0, 8.0), (3, -0.5, 0.25), (0, 4.9, 5.5)]
    clips, lens = sed_amd.cut_events(torch.from_numpy(x).to(dev), sr, events)
    want, wl = _expected(x, sr, events, clips.shape[1])
    assert clips.shape[1] % 4 == 0 and np.array_equal(lens.cpu().numpy(), wl)
    assert np.array_equal(clips.cpu().numpy(), want)
    small, lens2 = sed_amd.cut_events(torch.from_numpy(x).to(dev), sr, events, cap=1001)     # truncation, odd capacity
    want2, _ = _expected(x, sr, events, 1001)
    assert np.array_equal(small.cpu().numpy(), want2) and np.array_equal(lens2.cpu().numpy(), wl)
    s16 = of.f32_to_pcm16(x)
    c16, l16 = sed_amd.cut_events(torch.from_numpy(s16).to(dev), sr, events)
    w16, _ = _expected(s16, sr, events, c16.shape[1], div=32768.0)
    assert np.array_equal(c16.cpu().numpy(), w16) and np.array_equal(l16.cpu().numpy(), wl)
    e0, l0 = sed_amd.cut_events(torch.from_numpy(x).to(dev), sr, [])
    assert e0.shape[0] == 0 and l0.shape == (0,)
    with pytest.raises(RuntimeError):
        sed_amd.cut_events(torch.from_numpy(x), sr, events)


def test_cut_events_views_with_overlapping_or_repeated_rows():
    """Views whose rows are less than a row length apart (one clip expanded to several streams, overlapping windows of
    a long recording, a column slice of a wider batch) must be read as the tensor they present."""
    import sed_amd
    dev = torch.device("cuda", 0)
    sr = 16000
    long = torch.from_numpy(of.synth_batch(1, 6 * sr, sr, first=97)[0]).to(dev)
    events = [(0, 0.1, 0.9), (1, 0.0, 1.5), (2, 1.0, 2.0), (3, 0.25, 1.75), (2, 1.9, 2.5)]
    views = {
        "expand": long[:2 * sr][None].expand(4, -1),                              # stride(0) == 0
        "unfold": long.unfold(0, 2 * sr, sr // 2)[:4],                             # windows of 2 s, 0.5 s apart
        "as_strided": torch.as_strided(long, (4, 2 * sr), (123, 1)),
        "columns": long[:4 * (sr + 5)].reshape(4, sr + 5)[:, 3:sr],               # stride(0) > row length
        "single": long[None, :2 * sr],
    }
    for name, v in views.items():
        ev = [e for e in events if e[0] < v.shape[0]]
        clips, lens = sed_amd.cut_events(v, sr, ev)
        want, wl = _expected(v.cpu().numpy(), sr, ev, clips.shape[1])
        assert np.array_equal(lens.cpu().numpy(), wl), name
        assert np.array_equal(clips.cpu().numpy(), want), name


def test_cut_events_from_audio_names_and_rows():
    import sed_amd
    dev = torch.device("cuda", 0)
    sr = 32000
    x = of.synth_batch(2, 10 * sr, sr, first=95)
    events_list = {"a/soundscape_01.wav": [{"event_onset": 0.48, "event_offset": 1.92, "label": "siren"},
                                            {"event_onset": 5.76, "event_offset": 6.24, "label": "siren"}],
                   "b/soundscape_02.wav": [{"event_onset": 2.4, "event_offset": 10.08, "label": "cough"}]}
    clips, lens, upd = sed_amd.cut_events_from_audio(torch.from_numpy(x).to(dev), sr,
                                                     ["a/soundscape_01.wav", "b/soundscape_02.wav"], events_list)
    assert [e["extracted_audio_filename"] for e in upd["a/soundscape_01.wav"]] == \
        ["soundscape_01_0.48_1.92.wav", "soundscape_01_5.76_6.24.wav"]
    assert upd["b/soundscape_02.wav"][0]["extracted_audio_filename"] == "soundscape_02_2.40_10.08.wav"
    assert upd["b/soundscape_02.wav"][0]["label"] == "cough" and "clip_index" not in events_list["b/soundscape_02.wav"][0]
    for name, s in (("a/soundscape_01.wav", 0), ("b/soundscape_02.wav", 1)):
        for ev in upd[name]:
            sl = x[s][int(ev["event_onset"] * sr):int(ev["event_offset"] * sr)]
            row = clips[ev["clip_index"]].cpu().numpy()
            assert lens[ev["clip_index"]].item() == len(sl)
            assert np.array_equal(row[:len(sl)], sl) and not row[len(sl):].any()
    fe = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat())
    assert fe(clips).shape[0] == 3                                        # the rows feed the classifier front-end as is


def test_logmel_of_events_equals_cut_then_front_end():
    """The fused hand-off (the front-end kernel reads the events straight out of the resident recordings) against the two
    steps it replaces, cut_events then LogMelFrontend: the same log-mel bit for bit, for fp32 and int16 recordings, events
    that start on any sample, truncated and empty events, negative (from-the-end) times, strided rows, all three layouts."""
    import sed_amd
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(31)
    sr, S, n = 32000, 6, 10 * 32000
    x = torch.from_numpy((0.2 * rng.standard_normal((S, n + 24))).astype(np.float32)).to(dev)
    events = [(0, 0.0, 1.0), (0, 0.48, 2.88), (1, 5.279999999999999, 8.16), (2, 9.5, 12.0), (3, 1.00003125, 1.5),
              (5, 0.0, 10.0), (4, 3.0, 3.0), (4, 4.0, 3.5), (2, -2.0, -0.5), (1, 0.96, 0.96 + 1e-4), (3, 2.5, 2.5 + 1023 / sr),
              (5, 7.77, 9.99), (0, 9.999, 10.0)]
    for rows, cap in ((x[:, :n], 96000), (x[:, 3:n + 3], 64000), (x[:, 8:n + 8], 96000), (x[:, :n], 1025)):
        for layout in ("b1tm", "btm", "bmt"):
            fe = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat(), layout=layout)
            clips, lens = sed_amd.cut_events(rows, sr, events, cap=cap)
            want = fe(clips)
            got, glens = sed_amd.logmel_of_events(fe, rows, sr, events, cap)
            assert got.shape == want.shape and torch.equal(got, want), (cap, layout)
            assert torch.equal(glens, lens)
    # other flavours of the fused kernel (dynamic mel shape, n_fft 2048 with constant padding, magnitude spectrum)
    for cfg in (sed_amd.FrontendConfig.htsat(mel_bins=40, fmin=0, fmax=16000), sed_amd.FrontendConfig.crnn(), sed_amd.FrontendConfig.db_mel()):
        fe = sed_amd.LogMelFrontend(cfg, layout="btm")
        clips, _ = sed_amd.cut_events(x[:, :n], sr, events, cap=40000)
        got, _ = sed_amd.logmel_of_events(fe, x[:, :n], sr, events, 40000)
        assert torch.equal(got, fe(clips))
    # int16 recordings: the front-end's own int16 path with the divisor of the hand-off
    xi = torch.from_numpy(rng.integers(-20000, 20000, (S, n + 8), dtype=np.int16)).to(dev)
    fe16 = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat(pcm_kind=1, pcm_divisor=32768.0), layout="btm")
    fe32 = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat(), layout="btm")
    for rows in (xi[:, :n], xi[:, 1:n + 1]):
        clips, lens = sed_amd.cut_events(rows, sr, events, cap=96000, pcm_divisor=32768.0)
        got, glens = sed_amd.logmel_of_events(fe16, rows, sr, events, 96000)
        assert torch.equal(got, fe32(clips)) and torch.equal(glens, lens)
    # segment tables already on the device are taken as they are; lengths beyond the row are cut to it inside the kernel
    flat = x[:, :n].contiguous()
    off = torch.tensor([0, 1 * n + 4000, 3 * n + 12345, 5 * n + 200000], dtype=torch.int64, device=dev)
    full = torch.tensor([50000, 96000, 120000, 70000], dtype=torch.int32, device=dev)
    got, _ = fe32.forward_segments(flat, off, full, 96000)
    want, _ = fe32.forward_segments(flat, off.cpu().numpy(), full.cpu().numpy(), 96000)    # host arrays: clamped on the host
    assert torch.equal(got, want)
    # no events, and refusals
    empty, elens = sed_amd.logmel_of_events(fe32, x[:, :n], sr, [], 96000)
    assert empty.shape == (0, fe32.num_frames(96000, dev), 64) and elens.numel() == 0
    with pytest.raises(ValueError):
        sed_amd.logmel_of_events(fe32, x[:, :n], sr, [(S, 0.0, 1.0)], 96000)
    with pytest.raises(RuntimeError):
        sed_amd.logmel_of_events(fe32, x[:, :n].cpu(), sr, events, 96000)
    with pytest.raises(RuntimeError):
        sed_amd.logmel_of_events(fe16, x[:, :n], sr, events, 96000)                  # sample type of the plan
    with pytest.raises(ValueError):
        sed_amd.logmel_of_events(sed_amd.LogMelFrontend(sed_amd.FrontendConfig.yamnet()), x[:, :n], 16000, events[:2], 48000)


def test_events_handed_to_the_audio_mamba_front_end():
    """Detector events -> Audio-Mamba input (main.py:123-127 cuts every event into a file of its own, which
    external/audio_mamba/src/dataloader.py:94-152, 219-221 then reads: fbank of the event's own samples, zero frames up to
    target_length, normalisation).  Here: cut_events gives the zero-padded batch and the lengths, wav2fbank(lengths=...) the
    features; compared event by event with the restatement on the event's own samples."""
    import sed_amd
    from oracle import frontend as of
    dev = torch.device("cuda", 0)
    sr = 44100                                                            # the rate of the reference's files
    x = of.synth_batch(3, 6 * sr, sr, first=330)
    events = [(0, 0.5, 2.75), (1, 1.0, 1.02), (2, 0.0, 5.99), (1, 3.3, 4.0), (0, 5.9, 7.5), (2, 2.0, 2.0)]  # one shorter than a frame, one running past the end, one empty
    clips, lengths = sed_amd.cut_events(torch.from_numpy(x).to(dev), sr, events)[:2]
    fb = sed_amd.wav2fbank(clips, sr, melbins=128, target_length=256, norm_mean=-4.2677393, norm_std=4.5689974,
                           lengths=lengths).cpu().numpy()
    assert fb.shape == (len(events), 256, 128)
    span = 60.0 / 10.0 * np.log(10.0) / (2 * 4.5689974)
    win, hop = int(sr * 0.025), int(sr * 0.010)
    pad_value = (0.0 + 4.2677393) / (2 * 4.5689974)
    for e, (s, a, b) in enumerate(events):
        piece = x[s, int(a * sr):int(b * sr)]
        assert int(lengths[e]) == len(piece)
        if len(piece) < win:
            assert np.allclose(fb[e], pad_value, rtol=0, atol=1e-6)        # no frame at all: zeros before the normalisation
            continue
        want = of.aum_wav2fbank(piece[None], sr, 128, 256, 10, -4.2677393, 4.5689974, mode="gold")[0]
        T = min(256, 1 + (len(piece) - win) // hop)
        near = want[:T] >= want[:T].max(-1, keepdims=True) - span
        assert np.max(np.abs(fb[e, :T] - want[:T])[near]) < 1e-3 / (2 * 4.5689974) + 1e-6, e
        assert np.allclose(fb[e, T:], want[T:], rtol=0, atol=1e-6), e     # zero frames behind the event
    # the two calls as one, and the fused form (the filterbank kernel reads the events out of the recordings)
    xd = torch.from_numpy(x).to(dev)
    fb2, len2 = sed_amd.fbank_of_events(xd, sr, events, melbins=128, target_length=256, norm_mean=-4.2677393,
                                        norm_std=4.5689974, fused=False)
    assert torch.equal(len2, lengths) and np.array_equal(fb2.cpu().numpy(), fb)
    fb3, len3 = sed_amd.fbank_of_events(xd, sr, events, melbins=128, target_length=256, norm_mean=-4.2677393, norm_std=4.5689974)
    assert torch.equal(len3, lengths) and fb3.shape == fb2.shape
    fb3 = fb3.cpu().numpy()
    worst = 0.0
    for e, (s, a, b) in enumerate(events):
        piece = x[s, int(a * sr):int(b * sr)]
        if len(piece) < win:
            assert np.allclose(fb3[e], pad_value, rtol=0, atol=1e-6)
            continue
        want = of.aum_wav2fbank(piece[None], sr, 128, 256, 10, -4.2677393, 4.5689974, mode="gold")[0]
        T = min(256, 1 + (len(piece) - win) // hop)
        near = want[:T] >= want[:T].max(-1, keepdims=True) - span
        assert np.max(np.abs(fb3[e, :T] - want[:T])[near]) < 1e-3 / (2 * 4.5689974) + 1e-6, e
        assert np.allclose(fb3[e, T:], want[T:], rtol=0, atol=1e-6), e
        worst = max(worst, float(np.max(np.abs(fb3[e, :T] - fb[e, :T])[near])) * 2 * 4.5689974)
    assert worst < 1e-4, worst                                            # fused against cut + wav2fbank, in nepers (measured 2.3e-5)
    for fused in (True, False):
        short = sed_amd.fbank_of_events(xd, sr, [events[1], events[5]], target_length=64, fused=fused)[0]
        assert short.shape == (2, 64, 128) and not short.any()            # nothing but events shorter than a window
        assert sed_amd.fbank_of_events(xd, sr, [], target_length=64, fused=fused)[0].shape == (0, 64, 128)
    # int16 recordings, as the wav files hold them
    xi = torch.from_numpy(of.f32_to_pcm16(x)).to(dev)
    f16a = sed_amd.fbank_of_events(xi, sr, events, target_length=256)[0]
    f16b = sed_amd.fbank_of_events(xi, sr, events, target_length=256, fused=False)[0]
    live = f16b > f16b.amax(dim=-1, keepdim=True) - 60.0 / 10.0 * np.log(10.0)
    assert float((f16a - f16b).abs()[live].max()) < 1e-4
    # without lengths the batch is what it looks like: clips of the padded length
    dense = sed_amd.wav2fbank(clips, sr, melbins=128, target_length=256)
    assert dense.shape == (len(events), 256, 128)


def test_events_as_files_of_their_own_for_the_htsat_classifier():
    """Stage 2 of the reference loads every written event at its natural length and feeds it alone
    (src/models/htsat/data.py:138-146, batch_size = 1 at src/models/htsat/classification.py:55-59): the STFT mirrors the
    event's own ends (htsat.py:763) and reshape_wav2img stretches the event's own 1 + len // hop frames to the image
    (htsat.py:729-744).  logmel_of_events(per_file=True) + reshape_wav2img(frames=...) do that for a batch of events of
    different lengths in two launches: compared event by event, bit for bit, with the front-end and the image kernel run on
    that event alone, and with the CPU restatement on the event's samples."""
    import sed_amd
    dev = torch.device("cuda", 0)
    sr = 44100
    S, n = 3, 12 * sr
    x = of.synth_batch(S, n, sr, first=77)
    xd = torch.from_numpy(x).to(dev)
    cfg = sed_amd.FrontendConfig.htsat(sample_rate=sr)
    hop, half = cfg.hop, cfg.n_fft // 2
    events = [(0, 0.5, 2.75), (1, 1.0, 1.0 + 513 / sr), (2, 0.0, 11.99), (1, 3.3, 4.0), (0, 11.0, 13.5), (2, 2.0, 2.0),
              (1, 5.0, 5.0 + 512 / sr), (0, 0.25, 0.25 + 4 * hop / sr), (2, 6.000011, 6.9), (1, 0.0, 12.0), (0, 7.0, 7.01)]
    # (513 samples: the shortest event the reflect padding takes; 512 and the empty one: none; one past the end of its file)
    cap = 10 * sr                                                          # the 12 s event is cut to the classifier's clip
    for layout in ("b1tm", "btm"):
        fe = sed_amd.LogMelFrontend(cfg, layout=layout)
        y, lengths, frames = sed_amd.logmel_of_events(fe, xd, sr, events, cap, per_file=True)
        T = fe.num_frames(cap, dev)
        y3 = y[:, 0] if layout == "b1tm" else y
        assert y3.shape == (len(events), T, 64) and frames.dtype == torch.int32
        for e, (s, a, b) in enumerate(events):
            piece = x[s, int(a * sr):int(b * sr)]
            assert int(lengths[e]) == len(piece)
            piece = piece[:cap]
            want_T = 1 + len(piece) // hop if len(piece) > half else 0
            assert int(frames[e]) == want_T, e
            assert not y3[e, want_T:].any(), e                             # nothing behind the event's own frames
            if want_T == 0:
                with pytest.raises((ValueError, RuntimeError)):            # as the reference's STFT refuses such a file
                    fe(torch.from_numpy(piece[None].copy()).to(dev))
                continue
            alone = fe(torch.from_numpy(piece[None].copy()).to(dev))
            alone3 = alone[:, 0] if layout == "b1tm" else alone
            assert torch.equal(y3[e, :want_T], alone3[0]), (layout, e)
    # against the restatement on the event's own samples, and the image of every event from its own frames
    fe = sed_amd.LogMelFrontend(cfg)
    y, lengths, frames = sed_amd.logmel_of_events(fe, xd, sr, events, cap, per_file=True)
    img = sed_amd.reshape_wav2img(y, 512, 8, frames=frames)
    assert img.shape == (len(events), 1, 512, 512)
    yn = y.cpu().numpy()
    for e, (s, a, b) in enumerate(events):
        piece = x[s, int(a * sr):int(b * sr)][:cap]
        Te = int(frames[e])
        if Te == 0:
            assert not img[e].any()
            continue
        gold = of.htsat_logmel(piece[None], sr=sr, mode="gold")
        assert gold.shape == (1, 1, Te, 64)
        near = gold[0, 0] >= gold[0, 0].max(-1, keepdims=True) - 60.0
        assert np.max(np.abs(yn[e, 0, :Te] - gold[0, 0])[near]) < 1e-3, e
        alone = sed_amd.reshape_wav2img(y[e:e + 1, :, :Te].contiguous(), 512, 8)
        assert torch.equal(img[e:e + 1], alone), e
        want = of.htsat_reshape_wav2img(yn[e:e + 1, :, :Te], 512, 8)
        assert np.max(np.abs(img[e:e + 1].cpu().numpy() - want)) <= 2e-6 * np.max(np.abs(yn[e, 0, :Te])), e
    # int16 recordings (the samples as the wav files hold them): the same, through the plan's own conversion
    xi = torch.from_numpy(np.random.default_rng(5).integers(-20000, 20000, (S, n), dtype=np.int16)).to(dev)
    fe16 = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat(sample_rate=sr, pcm_kind=1, pcm_divisor=32768.0), layout="btm")
    y16, _, fr16 = sed_amd.logmel_of_events(fe16, xi, sr, events, cap, per_file=True)
    assert torch.equal(fr16, frames)
    for e, (s, a, b) in enumerate(events):
        Te = int(fr16[e])
        assert not y16[e, Te:].any()
        if Te:
            assert torch.equal(y16[e, :Te], fe16(xi[s:s + 1, int(a * sr):int(b * sr)][:, :cap].contiguous())[0]), e
    # the zero-padded-batch reading stays the default, and differs where the padding rule matters
    y0, _ = sed_amd.logmel_of_events(fe, xd, sr, events, cap)
    assert not torch.equal(y0[0], y[0])
    with pytest.raises(TypeError):
        sed_amd.reshape_wav2img(y, 512, 8, frames=frames.cpu())
    none = sed_amd.logmel_of_events(fe, xd, sr, [], cap, per_file=True)                # no event at all
    assert none[0].shape == (0, 1, fe.num_frames(cap, dev), 64) and none[1].numel() == 0 and none[2].numel() == 0
    assert sed_amd.reshape_wav2img(none[0], 512, 8, frames=none[2]).shape == (0, 1, 512, 512)
    # segment tables already on the device: the frame counts are formed there
    off = torch.tensor([1000, n + 3], dtype=torch.int64, device=dev)
    ln = torch.tensor([5000, 600000], dtype=torch.int32, device=dev)
    yd, _, fd = fe.forward_segments(xd, off, ln, cap, own_edges=True)
    assert fd.tolist() == [1 + 5000 // hop, 1 + cap // hop]
    assert torch.equal(yd[0, :, :int(fd[0])], fe(xd[0:1, 1000:6000].contiguous())[0])
    assert torch.equal(yd[1], fe(xd[1:2, 3:3 + cap].contiguous())[0])
    # plans with the YAMNet tail have no segment mode
    with pytest.raises(ValueError):
        sed_amd.logmel_of_events(sed_amd.LogMelFrontend(sed_amd.FrontendConfig.yamnet()), xd, 16000, events[:2], 48000, per_file=True)


def test_two_stage_chain_as_main_runs_it():
    """main.py's two-stage flow end to end on resident audio, with stand-ins for the two networks: recordings at 44.1 kHz
    -> 16 kHz (librosa.load(sr=16000), yamnet_train.py:46) -> YAMNet patches -> [detector stand-in: patch energy] ->
    detect_onsets_offsets (yamnet_train.py:55-71) -> events handed over as files of their own (cut_audio.py:13-53,
    htsat/data.py:138-146) -> HTS-AT log-mel + reshape_wav2img (htsat.py:763-764, 729-744) -> [classifier stand-in:
    strongest mel band] -> labelled events -> calculate_metrics (main.py:29-48).  The same chain is walked on the CPU with
    the restatements, fed with the same probabilities: events, labels and metrics must come out the same."""
    import sed_amd
    from oracle import decode as od, metrics as om
    dev = torch.device("cuda", 0)
    sr, S, secs = 44100, 4, 20
    n = sr * secs
    tones = {"cough": 400.0, "dog_bark": 900.0, "gun_shot": 1800.0, "siren": 3500.0, "car_horn": 7000.0}
    classes = list(tones)
    rng = np.random.default_rng(12)
    x = (0.002 * rng.standard_normal((S, n))).astype(np.float32)
    names, gt = [f"rec_{s}.wav" for s in range(S)], {}
    t = np.arange(n) / sr
    for s, name in enumerate(names):
        gt[name] = []
        for k, start in enumerate((3.0 + 0.7 * s, 11.5 + 0.4 * s)):
            label = classes[(2 * s + k) % 5]
            dur = 2.4 + 0.48 * ((s + k) % 3)
            m = (t >= start) & (t < start + dur)
            x[s, m] += (0.3 * np.sin(2 * np.pi * tones[label] * t[m])).astype(np.float32)
            gt[name].append({"file": name, "event_onset": start, "event_offset": start + dur, "event_label": label})
    pcm = torch.from_numpy(x).to(dev)
    # stage 1: detector features and decode
    x16 = sed_amd.resample(pcm, orig_sr=sr, target_sr=16000)
    _, patches = sed_amd.yamnet_log_mel(x16)
    assert patches.shape[:2] == (S, 41)                                    # N_FRAMES of single_stage_yamnet_frame.py:20
    probs = torch.sigmoid(2.0 * (patches.amax(dim=3).mean(dim=2) - 1.5)).cpu().numpy().astype(np.float32)
    got = sed_amd.detect_onsets_offsets_batch(probs, 0.5, 0.48, True)
    events = []
    for s in range(S):
        on, off = od.detect_onsets_offsets(probs[s], 0.5, 0.48, True)
        assert np.array_equal(np.asarray(got[s][0]), on) and np.array_equal(np.asarray(got[s][1]), off)
        assert len(on) == len(off) == 2, (s, on, off)
        events += [(s, float(a), float(b)) for a, b in zip(on, off)]
    # stage 2: every event as a file of its own through the classifier's front-end
    cfg = sed_amd.FrontendConfig.htsat(sample_rate=sr)
    fe = sed_amd.LogMelFrontend(cfg)
    y, lengths, frames = sed_amd.logmel_of_events(fe, pcm, sr, events, 20 * sr, per_file=True)
    img = sed_amd.reshape_wav2img(y, 512, 8, frames=frames)
    centres = of.mel_to_hz_slaney(np.linspace(of.hz_to_mel_slaney(50.0), of.hz_to_mel_slaney(14000.0), 66))[1:-1]
    class_bin = np.array([int(np.argmin(np.abs(centres - f))) for f in tones.values()])

    def label_of(image):                                                   # (512, 512): 8 bands of 64 mel rows
        band = image.reshape(8, 64, 512).mean(axis=(0, 2))
        return classes[int(np.argmin(np.abs(class_bin - int(np.argmax(band)))))]

    pred = {name: [] for name in names}
    for e, (s, a, b) in enumerate(events):
        piece = x[s, int(a * sr):int(b * sr)]
        assert int(lengths[e]) == len(piece) and int(frames[e]) == 1 + len(piece) // cfg.hop
        gold = of.htsat_reshape_wav2img(of.htsat_logmel(piece[None], sr=sr, mode="gold").astype(np.float32), 512, 8)
        mine = img[e, 0].cpu().numpy()
        assert label_of(mine) == label_of(gold[0, 0]), e
        loud = gold[0, 0] > gold.max() - 40.0
        assert np.max(np.abs(mine - gold[0, 0])[loud]) < 2e-3, e
        pred[names[s]].append({"file": names[s], "event_onset": a, "event_offset": b, "event_label": label_of(mine)})
    ev, seg = sed_amd.calculate_metrics(pred, gt, time_resolution=1.0, t_collar=1.0, classes=classes, verbose=False)
    oe, osg = om.EventBasedMetrics(classes, t_collar=1.0), om.SegmentBasedMetrics(classes, 1.0)
    for name in names:
        oe.evaluate(gt[name], pred[name]); osg.evaluate(gt[name], pred[name])
    assert ev.results()["overall"] == oe.results_overall_metrics() and seg.results()["overall"] == osg.results_overall_metrics()
    assert ev.overall["Nref"] == 2 * S and ev.overall["Nsubs"] == 0 and seg.overall["S"] == 0      # every label right
    assert ev.overall["Ntp"] == 2 * S and seg.results()["overall"]["f_measure"]["f_measure"] > 0.7
