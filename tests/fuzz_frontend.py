#!/usr/bin/env python3
"""Wide random sweep of the HIP front-end against the CPU oracle (run by name on a GPU box; not collected by pytest).

    python3 tests/fuzz_frontend.py [cases] [seed]

Every case draws n_fft, any hop, a window (name or parameterised) with win_length <= n_fft, centring and padding mode,
magnitude or power, mel count and band edges, batch, length and sample type, and checks
  * Spectrogram against the fp64 transform (relative to the frame's largest bin),
  * LogmelFilterBank(Spectrogram) and the fused module against each other and against the fp64 log-mel at the bar of
    tests/test_gpu_frontend.py,
  * the segment mode (events read out of resident recordings) against cut + front-end (bit for bit),
  * Kaldi filterbanks with random frame length / shift / mel count against the oracle,
  * every eighth case: the functional entry points (extract_mbe, get_mel_spec_db, yamnet_log_mel) with arguments and
    lengths the reference does not use, magnitude spectrograms, the whole-tensor top_db, int16 input against the same
    samples as fp32 and a strided batch against a contiguous one (both bit for bit).
Prints one line per case and a summary; exits 1 on the first failure with the case's parameters.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sound-event-detection_amd"))

from oracle import frontend as of  # noqa: E402
import sed_amd  # noqa: E402
from scipy.signal import get_window  # noqa: E402

NEAR_DB = 60.0


def check(hip, gold, ref32, near_span, tol=1e-3, deep_factor=1.5, per_unit=10.0 / np.log(10.0)):
    """The bar of tests/test_gpu_frontend.py.  On bins deeper than `near_span` below the frame's peak the comparison with
    the fp32 formulation of the reference is a statistical one (both carry an error of about 1e-7 of the frame's peak,
    whatever the bin's own level); over hundreds of random cases one of them is bound to come out a few times worse than
    the other, so a deep bin also passes when its error is below 4e-7 of the frame's peak in the linear domain; the same
    goes for a bin just inside `near_span` (at -60 dB an amplitude error of 1e-7 of the peak is 1e-3 dB), and for a frame
    whose spectral peak lies outside the mel band (every mel bin is then deep below the largest spectral bin, whose
    rounding error it carries): there the fp32 formulation of the reference is the yardstick for near bins too.
    per_unit: log-domain units per neper of the linear quantity (10 / ln 10 for dB of power, 1 for ln)."""
    hip, gold, ref32 = np.asarray(hip, np.float64), np.asarray(gold, np.float64), np.asarray(ref32, np.float64)
    assert hip.shape == gold.shape, (hip.shape, gold.shape)
    peak = gold.max(axis=-1, keepdims=True)
    near = gold >= peak - near_span
    err, rerr = np.abs(hip - gold), np.abs(ref32 - gold)
    near_max = float(err[near].max())
    deep_max = float(err[~near].max()) if (~near).any() else 0.0
    deep_ref = float(rerr[~near].max()) if (~near).any() else 0.0
    near_ref = float(rerr[near].max())
    lin = np.abs(np.expm1((hip - gold) / per_unit)) * np.exp((gold - peak) / per_unit)       # error over the frame's peak
    assert near_max <= tol or near_max <= deep_factor * near_ref or float(lin[near].max()) <= 4e-7, \
        f"near-bin error {near_max:.3e} (fp32 formulation {near_ref:.3e}), {float(lin[near].max()):.2e} of the frame peak"
    if deep_max > max(tol, deep_factor * deep_ref):
        worst = float(lin[~near].max())
        assert worst <= 4e-7, f"deep-bin error {deep_max:.3e} (fp32 formulation {deep_ref:.3e}), {worst:.2e} of the frame peak"
    return near_max, deep_max


def one_case(rng, case, dev):
    n_fft = int(rng.choice([512, 1024, 2048]))
    sr = int(rng.choice([8000, 16000, 22050, 32000, 44100, 48000]))
    hop = int(rng.integers(16, n_fft + 1)) if rng.random() < 0.7 else 4 * int(rng.integers(8, n_fft // 4 + 1))
    win_length = n_fft if rng.random() < 0.5 else int(rng.integers(n_fft // 4, n_fft + 1))
    window = [("hann"), ("hamming"), ("blackman"), ("kaiser", float(rng.uniform(2, 12))), ("tukey", 0.3), ("bartlett")][int(rng.integers(6))]
    center = bool(rng.random() < 0.7)
    pad_mode = "reflect" if rng.random() < 0.6 else "constant"
    n_mels = int(rng.integers(1, 129))
    fmin = float(rng.choice([0.0, 20.0, 50.0, 125.0, 300.0]))
    fmax = float(rng.uniform(max(fmin + 500.0, 0.2 * sr), 0.5 * sr))
    lo = n_fft // 2 + 1 if center else n_fft
    n = int(rng.integers(lo, max(lo + 1, 3 * sr // 4)))
    if rng.random() < 0.15:
        n = lo + int(rng.integers(0, 3 * hop))                       # a handful of frames: every group touches an edge
    B = int(rng.integers(1, 5))
    desc = dict(case=case, n_fft=n_fft, sr=sr, hop=hop, win_length=win_length, window=window, center=center,
                pad_mode=pad_mode, n_mels=n_mels, fmin=fmin, fmax=round(fmax, 1), n=n, B=B)
    try:
        x = of.synth_batch(B, n, sr, first=1000 + case)
        xt = torch.from_numpy(x).to(dev)
        wfull = of.pad_center(get_window(window, win_length, fftbins=True).astype(np.float64), n_fft)
        sp = sed_amd.Spectrogram(n_fft=n_fft, hop_length=hop, win_length=win_length, window=window, center=center,
                                 pad_mode=pad_mode).to(dev)
        p = sp(xt)[:, 0].cpu().numpy().astype(np.float64)
        gold_p = of.power_spectrogram_gold(x, n_fft, hop, win_length, center=center, pad_mode=pad_mode, window_full=wfull)
        assert p.shape == gold_p.shape, (p.shape, gold_p.shape)
        rel = float(np.max(np.abs(p - gold_p) / np.maximum(gold_p.max(axis=-1, keepdims=True), 1e-30)))
        assert rel < 2e-6, f"spectrogram relative error {rel:.3e}"
        lm = sed_amd.LogmelFilterBank(sr=sr, n_fft=n_fft, n_mels=n_mels, fmin=fmin, fmax=fmax, top_db=None).to(dev)
        fe = sed_amd.LogMelFrontend.from_torchlibrosa(sp, lm)
        fused = fe(xt)
        d = float(torch.max(torch.abs(fused - lm(sp(xt)))).item())
        assert d < 2e-4, f"fused against unfused {d:.3e}"
        melW = of.librosa_mel(sr, n_fft, n_mels, fmin, fmax).T
        gold = of.power_to_db(gold_p @ melW.astype(np.float64))[:, None]
        ref32 = of.power_to_db(of.power_spectrogram_ref32(x, n_fft, hop, win_length, center=center, pad_mode=pad_mode,
                                                          window_full=wfull) @ melW)[:, None]
        try:
            near, deep = check(fused.cpu().numpy(), gold, ref32, NEAR_DB)
        except AssertionError:
            g, h = gold[:, 0], fused.cpu().numpy()[:, 0].astype(np.float64)
            nearm = g >= g.max(-1, keepdims=True) - NEAR_DB
            i = np.unravel_index(np.argmax(np.where(nearm, np.abs(h - g), 0)), g.shape)
            wcol = melW[:, i[2]]
            nz = np.nonzero(wcol)[0]
            print(f"     fmax {fmax!r}; worst near bin {tuple(int(v) for v in i)}: gold {g[i]:.2f} dB, hip {h[i]:.5f}, fp32 formulation "
                  f"{ref32[i[0], 0, i[1], i[2]]:.5f}, frame mel peak {g[i[0], i[1]].max():.1f} dB, frame spectral peak "
                  f"{10 * np.log10(gold_p[i[0], i[1]].max()):.1f} dB; filter bins {nz.tolist()} weights {wcol[nz].tolist()} "
                  f"powers {gold_p[i[0], i[1], nz].tolist()} hip powers {p[i[0], i[1], nz].tolist()}", flush=True)
            raise
        # segment mode: events out of the resident batch, against cut + front-end
        extra = ""
        if n > 3 * n_fft:
            fe_btm = sed_amd.LogMelFrontend.from_torchlibrosa(sp, lm)
            events = []
            for _ in range(int(rng.integers(1, 7))):
                a = float(rng.uniform(0, (n - n_fft) / sr))
                events.append((int(rng.integers(B)), a, a + float(rng.uniform(n_fft / sr, (n / sr - a)))))
            cap = int(rng.integers(n_fft, n + 1))
            clips, lengths = sed_amd.cut_events(xt, sr, events, cap)[:2]
            want = fe_btm(clips)
            got, glen = sed_amd.logmel_of_events(fe_btm, xt, sr, events, cap)
            assert torch.equal(glen.cpu(), lengths.cpu()), "segment lengths"
            assert torch.equal(got, want), f"segment mode differs from cut + front-end (max {float(torch.max(torch.abs(got - want))):.3e})"
            # every event framed as a file of its own: the front-end on the event's samples alone, bit for bit
            events.append((0, 0.0, (lo - 1) / sr))                   # one sample too short for a frame
            events.append((B - 1, 0.1, 0.1 + (lo + 0.5) / sr))       # the shortest that has one
            own, olen, frames = sed_amd.logmel_of_events(fe_btm, xt, sr, events, cap, per_file=True)
            for e, (s_, a, b) in enumerate(events):
                piece = x[s_, int(a * sr):int(b * sr)][:cap]
                least = (n_fft // 2 + 1 if pad_mode == "reflect" else 1) if center else n_fft   # (constant padding takes any length)
                Te = (1 + len(piece) // hop if center else 1 + (len(piece) - n_fft) // hop) if len(piece) >= least else 0
                assert int(frames[e]) == Te, f"own frames of event {e}: {int(frames[e])} for {Te}"
                assert not own[e, :, Te:].any(), f"rows behind event {e}"
                if Te:
                    alone = fe_btm(torch.from_numpy(piece[None].copy()).to(dev))
                    assert torch.equal(own[e, :, :Te], alone[0]), f"event {e} of {len(piece)} samples differs from the event alone"
            extra = f" seg{len(events)}"
        print(f"ok   {desc} rel {rel:.1e} near {near:.1e} deep {deep:.1e}{extra}", flush=True)
        return True
    except Exception as e:  # noqa: BLE001
        print(f"FAIL {desc}: {type(e).__name__}: {e}", flush=True)
        return False


def functions_case(rng, case, dev):
    """The functional entry points with arguments the reference does not use: extract_mbe and get_mel_spec_db with other
    n_fft / hop / mel counts and lengths, YAMNet features of arbitrary lengths (tail padding), magnitude spectrograms, the
    whole-tensor top_db of LogmelFilterBank, int16 input against the same samples as fp32."""
    n_fft = int(rng.choice([512, 1024, 2048]))
    sr = int(rng.choice([16000, 22050, 32000, 44100]))
    hop = int(rng.integers(32, n_fft + 1))
    n_mels = int(rng.integers(4, 129))
    n = int(rng.integers(n_fft, sr))
    B = int(rng.integers(1, 4))
    desc = dict(case=case, kind="functions", n_fft=n_fft, sr=sr, hop=hop, n_mels=n_mels, n=n, B=B)
    try:
        y = of.synth_batch(B, n, sr, first=5000 + case)
        ln_span = NEAR_DB / 20.0 * np.log(10.0)
        mbe = sed_amd.extract_mbe(y, sr, n_fft, n_mels, hop)
        check(np.swapaxes(mbe, -1, -2), np.swapaxes(of.crnn_extract_mbe(y, sr, n_fft, n_mels, hop, mode="gold"), -1, -2),
              np.swapaxes(of.crnn_extract_mbe(y, sr, n_fft, n_mels, hop, mode="ref32"), -1, -2), ln_span, per_unit=2.0)
        top_db = float(rng.choice([80.0, 60.0, 120.0]))
        db = sed_amd.get_mel_spec_db(y, sr, n_fft, hop, n_mels, top_db)
        want = of.librosa_db_mel(y, sr, n_fft, hop, n_mels, mode="gold", top_db=top_db)
        assert db.shape == want.shape and db.max() == 0.0 and db.min() >= -top_db
        # (values at the floor may sit on either side of it by the error of the deep bins)
        d = np.abs(db - want)[want > -top_db + 1.0]
        near = want >= want.max(axis=(-1, -2), keepdims=True) - NEAR_DB
        assert np.max(np.abs(db - want)[near]) < 1e-3, f"dB-mel near error {np.max(np.abs(db - want)[near]):.3e}"
        # YAMNet: any length, the tail padded to whole patches
        n16 = int(rng.integers(400, 60000))
        w = of.synth_batch(B, n16, 16000, first=6000 + case)
        lm, patches = sed_amd.yamnet_log_mel(torch.from_numpy(w))
        g_lm, g_p = of.yamnet_logmel(w, mode="gold")
        r_lm, _ = of.yamnet_logmel(w, mode="ref32")
        assert tuple(patches.shape) == g_p.shape, (tuple(patches.shape), g_p.shape)
        check(lm.cpu().numpy(), g_lm, r_lm, ln_span, per_unit=2.0)
        # magnitude spectrogram and whole-tensor top_db of the unfused modules; int16 against fp32
        xt = torch.from_numpy(y).to(dev)
        mag = sed_amd.Spectrogram(n_fft=n_fft, hop_length=hop, power=1.0).to(dev)(xt)[:, 0].cpu().numpy().astype(np.float64)
        gold_p = of.power_spectrogram_gold(y, n_fft, hop)
        rel = float(np.max(np.abs(mag ** 2 - gold_p) / np.maximum(gold_p.max(axis=-1, keepdims=True), 1e-30)))
        assert rel < 4e-6, f"magnitude spectrogram {rel:.3e}"
        sp = sed_amd.Spectrogram(n_fft=n_fft, hop_length=hop).to(dev)
        lm_top = sed_amd.LogmelFilterBank(sr=sr, n_fft=n_fft, n_mels=n_mels, fmin=20, fmax=sr // 2, top_db=top_db).to(dev)
        got_top = lm_top(sp(xt)).cpu().numpy()[:, 0]
        melW = of.librosa_mel(sr, n_fft, n_mels, 20, sr // 2).T.astype(np.float64)
        want_top = of.power_to_db(gold_p @ melW, top_db=top_db)
        live = want_top > want_top.max() - top_db + 1.0
        assert got_top.min() >= got_top.max() - top_db - 1e-4
        near = want_top >= want_top.max(axis=-1, keepdims=True) - NEAR_DB
        assert np.max(np.abs(got_top - want_top)[near & live]) < 1e-3, "LogmelFilterBank top_db"
        s16 = of.f32_to_pcm16(y)
        c16 = sed_amd.FrontendConfig.htsat(sample_rate=sr, window_size=n_fft, hop_size=hop, mel_bins=n_mels, pcm_kind=1, pcm_divisor=32767.0)
        c32 = sed_amd.FrontendConfig.htsat(sample_rate=sr, window_size=n_fft, hop_size=hop, mel_bins=n_mels)
        a = sed_amd.LogMelFrontend(c16)(torch.from_numpy(s16).to(dev))
        b = sed_amd.LogMelFrontend(c32)(torch.from_numpy(of.pcm16_to_f32(s16)).to(dev))
        assert torch.equal(a, b), "int16 input differs from the same samples as fp32"
        # a strided batch (every second row of a larger buffer, rows longer than the clip)
        big = torch.zeros(2 * B, n + 37, device=dev)
        big[::2, :n] = xt
        assert torch.equal(sed_amd.LogMelFrontend(c32)(big[::2, :n]), sed_amd.LogMelFrontend(c32)(xt)), "strided batch"
        print(f"ok   {desc}", flush=True)
        return True
    except Exception as e:  # noqa: BLE001
        import traceback
        print(f"FAIL {desc}: {type(e).__name__}: {e}\n{traceback.format_exc(limit=2)}", flush=True)
        return False


def kaldi_case(rng, case):
    sr = int(rng.choice([16000, 22050, 32000, 44100]))                  # 8 kHz frames are 256-point transforms: not built
    frame_length = float(rng.choice([20.0, 25.0, 32.0]))
    frame_shift = float(rng.choice([5.0, 8.0, 10.0, 12.5]))
    mel = int(rng.integers(8, 129))
    n = int(rng.integers(int(sr * frame_length * 0.001) + 1, sr))
    B = int(rng.integers(1, 4))
    desc = dict(case=case, kaldi=True, sr=sr, frame_length=frame_length, frame_shift=frame_shift, mel=mel, n=n, B=B)
    try:
        x = of.synth_batch(B, n, sr, first=3000 + case) + 0.01
        got = sed_amd.kaldi_fbank(torch.from_numpy(x), sample_frequency=sr, num_mel_bins=mel, frame_length=frame_length,
                                  frame_shift=frame_shift, htk_compat=True)
        got = got.cpu().numpy().reshape(B, -1, mel)
        gold = of.kaldi_fbank(x, sr, mel, frame_length, frame_shift, mode="gold")
        ref32 = of.kaldi_fbank(x, sr, mel, frame_length, frame_shift, mode="ref32")
        near, deep = check(got, gold, ref32, NEAR_DB / 10.0 * np.log(10.0), deep_factor=2.5, per_unit=1.0)
        # segment mode of the Kaldi-framed kernels: events out of the resident batch, as a zero-padded batch (against cut +
        # front-end) and as files of their own (against the front-end on each event alone), both bit for bit
        dev = torch.device("cuda", 0)
        xt = torch.from_numpy(x).to(dev)
        fe = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.kaldi_fbank(sr, mel, frame_length, frame_shift), layout="btm")
        win, hop = fe.cfg.win_length, fe.cfg.hop
        events = []
        for _ in range(int(rng.integers(1, 6))):
            a = float(rng.uniform(0, (n - win) / sr))
            events.append((int(rng.integers(B)), a, a + float(rng.uniform(win / sr, n / sr - a))))
        events.append((0, 0.0, (win - 1) / sr))                          # one sample short of a frame
        events.append((B - 1, 0.001, 0.001 + (win + 0.5) / sr))         # exactly one frame
        cap = int(rng.integers(win, n + 1))
        clips, lengths = sed_amd.cut_events(xt, sr, events, cap)[:2]
        seg, glen = sed_amd.logmel_of_events(fe, xt, sr, events, cap)
        assert torch.equal(glen.cpu(), lengths.cpu()), "segment lengths"
        assert torch.equal(seg, fe(clips)), f"segment mode differs from cut + front-end (max {float(torch.max(torch.abs(seg - fe(clips)))):.3e})"
        own, _, frames = sed_amd.logmel_of_events(fe, xt, sr, events, cap, per_file=True)
        for e, (s_, a, b) in enumerate(events):
            piece = x[s_, int(a * sr):int(b * sr)][:cap]
            Te = 1 + (len(piece) - win) // hop if len(piece) >= win else 0
            assert int(frames[e]) == Te, f"own frames of event {e}: {int(frames[e])} for {Te}"
            assert not own[e, Te:].any(), f"rows behind event {e}"
            if Te:
                alone = fe(torch.from_numpy(piece[None].copy()).to(dev))
                assert torch.equal(own[e, :Te], alone[0]), f"event {e} of {len(piece)} samples differs from the event alone"
        print(f"ok   {desc} near {near:.1e} deep {deep:.1e} seg{len(events)}", flush=True)
        return True
    except Exception as e:  # noqa: BLE001
        print(f"FAIL {desc}: {type(e).__name__}: {e}", flush=True)
        return False


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(seed)
    bad = 0
    only_kaldi = len(sys.argv) > 3 and sys.argv[3] == "kaldi"                 # python3 tests/fuzz_frontend.py 100 7 kaldi
    for case in range(cases):
        ok = kaldi_case(rng, case) if case % 8 == 7 or only_kaldi else functions_case(rng, case, dev) if case % 8 == 3 else one_case(rng, case, dev)
        bad += not ok
    print(f"{cases - bad} of {cases} cases passed")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
