"""GPU parity: the HIP decoders (through the C ABI) against the golden fixtures generated from the reference
and against the CPU oracle on seeded inputs.  Everything here is bit exact."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import decode as od

pytestmark = pytest.mark.gpu

CLASSES = ["cough", "dog_bark", "gun_shot", "siren", "car_horn"]
NAMES6 = CLASSES + ["non_event"]


def _events(pack, prefix, files, labels):
    out = {f: [] for f in files}
    for fi, li, on, off in zip(pack[prefix + "_fi"], pack[prefix + "_li"], pack[prefix + "_on"], pack[prefix + "_off"]):
        f = files[int(fi)]
        out[f].append({"file": f, "event_onset": float(on), "event_offset": float(off), "event_label": labels[int(li)]})
    return out


@pytest.fixture(scope="module")
def kat(golden_dir):
    pack = np.load(os.path.join(golden_dir, "crnn_kat.npz"))
    return pack, [str(f) for f in pack["files"]], [str(l) for l in pack["labels"]]


@pytest.mark.parametrize("thr", ["0.3", "0.6", "1.0"])
def test_merge_kat(kat, thr):
    import sed_amd
    pack, files, labels = kat
    raw = _events(pack, "raw", files, labels)
    want = _events(pack, "m" + thr, files, labels)
    got = sed_amd.merge_close_events_batch(raw, float(thr))
    assert got == want and list(got) == files
    for f in files[:25] + files[-5:]:
        assert sed_amd.merge_close_events(raw[f], float(thr)) == want[f]
    assert sed_amd.merge_close_events([], 0.5) == []


def test_merge_random_overlapping_and_long_groups():
    import sed_amd
    rng = np.random.default_rng(3)
    data = {}
    for i in range(200):
        n = int(rng.integers(0, 400 if i % 20 == 0 else 40))
        on = np.round(rng.uniform(0, 60, n), 3)
        dur = np.round(rng.exponential(0.4, n) - (0.2 if i % 7 == 0 else 0.0), 3)   # some negative durations
        data[f"f{i}"] = [{"file": f"f{i}", "event_onset": float(a), "event_offset": float(a + d),
                          "event_label": str(rng.integers(0, 3)), "extra": k}
                         for k, (a, d) in enumerate(zip(on, dur))]
    for thr in (0.0, 0.3, 0.5, 2.0):
        got = sed_amd.merge_close_events_batch(data, thr)
        want = {k: od.merge_close_events(v, thr) for k, v in data.items()}
        assert got == want, thr
        assert all("extra" in e for v in got.values() for e in v)        # a merged event is a copy of its first event


def test_decode_predictions_roundtrip(kat, golden_dir):
    import sed_amd
    pack, files, labels = kat
    a = np.load(os.path.join(golden_dir, "crnn_activity.npz"))
    T = int(a["T"])
    act = np.unpackbits(a["act"], axis=1)[:, :T, :].astype(int)
    idx_to_label = dict(enumerate(labels))
    assert sed_amd.decode_predictions(act, files, idx_to_label, 43) == _events(pack, "raw", files, labels)
    cur = sed_amd.decode_predictions(act, files, idx_to_label, 43.06640625)
    assert np.array_equal([e["event_onset"] for f in files for e in cur[f]], a["fps_cur_on"])
    assert np.array_equal([e["event_offset"] for f in files for e in cur[f]], a["fps_cur_off"])
    # more predictions than names, numpy-str names, non-binary activity
    few = sed_amd.decode_predictions(act[:3], np.asarray(files[:2]), idx_to_label, 43)
    assert list(few) == files[:2]
    weird = np.array([[[0], [2], [3], [3], [2], [0], [1], [0]]])
    assert sed_amd.decode_predictions(weird, ["w"], {0: "x"}, 10) == od.decode_predictions(weird, ["w"], {0: "x"}, 10)


def test_full_crnn_chain_on_synthetic_probabilities(kat):
    """D2 + D3 + D4 at BASELINE.json configs[2] size (500 x 862 x 5) against the oracle."""
    import sed_amd
    from scipy.ndimage import gaussian_filter1d
    rng = np.random.default_rng(21)
    probs = 1.0 / (1.0 + np.exp(-4.0 * gaussian_filter1d(rng.standard_normal((500, 862, 5)), 6.0, axis=1) / 0.25))
    probs = probs.astype(np.float32)
    files = [f"clip_{i:04d}.wav" for i in range(500)]
    idx_to_label = {0: "car_horn", 1: "cough", 2: "dog_bark", 3: "siren", 4: "gun_shot"}
    fps = 44100 / 1024
    got = sed_amd.decode_probabilities(probs, files, idx_to_label, fps, 0.5)
    want = od.decode_predictions(od.smooth_and_threshold(probs, 0.5), files, idx_to_label, fps)
    assert got == want
    n_ev = sum(len(v) for v in want.values())
    assert n_ev > 2000
    for thr in (0.3, 0.6, 1.0):
        assert sed_amd.merge_close_events_batch(got, thr) == {k: od.merge_close_events(v, thr) for k, v in want.items()}
    # several gaps in one call (events read and grouped once)
    three = sed_amd.merge_close_events_batch(got, [0.3, 0.6, 1.0])
    assert three == [{k: od.merge_close_events(v, thr) for k, v in want.items()} for thr in (0.3, 0.6, 1.0)]
    assert sed_amd.merge_close_events_batch({}, [0.3, 0.6]) == [{}, {}]
    # label dicts that are not "every class in index order" take the general formatting path: a subset in another order,
    # fewer names than predictions, numpy names
    sub = {3: "siren", 0: "car_horn"}
    assert sed_amd.decode_probabilities(probs[:40], files[:40], sub, fps, 0.5) == \
        od.decode_predictions(od.smooth_and_threshold(probs[:40], 0.5), files[:40], sub, fps)
    assert sed_amd.decode_probabilities(probs[:9], np.asarray(files[:7]), idx_to_label, fps, 0.5) == \
        od.decode_predictions(od.smooth_and_threshold(probs[:9], 0.5), np.asarray(files[:7]), idx_to_label, fps)
    silent = np.zeros((3, 50, 5), np.float32)
    assert sed_amd.decode_probabilities(silent, files[:3], idx_to_label, fps, 0.5) == {f: [] for f in files[:3]}


def test_full_size_decode_properties():
    """Properties that hold whatever the size, on 500 x 862 x 5 probabilities: merging is idempotent, a wider gap never
    yields more events, merged events cover the raw ones, and the device records agree with the dictionary chain."""
    import sed_amd
    rng = np.random.default_rng(99)
    S, T, C, fps = 500, 862, 5, 44100 / 1024
    z = rng.standard_normal((S * C, T + 60)).astype(np.float32)
    z = np.stack([np.convolve(r, np.ones(61, np.float32) / 8.0, mode="valid") for r in z])
    probs = np.ascontiguousarray((1.0 / (1.0 + np.exp(-2.0 * z))).astype(np.float32).reshape(S, C, T).transpose(0, 2, 1))
    files = [f"c{i}.wav" for i in range(S)]
    labels = {0: "car_horn", 1: "cough", 2: "dog_bark", 3: "siren", 4: "gun_shot"}
    raw = sed_amd.decode_probabilities(probs, files, labels, fps, 0.5)
    gaps = [0.0, 0.3, 0.6, 1.0, 5.0]
    merged = sed_amd.merge_close_events_batch(raw, gaps)
    counts = [sum(len(v) for v in m.values()) for m in merged]
    assert counts[0] <= sum(len(v) for v in raw.values()) and all(a >= b for a, b in zip(counts, counts[1:])) and counts[-1] >= 1
    for g, m in zip(gaps, merged):
        assert sed_amd.merge_close_events_batch(m, g) == m                      # idempotent
        for f in files[::37]:
            for ev in raw[f]:                                                    # every raw event lies inside a merged one
                assert any(k["event_label"] == ev["event_label"] and k["event_onset"] <= ev["event_onset"]
                           and ev["event_offset"] <= k["event_offset"] for k in m[f]), (g, f, ev)
    # the device-only path: records of decode_and_exchange name the same events
    dev = torch.device("cuda", 0)
    table_host = [round(np.int64(k) * (1.0 / fps), 3) for k in range(T + 2)]
    table = torch.tensor(table_host, dtype=torch.float64, device=dev)
    class_of = {v: k for k, v in labels.items()}
    rec = sed_amd.sharding.decode_and_exchange(torch.from_numpy(probs).to(dev), table, 0.6, cap=128, smoothing="gauss").cpu().numpy()
    for i, f in enumerate(files):
        got = sorted((int(c), table_host[int(a)], table_host[int(b)]) for _, c, a, b in rec[i, 1:1 + rec[i, 0, 0]])
        want = sorted((class_of[e["event_label"]], e["event_onset"], e["event_offset"]) for e in merged[2][f])
        assert got == want and rec[i, 0, 1] == 0, f


def test_decode_predictions_rounding_at_other_frame_rates(golden_dir):
    """The reference's own outputs at eight frame rates (numpy rounding of np.int64 * float), every frame a boundary."""
    import sed_amd
    z = np.load(os.path.join(golden_dir, "d3_rounding.npz"))
    T = int(z["T"])
    act = np.zeros((1, T, 1), dtype=int)
    act[0, 0::2, 0] = 1
    for i, fps in enumerate(z["fps"]):
        ev = sed_amd.decode_predictions(act, ["f.wav"], {0: "e"}, float(fps))["f.wav"]
        assert [e["event_onset"] for e in ev] == z[f"on_{i}"].tolist(), fps
        assert [e["event_offset"] for e in ev] == z[f"off_{i}"].tolist(), fps


def test_gaussian_threshold_bits_near_the_threshold(golden_dir):
    """Traces built to sit within a few ulp of 0.5 after smoothing: the activity must match scipy's bits."""
    import sed_amd
    from sed_amd import decode as D
    z = np.load(os.path.join(golden_dir, "d2_vectors.npz"))
    x, y = z["x"], z["y"]
    dev = torch.device("cuda", 0)
    for thr in (0.5, float(np.median(y)), float(y[0, 100, 2]), float(np.nextafter(y[3, 400, 1], np.float32(2)))):
        runs, counts = D.threshold_runs(torch.from_numpy(x).to(dev), thr, "gauss")
        runs, counts = runs.cpu().numpy(), counts.cpu().numpy()
        act = (y > np.float32(thr)).astype(int)
        for b in range(x.shape[0]):
            for c in range(x.shape[2]):
                want = od.contiguous_regions(act[b, :, c])
                assert [tuple(r) for r in runs[b, c, :counts[b, c]]] == want, (thr, b, c)


def test_detect_onsets_offsets_vectors(golden_dir):
    import sed_amd
    exp = json.load(open(os.path.join(golden_dir, "d1_vectors.json")))["cases"]
    inp = np.load(os.path.join(golden_dir, "decode_inputs.npz"))
    for i, rec in enumerate(exp):
        p = inp[f"d1_{i}"]
        for mf in (1, 0):
            on, off = sed_amd.detect_onsets_offsets(p, 0.5, 0.48, bool(mf))
            assert on.dtype == np.float64 or on.size == 0
            assert [float(v).hex() for v in np.atleast_1d(on)] == rec[f"on_{mf}"], (i, mf)
            assert [float(v).hex() for v in np.atleast_1d(off)] == rec[f"off_{mf}"], (i, mf)


def test_detect_batch_4096_streams():
    """BASELINE.json configs[3] size: 4096 concurrent 10 s detector streams (20 patches each)."""
    import sed_amd
    rng = np.random.default_rng(31)
    p = rng.random((4096, 20), dtype=np.float32)
    p[::5] = (p[::5] > 0.5).astype(np.float32)
    got = sed_amd.detect_onsets_offsets_batch(p, 0.5, 0.48, True)
    for b in range(4096):
        on, off = od.detect_onsets_offsets(p[b], 0.5, 0.48, True)
        assert np.array_equal(got[b][0], on) and np.array_equal(got[b][1], off), b


def test_htsat_events_vectors(golden_dir):
    import sed_amd

    class Cfg:
        clip_duration = 20.0

    exp = json.load(open(os.path.join(golden_dir, "d5_vectors.json")))
    inp = np.load(os.path.join(golden_dir, "decode_inputs.npz"))
    for i, rec in enumerate(exp):
        batch = {"file": rec["files"], "pred_map": torch.from_numpy(inp[f"d5_{i}"])}
        assert sed_amd.get_events_from_batch(batch, Cfg) == rec["expected"], i
    batches = [{"file": exp[i]["files"], "pred_map": torch.from_numpy(inp[f"d5_{i}"])} for i in range(len(exp))]
    merged = {}
    for rec in exp:
        merged = merged | rec["expected"]
    assert sed_amd.get_events(batches, Cfg) == merged


def test_yamnet_frame_events_vectors(golden_dir):
    import sed_amd
    exp = json.load(open(os.path.join(golden_dir, "d6_vectors.json")))
    inp = np.load(os.path.join(golden_dir, "decode_inputs.npz"))
    for i, rec in enumerate(exp):
        for mf in (1, 0):
            got = sed_amd.predict_events(inp[f"d6_{i}"], f"/data/x/clip_{i}.wav", CLASSES, 0.48, bool(mf))
            assert got == rec[f"expected_{mf}"], (i, mf)


def test_htsat_clip_label_vectors(golden_dir):
    import sed_amd
    exp = json.load(open(os.path.join(golden_dir, "d7_vectors.json")))
    inp = np.load(os.path.join(golden_dir, "decode_inputs.npz"))
    for i, rec in enumerate(exp):
        assert sed_amd.get_classification_label(inp[f"d7_{i}"], NAMES6) == rec["label"], i


def test_argmax_runs_random_against_oracle():
    import sed_amd
    rng = np.random.default_rng(41)
    for T in (1, 2, 63, 64, 65, 128, 4096):
        pm = rng.random((3, T, 6)).astype(np.float32)
        lab = np.repeat(rng.integers(0, 6, T // 5 + 1), 5)[:T]
        pm[:, np.arange(T), lab] += 0.6

        class Cfg:
            clip_duration = 20.0
        for b in range(3):
            names = [f"c{b}.wav"]
            got = sed_amd.get_events_from_batch({"file": names, "pred_map": pm[b:b + 1]}, Cfg)
            assert got == od.htsat_events_from_batch(names, pm[b:b + 1], NAMES6, 5, 20.0), (T, b)
            for mf in (True, False):
                assert sed_amd.predict_events(pm[b], names[0], CLASSES, 0.48, mf) == \
                    od.yamnet_frame_events(pm[b], names[0], CLASSES, 0.48, mf), (T, b, mf)
            assert sed_amd.get_classification_label(pm[b], NAMES6) == NAMES6[od.htsat_clip_label(pm[b])]


def test_frame_run_merge_matches_time_merge():
    """sed_merge_runs (frame indices + host time table) gives the same events as merging the formatted times."""
    import sed_amd
    from sed_amd import decode as D
    rng = np.random.default_rng(51)
    from scipy.ndimage import gaussian_filter1d
    probs = (1.0 / (1.0 + np.exp(-16.0 * gaussian_filter1d(rng.standard_normal((64, 862, 5)), 5.0, axis=1)))).astype(np.float32)
    dev = torch.device("cuda", 0)
    runs, counts = D.threshold_runs(torch.from_numpy(probs).to(dev), 0.5, "gauss")
    fps = 44100 / 1024
    table = torch.tensor([round(k * (1.0 / fps), 3) for k in range(863)], dtype=torch.float64, device=dev)
    labels = {0: "car_horn", 1: "cough", 2: "dog_bark", 3: "siren", 4: "gun_shot"}
    files = [f"f{i}" for i in range(64)]
    events = sed_amd.decode_probabilities(probs, files, labels, fps, 0.5)
    for gap in (0.3, 0.6, 1.0):
        mr, mc = D.merge_runs(runs, counts, table, gap)
        mr, mc = mr.cpu().numpy(), mc.cpu().numpy()
        tt = table.cpu().numpy()
        for i, f in enumerate(files):
            got = sorted(((float(tt[s]), float(tt[e]), labels[c]) for c in range(5) for s, e in mr[i, c, :mc[i, c]]),
                         key=lambda t: t[0])
            want = [(e["event_onset"], e["event_offset"], e["event_label"]) for e in od.merge_close_events(events[f], gap)]
            assert sorted(got) == sorted(want), (gap, f)


def test_fused_threshold_merge_equals_two_launches():
    """sed_threshold_merge_runs against sed_threshold_runs + sed_merge_runs: identical counts and frame pairs."""
    from sed_amd import decode as D
    rng = np.random.default_rng(77)
    dev = torch.device("cuda", 0)
    for (B, T, C), smooth, gap, step in (((256, 20, 1), "median3", 0.6, 0.48), ((33, 41, 6), "median3", 0.3, 0.48),
                                         ((16, 500, 5), "gauss", 1.0, 1024 / 44100), ((5, 1, 2), None, 0.6, 0.48),
                                         ((7, 64, 3), None, 0.6, 0.1), ((3, 129, 4), "median3", 0.0, 0.25),
                                         ((6, 862, 5), "gauss", 0.3, 1024 / 44100), ((2, 4096, 6), "gauss", 0.05, 20.0 / 4096),
                                         ((3, 8190, 1), None, 0.01, 0.01), ((2, 511, 2), None, 0.0, 1.0)):
        probs = torch.from_numpy(rng.random((B, T, C)).astype(np.float32)).to(dev)
        probs[0] = 1.0                                                  # one fully active stream
        probs[-1] = 0.0                                                 # one silent stream
        table = torch.tensor([round(k * step, 3) for k in range(T + 2)], dtype=torch.float64, device=dev)
        runs, counts = D.threshold_runs(probs, 0.5, smooth)
        mruns, mcounts = D.merge_runs(runs, counts, table, gap)
        packed = D.threshold_merge_runs(probs, 0.5, table, gap, smooth).cpu().numpy()
        mruns, mcounts = mruns.cpu().numpy(), mcounts.cpu().numpy()
        assert packed.shape == (B, C, 1 + 2 * (T // 2 + 1))
        assert np.array_equal(packed[..., 0], mcounts)
        for b in range(B):
            for c in range(C):
                n = mcounts[b, c]
                assert np.array_equal(packed[b, c, 1:1 + 2 * n].reshape(n, 2), mruns[b, c, :n]), (B, T, C, b, c)
    with pytest.raises(ValueError):
        D.threshold_merge_runs(torch.zeros((1, 8192, 1), device=dev), 0.5, table, 0.6)  # beyond the LDS capacity
    # the kernels read the time table as fp64 on the device of the data: anything else is refused before the launch
    p1 = torch.zeros((1, 8, 1), device=dev)
    good = torch.zeros(10, dtype=torch.float64, device=dev)
    for bad, exc in ((good.float(), TypeError), (good.cpu(), ValueError), (torch.zeros(20, dtype=torch.float64, device=dev)[::2], ValueError),
                     ([0.0] * 10, TypeError)):
        with pytest.raises(exc):
            D.threshold_merge_runs(p1, 0.5, bad, 0.6)
        with pytest.raises(exc):
            D.threshold_merge_records(p1, 0.5, bad, 0.6)
        with pytest.raises(exc):
            D.merge_runs(*D.threshold_runs(p1, 0.5), bad, 0.6)
    with pytest.raises(ValueError):
        D.threshold_merge_records(p1, 0.5, good, 0.6, smoothing="gaussian")


def test_framewise_interpolate_and_pad():
    """interpolate / pad_framewise_output (external/hts_audio_transformer/utils.py:158-192): exact copies."""
    import sed_amd
    from oracle import frontend as of
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(12)
    for (B, T, C, ratio, frames) in ((1, 128, 6, 32, 4096), (3, 7, 6, 32, 250), (2, 31, 5, 1, 40), (4, 1, 3, 8, 8),
                                     (1, 33, 527, 32, 1056), (2, 50, 300, 4, 230), (1, 3, 9000, 2, 9),
                                     (5, 100, 1, 32, 3300), (2, 9, 64, 32, 288), (3, 40, 17, 5, 333)):
        p = rng.random((B, T, C)).astype(np.float32)
        pd = torch.from_numpy(p).to(dev)
        up = sed_amd.interpolate(pd, ratio)
        want_up = of.htsat_interpolate(p, ratio)
        assert up.shape == want_up.shape and np.array_equal(up.cpu().numpy(), want_up)
        want = of.htsat_pad_framewise_output(want_up, frames)
        assert np.array_equal(sed_amd.pad_framewise_output(up, frames).cpu().numpy(), want)
        assert np.array_equal(sed_amd.interpolate_and_pad(pd, ratio, frames).cpu().numpy(), want)
    with pytest.raises(RuntimeError):
        sed_amd.pad_framewise_output(up, up.shape[1] - 1)
    with pytest.raises(RuntimeError):
        sed_amd.interpolate(torch.zeros(1, 4, 6), 32)                                  # host tensor: no CPU path
    # the HTS-AT framewise output feeds the combined decoder: expansion then smoothing + argmax runs
    one_hot = np.zeros((1, 128, 6), np.float32)
    lab = np.repeat(np.array([5, 3, 5, 1, 5, 0, 5, 2]), 16)
    one_hot[0, np.arange(128), lab] = 1.0
    full = sed_amd.interpolate(torch.from_numpy(one_hot).to(dev), 32)
    assert full.shape == (1, 4096, 6) and torch.equal(full[0, ::32], torch.from_numpy(one_hot[0]).to(dev))


def test_decoded_events_scored_like_main():
    """The chain of main.py for the single-stage CRNN: probabilities -> events (crnn.py:418-459) -> calculate_metrics
    (main.py:29-48).  Probabilities that are 0.9 inside the true events and 0.1 elsewhere decode to the truth quantised to
    frames, which every segment of 1 s and every event within the 0.25 s collar must score as correct; one clip with its
    label swapped gives exactly one substitution."""
    import sed_amd
    from oracle import metrics as om
    fps, T, N = 43.06640625, 862, 12
    labels = {0: "car_horn", 1: "cough", 2: "dog_bark", 3: "siren", 4: "gun_shot"}           # crnn/utils.py:77-89
    rng = np.random.default_rng(3)
    probs = np.full((N, T, 5), 0.1, np.float32)
    names, gt = [f"clip_{i:02d}.wav" for i in range(N)], {}
    for i, name in enumerate(names):
        c = int(rng.integers(5))
        a = int(rng.integers(40, 400))
        b = a + int(rng.integers(60, 300))
        probs[i, a:b, c] = 0.9
        gt[name] = [{"file": name, "event_onset": a / fps, "event_offset": b / fps,
                     "event_label": labels[c if i else (c + 1) % 5]}]                         # clip 0: another label
    pred = sed_amd.decode_probabilities(probs, names, labels, fps)
    assert all(len(pred[n]) == 1 for n in names)
    ev, seg = sed_amd.calculate_metrics(pred, gt, time_resolution=1.0, t_collar=0.25, verbose=False)
    assert (ev.overall["Nref"], ev.overall["Ntp"], ev.overall["Nsubs"], ev.overall["Nfp"], ev.overall["Nfn"]) == (N, N - 1, 1, 0, 0)
    assert seg.overall["Nfp"] == seg.overall["Nfn"] == seg.overall["S"] > 0 and seg.overall["D"] == seg.overall["I"] == 0
    assert seg.overall["Ntp"] == seg.overall["Nref"] - seg.overall["S"]
    oe, osg = om.EventBasedMetrics(sed_amd.decode.CLASSES, t_collar=0.25), om.SegmentBasedMetrics(sed_amd.decode.CLASSES, 1.0)
    for n in names:
        oe.evaluate(gt[n], pred[n]); osg.evaluate(gt[n], pred[n])
    assert ev.results()["overall"] == oe.results_overall_metrics() and seg.results()["overall"] == osg.results_overall_metrics()
    assert "Event based metrics (onset-offset)" in str(ev) and "Segment based metrics" in str(seg)


def test_get_classification_events_like_the_reference_loop():
    """get_classification_events (src/models/htsat/classification.py:15-36) with its signature: batches of clips, the
    events of their files relabelled in place; against the restatement, for a ClassLabel-like object and a plain list."""
    import copy
    import sed_amd
    rng = np.random.default_rng(77)
    files = [f"rec_{k}.wav" for k in range(5)]
    events = {f: [{"extracted_audio_filename": f"{f[:-4]}_{j}.00_{j + 1}.50.wav", "event_label": "unknown", "ground_truth": "siren"}
                  for j in range(int(rng.integers(1, 6)))] for f in files}
    events[files[0]].append(dict(events[files[0]][0]))                               # a repeated name: only the first is labelled
    clips = [(f, ev["extracted_audio_filename"]) for f in files for ev in events[f][:4]]
    batches = []
    for k in range(0, len(clips), 4):
        part = clips[k:k + 4]
        pm = rng.random((len(part), int(rng.choice([64, 1024, 4096])), 6)).astype(np.float32)
        pm[0, :, :5] = 0.0
        pm[0, ::2, 3] = 1.0                                                           # a tie between two classes
        pm[0, 1::2, 1] = 1.0
        batches.append({"file": [f for f, _ in part], "file_path": [f"/tmp/extracted/{e}" for _, e in part],
                        "pred_map": torch.from_numpy(pm)})
    want = od.htsat_classification_events(batches, events, NAMES6)

    class Labels:                                                                     # datasets.ClassLabel's part in it
        def int2str(self, k):
            return NAMES6[k]

    for labels in (Labels(), NAMES6):
        mine = copy.deepcopy(events)
        got = sed_amd.get_classification_events(batches, mine, labels)
        assert got is mine and got == want
    assert any(ev["event_label"] == "unknown" for ev in want[files[0]])               # the repeated name stayed as it was
    with pytest.raises(KeyError):
        sed_amd.get_classification_events([{"file": ["nowhere.wav"], "file_path": ["/tmp/n_0.wav"],
                                            "pred_map": torch.zeros(1, 8, 6)}], copy.deepcopy(events), NAMES6)
