"""The decode oracle against the golden fixtures generated from the reference (CPU only)."""
import json
import os

import numpy as np
import pytest

from oracle import decode as od


def _events(pack, prefix, files, labels):
    out = {f: [] for f in files}
    for fi, li, on, off in zip(pack[prefix + "_fi"], pack[prefix + "_li"], pack[prefix + "_on"], pack[prefix + "_off"]):
        f = files[int(fi)]
        out[f].append({"file": f, "event_onset": float(on), "event_offset": float(off), "event_label": labels[int(li)]})
    return out


@pytest.fixture(scope="module")
def kat(golden_dir):
    pack = np.load(os.path.join(golden_dir, "crnn_kat.npz"))
    files = [str(f) for f in pack["files"]]
    labels = [str(l) for l in pack["labels"]]
    return pack, files, labels


@pytest.mark.parametrize("thr", ["0.3", "0.6", "1.0"])
def test_merge_kat(kat, thr):
    pack, files, labels = kat
    raw = _events(pack, "raw", files, labels)
    want = _events(pack, "m" + thr, files, labels)
    got = {f: od.merge_close_events(raw[f], float(thr)) for f in files}
    assert got == want
    assert sum(len(v) for v in got.values()) == {"0.3": 2335, "0.6": 1855, "1.0": 1635}[thr]


def test_merge_empty_and_overlap():
    assert od.merge_close_events([], 0.5) == []
    ev = [dict(file="a", event_onset=1.0, event_offset=5.0, event_label="x"),
          dict(file="a", event_onset=2.0, event_offset=3.0, event_label="x"),
          dict(file="a", event_onset=5.4, event_offset=6.0, event_label="x"),
          dict(file="a", event_onset=0.5, event_offset=0.6, event_label="y")]
    got = od.merge_close_events(ev, 0.5)
    assert [(e["event_onset"], e["event_offset"], e["event_label"]) for e in got] == [(0.5, 0.6, "y"), (1.0, 6.0, "x")]


def test_decode_predictions_roundtrip(kat, golden_dir):
    pack, files, labels = kat
    a = np.load(os.path.join(golden_dir, "crnn_activity.npz"))
    T = int(a["T"])
    act = np.unpackbits(a["act"], axis=1)[:, :T, :].astype(int)
    idx_to_label = dict(enumerate(labels))
    assert od.decode_predictions(act, files, idx_to_label, 43) == _events(pack, "raw", files, labels)
    cur = od.decode_predictions(act, files, idx_to_label, 43.06640625)
    on = np.asarray([e["event_onset"] for f in files for e in cur[f]])
    off = np.asarray([e["event_offset"] for f in files for e in cur[f]])
    assert np.array_equal(on, a["fps_cur_on"]) and np.array_equal(off, a["fps_cur_off"])


def test_decode_predictions_rounding_at_other_frame_rates(golden_dir):
    """d3_rounding.npz: the reference's decode_predictions at eight frame rates, every frame index a boundary.  The
    reference rounds an np.float64 (numpy's rint(x * 1000) / 1000), which differs from rounding a Python float for
    thousands of (frame, rate) pairs: at 2000 frames/s frame 1 is 0.0, not 0.001."""
    z = np.load(os.path.join(golden_dir, "d3_rounding.npz"))
    T = int(z["T"])
    act = np.zeros((1, T, 1), dtype=int)
    act[0, 0::2, 0] = 1
    n_python_differs = 0
    for i, fps in enumerate(z["fps"]):
        ev = od.decode_predictions(act, ["f.wav"], {0: "e"}, float(fps))["f.wav"]
        assert [e["event_onset"] for e in ev] == z[f"on_{i}"].tolist()
        assert [e["event_offset"] for e in ev] == z[f"off_{i}"].tolist()
        res = 1.0 / float(fps)
        n_python_differs += sum(round(k * res, 3) != v for k, v in zip(range(1, T + 1, 2), z[f"off_{i}"].tolist()))
    assert n_python_differs > 0


def test_yamnet_stage2_quantisation(golden_dir):
    z = np.load(os.path.join(golden_dir, "yamnet_stage2_times.npz"))
    grid = np.arange(0, 43) * 0.48
    for arr in (z["on"], z["off"]):
        k = np.rint(arr / 0.48).astype(int)
        assert k.max() <= 42 and np.array_equal(grid[k], arr)


def test_detect_onsets_offsets_vectors(golden_dir):
    exp = json.load(open(os.path.join(golden_dir, "d1_vectors.json")))["cases"]
    inp = np.load(os.path.join(golden_dir, "decode_inputs.npz"))
    for i, rec in enumerate(exp):
        p = inp[f"d1_{i}"]
        for mf in (1, 0):
            on, off = od.detect_onsets_offsets(p, 0.5, 0.48, bool(mf))
            assert [float(v).hex() for v in np.atleast_1d(on)] == rec[f"on_{mf}"], (i, mf)
            assert [float(v).hex() for v in np.atleast_1d(off)] == rec[f"off_{mf}"], (i, mf)


def test_detect_quirks():
    on, off = od.detect_onsets_offsets(np.array([1, 1, 0, 0, 1, 1], np.float32), med_filter=False)
    assert on.tolist() == [4 * 0.48] and off.tolist() == [2 * 0.48]          # reversed pair
    on, off = od.detect_onsets_offsets(np.ones(5, np.float32), med_filter=False)
    assert on.size == 0 and off.size == 0                                      # all-active -> nothing
    on, off = od.detect_onsets_offsets(np.array([0, 0, 1, 1], np.float32), med_filter=False)
    assert off.tolist() == [5 * 0.48]                                          # one hop past the end


def test_gaussian_bits(golden_dir):
    z = np.load(os.path.join(golden_dir, "d2_vectors.npz"))
    y = od.gaussian_smooth_time(z["x"])
    assert y.dtype == np.float32 and np.array_equal(y.view(np.uint32), z["y"].view(np.uint32))
    assert od.gaussian_half_kernel(3.0) == od.GAUSS_SIGMA3_HALF


def test_gaussian_vs_scipy_live():
    sp = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(5)
    x = rng.random((3, 100, 6), dtype=np.float32)
    want = sp.gaussian_filter1d(x, sigma=3, axis=1, mode="nearest")
    assert np.array_equal(od.gaussian_smooth_time(x).view(np.uint32), want.view(np.uint32))
    x1 = rng.random((41, 6), dtype=np.float32)
    assert np.array_equal(od.median3_time(x1, 0), sp.median_filter(x1, size=(3, 1)))
    x2 = rng.random(7, dtype=np.float32)
    assert np.array_equal(od.median3_time(x2, 0), sp.median_filter(x2, size=3))


def test_htsat_events_vectors(golden_dir):
    exp = json.load(open(os.path.join(golden_dir, "d5_vectors.json")))
    inp = np.load(os.path.join(golden_dir, "decode_inputs.npz"))
    names = ["cough", "dog_bark", "gun_shot", "siren", "car_horn", "non_event"]
    for i, rec in enumerate(exp):
        got = od.htsat_events_from_batch(rec["files"], inp[f"d5_{i}"], names, 5, 20.0)
        assert got == rec["expected"], i


def test_yamnet_frame_events_vectors(golden_dir):
    exp = json.load(open(os.path.join(golden_dir, "d6_vectors.json")))
    inp = np.load(os.path.join(golden_dir, "decode_inputs.npz"))
    classes = ["cough", "dog_bark", "gun_shot", "siren", "car_horn"]
    for i, rec in enumerate(exp):
        for mf in (1, 0):
            got = od.yamnet_frame_events(inp[f"d6_{i}"], f"clip_{i}.wav", classes, 0.48, bool(mf))
            assert got == rec[f"expected_{mf}"], (i, mf)


def test_htsat_clip_label_vectors(golden_dir):
    exp = json.load(open(os.path.join(golden_dir, "d7_vectors.json")))
    inp = np.load(os.path.join(golden_dir, "decode_inputs.npz"))
    names = ["cough", "dog_bark", "gun_shot", "siren", "car_horn", "non_event"]
    for i, rec in enumerate(exp):
        assert names[od.htsat_clip_label(inp[f"d7_{i}"])] == rec["label"], i


def test_htsat_classification_events_vectors(golden_dir):
    """The golden labels came out of the reference's get_classification_events (classification.py:15-36) for a file with
    one event; here all twelve clips go through the restatement of that function in batches of three, as events of two
    files, with a decoy event that repeats a clip's name (only the first match is labelled) and one event no clip names."""
    exp = json.load(open(os.path.join(golden_dir, "d7_vectors.json")))
    inp = np.load(os.path.join(golden_dir, "decode_inputs.npz"))
    names = ["cough", "dog_bark", "gun_shot", "siren", "car_horn", "non_event"]
    by_len = {}
    for i in range(len(exp)):
        by_len.setdefault(inp[f"d7_{i}"].shape[0], []).append(i)
    events = {"a.wav": [], "b.wav": []}
    batches = []
    for n, idx in by_len.items():
        for k in range(0, len(idx), 3):
            part = idx[k:k + 3]
            files = ["a.wav" if i % 2 == 0 else "b.wav" for i in part]
            batches.append({"file": files, "file_path": [f"/x/y/{f[0]}_{i}.00_{i + 1}.00.wav" for f, i in zip(files, part)],
                            "pred_map": np.stack([inp[f"d7_{i}"] for i in part])})
            for f, i in zip(files, part):
                events[f].append({"extracted_audio_filename": f"{f[0]}_{i}.00_{i + 1}.00.wav", "event_label": None, "k": i})
    events["a.wav"].append({"extracted_audio_filename": "a_0.00_1.00.wav", "event_label": "decoy"})
    events["b.wav"].append({"extracted_audio_filename": "b_99.00_100.00.wav", "event_label": None})
    out = od.htsat_classification_events(batches, events, names)
    seen = 0
    for f in ("a.wav", "b.wav"):
        for ev in out[f]:
            if "k" in ev:
                assert ev["event_label"] == exp[ev["k"]]["label"], ev
                seen += 1
    assert seen == len(exp)
    assert out["a.wav"][-1]["event_label"] == "decoy" and out["b.wav"][-1]["event_label"] is None
    assert events["a.wav"][0]["event_label"] is None                                  # the input is left alone
