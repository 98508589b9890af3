#!/usr/bin/env python3
"""Generate the committed golden fixtures from the reference's own code and result files.

Runs ONLY in the build container (needs /root/reference); the outputs next to this script are what
the tests read.  Nothing here is imported by tests, smoke() or bench.py.

Sources of truth used:
  * events_list_results/*.json                      (committed known-answer files)
  * src/utils/merge_events.py, src/models/crnn/utils.py   imported as modules
  * detect_onsets_offsets / get_events_from_batch / predict_events / get_classification_events:
    their modules need TensorFlow / librosa / lightning at import, so only the function bodies are
    compiled (ast) and executed with the numpy / scipy / torch / datasets globals they use.
"""
import ast
import json
import os
import sys
import textwrap

import numpy as np
import torch
from scipy.ndimage import gaussian_filter1d, median_filter

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

from src.utils.merge_events import merge_close_events as ref_merge  # noqa: E402
from src.models.crnn.utils import decode_predictions as ref_decode  # noqa: E402


def extract_function(path, name, extra_globals):
    src = open(path).read()
    tree = ast.parse(src)
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == name:
            seg = textwrap.dedent(ast.get_source_segment(src, node))
            ns = dict(extra_globals)
            exec(compile(seg, path, "exec"), ns)
            return ns[name]
    raise KeyError(name)


def compact(events_by_file, files, labels):
    fi, li, on, off = [], [], [], []
    for f in files:
        for e in events_by_file[f]:
            assert e["file"] == f
            fi.append(files.index(f)); li.append(labels.index(e["event_label"]))
            on.append(e["event_onset"]); off.append(e["event_offset"])
    return (np.asarray(fi, np.int32), np.asarray(li, np.int8), np.asarray(on, np.float64), np.asarray(off, np.float64))


def main():
    manifest = {}
    res_dir = os.path.join(REF, "events_list_results")

    # ---- D3/D4 KAT: crnn_results.json and its three merged files --------------------------------------
    raw = json.load(open(os.path.join(res_dir, "crnn_results.json")))
    files = list(raw.keys())
    labels = ["car_horn", "cough", "dog_bark", "siren", "gun_shot"]  # crnn.py:445 idx_to_label order
    pack = {"files": np.asarray(files), "labels": np.asarray(labels)}
    for k, v in zip(("fi", "li", "on", "off"), compact(raw, files, labels)):
        pack["raw_" + k] = v
    for thr in ("0.3", "0.6", "1.0"):
        merged = json.load(open(os.path.join(res_dir, f"crnn_results_merged_{thr}.json")))
        assert list(merged.keys()) == files
        mine = {f: ref_merge(raw[f], float(thr)) for f in files}
        assert mine == merged, f"reference merge does not reproduce merged_{thr}"
        for k, v in zip(("fi", "li", "on", "off"), compact(merged, files, labels)):
            pack[f"m{thr}_" + k] = v
        manifest[f"merged_{thr}_events"] = int(sum(len(v) for v in merged.values()))
    manifest["raw_events"] = int(len(pack["raw_on"]))
    np.savez_compressed(os.path.join(OUT, "crnn_kat.npz"), **pack)

    # activity (499, 862, 5) behind crnn_results.json (written with frames_per_sec = 43)
    T = 862
    act = np.zeros((len(files), T, len(labels)), dtype=np.uint8)
    k_of = {round(k * (1.0 / 43), 3): k for k in range(T + 1)}
    for i, f in enumerate(files):
        for e in raw[f]:
            s, t = k_of[e["event_onset"]], k_of[e["event_offset"]]
            act[i, s:t, labels.index(e["event_label"])] = 1
    idx_to_label = dict(enumerate(labels))
    assert ref_decode(act.astype(int), files, idx_to_label, 43) == raw
    cur = ref_decode(act.astype(int), files, idx_to_label, 43.06640625)
    fi, li, on, off = compact(cur, files, labels)
    np.savez_compressed(os.path.join(OUT, "crnn_activity.npz"), act=np.packbits(act, axis=1), T=T,
                        fps_cur_on=on, fps_cur_off=off, fps_cur_fi=fi, fps_cur_li=li)

    # ---- D1 quantisation KAT ---------------------------------------------------------------------------
    st2 = json.load(open(os.path.join(res_dir, "events_list_stage_2_yamnet_mamba.json")))
    on = np.asarray([e["event_onset"] for v in st2.values() for e in v], np.float64)
    off = np.asarray([e["event_offset"] for v in st2.values() for e in v], np.float64)
    counts = np.asarray([len(v) for v in st2.values()], np.int32)
    np.savez_compressed(os.path.join(OUT, "yamnet_stage2_times.npz"), on=on, off=off, counts=counts)
    manifest["yamnet_stage2_times"] = int(on.size + off.size)

    # ---- D1 vectors from the reference function -------------------------------------------------------
    det = extract_function(os.path.join(REF, "src/models/yamnet_train.py"), "detect_onsets_offsets",
                           {"np": np, "median_filter": median_filter})
    rng = np.random.default_rng(7)
    cases = [np.array(c, np.float32) for c in (
        [1, 1, 0, 0, 1, 1], [1, 1, 1, 1], [0, 0, 0, 0], [0, 1, 1, 0], [1, 0, 0, 0], [0, 0, 0, 1],
        [0.5, 0.5, 0.50001, 0.5], [0, 1, 0, 1, 0, 1, 0], [1, 0, 1, 0, 1], [0.9], [0.1], [0.9, 0.1], [0.1, 0.9])]
    for n in (20, 41, 41, 41, 64, 100, 7, 3, 2):
        for _ in range(22):
            base = rng.random(n) < rng.uniform(0.1, 0.9)
            p = np.where(base, rng.uniform(0.5, 1.0, n), rng.uniform(0.0, 0.5, n))
            if rng.random() < 0.3:
                p = np.repeat(p[: max(1, n // 4)], 4)[:n]
            cases.append(p.astype(np.float32))
    arrays = {}
    recs = []
    for i, p in enumerate(cases):
        arrays[f"d1_{i}"] = p
        rec = {}
        for mf in (True, False):
            o, f = det(p.copy(), threshold=0.5, hop_time=0.48, med_filter=mf)
            rec[f"on_{int(mf)}"] = [float(v).hex() for v in np.atleast_1d(o)]
            rec[f"off_{int(mf)}"] = [float(v).hex() for v in np.atleast_1d(f)]
        recs.append(rec)
    json.dump({"cases": recs}, open(os.path.join(OUT, "d1_vectors.json"), "w"))
    manifest["d1_cases"] = len(cases)

    # ---- D2 vectors: scipy gaussian_filter1d bits ----------------------------------------------------
    rng = np.random.default_rng(11)
    x = rng.random((4, 862, 5), dtype=np.float32)
    x[1] = (x[1] > 0.7).astype(np.float32)
    x[2, :30] = 1.0
    x[3] = 1.0 / (1.0 + np.exp(-6.0 * gaussian_filter1d(rng.standard_normal((862, 5)), 8.0, axis=0))).astype(np.float32)
    y = gaussian_filter1d(input=x, sigma=3, axis=1, mode="nearest")
    assert y.dtype == np.float32
    np.savez_compressed(os.path.join(OUT, "d2_vectors.npz"), x=x, y=y)

    # ---- D5 vectors ------------------------------------------------------------------------------------
    from datasets import ClassLabel
    classes = ["cough", "dog_bark", "gun_shot", "siren", "car_horn"]  # src/config.py:4
    CLASS_LABELS = ClassLabel(names=classes + ["non_event"])
    gefb = extract_function(os.path.join(REF, "src/models/htsat/combined.py"), "get_events_from_batch",
                            {"torch": torch, "gaussian_filter1d": gaussian_filter1d, "CLASS_LABELS": CLASS_LABELS,
                             "NON_EVENT_LABEL": "non_event"})

    class Cfg:
        clip_duration = 20.0

    d5 = []
    rng = np.random.default_rng(13)
    for n_frame, bs in ((40, 2), (4096, 1), (1024, 1), (512, 3), (64, 1), (64, 1)):
        lab = np.full(n_frame, 5)
        pos = 0
        while pos < n_frame:
            ln = int(rng.integers(1, max(2, n_frame // 6)))
            lab[pos:pos + ln] = rng.integers(0, 6)
            pos += ln
        if len(d5) == 4:
            lab[:] = 5
        if len(d5) == 5:
            lab[:] = 2
        pm = rng.random((bs, n_frame, 6)).astype(np.float32) * 0.6
        for b in range(bs):
            pm[b, np.arange(n_frame), np.roll(lab, 3 * b)] += 0.5
        names = [f"clip_{len(d5)}_{b}.wav" for b in range(bs)]
        ev = gefb({"file": names, "pred_map": torch.from_numpy(pm)}, Cfg)
        arrays[f"d5_{len(d5)}"] = pm
        d5.append({"files": names, "expected": ev})
    json.dump(d5, open(os.path.join(OUT, "d5_vectors.json"), "w"))

    # ---- D6 vectors ------------------------------------------------------------------------------------
    pe = extract_function(os.path.join(REF, "src/models/single_stage_yamnet_frame.py"), "predict_events",
                          {"os": os, "YAMNET_HOP_SEC": 0.48})

    class Stub:
        def __init__(self, probs):
            self.probs = probs
            self.classes = classes

        def predict_class(self, file, med_filter=False):
            fp = self.probs
            if med_filter:
                fp = median_filter(fp, size=(3, 1))
            return np.argmax(fp, axis=-1)

    d6 = []
    rng = np.random.default_rng(17)
    for case in range(24):
        n = (41, 20, 41, 5, 2, 1)[case % 6]
        lab = np.repeat(rng.integers(0, 6, n), rng.integers(1, 6))[:n]
        if lab.shape[0] < n:
            lab = np.resize(lab, n)
        fp = rng.random((n, 6)).astype(np.float32) * 0.7
        fp[np.arange(n), lab] += 0.4
        arrays[f"d6_{case}"] = fp
        rec = {}
        for mf in (True, False):
            rec[f"expected_{int(mf)}"] = pe(Stub(fp), f"/data/x/clip_{case}.wav", med_filter=mf)
        d6.append(rec)
    json.dump(d6, open(os.path.join(OUT, "d6_vectors.json"), "w"))

    # ---- D7 vectors ------------------------------------------------------------------------------------
    gce = extract_function(os.path.join(REF, "src/models/htsat/classification.py"), "get_classification_events",
                           {"os": os, "torch": torch})
    d7 = []
    rng = np.random.default_rng(19)
    for case in range(12):
        n = (1024, 64, 7)[case % 3]
        pm = rng.random((n, 6)).astype(np.float32)
        if case >= 6:                                   # force ties between two classes
            pm[:, :5] = 0.0
            half = n // 2
            pm[:half, 3] = 1.0
            pm[half:2 * half, 1] = 1.0
            pm[2 * half:, 0] = 1.0
        evl = {"a.wav": [{"extracted_audio_filename": "a_0.00_1.00.wav", "event_label": None}]}
        out = gce([{"file": ["a.wav"], "file_path": ["/tmp/a_0.00_1.00.wav"], "pred_map": torch.from_numpy(pm)[None]}],
                  evl, CLASS_LABELS)
        arrays[f"d7_{case}"] = pm
        d7.append({"label": out["a.wav"][0]["event_label"]})
    json.dump(d7, open(os.path.join(OUT, "d7_vectors.json"), "w"))

    np.savez_compressed(os.path.join(OUT, "decode_inputs.npz"), **arrays)
    manifest["scipy"] = __import__("scipy").__version__
    manifest["numpy"] = np.__version__
    json.dump(manifest, open(os.path.join(OUT, "MANIFEST.json"), "w"), indent=1)
    print(manifest)


if __name__ == "__main__":
    main()
