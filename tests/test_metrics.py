"""Segment-based and event-based detection metrics (main.py:29-48): `sed_amd.metrics` against the one-file-at-a-time
restatement of sed_eval 0.2.1 in oracle/metrics.py, against hand-worked cases, and against the reports the reference
printed (notebooks/evaluation_test.ipynb, copied into tests/golden/sed_eval_reports.json).

PARITY UNPINNED for the counting itself (rasterisation, matching): sed_eval is not installed and the ground truth behind
the printed reports is not in the reference.  The reports do pin everything from counts to printed figures: the test
below recovers per-class counts from each report's own columns, finds the free quantities (total segments, substituted
events) by search, and requires the rendered report to equal the printed one figure for figure.
"""
import importlib.util
import itertools
import json
import os
import re
import sys

import numpy as np
import pytest

from oracle import metrics as om

HERE = os.path.dirname(os.path.abspath(__file__))
CLASSES = ["cough", "dog_bark", "gun_shot", "siren", "car_horn"]


def _load_metrics():
    """sed_amd/metrics.py is host code: load it without the package, whose import needs the HIP library and a GPU build."""
    path = os.path.join(HERE, "..", "sound-event-detection_amd", "sed_amd", "metrics.py")
    spec = importlib.util.spec_from_file_location("sed_amd_metrics", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


pm = _load_metrics()


def ev(label, on, off, file="a.wav"):
    return {"file": file, "event_onset": on, "event_offset": off, "event_label": label}


# ---------------------------------------------------------------------------------------------------------------------
# hand-worked cases
# ---------------------------------------------------------------------------------------------------------------------

def test_segment_based_by_hand():
    ref = [ev("siren", 1.56, 5.41)]                                             # segments 1 .. 5
    est = [ev("siren", 0.96, 3.84), ev("car_horn", 3.84, 5.76)]                 # siren 0 .. 3, car_horn 3 .. 5
    for cls in (pm.SegmentBasedMetrics, om.SegmentBasedMetrics):
        m = cls(CLASSES, time_resolution=1.0).evaluate(ref, est)
        assert m.evaluated_length_seconds == 5.76 and m.evaluated_files == 1
        s = m.class_wise["siren"]
        assert (s["Nref"], s["Nsys"], s["Ntp"], s["Nfp"], s["Nfn"], s["Ntn"]) == (5, 4, 3, 1, 2, 0)
        c = m.class_wise["car_horn"]
        assert (c["Nref"], c["Nsys"], c["Ntp"], c["Nfp"], c["Nfn"], c["Ntn"]) == (0, 3, 0, 3, 0, 3)
        o = m.overall
        # segments: 0 {sys siren}: I; 1, 2 {both}: TP; 3 {ref siren, sys siren + horn}: TP + I; 4, 5 {ref siren, sys horn}: S
        assert (o["Ntp"], o["Nfp"], o["Nfn"], o["S"], o["D"], o["I"], o["Nref"], o["Nsys"]) == (3, 4, 2, 2, 0, 2, 5, 7)
        r = m.results_overall_metrics()
        assert r["f_measure"]["precision"] == pytest.approx(3 / 7) and r["f_measure"]["recall"] == pytest.approx(3 / 5)
        assert r["error_rate"]["error_rate"] == pytest.approx(4 / 5)
        assert r["accuracy"]["accuracy"] == pytest.approx((3 + 21) / 30)


def test_event_based_by_hand():
    ref = [ev("siren", 1.0, 5.0), ev("cough", 6.0, 6.5), ev("dog_bark", 8.0, 9.0)]
    est = [ev("siren", 1.2, 6.5),      # onset within 0.25, offset 1.5 off <= max(0.25, 0.5 * 4): hit
           ev("cough", 6.1, 7.0),      # offset 0.5 off > max(0.25, 0.25): miss, and no substitution either (times disagree)
           ev("gun_shot", 8.1, 9.2),   # times agree with the dog bark, label does not: substitution
           ev("car_horn", 12.0, 13.0)]  # insertion
    for cls in (pm.EventBasedMetrics, om.EventBasedMetrics):
        m = cls(CLASSES, t_collar=0.25).evaluate(ref, est)
        o = m.overall
        assert (o["Nref"], o["Nsys"], o["Ntp"], o["Nsubs"], o["Nfp"], o["Nfn"]) == (3, 4, 1, 1, 2, 1)
        assert m.evaluated_length == 9.0
        cw = m.class_wise
        assert (cw["siren"]["Ntp"], cw["cough"]["Nfp"], cw["cough"]["Nfn"], cw["dog_bark"]["Nfn"], cw["gun_shot"]["Nfp"]) == (1, 1, 1, 1, 1)
        r = m.results_overall_metrics()
        assert r["error_rate"]["substitution_rate"] == pytest.approx(1 / 3)
        assert r["error_rate"]["error_rate"] == pytest.approx(4 / 3)


def test_matching_is_maximum_not_greedy():
    """Two estimated events both hit reference 0, only the first also hits reference 1: the maximum matching pairs both."""
    ref = [ev("siren", 1.0, 2.0), ev("siren", 1.3, 2.3)]
    est = [ev("siren", 1.15, 2.15), ev("siren", 0.8, 1.8)]
    for cls in (pm.EventBasedMetrics, om.EventBasedMetrics):
        m = cls(CLASSES, t_collar=0.2, percentage_of_length=0.0).evaluate(ref, est)
        assert m.overall["Ntp"] == 2 and m.class_wise["siren"]["Ntp"] == 2


# ---------------------------------------------------------------------------------------------------------------------
# against the file-at-a-time restatement
# ---------------------------------------------------------------------------------------------------------------------

def _random_files(rng, n_files, max_ref, max_est, quantum=None):
    gt, pred = {}, {}
    for f in range(n_files):
        name = f"f{f:03d}.wav"
        def events(n):
            out = []
            for _ in range(n):
                on = float(rng.uniform(0, 18))
                off = on + float(rng.exponential(1.5)) + 0.05
                if quantum:
                    on, off = round(round(on / quantum) * quantum, 3), round(round(off / quantum) * quantum, 3)
                if rng.random() < 0.03:
                    on, off = off, on                                   # the detector decode can emit such a pair
                out.append(ev(CLASSES[int(rng.integers(len(CLASSES)))], on, off, name))
            return out
        gt[name] = events(int(rng.integers(0 if max_ref > 1 else 1, max_ref + 1)))
        pred[name] = events(int(rng.integers(0, max_est + 1)))
    return gt, pred


@pytest.mark.parametrize("seed,max_ref,max_est,quantum,res,collar", [
    (1, 1, 20, 0.48, 1.0, 0.5),       # the reference's shape: one reference event per file, detector hop 0.48 s
    (2, 6, 8, None, 1.0, 0.25),
    (3, 12, 12, 0.1, 0.5, 0.2),       # many events on a coarse grid: many candidates per event
    (4, 4, 30, 0.023, 0.1, 1.0),
])
def test_files_in_one_pass_equal_one_file_at_a_time(seed, max_ref, max_est, quantum, res, collar):
    rng = np.random.default_rng(seed)
    gt, pred = _random_files(rng, 60, max_ref, max_est, quantum)
    seg = pm.SegmentBasedMetrics(CLASSES, time_resolution=res).evaluate_files(gt, pred)
    evb = pm.EventBasedMetrics(CLASSES, t_collar=collar).evaluate_files(gt, pred)
    oseg, oevb = om.SegmentBasedMetrics(CLASSES, time_resolution=res), om.EventBasedMetrics(CLASSES, t_collar=collar)
    seg1, evb1 = pm.SegmentBasedMetrics(CLASSES, time_resolution=res), pm.EventBasedMetrics(CLASSES, t_collar=collar)
    for f in pred:
        oseg.evaluate(gt[f], pred[f]); oevb.evaluate(gt[f], pred[f])
        seg1.evaluate(reference_event_list=gt[f], estimated_event_list=pred[f])
        evb1.evaluate(reference_event_list=gt[f], estimated_event_list=pred[f])
    for mine in (seg, seg1):
        assert mine.overall == oseg.overall and mine.class_wise == oseg.class_wise
        assert mine.evaluated_length_seconds == oseg.evaluated_length_seconds and mine.evaluated_files == 60
        assert mine.results_overall_metrics() == oseg.results_overall_metrics()
        assert mine.results_class_wise_metrics() == oseg.results_class_wise_metrics()
        assert mine.results_class_wise_average_metrics() == om.class_wise_average(oseg.results_class_wise_metrics())
    for mine in (evb, evb1):
        assert mine.overall == oevb.overall and mine.class_wise == oevb.class_wise
        assert mine.evaluated_length == oevb.evaluated_length
        assert mine.results_overall_metrics() == oevb.results_overall_metrics()
        assert mine.results_class_wise_metrics() == oevb.results_class_wise_metrics()
    assert oevb.overall["Ntp"] > 0 and oseg.overall["S"] > 0


def test_evaluated_length_given_and_event_options():
    rng = np.random.default_rng(9)
    gt, pred = _random_files(rng, 20, 3, 6, 0.25)
    for f in pred:
        a = pm.SegmentBasedMetrics(CLASSES, 0.5).evaluate(gt[f], pred[f], evaluated_length_seconds=7.3)
        b = om.SegmentBasedMetrics(CLASSES, 0.5).evaluate(gt[f], pred[f], evaluated_length_seconds=7.3)
        assert a.overall == b.overall and a.class_wise == b.class_wise
    for onset, offset, pct in ((True, False, 0.5), (False, True, 0.2), (True, True, 0.0)):
        a = pm.EventBasedMetrics(CLASSES, evaluate_onset=onset, evaluate_offset=offset, t_collar=0.3, percentage_of_length=pct)
        b = om.EventBasedMetrics(CLASSES, evaluate_onset=onset, evaluate_offset=offset, t_collar=0.3, percentage_of_length=pct)
        for f in pred:
            a.evaluate(gt[f], pred[f]); b.evaluate(gt[f], pred[f])
        assert a.overall == b.overall and a.class_wise == b.class_wise
    with pytest.raises(KeyError):
        pm.SegmentBasedMetrics(CLASSES).evaluate([ev("glass_break", 0.0, 1.0)], [])
    with pytest.raises(ValueError):
        pm.EventBasedMetrics(CLASSES, event_matching_type="greedy")


def test_bipartite_matching_follows_the_recipe_pair_for_pair():
    """Not only the size of the matching: which events stay unmatched decides the substitution count."""
    rng = np.random.default_rng(5)
    for _ in range(400):
        n_ref, n_est = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        hit = rng.random((n_ref, n_est)) < rng.uniform(0.15, 0.7)
        graph, adj = {}, {}
        for j, i in zip(*np.nonzero(hit)):
            graph.setdefault(int(i), []).append(int(j))
            adj.setdefault(int(i), []).append(int(j))
        want = om.bipartite_match(graph)                                # reference event -> estimated event
        got = pm._bipartite_match(adj, n_ref)
        assert {j: i for j, i in enumerate(got) if i >= 0} == want
        # a maximum matching: no brute-force assignment is larger
        best = 0
        for perm in itertools.permutations(range(max(n_ref, n_est)), min(n_ref, n_est)) if max(n_ref, n_est) <= 6 else ():
            pairs = zip(range(n_ref), perm) if n_ref <= n_est else zip(perm, range(n_est))
            best = max(best, sum(1 for j, i in pairs if j < n_ref and i < n_est and hit[j, i]))
        if max(n_ref, n_est) <= 6:
            assert len(want) == best


# ---------------------------------------------------------------------------------------------------------------------
# the reports the reference printed
# ---------------------------------------------------------------------------------------------------------------------

def _norm(text):
    return [" ".join(l.split()) for l in text.splitlines() if l.strip()]


def _rows(block):
    rows = {}
    for l in block.splitlines():
        cells = [c.strip() for c in l.split("|")]
        if len(cells) >= 4 and cells[0] in CLASSES:
            rows[cells[0]] = [[float(v.rstrip("%")) for v in c.split()] for c in cells[1:] if c]
    return rows


def _field(block, name):
    return float(re.search(re.escape(name) + r"\s*:\s*([-\d.]+)", block).group(1))


def _class_tp(nref, nsys, f, p, r):
    sol = [tp for tp in range(0, int(min(nref, nsys)) + 1)
           if f"{100 * pm._ratio(tp, nsys):.1f}" == f"{p:.1f}" and f"{100 * pm._ratio(tp, nref):.1f}" == f"{r:.1f}"]
    assert len(sol) == 1, (nref, nsys, p, r, sol)
    return sol[0]


REPORTS = json.load(open(os.path.join(HERE, "golden", "sed_eval_reports.json")))


@pytest.mark.parametrize("name", sorted(REPORTS))
def test_reports_printed_by_the_reference(name):
    text = REPORTS[name]
    ev_text, seg_text = text[:text.index("Segment based metrics")], text[text.index("Segment based metrics"):]

    # event-based: per-class true positives from the printed precision / recall; substituted events by search
    rows = _rows(ev_text)
    m = pm.EventBasedMetrics(CLASSES, t_collar=_field(ev_text, "T collar") / 1000.0,
                             percentage_of_length=_field(ev_text, "Offset (length)") / 100.0)
    m.evaluated_length, m.evaluated_files = _field(ev_text, "Evaluated length"), int(_field(ev_text, "Evaluated files"))
    for c in CLASSES:
        (nref, nsys), (f, p, r), _ = rows[c]
        tp = _class_tp(nref, nsys, f, p, r)
        m.class_wise[c].update(Nref=nref, Nsys=nsys, Ntp=tp, Nfp=nsys - tp, Nfn=nref - tp)
    tot = {k: sum(m.class_wise[c][k] for c in CLASSES) for k in ("Nref", "Nsys", "Ntp")}
    assert tot["Nref"] == 500                                            # one reference event per test file
    found = False
    for subs in range(0, int(min(tot["Nref"], tot["Nsys"]) - tot["Ntp"]) + 1):
        m.overall.update(tot, Nsubs=subs, Nfp=tot["Nsys"] - tot["Ntp"] - subs, Nfn=tot["Nref"] - tot["Ntp"] - subs)
        if _norm(str(m)) == _norm(ev_text):
            found = True
            break
    assert found, "\n".join(l for l in _norm(str(m)) if l not in _norm(ev_text))

    # segment-based: true negatives follow from the total number of segments (the same for every class), found by search;
    # per segment, misses split into substitutions and deletions, false alarms into substitutions and insertions
    rows = _rows(seg_text)
    s = pm.SegmentBasedMetrics(CLASSES, time_resolution=_field(seg_text, "Segment length"))
    s.evaluated_length_seconds, s.evaluated_files = _field(seg_text, "Evaluated length"), int(_field(seg_text, "Evaluated files"))
    for c in CLASSES:
        (nref, nsys), (f, p, r) = rows[c][0], rows[c][1]
        tp = _class_tp(nref, nsys, f, p, r)
        s.class_wise[c].update(Nref=nref, Nsys=nsys, Ntp=tp, Nfp=nsys - tp, Nfn=nref - tp)
    tot = {k: sum(s.class_wise[c][k] for c in CLASSES) for k in ("Nref", "Nsys", "Ntp", "Nfp", "Nfn")}
    lo = int(np.ceil(s.evaluated_length_seconds))
    found = False
    for n_seg in range(lo, lo + 500):                                   # sum over 500 files of ceil(length of the file)
        for c in CLASSES:
            cw = s.class_wise[c]
            cw["Ntn"] = n_seg - cw["Ntp"] - cw["Nfp"] - cw["Nfn"]
        table = _norm(pm._report_class_table(s.results_class_wise_metrics()))
        if not all(l in _norm(seg_text) for l in table):
            continue
        for subs in range(0, int(min(tot["Nfp"], tot["Nfn"])) + 1):
            s.overall.update(tot, Ntn=sum(s.class_wise[c]["Ntn"] for c in CLASSES), S=subs, D=tot["Nfn"] - subs, I=tot["Nfp"] - subs)
            if _norm(str(s)) == _norm(seg_text):
                found = True
                break
        if found:
            break
    assert found
