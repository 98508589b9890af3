"""CPU oracle for the DCASE sound-event-detection metrics behind the decoders.  TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module; the product package never does.

PARITY UNPINNED.  The reference scores its event lists with sed_eval (requirements.txt:9 pins sed_eval==0.2.1;
main.py:29-48 builds `EventBasedMetrics(CLASSES, t_collar=...)` and `SegmentBasedMetrics(CLASSES, time_resolution=...)`,
calls `evaluate(reference_event_list=..., estimated_event_list=...)` once per file and prints both objects).  sed_eval is
neither vendored in the reference nor installed here, and the ground-truth onsets behind the reports printed in
notebooks/evaluation_test.ipynb are not in the reference either, so this file restates sed_eval's published algorithm
(sed_eval/sound_event.py, sed_eval/metric.py, sed_eval/util/event_roll.py, sed_eval/util/event_matching.py of release
0.2.1) one file and one event at a time, as that library does.  What the printed reports do pin are the formulas that
turn counts into rates (tests/test_oracle_metrics.py re-derives their columns from one another).

Event lists are lists of dicts with 'event_onset', 'event_offset', 'event_label' (the reference's own format,
src/models/crnn/utils.py:118-127).
"""
from __future__ import annotations

import math

import numpy as np

EPS = float(np.spacing(1))


# ----------------------------------------------------------------------------
# sed_eval/metric.py
# ----------------------------------------------------------------------------

def precision(Ntp, Nsys):
    return float(Ntp / float(Nsys + EPS))


def recall(Ntp, Nref):
    return float(Ntp / float(Nref + EPS))


def f_measure(p, r, beta=1.0):
    if p == 0 and r == 0:
        return 0.0
    return float((1 + beta ** 2) * p * r / ((beta ** 2) * p + r))


def sensitivity(Ntp, Nfn):
    return float(Ntp / (Ntp + Nfn + EPS))


def specificity(Ntn, Nfp):
    return float(Ntn / (Ntn + Nfp + EPS))


def balanced_accuracy(sens, spec, factor=0.5):
    return float(((1 - factor) * sens) + (factor * spec))


def accuracy(Ntp, Ntn, Nfp, Nfn):
    return float((Ntp + Ntn) / (Ntp + Ntn + Nfn + Nfp + EPS))


def rate(N, Nref):
    return float(N / (Nref + EPS))


# ----------------------------------------------------------------------------
# sed_eval/util/event_roll.py
# ----------------------------------------------------------------------------

def max_offset(event_list):
    m = 0
    for e in event_list:
        if e["event_offset"] > m:
            m = e["event_offset"]
    return m


def event_list_to_event_roll(event_list, labels, time_resolution):
    roll = np.zeros((int(math.ceil(max_offset(event_list) * 1 / time_resolution)), len(labels)))
    for e in event_list:
        pos = labels.index(e["event_label"])
        onset = int(math.floor(e["event_onset"] * 1 / time_resolution))
        offset = int(math.ceil(e["event_offset"] * 1 / time_resolution))
        roll[onset:offset, pos] = 1
    return roll


def pad_event_roll(roll, length):
    if length > roll.shape[0]:
        roll = np.vstack((roll, np.zeros((length - roll.shape[0], roll.shape[1]))))
    return roll


def match_event_roll_lengths(a, b, length=None):
    length = max(a.shape[0], b.shape[0]) if length is None else int(length)
    a = pad_event_roll(a, length) if length >= a.shape[0] else a[:length]
    b = pad_event_roll(b, length) if length >= b.shape[0] else b[:length]
    return a, b


# ----------------------------------------------------------------------------
# sed_eval/util/event_matching.py: Hopcroft-Karp as in D. Eppstein's recipe.  graph: u -> list of v; returns v -> u.
# ----------------------------------------------------------------------------

def bipartite_match(graph):
    matching = {}
    for u in graph:
        for v in graph[u]:
            if v not in matching:
                matching[v] = u
                break
    while True:
        preds = {}
        unmatched = []
        pred = dict([(u, unmatched) for u in graph])
        for v in matching:
            del pred[matching[v]]
        layer = list(pred)
        while layer and not unmatched:
            new_layer = {}
            for u in layer:
                for v in graph[u]:
                    if v not in preds:
                        new_layer.setdefault(v, []).append(u)
            layer = []
            for v in new_layer:
                preds[v] = new_layer[v]
                if v in matching:
                    layer.append(matching[v])
                    pred[matching[v]] = v
                else:
                    unmatched.append(v)
        if not unmatched:
            return matching

        def recurse(v):
            if v in preds:
                L = preds[v]
                del preds[v]
                for u in L:
                    if u in pred:
                        pu = pred[u]
                        del pred[u]
                        if pu is unmatched or recurse(pu):
                            matching[v] = u
                            return 1
            return 0

        for v in unmatched:
            recurse(v)


# ----------------------------------------------------------------------------
# sed_eval/sound_event.py: SegmentBasedMetrics
# ----------------------------------------------------------------------------

class SegmentBasedMetrics:
    def __init__(self, event_label_list, time_resolution=1.0):
        self.event_label_list = list(event_label_list)
        self.time_resolution = time_resolution
        self.evaluated_length_seconds = 0.0
        self.evaluated_files = 0
        self.overall = dict(Ntp=0.0, Ntn=0.0, Nfp=0.0, Nfn=0.0, Nref=0.0, Nsys=0.0, ER=0.0, S=0.0, D=0.0, I=0.0)
        self.class_wise = {c: dict(Ntp=0.0, Ntn=0.0, Nfp=0.0, Nfn=0.0, Nref=0.0, Nsys=0.0) for c in self.event_label_list}

    def evaluate(self, reference_event_list, estimated_event_list, evaluated_length_seconds=None):
        ref_roll = event_list_to_event_roll(reference_event_list, self.event_label_list, self.time_resolution)
        est_roll = event_list_to_event_roll(estimated_event_list, self.event_label_list, self.time_resolution)
        if evaluated_length_seconds is None:
            evaluated_length_seconds = max(max_offset(reference_event_list), max_offset(estimated_event_list))
        segments = int(math.ceil(evaluated_length_seconds * 1 / float(self.time_resolution)))
        self.evaluated_length_seconds += evaluated_length_seconds
        self.evaluated_files += 1
        ref_roll, est_roll = match_event_roll_lengths(ref_roll, est_roll, segments)
        for s in range(ref_roll.shape[0]):
            ann, sys_ = ref_roll[s, :], est_roll[s, :]
            Ntp = sum(sys_ + ann > 1)
            Ntn = sum(sys_ + ann == 0)
            Nfp = sum(sys_ - ann > 0)
            Nfn = sum(ann - sys_ > 0)
            Nref = sum(ann)
            Nsys = sum(sys_)
            o = self.overall
            o["Ntp"] += Ntp; o["Ntn"] += Ntn; o["Nfp"] += Nfp; o["Nfn"] += Nfn; o["Nref"] += Nref; o["Nsys"] += Nsys
            o["S"] += min(Nref, Nsys) - Ntp
            o["D"] += max(0, Nref - Nsys)
            o["I"] += max(0, Nsys - Nref)
            o["ER"] += max(Nref, Nsys) - Ntp
        for k, label in enumerate(self.event_label_list):
            ann, sys_ = ref_roll[:, k], est_roll[:, k]
            c = self.class_wise[label]
            c["Ntp"] += sum(sys_ + ann > 1)
            c["Ntn"] += sum(sys_ + ann == 0)
            c["Nfp"] += sum(sys_ - ann > 0)
            c["Nfn"] += sum(ann - sys_ > 0)
            c["Nref"] += sum(ann)
            c["Nsys"] += sum(sys_)
        return self

    def results_overall_metrics(self):
        o = self.overall
        p, r = precision(o["Ntp"], o["Nsys"]), recall(o["Ntp"], o["Nref"])
        s, d, i = rate(o["S"], o["Nref"]), rate(o["D"], o["Nref"]), rate(o["I"], o["Nref"])
        sens, spec = sensitivity(o["Ntp"], o["Nfn"]), specificity(o["Ntn"], o["Nfp"])
        return {
            "f_measure": {"f_measure": f_measure(p, r), "precision": p, "recall": r},
            "error_rate": {"error_rate": float(s + d + i), "substitution_rate": s, "deletion_rate": d, "insertion_rate": i},
            "accuracy": {"accuracy": accuracy(o["Ntp"], o["Ntn"], o["Nfp"], o["Nfn"]), "sensitivity": sens,
                         "specificity": spec, "balanced_accuracy": balanced_accuracy(sens, spec)},
        }

    def results_class_wise_metrics(self):
        out = {}
        for label in self.event_label_list:
            c = self.class_wise[label]
            p, r = precision(c["Ntp"], c["Nsys"]), recall(c["Ntp"], c["Nref"])
            d, i = rate(c["Nfn"], c["Nref"]), rate(c["Nfp"], c["Nref"])
            sens, spec = sensitivity(c["Ntp"], c["Nfn"]), specificity(c["Ntn"], c["Nfp"])
            out[label] = {
                "f_measure": {"f_measure": f_measure(p, r), "precision": p, "recall": r},
                "error_rate": {"error_rate": float(d + i), "deletion_rate": d, "insertion_rate": i},
                "count": {"Nref": c["Nref"], "Nsys": c["Nsys"]},
                "accuracy": {"accuracy": accuracy(c["Ntp"], c["Ntn"], c["Nfp"], c["Nfn"]), "sensitivity": sens,
                             "specificity": spec, "balanced_accuracy": balanced_accuracy(sens, spec)},
            }
        return out


# ----------------------------------------------------------------------------
# sed_eval/sound_event.py: EventBasedMetrics (event_matching_type 'optimal', the default)
# ----------------------------------------------------------------------------

class EventBasedMetrics:
    def __init__(self, event_label_list, evaluate_onset=True, evaluate_offset=True, t_collar=0.200, percentage_of_length=0.5):
        self.event_label_list = list(event_label_list)
        self.evaluate_onset, self.evaluate_offset = evaluate_onset, evaluate_offset
        self.t_collar, self.percentage_of_length = t_collar, percentage_of_length
        self.evaluated_length = 0.0
        self.evaluated_files = 0
        self.overall = dict(Nref=0.0, Nsys=0.0, Nsubs=0.0, Ntp=0.0, Nfp=0.0, Nfn=0.0)
        self.class_wise = {c: dict(Nref=0.0, Nsys=0.0, Ntp=0.0, Ntn=0.0, Nfp=0.0, Nfn=0.0) for c in self.event_label_list}

    def validate_onset(self, ref, est):
        return math.fabs(ref["event_onset"] - est["event_onset"]) <= self.t_collar

    def validate_offset(self, ref, est):
        length = ref["event_offset"] - ref["event_onset"]
        return math.fabs(ref["event_offset"] - est["event_offset"]) <= max(self.t_collar, self.percentage_of_length * length)

    def _timing(self, ref, est):
        return ((not self.evaluate_onset) or self.validate_onset(ref, est)) and \
               ((not self.evaluate_offset) or self.validate_offset(ref, est))

    def _match(self, refs, ests, with_label):
        hit = np.zeros((len(refs), len(ests)), dtype=bool)
        for j, r in enumerate(refs):
            for i, e in enumerate(ests):
                hit[j, i] = ((not with_label) or r["event_label"] == e["event_label"]) and self._timing(r, e)
        G = {}
        for ref_i, est_i in zip(*np.where(hit)):
            G.setdefault(int(est_i), []).append(int(ref_i))
        return sorted(bipartite_match(G).items())

    def evaluate(self, reference_event_list, estimated_event_list):
        refs, ests = list(reference_event_list), list(estimated_event_list)
        self.evaluated_length += max_offset(refs)
        self.evaluated_files += 1
        Nref, Nsys = len(refs), len(ests)
        matching = self._match(refs, ests, with_label=True)
        ref_ok, sys_ok = np.zeros(Nref, dtype=bool), np.zeros(Nsys, dtype=bool)
        for j, i in matching:
            ref_ok[j] = True
            sys_ok[i] = True
        Ntp = len(matching)
        Nsubs = 0
        counted = np.zeros(Nsys, dtype=bool)
        for j in np.nonzero(~ref_ok)[0]:
            for i in np.nonzero(~sys_ok)[0]:
                if not counted[i] and self._timing(refs[j], ests[i]):
                    counted[i] = True
                    Nsubs += 1
                    break
        o = self.overall
        o["Nref"] += Nref; o["Nsys"] += Nsys; o["Ntp"] += Ntp; o["Nsubs"] += Nsubs
        o["Nfp"] += Nsys - Ntp - Nsubs
        o["Nfn"] += Nref - Ntp - Nsubs
        for label in self.event_label_list:
            cr = [e for e in refs if e["event_label"] == label]
            ce = [e for e in ests if e["event_label"] == label]
            tp = len(self._match(cr, ce, with_label=False))
            c = self.class_wise[label]
            c["Nref"] += len(cr); c["Nsys"] += len(ce); c["Ntp"] += tp
            c["Nfp"] += len(ce) - tp
            c["Nfn"] += len(cr) - tp
        return self

    def results_overall_metrics(self):
        o = self.overall
        p, r = precision(o["Ntp"], o["Nsys"]), recall(o["Ntp"], o["Nref"])
        s, d, i = rate(o["Nsubs"], o["Nref"]), rate(o["Nfn"], o["Nref"]), rate(o["Nfp"], o["Nref"])
        return {
            "f_measure": {"f_measure": f_measure(p, r), "precision": p, "recall": r},
            "error_rate": {"error_rate": float(s + d + i), "substitution_rate": s, "deletion_rate": d, "insertion_rate": i},
        }

    def results_class_wise_metrics(self):
        out = {}
        for label in self.event_label_list:
            c = self.class_wise[label]
            p, r = precision(c["Ntp"], c["Nsys"]), recall(c["Ntp"], c["Nref"])
            d, i = rate(c["Nfn"], c["Nref"]), rate(c["Nfp"], c["Nref"])
            out[label] = {
                "f_measure": {"f_measure": f_measure(p, r), "precision": p, "recall": r},
                "error_rate": {"error_rate": float(d + i), "deletion_rate": d, "insertion_rate": i},
                "count": {"Nref": c["Nref"], "Nsys": c["Nsys"]},
            }
        return out


def class_wise_average(class_wise):
    """results_class_wise_average_metrics: the unweighted mean over the classes of every class-wise figure."""
    out = {}
    for group in next(iter(class_wise.values())):
        if group == "count":
            continue
        out[group] = {k: float(np.nanmean([class_wise[c][group][k] for c in class_wise])) for k in next(iter(class_wise.values()))[group]}
    return out
