"""DCASE sound-event-detection metrics behind the decoders: segment-based and event-based F-measure / error rate.

Same class names, constructor arguments, `evaluate(reference_event_list=..., estimated_event_list=...)` calls, result
dictionaries and printed report as the `sed_eval.sound_event` classes the reference scores its event lists with
(main.py:29-48, requirements.txt:9 sed_eval==0.2.1), so `calculate_metrics` below is that function with the import
changed.  Event lists are the decoders' own output: lists of dicts with 'event_onset', 'event_offset', 'event_label'.

This is host code, like the formatting of the decoders' records into dicts: one reference event and a few dozen
estimated events per file.  It differs from a file-at-a-time loop in that `evaluate_files` scores a whole
{file: events} dictionary in one pass: every event of every file is rasterised into one (segments of all files, classes)
activity array through a difference array and a running sum, and the per-segment and per-class counts are array
reductions; the event-based matching forms its hit matrices by broadcasting and only runs the bipartite matching for
files where an event has more than one candidate.
"""
from __future__ import annotations

import math
from typing import Dict, List, Mapping, Sequence

import numpy as np

_EPS = float(np.spacing(1))


def _ratio(num, den) -> float:
    return float(num / float(den + _EPS))


def _f_measure(p: float, r: float) -> float:
    if p == 0 and r == 0:
        return 0.0
    return float(2.0 * p * r / (p + r))


def _times(events: Sequence[Mapping]) -> tuple:
    on = np.array([e["event_onset"] for e in events], dtype=np.float64)
    off = np.array([e["event_offset"] for e in events], dtype=np.float64)
    return on, off


def _max_offset(events: Sequence[Mapping]):
    m = 0
    for e in events:
        if e["event_offset"] > m:
            m = e["event_offset"]
    return m


def _class_average(class_wise: Dict[str, dict]) -> dict:
    out: dict = {}
    first = next(iter(class_wise.values()))
    for group, fields in first.items():
        if group == "count":
            continue
        out[group] = {k: float(np.nanmean([class_wise[c][group][k] for c in class_wise])) for k in fields}
    return out


# ---------------------------------------------------------------------------------------------------------------------
# report text (the layout sed_eval prints; compared with the reports of notebooks/evaluation_test.ipynb in the tests)
# ---------------------------------------------------------------------------------------------------------------------

def _line(name: str, value, unit: str = "", indent: int = 2) -> str:
    pad = " " * indent
    if isinstance(value, float):
        value = f"{value:.2f}"
    return f"{pad}{name:<{36 - indent}s}: {value} {unit}\n"


def _pct(v: float) -> float:
    return v * 100.0


def _report_overall(res: dict, title: str, with_substitution: bool) -> str:
    out = f"  {title}\n  ======================================\n  F-measure\n"
    out += _line("F-measure (F1)", _pct(res["f_measure"]["f_measure"]), "%", 4)
    out += _line("Precision", _pct(res["f_measure"]["precision"]), "%", 4)
    out += _line("Recall", _pct(res["f_measure"]["recall"]), "%", 4)
    out += "  Error rate\n"
    out += _line("Error rate (ER)", res["error_rate"]["error_rate"], "", 4)
    if with_substitution:
        out += _line("Substitution rate", res["error_rate"]["substitution_rate"], "", 4)
    out += _line("Deletion rate", res["error_rate"]["deletion_rate"], "", 4)
    out += _line("Insertion rate", res["error_rate"]["insertion_rate"], "", 4)
    if "accuracy" in res:
        out += "  Accuracy\n"
        out += _line("Sensitivity", _pct(res["accuracy"]["sensitivity"]), "%", 4)
        out += _line("Specificity", _pct(res["accuracy"]["specificity"]), "%", 4)
        out += _line("Balanced accuracy", _pct(res["accuracy"]["balanced_accuracy"]), "%", 4)
        out += _line("Accuracy", _pct(res["accuracy"]["accuracy"]), "%", 4)
    return out


def _report_class_table(class_wise: Dict[str, dict]) -> str:
    acc = "accuracy" in next(iter(class_wise.values()))
    head = "    Event label  | Nref    Nsys  | F        Pre      Rec    | ER       Del      Ins    |"
    rule = "    ------------ | -----   ----- | ------   ------   ------ | ------   ------   ------ |"
    if acc:
        head += " Sens     Spec     Bacc     Acc     "
        rule += " ------   ------   ------   ------  "
    out = "  Class-wise metrics\n  ======================================\n" + head + "\n" + rule + "\n"
    for label, r in class_wise.items():
        def pc(v):
            return f"{_pct(v):.1f}%"
        row = f"    {label[:12]:<12s} | {int(r['count']['Nref']):<7d} {int(r['count']['Nsys']):<5d} | "
        row += f"{pc(r['f_measure']['f_measure']):<8s} {pc(r['f_measure']['precision']):<8s} {pc(r['f_measure']['recall']):<6s} | "
        row += f"{r['error_rate']['error_rate']:<8.2f} {r['error_rate']['deletion_rate']:<8.2f} {r['error_rate']['insertion_rate']:<6.2f} |"
        if acc:
            a = r["accuracy"]
            row += f" {pc(a['sensitivity']):<8s} {pc(a['specificity']):<8s} {pc(a['balanced_accuracy']):<8s} {pc(a['accuracy']):<7s} "
        out += row + "\n"
    return out


# ---------------------------------------------------------------------------------------------------------------------
# segment-based metrics
# ---------------------------------------------------------------------------------------------------------------------

class SegmentBasedMetrics:
    """sed_eval.sound_event.SegmentBasedMetrics: activity per (segment of `time_resolution` seconds, class)."""

    def __init__(self, event_label_list: Sequence[str], time_resolution: float = 1.0):
        self.event_label_list = list(event_label_list)
        self._index = {c: k for k, c in enumerate(self.event_label_list)}
        self.time_resolution = time_resolution
        self.reset()

    def reset(self):
        self.evaluated_length_seconds = 0.0
        self.evaluated_files = 0
        self.overall = dict(Ntp=0.0, Ntn=0.0, Nfp=0.0, Nfn=0.0, Nref=0.0, Nsys=0.0, ER=0.0, S=0.0, D=0.0, I=0.0)
        self.class_wise = {c: dict(Ntp=0.0, Ntn=0.0, Nfp=0.0, Nfn=0.0, Nref=0.0, Nsys=0.0) for c in self.event_label_list}
        return self

    def _roll(self, lists: Sequence[Sequence[Mapping]], base: np.ndarray, nseg: np.ndarray) -> np.ndarray:
        """Activity (segments of all files, classes) of one event list per file: +1 at the first segment of an event, -1
        behind its last one, running sum > 0.  An event is cut at its file's evaluated length; one whose offset precedes its
        onset (the detector decode can emit such a pair) covers nothing, as the slice of the event roll it stands for."""
        total = int(base[-1] + nseg[-1]) if len(nseg) else 0
        delta = np.zeros((total + 1, len(self.event_label_list)), dtype=np.int32)
        fidx, cls, on, off = [], [], [], []
        for f, events in enumerate(lists):
            for e in events:
                fidx.append(f)
                cls.append(self._index[e["event_label"]])      # KeyError for a label outside the list, like list.index
                on.append(e["event_onset"])
                off.append(e["event_offset"])
        if fidx:
            fidx = np.asarray(fidx)
            cls = np.asarray(cls)
            res = self.time_resolution
            first = np.floor(np.asarray(on, dtype=np.float64) * 1 / res).astype(np.int64)
            last = np.ceil(np.asarray(off, dtype=np.float64) * 1 / res).astype(np.int64)
            lim = nseg[fidx]
            first = np.clip(first, 0, lim)
            last = np.clip(np.maximum(last, first), 0, lim)
            np.add.at(delta, (base[fidx] + first, cls), 1)
            np.add.at(delta, (base[fidx] + last, cls), -1)
        return np.cumsum(delta, axis=0)[:total] > 0

    def evaluate_files(self, reference_event_dict: Mapping[str, Sequence[Mapping]],
                       estimated_event_dict: Mapping[str, Sequence[Mapping]]):
        """Every file of `estimated_event_dict` against its entry of `reference_event_dict`, in one pass."""
        files = list(estimated_event_dict)
        refs = [reference_event_dict[f] for f in files]
        ests = [estimated_event_dict[f] for f in files]
        return self._evaluate_lists(refs, ests, None)

    def evaluate(self, reference_event_list, estimated_event_list, evaluated_length_seconds=None):
        return self._evaluate_lists([reference_event_list], [estimated_event_list], evaluated_length_seconds)

    def _evaluate_lists(self, refs, ests, evaluated_length_seconds):
        if not refs:
            return self
        if evaluated_length_seconds is None:
            lengths = [max(_max_offset(r), _max_offset(e)) for r, e in zip(refs, ests)]
        else:
            lengths = [evaluated_length_seconds] * len(refs)
        nseg = np.array([int(math.ceil(s * 1 / float(self.time_resolution))) for s in lengths], dtype=np.int64)
        base = np.concatenate(([0], np.cumsum(nseg)[:-1]))
        for s in lengths:                                        # summed in file order, as one evaluate() per file does
            self.evaluated_length_seconds += s
        self.evaluated_files += len(refs)
        ann = self._roll(refs, base, nseg)
        sys_ = self._roll(ests, base, nseg)
        tp, tn = ann & sys_, ~ann & ~sys_
        fp, fn = sys_ & ~ann, ann & ~sys_
        ntp, nref, nsys = tp.sum(1), ann.sum(1), sys_.sum(1)    # per segment
        o = self.overall
        o["Ntp"] += float(ntp.sum()); o["Ntn"] += float(tn.sum()); o["Nfp"] += float(fp.sum()); o["Nfn"] += float(fn.sum())
        o["Nref"] += float(nref.sum()); o["Nsys"] += float(nsys.sum())
        o["S"] += float((np.minimum(nref, nsys) - ntp).sum())
        o["D"] += float(np.maximum(0, nref - nsys).sum())
        o["I"] += float(np.maximum(0, nsys - nref).sum())
        o["ER"] += float((np.maximum(nref, nsys) - ntp).sum())
        for k, label in enumerate(self.event_label_list):
            c = self.class_wise[label]
            c["Ntp"] += float(tp[:, k].sum()); c["Ntn"] += float(tn[:, k].sum())
            c["Nfp"] += float(fp[:, k].sum()); c["Nfn"] += float(fn[:, k].sum())
            c["Nref"] += float(ann[:, k].sum()); c["Nsys"] += float(sys_[:, k].sum())
        return self

    @staticmethod
    def _accuracy(c) -> dict:
        sens = float(c["Ntp"] / (c["Ntp"] + c["Nfn"] + _EPS))
        spec = float(c["Ntn"] / (c["Ntn"] + c["Nfp"] + _EPS))
        return {"accuracy": float((c["Ntp"] + c["Ntn"]) / (c["Ntp"] + c["Ntn"] + c["Nfn"] + c["Nfp"] + _EPS)),
                "sensitivity": sens, "specificity": spec, "balanced_accuracy": float(0.5 * sens + 0.5 * spec)}

    def results_overall_metrics(self) -> dict:
        o = self.overall
        p, r = _ratio(o["Ntp"], o["Nsys"]), _ratio(o["Ntp"], o["Nref"])
        s, d, i = _ratio(o["S"], o["Nref"]), _ratio(o["D"], o["Nref"]), _ratio(o["I"], o["Nref"])
        return {"f_measure": {"f_measure": _f_measure(p, r), "precision": p, "recall": r},
                "error_rate": {"error_rate": float(s + d + i), "substitution_rate": s, "deletion_rate": d, "insertion_rate": i},
                "accuracy": self._accuracy(o)}

    def results_class_wise_metrics(self) -> dict:
        out = {}
        for label in self.event_label_list:
            c = self.class_wise[label]
            p, r = _ratio(c["Ntp"], c["Nsys"]), _ratio(c["Ntp"], c["Nref"])
            d, i = _ratio(c["Nfn"], c["Nref"]), _ratio(c["Nfp"], c["Nref"])
            out[label] = {"f_measure": {"f_measure": _f_measure(p, r), "precision": p, "recall": r},
                          "error_rate": {"error_rate": float(d + i), "deletion_rate": d, "insertion_rate": i},
                          "count": {"Nref": c["Nref"], "Nsys": c["Nsys"]},
                          "accuracy": self._accuracy(c)}
        return out

    def results_class_wise_average_metrics(self) -> dict:
        return _class_average(self.results_class_wise_metrics())

    def results(self) -> dict:
        return {"overall": self.results_overall_metrics(), "class_wise": self.results_class_wise_metrics(),
                "class_wise_average": self.results_class_wise_average_metrics()}

    def __str__(self) -> str:
        out = "Segment based metrics\n========================================\n"
        out += _line("Evaluated length", float(self.evaluated_length_seconds), "sec")
        out += _line("Evaluated files", self.evaluated_files)
        out += _line("Segment length", float(self.time_resolution), "sec")
        out += "\n" + _report_overall(self.results_overall_metrics(), "Overall metrics (micro-average)", True)
        out += "\n" + _report_overall(self.results_class_wise_average_metrics(), "Class-wise average metrics (macro-average)", False)
        out += "  \n\n" + _report_class_table(self.results_class_wise_metrics()) + "\n"
        return out


# ---------------------------------------------------------------------------------------------------------------------
# event-based metrics
# ---------------------------------------------------------------------------------------------------------------------

def _bipartite_match(adj: Dict[int, List[int]], n_ref: int) -> List[int]:
    """Maximum matching of estimated events (keys of `adj`, in the order they are to be tried) to reference events (the
    lists, ascending): Hopcroft-Karp in the form sed_eval uses (D. Eppstein's recipe: a greedy start, then rounds of a
    layered breadth-first search from the free estimated events and a depth-first search back through the layers).  Which
    of several maximum matchings comes out decides the events left over for the substitution count, so the order of every
    loop is that recipe's.  Returns the estimated event of every reference event, -1 for none."""
    match = [-1] * n_ref
    for u, vs in adj.items():
        for v in vs:
            if match[v] < 0:
                match[v] = u
                break
    free = object()
    while True:
        preds: Dict[int, List[int]] = {}
        unmatched: List[int] = []
        taken = {u for u in match if u >= 0}
        pred: Dict[int, object] = {u: free for u in adj if u not in taken}
        layer = list(pred)
        while layer and not unmatched:
            new_layer: Dict[int, List[int]] = {}
            for u in layer:
                for v in adj[u]:
                    if v not in preds:
                        new_layer.setdefault(v, []).append(u)
            layer = []
            for v, us in new_layer.items():
                preds[v] = us
                if match[v] >= 0:
                    layer.append(match[v])
                    pred[match[v]] = v
                else:
                    unmatched.append(v)
        if not unmatched:
            return match

        def back(v) -> bool:
            us = preds.pop(v, None)
            if us is None:
                return False
            for u in us:
                if u in pred:
                    pu = pred.pop(u)
                    if pu is free or back(pu):
                        match[v] = u
                        return True
            return False

        for v in unmatched:
            back(v)


class EventBasedMetrics:
    """sed_eval.sound_event.EventBasedMetrics with the optimal (bipartite) event matching."""

    def __init__(self, event_label_list: Sequence[str], evaluate_onset: bool = True, evaluate_offset: bool = True,
                 t_collar: float = 0.200, percentage_of_length: float = 0.5, event_matching_type: str = "optimal"):
        if event_matching_type != "optimal":
            raise ValueError("only the optimal event matching (sed_eval's default, the one the reference uses) is built")
        self.event_label_list = list(event_label_list)
        self.evaluate_onset, self.evaluate_offset = evaluate_onset, evaluate_offset
        self.t_collar, self.percentage_of_length = t_collar, percentage_of_length
        self.reset()

    def reset(self):
        self.evaluated_length = 0.0
        self.evaluated_files = 0
        self.overall = dict(Nref=0.0, Nsys=0.0, Nsubs=0.0, Ntp=0.0, Nfp=0.0, Nfn=0.0)
        self.class_wise = {c: dict(Nref=0.0, Nsys=0.0, Ntp=0.0, Ntn=0.0, Nfp=0.0, Nfn=0.0) for c in self.event_label_list}
        return self

    def _timing(self, ron, roff, eon, eoff) -> np.ndarray:
        """(reference, estimated) matrix: onsets within the collar, offsets within max(collar, share of the reference length)."""
        hit = np.ones((len(ron), len(eon)), dtype=bool)
        if self.evaluate_onset:
            hit &= np.abs(ron[:, None] - eon[None, :]) <= self.t_collar
        if self.evaluate_offset:
            hit &= np.abs(roff[:, None] - eoff[None, :]) <= np.maximum(self.t_collar, self.percentage_of_length * (roff - ron))[:, None]
        return hit

    @staticmethod
    def _matching(hit: np.ndarray) -> List[int]:
        """Estimated event matched to every reference event (-1: none)."""
        n_ref = hit.shape[0]
        if hit.size == 0 or not hit.any():
            return [-1] * n_ref
        if hit.sum(0).max() <= 1 and hit.sum(1).max() <= 1:     # nobody has a choice: the hits are the matching
            m = np.full(n_ref, -1, dtype=np.int64)
            j, i = np.nonzero(hit)
            m[j] = i
            return m.tolist()
        adj: Dict[int, List[int]] = {}
        for j, i in zip(*np.nonzero(hit)):                       # row-major: estimated events in order of first hit
            adj.setdefault(int(i), []).append(int(j))
        return _bipartite_match(adj, n_ref)

    def evaluate_files(self, reference_event_dict, estimated_event_dict):
        for f, est in estimated_event_dict.items():
            self.evaluate(reference_event_dict[f], est)
        return self

    def evaluate(self, reference_event_list, estimated_event_list):
        refs, ests = list(reference_event_list), list(estimated_event_list)
        self.evaluated_length += _max_offset(refs)
        self.evaluated_files += 1
        n_ref, n_sys = len(refs), len(ests)
        ron, roff = _times(refs)
        eon, eoff = _times(ests)
        ids = {c: k for k, c in enumerate(self.event_label_list)}   # labels outside the list count overall, not per class
        rlab = np.array([ids.setdefault(e["event_label"], len(ids)) for e in refs], dtype=np.int64)
        elab = np.array([ids.setdefault(e["event_label"], len(ids)) for e in ests], dtype=np.int64)
        timing = self._timing(ron, roff, eon, eoff)
        same = rlab[:, None] == elab[None, :]
        match = np.asarray(self._matching(timing & same), dtype=np.int64)
        n_tp = int((match >= 0).sum())
        # substitutions: left-over pairs whose times agree although their labels do not, first come first served
        n_subs = 0
        ref_left = np.nonzero(match < 0)[0]
        sys_free = np.ones(n_sys, dtype=bool)
        sys_free[match[match >= 0]] = False
        for j in ref_left:
            cand = np.nonzero(sys_free & timing[j])[0]
            if len(cand):
                sys_free[cand[0]] = False
                n_subs += 1
        o = self.overall
        o["Nref"] += n_ref; o["Nsys"] += n_sys; o["Ntp"] += n_tp; o["Nsubs"] += n_subs
        o["Nfp"] += n_sys - n_tp - n_subs
        o["Nfn"] += n_ref - n_tp - n_subs
        for k, label in enumerate(self.event_label_list):
            rj, ei = np.nonzero(rlab == k)[0], np.nonzero(elab == k)[0]
            tp = 0
            if len(rj) and len(ei):
                tp = sum(1 for m in self._matching(timing[np.ix_(rj, ei)]) if m >= 0)
            c = self.class_wise[label]
            c["Nref"] += len(rj); c["Nsys"] += len(ei); c["Ntp"] += tp
            c["Nfp"] += len(ei) - tp
            c["Nfn"] += len(rj) - tp
        return self

    def results_overall_metrics(self) -> dict:
        o = self.overall
        p, r = _ratio(o["Ntp"], o["Nsys"]), _ratio(o["Ntp"], o["Nref"])
        s, d, i = _ratio(o["Nsubs"], o["Nref"]), _ratio(o["Nfn"], o["Nref"]), _ratio(o["Nfp"], o["Nref"])
        return {"f_measure": {"f_measure": _f_measure(p, r), "precision": p, "recall": r},
                "error_rate": {"error_rate": float(s + d + i), "substitution_rate": s, "deletion_rate": d, "insertion_rate": i}}

    def results_class_wise_metrics(self) -> dict:
        out = {}
        for label in self.event_label_list:
            c = self.class_wise[label]
            p, r = _ratio(c["Ntp"], c["Nsys"]), _ratio(c["Ntp"], c["Nref"])
            d, i = _ratio(c["Nfn"], c["Nref"]), _ratio(c["Nfp"], c["Nref"])
            out[label] = {"f_measure": {"f_measure": _f_measure(p, r), "precision": p, "recall": r},
                          "error_rate": {"error_rate": float(d + i), "deletion_rate": d, "insertion_rate": i},
                          "count": {"Nref": c["Nref"], "Nsys": c["Nsys"]}}
        return out

    def results_class_wise_average_metrics(self) -> dict:
        return _class_average(self.results_class_wise_metrics())

    def results(self) -> dict:
        return {"overall": self.results_overall_metrics(), "class_wise": self.results_class_wise_metrics(),
                "class_wise_average": self.results_class_wise_average_metrics()}

    def __str__(self) -> str:
        mode = "onset-offset" if self.evaluate_onset and self.evaluate_offset else "onset only" if self.evaluate_onset else "offset only"
        out = f"Event based metrics ({mode})\n========================================\n"
        out += _line("Evaluated length", float(self.evaluated_length), "sec")
        out += _line("Evaluated files", self.evaluated_files)
        out += _line("Evaluate onset", self.evaluate_onset)
        out += _line("Evaluate offset", self.evaluate_offset)
        out += _line("T collar", float(self.t_collar * 1000), "ms")
        out += _line("Offset (length)", float(self.percentage_of_length * 100), "%")
        out += "\n" + _report_overall(self.results_overall_metrics(), "Overall metrics (micro-average)", True)
        out += "\n" + _report_overall(self.results_class_wise_average_metrics(), "Class-wise average metrics (macro-average)", False)
        out += "  \n\n" + _report_class_table(self.results_class_wise_metrics()) + "\n"
        return out


def calculate_metrics(pred_event_dict, gt_event_dict, time_resolution: float = 1.0, t_collar: float = 0.25,
                      classes: Sequence[str] = ("cough", "dog_bark", "gun_shot", "siren", "car_horn"), verbose: bool = True):
    """main.py:29-48: both metric objects over every file of `pred_event_dict`, printed; also returned."""
    event_based = EventBasedMetrics(list(classes), t_collar=t_collar).evaluate_files(gt_event_dict, pred_event_dict)
    segment_based = SegmentBasedMetrics(list(classes), time_resolution=time_resolution).evaluate_files(gt_event_dict, pred_event_dict)
    if verbose:
        print(event_based)
        print(segment_based)
    return event_based, segment_based
