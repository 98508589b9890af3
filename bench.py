#!/usr/bin/env python3
"""Benchmark of the log-mel front-end hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): HTS-AT front-end, 32 kHz, n_fft 1024, hop 320, 64 mel, 256 clips of 10 s
per GPU, fp32 PCM resident in HBM.  One step = one pass of the hot path over the rank's batch: the fused
log-mel kernel, then the detector-side tail of the same 256 streams, sed_amd.sharding.decode_and_exchange (median-3 +
threshold + run extraction + 0.6 s gap merge on synthetic 20-frame probabilities, the (stream, class, onset, offset)
records, and with more than one rank the all-gather of the fixed-size record buffers, asynchronous and overlapped
with the next step).  Everything in the step stays on the device; formatting events as host dictionaries is outside
it and is timed separately ("decode_with_host_formatting_ms").  Ranks own independent streams (weak scaling).
Rank 0 prints one JSON line.

`--workload yamnet` and `--workload crnn_chain` time the other single-GPU configurations BASELINE.json names (one JSON
line each, N = 1 only): the YAMNet feature front-end on 64 and 512 clips of 10 s at 16 kHz (configs[0]) beside its CPU
formulation, and the CRNN decode chain on 500 files x 862 frames x 5 classes (configs[2]): gaussian smoothing, threshold,
region decode and the 0.3 / 0.6 / 1.0 s gap merges INCLUDING the host formatting of the event dictionaries, beside the
pure-Python formulation of the reference on the host.  The default line is unchanged by them.

`--workload server` times one GPU's share of the two-stage stream server (configs[3]: 4096 concurrent 10 s streams over 8
GPUs = 512 per GPU): YAMNet features of the 512 streams at 16 kHz, the detector decode and the exchange of its event
records, the cut of the detected events out of the resident 32 kHz PCM into the classifier batch, and the HTS-AT log-mel
of that batch.  It takes --gpus N like the headline workload (512 streams per rank, weak scaling).

`--gpus N` without a torchrun environment starts the N ranks itself (python -m torch.distributed.run as a child
process, before this process touches the GPU) and relays rank 0's line.
"""
import argparse
import contextlib
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sound-event-detection_amd"))

SR, CLIP_S, B_PER_GPU = 32000, 10, 256
EVENT_CAP = 16              # merged events per stream the exchange keeps (a 20-frame trace has at most 10)
HBM_PEAK_GBS = 8000.0
PREWARM_S = float(os.environ.get("SED_BENCH_PREWARM_S", "0.3"))   # (the environment override is for experiments)


def synth_pcm(n_clips, n_samples, device, seed, sr=SR):
    """Seeded synthetic audio generated on the device: noise + two sine bursts per clip + 0.5 s of silence."""
    g = torch.Generator(device=device).manual_seed(seed)
    x = 0.1 * torch.randn((n_clips, n_samples), generator=g, device=device)
    t = torch.arange(n_samples, device=device) / sr
    f = 200.0 + 5800.0 * torch.rand((n_clips, 2), generator=g, device=device)
    a = 0.05 + 0.45 * torch.rand((n_clips, 2), generator=g, device=device)
    s0 = torch.rand((n_clips, 2), generator=g, device=device) * (CLIP_S - 3.0)
    for j in range(2):
        m = (t[None] >= s0[:, j:j + 1]) & (t[None] < s0[:, j:j + 1] + 2.0)
        x += m * a[:, j:j + 1] * torch.sin(2 * torch.pi * f[:, j:j + 1] * t[None])
    sil = (torch.rand((n_clips, 1), generator=g, device=device) * (CLIP_S - 0.5) * sr).long()
    idx = torch.arange(n_samples, device=device)[None]
    x.masked_fill_((idx >= sil) & (idx < sil + sr // 2), 0.0)
    return x.clamp_(-1.0, 1.0).contiguous()


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            return next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        return ""


def measure_h2d(x):
    """Pinned host -> device copy of the batch, fp32 and int16, outside the timed region (the step's inputs are resident
    in HBM; a caller that hands over host PCM pays this on top).  Best of three copies each, HIP events."""
    out = {}
    for name, dtype in (("f32", torch.float32), ("i16", torch.int16)):
        host = torch.empty(x.shape, dtype=dtype, pin_memory=True)
        host.zero_()
        dst = torch.empty(x.shape, dtype=dtype, device=x.device)
        best = None
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dst.copy_(host, non_blocking=True)
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1)
            best = ms if best is None else min(best, ms)
        out[name] = best
        out[name + "_GBps"] = host.numel() * host.element_size() / (best * 1e-3) / 1e9
        del host, dst
    out["note"] = "pinned host -> device copy of the 256-clip batch, best of 4, not part of any timed step"
    return out


def time_kernel(fn, steps, warmup):
    """Mean and percentiles of one launch timed with HIP events on the current stream, after `warmup` untimed calls."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return sum(ms) / len(ms), [ms[int(q * (len(ms) - 1))] for q in (0.1, 0.5, 0.9)], wall


def recorded_traffic(key):
    """HBM bytes per launch from the PMC passes kept under profiles/ (collected with this kernel, not in this run), and
    whether the kernel sources still are the ones that library was built from."""
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        ent = json.load(open(tpath)).get(key, {})
        sys.path.insert(0, os.path.join(ROOT, "profiles", "tools"))
        from update_traffic import kernel_source_fingerprint
        return ent.get("hbm_bytes_per_launch"), ent.get("kernel_source_fingerprint") == kernel_source_fingerprint()
    except Exception:
        return None, None


def bench_yamnet(args):
    """BASELINE.json configs[0]: YAMNet features (16 kHz, 400/160/512, 64 HTK mel, ln(x + 0.001)) on 64 clips of 10 s,
    and on the 512-clip shard of the stream server; the CPU line is the TensorFlow formulation of the same features
    (frame, Hann, rFFT 512, magnitude, mel matmul, log) on torch's CPU kernels."""
    import sed_amd
    dev = torch.device("cuda", 0)
    sr, L = 16000, 160000
    fe = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.yamnet(), layout="btm")
    t_pre = time.perf_counter()
    xs = {B: synth_pcm(B, L, dev, 77 + B, sr=sr) for B in (64, 512)}
    while time.perf_counter() - t_pre < PREWARM_S:
        fe(xs[512])
        torch.cuda.synchronize()
    rows = {}
    for B, x in xs.items():
        T = fe.num_frames(L, dev)
        kern_ms, pcts, wall = time_kernel(lambda: fe(x), args.steps, args.warmup)
        alg = B * (L * 4 + T * 64 * 4)
        rows[B] = {"clips": B, "frames_per_clip": T, "kernel_ms": kern_ms, "kernel_ms_p10_p50_p90": pcts,
                   "ms_per_step": wall / args.steps * 1e3, "audio_hours_per_s": B * 10 / 3600.0 / (wall / args.steps),
                   "algorithmic_bytes_per_launch": alg, "achieved_GBps": alg / (kern_ms * 1e-3) / 1e9}
    big = rows[512]
    traffic, traffic_current = recorded_traffic("flavour_yamnet")
    res = {"metric": "log-mel audio-hours/sec", "value": big["audio_hours_per_s"], "unit": "audio-hours/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": big["ms_per_step"], "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "YAMNet feature front-end 16 kHz 25/10 ms 64 mel, 10 s clips (BASELINE.json configs[0]); value "
                                  "is the 512-clip shard of the stream server, the 64-clip batch of configs[0] is in batches",
                      "batches": [rows[64], rows[512]], "pcm": "f32", "parallelism": "1 GPU"},
           "roofline": {"bound": "hbm", "kernel": "stft_mel_kernel<n_fft=512, magnitude, mel+log>", "achieved": big["achieved_GBps"],
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": big["achieved_GBps"] / HBM_PEAK_GBS, "traffic": traffic,
                        "traffic_source": "profiles/hbm_traffic.json (rocprofv3 PMC passes of kbench -c yamnet, not this run)",
                        "traffic_is_of_these_kernel_sources": traffic_current,
                        "algorithmic_bytes_per_launch": big["algorithmic_bytes_per_launch"], "kernel_ms": big["kernel_ms"]}}
    if not args.no_cpu_baseline:
        from oracle.cpu_baseline import YamnetCpuFrontend, time_cpu_frontend
        sample = xs[64].cpu()
        threads = torch.get_num_threads()
        v, el, n = time_cpu_frontend(YamnetCpuFrontend(), sample, sr, budget_s=10.0)
        torch.set_num_threads(1)
        v1, el1, n1 = time_cpu_frontend(YamnetCpuFrontend(), sample[:8], sr, budget_s=5.0)
        torch.set_num_threads(threads)
        res["cpu_baseline"] = {"value": v, "unit": "audio-hours/s", "cores": threads, "kind": "port",
                               "sample": f"{n} passes over the 64 clips of configs[0] (10 s each), {el:.1f} s of CPU work, TensorFlow "
                                         "formulation (frame, Hann, rFFT 512, magnitude, mel matmul, log) on torch CPU",
                               "single_thread_value": v1, "host_cpus": os.cpu_count(), "cpu_model": cpu_model_name()}
    return res


def crnn_probabilities(n_files, n_frames, n_classes, seed):
    """Sigmoid of smoothed noise with about 18 events per file, the density of the reference's crnn_results.json."""
    import numpy as np
    rng = np.random.default_rng(seed)
    z = rng.standard_normal((n_files * n_classes, n_frames + 140)).astype(np.float32)
    k = np.hanning(141).astype(np.float32)
    z = np.stack([np.convolve(r, k / np.sqrt((k * k).sum()), mode="valid") for r in z])
    p = 1.0 / (1.0 + np.exp(-3.0 * (z - 0.45)))
    return np.ascontiguousarray(p.astype(np.float32).reshape(n_files, n_classes, n_frames).transpose(0, 2, 1))


def bench_crnn_chain(args):
    """BASELINE.json configs[2], the part behind the CRNN: sigmoid outputs (500 files, 862 frames, 5 classes) ->
    gaussian smoothing + threshold + region decode (src/models/crnn/crnn.py:418-425, 454-459) -> gap merge at 0.3 / 0.6 /
    1.0 s (src/utils/merge_events.py), with the event dictionaries formed on the host as the reference returns them.
    One step = decode_probabilities + merge_close_events_batch at the three gaps (events read and grouped once); the 8776-event result file of the
    reference (tests/golden/crnn_kat.npz) is merged at the three gaps beside it."""
    import numpy as np
    import sed_amd
    from sed_amd import decode as D
    dev = torch.device("cuda", 0)
    S, T, C, fps = 500, 862, 5, 44100 / 1024
    labels = {0: "car_horn", 1: "cough", 2: "dog_bark", 3: "siren", 4: "gun_shot"}
    files = [f"soundscape_{i:04d}.wav" for i in range(S)]
    probs_host = crnn_probabilities(S, T, C, 418)
    probs = torch.from_numpy(probs_host).to(dev)
    gaps = (0.3, 0.6, 1.0)
    table = torch.tensor([round(np.int64(k) * (1.0 / fps), 3) for k in range(T + 2)], dtype=torch.float64, device=dev)

    def chain():
        raw = D.decode_probabilities(probs, files, labels, fps)
        return raw, D.merge_close_events_batch(raw, list(gaps))

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, out

    chain_ms, (raw, merged) = timed(chain, args.steps, args.warmup)
    decode_ms, _ = timed(lambda: D.decode_probabilities(probs, files, labels, fps), args.steps, 1)
    merge_ms, _ = timed(lambda: D.merge_close_events_batch(raw, 0.6), args.steps, 1)
    records_ms, rec = timed(lambda: [sed_amd.sharding.decode_and_exchange(probs, table, g, cap=64, smoothing="gauss") for g in gaps],
                            args.steps, 1)
    kern_ms, pcts, _ = time_kernel(lambda: D.threshold_runs(probs, 0.5, "gauss"), max(args.steps, 50), 5)
    n_raw = sum(len(v) for v in raw.values())
    n_merged = [sum(len(v) for v in m.values()) for m in merged]
    # the reference's own result file through the merge at the three gaps
    kat_path = os.path.join(ROOT, "tests", "golden", "crnn_kat.npz")
    kat = None
    if os.path.exists(kat_path):
        pack = np.load(kat_path)
        kfiles, klabels = [str(f) for f in pack["files"]], [str(l) for l in pack["labels"]]
        kraw = {f: [] for f in kfiles}
        for fi, li, on, off in zip(pack["raw_fi"], pack["raw_li"], pack["raw_on"], pack["raw_off"]):
            kraw[kfiles[int(fi)]].append({"file": kfiles[int(fi)], "event_onset": float(on), "event_offset": float(off),
                                          "event_label": klabels[int(li)]})
        kat_ms, kout = timed(lambda: D.merge_close_events_batch(kraw, list(gaps)), args.steps, args.warmup)
        kat = {"events": sum(len(v) for v in kraw.values()), "files": len(kfiles), "merged_events": [sum(len(v) for v in m.values()) for m in kout],
               "ms_three_gaps": kat_ms}
    alg = S * T * C * 4
    hours = S * 20.0 / 3600.0
    res = {"metric": "log-mel audio-hours/sec", "value": hours / (chain_ms * 1e-3), "unit": "audio-hours/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": chain_ms, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64 smoothing, f32 compare, int32 frames", "data": "synthetic",
           "config": {"workload": "CRNN decode chain (BASELINE.json configs[2]): 500 files x 862 frames x 5 classes of 20 s audio -> gaussian "
                                  "smoothing, threshold, region decode, gap merge at 0.3 / 0.6 / 1.0 s, host dictionaries included; value = "
                                  "hours of audio whose frame probabilities are decoded per second",
                      "raw_events": n_raw, "merged_events": n_merged, "decode_probabilities_ms": decode_ms,
                      "merge_close_events_batch_ms_one_gap": merge_ms,
                      "device_only_records_three_gaps_ms": records_ms,
                      "device_only_note": "decode_and_exchange x 3: smoothing, threshold, runs, gap merge and (file, class, onset, offset) "
                                          "frame records on the device, no host dictionaries",
                      "reference_result_file": kat, "parallelism": "1 GPU"},
           "roofline": {"bound": "hbm", "kernel": "threshold_runs_kernel<gauss25>", "achieved": alg / (kern_ms * 1e-3) / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "algorithmic_bytes_per_launch": alg, "kernel_ms": kern_ms, "kernel_ms_p10_p50_p90": pcts,
                        "note": "8.6 MB per launch: launch-bound, the chain's time is host formatting"}}
    if not args.no_cpu_baseline:
        from oracle.cpu_baseline import time_crnn_chain_cpu
        v_ms, n, el, same = time_crnn_chain_cpu(probs_host, files, labels, fps, gaps, raw, merged, budget_s=10.0)
        res["cpu_baseline"] = {"value": hours / (v_ms * 1e-3), "unit": "audio-hours/s", "cores": 1, "kind": "port",
                               "ms_per_chain": v_ms, "same_events_as_device_chain": same,
                               "sample": f"{n} passes over the same 500 files, {el:.1f} s of CPU work: scipy gaussian_filter1d + threshold + "
                                         "the reference's per-file, per-class Python decode + its pure-Python merge at the three gaps",
                               "host_cpus": os.cpu_count(), "cpu_model": cpu_model_name()}
        if kat is not None:
            from oracle import decode as od
            t0 = time.perf_counter()
            reps = 0
            while time.perf_counter() - t0 < 2.0:
                for g in gaps:
                    {k: od.merge_close_events(v, g) for k, v in kraw.items()}
                reps += 1
            kat["cpu_ms_three_gaps"] = (time.perf_counter() - t0) / reps * 1e3
    return res


def bench_server(args, world, rank, dev, dist):
    """One GPU's share of BASELINE.json configs[3] (two-stage server, 4096 concurrent 10 s detector streams over 8 GPUs):
    per step and rank, 512 streams go through the detector front-end (YAMNet features, src/models/yamnet_train.py:46-47),
    the detector decode + exchange (yamnet_train.py:55-71, src/utils/merge_events.py), the hand-off of the detected events
    to the classifier (src/utils/cut_audio.py:13-53, on the device) and the classifier front-end (HTS-AT log-mel,
    external/hts_audio_transformer/model/htsat.py:763-764).  The models between the stages are not part of the path: the
    detector probabilities are fixed synthetic inputs, and the event list the hand-off cuts is their decode."""
    import sed_amd
    from sed_amd import decode as D
    S, secs, cap_s = 512, 10, 3.0
    det_fe = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.yamnet(), layout="btm")
    cls_fe = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat())
    x16 = synth_pcm(S, 16000 * secs, dev, 500 + rank, sr=16000)
    x32 = synth_pcm(S, 32000 * secs, dev, 900 + rank, sr=32000)
    g = torch.Generator(device=dev).manual_seed(7 + rank)
    det = torch.rand((S, 20, 1), generator=g, device=dev)
    table = torch.tensor([k * 0.48 for k in range(22)], dtype=torch.float64, device=dev)
    events = []
    for s_idx, (on, off) in enumerate(D.detect_onsets_offsets_batch(det[:, :, 0].contiguous(), 0.5, 0.48, True)):
        events += [(s_idx, float(a), float(min(b, secs))) for a, b in zip(on, off) if b > a]
    import numpy as np
    events = np.asarray(events, dtype=np.float64).reshape(-1, 3)     # (stream, start s, end s) rows, as a server would hold them
    cap = int(cap_s * 32000)
    collective = world > 1 or args.force_collective
    side = torch.cuda.Stream(device=dev) if collective and args.decode_stream != "same" else None
    pending = []
    stage_ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps)]

    def step(i=None, fused=True):
        ev = stage_ev[i] if i is not None else None
        if ev:
            ev[0].record()
        feat = det_fe(x16)
        if ev:
            ev[1].record()
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
            out, work = sed_amd.sharding.decode_and_exchange(det, table, 0.6, cap=EVENT_CAP, first_stream=rank * S, async_op=True,
                                                             host_staging=(args.backend != "nccl"),
                                                             force_collective=args.force_collective)
            if work is not None:
                pending.append((out, work))
                if len(pending) > 4:
                    pending.pop(0)[1].wait()
        if ev:
            ev[2].record()
        if fused == "own":   # every event framed as the file of its own that stage 2 of the reference loads, and its image
            if ev:
                ev[3].record()
            y, lengths, frames = sed_amd.logmel_of_events(cls_fe, x32, 32000, events, cap, per_file=True)
            y = sed_amd.reshape_wav2img(y, 256, 4, frames=frames)
        elif fused:    # the classifier front-end reads the events straight out of the resident recordings
            if ev:
                ev[3].record()
            y, lengths = sed_amd.logmel_of_events(cls_fe, x32, 32000, events, cap)
        else:          # the two steps it replaces: cut into a zero-padded batch, then the front-end on the batch
            clips, lengths = sed_amd.cut_events(x32, 32000, events, cap=cap)
            if ev:
                ev[3].record()
            y = cls_fe(clips)
        if ev:
            ev[4].record()
        return feat, y

    def fence():
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
            while pending:
                pending.pop(0)[1].wait()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(fused):
        for _ in range(args.warmup):
            step(fused=fused)
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = step(i, fused)
        fence()
        el = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        return el, out

    names = ["detector_features_ms", "decode_exchange_ms_on_main_stream", "cut_events_ms", "classifier_features_ms"]
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < PREWARM_S:
        step()
        torch.cuda.synchronize()
    two_step_elapsed, (_, y2) = timed(False)
    two_step_stages = {n: sum(e[k].elapsed_time(e[k + 1]) for e in stage_ev) / len(stage_ev) for k, n in enumerate(names)}
    own_elapsed, (_, img) = timed("own")
    own_stages = {n: sum(e[k].elapsed_time(e[k + 1]) for e in stage_ev) / len(stage_ev) for k, n in enumerate(names)}
    elapsed, (feat, y) = timed(True)
    same_bits = bool(torch.equal(y, y2))
    names = ["detector_features_ms", "decode_exchange_ms_on_main_stream", "cut_events_ms", "classifier_features_ms"]
    stages = {n: sum(e[k].elapsed_time(e[k + 1]) for e in stage_ev) / len(stage_ev) for k, n in enumerate(names)}
    E = len(events)
    T16, T32 = det_fe.num_frames(16000 * secs, dev), cls_fe.num_frames(cap, dev)
    cut_bytes = int(sum(min(int(b * 32000) - int(a * 32000), cap) for _, a, b in events.tolist())) * 4 + E * cap * 4
    alg = S * (16000 * secs * 4 + T16 * 64 * 4) + S * 20 * 4 + cut_bytes + E * (cap * 4 + T32 * 64 * 4)
    det_alg = S * (16000 * secs * 4 + T16 * 64 * 4)
    det_ms = stages["detector_features_ms"]
    return {"metric": "log-mel audio-hours/sec", "value": world * S * secs / 3600.0 * args.steps / elapsed, "unit": "audio-hours/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "two-stage stream server, one GPU's share of BASELINE.json configs[3]: 512 concurrent 10 s streams per GPU -> "
                                   "YAMNet features (16 kHz) -> detector decode + event exchange -> events cut out of the resident 32 kHz PCM "
                                   "(3 s rows) -> HTS-AT log-mel of the events; value = hours of stream audio served per second",
                       "streams_per_gpu": S, "events_per_step": E, "classifier_batch": [E, cap], "stage_ms": stages,
                       "handoff": "fused: the classifier front-end reads the events out of the resident recordings (sed_logmel_forward_segments)",
                       "two_step_handoff": {"ms_per_step": two_step_elapsed / args.steps * 1e3, "stage_ms": two_step_stages,
                                            "note": "cut_events into a zero-padded (events, 96000) batch, then the front-end on the batch",
                                            "same_log_mel_bits_as_fused": same_bits},
                       "per_event_handoff": {"ms_per_step": own_elapsed / args.steps * 1e3, "stage_ms": own_stages,
                                             "image": list(img.shape),
                                             "note": "every event framed as a file of its own (reflect padding at its own end, its own "
                                                     "frame count), then reshape_wav2img from each event's own rows to the 256 x 256 "
                                                     "image: what the reference's batch-of-one second stage computes; "
                                                     "classifier_features_ms includes the image here"},
                       "decode_stream": "side" if side is not None else "same", "forced_one_rank_collective": bool(args.force_collective),
                       "algorithmic_bytes_per_step": alg,
                       "parallelism": f"streams sharded over {world} GPU(s), all-gather of event record buffers"},
            "roofline": {"bound": "hbm", "kernel": "stft_mel_kernel<n_fft=512, magnitude, mel+log> (detector features, the longest launch of the step)",
                         "achieved": det_alg / (det_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": det_alg / (det_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": det_alg, "kernel_ms": det_ms,
                         "whole_step_achieved_GBps": alg / (elapsed / args.steps) / 1e9}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (default: WORLD_SIZE under a launcher, else 1)")
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 1000; 20 for --workload crnn_chain)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 100; 3 for --workload crnn_chain)")
    ap.add_argument("--workload", choices=["htsat", "yamnet", "crnn_chain", "server"], default="htsat",
                    help="htsat: BASELINE.json configs[1], the headline line; yamnet: configs[0]; crnn_chain: configs[2]; "
                         "server: one GPU's share (512 streams) of the two-stage stream server of configs[3], any --gpus")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                    "rehearse the multi-rank control flow on a box with one GPU, together with --one-device)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1 only: run the all-gather on a one-rank RCCL group (measures what the exchange adds to a step)")
    ap.add_argument("--decode-stream", choices=["auto", "side", "same"], default="auto",
                    help="side: the detector decode and the exchange of a step run on a second HIP stream beside the front-end "
                         "kernel (their input does not depend on it); same: everything on one stream; auto (default): side when "
                         "the step holds a collective (N > 1 or --force-collective), where it removes the cross-stream "
                         "dependency of the all-gather from the front-end's stream, else same")
    args = ap.parse_args()
    if args.gpus is None:                      # under a launcher the rank count is the launcher's
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.steps is None:
        args.steps = 20 if args.workload == "crnn_chain" else 1000
    if args.warmup is None:
        args.warmup = 3 if args.workload == "crnn_chain" else 100
    if args.workload in ("yamnet", "crnn_chain"):
        if args.gpus != 1 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
            raise SystemExit(f"--workload {args.workload} is a single-GPU line: run it with --gpus 1")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a ROCm device: the hot path has no CPU fallback")
        torch.cuda.set_device(0)
        res = bench_yamnet(args) if args.workload == "yamnet" else bench_crnn_chain(args)
        print(json.dumps(res), flush=True)
        return

    # Libraries below (RCCL prints a version banner when a communicator is made) write to stdout: from here on file
    # descriptor 1 is stderr, and the one JSON line goes to the saved descriptor of the real stdout.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    def emit(obj):
        real_stdout.write(json.dumps(obj) + "\n")
        real_stdout.flush()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Not under torchrun: start the ranks as a fresh child process tree (nothing here has touched the GPU yet) and
        # pass its output through; rank 0 of the children prints the JSON line.
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, stdout=real_stdout).returncode)

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s): the JSON line would misreport n_gpus")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device: the hot path has no CPU fallback")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    elif args.force_collective:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    import sed_amd
    from sed_amd import decode as D

    if args.workload == "server":
        res = bench_server(args, world, rank, dev, dist)
        if rank == 0:
            emit(res)
        if world > 1 or args.force_collective:
            dist.destroy_process_group()
        return

    trace = os.environ.get("SED_BENCH_TRACE")

    def say(msg):
        if trace:
            print(f"[rank {rank}] {msg}", file=sys.stderr, flush=True)

    L = SR * CLIP_S
    fe = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat())
    x = synth_pcm(B_PER_GPU, L, dev, 1234 + rank)
    g = torch.Generator(device=dev).manual_seed(99 + rank)
    det = torch.rand((B_PER_GPU, 20, 1), generator=g, device=dev)          # detector probabilities, 0.48 s hop
    table = torch.tensor([k * 0.48 for k in range(22)], dtype=torch.float64, device=dev)
    T = fe.num_frames(L, dev)
    alg_bytes = B_PER_GPU * (L * 4 + T * 64 * 4)                            # read PCM once + write log-mel once

    k_events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    pending = []
    if args.decode_stream == "auto":
        args.decode_stream = "side" if (world > 1 or args.force_collective) else "same"
    side = torch.cuda.Stream(device=dev) if args.decode_stream == "side" else None

    def step(i=None):
        if i is not None:
            k_events[i][0].record()
        y = fe(x)
        if i is not None:
            k_events[i][1].record()
        # decode -> records -> all-gather: the same function tests/test_gpu_sharding.py checks against the oracle
        # (the gloo rehearsal gathers host copies: gloo's device-tensor path stalls after a few async gathers)
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
            out, work = sed_amd.sharding.decode_and_exchange(det, table, 0.6, cap=EVENT_CAP, first_stream=rank * B_PER_GPU,
                                                             async_op=True, host_staging=(args.backend != "nccl"),
                                                             force_collective=args.force_collective)
            # (work.wait() makes the CURRENT stream wait for the collective: it is called on the tail's own stream, so the
            # front-end's stream never carries a dependency on the exchange)
            if work is not None:
                pending.append((out, work))
                say(f"step {i}: gather issued, {len(pending)} pending")
                if len(pending) > 4:
                    pending.pop(0)[1].wait()
                    say(f"step {i}: oldest gather done")
        return y, out

    def fence():
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
            while pending:
                pending.pop(0)[1].wait()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    say("setup done")
    # Device pre-warm (part of set-up, not of the W warm-up steps): the GPU raises its clocks only after some tens of
    # milliseconds of load, so a short run would otherwise time the ramp instead of the steady state.
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < PREWARM_S:
        for _ in range(50):
            fe(x)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    say("warm-up issued")
    fence()
    say("warm-up fenced")
    t0 = time.perf_counter()
    for i in range(args.steps):
        y, buf = step(i)
    say("steps issued")
    fence()
    say("steps fenced")
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if os.environ.get("SED_BENCH_DUMP"):
        print("kernel ms in launch order:", " ".join(f"{a.elapsed_time(b):.4f}" for a, b in k_events[:64]), file=sys.stderr)
    k_ms = sorted(a.elapsed_time(b) for a, b in k_events)
    kern_ms = sum(k_ms) / max(1, len(k_ms))

    def pct(q):
        return k_ms[min(len(k_ms) - 1, int(q * len(k_ms)))] if k_ms else None
    hours_per_step = world * B_PER_GPU * CLIP_S / 3600.0
    value = hours_per_step * args.steps / elapsed
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9

    # the full detector decode including what the step leaves out: device -> host copy of the runs and the Python
    # formatting of the reference's (onsets, offsets) arrays for 256 streams
    det2 = det[:, :, 0].contiguous()
    for _ in range(3):
        D.detect_onsets_offsets_batch(det2, 0.5, 0.48, True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    n_fmt = 20
    for _ in range(n_fmt):
        D.detect_onsets_offsets_batch(det2, 0.5, 0.48, True)
    host_fmt_ms = (time.perf_counter() - t1) / n_fmt * 1e3

    h2d = measure_h2d(x) if rank == 0 else None

    traffic, traffic_current = recorded_traffic("htsat_b256_fp32")

    res = {
        "metric": "log-mel audio-hours/sec", "value": value, "unit": "audio-hours/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "HTS-AT front-end 32 kHz n_fft=1024 hop=320 64 mel, 256 x 10 s clips per GPU "
                               "(BASELINE.json configs[1]) + detector decode of the same streams",
                   "clips_per_gpu": B_PER_GPU, "device_prewarm_s": PREWARM_S, "samples_per_clip": L, "frames_per_clip": T, "pcm": "f32",
                   "parallelism": f"streams sharded over {world} GPU(s), all-gather of event record buffers",
                   "decode_stream": args.decode_stream, "forced_one_rank_collective": bool(args.force_collective),
                   "step_covers": "fused log-mel kernel + one decode kernel (median, threshold, runs, gap merge, event records) (+ all-gather for N > 1), "
                                  "all on the device; formatting events as host dictionaries is outside the step",
                   "h2d_ms": h2d,
                   "decode_with_host_formatting_ms": host_fmt_ms,
                   "decode_with_host_formatting_note": "detect_onsets_offsets_batch on the same 256 streams: kernel + "
                                                       "device-to-host copy + Python formatting, per call"},
        "roofline": {"bound": "hbm", "kernel": "stft_mel_kernel<n_fft=1024, power, mel+log>", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": "profiles/hbm_traffic.json (rocprofv3 PMC passes of this command, not this run)",
                     "traffic_is_of_these_kernel_sources": traffic_current,
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kern_ms,
                     "kernel_ms_p10_p50_p90": [pct(0.1), pct(0.5), pct(0.9)]},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.cpu_baseline import time_htsat_cpu
        sample = x[:8].cpu()
        v, el, n = time_htsat_cpu(sample, budget_s=12.0)
        threads = torch.get_num_threads()
        torch.set_num_threads(1)
        v1, el1, n1 = time_htsat_cpu(sample[:2], budget_s=5.0)
        torch.set_num_threads(threads)
        model = cpu_model_name()
        res["cpu_baseline"] = {"value": v, "unit": "audio-hours/s", "cores": threads, "kind": "port",
                               "sample": f"{n} passes over 8 of the 256 clips (10 s each), {el:.1f} s of CPU work, "
                                         "torchlibrosa formulation (fp32 conv-DFT + mel matmul + 10log10) on torch CPU",
                               "single_thread_value": v1, "host_cpus": os.cpu_count(), "cpu_model": model}
    if rank == 0:
        emit(res)
    if world > 1 or args.force_collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
