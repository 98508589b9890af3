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
PREWARM_S = 0.3


def synth_pcm(n_clips, n_samples, device, seed):
    """Seeded synthetic audio generated on the device: noise + two sine bursts per clip + 0.5 s of silence."""
    g = torch.Generator(device=device).manual_seed(seed)
    x = 0.1 * torch.randn((n_clips, n_samples), generator=g, device=device)
    t = torch.arange(n_samples, device=device) / SR
    f = 200.0 + 5800.0 * torch.rand((n_clips, 2), generator=g, device=device)
    a = 0.05 + 0.45 * torch.rand((n_clips, 2), generator=g, device=device)
    s0 = torch.rand((n_clips, 2), generator=g, device=device) * (CLIP_S - 3.0)
    for j in range(2):
        m = (t[None] >= s0[:, j:j + 1]) & (t[None] < s0[:, j:j + 1] + 2.0)
        x += m * a[:, j:j + 1] * torch.sin(2 * torch.pi * f[:, j:j + 1] * t[None])
    sil = (torch.rand((n_clips, 1), generator=g, device=device) * (CLIP_S - 0.5) * SR).long()
    idx = torch.arange(n_samples, device=device)[None]
    x.masked_fill_((idx >= sil) & (idx < sil + SR // 2), 0.0)
    return x.clamp_(-1.0, 1.0).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                    "rehearse the multi-rank control flow on a box with one GPU, together with --one-device)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--decode-stream", choices=["side", "same"], default="same",
                    help="side: the detector decode of a step runs on a second HIP stream beside the front-end kernel")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Not under torchrun: start the ranks as a fresh child process tree (nothing here has touched the GPU yet) and
        # pass its output through; rank 0 of the children prints the JSON line.
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

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

    import sed_amd
    from sed_amd import decode as D

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
                                                             async_op=True, host_staging=(args.backend != "nccl"))
        if work is not None:
            pending.append((out, work))
            say(f"step {i}: gather issued, {len(pending)} pending")
            if len(pending) > 4:
                pending.pop(0)[1].wait()
                say(f"step {i}: oldest gather done")
        return y, out

    def fence():
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

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("htsat_b256_fp32", {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    res = {
        "metric": "log-mel audio-hours/sec", "value": value, "unit": "audio-hours/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "HTS-AT front-end 32 kHz n_fft=1024 hop=320 64 mel, 256 x 10 s clips per GPU "
                               "(BASELINE.json configs[1]) + detector decode of the same streams",
                   "clips_per_gpu": B_PER_GPU, "device_prewarm_s": PREWARM_S, "samples_per_clip": L, "frames_per_clip": T, "pcm": "f32",
                   "parallelism": f"streams sharded over {world} GPU(s), all-gather of event record buffers",
                   "step_covers": "fused log-mel kernel + one decode kernel (median, threshold, runs, gap merge, event records) (+ all-gather for N > 1), "
                                  "all on the device; formatting events as host dictionaries is outside the step",
                   "decode_with_host_formatting_ms": host_fmt_ms,
                   "decode_with_host_formatting_note": "detect_onsets_offsets_batch on the same 256 streams: kernel + "
                                                       "device-to-host copy + Python formatting, per call"},
        "roofline": {"bound": "hbm", "kernel": "stft_mel_kernel<n_fft=1024, power, mel+log>", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": "profiles/hbm_traffic.json (rocprofv3 PMC passes of this command, not this run)",
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
        model = ""
        try:
            with open("/proc/cpuinfo") as f:
                model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
        except OSError:
            pass
        res["cpu_baseline"] = {"value": v, "unit": "audio-hours/s", "cores": threads, "kind": "port",
                               "sample": f"{n} passes over 8 of the 256 clips (10 s each), {el:.1f} s of CPU work, "
                                         "torchlibrosa formulation (fp32 conv-DFT + mel matmul + 10log10) on torch CPU",
                               "single_thread_value": v1, "host_cpus": os.cpu_count(), "cpu_model": model}
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
