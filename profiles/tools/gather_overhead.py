"""Host-side cost of the per-batch event all-gather: the bench step with and without a (world size 1) RCCL
all_gather_into_tensor behind it.  With one rank the collective is a device copy, but the launch path through
torch.distributed and RCCL is the one the N > 1 bench takes."""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sound-event-detection_amd"))
import sed_amd
from sed_amd import decode as D

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

fe = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat())
x = torch.randn(256, 320000, device=dev) * 0.1
det = torch.rand(256, 20, 1, device=dev)
table = torch.tensor([k * 0.48 for k in range(22)], dtype=torch.float64, device=dev)


def run(gather, steps=300, depth=4):
    pending = []
    for it in range(steps + 30):
        if it == 30:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        fe(x)
        buf = D.threshold_merge_runs(det, 0.5, table, 0.6, "median3")
        if gather:
            out = torch.empty_like(buf)
            pending.append((out, dist.all_gather_into_tensor(out, buf, async_op=True)))
            if len(pending) > depth:
                pending.pop(0)[1].wait()
    for _, w in pending:
        w.wait()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


def host_only(gather, steps=300):
    """Issue time alone (no device work to wait for is excluded by measuring before the final synchronize)."""
    pending = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):
        fe(x)
        buf = D.threshold_merge_runs(det, 0.5, table, 0.6, "median3")
        if gather:
            out = torch.empty_like(buf)
            pending.append((out, dist.all_gather_into_tensor(out, buf, async_op=True)))
            if len(pending) > 4:
                pending.pop(0)[1].wait()
    t1 = time.perf_counter()
    for _, w in pending:
        w.wait()
    torch.cuda.synchronize()
    return (t1 - t0) / steps * 1e6


for g in (False, True, False, True):
    print(f"gather={g}: {run(g):7.1f} us per step, host issue {host_only(g):7.1f} us per step", flush=True)
print(f"gather, no waits before the end: {run(True, depth=10**9):7.1f} us per step")
print(f"gather, 1 in flight:              {run(True, depth=1):7.1f} us per step")
dist.destroy_process_group()
