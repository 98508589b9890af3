"""The default bench step (fused HTS-AT front-end of 256 x 10 s + record decode of 256 detector traces) launched eagerly and
replayed from a HIP graph: time per step.  python profiles/tools/graph_step.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "sound-event-detection_amd"))
import sed_amd  # noqa: E402
from sed_amd import decode as D  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
x = 0.1 * torch.randn((256, 320000), device=dev)
det = torch.rand((256, 20, 1), device=dev)
table = torch.tensor([k * 0.48 for k in range(22)], dtype=torch.float64, device=dev)
fe = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat(), layout="btm")


def step():
    y = fe(x)
    r = D.threshold_merge_records(det, 0.5, table, 0.6, "median3", 16, 0)
    return y, r


def timed(fn, n=1000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    return best


step()
torch.cuda.synchronize()
print(f"eager       {timed(step):7.1f} us per step")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
print(f"graph       {timed(g.replay):7.1f} us per step")
g8 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g8):
    for _ in range(8):
        out = step()
print(f"graph of 8  {timed(g8.replay, 125) / 8:7.1f} us per step")
