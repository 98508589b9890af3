"""Two ranks on the real kernels.  Each rank is a fresh child process (started before this process has touched the GPU:
this file is collected first) that runs sed_amd.sharding.decode_and_exchange on its shard_bounds block of the batch with
the HIP kernels on cuda:0 and exchanges the record buffers over gloo; the gathered (world * S, cap + 1, 4) buffer of
every rank must equal the records of the whole batch decoded by one process: rank offsets, the padding of the last shard
and the record order survive a real exchange.  (The 8-GPU RCCL run itself is the driver's.)"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_ranks_decode_and_exchange_equals_one_process(tmp_path):
    if torch.cuda.is_initialized():
        pytest.skip("needs a process that has not touched the GPU yet (run this file first, as the default order does)")
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "two_rank_worker.py"), str(r), str(world), str(port), str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=300)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"
    sys.path.insert(0, HERE)
    from two_rank_worker import CASES
    r0 = np.load(os.path.join(tmp_path, "rank0.npz"))
    r1 = np.load(os.path.join(tmp_path, "rank1.npz"))
    for ci, (S, T, C, cap, gap, smoothing, dt) in enumerate(CASES):
        g0, g1, single = r0[f"gathered{ci}"], r1[f"gathered{ci}"], r0[f"single{ci}"]
        per_rank = -(-S // world)
        assert g0.shape == (world * per_rank, cap + 1, 4) and g0.dtype == np.int32
        assert np.array_equal(g0, g1), f"case {ci}: the ranks hold different gathered buffers"
        assert np.array_equal(g0, single), f"case {ci}: gathered records differ from the one-process decode"
        kept = g0[:S, 0, 0]
        assert kept.sum() > S // 2                                           # the batch does hold events
        for s in (0, per_rank - 1, per_rank, S - 1):                         # stream numbers carry the rank offset
            assert np.all(g0[s, 1:1 + g0[s, 0, 0], 0] == s)
        assert np.all(g0[S:, 0, 0] == 0)                                     # padding streams are silent
