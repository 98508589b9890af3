"""Multi-rank path on CPU: two gloo ranks shard the stream batch, decode their block and form the event records (the
oracle stands in for the GPU kernels, which have no CPU path: threshold_runs' output layout and pack_events' record
format) and all-gather the fixed-size record buffers; every rank must end up with the events of the whole batch in
stream order, identical to a single-process decode.  The GPU side of the same path (the HIP pack kernel against this
oracle, gathered over RCCL) is tests/test_gpu_sharding.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import decode as od


def _runs_from_oracle(probs, threshold, cap_in):
    """Same layout sed_threshold_runs writes: runs (S, C, cap_in, 2), counts (S, C)."""
    S, T, C = probs.shape
    runs = np.zeros((S, C, cap_in, 2), np.int32)
    counts = np.zeros((S, C), np.int32)
    act = (od.median3_time(probs, 1) > np.float32(threshold)).astype(int)
    for s in range(S):
        for c in range(C):
            reg = od.contiguous_regions(act[s, :, c])
            counts[s, c] = len(reg)
            for k, (a, b) in enumerate(reg[:cap_in]):
                runs[s, c, k] = (a, b)
    return torch.from_numpy(runs), torch.from_numpy(counts)


def _batch(n_streams, T=20, C=2):
    rng = np.random.default_rng(77)
    return rng.random((n_streams, T, C), dtype=np.float32)


def _worker(rank, world, port, n_streams, cap, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sed_amd import sharding
    probs = _batch(n_streams)
    lo, hi = sharding.shard_bounds(n_streams, rank, world)
    per_rank = -(-n_streams // world)
    mine = np.zeros((per_rank,) + probs.shape[1:], np.float32)           # last shard padded with silent streams
    mine[:hi - lo] = probs[lo:hi]
    runs, counts = _runs_from_oracle(mine, 0.5, probs.shape[1] // 2 + 1)
    packed = torch.from_numpy(od.pack_events(runs.numpy(), counts.numpy(), cap, first_stream=rank * per_rank))
    gathered, work = sharding.all_gather_events(packed, async_op=True)
    work.wait()
    torch.save(gathered, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_streams,cap", [(10, 16), (7, 3)])
def test_two_rank_event_gather(tmp_path, n_streams, cap):
    from sed_amd import sharding
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world = 2
    mp.spawn(_worker, args=(world, port, n_streams, cap, str(tmp_path)), nprocs=world, join=True)
    g0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    g1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    assert torch.equal(g0, g1)
    per_rank = -(-n_streams // world)
    assert g0.shape == (world * per_rank, cap + 1, 4)
    probs = _batch(n_streams)
    runs, counts = _runs_from_oracle(probs, 0.5, probs.shape[1] // 2 + 1)
    events = sharding.unpack_events(g0)
    for s in range(n_streams):
        want = [(s, c, int(a), int(b)) for c in range(probs.shape[2]) for a, b in runs[s, c, :counts[s, c]].tolist()]
        assert events[s] == want[:cap]
        assert int(g0[s, 0, 0]) == min(len(want), cap) and int(g0[s, 0, 1]) == max(0, len(want) - cap)
    for s in range(n_streams, world * per_rank):
        assert events[s] == []


def test_shard_bounds_cover_everything():
    from sed_amd import sharding
    for n in (0, 1, 7, 8, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_bounds(4, 2, 2)


def test_single_process_gather_is_identity():
    from sed_amd import sharding
    packed = torch.arange(2 * 5 * 4, dtype=torch.int32).reshape(2, 5, 4)
    assert sharding.all_gather_events(packed) is packed
