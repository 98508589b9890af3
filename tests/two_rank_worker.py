"""One rank of tests/test_00_gpu_two_ranks.py: python two_rank_worker.py RANK WORLD PORT OUT_DIR.
Every rank uses cuda:0 (the test box has one GPU) and runs the HIP decode kernels on its block of the stream batch;
the record buffers are exchanged over gloo through host copies.  Rank 0 also decodes the whole batch in one call."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sound-event-detection_amd"))

# (streams, frames, classes, record capacity, gap in seconds, smoothing, seconds per frame)
CASES = [(37, 862, 5, 64, 0.6, "gauss", 1024 / 44100), (9, 20, 1, 16, 0.6, "median3", 0.48), (512, 41, 6, 8, 1.0, "median3", 0.48)]


def batch(case_idx):
    S, T, C = CASES[case_idx][:3]
    rng = np.random.default_rng(1000 + case_idx)
    z = rng.standard_normal((S, C, T + 8)).astype(np.float32)
    z = np.stack([np.convolve(r, np.ones(9, np.float32) / 3.0, mode="valid") for r in z.reshape(S * C, -1)]).reshape(S, C, T)
    return np.ascontiguousarray((1.0 / (1.0 + np.exp(-2.0 * z))).astype(np.float32).transpose(0, 2, 1))


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sed_amd
    from sed_amd import decode as D
    dev = torch.device("cuda", 0)
    out = {}
    for ci, (S, T, C, cap, gap, smoothing, dt) in enumerate(CASES):
        probs = batch(ci)
        table = torch.tensor([round(k * dt, 3) for k in range(T + 2)], dtype=torch.float64, device=dev)
        lo, hi = sed_amd.sharding.shard_bounds(S, rank, world)
        per_rank = -(-S // world)
        mine = np.zeros((per_rank, T, C), np.float32)                     # the last shard is padded with silent streams
        mine[:hi - lo] = probs[lo:hi]
        gathered = sed_amd.sharding.decode_and_exchange(torch.from_numpy(mine).to(dev), table, gap, cap=cap,
                                                        first_stream=rank * per_rank, smoothing=smoothing, host_staging=True)
        out[f"gathered{ci}"] = gathered.cpu().numpy()
        if rank == 0:
            whole = np.zeros((world * per_rank, T, C), np.float32)
            whole[:S] = probs
            out[f"single{ci}"] = D.threshold_merge_records(torch.from_numpy(whole).to(dev), 0.5, table, gap, smoothing, cap, 0).cpu().numpy()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
