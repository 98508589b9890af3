"""Multi-GPU layout: independent streams sharded over ranks, one all-gather of fixed-size event buffers.

Every clip/stream is independent through the front-end and the decoders (the reference processes files one
by one: src/models/yamnet_train.py:227, src/models/crnn/utils.py:99), so ranks own contiguous blocks of the
stream batch and the only exchange is the gather of decoded events at the end of a batch.  One process per
GPU; the collective is torch.distributed's all_gather_into_tensor (RCCL over xGMI with backend "nccl",
gloo on CPU tensors in the tests).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


FUSED_MAX_FRAMES = 8190   # longest trace whose runs the one-launch decode kernels keep in LDS (sed_threshold_merge_records)


def shard_bounds(n_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block partition: rank r owns [lo, hi); the first n_items % world_size ranks hold one more."""
    if world_size < 1 or not 0 <= rank < world_size or n_items < 0:
        raise ValueError("bad shard request")
    q, r = divmod(n_items, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def pack_events(runs: torch.Tensor, counts: Optional[torch.Tensor] = None, cap: int = 64, first_stream: int = 0) -> torch.Tensor:
    """Per-stream event records for the exchange between GPUs, formed on the device in one launch (no host
    synchronisation, fixed shapes, so the step can be captured in a HIP graph).

    runs (S, C, cap_in, 2) with counts (S, C), as threshold_runs / merge_runs return them, or the packed
    (S, C, 1 + 2*cap_in) buffer of threshold_merge_runs with counts=None.  Returns int32 (S, cap + 1, 4): row 0 of
    every stream is (n_records, n_dropped, 0, 0), rows 1.. are (stream, class, onset frame, offset frame) records,
    class-major and onset-ascending, i.e. the order decode_predictions emits (src/models/crnn/utils.py:99-134).
    Events beyond `cap` per stream are dropped and counted.
    """
    from . import _lib as L
    if not runs.is_cuda:
        raise RuntimeError("pack_events runs on a ROCm device: there is no CPU path")
    if runs.dtype != torch.int32:
        raise TypeError("runs must be int32")
    runs = runs.contiguous()
    if counts is None:
        if runs.dim() != 3 or runs.shape[2] % 2 != 1:
            raise ValueError("packed buffer must be (S, C, 1 + 2*cap_in)")
        S, C, w = runs.shape
        cap_in, stride = (w - 1) // 2, w
        runs_ptr, counts_ptr, runs_stride, counts_stride = runs.data_ptr() + 4, runs.data_ptr(), stride, stride
    else:
        if runs.dim() != 4 or runs.shape[3] != 2 or tuple(counts.shape) != tuple(runs.shape[:2]):
            raise ValueError("runs must be (S, C, cap_in, 2) and counts (S, C)")
        if counts.dtype != torch.int32 or counts.device != runs.device:
            raise TypeError("counts must be int32 on the device of runs")
        counts = counts.contiguous()
        S, C, cap_in, _ = runs.shape
        runs_ptr, counts_ptr, runs_stride, counts_stride = runs.data_ptr(), counts.data_ptr(), 2 * cap_in, 1
    out = torch.empty((S, cap + 1, 4), dtype=torch.int32, device=runs.device)
    with torch.cuda.device(runs.device):
        L.check(L.lib.sed_pack_events(runs_ptr, runs_stride, counts_ptr, counts_stride, S, C, cap_in, int(first_stream),
                                      out.data_ptr(), int(cap), L.stream_ptr(runs.device)), "sed_pack_events")
    return out


def all_gather_events(packed: torch.Tensor, group=None, async_op: bool = False, force_collective: bool = False):
    """All ranks hold the same number of streams (pad the last shard): (S, cap+1, 4) -> (world*S, cap+1, 4).
    With one rank nothing is exchanged unless force_collective is set (tests run the collective on a one-rank group)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (force_collective and dist.is_initialized()):
        return (packed, None) if async_op else packed
    out = torch.empty((world * packed.shape[0],) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
    work = dist.all_gather_into_tensor(out, packed.contiguous(), group=group, async_op=async_op)
    return (out, work) if async_op else out


def decode_and_exchange(det_probs: torch.Tensor, time_table: torch.Tensor, gap: float, cap: int = 64,
                        first_stream: int = 0, threshold: float = 0.5, smoothing: str = "median3", group=None,
                        async_op: bool = False, force_collective: bool = False, host_staging: bool = False):
    """The detector-side tail of one batch on one rank, all on the device and without host synchronisation:
    smoothing + threshold + run extraction + gap merge (src/models/yamnet_train.py:55-71, src/utils/merge_events.py:36-55)
    of this rank's streams and their (stream, class, onset, offset) records in one launch, then the all-gather of the
    record buffers.  det_probs (S, T, C) fp32 on the device; one launch for T <= 8190 (covers the 862-frame CRNN
    traces of src/models/crnn/crnn.py:181-182 and HTS-AT's 4096-frame maps), three launches with the same records beyond;
    returns what all_gather_events returns.
    host_staging copies the records to the host before the collective (for backends without a device path)."""
    from . import decode as D
    if det_probs.dim() == 3 and det_probs.shape[1] > FUSED_MAX_FRAMES:
        # traces too long for the LDS buffers of the one-launch kernel: the same records from three launches
        runs, counts = D.threshold_runs(det_probs.contiguous(), threshold, smoothing)
        runs, counts = D.merge_runs(runs, counts, time_table, gap)
        records = pack_events(runs, counts, cap, first_stream)
    else:
        records = D.threshold_merge_records(det_probs, threshold, time_table, gap, smoothing, cap, first_stream)
    if host_staging and dist.is_initialized() and dist.get_world_size(group) > 1:
        records = records.cpu()          # rehearsals on a backend without a device path (gloo)
    return all_gather_events(records, group=group, async_op=async_op, force_collective=force_collective)


def unpack_events(gathered: torch.Tensor) -> List[List[Tuple[int, int, int, int]]]:
    """Host side: per stream, the list of (stream, class, onset_frame, offset_frame) records."""
    g = gathered.cpu().numpy()
    return [[tuple(int(v) for v in rec) for rec in g[s, 1:1 + g[s, 0, 0]]] for s in range(g.shape[0])]
