"""Multi-GPU layout: independent streams sharded over ranks, one all-gather of fixed-size event buffers.

Every clip/stream is independent through the front-end and the decoders (the reference processes files one
by one: src/models/yamnet_train.py:227, src/models/crnn/utils.py:99), so ranks own contiguous blocks of the
stream batch and the only exchange is the gather of decoded events at the end of a batch.  One process per
GPU; the collective is torch.distributed's all_gather_into_tensor (RCCL over xGMI with backend "nccl",
gloo on CPU tensors in the tests).
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block partition: rank r owns [lo, hi); the first n_items % world_size ranks hold one more."""
    if world_size < 1 or not 0 <= rank < world_size or n_items < 0:
        raise ValueError("bad shard request")
    q, r = divmod(n_items, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def pack_events(runs: torch.Tensor, counts: torch.Tensor, cap: int, first_stream: int) -> torch.Tensor:
    """runs (S, C, cap_in, 2), counts (S, C) -> records int32 (S, cap, 4) = (stream, class, onset, offset), and the
    per-stream number of records in column 0 of an extra leading row: returns (S, cap + 1, 4).

    Row 0 of every stream is (n_records, n_dropped, 0, 0); records are class-major, onset-ascending, i.e. the
    order decode_predictions emits.  Events beyond `cap` per stream are dropped and counted.
    """
    S, C, cap_in, _ = runs.shape
    dev = runs.device
    slot = torch.arange(cap_in, device=dev)[None, None, :]
    valid = slot < counts[:, :, None].clamp(max=cap_in)                              # (S, C, cap_in)
    n_per_class = valid.sum(-1)                                                       # (S, C)
    class_base = torch.cumsum(n_per_class, 1) - n_per_class                           # exclusive prefix over classes
    pos = class_base[:, :, None] + slot                                               # record index inside the stream
    keep = valid & (pos < cap)
    out = torch.zeros((S, cap + 1, 4), dtype=torch.int32, device=dev)
    total = n_per_class.sum(1)
    out[:, 0, 0] = total.clamp(max=cap).to(torch.int32)
    out[:, 0, 1] = (total - total.clamp(max=cap)).to(torch.int32)
    s_idx, c_idx, k_idx = torch.nonzero(keep, as_tuple=True)
    p_idx = pos[s_idx, c_idx, k_idx] + 1
    out[s_idx, p_idx, 0] = (s_idx + first_stream).to(torch.int32)
    out[s_idx, p_idx, 1] = c_idx.to(torch.int32)
    out[s_idx, p_idx, 2] = runs[s_idx, c_idx, k_idx, 0]
    out[s_idx, p_idx, 3] = runs[s_idx, c_idx, k_idx, 1]
    return out


def all_gather_events(packed: torch.Tensor, group=None, async_op: bool = False):
    """All ranks hold the same number of streams (pad the last shard): (S, cap+1, 4) -> (world*S, cap+1, 4)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return (packed, None) if async_op else packed
    out = torch.empty((world * packed.shape[0],) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
    work = dist.all_gather_into_tensor(out, packed.contiguous(), group=group, async_op=async_op)
    return (out, work) if async_op else out


def unpack_events(gathered: torch.Tensor) -> List[List[Tuple[int, int, int, int]]]:
    """Host side: per stream, the list of (stream, class, onset_frame, offset_frame) records."""
    g = gathered.cpu().numpy()
    return [[tuple(int(v) for v in rec) for rec in g[s, 1:1 + g[s, 0, 0]]] for s in range(g.shape[0])]
