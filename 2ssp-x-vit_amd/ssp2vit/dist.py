"""Multi-GPU sharding of the hot path: one process per GPU, weights replicated, calibration /
eval BATCHES dealt round-robin (batch i -> rank i % P), no collective inside the forward.

Exchange steps (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests):
  * stage 1: ONE all_gather of the per-batch score vectors [n_local_batches, L, ld] (B/16: 147 KB per batch).
    Every rank then adds the batch vectors in GLOBAL batch order, so the scores — and therefore the masks —
    are bit-identical for every world size (a plain all-reduce would change the summation order with P).
  * stage 2: ONE all_reduce(sum) of the int64 correct-counts [L+1] (+ total), exact.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch

# debugging aid: run the collectives even in a one-rank process group (exercises the RCCL calls on a 1-GPU box)
FORCE_COLLECTIVES = bool(os.environ.get("SSP2_FORCE_COLLECTIVES"))


def _initialised() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def world(group=None) -> Tuple[int, int]:
    """(rank, world_size); (0, 1) when torch.distributed is not initialised."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def owns(batch_index: int, rank: int, world_size: int) -> bool:
    return batch_index % world_size == rank


def rank_batch_indices(n_items: int, batch_size: int, rank: int, world_size: int, limit: Optional[int] = None):
    """Batch sampler of one rank under the round-robin ownership above: the index lists of global batches rank,
    rank + P, rank + 2P, ... of a dataset of n_items cut into batches of batch_size (the last one may be short).
    Use as `DataLoader(dataset, batch_sampler=rank_batch_indices(...))` together with `sharded=True` in core.*: a rank
    then decodes and copies only what it owns (the default mode walks the whole loader on every rank)."""
    n_batches = (n_items + batch_size - 1) // batch_size
    if limit is not None:
        n_batches = min(n_batches, limit)
    return [list(range(b * batch_size, min(n_items, (b + 1) * batch_size))) for b in range(rank, n_batches, world_size)]


# optional timing of the exchange steps (bench.py --config 2): (name, start event, end event) per collective
TIMING = False
EVENTS: List[Tuple[str, object, object]] = []


class _timed:
    def __init__(self, name, device):
        self.on = TIMING and torch.device(device).type == "cuda"
        self.name = name

    def __enter__(self):
        if self.on:
            self.a = torch.cuda.Event(enable_timing=True); self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if self.on:
            self.b.record()
            EVENTS.append((self.name, self.a, self.b))
        return False


def collective_ms() -> dict:
    """Sum of the recorded collectives' device time per name since the last call (synchronises)."""
    out = {}
    for name, a, b in EVENTS:
        b.synchronize()
        out[name] = out.get(name, 0.0) + a.elapsed_time(b)
    EVENTS.clear()
    return out


def gather_batch_vectors(local: Sequence[Tuple[int, torch.Tensor]], n_batches_total: int, group=None,
                         shape: Optional[Tuple[int, int]] = None) -> List[torch.Tensor]:
    """local = [(global batch index, tensor [L, ld])] owned by this rank (round-robin ownership).
    Returns the vectors of ALL batches in global batch order, identical on every rank.
    `shape` = (L, ld) when the caller knows it: ranks that own no batch then need no shape exchange, and nothing here
    makes the host wait for the device (the exchange reads a device scalar back)."""
    import torch.distributed as dist
    rank, ws = world(group)
    if ws == 1 and not (FORCE_COLLECTIVES and _initialised()):
        return [v for _, v in sorted(local, key=lambda p: p[0])]
    if n_batches_total == 0:
        return []
    slots = (n_batches_total + ws - 1) // ws
    proto = local[0][1] if local else None
    if shape is not None:
        L, ld = int(shape[0]), int(shape[1])
    else:
        sh = torch.tensor(list(proto.shape) if proto is not None else [0, 0], dtype=torch.int64,
                          device=proto.device if proto is not None else _default_device(group))
        dist.all_reduce(sh, op=dist.ReduceOp.MAX, group=group)         # ranks without a batch learn the shape
        L, ld = int(sh[0]), int(sh[1])
    dev = proto.device if proto is not None else _default_device(group)
    mine = torch.zeros(slots, L, ld, dtype=torch.float32, device=dev)
    for idx, v in local:
        assert idx % ws == rank, "batch not owned by this rank"
        mine[idx // ws].copy_(v)
    everyone = [torch.empty_like(mine) for _ in range(ws)]
    with _timed("stage1_all_gather", dev):
        dist.all_gather(everyone, mine, group=group)
    return [everyone[i % ws][i // ws] for i in range(n_batches_total)]


def all_reduce_counts(counts: torch.Tensor, group=None) -> torch.Tensor:
    """int64 tensor, summed over ranks in place (exact)."""
    import torch.distributed as dist
    _, ws = world(group)
    if ws > 1 or (FORCE_COLLECTIVES and _initialised()):
        with _timed("counts_all_reduce", counts.device):
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    return counts


def _default_device(group=None) -> torch.device:
    import torch.distributed as dist
    backend = dist.get_backend(group)
    return torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
