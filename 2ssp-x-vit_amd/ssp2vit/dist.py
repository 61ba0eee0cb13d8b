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


# Exchange buffers of gather_batch_vectors, allocated ONCE per (device, world size, slots, L, ld) and re-used by every later step:
# the send block [slots, L, ld] and ONE contiguous receive block [ws, slots, L, ld] (round 3 allocated ws + 1 tensors inside every
# step, on the timed path of every rank).  The vectors handed back are views of the receive block: valid until the next exchange
# of the same shape on that device — core.stage1_scores adds them up before it returns.
_GATHER_BUFFERS = {}
STATS = {"batches_owned": 0, "exchanges": 0, "buffer_allocations": 0}      # per-process counters (bench.py prints them per rank)


def _gather_buffers(dev: torch.device, ws: int, slots: int, L: int, ld: int):
    key = (str(dev), ws, slots, L, ld)
    buf = _GATHER_BUFFERS.get(key)
    if buf is None:
        if len(_GATHER_BUFFERS) >= 8:
            _GATHER_BUFFERS.pop(next(iter(_GATHER_BUFFERS)))
        buf = _GATHER_BUFFERS[key] = (torch.zeros(slots, L, ld, dtype=torch.float32, device=dev),
                                      torch.empty(ws, slots, L, ld, dtype=torch.float32, device=dev))
        STATS["buffer_allocations"] += 1
    return buf


def gather_batch_vectors(local: Sequence[Tuple[int, torch.Tensor]], n_batches_total: int, group=None,
                         shape: Optional[Tuple[int, int]] = None) -> List[torch.Tensor]:
    """local = [(global batch index, tensor [L, ld])] owned by this rank (round-robin ownership).
    Returns the vectors of ALL batches in global batch order, identical on every rank.
    `shape` = (L, ld) when the caller knows it: ranks that own no batch then need no shape exchange, and nothing here
    makes the host wait for the device (the exchange reads a device scalar back)."""
    import torch.distributed as dist
    rank, ws = world(group)
    STATS["batches_owned"] += len(local)
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
    mine, everyone = _gather_buffers(dev, ws, slots, L, ld)
    owned = set()
    for idx, v in local:
        assert idx % ws == rank, "batch not owned by this rank"
        mine[idx // ws].copy_(v)
        owned.add(idx // ws)
    for k in range(slots):                    # slots this rank does not fill this time (ragged ownership): zeros, as a fresh buffer had
        if k not in owned:
            mine[k].zero_()
    with _timed("stage1_all_gather", dev):
        dist.all_gather_into_tensor(everyone.view(ws * slots, L, ld), mine, group=group)      # concatenation along dim 0: the form every backend takes
    STATS["exchanges"] += 1
    return [everyone[i % ws][i // ws] for i in range(n_batches_total)]


def all_reduce_counts(counts: torch.Tensor, group=None) -> torch.Tensor:
    """int64 tensor, summed over ranks in place (exact)."""
    import torch.distributed as dist
    _, ws = world(group)
    if ws > 1 or (FORCE_COLLECTIVES and _initialised()):
        with _timed("counts_all_reduce", counts.device):
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    return counts


def device_for_backend(backend: str, cuda_index: Optional[int]) -> torch.device:
    """Where a rank's exchange tensors live: RCCL ("nccl") moves device memory, so they sit on the rank's current HIP device;
    every other backend (gloo in the CPU tests) takes host tensors.  Pure: no device is touched here."""
    if str(backend) == "nccl":
        if cuda_index is None:
            raise RuntimeError("backend nccl (RCCL) without a current HIP device")
        return torch.device("cuda", int(cuda_index))
    return torch.device("cpu")


def _default_device(group=None) -> torch.device:
    import torch.distributed as dist
    backend = dist.get_backend(group)
    return device_for_backend(backend, torch.cuda.current_device() if (str(backend) == "nccl" and torch.cuda.is_available()) else None)


def rank_summary(extra: Optional[dict] = None) -> str:
    """One line per rank for stderr: who ran where, what it owned, what the exchanges cost — so that a failed or slow
    multi-rank run can be read off the driver's log tail."""
    rank, ws = world()
    dev = "cpu"
    if torch.cuda.is_available():
        i = torch.cuda.current_device()
        dev = f"cuda:{i} ({torch.cuda.get_device_name(i)})"
    parts = [f"[ssp2vit rank {rank}/{ws}] device={dev}", f"backend={__import__('torch.distributed').distributed.get_backend() if _initialised() else 'none'}",
             f"batches_owned={STATS['batches_owned']}", f"exchanges={STATS['exchanges']}", f"gather_buffer_allocations={STATS['buffer_allocations']}"]
    for k, v in (extra or {}).items():
        parts.append(f"{k}={v}")
    return " ".join(parts)
