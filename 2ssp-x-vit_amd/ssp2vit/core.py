"""Engine-level drivers of the two hot loops.  They take a device engine (ssp2vit.engine.VitEngine) — or a
factory that builds one once the first batch size is known — plus plain metadata, so they serve both the
reference-named wrappers in `vit_pruning` (live nn.Module in, like the reference) and `bench.py` (flat weights).
"""
from __future__ import annotations

import contextlib
import os
from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import torch

from . import dist as _dist
from ._lib import SLAB_ALIGN
from .mask_parity import mask_parity_report


class _OwnedBatches:
    """Walks a dataloader and yields (global batch index, batch) for the batches THIS rank owns, up to a GLOBAL batch limit
    (the reference's `batch_limit` / `max_batches`, src/vit_pruning.py:174-176, :347-349, counts batches of the one loader).
    Plain loader: every rank walks all batches and keeps batch i iff i % P == rank.  `sharded` loader (yields only this rank's batches,
    its k-th is global batch k * P + rank): the limit applies to the GLOBAL index, so N ranks together cover `limit` batches, not
    N x limit.  `.batches` / `.samples` count what this rank SAW (all batches of a plain loader, its own of a sharded one)."""

    def __init__(self, dataloader, limit, rank, ws, sharded, progress=False, desc=""):
        self.dl, self.limit, self.rank, self.ws, self.sharded, self.progress, self.desc = dataloader, limit, rank, ws, sharded, progress, desc
        self.batches = self.samples = 0
        self.sizes: List[Tuple[int, int]] = []                # (global batch index, images) of every batch seen

    def __iter__(self):
        local_limit = self.limit
        if self.sharded and self.limit is not None:           # this rank's share of the first `limit` global batches
            local_limit = len(range(self.rank, int(self.limit), self.ws))
        for i, batch in iter_limited(self.dl, local_limit, self.progress, self.desc):
            self.batches += 1
            self.samples += int(batch["pixel_values"].size(0))
            self.sizes.append((i * self.ws + self.rank if self.sharded else i, int(batch["pixel_values"].size(0))))
            if self.sharded:
                yield i * self.ws + self.rank, batch
            elif _dist.owns(i, self.rank, self.ws):
                yield i, batch


def iter_limited(dataloader: Iterable, limit: Optional[int], progress: bool, desc: str):
    it = dataloader
    if progress:
        try:
            from tqdm.auto import tqdm
            it = tqdm(dataloader, total=limit, desc=desc, leave=False)
        except Exception:
            pass
    for i, batch in enumerate(it):
        if limit is not None and i >= limit:
            break
        yield i, batch


# Images per forward.  Stage 1 packs whole dataloader batches (each its own slab), the search cuts the eval stream into
# forwards of exactly this many images; both defaults are the sizes measured best on ViT-B/16 (DESIGN.md section 4:
# larger is better, the tile-quantisation tail of a 64-image launch costs more than cache residency buys).
DEFAULT_CHUNK_IMAGES = int(os.environ.get("SSP2_CHUNK_IMAGES", "512"))
DEFAULT_EVAL_CHUNK_IMAGES = int(os.environ.get("SSP2_EVAL_CHUNK_IMAGES", "320"))
MAX_SLABS = 16          # the engine's workspace carries slack for 16 padded slabs per call (csrc/engine.hip rows_cap)
# Round-3 host-side changes, each switchable for same-box A/B runs (none changes a result): one tail over all search slots,
# candidate l started out of place from the baseline's stream, batches embedded into their rows without a concatenated copy
TAIL_SLOTS = os.environ.get("SSP2_TAIL_SLOTS", "1") != "0"
OUT_OF_PLACE_START = os.environ.get("SSP2_OUT_OF_PLACE_START", "1") != "0"
BATCH_LISTS = os.environ.get("SSP2_BATCH_LISTS", "0") != "0"     # measured 0.3-0.5 ms SLOWER per step than one torch.cat (13 x 2 small embed launches): off


def _resolve(engine, min_images: int):
    return engine(min_images) if callable(engine) else engine


def workspace_budget_bytes(device=None) -> int:
    """What the layer-major search may spend on engine workspace + snapshots: SSP2_WORKSPACE_GB, else half of the free
    HBM (288 GB per MI355X; ViT-B/16 needs 13 GB for 12 x 320 images, ViT-H/14 about 75 GB for 32 x 320)."""
    gb = os.environ.get("SSP2_WORKSPACE_GB")
    if gb:
        return int(float(gb) * (1 << 30))
    if torch.cuda.is_available():
        free, _ = torch.cuda.mem_get_info(device)
        return free // 2
    return 0


def layer_major_images(engine, slots: int, n: int, device=None) -> int:
    """Engine capacity (images) to ask for when `slots` copies of an n-image chunk should run side by side; n when that
    does not fit the budget or the engine is a fixed object that is too small (the search then runs candidate-major)."""
    want = slots * n
    if not callable(engine):
        return want if engine.max_images >= want else n
    per_image = getattr(engine, "bytes_per_image", None)
    if per_image is None:
        return n
    return want if want * per_image() <= workspace_budget_bytes(device) else n


_COPY_STREAMS = {}
_COUNT_STREAMS = {}
_INDEX_CACHE = {}


def _device_index(values: Sequence[int], device) -> torch.Tensor:
    """A small int64 index tensor on the device, built once per (device, values): a host list turned into a device tensor inside
    the search would be a pageable H2D copy, which makes the host wait for everything already enqueued."""
    key = (str(device), tuple(int(v) for v in values))
    t = _INDEX_CACHE.get(key)
    if t is None:
        t = _INDEX_CACHE[key] = torch.tensor(list(key[1]), dtype=torch.int64).to(device)
        if len(_INDEX_CACHE) > 256:
            _INDEX_CACHE.pop(next(iter(_INDEX_CACHE)))
    return t


def _to_device(t: torch.Tensor, device, dtype) -> torch.Tensor:
    """Batch tensor -> device.  A tensor that already lives there is passed through.  A HOST tensor (what the
    reference's dataloaders yield, src/vit_pruning.py:177) is copied on a per-device COPY STREAM: the host runs ahead of
    the GPU, so the copies of later batches overlap the forward of earlier ones instead of sitting between the
    kernels of the compute stream (pinned memory makes them asynchronous; pageable memory still works, staged by the
    runtime).  The compute stream waits for the copy's event before the first use."""
    device = torch.device(device)
    if t.device == device or device.type != "cuda":
        return t.to(device, dtype, non_blocking=True)
    key = (device.index if device.index is not None else torch.cuda.current_device())
    cs = _COPY_STREAMS.get(key)
    if cs is None:
        cs = _COPY_STREAMS[key] = torch.cuda.Stream(device=device)
    cur = torch.cuda.current_stream(device)
    with torch.cuda.stream(cs):
        out = t.to(device, dtype, non_blocking=True)
    cur.wait_stream(cs)
    out.record_stream(cur)
    return out


def _pixels_to_device(batch, device) -> torch.Tensor:
    """`pixel_values` of a dataloader batch -> fp32 NCHW on the device.
    fp32 batches (what the reference's loaders yield after their torchvision chain) go through `_to_device`.
    uint8 HWC batches — the raw images, for a loader that leaves the transform to the GPU — carry their
    `ssp2vit.preprocess.GpuPreprocessor` under batch["preprocess"] (and optional per-image flips under "hflip"): the
    3-byte pixels cross PCIe on the copy stream (a quarter of the fp32 bytes at equal resolution, 1/200 for CIFAR) and
    the reference's Resize(BICUBIC) -> flip -> ToTensor -> Normalize (auto_2ssp.py:290-301) runs there as
    ssp2_preproc_run, bit-identical to the Pillow / torchvision chain; the compute stream waits per batch."""
    px = batch["pixel_values"]
    if px.dtype != torch.uint8:
        return _to_device(px, device, torch.float32)
    pp = batch.get("preprocess") if hasattr(batch, "get") else None
    if pp is None:
        raise ValueError("uint8 pixel_values need batch['preprocess'] (an ssp2vit.preprocess.GpuPreprocessor)")
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device())
    cs = _COPY_STREAMS.get(key)
    if cs is None:
        cs = _COPY_STREAMS[key] = torch.cuda.Stream(device=device)
    cur = torch.cuda.current_stream(device)
    # (no cs.wait_stream(cur): the preprocessor's tables were uploaded synchronously when it was created, and waiting for
    # the compute stream here would park every later batch's copy behind the forwards already enqueued)
    with torch.cuda.stream(cs):
        out = pp(px, batch.get("hflip"))
    cur.wait_stream(cs)
    out.record_stream(cur)
    return out


class _Chunker:
    """Packs consecutive dataloader batches into one device forward of up to `capacity` images.

    GEMM tiles are 128 rows of (image, token) pairs; one 64-image batch of ViT-B/16 gives 99 row tiles, which fills
    the 256 CUs unevenly.  Several batches per launch fix the tail and amortise launches.  Batches inside a chunk
    must all have the size of the first one, except the last (a ragged final batch closes the chunk), so that group
    g of the chunk is exactly dataloader batch g (needed for the per-batch score sums)."""

    def __init__(self, capacity: int, device, max_batches: int = 1 << 30):
        self.capacity, self.device, self.max_batches = capacity, device, max_batches
        self.items = []          # (global batch index, pixels on device, labels or None)
        self.count = 0

    def full_for(self, n: int) -> bool:
        if not self.items:
            return False
        g = self.items[0][1].size(0)
        return (n > g or self.count + n > self.capacity or self.items[-1][1].size(0) < g
                or len(self.items) >= self.max_batches)

    def add(self, idx, batch, labels=None):
        self.items.append((idx, _pixels_to_device(batch, self.device),
                           None if labels is None else _to_device(labels, self.device, torch.int64)))
        self.count += int(batch["pixel_values"].size(0))

    def take(self, as_list: bool = False):
        items, self.items, self.count = self.items, [], 0
        px = items[0][1] if len(items) == 1 else ([it[1] for it in items] if as_list else torch.cat([it[1] for it in items], 0))
        labels = None
        if items[0][2] is not None:
            labels = items[0][2] if len(items) == 1 else torch.cat([it[2] for it in items], 0)
        return [it[0] for it in items], int(items[0][1].size(0)), px, labels


@torch.no_grad()
def stage1_scores(engine, dataloader, d_ints: Sequence[int], site: str, *, batch_limit: Optional[int] = None,
                  progress: bool = False, score_chain: str = "fp32", process_group=None,
                  chunk_images: Optional[int] = None, defer: bool = False, sharded: bool = False):
    """Reference src/vit_pruning.py:111-201.  `engine`: VitEngine or callable(min_images) -> VitEngine.

    Multi-rank: by default every rank walks the SAME dataloader and keeps batch i iff i % P == rank (the others are
    only counted).  `sharded=True` says the loader already yields ONLY this rank's batches, in order (its k-th batch
    is global batch k*P + rank — `dist.rank_batch_indices` builds such a batch sampler): non-owned batches are then
    never decoded or copied, and the global batch / sample counts come from one extra int64 all_reduce.

    Several dataloader batches share one forward (`chunk_images`, default SSP2_CHUNK_IMAGES).  A sample's sum of
    squares is folded per 128-row GEMM tile, so its fp32 rounding depends on where the sample sits relative to the
    tile grid; every batch is therefore laid out as its own 128-row-aligned slab (ssp2_rows / RowMap), which pins
    that position: scores are bit-identical for every packing and every world size.

    `defer=True` (fp32 chain) returns a zero-argument callable instead of the list: all device work is enqueued, the
    device-to-host copy and the wait happen when it is called — a caller can enqueue stage 2 first and run its host
    mask step while the GPU works (bench.py does)."""
    rank, ws = _dist.world(process_group)
    sharded = sharded or bool(getattr(dataloader, "sharded", False))     # a loader that deals the batches itself says so (ssp2vit.local_data)
    chunk_images = DEFAULT_CHUNK_IMAGES if chunk_images is None else chunk_images
    local: List[Tuple[int, torch.Tensor]] = []
    eng = None
    ch = None

    ramp = []                    # capacities of the launches that follow the first (host-fed batches only, see below)

    def flush():
        idxs, group, px, _ = ch.take(as_list=BATCH_LISTS and getattr(eng, "batch_lists", False))     # VitEngine embeds each batch into its slab: no cat
        vec = eng.forward_scores(px, site, score_chain, group)      # [len(idxs), L, ld]
        for k, i in enumerate(idxs):
            local.append((i, vec[k]))
        if ramp:
            ch.capacity = ramp.pop(0)

    walk = _OwnedBatches(dataloader, batch_limit, rank, ws, sharded, progress, "S1 activations")
    for gi, batch in walk:                                         # gi: global batch index
        px = batch["pixel_values"]
        n = int(px.size(0))
        if eng is None or (callable(engine) and n > eng.max_images):
            if ch is not None and ch.items:
                flush()
            eng = _resolve(engine, max(chunk_images, n))
            cap = min(eng.max_images, max(chunk_images, n))
            first = cap
            if px.device.type == "cpu" and px.dtype != torch.uint8 and torch.device(eng.device).type == "cuda" and cap >= 8 * n:
                # fp32 host batches: nothing can run before the first launch's pixels have crossed PCIe, so the first launch
                # takes only two batches and the copies of the rest overlap its forward (scores do not depend on the
                # packing: every batch is its own slab).  Same-box A/B on the 512-image calibration set: 114.7 ms per step
                # against 116.7 with two equal launches (profiles/r02_g_host_launch_split_ab.txt); uint8 batches are a
                # quarter of the bytes and keep the single large launch.
                first, ramp[:] = 2 * n, [cap]
            ch = _Chunker(first, eng.device, max_batches=MAX_SLABS)
        if ch.full_for(n):
            flush()
        ch.add(gi, batch)
    if ch is not None and ch.items:
        flush()
    return _reduce_scores(local, walk.batches, walk.samples, d_ints, score_chain, process_group, defer, sharded)


def _reduce_scores(local, n_batches: int, n_samples: int, d_ints: Sequence[int], score_chain: str, process_group, defer: bool,
                   sharded: bool):
    """Per-batch score sums of this rank [(global batch index, f32 [L, ld])] -> the stage-1 importances: ONE all_gather of the vectors,
    added in GLOBAL batch order on every rank (identical bits for every world size and packing), divided by the global sample count
    (reference src/vit_pruning.py:154-157, :194-200).  n_batches / n_samples: what this rank saw (see _OwnedBatches)."""
    rank, ws = _dist.world(process_group)
    if sharded and (ws > 1 or (_dist.FORCE_COLLECTIVES and _dist._initialised())):
        dev = _dist._default_device(process_group)
        if dev.type == "cuda":
            # The global batch / sample counts must come BACK to the host (they size the exchange below).  On the compute stream that
            # read-back would wait for everything already enqueued — in the one-pass prune the whole search — before the exchange, the
            # scores' copy and the tails could even be enqueued.  The two integers have no producer on the device, so their little
            # all-reduce runs on a side stream of its own and the host waits for that stream only.
            key = dev.index if dev.index is not None else torch.cuda.current_device()
            ent = _COUNT_STREAMS.get(key)
            if ent is None:             # (stream, pinned host pair, device pair): made once per device
                ent = _COUNT_STREAMS[key] = (torch.cuda.Stream(device=dev), torch.zeros(2, dtype=torch.int64).pin_memory(),
                                             torch.zeros(2, dtype=torch.int64, device=dev))
            side, pin, tot = ent
            pin[0], pin[1] = int(n_batches), int(n_samples)
            with torch.cuda.stream(side):
                tot.copy_(pin, non_blocking=True)
                _dist.all_reduce_counts(tot, process_group)
                pin.copy_(tot, non_blocking=True)
                side.synchronize()
            n_batches, n_samples = int(pin[0]), int(pin[1])
        else:
            tot = torch.tensor([n_batches, n_samples], dtype=torch.int64, device=dev)
            n_batches, n_samples = (int(v) for v in _dist.all_reduce_counts(tot, process_group).to("cpu"))
    ld = max((int(d) + 63) // 64 * 64 for d in d_ints) if d_ints else 0          # = VitEngine.score_ld
    vecs = _dist.gather_batch_vectors(local, n_batches, process_group,
                                      shape=(local[0][1].shape if local else (len(d_ints), ld)))
    denom = max(1, n_samples)
    if not vecs:
        empty = [torch.zeros(d) for d in d_ints]
        return (lambda: empty) if defer else empty
    if score_chain == "fp32":
        total = torch.zeros_like(vecs[0])
        for v in vecs:                          # global batch order: identical on every rank / world size / chunking
            total += v
        total = total / denom
        if defer and total.is_cuda:
            # the copy rides a side stream behind an event recorded HERE, so that waiting for the scores does not wait
            # for whatever the caller enqueues on the compute stream afterwards (stage 2)
            main = torch.cuda.current_stream(total.device)
            side = torch.cuda.Stream(total.device)
            host = torch.empty(total.shape, dtype=total.dtype, pin_memory=True)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                host.copy_(total, non_blocking=True)
                total.record_stream(side)
                done = torch.cuda.Event()
                done.record(side)

            def finish_async():
                done.synchronize()
                return [host[l, :d].clone() for l, d in enumerate(d_ints)]
            return finish_async

        def finish():
            host = total.to("cpu")
            return [host[l, :d].clone() for l, d in enumerate(d_ints)]
        return finish if defer else finish()
    # bf16_ref: the cross-batch `+=` and the final division happen in bf16 exactly as reference :154-157, :200
    host = [v.to("cpu") for v in vecs]
    imps: List[torch.Tensor] = []
    for l, d in enumerate(d_ints):
        run = None
        for v in host:
            acc = v[l, :d].to(torch.bfloat16)   # exact: the kernel already rounded the batch sum to bf16
            if run is None:
                run = acc.clone()
            else:
                run += acc
        imps.append(run / denom)
    return imps


def _chunks(engine, dataloader, limit, progress, desc, rank, ws, chunk_images, capacity=None, sharded=False):
    """Yields (engine, pixels, labels) chunks of EXACTLY `chunk_images` images (the last one may be short) cut from
    the concatenation of the batches this rank owns.  Evaluation results are integer counts, so where the cuts fall
    cannot change them; the chunk size is chosen for the GEMM tile grid (see best_eval_chunk)."""
    eng = None
    px_buf, lb_buf, count = [], [], 0
    sharded = sharded or bool(getattr(dataloader, "sharded", False))     # a loader that deals the batches itself says so (ssp2vit.local_data)

    def cut(k):
        nonlocal px_buf, lb_buf, count
        lb = lb_buf[0] if len(lb_buf) == 1 else torch.cat(lb_buf, 0)
        if BATCH_LISTS and getattr(eng, "batch_lists", False):
            # the chunk's pixels stay a LIST of (views of) the loader's batches: the engine embeds each into its rows of the
            # token matrix — no concatenated copy of the pixels (193 MB per 320-image chunk of ViT-B/16)
            take, rest, left = [], [], k
            for p in px_buf:
                m = int(p.shape[0])
                if left >= m:
                    take.append(p); left -= m
                elif left > 0:
                    take.append(p[:left]); rest.append(p[left:]); left = 0
                else:
                    rest.append(p)
            out = (eng, take[0] if len(take) == 1 else take, lb[:k])
            px_buf, lb_buf, count = (rest, [lb[k:]], count - k) if k < count else ([], [], 0)
            return out
        px = px_buf[0] if len(px_buf) == 1 else torch.cat(px_buf, 0)
        out = (eng, px[:k], lb[:k])
        px_buf, lb_buf, count = ([px[k:]], [lb[k:]], count - k) if k < count else ([], [], 0)
        return out

    for _, batch in _OwnedBatches(dataloader, limit, rank, ws, sharded, progress, desc):
        px, labels = batch["pixel_values"], batch["labels"]
        if eng is None:
            need = max(chunk_images, int(px.size(0)))
            eng = _resolve(engine, max(need, capacity(need) if capacity is not None else 0))
        cap = min(eng.max_images, chunk_images) if chunk_images > 0 else min(eng.max_images, int(px.size(0)))
        px_buf.append(_pixels_to_device(batch, eng.device))
        lb_buf.append(_to_device(labels, eng.device, torch.int64))
        count += int(px.size(0))
        while count >= cap:
            yield cut(cap)
    if count:
        yield cut(count)


def best_eval_chunk(tokens: int, cap: int, n_cu: int = 256, tile_m: int = 256, col_tiles=(3, 3, 9, 12)) -> int:
    """Images per evaluation forward that waste the fewest CU-rounds of the persistent 256x256 GEMMs: the four
    projections of a block have col_tiles column tiles each (B/16: proj 3, fc2 3, QKV 9, fc1 12) and
    ceil(n*tokens/256) row tiles; a launch of T tiles occupies ceil(T / n_cu) rounds of the chip.  Among the chunk
    sizes whose utilisation is within 2 % of the best, the SMALLEST wins (its activations stay in the 256 MiB
    Infinity Cache between producer and consumer kernels)."""
    best, scored = None, []
    for n in range(min(16, cap), cap + 1):
        rows = -(-n * tokens // tile_m)
        used = sum(rows * c for c in col_tiles)
        paid = sum(-(-rows * c // n_cu) * n_cu for c in col_tiles)
        scored.append((used / paid * (n * tokens) / (rows * tile_m), n))
    if not scored:
        return cap
    top = max(u for u, _ in scored)
    return min(n for u, n in scored if u >= top - 0.02)


@torch.no_grad()
def top1_counts(engine, dataloader, *, max_batches=None, progress=False, process_group=None,
                attn_skip: Optional[Sequence[int]] = None, chunk_images: Optional[int] = None,
                sharded: bool = False) -> Tuple[int, int]:
    """Reference src/vit_pruning.py:325-373 as integer counts (correct, total)."""
    rank, ws = _dist.world(process_group)
    correct_dev = None
    total = 0
    for eng, px, labels in _chunks(engine, dataloader, max_batches, progress, "eval", rank, ws,
                                   chunk_images or DEFAULT_EVAL_CHUNK_IMAGES, sharded=sharded):
        if correct_dev is None:
            correct_dev = torch.zeros(1, dtype=torch.int64, device=eng.device)
        n = int(labels.size(0))
        x = eng.embed(px)
        eng.layers(x, n, 0, eng.depth - 1, attn_skip)
        eng.tail(x, n, attn_skip, labels=labels, correct=correct_dev)      # last block + head on the CLS rows
        total += n
    if correct_dev is not None:       # (no host list -> device tensor here: that H2D copy would wait for the stream)
        counts = torch.cat([correct_dev, torch.full((1,), total, dtype=torch.int64, device=correct_dev.device)])
    else:
        counts = torch.tensor([0, total], dtype=torch.int64)
    if ws > 1 or (_dist.FORCE_COLLECTIVES and _dist._initialised()):
        counts = _dist.all_reduce_counts(counts.to(_dist._default_device(process_group)), process_group)
    c = counts.to("cpu")
    return int(c[0]), int(c[1])


def slab_rows(tokens: int, n: int, group: int) -> int:
    """Rows one SLOT of n images takes in the slab layout with `group` images per slab, every slab padded (n a multiple of group):
    slot s of a layer-major launch then begins at slab s * n / group of the launch's row map (csrc/common.hip.h RowMap)."""
    mpad = -(-group * tokens // SLAB_ALIGN) * SLAB_ALIGN
    return (n // group) * mpad


def _search_chunk(eng, px, labels, n: int, L: int, cands: Sequence[int], removed: Sequence[int], counts_dev: torch.Tensor, *,
                  slots: int, lm: bool, group: int = 0, score=None, aux_engine=None, aux_stream=None, aux_lead: float = 0.0,
                  defer_tail: bool = False, extra_px=None):
    """One chunk of the depth search: the baseline and every candidate of `cands` over the n images `px` (a tensor or a list of
    batches); correct counts are ADDED into counts_dev ([L + 1]: candidates, then the baseline).
    `group` > 0: the streams are in the SLAB layout (one slab per dataloader batch of `group` images, n a multiple of it).
    `score` = (site, chain): the baseline is the stage-1 pass too — its fc1 launches carry the hook (slot 0 of the layer-major launch
    through ssp2_layers_prefix) and the per-batch score sums f32 [n / group, L, score_ld] are returned.
    `extra_px` (scored layer-major chunks only): n_x more images (whole slabs of `group`) that are ONLY hooked — calibration batches the search
    does not take.  They ride in FRONT of slot 0 in every launch of the baseline (buffer: extras | slot 0 | slot 1 ...; the hooked prefix of a
    launch is extras + slot 0), so they need no launches of their own; their score rows come first in the returned tensor.
    Returns (scores or None, tails or None): with `defer_tail` (scored layer-major chunks) the CLS-only tails — which only the counts
    need — are handed back as a callable instead of being enqueued, so that a caller can send the finished scores on their way first."""
    cand_set, removed_set = set(cands), set(removed)
    removed = sorted(removed_set)
    if score is not None and (removed or group <= 0 or n % group):
        raise ValueError("a scored baseline is the DENSE model in whole slabs")
    site, chain = score if score is not None else ("none", "fp32")
    g = int(group)
    kw = {"score_group": g} if g > 0 else {}                      # (a stand-in engine of the CPU tests knows no layouts)
    tkw = {"group": g} if g > 0 else {}
    rows = slab_rows(eng.tokens, n, g) if g > 0 else None
    n_x = 0
    if extra_px is not None:
        n_x = sum(int(p.shape[0]) for p in extra_px) if isinstance(extra_px, (list, tuple)) else int(extra_px.shape[0])
        if not (lm and score is not None and g > 0 and n_x % g == 0 and getattr(eng, "batch_lists", False)):
            raise ValueError("extra (hook-only) images ride on a scored layer-major chunk, in whole slabs")
    x0 = slab_rows(eng.tokens, n_x, g) if n_x else 0          # rows the extras take in front of slot 0
    bs = eng.new_scores((n + n_x) // g) if score is not None else None
    xb = None
    if lm and getattr(eng, "batch_lists", False):
        # the embedding lands straight in slot 0 of the slot buffer (round 2 embedded into its own tensor and copied it over)
        rows = eng.rows(n) if rows is None else rows
        spare = (x0 + rows) if score is not None else 0       # behind the slots: where the hooked LAST block of the baseline (and the extras) runs out of place
        xb = torch.empty(x0 + slots * rows + spare, eng.dim, dtype=torch.float32, device=eng.device)
        if g > 0:
            pad = rows - (eng.rows(n, g) if g < n else n * eng.tokens)      # the last slab's pad rows of every slot: finite values for the row-wise kernels
            if pad:
                xb[x0:x0 + slots * rows].view(slots, rows, eng.dim)[:, rows - pad:, :].zero_()
                if x0:
                    xb[x0 - pad:x0].zero_()
                if spare:
                    xb[-pad:].zero_()
        if n_x:
            both = (list(extra_px) if isinstance(extra_px, (list, tuple)) else [extra_px]) + (list(px) if isinstance(px, (list, tuple)) else [px])
            eng.embed(both if BATCH_LISTS else torch.cat(both, 0), x=xb[:x0 + rows], group=g)
            x = xb[x0:x0 + rows]
        else:
            x = eng.embed(px, x=xb[:rows], group=g)
    else:
        x = eng.embed(px, group=g) if g > 0 else eng.embed(px)
    if lm:
        # Layer-major search: the baseline (slot 0) and the snapshots (the k-th candidate to start in slot k) sit
        # side by side in one buffer, and at block l the baseline and every candidate already under way (c < l) run
        # the block in ONE launch of (k + 1)*n images — same per-image arithmetic in the same order, l + 2 launches
        # per block become 2, and the persistent GEMMs lose their partial last round ((l+1)*246.25 row panels
        # instead of 246.25).  Candidate l itself runs block l alone (its attention is bypassed).  Blocks in
        # `removed` (earlier rounds of the iterative search) are bypassed for every slot alike.  Needs an engine
        # workspace for slots*n images.
        if xb is None:
            rows = x.shape[0]
            xb = torch.empty(slots * rows, x.shape[1], dtype=x.dtype, device=x.device)
            xb[:rows].copy_(x)
        started = []
        s0 = xb[x0:]                                          # the slots (slot 0 first); the extras sit in xb[:x0], in front of them
        for l in range(L - 1):
            k = len(started)
            if l in cand_set:
                # candidate l: block l without its attention, started STRAIGHT from the baseline's stream (slot 0 still holds
                # the input of block l) into its own slot — the fc2 epilogue reads slot 0 and writes slot k + 1
                # (ssp2_layers_from).  Round 2 copied the 194 MB stream into the slot first: 11 copies per step.
                if OUT_OF_PLACE_START:
                    eng.layers(s0[(k + 1) * rows:(k + 2) * rows], n, l, l + 1, [l], x_in=s0[:rows], **kw)
                else:
                    s0[(k + 1) * rows:(k + 2) * rows].copy_(s0[:rows])
                    eng.layers(s0[(k + 1) * rows:(k + 2) * rows], n, l, l + 1, [l], **kw)
            skip = [l] if l in removed_set else None
            if score is not None:
                eng.layers(xb[:x0 + (k + 1) * rows], n_x + (k + 1) * n, l, l + 1, skip, site, chain, bs, g, score_images=n_x + n)
            else:
                eng.layers(xb[:(k + 1) * rows], (k + 1) * n, l, l + 1, skip, **kw)
            if l in cand_set:
                started.append(l)
        if score is not None:
            # the hook of the LAST block, out of place into the spare area (slot 0 stays what the tails read) and BEFORE the tails: the
            # scores are then complete while the latency-bound tails still run — a caller's host mask step overlaps them
            at = x0 + slots * rows
            spare_x = xb[at:at + x0 + rows] if xb.shape[0] >= at + x0 + rows else torch.empty(x0 + rows, xb.shape[1], dtype=xb.dtype, device=xb.device)
            eng.layers(spare_x, n_x + n, L - 1, L, None, site, chain, bs, g, scores_only=True, x_in=xb[:x0 + rows])
        xb = s0                                               # from here on: the slots only (what the tails read)

        def tails():
            # the baseline (slot 0) and every candidate under way meet the same last block and classifier: ONE tail over all the
            # slots (ssp2_tail_slots) instead of one per slot — the tail's launches on n CLS rows are latency-bound, thirteen
            # of them per chunk were ~3 % of the step.  Slot s is counted in slot_counts[s] and added to its candidate's entry.
            if TAIL_SLOTS and getattr(eng, "batch_lists", False):
                slot_counts = torch.zeros(len(started) + 1, dtype=torch.int64, device=counts_dev.device)
                eng.tail(xb[:(len(started) + 1) * rows], n, removed, labels=labels, correct=slot_counts, slots=len(started) + 1, **tkw)
                counts_dev.index_add_(0, _device_index([L] + started, counts_dev.device), slot_counts)
            else:
                eng.tail(xb[:rows], n, removed, labels=labels, correct=counts_dev[L:L + 1], **tkw)
                for k, c in enumerate(started):
                    eng.tail(xb[(k + 1) * rows:(k + 2) * rows], n, removed, labels=labels, correct=counts_dev[c:c + 1], **tkw)
            if (L - 1) in cand_set:
                eng.tail(xb[:rows], n, removed + [L - 1], labels=labels, correct=counts_dev[L - 1:L], **tkw)
        if defer_tail:
            return bs, tails
        tails()
        return bs, None
    cache = {}
    for l in range(L - 1):
        if l in cand_set:
            cache[l] = x.clone()
        if score is not None:
            eng.layers(x, n, l, l + 1, None, site, chain, bs, g)
        else:
            eng.layers(x, n, l, l + 1, removed, **kw)
    # x now enters the last block: every pass finishes with the CLS-only tail, which leaves x untouched
    eng.tail(x, n, removed, labels=labels, correct=counts_dev[L:L + 1], **tkw)
    on_aux = set()
    if aux_engine is not None and aux_stream is not None and n <= aux_engine.max_images:
        load_main, load_aux = float(L - 1), float(aux_lead)          # greedy split, longest candidates first
        for c in sorted(cand_set):
            cost = float(L - 1 - c) + 0.2
            if load_aux + cost < load_main:
                on_aux.add(c); load_aux += cost
            else:
                load_main += cost
        main = torch.cuda.current_stream(eng.device)
        aux_stream.wait_stream(main)                                   # snapshots and labels are ready
    for c in cands:
        e, ctx = (aux_engine, torch.cuda.stream(aux_stream)) if c in on_aux else (eng, contextlib.nullcontext())
        with ctx:
            if c == L - 1:
                e.tail(x, n, removed + [c], labels=labels, correct=counts_dev[c:c + 1], **tkw)
                continue
            xc = cache.pop(c)
            if c in on_aux:
                xc.record_stream(aux_stream)
            e.layers(xc, n, c, L - 1, removed + [c], **kw)
            e.tail(xc, n, removed + [c], labels=labels, correct=counts_dev[c:c + 1], **tkw)
    if on_aux:
        x.record_stream(aux_stream); labels.record_stream(aux_stream)
        torch.cuda.current_stream(eng.device).wait_stream(aux_stream)
    if score is not None:                  # the hook of the LAST block on the baseline's stream, which nothing reads any more
        eng.layers(x, n, L - 1, L, None, site, chain, bs, g, scores_only=True)
    return bs, None


@torch.no_grad()
def depth_search_counts(engine, dataloader, depth: int, *, batch_limit: Optional[int] = 5, process_group=None,
                        removed: Sequence[int] = (), candidates: Optional[Sequence[int]] = None,
                        chunk_images: Optional[int] = None, defer: bool = False, aux_engine=None, aux_stream=None,
                        aux_lead: float = 0.0, batch_candidates="auto", sharded: bool = False):
    """One pass over the eval batches that yields the baseline AND every candidate's correct-count.

    The reference deep-copies the model and re-runs the whole forward per candidate (mask_conjunction.py:339-355,
    src/vit_pruning.py:477-494).  Here the residual stream entering each block is cached during the baseline
    forward, and candidate i (attention of block i bypassed) restarts from the cached input of block i: blocks
    0..i-1 are bit-identical to the baseline, so the result equals a full re-run while executing
    L(L+1)/2 + L block passes per batch instead of L(L+1).
    Returns (baseline_correct, [candidate_correct per block], total); with `defer=True` a zero-argument callable
    that waits for the device and returns that tuple.

    `aux_engine` + `aux_stream`: a second engine (same weights, own workspace) on a second HIP stream takes a share
    of the candidates once the baseline has produced their snapshots — candidates are independent, and two streams
    let the memory-bound kernels of one (LayerNorm, epilogue tails) run beside the matrix-bound kernels of the other.
    `aux_lead` = work already queued on that stream, in block passes of this chunk size (bench: the stage-1 launch);
    the split balances (baseline + main candidates) against (lead + aux candidates).  Integer counts: same result.

    `batch_candidates`: "auto" (default) runs the LAYER-MAJOR order below whenever the engine's workspace can hold the
    baseline and all candidates side by side (an engine factory is asked for that capacity if it fits the budget of
    `workspace_budget_bytes`), True requires it when the engine allows, False keeps the candidate-major order."""
    rank, ws = _dist.world(process_group)
    L = depth
    counts_dev = None
    total = 0
    removed = sorted(set(int(r) for r in removed))
    cands = list(range(L)) if candidates is None else [int(c) for c in candidates]
    cand_set, removed_set = set(cands), set(removed)
    chunk = chunk_images or DEFAULT_EVAL_CHUNK_IMAGES
    slots = 1 + sum(1 for c in cand_set if c < L - 1)         # the baseline + every candidate that starts before the tail
    want_lm = (batch_candidates is True or batch_candidates == "auto") and aux_engine is None and slots > 1
    cap_fn = (lambda need: layer_major_images(engine, slots, need)) if want_lm else None
    for eng, px, labels in _chunks(engine, dataloader, batch_limit, False, "attn search", rank, ws, chunk, cap_fn, sharded):
        if counts_dev is None:
            counts_dev = torch.zeros(L + 1, dtype=torch.int64, device=eng.device)
        n = int(labels.size(0))
        _search_chunk(eng, px, labels, n, L, cands, removed, counts_dev, slots=slots, lm=want_lm and eng.max_images >= slots * n,
                      aux_engine=aux_engine, aux_stream=aux_stream, aux_lead=aux_lead)
        total += n
    if counts_dev is None:
        counts = torch.zeros(L + 2, dtype=torch.int64)
    else:
        # torch.full, not torch.tensor([...], device=...): a host list becomes a pageable H2D copy, which makes the host
        # wait for everything enqueued before it (the whole search)
        counts = torch.cat([counts_dev, torch.full((1,), total, dtype=torch.int64, device=counts_dev.device)])
    if ws > 1 or (_dist.FORCE_COLLECTIVES and _dist._initialised()):
        counts = _dist.all_reduce_counts(counts.to(_dist._default_device(process_group)), process_group)

    def finish():
        c = counts.to("cpu").tolist()
        return c[-2], c[:-2], c[-1]
    return finish if defer else finish()


# what the last prune_pass of this process did (bench.py prints it: a run that silently fell back to the candidate-major order,
# because its engine was sized without the slab padding, is then visible)
# hook-only calibration batches (inside the score limit, outside the search's) ride in front of slot 0 of the last scored chunk instead of
# getting launches of their own (0: the scores-only forward of rounds 1-5a; same scores either way)
FUSE_EXTRAS = os.environ.get("SSP2_FUSE_EXTRAS", "1") != "0"
PASS_STATS = {"fused_chunks": 0, "fused_layer_major": 0, "search_only_chunks": 0, "scores_only_launches": 0, "hook_only_batches_fused": 0,
              "score_batches_owned": 0, "search_batches_owned": 0}      # the last two: 0 on a rank that idles in that stage (fewer batches than ranks)


def lm_capacity_images(tokens: int, slots: int, n: int, group: int) -> int:
    """Engine capacity (images) whose workspace holds `slots` streams of n images side by side in the slab layout
    (csrc/engine.hip: rows_cap = max_images * tokens + 16 * 256)."""
    rows = slots * slab_rows(tokens, n, group)
    return max(slots * n, -(-(rows - 16 * 256) // tokens))


@torch.no_grad()
def prune_pass(engine, dataloader, d_ints: Sequence[int], site: str, depth: int, *, score_limit: Optional[int] = None,
               search_limit: Optional[int] = 5, score_chain: str = "fp32", process_group=None,
               chunk_images: Optional[int] = None, eval_chunk_images: Optional[int] = None, defer: bool = False,
               sharded: bool = False, candidates: Optional[Sequence[int]] = None, progress: bool = False,
               batch_candidates="auto"):
    """ONE walk over ONE loader for BOTH stages: the stage-1 scores (stage1_scores) and the depth search's counts
    (depth_search_counts), with the search's dense baseline forward doubling as the stage-1 pass.

    The reference's plug-in walks one loader with one batch_limit for both: `Auto2SSPInterface._compute_mlp_importance` hooks
    `self.dl[:batch_limit]` (adaptation-for-Pures-framework/mask_conjunction.py:276-281), `_compute_att_depth_importance` evaluates the
    dense model (:327) and every candidate (:345) on `self.dl[:batch_limit]` again; `fit()` runs both (:359-362) and the CLI builds
    the object once (auto_2ssp.py:765-775).  The dense forward over those batches is computed twice there (and L more times with one
    attention bypassed).  Here a batch that both stages want goes through ONE dense forward: it is laid out as its own 128-row-aligned
    slab (what pins a sample's partial sums of squares, see stage1_scores), the fc1 launches of that baseline carry the stage-1 hook,
    and the candidates of the layer-major search start from its per-block stream as before.  Scores are bit-identical to
    stage1_scores (slab position) and counts to depth_search_counts (no result of a row depends on the launch it is part of).

    `score_limit` / `search_limit`: GLOBAL batch limits of the two stages (None = the whole loader).  Batches inside both go through
    the fused forward; batches only stage 1 wants get the scores-only forward, batches only the search wants the plain search.
    A loader that reshuffles per epoch (the reference's calibration loader, auto_2ssp.py:348) is iterated ONCE and that one order
    feeds both stages; the reference draws a fresh order per walk, so there the two stages see different random subsets.
    Batches need "labels" wherever the search takes them.

    Returns (scores, (baseline_correct, [candidate_correct], total)); with `defer=True` two zero-argument callables that wait."""
    rank, ws = _dist.world(process_group)
    sharded = sharded or bool(getattr(dataloader, "sharded", False))
    L = depth
    cands = list(range(L)) if candidates is None else [int(c) for c in candidates]
    slots = 1 + sum(1 for c in set(cands) if c < L - 1)
    want_lm = (batch_candidates is True or batch_candidates == "auto") and slots > 1
    chunk_images = DEFAULT_CHUNK_IMAGES if chunk_images is None else chunk_images
    eval_chunk = eval_chunk_images or DEFAULT_EVAL_CHUNK_IMAGES
    limit = None if (score_limit is None or search_limit is None) else max(int(score_limit), int(search_limit))
    walk = _OwnedBatches(dataloader, limit, rank, ws, sharded, progress, "2SSP pass")
    for k_ in PASS_STATS:
        PASS_STATS[k_] = 0
    local: List[Tuple[int, torch.Tensor]] = []
    seen_score = [0, 0]                      # batches / samples this rank SAW inside the score limit
    state = {"eng": None, "counts": None, "total": 0, "s1": None, "tails": None}
    fused: List[Tuple[int, torch.Tensor, torch.Tensor, bool]] = []      # (global index, pixels, labels, scored?) of the open search chunk
    extras: List[Tuple[int, torch.Tensor]] = []                          # hook-only batches that ride on it (global index, pixels)

    def resolve(n: int):
        base = max(chunk_images, n)
        n_chunk = max(1, eval_chunk // n) * n
        tokens = getattr(engine, "tokens", None)
        if callable(engine):
            need = base
            per_image = getattr(engine, "bytes_per_image", None)
            if want_lm and per_image is not None:
                tk = tokens() if callable(tokens) else tokens
                lm_imgs = lm_capacity_images(int(tk), slots, n_chunk, n) if tk else int(slots * n_chunk * 1.05) + 64
                if lm_imgs * per_image() <= workspace_budget_bytes():
                    need = max(need, lm_imgs)
                    # ... and room for hook-only batches in front of slot 0 (up to one slot's worth), when stage 1 takes more batches than the search
                    more = (lm_capacity_images(int(tk), slots + 1, n_chunk, n) if tk else lm_imgs + n_chunk + 64)
                    if FUSE_EXTRAS and (score_limit is None or search_limit is None or int(score_limit) > int(search_limit)) \
                            and more * per_image() <= workspace_budget_bytes():
                        need = max(need, more)
            eng = engine(need)
        else:
            eng = engine
        state["eng"] = eng
        state["counts"] = torch.zeros(L + 1, dtype=torch.int64, device=eng.device)
        state["s1"] = _Chunker(min(eng.max_images, base), eng.device, max_batches=MAX_SLABS)
        return eng

    def fits_lm(eng, n: int, g: int, n_x: int = 0) -> bool:
        if not want_lm or eng.max_images < slots * n + n_x:
            return False
        if not getattr(eng, "prefix_scoring", False):
            return n_x == 0
        return eng.max_images * eng.tokens + 16 * 256 >= slots * slab_rows(eng.tokens, n, g) + (slab_rows(eng.tokens, n_x, g) if n_x else 0)

    def run_pending():          # the CLS-only tails of the previous scored chunk (held back so that its scores could leave first)
        t, state["tails"] = state["tails"], None
        if t is not None:
            t()

    def flush_fused():
        if not fused:
            return
        eng = state["eng"]
        items = list(fused); fused.clear()
        g = int(items[0][1].size(0))
        n = sum(int(it[1].size(0)) for it in items)
        scored = items[0][3]
        as_list = BATCH_LISTS and getattr(eng, "batch_lists", False)
        px = items[0][1] if len(items) == 1 else ([it[1] for it in items] if as_list else torch.cat([it[1] for it in items], 0))
        labels = items[0][2] if len(items) == 1 else torch.cat([it[2] for it in items], 0)
        run_pending()
        ex = list(extras); extras.clear()
        n_x = sum(int(e[1].size(0)) for e in ex)
        if ex and not (scored and getattr(eng, "prefix_scoring", False) and fits_lm(eng, n, g, n_x)):
            for e in ex:                                      # (cannot ride after all: their own scores-only forward, as before)
                vec = eng.forward_scores(e[1], site, score_chain, int(e[1].size(0)))
                PASS_STATS["scores_only_launches"] += 1
                local.append((e[0], vec[0]))
            ex, n_x = [], 0
        if getattr(eng, "prefix_scoring", False):
            if scored:
                lm_now = fits_lm(eng, n, g, n_x)
                PASS_STATS["fused_chunks"] += 1; PASS_STATS["fused_layer_major"] += int(lm_now); PASS_STATS["hook_only_batches_fused"] += len(ex)
                bs, state["tails"] = _search_chunk(eng, px, labels, n, L, cands, (), state["counts"], slots=slots, lm=lm_now,
                                                   group=g, score=(site, score_chain), defer_tail=True,
                                                   extra_px=([e[1] for e in ex] if len(ex) > 1 else ex[0][1]) if ex else None)
                for k, e in enumerate(ex):                    # the extras' score rows come first
                    local.append((e[0], bs[k]))
                for k, it in enumerate(items):
                    local.append((it[0], bs[len(ex) + k]))
            else:
                PASS_STATS["search_only_chunks"] += 1
                _search_chunk(eng, px, labels, n, L, cands, (), state["counts"], slots=slots, lm=want_lm and eng.max_images >= slots * n)
        else:
            # an engine without the prefix hook (the stand-in engine of the CPU tests): two forwards, the same results
            if scored:
                vec = eng.forward_scores(px if not isinstance(px, list) else torch.cat(px, 0), site, score_chain, g)
                for k, it in enumerate(items):
                    local.append((it[0], vec[k]))
            _search_chunk(eng, px, labels, n, L, cands, (), state["counts"], slots=slots, lm=want_lm and eng.max_images >= slots * n)
        state["total"] += n

    def flush_s1():
        ch, eng = state["s1"], state["eng"]
        if ch is None or not ch.items:
            return
        idxs, group, px, _ = ch.take(as_list=BATCH_LISTS and getattr(eng, "batch_lists", False))
        vec = eng.forward_scores(px, site, score_chain, group)
        PASS_STATS["scores_only_launches"] += 1
        for k, i in enumerate(idxs):
            local.append((i, vec[k]))

    for gi, batch in walk:
        n = int(batch["pixel_values"].size(0))
        in_search = search_limit is None or gi < int(search_limit)
        in_score = score_limit is None or gi < int(score_limit)
        eng = state["eng"] or resolve(n)
        if callable(engine) and n > eng.max_images:       # a batch larger than everything seen so far: what is open runs on the old engine, then a larger one
            flush_fused(); flush_s1(); run_pending()
            counts_so_far = state["counts"]
            eng = resolve(n)
            state["counts"] = counts_so_far.to(eng.device)
        PASS_STATS["score_batches_owned"] += int(in_score); PASS_STATS["search_batches_owned"] += int(in_search)
        if in_search:
            if "labels" not in batch:
                raise KeyError("prune_pass: a batch inside the search limit carries no 'labels'")
            cap = min(max(1, eval_chunk // n) * n, eng.max_images)
            if fused and (int(fused[0][1].size(0)) != n or fused[0][3] != in_score
                          or sum(int(it[1].size(0)) for it in fused) + n > cap):
                flush_fused()
            fused.append((gi, _pixels_to_device(batch, eng.device), _to_device(batch["labels"], eng.device, torch.int64), in_score))
        elif in_score:
            n_open = sum(int(it[1].size(0)) for it in fused)
            n_x = sum(int(e[1].size(0)) for e in extras)
            if (FUSE_EXTRAS and fused and fused[0][3] and int(fused[0][1].size(0)) == n and n_x + n <= n_open
                    and getattr(eng, "prefix_scoring", False) and fits_lm(eng, n_open, n, n_x + n)):
                # a hook-only batch while a scored chunk is still open (the search's batches come first in the loader): it rides in front of
                # that chunk's slot 0 — no launches of its own — as long as the extras stay within one slot's worth and the workspace holds them
                extras.append((gi, _pixels_to_device(batch, eng.device)))
                continue
            flush_fused()
            if state["s1"].full_for(n):
                flush_s1()
            state["s1"].add(gi, batch)
    flush_fused()
    flush_s1()
    # what this rank saw inside the score limit (all batches of a plain loader, its own of a sharded one)
    for gi_seen, n_seen in walk.sizes:
        if score_limit is None or gi_seen < int(score_limit):
            seen_score[0] += 1; seen_score[1] += n_seen
    scores = _reduce_scores(local, seen_score[0], seen_score[1], d_ints, score_chain, process_group, defer, sharded)
    run_pending()

    if state["counts"] is None:
        counts = torch.zeros(L + 2, dtype=torch.int64)
    else:
        counts = torch.cat([state["counts"], torch.full((1,), state["total"], dtype=torch.int64, device=state["counts"].device)])
    if ws > 1 or (_dist.FORCE_COLLECTIVES and _dist._initialised()):
        counts = _dist.all_reduce_counts(counts.to(_dist._default_device(process_group)), process_group)

    def finish_search():
        c = counts.to("cpu").tolist()
        return c[-2], c[:-2], c[-1]
    return (scores, finish_search) if defer else (scores, finish_search())


def impacts_from_counts(base: int, cand: Sequence[int], total: int) -> List[float]:
    """impact_i = max(0, baseline - acc_i) with the reference's float arithmetic (mask_conjunction.py:329-348)."""
    baseline = float(base / max(1, total))
    return [max(0.0, baseline - float(c / max(1, total))) for c in cand]


_MASK_POOL = None
MASK_THREADS = os.environ.get("SSP2_MASK_THREADS", "1") != "0"       # 0: the blocks' mask steps one after the other (same results; for A/B runs)


def mask_pool():
    """Four worker threads, started once per process: the a7 mask step of the blocks is independent host work (torch releases the
    interpreter lock inside argsort / sort), and starting threads costs more than the step itself."""
    global _MASK_POOL
    if _MASK_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _MASK_POOL = ThreadPoolExecutor(max_workers=4, thread_name_prefix="ssp2-mask")
    return _MASK_POOL


def cut_masks(imps: Sequence[torch.Tensor], n_prune: Sequence[int]) -> List[torch.Tensor]:
    """The a7 mask step (reference src/vit_pruning.py:286-295) of every block: keep = sort(argsort(imp, descending)[:d_int - t]), int16 mask with
    1 = prune — the same torch calls on the same 1-D tensors, blocks side by side on the mask pool (12 x 0.2 ms in a row were 2.5 ms of host time
    at the end of a prune, more than the CLS-only tails they are meant to hide behind)."""
    def one(args):
        imp, t = args
        width = imp.numel()
        keep, _ = torch.sort(torch.argsort(imp, descending=True)[: width - int(t)])
        m = torch.ones(width, dtype=torch.int16)
        m[keep] = 0
        return m
    work = list(zip(imps, n_prune))
    if len(work) >= 4 and MASK_THREADS and all(i.device.type == "cpu" for i in imps):
        return list(mask_pool().map(one, work))
    return [one(w_) for w_ in work]


def select_for_targets(imps: Sequence[torch.Tensor], impact: torch.Tensor, plans: Sequence, min_remaining: int = 256,
                       site: Optional[str] = None) -> List[dict]:
    """The host half of a prune for one or several targets from ONE stage-1 pass and ONE search (BASELINE configs[2]:
    25 / 37.5 / 50 % — the sweep convention of main.py:152-157): per plan the a7 mask step on the final score vectors
    (reference src/vit_pruning.py:273-295: keep = sort(argsort(imp, descending)[:d_int - t]), 1 = prune; t clamped per block so
    that at least `min_remaining` neurons stay, :279-281 — the default is prune_vit_mlp_width's), the a9 block
    selection `torch.argsort(att_imp)[:K]` (auto_2ssp.py:857) and the cut-margin table of those masks (`site`: the hook site
    the scores were taken at — the post-GELU site gets the wider tie band, ssp2vit/mask_parity.py).
    Returns [{"target", "masks": [int16 [d_int]] * L, "blocks": sorted [int], "mask_parity": {...}}]."""
    impact = torch.as_tensor(impact, dtype=torch.float32)
    out = []
    for p in plans:
        t = int(p.per_block_neurons_to_prune)
        masks = cut_masks(imps, [max(0, min(t, imp.numel() - int(min_remaining))) for imp in imps])          # reference :279-281 (the clamp), :286-295
        blocks = sorted(int(i) for i in torch.argsort(impact)[: int(p.blocks_to_prune)])
        out.append({"target": float(p.target_sparsity), "masks": masks, "blocks": blocks,
                    "mask_parity": mask_parity_report(imps, [t] * len(imps), min_remaining=int(min_remaining), site=site)})
    return out
