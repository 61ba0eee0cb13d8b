"""Host-side mirror of the reference's function API (/root/reference/src/vit_pruning.py `__all__`, :10-21):
same names, argument meaning, return types and error behaviour — but the two hot loops (stage-1 activation
scoring, stage-2 attention-removal search) run on the MI355X through libssp2vit instead of calling
`model(px)` under autocast.

What stays on the host, unchanged in meaning (SURVEY.md §1 "seams"): the mask step on the final 12x3072 score
vectors (same `torch.argsort` call, so equal scores give equal masks), the in-place weight slicing of the user's
module, module surgery for the attention bypass, and the planner arithmetic.

Extra keyword-only arguments (not in the reference): `score_chain`, `process_group`, `engine`.
"""
from __future__ import annotations

import json
import os
import time
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import core as _core
from . import dist as _dist
from . import weights as _weights
from ._lib import check as _lib_check
from .mask_parity import MASK_PARITY_EPS, mask_parity_report
from .planner import ModelStats, TwoSSPPlan, plan_from_stats

__all__ = [
    "prune_vit_mlp_width", "evaluate_top1", "prune_vit_attention_blocks", "plan_2ssp_allocation",
    "count_total_params", "count_block_params", "compute_actual_sparsity", "save_cifar_adapter", "load_cifar_adapter", "save_report", "TwoSSPPlan",
    "mask_parity_report", "MASK_PARITY_EPS",
]


# ----------------------------------------------------------------------------- anatomy (reference :27-75)
def _get_encoder(vit_model):
    base = getattr(vit_model, "vit", None)
    if base is None:
        base = getattr(vit_model, "base_model", None) or vit_model
    return base.encoder if hasattr(base, "encoder") else base


def _blocks(vit_model):
    """(blocks, kind).  kind "hf" / "timm" are the two anatomies the reference duck-types over (:27-45); "hf5" is the
    renamed layout of transformers >= 5 (`vit.layers[i].{attention.{q,k,v,o}_proj, mlp.fc1, mlp.fc2}`), on which the
    reference itself raises AttributeError — accepted here because it is the only HF ViT installable today."""
    enc = _get_encoder(vit_model)
    if hasattr(enc, "layer"):
        return list(enc.layer), "hf"
    if hasattr(enc, "blocks"):
        return list(enc.blocks), "timm"
    if hasattr(enc, "layers") and all(hasattr(b, "mlp") and hasattr(b.mlp, "fc1") for b in enc.layers):
        return list(enc.layers), "hf5"
    raise AttributeError("Unsupported ViT model structure: expected encoder.layer or blocks")


def _gather_mlp_pairs(vit_model) -> List[Tuple[nn.Linear, nn.Linear]]:
    blocks, kind = _blocks(vit_model)
    if kind == "hf":
        return [(b.intermediate.dense, b.output.dense) for b in blocks]
    return [(b.mlp.fc1, b.mlp.fc2) for b in blocks]          # timm and hf5 name the pair alike


def _get_hidden_and_inter_sizes(vit_model) -> Tuple[int, List[int]]:
    pairs = _gather_mlp_pairs(vit_model)
    hidden = pairs[0][0].weight.size(1) if pairs else getattr(vit_model.config, "hidden_size", None)
    return hidden, [p[0].weight.size(0) for p in pairs]


def count_total_params(model: nn.Module) -> int:
    return sum(p.numel() for p in model.parameters())


def count_block_params(model: nn.Module) -> List[int]:
    return [sum(p.numel() for p in b.parameters()) for b in _blocks(model)[0]]


def compute_actual_sparsity(before_params: int, after_params: int) -> float:
    return 0.0 if before_params <= 0 else (before_params - after_params) / before_params


def _attn_module(block, kind):
    return getattr(block, "attn" if kind == "timm" else "attention", None)


def _model_stats(vit_model) -> ModelStats:
    blocks, kind = _blocks(vit_model)
    hidden, inters = _get_hidden_and_inter_sizes(vit_model)
    attn = []
    for b in blocks:
        a = _attn_module(b, kind)
        attn.append(0 if a is None else sum(p.numel() for p in a.parameters()))
    ffn = [sum(p.numel() for p in i.parameters()) + sum(p.numel() for p in o.parameters())
           for i, o in _gather_mlp_pairs(vit_model)]
    return ModelStats(count_total_params(vit_model), hidden, inters, attn, ffn)


def plan_2ssp_allocation(vit_model, target_sparsity: float, min_remaining: int = 256,
                         forced_blocks: Optional[int] = None) -> TwoSSPPlan:
    """Reference :585-769.  Pure host arithmetic on parameter counts."""
    st = _model_stats(vit_model)
    if st.hidden is None or len(st.inter_sizes) != len(count_block_params(vit_model)):
        raise RuntimeError("Unable to determine hidden/intermediate sizes for planning.")
    return plan_from_stats(st, target_sparsity, min_remaining, forced_blocks)


# ----------------------------------------------------------------------------- engine cache
_ENGINES: Dict[int, Tuple[Tuple, Any]] = {}


def _fingerprint(model) -> Tuple:
    return tuple((id(p), p._version, tuple(p.shape)) for p in model.parameters()) + \
           tuple(type(m).__name__ for m in model.modules())


# Arithmetic of the engines the reference-named functions build: "bf16" (the reference's CPU-autocast arithmetic, the parity mode) or,
# opt-in, "fp8" (BASELINE configs[4]: QKV / fc1 / fc2 / out-projection of launches with >= 4096 token rows on e4m3 MFMA operands — a
# tolerance mode).  SSP2_PRECISION sets the default; the CLI's --precision sets this variable.
DEFAULT_PRECISION = os.environ.get("SSP2_PRECISION", "bf16")
_FP8_SCALES: Dict[int, List[float]] = {}          # id(model) -> calibrated attention hand-off scales (calibrate_fp8), re-applied when an engine is rebuilt


def engine_for(model, device="cuda", max_images: int = 64, precision: Optional[str] = None):
    """Build (or reuse) the HIP engine holding `model`'s current weights.  Raises without a GPU."""
    from .engine import VitEngine
    precision = precision or DEFAULT_PRECISION
    key = id(model)
    fp = _fingerprint(model) + (precision,)
    hit = _ENGINES.get(key)
    if hit is not None and hit[0] == fp and hit[1].max_images >= max_images:
        return hit[1]
    if hit is not None:                                   # this module's engine is stale (weights changed, or too small): it goes to the pool,
        _ENGINES.pop(key)                                 # where the lookup below finds it again if the geometry is still the same
        if ENGINE_POOL_MAX > 0 and hit[1].precision == "bf16" and getattr(hit[1], "h", None):
            _POOL.append(hit[1])
        else:
            hit[1].close()
    dev = torch.device(device if str(device) != "cuda" else f"cuda:{torch.cuda.current_device()}") \
        if torch.cuda.is_available() else torch.device(device)
    flat = _weights.from_module(model)
    eng = _from_pool(flat, dev, max(int(max_images), 1), precision)
    while len(_POOL) > ENGINE_POOL_MAX:
        _POOL.pop(0).close()
    if eng is None:
        eng = VitEngine(flat, device=dev, max_images=max(int(max_images), 1), precision=precision)
    eng.layout = _weights.detect_layout(model)
    if precision == "fp8" and key in _FP8_SCALES and len(_FP8_SCALES[key]) == eng.depth:
        for l, sc in enumerate(_FP8_SCALES[key]):
            _lib_check(eng.lib.ssp2_fp8_set_attn_scale(eng.h, l, float(sc)))
    _ENGINES[key] = (fp, eng)
    return eng


def calibrate_fp8(model, pixel_values: torch.Tensor, device="cuda", headroom: float = 4.0) -> List[float]:
    """fp8 engines only: measure the attention outputs of `pixel_values` (>= 4096 token rows per launch, e.g. 32 images of 224 x 224) on
    `model`'s engine and fix each block's e4m3 hand-off scale (VitEngine.calibrate_fp8).  The scales are remembered for this model
    object and re-applied whenever its engine is rebuilt (a larger workspace, changed weights keep the measured ranges)."""
    eng = engine_for(model, device, max_images=int(pixel_values.shape[0]), precision="fp8")
    scales = eng.calibrate_fp8(pixel_values.to(eng.device), headroom)
    _FP8_SCALES[id(model)] = scales
    return scales


# Engines that `release_engines()` set free, kept for the NEXT model of the same geometry (VitEngine.reload): a sweep that builds a fresh
# module per target (the CLI's --sparsity_rate -2, bench.py's API leg) then pays a weight ingest per prune, not an engine build plus the
# first use of 13 GB of fresh workspace.  At most ENGINE_POOL_MAX engines wait here (each holds its workspace in HBM); 0 switches the
# pool off (SSP2_ENGINE_POOL); `release_engines(free=True)` closes everything.
ENGINE_POOL_MAX = int(os.environ.get("SSP2_ENGINE_POOL", "1"))
_POOL: List[Any] = []


def _from_pool(flat, dev, max_images: int, precision: str):
    if precision != "bf16":
        return None
    for i, e in enumerate(_POOL):
        try:
            same = (str(e.device) == str(dev) and e.precision == precision and e.max_images >= max_images and e.lib_variant is None
                    and (e.img, e.patch, e.dim, e.heads, e.depth, e.classes) == tuple(int(flat[k]) for k in ("img", "patch", "dim", "heads", "depth", "classes"))
                    and abs(e.eps - float(flat.get("eps", 1e-6))) < 1e-15
                    and [int(d) for d in e.d_int] == [int(flat[f"fc1_w.{l}"].shape[0]) for l in range(e.depth)])
        except Exception:
            same = False
        if same:
            _POOL.pop(i)
            return e.reload(flat)
    return None


def release_engines(free: bool = False) -> None:
    """The engines built for live modules are set free: closed, or — bf16 engines, up to ENGINE_POOL_MAX of them — parked for the next
    model of the same geometry.  `free=True` closes the parked ones too."""
    for _, (_, e) in list(_ENGINES.items()):
        if not free and ENGINE_POOL_MAX > 0 and e.precision == "bf16" and getattr(e, "h", None):
            _POOL.append(e)
        else:
            e.close()
    _ENGINES.clear()
    while len(_POOL) > (0 if free else ENGINE_POOL_MAX):
        _POOL.pop(0).close()


def _tokens_of(model) -> int:
    for path in ("pos_embed", "vit.embeddings.position_embeddings", "embeddings.position_embeddings"):
        obj = model
        try:
            for name in path.split("."):
                obj = getattr(obj, name)
            return int(obj.shape[1])
        except AttributeError:
            continue
    return 197


class _EngineFactory:
    """callable(min_images) -> VitEngine for `model` (cached per model, rebuilt when its weights change), plus the
    workspace price of one image, which lets core.depth_search_counts decide whether the layer-major search fits."""

    def __init__(self, model, device):
        self.model, self.device = model, device
        self.tokens = _tokens_of(model)

    def __call__(self, n: int):
        return engine_for(self.model, self.device, max_images=n)

    def bytes_per_image(self) -> int:
        hidden, inters = _get_hidden_and_inter_sizes(self.model)
        tokens = _tokens_of(self.model)
        ld_int = (max(inters) + 63) // 64 * 64
        # csrc/engine.hip ssp2_create: LN output + qkv + attention output + FFN activation (bf16) per token row, the
        # im2col matrix, and the fp32 residual-stream snapshot the search keeps per slot
        return tokens * (2 * hidden + 6 * hidden + 2 * hidden + 2 * ld_int) + tokens * 1536 + tokens * hidden * 4


def _engine_factory(model, device, engine):
    if engine is not None:
        return engine
    return _EngineFactory(model, device)


# ----------------------------------------------------------------------------- a1/a2 stage-1 scores
@torch.no_grad()
def _compute_ffn_activation_importance(vit_model, dataloader, device: str = "cuda", batch_limit: Optional[int] = None,
                                       progress: bool = False, *, score_chain: str = "fp32", process_group=None,
                                       engine=None, defer: bool = False, sharded: bool = False) -> List[torch.Tensor]:
    """Reference :111-201 — mean over calibration samples of the per-sample token-L2 of every block's FFN
    intermediate activation (pre-GELU for timm-layout models, post-GELU for HF-layout ones).

    Returns List[L] of CPU tensors [d_int]: float32 with score_chain="fp32" (default; fp32 accumulators end to
    end), bfloat16 with score_chain="bf16_ref" (the reference's CPU-autocast rounding points)."""
    vit_model.eval()
    _, kind = _blocks(vit_model)                # raises AttributeError on unknown anatomy like the reference
    d_ints = [p[0].out_features for p in _gather_mlp_pairs(vit_model)]
    # a container built from a CONVERTED checkpoint (modules.EngineViT over an HF state dict) keeps the hook site of the
    # anatomy the weights came from: the reference would have hooked the HF module post-GELU (:135)
    site = getattr(vit_model, "ssp2_score_site", None) or _weights.score_site_for("timm" if kind == "timm" else "hf")
    return _core.stage1_scores(_engine_factory(vit_model, device, engine), dataloader, d_ints, site,
                               batch_limit=batch_limit, progress=progress, score_chain=score_chain,
                               process_group=process_group, defer=defer, sharded=sharded)


# ----------------------------------------------------------------------------- a7/a8 width prune (host consumer)
# Descending argsort of importance vectors, computed AHEAD of the mask step while the device is still busy: Auto2SSPInterface.fit() has the
# stage-1 scores on the host before the search's CLS-only tails have finished, and the a7 mask step that follows in prune_vit_mlp_width
# starts with exactly this argsort (reference :286) — 55 % of its host time, paid with the card idle when it runs inside the later call.
# An entry serves the very tensor it was computed from (same object, same in-place version counter) and nothing else; it is the result
# of the same torch call on the same data, so masks do not depend on whether the cache was hit (tests/test_host_cpu.py).
_ORDER_CACHE: Dict[int, tuple] = {}


def precompute_orders(imps: Sequence[torch.Tensor]) -> None:
    import weakref
    _ORDER_CACHE.clear()                                   # one prune's worth: the previous fit()'s tensors are not coming back
    work = [t for t in imps if isinstance(t, torch.Tensor) and t.device.type == "cpu" and t.dim() == 1]
    if len(work) < 4:
        return
    orders = list(_core.mask_pool().map(lambda t: torch.argsort(t, descending=True), work))
    for t, o in zip(work, orders):
        _ORDER_CACHE[id(t)] = (weakref.ref(t), t._version, o)


def _order_of(imp: torch.Tensor) -> torch.Tensor:
    ent = _ORDER_CACHE.get(id(imp))
    if ent is not None and ent[0]() is imp and ent[1] == imp._version:
        return ent[2]
    return torch.argsort(imp, descending=True)


_LAPS: Optional[dict] = None                 # scripts/api_profile.py sets a dict: host seconds of the a7 / a8 sub-steps are added into it


def _lap(name: str, t0: float) -> float:
    if _LAPS is None:
        return 0.0
    import time
    t1 = time.perf_counter()
    if name:
        _LAPS[name] = _LAPS.get(name, 0.0) + (t1 - t0)
    return t1


@torch.no_grad()
def prune_vit_mlp_width(vit_model, sparsity: Optional[float] = None, strategy: str = "l1", min_remaining: int = 256,
                        n_to_prune_per_block: Optional[List[int]] = None, dataloader=None, device: str = "cuda",
                        batch_limit: Optional[int] = None, progress: bool = False, collect_masks: bool = False,
                        precomputed_importance: Optional[List[torch.Tensor]] = None, *, score_chain: str = "fp32",
                        process_group=None):
    """Reference :203-319.  Scores come from the HIP engine (strategy "act_l2"), a caller, or fc1 row-L1; the
    keep-set is `sort(argsort(imp, descending=True)[:n-n_prune])`; the module is sliced in place and returned."""
    pairs = _gather_mlp_pairs(vit_model)
    if n_to_prune_per_block is not None:
        if len(n_to_prune_per_block) != len(pairs):
            raise ValueError("n_to_prune_per_block length must match number of blocks")
    else:
        if sparsity is None:
            raise ValueError("Provide either sparsity or n_to_prune_per_block")
        if not (0.0 <= sparsity < 1.0):
            raise AssertionError("sparsity must be in [0,1)")

    scores: Optional[List[torch.Tensor]] = None
    if precomputed_importance is not None:
        if len(precomputed_importance) != len(pairs):
            raise ValueError("precomputed_importance length must match number of blocks")
        scores = precomputed_importance
    elif strategy == "act_l2" and dataloader is not None:
        scores = _compute_ffn_activation_importance(vit_model, dataloader, device=device, batch_limit=batch_limit,
                                                    progress=progress, score_chain=score_chain,
                                                    process_group=process_group)

    t_lap = _lap("", 0.0)
    all_idx: List[List[int]] = []
    all_masks: List[List[int]] = []
    used_scores: List[torch.Tensor] = []
    used_drop: List[int] = []
    todo = []                                              # (block, keep index tensor) in block order
    work = []                                              # (block, importance, width, drop) of the blocks that lose neurons
    for b, (fc1, fc2) in enumerate(pairs):
        w1 = fc1.weight
        width = w1.size(0)
        if scores is not None:
            # The mask step runs where the importance lives.  The reference moves it to the weights' device first (:264) — on its CPU
            # path that is the HOST, and torch.argsort's order among tied scores differs between the host and a device sort: importances
            # that arrive as CPU tensors (what Auto2SSPInterface hands over) are therefore sorted on the host, exactly as the CPU
            # reference sorts them, and only the kept-index list crosses to the device (round 4 sorted them on the device: ~8 small
            # launches and two synchronisations per block inside the prune bracket).
            imp = scores[b] if scores[b].device.type == "cpu" else scores[b].to(w1.device)
            if imp.numel() != width:
                raise RuntimeError("precomputed/act_l2 importance size mismatch with intermediate width")
        elif strategy == "l1":
            imp = w1.abs().sum(dim=1)
        elif strategy == "act_l2":
            raise RuntimeError("act_l2 importance requested but no dataloader/importance available")
        else:
            raise ValueError(f"Unknown strategy {strategy}")
        drop = int(n_to_prune_per_block[b]) if n_to_prune_per_block is not None else int(width * sparsity)
        if width - drop < min_remaining:
            drop = max(0, width - min_remaining)
        used_scores.append(imp.detach()); used_drop.append(drop)
        if drop > 0:
            work.append((b, imp, width, drop))

    def cut(item):
        """The reference's mask step for one block (:286-295): the same torch calls on the same 1-D tensor — ties fall as they fall there."""
        b, imp, width, drop = item
        keep, _ = torch.sort(_order_of(imp)[: width - drop])
        if not collect_masks:
            return b, keep, None, None
        mask = torch.ones(width, dtype=torch.int16, device=keep.device)
        mask[keep] = 0                                     # 1 = prune, 0 = keep
        return b, keep, mask.cpu().tolist(), torch.nonzero(mask == 1).view(-1).tolist()
    # host importances: the blocks are independent and torch releases the interpreter lock inside each call, so a few threads
    # cut them side by side (12 x ~0.26 ms in a row were 3 ms of the prune bracket of the reference-named API); order and results unchanged
    t_lap = _lap("width: checks", t_lap)
    if len(work) >= 4 and all(w_[1].device.type == "cpu" for w_ in work):
        done = list(_core.mask_pool().map(cut, work))
    else:
        done = [cut(w_) for w_ in work]
    t_lap = _lap("width: a7 mask step (pool)", t_lap)
    for b, keep, mask_list, idx_list in done:
        if collect_masks:
            all_masks.append(mask_list); all_idx.append(idx_list)
        todo.append((b, keep))
    # a8: the slicing itself (reference :297-311), one gather per tensor on the weights' device.  Host-made index lists go up in ONE copy.
    if todo:
        dev = pairs[todo[0][0]][0].weight.device
        host = [k for _, k in todo if k.device.type == "cpu"]
        if host and dev.type != "cpu":
            flat = torch.cat(host).to(dev)
            off, moved = 0, []
            for k in host:
                moved.append(flat[off:off + k.numel()]); off += k.numel()
            it = iter(moved)
            todo = [(b, next(it) if k.device.type == "cpu" else k) for b, k in todo]
    t_lap = _lap("width: keep lists to the device", t_lap)
    for b, keep in todo:
        fc1, fc2 = pairs[b]
        w1, b1, w2 = fc1.weight, fc1.bias, fc2.weight
        keep = keep.to(w1.device)
        fc1.weight = nn.Parameter(torch.index_select(w1, 0, keep))       # == w1[keep].clone()
        if b1 is not None:
            fc1.bias = nn.Parameter(torch.index_select(b1, 0, keep))
        fc1.out_features = int(keep.numel())
        fc1.in_features = w1.size(1)
        fc2.weight = nn.Parameter(torch.index_select(w2, 1, keep))       # == w2[:, keep].clone()
        fc2.in_features = int(keep.numel())
    t_lap = _lap("width: a8 gathers + parameters", t_lap)
    if collect_masks:
        site_now = _score_site_of(vit_model)
        # the reference's three keys (:313-318) + this build's cut-margin table for the very scores the masks were cut from
        return _WidthPruneResult({"model": vit_model, "ffn_pruned_indices": all_idx, "ffn_prune_masks": all_masks},
                                 lambda: mask_parity_report(used_scores, used_drop, min_remaining=0, site=site_now))
    return vit_model


class _WidthPruneResult(dict):
    """prune_vit_mlp_width(collect_masks=True): the reference's three keys (:313-318) as a plain dict, plus this build's fourth,
    `mask_parity` — the cut-margin table of the very scores the masks were cut from — computed when it is first asked for
    (`res["mask_parity"]`): it is host arithmetic nobody who wants the reference's result pays for inside the prune bracket."""

    def __init__(self, items, report):
        super().__init__(items)
        self._report = report

    def __missing__(self, key):
        if key != "mask_parity":
            raise KeyError(key)
        self[key] = self._report()
        return self[key]

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default


def _score_site_of(vit_model) -> str:
    """Hook site of the stage-1 scores for this module's anatomy (reference :130 HF post-GELU, :135 timm pre-GELU)."""
    return getattr(vit_model, "ssp2_score_site", None) or _weights.score_site_for("timm" if _blocks(vit_model)[1] == "timm" else "hf")


# ----------------------------------------------------------------------------- a4 top-1
@torch.no_grad()
def _top1_counts(model, dataloader, device="cuda", max_batches=None, progress=False, *, process_group=None,
                 engine=None, attn_skip: Optional[Sequence[int]] = None, sharded: bool = False) -> Tuple[int, int]:
    return _core.top1_counts(_engine_factory(model, device, engine), dataloader, max_batches=max_batches,
                             progress=progress, process_group=process_group, attn_skip=attn_skip, sharded=sharded)


@torch.no_grad()
def evaluate_top1(model, dataloader, device: str = "cuda", max_batches: int | None = None, progress: bool = False,
                  *, process_group=None, engine=None):
    """Reference :325-373 — returns correct / max(1, total) as a Python float."""
    model.eval()
    c, t = _top1_counts(model, dataloader, device, max_batches, progress, process_group=process_group, engine=engine)
    return c / max(1, t)


# ----------------------------------------------------------------------------- a5 stage-2 search
@torch.no_grad()
def depth_search_counts(model, dataloader, device="cuda", batch_limit: Optional[int] = 5, *, process_group=None,
                        engine=None, removed: Sequence[int] = (), candidates: Optional[Sequence[int]] = None,
                        defer: bool = False, chunk_images: Optional[int] = None, batch_candidates="auto",
                        sharded: bool = False):
    """(baseline_correct, [candidate_correct], total) — see core.depth_search_counts (prefix-cached, layer-major
    whenever the workspace budget allows; `defer=True` returns a callable that waits for the device)."""
    return _core.depth_search_counts(_engine_factory(model, device, engine), dataloader, len(_blocks(model)[0]),
                                     batch_limit=batch_limit, process_group=process_group, removed=removed,
                                     candidates=candidates, defer=defer, chunk_images=chunk_images,
                                     batch_candidates=batch_candidates, sharded=sharded)


@torch.no_grad()
def importances_one_pass(vit_model, dataloader, device="cuda", batch_limit: Optional[int] = 5, *, score_limit="same",
                         score_chain: str = "fp32", process_group=None, engine=None, defer: bool = False,
                         sharded: bool = False, chunk_images: Optional[int] = None, eval_chunk_images: Optional[int] = None,
                         batch_candidates="auto"):
    """Both importances of `Auto2SSPInterface.fit()` from ONE walk over the loader (core.prune_pass): the dense forward of the
    search's baseline is the stage-1 pass as well (reference: one loader, one batch_limit, two separate dense passes —
    adaptation-for-Pures-framework/mask_conjunction.py:276-281, :327, :359-362).  `score_limit`: "same" = batch_limit (the plug-in's
    rule), an int or None for a different stage-1 limit (batches beyond the search's get a scores-only forward).
    Returns (scores, (baseline_correct, [candidate_correct], total)) — or, with defer=True, two callables that wait."""
    vit_model.eval()
    blocks, kind = _blocks(vit_model)
    d_ints = [p[0].out_features for p in _gather_mlp_pairs(vit_model)]
    site = getattr(vit_model, "ssp2_score_site", None) or _weights.score_site_for("timm" if kind == "timm" else "hf")
    return _core.prune_pass(_engine_factory(vit_model, device, engine), dataloader, d_ints, site, len(blocks),
                            score_limit=batch_limit if score_limit == "same" else score_limit, search_limit=batch_limit,
                            score_chain=score_chain, process_group=process_group, chunk_images=chunk_images,
                            eval_chunk_images=eval_chunk_images, defer=defer, sharded=sharded, batch_candidates=batch_candidates)


class HFAttentionBypass(nn.Module):
    """Zero-output attention, HF flavour: returns a tuple (reference :416-423)."""
    def forward(self, hidden_states, head_mask=None, output_attentions: bool = False, *args, **kwargs):
        z = torch.zeros_like(hidden_states)
        return (z, None) if output_attentions else (z,)


class TimmAttentionBypass(nn.Module):
    """Zero-output attention, timm flavour (reference :425-429)."""
    def forward(self, x, *args, **kwargs):
        return torch.zeros_like(x)


class HF5AttentionBypass(nn.Module):
    """Zero-output attention for the transformers >= 5 layer, whose attention always returns the pair
    (attn_output, attn_weights) and is called as `attention(hidden_states, attention_mask, **kwargs)`."""
    def forward(self, hidden_states, attention_mask=None, *args, **kwargs):
        return torch.zeros_like(hidden_states), None


def _apply_bypass(vit_model, idx: int) -> None:
    blocks, kind = _blocks(vit_model)
    if kind == "hf" and hasattr(blocks[idx], "attention"):
        blocks[idx].attention = HFAttentionBypass()
    elif kind == "hf5" and hasattr(blocks[idx], "attention"):
        blocks[idx].attention = HF5AttentionBypass()
    elif kind == "timm" and hasattr(blocks[idx], "attn"):
        blocks[idx].attn = TimmAttentionBypass()


@torch.no_grad()
def prune_vit_attention_blocks(vit_model, sparsity: float, dataloader=None, device: str = "cuda", batch_limit: int = 5,
                               metric_fn=None, importance_mode: str = "copy", show_progress: bool = True,
                               num_to_prune: Optional[int] = None, selected_indices: Optional[List[int]] = None,
                               *, process_group=None, search: str = "one_shot") -> Dict[str, Any]:
    """Reference :379-520.  `search="iterative"` adds the greedy K-round variant of the LLM code
    (src/utilities.py:446-505) with top-1 as the metric."""
    assert 0.0 <= sparsity < 1.0, "sparsity must be in [0,1)"
    vit_model.eval()
    try:
        num_blocks = len(_blocks(vit_model)[0])
    except AttributeError:
        num_blocks = 0
    if num_to_prune is None:
        num_to_prune = int(round(num_blocks * sparsity))
    num_to_prune = max(0, min(num_blocks - 1, int(num_to_prune)))
    if num_to_prune == 0:
        return {"model": vit_model, "pruned_indices": [], "original_metrics": None, "final_metrics": None}

    original = final = None
    if selected_indices is not None:
        to_prune = sorted(set(i for i in selected_indices if 0 <= i < num_blocks))[:num_to_prune]
    elif dataloader is None or (isinstance(importance_mode, str) and importance_mode.lower() == "heuristic"):
        score = [(i if i < num_blocks / 2 else num_blocks - i) for i in range(num_blocks)]
        to_prune = sorted(range(num_blocks), key=lambda i: score[i])[:num_to_prune]
    elif search == "iterative":
        base, _, tot = depth_search_counts(vit_model, dataloader, device, batch_limit, process_group=process_group,
                                           candidates=[])
        original = base / max(1, tot)
        to_prune = []
        for _ in range(num_to_prune):
            rest = [i for i in range(num_blocks) if i not in to_prune]
            _, cc, tot = depth_search_counts(vit_model, dataloader, device, batch_limit, process_group=process_group,
                                             removed=to_prune, candidates=rest)
            best = max(rest, key=lambda i: (cc[i], -i))       # highest remaining top-1, ties -> lower index
            to_prune.append(best)
    else:
        base, cc, tot = depth_search_counts(vit_model, dataloader, device, batch_limit, process_group=process_group)
        original = base / max(1, tot)
        impact = [max(0.0, original - (c / max(1, tot))) for c in cc]
        if show_progress:
            for i, v in enumerate(impact):
                print(f"[Attn] Block {i} impact: {v:.4f}", flush=True)
        to_prune = sorted(range(num_blocks), key=lambda i: impact[i])[:num_to_prune]   # stable: ties -> lower index

    for idx in to_prune:
        _apply_bypass(vit_model, idx)
    if dataloader is not None:
        final = evaluate_top1(vit_model, dataloader, device, max_batches=batch_limit, process_group=process_group)
    return {"model": vit_model, "pruned_indices": sorted(list(to_prune)), "original_metrics": original,
            "final_metrics": final}


# ----------------------------------------------------------------------------- report (reference :877-946)
def _jsonable(o):
    try:
        json.dumps(o)
        return o
    except Exception:
        if isinstance(o, (list, tuple)):
            return [_jsonable(v) for v in o]
        if isinstance(o, dict):
            return {str(k): _jsonable(v) for k, v in o.items()}
        return str(o)


# ----------------------------------------------------------------------------- classifier head / adapter files (reference :774-875)
_ADAPTER_KEYS = ("state_dict", "classifier_type", "num_labels", "hidden_size", "timestamp", "extra")


@torch.no_grad()
def save_cifar_adapter(model: nn.Module, out_dir: str, filename: str = "adapter.pt", extra: Optional[Dict[str, Any]] = None) -> str:
    """Reference :775-798 — the classifier (a Linear head, or the Linear -> GELU -> Linear adapter) of `model` as one file:
    {"state_dict", "classifier_type", "num_labels", "hidden_size", "timestamp", "extra"}.  Host-only; returns the path."""
    os.makedirs(out_dir, exist_ok=True)
    head = model.classifier
    cfg = getattr(model, "config", None)
    values = (head.state_dict(), type(head).__name__, getattr(cfg, "num_labels", None), getattr(cfg, "hidden_size", None),
              time.strftime("%Y-%m-%d %H:%M:%S"), dict(extra) if extra else {})
    path = os.path.join(out_dir, filename)
    torch.save(dict(zip(_ADAPTER_KEYS, values)), path)
    return path


@torch.no_grad()
def load_cifar_adapter(path: str, model: nn.Module) -> nn.Module:
    """Reference :800-875 — rebuilds the saved head on `model` (in place, returned): a file whose state dict holds "weight" is a Linear
    classifier, one with "0.weight" / "2.weight" the bottleneck adapter Sequential(Linear(hidden, r, bias=False), GELU, Linear(r, labels));
    shapes missing from the metadata are read off the tensors; `model.config.num_labels` follows.  RuntimeError where the reference
    raises one (no hidden size, no label count, an adapter whose layers cannot be told).  The file is read with a loader that executes
    nothing from it (weights_only)."""
    blob = torch.load(path, map_location="cpu", weights_only=True)
    sd = blob.get("state_dict", {})
    kind = blob.get("classifier_type", "Linear")
    labels, hidden_saved = blob.get("num_labels"), blob.get("hidden_size")
    seen_hidden = seen_rank = seen_out = None
    if "weight" in sd:
        seen_out, seen_hidden = (int(v) for v in sd["weight"].shape)
    elif "0.weight" in sd and "2.weight" in sd:
        seen_rank, seen_hidden = (int(v) for v in sd["0.weight"].shape)
        seen_out = int(sd["2.weight"].shape[0])
    hidden = hidden_saved or getattr(getattr(model, "config", None), "hidden_size", None) or seen_hidden
    if hidden is None:
        raise RuntimeError("Cannot determine hidden size for adapter loading.")
    if labels is None:
        labels = seen_out
    if kind == "Linear" or ("weight" in sd and "bias" in sd):
        if labels is None:
            raise RuntimeError("num_labels is None for Linear classifier.")
        head: nn.Module = nn.Linear(int(hidden), int(labels))
    else:
        if seen_rank is None and "0.weight" in sd:
            seen_rank = int(sd["0.weight"].shape[0])
        if labels is None and "2.weight" in sd:
            labels = int(sd["2.weight"].shape[0])
        if seen_rank is None or labels is None:
            raise RuntimeError("Cannot reconstruct adapter architecture from payload/state_dict.")
        head = nn.Sequential(nn.Linear(int(hidden), seen_rank, bias=False), nn.GELU(), nn.Linear(seen_rank, int(labels), bias=True))
    head.load_state_dict(sd)
    model.classifier = head
    model.config.num_labels = int(labels)
    return model


def save_report(report: Dict[str, Any], out_dir: str, run_id: Optional[str] = None) -> Dict[str, str]:
    """JSON + Markdown report with the reference's file names and section layout."""
    os.makedirs(out_dir, exist_ok=True)
    run_id = run_id or time.strftime("%Y%m%d-%H%M%S")
    jp = os.path.join(out_dir, f"report-{run_id}.json")
    mp = os.path.join(out_dir, f"report-{run_id}.md")
    with open(jp, "w", encoding="utf-8") as f:
        json.dump(_jsonable(report), f, indent=2, ensure_ascii=False)
    md = [f"# 2SSP ViT Pruning Report ({run_id})", ""]
    if "config" in report:
        md += ["## Config"] + [f"- {k}: {v}" for k, v in report["config"].items()] + [""]
    m = report.get("metrics")
    if m is not None:
        g = m.get
        md += ["## Parameters reduction",
               f"- Stage-1 (Width): {g('params_before_stage1_millions')}M -> {g('params_after_stage1_millions')}M ({g('stage1_reduction_percent')}%)",
               f"- Stage-2 (Depth): {g('params_after_stage1_millions')}M -> {g('params_after_stage2_millions')}M ({g('stage2_reduction_percent')}%)",
               f"- Final result: {g('params_before_stage1_millions')}M -> {g('params_after_stage2_millions')}M ({g('total_reduction_percent')}%)",
               "", "## Latency", f"- Baseline: {g('latency_baseline_ms')} ms",
               f"- Stage-1 (Width): {g('latency_stage1_ms')} ms ({g('latency_stage1_change_percent')}%)",
               f"- Stage-2 (Depth): {g('latency_stage2_ms')} ms ({g('latency_stage2_change_percent')}%)",
               f"- Final change: {g('latency_total_change_percent')}%", "", "## Accuracy",
               f"- Baseline: {g('acc_baseline')}",
               f"- Stage-1 (Width): {g('acc_stage1')} (drop: {g('acc_drop_stage1_percent')}%)",
               f"- Stage-2 (Depth): {g('acc_stage2')} (drop: {g('acc_drop_stage2_percent')}%)",
               f"- Final change: {g('acc_total_drop_percent')}%", ""]
    p = report.get("plan")
    if p is not None:
        g = p.get
        md += ["## Auto-allocation plan", f"- Target sparsity: {g('target_sparsity')}", f"- Blocks total: {g('num_blocks_total')}",
               f"- Blocks to prune (Stage-2): {g('blocks_to_prune')} ({g('stage2_fraction'):.4f})",
               f"- Per-block neurons to prune (Stage-1): {g('per_block_neurons_to_prune')}",
               f"- Estimated total removed params: {g('estimated_total_removed_params')}",
               f"- Estimation error (params): {g('est_error_params')}", ""]
    mp_ = report.get("mask_parity")
    if mp_:
        md += ["## Mask parity (cut margins of the stage-1 masks)",
               f"- eps (twice the score error bound): {mp_.get('eps')}",
               f"- blocks whose mask is guaranteed equal to a CPU run's: {mp_.get('blocks_guaranteed')} of {mp_.get('blocks_total')}",
               f"- smallest relative cut margin: {mp_.get('min_margin')}"]
        md += [f"- block {b['block']}: margin {b['cut_margin']}, tie band {b['tie_band']}, exact ties {b['exact_ties']}"
               for b in mp_.get("blocks", []) if not b.get("guaranteed", True)] + [""]
    if "artifacts" in report:
        md += ["## Artifacts"] + [f"- {k}: {v}" for k, v in report["artifacts"].items()] + [""]
    with open(mp, "w", encoding="utf-8") as f:
        f.write("\n".join(md))
    return {"json": jp, "md": mp}
