"""TEST INFRASTRUCTURE — CPU oracle for the 2SSP-for-ViT hot path.

A restatement, in plain PyTorch CPU ops, of the reference algorithm the HIP engine replaces.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file;
the product package (``2ssp-x-vit_amd/ssp2vit``) never does and fails loudly without its HIP library.

Pinning: every function here is checked bit-for-bit against outputs of the real reference
(imported from /root/reference in the build container only) by ``tests/test_oracle_golden.py`` using
the fixtures that ``tests/golden/make_golden.py`` wrote.  Each function cites the reference lines it
follows (paths relative to /root/reference).
"""
from __future__ import annotations

import copy
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn


# ----------------------------------------------------------------------------- anatomy
def _blocks_of(model) -> Sequence[nn.Module]:
    """src/vit_pruning.py:27-45 — HF `.vit.encoder.layer` or timm `.blocks`."""
    base = getattr(model, "vit", None) or getattr(model, "base_model", None) or model
    enc = getattr(base, "encoder", base)
    if hasattr(enc, "layer"):
        return enc.layer
    if hasattr(enc, "blocks"):
        return enc.blocks
    raise AttributeError("Unsupported ViT model structure: expected encoder.layer or blocks")


def _is_hf(model) -> bool:
    base = getattr(model, "vit", None) or getattr(model, "base_model", None) or model
    return hasattr(getattr(base, "encoder", base), "layer")


def _score_tap(block, hf: bool) -> nn.Module:
    """Module whose OUTPUT the stage-1 score is taken from (src/vit_pruning.py:130 / :135)."""
    return block.intermediate if hf else block.mlp.fc1


def _d_int(block, hf: bool) -> int:
    return (block.intermediate.dense if hf else block.mlp.fc1).out_features


def _call(model, px):
    """src/vit_pruning.py:179-185, :353-369 — kwarg call, positional fallback, normalise to logits."""
    try:
        out = model(pixel_values=px)
    except TypeError:
        out = model(px)
    if isinstance(out, torch.Tensor):
        return out
    if hasattr(out, "logits"):
        return out.logits
    if isinstance(out, (tuple, list)) and out and isinstance(out[0], torch.Tensor):
        return out[0]
    raise RuntimeError("Model forward output is not a tensor or does not contain logits")


# ----------------------------------------------------------------------------- a1/a2  stage-1 scores
@torch.no_grad()
def ffn_activation_importance(model, loader: Iterable[Dict], batch_limit: Optional[int] = None,
                              chain: str = "autocast") -> List[torch.Tensor]:
    """src/vit_pruning.py:111-201.

    per block: sum over batches of [ sum over samples of ||act[s,:,j]||_2 over tokens ] / n_samples.
    ``chain="autocast"`` keeps every intermediate in the autocast dtype exactly like the reference on CPU
    (bf16: norm -> bf16, batch-sum -> bf16, cross-batch += in bf16, /count in bf16; lines 151-157, 200).
    ``chain="fp32"`` takes the same bf16 activations but carries the score arithmetic in fp32 (the engine's
    default ``score_chain``; not a reference mode — used to check the engine's fp32 accumulators).
    """
    model.eval()
    blocks = _blocks_of(model)
    hf = _is_hf(model)
    running: List[Optional[torch.Tensor]] = [None] * len(blocks)
    seen = 0

    def tap(idx):
        def fn(_m, _i, out):
            act = out[0] if isinstance(out, (tuple, list)) else out
            if chain == "fp32":
                act = act.float()
            contrib = torch.linalg.vector_norm(act, ord=2, dim=1).sum(dim=0).detach().to("cpu")
            if running[idx] is None:
                running[idx] = contrib
            else:
                running[idx] += contrib
        return fn

    hooks = [_score_tap(b, hf).register_forward_hook(tap(i)) for i, b in enumerate(blocks)]
    try:
        for bi, batch in enumerate(loader):
            if batch_limit is not None and bi >= batch_limit:
                break
            px = batch["pixel_values"]
            with torch.autocast(device_type="cpu", enabled=True):
                _call(model, px)
            seen += px.size(0)
    finally:
        for h in hooks:
            h.remove()
    denom = max(1, seen)
    return [torch.zeros(_d_int(b, hf)) if running[i] is None else running[i] / denom
            for i, b in enumerate(blocks)]


# ----------------------------------------------------------------------------- a4  top-1
@torch.no_grad()
def top1_counts(model, loader: Iterable[Dict], max_batches: Optional[int] = None):
    """src/vit_pruning.py:325-373 — returns (correct, total); accuracy = correct / max(1,total)."""
    model.eval()
    correct = total = 0
    for bi, batch in enumerate(loader):
        if max_batches is not None and bi >= max_batches:
            break
        with torch.autocast(device_type="cpu", enabled=True):
            logits = _call(model, batch["pixel_values"])
        correct += int((logits.argmax(dim=-1) == batch["labels"]).sum().item())
        total += int(batch["labels"].size(0))
    return correct, total


def evaluate_top1(model, loader, max_batches: Optional[int] = None) -> float:
    c, t = top1_counts(model, loader, max_batches)
    return c / max(1, t)


@torch.no_grad()
def logits_of(model, px: torch.Tensor) -> torch.Tensor:
    model.eval()
    with torch.autocast(device_type="cpu", enabled=True):
        return _call(model, px)


# ----------------------------------------------------------------------------- a6  bypass
class _ZeroAttnTuple(nn.Module):
    """HF flavour: attention returns a tuple (src/vit_pruning.py:416-423)."""
    def forward(self, hidden_states, head_mask=None, output_attentions=False, *a, **k):
        z = torch.zeros_like(hidden_states)
        return (z, None) if output_attentions else (z,)


class _ZeroAttnTensor(nn.Module):
    """timm flavour (src/vit_pruning.py:425-429)."""
    def forward(self, x, *a, **k):
        return torch.zeros_like(x)


def bypass_attention_(model, idx: int) -> None:
    blk = _blocks_of(model)[idx]
    if _is_hf(model):
        blk.attention = _ZeroAttnTuple()
    else:
        blk.attn = _ZeroAttnTensor()


# ----------------------------------------------------------------------------- a5  one-shot depth search
@torch.no_grad()
def att_depth_importance(model, loader, batch_limit: Optional[int] = 5) -> torch.Tensor:
    """adaptation-for-Pures-framework/mask_conjunction.py:298-357 (copy mode).

    impact_i = max(0, baseline - top1(model with attention i bypassed)); float32 tensor [L].
    """
    base = float(evaluate_top1(model, loader, batch_limit))
    out = []
    for i in range(len(_blocks_of(model))):
        trial = copy.deepcopy(model)
        bypass_attention_(trial, i)
        out.append(max(0.0, base - float(evaluate_top1(trial, loader, batch_limit))))
    return torch.tensor(out, dtype=torch.float32)


def heuristic_depth_scores(n_blocks: int) -> torch.Tensor:
    """mask_conjunction.py:301-304 — position heuristic."""
    return torch.tensor([(i if i < n_blocks / 2 else n_blocks - i) for i in range(n_blocks)],
                        dtype=torch.float32)


def select_blocks_python_sort(impact: Sequence[float], k: int) -> List[int]:
    """src/vit_pruning.py:496 — Python stable sort, ties -> lower index."""
    return sorted(range(len(impact)), key=lambda i: impact[i])[:k]


def select_blocks_torch_argsort(att_imp: torch.Tensor, k: int) -> List[int]:
    """adaptation-for-Pures-framework/auto_2ssp.py:857 then src/vit_pruning.py:452-454."""
    chosen = [int(i) for i in torch.argsort(att_imp)[:k]]
    return sorted(set(chosen))[:k]


@torch.no_grad()
def greedy_depth_search(model, loader, k: int, batch_limit: Optional[int] = 5):
    """src/utilities.py:446-505 semantics (LLM code) with top-1 as the metric: K rounds, each round
    tries every not-yet-removed block with the earlier removals in place and commits the one that keeps
    top-1 highest (ties -> lower index)."""
    work = copy.deepcopy(model)
    removed: List[int] = []
    trace = []
    for _ in range(k):
        best_i, best_acc = None, -1.0
        for i in range(len(_blocks_of(work))):
            if i in removed:
                continue
            trial = copy.deepcopy(work)
            bypass_attention_(trial, i)
            acc = float(evaluate_top1(trial, loader, batch_limit))
            if acc > best_acc:
                best_i, best_acc = i, acc
        bypass_attention_(work, best_i)
        removed.append(best_i)
        trace.append((best_i, best_acc))
    return removed, trace


# ----------------------------------------------------------------------------- a7  mask step
def width_prune_selection(importance: Sequence[torch.Tensor], n_prune_per_block: Sequence[int],
                          min_remaining: int = 256):
    """src/vit_pruning.py:273-295 — per block: keep = sort(argsort(imp, descending)[:n-n_prune]);
    mask 1 = prune.  Blocks with n_prune <= 0 contribute NO entry (the reference `continue`s, line 283)."""
    masks, pruned = [], []
    for imp, want in zip(importance, n_prune_per_block):
        n = imp.numel()
        t = int(want)
        if n - t < min_remaining:
            t = max(0, n - min_remaining)
        if t <= 0:
            continue
        keep, _ = torch.sort(torch.argsort(imp, descending=True)[: n - t])
        m = torch.ones(n, dtype=torch.int16)
        m[keep] = 0
        masks.append(m.tolist())
        pruned.append(torch.nonzero(m == 1).view(-1).tolist())
    return masks, pruned


# ----------------------------------------------------------------------------- standalone kernel oracle
def act_l2_accum_f64(act: np.ndarray) -> np.ndarray:
    """Exact (float64) statement of the hook arithmetic on one activation tensor [n, N, d_int]:
    out[j] = sum_s sqrt(sum_t act[s,t,j]^2)   (src/vit_pruning.py:151-152)."""
    a = act.astype(np.float64)
    return np.sqrt((a * a).sum(axis=1)).sum(axis=0)


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)
