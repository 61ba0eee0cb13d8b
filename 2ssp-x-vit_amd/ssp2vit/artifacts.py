"""Score / mask artifact I/O in the reference's file formats — host-side JSON only.

Schemas (all restated from the reference, 1 = prune, 0 = keep):
  * `ffn_prune_masks.json`          {"ffn_masks": [[0/1]*d_int]*L}                        auto_2ssp.py:921-931
  * `<prefix>_scores.json`          {"ffn": {"l:i": f}, "heads": {"l:h": f}, "qkv_dim": {"l:d": f}}   auto_2ssp.py:71-82
  * `<prefix>_masks.json`           {"ffn": {"l": [..]}, "heads": {"l": [..]}, "qkv_dim": {"l": [..]}} auto_2ssp.py:84-89
    head / qkv_dim scores are the block's depth importance broadcast, their masks all-ones for removed blocks
    (auto_2ssp.py:139-175)
  * the OLDER CLI's three files (`format_version` 1 masks + indices, attention indices, "b:j" importances):  save_v1_artifacts,
    experiments/vit_pruning/auto_2ssp.py:769-829
  * any JSON tree whose leaves are {"i:j": number} is a valid score or mask file for the consumers
    (experiments/vit_pruning/apply_mask_prune.py:206-256, manual-experiments/*.py)

Combiners follow manual-experiments/normalize_scores.py:44-85 (global raw min-max), aggregate_and_mask-summation.py
:208-269 (per-block bottom-K of the summed scores with one common K) and consensus_mask.py:175-298 (intersection of
per-file bottom-k sets, k grown until every block reaches the common K).
"""
from __future__ import annotations

import json
import math
import os
import re
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple

import torch

IJ = re.compile(r"^(\d+):(\d+)$")
PathT = Tuple[str, ...]


# ----------------------------------------------------------------------------- writers
def save_ffn_prune_masks(path: str, masks: Sequence[Sequence[int]]) -> str:
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w", encoding="utf-8") as f:
        json.dump({"ffn_masks": [[int(v) for v in (m.tolist() if isinstance(m, torch.Tensor) else m)] for m in masks]},
                  f, indent=2)
    return path


def save_v1_artifacts(art_dir: str, *, mlp_imp: Optional[Sequence] = None, ffn_masks: Optional[Sequence[Sequence[int]]] = None,
                      ffn_indices: Optional[Sequence[Sequence[int]]] = None, pruned_block_indices: Optional[Sequence[int]] = None,
                      min_remaining: int = 256, s1_sparsity: Optional[float] = None, block_inter_sizes: Optional[Sequence[int]] = None,
                      strategy: str = "act_l2") -> Dict[str, str]:
    """The OLDER CLI's artifact files (experiments/vit_pruning/auto_2ssp.py:769-829), by their names:
      iterative_vit_b16_ffn_importances.json   {"ffn": {"b:j": float}}                                          (:770-785; indent 2, non-ASCII kept)
      ffn_prune_masks.json                      {"format_version": 1, "stage": "s1", "strategy", "min_remaining", "s1_sparsity",
                                                 "block_inter_sizes", "masks": [[0/1]], "indices": [[pruned j]]}  (:788-806; 1 = prune)
      attention_pruned_indices.json             {"format_version": 1, "stage": "s2", "indices": [blocks]}        (:808-816; only when non-empty)
    Returns {artifact key of the reference's report: path} for the files written (:818-828)."""
    os.makedirs(art_dir, exist_ok=True)
    out: Dict[str, str] = {}
    if mlp_imp is not None:
        ffn_map = {}
        for b, imp in enumerate(mlp_imp):
            vals = imp.detach().cpu().flatten().tolist() if isinstance(imp, torch.Tensor) else [float(x) for x in imp]
            for j, v in enumerate(vals):
                ffn_map[f"{b}:{j}"] = float(v)
        path = os.path.join(art_dir, "iterative_vit_b16_ffn_importances.json")
        with open(path, "w", encoding="utf-8") as f:
            json.dump({"ffn": ffn_map}, f, ensure_ascii=False, indent=2)
        out["ffn_importances_path"] = path
    if ffn_masks is not None:
        masks = [[int(v) for v in (m.tolist() if isinstance(m, torch.Tensor) else m)] for m in ffn_masks]
        if ffn_indices is None:
            ffn_indices = [[j for j, bit in enumerate(m) if bit == 1] for m in masks]
        path = os.path.join(art_dir, "ffn_prune_masks.json")
        with open(path, "w", encoding="utf-8") as f:
            json.dump({"format_version": 1, "stage": "s1", "strategy": strategy, "min_remaining": min_remaining,
                       "s1_sparsity": s1_sparsity, "block_inter_sizes": None if block_inter_sizes is None else [int(v) for v in block_inter_sizes],
                       "masks": masks, "indices": [[int(j) for j in ix] for ix in ffn_indices]}, f, indent=2)
        out["ffn_prune_masks_path"] = path
    if pruned_block_indices:
        path = os.path.join(art_dir, "attention_pruned_indices.json")
        with open(path, "w", encoding="utf-8") as f:
            json.dump({"format_version": 1, "stage": "s2", "indices": [int(i) for i in pruned_block_indices]}, f, indent=2)
        out["attn_pruned_indices_path"] = path
    return out


def build_framework_exports(prefix: str, n_blocks: int, hidden: int, num_heads: int,
                            mlp_imp_list: Optional[Sequence], att_imp, ffn_masks_list: Optional[Sequence[Sequence[int]]],
                            pruned_attn_block_indices: Optional[Iterable[int]], write: bool = True) -> Dict[str, Dict]:
    """Framework JSON pair.  Takes the model facts (blocks, hidden, heads) instead of the module itself."""
    def tolist(v):
        return v.detach().cpu().tolist() if isinstance(v, torch.Tensor) else list(v)

    ffn_imp = {f"{l}:{i}": float(s) for l, vec in enumerate(mlp_imp_list or []) for i, s in enumerate(tolist(vec))}
    att = tolist(att_imp) if att_imp is not None else []
    att = (att + [0.0] * n_blocks)[:n_blocks]
    heads = {f"{l}:{h}": float(att[l]) for l in range(n_blocks) for h in range(num_heads)}
    qkv = {f"{l}:{d}": float(att[l]) for l in range(n_blocks) for d in range(hidden)}
    if ffn_masks_list is not None and len(ffn_masks_list) == n_blocks:
        ffn_mask = {str(l): [int(v) for v in m] for l, m in enumerate(ffn_masks_list)}
    else:   # reference fallback: all-keep masks sized from the score vectors
        ffn_mask = {str(l): [0] * (len(mlp_imp_list[l]) if mlp_imp_list and l < len(mlp_imp_list) else hidden * 4)
                    for l in range(n_blocks)}
    gone = set(int(i) for i in (pruned_attn_block_indices or []))
    head_mask = {str(l): [1 if l in gone else 0] * num_heads for l in range(n_blocks)}
    qkv_mask = {str(l): [1 if l in gone else 0] * hidden for l in range(n_blocks)}
    scores = {"ffn": ffn_imp, "heads": heads, "qkv_dim": qkv}
    masks = {"ffn": ffn_mask, "heads": head_mask, "qkv_dim": qkv_mask}
    if write:
        os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
        with open(prefix + "_scores.json", "w") as f:
            json.dump(scores, f, indent=2)
        with open(prefix + "_masks.json", "w") as f:
            json.dump(masks, f, indent=2)
    return {"scores": scores, "masks": masks}


def scores_to_ij(imps: Sequence[torch.Tensor]) -> Dict[str, Dict[str, float]]:
    """{"ffn": {"b:j": score}} — the layout of manual-experiments/2ssp_vit_b16_ffn_importances.json."""
    return {"ffn": {f"{b}:{j}": float(v) for b, t in enumerate(imps) for j, v in enumerate(t.tolist())}}


# ----------------------------------------------------------------------------- readers
def _is_ij_leaf(d: Any) -> bool:
    return isinstance(d, dict) and bool(d) and all(
        isinstance(k, str) and IJ.match(k) and isinstance(v, (int, float)) for k, v in d.items())


def find_ij_leaves(obj: Any, path: Optional[List[str]] = None, out: Optional[List] = None) -> List[Tuple[PathT, Dict[str, float]]]:
    path = [] if path is None else path
    out = [] if out is None else out
    if isinstance(obj, dict):
        if _is_ij_leaf(obj):
            out.append((tuple(path), {k: float(v) for k, v in obj.items()}))
            return out
        for k, v in obj.items():
            find_ij_leaves(v, path + [str(k)], out)
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            find_ij_leaves(v, path + [f"[{i}]"], out)
    return out


def load_mask(path: str) -> Dict[int, Dict[int, int]]:
    """block -> {neuron -> 0/1}; several ij-leaves are merged (apply_mask_prune.py:235-256)."""
    with open(path, "r", encoding="utf-8") as f:
        leaves = find_ij_leaves(json.load(f))
    if not leaves:
        raise RuntimeError(f"Mask file has no ij-leaf dicts: {path}")
    blocks: Dict[int, Dict[int, int]] = {}
    for _p, leaf in leaves:
        for k, v in leaf.items():
            m = IJ.match(k)
            blocks.setdefault(int(m.group(1)), {})[int(m.group(2))] = 1 if int(round(float(v))) != 0 else 0
    return blocks


def mask_to_importance_and_counts(blocks_mask: Dict[int, Dict[int, int]], inter_sizes: Sequence[int]):
    """+1 keep / -1 prune pseudo-importance and the prune count per block, ready for
    `prune_vit_mlp_width(precomputed_importance=..., n_to_prune_per_block=...)` (apply_mask_prune.py:259-280)."""
    imp, counts = [], []
    for i, d in enumerate(inter_sizes):
        vec = torch.ones(d, dtype=torch.float32)
        bm = blocks_mask.get(i, {})
        idx = [j for j in range(d) if bm.get(j, 0) == 1]
        if idx:
            vec[idx] = -1.0
        imp.append(vec)
        counts.append(len(idx))
    return imp, counts


# ----------------------------------------------------------------------------- combiners
def _is_num(x: Any) -> bool:
    return isinstance(x, (int, float)) and not isinstance(x, bool)


def minmax_normalize(obj: Any) -> Any:
    """Global raw min-max over every number in the tree -> [0,1]; constant input -> 0.0 (normalize_scores.py)."""
    lo, hi = math.inf, -math.inf
    stack = [obj]
    while stack:
        cur = stack.pop()
        if _is_num(cur):
            lo, hi = min(lo, float(cur)), max(hi, float(cur))
        elif isinstance(cur, list):
            stack.extend(cur)
        elif isinstance(cur, dict):
            stack.extend(cur.values())
    if lo is math.inf:
        return obj

    def walk(o):
        if _is_num(o):
            return 0.0 if hi == lo else (float(o) - lo) / (hi - lo)
        if isinstance(o, list):
            return [walk(x) for x in o]
        if isinstance(o, dict):
            return {k: walk(v) for k, v in o.items()}
        return o
    return walk(obj)


def _rounder(name: str):
    return {"round": lambda x: int(round(x)), "floor": lambda x: int(math.floor(x)), "ceil": lambda x: int(math.ceil(x))}[name]


def _key_order(k: str):
    m = IJ.match(k)
    return (int(m.group(1)), int(m.group(2))) if m else (1 << 30, 1 << 30)


def sum_leaves(leaves: Sequence[Dict[str, float]]) -> Dict[str, float]:
    out: Dict[str, float] = {}
    for leaf in leaves:
        for k, v in leaf.items():
            out[k] = out.get(k, 0.0) + float(v)
    return out


def bottom_k_mask(leaf: Dict[str, float], prune_fraction: float, rounding: str = "round",
                  per_block_k: Optional[int] = None) -> Dict[str, int]:
    """Per block the K smallest scores -> 1, one common K = min_i round(fraction * N_i) (or `per_block_k`);
    Python's stable sort breaks ties by insertion order; keys come back ordered by (block, neuron)."""
    groups: Dict[int, List[Tuple[str, float]]] = {}
    for k, v in leaf.items():
        m = IJ.match(k)
        if m:
            groups.setdefault(int(m.group(1)), []).append((k, float(v)))
    ordered = sorted(leaf.keys(), key=_key_order)
    if not groups:
        return {k: 0 for k in ordered}
    if per_block_k is None:
        rf = _rounder(rounding)
        common = min(max(0, min(len(it), rf(prune_fraction * len(it)))) for it in groups.values())
    else:
        common = max(0, per_block_k)
    pruned = set()
    for items in groups.values():
        pruned |= {k for k, _ in sorted(items, key=lambda kv: kv[1])[: min(common, len(items))]}
    return {k: (1 if k in pruned else 0) for k in ordered}


def consensus_mask(leaves: Sequence[Dict[str, float]], prune_fraction: float, rounding: str = "round") -> Dict[str, int]:
    """Intersection of the per-file bottom-k sets per block; the internal fraction t grows x1.2 from `prune_fraction`
    until every block's intersection reaches the common K (or t = 1, at most 100 steps); surplus keys are cut to K by
    smallest mean (ties by key order)."""
    rf = _rounder(rounding)
    per_file: List[Dict[int, Dict[str, float]]] = []
    for leaf in leaves:
        b: Dict[int, Dict[str, float]] = {}
        for k, v in leaf.items():
            m = IJ.match(k)
            if m:
                b.setdefault(int(m.group(1)), {})[k] = float(v)
        per_file.append(b)
    blocks = sorted(set().union(*[set(b.keys()) for b in per_file])) if per_file else []
    common_keys = {i: sorted(set.intersection(*[set(fb.get(i, {}).keys()) for fb in per_file]), key=_key_order) for i in blocks}
    if not blocks:
        return {}
    k_common = min(max(0, min(len(common_keys[i]), rf(prune_fraction * len(common_keys[i])))) for i in blocks)
    if k_common <= 0:
        return {k: 0 for i in blocks for k in common_keys[i]}

    def inter_at(t: float) -> Dict[int, List[str]]:
        out = {}
        for i in blocks:
            keys = common_keys[i]
            k = max(0, min(len(keys), rf(t * len(keys)))) if keys else 0
            if k == 0:
                out[i] = []
                continue
            sets = [set(sorted(keys, key=lambda kk: (fb.get(i, {}).get(kk, float("inf")), _key_order(kk)))[:k]) for fb in per_file]
            out[i] = sorted(set.intersection(*sets), key=_key_order)
        return out

    t = max(0.0, prune_fraction)
    inter = inter_at(t)
    it = 0
    while min((len(v) for v in inter.values()), default=0) < k_common and t < 1.0 and it < 100:
        t = min(1.0, t * 1.2 if t > 0 else 0.02)
        inter = inter_at(t)
        it += 1
    mask: Dict[str, int] = {}
    for i in blocks:
        for k in common_keys[i]:
            mask[k] = 0
        keys = inter.get(i, [])
        if len(keys) > k_common:
            means = [(k, sum(fb.get(i, {}).get(k, float("inf")) for fb in per_file) / max(1, len(per_file))) for k in keys]
            keys = [k for k, _ in sorted(means, key=lambda kv: (kv[1], _key_order(kv[0])))[:k_common]]
        for k in keys:
            mask[k] = 1
    return mask
