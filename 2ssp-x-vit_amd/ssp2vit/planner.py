"""(K blocks, t neurons) allocation for a single target sparsity — host arithmetic only.

Restates the decision procedure of /root/reference/src/vit_pruning.py:585-769 (`plan_2ssp_allocation`) on
plain parameter counts, so it also runs without a live module.  Known answers captured from the reference are
in tests/golden/planner.json.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence


@dataclass
class TwoSSPPlan:  # field names and order as the reference dataclass (src/vit_pruning.py:564-572)
    target_sparsity: float
    num_blocks_total: int
    blocks_to_prune: int
    per_block_neurons_to_prune: int
    stage2_fraction: float
    estimated_total_removed_params: int
    est_error_params: int


@dataclass
class ModelStats:
    total_params: int
    hidden: int
    inter_sizes: List[int]
    attn_params: List[int]     # per block, attention sub-module only
    ffn_params: List[int]      # per block, fc1 + fc2 incl. biases


ALPHA = 1.5  # paper Eq.: N_attn = round(B * s^(|W_FFN| / (alpha * |W_Attn|)))   (reference :636-639)


@dataclass
class _Cand:
    err: int
    k: int
    t: int
    removed: int


def _prefer(new: _Cand, old: Optional[_Cand], tol: int) -> bool:
    """Strictly smaller error wins; errors within `tol` of each other -> more attention blocks wins (:690)."""
    if old is None:
        return True
    return (new.err < old.err - tol) or (abs(new.err - old.err) <= tol and new.k > old.k)


def plan_from_stats(st: ModelStats, target_sparsity: float, min_remaining: int = 256,
                    forced_blocks: Optional[int] = None) -> TwoSSPPlan:
    assert 0.0 < target_sparsity < 1.0, "target_sparsity must be in (0,1)"
    B = len(st.inter_sizes)
    goal = int(round(st.total_params * target_sparsity))
    t_cap = min([max(0, d - min_remaining) for d in st.inter_sizes]) if st.inter_sizes else 0
    per_t = 2 * st.hidden + 1                      # params removed per neuron per block (fc1 row+bias, fc2 col)
    unit = B * per_t
    tol = max(1, int(0.02 * goal))
    attn_mean = sum(st.attn_params) / max(1, B)
    ffn_mean = sum(st.ffn_params) / max(1, B)

    def width_only(k: int, t: int) -> _Cand:
        depth_part = int(round(k * attn_mean))
        removed = depth_part + (t * per_t if t > 0 else 0) * B
        return _Cand(abs(goal - removed), k, t, removed)

    def base_t(k: int) -> int:
        left = max(0, goal - int(round(k * attn_mean)))
        t = int(round(left / unit)) if unit > 0 else 0
        return max(0, min(t, t_cap))

    if forced_blocks is not None:
        ks = [max(0, min(B - 1, int(forced_blocks)))]
    else:
        k0 = int(round(B * (target_sparsity ** (ffn_mean / (ALPHA * attn_mean))))) if attn_mean > 0 else 0
        k0 = max(0, min(B - 1, k0))
        ks = [k for k in sorted({k0 + d for d in (-2, -1, 0, 1, 2)}) if 0 <= k <= B - 1]

    best: Optional[_Cand] = None
    for k in ks:
        t = base_t(k)
        for tt in (t, t - 1, t + 1, t + 2, t - 2):          # same visiting order as the reference (:684, :694)
            c = width_only(k, max(0, min(tt, t_cap)))
            if _prefer(c, best, tol):
                best = c

    # all-width result although the budget is worth at least half an attention block: look for K >= 1 (:710-738)
    if best is not None and forced_blocks is None and best.k == 0 and attn_mean > 0 and goal >= 0.5 * attn_mean:
        k_guess = max(1, int(round(goal / max(1, attn_mean))))
        alt: Optional[_Cand] = None
        for k in range(1, min(B - 1, k_guess + 2) + 1):
            c = width_only(k, base_t(k))
            if _prefer(c, alt, tol):
                alt = c
        if alt is not None and ((alt.err < best.err - tol) or abs(alt.err - best.err) <= tol):
            best = alt

    if best is None:
        return TwoSSPPlan(target_sparsity, B, 0, 0, 0.0, 0, goal)
    return TwoSSPPlan(target_sparsity, B, best.k, best.t, (best.k / B) if B > 0 else 0.0, best.removed, int(best.err))


def stats_from_shapes(dim: int, depth: int, inter: Sequence[int] | int, classes: int, tokens: int, patch: int) -> ModelStats:
    """Parameter counts of a standard ViT (fused or split qkv count the same)."""
    inters = list(inter) if not isinstance(inter, int) else [inter] * depth
    attn = [4 * dim * dim + 4 * dim] * depth
    ffn = [d * dim + d + dim * d + dim for d in inters]
    blocks = sum(a + f + 4 * dim for a, f in zip(attn, ffn))
    total = blocks + (3 * patch * patch * dim + dim) + dim + tokens * dim + 2 * dim + (classes * dim + classes)
    return ModelStats(total, dim, inters, attn, ffn)
