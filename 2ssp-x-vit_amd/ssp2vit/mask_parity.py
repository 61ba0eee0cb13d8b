"""Cut-margin / tie-band diagnostics of the mask step (SURVEY.md section 7 "hard parts" (iii); reference
src/vit_pruning.py:273-295).  Host arithmetic only."""
from __future__ import annotations

import os
from typing import Any, Dict, List, Optional, Sequence

import torch

# The mask step (reference :273-295) is a discrete function of the scores: `argsort(imp, descending)[:keep]`.  Two runs
# whose scores differ by a relative error of at most e give the SAME mask whenever the gap between the weakest kept and
# the strongest pruned neuron exceeds 2e (no pair can then change sides), and may differ otherwise; with the
# reference's own bf16 score chain exact ties across the cut are decided by an unstable sort (:286) and nothing can be
# promised at all.  MASK_PARITY_EPS is that gap threshold: twice the 5e-4 bound on the engine's fp32-chain score error
# against the CPU restatement (measured on MI355X: <= 4.6e-4 ViT-Ti/16, <= 3.0e-4 ViT-B/16).
MASK_PARITY_EPS = float(os.environ.get("SSP2_MASK_PARITY_EPS", "1e-3"))
# The POST-GELU hook site (old-HF / HF >= 5 anatomy, reference :130) is noisier than the pre-GELU one (timm, :135): a weak neuron's
# score is a norm of GELU-tail values, whose relative sensitivity to a one-ulp bf16 flip of the pre-activation is several times
# that of the pre-activation itself.  Its band is EMPIRICAL and the report says so (`basis`): largest per-element relative error of the
# fp32 chain against the CPU restatement over every (model, images) sample measured on MI355X so far — ViT-B/16 HF 64 samples 1.6e-3 /
# 1.9e-3 (two weight sets), 24 samples 2.4e-3; ViT-L/16 HF 24 samples 4.3e-3 .. 4.7e-3 (two seeds; the fewer samples, the less the
# outliers average out) — pre-GELU on the same weights 2.7e-4 .. 6.6e-4 (profiles/r04_c_site_diag.txt, r05_e_decisive_stage2.log).
# Error bound = the largest of them + 25 % = 5.9e-3; the band is twice that.  (Round 4 used 1e-2 = 2.1 x the worst sample)
POST_GELU_ERROR_BOUND = 5.9e-3
MASK_PARITY_EPS_POST_GELU = float(os.environ.get("SSP2_MASK_PARITY_EPS_POST_GELU", str(2 * POST_GELU_ERROR_BOUND)))


def eps_for_site(site: Optional[str]) -> float:
    return MASK_PARITY_EPS_POST_GELU if site == "post_gelu" else MASK_PARITY_EPS


def mask_parity_report(scores: Sequence[torch.Tensor], n_prune_per_block: Sequence[int], min_remaining: int = 256,
                       eps: Optional[float] = None, site: Optional[str] = None) -> Dict[str, Any]:
    """Per block, for the cut the mask step is about to make on `scores`:
      cut_margin        (weakest kept - strongest pruned) / weakest kept, relative
      tie_band          neurons within the +-eps band of the cut: kept ones at most (1+eps) x the strongest pruned score
                        plus pruned ones at least (1-eps) x the weakest kept score — the only neurons whose side a score
                        error below eps/2 could change (0 <=> cut_margin > eps)
      exact_ties        neurons whose score EQUALS a score on the other side of the cut (bf16 chains: sort-order lottery)
      guaranteed        tie_band == 0: the mask equals the one a CPU run of the reference algorithm makes from its own
                        fp32 scores, as long as the two score vectors agree to eps/2
    plus `eps`, `blocks_guaranteed`, `min_margin`.  Pure host arithmetic on the final [L][d_int] score vectors."""
    eps = eps_for_site(site) if eps is None else float(eps)          # `site`: the hook site the scores were taken at ("pre_gelu" default)
    blocks: List[Dict[str, Any]] = []
    # equal widths (every model before a width prune): ONE descending sort of the [L, d_int] matrix instead of L of them — the report is
    # this build's addition to prune_vit_mlp_width's result and sits inside the prune bracket of the reference-named API (3.8 -> 0.6 ms)
    pre_sorted = None
    widths = {int(t.numel()) for t in scores}
    if len(scores) > 1 and len(widths) == 1:
        # (sorted as fp32 — every score chain yields fp32-representable values — and widened afterwards: torch's 2-D float64 sort is 40 x slower)
        pre_sorted = torch.sort(torch.stack([t.detach().to("cpu", torch.float32).view(-1) for t in scores]), dim=1, descending=True).values.double()
    for b, imp in enumerate(scores):
        width = int(imp.numel())
        drop = int(n_prune_per_block[b])
        if width - drop < min_remaining:
            drop = max(0, width - min_remaining)
        if drop <= 0 or drop >= width:
            blocks.append({"block": b, "pruned": max(0, min(drop, width)), "cut_margin": None, "tie_band": 0,
                           "exact_ties": 0, "guaranteed": True})
            continue
        s = pre_sorted[b] if pre_sorted is not None else torch.sort(imp.detach().to("cpu", torch.float64).view(-1), descending=True).values
        kept_min, pruned_max = float(s[width - drop - 1]), float(s[width - drop])
        margin = (kept_min - pruned_max) / kept_min if kept_min > 0 else 0.0
        kept, pruned = s[: width - drop], s[width - drop:]
        band = int((kept <= pruned_max * (1.0 + eps)).sum()) + int((pruned >= kept_min * (1.0 - eps)).sum())
        ties = (int((kept == pruned_max).sum()) + int((pruned == kept_min).sum())) if kept_min == pruned_max else 0
        blocks.append({"block": b, "pruned": drop, "cut_margin": margin, "tie_band": band, "exact_ties": ties,
                       "guaranteed": band == 0})
    margins = [x["cut_margin"] for x in blocks if x["cut_margin"] is not None]
    basis = ("empirical: twice (the largest fp32-chain score error measured on MI355X against the CPU restatement + 25 %); not a derived bound"
             if site == "post_gelu" else "twice the 5e-4 bound on the fp32-chain score error (measured <= 4.6e-4 on MI355X)")
    return {"eps": eps, "score_site": site or "pre_gelu", "basis": basis, "blocks": blocks, "blocks_guaranteed": sum(1 for x in blocks if x["guaranteed"]),
            "blocks_total": len(blocks), "min_margin": min(margins) if margins else None,
            "rule": "mask == CPU-reference mask from fp32 scores wherever tie_band == 0 (cut_margin > eps = 2 x score error bound)"}
