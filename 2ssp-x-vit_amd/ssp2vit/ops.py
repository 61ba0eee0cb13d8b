"""torch.ops.ssp2vit.* — the custom-op face of the engine (csrc/torch_ops.cpp, a TORCH_LIBRARY shim over the C ABI).
These helpers only translate Python-side names (engine object, score-site strings) into the ops' plain arguments; the
same device work is reachable through ctypes (engine.VitEngine) — both end in the same extern "C" entry points."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import SCORE_CHAIN, SCORE_SITE


def _ops():
    return _lib.load_torch_ops()


def forward(engine, pixels: torch.Tensor, attn_skip: Optional[Sequence[int]] = None, score_site: str = "none",
            score_chain: str = "fp32", score_group: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """(logits f32 [n, classes], scores f32 [groups, depth, score_ld] — empty when score_site == "none")."""
    skip = sorted(set(int(i) for i in (attn_skip or ())) | {i for i, a in enumerate(engine.absent) if a})
    return _ops().forward(int(engine.h.value), pixels, skip, SCORE_SITE[score_site], SCORE_CHAIN[score_chain], int(score_group))


def act_l2_accum(act: torch.Tensor, score_chain: str = "fp32") -> torch.Tensor:
    return _ops().act_l2_accum(act, SCORE_CHAIN[score_chain])


def top1_count(engine, pixels: torch.Tensor, labels: torch.Tensor, attn_skip: Optional[Sequence[int]] = None) -> torch.Tensor:
    skip = sorted(set(int(i) for i in (attn_skip or ())) | {i for i, a in enumerate(engine.absent) if a})
    return _ops().top1_count(int(engine.h.value), pixels, labels, skip)
