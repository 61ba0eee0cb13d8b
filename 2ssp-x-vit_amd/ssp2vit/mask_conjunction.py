"""Host-side mirror of the framework plug-in contract (/root/reference/adaptation-for-Pures-framework/
mask_conjunction.py): `PruningTypes` (:32-36), `PruningInterface` (:38-88) and this method's implementation
`Auto2SSPInterface` (:236-362).  `fit()` returns `(att_importance, mlp_importance)`:
Tensor[n_blocks] float32 for DEPTH attention pruning and List[Tensor[d_int]] for WIDTH MLP pruning, detached CPU
tensors, "lower importance => pruned earlier".  Both importances are computed on the MI355X by libssp2vit.

Deviation kept on purpose: the reference silently falls back to weight-L1 scores when the activation pass raises
(:286-287).  Here an engine failure (no GPU, missing library, unsupported shape) propagates — a silent fallback
would hide that the native path did not run.  The L1 fallback is kept only for `pruning_dataloader=None`.
"""
from __future__ import annotations

from enum import Enum
from typing import List

import torch

from . import vit_pruning as _vp


class PruningTypes(Enum):
    DEPTH = 0
    WIDTH = 1
    HEAD = 2   # attention only
    NONE = 3   # structure not pruned by this method


class PruningInterface:
    def __init__(self, model, pruning_dataloader):
        self.nn = model
        self.dl = pruning_dataloader
        self.att_prune_type = PruningTypes.DEPTH
        self.mlp_prune_type = PruningTypes.WIDTH

    def fit(self):  # pragma: no cover - abstract in spirit
        raise NotImplementedError


class Auto2SSPInterface(PruningInterface):
    def __init__(self, model, pruning_dataloader, device=None, importance_mode="copy", batch_limit=5,
                 min_remaining=256, error_policy="raise", *, score_chain="fp32", process_group=None):
        super().__init__(model, pruning_dataloader)
        self.att_prune_type = PruningTypes.DEPTH
        self.mlp_prune_type = PruningTypes.WIDTH
        self.device = device or "cuda"
        self.importance_mode = importance_mode
        self.batch_limit = batch_limit
        self.min_remaining = min_remaining
        self.error_policy = error_policy
        self.score_chain = score_chain
        self.process_group = process_group

    def _num_blocks(self) -> int:
        return len(_vp._blocks(self.nn)[0])

    def _mlp_importance_deferred(self):
        """All device work enqueued; the returned callable waits and hands back the CPU tensors."""
        if self.dl is not None:
            fin = _vp._compute_ffn_activation_importance(self.nn, self.dl, device=self.device,
                                                         batch_limit=self.batch_limit, progress=False,
                                                         score_chain=self.score_chain,
                                                         process_group=self.process_group,
                                                         defer=(self.score_chain == "fp32"))
            if callable(fin):
                return lambda: [t.detach().to("cpu") for t in fin()]
            return lambda: [t.detach().to("cpu") for t in fin]
        imps = [fc1.weight.abs().sum(dim=1).detach().to("cpu") for fc1, _ in _vp._gather_mlp_pairs(self.nn)]
        return lambda: imps

    def _compute_mlp_importance(self) -> List[torch.Tensor]:
        return self._mlp_importance_deferred()()

    def _heuristic(self) -> torch.Tensor:
        B = self._num_blocks()
        return torch.tensor([(i if i < B / 2 else B - i) for i in range(B)], dtype=torch.float32)

    def _att_importance_deferred(self):
        if self.importance_mode.lower() == "heuristic" or self.dl is None:
            h = self._heuristic()
            return lambda: h
        try:
            fin = _vp.depth_search_counts(self.nn, self.dl, self.device, self.batch_limit,
                                          process_group=self.process_group, defer=True)
        except Exception:
            if getattr(self, "error_policy", "raise") == "raise":
                raise
            h = self._heuristic()
            return lambda: h

        def finish():
            try:
                base, cand, total = fin()
            except Exception:
                if getattr(self, "error_policy", "raise") == "raise":
                    raise
                return self._heuristic()
            baseline = float(base / max(1, total))
            return torch.tensor([max(0.0, baseline - float(c / max(1, total))) for c in cand], dtype=torch.float32)
        return finish

    def _compute_att_depth_importance(self) -> torch.Tensor:
        return self._att_importance_deferred()()

    def fit(self):
        """Attention first, then MLP, as the reference orders them (:359-362) — but both stages are ENQUEUED before the
        host waits for either: the two are independent (stage 2 evaluates the dense model), so the GPU never idles
        between them and the engine (built once, with the layer-major search's workspace) serves both."""
        att = self._att_importance_deferred()
        mlp = self._mlp_importance_deferred()
        self.att_importance = att()
        self.mlp_importance = mlp()
        return self.att_importance, self.mlp_importance
