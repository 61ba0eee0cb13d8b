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
                 min_remaining=256, error_policy="raise", *, score_chain="fp32", process_group=None, one_pass=True, score_batch_limit="same"):
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
        self.score_batch_limit = score_batch_limit      # extension: a different batch limit for the stage-1 hook ("same": batch_limit, the reference's one rule)
        self.one_pass = one_pass            # fit(): one dense pass over self.dl feeds both importances (False: two passes, as the reference runs them)

    def _num_blocks(self) -> int:
        return len(_vp._blocks(self.nn)[0])

    def _mlp_importance_deferred(self):
        """All device work enqueued; the returned callable waits and hands back the CPU tensors."""
        if self.dl is not None:
            fin = _vp._compute_ffn_activation_importance(self.nn, self.dl, device=self.device,
                                                         batch_limit=self.batch_limit if self.score_batch_limit == "same" else self.score_batch_limit,
                                                         progress=False,
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

    def _both_deferred(self):
        """One walk over self.dl for both importances (vit_pruning.importances_one_pass): the reference hooks self.dl[:batch_limit]
        (:276-281) and evaluates the dense baseline on self.dl[:batch_limit] again (:327) — the same batches, the same dense forward.
        None when the two do not share a loader pass (heuristic depth scores, no loader, bf16_ref chain without labels ...)."""
        if self.dl is None or self.importance_mode.lower() == "heuristic" or not self.one_pass:
            return None
        try:
            sc, se = _vp.importances_one_pass(self.nn, self.dl, self.device, self.batch_limit, score_limit=self.score_batch_limit,
                                              score_chain=self.score_chain, process_group=self.process_group, defer=True)
        except KeyError:                # a loader without labels cannot feed the search: the reference would fail there too; keep its order of failure
            return None

        def att():
            base, cand, total = se()
            baseline = float(base / max(1, total))
            return torch.tensor([max(0.0, baseline - float(c / max(1, total))) for c in cand], dtype=torch.float32)

        def mlp():
            got = sc() if callable(sc) else sc
            return [t.detach().to("cpu") for t in got]
        return att, mlp

    def fit(self):
        """Attention first, then MLP, as the reference orders them (:359-362).  Both importances come from ONE walk over the
        loader whose dense forward serves the stage-1 hook and the search's baseline alike (`one_pass`, default on); everything is
        ENQUEUED before the host waits, and the engine (built once, with the layer-major search's workspace) serves both."""
        both = None
        try:
            both = self._both_deferred()
        except Exception:
            if getattr(self, "error_policy", "raise") == "raise":
                raise
        if both is not None:
            att, mlp = both
            # The scores reach the host before the search's CLS-only tails have finished (core.prune_pass sends them first), so the host
            # takes them FIRST and sorts them while the card is still busy: the descending argsort the a7 mask step of a following
            # prune_vit_mlp_width(precomputed_importance=...) starts with (vit_pruning.precompute_orders).  Results and the order of the
            # returned pair are the reference's (:359-362: attention, then MLP); a failure of the MLP half still surfaces after the
            # attention half's, as it would there.
            mlp_err = None
            try:
                self.mlp_importance = mlp()
                _vp.precompute_orders(self.mlp_importance)
            except Exception as e:                            # noqa: BLE001 - re-raised below, in the reference's order
                mlp_err = e
            try:
                self.att_importance = att()
            except Exception:
                if getattr(self, "error_policy", "raise") == "raise":
                    raise
                self.att_importance = self._heuristic()
            if mlp_err is not None:
                raise mlp_err
            return self.att_importance, self.mlp_importance
        att = self._att_importance_deferred()
        mlp = self._mlp_importance_deferred()
        self.att_importance = att()
        self.mlp_importance = mlp()
        return self.att_importance, self.mlp_importance
