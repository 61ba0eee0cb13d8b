"""Property tests (hypothesis; SURVEY.md §4 test plan item 3) of the host-side rows of §8: planner (f3), mask step (a7 / a8),
stage-2 selection (a9), the dealing of batches over ranks (e).

Where the REAL reference is importable (this build container: /root/reference; it never travels) every generated case is also run
through the reference's own function and the results must be EQUAL — random geometries, scores with ties, clamps and forced
block counts that the committed known-answer fixtures do not reach.  Elsewhere only the invariants are checked.
Derandomised: the same cases on every run; no example database is written.
"""
import copy
import os
import sys

import pytest
import torch

from conftest import ROOT  # noqa: F401  (import paths)

hypothesis = pytest.importorskip("hypothesis")
from hypothesis import given, settings, strategies as st  # noqa: E402

REF = "/root/reference"
_ref_vp = None


def ref_vp():
    """The reference's src/vit_pruning module, or None where /root/reference does not exist (every box but the build container)."""
    global _ref_vp
    if _ref_vp is None:
        p = os.path.join(REF, "adaptation-for-Pures-framework")
        if not os.path.isdir(p):
            _ref_vp = False
        else:
            sys.path.insert(0, p)
            try:
                from src import vit_pruning as m
                _ref_vp = m
            except Exception:                    # an ordinary import error: the invariants still run
                _ref_vp = False
            finally:
                sys.path.remove(p)
    return _ref_vp or None


def quiet(fn, *a, **k):
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def tiny_model(layout, heads, head_dim, inter, depth, classes, seed):
    from oracle.vit_modules import HFLayoutViT, TimmLayoutViT
    torch.manual_seed(seed)
    cls = TimmLayoutViT if layout == "timm" else HFLayoutViT
    return cls(img=32, patch=16, dim=heads * head_dim, heads=heads, inter=inter, depth=depth, classes=classes).eval()


CFG = settings(max_examples=60, deadline=None, database=None, derandomize=True)
geometry = dict(layout=st.sampled_from(["timm", "hf"]), heads=st.sampled_from([1, 2, 4]), head_dim=st.sampled_from([4, 8]),
                inter=st.integers(6, 96), depth=st.integers(2, 9), classes=st.sampled_from([3, 10]))


@CFG
@given(target=st.floats(0.02, 0.95), min_remaining=st.sampled_from([0, 4, 16, 64]), forced=st.one_of(st.none(), st.integers(0, 10)), **geometry)
def test_planner_invariants_and_equality_with_the_reference(layout, heads, head_dim, inter, depth, classes, target, min_remaining, forced):
    """`plan_2ssp_allocation` (reference src/vit_pruning.py:585-769): K stays below the depth, t respects the `min_remaining` clamp, the
    bookkeeping fields are consistent — and every field equals the reference's on the same module."""
    from ssp2vit import vit_pruning as vp
    m = tiny_model(layout, heads, head_dim, inter, depth, classes, 0)
    plan = vp.plan_2ssp_allocation(m, target, min_remaining=min_remaining, forced_blocks=forced)
    total = sum(p.numel() for p in m.parameters())
    dim = heads * head_dim
    assert plan.num_blocks_total == depth and 0 <= plan.blocks_to_prune <= depth - 1
    assert 0 <= plan.per_block_neurons_to_prune <= max(0, inter - min_remaining)
    if forced is not None:
        assert plan.blocks_to_prune == max(0, min(depth - 1, forced))
    attn = 4 * dim * dim + 4 * dim
    removed = plan.blocks_to_prune * attn + depth * plan.per_block_neurons_to_prune * (2 * dim + 1)
    assert plan.estimated_total_removed_params == removed
    assert plan.est_error_params == abs(int(round(total * target)) - removed)
    assert plan.stage2_fraction == plan.blocks_to_prune / depth
    r = ref_vp()
    if r is not None:
        want = quiet(r.plan_2ssp_allocation, m, target, min_remaining=min_remaining, forced_blocks=forced)
        assert plan.__dict__ == want.__dict__


@CFG
@given(t=st.integers(0, 200), min_remaining=st.sampled_from([1, 4, 16]), levels=st.integers(1, 40), seed=st.integers(0, 10 ** 6), **geometry)
def test_mask_step_with_ties_equals_the_reference(layout, heads, head_dim, inter, depth, classes, t, min_remaining, levels, seed):
    """a7 / a8 (reference :256-311) on importances drawn from `levels` distinct values — TIES at the cut are the rule, not the exception —
    with a different count per block, some beyond the `min_remaining` clamp, some zero: exactly n_prune ones per mask, no pruned score
    above a kept one, the compacted fc1 is the kept rows in ascending order, an untouched block has no mask entry — and masks, index
    lists and every weight tensor equal the reference's on a copy of the same module."""
    from ssp2vit import vit_pruning as vp
    m = tiny_model(layout, heads, head_dim, inter, depth, classes, seed % 7)
    g = torch.Generator().manual_seed(seed)
    imps = [torch.randint(0, levels, (inter,), generator=g).float() for _ in range(depth)]
    m_ref = copy.deepcopy(m)
    fc1_before = [p.detach().clone() for n, p in m.named_parameters() if n.endswith(("fc1.weight", "intermediate.dense.weight"))]
    ts = [(t + 5 * b) % (inter + 20) for b in range(depth)]          # a different count per block, some beyond the clamp
    out = quiet(vp.prune_vit_mlp_width, m, n_to_prune_per_block=list(ts), min_remaining=min_remaining, precomputed_importance=[i.clone() for i in imps],
                collect_masks=True, device="cpu")
    drops = [max(0, min(ts[b], inter - min_remaining)) for b in range(depth)]
    pruned_blocks = [b for b in range(depth) if drops[b] > 0]          # a block that loses nothing gets NO mask entry (reference :282-283)
    assert len(out["ffn_prune_masks"]) == len(out["ffn_pruned_indices"]) == len(pruned_blocks)
    fc1_after = [p for n, p in out["model"].named_parameters() if n.endswith(("fc1.weight", "intermediate.dense.weight"))]
    for b in range(depth):
        if drops[b] == 0:
            assert torch.equal(fc1_after[b], fc1_before[b])
    for b, mask, idx in zip(pruned_blocks, out["ffn_prune_masks"], out["ffn_pruned_indices"]):
        mask_t = torch.as_tensor(mask)
        assert int(mask_t.sum()) == drops[b] == len(idx) and sorted(idx) == list(idx)
        assert torch.nonzero(mask_t).flatten().tolist() == list(idx)
        if drops[b] < inter:
            assert float(imps[b][mask_t.bool()].max()) <= float(imps[b][~mask_t.bool()].min())
        assert torch.equal(fc1_after[b], fc1_before[b][~mask_t.bool()])
    r = ref_vp()
    if r is not None:
        want = quiet(r.prune_vit_mlp_width, m_ref, n_to_prune_per_block=list(ts), min_remaining=min_remaining,
                     precomputed_importance=[i.clone() for i in imps], collect_masks=True, device="cpu")
        assert [list(map(int, a)) for a in want["ffn_prune_masks"]] == [list(map(int, a)) for a in out["ffn_prune_masks"]]
        assert [list(map(int, a)) for a in want["ffn_pruned_indices"]] == [list(map(int, a)) for a in out["ffn_pruned_indices"]]
        for (na, pa), (nb, pb) in zip(want["model"].named_parameters(), out["model"].named_parameters()):
            assert na == nb and torch.equal(pa, pb), na


@CFG
@given(depth=st.integers(2, 32), k=st.integers(0, 31), levels=st.integers(1, 6), seed=st.integers(0, 10 ** 6))
def test_stage2_selection_is_the_references_argsort_prefix(depth, k, levels, seed):
    """a9 (reference auto_2ssp.py:857 `torch.argsort(att_imp)[:K]`, applied sorted): with impacts full of ties the selection is what
    `core.select_for_targets` returns for a plan with K blocks — the K lowest impacts, no selected impact above an unselected one."""
    from ssp2vit import core
    from ssp2vit.planner import TwoSSPPlan
    g = torch.Generator().manual_seed(seed)
    impact = (torch.randint(0, levels, (depth,), generator=g).float() / 64.0)
    k = min(k, depth - 1)
    imps = [torch.rand(8, generator=g) for _ in range(depth)]
    plan = TwoSSPPlan(0.3, depth, k, 2, k / depth, 0, 0)
    (sel,) = core.select_for_targets(imps, impact, [plan], min_remaining=1)
    chosen = sel["blocks"]
    want = sorted(int(i) for i in torch.argsort(impact)[:k])
    assert chosen == want and len(set(chosen)) == k
    if 0 < k < depth:
        rest = [i for i in range(depth) if i not in chosen]
        assert float(impact[chosen].max()) <= float(impact[rest].min())
    assert len(sel["masks"]) == depth and all(int(m.sum()) == 2 for m in sel["masks"])


@CFG
@given(n=st.integers(0, 500), batch=st.integers(1, 70), world=st.integers(1, 9), limit=st.one_of(st.none(), st.integers(0, 12)))
def test_batches_are_dealt_to_ranks_without_loss_or_overlap(n, batch, world, limit):
    """§8(e): batch b -> rank b % P.  Over all ranks every item of the first `limit` batches appears exactly once, a rank's batches keep
    the global order, batch b is owned by rank b % P, and only the globally last batch may be ragged."""
    from ssp2vit import dist as D
    per_rank = [D.rank_batch_indices(n, batch, r, world, limit) for r in range(world)]
    n_batches = (n + batch - 1) // batch
    if limit is not None:
        n_batches = min(n_batches, limit)
    flat = sorted(i for rb in per_rank for b in rb for i in b)
    assert flat == list(range(min(n, n_batches * batch)))
    for r, rb in enumerate(per_rank):
        starts = [b[0] for b in rb]
        assert starts == sorted(starts)
        for b in rb:
            gb = b[0] // batch
            assert gb % world == r and D.owns(gb, r, world) and list(b) == list(range(b[0], b[0] + len(b)))
            assert len(b) == batch or gb == (n + batch - 1) // batch - 1
        assert len(rb) == len(range(r, n_batches, world))


def test_the_reference_is_really_compared_where_it_exists():
    """The equality halves above must not vanish silently: where /root/reference is present its module has to import."""
    if os.path.isdir(REF):
        r = ref_vp()
        assert r is not None and hasattr(r, "plan_2ssp_allocation") and hasattr(r, "prune_vit_mlp_width")
        assert os.path.realpath(r.__file__).startswith(os.path.realpath(REF))
        assert hasattr(r, "_compute_ffn_activation_importance") and hasattr(r, "evaluate_top1")
    else:
        assert ref_vp() is None


_ref_mc = None


def ref_mc():
    """The reference's mask_conjunction module (Auto2SSPInterface), or None where /root/reference does not exist."""
    global _ref_mc
    if _ref_mc is None:
        p = os.path.join(REF, "adaptation-for-Pures-framework")
        if not os.path.isdir(p):
            _ref_mc = False
        else:
            sys.path.insert(0, p)
            try:
                import mask_conjunction as m
                _ref_mc = m
            except Exception:
                _ref_mc = False
            finally:
                sys.path.remove(p)
    return _ref_mc or None


ORACLE_CFG = settings(max_examples=25, deadline=None, database=None, derandomize=True)


@ORACLE_CFG
@given(layout=st.sampled_from(["timm", "hf"]), heads=st.sampled_from([1, 2, 4]), head_dim=st.sampled_from([8, 16]), inter=st.integers(8, 80),
       depth=st.integers(1, 5), sizes=st.lists(st.integers(1, 7), min_size=1, max_size=4), limit=st.one_of(st.none(), st.integers(0, 5)),
       seed=st.integers(0, 10 ** 6))
def test_the_oracle_equals_the_reference_on_random_tiny_models(layout, heads, head_dim, inter, depth, sizes, limit, seed):
    """§8(c): the oracle (oracle/ref_cpu.py — the checker of every GPU parity test) against the reference ITSELF on geometries and
    loaders the committed fixtures do not hold: both anatomies, 1-5 blocks, ragged batches of 1-7 images, batch limits from 0 (no batch
    at all) upwards.  Stage-1 scores in the reference's bf16 chain must be the same BITS and dtype, top-1 the same float, the
    depth-importance vector the same tensor.  Without the reference (every box but the build container) the oracle's own
    consistency is checked: a limit beyond the loader changes nothing, a limit of 0 gives zeros."""
    from oracle import ref_cpu
    torch.set_num_threads(2)
    m = tiny_model(layout, heads, head_dim, inter, depth, 10, seed % 5)
    g = torch.Generator().manual_seed(seed)
    batches = [{"pixel_values": torch.randn(n, 3, 32, 32, generator=g), "labels": torch.randint(0, 10, (n,), generator=g)} for n in sizes]
    imps = ref_cpu.ffn_activation_importance(m, batches, batch_limit=limit)
    acc = ref_cpu.evaluate_top1(m, batches, limit)
    assert len(imps) == depth and all(t.shape == (inter,) for t in imps)
    if limit == 0:
        assert all(float(t.abs().sum()) == 0.0 for t in imps) and acc == 0.0
    if limit is not None and limit >= len(sizes):
        again = ref_cpu.ffn_activation_importance(m, batches, batch_limit=None)
        assert all(torch.equal(a, b) for a, b in zip(imps, again))
    rv, rm = ref_vp(), ref_mc()
    if rv is not None:
        want = quiet(rv._compute_ffn_activation_importance, m, batches, device="cpu", batch_limit=limit)
        for a, b in zip(want, imps):
            assert a.dtype == b.dtype and torch.equal(a, b)
        assert quiet(rv.evaluate_top1, m, batches, device="cpu", max_batches=limit) == acc
    if rm is not None and depth >= 2:
        iface = rm.Auto2SSPInterface(m, batches, device="cpu", importance_mode="copy", batch_limit=limit if limit is not None else 5)
        want = quiet(iface._compute_att_depth_importance)
        got = ref_cpu.att_depth_importance(m, batches, batch_limit=limit if limit is not None else 5)
        assert want.dtype == got.dtype and torch.equal(want, got)


def test_the_reference_interface_is_really_compared_where_it_exists():
    if os.path.isdir(REF):
        m = ref_mc()
        assert m is not None and hasattr(m, "Auto2SSPInterface") and os.path.realpath(m.__file__).startswith(os.path.realpath(REF))
    else:
        assert ref_mc() is None


@pytest.mark.parametrize("kind", ["linear", "adapter"])
def test_classifier_head_files_round_trip_and_are_the_references(kind, tmp_path):
    """save_cifar_adapter / load_cifar_adapter (reference src/vit_pruning.py:774-875, in its `__all__`): the head of an HF-layout model — a
    Linear classifier or the bottleneck adapter Sequential(Linear(bias=False), GELU, Linear) — goes to one file and comes back onto another
    model with the same logits; the file's keys are the reference's; metadata missing from the file is read off the tensors; the three
    RuntimeErrors.  Where the real reference is importable its loader must read THIS build's file and this loader the reference's, with
    equal tensors."""
    from ssp2vit import vit_pruning as vp
    import torch.nn as nn
    src = tiny_model("hf", 2, 8, 24, 2, 10, seed=3)
    if kind == "adapter":
        torch.manual_seed(4)
        src.classifier = nn.Sequential(nn.Linear(16, 5, bias=False), nn.GELU(), nn.Linear(5, 7, bias=True))
        src.config.num_labels = 7
    path = vp.save_cifar_adapter(src, str(tmp_path), "head.pt", extra={"note": 1})
    blob = torch.load(path, map_location="cpu", weights_only=True)
    assert sorted(blob) == sorted(["state_dict", "classifier_type", "num_labels", "hidden_size", "timestamp", "extra"])
    assert blob["classifier_type"] == ("Linear" if kind == "linear" else "Sequential") and blob["extra"] == {"note": 1}
    assert blob["num_labels"] == (10 if kind == "linear" else 7) and blob["hidden_size"] == 16
    dst = tiny_model("hf", 2, 8, 24, 2, 3, seed=9)
    assert vp.load_cifar_adapter(path, dst) is dst and dst.config.num_labels == blob["num_labels"]
    feats = torch.randn(5, 16, generator=torch.Generator().manual_seed(1))
    assert torch.equal(dst.classifier(feats), src.classifier(feats))
    # metadata missing: shapes come from the tensors
    bare = {"state_dict": blob["state_dict"], "classifier_type": blob["classifier_type"]}
    torch.save(bare, tmp_path / "bare.pt")
    again = tiny_model("hf", 2, 8, 24, 2, 3, seed=10)
    vp.load_cifar_adapter(str(tmp_path / "bare.pt"), again)
    assert torch.equal(again.classifier(feats), src.classifier(feats)) and again.config.num_labels == blob["num_labels"]
    # the reference's failures
    torch.save({"state_dict": {}, "classifier_type": "Linear"}, tmp_path / "empty.pt")
    nohid = tiny_model("hf", 2, 8, 24, 2, 3, seed=11); nohid.config.hidden_size = None
    with pytest.raises(RuntimeError, match="hidden size"):
        vp.load_cifar_adapter(str(tmp_path / "empty.pt"), nohid)
    with pytest.raises(RuntimeError, match="num_labels is None"):
        vp.load_cifar_adapter(str(tmp_path / "empty.pt"), tiny_model("hf", 2, 8, 24, 2, 3, seed=12))
    torch.save({"state_dict": {}, "classifier_type": "Sequential"}, tmp_path / "empty_seq.pt")
    with pytest.raises(RuntimeError, match="reconstruct adapter"):
        vp.load_cifar_adapter(str(tmp_path / "empty_seq.pt"), tiny_model("hf", 2, 8, 24, 2, 3, seed=13))
    R = ref_vp()
    if R is not None:
        theirs = tiny_model("hf", 2, 8, 24, 2, 3, seed=14)
        R.load_cifar_adapter(path, theirs)                                  # the reference reads this build's file
        assert torch.equal(theirs.classifier(feats), src.classifier(feats)) and theirs.config.num_labels == blob["num_labels"]
        rpath = R.save_cifar_adapter(src, str(tmp_path), "ref_head.pt", extra={"note": 1})
        rblob = torch.load(rpath, map_location="cpu", weights_only=True)
        assert sorted(rblob) == sorted(blob) and all(torch.equal(rblob["state_dict"][k], blob["state_dict"][k]) for k in blob["state_dict"])
        assert {k: rblob[k] for k in ("classifier_type", "num_labels", "hidden_size", "extra")} == {k: blob[k] for k in ("classifier_type", "num_labels", "hidden_size", "extra")}
        ours = tiny_model("hf", 2, 8, 24, 2, 3, seed=15)
        vp.load_cifar_adapter(rpath, ours)                                  # and this build reads the reference's
        assert torch.equal(ours.classifier(feats), src.classifier(feats))
