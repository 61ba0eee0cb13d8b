"""Host logic of the ONE-PASS prune (core.prune_pass) on CPU, through the oracle-backed stand-in engine of test_dist_cpu.py (test
infrastructure: the product never constructs it; it has no prefix hook, so prune_pass takes its two-forward route — what is tested
here is the walking, routing, limits, dealing over ranks and the two exchange steps, which are the same code on the GPU).

Reference semantics being mirrored: Auto2SSPInterface walks ONE loader with ONE batch_limit for both stages
(adaptation-for-Pures-framework/mask_conjunction.py:276-281, :327, :345, :359-362)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import PKG, ROOT, load_tiny_golden
from test_dist_cpu import OracleBackedEngine, _batches, _free_port


def _setup(max_images=64):
    from oracle.vit_modules import build_from_flat
    w, batches = _batches()
    model = build_from_flat(w, "timm")
    return model, OracleBackedEngine(model, max_images=max_images), batches, [b.mlp.fc1.out_features for b in model.blocks]


@pytest.mark.parametrize("max_images", [64, 4])                  # layer-major stand-in path / candidate-major (one batch fits)
def test_one_pass_equals_the_two_passes_for_every_pair_of_limits(max_images):
    from ssp2vit import core
    model, eng, batches, d_ints = _setup(max_images)
    L = eng.depth
    for s_lim in (None, 0, 2, 5, 9):
        for q_lim in (None, 0, 1, 3, 5):
            ref_s = core.stage1_scores(eng, batches, d_ints, "pre_gelu", batch_limit=s_lim)
            ref_c = core.depth_search_counts(eng, batches, L, batch_limit=q_lim)
            got_s, got_c = core.prune_pass(eng, batches, d_ints, "pre_gelu", L, score_limit=s_lim, search_limit=q_lim, eval_chunk_images=8)
            assert all(torch.equal(a, b) for a, b in zip(got_s, ref_s)), (s_lim, q_lim)
            assert got_c == ref_c, (s_lim, q_lim)
    fs, fc = core.prune_pass(eng, batches, d_ints, "pre_gelu", L, score_limit=None, search_limit=3, defer=True)
    assert callable(fs) and callable(fc)
    assert all(torch.equal(a, b) for a, b in zip(fs(), core.stage1_scores(eng, batches, d_ints, "pre_gelu")))
    assert fc() == core.depth_search_counts(eng, batches, L, batch_limit=3)


def test_one_pass_walks_a_reshuffling_loader_once():
    """The reference's calibration loader reshuffles on every walk (auto_2ssp.py:348).  prune_pass iterates it ONCE; the order of that
    one walk feeds both stages (documented deviation: the reference's two stages see two different random orders)."""
    from ssp2vit import core

    class Reshuffling:
        def __init__(self, batches):
            self.batches, self.walks = batches, 0

        def __iter__(self):
            self.walks += 1
            order = list(range(len(self.batches)))
            order = order[self.walks % len(order):] + order[: self.walks % len(order)]      # another order on every walk
            return iter([self.batches[i] for i in order])

    model, eng, batches, d_ints = _setup()
    batches = batches[:4]                                        # equal sizes: any order is a valid dealing
    dl = Reshuffling(batches)
    got_s, got_c = core.prune_pass(eng, dl, d_ints, "pre_gelu", eng.depth, score_limit=3, search_limit=2)
    assert dl.walks == 1
    order = [1, 2, 3, 0]                                         # the order of walk 1
    ref_s = core.stage1_scores(eng, [batches[i] for i in order], d_ints, "pre_gelu", batch_limit=3)
    ref_c = core.depth_search_counts(eng, [batches[i] for i in order], eng.depth, batch_limit=2)
    assert all(torch.equal(a, b) for a, b in zip(got_s, ref_s)) and got_c == ref_c


def test_search_batches_need_labels_and_scores_only_batches_do_not():
    from ssp2vit import core
    model, eng, batches, d_ints = _setup()
    bare = [{"pixel_values": b["pixel_values"]} for b in batches]
    mixed = batches[:2] + bare[2:]
    got_s, got_c = core.prune_pass(eng, mixed, d_ints, "pre_gelu", eng.depth, score_limit=None, search_limit=2)
    assert got_c == core.depth_search_counts(eng, batches, eng.depth, batch_limit=2)
    assert all(torch.equal(a, b) for a, b in zip(got_s, core.stage1_scores(eng, bare, d_ints, "pre_gelu")))
    with pytest.raises(KeyError):
        core.prune_pass(eng, bare, d_ints, "pre_gelu", eng.depth, score_limit=None, search_limit=2)


def test_fit_takes_one_pass_and_keeps_the_reference_results():
    """Auto2SSPInterface.fit(): one walk (default) == the reference's order of two walks (one_pass=False) == the two private methods."""
    from ssp2vit import vit_pruning as vp
    from ssp2vit.mask_conjunction import Auto2SSPInterface
    model, eng, batches, _ = _setup()
    real = vp._engine_factory
    vp._engine_factory = lambda m, d, e: eng                     # the interface builds its engine through this seam
    try:
        a = Auto2SSPInterface(model, batches, device="cpu", batch_limit=3)
        att1, mlp1 = a.fit()
        b = Auto2SSPInterface(model, batches, device="cpu", batch_limit=3, one_pass=False)
        att2, mlp2 = b.fit()
        att3, mlp3 = b._compute_att_depth_importance(), b._compute_mlp_importance()
        assert torch.equal(att1, att2) and torch.equal(att1, att3)
        for x, y, z in zip(mlp1, mlp2, mlp3):
            assert torch.equal(x, y) and torch.equal(x, z)
        h = Auto2SSPInterface(model, batches, device="cpu", importance_mode="heuristic", batch_limit=3)
        atth, mlph = h.fit()                                     # heuristic depth scores: no search, the stage-1 pass alone
        assert atth.tolist() == [0.0, 1.0, 2.0, 1.0] and all(torch.equal(x, y) for x, y in zip(mlph, mlp1))
    finally:
        vp._engine_factory = real


# ------------------------------------------------------------------------------------------ BASELINE configs[2] dealt over 8 ranks
N_CAL_B, N_EV_B, BSZ = 32, 40, 2


def _config2_loader(rank, world):
    """configs[2] in miniature: 32 calibration + 40 evaluation batches in ONE loader of 40 batches (the first 32 are hooked, all 40
    searched), batch b a function of b alone; a rank's sharded loader yields global batches rank, rank + P, ..."""
    out = []
    for b in range(rank, N_EV_B, world):
        g = torch.Generator().manual_seed(1000 + b)
        out.append({"pixel_values": torch.randn(BSZ, 3, 32, 32, generator=g), "labels": torch.randint(0, 10, (BSZ,), generator=g)})
    return out


def _config2_run(rank, world, pg):
    from oracle.vit_modules import build_from_flat
    from ssp2vit import core
    w, _, _ = load_tiny_golden("timm")
    model = build_from_flat(w, "timm")
    eng = OracleBackedEngine(model)
    d_ints = [b.mlp.fc1.out_features for b in model.blocks]
    s, c = core.prune_pass(eng, _config2_loader(rank, world), d_ints, "pre_gelu", eng.depth, score_limit=N_CAL_B, search_limit=N_EV_B,
                           process_group=pg, sharded=True, eval_chunk_images=4 * BSZ)
    # fewer search batches than ranks (the CLI's default --eval-batches 5 on 8 GPUs): the ranks beyond the fifth own no search batch
    # and idle in the search (policy: a batch is the dealing unit — a slab pins the scores' bits — and the rank summary says so)
    s5, c5 = core.prune_pass(eng, _config2_loader(rank, world), d_ints, "pre_gelu", eng.depth, score_limit=5, search_limit=5,
                             process_group=pg, sharded=True)
    return s, c, s5, c5, dict(core.PASS_STATS)


def _config2_worker(rank, world, port, out_dir):
    for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        torch.save(_config2_run(rank, world, dist.group.WORLD), os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_config2_dealing_over_eight_gloo_ranks_equals_one_rank(tmp_path):
    """world size 8 through the real core.* host code — 32 + 40 batches dealt round-robin (5 per rank, 4 of them
    hooked), global limits, one all_gather of per-batch vectors added in global order, one int64 all_reduce: every rank holds the
    single-rank scores and counts bit for bit; and with 5 search batches on 8 ranks three ranks idle in the search (stated policy)."""
    torch.set_num_threads(1)
    ref = _config2_run(0, 1, None)
    assert ref[1][2] == N_EV_B * BSZ and ref[3][2] == 5 * BSZ
    mp.spawn(_config2_worker, args=(8, _free_port(), str(tmp_path)), nprocs=8, join=True)
    for r in range(8):
        s, c, s5, c5, stats = torch.load(os.path.join(tmp_path, f"r{r}.pt"))
        assert all(torch.equal(a, b) for a, b in zip(s, ref[0])) and c == ref[1], r
        assert all(torch.equal(a, b) for a, b in zip(s5, ref[2])) and c5 == ref[3], r
        assert stats["search_batches_owned"] == (1 if r < 5 else 0), (r, stats)
