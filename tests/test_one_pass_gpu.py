"""GPU tests of the ONE-PASS prune (core.prune_pass): the dense forward of the depth search's baseline carries the stage-1 hook, as
the reference's one-loader semantics allow (adaptation-for-Pures-framework/mask_conjunction.py:276-281, :327, :345, :359-362: one
`self.dl`, one `self.batch_limit`, both stages).  Everything here is BIT-EXACT: scores against core.stage1_scores, counts against
core.depth_search_counts, streams / logits against the separate launches — no tolerance appears in this file."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_tiny_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no HIP device is visible")
    return torch.device("cuda:0")


def _teacher_batches(eng, sizes, img, seed, flip_every=5):
    g = torch.Generator().manual_seed(seed)
    out = []
    for n in sizes:
        px = torch.randn(n, 3, img, img, generator=g)
        x = eng.embed(px.cuda()); eng.layers(x, n)
        lb = eng.head(x, n, want_pred=True)[1].long().cpu()
        lb[::flip_every] = (lb[::flip_every] + 1) % eng.classes          # a few wrong teachers: the baseline is not 100 %
        out.append({"pixel_values": px, "labels": lb})
    return out


@pytest.mark.parametrize("cfg,n,g,k,site", [("vit_base_patch16_224_d3", 24, 8, 3, "pre_gelu"),      # 4728 rows per slot: the 256 x 256 kernel
                                            ("vit_base_patch16_224_d3", 8, 4, 2, "post_gelu"),      # below 4096 rows in all: the 128 x 128 kernel
                                            ("vit_huge_patch14_224_d2", 16, 16, 3, "pre_gelu"),     # one slab per slot, 257 tokens, d_h 80
                                            ("vit_large_patch16_224_d2", 24, 12, 2, "post_gelu")])
def test_prefix_hook_and_slab_tail_primitives(gpu, cfg, n, g, k, site):
    """ssp2_layers_prefix / ssp2_tail_group on their own.  k streams of n images each, in slabs of g images, side by side in ONE
    launch with the stage-1 hook on the first stream only  ==  k separate launches (the first hooked): the same stream bits in
    every slot, the same per-batch score sums; and the slab-layout tail over the k slots == k contiguous tails on the streams'
    valid rows (logits, predictions, counts), with and without the last block's attention."""
    from ssp2vit import core
    from ssp2vit.engine import VitEngine
    from ssp2vit.weights import synthetic_weights
    w = synthetic_weights(cfg, classes=10, seed=9, std=0.03, eps=1e-6, bias_std=0.02, spread=4.0)
    T = (224 // (14 if "patch14" in cfg else 16)) ** 2 + 1
    eng = VitEngine(w, max_images=core.lm_capacity_images(T, k, n, g))
    assert eng.tokens == T
    L, D = eng.depth, eng.dim
    gen = torch.Generator().manual_seed(3)
    rows = core.slab_rows(T, n, g)
    px = [torch.randn(n, 3, 224, 224, generator=gen).to(gpu) for _ in range(k)]
    # k separate launches (slot 0 hooked)
    sep, sep_scores = [], None
    for s, p in enumerate(px):
        x = eng.embed(p, group=g)
        if s == 0:
            sep_scores = eng.layers(x, n, 0, L - 1, None, site, "fp32", None, g)
        else:
            eng.layers(x, n, 0, L - 1, score_group=g)
        sep.append(x.clone())
    # one launch over all slots, hook on the leading n images
    xb = torch.zeros(k * rows, D, dtype=torch.float32, device=gpu)
    for s, p in enumerate(px):
        eng.embed(p, x=xb[s * rows:(s + 1) * rows], group=g)
    bs = eng.new_scores(n // g)
    eng.layers(xb, k * n, 0, L - 1, None, site, "fp32", bs, g, score_images=n)
    valid = eng.rows(n, g) if g < n else n * T
    for s in range(k):
        assert torch.equal(xb[s * rows:s * rows + valid], sep[s][:valid]), (cfg, s)
    assert torch.equal(bs[:, : L - 1], sep_scores[:, : L - 1])
    assert float(bs[:, : L - 1].abs().sum()) > 0
    # the slab-layout tail over k slots against contiguous tails on the de-slabbed streams
    labels = torch.randint(0, 10, (n,), generator=gen).to(gpu)
    mpad = rows // (n // g)
    for skip in (None, [L - 1]):
        lg, pr, cc = eng.tail(xb, n, skip, labels=labels, want_logits=True, want_pred=True, slots=k, group=g)
        for s in range(k):
            xc = torch.cat([xb[s * rows + b * mpad: s * rows + b * mpad + g * T] for b in range(n // g)], 0).contiguous()
            l1, p1, c1 = eng.tail(xc, n, skip, labels=labels, want_logits=True, want_pred=True)
            assert torch.equal(lg[s * n:(s + 1) * n], l1) and torch.equal(pr[s * n:(s + 1) * n], p1), (cfg, s, skip)
            assert int(cc[s]) == int(c1[0])
    # whole-slab rule and argument checks
    from ssp2vit._lib import Ssp2Error
    if g < n:
        with pytest.raises(Ssp2Error):
            eng.layers(xb, k * n, 0, 1, None, site, "fp32", bs, g, score_images=n - 1)
    with pytest.raises(Ssp2Error):
        eng.layers(xb, k * n, 0, 1, None, site, "fp32", bs, g, score_images=k * n + 1)
    eng.close()


def _same(a, b):
    return len(a) == len(b) and all(x.dtype == y.dtype and torch.equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("cfg,site,precision", [("vit_tiny_patch16_224", "pre_gelu", "bf16"),
                                                ("vit_base_patch16_224_d3", "post_gelu", "bf16"),
                                                ("vit_huge_patch14_224_d2", "pre_gelu", "bf16"),
                                                ("vit_large_patch16_224_d2", "post_gelu", "fp8")])
def test_one_pass_equals_the_two_separate_passes_bit_for_bit(gpu, cfg, site, precision):
    """core.prune_pass against core.stage1_scores + core.depth_search_counts on the same loader: score tensors and counts EQUAL,
    for equal and unequal batch limits of the two stages (batches only stage 1 wants get the scores-only forward, batches only the
    search wants the plain search), a ragged last batch (its own chunk), both score chains, the layer-major search (a workspace
    for all slots) and the candidate-major one (an engine that only holds one chunk), chunk sizes of one and several batches."""
    from ssp2vit import core
    from ssp2vit.engine import VitEngine
    from ssp2vit.weights import VIT_CONFIGS, synthetic_weights
    w = synthetic_weights(cfg, classes=10, seed=4, std=0.04, eps=1e-6, bias_std=0.02, spread=4.0)
    img, patch, dim, heads, d_int, depth = VIT_CONFIGS[cfg]
    T = (img // patch) ** 2 + 1
    bsz = 24 if T * 24 >= 4096 else 8
    sizes = [bsz, bsz, bsz, bsz - 3]
    slots = depth                                                            # baseline + candidates 0 .. depth - 2
    big = VitEngine(w, max_images=core.lm_capacity_images(T, slots, 2 * bsz, bsz), precision=precision)
    small = VitEngine(w, max_images=2 * bsz, precision=precision)
    batches = _teacher_batches(big, sizes, img, seed=31)
    d_ints = [d_int] * depth
    for eng, lm in ((big, True), (small, False)):
        for chain in ("fp32", "bf16_ref"):
            for s_lim, q_lim, chunk in ((None, None, 2 * bsz), (4, 2, bsz), (2, 3, 2 * bsz), (3, 3, 2 * bsz)):
                ref_s = core.stage1_scores(eng, batches, d_ints, site, batch_limit=s_lim, score_chain=chain, chunk_images=2 * bsz)
                ref_c = core.depth_search_counts(eng, batches, depth, batch_limit=q_lim, chunk_images=chunk)
                got_s, got_c = core.prune_pass(eng, batches, d_ints, site, depth, score_limit=s_lim, search_limit=q_lim, score_chain=chain,
                                               chunk_images=2 * bsz, eval_chunk_images=chunk)
                assert _same(got_s, ref_s), (cfg, lm, chain, s_lim, q_lim)
                assert got_c == ref_c, (cfg, lm, chain, s_lim, q_lim, got_c, ref_c)
                assert 0 < ref_c[0] < ref_c[2] and len(set(ref_c[1])) > 1
        # deferred form: two callables, the same results
        fs, fc = core.prune_pass(eng, batches, d_ints, site, depth, score_limit=None, search_limit=2, defer=True, eval_chunk_images=2 * bsz)
        assert _same(fs(), core.stage1_scores(eng, batches, d_ints, site)) and fc() == core.depth_search_counts(eng, batches, depth, batch_limit=2)
    big.close(); small.close()


def test_one_pass_on_the_vit_b16_reference_golden(gpu):
    """The headline geometry on the fixture made by the REAL reference (tests/golden/vit_b16_2x32.npz: ViT-B/16, 2 x 32 images, the
    reference's one loader feeding both stages): one pass gives the very scores (both chains) and counts of the two separate passes,
    which tests/test_gpu_parity.py holds against the reference's outputs."""
    from ssp2vit import core
    from ssp2vit.engine import VitEngine
    from ssp2vit.weights import synthetic_weights
    z = dict(np.load(os.path.join(GOLDEN, "vit_b16_2x32.npz")))
    w = synthetic_weights("vit_base_patch16_224", classes=1000, seed=0, std=0.02, eps=1e-6, spread=4.0)
    g = torch.Generator().manual_seed(1)
    batches = [{"pixel_values": torch.randn(32, 3, 224, 224, generator=g), "labels": torch.from_numpy(z[f"labels.{i}"])} for i in range(2)]
    eng = VitEngine(w, max_images=core.lm_capacity_images(197, 12, 64, 32))
    d_ints = [3072] * 12
    for chain in ("fp32", "bf16_ref"):
        ref_s = core.stage1_scores(eng, batches, d_ints, "pre_gelu", score_chain=chain)
        ref_c = core.depth_search_counts(eng, batches, 12, batch_limit=5, chunk_images=64)
        got_s, got_c = core.prune_pass(eng, batches, d_ints, "pre_gelu", 12, score_limit=5, search_limit=5, score_chain=chain,
                                       eval_chunk_images=64)
        assert _same(got_s, ref_s) and got_c == ref_c, chain
    assert got_c[2] == 64 and abs(got_c[0] / 64 - float(z["top1"])) <= 1 / 64 + 1e-9
    eng.close()


def test_fit_and_the_reference_named_wrappers_take_one_pass(gpu):
    """Auto2SSPInterface.fit() (one walk) == the two private methods called one after the other (two walks), on a live module of
    the timm anatomy: the att / mlp importances are the same tensors, and `one_pass=False` keeps the reference's two-walk order."""
    from ssp2vit import vit_pruning as vp
    from ssp2vit.mask_conjunction import Auto2SSPInterface
    from ssp2vit.modules import EngineViT
    from ssp2vit.weights import synthetic_weights
    w = synthetic_weights("vit_tiny_patch16_224", classes=10, seed=2, std=0.05, eps=1e-6, bias_std=0.02)
    model = EngineViT(w).to(gpu)
    g = torch.Generator().manual_seed(8)
    batches = []
    for n in (16, 16, 16, 9):
        px = torch.randn(n, 3, 224, 224, generator=g)
        lb = model(px.to(gpu)).argmax(-1).cpu()
        lb[::4] = (lb[::4] + 1) % 10
        batches.append({"pixel_values": px, "labels": lb})
    for limit in (3, None):
        a = Auto2SSPInterface(model, batches, device="cuda", batch_limit=limit)
        att1, mlp1 = a.fit()
        b = Auto2SSPInterface(model, batches, device="cuda", batch_limit=limit, one_pass=False)
        att2, mlp2 = b._compute_att_depth_importance(), b._compute_mlp_importance()
        att3, mlp3 = b.fit()
        assert torch.equal(att1, att2) and torch.equal(att1, att3) and float(att1.max()) > 0
        assert _same(mlp1, mlp2) and _same(mlp1, mlp3)
    # a different stage-1 limit (extension): scores over all four batches, the search over two
    c = Auto2SSPInterface(model, batches, device="cuda", batch_limit=2, score_batch_limit=None)
    att, mlp = c.fit()
    assert _same(mlp, vp._compute_ffn_activation_importance(model, batches, device="cuda"))
    base, cand, tot = vp.depth_search_counts(model, batches, "cuda", 2)
    assert tot == 32 and torch.equal(att, torch.tensor([max(0.0, base / tot - cc / tot) for cc in cand], dtype=torch.float32))
    vp.release_engines()


def test_engine_pool_hands_the_next_model_of_the_same_geometry_a_reloaded_engine(gpu):
    """vit_pruning's engine pool (release_engines parks a bf16 engine; engine_for of the next module of the same geometry reloads it instead
    of building one): the second model's results are those of an engine built for it from scratch — every weight is ingested again, the
    attention flags are reset — and a module whose FFN widths differ gets a new engine."""
    from ssp2vit import core, vit_pruning as vp
    from ssp2vit.engine import VitEngine
    from ssp2vit.modules import EngineViT
    from ssp2vit.weights import synthetic_weights
    vp.release_engines(free=True)
    g = torch.Generator().manual_seed(5)
    batches = [{"pixel_values": torch.randn(8, 3, 224, 224, generator=g), "labels": torch.randint(0, 10, (8,), generator=g)} for _ in range(3)]
    w1 = synthetic_weights("vit_tiny_patch16_224", classes=10, seed=1, std=0.05, eps=1e-6, bias_std=0.02)
    w2 = synthetic_weights("vit_tiny_patch16_224", classes=10, seed=2, std=0.05, eps=1e-6, bias_std=0.02)
    m1 = EngineViT(w1).to(gpu)
    vp.importances_one_pass(m1, batches, "cuda", 3)                                      # builds m1's engine with the layer-major workspace
    e1 = vp.engine_for(m1, "cuda", 8)
    e1.drop_attention([2, 5])                                                            # ... which ends its life with two blocks bypassed
    assert any(e1.absent)
    vp.release_engines()
    assert len(vp._POOL) == 1 and vp._POOL[0] is e1
    m2 = EngineViT(w2).to(gpu)
    s2, c2 = vp.importances_one_pass(m2, batches, "cuda", 3)
    e2 = vp.engine_for(m2, "cuda", 8)
    assert e2 is e1 and not any(e2.absent) and len(vp._POOL) == 0                        # the parked engine, reloaded and reset
    fresh = VitEngine(w2, max_images=e2.max_images)
    rs, rc = core.prune_pass(fresh, batches, [768] * 12, "pre_gelu", 12, score_limit=3, search_limit=3)
    assert _same(s2, rs) and c2 == rc
    fresh.close()
    res = vp.prune_vit_mlp_width(m2, n_to_prune_per_block=[100] * 12, min_remaining=256, strategy="l1")          # other widths: the pooled geometry no longer fits
    e3 = vp.engine_for(m2, "cuda", 8)
    assert e3 is not e2 and e3.d_int == [668] * 12
    vp.release_engines(free=True)
    assert len(vp._POOL) == 0
