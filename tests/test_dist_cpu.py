"""N>1 path on CPU: two gloo ranks shard the calibration / eval batches exactly as the RCCL path does on GPUs
(batch i -> rank i % P, one all_gather of per-batch score vectors, one all_reduce of int64 correct-counts).
The device engine is replaced by an oracle-backed stand-in (TEST infrastructure only; the product never
constructs it), so what is tested is the host sharding/ordering logic: results must be bit-identical to ws=1."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import PKG, ROOT, load_tiny_golden


class OracleBackedEngine:
    """Duck-types ssp2vit.engine.VitEngine on CPU through the oracle's TimmLayoutViT."""

    def __init__(self, model, max_images=64):
        self.m = model.eval()
        self.depth = len(model.blocks)
        self.device = torch.device("cpu")
        self.max_images = max_images
        self.tokens = model.pos_embed.shape[1]
        self.dim = model.pos_embed.shape[2]

    @torch.no_grad()
    def embed(self, px):
        with torch.autocast("cpu", enabled=True):
            t = self.m.patch_embed(px)
            t = torch.cat((self.m.cls_token.expand(t.shape[0], -1, -1), t), 1) + self.m.pos_embed
        return t.float().reshape(-1, self.dim).contiguous()

    @torch.no_grad()
    def layers(self, x, n, l_begin=0, l_end=None, attn_skip=None, score_site="none", score_chain="fp32", batch_scores=None, x_in=None):
        l_end = self.depth if l_end is None else l_end
        skip = set(int(i) for i in (attn_skip or []))
        t = (x if x_in is None else x_in[: x.shape[0]]).view(n, self.tokens, self.dim)
        with torch.autocast("cpu", enabled=True):
            for l in range(l_begin, l_end):
                b = self.m.blocks[l]
                if l not in skip:
                    t = t + b.attn(b.norm1(t))
                t = t + b.mlp(b.norm2(t))
        x.copy_(t.float().reshape(-1, self.dim))

    @torch.no_grad()
    def head(self, x, n, labels=None, correct=None, want_logits=False, want_pred=False):
        with torch.autocast("cpu", enabled=True):
            lg = self.m.head(self.m.norm(x.view(n, self.tokens, self.dim))[:, 0])
        if labels is not None:
            correct += (lg.argmax(-1) == labels).sum()
        return lg, lg.argmax(-1), correct

    @torch.no_grad()
    def tail(self, x, n, attn_skip=None, labels=None, correct=None, want_logits=False, want_pred=False):
        y = x.clone()
        self.layers(y, n, self.depth - 1, self.depth, attn_skip)
        return self.head(y, n, labels=labels, correct=correct)

    @torch.no_grad()
    def forward_scores(self, px, site, chain, group=0):
        from oracle import ref_cpu
        group = group if group and group > 0 else px.shape[0]
        out = []
        for s in range(0, px.shape[0], group):                   # one row of un-normalised sums per batch
            part = px[s:s + group]
            imps = ref_cpu.ffn_activation_importance(self.m, [{"pixel_values": part}], chain="fp32")
            out.append(torch.stack([t * part.shape[0] for t in imps]))
        return torch.stack(out)                                  # [groups, L, d_int]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batches():
    w, _, _ = load_tiny_golden("timm")
    g = torch.Generator().manual_seed(11)
    out = []
    for i in range(5):                                          # 5 batches over 2 ranks: ragged ownership (3 + 2)
        n = 4 if i < 4 else 3                                   # ragged last batch
        px = torch.randn(n, 3, 32, 32, generator=g)
        out.append({"pixel_values": px, "labels": torch.randint(0, 10, (n,), generator=g)})
    return w, out


def _run_all(process_group=None, sharded=False):
    from oracle.vit_modules import build_from_flat
    from ssp2vit import dist as D, vit_pruning as vp
    w, batches = _batches()
    model = build_from_flat(w, "timm")
    eng = OracleBackedEngine(model)
    if sharded:
        # the loader hands every rank ONLY its own batches (what DataLoader(batch_sampler=D.rank_batch_indices(...)) does):
        # 19 items in batches of 4 -> the same 5 batches, dealt round-robin
        rank, ws = D.world(process_group)
        px_all = torch.cat([b["pixel_values"] for b in batches]); lb_all = torch.cat([b["labels"] for b in batches])
        mine = D.rank_batch_indices(19, 4, rank, ws)
        assert [len(ix) for ix in D.rank_batch_indices(19, 4, 0, 1)] == [4, 4, 4, 4, 3]
        batches = [{"pixel_values": px_all[ix], "labels": lb_all[ix]} for ix in mine]
    kw = dict(engine=eng, process_group=process_group, sharded=sharded)
    imps = vp._compute_ffn_activation_importance(model, batches, device="cpu", **kw)
    # a batch limit is GLOBAL in both loader modes (round 5; a sharded loader used to count it per rank — ADVICE r04)
    imps3 = vp._compute_ffn_activation_importance(model, batches, device="cpu", batch_limit=3, **kw)
    counts = vp.depth_search_counts(model, batches, "cpu", None, **kw)
    top1 = vp._top1_counts(model, batches, "cpu", None, **kw)
    # ... and the same limit on the two integer loops, and ONE pass for both stages (core.prune_pass through the reference-named wrapper)
    counts3 = vp.depth_search_counts(model, batches, "cpu", 3, **kw)
    top3 = vp._top1_counts(model, batches, "cpu", 3, **kw)
    one = vp.importances_one_pass(model, batches, "cpu", 3, score_limit=None, **kw)
    return imps, imps3, counts, top1, counts3, top3, one


def _check_one_pass(res):
    """ONE pass (scores over all 5 batches, the search over the first 3) == the separate passes of the same run."""
    imps, _, _, _, counts3, _, one = res
    for a, b in zip(one[0], imps):
        assert torch.equal(a, b)
    assert tuple(one[1]) == tuple(counts3)


def _worker(rank, world, port, out_dir, sharded=False):
    for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        res = _run_all(sharded=sharded)
        torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_results_equal_single_process(tmp_path):
    torch.set_num_threads(2)
    ref = _run_all()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    _check_one_pass(ref)
    for r in range(2):
        imps, imps3, counts, top1, counts3, top3, one = torch.load(os.path.join(tmp_path, f"r{r}.pt"))
        for a, b in zip(imps, ref[0]):
            assert torch.equal(a, b)                            # bitwise: global batch order, not arrival order
        for a, b in zip(imps3, ref[1]):
            assert torch.equal(a, b)
        assert counts == ref[2] and top1 == ref[3]
        assert counts3 == ref[4] and top3 == ref[5]
        _check_one_pass((imps, imps3, counts, top1, counts3, top3, one))
    assert ref[2][2] == 19 and ref[3][1] == 19                  # every image counted exactly once


@pytest.mark.timeout(300)
@pytest.mark.parametrize("sharded", [False, True])
def test_four_rank_gloo_ragged_ownership_equals_single_process(tmp_path, sharded):
    """world_size 4 over 5 batches (ownership 2 + 1 + 1 + 1, ragged last batch of 3 images on rank 0), in both loader
    modes: every rank walks the whole loader / every rank is handed only its own batches (sharded=True: the totals
    come from one more int64 all_reduce).  Scores bit-identical to one rank, every image counted once."""
    torch.set_num_threads(1)
    ref = _run_all()
    mp.spawn(_worker, args=(4, _free_port(), str(tmp_path), sharded), nprocs=4, join=True)
    for r in range(4):
        imps, imps3, counts, top1, counts3, top3, one = torch.load(os.path.join(tmp_path, f"r{r}.pt"))
        for a, b in zip(imps, ref[0]):
            assert torch.equal(a, b)
        for a, b in zip(imps3, ref[1]):                         # limit 3 = the first three GLOBAL batches, plain and sharded loaders alike
            assert torch.equal(a, b)
        assert counts == ref[2] and top1 == ref[3]
        assert counts3 == ref[4] and top3 == ref[5] and counts3[2] == 12
        _check_one_pass((imps, imps3, counts, top1, counts3, top3, one))
    assert ref[2][2] == 19


def _starved_worker(rank, world, port, out_dir):
    """4 ranks, 2 batches: ranks 2 and 3 own NOTHING — their exchange tensors come from dist._default_device and the caller-known
    shape (no shape exchange); three steps in a row re-use ONE pair of exchange buffers."""
    for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from oracle.vit_modules import build_from_flat
    from ssp2vit import core, dist as D
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        w, batches = _batches()
        batches = batches[:2]
        model = build_from_flat(w, "timm")
        eng = OracleBackedEngine(model)
        d_ints = [b.mlp.fc1.out_features for b in model.blocks]
        assert D._default_device() == torch.device("cpu")
        outs = [core.stage1_scores(eng, batches, d_ints, "pre_gelu") for _ in range(3)]
        counts = core.depth_search_counts(eng, batches, eng.depth, batch_limit=None)
        summary = D.rank_summary({"note": "test"})
        torch.save((outs, counts, dict(D.STATS), summary), os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ranks_without_a_batch_and_buffer_reuse_over_steps(tmp_path):
    """world size 4 over TWO batches — two ranks own nothing, so `_default_device` and the `shape=` fast path
    of gather_batch_vectors run at ws > 1; three consecutive steps allocate the exchange buffers once; every rank reports the
    one-rank result bit for bit and prints a one-line summary."""
    from oracle.vit_modules import build_from_flat
    from ssp2vit import core
    torch.set_num_threads(1)
    w, batches = _batches()
    model = build_from_flat(w, "timm")
    eng = OracleBackedEngine(model)
    d_ints = [b.mlp.fc1.out_features for b in model.blocks]
    ref = core.stage1_scores(eng, batches[:2], d_ints, "pre_gelu")
    ref_counts = core.depth_search_counts(eng, batches[:2], eng.depth, batch_limit=None)
    mp.spawn(_starved_worker, args=(4, _free_port(), str(tmp_path)), nprocs=4, join=True)
    for r in range(4):
        outs, counts, stats, summary = torch.load(os.path.join(tmp_path, f"r{r}.pt"))
        for o in outs:
            for a, b in zip(o, ref):
                assert torch.equal(a, b)
        assert counts == ref_counts
        assert stats["exchanges"] == 3 and stats["buffer_allocations"] == 1, stats
        assert stats["batches_owned"] == (3 if r < 2 else 0)
        assert summary.startswith(f"[ssp2vit rank {r}/4]") and "backend=gloo" in summary and "batches_owned=" in summary


def test_exchange_tensors_follow_the_backend():
    """RCCL ("nccl") exchanges device memory on the rank's current HIP device, gloo host memory (dist.device_for_backend is pure:
    torch.device objects only, no device is touched — so the nccl branch is checked here, without a GPU)."""
    from ssp2vit import dist as D
    assert D.device_for_backend("nccl", 3) == torch.device("cuda", 3)
    assert D.device_for_backend("gloo", None) == torch.device("cpu") and D.device_for_backend("gloo", 5) == torch.device("cpu")
    with pytest.raises(RuntimeError):
        D.device_for_backend("nccl", None)
    assert "rank 0/1" in D.rank_summary() and "backend=none" in D.rank_summary()


def test_gather_is_identity_without_process_group():
    from ssp2vit import dist as D
    v = [(2, torch.ones(1, 2) * 2), (0, torch.zeros(1, 2)), (1, torch.ones(1, 2))]
    out = D.gather_batch_vectors(v, 3)
    assert [float(t[0, 0]) for t in out] == [0.0, 1.0, 2.0]
    assert D.world() == (0, 1) and D.owns(3, 0, 1)


def test_eval_chunks_cut_the_image_stream_exactly_and_counts_do_not_move():
    """core._chunks re-cuts the owned batches into forwards of exactly `chunk_images` images (batches are split and
    joined as needed); integer results must not depend on where the cuts fall."""
    from oracle.vit_modules import build_from_flat
    from ssp2vit import core
    w, batches = _batches()
    eng = OracleBackedEngine(build_from_flat(w, "timm"))
    sizes = [int(px.shape[0]) for _, px, _ in core._chunks(eng, batches, None, False, "t", 0, 1, 5)]
    assert sizes == [5, 5, 5, 4]                                # 4+4+4+4+3 images re-cut
    assert [int(px.shape[0]) for _, px, _ in core._chunks(eng, batches, 2, False, "t", 0, 1, 64)] == [8]
    ref = core.depth_search_counts(eng, batches, eng.depth, batch_limit=None, chunk_images=64)
    for c in (1, 3, 5, 7, 19):
        assert core.depth_search_counts(eng, batches, eng.depth, batch_limit=None, chunk_images=c) == ref
        assert core.top1_counts(eng, batches, chunk_images=c) == core.top1_counts(eng, batches, chunk_images=64)


def test_best_eval_chunk_prefers_full_rounds_of_the_persistent_gemm():
    from ssp2vit.core import best_eval_chunk
    n = best_eval_chunk(197, 320)
    rows = -(-n * 197 // 256)
    assert 16 <= n <= 320 and (rows * 3) % 256 <= 8 or (rows * 3) % 256 >= 240 or rows * 3 <= 256   # ~whole rounds of 256 CUs
    assert best_eval_chunk(197, 8) == 8                          # cap below the search range: the cap itself


def test_deferred_results_equal_immediate_ones():
    """defer=True only postpones the device-to-host copy and the wait (bench.py enqueues stage 2 before it waits for
    stage 1): same values."""
    from oracle.vit_modules import build_from_flat
    from ssp2vit import core
    w, batches = _batches()
    model = build_from_flat(w, "timm")
    eng = OracleBackedEngine(model)
    d_ints = [b.mlp.fc1.out_features for b in model.blocks]
    now = core.stage1_scores(eng, batches, d_ints, "pre_gelu")
    later = core.stage1_scores(eng, batches, d_ints, "pre_gelu", defer=True)
    assert callable(later)
    for a, b in zip(now, later()):
        assert torch.equal(a, b)
    assert core.depth_search_counts(eng, batches, eng.depth, batch_limit=None, defer=True)() == \
        core.depth_search_counts(eng, batches, eng.depth, batch_limit=None)
    assert [float(t.sum()) for t in core.stage1_scores(eng, [], d_ints, "pre_gelu", defer=True)()] == [0.0] * len(d_ints)


def test_layer_major_batched_search_gives_the_same_counts():
    """batch_candidates=True runs every block once for all candidates already under way (one launch of l*n images);
    per candidate the arithmetic and its order are unchanged, so the integers are."""
    from oracle.vit_modules import build_from_flat
    from ssp2vit import core
    w, batches = _batches()
    eng = OracleBackedEngine(build_from_flat(w, "timm"), max_images=1000)
    ref = core.depth_search_counts(eng, batches, eng.depth, batch_limit=None, chunk_images=64)
    assert core.depth_search_counts(eng, batches, eng.depth, batch_limit=None, chunk_images=64, batch_candidates=True) == ref
    assert core.depth_search_counts(eng, batches, eng.depth, batch_limit=None, chunk_images=7, batch_candidates=True) == ref
