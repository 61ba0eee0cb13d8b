"""Worker of tests/test_gpu_parity.py::test_two_ranks_sharing_the_card_equal_the_single_rank_result.

Two (or three) ranks, each with its OWN VitEngine on the one card of the GPU box, meet over gloo (RCCL refuses two ranks on one
device, so the exchange tensors take the host route of `dist.device_for_backend`); every rank is handed only the batches it owns
(`dist.rank_batch_indices`) and runs the product's sharded stage 1 + depth search.  What the parent checks: every rank's scores
and counts are bit-identical to the single-rank run of the same engine code on the same card.
"""
import os
import sys

import torch


def run(engine_factory, data, rank, world, process_group, sharded):
    from ssp2vit import core, dist as D
    px, labels, batch, depth, d_int = data["px"], data["labels"], data["batch"], data["depth"], data["d_int"]
    n_cal, n_ev = data["n_calib"], data["n_eval"]
    cal_ix = D.rank_batch_indices(n_cal, batch, rank, world)
    ev_ix = D.rank_batch_indices(n_ev, batch, rank, world)
    calib = [{"pixel_values": px[ix]} for ix in cal_ix]
    evalb = [{"pixel_values": px[n_cal:][ix], "labels": labels[ix]} for ix in ev_ix]
    eng = engine_factory()
    kw = dict(process_group=process_group, sharded=sharded)
    imps = core.stage1_scores(eng, calib, [d_int] * depth, "pre_gelu", score_chain="fp32", **kw)
    imps_bf = core.stage1_scores(eng, calib, [d_int] * depth, "post_gelu", score_chain="bf16_ref", **kw)
    base, cand, total = core.depth_search_counts(eng, evalb, depth, batch_limit=None, **kw)
    top1 = core.top1_counts(eng, evalb, **kw)
    out = {"imps": [t.cpu() for t in imps], "imps_bf": [t.cpu() for t in imps_bf], "base": int(base), "cand": [int(c) for c in cand],
           "total": int(total), "top1": tuple(int(v) for v in top1)}
    if "labels_cal" in data:
        # ONE pass for both stages over the calibration loader (core.prune_pass): the search takes its first `search_limit` GLOBAL batches
        calib_l = [{"pixel_values": px[ix], "labels": data["labels_cal"][ix]} for ix in cal_ix]
        lim = int(data["search_limit"])
        s1, (b1, c1, t1) = core.prune_pass(eng, calib_l, [d_int] * depth, "pre_gelu", depth, score_limit=None, search_limit=lim, **kw)
        b2, c2, t2 = core.depth_search_counts(eng, calib_l, depth, batch_limit=lim, **kw)          # the same search, its own pass
        s3 = core.stage1_scores(eng, calib_l, [d_int] * depth, "pre_gelu", batch_limit=lim, **kw)   # a GLOBAL limit on a sharded loader
        out.update(one_imps=[t.cpu() for t in s1], one_counts=(int(b1), [int(c) for c in c1], int(t1)),
                   two_counts=(int(b2), [int(c) for c in c2], int(t2)), imps_lim=[t.cpu() for t in s3])
    out["stats"] = dict(D.STATS)
    return out


def make_engine(model, cap):
    from ssp2vit.engine import VitEngine
    from ssp2vit.weights import synthetic_weights
    return VitEngine(synthetic_weights(model, classes=1000, seed=0, std=0.02, eps=1e-6, spread=4.0), device="cuda:0", max_images=cap)


def worker(rank, world, port, data_path, out_dir):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "2ssp-x-vit_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        data = torch.load(data_path, weights_only=True)
        res = run(lambda: make_engine(data["model"], data["cap"]), data, rank, world, dist.group.WORLD, True)
        torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()
