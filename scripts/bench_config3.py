#!/usr/bin/env python3
"""BASELINE configs[3] as a measured line (not the driver's bench): ViT-L/16, stage-2 ITERATIVE attention-block removal, the full depth
sweep K = 1 .. 23 (semantics of the reference's LLM loop, src/utilities.py:446-505, with top-1 as the metric; `search="iterative"` of
ssp2vit.vit_pruning.prune_vit_attention_blocks), evaluation batches of this rank resident in HBM.  A step = one whole sweep: round r
evaluates the model with the r blocks removed so far (the round's baseline) and every remaining candidate on top of it — 322 evaluation
passes of n_eval images in reference terms, executed layer-major with the prefix cache.
    python scripts/bench_config3.py [--eval-batches 5] [--steps 2]        (one GPU; under torch.distributed.run: one rank per GPU, counts all-reduced)"""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "2ssp-x-vit_amd"))
from ssp2vit import core
from ssp2vit.engine import VitEngine
from ssp2vit.weights import synthetic_weights, VIT_CONFIGS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="vit_large_patch16_224")
    ap.add_argument("--eval-batches", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    args = ap.parse_args()
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        pg = dist.group.WORLD
    img, patch, dim, heads, d_int, depth = VIT_CONFIGS[args.model]
    w = synthetic_weights(args.model, classes=1000, seed=0, std=0.02, eps=1e-6, spread=4.0)
    n_eval = args.eval_batches * args.batch
    eng = VitEngine(w, device=dev, max_images=depth * n_eval)
    g = torch.Generator(device=dev).manual_seed(1 + rank)
    evalb = []
    for _ in range(args.eval_batches):
        px = torch.randn(args.batch, 3, img, img, generator=g, device=dev)
        x = eng.embed(px); eng.layers(x, args.batch)
        evalb.append({"pixel_values": px, "labels": eng.head(x, args.batch, want_pred=True)[1].long()})

    def sweep():
        removed, trace = [], []
        for r in range(depth - 1):
            rest = [i for i in range(depth) if i not in removed]
            base, cc, tot = core.depth_search_counts(eng, evalb, depth, batch_limit=None, removed=removed, candidates=rest, process_group=pg,
                                                     chunk_images=n_eval, batch_candidates=True, sharded=True)
            best = max(rest, key=lambda i: (cc[i], -i))                      # greedy: the block whose removal hurts top-1 least; ties -> lower index
            removed.append(best); trace.append(cc[best])
        return removed, trace, tot

    for _ in range(args.warmup):
        sweep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        removed, trace, tot = sweep()
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if pg is not None:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    if rank == 0:
        passes = sum(depth - r + 1 for r in range(depth - 1))                # per round: the baseline + every remaining candidate
        s = float(el[0]) / args.steps
        print(json.dumps({"metric": "2ssp_iterative_depth_sweep_image_forwards_per_sec", "config": "BASELINE.json configs[3]", "model": args.model,
                          "n_gpus": world, "eval_images_per_gpu": n_eval, "rounds": depth - 1, "reference_equivalent_eval_passes": passes,
                          "s_per_sweep": round(s, 4), "value": round(passes * n_eval * world / s, 1), "unit": "image-forwards/s",
                          "removal_order": removed, "correct_after_each_round_of": tot, "correct_after_each_round": trace,
                          "dtype": "bf16", "data": "synthetic", "search": "layer-major, prefix-cached, CLS-only last block"}), flush=True)
    if pg is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
