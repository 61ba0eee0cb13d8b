#!/usr/bin/env python3
"""Headline benchmark: end-to-end 2SSP prune of ViT-B/16 @ 37.5 % (BASELINE.json configs[1]) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One STEP = one full pass of the hot path over one synthetic calibration+eval set resident in HBM:
    stage 1  per-neuron activation-L2 scores over `--calib` images (default 512, batches of 64)
    stage 2  one-shot attention-removal search: baseline + L candidates over `--eval-batches` x 64 images
    apply    plan (K=5 blocks, t=1120 neurons), mask selection (host argsort on 12x3072 scores), block selection
This is the reference CLI's "prune time" bracket (importance computation .. stage-2 selection), model/data load
and post-prune evaluation excluded.  `value` = reference-equivalent image-forwards per second, whole job:
    N_gpus * (calib + (L+1) * eval_images) * K / max-over-ranks(time of K steps)
(the reference runs L+1 FULL eval passes; the engine's prefix-cached search executes fewer block passes for the
same result — `executed_block_pass_fraction` says how many).  Multi-GPU is weak scaling: every rank holds its own
shard of `calib`/`eval` images, weights replicated; the only collectives are one all_gather of per-batch score
vectors and one all_reduce of int64 counts per step (RCCL).

Extra objects on the JSON line: `roofline` (dominant kernel = the fused fc1+GELU+L2 GEMM, HIP events around every
launch inside the timed region) and `cpu_baseline` (the CPU oracle timed on this box's host cores, rank 0, N=1).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "2ssp-x-vit_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"


def pmc_traffic():
    """HBM bytes per fc1 launch (launch-weighted over the same launch mix), from the committed rocprofv3 PMC passes
    (scripts/pmc_traffic.sh -> profiles/r<round>_<tag>_pmc_traffic.json, newest file; counters cannot be read from inside the process)."""
    import glob
    try:
        newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]
        with open(newest) as f:
            return int(json.load(f)["fc1_family"]["avg_hbm_bytes_per_launch"])
    except Exception:
        return None


def pmc_mfma():
    """Matrix-pipe busy fraction (of the cycles the part actually ran) and shader clock of the fc1 search kernel, from the
    newest committed PMC pass (scripts/pmc_mfma.py -> profiles/r<round>_<tag>_pmc_mfma.json); None when absent."""
    import glob
    try:
        newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_mfma.json")))[-1]
        with open(newest) as f:
            ks = json.load(f)["kernels"]
        k = [r for r in ks if "gemm256_bf16_kernel<2, 0>" in r["kernel"]][0]
        return {"mfma_busy_frac_of_cycles": k["mfma_util_of_cycles"], "shader_clock_ghz": k["shader_clock_ghz"],
                "source": os.path.basename(newest)}
    except Exception:
        return None


class _MetaBatch(dict):
    """Placeholder for a batch owned by another rank: only its size is ever read."""


def make_loader(own_batches, rank, world):
    """Global batch list under round-robin ownership: entry i belongs to rank i % world."""
    n_local = len(own_batches)
    out = []
    for j in range(n_local):
        for r in range(world):
            if r == rank:
                out.append(own_batches[j])
            else:
                b = own_batches[j]
                out.append(_MetaBatch(pixel_values=torch.empty(b["pixel_values"].shape, device="meta"),
                                      labels=torch.empty(b["pixel_values"].shape[0], device="meta")))
    return out


def cpu_baseline(args, weights):
    """Oracle (kind="port": bit-exact restatement of the reference, pinned by tests/test_oracle_golden.py) timed on
    the host cores over a bounded sample; converted to the metric's unit with the step's own mix of passes."""
    from oracle import ref_cpu
    from oracle.vit_modules import build_from_flat
    model = build_from_flat(weights, "timm")
    g = torch.Generator().manual_seed(123)
    n1, n2, bs = args.cpu_sample, args.cpu_sample, 32
    calib = [{"pixel_values": torch.randn(bs, 3, 224, 224, generator=g)} for _ in range(n1 // bs)]
    evalb = [{"pixel_values": torch.randn(bs, 3, 224, 224, generator=g), "labels": torch.zeros(bs, dtype=torch.int64)}
             for _ in range(n2 // bs)]
    ref_cpu.ffn_activation_importance(model, calib[:1])            # warm-up (oneDNN primitive cache)
    t0 = time.time(); ref_cpu.ffn_activation_importance(model, calib); t1 = time.time()
    ref_cpu.top1_counts(model, evalb); t2 = time.time()
    r1, r2 = n1 / (t1 - t0), n2 / (t2 - t1)
    L = int(weights["depth"])
    n_eval = args.eval_batches * args.batch
    units = args.calib + (L + 1) * n_eval
    step_s = args.calib / r1 + (L + 1) * n_eval / r2
    return {"value": round(units / step_s, 3), "unit": "image-forwards/s", "cores": torch.get_num_threads(),
            "kind": "port", "prune_time_s_extrapolated": round(step_s, 1),
            "stage1_img_per_s": round(r1, 2), "eval_img_per_s": round(r2, 2),
            "sample": f"ViT-B/16 bf16-autocast oracle: stage-1 scoring of {n1} images + top-1 eval of {n2} images "
                      f"(batch {bs}); extrapolated to {args.calib} calib + {L + 1}x{n_eval} eval image-forwards; "
                      f"model deep-copies of the reference not counted"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="vit_base_patch16_224")
    ap.add_argument("--calib", type=int, default=512, help="calibration images per GPU")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--eval-batches", type=int, default=5)
    ap.add_argument("--calib-chunk", type=int, default=512, help="images per stage-1 forward (0 = one batch; see core.stage1_scores)")
    ap.add_argument("--eval-chunk", type=int, default=0, help="images per search forward (0 = all eval images of the rank)")
    ap.add_argument("--target", type=float, default=0.375)
    ap.add_argument("--cpu-sample", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-batch-candidates", action="store_true",
                    help="candidate-major search (one launch per candidate and block) instead of the layer-major one, "
                         "in which all candidates under way run a block in ONE launch (engine workspace for "
                         "(depth-1) x eval images)")
    ap.add_argument("--host-inputs", action="store_true",
                    help="keep the batches in pinned HOST memory, as a dataloader hands them over: every step then pays "
                         "the PCIe copy of 602 KB per image (the PCIe-inclusive rate of DESIGN.md; never the default)")
    ap.add_argument("--two-streams", action="store_true",
                    help="run stage 1 and a share of the search candidates on a second HIP stream with its own engine "
                         "workspace (2-5 %% faster end to end; per-launch durations then include the share of the "
                         "machine lent to the other stream, so the roofline object is not a clean kernel figure)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    if world > 1 or ("RANK" in os.environ and os.environ.get("SSP2_FORCE_COLLECTIVES")):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)          # "nccl" is RCCL on ROCm
        pg = dist.group.WORLD

    from ssp2vit import core
    from ssp2vit.engine import VitEngine
    from ssp2vit.planner import plan_from_stats, stats_from_shapes
    from ssp2vit.weights import VIT_CONFIGS, synthetic_weights

    img, patch, dim, heads, d_int, depth = VIT_CONFIGS[args.model]
    tokens = (img // patch) ** 2 + 1
    weights = synthetic_weights(args.model, classes=1000, seed=0, std=0.02, eps=1e-6, spread=4.0)
    cap = max(args.batch, args.eval_batches * args.batch, args.calib_chunk)
    args.batch_candidates = not args.no_batch_candidates and not args.two_streams
    if args.batch_candidates:
        cap = max(cap, depth * (args.eval_chunk or args.eval_batches * args.batch))
    eng = VitEngine(weights, device=dev, max_images=cap)
    # --two-streams: stage 1 (calibration scores) and stage 2 (depth search on the dense model) are independent, and so
    # are the search candidates: with a second engine workspace (weights uploaded twice, 173 MB) on a second HIP stream
    # the memory-bound kernels of one stream (LayerNorm, attention, epilogue tails) overlap the matrix-bound kernels of
    # the other and the partial last round of a persistent GEMM is filled by the other stream's workgroups.
    eng1 = VitEngine(weights, device=dev, max_images=max(args.batch, args.calib_chunk)) if args.two_streams else eng
    side = torch.cuda.Stream(dev) if args.two_streams else None
    d_ints = [d_int] * depth
    plan = plan_from_stats(stats_from_shapes(dim, depth, d_int, 1000, tokens, patch), args.target, min_remaining=512)

    # synthetic ImageNet-shape inputs, resident in HBM before the timed region; every rank its own shard
    g = torch.Generator(device=dev).manual_seed(1 + rank)
    n_cal_b = args.calib // args.batch
    calib = [{"pixel_values": torch.randn(args.batch, 3, img, img, generator=g, device=dev)} for _ in range(n_cal_b)]
    evalb = []
    for _ in range(args.eval_batches):
        px = torch.randn(args.batch, 3, img, img, generator=g, device=dev)
        x = eng.embed(px); eng.layers(x, args.batch)
        _, pred, _ = eng.head(x, args.batch, want_pred=True)
        evalb.append({"pixel_values": px, "labels": pred.long()})     # teacher labels: dense model's own argmax
    if args.host_inputs:
        calib = [{k: v.cpu().pin_memory() for k, v in b.items()} for b in calib]
        evalb = [{k: v.cpu().pin_memory() for k, v in b.items()} for b in evalb]
    calib_loader, eval_loader = make_loader(calib, rank, world), make_loader(evalb, rank, world)
    n_eval = args.eval_batches * args.batch

    def step():
        # both stages are enqueued before the host waits for either: the a7 mask step runs on the CPU while the GPU
        # is still searching (the two stages are independent: stage 2 evaluates the dense model)
        if side is None:
            scores = core.stage1_scores(eng1, calib_loader, d_ints, "pre_gelu", score_chain="fp32", process_group=pg,
                                        chunk_images=args.calib_chunk, defer=True)
        else:
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                scores = core.stage1_scores(eng1, calib_loader, d_ints, "pre_gelu", score_chain="fp32", process_group=pg,
                                            chunk_images=args.calib_chunk, defer=True)
        # the side stream also takes a share of the search candidates behind its stage-1 launch (lead ~ the stage-1
        # work expressed in block passes of the search chunk)
        search = core.depth_search_counts(eng, eval_loader, depth, batch_limit=None, process_group=pg,
                                          chunk_images=args.eval_chunk or n_eval, defer=True,
                                          aux_engine=None if side is None else eng1, aux_stream=side,
                                          aux_lead=args.calib * depth / max(1, args.eval_chunk or n_eval),
                                          batch_candidates=args.batch_candidates and side is None)
        imps = scores()
        masks = []
        t = plan.per_block_neurons_to_prune
        for imp in imps:                                              # a7 mask step (host, 12 x 3072)
            keep, _ = torch.sort(torch.argsort(imp, descending=True)[: imp.numel() - t])
            m = torch.ones(imp.numel(), dtype=torch.int16); m[keep] = 0
            masks.append(m)
        base, cand, total = search()
        impact = torch.tensor(core.impacts_from_counts(base, cand, total), dtype=torch.float32)
        blocks = sorted(int(i) for i in torch.argsort(impact)[: plan.blocks_to_prune])   # a9 (auto_2ssp.py:857)
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)
        return imps, impact, masks, blocks

    def sync_all():
        if pg is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    prof = None
    if not args.no_roofline:
        prof = eng.profile("gemm_fc1"); prof.__enter__()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    t0 = time.perf_counter()
    s1_ms = 0.0
    for _ in range(args.steps):
        out = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if prof is not None:
        prof.__exit__(None, None, None)

    # stage-1-only rate (secondary figure, separate timed loop so the headline region stays untouched)
    sync_all(); t1 = time.perf_counter()
    core.stage1_scores(eng1, calib_loader, d_ints, "pre_gelu", score_chain="fp32", process_group=pg, chunk_images=args.calib_chunk)
    sync_all(); s1_s = time.perf_counter() - t1

    el = torch.tensor([elapsed, s1_s], dtype=torch.float64, device=dev)
    if pg is not None:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    elapsed, s1_s = float(el[0]), float(el[1])

    if rank == 0:
        units_step = args.calib + (depth + 1) * n_eval
        value = world * units_step * args.steps / elapsed
        tail = 2.0 * dim / (4 * dim + 2 * d_int + 2 * tokens)                            # cost of the CLS-only last block / a full block
        executed = args.calib * depth + n_eval * ((depth - 1) + (depth - 1) * depth // 2 + (depth + 1) * tail)   # block passes per step
        reference_equiv = args.calib * depth + n_eval * depth * (depth + 1)
        line = {
            "metric": "2ssp_prune_image_forwards_per_sec", "value": round(value, 1), "unit": "image-forwards/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic, pinned host batches copied over PCIe inside the timed region" if args.host_inputs else "synthetic",
            "config": {"workload": f"{args.model}, {args.calib} calib images/GPU, full 2SSP @ {args.target} "
                                   f"(K={plan.blocks_to_prune} blocks, t={plan.per_block_neurons_to_prune} neurons), "
                                   f"one-shot depth search over {n_eval} eval images/GPU, batch {args.batch}",
                       "weights": "random-init trunc-normal(0.02), fc1 rows log-uniform x[1/4,4], seed 0",
                       "parallelism": f"dp{world} (batches round-robin, weights replicated)"},
            "prune_time_s": round(elapsed / args.steps, 4),
            "calib_images_per_sec": round(world * args.calib / s1_s, 1),
            "executed_block_pass_fraction": round(executed / reference_equiv, 4),
            "selected_blocks": out[3], "pruned_neurons_per_block": plan.per_block_neurons_to_prune,
            "streams": 2 if args.two_streams else 1, "search": "layer-major" if args.batch_candidates else "candidate-major",
        }
        if prof is not None and prof.launches:
            # dominant kernel family: fc1 (+bias +erf-GELU; + fused activation-L2 partials in stage 1).
            # achieved = algorithmic flops (2*M*N*K summed over the recorded launches) / summed HIP-event durations.
            ach = prof.flops / (prof.total_ms * 1e-3) / 1e12
            lm = f", x 1..{depth - 1} in the layer-major search" if args.batch_candidates else ""
            line["roofline"] = {"bound": "mfma", "kernel": "fc1 GEMM family: gemm256_bf16_kernel<EPI_FC1,SCORE> persistent 256x256 (stage 1: + fused activation-L2 partials; search passes: SCORE=0) and gemm_bf16_kernel<EPI_FC1> 128x128 (CLS tail); bias + erf-GELU fused",
                                "achieved": round(ach, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(ach / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": pmc_traffic(), "pmc": pmc_mfma(),
                                "launches": prof.launches, "avg_launch_us": round(prof.total_ms * 1e3 / prof.launches, 2),
                                "flops_per_launch_avg": prof.flops / prof.launches,
                                "shapes": f"[{eng.rows(min(args.calib, args.calib_chunk or args.batch), args.batch)} | {(args.eval_chunk or n_eval) * tokens} | {n_eval}] x {d_int} x {dim} (stage-1 launch in {args.batch}-image slabs | search chunk{lm} | CLS tail)"}
        if not args.no_roofline:
            # the HBM-bound kernel of the path on its own: the standalone activation-L2 accumulate (a2) over one
            # layer's activation of one calibration batch, outside the timed region (in the step it is fused into the
            # fc1 epilogue and reads nothing from HBM).  Algorithmic bytes = n*N*d_int*2 read once.
            act = torch.randn(args.batch, tokens, d_int, device=dev, dtype=torch.float32).to(torch.bfloat16)
            for _ in range(3):
                eng.act_l2_accum(act)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for _ in range(reps):
                eng.act_l2_accum(act)
            e1.record(); e1.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            gbps = act.numel() * 2 / (us * 1e-6) / 1e9
            line["act_l2_kernel"] = {"bound": "hbm", "achieved": round(gbps, 1), "peak": 8000.0, "unit": "GB/s",
                                     "frac": round(gbps / 8000.0, 4), "bytes_per_launch": act.numel() * 2,
                                     "avg_launch_us": round(us, 2),
                                     "kernel": "act_l2_norms_kernel<bf16> + score_colsum_kernel (standalone a2; 2 launches per call)"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, weights)
        print(json.dumps(line), flush=True)
    if pg is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
