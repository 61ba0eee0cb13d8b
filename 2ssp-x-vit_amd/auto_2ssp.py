#!/usr/bin/env python3
"""Auto 2SSP for ViT with a single TARGET sparsity — MI355X-native driver with the flag surface of the reference CLI
(/root/reference/adaptation-for-Pures-framework/auto_2ssp.py:1037-1088) for everything on the prune path.

Flow = reference `run()` (:628-1034): baseline metrics -> plan -> importances (stage-1 activation scores and stage-2
attention-removal impacts, both on the GPU engine) -> stage-1 width prune -> metrics -> stage-2 attention bypass ->
metrics -> artifacts (`ffn_prune_masks.json`, `<prefix>_scores.json` / `_masks.json`, `report-<id>.json|.md`).

Out of scope here (network): HF / timm checkpoint and CIFAR loading, fine-tuning, adapters.  Models are seeded
random-init architectures (`--model vit_base_patch16_224` ...) and data is synthetic ImageNet-shape batches with
teacher labels; a caller with a real module + dataloader uses `ssp2vit.vit_pruning` directly.
Extra flags: --sparsity_rate (-2 = the main-table sweep 0.25/0.375/0.5 of main.py:152-157), --seed, --synthetic-*.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))

import torch  # noqa: E402

from ssp2vit import artifacts, vit_pruning as vp  # noqa: E402
from ssp2vit.mask_conjunction import Auto2SSPInterface  # noqa: E402
from ssp2vit.modules import EngineViT  # noqa: E402
from ssp2vit.weights import VIT_CONFIGS, synthetic_weights  # noqa: E402


def measure_latency(model, device, warmup=3, iters=10, img_size=224) -> float:
    """Reference :196-221 — bs=1 forward, seconds per image."""
    x = torch.randn(1, 3, img_size, img_size, device=device)
    for _ in range(warmup):
        model(x)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters):
        model(x)
    torch.cuda.synchronize()
    return (time.time() - t0) / iters


def synthetic_loaders(model, img, batch, n_calib, n_eval_batches, seed, device):
    g = torch.Generator(device=device).manual_seed(seed)
    calib = [{"pixel_values": torch.randn(batch, 3, img, img, generator=g, device=device)} for _ in range(max(1, n_calib // batch))]
    test = []
    for _ in range(n_eval_batches):
        px = torch.randn(batch, 3, img, img, generator=g, device=device)
        test.append({"pixel_values": px, "labels": model(px).argmax(-1)})      # teacher labels
    for b in calib:
        b["labels"] = model(b["pixel_values"]).argmax(-1)
    return test, calib


def local_loaders(args, img, device, pg):
    """--calib-data / --eval-data: (test_loader, cal_loader) over uint8 arrays on disk (ssp2vit.local_data).  Every rank sees the
    same seeded order and yields only the batches it owns (dist.rank_batch_indices); the hot loops pick `sharded` up from the loader."""
    from ssp2vit.local_data import Uint8BatchLoader, load_uint8_dataset
    rank, world = (torch.distributed.get_rank(), torch.distributed.get_world_size()) if pg is not None else (0, 1)
    mean = tuple(float(v) for v in args.data_mean.split(","))
    std = tuple(float(v) for v in args.data_std.split(","))
    ev_x, ev_y = load_uint8_dataset(args.eval_data, args.eval_labels)
    ca_x, ca_y = load_uint8_dataset(args.calib_data, args.calib_labels)
    common = dict(out_size=img, mean=mean, std=std, device=device, rank=rank, world=world)
    test_loader = Uint8BatchLoader(ev_x, ev_y, 64, shuffle=False, random_flip=False, seed=args.seed, **common)               # reference :346
    cal_loader = Uint8BatchLoader(ca_x, ca_y, 64, shuffle=True, random_flip=True, seed=args.seed, **common)                  # reference :347
    return test_loader, cal_loader


def run_one(args, target, run_id):
    device = "cuda"
    if not torch.cuda.is_available():
        raise SystemExit("auto_2ssp needs an MI355X: the product path has no CPU fallback")
    pg = _process_group()
    rank0 = pg is None or torch.distributed.get_rank() == 0
    vp.DEFAULT_PRECISION = "bf16"          # (module-level switch: a previous fp8 run in this process must not leak into this one's set-up)
    if args.weights:
        # a LOCAL checkpoint (reference :636-667 loads its model with from_pretrained / timm.create_model; the network
        # fetches stay out): .safetensors / .pth state dict or an HF save_pretrained directory, any of the three key layouts
        from ssp2vit.weights import load_checkpoint
        flat = load_checkpoint(args.weights, heads=args.heads)
        name = f"{os.path.basename(os.path.normpath(args.weights))} ({flat['layout']} layout, dim {flat['dim']}, depth {flat['depth']})"
        img = int(flat["img"])
    else:
        name = args.model if args.model in VIT_CONFIGS else "vit_base_patch16_224"
        img = VIT_CONFIGS[name][0]
        flat = synthetic_weights(name, classes=args.num_classes, seed=args.seed, std=0.02, spread=4.0)
    model = EngineViT(flat)
    dataset_desc = "synthetic"
    if args.calib_data or args.eval_data:
        # LOCAL image bytes (uint8 HWC arrays on disk) through the GPU input pipeline, with the reference's loader semantics
        # (load_cifar, reference :268-350: test batch 64 unshuffled / test transform; calibration batch 64 shuffled / train transform
        # with the random flip) — the network fetch in front of them stays out of scope
        if not (args.calib_data and args.eval_data):
            raise SystemExit("--calib-data and --eval-data go together (the search evaluates, the report needs a test set)")
        test_loader, cal_loader = local_loaders(args, img, device, pg)
        dataset_desc = f"local uint8: calib {args.calib_data} ({cal_loader.n} images), eval {args.eval_data} ({test_loader.n} images)"
        if int(test_loader.labels.max()) >= int(flat["classes"]) or int(cal_loader.labels.max()) >= int(flat["classes"]):
            raise SystemExit(f"labels reach {int(max(test_loader.labels.max(), cal_loader.labels.max()))} but the model has {int(flat['classes'])} classes (--num-classes)")
    else:
        test_loader, cal_loader = synthetic_loaders(model, img, args.batch_size, args.synthetic_calib, max(args.eval_batches, 1),
                                                    args.seed + 1, device)
    if rank0:
        bs = getattr(cal_loader, "batch_size", args.batch_size)            # local data: the reference's loaders fix 64 (:346-347)
        print(f"[INFO] Using device: {device}; model={name}; data={dataset_desc}; calib batches={len(cal_loader)} eval batches={len(test_loader)} "
              f"of {bs} (this rank's); ranks={1 if pg is None else torch.distributed.get_world_size()}")

    if args.precision == "fp8":
        # opt-in e4m3 arithmetic for the large projections (BASELINE configs[4]); --fp8-calibrate measures the attention outputs of the
        # first calibration batches and fixes the e4m3 hand-off scale of every block before anything else runs
        vp.DEFAULT_PRECISION = "fp8"
        if args.fp8_calibrate:
            from ssp2vit.core import _pixels_to_device
            # EVERY rank measures the same images — global batch 0 of the calibration order, whoever owns it — so that all ranks set
            # the same e4m3 scales and a batch's scores do not depend on the rank that ran it; the peek leaves the loader's epoch alone
            peek = cal_loader.peek_global(1) if hasattr(cal_loader, "peek_global") else list(cal_loader)[:1]
            px = [_pixels_to_device(b, torch.device(device, torch.cuda.current_device())) for b in peek]
            scales = vp.calibrate_fp8(model, torch.cat(px, 0), device)
            if rank0:
                print(f"[FP8] calibrated attention hand-off scales: {scales}")
    else:
        vp.DEFAULT_PRECISION = "bf16"
    params_before = vp.count_total_params(model)
    latency_baseline = measure_latency(model, device, img_size=img)
    maxb = args.baseline_eval_batches if args.baseline_eval_batches is not None else args.eval_batches
    acc_baseline = vp.evaluate_top1(model, test_loader, device, max_batches=maxb, process_group=pg) if (maxb is None or maxb > 0) else None
    print(f"[STEP] Baseline computed: params={params_before}, latency_ms={round(latency_baseline * 1000, 2)}, acc={acc_baseline}")

    plan = None
    if args.stage == "both":
        plan = vp.plan_2ssp_allocation(model, target, min_remaining=args.min_remaining, forced_blocks=args.force_depth_blocks)
        print(f"[PLAN] target={plan.target_sparsity:.3f}, blocks_to_prune={plan.blocks_to_prune}, "
              f"per_block_neurons_to_prune={plan.per_block_neurons_to_prune}")
    B = len(model.blocks)

    t_prune0 = time.time()
    imp_mode = "heuristic" if args.stage == "s1" else args.depth_importance
    iface = Auto2SSPInterface(model, cal_loader, device=device, importance_mode=imp_mode, batch_limit=args.eval_batches,
                              min_remaining=args.min_remaining, score_chain=args.score_chain, process_group=pg)
    if args.stage == "both" and args.s1_importance == "act" and not args.two_pass:
        # the reference calls the two private methods one after the other (:774-775), each walking cal_loader[:eval_batches] with its own
        # dense forward; fit() takes both from ONE walk whose dense forward serves the hook and the search's baseline alike
        att_imp, mlp_imp = iface.fit()
    else:
        mlp_imp = iface._compute_mlp_importance() if (args.stage in ("both", "s1") and args.s1_importance == "act") else None
        att_imp = iface._compute_att_depth_importance() if args.stage in ("both", "s2") else None

    ffn_masks = ffn_indices = mask_parity = None
    inter_before = [int(d) for d in vp._get_hidden_and_inter_sizes(model)[1]]
    if args.stage in ("both", "s1"):
        if args.stage == "both":
            n_prune = [plan.per_block_neurons_to_prune] * B
        else:
            if args.s1_sparsity is None:
                raise ValueError("When --stage s1, you must provide --s1-sparsity (fraction of FFN params per block to remove).")
            _, inter = vp._get_hidden_and_inter_sizes(model)
            n_prune = [max(0, min(int(round(args.s1_sparsity * d)), max(0, d - args.min_remaining))) for d in inter]
        res = vp.prune_vit_mlp_width(model, n_to_prune_per_block=n_prune, min_remaining=args.min_remaining, strategy="l1",
                                     collect_masks=True,
                                     precomputed_importance=[x.to(torch.float32) for x in mlp_imp] if mlp_imp is not None else None)
        ffn_masks = res["ffn_prune_masks"]
        ffn_indices = res["ffn_pruned_indices"]
        mask_parity = res.get("mask_parity")
    params_s1 = vp.count_total_params(model) if args.stage != "s2" else params_before
    t_s1 = time.time()
    latency_s1 = measure_latency(model, device, img_size=img)
    acc_s1 = vp.evaluate_top1(model, test_loader, device, max_batches=args.eval_batches, process_group=pg)
    print(f"[STAGE-1 DONE] params={params_s1}, latency_ms={round(latency_s1 * 1000, 2)}, acc={acc_s1}")

    pruned_indices = []
    t_s2a = time.time()
    if args.stage in ("both", "s2"):
        if args.stage == "both":
            k = args.force_depth_blocks if args.force_depth_blocks is not None else plan.blocks_to_prune
            frac = plan.blocks_to_prune / max(1, B)
        else:
            if args.s2_sparsity is None:
                raise ValueError("When --stage s2, you must provide --s2-sparsity (fraction of Attention params / blocks to remove).")
            k = max(0, min(B - 1, int(round(B * args.s2_sparsity))))
            frac = k / max(1, B)
        sel = [int(i) for i in torch.argsort(att_imp)[:k]]                    # reference :857
        res = vp.prune_vit_attention_blocks(model, sparsity=frac, dataloader=test_loader, device=device,
                                            batch_limit=args.eval_batches, importance_mode=args.depth_importance,
                                            show_progress=False, num_to_prune=k, selected_indices=sel, process_group=pg)
        pruned_indices = res["pruned_indices"]
    t_prune1 = time.time()
    params_s2 = vp.count_total_params(model)
    latency_s2 = measure_latency(model, device, img_size=img)
    acc_s2 = vp.evaluate_top1(model, test_loader, device, max_batches=args.eval_batches, process_group=pg)
    print(f"[STAGE-2 DONE] params={params_s2}, latency_ms={round(latency_s2 * 1000, 2)}, acc={acc_s2}, pruned_blocks={pruned_indices}")

    if not rank0:                      # every rank holds the same scores, masks and selection; rank 0 writes the artefacts
        vp.release_engines()
        return None
    s1 = vp.compute_actual_sparsity(params_before, params_s1)
    s2 = vp.compute_actual_sparsity(params_s1, params_s2)
    st = vp.compute_actual_sparsity(params_before, params_s2)
    out_root = Path(args.output_dir)
    art_dir = out_root / "artifacts" / run_id
    art_dir.mkdir(parents=True, exist_ok=True)
    arte = {"pruned_block_indices": pruned_indices}
    if args.artifact_format == "v1":
        # the older CLI's files (experiments/vit_pruning/auto_2ssp.py:769-829): format_version 1 masks + indices, attention indices, "b:j" importances
        arte.update(artifacts.save_v1_artifacts(str(art_dir), mlp_imp=mlp_imp, ffn_masks=ffn_masks, ffn_indices=ffn_indices,
                                                pruned_block_indices=pruned_indices, min_remaining=args.min_remaining,
                                                s1_sparsity=args.s1_sparsity, block_inter_sizes=inter_before))
    elif ffn_masks is not None:
        arte["ffn_prune_masks_path"] = artifacts.save_ffn_prune_masks(str(art_dir / "ffn_prune_masks.json"), ffn_masks)
    if args.fw_export_prefix:
        artifacts.build_framework_exports(args.fw_export_prefix, B, model.config.hidden_size, model.config.num_attention_heads,
                                          mlp_imp, att_imp, ffn_masks, pruned_indices)
        print(f"[FRAMEWORK] Exported to: {args.fw_export_prefix}_scores.json and {args.fw_export_prefix}_masks.json")
    if args.save_pruned_model:
        from ssp2vit import export
        if args.save_format == "timm":                                               # reference :881-894 (timm / SRP branch)
            arte["pruned_model_dir"] = export.save_timm_state_dict(model, args.pruned_output_dir, run_id)
        else:                                                                        # reference :415-424, :896-901 (HF directory)
            arte["pruned_model_dir"] = export.save_pruned_model_and_processor(model, None, Path(args.pruned_output_dir), run_id)
    pct = lambda a, b: round((a / max(1e-12, b) - 1) * 100, 1)
    drop = lambda a, b: round(((a - b) / max(1e-12, a)) * 100, 2) if (a is not None and b is not None) else None
    report = {
        "config": {"model": name, "target_sparsity": target, "stage": args.stage, "s1_sparsity": args.s1_sparsity,
                   "s2_sparsity": args.s2_sparsity, "freeze_backbone": False, "replace_classifier": False, "use_adapter": False,
                   "adapter_reduction": None, "eval_batches": args.eval_batches, "min_remaining": args.min_remaining,
                   "cifar_load": False, "dataset": dataset_desc, "weights": args.weights, "precision": args.precision,
                   "gpus": 1 if pg is None else torch.distributed.get_world_size()},
        "metrics": {
            "params_before_stage1": params_before, "params_after_stage1": params_s1, "params_after_stage2": params_s2,
            "params_before_stage1_millions": round(params_before / 1e6, 2), "params_after_stage1_millions": round(params_s1 / 1e6, 2),
            "params_after_stage2_millions": round(params_s2 / 1e6, 2),
            "stage1_reduction_percent": round(s1 * 100, 1), "stage2_reduction_percent": round(s2 * 100, 1),
            "total_reduction_percent": round(st * 100, 1),
            "latency_baseline_ms": round(latency_baseline * 1000, 2), "latency_stage1_ms": round(latency_s1 * 1000, 2),
            "latency_stage2_ms": round(latency_s2 * 1000, 2), "latency_stage1_change_percent": pct(latency_s1, latency_baseline),
            "latency_stage2_change_percent": pct(latency_s2, latency_s1), "latency_total_change_percent": pct(latency_s2, latency_baseline),
            "acc_baseline": round(acc_baseline, 4) if acc_baseline is not None else None,
            "acc_stage1": round(acc_s1, 4), "acc_stage2": round(acc_s2, 4),
            "acc_drop_stage1_percent": drop(acc_baseline, acc_s1), "acc_drop_stage2_percent": drop(acc_s1, acc_s2),
            "acc_total_drop_percent": drop(acc_baseline, acc_s2),
            # additions of this build: the prune-time bracket of main.py:164-198 (importance .. stage-2 apply)
            "prune_time_s": round((t_s1 - t_prune0) + (t_prune1 - t_s2a), 4),
        },
        "artifacts": arte,
    }
    if plan is not None:
        report["plan"] = dict(plan.__dict__)
    if mask_parity is not None:          # per-block cut margin / tie band of the masks just cut (ssp2vit/mask_parity.py)
        report["mask_parity"] = dict(mask_parity, score_chain=args.score_chain)
    saved = vp.save_report(report, str(out_root / "reports"), run_id=run_id)
    print("[SUMMARY]")
    print(json.dumps(report["metrics"], indent=2))
    print(f"[INFO] Report saved to: {saved['json']} and {saved['md']}")
    vp.release_engines()
    return report


def _process_group():
    return torch.distributed.group.WORLD if (torch.distributed.is_available() and torch.distributed.is_initialized()) else None


def launcher_plan(argv, environ):
    """`--gpus N` (N > 1) without a torch.distributed environment: the N-rank job to START — one process per GPU over RCCL,
    the calibration / evaluation batches dealt round-robin (ssp2vit/dist.py) — or None when this process is a rank itself."""
    import socket
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1 or "RANK" in environ or "WORLD_SIZE" in environ:
        return None
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return {"cmd": [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                    "--master-port", port, str(Path(__file__).resolve()), *argv], "env": env, "n": n}


def build_argparser():
    p = argparse.ArgumentParser(description="Auto 2SSP for ViT with single TARGET sparsity (MI355X-native engine).")
    p.add_argument("--model", type=str, default="vit_base_patch16_224", help=f"architecture: one of {sorted(VIT_CONFIGS)}")
    p.add_argument("--target", type=float, required=False)
    p.add_argument("--sparsity_rate", type=float, default=None, help="alias of --target; -2 sweeps 0.25/0.375/0.5")
    p.add_argument("--stage", type=str, default="both", choices=["both", "s1", "s2"])
    p.add_argument("--s1-sparsity", type=float, default=None)
    p.add_argument("--s2-sparsity", type=float, default=None)
    p.add_argument("--min-remaining", type=int, default=512)
    p.add_argument("--eval-batches", type=int, default=5)
    p.add_argument("--baseline-eval-batches", type=int, default=None)
    p.add_argument("--s1-importance", type=str, default="act", choices=["act", "l1"])
    p.add_argument("--depth-importance", type=str, default="copy", choices=["copy", "heuristic"])
    p.add_argument("--force-depth-blocks", type=int, default=None)
    p.add_argument("--force-copy-eval", action="store_true", help="accepted for compatibility (no MPS here)")
    p.add_argument("--save-pruned-model", action="store_true")
    p.add_argument("--pruned-output-dir", type=str, default=str(HERE / "pruned_models"))
    p.add_argument("--save-format", type=str, default="timm", choices=["timm", "hf"],
                   help="timm: state_dict file (the reference's --use-srp-checkpoint branch); hf: save_pretrained-style directory")
    p.add_argument("--fw-export-prefix", type=str, default=None)
    p.add_argument("--artifact-format", type=str, default="current", choices=["current", "v1"],
                   help="v1: the older CLI's artifact files (experiments/vit_pruning/auto_2ssp.py:769-829): ffn_prune_masks.json with "
                        "format_version 1 / masks / indices, attention_pruned_indices.json, iterative_vit_b16_ffn_importances.json")
    p.add_argument("--output-dir", type=str, default=str(HERE / "runs"), help="reports/ and artifacts/ are created below it")
    # synthetic stand-ins for the network-loaded model/data of the reference
    p.add_argument("--weights", type=str, default=None,
                   help="LOCAL checkpoint instead of random init: .safetensors / .pth state dict or an HF save_pretrained directory "
                        "(timm, transformers<5 or transformers>=5 key layout; width-pruned checkpoints load as they are)")
    p.add_argument("--heads", type=int, default=None, help="attention heads, when the checkpoint carries no config.json")
    p.add_argument("--gpus", type=int, default=1, help="N > 1: one process per GPU over RCCL, batches dealt round-robin (starts the ranks itself)")
    p.add_argument("--calib-data", type=str, default=None,
                   help="LOCAL calibration images: .npz (images/x/data + labels/y) or .npy (+ --calib-labels or <stem>_labels.npy), uint8 [n,H,W,3]; "
                        "shuffled by --seed, random flip, batch 64 (the reference's cal_loader), resized / normalised on the GPU")
    p.add_argument("--eval-data", type=str, default=None, help="LOCAL evaluation images, same formats; batch 64, not shuffled (the reference's test_loader)")
    p.add_argument("--calib-labels", type=str, default=None)
    p.add_argument("--eval-labels", type=str, default=None)
    p.add_argument("--data-mean", type=str, default="0.5,0.5,0.5", help="Normalize mean (the HF ViT processor's image_mean)")
    p.add_argument("--data-std", type=str, default="0.5,0.5,0.5")
    p.add_argument("--num-classes", type=int, default=1000)
    p.add_argument("--batch-size", type=int, default=64)
    p.add_argument("--synthetic-calib", type=int, default=512)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--score-chain", type=str, default="fp32", choices=["fp32", "bf16_ref"])
    p.add_argument("--precision", type=str, default="bf16", choices=["bf16", "fp8"],
                   help="fp8: QKV / out-proj / fc1 / fc2 of launches with >= 4096 token rows on e4m3 MFMA operands (opt-in tolerance mode, configs[4])")
    p.add_argument("--two-pass", action="store_true", help="compute the two importances in two separate walks over the calibration loader, as the reference "
                                                           "does (default: one walk whose dense forward serves both; a reshuffling loader then feeds both stages the same order)")
    p.add_argument("--fp8-calibrate", action="store_true", help="with --precision fp8: measure the attention outputs of the first calibration images and set the e4m3 hand-off scales")
    # accepted and ignored (data / fine-tuning flags of the reference that need the network)
    for flag in ("--load-cifar", "--do-finetune", "--freeze-backbone", "--replace-classifier", "--use-adapter", "--save-adapter",
                 "--use-srp-checkpoint"):
        p.add_argument(flag, action="store_true", help=argparse.SUPPRESS)
    for flag in ("--dataset", "--load-adapter", "--srp-model-type", "--srp-dataset", "--srp-index-csv", "--srp-models-dir",
                 "--srp-checkpoint-npz"):
        p.add_argument(flag, type=str, default=None, help=argparse.SUPPRESS)
    for flag, typ in (("--calib-per-class", int), ("--cifar-train-pct", float), ("--cifar-test-pct", float), ("--ft-epochs", int),
                      ("--ft-lr", float), ("--adapter-reduction", int), ("--srp-res", int)):
        p.add_argument(flag, type=typ, default=None, help=argparse.SUPPRESS)
    return p


def main(argv=None):
    args = build_argparser().parse_args(argv)
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("SSP2_REHEARSE_ONE_CARD") == "1":
            # a correctness rehearsal of the N-rank job on a one-GPU box (as bench.py's): every rank computes on cuda:0 and the exchanges go over
            # gloo (RCCL refuses two ranks on one device; the collectives are the same calls, their tensors take dist.device_for_backend's host route)
            torch.cuda.set_device(0)
            torch.distributed.init_process_group("gloo")
        else:
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))   # RCCL
    rate = args.sparsity_rate if args.sparsity_rate is not None else args.target
    targets = [0.25, 0.375, 0.5] if rate == -2 else [rate]
    if args.stage == "both" and any(t is None for t in targets):
        raise SystemExit("--target (or --sparsity_rate) is required when --stage both")
    stamp = time.strftime("%Y%m%d-%H%M%S")
    return [run_one(args, t, f"{stamp}-s{t}" if len(targets) > 1 else stamp) for t in targets]


if __name__ == "__main__":
    _plan = launcher_plan(sys.argv[1:], os.environ)
    if _plan is not None:                       # the parent only starts the ranks (children, no exec) and passes their exit code on
        import subprocess
        sys.exit(subprocess.call(_plan["cmd"], env=_plan["env"]))
    main()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
