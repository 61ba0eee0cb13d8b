#!/usr/bin/env python3
"""Where the fp32-chain score error of the full-depth goldens sits (GPU box): per block the largest relative errors with the score's
rank inside its block, and percentiles — is the tail a few tiny-score neurons or a uniform shift?
    python scripts/deep_golden_diag.py vit_large_patch16_224 vit_l16_2x12 hf"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "2ssp-x-vit_amd"))
from ssp2vit import core
from ssp2vit.engine import VitEngine
from ssp2vit.weights import synthetic_weights, VIT_CONFIGS
name, tag, layout = sys.argv[1:4]
z = dict(np.load(os.path.join(ROOT, "tests", "golden", tag + ".npz")))
img, patch, dim, heads, inter, depth = VIT_CONFIGS[name]
nb = int(z["n_per_batch"])
w = synthetic_weights(name, classes=1000, seed=0, std=0.02, eps=1e-6 if layout == "timm" else 1e-12, spread=4.0)
g = torch.Generator().manual_seed(1)
batches = [{"pixel_values": torch.randn(nb, 3, img, img, generator=g)} for _ in range(2)]
eng = VitEngine(w, max_images=2 * nb)
site = "pre_gelu" if layout == "timm" else "post_gelu"
for opt in (4096, 1 << 30):
    eng.set_option("big_tile_min_rows", opt)
    got = core.stage1_scores(eng, batches, [inter] * depth, site)
    print(f"--- big_tile_min_rows = {opt}")
    for l in range(depth):
        ref = torch.from_numpy(z[f"oracle_fp32.{l}"])
        rel = (got[l] - ref).abs() / ref.abs().clamp_min(1e-6)
        signed = ((got[l] - ref) / ref.abs().clamp_min(1e-6))
        q = torch.quantile(rel, torch.tensor([0.5, 0.99, 0.999, 0.9999]))
        top = torch.topk(rel, 3)
        rank = [int((ref < ref[i]).sum()) for i in top.indices]
        print(f"block {l:2d}: median {q[0]:.1e} p99 {q[1]:.1e} p99.9 {q[2]:.1e} p99.99 {q[3]:.1e} max {float(rel.max()):.2e} | mean signed {float(signed.mean()):+.1e} | "
              f"top-3 at score rank {rank} of {inter} (scores {[round(float(ref[i]), 4) for i in top.indices]}, block median {float(ref.median()):.4f})")
