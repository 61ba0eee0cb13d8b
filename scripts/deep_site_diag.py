#!/usr/bin/env python3
"""Is the larger fp32-chain score error of the ViT-L/16 golden a property of the post-GELU hook site (old-HF anatomy) or of the
geometry?  Same weights, same 24 images, both anatomies: oracle (CPU, live) against the engine, per-site error distribution.
    python scripts/deep_site_diag.py [model] [images per batch]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "2ssp-x-vit_amd"))
from oracle import ref_cpu
from oracle.vit_modules import build_from_flat
from ssp2vit import core
from ssp2vit.engine import VitEngine
from ssp2vit.weights import synthetic_weights, VIT_CONFIGS
name = sys.argv[1] if len(sys.argv) > 1 else "vit_large_patch16_224"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 12
img, patch, dim, heads, inter, depth = VIT_CONFIGS[name]
torch.set_num_threads(16)
for layout, site, eps in (("timm", "pre_gelu", 1e-6), ("hf", "post_gelu", 1e-12)):
    w = synthetic_weights(name, classes=1000, seed=0, std=0.02, eps=eps, spread=4.0)
    g = torch.Generator().manual_seed(1)
    batches = [{"pixel_values": torch.randn(nb, 3, img, img, generator=g)} for _ in range(2)]
    ref = ref_cpu.ffn_activation_importance(build_from_flat(w, layout), batches, chain="fp32")
    eng = VitEngine(w, max_images=2 * nb)
    got = core.stage1_scores(eng, batches, [inter] * depth, site)
    rel = torch.stack([(got[l] - ref[l]).abs() / ref[l].abs().clamp_min(1e-6) for l in range(depth)])
    q = torch.quantile(rel.flatten().float(), torch.tensor([0.5, 0.99, 0.9999]))
    print(f"{name} {layout:4s} {site:9s} n={2 * nb}: median {q[0]:.1e} p99 {q[1]:.1e} p99.99 {q[2]:.1e} max {float(rel.max()):.2e} | per-block max "
          f"{[f'{float(rel[l].max()):.1e}' for l in range(0, depth, max(1, depth // 8))]}", flush=True)
    eng.close()
