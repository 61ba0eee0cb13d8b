#!/usr/bin/env python3
"""Where the HOST time of a bench step goes (enqueue vs wait): python scripts/host_timing.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "2ssp-x-vit_amd")):
    sys.path.insert(0, p)
import torch
from ssp2vit import core
from ssp2vit.engine import VitEngine
from ssp2vit.weights import VIT_CONFIGS, synthetic_weights

model = "vit_base_patch16_224"
img, patch, dim, heads, d_int, depth = VIT_CONFIGS[model]
w = synthetic_weights(model, classes=1000, seed=0, std=0.02, eps=1e-6, spread=4.0)
dev = torch.device("cuda", 0)
eng = VitEngine(w, device=dev, max_images=512)
g = torch.Generator(device=dev).manual_seed(1)
calib = [{"pixel_values": torch.randn(64, 3, img, img, generator=g, device=dev)} for _ in range(8)]
evalb = [{"pixel_values": torch.randn(64, 3, img, img, generator=g, device=dev), "labels": torch.zeros(64, dtype=torch.int64, device=dev)} for _ in range(5)]
d_ints = [d_int] * depth
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = core.stage1_scores(eng, calib, d_ints, "pre_gelu", chunk_images=512, defer=True); t1 = time.perf_counter()
    q = core.depth_search_counts(eng, evalb, depth, batch_limit=None, chunk_images=320, defer=True); t2 = time.perf_counter()
    imps = s(); t3 = time.perf_counter()
    for imp in imps:
        keep, _ = torch.sort(torch.argsort(imp, descending=True)[: imp.numel() - 1120])
        m = torch.ones(imp.numel(), dtype=torch.int16); m[keep] = 0
    t4 = time.perf_counter()
    q(); t5 = time.perf_counter()
    print(f"iter {it}: enqueue stage1 {1e3*(t1-t0):.1f} ms | enqueue stage2 {1e3*(t2-t1):.1f} | wait stage1 {1e3*(t3-t2):.1f} | mask step {1e3*(t4-t3):.1f} | wait stage2 {1e3*(t5-t4):.1f} | total {1e3*(t5-t0):.1f}")
