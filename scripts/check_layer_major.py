#!/usr/bin/env python3
"""Layer-major search vs candidate-major search at full size (counts must be identical), incl. launches whose
activations exceed 4 GiB:  python scripts/check_layer_major.py [model] [n_eval]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "2ssp-x-vit_amd")):
    sys.path.insert(0, p)
import torch
from ssp2vit import core
from ssp2vit.engine import VitEngine
from ssp2vit.weights import VIT_CONFIGS, synthetic_weights

model = sys.argv[1] if len(sys.argv) > 1 else "vit_large_patch16_224"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 320
img, patch, dim, heads, d_int, L = VIT_CONFIGS[model]
w = synthetic_weights(model, classes=1000, seed=0, std=0.02, eps=1e-6, spread=4.0)
dev = torch.device("cuda", 0)
eng = VitEngine(w, device=dev, max_images=L * n)
g = torch.Generator(device=dev).manual_seed(3)
batches = []
for s in range(0, n, 64):
    px = torch.randn(min(64, n - s), 3, img, img, generator=g, device=dev)
    x = eng.embed(px); eng.layers(x, px.shape[0])
    batches.append({"pixel_values": px, "labels": eng.head(x, px.shape[0], want_pred=True)[1].long()})
a = core.depth_search_counts(eng, batches, L, batch_limit=None, chunk_images=n, batch_candidates=True)
b = core.depth_search_counts(eng, batches, L, batch_limit=None, chunk_images=n, batch_candidates=False)
rows = n * ((img // patch) ** 2 + 1)
print(f"{model}: {n} images, largest launch {(L - 1) * n} images = {(L - 1) * rows} rows, fc2 operand {(L - 1) * rows * d_int * 2 / 2**30:.1f} GiB")
print("layer-major    ", a)
print("candidate-major", b)
print("IDENTICAL" if a == b else "MISMATCH  <-- FAIL")
sys.exit(0 if a == b else 1)
