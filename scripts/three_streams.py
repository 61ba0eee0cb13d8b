#!/usr/bin/env python3
"""Experiment: the depth search over 1 / 2 / 3 HIP streams (one engine workspace each), ViT-B/16, 320 images."""
import os, sys, time, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "2ssp-x-vit_amd")):
    sys.path.insert(0, p)
import torch
from ssp2vit.engine import VitEngine
from ssp2vit.weights import VIT_CONFIGS, synthetic_weights
model = "vit_base_patch16_224"
img, patch, dim, heads, d_int, L = VIT_CONFIGS[model]
w = synthetic_weights(model, classes=1000, seed=0, std=0.02, eps=1e-6, spread=4.0)
dev = torch.device("cuda", 0)
n = 320
engs = [VitEngine(w, device=dev, max_images=n) for _ in range(3)]
streams = [torch.cuda.current_stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
g = torch.Generator(device=dev).manual_seed(1)
px = torch.randn(n, 3, img, img, generator=g, device=dev)
labels = torch.zeros(n, dtype=torch.int64, device=dev)
def search(k):
    counts = torch.zeros(L + 1, dtype=torch.int64, device=dev)
    e = engs[0]
    x = e.embed(px); cache = {}
    for l in range(L - 1):
        cache[l] = x.clone(); e.layers(x, n, l, l + 1)
    e.tail(x, n, None, labels=labels, correct=counts[L:L + 1])
    loads = [float(L - 1)] + [0.0] * (k - 1)
    assign = {}
    for c in range(L):
        j = min(range(k), key=lambda i: loads[i]); assign[c] = j; loads[j] += L - 1 - c + 0.2
    for s in streams[1:k]:
        s.wait_stream(streams[0])
    for c in range(L):
        j = assign[c]
        with (torch.cuda.stream(streams[j]) if j else contextlib.nullcontext()):
            if c == L - 1:
                engs[j].tail(x, n, [c], labels=labels, correct=counts[c:c + 1]); continue
            xc = cache.pop(c); xc.record_stream(streams[j])
            engs[j].layers(xc, n, c, L - 1, [c]); engs[j].tail(xc, n, [c], labels=labels, correct=counts[c:c + 1])
    for s in streams[1:k]:
        streams[0].wait_stream(s)
    return counts
for k in (1, 2, 3, 1, 2, 3):
    search(k); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        c = search(k)
    torch.cuda.synchronize()
    print(f"{k} stream(s): {1e3 * (time.perf_counter() - t0) / 3:.2f} ms per search, counts {c.tolist()[:4]}...")
