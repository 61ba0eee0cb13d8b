"""Experiment: do two engines with HALF the CUs each, on two streams, beat one engine with all CUs?
Each configuration runs the same total work: 12 blocks over 320 images (as 1 x 320 or 2 x 160)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "2ssp-x-vit_amd")]
import torch
from ssp2vit.engine import VitEngine
from ssp2vit.weights import synthetic_weights

w = synthetic_weights("vit_base_patch16_224", classes=1000, seed=0, std=0.02, eps=1e-6, spread=4.0)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
px = torch.randn(320, 3, 224, 224, generator=g, device=dev)
one = VitEngine(w, max_images=320)
a, b = VitEngine(w, max_images=160), VitEngine(w, max_images=160)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def run_one(reps):
    for _ in range(reps):
        x = one.embed(px); one.layers(x, 320)


def run_two(reps, limit):
    a.set_cu_limit(limit); b.set_cu_limit(limit)
    cur = torch.cuda.current_stream(dev)
    sa.wait_stream(cur); sb.wait_stream(cur)
    for _ in range(reps):
        with torch.cuda.stream(sa):
            x = a.embed(px[:160]); a.layers(x, 160)
        with torch.cuda.stream(sb):
            y = b.embed(px[160:]); b.layers(y, 160)
    cur.wait_stream(sa); cur.wait_stream(sb)


def timed(fn, *args):
    fn(*args); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(*args); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for rnd in range(3):
    print(f"round {rnd}: one engine, 256 CUs, 320 images x 5: {timed(run_one, 5):.2f} ms | "
          f"two engines x 160 images, 128 CUs each: {timed(run_two, 5, 128):.2f} ms | two engines, 256-WG grids each (take turns): {timed(run_two, 5, 0):.2f} ms | "
          f"two engines, 192 each: {timed(run_two, 5, 192):.2f} ms", flush=True)
