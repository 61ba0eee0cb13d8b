"""Sustained time of the residual projections on the persistent GEMM, direct (SSP2_DEFER_RESID=0) and deferred (=1), for the
library SSP2_LIB_VARIANT names (default: the product's).  Timing only; scripts/dg_check.py compares the bits."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "2ssp-x-vit_amd"))
from ssp2vit.engine import VitEngine
from ssp2vit.weights import synthetic_weights
dev = torch.device("cuda:0")
eng = VitEngine(synthetic_weights("vit_test_patch16_32", classes=10, seed=0, std=0.02, eps=1e-6), device=dev, max_images=8)
g = torch.Generator(device=dev).manual_seed(3)
flags = sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1", "0", "1"]
for M, N, K in [(63040, 768, 3072), (63040, 768, 768), (20000, 1280, 5120), (630400, 768, 768)]:
    a = (torch.randn(M, K, generator=g, device=dev) * 0.5).bfloat16()
    wt = torch.randn(N, K, generator=g, device=dev) * 0.03
    b = torch.randn(N, generator=g, device=dev) * 0.1
    xs = torch.randn(M, N, generator=g, device=dev)
    for flag in flags:
        os.environ["SSP2_DEFER_RESID"] = flag
        for _ in range(3): eng.linear(a, wt, b, "resid", x=xs, kernel="big")
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(30): eng.linear(a, wt, b, "resid", x=xs, kernel="big")
        ev[1].record(); torch.cuda.synchronize()
        print(f"{os.environ.get('SSP2_LIB_VARIANT','default')} {M}x{N}x{K} defer={flag}: {ev[0].elapsed_time(ev[1]) / 30 * 1e3:.1f} us", flush=True)
    del a, wt, b, xs
