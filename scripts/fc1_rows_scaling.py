"""Does the fc1 GEMM's over-fetch (FETCH_SIZE 5.6 x the algorithmic bytes, profiles/r04_zz_pmc_traffic.json) reach HBM?
If the re-reads of the activation panel were served by HBM, a launch whose operands fit the 256 MiB Infinity Cache would run
faster PER ROW than one ten times as large (A = 968 MB, out = 3.9 GB).  Sustained time per launch, fc1 + GELU on the persistent
256 x 256 kernel and QKV (bf16 epilogue), at 63 040 / 315 200 / 630 400 rows.
    python scripts/fc1_rows_scaling.py > profiles/r04_w_fc1_rows_scaling.txt"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "2ssp-x-vit_amd"))
from ssp2vit.engine import VitEngine
from ssp2vit.weights import synthetic_weights
dev = torch.device("cuda:0")
eng = VitEngine(synthetic_weights("vit_test_patch16_32", classes=10, seed=0, std=0.02, eps=1e-6), device=dev, max_images=8)
g = torch.Generator(device=dev).manual_seed(3)
for name, N, K, epi in (("fc1 + GELU", 3072, 768, "gelu"), ("QKV", 2304, 768, "bf16")):
    wt = torch.randn(N, K, generator=g, device=dev) * 0.03
    b = torch.randn(N, generator=g, device=dev) * 0.1
    base = None
    for M in (63040, 315200, 630400):
        a = (torch.randn(M, K, generator=g, device=dev) * 0.5).bfloat16()
        for _ in range(3): eng.linear(a, wt, b, epi, kernel="big")
        reps = max(4, 2000000 // M)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(reps): out = eng.linear(a, wt, b, epi, kernel="big")
        ev[1].record(); torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) / reps * 1e3
        base = base or us / M
        print(f"{name:11s} {M:7d} rows: {us:8.1f} us per launch = {us / M * 1e3:.3f} ns per row ({us / M / base:.3f} x the 63 040-row launch), "
              f"{2.0 * M * N * K / us * 1e-6:.0f} TF; A {M * K * 2 / 1e6:.0f} MB, out {M * N * 2 / 1e6:.0f} MB", flush=True)
        del a, out
