"""VERDICT r04 item 3, cheapest experiment first: the residual projections (out-proj K = 768, fc2 K = 3072) on the 128 x 128 kernel
(two independent workgroups per CU: one's fp32 x read-add-write can overlap the other's main loop) against the persistent 256 x 256
kernel, sustained, at the 63 040-row chunk and the 630 400-row layer-major launch.  tools/gemm_bench epi 1 / 11.

  python3 scripts/proj_small_tile_probe.py [seconds per run]  ->  JSON lines"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sustained_yardstick as sy
from mfma16_probe import run


def main():
    sampler = sy.Sampler(); sampler.start()
    print(json.dumps({"device": torch.cuda.get_device_name(0), "seconds_per_run": sy.SECONDS}), flush=True)
    for name, M, N, K in (("out-proj + residual", 63040, 768, 768), ("out-proj + residual, layer-major", 630400, 768, 768),
                          ("fc2 + residual", 63040, 768, 3072), ("fc2 + residual, layer-major", 630400, 768, 3072),
                          ("H/14 out-proj", 82240, 1280, 1280), ("L/16 out-proj", 63040, 1024, 1024)):
        est = 2.0 * M * N * K / 600e12 * 1e6
        row = {"shape": name, "M": M, "N": N, "K": K}
        for rnd in range(2):
            for label, epi in (("256x256 persistent", 11), ("128x128, 2 workgroups per CU", 1)):
                row[f"{label} #{rnd}"] = run("gemm_bench_nt3.bin", M, N, K, epi, sampler, est)
        a = min(row[f"256x256 persistent #{r}"].get("sustained_us", 1e9) for r in range(2))
        b = min(row[f"128x128, 2 workgroups per CU #{r}"].get("sustained_us", 1e9) for r in range(2))
        row["speedup_small_tiles"] = round(a / b, 4)
        print(json.dumps(row), flush=True)
    sampler.stop_flag = True


if __name__ == "__main__":
    main()
