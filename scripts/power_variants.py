"""VERDICT r03 item 3 — what do the extra joules buy?  SUSTAINED rates (launches back to back for `seconds`) of the persistent 256 x 256
GEMM in its build / order variants, with the card's power and shader clock sampled beside each, next to hipBLASLt on the same
shape in the same process:

  plain            default build, N-fastest tile order
  order 8x6        super-tiles of 8 row panels x 6 column tiles (PMC: FETCH - 6 % on fc1)
  order Mx4        column groups of 4 tiles over ALL row panels (an XCD keeps 4 weight panels per round)
  nt loads         activation-panel LDS-DMA marked non-temporal (PMC round 1: QKV fetch halves)            [gemm_bench_nt2.bin]
  nt stores        output stores non-temporal                                                               [gemm_bench_nt3.bin]
  nt both                                                                                                    [gemm_bench_nt1.bin]
  four waves       one wave per SIMD, 128 x 128 per wave: two thirds of the ds_read_b128 traffic            (tools/gemm256w4.hip.h)

  python3 scripts/power_variants.py [seconds per run, default 2.5]  ->  JSON lines on stdout (MHz held, W, TF, TF per W)"""
import json, os, subprocess, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sustained_yardstick as sy

TOOLS = os.path.join(sy.ROOT, "2ssp-x-vit_amd", "csrc", "tools")


def run(binary, M, N, K, epi, group, sampler, est_us):
    n = max(50, int(sy.SECONDS / (est_us * 1e-6)))
    env = dict(os.environ, GEMM_SUSTAIN=str(n))
    t0 = time.time()
    p = subprocess.run([os.path.join(TOOLS, binary), str(M), str(N), str(K), str(epi), "20", "197", str(group)], env=env, capture_output=True, text=True, timeout=300)
    t1 = time.time()
    res = {"ok": p.returncode == 0 and "FAIL" not in p.stdout}
    for l in p.stdout.splitlines():
        if "sustained" in l:
            res["sustained_us"] = float(l.split("):")[1].split("us")[0])
            res["sustained_tflops"] = float(l.split("->")[1].split("TFLOP")[0])
            res.update(sampler.window(max(t0, t1 - sy.SECONDS * 1.1), t1))
    if res.get("power_w") and res.get("sustained_tflops"):
        res["tflops_per_kw"] = round(res["sustained_tflops"] / res["power_w"] * 1e3, 1)
    if not res["ok"]:
        res["tail"] = (p.stdout + p.stderr)[-300:]
    return res


def main():
    sampler = sy.Sampler(); sampler.start()
    print(json.dumps({"device": torch.cuda.get_device_name(0), "seconds_per_run": sy.SECONDS}), flush=True)
    shapes = [("fc1 + erf-GELU", 63040, 3072, 768, 12, 42), ("fc1 bias only", 63040, 3072, 768, 10, 40), ("QKV bias only", 63040, 2304, 768, 10, 40),
              ("fc2 + residual", 63040, 768, 3072, 11, 41), ("out-proj + residual", 63040, 768, 768, 11, 41)]
    for name, M, N, K, epi, epi_w4 in shapes:
        lib = sy.library(M, N, K, sampler)
        if lib.get("power_w"):
            lib["tflops_per_kw"] = round(lib["sustained_tflops"] / lib["power_w"] * 1e3, 1)
        torch.cuda.empty_cache()
        row = {"shape": name, "M": M, "N": N, "K": K, "library (bias only)": lib}
        est = lib["sustained_us"] * 1.4
        for label, binary, e, group in (("plain", "gemm_bench.bin", epi, 0), ("order 8x6", "gemm_bench.bin", epi, 806), ("order Mx4", "gemm_bench.bin", epi, 4),
                                        ("nt loads", "gemm_bench_nt2.bin", epi, 0), ("nt stores", "gemm_bench_nt3.bin", epi, 0), ("nt both", "gemm_bench_nt1.bin", epi, 0),
                                        ("nt loads + order Mx4", "gemm_bench_nt2.bin", epi, 4), ("four waves", "gemm_bench.bin", epi_w4, 0), ("plain (again)", "gemm_bench.bin", epi, 0)):
            if not os.path.exists(os.path.join(TOOLS, binary)):
                continue
            row[label] = run(binary, M, N, K, e, group, sampler, est)
        print(json.dumps(row), flush=True)
    sampler.stop_flag = True


if __name__ == "__main__":
    main()
