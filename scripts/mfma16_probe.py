"""VERDICT r04 item 2, step 1 — TIMING probe of the v_mfma_f32_16x16x32_bf16 main loop in the persistent 256 x 256 GEMM
(csrc/gemm256.hip.h, -DGEMM_MFMA16=1: same ring, same 12 ds_read_b128 and 4 LDS-DMA pieces per unit, 32 MFMAs of 16 cycles instead of
16 of 32; the epilogue's lane map is NOT adapted, so the probe binary's results are invalid and its checks are expected to fail —
only its sustained rate, power and clock are read).  Both arms are built with non-temporal output stores (-DGEMM_NT=3 = the product's
SSP2_OPT_NT_STORES default) and run interleaved, `seconds` back to back per cell, on random data.

  python3 scripts/mfma16_probe.py [seconds per run, default 2.5]  ->  JSON lines on stdout"""
import json, os, subprocess, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sustained_yardstick as sy

TOOLS = os.path.join(sy.ROOT, "2ssp-x-vit_amd", "csrc", "tools")


def run(binary, M, N, K, epi, sampler, est_us):
    n = max(50, int(sy.SECONDS / (est_us * 1e-6)))
    env = dict(os.environ, GEMM_SUSTAIN=str(n))
    t0 = time.time()
    p = subprocess.run([os.path.join(TOOLS, binary), str(M), str(N), str(K), str(epi), "20", "197", "0"], env=env, capture_output=True, text=True, timeout=300)
    t1 = time.time()
    res = {}
    for l in p.stdout.splitlines():
        if "sustained" in l:
            res["sustained_us"] = float(l.split("):")[1].split("us")[0])
            res["sustained_tflops"] = float(l.split("->")[1].split("TFLOP")[0])
            res.update(sampler.window(max(t0, t1 - sy.SECONDS * 1.1), t1))
    if "sustained_us" not in res:
        res["tail"] = (p.stdout + p.stderr)[-300:]
    return res


def main():
    sampler = sy.Sampler(); sampler.start()
    print(json.dumps({"device": torch.cuda.get_device_name(0), "seconds_per_run": sy.SECONDS}), flush=True)
    shapes = [("QKV bias only", 63040, 2304, 768, 10), ("fc1 bias only", 63040, 3072, 768, 10), ("fc1 + erf-GELU", 63040, 3072, 768, 12),
              ("fc2 + residual", 63040, 768, 3072, 11), ("out-proj + residual", 63040, 768, 768, 11),
              ("QKV bias only, 630400 rows", 630400, 2304, 768, 10), ("H/14 fc1 + erf-GELU", 82240, 5120, 1280, 12)]
    for name, M, N, K, epi in shapes:
        lib = sy.library(M, N, K, sampler)
        torch.cuda.empty_cache()
        row = {"shape": name, "M": M, "N": N, "K": K, "library (bias only)": lib}
        est = lib["sustained_us"] * 1.4
        for rnd in range(2):
            for label, binary in (("32x32x16", "gemm_bench_nt3.bin"), ("16x16x32 (probe)", "gemm_bench_nt3_m16.bin"),
                                  # second question: with 16-cycle MFMAs a DMA piece between two of them stalls the matrix pipe for ~(60 - 16) cycles
                                  # instead of ~(60 - 32): do some of the unit's four pieces belong in the LOAD phase now?  (PP_NL, see gemm256.hip.h)
                                  ("16x16x32, 1 piece in LOAD", "gemm_bench_nt3_m16_nl1.bin"), ("16x16x32, 2 pieces in LOAD", "gemm_bench_nt3_m16_nl2.bin"),
                                  ("16x16x32, 3 pieces in LOAD", "gemm_bench_nt3_m16_nl3.bin")):
                if os.path.exists(os.path.join(TOOLS, binary)):
                    row[f"{label} #{rnd}"] = run(binary, M, N, K, epi, sampler, est)
        a = min(row[f"32x32x16 #{r}"].get("sustained_us", 1e9) for r in range(2))
        b = min(row[f"16x16x32 (probe) #{r}"].get("sustained_us", 1e9) for r in range(2))
        row["speedup_16x16x32"] = round(a / b, 4)
        print(json.dumps(row), flush=True)
    sampler.stop_flag = True


if __name__ == "__main__":
    main()
