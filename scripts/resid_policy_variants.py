"""Cache policy of the fp32 residual tile in the residual epilogue (GEMM_XNT builds of tools/gemm_bench), sustained, with power / clock:
   python3 scripts/resid_policy_variants.py [seconds]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sustained_yardstick as sy
import power_variants as pv

def main():
    sampler = sy.Sampler(); sampler.start()
    for name, M, N, K in (("out-proj + residual", 63040, 768, 768), ("fc2 + residual", 63040, 768, 3072), ("out-proj + residual, layer-major", 630400, 768, 768)):
        row = {"shape": name, "M": M, "N": N, "K": K}
        est = 2.0 * M * N * K / 600e12 * 1e6
        for label, binary in (("plain", "gemm_bench.bin"), ("x loads nt", "gemm_bench_xnt1.bin"), ("x stores nt", "gemm_bench_xnt2.bin"), ("x both nt", "gemm_bench_xnt3.bin"), ("plain (again)", "gemm_bench.bin")):
            if os.path.exists(os.path.join(pv.TOOLS, binary)):
                row[label] = pv.run(binary, M, N, K, 11, 0, sampler, est)
        print(json.dumps(row), flush=True)
    sampler.stop_flag = True

if __name__ == "__main__":
    main()
