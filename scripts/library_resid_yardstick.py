"""What does AMD's library pay for a FUSED residual add?  torch.addmm(x, a, w^T) = hipBLASLt with beta = 1 — only available with a residual of the
operands' dtype (bf16: HALF the x bytes this path moves, its residual stream is fp32) — next to the bias-only kernel and to the unfused form a
caller would need for an fp32 stream (library GEMM with bf16 output, then x += y as a separate fp32 pass).  Sustained, out-proj and fc2 shapes.
    python3 scripts/library_resid_yardstick.py"""
import json, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sustained_yardstick as sy
from library_gelu_yardstick import sustained

def main():
    dev = torch.device("cuda:0")
    sampler = sy.Sampler(); sampler.start()
    for name, M, N, K in (("out-proj", 63040, 768, 768), ("fc2", 63040, 768, 3072), ("out-proj layer-major", 630400, 768, 768)):
        g = torch.Generator(device=dev).manual_seed(1)
        a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        w = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) * 0.02).to(torch.bfloat16)
        b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
        wt = w.t()
        xb = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        xf = torch.zeros(M, N, device=dev, dtype=torch.float32)
        row = {"shape": name, "M": M, "N": N, "K": K}
        row["library bias only (bf16 out)"] = sustained(lambda: torch.nn.functional.linear(a, w, b), sampler)
        row["library beta = 1, bf16 residual in place"] = sustained(lambda: torch.addmm(xb, a, wt, out=xb), sampler)
        row["library bias only, then fp32 x += y pass"] = sustained(lambda: xf.add_(torch.nn.functional.linear(a, w, b)), sampler)
        flops = 2.0 * M * N * K
        for k, v in row.items():
            if isinstance(v, dict) and "us" in v:
                v["tflops"] = round(flops / (v["us"] * 1e-6) / 1e12, 1)
        print(json.dumps(row), flush=True)
        del a, w, xb, xf
        torch.cuda.empty_cache()
    sampler.stop_flag = True

if __name__ == "__main__":
    main()
