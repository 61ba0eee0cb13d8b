"""What does AMD's library pay for a FUSED GELU?  hipBLASLt's bias + GELU epilogue (tanh approximation — not the exact erf form the reference
computes, so not usable here; a yardstick only) through torch._addmm_activation(use_gelu=True), next to its bias-only kernel and to an unfused
bias-only GEMM followed by torch's exact-erf GELU pass, sustained, on the fc1 shape of the 320-image search chunk.
    python3 scripts/library_gelu_yardstick.py"""
import json, sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sustained_yardstick as sy

def sustained(fn, sampler, seconds=2.0):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); e1.synchronize()
    n = max(50, int(seconds / (e0.elapsed_time(e1) / 20 * 1e-3)))
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time(); a0.record()
    for _ in range(n): fn()                                   # (outputs are dropped at once: 387 MB each)
    a1.record(); a1.synchronize(); t1 = time.time()
    return {"us": round(a0.elapsed_time(a1) / n * 1e3, 1), **sampler.window(t0, t1)}

def main():
    dev = torch.device("cuda:0")
    sampler = sy.Sampler(); sampler.start()
    for name, M, N, K in (("fc1", 63040, 3072, 768), ("H/14 fc1", 82240, 5120, 1280)):
        g = torch.Generator(device=dev).manual_seed(1)
        a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        w = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) * 0.05).to(torch.bfloat16)
        b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
        wt = w.t()
        row = {"shape": name, "M": M, "N": N, "K": K}
        row["library bias only"] = sustained(lambda: torch.nn.functional.linear(a, w, b), sampler)
        try:
            row["library bias + GELU epilogue (tanh form)"] = sustained(lambda: torch._addmm_activation(b, a, wt, use_gelu=True), sampler)
        except Exception as exc:
            row["library bias + GELU epilogue (tanh form)"] = {"error": repr(exc)[:200]}
        row["library bias only, then torch exact-erf GELU pass"] = sustained(lambda: torch.nn.functional.gelu(torch.nn.functional.linear(a, w, b)), sampler)
        flops = 2.0 * M * N * K
        for k, v in row.items():
            if isinstance(v, dict) and "us" in v:
                v["tflops"] = round(flops / (v["us"] * 1e-6) / 1e12, 1)
        print(json.dumps(row), flush=True)
    sampler.stop_flag = True

if __name__ == "__main__":
    main()
