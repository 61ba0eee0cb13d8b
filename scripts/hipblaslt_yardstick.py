"""Yardstick only (not a product path): what PyTorch-ROCm's library GEMM (hipBLASLt / rocBLAS behind torch.nn.functional.linear)
reaches on the step's projection shapes, bf16 in / bf16 out, with and without a bias — next to libssp2vit's persistent kernel
(tools/gemm_bench, same shapes).  The library call has no GELU / residual / scoring epilogue: those would be extra passes."""
import json, sys, torch
dev = torch.device("cuda:0")
shapes = [("QKV", 63040, 2304, 768), ("out-proj", 63040, 768, 768), ("fc1", 63040, 3072, 768), ("fc2", 63040, 768, 3072),
          ("fc1 layer-major", 315200, 3072, 768), ("H/14 QKV", 82240, 3840, 1280), ("H/14 fc1", 82240, 5120, 1280)]
out = []
for name, M, N, K in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) * 0.05).to(torch.bfloat16)
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    res = {}
    for label, fn in (("linear", lambda: torch.nn.functional.linear(a, w)), ("linear+bias", lambda: torch.nn.functional.linear(a, w, b))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
        ts.sort(); med = ts[len(ts) // 2]
        res[label] = {"median_us": round(med * 1e3, 1), "tflops": round(2.0 * M * N * K / (med * 1e-3) / 1e12, 1)}
    out.append({"shape": name, "M": M, "N": N, "K": K, **res})
    print(json.dumps(out[-1]), flush=True)
