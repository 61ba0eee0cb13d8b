"""Deferred residual (SSP2_OPT_DEFER_RESID / SSP2_DEFER_RESID=1) against the direct epilogue: bits and time.
  python scripts/dg_check.py [reps]
1. ssp2_linear_bf16 with the residual epilogue on the shapes of the models (rows full / ragged, K = 768 .. 5120): x must be identical.
2. a ViT-B/16 engine: logits and one-shot depth-search counts with the option off / on must be identical.
3. sustained time of both forms per shape (median of `reps` launches in a row, HIP events)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "2ssp-x-vit_amd"))
from ssp2vit import core  # noqa: E402
from ssp2vit.engine import VitEngine  # noqa: E402
from ssp2vit.weights import synthetic_weights  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    w = synthetic_weights("vit_base_patch16_224", classes=1000, seed=0, std=0.02, eps=1e-6, spread=4.0)
    eng = VitEngine(w, device=dev, max_images=12 * 64)
    g = torch.Generator(device=dev).manual_seed(3)
    ok = True
    shapes = [(63040, 768, 768), (63040, 768, 3072), (63040 - 77, 768, 768), (4096 + 300, 1024, 1024), (20000, 1024, 4096),
              (20000, 1280, 1280), (20000, 1280, 5120), (4096, 768, 256), (300000, 768, 768)]
    for M, N, K in shapes:
        a = (torch.randn(M, K, generator=g, device=dev) * 0.5).bfloat16()
        wt = torch.randn(N, K, generator=g, device=dev) * 0.03
        b = torch.randn(N, generator=g, device=dev) * 0.1
        x0 = torch.randn(M + 5, N, generator=g, device=dev)
        outs, times = [], []
        for flag in ("0", "1"):
            os.environ["SSP2_DEFER_RESID"] = flag
            x = x0.clone()
            eng.linear(a, wt, b, "resid", x=x, kernel="big")
            torch.cuda.synchronize()
            outs.append(x)
            xs = x0.clone()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            for _ in range(3):
                eng.linear(a, wt, b, "resid", x=xs, kernel="big")
            ev[0].record()
            for _ in range(reps):
                eng.linear(a, wt, b, "resid", x=xs, kernel="big")
            ev[1].record(); torch.cuda.synchronize()
            times.append(ev[0].elapsed_time(ev[1]) / reps * 1e3)
        same = torch.equal(outs[0], outs[1])
        ok &= same
        nbad = int((outs[0] != outs[1]).sum())
        print(f"resid {M} x {N} x {K}: identical {same} (differing elements {nbad}); direct {times[0]:.1f} us, deferred {times[1]:.1f} us "
              f"({(times[1] / times[0] - 1) * 100:+.1f} %)", flush=True)
    os.environ["SSP2_DEFER_RESID"] = "0"
    # engine: logits + search counts
    px = torch.randn(128, 3, 224, 224, generator=g, device=dev)
    res = []
    for flag in (0, 1):
        eng.set_option("defer_resid", flag)
        lg = eng.forward_logits(px)
        labels = lg.argmax(-1)
        batches = [{"pixel_values": px[i:i + 64], "labels": labels[i:i + 64]} for i in range(0, 128, 64)]
        counts = core.depth_search_counts(eng, batches, eng.depth, batch_limit=None)
        sc = core.stage1_scores(eng, batches, [3072] * 12, "pre_gelu", score_chain="fp32")
        res.append((lg, counts, sc))
    same = torch.equal(res[0][0], res[1][0]) and res[0][1] == res[1][1] and all(torch.equal(a, b) for a, b in zip(res[0][2], res[1][2]))
    ok &= same
    print("engine: logits, search counts and stage-1 scores identical:", same, res[1][1])
    print("DG CHECK", "OK" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
