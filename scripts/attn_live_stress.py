"""Race screen of the compile-time-padding attention instantiations (attn64_persist_kernel<7, 1>, attn80_persist_kernel<9, 1>): the same
two-block forward ROUNDS times, every residual stream compared with the first — the kernels carry hand-counted waits, so a missing one
shows up as run-to-run differences — and once against SSP2_OPT_ATTN_LIVE = 0.   python3 scripts/attn_live_stress.py [rounds]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "2ssp-x-vit_amd"))
from ssp2vit.engine import VitEngine
from ssp2vit.weights import synthetic_weights
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for cfg, n in (("vit_small_patch16_224_d2", 384), ("vit_huge_patch14_224_d2", 192)):
    w = synthetic_weights(cfg, classes=10, seed=3, std=0.05, eps=1e-6, bias_std=0.02)
    eng = VitEngine(w, max_images=n)
    px = torch.randn(n, 3, 224, 224, generator=torch.Generator().manual_seed(4)).cuda()
    x0 = eng.embed(px)
    first, bad = None, 0
    for r in range(rounds):
        x = x0.clone()
        eng.layers(x, n)
        if first is None:
            first = x.clone()
        elif not torch.equal(x, first):
            bad += 1
    eng.set_option("attn_live", 0)
    x = x0.clone(); eng.layers(x, n)
    same = torch.equal(x, first)
    torch.cuda.synchronize()
    print(f"{cfg}: {n} images x {eng.heads} heads, {rounds} launches of two blocks: {bad} differ from the first; general instantiation equal: {same}")
    eng.close()
    assert bad == 0 and same
