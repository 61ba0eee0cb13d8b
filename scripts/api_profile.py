import os, sys, time, torch
sys.path.insert(0, "/root/repo/2ssp-x-vit_amd"); sys.path.insert(0, "/root/repo")
from ssp2vit import vit_pruning as vp, engine as E
from ssp2vit.mask_conjunction import Auto2SSPInterface
from ssp2vit.modules import EngineViT
from ssp2vit.weights import synthetic_weights
from ssp2vit.planner import plan_from_stats, stats_from_shapes
dev = torch.device("cuda:0")
w = synthetic_weights("vit_base_patch16_224", classes=1000, seed=0, std=0.02, eps=1e-6, spread=4.0)
plan = plan_from_stats(stats_from_shapes(768, 12, 3072, 1000, 197, 16), 0.375, min_remaining=512)
g = torch.Generator(device=dev).manual_seed(1)
calib = [{"pixel_values": torch.randn(64, 3, 224, 224, generator=g, device=dev)} for _ in range(8)]
unused_evalb = [{"pixel_values": torch.randn(64, 3, 224, 224, generator=g, device=dev), "labels": torch.randint(0, 1000, (64,), generator=g, device=dev)} for _ in range(5)]
T = {}
orig_init = E.VitEngine.__init__
def timed_init(self, *a, **k):
    torch.cuda.synchronize(); t = time.perf_counter(); orig_init(self, *a, **k); torch.cuda.synchronize()
    T.setdefault("engine_init", []).append(time.perf_counter() - t)
E.VitEngine.__init__ = timed_init
orig_load = E.VitEngine._load
def timed_load(self, *a, **k):
    t = time.perf_counter(); r = orig_load(self, *a, **k); T.setdefault("VitEngine._load (ingest incl. its wait)", []).append(time.perf_counter() - t); return r
E.VitEngine._load = timed_load
orig_reload = E.VitEngine.reload
def timed_reload(self, *a, **k):
    t = time.perf_counter(); r = orig_reload(self, *a, **k); T.setdefault("VitEngine.reload (pool hit)", []).append(time.perf_counter() - t); return r
E.VitEngine.reload = timed_reload
def lap(name, t0):
    torch.cuda.synchronize(); T.setdefault(name, []).append(time.perf_counter() - t0); return time.perf_counter()
def teacher(px):
    m = EngineViT(w).to(dev)
    with torch.no_grad():
        lb = m(px).argmax(-1)
    vp.release_engines()
    return lb
for b in calib[:5]:                 # round 5: ONE loader, its first five batches carry labels (the search evaluates them)
    b["labels"] = teacher(b["pixel_values"])
orig_ef = vp.engine_for
def timed_ef(*a, **k):
    torch.cuda.synchronize(); t = time.perf_counter(); r = orig_ef(*a, **k); torch.cuda.synchronize(); T.setdefault("engine_for (from_module + VitEngine)", []).append(time.perf_counter() - t); return r
vp.engine_for = timed_ef
import ssp2vit.weights as W_
orig_fm = W_.from_module
def timed_fm(*a, **k):
    t = time.perf_counter(); r = orig_fm(*a, **k); T.setdefault("weights.from_module (host)", []).append(time.perf_counter() - t); return r
W_.from_module = timed_fm
orig_pp = vp._core.prune_pass
def timed_pp(*a, **k):
    t = time.perf_counter(); r = orig_pp(*a, **k); T.setdefault("prune_pass host (enqueue everything)", []).append(time.perf_counter() - t); return r
vp._core.prune_pass = timed_pp
LAPS = {}
PREBUILT = os.environ.get("API_PREBUILT", "1") != "0"      # 1 (as bench.py's api figure): the modules exist before the first bracket, brackets back to back
models = [EngineViT(w).to(dev) for _ in range(6)] if PREBUILT else None
for it in range(6):
    model = models[it] if PREBUILT else EngineViT(w).to(dev); torch.cuda.synchronize()
    vp._LAPS = {} if it >= 1 else None
    T_it0 = t = time.perf_counter()
    iface = Auto2SSPInterface(model, calib, device=dev, importance_mode="copy", batch_limit=5, min_remaining=512, score_batch_limit=None)
    att, mlp = iface.fit(); t = lap("fit (engine build + one pass + wait)", t)
    if it >= 2:     # the same call again on the engine that now exists (outside the prune's own total): what a fresh engine costs beyond its build
        iface.fit(); lap("fit again, cached engine (not part of total)", t)
        T_it0 += time.perf_counter() - t; t = time.perf_counter()
    res = vp.prune_vit_mlp_width(model, n_to_prune_per_block=[plan.per_block_neurons_to_prune] * 12, min_remaining=512, strategy="l1", collect_masks=True, precomputed_importance=[x.to(torch.float32) for x in mlp]); t = lap("width (host mask step + 36 gathers)", t)
    out = vp.prune_vit_attention_blocks(res["model"], sparsity=5 / 12, dataloader=None, device=dev, num_to_prune=5, show_progress=False, selected_indices=[int(i) for i in torch.argsort(att)[:5]]); t = lap("depth", t)
    T.setdefault("total", []).append(time.perf_counter() - T_it0)
    for k_, v_ in (vp._LAPS or {}).items():
        LAPS.setdefault(k_, []).append(v_)
    vp.release_engines(); del model, res, out
    if PREBUILT: models[it] = None
for k, v in LAPS.items():
    T[k] = v
for k, v in T.items():
    print(f"{k:24s}", " ".join(f"{x*1e3:7.2f}" for x in v), "ms")
