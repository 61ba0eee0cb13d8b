#!/usr/bin/env python3
"""Headline benchmark: end-to-end 2SSP prune of ViT-B/16 @ 37.5 % (BASELINE.json configs[1]) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One STEP = one full pass of the hot path over one synthetic calibration+eval set resident in HBM:
    stage 1  per-neuron activation-L2 scores over `--calib` images (default 512, batches of 64)
    stage 2  one-shot attention-removal search: baseline + L candidates over `--eval-batches` x 64 images
    select   plan (K=5 blocks, t=1120 neurons), mask step (host argsort on 12x3072 scores), block selection
    apply    the kept FFN neurons are gathered and the chosen attention blocks dropped INTO A SECOND ENGINE
             (engine.apply_into: the dense engine stays intact for the next step, as the reference's deep copy does)
This is the reference CLI's "prune time" bracket (importance computation .. just after stage-2 apply), model /
data load and post-prune evaluation excluded.  `value` = reference-equivalent image-forwards per second, whole job:
    N_gpus * (calib + (L+1) * eval_images) * K / max-over-ranks(time of K steps)
(the reference runs L+1 FULL eval passes; the engine's prefix-cached search executes fewer block passes for the
same result — `executed_block_pass_fraction` says how many).  Multi-GPU, default: weak scaling, every rank holds its
own shard of `calib`/`eval` images (its loader yields only its own batches), weights replicated; the only collectives
are one all_gather of per-batch score vectors and int64 count all_reduces per step (RCCL).

`--config 2` = BASELINE.json configs[2]: 2048 calibration + 2560 evaluation images IN TOTAL, dealt to the ranks batch by
batch (strong scaling), ONE stage-1 pass and ONE search, then plan + masks + selection + apply for each of the targets
0.25 / 0.375 / 0.5; the line carries the collectives' device time per step.

Extra objects on the JSON line: `roofline` (dominant kernel = the fc1 GEMM family, HIP events around every launch inside
the timed region; `traffic` / `pmc` only from PMC summaries recorded at THIS source hash), `act_l2_kernel` (the HBM-bound
standalone hook-body kernel over a rotation of activations larger than the Infinity Cache), `api` (the same prune
through the reference-named Python API on a live module, engine build included) and `cpu_baseline` (the CPU oracle
timed on this box's host cores, rank 0, N=1).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "2ssp-x-vit_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


# ------------------------------------------------------------------------------------------------ self-launch (N > 1)
def launcher_plan(argv, environ):
    """`python bench.py --gpus N` with N > 1 and no torch.distributed environment: the command line and environment of
    the N-rank job this process should START (one process per GPU over RCCL), or None when this process is itself a rank
    (RANK / WORLD_SIZE set: started by torch.distributed.run, e.g. by the driver) or N == 1.  Pure function of its
    arguments — the parent never imports torch, so it never initialises HIP before (or after) the children exist."""
    import socket
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1 or "RANK" in environ or "WORLD_SIZE" in environ:
        return None
    port = environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    env = dict(environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__), *argv]
    return {"cmd": cmd, "env": env, "n": n}


def run_launcher(plan) -> int:
    """Start the N ranks as CHILD processes (no exec), relay their output, return the job's exit code; non-zero when any
    rank failed or rank 0 printed no result line."""
    import subprocess
    proc = subprocess.Popen(plan["cmd"], env=plan["env"], stdout=subprocess.PIPE, text=True, bufsize=1)
    got_line = False
    for line in proc.stdout:
        sys.stdout.write(line); sys.stdout.flush()
        if line.lstrip().startswith("{") and '"metric"' in line:
            got_line = True
    rc = proc.wait()
    if rc == 0 and not got_line:
        print("bench.py launcher: the ranks exited cleanly but rank 0 printed no result line", file=sys.stderr)
        return 1
    return rc


if __name__ == "__main__":
    _plan = launcher_plan(sys.argv[1:], os.environ)
    if _plan is not None:
        sys.exit(run_launcher(_plan))

import torch  # noqa: E402

BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
FP8_MFMA_PEAK_TFLOPS = 5000.0    # dense, same table ("Peak FP8 MFMA")


def _newest_summary(pattern, model="vit_base_patch16_224", precision="bf16"):
    """Newest committed PMC summary of that kind recorded (a) at the running library's source hash — counters cannot be
    read from inside the process, and a summary of another kernel revision is not evidence — and (b) on THIS model and
    precision (round 2 keyed on the hash alone, so the ViT-L / ViT-H / fp8 lines carried ViT-B/16's busy fraction).
    Summaries without the two fields predate them and were all taken on ViT-B/16 in bf16."""
    import glob
    from ssp2vit import _lib
    want = _lib._source_hash()
    stale = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            with open(path) as f:
                js = json.load(f)
        except Exception:
            continue
        if js.get("model", "vit_base_patch16_224") != model or js.get("precision", "bf16") != precision:
            continue
        if js.get("lib_source_hash") == want:
            return js, os.path.basename(path), None
        stale = stale or os.path.basename(path)
    return None, None, stale


def pmc_traffic(model="vit_base_patch16_224", precision="bf16"):
    js, name, stale = _newest_summary("r*_pmc_traffic*.json", model, precision)
    if js is None:
        return None, {"stale_summary_ignored": stale} if stale else None
    return int(js["fc1_family"]["avg_hbm_bytes_per_launch"]), {"source": name}


def pmc_mfma(model="vit_base_patch16_224", precision="bf16"):
    js, name, stale = _newest_summary("r*_pmc_mfma*.json", model, precision)
    if js is None:
        return {"stale_summary_ignored": stale} if stale else None
    try:
        k = [r for r in js["kernels"] if "gemm256_bf16_kernel<2, 0" in r["kernel"]][0]
        return {"mfma_busy_frac_of_cycles": k["mfma_util_of_cycles"], "shader_clock_ghz": k["shader_clock_ghz"], "source": name,
                "model": model, "precision": precision}
    except Exception:
        return None


def library_yardstick(shape="fc1"):
    """The committed SUSTAINED library measurement of the dominant family's shape (scripts/sustained_yardstick.py: hipBLASLt,
    bias-only epilogue, 2.5 s of back-to-back launches, the card's power sampled beside it) — context for `roofline.frac`: on
    random data these GEMMs run at the card's power cap, and what the vendor's own kernel sustains there is the practical
    ceiling.  A recorded number of ANOTHER box and run, never part of this run's measurement; None if the file is not there."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sustained_library_yardstick.jsonl")), reverse=True):
        try:
            for line in open(path):
                if not line.startswith('{"shape"'):
                    continue
                row = json.loads(line)
                if row.get("shape") == shape:
                    lib = row["library(linear+bias)"]
                    return {"hipblaslt_bias_only_sustained_tflops": lib["sustained_tflops"], "power_w": lib["power_w"], "sclk_mhz": lib["sclk_mhz"],
                            "shape": [row["M"], row["N"], row["K"]], "source": os.path.basename(path),
                            "note": "recorded on another box; the library call has no GELU / scoring epilogue"}
        except Exception:
            continue
    return None


FAMILIES = (("fc1", ("gemm_fc1",), "fc1 (+bias +erf-GELU; stage 1: + fused activation-L2 partials)"),
            ("resid", ("gemm_fc2", "gemm_proj"), "attention out-projection + fc2 (+bias, fp32 residual read-add-write)"),
            ("qkv", ("gemm_qkv",), "QKV projection (+bias)"),
            ("attention", ("attn",), "softmax(QK^T / sqrt(d_h)) V per (image, head)"),
            ("layernorm", ("ln",), "standalone LayerNorm (fp32 row in, bf16 / e4m3 row out)"),
            ("other", ("gemm_patch", "gemm_head", "score_finish", "act_l2", "other"), "patch embed, head, score finish, argmax, im2col"))


def roofline_by_family(by_class, precision):
    """One extra, UNTIMED step with HIP events around every launch (VitEngine.profile("all")): per kernel family the
    device time, its share, algorithmic flops / bytes and the fraction of the roofline that bounds it.  The GEMM families
    on e4m3 operands are priced against the fp8 peak, the out-projection inside `resid` stays bf16 (priced as fp8 there:
    a lower bound on the fraction)."""
    tot = sum(v["ms"] for v in by_class.values()) or 1.0
    peak = BF16_MFMA_PEAK_TFLOPS if precision == "bf16" else FP8_MFMA_PEAK_TFLOPS
    out = {}
    for name, classes, what in FAMILIES:
        ms = sum(by_class.get(c, {}).get("ms", 0.0) for c in classes)
        if ms <= 0:
            continue
        fl = sum(by_class.get(c, {}).get("flops", 0.0) for c in classes)
        by = sum(by_class.get(c, {}).get("bytes", 0.0) for c in classes)
        n = sum(by_class.get(c, {}).get("launches", 0) for c in classes)
        row = {"kernels": what, "launches": int(n), "ms": round(ms, 3), "share_of_kernel_time": round(ms / tot, 4)}
        if name in ("fc1", "resid", "qkv"):
            tf = fl / (ms * 1e-3) / 1e12
            row.update(bound="mfma", flops=fl, achieved=round(tf, 1), peak=peak, unit="TFLOP/s", frac=round(tf / peak, 4))
        elif name == "attention":
            tf = fl / (ms * 1e-3) / 1e12
            gb = by / (ms * 1e-3) / 1e9
            row.update(bound="hbm", bytes=by, achieved=round(gb, 1), peak=8000.0, unit="GB/s", frac=round(gb / 8000.0, 4),
                       flops=fl, tflops=round(tf, 1), frac_of_bf16_mfma_peak=round(tf / BF16_MFMA_PEAK_TFLOPS, 4))
        elif name == "layernorm":
            gb = by / (ms * 1e-3) / 1e9
            row.update(bound="hbm", bytes=by, achieved=round(gb, 1), peak=8000.0, unit="GB/s", frac=round(gb / 8000.0, 4))
        out[name] = row
    return out


class _PowerSampler:
    """Card power (uW) and shader clock (Hz) from sysfs every 20 ms, the card picked by the PCI address HIP reports (sysfs lists
    every card of the host).  Missing files (no sysfs access) simply give None."""

    def __init__(self, dev):
        import glob
        import threading
        cards = sorted(glob.glob("/sys/class/drm/card*/device"))
        try:
            pr = torch.cuda.get_device_properties(dev)
            addr = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            cards = [c for c in cards if os.path.realpath(c).endswith(addr)] or cards
        except Exception:
            pass
        self.pw, self.ck = [], []
        for c in cards:
            for name in ("power1_average", "power1_input"):
                self.pw += glob.glob(os.path.join(c, "hwmon", "hwmon*", name))
            self.ck += glob.glob(os.path.join(c, "hwmon", "hwmon*", "freq1_input"))
        self.rows, self.stop = [], False
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    @staticmethod
    def _read(files):
        best = None
        for f in files:
            try:
                v = float(open(f).read().strip())
                best = v if best is None else max(best, v)
            except (OSError, ValueError):
                pass
        return best

    def _run(self):
        while not self.stop:
            self.rows.append((time.time(), self._read(self.pw), self._read(self.ck)))
            time.sleep(0.02)

    def window(self, t0, t1):
        rows = [r for r in self.rows if t0 + 0.3 * (t1 - t0) <= r[0] <= t1]          # the settled part of the run
        pw = [r[1] for r in rows if r[1] is not None]
        ck = [r[2] for r in rows if r[2] is not None]
        return {"power_w": round(sum(pw) / len(pw) / 1e6, 1) if pw else None, "sclk_mhz": round(sum(ck) / len(ck) / 1e6, 1) if ck else None}


def sustained_families(eng, dev, rows, dim, d_int, seconds=0.8):
    """what each GEMM family of the step sustains ON ITS OWN — its shape launched back to back for `seconds`
    through the C ABI's projection operator (ssp2_linear_bf16: the very kernels of the forward), with the card's power and shader
    clock sampled beside it — and, in the same process on the same box, what AMD's library (hipBLASLt behind
    torch.nn.functional.linear, bias-only epilogue: it has no fused GELU / fp32-residual form) sustains on that shape: the realistic
    ceiling of a bf16 GEMM of this shape on this card under its power cap.  Outside the timed region; bf16 only."""
    import ctypes as C
    sampler = _PowerSampler(dev)
    g = torch.Generator(device=dev).manual_seed(7)
    out = {}
    shapes = (("qkv", rows, 3 * dim, dim, 0, "bias, bf16 out"), ("proj", rows, dim, dim, 1, "bias + fp32 residual"),
              ("fc1", rows, d_int, dim, 2, "bias + erf-GELU"), ("fc2", rows, dim, d_int, 1, "bias + fp32 residual"))
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for name, M, N, K, epi, what in shapes:
        a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        npad = (N + 255) // 256 * 256
        w = torch.zeros(npad, K, device=dev, dtype=torch.bfloat16)
        w[:N] = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) * 0.05).to(torch.bfloat16)
        b = torch.zeros(npad, device=dev, dtype=torch.float32)
        o = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi != 1 else None
        x = torch.zeros(M, N, device=dev, dtype=torch.float32) if epi == 1 else None
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)

        def ours():
            eng.lib.ssp2_linear_bf16(stream, epi, p(a), K, p(w), K, p(b), M, N, K, p(o), N, p(x), N, 2)
        bl = b[:N].to(torch.bfloat16)
        lib = lambda: torch.nn.functional.linear(a, w[:N], bl)
        row = {"shape": [M, N, K], "epilogue": what}
        runs = [("ours", ours), ("library_bias_only", lib)]
        if epi == 2:
            # the library's OWN fused GELU (hipBLASLt bias + GELU epilogue, the cheaper tanh form — not the exact erf the reference computes, so a
            # yardstick only): what a fused activation costs AMD's hand-written kernel on this shape
            wt = w[:N].t()
            runs.append(("library_fused_gelu_tanh", lambda: torch._addmm_activation(bl, a, wt, use_gelu=True)))
        for label, fn in runs:
            for _ in range(5):
                fn()
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); e1.synchronize()
            n = max(40, int(seconds / (e0.elapsed_time(e1) / 20 * 1e-3)))
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.time()
            a0.record()
            for _ in range(n):
                fn()
            a1.record(); a1.synchronize()
            t1 = time.time()
            us = a0.elapsed_time(a1) / n * 1e3
            tf = 2.0 * M * N * K / (us * 1e-6) / 1e12
            w_ = sampler.window(t0, t1)
            row[label] = {"us": round(us, 1), "tflops": round(tf, 1), **w_,
                          "tflops_per_kw": round(tf / w_["power_w"] * 1e3, 1) if w_.get("power_w") else None}
        row["ours_over_library"] = round(row["ours"]["tflops"] / row["library_bias_only"]["tflops"], 3)
        if "library_fused_gelu_tanh" in row:
            row["ours_over_library_fused_gelu"] = round(row["ours"]["tflops"] / row["library_fused_gelu_tanh"]["tflops"], 3)
        out[name] = row
        del a, w, o, x
        torch.cuda.empty_cache()
    sampler.stop = True
    out["note"] = ("each family's shape of the 320-image search chunk, launched back to back for %.1f s (ours: ssp2_linear_bf16 on the persistent 256 x 256 kernel, "
                   "fused epilogue as in the step; library: hipBLASLt, bias only); power / clock from sysfs, None where it is not readable" % seconds)
    return out


def cpu_budget():
    """(threads to use, facts): the PHYSICAL cores this process may really run on — the smallest of the physical-core count, the
    scheduler affinity mask and the cgroup CPU quota.  Round 3 let PyTorch take every logical CPU of the host (128) on a box whose
    share is smaller; the oversubscribed runs differed by 50 % between boxes (9.4 / 13.0 / 14.3 image-forwards/s)."""
    logical = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except Exception:
        affinity = logical
    physical = logical
    try:
        cores = set()
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
        if cores:
            physical = len(cores)
    except Exception:
        pass
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
            if q != "max":
                quota = max(1, int(int(q) / int(per)))
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                q, per = int(f.read()), int(f2.read())
                if q > 0:
                    quota = max(1, q // per)
        except Exception:
            pass
    threads = max(1, min(v for v in (physical, affinity, quota) if v))
    return threads, {"nproc": logical, "affinity": affinity, "physical_cores": physical, "cgroup_quota": quota}


def cpu_baseline(args, weights, n_eval, calib_n):
    """Oracle (kind="port": bit-exact restatement of the reference, pinned by tests/test_oracle_golden.py) timed on
    the host cores over a bounded sample; converted to the metric's unit with the step's own mix of passes.
    Threads are pinned to the physical cores the process may use (cpu_budget), the sample is run THREE times after a warm-up and
    the MEDIAN pass is reported with the min-max spread beside it."""
    from oracle import ref_cpu
    from oracle.vit_modules import build_from_flat
    threads, facts = cpu_budget()
    old_threads = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        model = build_from_flat(weights, "timm")
        g = torch.Generator().manual_seed(123)
        n1 = n2 = max(8, args.cpu_sample)
        bs = min(32, n1)
        img = int(weights["img"])
        calib = [{"pixel_values": torch.randn(bs, 3, img, img, generator=g)} for _ in range(n1 // bs)]
        evalb = [{"pixel_values": torch.randn(bs, 3, img, img, generator=g), "labels": torch.zeros(bs, dtype=torch.int64)}
                 for _ in range(n2 // bs)]
        n1 = n2 = bs * len(calib)
        ref_cpu.ffn_activation_importance(model, calib[:1])            # warm-up (oneDNN primitive cache, page faults)
        ref_cpu.top1_counts(model, evalb[:1])
        r1s, r2s = [], []
        for _ in range(3):
            t0 = time.time(); ref_cpu.ffn_activation_importance(model, calib); t1 = time.time()
            ref_cpu.top1_counts(model, evalb); t2 = time.time()
            r1s.append(n1 / (t1 - t0)); r2s.append(n2 / (t2 - t1))
    finally:
        torch.set_num_threads(old_threads)
    L = int(weights["depth"])
    units = calib_n + (L + 1) * n_eval

    def rate(r1, r2):
        return units / (calib_n / r1 + (L + 1) * n_eval / r2)
    vals = sorted(rate(a, b) for a, b in zip(r1s, r2s))
    r1, r2 = sorted(r1s)[1], sorted(r2s)[1]
    step_s = calib_n / r1 + (L + 1) * n_eval / r2
    return {"value": round(vals[1], 3), "unit": "image-forwards/s", "cores": threads, "threads": threads, **facts,
            "kind": "port", "passes": 3, "statistic": "median", "spread": [round(vals[0], 3), round(vals[2], 3)],
            "spread_rel": round((vals[2] - vals[0]) / vals[1], 4),
            "prune_time_s_extrapolated": round(step_s, 1),
            "stage1_img_per_s": round(r1, 2), "eval_img_per_s": round(r2, 2),
            "sample": f"bf16-autocast oracle on {threads} threads (= physical cores available to the process): 3 passes of stage-1 "
                      f"scoring of {n1} images + top-1 eval of {n2} images (batch {bs}) after a warm-up batch, median pass; "
                      f"extrapolated to {calib_n} calib + {L + 1}x{n_eval} eval image-forwards; "
                      f"model deep-copies of the reference not counted"}


def act_l2_figure(eng, batch, tokens, d_int, dev, n_images=512):
    """The HBM-bound kernel of the path on its own: the standalone activation-L2 accumulate (a2) over one layer's
    activation of one stage-1 launch (`n_images` = 8 calibration batches of 64, scored per batch exactly as the unfused
    path of ssp2_layers does), outside the timed region (in the step it is fused into the fc1 epilogue and reads nothing
    from HBM).  Algorithmic bytes = n*N*d_int*2 read once: 620 MB per call for ViT-B/16, more than the 256 MiB
    Infinity Cache, and the calls alternate between two such tensors — the figure is HBM, not cache replay.
    `achieved` counts the whole call (both kernels and the boundary between them); the rocprofv3 summary recorded at this
    source hash (scripts/pmc_act_l2.sh) gives the norms kernel's own duration and its FETCH_SIZE bytes."""
    import ctypes as C
    one = n_images * tokens * d_int * 2
    acts = [torch.randn(n_images, tokens, d_int, device=dev, dtype=torch.bfloat16) for _ in range(2)]
    groups = (n_images + batch - 1) // batch
    ws = torch.empty(2, n_images, d_int, dtype=torch.float32, device=dev)
    outv = torch.empty(groups, d_int, dtype=torch.float32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ptrs = [C.c_void_p(a.data_ptr()) for a in acts]

    def call(i):
        eng.lib.ssp2_act_l2_accum(stream, ptrs[i % 2], 0, n_images, tokens, d_int, d_int, 0, batch, C.c_void_p(ws.data_ptr()),
                                  C.c_void_p(outv.data_ptr()), d_int)
    for i in range(4):
        call(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for i in range(reps):
        call(i)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    gbps = one / (us * 1e-6) / 1e9
    js, name, stale = _newest_summary("r*_pmc_act_l2.json")
    k = (js or {}).get("act_l2_norms_kernel", {})
    return {"bound": "hbm", "achieved": round(gbps, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbps / 8000.0, 4),
            "bytes_per_launch": one, "avg_launch_us": round(us, 2),
            "workload": f"{n_images} x {tokens} x {d_int} bf16 activation ({one >> 20} MiB > 256 MiB Infinity Cache), two tensors alternating, scored in groups of {batch}",
            "traffic": k.get("hbm_bytes_per_launch"), "kernel_only_us_rocprof": k.get("avg_us"), "kernel_only_gbps_rocprof": k.get("hbm_gbps_algorithmic"),
            "traffic_source": name if js else ({"stale_summary_ignored": stale} if stale else None),
            "kernel": "act_l2_norms_kernel<bf16> + score_colsum_halves_kernel (standalone a2; 2 launches per call)"}


def api_level(args, weights, calib, evalb, plan, dev, steps=5, search_batches=None):
    """The same prune through the reference-named API on a LIVE module that sits on the device (as the reference keeps
    it): Auto2SSPInterface importances (attention first, then MLP, both enqueued before either is waited for) ->
    prune_vit_mlp_width(precomputed_importance) -> prune_vit_attention_blocks(selected_indices).  The engine build
    (weight ingest + workspace) happens inside the bracket: a fresh module per step has no cached engine; the modules themselves
    are built before the first bracket (see below)."""
    from ssp2vit import vit_pruning as vp
    from ssp2vit.mask_conjunction import Auto2SSPInterface
    from ssp2vit.modules import EngineViT
    times, last = [], None
    B = int(weights["depth"])
    # the fresh modules of all steps are built BEFORE the first bracket, so that the brackets follow one another as the core step's do:
    # building one between two brackets (a 344 MB host-to-device copy) leaves the card idle for tens of ms, and the first ~3 ms of the next
    # prune then run at the clocks of an idle card (scripts/api_profile.py: fit 96.0 against 91.5 ms right after another fit) — a property
    # of the measuring loop, not of the API layer whose cost this figure is meant to show
    models = [EngineViT(weights).to(dev) for _ in range(steps + 1)]
    torch.cuda.synchronize()
    for model in models:                                         # first pass = warm-up
        t0 = time.perf_counter()
        if evalb is None:
            # ONE interface over ONE loader, as the reference's CLI builds it (auto_2ssp.py:765-775): fit() takes both importances from one
            # walk; the search takes the first `search_batches` batches, the stage-1 hook all of them (score_batch_limit, an extension)
            iface = Auto2SSPInterface(model, calib, device=dev, importance_mode="copy", batch_limit=search_batches, min_remaining=512,
                                      score_batch_limit=None)
            att, mlp = iface.fit()
        else:
            s2 = Auto2SSPInterface(model, evalb, device=dev, importance_mode="copy", batch_limit=len(evalb), min_remaining=512)
            s1 = Auto2SSPInterface(model, calib, device=dev, batch_limit=None, min_remaining=512)
            att_fin = s2._att_importance_deferred()                  # fit(): attention first (mask_conjunction.py:359-362) ...
            mlp_fin = s1._mlp_importance_deferred()                  # ... then MLP; both enqueued, then awaited
            att, mlp = att_fin(), mlp_fin()
        res = vp.prune_vit_mlp_width(model, n_to_prune_per_block=[plan.per_block_neurons_to_prune] * B, min_remaining=512,
                                     strategy="l1", collect_masks=True, precomputed_importance=[x.to(torch.float32) for x in mlp])
        K = plan.blocks_to_prune
        out = vp.prune_vit_attention_blocks(res["model"], sparsity=K / B, dataloader=None, device=dev, num_to_prune=K,
                                            show_progress=False, selected_indices=[int(i) for i in torch.argsort(att)[:K]])
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        last = out["pruned_indices"]
        vp.release_engines()
        del res, out
    del models, model
    vp.release_engines(free=True)
    t = sorted(times[1:])
    return {"prune_time_s": round(t[len(t) // 2], 4), "steps": steps, "all_s": [round(x, 4) for x in times[1:]],
            "selected_blocks": last,
            "path": ("EngineViT(weights).to(device) -> Auto2SSPInterface.fit() (ONE walk over one loader: the search's baseline carries the stage-1 hook)"
                     if evalb is None else "EngineViT(weights).to(device) -> Auto2SSPInterface (deferred att + mlp importances, two loaders)")
                    + " -> prune_vit_mlp_width(precomputed_importance) -> prune_vit_attention_blocks(selected_indices); engine build inside the bracket"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="vit_base_patch16_224")
    ap.add_argument("--config", type=int, default=1, choices=(1, 2),
                    help="1: BASELINE configs[1], --calib/--eval-batches PER GPU (weak scaling); 2: configs[2], 2048 calib + 2560 eval "
                         "images in total over the ranks, targets 0.25/0.375/0.5 from one stage-1 pass and one search (strong scaling)")
    ap.add_argument("--calib", type=int, default=512, help="calibration images per GPU (config 1)")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--eval-batches", type=int, default=5, help="evaluation batches per GPU (config 1)")
    ap.add_argument("--calib-chunk", type=int, default=512, help="images per stage-1 forward (0 = one batch; see core.stage1_scores)")
    ap.add_argument("--eval-chunk", type=int, default=0, help="images per search forward (0 = min(all eval images of the rank, 320))")
    ap.add_argument("--target", type=float, default=0.375)
    ap.add_argument("--precision", default="bf16", choices=("bf16", "fp8"),
                    help="fp8: QKV / fc1 / fc2 on e4m3 MFMA (opt-in, BASELINE configs[4]; its own tolerance, not the parity mode)")
    ap.add_argument("--cpu-sample", type=int, default=64, help="images per leg (stage-1 scoring / top-1 eval) per pass of the CPU baseline; three passes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-sustained", action="store_true", help="skip the per-family sustained / library-ceiling measurement (about 10 s)")
    ap.add_argument("--no-api", action="store_true", help="skip the API-level secondary measurement")
    ap.add_argument("--act-l2-only", action="store_true", help="only the standalone activation-L2 kernel loop (PMC passes)")
    ap.add_argument("--no-batch-candidates", action="store_true",
                    help="candidate-major search (one launch per candidate and block) instead of the layer-major one, "
                         "in which all candidates under way run a block in ONE launch (engine workspace for "
                         "(depth-1) x eval images)")
    ap.add_argument("--host-inputs", action="store_true",
                    help="keep the batches in pinned HOST memory, as a dataloader hands them over: every step then pays "
                         "the PCIe copy (the PCIe-inclusive rate of DESIGN.md; never the default)")
    ap.add_argument("--uint8", action="store_true",
                    help="synthetic uint8 HWC images run through the GPU input pipeline (f4: resize -> ToTensor -> Normalize); with "
                         "--host-inputs the 3-byte pixels cross PCIe and the pipeline runs inside the timed region")
    ap.add_argument("--two-streams", action="store_true",
                    help="run stage 1 and a share of the search candidates on a second HIP stream with its own engine "
                         "workspace (per-launch durations then include the share of the machine lent to the other stream, "
                         "so the roofline object is not a clean kernel figure)")
    ap.add_argument("--no-overlap-figure", action="store_true", help="(accepted; the secondary two-stream figure is opt-in since round 5: --overlap-figure)")
    ap.add_argument("--overlap-figure", action="store_true",
                    help="also time the round-4 form with stage 1 on a second HIP stream beside the search (two passes; slower than the one-pass default "
                         "since round 5: 100.4 against 93.5 ms) and attach it as `stage1_beside_search`")
    ap.add_argument("--two-pass", action="store_true",
                    help="rounds 1-4: stage 1 and the depth search as two independent passes over DISJOINT calibration / evaluation sets. "
                         "Default: the reference's one-loader semantics (mask_conjunction.py:276-281, :327) — the evaluation batches are the "
                         "first --eval-batches calibration batches and ONE dense pass over them serves the stage-1 hook and the search's baseline")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="no GPU work: the ranks meet over gloo and rank 0 prints a line with the world size (CPU test of --gpus N)")
    ap.add_argument("--overlap-stage1", action="store_true",
                    help="stage 1 on a second HIP stream (own engine workspace) BESIDE the layer-major search on the main one; "
                         "unlike --two-streams the search keeps its layer-major form and its candidates are not shared out")
    args = ap.parse_args()
    if args.overlap_stage1 and args.two_streams:
        raise SystemExit("--overlap-stage1 and --two-streams are two forms of the same idea: pick one")

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.selftest_launcher:
        import torch.distributed as dist
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
            t = torch.ones(1); dist.all_reduce(t)
            assert int(t) == world
            dist.destroy_process_group()
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        if os.environ.get("SSP2_SELFTEST_FAIL_RANK") == str(rank):
            raise SystemExit(3)
        if rank == 0:
            print(json.dumps({"metric": "launcher_selftest", "n_gpus": world, "cuda_initialised": torch.cuda.is_initialized()}), flush=True)
        return
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # SSP2_REHEARSE_ONE_CARD=1: every rank takes cuda:0 and the ranks meet over gloo (RCCL refuses two ranks on one device).  A
    # correctness rehearsal of the N-rank job on a one-GPU box — launcher, dealing of the batches, exchange steps, one result line —
    # whose line says so and is never a scaling figure (the ranks share one card's time).
    one_card = world > 1 and os.environ.get("SSP2_REHEARSE_ONE_CARD") == "1"
    if one_card:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    if world > 1 or ("RANK" in os.environ and os.environ.get("SSP2_FORCE_COLLECTIVES")):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_card:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)          # "nccl" is RCCL on ROCm
        pg = dist.group.WORLD

    from ssp2vit import core, dist as sdist
    from ssp2vit.engine import VitEngine
    from ssp2vit.planner import plan_from_stats, stats_from_shapes
    from ssp2vit.weights import VIT_CONFIGS, synthetic_weights

    img, patch, dim, heads, d_int, depth = VIT_CONFIGS[args.model]
    tokens = (img // patch) ** 2 + 1
    weights = synthetic_weights(args.model, classes=1000, seed=0, std=0.02, eps=1e-6, spread=4.0)

    # ---- how many batches this rank owns
    if args.config == 2:
        tot_cal_b, tot_ev_b = 2048 // args.batch, 2560 // args.batch
        n_cal_b = len(range(rank, tot_cal_b, world)); n_ev_b = len(range(rank, tot_ev_b, world))
        targets = [0.25, 0.375, 0.5]
        calib_total, eval_total = tot_cal_b * args.batch, tot_ev_b * args.batch
    else:
        n_cal_b, n_ev_b = args.calib // args.batch, args.eval_batches
        targets = [args.target]
        calib_total, eval_total = world * n_cal_b * args.batch, world * n_ev_b * args.batch
    n_calib, n_eval = n_cal_b * args.batch, n_ev_b * args.batch
    eval_chunk = args.eval_chunk or min(n_eval, 320)
    calib_chunk = min(args.calib_chunk or args.batch, max(n_calib, args.batch))

    stats = stats_from_shapes(dim, depth, d_int, 1000, tokens, patch)
    plans = [plan_from_stats(stats, t, min_remaining=512) for t in targets]
    plan = plans[targets.index(args.target)] if args.target in targets else plans[0]

    if args.act_l2_only:
        eng = VitEngine(weights, device=dev, max_images=args.batch)
        print(json.dumps(act_l2_figure(eng, args.batch, tokens, d_int, dev)), flush=True)
        return
    cap = max(args.batch, eval_chunk, calib_chunk)
    args.batch_candidates = not args.no_batch_candidates and not args.two_streams
    if args.batch_candidates:
        cap = max(cap, depth * eval_chunk)
        if not args.two_pass and not (args.two_streams or args.overlap_stage1):
            # one pass: the search's streams are in the slab layout of the stage-1 hook (every batch padded to a multiple of 128 rows)
            cap = max(cap, core.lm_capacity_images(tokens, depth, max(1, eval_chunk // args.batch) * args.batch, args.batch))
            if n_calib > n_eval:       # + the calibration batches the search does not take: they ride in front of slot 0 (hook only)
                cap += -(-core.slab_rows(tokens, min(n_calib - n_eval, eval_chunk) // args.batch * args.batch, args.batch) // tokens) + args.batch
    eng = VitEngine(weights, device=dev, max_images=cap, precision=args.precision)
    # --two-streams: stage 1 (calibration scores) and stage 2 (depth search on the dense model) are independent, and so
    # are the search candidates: with a second engine workspace (weights uploaded twice, 173 MB) on a second HIP stream
    # the memory-bound kernels of one stream (LayerNorm, attention, epilogue tails) overlap the matrix-bound kernels of
    # the other and the partial last round of a persistent GEMM is filled by the other stream's workgroups.
    second = args.two_streams or args.overlap_stage1
    eng1 = VitEngine(weights, device=dev, max_images=max(args.batch, calib_chunk), precision=args.precision) if second else eng
    side = torch.cuda.Stream(dev) if second else None
    d_ints = [d_int] * depth
    twins = [eng.pruned_twin([d_int - p.per_block_neurons_to_prune] * depth, max_images=args.batch) for p in plans]

    # ---- synthetic ImageNet-shape inputs, resident in HBM before the timed region; every rank its own shard.
    # Batch b of the GLOBAL calibration / eval set is a function of b alone (seeded per batch), and this rank's k-th batch is global
    # batch k * world + rank: at a fixed total (--config 2) every world size sees the same images, so the selections must agree.
    g = torch.Generator(device=dev)
    pp = None
    if args.uint8:
        from ssp2vit.preprocess import GpuPreprocessor
        pp = GpuPreprocessor((img, img), img, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5), device=dev)

    def fresh_pixels(kind, global_batch):
        g.manual_seed(1 + 2 * global_batch + kind)
        if pp is None:
            return torch.randn(args.batch, 3, img, img, generator=g, device=dev), None
        u8 = torch.randint(0, 256, (args.batch, img, img, 3), generator=g, device=dev, dtype=torch.uint8)
        return pp(u8), u8

    one_pass = not args.two_pass and not second
    calib, evalb = [], []

    def teacher(px):
        x = eng.embed(px); eng.layers(x, args.batch)
        return eng.head(x, args.batch, want_pred=True)[1].long()              # teacher labels: dense model's own argmax

    for k in range(max(n_cal_b, n_ev_b) if one_pass else n_cal_b):
        px, u8 = fresh_pixels(0, k * world + rank)
        calib.append({"pixel_values": px, "u8": u8})
        if one_pass and k < n_ev_b:                                            # ONE loader: its first batches are what the search evaluates
            calib[-1]["labels"] = teacher(px)
    for k in range(0 if one_pass else n_ev_b):
        px, u8 = fresh_pixels(1, k * world + rank)
        evalb.append({"pixel_values": px, "labels": teacher(px), "u8": u8})

    def as_loader(batches):
        out = []
        for b in batches:
            d = {k: v for k, v in b.items() if k != "u8"}
            if args.host_inputs:
                if b["u8"] is not None:                                           # raw 3-byte pixels + the GPU pipeline
                    d["pixel_values"] = b["u8"].cpu().pin_memory(); d["preprocess"] = pp
                else:
                    d["pixel_values"] = b["pixel_values"].cpu().pin_memory()
                if "labels" in d:
                    d["labels"] = d["labels"].cpu().pin_memory()
            out.append(d)
        return out
    calib_loader, eval_loader = as_loader(calib), as_loader(evalb)            # this rank's batches only (sharded=True)

    sdist.TIMING = pg is not None              # device time of the exchange steps, both configs

    # GLOBAL batch limits of the two stages (this rank's k-th batch is global batch k * world + rank)
    g_score_limit = tot_cal_b if args.config == 2 else world * n_cal_b
    g_search_limit = tot_ev_b if args.config == 2 else world * n_ev_b

    def step(s1_eng=eng1, sd=side, share=args.two_streams):
        # both stages are enqueued before the host waits for either: the a7 mask step runs on the CPU while the GPU
        # is still searching (the two stages are independent: stage 2 evaluates the dense model)
        if one_pass and sd is None:
            # ONE walk over ONE loader: the dense forward over the batches both stages take is the stage-1 pass AND the search's baseline
            scores, search = core.prune_pass(eng, calib_loader, d_ints, "pre_gelu", depth, score_limit=g_score_limit,
                                             search_limit=g_search_limit, score_chain="fp32", process_group=pg, chunk_images=calib_chunk,
                                             eval_chunk_images=eval_chunk, defer=True, sharded=True, batch_candidates=args.batch_candidates)
        elif sd is None:
            scores = core.stage1_scores(s1_eng, calib_loader, d_ints, "pre_gelu", score_chain="fp32", process_group=pg,
                                        chunk_images=calib_chunk, defer=True, sharded=True, batch_limit=g_score_limit if one_pass else None)
        else:
            sd.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(sd):
                scores = core.stage1_scores(s1_eng, calib_loader, d_ints, "pre_gelu", score_chain="fp32", process_group=pg,
                                            chunk_images=calib_chunk, defer=True, sharded=True, batch_limit=g_score_limit if one_pass else None)
        # the side stream also takes a share of the search candidates behind its stage-1 launch (lead ~ the stage-1
        # work expressed in block passes of the search chunk)
        if not (one_pass and sd is None):
            search = core.depth_search_counts(eng, calib_loader if one_pass else eval_loader, depth, batch_limit=g_search_limit if one_pass else None,
                                              process_group=pg, chunk_images=eval_chunk, defer=True,
                                              aux_engine=s1_eng if share else None, aux_stream=sd if share else None,
                                              aux_lead=n_calib * depth / max(1, eval_chunk),
                                              batch_candidates=args.batch_candidates, sharded=True)
        imps = scores()
        all_masks = []
        for p in plans:                                                   # a7 mask step per target (host, 12 x 3072; the blocks side by side on four threads)
            all_masks.append(core.cut_masks(imps, [p.per_block_neurons_to_prune] * len(imps)))
        for masks, twin in zip(all_masks, twins):                          # a8 apply: the gathers into the pruned twin do not depend on the search —
            eng.apply_ffn_into(twin, masks)                                 # queued behind it NOW, so the prune does not end on the host's latency
        base, cand, total = search()
        impact = torch.tensor(core.impacts_from_counts(base, cand, total), dtype=torch.float32)
        if sd is not None:
            torch.cuda.current_stream(dev).wait_stream(sd)
        chosen = []
        for p, masks, twin in zip(plans, all_masks, twins):
            blocks = sorted(int(i) for i in torch.argsort(impact)[: p.blocks_to_prune])   # a9 (auto_2ssp.py:857)
            eng.apply_attention_into(twin, blocks)                         # a9 apply: the chosen blocks lose their attention (flags)
            chosen.append(blocks)
        return imps, impact, all_masks, chosen

    def sync_all():
        if pg is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    if sdist.TIMING:
        sdist.collective_ms()
    prof = None
    if not args.no_roofline:
        prof = eng.profile("gemm_fc1"); prof.__enter__()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if prof is not None:
        prof.__exit__(None, None, None)
    coll = sdist.collective_ms() if sdist.TIMING else None
    if world > 1 or "RANK" in os.environ:
        # one line per rank on stderr: a failed or slow multi-rank run is then readable from the tail of the driver's log
        print(sdist.rank_summary({"local_ms_per_step": round(elapsed / args.steps * 1e3, 3), "steps": args.steps,
                                  "calib_batches": n_cal_b, "eval_batches": n_ev_b,
                                  **({"pass": ",".join(f"{k}={v}" for k, v in core.PASS_STATS.items())} if one_pass else {}),
                                  "collective_ms_per_step": {k: round(v / max(1, args.steps), 3) for k, v in (coll or {}).items()}}),
              file=sys.stderr, flush=True)

    # every kernel family of the step: one more step, untimed, with HIP events around EVERY launch (the timed region above
    # carries events on the fc1 family only, as in rounds 1-2, so `value` stays comparable)
    families = None
    if not args.no_roofline and not second:
        with eng.profile("all") as pall:
            step()
        families = roofline_by_family(pall.by_class, args.precision)
    # stage-1-only rate (secondary figure, separate timed loop so the headline region stays untouched)
    sync_all(); t1 = time.perf_counter()
    core.stage1_scores(eng1, calib_loader, d_ints, "pre_gelu", score_chain="fp32", process_group=pg, chunk_images=calib_chunk, sharded=True,
                       batch_limit=g_score_limit if one_pass else None)
    sync_all(); s1_s = time.perf_counter() - t1

    # secondary figure: the same step with stage 1 on a second HIP stream (own engine workspace) beside the layer-major search —
    # the two stages are independent.  Not the headline: kernels of two streams share the CUs, so per-launch durations (the
    # roofline object) are not clean kernel figures in such a run.
    overlap = None
    if world == 1 and args.config == 1 and not second and not args.no_roofline and args.overlap_figure and not args.no_overlap_figure:
        e2 = VitEngine(weights, device=dev, max_images=max(args.batch, calib_chunk), precision=args.precision)
        st2 = torch.cuda.Stream(dev)
        step(e2, st2, False); sync_all()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            o2 = step(e2, st2, False)
        sync_all()
        overlap = (time.perf_counter() - t2, o2[3])
        e2.close()

    el = torch.tensor([elapsed, s1_s], dtype=torch.float64, device=sdist._default_device(pg) if pg is not None else dev)   # nccl: this card; gloo: host
    if pg is not None:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    elapsed, s1_s = float(el[0]), float(el[1])

    if rank == 0:
        units_step = calib_total + (depth + 1) * eval_total                             # whole job, all ranks
        value = units_step * args.steps / elapsed
        tail = 2.0 * dim / (4 * dim + 2 * d_int + 2 * tokens)                            # cost of the CLS-only last block / a full block
        executed = calib_total * depth + eval_total * ((depth - 1) + (depth - 1) * depth // 2 + (depth + 1) * tail)   # block passes per step
        if one_pass:       # the baseline's dense pass IS the stage-1 pass of those images (its last block runs once more, hooked, beside the CLS-only tail)
            executed -= min(calib_total, eval_total) * (depth - 1)
        reference_equiv = calib_total * depth + eval_total * depth * (depth + 1)
        tdesc = ", ".join(f"{t} (K={p.blocks_to_prune}, t={p.per_block_neurons_to_prune})" for t, p in zip(targets, plans))
        per = "in total over the ranks" if args.config == 2 else "/GPU"
        line = {
            "metric": "2ssp_prune_image_forwards_per_sec", "value": round(value, 1), "unit": "image-forwards/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True,
            "scaling": "strong" if args.config == 2 else "weak",
            "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "fp8 (e4m3 QKV/fc1/fc2 GEMMs, bf16 elsewhere)",
            "data": ("synthetic" + (", uint8 HWC through the GPU input pipeline" if args.uint8 else "")
                     + (", pinned host batches copied over PCIe inside the timed region" if args.host_inputs else "")),
            "config": {"workload": f"{args.model}, {calib_total if args.config == 2 else n_calib} calib images {per}, full 2SSP @ {tdesc}, "
                                   f"one-shot depth search over {eval_total if args.config == 2 else n_eval} eval images {per}"
                                   + (" = the FIRST batches of the calibration loader (the reference's one-loader semantics, mask_conjunction.py:276-281, :327: "
                                      "one dense pass over them serves the stage-1 hook and the search's baseline)" if one_pass else " (a disjoint set, two passes)")
                                   + f", batch {args.batch}, select + apply into a pruned engine inside the step",
                       "baseline_config": f"BASELINE.json configs[{args.config}]",
                       "weights": "random-init trunc-normal(0.02), fc1 rows log-uniform x[1/4,4], seed 0",
                       "parallelism": f"dp{world} (batches dealt round-robin, every rank's loader yields only its own, weights replicated)"},
            "prune_time_s": round(elapsed / args.steps, 4),
            "calib_images_per_sec": round(calib_total / s1_s, 1),
            "executed_block_pass_fraction": round(executed / reference_equiv, 4),
            "selected_blocks": out[3][targets.index(args.target)] if args.target in targets else out[3][0],
            "selected_blocks_per_target": {str(t): b for t, b in zip(targets, out[3])},
            "pruned_neurons_per_block": plan.per_block_neurons_to_prune,
            "passes": "one (stage-1 hook on the search's baseline forward)" if one_pass else "two",
            "pass_stats": dict(core.PASS_STATS) if one_pass else None,
            "streams": 2 if (args.two_streams or args.overlap_stage1) else 1, "stage1_beside_search": bool(args.overlap_stage1), "search": "layer-major" if args.batch_candidates else "candidate-major",
        }
        if one_card:
            line["rehearsal"] = f"{world} ranks share ONE card over gloo (SSP2_REHEARSE_ONE_CARD=1): a correctness rehearsal, not a scaling figure"
        if overlap is not None:
            line["stage1_beside_search"] = {"ms_per_step": round(1e3 * overlap[0] / args.steps, 2), "value": round(units_step * args.steps / overlap[0], 1),
                                            "streams": 2, "same_selection": overlap[1] == out[3],
                                            "note": "stage 1 on a second HIP stream with its own engine workspace beside the layer-major search; secondary figure (bench.py --overlap-stage1 makes it the timed region)"}
        if coll is not None:
            line["collectives"] = {"backend": "gloo (one-card rehearsal)" if one_card else "nccl (RCCL)", "world_size": world,
                                   "device_ms_per_step": {k: round(v / args.steps, 3) for k, v in coll.items()}}
        if prof is not None and prof.launches:
            # dominant kernel family: fc1 (+bias +erf-GELU; + fused activation-L2 partials in stage 1).
            # achieved = algorithmic flops (2*M*N*K summed over the recorded launches) / summed HIP-event durations.
            ach = prof.flops / (prof.total_ms * 1e-3) / 1e12
            peak = BF16_MFMA_PEAK_TFLOPS if args.precision == "bf16" else FP8_MFMA_PEAK_TFLOPS
            lm = f", x 1..{depth - 1} in the layer-major search" if args.batch_candidates else ""
            traffic, tsrc = pmc_traffic(args.model, args.precision)
            line["roofline"] = {"bound": "mfma", "kernel": "fc1 GEMM family: gemm256 persistent 256x256 <EPI_FC1,SCORE> (stage 1: + fused activation-L2 partials; search passes: SCORE=0) and gemm_bf16_kernel<EPI_FC1> 128x128 (CLS tail); bias + erf-GELU fused",
                                "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s",
                                "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": tsrc, "pmc": pmc_mfma(args.model, args.precision),
                                "launches": prof.launches, "avg_launch_us": round(prof.total_ms * 1e3 / prof.launches, 2),
                                "flops_per_launch_avg": prof.flops / prof.launches,
                                "library_yardstick": library_yardstick("fc1") if args.precision == "bf16" and args.model == "vit_base_patch16_224" else None,
                                "shapes": (f"[{core.slab_rows(tokens, max(1, eval_chunk // args.batch) * args.batch + (core.PASS_STATS['hook_only_batches_fused'] * args.batch), args.batch)} hooked | "
                                           f"{core.slab_rows(tokens, max(1, eval_chunk // args.batch) * args.batch, args.batch)} x 0..{depth - 2} unhooked | {eval_chunk}] x {d_int} x {dim} "
                                           f"(the search chunk's baseline in {args.batch}-image slabs + the {core.PASS_STATS['hook_only_batches_fused']} calibration batches the search does not take, "
                                           f"all carrying the stage-1 hook | the candidates under way, second launch of the same block | CLS tail)"
                                           if one_pass else
                                           f"[{eng.rows(min(n_calib, calib_chunk), args.batch)} | {eval_chunk * tokens} | {eval_chunk}] x {d_int} x {dim} (stage-1 launch in {args.batch}-image slabs | search chunk{lm} | CLS tail)")}
        if args.precision == "fp8":
            line["fp8_saturation_events"] = eng.fp8_saturation()       # waves that clipped an attention output at the e4m3 range (0 = none)
        if families is not None:
            dom = max((k for k in families if k != "other"), key=lambda k: families[k]["ms"])
            line["roofline_by_family"] = dict(families, time_dominant=dom,
                                              note="one untimed step with HIP events around every launch; `roofline` above is the fc1 family inside the timed region")
        if not args.no_roofline:
            line["act_l2_kernel"] = act_l2_figure(eng, args.batch, tokens, d_int, dev)
        if families is not None and world == 1 and args.precision == "bf16" and not args.no_sustained:
            try:
                sus = sustained_families(eng, dev, eval_chunk * tokens, dim, d_int)
                line["sustained_by_family"] = sus
                for fam, keys in (("fc1", ("fc1",)), ("qkv", ("qkv",)), ("resid", ("proj", "fc2"))):     # what VERDICT asks for inside roofline_by_family
                    if fam in line.get("roofline_by_family", {}):
                        line["roofline_by_family"][fam]["sustained"] = {k: {"ours_tflops": sus[k]["ours"]["tflops"], "power_w": sus[k]["ours"]["power_w"],
                                                                               "sclk_mhz": sus[k]["ours"]["sclk_mhz"],
                                                                               "library_ceiling_tflops": sus[k]["library_bias_only"]["tflops"],
                                                                               "library_power_w": sus[k]["library_bias_only"]["power_w"],
                                                                               "library_sclk_mhz": sus[k]["library_bias_only"]["sclk_mhz"],
                                                                               **({"library_fused_gelu_tanh_tflops": sus[k]["library_fused_gelu_tanh"]["tflops"]}
                                                                                  if "library_fused_gelu_tanh" in sus[k] else {})} for k in keys}
            except Exception as exc:                                     # a yardstick, never a reason to lose the line
                line["sustained_by_family"] = {"error": repr(exc)[:200]}
        if world == 1 and not args.no_api and args.config == 1 and args.precision == "bf16":
            eng.close(); eng1.close()
            for t in twins:
                t.close()
            torch.cuda.empty_cache()
            if one_pass:
                line["api"] = api_level(args, weights, calib_loader, None, plan, dev, search_batches=n_ev_b)
            else:
                api_calib = [{"pixel_values": b["pixel_values"]} for b in calib_loader]
                line["api"] = api_level(args, weights, api_calib, eval_loader, plan, dev)
            line["api"]["vs_core_step"] = round(line["api"]["prune_time_s"] / line["prune_time_s"], 3)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, weights, n_eval, n_calib)
        print(json.dumps(line), flush=True)
    if pg is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
