#!/bin/bash
# HBM traffic of the standalone activation-L2 kernel (north_star: "rocprof reports achieved HBM GB/s for the L2-accum
# kernel"): rocprofv3 --pmc FETCH_SIZE over `bench.py --act-l2-only`, which rotates over > 256 MiB of distinct
# activations (so the counter sees HBM, not Infinity-Cache replay).  FETCH_SIZE is in KiB and reads HALF of a wide
# coalesced streaming read on gfx950 (MI355X_MICROARCH.md, HBM): doubled in the summary.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_act_l2 -- python3 bench.py --act-l2-only > gpurun_out/pmc_act_l2.log 2>&1
python3 - <<'P' > gpurun_out/pmc_act_l2.json
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "2ssp-x-vit_amd"))
from ssp2vit import _lib
d = "gpurun_out/pmc_act_l2"
cc = sorted(glob.glob(d + "/*/*_counter_collection.csv"), key=os.path.getmtime)[-1]
kt = sorted(glob.glob(d + "/*/*_kernel_trace.csv"), key=os.path.getmtime)[-1]
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
acc = collections.defaultdict(lambda: [0.0, 0, 0])
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "FETCH_SIZE":
        continue
    k = "act_l2_norms_kernel" if "act_l2_norms_kernel" in r["Kernel_Name"] else r["Kernel_Name"].split("(")[0].split("<")[0][-40:]
    a = acc[k]; a[0] += float(r["Counter_Value"]); a[1] += 1; a[2] += dur.get(r["Dispatch_Id"], 0)
out = {"lib_source_hash": _lib._source_hash(),
       "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py --act-l2-only",
       "unit": "bytes per launch; FETCH_SIZE KiB -> B and x2 (gfx950 streaming-read correction); avg_us from the kernel trace of the same (profiled) run"}
for k, (v, n, ns) in acc.items():
    if "act_l2" in k or "colsum" in k:
        b = v / n * 1024 * 2
        out[k] = {"launches": n, "hbm_bytes_per_launch": round(b), "avg_us": round(ns / n / 1e3, 2), "hbm_gbps": round(b / (ns / n), 1)}
        if k == "act_l2_norms_kernel":
            algo = 512 * 197 * 3072 * 2          # bench.py --act-l2-only: ViT-B/16, 512 images
            out[k]["algorithmic_bytes"] = algo
            out[k]["hbm_gbps_algorithmic"] = round(algo / (ns / n), 1)
print(json.dumps(out, indent=1))
P
cat gpurun_out/pmc_act_l2.json
