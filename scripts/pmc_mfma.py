"""Per-kernel matrix-pipe utilisation in CYCLES and the shader clock, from one rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_WAVE_CYCLES, SQ_ACTIVE_INST_VALU, SQ_WAIT_INST_ANY) joined with the
kernel trace of the same run (durations).

    python3 scripts/pmc_mfma.py gpurun_out/pmc_mfma > profiles/rNN_x_pmc_mfma.json

SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs (32 per v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md
'rocprofv3 PMC slots'); GRBM_GUI_ACTIVE counts the shader-clock cycles the dispatch was active (reported here as is AND
divided by the XCD count, since the value is summed over the 8 XCDs when it exceeds duration x 2.5 GHz).
mfma_util = MFMA_BUSY / (active cycles x 256 CUs x 4 SIMDs); clock = active cycles / duration.
"""
import collections, csv, glob, json, os, sys

d = sys.argv[1]
cc = sorted(glob.glob(d + "/*/*_counter_collection.csv"), key=os.path.getmtime)[-1]
kt = sorted(glob.glob(d + "/*/*_kernel_trace.csv"), key=os.path.getmtime)[-1]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
per = collections.defaultdict(lambda: collections.defaultdict(float))
seen = collections.defaultdict(set)
for r in csv.DictReader(open(cc)):
    k = r["Kernel_Name"].split("(")[0][-44:]
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen[k]:
        seen[k].add(r["Dispatch_Id"])
        per[k]["_ns"] += dur.get(r["Dispatch_Id"], 0)
rows = []
for k, c in per.items():
    ns = c["_ns"]
    if ns <= 0 or c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0:
        continue
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    xcd = 8 if gui > ns * 2.6 else 1                      # summed over the XCDs?
    cyc = gui / xcd
    rows.append({"kernel": k, "launches": len(seen[k]), "total_ms": round(ns / 1e6, 3),
                 "mfma_busy_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"], "gui_active": gui, "gui_active_divisor": xcd,
                 "shader_clock_ghz": round(cyc / ns, 3) if cyc else None,
                 "mfma_util_of_cycles": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 4) if cyc else None,
                 "mfma_busy_per_ns_per_simd": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / ns / 1024, 4),
                 "other": {n: v for n, v in c.items() if n not in ("_ns", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")}})
rows.sort(key=lambda r: -r["total_ms"])
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "2ssp-x-vit_amd"))
from ssp2vit import _lib   # noqa: E402  (bench.py only trusts a summary recorded at the running library's source hash)
print(json.dumps({"lib_source_hash": _lib._source_hash(), "model": os.environ.get("PMC_MODEL", "vit_base_patch16_224"), "precision": os.environ.get("PMC_PRECISION", "bf16"), "command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline",
                  "note": "mfma_busy_per_ns_per_simd x (1 / 2.4) = fraction of the 2.4 GHz peak rate; mfma_util_of_cycles is relative to the cycles the part actually ran",
                  "kernels": rows[:12]}, indent=1))
