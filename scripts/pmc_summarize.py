"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), per launch.

Units / corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM: both counters are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide coalesced streaming read (16 B/lane, global_load and LDS-DMA alike), so
it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.  Infinity-Cache hits are counted as traffic.
"""
import collections, csv, glob, json, os, sys


def load(d, counter):
    import os
    f = sorted(glob.glob(d + "/*/*_counter_collection.csv") or glob.glob(d + "/*_counter_collection.csv"), key=os.path.getmtime, reverse=True)
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter:
            key = (r["Kernel_Name"].split("(")[0], r.get("Grid_Size", r.get("Grid_Size_X", "")))
            out[key].append(float(r["Counter_Value"]))
    return out


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
rows = []
for k in fetch:
    f = sum(fetch[k]) / len(fetch[k]) * 1024 * 2          # KiB -> B, x2 gfx950 correction
    w = sum(write.get(k, [0])) / max(1, len(write.get(k, [0]))) * 1024
    rows.append({"kernel": k[0][-48:], "grid": k[1], "launches": len(fetch[k]), "fetch_bytes_per_launch": round(f),
                 "write_bytes_per_launch": round(w), "hbm_bytes_per_launch": round(f + w)})
rows.sort(key=lambda r: -r["hbm_bytes_per_launch"] * r["launches"])
fc1 = [r for r in rows if "gemm_bf16_kernel<2," in r["kernel"] or "gemm256_bf16_kernel<2" in r["kernel"]]   # EPI_FC1, both tile shapes
n = sum(r["launches"] for r in fc1)
avg = sum(r["launches"] * r["hbm_bytes_per_launch"] for r in fc1) / max(1, n)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "2ssp-x-vit_amd"))
from ssp2vit import _lib   # noqa: E402  (bench.py only trusts a summary recorded at the running library's source hash)
print(json.dumps({"lib_source_hash": _lib._source_hash(), "model": os.environ.get("PMC_MODEL", "vit_base_patch16_224"), "precision": os.environ.get("PMC_PRECISION", "bf16"), "unit": "bytes per launch; FETCH_SIZE doubled (gfx950), KiB->B; Infinity-Cache hits count as traffic",
                  "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 1 --warmup 1 (two passes)",
                  "fc1_family": {"launches": n, "avg_hbm_bytes_per_launch": round(avg)}, "kernels": rows[:14]}, indent=1))
