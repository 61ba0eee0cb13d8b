#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_yard -- python3 scripts/hipblaslt_yardstick.py > $O/r02_yardstick_prof.log 2>&1
F=$(ls -t $O/prof_yard/*/*_kernel_stats.csv | head -1)
cut -d, -f1-4 $F | head -20 > $O/r02_yardstick_kernels.txt; cat $O/r02_yardstick_kernels.txt
T=$(ls -t $O/prof_yard/*/*_kernel_trace.csv | head -1)
python3 - "$T" <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
seen = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"]
    if k.startswith("Cijk") or "gemm" in k.lower():
        key = (k, r.get("Grid_Size_X"), r.get("Workgroup_Size_X"), r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("Scratch_Size"))
        seen[key] = seen.get(key, 0) + 1
for k, n in seen.items():
    print(n, "x grid", k[1], "wg", k[2], "lds", k[3], "vgpr", k[4], "agpr", k[5], "sgpr", k[6], "\n   ", k[0][:400])
P
