#!/bin/bash
# L2 hit rate and HBM-side traffic of the persistent 256x256 GEMM on the bench's shapes (rocprofv3 PMC passes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=2ssp-x-vit_amd/csrc/tools/gemm_bench.bin
for shape in "63040 2304 768 10" "63040 768 3072 11" "63040 3072 768 12"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/gpmc_hit_$tag -- $B $shape 5 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/gpmc_fetch_$tag -- $B $shape 5 > /dev/null 2>&1
done
python3 - <<'P'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/gpmc_*')):
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            if 'gemm256' not in k: continue
            print(d.split('/')[-1], k, {c: sum(x) / len(x) for c, x in v.items()}, 'launches', len(next(iter(v.values()))))
P
