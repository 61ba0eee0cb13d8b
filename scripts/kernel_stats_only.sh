#!/bin/bash
# rocprofv3 kernel stats + trace gaps of the default step alone (no secondary figures), TAG in the names
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r02_g}; O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG} -- python3 bench.py --no-cpu-baseline --no-api --no-overlap-figure --no-sustained > $O/${TAG}_bench_prof.log 2>&1
cp $(ls -t $O/prof_${TAG}/*/*_kernel_stats.csv | head -1) $O/${TAG}_bench_kernel_stats.csv
python3 scripts/trace_gaps.py $(ls -t $O/prof_${TAG}/*/*_kernel_trace.csv | head -1) --json $O/${TAG}_trace_gaps.json > /dev/null 2>&1
head -8 $O/${TAG}_bench_kernel_stats.csv | cut -c1-120; head -5 $O/${TAG}_trace_gaps.json
