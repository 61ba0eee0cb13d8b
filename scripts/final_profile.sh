#!/bin/bash
# Round-end evidence on the GPU box: tests, the default bench line, rocprofv3 kernel stats of the same command, and the
# two PMC passes for HBM traffic.  Outputs under gpurun_out/ (copied to profiles/ by hand).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-e}
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_final_$TAG.log 2>&1; tail -2 gpurun_out/pytest_gpu_final_$TAG.log
timeout -k 10 400 python3 bench.py > gpurun_out/bench_final_$TAG.jsonl 2> gpurun_out/bench_final_$TAG.err || exit 1
cut -c1-300 gpurun_out/bench_final_$TAG.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final_$TAG -- python3 bench.py --no-cpu-baseline > gpurun_out/bench_prof_final_$TAG.log 2>&1
timeout -k 10 300 bash scripts/pmc_traffic.sh > gpurun_out/pmc_traffic_$TAG.log 2>&1; tail -3 gpurun_out/pmc_traffic_$TAG.log
