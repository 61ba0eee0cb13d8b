#!/bin/bash
# Quick GPU check of a source state: the GPU suite, the default bench line and (optional) one A/B bench with extra environment.
#   bash scripts/gpu_check.sh TAG ["ENV=1 ENV2=2" for the A/B line]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-check}; AB=$2; O=gpurun_out
timeout -k 10 1000 python3 -m pytest tests -x -q -s -m gpu > $O/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/${TAG}_pytest_gpu.log
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 > $O/${TAG}_bench.jsonl 2> $O/${TAG}_bench.err || { echo "bench failed"; tail -5 $O/${TAG}_bench.err; exit 1; }
cut -c1-400 $O/${TAG}_bench.jsonl
if [ -n "$AB" ]; then
  env $AB timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-api --no-cpu-baseline > $O/${TAG}_bench_ab.jsonl 2>> $O/${TAG}_bench.err; echo "A/B ($AB) rc=$?"
  cut -c1-300 $O/${TAG}_bench_ab.jsonl
fi
