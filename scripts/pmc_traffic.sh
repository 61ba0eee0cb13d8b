#!/bin/bash
# HBM traffic of the dominant kernels (run on the GPU box): two separate rocprofv3 --pmc passes (FETCH_SIZE and
# WRITE_SIZE do not fit one pass on gfx950), kernel-trace only.  Summarised by scripts/pmc_summarize.py.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-api > gpurun_out/pmc_$c.log 2>&1
done
python3 scripts/pmc_summarize.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/pmc_traffic.json
cat gpurun_out/pmc_traffic.json
