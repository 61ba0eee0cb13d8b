#!/bin/bash
# HBM traffic of the dominant kernels (run on the GPU box): two separate rocprofv3 --pmc passes (FETCH_SIZE and
# WRITE_SIZE do not fit one pass on gfx950), kernel-trace only.  Summarised by scripts/pmc_summarize.py.
#   PMC_MODEL / PMC_PRECISION / PMC_TARGET select another model (defaults: vit_base_patch16_224, bf16, 0.375); the summary
#   records them and bench.py only attaches a summary to a line of the same model and precision.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PMC_MODEL=${PMC_MODEL:-vit_base_patch16_224} PMC_PRECISION=${PMC_PRECISION:-bf16}
ARGS="--model $PMC_MODEL --precision $PMC_PRECISION --target ${PMC_TARGET:-0.375} --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-api"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py $ARGS > gpurun_out/pmc_$c.log 2>&1
done
python3 scripts/pmc_summarize.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/pmc_traffic.json
cat gpurun_out/pmc_traffic.json
