#!/bin/bash
# Matrix-pipe utilisation in cycles and the shader clock per kernel: ONE rocprofv3 --pmc pass (its own run: no other trace domain)
# over a one-step bench, summarised by scripts/pmc_mfma.py.   bash scripts/pmc_mfma.sh OUT.json   (PMC_MODEL / PMC_PRECISION / PMC_TARGET as in pmc_traffic.sh)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PMC_MODEL=${PMC_MODEL:-vit_base_patch16_224} PMC_PRECISION=${PMC_PRECISION:-bf16}
ARGS="--model $PMC_MODEL --precision $PMC_PRECISION --target ${PMC_TARGET:-0.375} --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-api"
rm -rf gpurun_out/pmc_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -- python3 bench.py $ARGS > gpurun_out/pmc_mfma.log 2>&1
python3 scripts/pmc_mfma.py gpurun_out/pmc_mfma > "$1"
python3 - "$1" <<'P'
import json, sys
j = json.load(open(sys.argv[1]))
print(j["model"], j["precision"], [(k["kernel"][-34:], k["mfma_util_of_cycles"], k["shader_clock_ghz"]) for k in j["kernels"][:5]])
P
