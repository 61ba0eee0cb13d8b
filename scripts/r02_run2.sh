#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r02_pytest_gpu3.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02_pytest_gpu3.log
tail -3 gpurun_out/r02_pytest_gpu3.log
timeout -k 10 60 2ssp-x-vit_amd/csrc/tools/fp8_probe.bin > gpurun_out/r02_fp8_probe.txt 2>&1; echo "probe rc=$?"; cat gpurun_out/r02_fp8_probe.txt
timeout -k 10 200 python bench.py --act-l2-only > gpurun_out/r02_act_l2.json 2>/dev/null; cat gpurun_out/r02_act_l2.json
timeout -k 10 300 bash scripts/pmc_act_l2.sh > gpurun_out/r02_pmc_act_l2.log 2>&1; tail -12 gpurun_out/r02_pmc_act_l2.log
