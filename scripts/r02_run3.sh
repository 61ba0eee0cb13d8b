#!/bin/bash
cd "$GRAFT_REPO_ROOT"
B=2ssp-x-vit_amd/csrc/tools/gemm_bench.bin
timeout -k 10 60 2ssp-x-vit_amd/csrc/tools/fp8_probe.bin > gpurun_out/r02_fp8_probe.txt 2>&1; echo "probe rc=$?"; head -8 gpurun_out/r02_fp8_probe.txt
for shape in "4096 256 128 30" "4100 768 768 30" "5000 320 256 31" "4100 3072 768 32" "4100 3072 768 33" "63040 2304 768 30" "63040 768 3072 31" "63040 3072 768 32" "65792 3840 1280 30" "65792 5120 1280 32" "65792 1280 5120 31"; do
  timeout -k 5 200 $B $shape 10 2>&1 | tail -4
done > gpurun_out/r02_gemm_fp8.txt 2>&1
cat gpurun_out/r02_gemm_fp8.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r02_pytest_gpu3.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02_pytest_gpu3.log
tail -3 gpurun_out/r02_pytest_gpu3.log
timeout -k 10 200 python bench.py --act-l2-only > gpurun_out/r02_act_l2.json 2>/dev/null; cat gpurun_out/r02_act_l2.json
timeout -k 10 300 bash scripts/pmc_act_l2.sh > gpurun_out/r02_pmc_act_l2.log 2>&1; tail -14 gpurun_out/r02_pmc_act_l2.log
