#!/bin/bash
cd "$GRAFT_REPO_ROOT"
B=2ssp-x-vit_amd/csrc/tools/gemm_bench.bin
for shape in "4096 256 128 30" "4100 768 768 30" "5000 320 256 31" "4100 3072 768 33" "63040 2304 768 30" "63040 768 3072 31" "65792 1280 5120 31"; do
  timeout -k 5 200 $B $shape 10 2>&1 | tail -3
done > gpurun_out/r02_gemm_fp8.txt 2>&1
grep -c "(ok)" gpurun_out/r02_gemm_fp8.txt; grep "FAIL" gpurun_out/r02_gemm_fp8.txt | head -5
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r02_pytest_gpu4.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02_pytest_gpu4.log
tail -3 gpurun_out/r02_pytest_gpu4.log; grep "\[fp8\]" gpurun_out/r02_pytest_gpu4.log
timeout -k 10 400 python bench.py --config 2 --no-cpu-baseline --no-roofline --steps 2 > gpurun_out/r02_bench_config2_n1.jsonl 2> gpurun_out/r02_bench_config2_n1.err; echo "config2 rc=$?"; cut -c1-300 gpurun_out/r02_bench_config2_n1.jsonl
for prec in bf16 fp8; do
  timeout -k 10 500 python bench.py --model vit_huge_patch14_224 --target 0.5 --precision $prec --steps 2 --warmup 1 --no-api --no-cpu-baseline >> gpurun_out/r02_bench_h14.jsonl 2>> gpurun_out/r02_bench_h14.err; echo "h14 $prec rc=$?"
done
cut -c1-260 gpurun_out/r02_bench_h14.jsonl
