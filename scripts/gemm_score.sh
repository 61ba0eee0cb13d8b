#!/bin/bash
cd "$GRAFT_REPO_ROOT"
B=2ssp-x-vit_amd/csrc/tools/gemm_bench.bin
for shape in "102400 3072 768 13" "102400 3072 768 2" "102400 3072 768 14" "102400 3072 768 5" "12608 3072 768 13" "12608 3072 768 2" "25000 1984 768 13" "7168 768 192 14"; do
  timeout -k 5 120 $B $shape 20 || exit 1
done
