#!/bin/bash
cd "$GRAFT_REPO_ROOT"
B=2ssp-x-vit_amd/csrc/tools/gemm_bench.bin
for rep in 1 2; do
for shape in "63040 768 768 11" "63040 768 768 21" "63040 768 3072 11" "63040 768 3072 21" "63000 768 768 11" "12608 768 3072 11" "12608 768 3072 21" "7000 192 192 11" "102400 768 768 11"; do
  timeout -k 5 120 $B $shape 30 || exit 1
done
done
