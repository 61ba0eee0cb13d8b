#!/bin/bash
# Experiment: XCD-level start stagger of the persistent GEMM (gemm_stagger.bin = gemm_bench built with -DGEMM_STAGGER;
# 7th argument = 100 x U: XCD group x starts x * U * 256 cycles late).
cd "$GRAFT_REPO_ROOT"
B=2ssp-x-vit_amd/csrc/tools/gemm_stagger.bin
for rep in 1 2; do
for shape in "63040 2304 768 10" "63040 3072 768 12" "63040 768 768 11" "63040 768 3072 11" "315200 3072 768 12"; do
  for u in 0 4 8 16 24 32 0; do
    echo -n "stagger $u: "; timeout -k 5 120 $B $shape 30 197 $((u * 100)) | grep median || exit 1
  done
done
done
