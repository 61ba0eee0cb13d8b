#!/bin/bash
# Experiment: residual-epilogue GEMM with every second workgroup of an XCD started late (gemm_stagger.bin = gemm_bench
# built with -DGEMM_STAGGER; 7th argument = delay in units of 1024 cycles).
cd "$GRAFT_REPO_ROOT"
B=2ssp-x-vit_amd/csrc/tools/gemm_stagger.bin
for shape in "63040 768 768" "630400 768 768" "63040 768 3072" "315200 768 3072"; do
  for d in 0 16 32 48 0; do
    [ "${shape##* }" = "3072" ] && [ $d -ne 0 ] && d=$((d * 3))
    echo "== $shape stagger $d"
    timeout -k 5 120 $B $shape 11 20 197 $d | tail -2 || exit 1
  done
done
