#!/bin/bash
cd "$GRAFT_REPO_ROOT"
T=2ssp-x-vit_amd/csrc/tools
bash scripts/gemm_edge.sh || exit 1
for shape in "4096 256 128 30" "4100 768 768 30" "5000 320 256 31" "4100 3072 768 32" "4100 3072 768 33"; do timeout -k 5 100 $T/gemm_bench.bin $shape 5 | grep "fp8 epi"; done
for rep in 1 2 3; do
for shape in "63040 2304 768 10" "63040 768 768 11" "63040 768 3072 11" "63040 3072 768 12" "315200 3072 768 12" "65792 3840 1280 10"; do
  for b in gemm_bench gemm_oldwait; do
    echo -n "$b: "; timeout -k 5 200 $T/$b.bin $shape 30 | grep "median" || exit 1
  done
done
done
for s in "63040 2304 768 10" "63040 3072 768 12"; do timeout -k 5 120 $T/gemm_stamps.bin $s 10 2>&1 | grep "third tile\|epilogue of"; done
