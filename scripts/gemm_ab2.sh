#!/bin/bash
cd "$GRAFT_REPO_ROOT"
T=2ssp-x-vit_amd/csrc/tools
for rep in 1 2 3; do
for v in gemm_bench gemm_bench_old; do
  echo "== $v"
  for shape in "63040 2304 768 10" "63040 768 3072 11" "63040 3072 768 12" "63040 768 768 11"; do
    timeout -k 5 120 $T/$v.bin $shape 30 | grep -v "verify\|^  " || exit 1
  done
done
done
