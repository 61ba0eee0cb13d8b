#!/bin/bash
# A/B of the persistent 256x256 GEMM (epi 10/11/12) against its round-1 version (20/21/22) on the bench's shapes.
cd "$GRAFT_REPO_ROOT"
B=2ssp-x-vit_amd/csrc/tools/gemm_bench.bin
for shape in "63040 2304 768 0" "63040 768 768 1" "63040 768 3072 1" "63040 3072 768 2" "12608 2304 768 0" "12608 768 3072 1" "21276 768 768 1" "63000 768 768 1" "63040 1984 768 2" "63040 768 1984 1"; do
  set -- $shape
  for base in 10 20; do
    timeout -k 5 120 $B $1 $2 $3 $((base + $4)) 30 || exit 1
  done
done
