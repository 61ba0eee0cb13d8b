#!/bin/bash
# A/B in one call: gemm_bench.bin (HEAD) against gemm_oldwait.bin (same source, -DGEMM_TILE_WAIT_STORES = round-1 tile-start wait)
cd "$GRAFT_REPO_ROOT"
T=2ssp-x-vit_amd/csrc/tools
for rep in 1 2 3; do
for shape in "63040 2304 768 10" "63040 768 768 11" "63040 768 3072 11" "63040 3072 768 12" "102400 3072 768 13" "315200 3072 768 12" "65792 3840 1280 10" "63040 2304 768 30" "63040 3072 768 32"; do
  for b in gemm_bench gemm_oldwait; do
    echo -n "$b: "; timeout -k 5 200 $T/$b.bin $shape 30 | grep "median" || exit 1
  done
done
done
for s in "63040 2304 768 10" "63040 768 3072 11" "63040 3072 768 12"; do timeout -k 5 120 $T/gemm_stamps.bin $s 10 2>&1 | grep "third tile\|persistent"; done
