#!/bin/bash
# Round-3 scoring epilogue (swapped operands + LDS pass sums) against the round-2 binary (plain operand order) on one box:
# bit-checks of the new kernels (256 x 256 vs 128 x 128, outputs and slab), then interleaved timings.
#   tools/gemm_bench_r02.bin = csrc/tools/gemm_bench.hip built from the round-2 tree (git worktree of c19432f..9dee565)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=2ssp-x-vit_amd/csrc/tools
for s in "9000 192 192 13" "9001 320 192 14" "4100 1984 768 13" "12608 768 3072 13 5 197" "25216 3072 768 13 5 197" "25216 3072 768 14 5 197" "33000 5120 1280 13 5 257" "4100 3072 768 33"; do
  echo "== $s"; timeout -k 5 120 $T/gemm_bench.bin $s | grep -E "verify|slab|fp8 epi|FAIL"
done
for rep in 1 2 3; do
  for s in "102400 3072 768 13" "102400 3072 768 14" "102400 3072 768 12" "63040 3072 768 12"; do
    for b in gemm_bench gemm_bench_r02; do
      [ -x $T/$b.bin ] || continue
      echo -n "$b: "; timeout -k 5 200 $T/$b.bin $s 30 | grep "median"
    done
  done
done
