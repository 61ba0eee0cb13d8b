#!/bin/bash
# What the stage-1 hook costs the fc1 launch: gemm_bench on the hooked launch's shape (101 376 rows = 8 slabs of 64 ViT-B/16 images), sustained,
# epi 12 (bias + erf-GELU, swapped operands: the search's fc1) against 13 / 14 (the same + pre- / post-GELU hook, plain operand order), interleaved.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=2ssp-x-vit_amd/csrc/tools/gemm_bench.bin
O=gpurun_out/${1:-score_gap}.txt; : > $O
for r in 1 2 3; do
  for epi in 12 13 14; do
    GEMM_SUSTAIN=300 $B 101376 3072 768 $epi 5 197 0 2>&1 | grep -E "sustained|BAD|mismatch" | sed "s/^/round $r epi $epi: /" | tee -a $O
  done
done
