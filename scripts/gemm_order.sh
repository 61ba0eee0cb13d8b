#!/bin/bash
# Experiment: tile order of the persistent GEMM — blocks of GM row panels x column groups of GN tiles
# (7th argument of gemm_bench = 100 * GM + GN; 0 = plain N-fastest order).
cd "$GRAFT_REPO_ROOT"
B=2ssp-x-vit_amd/csrc/tools/gemm_bench.bin
run() { echo "== $*"; timeout -k 5 120 $B $* | tail -2 | cut -c1-160 || exit 1; }
for o in 0 4 6404 12804 6403 6406 3204 0; do run 63040 3072 768 12 20 197 $o; done
for o in 0 4 6404 12804 6406 0; do run 315200 3072 768 12 20 197 $o; done
for o in 0 6404 0 6404; do run 102400 3072 768 13 20 197 $o; done
for o in 0 3 6403 8603 6405 0; do run 63040 2304 768 10 20 197 $o; done
for o in 0 6403 8603 0; do run 315200 2304 768 10 20 197 $o; done
