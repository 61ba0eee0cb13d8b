#!/bin/bash
# A/B of two builds of csrc/tools/gemm_bench on the bench's shapes, interleaved, three rounds, medians only.
#   bash scripts/gemm_ab.sh [BIN_A [BIN_B ["M N K epi" ...]]]
# BIN_*: names of binaries under csrc/tools (without .bin; build variants with ssp2vit._lib.build_tool(name, defines, out)).
# Defaults: gemm_bench against gemm_oldwait (same source, -DGEMM_TILE_WAIT_STORES) on QKV / out-proj / fc2 / fc1 / scoring fc1
# / a layer-major fc1 / ViT-H QKV / the two fp8 shapes.  (Rounds 1-2 kept four near-identical copies of this loop.)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
T=2ssp-x-vit_amd/csrc/tools
A=${1:-gemm_bench}; B=${2:-gemm_oldwait}; shift 2 2>/dev/null
if [ $# -eq 0 ]; then
  set -- "63040 2304 768 10" "63040 768 768 11" "63040 768 3072 11" "63040 3072 768 12" "102400 3072 768 13" "315200 3072 768 12" \
         "65792 3840 1280 10" "63040 2304 768 30" "63040 3072 768 32"
fi
for rep in 1 2 3; do
  for shape in "$@"; do
    for b in $A $B; do
      [ -x $T/$b.bin ] || continue
      echo -n "$b: "; timeout -k 5 200 $T/$b.bin $shape 30 | grep "median" || exit 1
    done
  done
done
