#!/bin/bash
cd "$GRAFT_REPO_ROOT"
T=2ssp-x-vit_amd/csrc/tools
for shape in "63040 2304 768 10" "63040 768 768 11" "63040 768 3072 11" "63040 3072 768 12" "102400 3072 768 13"; do
  timeout -k 5 120 $T/gemm_stamps.bin $shape 20 | grep -v "block \|verify\|slab" || exit 1
done
