#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; T=2ssp-x-vit_amd/csrc/tools
{
for sh in "82240 1280 1280" "82240 1280 1280" "41120 1280 1280" "20000 1280 1280" "82240 1280 768"; do
  echo "== $sh"; timeout -k 10 200 $T/gemm_bench.bin $sh 15 3 | grep -v "^  out\|^  x\[\|^  h\[" | head -40
done
} > $O/r03_q5_ln_dbg.txt 2>&1; cat $O/r03_q5_ln_dbg.txt
