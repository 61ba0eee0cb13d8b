#!/bin/bash
# Is fc2's main loop (K = 3072, A = the 387 MB activation) bound by the A stream's HBM latency?  Same K, shrinking A.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=2ssp-x-vit_amd/csrc/tools/gemm_bench.bin
for sh in "63040 768 3072 11" "63040 768 3072 10" "32768 768 3072 11" "32768 768 3072 10" "21760 768 3072 10" "63040 768 768 10" "63040 3072 3072 10" "63040 2304 768 10"; do
  timeout -k 10 120 $B $sh 20 | grep "median"
done
