#!/bin/bash
# Where the fused LayerNorm's time goes: epi 15 (resid + LN), the same without the LN phase (gemm_noln: queues + arrivals only), epi 11.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r03_q2}; O=gpurun_out; T=2ssp-x-vit_amd/csrc/tools
{
for sh in "63040 768 768" "63040 768 3072" "100864 768 768" "315200 768 768" "315200 768 3072" "630400 768 768" "82240 1280 1280" "82240 1280 5120"; do
  n=400; case "$sh" in 3152*|6304*) n=100;; esac
  for b in "gemm_bench 15" "gemm_noln 15" "gemm_bench 11"; do
    set -- $b
    echo -n "$sh $1 epi $2: "; GEMM_SUSTAIN=$n timeout -k 10 200 $T/$1.bin $sh $2 5 | grep -E "sustained|FAIL" || { echo "rc=$?"; exit 1; }
  done
done
} > $O/${TAG}_ln_probe.txt 2>&1; rc=$?; cat $O/${TAG}_ln_probe.txt; exit $rc
