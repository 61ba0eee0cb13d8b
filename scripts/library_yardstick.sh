#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; B=2ssp-x-vit_amd/csrc/tools/gemm_bench.bin
timeout -k 10 300 python3 scripts/hipblaslt_yardstick.py > $O/r02_yardstick.jsonl 2> $O/r02_yardstick.err; echo "rc=$?"; cat $O/r02_yardstick.jsonl
{
for sh in "63040 2304 768 10" "63040 768 768 11" "63040 3072 768 12" "63040 768 3072 11" "315200 3072 768 12" "82240 3840 1280 10" "82240 5120 1280 12"; do
  timeout -k 10 120 $B $sh 20 | grep "median"
done
} > $O/r02_yardstick_ours.txt 2>&1; cat $O/r02_yardstick_ours.txt
