#!/bin/bash
# Sustained (back-to-back) library-vs-ours GEMM rates with power / clock samples: scripts/sustained_yardstick.py, one call.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; TAG=${1:-r03}
ls /sys/class/drm/ > $O/${TAG}_sysfs_ls.txt 2>&1
timeout -k 10 900 python3 scripts/sustained_yardstick.py ${2:-2.5} > $O/${TAG}_sustained_yardstick.jsonl 2> $O/${TAG}_sustained_yardstick.err
echo "rc=$?"; cat $O/${TAG}_sustained_yardstick.jsonl; tail -5 $O/${TAG}_sustained_yardstick.err
