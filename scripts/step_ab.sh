#!/bin/bash
# Same-box A/B of the whole step: the default against variants given as "ENV=.. ENV2=.." strings, interleaved, ROUNDS rounds each.
#   bash scripts/step_ab.sh TAG ROUNDS "SSP2_TAIL_SLOTS=0" "SSP2_OUT_OF_PLACE_START=0" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; R=$2; shift 2
O=gpurun_out/${TAG}_step_ab.txt; : > $O
for r in $(seq 1 $R); do
  for v in "" "$@"; do
    ms=$(env $v timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-api --no-cpu-baseline --no-roofline --no-overlap-figure 2>/dev/null | python3 -c "import sys, json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "round $r  ${v:-default}  $ms ms" | tee -a $O
  done
done
