#!/bin/bash
# stream-overlap variants of the step on one box, alternating
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; rm -f $O/r02_streams.jsonl
for i in 1 2; do
  for v in "" "--two-streams" "--overlap-stage1"; do
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-api --no-roofline $v >> $O/r02_streams.jsonl 2>> $O/r02_streams.err; echo "bench [$v] rc=$?"
  done
done
python3 - <<'P'
import json
for l in open("gpurun_out/r02_streams.jsonl"):
    if l.startswith("{"):
        d = json.loads(l); print(d["ms_per_step"], d["streams"], d.get("stage1_beside_search"), d["search"], d["selected_blocks"])
P
