#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; rm -f $O/r02_streams2.jsonl
for v in "" "--overlap-stage1" "" "--overlap-stage1"; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-api $v >> $O/r02_streams2.jsonl 2>> $O/r02_streams2.err; echo "bench [$v] rc=$?"
done
python3 - <<'P'
import json
for l in open("gpurun_out/r02_streams2.jsonl"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]; print(d["ms_per_step"], d["streams"], r["achieved"], r["frac"], r["launches"], r["avg_launch_us"])
P
