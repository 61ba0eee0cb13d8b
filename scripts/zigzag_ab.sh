#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; rm -f $O/r02_zig.jsonl
SSP2_ZIGZAG=1 timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "golden or full_size or edge_shapes or chunking or prefix_cache" > $O/r02_zig_pytest.log 2>&1; echo "pytest(zigzag) rc=$?"; tail -2 $O/r02_zig_pytest.log
for i in 1 2 3; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-api --no-roofline >> $O/r02_zig.jsonl 2>> $O/r02_zig.err
  SSP2_ZIGZAG=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-api --no-roofline >> $O/r02_zig.jsonl 2>> $O/r02_zig.err
  SSP2_ZIGZAG=ln timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-api --no-roofline >> $O/r02_zig.jsonl 2>> $O/r02_zig.err
done
python3 - <<'P'
import json
v=[json.loads(l)["ms_per_step"] for l in open("gpurun_out/r02_zig.jsonl") if l.startswith("{")]
print("default", v[0::3], "zigzag", v[1::3], "ln-only", v[2::3])
P
