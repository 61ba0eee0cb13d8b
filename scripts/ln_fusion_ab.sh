#!/bin/bash
# LayerNorm fused behind the residual GEMMs: tool-level bit checks + timing, GPU tests, step A/B; act_l2 NT loads A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
B=2ssp-x-vit_amd/csrc/tools/gemm_bench.bin
{
for sh in "63040 768 768" "63040 768 3072" "5000 768 768" "300 768 768" "100864 768 768" "63040 1024 1024" "16448 1280 1280" "16448 1280 5120"; do
  echo "== $sh epi 15 (resid + LN)"; timeout -k 10 120 $B $sh 15 20 || echo "rc=$?"
  echo "== $sh epi 11 (resid)"; timeout -k 10 120 $B $sh 11 20 || echo "rc=$?"
done
} > $O/r02_gemm_ln.txt 2>&1
grep -c "FAIL" $O/r02_gemm_ln.txt; grep "FAIL" $O/r02_gemm_ln.txt | head
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -s > $O/r02_pytest_gpu5.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r02_pytest_gpu5.log
for i in 1 2; do
  SSP2_LN_FUSION=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-api --no-roofline >> $O/r02_bench_ln.jsonl 2>> $O/r02_bench_ln.err
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-api --no-roofline >> $O/r02_bench_noln.jsonl 2>> $O/r02_bench_ln.err
done
python3 - <<'P'
import json
for f in ("gpurun_out/r02_bench_ln.jsonl", "gpurun_out/r02_bench_noln.jsonl"):
    print(f, [json.loads(l)["ms_per_step"] for l in open(f) if l.startswith("{")])
P
timeout -k 10 200 python3 bench.py --act-l2-only > $O/r02_act_l2_now.json 2>&1
tail -1 $O/r02_act_l2_now.json | cut -c1-250
