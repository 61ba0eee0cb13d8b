#!/bin/bash
# round-2 GPU pass: tests, default bench line, input-path variants
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r02_pytest_gpu2.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02_pytest_gpu2.log
tail -3 gpurun_out/r02_pytest_gpu2.log
timeout -k 10 500 python bench.py > gpurun_out/r02_bench_a.jsonl 2> gpurun_out/r02_bench_a.err; echo "bench rc=$?"
cut -c1-400 gpurun_out/r02_bench_a.jsonl
for v in "--host-inputs" "--uint8" "--host-inputs --uint8"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-api --no-roofline $v >> gpurun_out/r02_bench_inputs.jsonl 2>> gpurun_out/r02_bench_inputs.err; echo "bench $v rc=$?"
done
cut -c1-200 gpurun_out/r02_bench_inputs.jsonl
