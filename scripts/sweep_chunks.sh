#!/bin/bash
# Chunk-size sweep of the headline step on the GPU box (no CPU baseline): results to gpurun_out/sweep.jsonl
cd "$GRAFT_REPO_ROOT"
: > gpurun_out/sweep.jsonl
run() {
  python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --calib-chunk $1 --eval-chunk $2 2>>gpurun_out/sweep.err | python3 -c "import sys,json; l=json.loads(sys.stdin.readline()); print(json.dumps({'calib_chunk':$1,'eval_chunk':$2,'ms':l['ms_per_step'],'calib_ips':l['calib_images_per_sec'],'tf':l['roofline']['achieved']}))" >> gpurun_out/sweep.jsonl || exit 1
}
for cc in 64 256 512; do run $cc 0; done
for ec in 107 160 214; do run 512 $ec; done
cat gpurun_out/sweep.jsonl
