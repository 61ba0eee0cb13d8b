#!/bin/bash
# Chunk-size sweep of the headline step on the GPU box (no CPU baseline): results to gpurun_out/sweep.jsonl
cd "$GRAFT_REPO_ROOT"
: > gpurun_out/sweep.jsonl
for cc in 64 192 256 320 512; do
  python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --calib-chunk $cc --eval-chunk 0 2>>gpurun_out/sweep.err | python3 -c "import sys,json; l=json.loads(sys.stdin.readline()); print(json.dumps({'calib_chunk':$cc,'eval_chunk':0,'ms':l['ms_per_step'],'calib_ips':l['calib_images_per_sec'],'tf':l['roofline']['achieved']}))" >> gpurun_out/sweep.jsonl || exit 1
done
for ec in 107 108 110 160 214 220; do
  python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --calib-chunk 256 --eval-chunk $ec 2>>gpurun_out/sweep.err | python3 -c "import sys,json; l=json.loads(sys.stdin.readline()); print(json.dumps({'calib_chunk':256,'eval_chunk':$ec,'ms':l['ms_per_step'],'calib_ips':l['calib_images_per_sec'],'tf':l['roofline']['achieved']}))" >> gpurun_out/sweep.jsonl || exit 1
done
cat gpurun_out/sweep.jsonl
