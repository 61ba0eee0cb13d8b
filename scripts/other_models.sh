#!/bin/bash
# Secondary models / precisions with the per-family breakdown (roofline_by_family): ViT-H/14 (configs[4] shard) bf16 and fp8, ViT-L/16, ViT-S/16.
#   bash scripts/other_models.sh TAG ["model target precision" ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r03}; shift
# (a fourth field = calibration images per GPU; BASELINE configs[4] states 4096 for ViT-H/14)
[ $# -eq 0 ] && set -- "vit_huge_patch14_224 0.5 bf16" "vit_huge_patch14_224 0.5 fp8" "vit_huge_patch14_224 0.5 bf16 4096" "vit_huge_patch14_224 0.5 fp8 4096" "vit_large_patch16_224 0.375 bf16" "vit_small_patch16_224 0.375 bf16"
O=gpurun_out; rm -f $O/${TAG}_other_models.jsonl
for m in "$@"; do
  set -- $m
  timeout -k 10 500 python3 bench.py --model $1 --target $2 --precision $3 --calib ${4:-512} --steps 2 --warmup 1 --no-api --no-cpu-baseline --no-overlap-figure --no-sustained >> $O/${TAG}_other_models.jsonl 2>> $O/${TAG}_other_models.err; echo "$m rc=$?"
done
python3 - <<P
import json
for l in open("$O/${TAG}_other_models.jsonl"):
    if not l.startswith("{"): continue
    j = json.loads(l); f = j.get("roofline_by_family", {})
    print(j["config"]["workload"][:40], j["dtype"][:4], j["ms_per_step"], "ms |", " ".join(f"{k}:{v['ms']:.1f}ms/{v.get('achieved')}" for k, v in f.items() if isinstance(v, dict)))
P
