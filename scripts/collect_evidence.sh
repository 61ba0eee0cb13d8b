#!/bin/bash
# After `gpurun -- bash scripts/final_profile.sh TAG` has merged gpurun_out/ back: copy what is to be judged into profiles/ (tracked).
# The PMC summaries the script wrote into profiles/ on the GPU box do not travel back by themselves — only gpurun_out/ does.
#   bash scripts/collect_evidence.sh r04_zz
TAG=${1:?tag}
cd "$(dirname "$0")/.."
for f in bench.jsonl bench_kernel_stats.csv bench_variants.jsonl bench_config2_n1.jsonl other_models.jsonl pytest_gpu.log trace_gaps.json h14_bf16_kernel_stats.csv h14_fp8_kernel_stats.csv; do
  [ -f gpurun_out/${TAG}_$f ] && cp gpurun_out/${TAG}_$f profiles/${TAG}_$f
done
cp gpurun_out/${TAG}_pmc_*.json profiles/ 2>/dev/null
python3 - "$TAG" <<'PY'
import glob, json, sys
sys.path.insert(0, "2ssp-x-vit_amd")
from ssp2vit import _lib
h = _lib._source_hash()
bad = [f for f in sorted(glob.glob(f"profiles/{sys.argv[1]}_pmc_*.json")) if json.load(open(f)).get("lib_source_hash") != h]
print("source hash", h[:16], "| PMC summaries at another hash:", bad or "none")
sys.exit(1 if bad else 0)
PY
