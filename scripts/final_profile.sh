#!/bin/bash
# Round-end evidence on the GPU box at ONE source state: GPU tests, the PMC passes (HBM traffic of the GEMMs, matrix-pipe
# utilisation — ViT-B/16 bf16 AND ViT-H/14 bf16 / fp8, each summary keyed on model + precision + source hash), the default bench
# line, rocprofv3 kernel stats of the same command, the input-path variants, BASELINE configs[2] at N = 1 and the other models.
# Outputs under gpurun_out/ (TAG in the names) and, for what is to be judged, profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r03_z}
O=gpurun_out
mkdir -p profiles
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -s > $O/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/${TAG}_pytest_gpu.log; tail -2 $O/${TAG}_pytest_gpu.log
# PMC summaries first: the bench lines below then carry traffic / pmc from THIS source hash
timeout -k 10 400 bash scripts/pmc_traffic.sh > $O/${TAG}_pmc_traffic.log 2>&1; cp $O/pmc_traffic.json profiles/${TAG}_pmc_traffic.json
timeout -k 10 300 bash scripts/pmc_mfma.sh profiles/${TAG}_pmc_mfma.json
PMC_MODEL=vit_huge_patch14_224 PMC_TARGET=0.5 PMC_PRECISION=bf16 timeout -k 10 400 bash scripts/pmc_mfma.sh profiles/${TAG}_pmc_mfma_h14_bf16.json
PMC_MODEL=vit_huge_patch14_224 PMC_TARGET=0.5 PMC_PRECISION=fp8 timeout -k 10 400 bash scripts/pmc_mfma.sh profiles/${TAG}_pmc_mfma_h14_fp8.json
PMC_MODEL=vit_huge_patch14_224 PMC_TARGET=0.5 PMC_PRECISION=fp8 timeout -k 10 500 bash scripts/pmc_traffic.sh > $O/${TAG}_pmc_traffic_h14_fp8.log 2>&1; cp $O/pmc_traffic.json profiles/${TAG}_pmc_traffic_h14_fp8.json
timeout -k 10 300 bash scripts/pmc_act_l2.sh > $O/${TAG}_pmc_act_l2.log 2>&1; cp $O/pmc_act_l2.json profiles/${TAG}_pmc_act_l2.json
cp profiles/${TAG}_pmc_*.json $O/
timeout -k 10 500 python3 bench.py > $O/${TAG}_bench.jsonl 2> $O/${TAG}_bench.err || exit 1
cut -c1-300 $O/${TAG}_bench.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG} -- python3 bench.py --no-cpu-baseline --no-api --no-overlap-figure --no-sustained > $O/${TAG}_bench_prof.log 2>&1
cp $(ls -t $O/prof_${TAG}/*/*_kernel_stats.csv | head -1) $O/${TAG}_bench_kernel_stats.csv
python3 scripts/trace_gaps.py $(ls -t $O/prof_${TAG}/*/*_kernel_trace.csv | head -1) --json $O/${TAG}_trace_gaps.json > /dev/null 2>&1
for v in "--host-inputs" "--uint8" "--host-inputs --uint8" "--two-streams"; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-api --no-roofline $v >> $O/${TAG}_bench_variants.jsonl 2>> $O/${TAG}_bench_variants.err; echo "bench $v rc=$?"
done
timeout -k 10 400 python3 bench.py --config 2 --no-cpu-baseline --no-roofline --steps 2 > $O/${TAG}_bench_config2_n1.jsonl 2>> $O/${TAG}_bench_variants.err; echo "config2 rc=$?"
bash scripts/other_models.sh ${TAG}
# rocprofv3 kernel stats of the configs[4] shard too (ViT-H/14, bf16 and fp8): the per-kernel averages behind the H/14 claims
for prec in bf16 fp8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_h14_${prec} -- python3 bench.py --model vit_huge_patch14_224 --target 0.5 --precision ${prec} --steps 1 --warmup 1 --no-cpu-baseline --no-api --no-overlap-figure --no-sustained > $O/${TAG}_bench_prof_h14_${prec}.log 2>&1
  cp $(ls -t $O/prof_${TAG}_h14_${prec}/*/*_kernel_stats.csv | head -1) $O/${TAG}_h14_${prec}_kernel_stats.csv; echo "h14 ${prec} kernel stats rc=$?"
done
