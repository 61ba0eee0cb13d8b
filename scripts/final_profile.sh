#!/bin/bash
# Round-end evidence on the GPU box at ONE source state: GPU tests, the default bench line, rocprofv3 kernel stats of the
# same command, the PMC passes (HBM traffic of the GEMMs, matrix-pipe utilisation, the activation-L2 kernel), the
# input-path variants, BASELINE configs[2] at N = 1 and configs[4] (ViT-H/14, bf16 and fp8).  Outputs under gpurun_out/
# (TAG in the names); copy what is to be judged into profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r02_b}
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -s > $O/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/${TAG}_pytest_gpu.log; tail -2 $O/${TAG}_pytest_gpu.log
# PMC summaries first: the bench line below then carries traffic / pmc from THIS source hash
timeout -k 10 300 bash scripts/pmc_traffic.sh > $O/${TAG}_pmc_traffic.log 2>&1; cp $O/pmc_traffic.json $O/${TAG}_pmc_traffic.json; mkdir -p profiles; cp $O/pmc_traffic.json profiles/${TAG}_pmc_traffic.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-api > $O/${TAG}_pmc_mfma.log 2>&1
python3 scripts/pmc_mfma.py $O/pmc_mfma > profiles/${TAG}_pmc_mfma.json 2>> $O/${TAG}_pmc_mfma.log; cp profiles/${TAG}_pmc_mfma.json $O/
timeout -k 10 300 bash scripts/pmc_act_l2.sh > $O/${TAG}_pmc_act_l2.log 2>&1; cp $O/pmc_act_l2.json profiles/${TAG}_pmc_act_l2.json; cp $O/pmc_act_l2.json $O/${TAG}_pmc_act_l2.json
timeout -k 10 500 python3 bench.py > $O/${TAG}_bench.jsonl 2> $O/${TAG}_bench.err || exit 1
cut -c1-300 $O/${TAG}_bench.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG} -- python3 bench.py --no-cpu-baseline --no-api --no-overlap-figure > $O/${TAG}_bench_prof.log 2>&1
cp $(ls -t $O/prof_${TAG}/*/*_kernel_stats.csv | head -1) $O/${TAG}_bench_kernel_stats.csv
python3 scripts/trace_gaps.py $(ls -t $O/prof_${TAG}/*/*_kernel_trace.csv | head -1) --json $O/${TAG}_trace_gaps.json > /dev/null 2>&1
for v in "--host-inputs" "--uint8" "--host-inputs --uint8" "--two-streams"; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-api --no-roofline $v >> $O/${TAG}_bench_variants.jsonl 2>> $O/${TAG}_bench_variants.err; echo "bench $v rc=$?"
done
timeout -k 10 400 python3 bench.py --config 2 --no-cpu-baseline --no-roofline --steps 2 > $O/${TAG}_bench_config2_n1.jsonl 2>> $O/${TAG}_bench_variants.err; echo "config2 rc=$?"
for m in "vit_huge_patch14_224 0.5 bf16" "vit_huge_patch14_224 0.5 fp8" "vit_large_patch16_224 0.375 bf16" "vit_small_patch16_224 0.375 bf16"; do
  set -- $m
  timeout -k 10 500 python3 bench.py --model $1 --target $2 --precision $3 --steps 2 --warmup 1 --no-api --no-cpu-baseline --no-overlap-figure >> $O/${TAG}_other_models.jsonl 2>> $O/${TAG}_bench_variants.err; echo "$m rc=$?"
done
