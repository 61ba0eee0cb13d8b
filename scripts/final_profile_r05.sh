#!/bin/bash
# Round-5 evidence on the GPU box at ONE source state, in parts that each fit one gpurun call (<= 1200 s):
#   bash scripts/final_profile_r05.sh TAG a   GPU tests, PMC passes of ViT-B/16 (HBM traffic, matrix pipe, activation-L2), the driver's bench line
#                                             (20 steps, 3 warm-up), rocprofv3 kernel stats + trace gaps of the same command
#   bash scripts/final_profile_r05.sh TAG b   the same step as two passes (A/B), input-path variants, BASELINE configs[2] at N = 1, a two-rank
#                                             one-card rehearsal of configs[2], ViT-L/16 / ViT-S/16, BASELINE configs[3]
#   bash scripts/final_profile_r05.sh TAG c   ViT-H/14 (configs[4]): bf16 / fp8 at 512 and 4096 calibration images, PMC + kernel stats
# then here: bash scripts/collect_evidence.sh TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r05_z}; PART=${2:-a}
O=gpurun_out
mkdir -p profiles
if [ "$PART" = a ]; then
  timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -s > $O/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/${TAG}_pytest_gpu.log; tail -2 $O/${TAG}_pytest_gpu.log
  timeout -k 10 200 bash scripts/pmc_traffic.sh > $O/${TAG}_pmc_traffic.log 2>&1; cp $O/pmc_traffic.json profiles/${TAG}_pmc_traffic.json
  timeout -k 10 200 bash scripts/pmc_mfma.sh profiles/${TAG}_pmc_mfma.json > $O/${TAG}_pmc_mfma.log 2>&1
  timeout -k 10 150 bash scripts/pmc_act_l2.sh > $O/${TAG}_pmc_act_l2.log 2>&1; cp $O/pmc_act_l2.json profiles/${TAG}_pmc_act_l2.json
  cp profiles/${TAG}_pmc_*.json $O/
  timeout -k 10 400 python3 bench.py --steps 20 --warmup 3 > $O/${TAG}_bench.jsonl 2> $O/${TAG}_bench.err || exit 1
  cut -c1-300 $O/${TAG}_bench.jsonl
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG} -- python3 bench.py --no-cpu-baseline --no-api --no-overlap-figure --no-sustained > $O/${TAG}_bench_prof.log 2>&1
  cp $(ls -t $O/prof_${TAG}/*/*_kernel_stats.csv | head -1) $O/${TAG}_bench_kernel_stats.csv
  python3 scripts/trace_gaps.py $(ls -t $O/prof_${TAG}/*/*_kernel_trace.csv | head -1) --json $O/${TAG}_trace_gaps.json > /dev/null 2>&1
  rm -rf $O/prof_${TAG} $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
fi
if [ "$PART" = b ]; then
  rm -f $O/${TAG}_bench_variants.jsonl
  for v in "--two-pass" "" "--two-pass" "" "--host-inputs" "--uint8" "--host-inputs --uint8"; do
    timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-api --no-roofline $v >> $O/${TAG}_bench_variants.jsonl 2>> $O/${TAG}_bench_variants.err; echo "bench $v rc=$?"
  done
  timeout -k 10 300 python3 bench.py --config 2 --no-cpu-baseline --no-roofline --steps 2 > $O/${TAG}_bench_config2_n1.jsonl 2>> $O/${TAG}_bench_variants.err; echo "config2 rc=$?"
  SSP2_REHEARSE_ONE_CARD=1 timeout -k 10 300 python3 bench.py --gpus 2 --config 2 --no-cpu-baseline --no-roofline --steps 1 --warmup 1 > $O/${TAG}_rehearsal_config2_2ranks.jsonl 2> $O/${TAG}_rehearsal_config2_2ranks.err; echo "rehearsal rc=$?"
  bash scripts/other_models.sh ${TAG} "vit_large_patch16_224 0.375 bf16" "vit_small_patch16_224 0.375 bf16"
  timeout -k 10 300 python3 scripts/bench_config3.py > $O/${TAG}_bench_config3.jsonl 2> $O/${TAG}_bench_config3.err; echo "config3 rc=$?"
fi
if [ "$PART" = c ]; then
  bash scripts/other_models.sh ${TAG}_h14 "vit_huge_patch14_224 0.5 bf16" "vit_huge_patch14_224 0.5 fp8" "vit_huge_patch14_224 0.5 bf16 4096" "vit_huge_patch14_224 0.5 fp8 4096"
  PMC_MODEL=vit_huge_patch14_224 PMC_TARGET=0.5 PMC_PRECISION=fp8 timeout -k 10 300 bash scripts/pmc_mfma.sh profiles/${TAG}_pmc_mfma_h14_fp8.json > $O/${TAG}_pmc_mfma_h14_fp8.log 2>&1
  cp profiles/${TAG}_pmc_*.json $O/
  for prec in bf16 fp8; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_h14_${prec} -- python3 bench.py --model vit_huge_patch14_224 --target 0.5 --precision ${prec} --steps 1 --warmup 1 --no-cpu-baseline --no-api --no-overlap-figure --no-sustained > $O/${TAG}_bench_prof_h14_${prec}.log 2>&1
    cp $(ls -t $O/prof_${TAG}_h14_${prec}/*/*_kernel_stats.csv | head -1) $O/${TAG}_h14_${prec}_kernel_stats.csv; echo "h14 ${prec} kernel stats rc=$?"
    rm -rf $O/prof_${TAG}_h14_${prec}
  done
fi
