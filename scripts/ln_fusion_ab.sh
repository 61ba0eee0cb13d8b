#!/bin/bash
# LayerNorm fused behind the residual GEMMs (last-arriver form): XCC_ID probe, tool-level bit checks + timing, the GPU test, step A/B.
#   bash scripts/ln_fusion_ab.sh TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r03_q}; O=gpurun_out
T=2ssp-x-vit_amd/csrc/tools
timeout -k 5 60 $T/xcc_probe.bin 1024 > $O/${TAG}_xcc_probe.txt 2>&1; echo "probe rc=$?"; cat $O/${TAG}_xcc_probe.txt
{
for sh in "63040 768 768" "63040 768 3072" "5000 768 768" "300 768 768" "100864 768 768" "63040 1024 1024" "16448 1280 1280" "16448 1280 5120"; do
  echo "== $sh epi 15 (resid + LN)"; GEMM_SUSTAIN=400 timeout -k 10 120 $T/gemm_bench.bin $sh 15 20 || { echo "rc=$?"; exit 1; }
  echo "== $sh epi 11 (resid)"; GEMM_SUSTAIN=400 timeout -k 10 120 $T/gemm_bench.bin $sh 11 20 || { echo "rc=$?"; exit 1; }
done
} > $O/${TAG}_gemm_ln.txt 2>&1 || { tail -20 $O/${TAG}_gemm_ln.txt; exit 1; }
echo "FAIL lines: $(grep -c FAIL $O/${TAG}_gemm_ln.txt)"; grep "FAIL" $O/${TAG}_gemm_ln.txt | head
grep -E "^==|sustained|fused LayerNorm" $O/${TAG}_gemm_ln.txt
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "layernorm_fused or zigzag or config2 or fp8" > $O/${TAG}_pytest_ln.log 2>&1 || { tail -30 $O/${TAG}_pytest_ln.log; exit 1; }
tail -3 $O/${TAG}_pytest_ln.log
bash scripts/step_ab.sh ${TAG} 3 "SSP2_LN_FUSION=0"
