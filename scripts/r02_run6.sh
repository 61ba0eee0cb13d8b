#!/bin/bash
# gemm_bench LayerNorm-fusion checks (bit identity + the standalone kernel's time), then the round-end evidence at one source state
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
B=2ssp-x-vit_amd/csrc/tools/gemm_bench.bin
{
for sh in "63040 768 768" "63040 768 3072" "5000 768 768" "300 768 768" "63040 1024 1024" "16448 1280 1280"; do
  echo "== $sh epi 15 (resid + LN)"; timeout -k 10 120 $B $sh 15 20 || echo "rc=$?"
  echo "== $sh epi 11 (resid)"; timeout -k 10 120 $B $sh 11 20 || echo "rc=$?"
done
} > $O/r02_gemm_ln2.txt 2>&1
grep -c "FAIL" $O/r02_gemm_ln2.txt; grep "standalone" $O/r02_gemm_ln2.txt
bash scripts/final_profile.sh ${1:-r02_e}
