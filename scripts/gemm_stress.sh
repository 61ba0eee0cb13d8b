#!/bin/bash
# Race screen of the persistent GEMM: 60 launches per shape, every result (output / x / slab) compared with the first.
cd "$GRAFT_REPO_ROOT"
B=${BENCH_BIN:-2ssp-x-vit_amd/csrc/tools/gemm_bench.bin}     # BENCH_BIN=.../gemm_bench_product.bin: the PRODUCT build of the kernels (-USSP2_LAB)
fail=0
for shape in "63040 2304 768 10" "63040 768 768 11" "63040 768 3072 11" "63040 3072 768 12" "102400 3072 768 13" "25000 1984 768 14" "21276 768 768 11" "12608 2304 768 10"; do
  out=$(GEMM_STRESS=${REPS:-60} timeout -k 5 200 $B $shape 3) || { echo "CRASH $shape"; fail=1; continue; }
  echo "$shape: $(echo "$out" | grep -E "stress|verify" | tr '\n' ' ')"
  echo "$out" | grep -q "FAIL" && fail=1
done
exit $fail
