#!/bin/bash
# Edge shapes of the persistent 256x256 GEMM against the 128x128 kernel: K of one / two / three K-tiles, partial column
# tiles, partial row tiles, every epilogue incl. the scoring variant (tokens 197), and one launch whose activation
# operand exceeds 4 GiB (720000 x 3072 bf16: per-lane offsets are tile-relative).
cd "$GRAFT_REPO_ROOT"
B=2ssp-x-vit_amd/csrc/tools/gemm_bench.bin
fail=0
for shape in "8192 128 64 10" "8192 128 64 11" "8192 128 64 12" "8200 64 128 11" "8200 128 128 10" "9000 192 192 12" "9000 192 192 13" "9001 320 192 14" "4096 64 64 11" "5000 2304 768 10" "4100 1984 768 13" "6000 768 1984 11" "300000 64 64 10" "720000 768 3072 11"; do
  out=$(timeout -k 5 400 $B $shape 3) || { echo "CRASH $shape"; fail=1; continue; }
  echo "$out" | grep -q "FAIL" && { echo "FAIL $shape"; echo "$out" | head -5; fail=1; } || echo "ok   $shape  $(echo "$out" | grep -c bit-identical) checks"
done
exit $fail
