#!/bin/bash
# Four-wave 256x256 GEMM (gemm256w4.hip.h, epi 40/41/42) against the 8-wave ping-pong kernel (10/11/12): bit-compare with
# the 128x128 kernel on edge shapes, then interleaved timings on the bench's shapes.
cd "$GRAFT_REPO_ROOT"
B=2ssp-x-vit_amd/csrc/tools/gemm_bench.bin
fail=0
for shape in "8192 128 64 40" "8192 128 64 41" "8192 128 64 42" "8200 64 128 41" "8200 128 128 40" "9000 192 192 42" "4096 64 64 41" "5000 2304 768 40" "6000 768 1984 41" "12608 768 3072 41" "4100 3072 768 42"; do
  out=$(timeout -k 5 200 $B $shape 3) || { echo "CRASH $shape"; fail=1; break; }
  echo "$out" | grep -q "FAIL" && { echo "FAIL $shape"; echo "$out" | head -6; fail=1; } || echo "ok   $shape  $(echo "$out" | grep -c bit-identical) checks"
done
[ $fail = 1 ] && exit 1
for rep in 1 2; do
for shape in "63040 2304 768 0" "63040 768 768 1" "63040 768 3072 1" "63040 3072 768 2" "315200 3072 768 2" "65792 3840 1280 0" "65792 5120 1280 2"; do
  set -- $shape
  for base in 10 40; do
    timeout -k 5 200 $B $1 $2 $3 $((base + $4)) 30 | grep "median" || exit 1
  done
done
done
