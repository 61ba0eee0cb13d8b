#!/bin/bash
# Attention micro-benchmark on the GPU box: one-item vs persistent kernel (bit-compared), phase stamps, clock calibration.
# Build first (CPU is fine):  cd 2ssp-x-vit_amd/csrc/tools && hipcc --offload-arch=gfx950 -O3 -std=c++17 attn_bench.hip -o attn_bench.bin
#                             hipcc --offload-arch=gfx950 -O3 -std=c++17 -DATTN_STAMPS attn_bench.hip -o attn_stamps.bin
cd "$GRAFT_REPO_ROOT"
T=2ssp-x-vit_amd/csrc/tools
timeout -k 10 60 $T/attn_bench.bin calib || exit 1
for n in 320 512 3520; do timeout -k 10 60 $T/attn_bench.bin $n 12 197 || exit 1; done
[ -x $T/attn_stamps.bin ] && timeout -k 10 60 $T/attn_stamps.bin 512 12 197 | head -20
