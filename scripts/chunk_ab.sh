#!/bin/bash
# BASELINE configs[2] at one GPU (2048 calib + 2560 eval images, three targets): the search's chunk size (images per layer-major launch group)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03_t}_chunk_ab.txt; : > $O
for r in 1 2; do
  for c in 320 640 1280 2560; do
    ms=$(timeout -k 10 400 python3 bench.py --config 2 --eval-chunk $c --steps 3 --warmup 1 --no-api --no-cpu-baseline --no-roofline --no-overlap-figure 2>/dev/null | python3 -c "import sys, json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "round $r  eval-chunk $c  $ms ms" | tee -a $O
  done
done
