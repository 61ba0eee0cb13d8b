#!/bin/bash
# N-rank rehearsal on a ONE-GPU box: every rank its own engine on cuda:0, exchange over gloo (SSP2_REHEARSE_ONE_CARD=1).
# Correctness only — launcher, dealing of the batches, exchange steps, one result line; the ranks share one card's time.
#   bash scripts/one_card_rehearsal.sh TAG
# config 2 fixes the TOTAL (2048 calibration / 2560 eval images), and batch b is a function of b alone: the selections at 1, 2 and 4
# ranks must be identical.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; O=gpurun_out/${TAG}
python3 bench.py --config 2 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > ${O}_config2_1rank.jsonl 2> ${O}_config2_1rank.err || exit 1
for n in 2 4; do
  SSP2_REHEARSE_ONE_CARD=1 timeout -k 10 500 python3 bench.py --gpus $n --config 2 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > ${O}_config2_${n}ranks_one_card.jsonl 2> ${O}_config2_${n}ranks_one_card.err || exit 1
  grep "ssp2vit rank" ${O}_config2_${n}ranks_one_card.err
done
SSP2_REHEARSE_ONE_CARD=1 timeout -k 10 500 python3 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > ${O}_config1_2ranks_one_card.jsonl 2> ${O}_config1_2ranks_one_card.err || exit 1
grep "ssp2vit rank" ${O}_config1_2ranks_one_card.err
python3 - "$O" <<'PY'
import json, sys
o = sys.argv[1]
def last(p): return json.loads(open(p).read().strip().splitlines()[-1])
one = last(f"{o}_config2_1rank.jsonl")
ok = True
for n in (2, 4):
    d = last(f"{o}_config2_{n}ranks_one_card.jsonl")
    same = d["selected_blocks_per_target"] == one["selected_blocks_per_target"]
    ok &= same
    print(f"config 2, {n} ranks on one card: selections {d['selected_blocks_per_target']}  same as 1 rank: {same}  | {d['collectives']}")
print("config 2, 1 rank:", one["selected_blocks_per_target"])
print("REHEARSAL", "OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
PY
