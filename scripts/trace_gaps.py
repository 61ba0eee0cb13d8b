#!/usr/bin/env python3
"""Where the wall time of a bench run goes: reads a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv) and prints
busy time, idle gaps between consecutive kernels (by size class and by the kernel that FOLLOWS the gap), for the
window that holds the last `--tail` fraction of the kernels (the timed steps sit at the end of a bench run).

    python scripts/trace_gaps.py gpurun_out/trace/<host>/<pid>_kernel_trace.csv [--tail 0.6]
"""
import argparse
import csv
import collections
import json


def short(name: str) -> str:
    name = name.replace("void ", "")
    return name[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--tail", type=float, default=0.6)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    rows = rows[int(len(rows) * (1 - a.tail)):]
    span = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e, _ in rows)
    gaps = collections.Counter()
    gap_ns = collections.Counter()
    after = collections.Counter()
    prev_end = rows[0][1]
    for (s, e, n), (ps, pe, pn) in zip(rows[1:], rows[:-1]):
        g = s - max(prev_end, pe)
        prev_end = max(prev_end, e)
        if g <= 0:
            continue
        cls = "<2us" if g < 2000 else "<10us" if g < 10000 else "<100us" if g < 100000 else "<1ms" if g < 1000000 else ">=1ms"
        gaps[cls] += 1
        gap_ns[cls] += g
        after[short(pn) + "  ->  " + short(n)] += g
    out = {"kernels": len(rows), "span_ms": span / 1e6, "busy_ms": busy / 1e6, "idle_frac": 1 - busy / span,
           "gaps": {k: {"count": gaps[k], "ms": gap_ns[k] / 1e6} for k in gaps},
           "top_gap_transitions_ms": {k: v / 1e6 for k, v in after.most_common(12)}}
    per = collections.Counter()
    for s, e, n in rows:
        per[short(n)] += e - s
    out["busy_by_kernel_ms"] = {k: v / 1e6 for k, v in per.most_common(14)}
    print(json.dumps(out, indent=1))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
