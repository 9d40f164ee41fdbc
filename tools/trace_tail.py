#!/usr/bin/env python3
"""Per-kernel time of the LAST part of a run from a rocprofv3 kernel trace (…_kernel_trace.csv):

    python tools/trace_tail.py <kernel_trace.csv> [fraction of the run, default 0.1]

A densifying training run changes its scene while it runs: the averages of `--stats` mix its fast start with its slow end.
This prints, for the dispatches whose start lies in the last `fraction` of the traced interval, calls / average / maximum /
total per kernel and the share of the interval the kernels were busy."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
rows = []
with open(path, newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
t0, t1 = min(r[0] for r in rows), max(r[1] for r in rows)
cut = t1 - frac * (t1 - t0)
agg = defaultdict(lambda: [0, 0, 0])
for s, e, n in rows:
    if s >= cut:
        n = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        a = agg[n]
        a[0] += 1
        a[1] += e - s
        a[2] = max(a[2], e - s)
busy = sum(a[1] for a in agg.values())
print(f"last {frac:.0%} of the traced interval: {(t1 - cut) / 1e9:.3f} s, kernels busy {busy / 1e9:.3f} s")
for n, (c, tot, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"  {n[:70]:70s} calls {c:6d}  avg {tot / c / 1e3:9.1f} us  max {mx / 1e3:9.1f} us  total {tot / 1e9:6.3f} s")
