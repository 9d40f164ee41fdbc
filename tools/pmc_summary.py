#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean counter value per dispatch.

    python tools/pmc_summary.py <counter_collection.csv ...>  >  pmc_summary.csv

Kernel names are written as QUOTED csv fields (template arguments contain commas: `raster_backward_kernel<27, true,
false>`).  FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 derives them (TCC_EA0_RDREQ x 64 B / 1024 ...); on gfx950
FETCH_SIZE under-reports wide streaming reads by 2x (calibration in profiles/traffic.json) -- this tool reports the raw
values, tools/make_traffic.py applies the correction."""
import csv
import sys
from collections import defaultdict


def short_name(k: str) -> str:
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    depth, cut = 0, len(k)
    for i, ch in enumerate(k):  # cut at the argument list's "(" -- the first one outside <...>
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return k[:cut].strip()


def summarise(paths):
    acc = defaultdict(lambda: defaultdict(list))
    for p in paths:
        with open(p, newline="") as f:
            for r in csv.DictReader(f):
                acc[short_name(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main(paths):
    acc = summarise(paths)
    names = sorted({c for k in acc for c in acc[k]})
    w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["kernel"] + names + ["dispatches"])
    for k in sorted(acc):
        row, n = [], 0
        for c in names:
            v = acc[k].get(c, [])
            n = max(n, len(v))
            row.append(round(sum(v) / len(v), 1) if v else "")
        w.writerow([k] + row + [n])


if __name__ == "__main__":
    main(sys.argv[1:])
