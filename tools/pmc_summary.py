#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean counter value per
dispatch.  FETCH_SIZE / WRITE_SIZE are reported in bytes with the gfx950 corrections of
MI355X_MICROARCH.md section HBM: rocprofv3 reports them in units of 1 KiB... no: FETCH_SIZE here is
in KB (derived: TCC_EA0_RDREQ*64B/1024); wide coalesced reads are under-reported 2x (x2 column)."""
import csv
import sys
from collections import defaultdict


def main(paths):
    acc = defaultdict(lambda: defaultdict(list))
    for p in paths:
        with open(p) as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
                k = k.split("(")[0]
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = sorted({c for k in acc for c in acc[k]})
    print("kernel," + ",".join(names) + ",dispatches")
    for k in sorted(acc):
        row = []
        n = 0
        for c in names:
            v = acc[k].get(c, [])
            n = max(n, len(v))
            row.append(f"{sum(v) / len(v):.1f}" if v else "")
        print(k[:60] + "," + ",".join(row) + f",{n}")


if __name__ == "__main__":
    main(sys.argv[1:])
