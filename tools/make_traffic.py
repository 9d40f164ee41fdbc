#!/usr/bin/env python3
"""profiles/traffic.json from PMC summaries:  python tools/make_traffic.py cfg5=profiles/r02_a_cfg5_pmc_summary.csv:6950364 ...

Per kernel of each config: FETCH_SIZE / WRITE_SIZE (KiB per launch, raw) and traffic_bytes = 2 x FETCH + WRITE -- the
gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts a 128-byte request as 64 B), calibrated in
profiles/traffic.json's note on kernels with known streaming reads."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "profiles", "traffic.json")
data = json.load(open(path)) if os.path.exists(path) else {}
for arg in sys.argv[1:]:
    cfg, rest = arg.split("=", 1)
    fn, pairs = rest.rsplit(":", 1)
    entry = {"tile_pairs": int(pairs), "source": os.path.relpath(fn, ROOT)}
    with open(fn, newline="") as f:
        for r in csv.DictReader(f):
            if r.get("FETCH_SIZE") in (None, "") or r.get("WRITE_SIZE") in (None, ""):
                continue
            fk, wk = float(r["FETCH_SIZE"]), float(r["WRITE_SIZE"])
            name = r["kernel"].split("<")[0]
            if name in entry:  # several instantiations of one template: keep the one that moved the most
                if entry[name]["traffic_bytes"] >= int((2 * fk + wk) * 1024):
                    continue
            entry[name] = {"kernel": r["kernel"], "fetch_kib": fk, "write_kib": wk,
                           "traffic_bytes": int((2 * fk + wk) * 1024)}
    data[cfg] = entry
json.dump(data, open(path, "w"), indent=1)
print(json.dumps({k: {kk: vv.get("traffic_bytes") for kk, vv in v.items() if isinstance(vv, dict)}
                  for k, v in data.items() if isinstance(v, dict)}, indent=1))
