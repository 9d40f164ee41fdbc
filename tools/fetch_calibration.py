#!/usr/bin/env python3
"""profiles/r04_fetch_calibration.json from one run of tools/ubench/fetch_gather under rocprofv3 (tools/batches/gpu_r4b.sh):

    python tools/fetch_calibration.py gpurun_out/r4b/fetch_gather.jsonl gpurun_out/r4b/fetch_gather_pmc_summary.csv

Per access pattern: the bytes the kernel really moved (known by construction: every record / row is touched exactly once,
the array is four times the Infinity Cache) against what FETCH_SIZE / WRITE_SIZE report, and the raw request counters.
Result (MI355X, ROCm 7.2): FETCH_SIZE = 64 B x TCC_EA0_RDREQ -- a request is 64 B for an isolated line (gathers: exact)
and 128 B for two adjacent lines asked for together (streams: reported at half); nothing on gfx950 tells the two apart
(TCC_EA0_RDREQ_32B stays 0, the 128-byte term of rocprofv3's formula, TCC_BUBBLE, is not counted).  WRITE_SIZE is exact
(32-byte partial and 64-byte full-line requests are told apart)."""
import csv
import json
import sys

jl, pmc = sys.argv[1], sys.argv[2]
known = {}
for line in open(jl):
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        known[d["kernel"]] = d
rows = {r["kernel"]: r for r in csv.DictReader(open(pmc))}
WHAT = {
    "k_stream16": "coalesced 16 B per lane (every streaming kernel of the path)",
    "k_stream4": "coalesced 4 B per lane",
    "k_gather64_quad": "64-byte records in pseudo-random order, four lanes x 16 B (compositing kernels: sorted id -> record)",
    "k_gather64_lane": "64-byte records in pseudo-random order, one lane reads all four quarters",
    "k_gather16_of64": "16 B out of every 64-byte record, one lane per record (rectangle / rec_geom gathers)",
    "k_gather4": "4 B per lane at pseudo-random places (pair_offsets[id])",
    "k_gather48_rows": "48-byte rows at stride 48 in pseudo-random order (round 3's rgb gradient rows, read side)",
    "k_wstream16": "coalesced 16 B per lane, store",
    "k_wscatter64_quad": "whole aligned 64-byte lines, four lanes x 16 B in one instruction (round 4's rgb gradient rows)",
    "k_wscatter48_lane": "48-byte rows at stride 48, one lane, three 16-byte stores (round 3's rgb gradient rows)",
    "k_wscatter1": "one byte per lane at pseudo-random places (round 3's row flags)",
    "k_wscatter8": "8 B per lane at pseudo-random places (table variant's pair scatter)",
}
out = {"_note": __doc__.split("\n\n", 2)[2].replace("\n", " "), "patterns": {}}
for k, d in known.items():
    r = rows.get(k)
    if r is None:
        continue
    write = k.startswith("k_w")
    counter_kib = float(r["WRITE_SIZE" if write else "FETCH_SIZE"])
    e = {"pattern": WHAT.get(k, ""), "useful_bytes": int(d["useful_bytes"]), "GBs": d["GBs"],
         ("WRITE_SIZE_KiB" if write else "FETCH_SIZE_KiB"): counter_kib,
         "useful_over_counter": round(d["useful_bytes"] / (counter_kib * 1024), 3)}
    if write:
        e["WRREQ"], e["WRREQ_64B"] = float(r["TCC_EA0_WRREQ_sum"]), float(r["TCC_EA0_WRREQ_64B_sum"])
    else:
        e["RDREQ"], e["RDREQ_32B"] = float(r["TCC_EA0_RDREQ_sum"]), float(r["TCC_EA0_RDREQ_32B_sum"])
        e["bytes_per_request_if_all_useful"] = round(d["useful_bytes"] / e["RDREQ"], 1)
    out["patterns"][k] = e
json.dump(out, sys.stdout, indent=1)
print()
