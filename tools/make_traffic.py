#!/usr/bin/env python3
"""profiles/traffic.json from PMC summaries:

    python tools/make_traffic.py cfg5=profiles/r03_af_cfg5_forward_pmc_summary.csv:6950364:profiles/r03_af_cfg5_forward_kernel_stats.csv ...

Per kernel of each config: FETCH_SIZE / WRITE_SIZE (KiB per launch, raw) and traffic_bytes = factor x FETCH + WRITE.
FETCH_SIZE is 64 B x read requests; a request is 128 B for two adjacent lines fetched together (coalesced streams: the
counter reports HALF the bytes, the correction of MI355X_MICROARCH.md) and 64 B for an isolated line (gathers of 64-byte
records by index: the counter is EXACT) -- profiles/r04_fetch_calibration.json (tools/ubench/fetch_gather.hip, round 4).
Round 3 applied x2 to every kernel; the factor is now chosen per kernel by what its reads are (FETCH_FACTOR below), and
every entry carries the bounds [1 x FETCH + WRITE, 2 x FETCH + WRITE] the truth lies between.  With a kernel_stats.csv of the same workload
(third field) also `issue_busy` = SQ_ACTIVE_INST_VALU x 4 / (average kernel time x 2.4 GHz x 1024 SIMDs): the fraction
of the SIMDs' cycles in which a VALU instruction was executing (SQ_ACTIVE_INST_VALU counts in units of 4 cycles, summed
over the SIMDs)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "profiles", "traffic.json")
data = json.load(open(path)) if os.path.exists(path) else {}
CLOCK_GHZ, SIMDS = 2.4, 1024
# kernel -> (factor on FETCH_SIZE, what its reads are).  Default: coalesced streams, x2.
FETCH_FACTOR = {
    "raster_forward_kernel": (1.0, "64-byte records gathered by sorted id, one isolated 64-B request each (exact); the id "
                                   "stream (4 B per composited step) is < 6 % of the bytes"),
    "raster_segment_kernel": (1.0, "as raster_forward_kernel"),
    "raster_backward_pixel_sh_kernel": (1.5, "mixed: 4-KiB checkpoints and image rows stream (x2), records / rectangles / "
                                             "pair offsets are gathered (x1); SH coefficients are contiguous 108 / 192-B "
                                             "runs per Gaussian (between the two)"),
    "raster_backward_pixel_kernel": (1.5, "as raster_backward_pixel_sh_kernel"),
    "raster_backward_rows_kernel": (1.5, "as raster_backward_pixel_sh_kernel (round 5: the rgb backward in the row layout)"),
    "raster_backward_mfma_sh_kernel": (1.5, "as raster_backward_pixel_sh_kernel (round 4: the SH backward of the frame path)"),
    "raster_backward_kernel": (1.5, "as raster_backward_pixel_sh_kernel"),
    "frame_project_backward_kernel": (1.5, "mixed: rectangles, offsets and raw parameters stream (x2), gradient rows are "
                                           "isolated 64-byte lines (rgb, x1) or 144 / 224-byte runs (SH)"),
}


def short(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("<")[0].split("(")[0].strip()


for arg in sys.argv[1:]:
    cfg, rest = arg.split("=", 1)
    parts = rest.split(":")
    fn, pairs = parts[0], parts[1]
    avg_ns = {}
    if len(parts) > 2:
        with open(parts[2], newline="") as f:
            for r in csv.DictReader(f):
                k = short(r["Name"])
                if k not in avg_ns or float(r["TotalDurationNs"]) > avg_ns[k][1]:
                    avg_ns[k] = (float(r["AverageNs"]), float(r["TotalDurationNs"]))
    entry = {"tile_pairs": int(pairs), "source": os.path.relpath(fn, ROOT)}
    with open(fn, newline="") as f:
        for r in csv.DictReader(f):
            if r.get("FETCH_SIZE") in (None, "") or r.get("WRITE_SIZE") in (None, ""):
                continue
            fk, wk = float(r["FETCH_SIZE"]), float(r["WRITE_SIZE"])
            name = r["kernel"].split("<")[0]
            if name.startswith("at::") or name.startswith("__amd"):
                continue  # torch helpers of the profiling script, not the path's kernels
            factor, why = FETCH_FACTOR.get(name, (2.0, "coalesced streams"))
            if name in entry:  # several instantiations of one template: keep the one that moved the most
                if entry[name]["traffic_bytes"] >= int((factor * fk + wk) * 1024):
                    continue
            e = {"kernel": r["kernel"], "fetch_kib": fk, "write_kib": wk, "traffic_bytes": int((factor * fk + wk) * 1024),
                 "traffic_bytes_bounds": [int((fk + wk) * 1024), int((2 * fk + wk) * 1024)],
                 "correction": {"fetch_factor": factor, "reads": why}}
            if name in avg_ns and r.get("SQ_ACTIVE_INST_VALU") not in (None, ""):
                e["kernel_us"] = round(avg_ns[name][0] / 1e3, 2)
                e["issue_busy"] = round(float(r["SQ_ACTIVE_INST_VALU"]) * 4 / (avg_ns[name][0] * CLOCK_GHZ * SIMDS), 3)
            entry[name] = e
    data[cfg] = entry
data["_note"] = (
    "HBM-side bytes per launch from rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE runs of "
    "tools/prof_target.py <config>; current set: the `source` file of every config). FETCH_SIZE / "
    "WRITE_SIZE are in KiB. FETCH_SIZE = 64 B x read requests: a coalesced stream asks for 128 B per request (x2, the "
    "correction of MI355X_MICROARCH.md), an isolated 64-byte line is one 64-B request (x1: exact) -- "
    "profiles/r04_fetch_calibration.json; every entry says which factor it used (`correction`) and gives the bounds. "
    "Streams were calibrated in round 1 on kernels with known streaming reads (bin_count_kernel: 16 B x 376,467 rect records = "
    "6.02 MB, FETCH_SIZE = 3.07 MB; bin_colscan_kernel: 8.03 MB table, FETCH_SIZE = 4.08 MB) and again in round 3 on "
    "frame_project_count_kernel at 2.4 M Gaussians: 56 B x 2.4 M = 134.4 MB of parameters read, FETCH_SIZE = 65,725 KiB = "
    "67.3 MB (ratio 0.50). WRITE_SIZE needs no correction (raster writes 12 B x 2,088,960 px = 25.07 MB, WRITE_SIZE = "
    "24.9 MB). issue_busy: see tools/make_traffic.py.")
json.dump(data, open(path, "w"), indent=1)
print(json.dumps({k: {kk: (vv.get("traffic_bytes"), vv.get("issue_busy")) for kk, vv in v.items() if isinstance(vv, dict)}
                  for k, v in data.items() if isinstance(v, dict)}, indent=1))
