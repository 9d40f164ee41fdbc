#!/bin/bash
# round 4, call t: the next frame's project stage on a side stream: train tests, then the probe over 1 / 2 / 3 slices (rgb, SH)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4t; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "train or trajectory or cli or splatter" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 python tools/exchange_probe.py cfg5 --slices 1,2,3 > "$OUT/probe_cfg5.jsonl" 2> "$OUT/probe_cfg5.err"; echo "probe5 rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 python tools/exchange_probe.py cfg4 --slices 1,2 --modes all_reduce > "$OUT/probe_cfg4.jsonl" 2> "$OUT/probe_cfg4.err"; echo "probe4 rc=$?" | tee -a "$OUT/steps.txt"
grep -E "passed|failed" "$OUT/pytest.log"; cat "$OUT/probe_cfg5.jsonl" "$OUT/probe_cfg4.jsonl"
