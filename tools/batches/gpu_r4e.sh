#!/bin/bash
# round 4, call e: SUM exchange + mean in the optimizer, direct process-group calls; probe over slice counts on the rgb and
# the SH scene; the trajectory test (prints its drift figures); full GPU suite
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4e; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=10 > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 python tools/exchange_probe.py cfg5 --slices 1,2,3,4 > "$OUT/probe_cfg5.jsonl" 2> "$OUT/probe_cfg5.err"
echo "probe5 rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 python tools/exchange_probe.py cfg4 --slices 1,2,4 > "$OUT/probe_cfg4.jsonl" 2> "$OUT/probe_cfg4.err"
echo "probe4 rc=$?" | tee -a "$OUT/steps.txt"
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" "$OUT/pytest.log" | tail -40; cat "$OUT/probe_cfg5.jsonl" "$OUT/probe_cfg4.jsonl"; tail -3 "$OUT/probe_cfg4.err"
