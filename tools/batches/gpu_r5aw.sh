#!/bin/bash
# round 5, call au: the LAST tree (bounded counter lag, long-list flag from 6,144 entries on): whole GPU suite, smoke, default bench
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5aw; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -q -rf --maxfail=30 -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
timeout 300 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
cat "$OUT/steps.txt"; tail -n 6 "$OUT/pytest.log" | cut -c1-300; tail -n 2 "$OUT/smoke.log"
