#!/bin/bash
# round 6, call e: the whole GPU tier + smoke + the default bench (all legs) on the current tree
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6e; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1800 python -m pytest tests -q -m gpu > "$OUT/gpu_tier.txt" 2>&1; echo "gpu tier rc=$?" | tee -a "$OUT/steps.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.txt" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/steps.txt"
timeout 1500 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
cat "$OUT/steps.txt"; grep -n "FAILED\|passed\|failed" "$OUT/gpu_tier.txt" | tail -12; tail -2 "$OUT/smoke.txt"
