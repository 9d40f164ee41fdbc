#!/bin/bash
# round 4, call m: full GPU suite + smoke on the final tree; sweep over N with and without the asynchronous counter read-back
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4m; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=10 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 python tools/sweep_n.py > "$OUT/sweep_n.jsonl" 2> "$OUT/sweep_n.err"; echo "sweep rc=$?" | tee -a "$OUT/steps.txt"
GS_SWEEP_ASYNC=1 timeout 600 python tools/sweep_n.py 10000 100000 > "$OUT/sweep_n_async.jsonl" 2> "$OUT/sweep_n_async.err"; echo "sweep_async rc=$?" | tee -a "$OUT/steps.txt"
grep -E "passed|failed" "$OUT/pytest.log"; tail -2 "$OUT/smoke.log"; cat "$OUT/sweep_n.jsonl" "$OUT/sweep_n_async.jsonl" | cut -c1-330
