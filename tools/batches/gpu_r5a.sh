#!/bin/bash
# round 5, call a: first tree of the round (SH rows in whole lines + stop keys, exec-row counters, bench legs cfg1 / soak,
# exchange self-check): whole GPU suite with -s (the full-size gradient parity tables and the seam test print their reports),
# default bench, kernel trace + PMC of the degree-2 / degree-3 forward + backward
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5a; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -q -rf --maxfail=30 -p no:cacheprovider -s > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 bash tools/profile_round.sh r5a_deg2 cfg4 fwdbwd --sh-degree 2 > "$OUT/profile_deg2.txt" 2>&1; echo "deg2 rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 bash tools/profile_round.sh r5a_deg3 cfg4 fwdbwd --sh-degree 3 > "$OUT/profile_deg3.txt" 2>&1; echo "deg3 rc=$?" | tee -a "$OUT/steps.txt"
cat "$OUT/steps.txt"; tail -n 25 "$OUT/pytest.log"; tail -c 1500 "$OUT/bench_default.json"
