#!/bin/bash
# round 5, call x: the tree with the fused optimizer step and the packed row steps / loss filters: whole GPU suite, smoke, default
# bench, then the profile sets of the kernels that changed (cfg5 forward + backward, cfg4 degree 2 and 3 forward + backward)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5x; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -q -rf --maxfail=30 -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
timeout 300 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 bash tools/profile_round.sh r5x_cfg5t cfg5 fwdbwd > "$OUT/profile_cfg5t.txt" 2>&1; echo "cfg5t rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 bash tools/profile_round.sh r5x_deg2 cfg4 fwdbwd --sh-degree 2 > "$OUT/profile_deg2.txt" 2>&1; echo "deg2 rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 bash tools/profile_round.sh r5x_deg3 cfg4 fwdbwd --sh-degree 3 > "$OUT/profile_deg3.txt" 2>&1; echo "deg3 rc=$?" | tee -a "$OUT/steps.txt"
cd "$R"; cat "$OUT/steps.txt"; tail -n 12 "$OUT/pytest.log" | cut -c1-300; tail -n 2 "$OUT/smoke.log"
