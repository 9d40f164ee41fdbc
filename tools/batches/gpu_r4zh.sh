#!/bin/bash
# round 4, call zh: final set on the round's tree: GPU suite, smoke(), default bench line, kernel trace + PMC of the
# degree-3 / degree-2 forward + backward at 2.4 M Gaussians, regression look at the rgb backward / long-list paths
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4zh; mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest.txt" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
grep -E "passed|failed" "$OUT/pytest.txt" | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.txt" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/steps.txt"; tail -2 "$OUT/smoke.txt"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 bash tools/profile_round.sh r4zh_deg3 cfg4 fwdbwd --sh-degree 3 > "$OUT/profile_deg3.txt" 2>&1; echo "deg3 rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 bash tools/profile_round.sh r4zh_deg2 cfg4 fwdbwd --sh-degree 2 > "$OUT/profile_deg2.txt" 2>&1; echo "deg2 rc=$?" | tee -a "$OUT/steps.txt"
grep "raster_backward\|raster_forward\|project_backward" "$R/gpurun_out/r4zh_deg3/kernel_stats.csv" "$R/gpurun_out/r4zh_deg2/kernel_stats.csv" | cut -d, -f1-4 | cut -c1-60,200-330
timeout 300 python tools/stage_profile.py cfg5 cfg2 > "$OUT/stage_profile_rgb.txt" 2>&1; tail -2 "$OUT/stage_profile_rgb.txt" | cut -c1-400
