#!/bin/bash
# round 4, call zd: measurement set with the SH backward on the matrix pipe: default bench line, kernel trace + PMC of the
# degree-3 and degree-2 forward + backward at 2.4 M Gaussians
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4zd; mkdir -p "$OUT"
cd "$R"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
tail -c 1500 "$OUT/bench_default.json"; echo
timeout 600 bash tools/profile_round.sh r4zd_deg3 cfg4 fwdbwd --sh-degree 3 > "$OUT/profile_deg3.txt" 2>&1; echo "deg3 rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 bash tools/profile_round.sh r4zd_deg2 cfg4 fwdbwd --sh-degree 2 > "$OUT/profile_deg2.txt" 2>&1; echo "deg2 rc=$?" | tee -a "$OUT/steps.txt"
grep "raster_backward\|raster_forward\|project_backward" "$R/gpurun_out/r4zd_deg3/kernel_stats.csv" "$R/gpurun_out/r4zd_deg2/kernel_stats.csv" | cut -c1-260
