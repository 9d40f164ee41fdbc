#!/bin/bash
# round 4, call zv: the round's last tree (SH big-row pre-pass, rgb cooperative sum from 64 rows on, long-list hand-over of
# the SH backward): default bench, kernel trace + PMC of the degree-3 / degree-2 forward + backward, soak runs (rgb with the
# render loop as in r04_p; SH degree 2 and 3)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4zv; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 bash tools/profile_round.sh r4zv_deg3 cfg4 fwdbwd --sh-degree 3 > "$OUT/profile_deg3.txt" 2>&1; echo "deg3 rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 bash tools/profile_round.sh r4zv_deg2 cfg4 fwdbwd --sh-degree 2 > "$OUT/profile_deg2.txt" 2>&1; echo "deg2 rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 python tools/soak.py 20000 3000 > "$OUT/soak_rgb.json" 2> "$OUT/soak_rgb.err"; echo "soak rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 python tools/soak.py 0 2000 2 > "$OUT/soak_sh2.json" 2> "$OUT/soak_sh2.err"; echo "soak2 rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 python tools/soak.py 0 2000 3 > "$OUT/soak_sh3.json" 2> "$OUT/soak_sh3.err"; echo "soak3 rc=$?" | tee -a "$OUT/steps.txt"
cat "$OUT/steps.txt"; for f in soak_rgb soak_sh2 soak_sh3; do tail -c 600 "$OUT/$f.json"; echo; done
