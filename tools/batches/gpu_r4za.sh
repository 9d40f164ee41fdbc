#!/bin/bash
# round 4, call za: MFMA SH backward with the LDS operands requested one pixel row ahead (in-tree) against loads-at-use
# (pf0) and the old kernel; degree 2 on the same kernel (m2); then kernel trace + PMC of the degree-3 forward + backward
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4za; mkdir -p "$OUT"
cd "$R"
timeout 600 python tools/mfma_bwd_check.py compare old,pf0 s3 cfg4_deg3 > "$OUT/compare.txt" 2> "$OUT/compare.err"; echo "compare rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 python tools/mfma_bwd_check.py compare m2 cfg4 >> "$OUT/compare.txt" 2>> "$OUT/compare.err"; echo "compare2 rc=$?" | tee -a "$OUT/steps.txt"
grep "raster_bwd\|rgb " "$OUT/compare.txt" | cut -c1-200
timeout 900 bash tools/profile_round.sh r4za_deg3 cfg4 fwdbwd --sh-degree 3 > "$OUT/profile.txt" 2>&1; echo "profile rc=$?" | tee -a "$OUT/steps.txt"
tail -30 "$OUT/profile.txt" | cut -c1-250
