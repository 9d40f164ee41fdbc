#!/bin/bash
# round 4, call zb: MFMA SH backward leaving out pixel rows whose 16 pixels have all stopped (in-tree) against the same
# kernel without the skip (rs0) and the old kernel; degree 2 on the new kernel (m2) against the old one
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4zb; mkdir -p "$OUT"
cd "$R"
timeout 600 python tools/mfma_bwd_check.py compare old,rs0 d3 cfg4_deg3 > "$OUT/compare.txt" 2> "$OUT/compare.err"; echo "compare rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 python tools/mfma_bwd_check.py compare m2 d2 cfg4 >> "$OUT/compare.txt" 2>> "$OUT/compare.err"; echo "compare2 rc=$?" | tee -a "$OUT/steps.txt"
grep "raster_bwd\|rgb \|pos " "$OUT/compare.txt" | cut -c1-200
