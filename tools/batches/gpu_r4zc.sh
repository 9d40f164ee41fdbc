#!/bin/bash
# round 4, call zc: heavy-tiles-first dispatch of the MFMA SH backward (in-tree) against raster order (ord0); 16 row loads
# in flight in the SH projection backward (u16) against 8; then the whole GPU suite on the in-tree build
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4zc; mkdir -p "$OUT"
cd "$R"
timeout 600 python tools/mfma_bwd_check.py compare ord0,u16 cfg4_deg3 cfg4 > "$OUT/compare.txt" 2> "$OUT/compare.err"; echo "compare rc=$?" | tee -a "$OUT/steps.txt"
grep "raster_bwd\|rgb " "$OUT/compare.txt" | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest.txt" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
tail -5 "$OUT/pytest.txt"
