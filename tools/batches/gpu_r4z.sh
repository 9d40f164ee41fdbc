#!/bin/bash
# round 4, call z: the MFMA SH backward with two tiles per workgroup (default for degree 3) against the old kernel and
# against one / three tiles per workgroup; then the whole GPU suite on the in-tree build
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4z; mkdir -p "$OUT"
cd "$R"
timeout 900 python tools/mfma_bwd_check.py compare old,mfma27,t1,w6t3 d3 cfg4_deg3 cfg4 > "$OUT/compare.txt" 2> "$OUT/compare.err"; echo "compare rc=$?" | tee -a "$OUT/steps.txt"
grep "raster_bwd\|rgb " "$OUT/compare.txt" | cut -c1-200
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/pytest.txt" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
tail -8 "$OUT/pytest.txt"
