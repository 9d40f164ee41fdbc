#!/bin/bash
# round 4, call ze: MFMA SH backward after the instruction trims (clamp modifier, -ln 2 folded into w, opacity sum from
# sum s / sigma(opa)) and without the bucket scan: gradients against the old kernel, SH tests of the suite
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4ze; mkdir -p "$OUT"
cd "$R"
timeout 600 python tools/mfma_bwd_check.py compare old d2 d3 cfg4_deg3 cfg4 > "$OUT/compare.txt" 2> "$OUT/compare.err"; echo "compare rc=$?" | tee -a "$OUT/steps.txt"
grep "raster_bwd\|rgb \|opa " "$OUT/compare.txt" | cut -c1-200
timeout 900 python -m pytest tests -m gpu -x -q -k "sh or deg3 or backward or long_lists or degenerate or trajectory or train" > "$OUT/pytest.txt" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
grep -E "passed|failed" "$OUT/pytest.txt" | tail -2
