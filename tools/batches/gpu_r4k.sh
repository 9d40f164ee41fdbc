#!/bin/bash
# round 4, call k: stop keys as two 32-bit arrays (depth bits first): stage times + the backward tests
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4k; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
for i in 1 2; do timeout 300 python tools/stage_profile.py cfg5 cfg2 >> "$OUT/stage.txt" 2>&1; done
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "backward or train or trajectory or smoke or splatter or densif or depth_cluster" > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
grep -E "bwd" "$OUT/stage.txt" | cut -c1-300; tail -4 "$OUT/pytest.log"
