#!/bin/bash
# round 4, call i: wave-cooperative stop-key lookup of the projection backward (in-tree) against the per-thread walk
# (build/variants/threadmask): stage times at cfg5 / cfg2, then the backward tests
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4i; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 400 python tools/ab_variants.py run cfg5 cfg2 > "$OUT/ab.txt" 2> "$OUT/ab.err"
echo "ab rc=$?" | tee -a "$OUT/steps.txt"
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "backward or train or trajectory or smoke or splatter or densif" > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
grep -E "bwd" "$OUT/ab.txt" | cut -c1-300; tail -4 "$OUT/pytest.log"
