#!/bin/bash
# round 4, call v: degree-3 SH backward: 2 waves per SIMD without spills, with / without the finished-half-tile skip
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4v; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 900 python tools/ab_variants.py run cfg4_deg3 > "$OUT/ab.txt" 2> "$OUT/ab.err"; echo "ab rc=$?" | tee -a "$OUT/steps.txt"
cut -c1-40,180-330 "$OUT/ab.txt"
