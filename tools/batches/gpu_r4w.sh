#!/bin/bash
# round 4, call w: timing-only build of the project stage with approximate reciprocals / roots in the tile rectangle
# (NOT bit-exact): the upper bound of what an exactness-guarded fast path could save
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4w; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 600 python tools/ab_variants.py run cfg5_fwd cfg2_fwd > "$OUT/ab.txt" 2> "$OUT/ab.err"; echo "ab rc=$?" | tee -a "$OUT/steps.txt"
cut -c1-170 "$OUT/ab.txt"
