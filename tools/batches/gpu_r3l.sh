#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r3l; mkdir -p "$OUT"; cd "$R"
timeout 900 python tools/ab_variants.py run cfg5_fwd > "$OUT/ab.txt" 2> "$OUT/ab.err"; grep -v amdgpu "$OUT/ab.txt"
