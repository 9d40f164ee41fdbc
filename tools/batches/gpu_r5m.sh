#!/bin/bash
# round 5, call m: the end state of the densifying SH soak -- list lengths, rows per Gaussian, stage times
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5m; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 600 python tools/soak_end_state.py 2 2000 > "$OUT/end_state_sh2.json" 2> "$OUT/end_state_sh2.err"; echo "rc=$?"
timeout 600 python tools/soak_end_state.py 0 3000 > "$OUT/end_state_rgb.json" 2> "$OUT/end_state_rgb.err"; echo "rc=$?"
cat "$OUT/end_state_sh2.json" "$OUT/end_state_rgb.json"; tail -5 "$OUT/end_state_sh2.err"
