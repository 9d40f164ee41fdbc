#!/bin/bash
# round 5, call l: the END of the densifying SH soak (124 it/s in its last block against 556 in its first): which kernels?
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5l; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_sh2" -o s -- python "$R/tools/soak.py" 0 2000 2 > "$OUT/soak_sh2.json" 2> "$OUT/soak_sh2.err"
T=$(find "$OUT/trace_sh2" -name '*kernel_trace.csv' | head -1)
python "$R/tools/trace_tail.py" "$T" 0.1 > "$OUT/tail_sh2_10pct.txt"; python "$R/tools/trace_tail.py" "$T" 0.03 > "$OUT/tail_sh2_3pct.txt"
rm -rf "$OUT/trace_sh2"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_rgb" -o s -- python "$R/tools/soak.py" 0 3000 0 > "$OUT/soak_rgb.json" 2> "$OUT/soak_rgb.err"
T=$(find "$OUT/trace_rgb" -name '*kernel_trace.csv' | head -1)
python "$R/tools/trace_tail.py" "$T" 0.1 > "$OUT/tail_rgb_10pct.txt"
rm -rf "$OUT/trace_rgb"
cat "$OUT/tail_sh2_10pct.txt" "$OUT/tail_sh2_3pct.txt" "$OUT/tail_rgb_10pct.txt"
