#!/bin/bash
# round 5, call j: where does the densifying run spend its time?  Kernel traces of tools/soak.py's training run (SH degree 2,
# 2,000 iterations; rgb, 3,000 iterations)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5j; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_sh2" -o s -- python "$R/tools/soak.py" 0 2000 2 > "$OUT/soak_sh2.json" 2> "$OUT/soak_sh2.err"
cp $(find "$OUT/trace_sh2" -name '*kernel_stats.csv' | head -1) "$OUT/kernel_stats_sh2.csv"; rm -rf "$OUT/trace_sh2"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_rgb" -o s -- python "$R/tools/soak.py" 0 3000 0 > "$OUT/soak_rgb.json" 2> "$OUT/soak_rgb.err"
cp $(find "$OUT/trace_rgb" -name '*kernel_stats.csv' | head -1) "$OUT/kernel_stats_rgb.csv"; rm -rf "$OUT/trace_rgb"
head -14 "$OUT/kernel_stats_sh2.csv" | cut -c1-80,200-300
