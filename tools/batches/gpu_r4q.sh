#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4q; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 300 python tools/scratch/small_train_probe.py 100000 > "$OUT/probe.txt" 2>&1
timeout 300 python tools/scratch/small_train_probe.py 10000 >> "$OUT/probe.txt" 2>&1
grep "^n=" "$OUT/probe.txt"
