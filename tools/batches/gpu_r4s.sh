#!/bin/bash
# round 4, call s: kernel traces of the forced-collective training loop on the final tree (SUM exchange): one slice and
# two slices -- which RCCL kernels are left on one rank, what the per-slice launches cost
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4s; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for K in 1 2; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace$K" -o s -- python "$R/tools/exchange_probe.py" cfg5 --slices $K --modes all_reduce --trace --repeats 3 > "$OUT/trace$K.json" 2> "$OUT/trace$K.err"
cp $(find "$OUT/trace$K" -name '*kernel_stats.csv' | head -1) "$OUT/trace${K}_kernel_stats.csv"
done
cd "$R"; for K in 1 2; do echo "== $K slice(s)"; cut -d, -f1-4 "$OUT/trace${K}_kernel_stats.csv" | cut -c1-110 | head -14; done
