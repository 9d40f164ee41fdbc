#!/bin/bash
# round 4, call d: full GPU tests again (quaternion sums uncontracted), exchange probe over slice counts, and a kernel trace
# of the forced-collective training loop (what do RCCL's one-rank kernels cost?)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4d; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=10 > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 python tools/exchange_probe.py cfg5 --slices 1,2,4,8 > "$OUT/probe_cfg5.jsonl" 2> "$OUT/probe_cfg5.err"
echo "probe rc=$?" | tee -a "$OUT/steps.txt"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace8" -o s -- python "$R/tools/exchange_probe.py" cfg5 --slices 8 --modes all_reduce --trace --repeats 3 > "$OUT/trace8.json" 2> "$OUT/trace8.err"
cp $(find "$OUT/trace8" -name '*kernel_stats.csv' | head -1) "$OUT/trace8_kernel_stats.csv"
cd "$R"
tail -5 "$OUT/pytest.log"; cat "$OUT/probe_cfg5.jsonl"; tail -3 "$OUT/probe_cfg5.err"; cut -c1-160 "$OUT/trace8_kernel_stats.csv" | head -30
