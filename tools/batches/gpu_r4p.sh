#!/bin/bash
# round 4, call p: final numbers with the fixed-scene training protocol (learning rate 0; the moving-scene figure next to
# it): default bench, forced-collective bench, kernel trace of the headline command, sweep over N; soak run (long render
# loop + densifying training) on the final tree
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4p; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python bench.py --force-collective --legs headline,multi_gpu > "$OUT/bench_fc.json" 2> "$OUT/bench_fc.err"; echo "bench_fc rc=$?" | tee -a "$OUT/steps.txt"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/benchprof" -o s -- python "$R/bench.py" --legs headline --no-cpu-baseline > "$OUT/benchprof.json" 2> "$OUT/benchprof.err"; cp $(find "$OUT/benchprof" -name '*kernel_stats.csv' | head -1) "$OUT/benchprof_kernel_stats.csv")
echo "benchprof rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 python tools/sweep_n.py > "$OUT/sweep_n.jsonl" 2> "$OUT/sweep_n.err"; echo "sweep rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 python tools/soak.py 20000 3000 > "$OUT/soak.json" 2> "$OUT/soak.err"; echo "soak rc=$?" | tee -a "$OUT/steps.txt"
cat "$OUT/steps.txt"; cat "$OUT/sweep_n.jsonl" | cut -c1-300; tail -c 600 "$OUT/soak.json"
