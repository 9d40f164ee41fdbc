#!/bin/bash
# round 4, call l: the round's measurement set on the final tree -- default bench, forced-collective bench (multi_gpu object
# with exposed_ms), kernel trace of the headline command, per-config kernel traces + PMC passes (traffic.json), sweep over N
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4l; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python bench.py --force-collective --legs headline,multi_gpu > "$OUT/bench_fc.json" 2> "$OUT/bench_fc.err"; echo "bench_fc rc=$?" | tee -a "$OUT/steps.txt"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/benchprof" -o s -- python "$R/bench.py" --legs headline --no-cpu-baseline > "$OUT/benchprof.json" 2> "$OUT/benchprof.err"; cp $(find "$OUT/benchprof" -name '*kernel_stats.csv' | head -1) "$OUT/benchprof_kernel_stats.csv")
echo "benchprof rc=$?" | tee -a "$OUT/steps.txt"
timeout 500 tools/profile_round.sh r4l/prof_cfg5 cfg5 fwd > "$OUT/prof_cfg5.log" 2>&1; echo "prof_cfg5 rc=$?" | tee -a "$OUT/steps.txt"
timeout 500 tools/profile_round.sh r4l/prof_cfg2 cfg2 fwd > "$OUT/prof_cfg2.log" 2>&1; echo "prof_cfg2 rc=$?" | tee -a "$OUT/steps.txt"
timeout 500 tools/profile_round.sh r4l/prof_cfg5t cfg5 fwdbwd > "$OUT/prof_cfg5t.log" 2>&1; echo "prof_cfg5t rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 tools/profile_round.sh r4l/prof_cfg4 cfg4 fwdbwd > "$OUT/prof_cfg4.log" 2>&1; echo "prof_cfg4 rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 tools/profile_round.sh r4l/prof_cfg4d3 cfg4 fwdbwd --sh-degree 3 > "$OUT/prof_cfg4d3.log" 2>&1; echo "prof_cfg4d3 rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 python tools/sweep_n.py > "$OUT/sweep_n.jsonl" 2> "$OUT/sweep_n.err"; echo "sweep rc=$?" | tee -a "$OUT/steps.txt"
cat "$OUT/steps.txt"; tail -c 1500 "$OUT/bench_default.json"; echo; cat "$OUT/sweep_n.jsonl"
