#!/bin/bash
# round 4, call zk: final set on the round's last tree (after the two-pass SH projection backward): GPU suite, default
# bench, forced-collective bench (exposed_ms, rgb and SH scene), kernel trace of the headline command, kernel trace + PMC of
# the degree-3 / degree-2 forward + backward, densifying SH training soak (degree 2 and 3)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4zk; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest.txt" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
grep -E "passed|failed" "$OUT/pytest.txt" | tail -2
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python bench.py --force-collective --legs headline,multi_gpu > "$OUT/bench_fc.json" 2> "$OUT/bench_fc.err"; echo "bench_fc rc=$?" | tee -a "$OUT/steps.txt"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/benchprof" -o s -- python "$R/bench.py" --legs headline --no-cpu-baseline > "$OUT/benchprof.json" 2> "$OUT/benchprof.err"; cp $(find "$OUT/benchprof" -name '*kernel_stats.csv' | head -1) "$OUT/benchprof_kernel_stats.csv")
echo "benchprof rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 bash tools/profile_round.sh r4zk_deg3 cfg4 fwdbwd --sh-degree 3 > "$OUT/profile_deg3.txt" 2>&1; echo "deg3 rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 bash tools/profile_round.sh r4zk_deg2 cfg4 fwdbwd --sh-degree 2 > "$OUT/profile_deg2.txt" 2>&1; echo "deg2 rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 python tools/soak.py 0 2000 2 > "$OUT/soak_sh2.json" 2> "$OUT/soak_sh2.err"; echo "soak2 rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 python tools/soak.py 0 2000 3 > "$OUT/soak_sh3.json" 2> "$OUT/soak_sh3.err"; echo "soak3 rc=$?" | tee -a "$OUT/steps.txt"
cat "$OUT/steps.txt"; tail -c 700 "$OUT/soak_sh2.json"; echo; tail -c 700 "$OUT/soak_sh3.json"; echo
head -7 "$OUT/benchprof_kernel_stats.csv" | cut -d, -f1-4 | cut -c1-50,150-400
