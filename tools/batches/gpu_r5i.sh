#!/bin/bash
# round 5, call i: the final tree -- whole GPU suite, smoke(), default bench, kernel trace of the headline command, the gradient
# exchange through RCCL on one rank (--force-collective), the metric over the scene size (tools/sweep_n.py)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5i; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -q -rf --maxfail=30 -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
timeout 300 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/benchprof" -o s -- python "$R/bench.py" --legs headline --no-cpu-baseline > "$OUT/benchprof.json" 2> "$OUT/benchprof.err"; cp $(find "$OUT/benchprof" -name '*kernel_stats.csv' | head -1) "$OUT/benchprof_kernel_stats.csv"; rm -rf "$OUT/benchprof"); echo "benchprof rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python bench.py --gpus 1 --force-collective --legs headline,multi_gpu > "$OUT/bench_force_collective.json" 2> "$OUT/bench_force_collective.err"; echo "force rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python tools/sweep_n.py > "$OUT/sweep_n.jsonl" 2> "$OUT/sweep_n.err"; echo "sweep rc=$?" | tee -a "$OUT/steps.txt"
cat "$OUT/steps.txt"; tail -n 4 "$OUT/pytest.log" | cut -c1-300; tail -n 2 "$OUT/smoke.log"; cat "$OUT/sweep_n.jsonl" | cut -c1-300
