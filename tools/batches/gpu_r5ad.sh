#!/bin/bash
# round 5, call ad: the side stream's fork / join events without the system-scope fence (in-tree) against plain events (build/variants/evsys)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5ad; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -rf --maxfail=30 -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
for i in 1 2; do
  timeout 600 python bench.py --legs headline,train > "$OUT/bench_train_new_$i.json" 2> "$OUT/bench_train_new_$i.err"; echo "bench new $i rc=$?" | tee -a "$OUT/steps.txt"
  GS_AMD_LIB=$R/build/variants/evsys/libgs_amd.so timeout 600 python bench.py --legs headline,train > "$OUT/bench_train_old_$i.json" 2> "$OUT/bench_train_old_$i.err"; echo "bench old $i rc=$?" | tee -a "$OUT/steps.txt"
done
cd /tmp; export TMPDIR=/tmp
for c in "cfg2" "cfg5"; do
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/tr_$c" -o t -- python "$R/tools/prof_target.py" $c --train --frames 30 > "$OUT/$c.json" 2> "$OUT/$c.err"
  f=$(find "$OUT/tr_$c" -name '*kernel_trace.csv' | head -1)
  python "$R/tools/step_timeline.py" "$f" > "$OUT/timeline_$c.txt" 2>&1
  rm -rf "$OUT/tr_$c"
done
tail -n 4 "$OUT/pytest.log" | cut -c1-300
