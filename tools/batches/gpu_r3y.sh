#!/bin/bash
# round 3 final: GPU suite, smoke, bench (default and forced collective), kernel trace of the headline, size sweep
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r3y; rm -rf "$OUT"; mkdir -p "$OUT"; cd "$R"
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v amdgpu.ids | tail -2
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$? lines=$(wc -l < $OUT/bench_default.json)"
timeout 900 python bench.py --gpus 1 --force-collective --legs headline,train,multi_gpu > "$OUT/bench_force_collective.json" 2> "$OUT/bench_fc.err"; echo "bench fc rc=$? lines=$(wc -l < $OUT/bench_force_collective.json)"
python bench.py --gpus 2 > "$OUT/bench_gpus2.out" 2>&1; echo "bench --gpus 2 rc=$? : $(tail -1 $OUT/bench_gpus2.out)"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/benchprof" -o s -- python "$R/bench.py" --legs headline --no-cpu-baseline > "$OUT/benchprof.json" 2> "$OUT/benchprof.err"; cp $(find "$OUT/benchprof" -name '*kernel_stats.csv' | head -1) "$OUT/benchprof_kernel_stats.csv")
timeout 900 python tools/sweep_n.py 2>/dev/null > "$OUT/sweep_auto.jsonl"; cat "$OUT/sweep_auto.jsonl"
head -c 700 "$OUT/bench_default.json"; echo
