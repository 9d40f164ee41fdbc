#!/bin/bash
# round 5, call ac: the frame counters' device-to-host copies on a copy stream of the renderer (in-tree) against the frame's own stream
# (GS_FRAME_COPY_STREAM=0): frame / train / splatter tests, training legs + rgb soak alternating, step timelines
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5ac; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests/test_gpu_frame.py tests/test_gpu_train.py tests/test_gpu_splatter.py tests/test_gpu_densify.py tests/test_gpu_trajectory.py -m gpu -q -rf --maxfail=30 -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
for i in 1 2; do
  timeout 600 python bench.py --legs headline,train > "$OUT/bench_train_new_$i.json" 2> "$OUT/bench_train_new_$i.err"; echo "bench new $i rc=$?" | tee -a "$OUT/steps.txt"
  GS_FRAME_COPY_STREAM=0 timeout 600 python bench.py --legs headline,train > "$OUT/bench_train_old_$i.json" 2> "$OUT/bench_train_old_$i.err"; echo "bench old $i rc=$?" | tee -a "$OUT/steps.txt"
  timeout 200 python tools/soak.py 0 3000 0 > "$OUT/soak_new_$i.json" 2> "$OUT/soak_new_$i.err"
  GS_FRAME_COPY_STREAM=0 timeout 200 python tools/soak.py 0 3000 0 > "$OUT/soak_old_$i.json" 2> "$OUT/soak_old_$i.err"
done
cd /tmp; export TMPDIR=/tmp
for c in "cfg2" "cfg5"; do
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/tr_$c" -o t -- python "$R/tools/prof_target.py" $c --train --frames 30 > "$OUT/$c.json" 2> "$OUT/$c.err"
  f=$(find "$OUT/tr_$c" -name '*kernel_trace.csv' | head -1)
  python "$R/tools/step_timeline.py" "$f" > "$OUT/timeline_$c.txt" 2>&1
  rm -rf "$OUT/tr_$c"
done
tail -n 8 "$OUT/pytest.log" | cut -c1-300
