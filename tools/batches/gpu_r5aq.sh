#!/bin/bash
# round 5, call aq: the host may run at most 8 frames ahead of the asynchronous counters (FrameRenderer.ASYNC_COUNTER_LAG): the two
# 7,001-iteration fits at their own pace (fused / two-kernel step, tools/fused_adam_bisect.py --sequential --no-log), train + frame
# tests, training legs
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5aq; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 400 python tools/fused_adam_bisect.py 7001 100 --sequential --demo-capacity --no-log > "$OUT/seq.json" 2> "$OUT/seq.err"; echo "bisect rc=$?" | tee -a "$OUT/steps.txt"
timeout 300 python tools/train_demo.py > "$OUT/demo_fused.json" 2> "$OUT/demo_fused.err"
GS_TRAIN_FUSE_ADAM=0 timeout 300 python tools/train_demo.py > "$OUT/demo_unfused.json" 2> "$OUT/demo_unfused.err"
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_frame.py tests/test_gpu_splatter.py tests/test_gpu_densify.py -m gpu -q -rf --maxfail=30 -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 python bench.py --legs headline,train > "$OUT/bench_train.json" 2> "$OUT/bench_train.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
cat "$OUT/seq.json"; cat "$OUT"/demo_*.json | cut -c1-400; tail -n 4 "$OUT/pytest.log" | cut -c1-300
