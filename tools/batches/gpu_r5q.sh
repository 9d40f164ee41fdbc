#!/bin/bash
# round 5, call q: the Adam step fused into the projection backward (gs_frame_backward_adam): GPU suite, then bench.py's training
# legs fused (in-tree default) against unfused (GS_TRAIN_FUSE_ADAM=0) on one box, and a kernel trace of the fused step
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5q; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -q -rf --maxfail=30 -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 python bench.py --legs headline,train > "$OUT/bench_train_fused.json" 2> "$OUT/bench_train_fused.err"; echo "bench fused rc=$?" | tee -a "$OUT/steps.txt"
GS_TRAIN_FUSE_ADAM=0 timeout 600 python bench.py --legs headline,train > "$OUT/bench_train_unfused.json" 2> "$OUT/bench_train_unfused.err"; echo "bench unfused rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 python bench.py --legs headline,train > "$OUT/bench_train_fused2.json" 2> "$OUT/bench_train_fused2.err"; echo "bench fused2 rc=$?" | tee -a "$OUT/steps.txt"
tail -n 12 "$OUT/pytest.log" | cut -c1-300
