#!/bin/bash
# round 5, call r: second version of the fused optimizer step (float4 accesses through an LDS hand-over): the training tests, then
# bench.py's training legs unfused (default) / fused (GS_TRAIN_FUSE_ADAM=1) alternating on one box, and a kernel trace of the fused step
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5r; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -rf --maxfail=30 -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
for i in 1 2; do
  timeout 600 python bench.py --legs headline,train > "$OUT/bench_train_unfused$i.json" 2> "$OUT/bench_train_unfused$i.err"; echo "bench unfused$i rc=$?" | tee -a "$OUT/steps.txt"
  GS_TRAIN_FUSE_ADAM=1 timeout 600 python bench.py --legs headline,train > "$OUT/bench_train_fused$i.json" 2> "$OUT/bench_train_fused$i.err"; echo "bench fused$i rc=$?" | tee -a "$OUT/steps.txt"
done
export TMPDIR=/tmp
GS_TRAIN_FUSE_ADAM=1 timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace_fused" -o t -- python bench.py --legs headline,train --quick > "$OUT/trace_fused.json" 2> "$OUT/trace_fused.err"; echo "trace fused rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace_unfused" -o t -- python bench.py --legs headline,train --quick > "$OUT/trace_unfused.json" 2> "$OUT/trace_unfused.err"; echo "trace unfused rc=$?" | tee -a "$OUT/steps.txt"
# keep the merge small: the per-dispatch csv only for the kernels of interest
for d in trace_fused trace_unfused; do
  f=$(find "$OUT/$d" -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && { head -1 "$f" > "$OUT/$d.kernels.csv"; grep -E 'frame_project_backward|adam_kernel' "$f" >> "$OUT/$d.kernels.csv"; rm -f "$f"; }
done
tail -n 12 "$OUT/pytest.log" | cut -c1-300
