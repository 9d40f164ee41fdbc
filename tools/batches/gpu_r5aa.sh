#!/bin/bash
# round 5, call aa: the bucket work list in two kernels (scan in one workgroup, records by a wave per tile; in-tree) against the one
# kernel that did both (build/variants/scan1wg): backward tests, kernel trace of whole training steps, training legs + rgb soak alternating
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5aa; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests/test_gpu_frame.py tests/test_gpu_train.py tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_splatter.py -m gpu -q -rf --maxfail=30 -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_new" -o s -- python "$R/tools/prof_target.py" cfg5 --train --frames 60 > "$OUT/trace_new.json" 2> "$OUT/trace_new.err"); echo "trace new rc=$?" | tee -a "$OUT/steps.txt"
(cd /tmp && GS_AMD_LIB=$R/build/variants/scan1wg/libgs_amd.so rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_old" -o s -- python "$R/tools/prof_target.py" cfg5 --train --frames 60 > "$OUT/trace_old.json" 2> "$OUT/trace_old.err"); echo "trace old rc=$?" | tee -a "$OUT/steps.txt"
for d in trace_new trace_old; do cp $(find "$OUT/$d" -name '*kernel_stats.csv' | head -1) "$OUT/$d.kernel_stats.csv"; rm -rf "$OUT/$d"; done
for i in 1 2; do
  timeout 600 python bench.py --legs headline,train > "$OUT/bench_train_new_$i.json" 2> "$OUT/bench_train_new_$i.err"; echo "bench new $i rc=$?" | tee -a "$OUT/steps.txt"
  GS_AMD_LIB=$R/build/variants/scan1wg/libgs_amd.so timeout 600 python bench.py --legs headline,train > "$OUT/bench_train_old_$i.json" 2> "$OUT/bench_train_old_$i.err"; echo "bench old $i rc=$?" | tee -a "$OUT/steps.txt"
  timeout 200 python tools/soak.py 0 3000 0 > "$OUT/soak_new_$i.json" 2> "$OUT/soak_new_$i.err"
  GS_AMD_LIB=$R/build/variants/scan1wg/libgs_amd.so timeout 200 python tools/soak.py 0 3000 0 > "$OUT/soak_old_$i.json" 2> "$OUT/soak_old_$i.err"
done
tail -n 8 "$OUT/pytest.log" | cut -c1-300
