#!/bin/bash
# round 5, call s: fused optimizer step, third version (every load of the step first) against the second (build/variants/adam_v2):
# the fused tests, then bench.py's training legs with either library alternating on one box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5s; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -rf --maxfail=30 -p no:cacheprovider -k "fused or trainer" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
for i in 1 2; do
  timeout 600 python bench.py --legs headline,train > "$OUT/bench_train_v3_$i.json" 2> "$OUT/bench_train_v3_$i.err"; echo "bench v3 $i rc=$?" | tee -a "$OUT/steps.txt"
  GS_AMD_LIB=$R/build/variants/adam_v2/libgs_amd.so timeout 600 python bench.py --legs headline,train > "$OUT/bench_train_v2_$i.json" 2> "$OUT/bench_train_v2_$i.err"; echo "bench v2 $i rc=$?" | tee -a "$OUT/steps.txt"
done
tail -n 6 "$OUT/pytest.log" | cut -c1-300
