#!/bin/bash
# round 5, call t: the row-layout rgb backward with its per-pixel algebra on pixel PAIRS (GS_BWD_ROWS_PK 1, in-tree) against the
# first version (build/variants/rows_pk0): the frame tests, stage times on one box (tools/ab_variants.py), the training legs
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5t; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1200 python -m pytest tests/test_gpu_frame.py tests/test_gpu_train.py -m gpu -q -rf --maxfail=30 -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 python tools/ab_variants.py run cfg5_rows cfg2_rows > "$OUT/ab.txt" 2> "$OUT/ab.err"; echo "ab rc=$?" | tee -a "$OUT/steps.txt"
for i in 1 2; do
  timeout 600 python bench.py --legs headline,train > "$OUT/bench_train_pk1_$i.json" 2> "$OUT/bench_train_pk1_$i.err"; echo "bench pk1 $i rc=$?" | tee -a "$OUT/steps.txt"
  GS_AMD_LIB=$R/build/variants/rows_pk0/libgs_amd.so timeout 600 python bench.py --legs headline,train > "$OUT/bench_train_pk0_$i.json" 2> "$OUT/bench_train_pk0_$i.err"; echo "bench pk0 $i rc=$?" | tee -a "$OUT/steps.txt"
done
tail -n 8 "$OUT/pytest.log" | cut -c1-300
