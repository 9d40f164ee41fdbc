#!/bin/bash
# round 5, call c: rgb backward with TWO work lists (row-layout kernel for the tiles whose pixels all saturated, pixel-parallel
# kernel for the others; GS_BWD_RGB_ROWS = 2), fused DPP blocks, PF = 0: GPU suite, same-box A/B against one kernel for all
# buckets (rows0 / rows1), then the training legs of the bench
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5c; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -q -rf --maxfail=30 -p no:cacheprovider -s > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
for rep in 0 1; do
  for V in base rows0 rows1; do
    L=""; [ "$V" != base ] && L="$R/build/variants/$V/libgs_amd.so"
    GS_AMD_LIB=$L timeout 300 python tools/stage_profile.py cfg5 cfg2 cfg3 2>> "$OUT/ab.err" | sed "s/^/[$V #$rep] /" >> "$OUT/ab_rgb.txt"
  done
done
echo "ab rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python bench.py --legs headline,train,cfg4,soak > "$OUT/bench_train.json" 2> "$OUT/bench_train.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
GS_AMD_LIB=$R/build/variants/rows0/libgs_amd.so timeout 900 python bench.py --legs headline,train > "$OUT/bench_train_rows0.json" 2> "$OUT/bench_train_rows0.err"; echo "bench0 rc=$?" | tee -a "$OUT/steps.txt"
cat "$OUT/steps.txt"; tail -n 4 "$OUT/pytest.log"; cat "$OUT/ab_rgb.txt" | cut -c1-400
