#!/bin/bash
# round 5, call d: why does the row-layout rgb backward win 13 % by tools/stage_profile.py and nothing in bench.py's train leg?
# Kernel traces (true kernel durations) of the cfg5 forward + backward for GS_BWD_RGB_ROWS = 2 (in-tree) / 0 / 1, and
# bench.py's training legs for the three builds on one box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5d; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for V in base rows0 rows1; do
  L=""; [ "$V" != base ] && L="$R/build/variants/$V/libgs_amd.so"
  GS_AMD_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$V" -o s -- python "$R/tools/prof_target.py" cfg5 --backward --frames 60 > "$OUT/target_$V.json" 2> "$OUT/trace_$V.err"
  cp $(find "$OUT/trace_$V" -name '*kernel_stats.csv' | head -1) "$OUT/kernel_stats_$V.csv"
done
cd "$R"
for V in base rows0 rows1; do
  L=""; [ "$V" != base ] && L="$R/build/variants/$V/libgs_amd.so"
  GS_AMD_LIB=$L timeout 600 python bench.py --legs headline,train > "$OUT/bench_train_$V.json" 2> "$OUT/bench_train_$V.err"; echo "bench $V rc=$?" | tee -a "$OUT/steps.txt"
done
for V in base rows0 rows1; do echo "== $V"; head -8 "$OUT/kernel_stats_$V.csv" | cut -c1-60,200-330; cat "$OUT/target_$V.json"; echo; done
