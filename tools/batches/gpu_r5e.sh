#!/bin/bash
# round 5, call e: the bucket kernels walk their work list with the grid as the stride, grids capped at 40,960 waves (the empty
# waves of a capacity-sized grid cost 22 - 30 us per launch: r5d); GS_BWD_RGB_ROWS = 2 (two lists) / 0 / 1 again: GPU suite,
# kernel traces at cfg5 and cfg2, bench.py's training legs, all on one box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5e; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -q -rf --maxfail=30 -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
cd /tmp && export TMPDIR=/tmp
for C in cfg5 cfg2; do
for V in base rows0 rows1; do
  L=""; [ "$V" != base ] && L="$R/build/variants/$V/libgs_amd.so"
  GS_AMD_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_${C}_$V" -o s -- python "$R/tools/prof_target.py" $C --backward --frames 60 > "$OUT/target_${C}_$V.json" 2> "$OUT/trace_${C}_$V.err"
  cp $(find "$OUT/trace_${C}_$V" -name '*kernel_stats.csv' | head -1) "$OUT/kernel_stats_${C}_$V.csv"
  rm -rf "$OUT/trace_${C}_$V"
done
done
cd "$R"
for V in base rows0 rows1; do
  L=""; [ "$V" != base ] && L="$R/build/variants/$V/libgs_amd.so"
  GS_AMD_LIB=$L timeout 600 python bench.py --legs headline,train > "$OUT/bench_train_$V.json" 2> "$OUT/bench_train_$V.err"; echo "bench $V rc=$?" | tee -a "$OUT/steps.txt"
done
tail -n 3 "$OUT/pytest.log"
