#!/bin/bash
# round 5, call g (= call f after the choice moved to synchronous stats only): the rgb backward kernel chosen per FRAME by the caller's flag (GS_FRAME_BWD_ROWS), FrameRenderer latches it
# from the share of saturated buckets: GPU suite (new: row layout vs oracle, the latch), kernel traces at cfg5 / cfg2 with
# the renderer's own choice, bench.py's training legs
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5g; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -q -rf --maxfail=30 -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
cd /tmp && export TMPDIR=/tmp
for C in cfg5 cfg2; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$C" -o s -- python "$R/tools/prof_target.py" $C --backward --frames 60 > "$OUT/target_$C.json" 2> "$OUT/trace_$C.err"
  cp $(find "$OUT/trace_$C" -name '*kernel_stats.csv' | head -1) "$OUT/kernel_stats_$C.csv"; rm -rf "$OUT/trace_$C"
done
cd "$R"
timeout 600 python bench.py --legs headline,train > "$OUT/bench_train.json" 2> "$OUT/bench_train.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
tail -n 15 "$OUT/pytest.log" | cut -c1-300; cat "$OUT"/target_*.json
