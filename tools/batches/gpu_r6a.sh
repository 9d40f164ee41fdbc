#!/bin/bash
# round 6, call a: the new parity tests (reference API element-wise + full size, calibration against the reference kernels' own error),
# the whole GPU tier, the trained-state scene with the long-list flag off / on under a kernel trace, bench legs headline + trained
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6a; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests/test_grad_calibration.py tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_compat_pipeline.py -q -m gpu -s > "$OUT/new_tests.txt" 2>&1; echo "new tests rc=$?" | tee -a "$OUT/steps.txt"
timeout 1500 python -m pytest tests -x -q -m gpu -s > "$OUT/gpu_tier.txt" 2>&1; echo "gpu tier rc=$?" | tee -a "$OUT/steps.txt"
cd /tmp && export TMPDIR=/tmp
for CFG in trained_rgb trained_sh; do for LL in off on; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/st_${CFG}_$LL" -o s -- python "$R/tools/prof_target.py" $CFG --backward --frames 40 --long-lists $LL > "$OUT/target_${CFG}_$LL.json" 2> "$OUT/st_${CFG}_$LL.err"
  cp $(find "$OUT/st_${CFG}_$LL" -name '*kernel_stats.csv' | head -1) "$OUT/kernel_stats_${CFG}_$LL.csv"; rm -rf "$OUT/st_${CFG}_$LL"
done; done
cd "$R"
timeout 900 python bench.py --legs headline,cfg2,trained --steps 20 --warmup 5 > "$OUT/bench_headline_trained.json" 2> "$OUT/bench.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
cat "$OUT/steps.txt"; grep -h "CALIB" "$OUT/new_tests.txt" | head -80; tail -5 "$OUT/new_tests.txt"; tail -3 "$OUT/gpu_tier.txt"
