#!/bin/bash
# round 6, call b: the fixed parity tests (calibration statistic, API full size on torch-activated inputs, sigmoid squash in double),
# GS_FRAME_LONG_SORT, eight ranks on one GPU; the whole GPU tier; bench legs headline + trained
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6b; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests/test_grad_calibration.py tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_compat_pipeline.py -q -m gpu -s > "$OUT/new_tests.txt" 2>&1; echo "new tests rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python -m pytest tests/test_gpu_frame.py -q -m gpu -k "sort_window or long_list or pile" > "$OUT/long_sort_tests.txt" 2>&1; echo "long sort tests rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -k "ranks_on_one_gpu" > "$OUT/ranks_tests.txt" 2>&1; echo "ranks tests rc=$?" | tee -a "$OUT/steps.txt"
timeout 1800 python -m pytest tests -q -m gpu -s > "$OUT/gpu_tier.txt" 2>&1; echo "gpu tier rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python bench.py --legs headline,trained --steps 20 --warmup 5 > "$OUT/bench_headline_trained.json" 2> "$OUT/bench.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
cat "$OUT/steps.txt"; grep -h "CALIB" "$OUT/new_tests.txt" | cut -c1-330 | head -60; tail -5 "$OUT/new_tests.txt"; tail -3 "$OUT/long_sort_tests.txt"; tail -3 "$OUT/ranks_tests.txt"; grep -n "FAILED\|passed\|failed" "$OUT/gpu_tier.txt" | tail -12
