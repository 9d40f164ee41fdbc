#!/bin/bash
# round 4, call r: the two-ranks-on-one-GPU test (gloo, real kernels), then the whole GPU suite
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4r; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -p no:cacheprovider -x -k "two_ranks" > "$OUT/pytest_two.log" 2>&1; echo "two rc=$?" | tee -a "$OUT/steps.txt"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=5 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" "$OUT/pytest_two.log" | tail -30 | cut -c1-300; grep -E "passed|failed" "$OUT/pytest.log"
