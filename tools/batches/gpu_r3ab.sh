#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r3ab; rm -rf "$OUT"; mkdir -p "$OUT"; cd "$R"
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -p no:cacheprovider -k "loss" > "$OUT/pytest_loss.log" 2>&1; echo "pytest loss rc=$?"; tail -5 "$OUT/pytest_loss.log"
timeout 600 python tools/loss_ab.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/loss_ab.txt"
