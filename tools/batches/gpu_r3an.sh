#!/bin/bash
# SH forward compositing with a wave-uniform branch around finished pixel pairs: parity tests, then same-box A/B
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r3an; rm -rf "$OUT"; mkdir -p "$OUT"; cd "$R"
timeout 1500 python -m pytest tests/test_gpu_frame.py tests/test_gpu_kernels.py tests/test_gpu_golden.py -m gpu -q -x -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log" | head -2
timeout 900 python tools/ab_variants.py run cfg4 cfg4_deg3 2>&1 | grep -v amdgpu.ids | cut -c1-22,100-330 | tee "$OUT/ab.txt"
