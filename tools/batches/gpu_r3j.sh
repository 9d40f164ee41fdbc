#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r3j; mkdir -p "$OUT"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_frame.py tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
timeout 900 python tools/ab_variants.py run cfg5_fwd cfg2_fwd cfg5 cfg2 > "$OUT/ab.txt" 2> "$OUT/ab.err"; grep -v amdgpu "$OUT/ab.txt"
