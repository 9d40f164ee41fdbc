#!/bin/bash
# project + count with frustum compaction: frame parity tests (records / lists bit-exact), then same-box A/B against the uncompacted kernel
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r3ai; rm -rf "$OUT"; mkdir -p "$OUT"; cd "$R"
timeout 1500 python -m pytest tests/test_gpu_frame.py tests/test_gpu_train.py -m gpu -q -x -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest.log"
timeout 600 python tools/ab_variants.py run cfg5_fwd cfg2_fwd 2>&1 | grep -v amdgpu.ids | cut -c1-250 | tee "$OUT/ab.txt"
