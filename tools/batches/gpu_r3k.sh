#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r3k; mkdir -p "$OUT"; cd "$R"
timeout 1500 python -m pytest tests/test_gpu_frame.py -m gpu -q -x -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest.log"
for rep in 1 2; do timeout 600 python tools/stage_profile.py ${STAGE_CFGS:-cfg5_fwd cfg2_fwd cfg5_fwd_keys} 2>&1 | grep -v amdgpu.ids; done | tee "$OUT/stages.txt"
