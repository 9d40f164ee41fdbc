#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r3i; mkdir -p "$OUT"; cd "$R"
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v amdgpu.ids | tail -3
timeout 900 python tools/ab_variants.py run cfg4 cfg4_deg3 > "$OUT/ab.txt" 2> "$OUT/ab.err"; grep -v amdgpu "$OUT/ab.txt"
