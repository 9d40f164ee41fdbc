#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r3g; mkdir -p "$OUT"; cd "$R"
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
for rep in 1 2; do timeout 600 python tools/stage_profile.py cfg5_fwd cfg2_fwd cfg5 2>&1 | grep -v amdgpu.ids; done | tee "$OUT/stages.txt"
echo "== r01h tree"; timeout 300 python build/r01h/tools/sweep_n.py 10000 100000 2>/dev/null | tee "$OUT/sweep_r01h_tree.jsonl"
echo "== current tree"; timeout 300 python tools/sweep_n.py 10000 100000 200000 2>/dev/null | tee "$OUT/sweep_current.jsonl"
