#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r3d; mkdir -p "$OUT"; cd "$R"
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
for rep in 1 2; do timeout 600 python tools/stage_profile.py cfg4 cfg4_deg3 2>&1 | grep -v amdgpu.ids; done | tee "$OUT/stages_sh.txt"
echo "== r01h tree, same box"; for rep in 1 2; do timeout 300 python build/r01h/tools/sweep_n.py 10000 100000 2>/dev/null; done | tee "$OUT/sweep_r01h_tree.jsonl"
echo "== current tree"; for rep in 1 2; do timeout 300 python tools/sweep_n.py 10000 100000 2>/dev/null; done | tee "$OUT/sweep_current.jsonl"
timeout 120 tools/ubench/mfma_reduce 2>&1 | tee "$OUT/ubench_mfma_reduce.txt"
