#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r3m; mkdir -p "$OUT"; cd "$R"
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
echo "== r01h tree"; timeout 300 python build/r01h/tools/sweep_n.py 10000 100000 2>/dev/null | tee "$OUT/sweep_r01h_tree.jsonl"
echo "== current tree"; timeout 300 python tools/sweep_n.py 10000 50000 100000 130000 2>/dev/null | tee "$OUT/sweep_current.jsonl"
