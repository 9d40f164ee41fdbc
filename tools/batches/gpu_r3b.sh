#!/bin/bash
# round 3, call B: full GPU suite + variant sweep (table / strip by size) + SH stage times + zero-change segments
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r3b; mkdir -p "$OUT"; cd "$R"
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider -s > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"; grep -E "gradient parity" "$OUT/pytest.log" | cut -c1-900
for V in --table --strip; do timeout 600 python tools/sweep_n.py $V 10000 100000 200000 300000 376467 506627 2>/dev/null; done > "$OUT/sweep_variants.jsonl"; cat "$OUT/sweep_variants.jsonl"
AB_CFGS="cfg4 cfg4_deg3 cfg5" timeout 900 python tools/ab_variants.py run cfg4 cfg4_deg3 cfg5 > "$OUT/ab.txt" 2> "$OUT/ab.err"; grep -v amdgpu "$OUT/ab.txt"
timeout 600 python tools/compat_fps.py cfg5 > "$OUT/compat_cfg5.txt" 2>&1; grep -v amdgpu "$OUT/compat_cfg5.txt"
