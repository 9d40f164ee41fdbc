#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r3e; mkdir -p "$OUT"; cd "$R"
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
for rep in 1 2; do timeout 600 python tools/stage_profile.py cfg4 cfg4_deg3 cfg5 2>&1 | grep -v amdgpu.ids; done | tee "$OUT/stages_sh.txt"
timeout 300 python tools/adam_bw.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/adam.jsonl"
timeout 900 tools/profile_round.sh r3e/prof_cfg5 cfg5 fwd > "$OUT/prof_cfg5.log" 2>&1; tail -30 "$OUT/prof_cfg5.log" | cut -c1-250
