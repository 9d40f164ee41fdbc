#!/bin/bash
# round 6, call p: whole GPU tier + smoke + default bench + kernel trace / PMC of the culled cfg5 forward (survivor-list tree)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6p; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1800 python -m pytest tests -q -m gpu > "$OUT/gpu_tier.txt" 2>&1; echo "gpu tier rc=$?" | tee -a "$OUT/steps.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.txt" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/steps.txt"
timeout 1500 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 bash tools/profile_round.sh r6p_cfg5_fwd cfg5 fwd > "$OUT/profile_cfg5.txt" 2>&1; echo "profile cfg5 rc=$?" | tee -a "$OUT/steps.txt"
cd "$R"; cat "$OUT/steps.txt"; grep -n "FAILED\|passed\|failed" "$OUT/gpu_tier.txt" | tail -8; tail -2 "$OUT/smoke.txt"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6p/bench_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","latency_fps","occlusion_cull","stages","stage_total_ms") if k in d})
print(d["roofline"])
print(d["moving_camera"])
PY
ls gpurun_out | head -30
