#!/bin/bash
# round 6, call f: GPU tier on the tree with the cost-model flag; bench legs headline + trained + soak
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6f; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1800 python -m pytest tests -q -m gpu > "$OUT/gpu_tier.txt" 2>&1; echo "gpu tier rc=$?" | tee -a "$OUT/steps.txt"
timeout 1500 python bench.py --legs headline,trained,soak > "$OUT/bench_soak.json" 2> "$OUT/bench.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
cat "$OUT/steps.txt"; grep -n "FAILED\|passed\|failed" "$OUT/gpu_tier.txt" | tail -12
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6f/bench_soak.json").read().strip().splitlines()[-1])
print(d["value"], d.get("leg_errors"))
for k,v in d["extra"]["soak_densifying"].items(): print(k, v["iters_per_s"], v["iters_per_s_blocks"])
ts=d["extra"]["trained_state"]
for tag in ts:
    for mode in ("auto","flag_off","flag_on"):
        r=ts[tag][mode]; print(tag, mode, r["flagged_long_lists"], r.get("flagged_long_sort"), r["render_fps"], r["fwd_bwd_iters_per_s"])
PY
