#!/bin/bash
# round 6, call an: GPU tier, then smoke, then the default bench with the flag log (does the densifying rgb run's late-block
# slowdown of r6o / r6p come back when the bench follows the test tier on the same box?)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6an; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1800 python -m pytest tests -q -m gpu > "$OUT/gpu_tier.txt" 2>&1; echo "gpu tier rc=$?" | tee -a "$OUT/steps.txt"
ps aux | grep -c python > "$OUT/ps_after_tests.txt"; ps aux | grep python | head -20 >> "$OUT/ps_after_tests.txt"
rocm-smi --showuse --showmemuse 2>/dev/null | head -20 >> "$OUT/ps_after_tests.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.txt" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/steps.txt"
GS_LOG_FLAGS=1 timeout 1500 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
cat "$OUT/steps.txt"; grep -n "FAILED\|passed\|failed" "$OUT/gpu_tier.txt" | tail -4; tail -1 "$OUT/smoke.txt"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6an/bench_default.json").read().strip().splitlines()[-1])
v=d["extra"]["soak_densifying"]["rgb"]
print(d["value"], v["iters_per_s"], v["iters_per_s_blocks"][10:]); print(v["flags_capacity_overflows_per_block"][10:])
PY
cat "$OUT/ps_after_tests.txt" | head -12
