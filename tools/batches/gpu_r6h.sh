#!/bin/bash
# round 6, call h: the whole GPU tier + smoke + the default bench (all legs) + kernel traces / PMC of the culled cfg5 forward and of the
# trained-like SH scene (forward + backward)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6h; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1800 python -m pytest tests -q -m gpu > "$OUT/gpu_tier.txt" 2>&1; echo "gpu tier rc=$?" | tee -a "$OUT/steps.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.txt" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/steps.txt"
timeout 1500 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 bash tools/profile_round.sh r6h_cfg5_fwd cfg5 fwd > "$OUT/profile_cfg5.txt" 2>&1; echo "profile cfg5 rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 bash tools/profile_round.sh r6h_trained_sh trained_sh fwdbwd > "$OUT/profile_trained_sh.txt" 2>&1; echo "profile trained_sh rc=$?" | tee -a "$OUT/steps.txt"
cd "$R"; cat "$OUT/steps.txt"; grep -n "FAILED\|passed\|failed" "$OUT/gpu_tier.txt" | tail -8; tail -2 "$OUT/smoke.txt"
