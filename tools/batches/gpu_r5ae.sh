#!/bin/bash
# round 5, call ae: the final tree (bucket work list in two kernels, events without the system fence): whole GPU suite, smoke, default
# bench, kernel trace + PMC of whole training steps at 2.4 M and 376 k Gaussians
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5ae; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -q -rf --maxfail=30 -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
timeout 300 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 bash tools/profile_round.sh r5ae_cfg5_train cfg5 train > "$OUT/profile_cfg5_train.txt" 2>&1; echo "cfg5 train rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 bash tools/profile_round.sh r5ae_cfg2_train cfg2 train > "$OUT/profile_cfg2_train.txt" 2>&1; echo "cfg2 train rc=$?" | tee -a "$OUT/steps.txt"
cd "$R"; cat "$OUT/steps.txt"; tail -n 12 "$OUT/pytest.log" | cut -c1-300; tail -n 2 "$OUT/smoke.log"
