#!/bin/bash
# round 5, call z: kernel trace + PMC of whole TRAINING steps (tools/prof_target.py --train: forward, loss, backward with the optimizer
# step fused in) at 2.4 M and at 376 k Gaussians
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5z; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 600 bash tools/profile_round.sh r5z_cfg5_train cfg5 train > "$OUT/profile_cfg5_train.txt" 2>&1; echo "cfg5 rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 bash tools/profile_round.sh r5z_cfg2_train cfg2 train > "$OUT/profile_cfg2_train.txt" 2>&1; echo "cfg2 rc=$?" | tee -a "$OUT/steps.txt"
cd "$R"; cat "$OUT/steps.txt"; cat gpurun_out/r5z_cfg5_train/target.json; tail -3 gpurun_out/r5z_cfg5_train/stats.err
