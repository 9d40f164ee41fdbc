#!/bin/bash
# round 4, call u: regression sweep of the features this round did not touch, on the final tree: a 100,000-Gaussian pile
# (long-list kernels), a moving camera, 10 M Gaussians forward, cfg2 / cfg5 / cfg4 stage times
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4u; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 300 python tools/long_list.py > "$OUT/long_list.txt" 2>&1; echo "long_list rc=$?" | tee -a "$OUT/steps.txt"
timeout 300 python tools/moving_camera_fps.py cfg2 cfg5 > "$OUT/moving_camera.txt" 2>&1; echo "moving rc=$?" | tee -a "$OUT/steps.txt"
timeout 400 python tools/stage_profile.py cfg6_fwd cfg2 cfg5 cfg4 cfg4_deg3 > "$OUT/stage.txt" 2>&1; echo "stage rc=$?" | tee -a "$OUT/steps.txt"
grep -v "^RCCL\|^HIP\|^ROCm\|amdgpu.ids" "$OUT/long_list.txt" | tail -8 | cut -c1-300; grep -v "amdgpu.ids" "$OUT/moving_camera.txt" | tail -6 | cut -c1-300; grep -v "amdgpu.ids" "$OUT/stage.txt" | cut -c1-330
