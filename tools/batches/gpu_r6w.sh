#!/bin/bash
# round 6, call w: dilated cuts under a moving camera -- depth scale and radius sweep
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6w; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
for cfg in "1.0 1" "1.25 1" "1.5 1" "2.0 1" "1.25 2" "1.5 2" "2.0 2"; do
  set -- $cfg
  for s in 0.05 0.2; do
    echo "scale $1 radius $2 step $s" | tee -a "$OUT/moving.jsonl"
    GS_CULL_DILATE_SCALE=$1 GS_CULL_DILATE_RADIUS=$2 GS_FRAME_CULL_MAX_SHIFT_PX=16 timeout 300 python tools/cull_moving.py $s 120 2>/dev/null | tee -a "$OUT/moving.jsonl"
  done
done
