#!/bin/bash
# round 6, call x: dilated cuts under a moving camera -- finer depth scale sweep over pan speeds
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6x; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
for sc in 1.125 1.25 1.375; do
  for s in 0.01 0.05 0.2 0.5 1.0; do
    echo "scale $sc radius 1 step $s" | tee -a "$OUT/moving.jsonl"
    GS_CULL_DILATE_SCALE=$sc GS_FRAME_CULL_MAX_SHIFT_PX=64 timeout 300 python tools/cull_moving.py $s 120 2>/dev/null | tee -a "$OUT/moving.jsonl"
  done
done
for s in 0.5 1.0; do echo "unculled step $s"; GS_FRAME_CULL_MAX_SHIFT_PX=0 timeout 300 python tools/cull_moving.py $s 120 2>/dev/null | tee -a "$OUT/moving.jsonl"; done
