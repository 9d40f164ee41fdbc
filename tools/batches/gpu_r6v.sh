#!/bin/bash
# round 6, call v: the cull under a moving camera with dilated cuts -- fallback rate and rate of a pan at several speeds
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6v; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_frame.py -q -m gpu -x -k "occlusion" > "$OUT/cull_tests.txt" 2>&1; echo "cull tests rc=$?"; tail -3 "$OUT/cull_tests.txt"
for s in 0.01 0.05 0.2 0.5; do
  GS_FRAME_CULL_MAX_SHIFT_PX=0 timeout 300 python tools/cull_moving.py $s 120 2>/dev/null | tee -a "$OUT/moving.jsonl"
  GS_FRAME_CULL_MAX_SHIFT_PX=16 timeout 300 python tools/cull_moving.py $s 120 2>/dev/null | tee -a "$OUT/moving.jsonl"
done
