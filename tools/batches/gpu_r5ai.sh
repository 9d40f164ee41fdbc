#!/bin/bash
# round 5, call ai: the final tree over the scene size (tools/sweep_n.py) and the 7,001-iteration fit of BASELINE configs[2] in miniature
# (tools/train_demo.py: the fused optimizer step is what its single-rank rgb Trainer takes), with and without the fused step
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5ai; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 900 python tools/sweep_n.py > "$OUT/sweep_n.jsonl" 2> "$OUT/sweep_n.err"; echo "sweep rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 python tools/train_demo.py > "$OUT/train_demo_fused.json" 2> "$OUT/train_demo_fused.err"; echo "demo fused rc=$?" | tee -a "$OUT/steps.txt"
GS_TRAIN_FUSE_ADAM=0 timeout 600 python tools/train_demo.py > "$OUT/train_demo_unfused.json" 2> "$OUT/train_demo_unfused.err"; echo "demo unfused rc=$?" | tee -a "$OUT/steps.txt"
cat "$OUT/steps.txt"; cat "$OUT/train_demo_fused.json" "$OUT/train_demo_unfused.json" | cut -c1-600
