#!/bin/bash
# round 3 final, part 2: per-config kernel traces + PMC passes, the size sweep, the cfg3 training demo
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r3z; mkdir -p "$OUT"; cd "$R"
timeout 600 tools/profile_round.sh r3z/prof_cfg2 cfg2 fwd > "$OUT/prof_cfg2.log" 2>&1; echo "prof cfg2 rc=$?"
timeout 900 tools/profile_round.sh r3z/prof_cfg4_deg2 cfg4 fwdbwd > "$OUT/prof_cfg4_deg2.log" 2>&1; echo "prof cfg4 deg2 rc=$?"
timeout 900 tools/profile_round.sh r3z/prof_cfg4_deg3 cfg4 fwdbwd --sh-degree 3 > "$OUT/prof_cfg4_deg3.log" 2>&1; echo "prof cfg4 deg3 rc=$?"
timeout 900 python tools/sweep_n.py 2>/dev/null > "$OUT/sweep_auto.jsonl"; cat "$OUT/sweep_auto.jsonl"
timeout 600 python tools/train_demo.py 2>/dev/null | tail -1 > "$OUT/train_demo_cfg3_7k.json"; cat "$OUT/train_demo_cfg3_7k.json"
timeout 300 python bench.py --gpus 1 --force-collective --legs headline,multi_gpu 2> /dev/null | wc -l
