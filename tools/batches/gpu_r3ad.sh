#!/bin/bash
# after the fused image loss and the flag-only long-list gating: GPU suite, smoke, default bench, loss kernel trace
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r3ao; rm -rf "$OUT"; mkdir -p "$OUT"; cd "$R"
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v amdgpu.ids | tail -2
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$? lines=$(wc -l < $OUT/bench_default.json)"
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/lossprof" -o s -- python "$R/tools/loss_profile.py" > "$OUT/loss.txt" 2> "$OUT/lossprof.err"; cp $(find "$OUT/lossprof" -name '*kernel_stats.csv' | head -1) "$OUT/loss_kernel_stats.csv")
cat "$OUT/loss.txt"; head -3 "$OUT/loss_kernel_stats.csv" | cut -c1-160
python - <<PY
import json
d=json.load(open("$OUT/bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
e=d["extra"]
print(json.dumps(e.get("train_cfg2"))[:400])
print(json.dumps(e.get("train_headline_scene"))[:400])
print(d.get("leg_errors"))
PY
