#!/bin/bash
# round 6, call g: cull tests after the policy fixes; the densifying SH run's end state (tools/soak_end_state.py); bench headline with the
# moving-camera figure
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6g; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_frame.py -q -m gpu -s -k "occlusion or sort_window or long_list or pile or graph" > "$OUT/cull_tests.txt" 2>&1; echo "cull tests rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python tools/soak_end_state.py 2 2000 > "$OUT/soak_end_sh2.json" 2> "$OUT/soak_end_sh2.err"; echo "soak end sh2 rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python tools/soak_end_state.py 0 3000 > "$OUT/soak_end_rgb.json" 2> "$OUT/soak_end_rgb.err"; echo "soak end rgb rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python bench.py --legs headline,cfg2,cfg1 > "$OUT/bench_headline.json" 2> "$OUT/bench.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
cat "$OUT/steps.txt"; tail -4 "$OUT/cull_tests.txt"; grep "occlusion cull over" "$OUT/cull_tests.txt"; cat "$OUT/soak_end_sh2.json" "$OUT/soak_end_rgb.json"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6g/bench_headline.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("latency_fps"), d.get("occlusion_cull"), d.get("moving_camera"), {k:v["ms"] for k,v in d["stages"].items()}, "cfg2", d["cfg2"]["render_fps"], d["cfg2"].get("moving_camera"), d["cfg2"]["occlusion_cull"], "cfg1", d["cfg1"]["render_fps"], d.get("leg_errors"))
PY
