#!/bin/bash
# round 6, call l: compacting project kernel of the occlusion-culled frame -- cull tests, headline bench with / without
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6l; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_frame.py -q -m gpu -x -k "occlusion or 2p4M_forward" > "$OUT/cull_tests.txt" 2>&1; echo "cull tests rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python bench.py --legs headline > "$OUT/bench_headline.json" 2> "$OUT/bench.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
GS_NO_CULL=1 timeout 900 python bench.py --legs headline > "$OUT/bench_headline_nocull.json" 2> "$OUT/bench_nocull.err"; echo "bench nocull rc=$?" | tee -a "$OUT/steps.txt"
tail -15 "$OUT/cull_tests.txt"
python - <<'PY'
import json
for f in ("bench_headline.json","bench_headline_nocull.json"):
    d=json.loads(open("gpurun_out/r6l/"+f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d.get("latency_fps"), d.get("occlusion_cull"), d["moving_camera"]["fps"], d["moving_camera"]["culled"], {k:v["ms"] for k,v in d["stages"].items()}, d["stage_total_ms"])
PY
