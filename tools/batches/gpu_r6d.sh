#!/bin/bash
# round 6, call d: occlusion cull with the cut table in LDS; cfg4 API parity with the basis-conditioned scale; headline bench A/B
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6d; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_frame.py -q -m gpu -s -k "occlusion or 2p4M_forward" > "$OUT/cull_tests.txt" 2>&1; echo "cull tests rc=$?" | tee -a "$OUT/steps.txt"
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -m gpu -s -k "full_size" > "$OUT/api_tests.txt" 2>&1; echo "api tests rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python bench.py --legs headline,cfg2,cfg1 --steps 20 --warmup 5 > "$OUT/bench_headline.json" 2> "$OUT/bench.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
GS_NO_CULL=1 timeout 900 python bench.py --legs headline,cfg2,cfg1 --steps 20 --warmup 5 > "$OUT/bench_headline_nocull.json" 2> "$OUT/bench_nocull.err"; echo "bench nocull rc=$?" | tee -a "$OUT/steps.txt"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/st_cfg5" -o s -- python "$R/tools/prof_target.py" cfg5 --frames 100 > "$OUT/target_cfg5.json" 2> "$OUT/st_cfg5.err"
F=$(find "$OUT/st_cfg5" -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" "$OUT/kernel_stats_cfg5_culled.csv"; rm -rf "$OUT/st_cfg5"
cd "$R"
cat "$OUT/steps.txt"; tail -8 "$OUT/cull_tests.txt"; grep "occlusion cull" "$OUT/cull_tests.txt"; tail -4 "$OUT/api_tests.txt"
python - <<'PY'
import json, csv
for f in ("bench_headline.json","bench_headline_nocull.json"):
    try:
        d=json.loads(open("gpurun_out/r6d/"+f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("latency_fps"), d.get("occlusion_cull"), {k:v["ms"] for k,v in d["stages"].items()}, d["stage_total_ms"], "cfg2", d.get("cfg2",{}).get("render_fps"), d.get("cfg2",{}).get("occlusion_cull"), "cfg1", d.get("cfg1",{}).get("render_fps"), d.get("leg_errors"))
    except Exception as e: print(f, "ERR", e)
for r in list(csv.DictReader(open("gpurun_out/r6d/kernel_stats_cfg5_culled.csv")))[:8]:
    print(r["Name"].replace("(anonymous namespace)::","")[:70], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
