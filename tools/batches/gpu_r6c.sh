#!/bin/bash
# round 6, call c: temporal occlusion cull (GS_FRAME_OCCLUSION_CULL) + tagged stats: its tests, the parity tests again, the GPU tier,
# headline bench, kernel trace of the culled cfg5 forward
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6c; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_frame.py -q -m gpu -s -k "occlusion or sort_window or long_list or pile or 2p4M_forward" > "$OUT/cull_tests.txt" 2>&1; echo "cull tests rc=$?" | tee -a "$OUT/steps.txt"
timeout 1500 python -m pytest tests/test_grad_calibration.py tests/test_gpu_kernels.py -q -m gpu -s > "$OUT/new_tests.txt" 2>&1; echo "parity tests rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python bench.py --legs headline,cfg2,cfg1 --steps 20 --warmup 5 > "$OUT/bench_headline.json" 2> "$OUT/bench.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
GS_NO_CULL=1 timeout 900 python bench.py --legs headline --steps 20 --warmup 5 > "$OUT/bench_headline_nocull.json" 2> "$OUT/bench_nocull.err"; echo "bench nocull rc=$?" | tee -a "$OUT/steps.txt"
timeout 1800 python -m pytest tests -q -m gpu > "$OUT/gpu_tier.txt" 2>&1; echo "gpu tier rc=$?" | tee -a "$OUT/steps.txt"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/st_cfg5" -o s -- python "$R/tools/prof_target.py" cfg5 --frames 100 > "$OUT/target_cfg5.json" 2> "$OUT/st_cfg5.err"
cp $(find "$OUT/st_cfg5" -name '*kernel_stats.csv' | head -1) "$OUT/kernel_stats_cfg5_culled.csv"; rm -rf "$OUT/st_cfg5"
cd "$R"
cat "$OUT/steps.txt"; tail -15 "$OUT/cull_tests.txt"; grep -h "CALIB cfg" "$OUT/new_tests.txt" | cut -c1-260 | head -40; tail -5 "$OUT/new_tests.txt"; grep -n "FAILED\|passed\|failed" "$OUT/gpu_tier.txt" | tail -12
python - <<'PY'
import json
for f in ("bench_headline.json","bench_headline_nocull.json"):
    try:
        d=json.loads(open("gpurun_out/r6c/"+f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("latency_fps"), d.get("occlusion_cull"), {k:v["ms"] for k,v in d["stages"].items()}, d["stage_total_ms"], d.get("cfg2",{}).get("render_fps"), d.get("cfg1",{}).get("render_fps"), d.get("leg_errors"))
    except Exception as e: print(f, "ERR", e)
PY
head -14 "$OUT/kernel_stats_cfg5_culled.csv" | cut -c1-200
