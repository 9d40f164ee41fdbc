#!/bin/bash
# round 6, call j: the fused backward + Adam step with SH colours -- parity tests, then the trained / soak legs with and without
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6j; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -k "fused_backward_adam" > "$OUT/fused_tests.txt" 2>&1; echo "fused tests rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python -m pytest tests/test_gpu_frame.py -q -m gpu -k "occlusion" > "$OUT/cull_tests.txt" 2>&1; echo "cull tests rc=$?" | tee -a "$OUT/steps.txt"
timeout 1200 python bench.py --legs headline,trained,soak > "$OUT/bench_fused.json" 2> "$OUT/bench_fused.err"; echo "bench fused rc=$?" | tee -a "$OUT/steps.txt"
GS_TRAIN_FUSE_ADAM=0 timeout 1200 python bench.py --legs headline,trained,soak > "$OUT/bench_unfused.json" 2> "$OUT/bench_unfused.err"; echo "bench unfused rc=$?" | tee -a "$OUT/steps.txt"
cat "$OUT/steps.txt"; tail -5 "$OUT/fused_tests.txt"; tail -3 "$OUT/cull_tests.txt"
python - <<'PY'
import json
for f in ("bench_fused.json","bench_unfused.json"):
    d=json.loads(open("gpurun_out/r6j/"+f).read().strip().splitlines()[-1])
    ts=d.get("trained_state",{})
    print(f, d["value"], d["moving_camera"]["fps"], d["moving_camera"].get("culled"))
    print("  trained:", json.dumps(ts)[:1500])
    print("  soak:", json.dumps({k:(v.get("iters_per_s"), v.get("iters_per_s_blocks")) if isinstance(v,dict) else v for k,v in d.get("soak",{}).items()})[:1200])
PY
