#!/bin/bash
# round 6, call k: SH fused Adam (float4 walk) -- parity tests, soak leg with / without
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6k; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -k "fused_backward_adam" > "$OUT/fused_tests.txt" 2>&1; echo "fused tests rc=$?" | tee -a "$OUT/steps.txt"
for i in 1 2; do
timeout 1200 python bench.py --legs headline,soak > "$OUT/bench_fused$i.json" 2> "$OUT/bench_fused.err"; echo "bench fused rc=$?" | tee -a "$OUT/steps.txt"
GS_TRAIN_FUSE_ADAM=0 timeout 1200 python bench.py --legs headline,soak > "$OUT/bench_unfused$i.json" 2> "$OUT/bench_unfused.err"; echo "bench unfused rc=$?" | tee -a "$OUT/steps.txt"
done
tail -5 "$OUT/fused_tests.txt"
python - <<'PY'
import json
for f in ("bench_fused1.json","bench_unfused1.json","bench_fused2.json","bench_unfused2.json"):
    d=json.loads(open("gpurun_out/r6k/"+f).read().strip().splitlines()[-1])
    e=d["extra"]["soak_densifying"]
    for k,v in e.items():
        if isinstance(v,dict): print(f,k,v.get("iters_per_s"),v.get("iters_per_s_blocks")[:3],v.get("iters_per_s_blocks")[-3:])
PY
