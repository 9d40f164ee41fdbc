#!/bin/bash
# round 4, call c: the slice pipeline on hardware: full GPU test-suite, then bench.py through RCCL on one rank
# (--force-collective: multi_gpu leg with exposed_ms on the rgb and the SH scene), stage times for the reworked
# projection backward
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4c; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
timeout 300 python tools/stage_profile.py cfg5 cfg2 > "$OUT/stage.txt" 2>&1
echo "stage rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python bench.py --force-collective --legs headline,train,multi_gpu > "$OUT/bench_fc.json" 2> "$OUT/bench_fc.err"
echo "bench rc=$?" | tee -a "$OUT/steps.txt"
tail -15 "$OUT/pytest.log"; cat "$OUT/stage.txt"; tail -5 "$OUT/bench_fc.err"
python - <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r4c/bench_fc.json"))
    print("value",d["value"],"ms",d["ms_per_step"])
    print(json.dumps(d.get("multi_gpu"),indent=1)[:3500])
    print(json.dumps({k:{kk:vv for kk,vv in v.items() if kk in("iters_per_s","ms_per_iter","ms_per_iter_min","ms_per_iter_max","repeats","backward_stage_ms","forward_ms","loss_ms","adam_ms")} for k,v in d["extra"].items() if k.startswith("train")},indent=1))
    print(d.get("leg_errors"))
except Exception as e: print("no json",e)
P
