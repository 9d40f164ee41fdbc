#!/bin/bash
# round 4, call f: full GPU suite with the new tests (prints: full-size gradient parity, trajectory pin), default bench
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4f; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=10 -s > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python bench.py --legs headline,train,cfg4,pipelined > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?" | tee -a "$OUT/steps.txt"
grep -E "gradient parity|trajectory|loss:|travelled|passed|failed|FAILED|Error" "$OUT/pytest.log" | cut -c1-900 | tail -60
python - <<'P'
import json
d=json.load(open("gpurun_out/r4f/bench.json"))
print("value",d["value"],"ms",d["ms_per_step"], d["ms_per_step_min"], d["ms_per_step_max"])
print(json.dumps(d["stages"]))
ex=d["extra"]
for k,v in ex.items():
    if k.startswith("train"): print(k, {kk:vv for kk,vv in v.items() if kk not in ("step","protocol","adam_roofline")})
    elif k=="cfg4_2p4M_sh_fwd_bwd":
        for kk,vv in v.items(): print(kk, {a:b for a,b in vv.items() if not a.startswith("roofline")})
    else: print(k, v)
print(d.get("leg_errors"))
P
