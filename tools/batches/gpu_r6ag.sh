#!/bin/bash
# round 6, call ag: SH compositing kernel leaves dead half tiles out of a group of steps -- SH parity tests, cfg4 leg
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6ag; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1200 python -m pytest tests/test_gpu_frame.py tests/test_gpu_kernels.py tests/test_gpu_golden.py -q -m gpu -k "sh or SH or cfg4 or deg" > "$OUT/tests.txt" 2>&1; grep -n "passed\|failed\|FAILED" "$OUT/tests.txt"
python bench.py --legs headline,cfg4,trained 2>"$OUT/b.err" > "$OUT/bench.json"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6ag/bench.json").read().strip().splitlines()[-1]); e=d["extra"]
print(d["value"], e.get("leg_errors"))
for k,v in e["cfg4_2p4M_sh_fwd_bwd"].items(): print(k, v["forward_ms"], v["raster_fwd_ms"], v["backward_ms"], v["fwd_bwd_iters_per_s"], v["render"])
ts=e["trained_state"]["sh_degree_2"]
print({m:(ts[m].get("render_fps"), ts[m].get("fwd_bwd_iters_per_s"), ts[m].get("forward_stage_ms",{}).get("raster")) for m in ts if isinstance(ts[m],dict) and "render_fps" in ts[m]})
PY
