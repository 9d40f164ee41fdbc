#!/bin/bash
# round 6, call z: one-pass segmented compositing -- long-list parity tests, then the trained-state leg with the one-pass and the
# NOTE: the experiment was dropped: GS_SEG_TWO_PASS no longer exists in the library; kept as the record of what was run
# two-pass scheme (GS_SEG_TWO_PASS=1)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6z; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_frame.py tests/test_gpu_train.py -q -m gpu -k "long or pile or segment or beyond_the_sort_window" > "$OUT/tests.txt" 2>&1; echo "tests rc=$?"; tail -15 "$OUT/tests.txt"
for tp in 0 1; do
  GS_SEG_TWO_PASS=$tp timeout 900 python bench.py --legs headline,trained > "$OUT/bench_tp$tp.json" 2> "$OUT/err$tp.txt"
  python - "$OUT/bench_tp$tp.json" $tp <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ts=d["extra"]["trained_state"]
for k,v in ts.items():
    if isinstance(v,dict) and "auto" in v:
        print("two_pass", sys.argv[2], k, {m:(v[m].get("flagged_long_lists"), v[m].get("render_fps"), v[m].get("fwd_bwd_iters_per_s"), v[m].get("forward_stage_ms",{}).get("raster")) for m in v if isinstance(v[m],dict) and "render_fps" in v[m]})
PY
done
