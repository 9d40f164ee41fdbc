#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r3n; mkdir -p "$OUT"; cd "$R"
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
echo "== current tree"; timeout 300 python tools/sweep_n.py 10000 100000 376467 2>/dev/null | tee "$OUT/sweep_current.jsonl"
timeout 600 python bench.py --legs headline,cfg2,pipelined --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['host_us_per_frame'], j['cfg2']['render_fps'], j['cfg2']['host_us_per_frame'], j['extra'])"
