#!/bin/bash
# round 6, call r: which leg in front of the densifying rgb run makes its late blocks slower (r6o / r6p: 430 against 515 it/s)?
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6r; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
for L in soak train,soak cfg4,soak trained,soak fit,soak cfg2,soak pipelined,soak; do
  timeout 900 python bench.py --legs headline,$L > "$OUT/b_${L//,/_}.json" 2> "$OUT/err.txt"
  python - "$OUT/b_${L//,/_}.json" "$L" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
v=d["extra"]["soak_densifying"]["rgb"]
print(sys.argv[2], v["iters_per_s"], v["iters_per_s_blocks"][10:])
PY
done
