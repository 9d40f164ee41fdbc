#!/bin/bash
# round 6, call n: timing-only switches of the compacting project kernel's phases
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6n; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
for d in 0 32 64 96; do
  GS_OCC_DIAG=$d timeout 300 python tools/cull_survivors.py > "$OUT/surv_diag$d.json" 2> "$OUT/err$d.txt"
  python - "$d" <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r6n/surv_diag{sys.argv[1]}.json"))
print("diag",sys.argv[1],{k:(v["stage_ms"]["project"],v["with_tiles"],v["no_tile_but_visible"],v["pairs"]) for k,v in d.items()})
PY
done
