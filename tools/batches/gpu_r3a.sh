#!/bin/bash
# round 3, call A: validate the tree + ordered-dispatch A/B + forced-collective bench
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r3a; mkdir -p "$OUT"; cd "$R"
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest.log"
for V in "GS_FWD_ORDER=0" "GS_FWD_ORDER=1" "GS_FWD_TICKET=4" "GS_FWD_TICKET=3" "GS_FWD_TICKET=8" "GS_FWD_ORDER=0" "GS_FWD_ORDER=1"; do
  echo "== $V"; env $V timeout 300 python tools/stage_profile.py cfg5_fwd cfg2_fwd cfg5 2>&1 | grep -v amdgpu.ids
done > "$OUT/order_ab.txt" 2>&1
cat "$OUT/order_ab.txt"
timeout 900 python bench.py --gpus 1 --force-collective > "$OUT/bench_force_collective.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; tail -c 1500 "$OUT/bench.err"
python bench.py --gpus 2 > "$OUT/bench_gpus2.out" 2>&1; echo "bench --gpus 2 rc=$?"; cat "$OUT/bench_gpus2.out"
cat "$OUT/bench_force_collective.json"
