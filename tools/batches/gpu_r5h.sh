#!/bin/bash
# round 5, call h: the round's kernels are settled -- (1) the full-size gradient parity tables and the seam / row-layout tests with
# -s (the asserted relative-error bounds are set from THIS run), (2) profile sets of the final tree: cfg5 forward (headline
# kernel), cfg5 forward + backward (rgb, the renderer's own kernel choice), cfg4 degree 2 and 3, cfg2 forward, (3) default bench
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5h; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_frame.py -m gpu -q -rf -p no:cacheprovider -s -k "full_size_backward_matches or hand_over or row_layout or repeatable" > "$OUT/pytest_s.log" 2>&1; echo "pytest_s rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -rf -p no:cacheprovider > "$OUT/pytest_train.log" 2>&1; echo "pytest_train rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 bash tools/profile_round.sh r5h_cfg5 cfg5 fwd > "$OUT/profile_cfg5.txt" 2>&1; echo "cfg5 rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 bash tools/profile_round.sh r5h_cfg5t cfg5 fwdbwd > "$OUT/profile_cfg5t.txt" 2>&1; echo "cfg5t rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 bash tools/profile_round.sh r5h_deg2 cfg4 fwdbwd --sh-degree 2 > "$OUT/profile_deg2.txt" 2>&1; echo "deg2 rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 bash tools/profile_round.sh r5h_deg3 cfg4 fwdbwd --sh-degree 3 > "$OUT/profile_deg3.txt" 2>&1; echo "deg3 rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 bash tools/profile_round.sh r5h_cfg2 cfg2 fwd > "$OUT/profile_cfg2.txt" 2>&1; echo "cfg2 rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?" | tee -a "$OUT/steps.txt"
cat "$OUT/steps.txt"; tail -n 5 "$OUT/pytest_s.log" | cut -c1-300; tail -n 5 "$OUT/pytest_train.log" | cut -c1-300
