#!/bin/bash
# round 5, call w: the SH matrix-pipe backward with the per-pixel algebra of its row step on pixel pairs (GS_BWD_MFMA_PK 1, in-tree)
# against the previous kernel (build/variants/mfma_pk0): frame + MFMA tests, then stage times of cfg4 at degree 2 and 3 on one box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5w; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests/test_gpu_frame.py tests/test_gpu_train.py -m gpu -q -rf --maxfail=30 -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python tools/ab_variants.py run cfg4 cfg4_deg3 > "$OUT/ab.txt" 2> "$OUT/ab.err"; echo "ab rc=$?" | tee -a "$OUT/steps.txt"
tail -n 8 "$OUT/pytest.log" | cut -c1-300
