#!/bin/bash
# round 4, call g: (1) where does the projection backward's time go (timing-only variants: no mask / no row loads),
# (2) full-size gradient parity report with the pure relative error percentiles, (3) PSNR noise: 5 seeds x 2 trees of the
# 7,001-iteration fit, (4) kernel trace + PMC of cfg5 forward + backward on the current tree
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4g; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 400 python tools/ab_variants.py run cfg5 > "$OUT/ab.txt" 2> "$OUT/ab.err"
echo "ab rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -s -k "full_size_backward" > "$OUT/pytest_full.log" 2>&1
echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
timeout 900 python tools/psnr_noise.py --seeds 1,2,3,4,5 --trees .,build/r03 > "$OUT/psnr_noise.jsonl" 2> "$OUT/psnr_noise.err"
echo "psnr rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 tools/profile_round.sh r4g/prof_cfg5t cfg5 fwdbwd > "$OUT/prof_cfg5t.log" 2>&1
echo "prof rc=$?" | tee -a "$OUT/steps.txt"
grep -E "bwd" "$OUT/ab.txt" | cut -c1-260; grep -E "gradient parity|passed|failed|Assertion" "$OUT/pytest_full.log" | cut -c1-1500; cat "$OUT/psnr_noise.jsonl"
grep -E "^\"kernel|frame_project_backward|raster_backward_pixel" "$OUT/prof_cfg5t/pmc_summary.csv"
