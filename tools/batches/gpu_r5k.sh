#!/bin/bash
# round 5, call k: the SH backward on the matrix pipe takes work items of <= 8 buckets (heavy tiles first) instead of one
# workgroup per tile: GPU suite; same-box A/B against one workgroup per tile (chunk0) and 16-bucket items (chunk16) --
# stage times at cfg4 / cfg4_deg3, the densifying SH soak (degree 2 and 3, 2,000 iterations)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5k; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -q -rf --maxfail=30 -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
for rep in 0 1; do
  for V in base chunk0 chunk16; do
    L=""; [ "$V" != base ] && L="$R/build/variants/$V/libgs_amd.so"
    GS_AMD_LIB=$L timeout 300 python tools/stage_profile.py cfg4 cfg4_deg3 2>> "$OUT/ab.err" | sed "s/^/[$V #$rep] /" >> "$OUT/ab_sh.txt"
  done
done
for V in base chunk0 chunk16; do
  L=""; [ "$V" != base ] && L="$R/build/variants/$V/libgs_amd.so"
  for D in 2 3; do
    GS_AMD_LIB=$L timeout 300 python tools/soak.py 0 2000 $D > "$OUT/soak_${V}_deg$D.json" 2>> "$OUT/soak.err"
  done
done
echo "ab rc=$?" | tee -a "$OUT/steps.txt"
tail -n 3 "$OUT/pytest.log" | cut -c1-200; sed "s/fwd {.*'total': \([0-9.]*\)} bwd/fwd_total \1 bwd/" "$OUT/ab_sh.txt" | cut -c1-200
for f in "$OUT"/soak_*.json; do echo "$f"; python -c "import json,sys; d=json.load(open('$f'))['train']; print(d['iters_per_s'], d['iters_per_s_median_block'], d['iters_per_s_min_block'])"; done
