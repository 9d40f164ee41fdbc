#!/bin/bash
# round 5, call n: where the serial walk of a flagged frame's long lists ends (GS_LONG_MIN) and how long a segment is
# (GS_SEG_LEN): base 2048 / 1024 against 512 / 512, 512 / 256, 1024 / 512 -- the densifying soaks (rgb 3,000 iterations, SH
# degree 2 2,000), the 100,000-Gaussian pile of tools/long_list.py (rgb and SH degree 2); plus the long-list tests on the
# in-tree build (no SH hand-over any more: the matrix-pipe kernel's work items take every bucket)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5n; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_frame.py -m gpu -q -rf -p no:cacheprovider -k "long or pile or segments or hand_over or dense or deep or screen_filling" > "$OUT/pytest_long.log" 2>&1; echo "pytest_long rc=$?" | tee -a "$OUT/steps.txt"
for V in base lm512s512 lm512s256 lm1024s512; do
  L=""; [ "$V" != base ] && L="$R/build/variants/$V/libgs_amd.so"
  GS_AMD_LIB=$L timeout 300 python tools/soak.py 0 3000 0 > "$OUT/soak_${V}_rgb.json" 2>> "$OUT/soak.err"
  GS_AMD_LIB=$L timeout 300 python tools/soak.py 0 2000 2 > "$OUT/soak_${V}_sh2.json" 2>> "$OUT/soak.err"
  GS_AMD_LIB=$L timeout 300 python tools/long_list.py 100000 0 > "$OUT/pile_${V}_rgb.txt" 2>> "$OUT/pile.err"
  GS_AMD_LIB=$L timeout 300 python tools/long_list.py 100000 2 > "$OUT/pile_${V}_sh2.txt" 2>> "$OUT/pile.err"
done
tail -n 3 "$OUT/pytest_long.log" | cut -c1-200
for f in "$OUT"/soak_*.json; do echo "$f"; python -c "import json,sys; d=json.load(open('$f'))['train']; print(d['iters_per_s'], d['iters_per_s_median_block'], d['iters_per_s_min_block'], d['iters_per_s_blocks'][-3:])"; done
for f in "$OUT"/pile_*.txt; do echo "$f"; cut -c1-400 "$f"; done
