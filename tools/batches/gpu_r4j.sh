#!/bin/bash
# round 4, call j: tile dispatch order with 2 x 2 blocks kept on one XCD (build/variants/blocks, blocks32) against the
# in-tree longest-first order: stage times (cfg5, cfg2 inference frames) and FETCH_SIZE of the compositing kernel
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4j; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 600 python tools/ab_variants.py run cfg5_fwd cfg2_fwd > "$OUT/ab.txt" 2> "$OUT/ab.err"
echo "ab rc=$?" | tee -a "$OUT/steps.txt"
cd /tmp && export TMPDIR=/tmp
for V in base blocks blocks32; do
  L=""; [ "$V" != base ] && L="$R/build/variants/$V/libgs_amd.so"
  GS_AMD_LIB=$L timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_$V" -o p -- python "$R/tools/prof_target.py" cfg5 --frames 12 --no-stage-times > /dev/null 2> "$OUT/pmc_$V.err"
  python "$R/tools/pmc_summary.py" $(find "$OUT/pmc_$V" -name '*counter_collection.csv') | grep -E "kernel|raster_forward" > "$OUT/fetch_$V.csv"
  GS_AMD_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tr_$V" -o s -- python "$R/tools/prof_target.py" cfg5 --frames 100 > "$OUT/tr_$V.json" 2> "$OUT/tr_$V.err"
  grep -E "raster_forward" $(find "$OUT/tr_$V" -name '*kernel_stats.csv' | head -1) | cut -d, -f2-4 > "$OUT/raster_$V.txt"
done
cd "$R"
cut -c1-230 "$OUT/ab.txt"; for V in base blocks blocks32; do echo $V; cat "$OUT/fetch_$V.csv" | cut -c1-200; cat "$OUT/raster_$V.txt"; done
