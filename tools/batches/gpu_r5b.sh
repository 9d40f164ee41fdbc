#!/bin/bash
# round 5, call b: rgb backward in the row layout (raster_backward_rows_kernel, default on), instruction diet of the SH MFMA
# kernel, SH projection backward with the row mask built once: GPU suite, then same-box A/Bs by hipEvent stage times:
#   base (rows kernel, diet) | rows0 (pixel-parallel rgb kernel) | rows_nopf | rows_nopf5 | nodiet | mfma_wpe2
# and a WRITE_SIZE pass of the SH backward without spills (mfma_wpe2: is the write excess over the row bytes scratch?)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5b; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -q -rf --maxfail=30 -p no:cacheprovider -s > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
for rep in 0 1; do
  for V in base rows0 rows_nopf rows_nopf5; do
    L=""; [ "$V" != base ] && L="$R/build/variants/$V/libgs_amd.so"
    GS_AMD_LIB=$L timeout 300 python tools/stage_profile.py cfg5 cfg2 2>> "$OUT/ab.err" | sed "s/^/[$V #$rep] /" >> "$OUT/ab_rgb.txt"
  done
  for V in base nodiet mfma_wpe2; do
    L=""; [ "$V" != base ] && L="$R/build/variants/$V/libgs_amd.so"
    GS_AMD_LIB=$L timeout 300 python tools/stage_profile.py cfg4 cfg4_deg3 2>> "$OUT/ab.err" | sed "s/^/[$V #$rep] /" >> "$OUT/ab_sh.txt"
  done
done
echo "ab rc=$?" | tee -a "$OUT/steps.txt"
cd /tmp && export TMPDIR=/tmp
for V in base mfma_wpe2; do
  L=""; [ "$V" != base ] && L="$R/build/variants/$V/libgs_amd.so"
  GS_AMD_LIB=$L timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_$V" -o p -- python "$R/tools/prof_target.py" cfg4 --backward --frames 12 --no-stage-times --sh-degree 2 > /dev/null 2> "$OUT/pmc_write_$V.err"
  python "$R/tools/pmc_summary.py" $(find "$OUT/pmc_write_$V" -name '*counter_collection.csv') > "$OUT/pmc_write_$V.csv"
done
cd "$R"
cat "$OUT/steps.txt"; tail -n 8 "$OUT/pytest.log"; cat "$OUT/ab_rgb.txt" "$OUT/ab_sh.txt" | cut -c1-400; grep -h "mfma_sh\|project_backward" "$OUT"/pmc_write_*.csv | cut -c1-200
