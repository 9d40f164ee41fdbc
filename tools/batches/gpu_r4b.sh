#!/bin/bash
# round 4, call b: (1) backward-related GPU tests on the reworked rgb gradient rows, (2) A/B of the projection
# backward's two row-fetch designs (GS_PB_DIRECT 1 = in-tree, 0 = build/variants/staged), (3) the FETCH_SIZE /
# WRITE_SIZE calibration micro-benchmark under separate --pmc passes, (4) kernel trace + PMC of cfg5 forward+backward
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4b; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "backward or train or smoke or dist or long_list or densif" > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
AB_CFGS="cfg5 cfg2" timeout 600 python tools/ab_variants.py run cfg5 cfg2 > "$OUT/ab.txt" 2> "$OUT/ab.err"
echo "ab rc=$?" | tee -a "$OUT/steps.txt"
cd /tmp && export TMPDIR=/tmp
timeout 120 "$R/tools/ubench/fetch_gather" > "$OUT/fetch_gather.jsonl" 2> "$OUT/fetch_gather.err"
echo "ubench rc=$?" | tee -a "$OUT/steps.txt"
rocprofv3 -L 2>/dev/null | grep -i -E "TCC_EA0_(RD|WR)REQ|TCC_BUBBLE|FETCH_SIZE|WRITE_SIZE|TCC_EA0_RD_UNCACHED|TCC_REQ" | head -60 > "$OUT/counters_available.txt"
i=0
for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --output-format csv -d "$OUT/ub_pmc_$i" -o p -- "$R/tools/ubench/fetch_gather" > /dev/null 2> "$OUT/ub_pmc_$i.err"
done
python "$R/tools/pmc_summary.py" $(find "$OUT" -path '*ub_pmc*' -name '*counter_collection.csv') > "$OUT/fetch_gather_pmc_summary.csv"
cd "$R"
timeout 600 tools/profile_round.sh r4b/prof_cfg5t cfg5 fwdbwd > "$OUT/prof_cfg5t.log" 2>&1
echo "prof rc=$?" | tee -a "$OUT/steps.txt"
tail -5 "$OUT/pytest.log"; grep -E "project_bwd|raster_bwd|backward" "$OUT/ab.txt" | head -40; cat "$OUT/fetch_gather.jsonl"; cat "$OUT/fetch_gather_pmc_summary.csv"
