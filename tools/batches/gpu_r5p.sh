#!/bin/bash
# round 5, call p: profile sets of the FINAL tree (after the work items / 512-512 changes): cfg4 degree 2 and 3 forward + backward,
# cfg5 forward + backward, cfg5 forward; the long-list pile with the final defaults
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5p; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 600 bash tools/profile_round.sh r5p_deg2 cfg4 fwdbwd --sh-degree 2 > "$OUT/profile_deg2.txt" 2>&1; echo "deg2 rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 bash tools/profile_round.sh r5p_deg3 cfg4 fwdbwd --sh-degree 3 > "$OUT/profile_deg3.txt" 2>&1; echo "deg3 rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 bash tools/profile_round.sh r5p_cfg5t cfg5 fwdbwd > "$OUT/profile_cfg5t.txt" 2>&1; echo "cfg5t rc=$?" | tee -a "$OUT/steps.txt"
timeout 600 bash tools/profile_round.sh r5p_cfg5 cfg5 fwd > "$OUT/profile_cfg5.txt" 2>&1; echo "cfg5 rc=$?" | tee -a "$OUT/steps.txt"
timeout 300 python tools/long_list.py 100000 0 > "$OUT/pile_rgb.txt" 2> "$OUT/pile.err"
timeout 300 python tools/long_list.py 100000 2 > "$OUT/pile_sh2.txt" 2>> "$OUT/pile.err"
timeout 300 python tools/long_list.py 100000 3 > "$OUT/pile_sh3.txt" 2>> "$OUT/pile.err"
cat "$OUT/steps.txt"; grep -v "'serial_long_lists': True" "$OUT"/pile_*.txt | cut -c1-360
