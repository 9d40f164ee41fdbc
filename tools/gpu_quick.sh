#!/bin/bash
# tools/gpu_quick.sh <tag> [pytest -k expression] -- frame tests + stage times (strip / table variants) + cfg5 kernel trace
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; K=${2:-}
OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$R"
if [ -n "$K" ]; then timeout 900 python -m pytest tests/test_gpu_frame.py -m gpu -q -x -p no:cacheprovider -k "$K" > "$OUT/pytest.log" 2>&1; else timeout 900 python -m pytest tests/test_gpu_frame.py -m gpu -q -x -p no:cacheprovider > "$OUT/pytest.log" 2>&1; fi
tail -12 "$OUT/pytest.log"
timeout 300 python tools/stage_profile.py ${STAGE_CFGS:-cfg5_fwd cfg5_fwd_tb cfg2_fwd cfg2_fwd_tb cfg5 cfg5_tb} > "$OUT/stages.txt" 2>&1; grep -v amdgpu.ids "$OUT/stages.txt"
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python "$R/tools/prof_target.py" ${PROF_CFG:-cfg5} --frames 100 > "$OUT/target.json" 2>/dev/null)
python - "$OUT" <<'P'
import csv,sys,glob
for f in glob.glob(sys.argv[1]+'/stats/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if float(r['Percentage'])>0.5: print(r['Name'][:64], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
P
