#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r3f; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
for T in cur r01h; do
  ROOT=$R; [ $T = r01h ] && ROOT=$R/build/r01h
  for N in 10000 100000; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tr_${T}_$N" -o s -- python "$R/tools/small_trace.py" "$ROOT" $N 300 > /dev/null 2>&1
    echo "== $T $N"; python - "$OUT/tr_${T}_$N" <<'P'
import csv,sys,glob
tot=0
for f in glob.glob(sys.argv[1]+'/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if int(r['Calls'])>=290:
            print('  ', r['Name'].replace('void (anonymous namespace)::','')[:60], r['Calls'], round(float(r['AverageNs'])/1e3,2)); tot+=float(r['AverageNs'])/1e3
print('   sum of per-frame kernels (us):', round(tot,2))
P
  done
done 2>&1 | tee "$OUT/small_scene_kernels.txt"
cd "$R"; timeout 900 tools/profile_round.sh r3f/prof_cfg4 cfg4 fwdbwd > "$OUT/prof_cfg4.log" 2>&1; tail -22 "$OUT/prof_cfg4.log" | cut -c1-300
