#!/bin/bash
# Collects one profiles/ set on the GPU box:  tools/profile_round.sh <tag> [bench flags...]
#   1. rocprofv3 --kernel-trace --stats of bench.py (kernel_stats.csv + the JSON line of the same run)
#   2. three separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ busy/instruction counters)
# Outputs land in gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
FLAGS=${*:---steps 200 --warmup 20 --no-cpu-baseline --no-extra --streams 1}
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python "$R/bench.py" $FLAGS > "$OUT/bench.json" 2> "$OUT/stats.err"
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  N=$(echo $C | cut -d' ' -f1)
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$N" -o p -- python "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-extra --streams 1 > /dev/null 2> "$OUT/pmc_$N.err"
done
python "$R/tools/pmc_summary.py" $(find "$OUT" -name '*counter_collection.csv') > "$OUT/pmc_summary.csv"
cp $(find "$OUT/stats" -name '*kernel_stats.csv') "$OUT/kernel_stats.csv"
tail -c 1500 "$OUT/bench.json"; echo; cat "$OUT/pmc_summary.csv"
