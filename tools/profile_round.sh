#!/bin/bash
# Collects one profiles/ set on the GPU box:  tools/profile_round.sh <tag> <config> [fwd|fwdbwd|train] [extra prof_target flags]
#   (mode `train`: whole steps of gs_train.Trainer -- forward, loss, backward, optimizer -- instead of bare frames)
#   1. rocprofv3 --kernel-trace --stats of tools/prof_target.py <config> (per-kernel averages of THAT workload)
#      + the JSON line of the same run (N, V, M, hipEvent stage times)
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ instruction / busy counters, SQ active / wait counters);
#      never combined with a trace domain
# Outputs land in gpurun_out/<tag>/ ; copy what should be judged into profiles/ (tools/README.md).
set -u
TAG=$1; CFG=$2; MODE=${3:-fwd}; shift; shift; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BW=""; [ "$MODE" = "fwdbwd" ] && BW="--backward"; [ "$MODE" = "train" ] && BW="--train"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python "$R/tools/prof_target.py" $CFG $BW --frames 100 "$@" > "$OUT/target.json" 2> "$OUT/stats.err"
i=0
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$i" -o p -- python "$R/tools/prof_target.py" $CFG $BW --frames 12 --no-stage-times "$@" > /dev/null 2> "$OUT/pmc_$i.err"
done
python "$R/tools/pmc_summary.py" $(find "$OUT" -name '*counter_collection.csv') > "$OUT/pmc_summary.csv"
cp $(find "$OUT/stats" -name '*kernel_stats.csv' | head -1) "$OUT/kernel_stats.csv"
cat "$OUT/target.json"; echo; head -40 "$OUT/kernel_stats.csv" | cut -c1-220; cat "$OUT/pmc_summary.csv"
