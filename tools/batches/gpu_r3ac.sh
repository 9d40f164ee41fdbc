#!/bin/bash
# kernel trace + PMC of the fused loss kernel alone
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r3ac; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python "$R/tools/loss_profile.py" > "$OUT/loss.txt" 2> "$OUT/stats.err"
i=0
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_IFETCH SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$i" -o p -- python "$R/tools/loss_profile.py" > /dev/null 2> "$OUT/pmc_$i.err"
done
python "$R/tools/pmc_summary.py" $(find "$OUT" -name '*counter_collection.csv') > "$OUT/pmc_summary.csv"
cp $(find "$OUT/stats" -name '*kernel_stats.csv' | head -1) "$OUT/kernel_stats.csv"
cat "$OUT/loss.txt"; head -8 "$OUT/kernel_stats.csv" | cut -c1-200; cat "$OUT/pmc_summary.csv"; tail -2 "$OUT"/pmc_5.err "$OUT"/pmc_6.err
