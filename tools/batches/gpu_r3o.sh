#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r3o; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python "$R/tools/loss_profile.py" > "$OUT/loss.txt" 2> /dev/null
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$i" -o p -- python "$R/tools/loss_profile.py" > /dev/null 2> "$OUT/pmc_$i.err"
done
python "$R/tools/pmc_summary.py" $(find "$OUT" -name '*counter_collection.csv') > "$OUT/pmc_summary.csv"
cat "$OUT/loss.txt"; cp $(find "$OUT/stats" -name '*kernel_stats.csv' | head -1) "$OUT/kernel_stats.csv"; grep -i "ssim\|loss" "$OUT/kernel_stats.csv" | cut -c1-200; grep -i "kernel\|ssim\|loss" "$OUT/pmc_summary.csv"
