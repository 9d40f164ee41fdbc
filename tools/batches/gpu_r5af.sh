#!/bin/bash
# round 5, call af: timelines of single steps at the END of the densifying runs (tools/soak.py under a kernel trace)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5af; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for c in "3000 0" "2000 2"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/tr_$tag" -o t -- python "$R/tools/soak.py" 0 $c > "$OUT/soak_$tag.json" 2> "$OUT/soak_$tag.err"
  f=$(find "$OUT/tr_$tag" -name '*kernel_trace.csv' | head -1)
  python "$R/tools/step_timeline.py" "$f" raster_forward_kernel 3 > "$OUT/timeline_$tag.txt" 2>&1
  python "$R/tools/step_timeline.py" "$f" raster_forward_kernel 40 > "$OUT/timeline_${tag}_b.txt" 2>&1
  rm -rf "$OUT/tr_$tag"
done
cat "$OUT"/timeline_3000_0.txt | cut -c1-180
