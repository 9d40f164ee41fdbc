#!/bin/bash
# round 5, call ab: timelines of single training steps (tools/step_timeline.py) -- rgb at 376 k and 2.4 M Gaussians, SH degree 2 at 2.4 M
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5ab; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for c in "cfg2" "cfg5" "cfg4 --sh-degree 2"; do
  tag=$(echo $c | tr -d ' -'); 
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/tr_$tag" -o t -- python "$R/tools/prof_target.py" $c --train --frames 30 > "$OUT/$tag.json" 2> "$OUT/$tag.err"
  f=$(find "$OUT/tr_$tag" -name '*kernel_trace.csv' | head -1)
  python "$R/tools/step_timeline.py" "$f" > "$OUT/timeline_$tag.txt" 2>&1
  rm -rf "$OUT/tr_$tag"
done
cat "$OUT"/timeline_*.txt | cut -c1-200
