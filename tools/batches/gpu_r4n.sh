#!/bin/bash
# round 4, call n: is the training step slower on small scenes than in round 3, or is it the timing protocol?  Round 3's
# sweep_n.py (its protocol: random target, iterations 5 .. 25) on the current tree and on round 3's tree (build/r03), next
# to the current protocol on both -- same box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4n; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
sed "s#'/root/repo/3d-gaussian-splatting_amd'#'/root/repo/build/r03/3d-gaussian-splatting_amd'#; s#\['/root/repo', #['/root/repo/build/r03', #" build/r03/tools/sweep_n.py > "$OUT/sweep_old_protocol_old_tree.py"
for rep in 1 2; do
timeout 300 python build/r03/tools/sweep_n.py 10000 100000 376467 >> "$OUT/old_protocol_new_tree.jsonl" 2>> "$OUT/err.txt"
timeout 300 python "$OUT/sweep_old_protocol_old_tree.py" 10000 100000 376467 >> "$OUT/old_protocol_old_tree.jsonl" 2>> "$OUT/err.txt"
done
for f in old_protocol_new_tree old_protocol_old_tree; do echo $f; cut -c1-260 "$OUT/$f.jsonl"; done; tail -3 "$OUT/err.txt"
