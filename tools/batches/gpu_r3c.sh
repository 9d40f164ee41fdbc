#!/bin/bash
# round 3, call C: fused project + count A/B, written-row mask, auto variant sweep, Adam stream rates
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r3c; mkdir -p "$OUT"; cd "$R"
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
for V in base nofuse; do
  echo "== $V"
  if [ $V = base ]; then unset GS_AMD_LIB; else export GS_AMD_LIB=$R/build/variants/$V/libgs_amd.so; fi
  for rep in 1 2; do timeout 600 python tools/stage_profile.py cfg5_fwd cfg2_fwd cfg5 cfg4 cfg4_deg3 2>&1 | grep -v amdgpu.ids; done
done > "$OUT/fuse_ab.txt" 2>&1; unset GS_AMD_LIB
cat "$OUT/fuse_ab.txt"
timeout 900 python tools/sweep_n.py 10000 100000 200000 376467 506627 1000000 2400000 2>/dev/null > "$OUT/sweep_auto.jsonl"; cat "$OUT/sweep_auto.jsonl"
echo "== adam base"; timeout 300 python tools/adam_bw.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/adam_base.jsonl"
echo "== adam nt"; GS_AMD_LIB=$R/build/variants/adamnt/libgs_amd.so timeout 300 python tools/adam_bw.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/adam_nt.jsonl"
