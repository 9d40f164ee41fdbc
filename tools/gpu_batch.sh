#!/bin/bash
# One gpurun call's worth of work:  tools/gpu_batch.sh <tag> <step> [<step> ...]
# steps: pytest probe ab prof_cfg5 prof_cfg4 prof_cfg2 bench ubench smoke benchprof
# Every step runs under its own timeout and logs to gpurun_out/<tag>/<step>.log, so one failure does not cost the rest.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
for STEP in "$@"; do
  T0=$(date +%s)
  case $STEP in
    pytest)    timeout 1200 python -m pytest tests -m gpu -q -rf --maxfail=25 -p no:cacheprovider > "$OUT/pytest.log" 2>&1 ;;
    pytest_s)  timeout 1200 python -m pytest tests -m gpu -q -rf --maxfail=25 -p no:cacheprovider -s -k "full_size_backward_matches" > "$OUT/pytest_s.log" 2>&1 ;;
    probe)     timeout 900 python tools/grad_parity_probe.py small small_sh cfg2 cfg4 > "$OUT/probe.log" 2>&1 ;;
    ab)        timeout 900 python tools/ab_variants.py run ${AB_CFGS:-cfg5 cfg2 cfg5_keys} > "$OUT/ab.txt" 2> "$OUT/ab.err" ;;
    prof_cfg5) timeout 600 tools/profile_round.sh $TAG/prof_cfg5 cfg5 fwd > "$OUT/prof_cfg5.log" 2>&1 ;;
    prof_cfg4) timeout 600 tools/profile_round.sh $TAG/prof_cfg4 cfg4 fwdbwd > "$OUT/prof_cfg4.log" 2>&1 ;;
    prof_cfg5t) timeout 600 tools/profile_round.sh $TAG/prof_cfg5t cfg5 fwdbwd > "$OUT/prof_cfg5t.log" 2>&1 ;;
    prof_cfg2) timeout 600 tools/profile_round.sh $TAG/prof_cfg2 cfg2 fwd > "$OUT/prof_cfg2.log" 2>&1 ;;
    bench)     timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" ;;
    benchprof) (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/benchprof" -o s -- python "$R/bench.py" --legs headline --no-cpu-baseline > "$OUT/benchprof.json" 2> "$OUT/benchprof.err"; cp $(find "$OUT/benchprof" -name '*kernel_stats.csv' | head -1) "$OUT/benchprof_kernel_stats.csv") ;;
    diag)      (cd /tmp && export TMPDIR=/tmp && for V in $(ls "$R/build/variants"); do GS_AMD_LIB="$R/build/variants/$V/libgs_amd.so" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/diag_$V" -o s -- python "$R/tools/prof_target.py" cfg5 --frames 60 > "$OUT/diag_$V.json" 2> "$OUT/diag_$V.err"; cp $(find "$OUT/diag_$V" -name '*kernel_stats.csv' | head -1) "$OUT/diag_${V}_kernel_stats.csv"; done) ;;
    ubench)    timeout 120 tools/ubench/sort_ops > "$OUT/ubench_sort_ops.txt" 2>&1 ;;
    smoke)     timeout 300 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1 ;;
    *)         echo "unknown step $STEP" ;;
  esac
  echo "$STEP rc=$? $(( $(date +%s) - T0 ))s" | tee -a "$OUT/steps.txt"
done
tail -n 30 "$OUT/pytest.log" 2>/dev/null
cat "$OUT/ab.txt" 2>/dev/null | tail -40
tail -c 3000 "$OUT/bench.json" 2>/dev/null
