#!/bin/bash
# round 5, call u: the image-loss kernel with its filter stages on packed fp32 pairs ((x, y), (xx, yy), (D_mu, D_xx)) against the
# previous kernel (build/variants/rows_pk0 still holds it): loss tests, then loss_ms of bench.py's training legs alternating
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r5u; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 1200 python -m pytest tests -m gpu -q -rf --maxfail=30 -p no:cacheprovider -k "loss or train" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
for i in 1 2; do
  timeout 600 python bench.py --legs headline,train > "$OUT/bench_train_new_$i.json" 2> "$OUT/bench_train_new_$i.err"; echo "bench new $i rc=$?" | tee -a "$OUT/steps.txt"
  GS_AMD_LIB=$R/build/variants/rows_pk0/libgs_amd.so timeout 600 python bench.py --legs headline,train > "$OUT/bench_train_old_$i.json" 2> "$OUT/bench_train_old_$i.err"; echo "bench old $i rc=$?" | tee -a "$OUT/steps.txt"
done
tail -n 8 "$OUT/pytest.log" | cut -c1-300
