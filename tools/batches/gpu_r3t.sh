#!/bin/bash
# the exact launch line the driver uses for N > 1, at N = 1: one JSON line on stdout, banners on stderr
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r3t; mkdir -p "$OUT"; cd "$R"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 50 --warmup 10 --legs headline,train,multi_gpu > "$OUT/torchrun_n1.stdout" 2> "$OUT/torchrun_n1.stderr"; echo "rc=$? lines on stdout: $(wc -l < $OUT/torchrun_n1.stdout)"
python -c "
import json,sys
j=json.loads(open('$OUT/torchrun_n1.stdout').read())
print(j['value'], j['n_gpus'], list(j['multi_gpu']['modes'].keys()), j['multi_gpu']['ranks_seen'], j.get('leg_errors'))
"
grep -c "RCCL version" "$OUT/torchrun_n1.stderr"
