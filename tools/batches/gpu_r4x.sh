#!/bin/bash
# round 4, call x: Trainer.tune_slices on two gloo ranks (one GPU) and through RCCL on one rank; train tests
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4x; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "train or trajectory" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
timeout 300 python - > "$OUT/tune.txt" 2>&1 <<'P'
import os, sys
sys.path[:0] = ['/root/repo', '/root/repo/3d-gaussian-splatting_amd', '/root/repo/tools']
import torch, torch.distributed as dist
from gs_frame import FrameRenderer
from gs_scene import CONFIGS, make_camera, make_scene
from gs_train import TrainOptions, Trainer
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29871')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
n, W, H, _ = CONFIGS['cfg5']; scene = make_scene(n, W, H, seed=2023); cam = make_camera(W, H)
params = [torch.from_numpy(x).to(dev) for x in (scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb)]
r = FrameRenderer(dev, max_pairs=1 << 20, auto_grow=True); img = r.forward(*params, cam)[0]; pairs = r.stats().pairs; del r
tr = Trainer(params, [cam], [img.clone()], TrainOptions(lr=0.0), max_pairs=int(pairs * 1.25) + 4096)
tr.flat.force_collective = True
for i in range(10): tr.train_step(i, 0, next_camera_id=0)
print('tuned slices on one rank (nothing travels):', tr.tune_slices(10, 0, candidates=(1, 2, 4), iters=8, next_camera_id=0))
dist.destroy_process_group()
P
echo "tune rc=$?" | tee -a "$OUT/steps.txt"
grep -E "passed|failed" "$OUT/pytest.log"; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu" "$OUT/tune.txt" | tail -3
