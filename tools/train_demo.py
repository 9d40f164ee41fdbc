"""BASELINE.json configs[2] with what this box has: a 7,001-iteration run of the reference's training loop
(train.py defaults: lr 0.003 x (10, 10, 1, 1, 1), 300 warm-up iterations, exponential decay to 1 %, L1 + 0.1 SSIM,
Adam(0.9, 0.99), one random training view per iteration, every 8th view held out) on a synthetic multi-view
problem -- there is no dataset here: the ground truth is the cfg3 scene (506,627 Gaussians) rendered from
17 yawed cameras at 1080p, and the model starts from a perturbed copy of it.  Prints one JSON object:
iterations/s over the whole loop (host work included) and PSNR / SSIM on the held-out views before and after."""
import json
import os
import sys
import time

import numpy as np

# python tools/train_demo.py [n_iters] [--seed S] [--tree PATH]
#   --seed S    : seed of the start perturbation AND of the random view order (default: 11 / 2023, the historical run)
#   --tree PATH : run another checkout of this repository (its own libgs_amd.so), e.g. build/r03 = round 3's final tree
_args = sys.argv[1:]
_seed = int(_args[_args.index("--seed") + 1]) if "--seed" in _args else None
_tree = os.path.abspath(_args[_args.index("--tree") + 1]) if "--tree" in _args else \
    os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_pos = [a for i, a in enumerate(_args) if not a.startswith("--") and (i == 0 or _args[i - 1] not in ("--seed", "--tree"))]
sys.path[:0] = [_tree, os.path.join(_tree, '3d-gaussian-splatting_amd')]
import torch  # noqa: E402

from gs_frame import FrameRenderer  # noqa: E402
from gs_scene import CONFIGS, make_camera, make_scene  # noqa: E402
from gs_train import TrainOptions, Trainer  # noqa: E402

dev = torch.device('cuda:0')
n_iters = int(_pos[0]) if _pos else 7001
n, W, H, _ = CONFIGS['cfg3']
scene = make_scene(n, W, H, seed=2023)
gt = [torch.from_numpy(a).to(dev) for a in (scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb)]
cams = [make_camera(W, H, yaw_deg=float(y)) for y in np.linspace(-16, 16, 17)]
r = FrameRenderer(dev, max_pairs=1 << 21)
targets = [r.forward(*gt, c)[0].clone() for c in cams]
del r
g = torch.Generator(device=dev).manual_seed(11 if _seed is None else _seed)
start = [t.clone() for t in gt]
start[4] += 0.5 * torch.randn(start[4].shape, device=dev, generator=g)    # colour logits
start[3] += 0.3 * torch.randn(start[3].shape, device=dev, generator=g)    # opacity logits
start[0] += 0.002 * torch.randn(start[0].shape, device=dev, generator=g)  # positions
start[2] *= 1.0 + 0.1 * torch.randn(start[2].shape, device=dev, generator=g)
tr = Trainer(start, cams, targets, TrainOptions(n_iters=n_iters), max_pairs=1 << 21)
test_split = np.arange(0, len(cams), 8)                                    # train.py:68-69
train_split = np.array(sorted(set(range(len(cams))) - set(test_split)))


def evaluate():
    m = [tr.test(int(c)) for c in test_split]
    return (float(np.mean([x["psnr"] for x in m])), float(np.mean([x["ssim"] for x in m])),
            len(m) / sum(x["render_time"] for x in m))


psnr0, ssim0, _ = evaluate()
rng = np.random.default_rng(2023 if _seed is None else 1000 + _seed)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n_iters):
    tr.train_step(i, int(rng.choice(train_split)))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
psnr1, ssim1, fps = evaluate()
print(json.dumps({"tree": os.path.relpath(_tree, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) or ".",
                  "seed": _seed, "workload": f"cfg3 scene ({n} Gaussians), 1080p, {len(train_split)} training + {len(test_split)} "
                              f"held-out synthetic views, {n_iters} iterations of train.py's step (no densification)",
                  "iters_per_s": round(n_iters / dt, 1), "wall_s": round(dt, 2),
                  "test_psnr_before_dB": round(psnr0, 2), "test_psnr_after_dB": round(psnr1, 2),
                  "test_ssim_before": round(ssim0, 4), "test_ssim_after": round(ssim1, 4),
                  "test_render_fps": round(fps, 1), "final_train_loss": round(float(tr._loss_for(H, W).values[0]), 5)}))
