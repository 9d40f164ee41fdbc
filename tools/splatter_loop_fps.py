"""The reference's training loop shape (torch.optim.Adam over gaussian_3ds' parameters, autograd L1 + SSIM-free loss
on ``Splatter.forward``) on this package's ``splatter.Splatter``, next to ``gs_train.Trainer`` (fused loss + Adam) on
the same capture.  Prints one JSON object.   python tools/splatter_loop_fps.py [n_points] [iters]"""
import json
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(HERE, "..", "3d-gaussian-splatting_amd")]
import torch  # noqa: E402

import make_synthetic_colmap as msc  # noqa: E402
from gs_train import TrainOptions, Trainer  # noqa: E402
from splatter import Splatter  # noqa: E402

n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 376_467
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
root = tempfile.mkdtemp()
msc.build(root, n=n_points, width=1920, height=1080, views=9, points=n_points, downsample=(1,), seed=2023)
sp = Splatter(os.path.join(root, "sparse", "0"), os.path.join(root, "images_1"), render_downsample=1,
              opa_init_value=0.3, scale_init_value=1, tile_culling_prob_thresh=0.05, max_pairs=1 << 21)
g = sp.gaussian_3ds
opt = torch.optim.Adam([{"params": g.opa, "lr": 0.03}, {"params": g.rgb, "lr": 0.03}, {"params": g.pos, "lr": 0.003},
                        {"params": g.scale, "lr": 0.003}, {"params": g.quat, "lr": 0.003}], betas=(0.9, 0.99))
rng = np.random.default_rng(0)


def step():
    opt.zero_grad()
    loss = (sp(int(rng.integers(1, 8))) - sp.ground_truth).abs().mean()
    loss.backward()
    opt.step()
    return loss


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    step()
torch.cuda.synchronize()
loop = iters / (time.perf_counter() - t0)
# the fused trainer on the same Gaussians / cameras / targets
import gs_colmap  # noqa: E402

scene = gs_colmap.load_scene(root, 1, sp.device)
tr = Trainer([t.detach().clone() for t in (g.pos, g.quat, g.scale, g.opa, g.rgb)], scene.cameras, scene.targets,
             TrainOptions(ssim_weight=0.0), max_pairs=1 << 21)
for i in range(20):
    tr.train_step(i, 1 + i % 7)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(iters):
    tr.train_step(20 + i, int(rng.integers(1, 8)))
torch.cuda.synchronize()
fused = iters / (time.perf_counter() - t0)
print(json.dumps({"gaussians": sp.n_gaussians, "tile_pairs": sp.n_tile_gaussians, "resolution": "1920x1080",
                  "reference_style_loop_iters_per_s": round(loop, 1),
                  "reference_style_loop": "splatter.Splatter forward/backward through autograd + torch L1 loss + "
                                          "torch.optim.Adam (5 groups)",
                  "fused_trainer_iters_per_s": round(fused, 1),
                  "fused_trainer": "gs_train.Trainer: same frame kernels + gs_loss_l1_ssim (L1 only) + gs_adam_step"}))
