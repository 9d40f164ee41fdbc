"""Soak test: a long render loop and a long training run with densification on the reference's schedule;
checks that results stay finite, memory does not creep and nothing stalls.  Prints one JSON object.

    python tools/soak.py [render frames] [training iterations] [SH degree: 0 (rgb logits, default) | 2 | 3]"""
import json
import sys
import time

import numpy as np

sys.path[:0] = ['/root/repo', '/root/repo/3d-gaussian-splatting_amd']
import torch  # noqa: E402

from gs_frame import FrameRenderer  # noqa: E402
from gs_scene import make_camera, make_scene  # noqa: E402
from gs_train import TrainOptions, Trainer  # noqa: E402

dev = torch.device('cuda:0')
W, H = 1920, 1080
out = {}
# ---- render: 3 frames in flight, changing cameras
sh_degree = int(sys.argv[3]) if len(sys.argv) > 3 else 0
scene = make_scene(376_467, W, H, seed=2023, use_sh=sh_degree > 0, sh_degree=sh_degree if sh_degree else 2)
out["sh_degree"] = sh_degree
params = [torch.from_numpy(a).to(dev) for a in (scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb)]
cams = [make_camera(W, H, yaw_deg=float(y)) for y in np.linspace(-10, 10, 9)]
rs = [FrameRenderer(dev, max_pairs=1 << 21, auto_grow="async") for _ in range(3)]
streams = [torch.cuda.Stream(device=dev) for _ in rs]
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60_000
torch.cuda.synchronize()
m0 = torch.cuda.memory_allocated()
t0 = time.perf_counter()
last = None
for k in range(n_frames):
    with torch.cuda.stream(streams[k % 3]):
        last = rs[k % 3].forward(*params, cams[k % len(cams)])[0]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
if n_frames:
    out["render"] = {"frames": n_frames, "fps": round(n_frames / dt, 1), "finite": bool(torch.isfinite(last).all()),
                     "memory_growth_MiB": round((torch.cuda.memory_allocated() - m0) / 2**20, 1)}
del rs, last
# ---- training with densification every 100 iterations (train.py schedule), 16 cameras
n_iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
gt = params
targets = []
r = FrameRenderer(dev, max_pairs=1 << 21)
for c in cams:
    targets.append(r.forward(*gt, c)[0].clone())
del r
g = torch.Generator(device=dev).manual_seed(3)
start = [t.clone() for t in gt]
start[4] += 0.5 * torch.randn(start[4].shape, device=dev, generator=g)
opt = TrainOptions(n_iters=n_iters, use_clone=1, delete_thresh=1.5, grad_thresh=2e-4)
tr = Trainer(start, cams, targets, opt, max_pairs=1 << 21, densify=True, generator=g)
rng = np.random.default_rng(0)
sizes, pairs, rates = [tr.n_gaussians], [], []
torch.cuda.synchronize()
t0 = t_win = time.perf_counter()
for i in range(n_iters):
    v = tr.train_step(i, int(rng.integers(len(cams))))
    if i % 500 == 0:
        sizes.append(tr.n_gaussians)
        pairs.append(tr.renderer.stats().pairs)  # synchronises: once per 500 iterations
        now = time.perf_counter()
        rates.append(round(500 / (now - t_win), 1) if i else 0.0)
        t_win = now
torch.cuda.synchronize()
dt = time.perf_counter() - t0
vals = v.cpu().numpy()
out["train"] = {"iters": n_iters, "iters_per_s": round(n_iters / dt, 1), "n_gaussians_every_500": sizes,
                "tile_pairs_every_500": pairs, "iters_per_s_every_500": rates,
                "final_loss": round(float(vals[0]), 5), "finite": bool(np.isfinite(vals).all()),
                "peak_memory_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}
print(json.dumps(out))
