#!/usr/bin/env python3
"""N Gaussians at 1080p, forward frames only, for a kernel trace of a small scene -- works on any tree of this repo:
    python tools/small_trace.py <tree root> [n_gaussians] [frames]"""
import os
import sys

ROOT = os.path.abspath(sys.argv[1])
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-gaussian-splatting_amd")]
import torch

from gs_frame import FrameRenderer
from gs_scene import make_camera, make_scene

n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 300
dev = torch.device("cuda:0")
W, H = 1920, 1080
scene, cam = make_scene(n, W, H, seed=2023), make_camera(W, H)
params = [torch.from_numpy(a).to(dev) for a in (scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb)]
r = FrameRenderer(dev, max_pairs=1 << 20, training=False, auto_grow=True)
r.forward(*params, cam)
r.auto_grow = False
for _ in range(frames):
    r.forward(*params, cam)
torch.cuda.synchronize()
print("done", n, frames)
