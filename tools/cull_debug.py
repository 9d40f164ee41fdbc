#!/usr/bin/env python3
"""Diagnostic (round 6): which frames of a static -> moving -> static camera sequence carry GS_FRAME_OCCLUSION_CULL."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-gaussian-splatting_amd")]
import torch
from gs_frame import FrameRenderer
from gs_scene import CONFIGS, make_camera, make_scene

dev = torch.device("cuda:0")
n, W, H, _ = CONFIGS["cfg5"]
scene = make_scene(n, W, H, seed=2023)
params = [torch.from_numpy(a).to(dev) for a in (scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb)]
cam = make_camera(W, H)
r = FrameRenderer(dev, max_pairs=8_000_000, auto_grow=False)
log = []
def note(tag):
    log.append((tag, bool(r._frame.flags & 256), r._cull_off_until, r._cull_settled, r._cull_full_pairs, getattr(r, "_cull_run", None)))
for i in range(6):
    r.forward(*params, cam); torch.cuda.synchronize(); note(f"static{i}")
pan = [make_camera(W, H, yaw_deg=0.01 * i) for i in range(1, 12)]
for i, c in enumerate(pan):
    r.forward(*params, c); torch.cuda.synchronize(); note(f"moving{i}")
r.forward(*params, cam); note("back")
for i in range(6):
    p = r.profile_forward(*params, cam); note(f"profile{i} sort={p['ranges']:.3f}")
for row in log:
    print(row)
