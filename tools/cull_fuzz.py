#!/usr/bin/env python3
"""Exactness of the occlusion cull along a random camera walk (round 6): every frame of a renderer with the cull on (its own
policy: own cuts at rest, dilated cuts within 8 px, nothing beyond; adaptive back-off) equals, bit for bit, the frame of a
renderer with the cull off.

    python tools/cull_fuzz.py [frames, default 600] [config, default cfg5] [seed]

Steps are drawn log-uniformly between 0.001 and 3 degrees (yaw) with rests and translations mixed in."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-gaussian-splatting_amd")]
import numpy as np
import torch
from gs_frame import FrameRenderer
from gs_scene import CONFIGS, make_camera, make_scene

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 600
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg5"
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
n, W, H, _ = CONFIGS[cfg]
scene = make_scene(n, W, H, seed=2023)
params = [torch.from_numpy(a).to(dev) for a in (scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb)]
r = FrameRenderer(dev, max_pairs=9_000_000, auto_grow=False)
off = FrameRenderer(dev, max_pairs=9_000_000, auto_grow=False, occlusion_cull=False)
rng = np.random.default_rng(seed)
yaw, tx = 0.0, 0.0
stats = {"frames": 0, "culled": 0, "dilated": 0, "near": 0, "fell_back": 0, "mismatch": 0, "overflow": 0}
for k in range(frames):
    u = rng.random()
    if u < 0.25:
        pass  # rest
    elif u < 0.9:
        yaw += float(np.exp(rng.uniform(np.log(0.001), np.log(3.0)))) * (1 if rng.random() < 0.5 else -1)
        yaw = float(np.clip(yaw, -25.0, 25.0))
    else:
        tx += float(rng.normal(0.0, 0.003))
    cam = make_camera(W, H, yaw_deg=yaw)
    cam.tran = np.asarray(cam.tran, np.float32) + np.array([tx, 0.0, 0.0], np.float32)
    img, _ = r.forward(*params, cam)
    st = r.stats()
    ref, _ = off.forward(*params, cam)
    fl = int(r._frame.flags)
    stats["frames"] += 1
    stats["culled"] += bool(fl & 256)
    stats["dilated"] += bool(fl & 512)
    stats["near"] += bool(fl & 1024)
    stats["fell_back"] += bool(st.cull_fallback)
    stats["overflow"] += bool(st.overflow or off.stats().overflow)
    if not torch.equal(img, ref):
        stats["mismatch"] += 1
        print("MISMATCH at frame", k, "yaw", yaw, "flags", fl, st, file=sys.stderr)
print(json.dumps({"config": cfg, "seed": seed, **stats}))
sys.exit(1 if stats["mismatch"] else 0)
