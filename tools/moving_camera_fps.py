#!/usr/bin/env python3
"""Render FPS with a camera that MOVES every frame (a viewer's case): every frame misses the descriptor cache and the
per-camera cache of FrameRenderer, so the host builds the tile grid, the ray basis and the frame descriptor anew.

    python tools/moving_camera_fps.py [cfg2 cfg5 ...]

Prints, per config: FPS with a fixed camera, FPS with a moving camera, and the host time per frame of the moving
case (the loop without the final synchronisation)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-gaussian-splatting_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gs_frame import FrameRenderer  # noqa: E402
from gs_scene import CONFIGS, make_camera, make_scene  # noqa: E402

dev = torch.device("cuda:0")
for cfg in (sys.argv[1:] or ["cfg2", "cfg5"]):
    n, W, H, use_sh = CONFIGS[cfg]
    scene = make_scene(n, W, H, seed=2023)
    params = [torch.from_numpy(a).to(dev) for a in (scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb)]
    frames = 400
    cams = [make_camera(W, H, yaw_deg=float(y)) for y in np.linspace(-2.0, 2.0, frames)]
    r = FrameRenderer(dev, max_pairs=1 << 20, auto_grow=True)
    for c in (cams[0], cams[-1], cams[frames // 2]):
        r.forward(*params, c)
    r.max_pairs = int(r.max_pairs * 1.2)
    r.auto_grow = False
    out = {"config": cfg}
    for name, seq in (("fixed", [cams[frames // 2]] * frames), ("moving", cams)):
        for c in seq[:20]:
            r.forward(*params, c)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c in seq:
            r.forward(*params, c)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out[name] = {"fps": round(frames / (t2 - t0), 1), "host_us_per_frame": round((t1 - t0) / frames * 1e6, 1)}
        r._cam_cache.clear()
    print(json.dumps(out), flush=True)
