#!/usr/bin/env python3
"""A pathological pile-up: cfg2's scene plus PILE low-opacity Gaussians inside one tile (what a degenerate densification
run produces, tools/soak.py 1000 12000).  Forward stage times with the segmented compositing / big-list sort of dense
frames and with the serial walk (GS_FRAME_SERIAL_LONG_LISTS):  python tools/long_list.py [pile=100000] [SH degree 0 | 2 | 3]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-gaussian-splatting_amd")]
import numpy as np
import torch

from gs_frame import FrameRenderer
from gs_scene import CONFIGS, make_camera, make_scene

dev = torch.device("cuda:0")
pile = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
n, W, H, _ = CONFIGS["cfg2"]
sh = int(sys.argv[2]) if len(sys.argv) > 2 else 0
scene = make_scene(n + pile, W, H, seed=2023, use_sh=sh > 0, sh_degree=sh if sh else 2)
cam = make_camera(W, H)
rng = np.random.default_rng(1)
idx = np.arange(n, n + pile)
centre = scene.pos[:n].mean(axis=0)
scene.pos[idx] = centre + rng.normal(scale=0.002, size=(pile, 3)).astype(np.float32)  # a few pixels wide
scene.scale[idx] = np.float32(0.002)
scene.opa[idx] = -7.0
params = [torch.from_numpy(a).to(dev) for a in (scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb)]
for training in ((True,) if sh else (False, True)):
    for serial in ((False,) if sh else (False, True)):
        r = FrameRenderer(dev, max_pairs=1 << 20, training=training, auto_grow=True, serial_long_lists=serial,
                          long_lists=True)
        r.forward(*params, cam)
        st = r.stats()
        r.max_pairs = max(int(st.pairs * 1.1) + 4096, 1100 * 8160)
        r.auto_grow = False
        img, _ = r.forward(*params, cam)
        ranges = r.debug_views()["tile_ranges"]
        longest = int((ranges[:, 1] - ranges[:, 0]).max())
        for _ in range(10):
            r.forward(*params, cam)
        prof = [r.profile_forward(*params, cam) for _ in range(12)][4:]
        fw = {k: round(float(np.median([p[k] for p in prof])), 4) for k in prof[0]}
        out = {"training": training, "serial_long_lists": serial, "pairs": st.pairs, "longest_list": longest, "fwd_ms": fw}
        if training:
            g = torch.sign(img - 0.5) / img.numel()
            pb = [r.profile_backward(g) for _ in range(8)][3:]
            out["bwd_ms"] = {k: round(float(np.median([p[k] for p in pb])), 4) for k in pb[0]}
        print(out, flush=True)
        del r
        torch.cuda.empty_cache()
