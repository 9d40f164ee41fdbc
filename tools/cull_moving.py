#!/usr/bin/env python3
"""Diagnostic (round 6): the occlusion cull under a MOVING camera (GS_FRAME_CULL_DILATE: cuts from the 3 x 3 tile neighbourhood).

    GS_FRAME_CULL_MAX_SHIFT_PX=8 python tools/cull_moving.py [degrees per frame, default 0.01] [frames, default 120]

Pans the cfg5 scene, synchronising after every frame: how many frames were culled, how many of those fell back (were rendered a
second time from the full lists), pairs emitted, and the free-running rate of the same pan."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-gaussian-splatting_amd")]
import torch
from gs_frame import FrameRenderer
from gs_scene import CONFIGS, make_camera, make_scene

step = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 120
dev = torch.device("cuda:0")
n, W, H, _ = CONFIGS["cfg5"]
scene = make_scene(n, W, H, seed=2023)
params = [torch.from_numpy(a).to(dev) for a in (scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb)]
r = FrameRenderer(dev, max_pairs=8_000_000, auto_grow=False)
r._cull_backoff = 0  # (count what happens in EVERY frame: no switching off)
cams = [make_camera(W, H, yaw_deg=step * i) for i in range(frames)]
r.forward(*params, cams[0])
culled = fell = 0
pairs = []
for c in cams[1:]:
    r._cull_off_until = 0
    r.forward(*params, c)
    st = r.stats()
    f = r._frame.flags
    culled += bool(f & 256)
    fell += bool(st.cull_fallback)
    pairs.append(st.pairs)
shift = r._camera_shift_px(cams[-1])
# free-running rate of the pan with the renderer's own policy
r2 = FrameRenderer(dev, max_pairs=8_000_000, auto_grow=False)
for c in cams[:20]:
    r2.forward(*params, c)
torch.cuda.synchronize()
t0 = time.perf_counter()
k = 0
for rep in range(6):
    for c in (cams if rep % 2 == 0 else cams[::-1]):
        r2.forward(*params, c)
        k += 1
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"deg_per_frame": step, "shift_px_estimate_per_frame": round(shift, 3), "frames": frames - 1, "culled": culled,
                  "fell_back": fell, "pairs_emitted_median": sorted(pairs)[len(pairs) // 2], "pairs_min": min(pairs),
                  "pairs_max": max(pairs), "free_running_fps": round(k / dt, 1),
                  "max_shift_px": FrameRenderer.CULL_MAX_SHIFT_PX}))
