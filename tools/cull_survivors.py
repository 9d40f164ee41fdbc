#!/usr/bin/env python3
"""Diagnostic (round 6): what the compacting project kernel of an occlusion-culled frame skips.

Renders the 2.4 M-Gaussian scene a few times from one pose, then reads the rectangle records of a culled frame: Gaussians
outside the frustum, Gaussians the occlusion test skipped (depth bits set, no tile), survivors, and -- from an unculled
frame's rectangles and the cut table's effect on the pair count -- what a perfect test could have skipped."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-gaussian-splatting_amd")]
import numpy as np
import torch
from gs_frame import FrameRenderer
from gs_scene import CONFIGS, make_camera, make_scene

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
dev = torch.device("cuda:0")
n, W, H, _ = CONFIGS[cfg]
scene = make_scene(n, W, H, seed=2023)
params = [torch.from_numpy(a).to(dev) for a in (scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb)]
cam = make_camera(W, H)
out = {}
for cull in (False, True):
    r = FrameRenderer(dev, max_pairs=8_000_000, auto_grow=False, occlusion_cull=cull)
    for _ in range(8):
        r.forward(*params, cam)
    torch.cuda.synchronize()
    rc = r._rects(allow_culled=True).cpu().numpy()  # (culled frame: fresh for the projected Gaussians only)
    st = r.stats()
    prof = [r.profile_forward(*params, cam) for _ in range(8)][3:]
    key = "culled" if cull else "plain"
    out[key] = {"flag": bool(r._frame.flags & 256), "visible": st.visible, "pairs": st.pairs,
                "outside_frustum": int((rc[:, 2] == 0).sum()), "no_tile_but_visible": int(((rc[:, 2] != 0) & (rc[:, 3] == 0)).sum()),
                "with_tiles": int((rc[:, 3] != 0).sum()), "rect_area_sum": int(rc[:, 3].astype(np.int64).sum()),
                "stage_ms": {k: round(float(np.median([p[k] for p in prof])), 4) for k in prof[0]}}
    if cull:
        # survivors whose every pair was trimmed all the same: the strip entries say (rects alone do not) -- estimate from
        # the emitted pairs against the survivors' rectangle areas
        out[key]["emitted_over_survivor_area"] = round(st.pairs / max(1, out[key]["rect_area_sum"]), 4)
print(json.dumps(out, indent=1))
