"""raster/sort stage time vs number of pairs at 1080p (slope = per-pair cost, intercept = per-tile overhead)."""
import sys
sys.path[:0] = ['/root/repo', '/root/repo/3d-gaussian-splatting_amd']
import numpy as np, torch
from gs_frame import FrameRenderer
from gs_scene import make_camera, make_scene
dev = torch.device('cuda:0')
W, H = 1920, 1080
for n in (1000, 50_000, 100_000, 200_000, 376_467, 750_000, 1_500_000):
    scene = make_scene(n, W, H, seed=2023)
    scene.opa = (scene.opa - 4.0).astype(np.float32)   # low opacity: no early termination
    cam = make_camera(W, H)
    params = [torch.from_numpy(a).to(dev) for a in (scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb)]
    r = FrameRenderer(dev, max_pairs=1 << 20, auto_grow=True)
    r.forward(*params, cam); st = r.stats(); r.max_pairs = int(st.pairs * 1.1) + 4096; r.auto_grow = False
    r.forward(*params, cam)
    prof = [r.profile_forward(*params, cam) for _ in range(12)][4:]
    fw = {k: round(float(np.median([p[k] for p in prof])) * 1e3, 1) for k in prof[0]}
    print(f"N={n:8d} M={st.pairs:8d} M/T={st.pairs/8160:7.1f}  us: {fw}", flush=True)
