"""Render FPS of the ZERO-CHANGE integration mode at cfg2: the reference's own per-frame call sequence
(splatter.py:513-655: torch masks, T x MAXP table, cumsum, two attribute gathers, torch.sort, host syncs)
driven through the drop-in gaussian / renderer modules -- against the fused frame path on the same scene."""
import sys
import time

sys.path[:0] = ['/root/repo', '/root/repo/3d-gaussian-splatting_amd', '/root/repo/tests']
import torch  # noqa: E402

from gs_frame import FrameRenderer  # noqa: E402
from gs_scene import CONFIGS, make_camera, make_scene  # noqa: E402
from gs_testutil import frame_scalars  # noqa: E402
from test_gpu_compat_pipeline import reference_style_frame  # noqa: E402

dev = torch.device('cuda:0')
n, W, H, _ = CONFIGS['cfg2']
scene, cam = make_scene(n, W, H, seed=2023), make_camera(W, H)
params = [torch.from_numpy(a).to(dev) for a in (scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb)]
grid, _, _, rays = frame_scalars(cam)
with torch.no_grad():
    for _ in range(3):
        img, max_tile, maxp = reference_style_frame(params, cam, grid, rays)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        reference_style_frame(params, cam, grid, rays)
    torch.cuda.synchronize()
    t_compat = (time.perf_counter() - t0) / 20
r = FrameRenderer(dev, max_pairs=1_300_000, auto_grow=False)
for _ in range(5):
    fused = r.forward(*params, cam)[0]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    r.forward(*params, cam)
torch.cuda.synchronize()
t_fused = (time.perf_counter() - t0) / 200
print(f"compat mode (reference call sequence on these kernels): {1 / t_compat:.1f} FPS ({t_compat * 1e3:.2f} ms), "
      f"max per tile {max_tile} / cap {maxp}; fused frame path: {1 / t_fused:.1f} FPS ({t_fused * 1e3:.3f} ms); "
      f"max |image difference| {float((fused - img).abs().max()):.2e}")
