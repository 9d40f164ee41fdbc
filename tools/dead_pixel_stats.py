#!/usr/bin/env python3
"""How much of the backward compositing falls on pixels that have already stopped, by granularity.

    GS_AMD_LIB=build/variants/diag/libgs_amd.so python tools/dead_pixel_stats.py [cfg5]     # diag = -DGS_DIAG_CKPT

Reads the forward's checkpoints (transmittance of every pixel at every 64th Gaussian of its tile's list) and counts, over
all (tile, bucket) pairs the backward works on and weighted with the bucket's Gaussians, the share of (pixel, Gaussian)
evaluations whose pixel / 16 x 1 pixel row / 4 x 4 pixel block / 16 x 8 half tile had stopped at the bucket's start.  The
SH backward on the matrix pipe leaves out 16 x 1 rows (at a granularity of 16 Gaussians: it sees more than this
64-Gaussian estimate); the row : block ratio says what another unit of a step would buy."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-gaussian-splatting_amd")]
import torch

from gaussian import _lib
from gs_frame import FrameRenderer
from gs_scene import CONFIGS, make_camera, make_scene

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
n, W, H, use_sh = CONFIGS[cfg]
dev = torch.device("cuda:0")
sc = make_scene(n, W, H, seed=2023, use_sh=use_sh)
cam = make_camera(W, H)
params = [torch.from_numpy(a).to(dev) for a in (sc.pos, sc.quat, sc.scale, sc.opa, sc.rgb)]
r = FrameRenderer(dev, max_pairs=1 << 20, training=True, auto_grow=True)
r.forward(*params, cam)
r.max_pairs = int(r.stats().pairs * 1.1) + 4096
r.auto_grow = False
r.forward(*params, cam)
torch.cuda.synchronize()
f = r._frame
lib = C.CDLL(_lib.LIB_PATH)
ck, mb = C.c_void_p(), C.c_int64()
lib.gs_diag_ckpt.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
assert lib.gs_diag_ckpt(C.byref(f), C.byref(ck), C.byref(mb)) == 0
ws = r._ws
off = ck.value - ws.data_ptr()
ckpt = ws[off:off + mb.value * 256 * 16].view(torch.float32).view(mb.value, 256, 4)
ptr = C.c_void_p()
_lib.check(_lib.gs_frame_debug_tile_nproc(C.byref(f), C.byref(ptr)), "nproc")
T = r._grid.n_tiles
o2 = ptr.value - ws.data_ptr()
nproc = ws[o2:o2 + 4 * T].view(torch.int32).to(torch.int64)
start = r.debug_views()["tile_ranges"][:, 0].to(torch.int64)
tot = {k: 0.0 for k in ("evals", "pixel", "row16x1", "block4x4", "half16x8", "quad8x8")}
maxb = int(((nproc + 63) // 64).max())
tiles = torch.arange(T, device=dev)
for b in range(1, maxb):  # bucket 0 starts from T = 1: nothing has stopped
    sel = (nproc + 63) // 64 > b
    if not bool(sel.any()):
        break
    t = tiles[sel]
    slot = start[sel] // 64 + t + b
    cnt = torch.clamp(nproc[sel] - 64 * b, max=64).to(torch.float64)          # Gaussians of the bucket
    dead = ckpt[slot, :, 0] <= 1e-4                                           # [tiles, 256], index 16 y + x
    d = dead.view(-1, 16, 16)
    w = cnt[:, None]
    tot["evals"] += float((cnt * 256).sum())
    tot["pixel"] += float((dead.sum(1).to(torch.float64) * cnt).sum())
    tot["row16x1"] += float((d.all(2).sum(1).to(torch.float64) * 16 * cnt).sum())
    blk = d.view(-1, 4, 4, 4, 4).permute(0, 1, 3, 2, 4).reshape(-1, 16, 16).all(2)
    tot["block4x4"] += float((blk.sum(1).to(torch.float64) * 16 * cnt).sum())
    tot["half16x8"] += float((d.view(-1, 2, 128).all(2).sum(1).to(torch.float64) * 128 * cnt).sum())
    q8 = d.view(-1, 2, 8, 2, 8).permute(0, 1, 3, 2, 4).reshape(-1, 4, 64).all(2)
    tot["quad8x8"] += float((q8.sum(1).to(torch.float64) * 64 * cnt).sum())
first = float((torch.clamp(nproc, max=64) * 256).sum())
all_evals = tot["evals"] + first
print(f"{cfg}: composited steps {int(nproc.sum())}, evaluations {all_evals:.3e} (first buckets {first / all_evals:.1%})")
for k in ("pixel", "row16x1", "block4x4", "quad8x8", "half16x8"):
    print(f"  stopped at the bucket's start, unit {k:9s}: {tot[k] / all_evals:.2%} of all evaluations")
