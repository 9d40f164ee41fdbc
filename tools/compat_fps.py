"""Render FPS of the ZERO-CHANGE integration mode: the reference's own per-frame call sequence
(splatter.py:513-655: torch masks, T x MAXP table, cumsum, two attribute gathers, torch.sort, host syncs)
driven through the drop-in gaussian / renderer modules -- against the fused frame path on the same scene.

    python tools/compat_fps.py [cfg2 cfg5 ...]

`reference_style_frame` is the restatement of that call sequence (also used by tests/test_gpu_compat_pipeline.py and
by bench.py's `compat_mode` leg); the reference's splatter.py itself cannot travel to the GPU box."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "3d-gaussian-splatting_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import torch  # noqa: E402


def reference_style_frame(params, cam, grid, rays, thresh=0.05, key_dtype=None, marks=None):
    """The steps of Splatter.forward with the reference's call signatures.  `marks` (a list) receives
    (segment name, seconds) pairs: wall time of every segment with a device synchronisation after it."""
    import gaussian
    import renderer

    t_last = [time.perf_counter()]

    def mark(name):
        if marks is not None:
            torch.cuda.synchronize()
            now = time.perf_counter()
            marks.append((name, now - t_last[0]))
            t_last[0] = now

    pos, quat, scale, opa, rgb = params
    dev = pos.device
    quat_n = quat / quat.norm(dim=1, keepdim=True)                       # splatter.py:519
    scale_a = scale.abs() + 1e-4                                          # :521
    half_w, half_h = grid.frustum_half_extents()
    rot, tran = torch.from_numpy(cam.rot).to(dev), torch.from_numpy(cam.tran).to(dev)
    mark("activations (torch)")
    pos_i, cov, mask = renderer.global_culling(pos, quat_n, scale_a, rot, tran, cam.near, half_w, half_h)
    mark("renderer.global_culling")
    keep = mask.bool()                                                    # :536-541
    pos_i, cov, rgb_k, opa_k = pos_i[keep], cov[keep], rgb[keep], opa[keep]
    V, T = pos_i.shape[0], grid.n_tiles
    mark("mask compactions (torch)")
    g3, ti = gaussian.Gaussian3ds(), gaussian.Tiles()
    g3.pos, g3.cov = pos_i.detach().contiguous(), cov.detach().contiguous()
    ti.top, ti.bottom, ti.left, ti.right = (torch.from_numpy(a).to(dev) for a in grid.tile_edges())
    maxp = max(V // 20, 8)                                                # :569
    tile_n_point = torch.zeros(T, dtype=torch.int32, device=dev)
    tile_list = torch.ones(T, maxp, dtype=torch.int32, device=dev) * -1   # :570
    mark("T x MAXP table fill (torch)")
    gaussian.calc_tile_list(g3, ti, tile_n_point, tile_list, thresh, 2, grid.tile_geo_length_x, grid.tile_geo_length_y,
                            grid.n_tile_x, grid.n_tile_y, grid.leftmost, grid.topmost)
    mark("gaussian.calc_tile_list")
    cnt = torch.min(tile_n_point, torch.ones_like(tile_n_point) * maxp)  # :586
    accum = torch.cat([torch.zeros(1, dtype=torch.int32, device=dev), torch.cumsum(cnt, 0).to(torch.int32)])
    M = int(accum[-1])
    gathered = torch.zeros(M, dtype=torch.int32, device=dev)
    tile_ids = torch.zeros(M, dtype=torch.int32, device=dev)
    max_cnt = int(cnt.max())
    mark("clamp / cumsum / allocations (torch, 2 host syncs)")
    gaussian.gather_gaussians(accum, tile_list, gathered, tile_ids, max_cnt)
    mark("gaussian.gather_gaussians")
    idx = gathered.long()
    t_pos, t_rgb, t_opa, t_cov = pos_i[idx], rgb_k[idx], opa_k[idx], cov[idx]   # :600-604
    depth = t_pos[:, 2]
    mark("attribute gather 1 (torch)")
    # :610-611 builds this key in fp32, which cannot hold (tile, depth) beyond a few dozen tiles;
    # the same formula in float64 (the default here) keeps the parity test about the modules, not about that rounding;
    # key_dtype=torch.float32 is the literal reference (what the timing of the zero-change mode uses)
    kd = key_dtype or torch.float64
    key = depth.to(kd) + tile_ids.to(kd) * (depth.max().to(kd) + 1)
    order = torch.sort(key)[1]
    mark("composite key + torch.sort")
    t_pos, t_rgb, t_opa, t_cov = t_pos[order], t_rgb[order], t_opa[order], t_cov[order]
    mark("attribute gather 2 (torch)")
    image = renderer.draw(t_pos, t_rgb.sigmoid(), t_opa.sigmoid(), t_cov, accum, grid.padded_height, grid.padded_width,
                          grid.focal_x, grid.focal_y, False, False, False, True,
                          *(torch.from_numpy(a).to(dev) for a in (rays.rays_o, rays.lefttop, rays.dx, rays.dy)))
    mark("sigmoids (torch) + renderer.draw")
    image = image.clamp(0, 1)                                             # :652
    top, left = grid.crop_offsets()
    out = image[top:top + grid.height, left:left + grid.width]
    mark("clamp + crop (torch)")
    return out, int(tile_n_point.max()), maxp


def measure(cfg, dev, frames=20, fused_frames=100):
    """FPS of the zero-change mode (literal fp32 sort key) and of the fused frame path on CONFIGS[cfg]; peak memory."""
    from gs_frame import FrameRenderer
    from gs_geometry import RayBasis, TileGrid
    from gs_scene import CONFIGS, make_camera, make_scene

    n, W, H, use_sh = CONFIGS[cfg]
    assert not use_sh, "the zero-change timing uses rgb logits"
    scene, cam = make_scene(n, W, H, seed=2023), make_camera(W, H)
    params = [torch.from_numpy(a).to(dev) for a in (scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb)]
    grid = TileGrid(cam.width, cam.height, cam.focal_x, cam.focal_y)
    rays = RayBasis.from_camera(cam.rot, cam.tran, grid.padded_height, grid.padded_width, grid.focal_x, grid.focal_y)
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(dev)
    base = torch.cuda.memory_allocated(dev)
    with torch.no_grad():
        img64, max_tile, maxp = reference_style_frame(params, cam, grid, rays)  # float64 key: the comparable image
        for _ in range(2):
            reference_style_frame(params, cam, grid, rays, key_dtype=torch.float32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            reference_style_frame(params, cam, grid, rays, key_dtype=torch.float32)
        torch.cuda.synchronize()
        t_compat = (time.perf_counter() - t0) / frames
    peak = torch.cuda.max_memory_allocated(dev) - base
    segs = {}
    with torch.no_grad():
        for _ in range(3):  # per-segment wall time, a device synchronisation after every segment (so their sum exceeds
            marks = []      # the free-running frame time): where the zero-change mode spends its frame
            reference_style_frame(params, cam, grid, rays, key_dtype=torch.float32, marks=marks)
            for k, v in marks:
                segs.setdefault(k, []).append(v)
    segments_ms = {k: round(sorted(v)[len(v) // 2] * 1e3, 3) for k, v in segs.items()}
    r = FrameRenderer(dev, max_pairs=1 << 20, auto_grow=True)
    fused = r.forward(*params, cam)[0]
    r.auto_grow = False
    for _ in range(5):
        r.forward(*params, cam)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(fused_frames):
        r.forward(*params, cam)
    torch.cuda.synchronize()
    t_fused = (time.perf_counter() - t0) / fused_frames
    return {"n_gaussians": n, "zero_change_fps": round(1 / t_compat, 1), "zero_change_ms": round(t_compat * 1e3, 3),
            "zero_change_peak_bytes": int(peak), "table_bytes": int(4 * grid.n_tiles * maxp),
            "max_per_tile": max_tile, "per_tile_cap": maxp, "fused_fps": round(1 / t_fused, 1),
            "fused_ms": round(t_fused * 1e3, 4), "max_abs_image_difference": float((fused - img64).abs().max()),
            "zero_change_segments_ms": segments_ms}


if __name__ == "__main__":
    import json

    for c in (sys.argv[1:] or ["cfg2"]):
        print(c, json.dumps(measure(c, torch.device("cuda:0"))), flush=True)
