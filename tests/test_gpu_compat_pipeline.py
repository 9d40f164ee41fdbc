"""GPU: the zero-change integration mode end to end.

The reference's per-frame flow (splatter.py:513-655: activations -> renderer.global_culling -> mask gathers ->
gaussian.calc_tile_list -> clamp / cumsum -> gaussian.gather_gaussians -> attribute gather -> torch.sort on the
fp32 composite key -> attribute gather -> renderer.draw -> clamp -> crop) is driven here through the drop-in
``gaussian`` and ``renderer`` modules exactly as splatter.py would call them, with plain torch ops in between, and
compared with the fused frame path and with the oracle -- image and parameter gradients."""
import numpy as np
import pytest
import torch

from gs_testutil import OracleFrame, rel_err, to_torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def reference_style_frame(params, cam, grid, rays, thresh=0.05):
    """The steps of Splatter.forward with the reference's call signatures."""
    import gaussian
    import renderer

    pos, quat, scale, opa, rgb = params
    dev = pos.device
    quat_n = quat / quat.norm(dim=1, keepdim=True)                       # splatter.py:519
    scale_a = scale.abs() + 1e-4                                          # :521
    half_w, half_h = grid.frustum_half_extents()
    rot, tran = torch.from_numpy(cam.rot).to(dev), torch.from_numpy(cam.tran).to(dev)
    pos_i, cov, mask = renderer.global_culling(pos, quat_n, scale_a, rot, tran, cam.near, half_w, half_h)
    keep = mask.bool()                                                    # :536-541
    pos_i, cov, rgb_k, opa_k = pos_i[keep], cov[keep], rgb[keep], opa[keep]
    V, T = pos_i.shape[0], grid.n_tiles
    g3, ti = gaussian.Gaussian3ds(), gaussian.Tiles()
    g3.pos, g3.cov = pos_i.detach().contiguous(), cov.detach().contiguous()
    ti.top, ti.bottom, ti.left, ti.right = (torch.from_numpy(a).to(dev) for a in grid.tile_edges())
    maxp = max(V // 20, 8)                                                # :569
    tile_n_point = torch.zeros(T, dtype=torch.int32, device=dev)
    tile_list = torch.ones(T, maxp, dtype=torch.int32, device=dev) * -1   # :570
    gaussian.calc_tile_list(g3, ti, tile_n_point, tile_list, thresh, 2, grid.tile_geo_length_x, grid.tile_geo_length_y,
                            grid.n_tile_x, grid.n_tile_y, grid.leftmost, grid.topmost)
    cnt = torch.min(tile_n_point, torch.ones_like(tile_n_point) * maxp)  # :586
    accum = torch.cat([torch.zeros(1, dtype=torch.int32, device=dev), torch.cumsum(cnt, 0).to(torch.int32)])
    M = int(accum[-1])
    gathered = torch.zeros(M, dtype=torch.int32, device=dev)
    tile_ids = torch.zeros(M, dtype=torch.int32, device=dev)
    gaussian.gather_gaussians(accum, tile_list, gathered, tile_ids, int(cnt.max()))
    idx = gathered.long()
    t_pos, t_rgb, t_opa, t_cov = pos_i[idx], rgb_k[idx], opa_k[idx], cov[idx]   # :600-604
    depth = t_pos[:, 2]
    # :610-611 builds this key in fp32, which cannot hold (tile, depth) beyond a few dozen tiles;
    # the same formula in float64 keeps the test about the modules, not about that rounding
    key = depth.double() + tile_ids.double() * (depth.max().double() + 1)
    order = torch.sort(key)[1]
    t_pos, t_rgb, t_opa, t_cov = t_pos[order], t_rgb[order], t_opa[order], t_cov[order]
    image = renderer.draw(t_pos, t_rgb.sigmoid(), t_opa.sigmoid(), t_cov, accum, grid.padded_height, grid.padded_width,
                          grid.focal_x, grid.focal_y, False, False, False, True,
                          *(torch.from_numpy(a).to(dev) for a in (rays.rays_o, rays.lefttop, rays.dx, rays.dy)))
    image = image.clamp(0, 1)                                             # :652
    top, left = grid.crop_offsets()
    return image[top:top + grid.height, left:left + grid.width], int(tile_n_point.max()), maxp


def test_reference_call_sequence_matches_frame_path_and_oracle(gpu):
    from gs_frame import FrameRenderer
    from gs_scene import make_camera, make_scene
    from gs_testutil import frame_scalars

    W, H = 250, 186  # not multiples of 16: padded to 256 x 192 (192 tiles), centred crop
    scene, cam = make_scene(3000, W, H, seed=5), make_camera(W, H, yaw_deg=2.0)
    grid, _, _, rays = frame_scalars(cam)
    of = OracleFrame(scene, cam)
    params = to_torch(scene, gpu, requires_grad=True)
    img, max_tile, maxp = reference_style_frame(params, cam, grid, rays)
    assert max_tile <= maxp, "keep the test below the reference's per-tile cap (it would drop Gaussians)"
    got = img.detach().cpu().numpy()
    assert np.abs(got - of.image).max() < 2e-4
    fused = FrameRenderer(gpu, max_pairs=1 << 16).forward(*[p.detach() for p in params], cam)[0]
    assert float((fused - img.detach()).abs().max()) < 2e-4
    gimg = np.random.default_rng(2).normal(size=of.image.shape).astype(np.float32)
    img.backward(torch.from_numpy(gimg).to(gpu))
    ref = of.backward(gimg)
    for t, name in zip(params, ("pos", "quat", "scale", "opa", "rgb")):
        assert t.grad is not None and bool(torch.isfinite(t.grad).all()), name
        assert rel_err(t.grad.cpu().numpy(), ref[name]) < 2e-3, (name, rel_err(t.grad.cpu().numpy(), ref[name]))
