"""GPU parity: every reference-API entry point of libgs_amd.so (through the `gaussian` /
`renderer` drop-in modules, i.e. through the C ABI) against the CPU oracle on seeded inputs.

Tolerances: integer / index results bit-exact; cull+project forward bit-exact (both sides are
IEEE fp32 in source order, no contraction); rasterizer images abs 5e-5 (v_exp_f32 vs expf and
the hoisted conic division); the rasterizer's gradients ELEMENT BY ELEMENT in units of each element's conditioning
scale plus a relative L2 bound per tensor (gs_testutil.assert_rows_close / assert_grads_close: the frame path's
standard, at the reference API's own boundary -- no tensor-max criterion on any function of SURVEY.md section 8a).
"""
import numpy as np
import pytest
import torch

import oracle
from gs_scene import make_camera, make_scene
from gs_testutil import (OracleFrame, activate, assert_error_no_worse_than, assert_grads_close, assert_rows_close,
                         assert_rows_error_no_worse_than, frame_scalars, grad_close, robust_padded_grad, to_torch)

pytestmark = pytest.mark.gpu

IMG_ATOL = 5e-5


def dev(a, device, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(device)


def small_case(n=20000, W=320, H=200, seed=7, use_sh=False):
    scene = make_scene(n, W, H, seed=seed, use_sh=use_sh)
    cam = make_camera(W, H, yaw_deg=3.0)
    cam.tran = np.array([0.05, -0.02, 0.1], np.float32)
    return scene, cam


# ------------------------------------------------------------------ K1 / K2 / legacy
def test_global_culling_forward_bit_exact(gpu):
    import gaussian

    scene, cam = small_case()
    qn, sn = activate(scene)
    grid, hw, hh, _ = frame_scalars(cam)
    rp, rc, mk = oracle.global_culling(scene.pos, qn, sn, cam.rot, cam.tran, cam.near, hw, hh)
    n = scene.n
    res_pos = torch.zeros(n, 3, device=gpu)
    res_cov = torch.zeros(n, 2, 2, device=gpu)
    mask = torch.zeros(n, dtype=torch.long, device=gpu)
    gaussian.global_culling(dev(scene.pos, gpu), dev(qn, gpu), dev(sn, gpu), dev(cam.rot, gpu), dev(cam.tran, gpu),
                            res_pos, res_cov, mask, cam.near, hw, hh)
    assert np.array_equal(mask.cpu().numpy(), mk)
    assert 0 < mk.sum() < n
    assert np.array_equal(res_pos.cpu().numpy().view(np.uint32), rp.view(np.uint32)), "pos_i bits differ"
    assert np.array_equal(res_cov.cpu().numpy().view(np.uint32), rc.view(np.uint32)), "cov bits differ"


def test_global_culling_backward(gpu):
    import gaussian

    scene, cam = small_case()
    qn, sn = activate(scene)
    grid, hw, hh, _ = frame_scalars(cam)
    _, _, mk = oracle.global_culling(scene.pos, qn, sn, cam.rot, cam.tran, cam.near, hw, hh)
    rng = np.random.default_rng(1)
    gop = rng.normal(size=(scene.n, 3)).astype(np.float32)
    goc = rng.normal(size=(scene.n, 2, 2)).astype(np.float32)
    ref = oracle.global_culling_backward(scene.pos, qn, sn, cam.rot, cam.tran, gop, goc, mk)
    outs = [torch.zeros(scene.n, k, device=gpu) for k in (3, 4, 3)]
    gaussian.global_culling_backward(dev(scene.pos, gpu), dev(qn, gpu), dev(sn, gpu), dev(cam.rot, gpu),
                                     dev(cam.tran, gpu), dev(gop, gpu), dev(goc, gpu), dev(mk, gpu), *outs)
    # element by element: |got - ref| <= 1e-5 |ref| + 2e-6 x the element's magnitude sum (the backward with every
    # product taken between magnitudes, oracle.global_culling_backward_scale) -- ~16 fp32 ulp of what is summed
    S = oracle.global_culling_backward_scale(scene.pos, qn, sn, cam.rot, cam.tran, np.abs(gop),
                                             np.abs(goc).reshape(-1, 4), mk)
    for o, r, sc, name in zip(outs, ref, S, ("pos", "quat", "scale")):
        o = o.cpu().numpy()
        assert np.all(o[mk == 0] == 0), name
        ok, worst, where, _ = grad_close(o, r, sc, rtol=1e-5, kappa=2e-6)
        assert ok, (name, "worst err/tol", worst, "at", where, float(o[where]), float(r[where]), float(sc[where]))


def test_world2camera_and_jacobian(gpu):
    import gaussian

    rng = np.random.default_rng(3)
    pos = rng.normal(size=(1001, 3)).astype(np.float32) + np.array([0, 0, 4], np.float32)
    _, cam = small_case()
    res = torch.zeros(1001, 3, device=gpu)
    gaussian.world2camera(dev(pos, gpu), dev(cam.rot, gpu), dev(cam.tran, gpu), res)
    ref = oracle.world2camera(pos, cam.rot, cam.tran)
    assert np.allclose(res.cpu().numpy(), ref, rtol=1e-6, atol=1e-6)
    gi = torch.zeros(1001, 3, device=gpu)
    gaussian.world2camera_backward(dev(pos, gpu), dev(cam.rot, gpu), gi)
    assert np.allclose(gi.cpu().numpy(), oracle.world2camera_backward(pos, cam.rot), rtol=1e-6, atol=1e-6)
    jac = torch.zeros(1001, 3, 3, device=gpu)
    gaussian.jacobian(dev(ref, gpu), jac)
    assert np.allclose(jac.cpu().numpy(), oracle.jacobian(ref), rtol=1e-5, atol=1e-6)


def test_renderer_world2camera_autograd(gpu):
    from renderer import world2camera_func

    _, cam = small_case()
    pos = torch.randn(257, 3, device=gpu, requires_grad=True)
    rot, tran = dev(cam.rot, gpu), dev(cam.tran, gpu)
    out = world2camera_func(pos, rot, tran)
    ref = pos @ rot.T + tran
    assert torch.allclose(out, ref, atol=1e-5)
    g = torch.randn_like(out)
    (gin,) = torch.autograd.grad(out, pos, g)
    (gref,) = torch.autograd.grad(ref, pos, g)
    assert torch.allclose(gin, gref, atol=1e-5)


# ------------------------------------------------------------------ K3/K4/K5/K6
def _projected(scene, cam):
    qn, sn = activate(scene)
    grid, hw, hh, _ = frame_scalars(cam)
    rp, rc, mk = oracle.global_culling(scene.pos, qn, sn, cam.rot, cam.tran, cam.near, hw, hh)
    keep = mk.astype(bool)
    return grid, rp[keep], rc[keep].reshape(-1, 4)


@pytest.mark.parametrize("method", [2, 1, 0])
def test_calc_tile_list_and_gather(gpu, method):
    import gaussian

    scene, cam = small_case(n=3000 if method != 2 else 20000, W=160, H=96)
    grid, pos_i, cov = _projected(scene, cam)
    V, T = pos_i.shape[0], grid.n_tiles
    maxp = max(V // 20, 8)  # splatter.py:569
    top, bottom, left, right = grid.tile_edges()
    thresh = 0.05 if method else (grid.tile_geo_length_x / 0.3) ** 2
    n_ref, list_ref = oracle.calc_tile_list(pos_i, cov, maxp, thresh, method, grid.tile_geo_length_x,
                                            grid.tile_geo_length_y, grid.n_tile_x, grid.n_tile_y, grid.leftmost,
                                            grid.topmost, top, bottom, left, right)
    g3 = gaussian.Gaussian3ds()
    g3.pos, g3.cov = dev(pos_i, gpu), dev(cov.reshape(-1, 2, 2), gpu)
    ti = gaussian.Tiles()
    ti.top, ti.bottom, ti.left, ti.right = (dev(a, gpu) for a in (top, bottom, left, right))
    tile_n_point = torch.zeros(T, dtype=torch.int32, device=gpu)
    tile_list = torch.full((T, maxp), -1, dtype=torch.int32, device=gpu)
    gaussian.calc_tile_list(g3, ti, tile_n_point, tile_list, thresh, method, grid.tile_geo_length_x,
                            grid.tile_geo_length_y, grid.n_tile_x, grid.n_tile_y, grid.leftmost, grid.topmost)
    cnt = torch.min(tile_n_point, torch.full_like(tile_n_point, maxp))  # splatter.py:586
    cnt_np = cnt.cpu().numpy()
    assert np.array_equal(cnt_np, np.minimum(n_ref, maxp))
    assert cnt_np.sum() > 0
    lst = tile_list.cpu().numpy()
    for t in range(T):
        if n_ref[t] < maxp:  # strictly below the cap the SET of Gaussians is defined (order is atomic order)
            assert np.array_equal(np.sort(lst[t, :cnt_np[t]]), np.sort(list_ref[t, :n_ref[t]])), t
    accum = torch.cat([torch.zeros(1, dtype=torch.int32, device=gpu), torch.cumsum(cnt, 0).to(torch.int32)])
    M = int(accum[-1])
    gathered = torch.empty(M, dtype=torch.int32, device=gpu)
    tile_ids = torch.empty(M, dtype=torch.int32, device=gpu)
    gaussian.gather_gaussians(accum, tile_list, gathered, tile_ids, int(cnt.max()))
    g_ref, t_ref = oracle.gather_gaussians(accum.cpu().numpy(), lst)
    assert np.array_equal(gathered.cpu().numpy(), g_ref)
    assert np.array_equal(tile_ids.cpu().numpy(), t_ref)


# ------------------------------------------------------------------ radix sort
@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 2047, 2048, 2049, 5000, 300001])
def test_sort_pairs(gpu, n):
    import ctypes as C

    from gaussian import _lib

    rng = np.random.default_rng(n + 11)
    cap = n + 1000
    tiles = rng.integers(0, 700, cap).astype(np.uint64)
    depth = rng.integers(0, 40, cap).astype(np.uint64) * np.uint64(0x01000193) % np.uint64(1 << 32)  # many ties
    keys = (tiles << np.uint64(32)) | depth
    vals = np.arange(cap, dtype=np.uint32)
    k0, v0 = dev(keys.view(np.int64), gpu), dev(vals.view(np.int32), gpu)
    k1, v1 = torch.empty_like(k0), torch.empty_like(v0)
    cnt = torch.tensor([n, 0], dtype=torch.int32, device=gpu)
    tmp = torch.empty(_lib.gs_sort_pairs_tmp_bytes(cap), dtype=torch.uint8, device=gpu)
    in1 = C.c_int(0)
    end_bit = 32 + 10
    _lib.check(_lib.gs_sort_pairs(k0.data_ptr(), v0.data_ptr(), k1.data_ptr(), v1.data_ptr(), cnt.data_ptr(), cap,
                                  end_bit, tmp.data_ptr(), tmp.numel(), C.byref(in1),
                                  torch.cuda.current_stream().cuda_stream), "gs_sort_pairs")
    ks, vs = (k1, v1) if in1.value else (k0, v0)
    order = np.argsort(keys[:n], kind="stable")
    assert np.array_equal(ks.cpu().numpy().view(np.uint64)[:n], keys[:n][order])
    assert np.array_equal(vs.cpu().numpy().view(np.uint32)[:n], vals[:n][order])  # stability


# ------------------------------------------------------------------ K7 / K8 through renderer.py
def _sorted_inputs(scene, cam, thresh=0.05):
    of = OracleFrame(scene, cam, thresh)
    return of


@pytest.mark.parametrize("use_sh", [False, True])
def test_draw_forward_backward(gpu, use_sh):
    from renderer import draw

    scene, cam = small_case(n=12000, W=200, H=120, seed=5, use_sh=use_sh)
    of = _sorted_inputs(scene, cam)
    grid, rays = of.grid, of.rays
    assert of.accum.max() > 0 and np.diff(of.accum).max() > 64  # multi-bucket tiles
    t = [dev(a, gpu).requires_grad_(True) for a in (of.s_pos, of.s_rgb, of.s_opa, of.s_cov.reshape(-1, 2, 2))]
    accum = dev(of.accum, gpu)
    img = draw(*t, accum, grid.padded_height, grid.padded_width, grid.focal_x, grid.focal_y, False, False, use_sh,
               True, dev(rays.rays_o, gpu), dev(rays.lefttop, gpu), dev(rays.dx, gpu), dev(rays.dy, gpu))
    err = np.abs(img.detach().cpu().numpy() - of.padded)
    assert err.max() < IMG_ATOL, err.max()
    rng = np.random.default_rng(9)
    gpad, _ = robust_padded_grad(of, rng.normal(size=of.padded.shape).astype(np.float32))
    img.backward(dev(gpad, gpu))
    ref, scale = oracle.draw_backward(of.s_pos, of.s_rgb, of.s_opa, of.s_cov, of.accum, of.padded, gpad, grid.focal_x,
                                      grid.focal_y, use_sh=use_sh, fast=True, rays_o=rays.rays_o,
                                      lefttop=rays.lefttop, vdx=rays.dx, vdy=rays.dy, with_scale=True)
    got = [x.grad.cpu().numpy() for x in t]
    assert np.all(got[0][:, 2] == 0)  # grad_pos z is never written
    print("draw_backward rows", "sh" if use_sh else "rgb", assert_rows_close(got, ref, scale, f"use_sh={use_sh}"))


@pytest.mark.parametrize("cfg", ["cfg2", "cfg4"])
def test_reference_api_full_size(gpu, cfg):
    """The reference's OWN call sequence at BASELINE.json full size -- 376,467 Gaussians rgb (configs[1]) and 2.4 M
    Gaussians SH degree 2 (configs[3]), 1080p: renderer.global_culling (renderer.py:116-158) -> the mask / sorted-id
    gathers of splatter.py:536-541, 604-613 (the oracle's canonical order for the ids) -> renderer.draw
    (renderer.py:8-87) -> backward through both autograd Functions.  Projection bit-exact; image 5e-5; the FOUR
    (tile, Gaussian)-row gradient tensors of draw_backward element by element; and the five raw-parameter gradients
    that come out of global_culling_backward + the torch activations, element by element too."""
    from gs_scene import CONFIGS
    from renderer import draw, global_culling

    n, W, H, use_sh = CONFIGS[cfg]
    scene, cam = make_scene(n, W, H, seed=2023, use_sh=use_sh), make_camera(W, H)
    params = to_torch(scene, gpu, requires_grad=True)
    pos, quat, scale_raw, opa_raw, rgb_raw = params
    qn = quat / torch.norm(quat, dim=1, keepdim=True)  # splatter.py:519-524 (train.py defaults)
    sn = torch.abs(scale_raw) + 1e-4
    # the oracle's frame on exactly the activated values torch produced (identical inputs for the bit-exact comparison)
    of = OracleFrame(scene, cam, activated=(qn.detach().cpu().numpy(), sn.detach().cpu().numpy()))
    grid, rays = of.grid, of.rays
    hw, hh = grid.frustum_half_extents()
    pos_i, cov, mask = global_culling(pos, qn, sn, dev(cam.rot, gpu), dev(cam.tran, gpu), cam.near, hw, hh)
    assert np.array_equal(mask.cpu().numpy(), of.mask)
    assert np.array_equal(pos_i.detach().cpu().numpy().view(np.uint32), of.pos_i.view(np.uint32))
    assert np.array_equal(cov.detach().cpu().numpy().view(np.uint32), of.cov.view(np.uint32))
    ids = dev(of.ids, gpu, torch.long)
    opa = torch.sigmoid(opa_raw)
    col = rgb_raw if use_sh else torch.sigmoid(rgb_raw)
    s = [pos_i[ids], col[ids], opa[ids], cov[ids]]
    for t in s:
        t.retain_grad()
    img = draw(*s, dev(of.accum, gpu), grid.padded_height, grid.padded_width, grid.focal_x, grid.focal_y, False, False,
               use_sh, True, dev(rays.rays_o, gpu), dev(rays.lefttop, gpu), dev(rays.dx, gpu), dev(rays.dy, gpu))
    assert np.abs(img.detach().cpu().numpy() - of.padded).max() < IMG_ATOL
    gimg = (np.sign(of.image - 0.5) / of.image.size).astype(np.float32)
    gimg, n_masked = of.robust_grad_image(gimg)
    assert n_masked < 0.002 * W * H
    ref, scale = of.backward(gimg, with_scale=True)  # (also leaves the oracle's row gradients in of.pair_grads)
    top, left = grid.crop_offsets()
    gpad = np.zeros_like(of.padded)
    gpad[top:top + H, left:left + W] = gimg
    gpad *= ((of.padded >= 0) & (of.padded <= 1))
    img.backward(dev(gpad, gpu))
    rows = [t.grad.cpu().numpy() for t in s]
    # rows are partial sums -- a Gaussian's terms split over its 3.7 tiles, 3 x as many elements as parameters --: the
    # element-wise tolerance in units of the conditioning scale is 1e-4 here where the parameter gradients get 3e-5
    # (measured at cfg4: ONE of 188 M SH-coefficient row elements, 1e-10 of the tensor's maximum, at 2.44 x the latter)
    print(cfg, "reference API rows:", assert_rows_close(rows, of.pair_grads, of.pair_scales, cfg, kappa=1e-4))
    got = [t.grad.cpu().numpy() for t in params]
    print(cfg, "reference API parameters:", assert_grads_close(got, ref, scale, cfg))
    # and the calibrated, scale-free statement (tests/test_grad_calibration.py): against the double-precision evaluation
    # the API's error quantiles are within CALIB_K x those of the reference's own fp32 arithmetic, rows and parameters
    rows64, par64 = of.backward_f64(gimg)
    for name, (qh, qr) in assert_rows_error_no_worse_than(rows, rows64, of.pair_grads, cfg).items():
        print(f"CALIB {cfg} API rows {name}: hip", ["%.2e" % v for v in qh], "reference arithmetic", ["%.2e" % v for v in qr])
    for name, (qh, qr) in assert_error_no_worse_than(got, par64, ref, cfg).items():
        print(f"CALIB {cfg} API {name}: hip", ["%.2e" % v for v in qh], "reference arithmetic", ["%.2e" % v for v in qr])


@pytest.mark.parametrize("use_sh", [False, True])
def test_draw_exact_exp_flavour(gpu, use_sh):
    """`fast=False` of draw / draw_backward (gaussian.cu:922-923, 596-603): the exponent's argument in the reference's
    own float order over the double 2 det + 1e-14, then a double-precision exp.  Against the oracle's fast=False the
    image agrees to 2e-6 -- more than ten times closer than the `fast` flavour (v_exp_f32 on the hoisted conic) is
    allowed to be -- and the two flavours really are different kernels (their images differ)."""
    from renderer import draw

    scene, cam = small_case(n=12000, W=200, H=120, seed=5, use_sh=use_sh)
    of = _sorted_inputs(scene, cam)
    grid, rays = of.grid, of.rays
    kw = dict(use_sh=use_sh, rays_o=rays.rays_o, lefttop=rays.lefttop, vdx=rays.dx, vdy=rays.dy)
    ref = {f: oracle.draw(of.s_pos, of.s_rgb, of.s_opa, of.s_cov, of.accum, grid.padded_height, grid.padded_width,
                          grid.focal_x, grid.focal_y, fast=f, **kw) for f in (False, True)}
    rb = [dev(a, gpu) for a in (rays.rays_o, rays.lefttop, rays.dx, rays.dy)]
    imgs, grads = {}, {}
    gpad, _ = robust_padded_grad(of, np.random.default_rng(9).normal(size=ref[False].shape).astype(np.float32))
    for fast in (False, True):
        t = [dev(a, gpu).requires_grad_(True) for a in (of.s_pos, of.s_rgb, of.s_opa, of.s_cov.reshape(-1, 2, 2))]
        img = draw(*t, dev(of.accum, gpu), grid.padded_height, grid.padded_width, grid.focal_x, grid.focal_y, False,
                   False, use_sh, fast, *rb)
        imgs[fast] = img.detach().cpu().numpy()
        img.backward(dev(gpad, gpu))
        grads[fast] = [x.grad.cpu().numpy() for x in t]
    assert np.abs(imgs[False] - ref[False]).max() < 2e-6, np.abs(imgs[False] - ref[False]).max()
    assert np.abs(imgs[True] - ref[True]).max() < IMG_ATOL
    assert np.abs(imgs[False] - imgs[True]).max() > 0
    gref, gscale = oracle.draw_backward(of.s_pos, of.s_rgb, of.s_opa, of.s_cov, of.accum, ref[False], gpad,
                                        grid.focal_x, grid.focal_y, fast=False, with_scale=True, **kw)
    assert_rows_close(grads[False], gref, gscale, f"fast=False use_sh={use_sh}")


def test_draw_flags_weight_normalize_and_sigmoid(gpu):
    import gaussian

    scene, cam = small_case(n=6000, W=96, H=64, seed=2)
    of = _sorted_inputs(scene, cam)
    grid = of.grid
    args = [dev(a, gpu) for a in (of.s_pos, of.s_rgb, of.s_opa, of.s_cov.reshape(-1, 2, 2))]
    accum = dev(of.accum, gpu)
    for wn, sg in ((True, False), (False, True), (True, True)):
        res = torch.zeros(grid.padded_height, grid.padded_width, 3, device=gpu)
        gaussian.draw(*args, accum, res, grid.focal_x, grid.focal_y, wn, sg, True, None, None, None, None, False)
        ref = oracle.draw(of.s_pos, of.s_rgb, of.s_opa, of.s_cov, of.accum, grid.padded_height, grid.padded_width,
                          grid.focal_x, grid.focal_y, weight_normalize=wn, sigmoid=sg, fast=True)
        assert np.abs(res.cpu().numpy() - ref).max() < 2e-4, (wn, sg)


@pytest.mark.parametrize("use_sh", [False, True])
@pytest.mark.parametrize("saturated", [False, True])
def test_draw_backward_sigmoid_flag(gpu, use_sh, saturated):
    """sigmoid=True through forward AND backward (gaussian.cu:593-594, 622-630, 727, 918, 930).  `saturated`:
    ordinary opacities, where p0 ~ 1e3 drives the squashed alpha to 1 and pixels stop after a few Gaussians;
    otherwise opacities scaled so that every pixel stays live (the regime tests/test_ref_live.py pins against
    the reference kernels)."""
    from renderer import draw

    scene, cam = small_case(n=9000, W=160, H=96, seed=14, use_sh=use_sh)
    of = _sorted_inputs(scene, cam)
    grid, rays = of.grid, of.rays
    assert np.diff(of.accum).max() > 64
    opa = of.s_opa
    if not saturated:
        det = of.s_cov[:, 0] * of.s_cov[:, 3] - of.s_cov[:, 1] * of.s_cov[:, 2]
        opa = (of.s_opa * 0.02 / (np.pi / 2 / np.sqrt(det + 1e-7))).astype(np.float32)
    t = [dev(a, gpu).requires_grad_(True) for a in (of.s_pos, of.s_rgb, opa, of.s_cov.reshape(-1, 2, 2))]
    accum = dev(of.accum, gpu)
    rv = [dev(v, gpu) for v in (rays.rays_o, rays.lefttop, rays.dx, rays.dy)]
    img = draw(*t, accum, grid.padded_height, grid.padded_width, grid.focal_x, grid.focal_y, False, True, use_sh,
               True, *rv)
    kw = dict(use_sh=use_sh, fast=True, sigmoid=True, rays_o=rays.rays_o, lefttop=rays.lefttop, vdx=rays.dx,
              vdy=rays.dy)
    want = oracle.draw(of.s_pos, of.s_rgb, opa, of.s_cov, of.accum, grid.padded_height, grid.padded_width,
                       grid.focal_x, grid.focal_y, **kw)
    assert np.abs(img.detach().cpu().numpy() - want).max() < 2e-4
    gpad = np.random.default_rng(10).normal(size=want.shape).astype(np.float32)
    img.backward(dev(gpad, gpu))
    # the backward replays the forward from ITS image argument: hand the oracle the same image the GPU produced
    ref, scale = oracle.draw_backward(of.s_pos, of.s_rgb, opa, of.s_cov, of.accum, img.detach().cpu().numpy(), gpad,
                                      grid.focal_x, grid.focal_y, with_scale=True, **kw)
    got = [x.grad.cpu().numpy() for x in t]
    for g, name in zip(got, ("pos", "rgb", "opa", "cov")):
        assert np.isfinite(g).all(), name
    print("sigmoid flag", use_sh, saturated, assert_rows_close(got, ref, scale, f"sigmoid use_sh={use_sh} saturated={saturated}"))


def test_draw_empty_and_errors(gpu):
    import gaussian

    res = torch.zeros(32, 48, 3, device=gpu)
    accum = torch.zeros(2 * 3 + 1, dtype=torch.int32, device=gpu)
    z = lambda *s: torch.zeros(*s, device=gpu)
    gaussian.draw(z(0, 3), z(0, 3), z(0), z(0, 2, 2), accum, res, 100.0, 100.0, False, False, True, None, None, None,
                  None, False)
    assert float(res.abs().max()) == 0.0
    with pytest.raises(RuntimeError):  # wrong dtype, like data_ptr<float>() in the reference
        gaussian.draw(z(0, 3).double(), z(0, 3), z(0), z(0, 2, 2), accum, res, 100.0, 100.0, False, False, True,
                      None, None, None, None, False)
    with pytest.raises(RuntimeError):  # CPU tensor
        gaussian.world2camera(torch.zeros(4, 3), z(3, 3), z(3), z(4, 3))
