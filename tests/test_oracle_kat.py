"""CPU tier: the oracle against closed-form known answers, torch.autograd on an independent
fp64 restatement (oracle/torch_ref.py) and finite differences -- SURVEY.md section 4's plan for
pinning what the reference itself never tested."""
import numpy as np
import pytest
import torch

import oracle
from oracle import torch_ref
from gs_geometry import TileGrid
from gs_scene import make_camera, make_scene
from gs_testutil import OracleFrame, activate, frame_scalars, rel_err


def test_single_isotropic_gaussian_closed_form():
    """One isotropic Gaussian on the optical axis: cov2d = (s/z)^2 I and
    pixel = rgb * opa * exp(-(px^2+py^2) / (2 (s/z)^2))."""
    s, z, opa = 0.05, 2.0, 0.7
    rgb = np.array([[0.2, 0.5, 0.9]], np.float32)
    pos = np.array([[0, 0, z]], np.float32)
    quat = np.array([[1, 0, 0, 0]], np.float32)
    scale = np.full((1, 3), s, np.float32)
    rot, tran = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    rp, rc, mk = oracle.global_culling(pos, quat, scale, rot, tran, 0.3, 10, 10)
    assert mk[0] == 1 and np.allclose(rp[0], [0, 0, z])
    assert np.allclose(rc[0], np.eye(2) * (s / z) ** 2, rtol=1e-6, atol=1e-12)
    W = H = 32
    f = 24.0
    accum = np.array([0] + [1] * 4, np.int32)  # the Gaussian is listed in all 4 tiles
    accum = np.arange(5, dtype=np.int32)
    rep = lambda a: np.repeat(a, 4, axis=0)
    img = oracle.draw(rep(rp), rep(rgb), np.full(4, opa, np.float32), rep(rc.reshape(1, 4)), accum, H, W, f, f, fast=True)
    ix, iy = np.meshgrid(np.arange(W), np.arange(H))
    px, py = (ix + 0.5 - W // 2) / f, (iy + 0.5 - H // 2) / f
    expect = opa * np.exp(-(px ** 2 + py ** 2) / (2 * (s / z) ** 2))
    assert np.allclose(img, expect[..., None] * rgb[0], rtol=2e-5, atol=1e-7)


def test_two_stacked_gaussians_front_to_back():
    """c1 a1 + c2 a2 (1 - a1) at the common centre; order matters."""
    cov = np.array([[1e-2, 0, 0, 1e-2]] * 2, np.float32)
    pos = np.array([[0.5 / 24, 0.5 / 24, 1.0], [0.5 / 24, 0.5 / 24, 2.0]], np.float32)  # centre of pixel (16,16)
    rgb = np.array([[1, 0, 0], [0, 1, 0]], np.float32)
    opa = np.array([0.6, 0.5], np.float32)
    accum = np.array([0, 0, 0, 0, 2], np.int32)  # both in tile 3 (pixels 16..31)
    img = oracle.draw(pos, rgb, opa, cov, accum, 32, 32, 24.0, 24.0, fast=True)
    assert np.allclose(img[16, 16], [0.6, 0.5 * 0.4, 0], atol=1e-6)
    img2 = oracle.draw(pos[::-1], rgb[::-1], opa[::-1], cov, accum, 32, 32, 24.0, 24.0, fast=True)
    assert np.allclose(img2[16, 16], [0.6 * 0.5, 0.5, 0], atol=1e-6)
    assert np.all(img[:16] == 0)  # other tiles are empty


def test_early_termination_threshold():
    """A pixel stops compositing once transmittance < 1e-4 (checked before each Gaussian)."""
    n = 6
    cov = np.tile(np.array([[1.0, 0, 0, 1.0]], np.float32), (n, 1))
    pos = np.zeros((n, 3), np.float32)
    rgb = np.ones((n, 3), np.float32)
    opa = np.full(n, 0.95, np.float32)  # T: 1, 0.05, 2.5e-3, 1.25e-4, 6.25e-6 (< 1e-4: stop)
    img = oracle.draw(pos, rgb, opa, cov, np.array([0, n], np.int32), 16, 16, 1e4, 1e4, fast=True)
    # G ~= 1 everywhere: T after k Gaussians = 0.05^k; stop when T < 1e-4 => exactly 4 contribute
    t = 1.0
    acc = 0.0
    for k in range(n):
        if t < 1e-4:
            break
        acc += 0.95 * t
        t *= 0.05
    assert k == 4 and abs(float(img[8, 8, 0]) - acc) < 1e-6


def test_weight_normalize_and_sigmoid_flags_follow_the_reference_formulas():
    cov = np.array([[4e-3, 1e-3, 1e-3, 6e-3]], np.float32)
    pos = np.array([[0.01, -0.02, 1.0]], np.float32)
    rgb = np.array([[0.3, 0.6, 0.9]], np.float32)
    opa = np.array([0.4], np.float32)
    accum = np.array([0, 1], np.int32)
    base = oracle.draw(pos, rgb, opa, cov, accum, 16, 16, 50.0, 50.0, fast=True)
    wn = oracle.draw(pos, rgb, opa, cov, accum, 16, 16, 50.0, 50.0, weight_normalize=True, fast=True)
    w = base[..., 0] / 0.3  # accumulated weight
    expect = np.where((w < 0.01)[..., None], base, base / np.maximum(w, 1e-30)[..., None])
    assert np.allclose(wn, expect, rtol=1e-5, atol=1e-7)
    sg = oracle.draw(pos, rgb, opa, cov, accum, 16, 16, 50.0, 50.0, sigmoid=True, fast=True)
    det = 4e-3 * 6e-3 - 1e-6
    p0 = 0.5 * np.pi / np.sqrt(det + 1e-7)  # gaussian.cu:918 (1.0/2*3.14.. == pi/2)
    a_raw = (base[..., 0] / 0.3 / 0.4) * p0 * 0.4
    expect_a = 2.0 / (np.exp(-a_raw) + 1) - 1
    assert np.allclose(sg[..., 0], 0.3 * expect_a, rtol=2e-4, atol=1e-6)


def test_cull_rules():
    """p_c.z <= near and |x/z| >= half_w / |y/z| >= half_h are culled; rows stay untouched."""
    pos = np.array([[0, 0, 0.3], [0, 0, 0.31], [1.0, 0, 1.0], [0.99, 0, 1.0], [0, -1.0, 1.0], [0, 0, -1]], np.float32)
    quat = np.tile(np.array([[1, 0, 0, 0]], np.float32), (6, 1))
    scale = np.full((6, 3), 0.01, np.float32)
    rp, rc, mk = oracle.global_culling(pos, quat, scale, np.eye(3, dtype=np.float32), np.zeros(3, np.float32), 0.3,
                                       1.0, 1.0)
    assert mk.tolist() == [0, 1, 0, 1, 0, 0]
    assert np.all(rp[mk == 0] == 0) and np.all(rc[mk == 0] == 0)
    assert np.isclose(rp[3, 2], np.sqrt(0.99 ** 2 + 1))  # depth is the Euclidean distance, not z


def test_cull_project_backward_matches_autograd_with_detached_jacobian():
    scene = make_scene(400, 96, 64, seed=3)
    cam = make_camera(96, 64, yaw_deg=4.0)
    cam.tran = np.array([0.1, 0.05, 0.3], np.float32)
    qn, sn = activate(scene)
    _, hw, hh, _ = frame_scalars(cam)
    rp, rc, mk = oracle.global_culling(scene.pos, qn, sn, cam.rot, cam.tran, cam.near, hw, hh)
    rng = np.random.default_rng(0)
    gop = rng.normal(size=(scene.n, 3)).astype(np.float32)
    goc = rng.normal(size=(scene.n, 2, 2)).astype(np.float32)
    got = oracle.global_culling_backward(scene.pos, qn, sn, cam.rot, cam.tran, gop, goc, mk)
    t = lambda a: torch.tensor(np.asarray(a, np.float64), requires_grad=True)
    p, q, s = t(scene.pos), t(qn), t(sn)
    pi, cv = torch_ref.project(p, q, s, torch.tensor(cam.rot, dtype=torch.float64),
                               torch.tensor(cam.tran, dtype=torch.float64))
    assert np.allclose(pi.detach().numpy()[mk == 1], rp[mk == 1], rtol=1e-5, atol=1e-6)
    m = torch.tensor(mk.astype(np.float64))
    loss = (pi * torch.tensor(gop, dtype=torch.float64) * m[:, None]).sum() + \
        (cv * torch.tensor(goc, dtype=torch.float64) * m[:, None, None]).sum()
    loss.backward()
    for g, ref in zip(got, (p.grad, q.grad, s.grad)):
        assert rel_err(g, ref.numpy()) < 2e-5, rel_err(g, ref.numpy())


@pytest.mark.parametrize("use_sh", [False])
def test_draw_backward_matches_autograd(use_sh):
    """Analytic rows of gaussian.cu:582-772 == autograd through A.7 (up to the +1e-7 in
    1/(1-alpha+1e-7)); low opacity keeps every pixel active (no early stop)."""
    scene = make_scene(250, 32, 32, seed=9)
    scene.opa -= 2.5
    cam = make_camera(32, 32)
    of = OracleFrame(scene, cam)
    grid = of.grid
    rng = np.random.default_rng(2)
    gpad = rng.normal(size=of.padded.shape).astype(np.float32)
    got = oracle.draw_backward(of.s_pos, of.s_rgb, of.s_opa, of.s_cov, of.accum, of.padded, gpad, grid.focal_x,
                               grid.focal_y, fast=True)
    t = lambda a: torch.tensor(np.asarray(a, np.float64), requires_grad=True)
    pos, rgb, opa, cov = t(of.s_pos), t(of.s_rgb), t(of.s_opa), t(of.s_cov)
    loss = 0.0
    for tile in range(grid.n_tiles):
        s, e = int(of.accum[tile]), int(of.accum[tile + 1])
        if e == s:
            continue
        tx, ty = tile % grid.n_tile_x, tile // grid.n_tile_x
        ix, iy = np.meshgrid(np.arange(16) + tx * 16, np.arange(16) + ty * 16)
        px = torch.tensor(((ix + 0.5 - grid.padded_width // 2) / grid.focal_x).reshape(-1))
        py = torch.tensor(((iy + 0.5 - grid.padded_height // 2) / grid.focal_y).reshape(-1))
        col = torch_ref.rasterize_tile(px, py, pos[s:e, 0], pos[s:e, 1], cov[s:e], opa[s:e], rgb[s:e])
        g = torch.tensor(gpad[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16].reshape(-1, 3).astype(np.float64))
        loss = loss + (col * g).sum()
    loss.backward()
    refs = (pos.grad.numpy(), rgb.grad.numpy(), opa.grad.numpy(), cov.grad.numpy())
    assert np.all(got[0][:, 2] == 0)
    for g, r, name in zip(got, refs, ("pos", "rgb", "opa", "cov")):
        assert rel_err(g, r) < 5e-5, (name, rel_err(g, r))


def test_frame_gradient_finite_differences():
    """End-to-end: d(sum(image * w))/d(opacity logit, rgb logit) of the oracle frame vs central
    differences of the oracle forward (parameters that do not move the tile lists)."""
    scene = make_scene(150, 32, 32, seed=4)
    scene.opa -= 1.0
    cam = make_camera(32, 32)
    of = OracleFrame(scene, cam)
    w = np.random.default_rng(1).normal(size=of.image.shape).astype(np.float32)
    g = of.backward(w)
    vis = np.nonzero(np.bincount(of.ids, minlength=scene.n))[0][:6]

    def f(sc):
        return float((OracleFrame(sc, cam).image.astype(np.float64) * w).sum())

    import copy
    for i in vis:
        for field, idx in (("opa", (i,)), ("rgb", (i, 1))):
            eps = 2e-2
            sp, sm = copy.deepcopy(scene), copy.deepcopy(scene)
            getattr(sp, field)[idx] += eps
            getattr(sm, field)[idx] -= eps
            fd = (f(sp) - f(sm)) / (2 * eps)
            an = float(g[field][idx])
            assert abs(fd - an) < 2e-3 * max(1.0, abs(an)) + 2e-3, (field, i, fd, an)


def test_float32_composite_key_loses_order_but_canonical_key_does_not():
    """ref_compat: splatter.py:610-612 sorts on depth + tile*(max_depth+1) in float32.  At
    1080p tile ids the key spacing exceeds small depth gaps, so near-equal depths collapse."""
    M = 64
    tile_ids = np.full(M, 8000, np.int32)
    depth = (5.0 + np.arange(M)[::-1] * 1e-4).astype(np.float32)  # strictly decreasing by 1e-4
    perm = oracle.sort_float32_key(depth, tile_ids)
    exact = np.argsort(depth, kind="stable")
    assert not np.array_equal(perm, exact)  # the float32 key cannot resolve 1e-4 at magnitude ~5e4
    keys = (np.uint64(8000) << np.uint64(32)) | depth.view(np.uint32).astype(np.uint64)
    assert np.array_equal(np.argsort(keys, kind="stable"), exact)


def test_sorted_pairs_edge_cases():
    g = TileGrid(64, 48, 48.0, 48.0)
    geom = (g.tile_geo_length_x, g.tile_geo_length_y, g.n_tile_x, g.n_tile_y, g.leftmost, g.topmost)
    # empty input
    k, i, a = oracle.sorted_pairs(np.zeros((0, 3), np.float32), np.zeros((0, 4), np.float32), None, 0.05, *geom)
    assert len(k) == 0 and np.all(a == 0)
    # det <= 0 is dropped; identical depths tie-break on the Gaussian index
    pos = np.array([[0, 0, 2.0], [0, 0, 2.0], [0, 0, 1.0], [0.1, 0.1, 3.0]], np.float32)
    cov = np.array([[1e-3, 0, 0, 1e-3]] * 3 + [[1e-3, 2e-3, 2e-3, 1e-3]], np.float32)  # last: det < 0
    k, i, a = oracle.sorted_pairs(pos, cov, None, 0.05, *geom)
    assert 3 not in i
    tiles = np.unique(k >> np.uint64(32))
    for t in tiles:
        seg = i[(k >> np.uint64(32)) == t]
        assert seg.tolist() == [2, 0, 1]  # depth 1.0 first, then the tie in index order
    # a huge Gaussian covers every tile exactly once
    k, i, a = oracle.sorted_pairs(np.array([[0, 0, 1.0]], np.float32), np.array([[100.0, 0, 0, 100.0]], np.float32),
                                  None, 0.05, *geom)
    assert len(k) == g.n_tiles and np.array_equal(np.diff(a), np.ones(g.n_tiles, np.int32))


# ------------------------------------------------------------------------------------------------
# SH degree 3 (extension: BASELINE config 4 names it, the reference stops at degree 2 -- gaussian.cu:405-426
# never reads its C3 table).  Pinned here by closed forms instead of reference outputs.
def test_sh_basis_is_the_real_spherical_harmonics_of_scipy():
    """All 16 functions (the reference's 9 and the 7 added ones) follow ONE rule: index l^2 + l + m holds
    sqrt(2) Im Y_l^|m| (m < 0), Y_l^0, sqrt(2) Re Y_l^m (m > 0) of scipy's complex harmonics -- so the degree-3
    band is the continuation of the reference's own convention, not a different one."""
    from scipy.special import sph_harm_y

    rng = np.random.default_rng(0)
    d = rng.normal(size=(200, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    polar, azim = np.arccos(d[:, 2]), np.arctan2(d[:, 1], d[:, 0])
    want = np.zeros((200, 16))
    for l in range(4):
        for m in range(-l, l + 1):
            y = sph_harm_y(l, abs(m), polar, azim)
            want[:, l * l + l + m] = y.real if m == 0 else np.sqrt(2) * (y.imag if m < 0 else y.real)
    got16 = oracle.calc_sh(16, d)
    assert np.abs(got16 - want).max() < 2e-6
    assert np.array_equal(oracle.calc_sh(9, d), got16[:, :9])


def _sh_scene(n, W, H, seed, degree):
    scene = make_scene(n, W, H, seed=seed, use_sh=True, sh_degree=degree)
    scene.opa -= 1.0
    return scene


def test_sh_degree3_with_zero_band3_is_degree2():
    """48 coefficients whose degree-3 band is zero render bit-identically to the 27-coefficient scene, and the
    gradients of the shared coefficients agree bit for bit."""
    import copy

    cam = make_camera(48, 32)
    s3 = _sh_scene(300, 48, 32, 5, 3)
    c = s3.rgb.reshape(-1, 3, 16)
    c[:, :, 9:] = 0
    s2 = copy.deepcopy(s3)
    s2.rgb = np.ascontiguousarray(c[:, :, :9]).reshape(-1, 27)
    o3, o2 = OracleFrame(s3, cam), OracleFrame(s2, cam)
    assert np.array_equal(o3.padded, o2.padded)
    w = np.random.default_rng(1).normal(size=o3.image.shape).astype(np.float32)
    g3, g2 = o3.backward(w), o2.backward(w)
    for k in ("pos", "quat", "scale", "opa"):
        assert np.array_equal(g3[k], g2[k]), k
    assert np.array_equal(g3["rgb"].reshape(-1, 3, 16)[:, :, :9], g2["rgb"].reshape(-1, 3, 9))
    assert np.abs(g3["rgb"].reshape(-1, 3, 16)[:, :, 9:]).max() > 0  # the band still receives a gradient


def test_sh_degree3_gradient_finite_differences():
    """Analytic SH-coefficient rows (all 16 per channel) vs central differences of the oracle forward."""
    import copy

    cam = make_camera(32, 32)
    scene = _sh_scene(120, 32, 32, 8, 3)
    of = OracleFrame(scene, cam)
    w = np.random.default_rng(2).normal(size=of.image.shape).astype(np.float32)
    g = of.backward(w)
    vis = np.nonzero(np.bincount(of.ids, minlength=scene.n))[0]
    # the most visible Gaussians: finite differences of an fp32 forward need a signal above its rounding noise
    strong = vis[np.argsort(-np.abs(g["rgb"][vis]).max(1))[:3]]

    def f(sc):
        return float((OracleFrame(sc, cam).image.astype(np.float64) * w).sum())

    eps = 2e-2
    for i in strong:
        for k in (0, 9, 12, 15, 16 + 10, 32 + 13, 47):
            hi, lo = copy.deepcopy(scene), copy.deepcopy(scene)
            hi.rgb[i, k] += eps
            lo.rgb[i, k] -= eps
            fd = (f(hi) - f(lo)) / (2 * eps)
            assert abs(fd - g["rgb"][i, k]) < 2e-2 * max(abs(fd), np.abs(g["rgb"][i]).max()) + 1e-5, (i, k, fd, g["rgb"][i, k])
