"""Shared helpers for the parity tests: oracle-side frame pipeline and comparisons."""
from __future__ import annotations

import numpy as np

import oracle
from gs_geometry import RayBasis, TileGrid
from gs_scene import Camera, Scene


def sigmoid32(x):
    x = np.asarray(x, np.float32)
    return (np.float32(1) / (np.float32(1) + np.exp(-x, dtype=np.float32))).astype(np.float32)


def activate(scene: Scene, scale_activation="abs"):
    """splatter.py:519-524 in fp32 (same expression order as the kernels)."""
    q = scene.quat.astype(np.float32)
    nr = np.sqrt(((q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1]) + q[:, 2] * q[:, 2]) + q[:, 3] * q[:, 3], dtype=np.float32)
    qn = (q / nr[:, None]).astype(np.float32)
    if scale_activation == "abs":
        sn = (np.abs(scene.scale) + np.float32(1e-4)).astype(np.float32)
    else:
        sn = np.exp(scene.scale, dtype=np.float32)
    return qn, sn


def frame_scalars(cam: Camera):
    grid = TileGrid(cam.width, cam.height, cam.focal_x, cam.focal_y)
    hw, hh = grid.frustum_half_extents()
    rays = RayBasis.from_camera(cam.rot, cam.tran, grid.padded_height, grid.padded_width, grid.focal_x, grid.focal_y)
    return grid, hw, hh, rays


class OracleFrame:
    """Oracle restatement of one forward frame on raw parameters, keeping every intermediate."""

    def __init__(self, scene: Scene, cam: Camera, thresh=0.05, scale_activation="abs", tile_culling_method="prob2",
                 dist_thresh=0.5, activated=None):
        """``activated`` = (normalised quaternions, activated scales) computed by the CALLER (the reference's torch ops,
        splatter.py:519-524, whose last bits differ from ``activate``'s expression order): the frame is then the oracle's
        on exactly those inputs -- what the reference-API tests need to compare bit for bit behind torch activations."""
        self.scene, self.cam = scene, cam
        self.dist_thresh = dist_thresh
        self.scale_activation = scale_activation
        grid, hw, hh, rays = frame_scalars(cam)
        self.grid, self.rays = grid, rays
        self.qn, self.sn = activate(scene, scale_activation) if activated is None else \
            tuple(np.ascontiguousarray(a, np.float32) for a in activated)
        self.pos_i, self.cov, self.mask = oracle.global_culling(scene.pos, self.qn, self.sn, cam.rot, cam.tran,
                                                                cam.near, hw, hh)
        if tile_culling_method == "prob2":
            self.keys, self.ids, self.accum = oracle.sorted_pairs(
                self.pos_i, self.cov.reshape(-1, 4), self.mask, thresh, grid.tile_geo_length_x,
                grid.tile_geo_length_y, grid.n_tile_x, grid.n_tile_y, grid.leftmost, grid.topmost)
        else:
            self.keys, self.ids, self.accum = self._pairs_from_table(tile_culling_method, thresh)
        self.opa_act = sigmoid32(scene.opa)
        self.col_act = scene.rgb if scene.use_sh else sigmoid32(scene.rgb)
        ids = self.ids
        self.s_pos, self.s_cov = self.pos_i[ids], self.cov.reshape(-1, 4)[ids]
        self.s_opa, self.s_rgb = self.opa_act[ids], self.col_act[ids]
        self.padded = oracle.draw(self.s_pos, self.s_rgb, self.s_opa, self.s_cov, self.accum, grid.padded_height,
                                  grid.padded_width, grid.focal_x, grid.focal_y, use_sh=scene.use_sh, fast=True,
                                  rays_o=rays.rays_o, lefttop=rays.lefttop, vdx=rays.dx, vdy=rays.dy)
        self.image = grid.crop(np.clip(self.padded, 0, 1))

    def robust_grad_image(self, grad_image, band=2e-5):
        """dL/dimage with the pixels zeroed whose early-stop decision is not robust in fp32 (their transmittance
        passes within its own fp32 uncertainty -- ``band`` plus what nearly opaque Gaussians in front add -- of the 1e-4
        threshold: oracle.draw_ambiguous) -- typically ~0.1 % of the pixels.
        Every gradient term is proportional to its pixel's dL/dimage, so with this input two evaluations that stop
        such a pixel one Gaussian apart still have the same set of terms to compare.  -> (grad_image, #pixels zeroed)"""
        grid = self.grid
        amb = oracle.draw_ambiguous(self.s_pos, self.s_opa, self.s_cov, self.accum, grid.padded_height,
                                    grid.padded_width, grid.focal_x, grid.focal_y, band)
        keep = ~grid.crop(amb[:, :, None])
        return np.ascontiguousarray(grad_image * keep, np.float32), int((~keep).sum())

    def _pairs_from_table(self, method, thresh):
        """'prob' / 'dist' (splatter.py:571-578): the oracle's restatement of calc_tile_list methods 1 / 0 (pinned
        bit for bit against the reference kernels) on the visible Gaussians, uncapped, then the canonical
        (tile, depth bits, Gaussian index) order."""
        grid = self.grid
        vis = np.nonzero(self.mask)[0]
        top, bottom, left, right = grid.tile_edges()
        th = (grid.tile_geo_length_x / self.dist_thresh) ** 2 if method == "dist" else thresh  # splatter.py:577
        cnt, lst = oracle.calc_tile_list(self.pos_i[vis], self.cov.reshape(-1, 4)[vis], len(vis), th,
                                         {"dist": 0, "prob": 1}[method], grid.tile_geo_length_x,
                                         grid.tile_geo_length_y, grid.n_tile_x, grid.n_tile_y, grid.leftmost,
                                         grid.topmost, top=top, bottom=bottom, left=left, right=right)
        tiles = np.repeat(np.arange(len(cnt)), cnt)
        ids = np.concatenate([vis[lst[t, :c]] for t, c in enumerate(cnt)]) if cnt.sum() else np.zeros(0, np.int64)
        dbits = self.pos_i[ids, 2].view(np.uint32).astype(np.uint64)
        keys = (tiles.astype(np.uint64) << np.uint64(32)) | dbits
        order = np.lexsort((ids, keys))
        accum = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
        return keys[order], ids[order].astype(np.int32), accum

    def backward(self, grad_image, with_scale=False, scale_w=0.05):
        """dL/d(image) -> dict of dL/d(raw parameter), the chain splatter.py's autograd runs.

        ``with_scale``: also return, per gradient element, its conditioning scale -- the oracle's per-row scale
        (gs_oracle.c, draw_backward_impl) summed over the Gaussian's (tile, Gaussian) rows and pushed through the
        magnitudes of the (linear) projection / activation backward.  ``grad_close`` states tolerances in it."""
        sc, cam, grid, rays = self.scene, self.cam, self.grid, self.rays
        n = sc.n
        top, left = grid.crop_offsets()
        gpad = np.zeros_like(self.padded)
        inside = ((self.padded >= 0) & (self.padded <= 1)).astype(np.float32)
        gpad[top:top + grid.height, left:left + grid.width] = grad_image
        gpad *= inside
        out = oracle.draw_backward(self.s_pos, self.s_rgb, self.s_opa, self.s_cov, self.accum, self.padded,
                                   gpad, grid.focal_x, grid.focal_y, use_sh=sc.use_sh, fast=True,
                                   rays_o=rays.rays_o, lefttop=rays.lefttop, vdx=rays.dx, vdy=rays.dy,
                                   with_scale=with_scale, scale_w=scale_w)
        (gp, gr, go, gc), cs = out if with_scale else (out, None)
        self.pair_grads, self.pair_scales = (gp, gr, go, gc), cs  # the (tile, Gaussian) rows, for the reference-API tests
        # index backward (index_put accumulate) in double
        d_pos_i, d_cov, d_opa, d_col = (_sum_by_id(self.ids, a, n) for a in (gp, gc, go, gr))
        grads = self._chain(d_pos_i, d_cov, d_opa, d_col)
        if not with_scale:
            return grads
        s_pos_i, s_cov, s_opa, s_col = (_sum_by_id(self.ids, a, n) for a in (cs[0], cs[3], cs[2], cs[1]))
        return grads, self._chain_scale(s_pos_i, s_cov, s_opa, s_col)

    def backward_f64(self, grad_image):
        """The same chain as ``backward`` evaluated in DOUBLE on the fp32 inputs: K8 by oracle.draw_backward_f64
        (its own final colour, exp, sums -- only the stop decisions are the fp32 chain's), the index backward as a
        double sum, the projection backward as torch.autograd (float64) on oracle/torch_ref.project -- A.4 with the
        Jacobian detached, i.e. gaussian.cu:1371-1576 derived independently --, the activations in double.
        The yardstick of tests/test_grad_calibration.py.  -> ((gp, gr, go, gc) rows, {name: parameter gradient})."""
        import torch

        from oracle import torch_ref

        sc, cam, grid, rays = self.scene, self.cam, self.grid, self.rays
        n = sc.n
        top, left = grid.crop_offsets()
        gpad = np.zeros_like(self.padded)
        inside = ((self.padded >= 0) & (self.padded <= 1)).astype(np.float32)
        gpad[top:top + grid.height, left:left + grid.width] = grad_image
        gpad *= inside
        rows = oracle.draw_backward_f64(self.s_pos, self.s_rgb, self.s_opa, self.s_cov, self.accum, gpad,
                                        grid.focal_x, grid.focal_y, use_sh=sc.use_sh, rays_o=rays.rays_o,
                                        lefttop=rays.lefttop, vdx=rays.dx, vdy=rays.dy)
        gp, gr, go, gc = rows
        d_pos_i, d_cov, d_opa, d_col = (_sum_by_id(self.ids, a, n) for a in (gp, gc, go, gr))
        vis = np.nonzero(self.mask)[0]
        f64 = lambda a: torch.from_numpy(np.asarray(a, np.float64))  # noqa: E731
        q = sc.quat.astype(np.float64)
        nr = np.linalg.norm(q, axis=1, keepdims=True)
        qh = q / nr
        # (the projection sees the fp32 activated values, like every fp32 evaluation: each stage's backward is taken
        # at the forward's stored fp32 intermediates)
        tp, tq, ts = (f64(a[vis]).requires_grad_(True) for a in (sc.pos, self.qn, self.sn))
        pos_i, cov = torch_ref.project(tp, tq, ts, f64(cam.rot), f64(cam.tran), detach_jacobian=True)
        obj = (pos_i * f64(d_pos_i[vis])).sum() + (cov.reshape(-1, 4) * f64(d_cov[vis])).sum()
        gpv, gqv, gsv = torch.autograd.grad(obj, (tp, tq, ts))
        g_pos, g_qn, g_sn = (np.zeros((n, k)) for k in (3, 4, 3))
        g_pos[vis], g_qn[vis], g_sn[vis] = gpv.numpy(), gqv.numpy(), gsv.numpy()
        g_q = (g_qn - qh * np.sum(qh * g_qn, axis=1, keepdims=True)) / nr
        g_s = g_sn * np.sign(sc.scale) if self.scale_activation == "abs" else g_sn * np.exp(np.clip(sc.scale, -1, 1))
        o = 1.0 / (1.0 + np.exp(-sc.opa.astype(np.float64)))
        g_o = d_opa * o * (1 - o)
        if sc.use_sh:
            g_c = d_col
        else:
            c = 1.0 / (1.0 + np.exp(-sc.rgb.astype(np.float64)))
            g_c = d_col * c * (1 - c)
        return rows, {"pos": g_pos, "quat": g_q, "scale": g_s, "opa": g_o, "rgb": g_c}

    def _chain(self, d_pos_i, d_cov, d_opa, d_col):
        sc, cam = self.scene, self.cam
        g_pos, g_qn, g_sn = oracle.global_culling_backward(sc.pos, self.qn, self.sn, cam.rot, cam.tran,
                                                           d_pos_i.astype(np.float32), d_cov.astype(np.float32),
                                                           self.mask)
        q = sc.quat.astype(np.float64)
        nr = np.linalg.norm(q, axis=1, keepdims=True)
        qh = q / nr
        g_q = (g_qn - qh * np.sum(qh * g_qn, axis=1, keepdims=True)) / nr
        if self.scale_activation == "abs":
            g_s = g_sn * np.sign(sc.scale)
        else:
            g_s = g_sn * np.exp(np.clip(sc.scale, -1, 1))
        o = self.opa_act.astype(np.float64)
        g_o = d_opa * o * (1 - o)
        if sc.use_sh:
            g_c = d_col
        else:
            c = self.col_act.astype(np.float64)
            g_c = d_col * c * (1 - c)
        return {"pos": g_pos.astype(np.float32), "quat": g_q.astype(np.float32), "scale": g_s.astype(np.float32),
                "opa": g_o.astype(np.float32), "rgb": g_c.astype(np.float32)}

    def _chain_scale(self, s_pos_i, s_cov, s_opa, s_col):
        """The scales pushed through K2 and the activations with every product taken between magnitudes and every
        difference replaced by the sum of its operands' magnitudes (oracle.global_culling_backward_scale)."""
        sc, cam = self.scene, self.cam
        S_pos, S_qn, S_sn = oracle.global_culling_backward_scale(sc.pos, self.qn, self.sn, cam.rot, cam.tran,
                                                                 s_pos_i, s_cov.reshape(-1, 4), self.mask)
        q = sc.quat.astype(np.float64)
        nr = np.linalg.norm(q, axis=1, keepdims=True)
        qh = np.abs(q / nr)
        S_q = (S_qn + qh * np.sum(qh * S_qn, axis=1, keepdims=True)) / nr
        S_s = S_sn if self.scale_activation == "abs" else S_sn * np.exp(np.clip(sc.scale, -1, 1))
        o = self.opa_act.astype(np.float64)
        S_o = s_opa * o * (1 - o)
        if sc.use_sh:
            S_c = s_col
        else:
            c = self.col_act.astype(np.float64)
            S_c = s_col * c * (1 - c)
        return {"pos": S_pos, "quat": S_q, "scale": S_s, "opa": S_o, "rgb": S_c}


def _sum_by_id(ids, rows, n):
    """index_put_(accumulate=True) (splatter.py:604-613 backward) in float64: rows[j] added to out[ids[j]]."""
    rows = np.asarray(rows, np.float64)
    flat = rows.reshape(len(ids), -1)
    out = np.zeros((n, flat.shape[1]), np.float64)
    if len(ids):
        order = np.argsort(ids, kind="stable")
        sid = np.asarray(ids)[order]
        first = np.flatnonzero(np.r_[True, sid[1:] != sid[:-1]])
        out[sid[first]] = np.add.reduceat(flat[order], first, axis=0)
    return out.reshape((n,) + rows.shape[1:])


# Element-wise gradient tolerance: |got - ref| <= GRAD_RTOL |ref| + GRAD_KAPPA scale, `scale` being the element's own
# conditioning scale from OracleFrame.backward(with_scale=True): the sum over its pixels and (tile, Gaussian) rows of
# |term| (1 + 0.25 x the cancellation inside exp's argument) + 0.05 x (the term with every other internal difference
# replaced by the magnitudes of its operands), pushed through the projection / activation backward in the same way.  A gradient element is a signed sum of thousands of
# fp32 terms, several of them differences of nearly equal numbers (T g.c against g.(C_final - C_run)/(1 - alpha) for
# a Gaussian deep in a tile's list): two correct fp32 evaluations in different orders agree to a number of ulp of
# that scale, however small the sum comes out -- so the tolerance is per element and NOT a fraction of the tensor's
# largest entry.  GRAD_KAPPA = 3e-5 is ~500 ulp of the plain term sum and ~25 ulp of the operand magnitudes (measured
# on the GPU, profiles/r02_a: at 1e-5 every element of cfg2 / cfg3 and all but ONE of the 65 M SH coefficients of cfg4
# pass -- that one, 1.4e-14 against a tensor maximum of ~1e-4 and a scale of 1e-12, sat at 2.05 x); what
# differs between the kernels and the oracle is the summation order, v_exp_f32 / v_rcp_f32 against expf / IEEE
# division, the conic hoisted out of the pixel loop, and the final image each side subtracts its running colour from
# (they agree to ~1e-6).  tools/grad_parity_probe.py prints the measured distribution of err / scale.
GRAD_RTOL = 1e-4
GRAD_KAPPA = 3e-5
GRAD_L2 = 2e-5  # ||got - ref||_2 / ||ref||_2 per tensor
# Independent of the oracle-supplied scale, and CALIBRATED rather than fitted: the PURE
# relative error |got - truth| / |truth| against a DOUBLE-precision evaluation of the same formulas
# (OracleFrame.backward_f64), over the elements above 1e-6 of the tensor's largest entry, compared quantile by quantile
# with the error the REFERENCE's own fp32 arithmetic has against that same truth:
#   * at the size where the reference's kernels are well defined, their outputs are committed
#     (tests/golden/calib_*.npz) and the HIP kernels' error quantiles are bounded by CALIB_K x theirs
#     (tests/test_grad_calibration.py) -- where the same file shows that the ORACLE's error distribution is the
#     reference kernels' (same fp32 terms; quantiles within a factor 2, measured within 20 %);
#   * at full size (tests/test_gpu_frame.py) the reference kernel is not defined (its chunk defect), and the oracle
#     -- fp32 terms in the reference's expression order, accumulated in double: never WORSE than the reference's
#     fp32 shuffles / atomics -- stands in for it: assert_error_no_worse_than.
# What the reference's own arithmetic achieves (calib fixtures, lists <= 166): median ~1e-6, 99.9th percentile 1e-4 ...
# 5e-4, maximum up to 1.8e-2 -- percent-level single elements are a property of fp32 on these sums, not of a kernel.
CALIB_QS = (0.5, 0.9, 0.99, 0.999, 1.0)
CALIB_K = 4.0  # error quantile (median ... 99.9 %) of a HIP kernel <= CALIB_K x the same quantile of the reference arithmetic's error
CALIB_K_MAX = 8.0  # the MAXIMUM of the relative error: a one-element statistic of an element that is the small difference of
#                    large terms -- which element that is, and how lucky the reference was on it, differs between evaluations
#                    (measured over the golden-size cases, round 6: quantiles up to 99.9 % within 3.6 x, maxima within 5.9 x)
CALIB_K_ELEM = 6.0  # per element: e_hip <= CALIB_K_ELEM x max(e_ref, the reference's 99.9th-percentile error on sums of that magnitude)
#                      (measured: worst element 4.7 x, 99.9th percentile 0.8 - 1.1 x)


def rel_error_quantiles(got, truth, qs=CALIB_QS, floor=1e-6):
    """Quantiles of |got - truth| / |truth| over the elements with |truth| > floor x max |truth|."""
    got, truth = np.asarray(got, np.float64), np.asarray(truth, np.float64)
    a = np.abs(truth)
    big = a > floor * (a.max() if a.size else 0.0)
    if not big.any():
        return [0.0] * len(qs)
    r = np.abs(got - truth)[big] / a[big]
    return [float(np.quantile(r, q)) for q in qs]


def elementwise_error_ratio(got, ref, truth, scale):
    """Element by element: e_got / max(e_ref, F x scale) -> (99.9th percentile, maximum, F, the same F for `got`).
    `scale` is the element's term-magnitude sum (the oracle's conditioning scale with scale_w = 0) and F the REFERENCE
    evaluation's own tail error per unit of scale, the 99.9th percentile of e_ref / scale over the tensor: where the
    reference happened to round luckily, an element is held to what the reference's arithmetic delivers at its 99.9th
    percentile on sums of that magnitude -- not to that luck.  (Errors of two independent fp32 evaluations are not
    correlated element by element: against max(e_ref, one ulp of scale) the ORACLE passes with 1.5, because it shares
    the reference's per-term arithmetic, while any kernel with another exp / rcp / summation order sits at 5 - 25.)"""
    got, ref, truth, scale = (np.asarray(a, np.float64).ravel() for a in (got, ref, truth, scale))
    e_got, e_ref = np.abs(got - truth), np.abs(ref - truth)
    pos = scale > 0
    F = float(np.quantile(e_ref[pos] / scale[pos], 0.999)) if pos.any() else 0.0
    F_got = float(np.quantile(e_got[pos] / scale[pos], 0.999)) if pos.any() else 0.0
    ratio = e_got / np.maximum(np.maximum(e_ref, F * scale), 1e-300)
    return float(np.quantile(ratio, 0.999)), float(ratio.max()), F, F_got


def assert_error_no_worse_than(grads, truth, ref, what="", k=CALIB_K):
    """Five parameter gradients `grads` and the reference arithmetic's `ref` (the oracle's, fp32 terms) against the
    double-precision `truth`: every quantile of CALIB_QS of the HIP error is within k x the reference arithmetic's.
    -> {name: (hip quantiles, reference quantiles)}"""
    report = {}
    for g, name in zip(grads, ("pos", "quat", "scale", "opa", "rgb")):
        qh = rel_error_quantiles(np.asarray(g), truth[name])
        qr = rel_error_quantiles(ref[name], truth[name])
        report[name] = (qh, qr)
        for a, b, q in zip(qh, qr, CALIB_QS):
            assert a <= (CALIB_K_MAX if q == 1.0 else k) * b, (what, name, "quantile", q, "hip error", a,
                                                               "reference arithmetic's error", b)
    return report


def assert_rows_error_no_worse_than(got, truth, ref, what="", k=CALIB_K):
    """The same statement for the four (tile, Gaussian)-row gradients of draw_backward (pos x / y, rgb, opa, cov)."""
    report = {}
    for g, t, r, name in zip(got, truth, ref, ROW_NAMES):
        cut = (lambda a: np.asarray(a).reshape(np.asarray(t).shape)[:, :2]) if name == "pos" else \
            (lambda a: np.asarray(a).reshape(np.asarray(t).shape))
        qh, qr = rel_error_quantiles(cut(g), cut(t)), rel_error_quantiles(cut(r), cut(t))
        report[name] = (qh, qr)
        for a, b, q in zip(qh, qr, CALIB_QS):
            assert a <= (CALIB_K_MAX if q == 1.0 else k) * b, (what, "rows", name, "quantile", q, "hip error", a,
                                                               "reference arithmetic's error", b)
    return report


# Entries 14 orders of magnitude below the tensor's largest are compared up to this floor: they are sums of products
# whose intermediates underflow in fp32 (the transcendental unit flushes subnormal operands, the CPU oracle keeps
# them) or whose dL/dimage factor is itself a rounding residue of image - target.  Measured at cfg4 (65 M SH
# coefficients): with the floor at 1e-16 exactly one element (ref 2.0e-18 against a tensor maximum of ~1e-4) sat at
# 2.2 x its tolerance; nothing an optimiser sees: torch.optim.Adam (train.py:56) adds eps = 1e-8 to sqrt(v), so a 1e-18 gradient moves nothing.
GRAD_FLOOR_REL = 1e-14
GRAD_FLOOR = 1e-30


def grad_close(got, ref, scale, rtol=GRAD_RTOL, kappa=GRAD_KAPPA):
    """-> (ok, worst ratio err / tol, index of the worst element, fraction of elements within rtol |ref| alone)."""
    got, ref, scale = (np.asarray(a, np.float64) for a in (got, ref, scale))
    err = np.abs(got - ref)
    tol = rtol * np.abs(ref) + kappa * scale + GRAD_FLOOR + GRAD_FLOOR_REL * (np.abs(ref).max() if ref.size else 0.0)
    bad = err > tol
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.where(err > 0, err / tol, 0.0)
    worst = int(np.argmax(ratio)) if ratio.size else 0
    pure = float(np.mean(err <= rtol * np.abs(ref))) if err.size else 1.0
    return (not bool(bad.any()) and bool(np.isfinite(got).all())), float(ratio.flat[worst]) if ratio.size else 0.0, \
        np.unravel_index(worst, ratio.shape) if ratio.size else (), pure


def assert_grads_close(grads, ref, scale, what="", rtol=GRAD_RTOL, kappa=GRAD_KAPPA, l2=GRAD_L2):
    """grads: five arrays (pos, quat, scale, opa, rgb) -> element-wise check of each against the oracle + a relative
    L2 bound per tensor + bounds on the pure relative error.  Returns {name: (worst err / tol, fraction of elements inside
    rtol |ref| alone, rel. L2, worst pure relative error over the elements above 1e-6 of the tensor's maximum, its 99.9th
    percentile)}."""
    report = {}
    for g, name in zip(grads, ("pos", "quat", "scale", "opa", "rgb")):
        g = np.asarray(g)
        ok, worst, where, pure = grad_close(g, ref[name], scale[name], rtol, kappa)
        rl2 = float(np.linalg.norm(g.astype(np.float64) - ref[name]) / (np.linalg.norm(ref[name].astype(np.float64)) + 1e-300))
        # worst PURE relative error |got - ref| / |ref| over the elements above 1e-6 of the tensor's largest (reported;
        # the asserted oracle-independent statement is assert_error_no_worse_than, against a double-precision truth)
        r64 = np.abs(np.asarray(ref[name], np.float64))
        big = r64 > 1e-6 * (r64.max() if r64.size else 0.0)
        rels = np.abs(g.astype(np.float64) - ref[name])[big] / r64[big] if big.any() else np.zeros(1)
        rel_big = float(rels.max())
        rel_p999 = float(np.quantile(rels, 0.999))
        report[name] = (round(worst, 3), round(pure, 5), rl2, rel_big, rel_p999)
        assert ok, (what, name, "worst err/tol", worst, "at", where, "got", float(g[where]), "ref",
                    float(ref[name][where]), "scale", float(scale[name][where]))
        assert rl2 <= l2, (what, name, "relative L2 error", rl2)
    return report


ROW_NAMES = ("pos", "rgb", "opa", "cov")


def assert_rows_close(got, ref, scale, what="", rtol=GRAD_RTOL, kappa=GRAD_KAPPA, l2=GRAD_L2):
    """The four (tile, Gaussian)-row gradients of draw_backward (pos, rgb, opa, cov -- renderer.py:46-58) against the
    oracle's, ELEMENT BY ELEMENT in units of each element's conditioning scale (oracle.draw_backward(with_scale=True)),
    plus a relative L2 bound per tensor: the standard of the frame path's parameter gradients (assert_grads_close)
    applied at the reference API's own boundary.  -> {name: (worst err / tol, fraction within rtol |ref| alone, rel. L2)}"""
    report = {}
    for g, r, sc, name in zip(got, ref, scale, ROW_NAMES):
        g = np.asarray(g).reshape(np.asarray(r).shape)
        ok, worst, where, pure = grad_close(g, r, sc, rtol, kappa)
        rl2 = float(np.linalg.norm(g.astype(np.float64) - r) / (np.linalg.norm(np.asarray(r, np.float64)) + 1e-300))
        report[name] = (round(worst, 3), round(pure, 5), rl2)
        assert ok, (what, name, "worst err/tol", worst, "at", where, "got", float(g[where]), "ref", float(r[where]),
                    "scale", float(sc[where]))
        assert rl2 <= l2, (what, name, "relative L2 error", rl2)
    return report


def robust_padded_grad(of, gpad, band=2e-5):
    """dL/d(padded image) zeroed on the pixels whose early-stop decision is not robust in fp32 (OracleFrame.robust_grad_image
    for the padded image the reference API's draw works on).  -> (gpad, #pixels zeroed)"""
    grid = of.grid
    amb = oracle.draw_ambiguous(of.s_pos, of.s_opa, of.s_cov, of.accum, grid.padded_height, grid.padded_width,
                                grid.focal_x, grid.focal_y, band)
    keep = ~amb[:, :, None].astype(bool)
    return np.ascontiguousarray(gpad * keep, np.float32), int(amb.sum())


def rel_err(a, b):
    """max |a-b| / (max |b| + tiny): scale-aware error for gradient tensors."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def to_torch(scene: Scene, device, requires_grad=False):
    import torch

    out = []
    for a in (scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(device)
        t.requires_grad_(requires_grad)
        out.append(t)
    return out
