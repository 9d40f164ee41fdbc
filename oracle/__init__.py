"""CPU oracle for the rasterizer hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker.  The product (``3d-gaussian-splatting_amd/``) never
does; it fails loudly when its HIP library is missing.

``oracle/gs_oracle.c`` is the restatement (each function cites the reference file:line it
follows); this module is a thin numpy/ctypes binding plus the build recipe.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "gs_oracle.c")
_SO = os.path.join(_HERE, "libgs_oracle.so")

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    """gcc -O2 -ffp-contract=off: IEEE fp32 in source order (see gs_oracle.c header).  -fopenmp: the pixel loop of
    K7 and the tile loop of K8 are independent work items; results do not depend on the thread count."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(
            ["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", "-o", _SO, _SRC, "-lm"]
        )
    return _SO


def num_threads() -> int:
    """Threads the OpenMP loops of the oracle will use (OMP_NUM_THREADS or every core)."""
    env = os.environ.get("OMP_NUM_THREADS", "")
    return int(env) if env.isdigit() and int(env) > 0 else (os.cpu_count() or 1)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.gso_sorted_pairs.restype = C.c_int64
        _lib.gso_render_forward.restype = C.c_int64
        _lib.gso_tile_rect.restype = C.c_int
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


_Z3 = np.zeros(3, np.float32)


# ------------------------------------------------------------------------------------------
def world2camera(pos, rot, tran):
    pos, rot, tran = _f(pos), _f(rot), _f(tran)
    res = np.zeros_like(pos)
    lib().gso_world2camera(_vp(pos), _vp(rot), _vp(tran), _vp(res), C.c_int64(pos.shape[0]))
    return res


def world2camera_backward(grad_out, rot):
    grad_out, rot = _f(grad_out), _f(rot)
    res = np.zeros_like(grad_out)
    lib().gso_world2camera_backward(_vp(grad_out), _vp(rot), _vp(res), C.c_int64(grad_out.shape[0]))
    return res


def jacobian(pos_cam):
    pos_cam = _f(pos_cam)
    jac = np.zeros((pos_cam.shape[0], 3, 3), np.float32)
    lib().gso_jacobian(_vp(pos_cam), _vp(jac), C.c_int64(pos_cam.shape[0]))
    return jac


def global_culling(pos, quat, scale, rot, tran, near, half_w, half_h):
    """K1 (gaussian.cu:1182-1336).  quat pre-normalised, scale pre-activated."""
    pos, quat, scale, rot, tran = _f(pos), _f(quat), _f(scale), _f(rot), _f(tran)
    n = pos.shape[0]
    res_pos = np.zeros((n, 3), np.float32)
    res_cov = np.zeros((n, 2, 2), np.float32)
    mask = np.zeros(n, np.int64)
    lib().gso_global_culling(_vp(pos), _vp(quat), _vp(scale), _vp(rot), _vp(tran), C.c_int64(n),
                             C.c_float(near), C.c_float(half_w), C.c_float(half_h),
                             _vp(res_pos), _vp(res_cov), _vp(mask))
    return res_pos, res_cov, mask


def global_culling_backward(pos, quat, scale, rot, tran, gradout_pos, gradout_cov, mask):
    """K2 (gaussian.cu:1371-1576)."""
    pos, quat, scale, rot, tran = _f(pos), _f(quat), _f(scale), _f(rot), _f(tran)
    gradout_pos, gradout_cov = _f(gradout_pos), _f(gradout_cov)
    mask = np.ascontiguousarray(mask, np.int64)
    n = pos.shape[0]
    gp = np.zeros((n, 3), np.float32)
    gq = np.zeros((n, 4), np.float32)
    gs = np.zeros((n, 3), np.float32)
    lib().gso_global_culling_backward(_vp(pos), _vp(quat), _vp(scale), _vp(rot), _vp(tran),
                                      C.c_int64(n), _vp(gradout_pos), _vp(gradout_cov), _vp(mask),
                                      _vp(gp), _vp(gq), _vp(gs))
    return gp, gq, gs


def global_culling_backward_scale(pos, quat, scale, rot, tran, s_pos, s_cov, mask):
    """Conditioning scales of K2's outputs from those of its inputs (float64 in / out; gs_oracle.c)."""
    pos, quat, scale, rot, tran = _f(pos), _f(quat), _f(scale), _f(rot), _f(tran)
    s_pos = np.ascontiguousarray(s_pos, np.float64)
    s_cov = np.ascontiguousarray(s_cov, np.float64)
    mask = np.ascontiguousarray(mask, np.int64)
    n = pos.shape[0]
    op, oq, os_ = np.zeros((n, 3)), np.zeros((n, 4)), np.zeros((n, 3))
    lib().gso_global_culling_backward_scale(_vp(pos), _vp(quat), _vp(scale), _vp(rot), _vp(tran), C.c_int64(n),
                                            _vp(s_pos), _vp(s_cov), _vp(mask), _vp(op), _vp(oq), _vp(os_))
    return op, oq, os_


def tile_rect(cx, cy, a, b, c, d, thresh, tlx, tly, ntx, nty, leftmost, topmost):
    rect = np.zeros(4, np.uint32)
    ok = lib().gso_tile_rect(C.c_float(cx), C.c_float(cy), C.c_float(a), C.c_float(b), C.c_float(c),
                             C.c_float(d), C.c_float(thresh), C.c_float(tlx), C.c_float(tly),
                             C.c_uint32(ntx), C.c_uint32(nty), C.c_float(leftmost),
                             C.c_float(topmost), _vp(rect))
    return (int(ok), rect)


def calc_tile_list(pos, cov, maxp, thresh, method, tlx, tly, ntx, nty, leftmost, topmost,
                   top=None, bottom=None, left=None, right=None):
    """K3/K4/K5 (gaussian.cu:101-335), serial (race-free) cap semantics."""
    pos, cov = _f(pos), _f(cov)
    T = ntx * nty
    tile_n_point = np.zeros(T, np.int32)
    lst = np.full((T, max(int(maxp), 0)), -1, np.int32)
    tb = [None if v is None else _f(v) for v in (top, bottom, left, right)]
    lib().gso_calc_tile_list(_vp(pos), _vp(cov), C.c_int64(pos.shape[0]), _vp(tb[0]), _vp(tb[1]),
                             _vp(tb[2]), _vp(tb[3]), _vp(tile_n_point), _vp(lst),
                             C.c_int64(maxp), C.c_float(thresh), C.c_int(method), C.c_float(tlx),
                             C.c_float(tly), C.c_int32(ntx), C.c_int32(nty), C.c_float(leftmost),
                             C.c_float(topmost))
    return tile_n_point, lst


def gather_gaussians(accum, lst):
    """K6 (gaussian.cu:337-381)."""
    accum = np.ascontiguousarray(accum, np.int32)
    lst = np.ascontiguousarray(lst, np.int32)
    M = int(accum[-1])
    gathered = np.zeros(M, np.int32)
    tile_ids = np.zeros(M, np.int32)
    lib().gso_gather_gaussians(_vp(accum), _vp(lst), _vp(gathered), _vp(tile_ids),
                               C.c_int64(accum.shape[0] - 1), C.c_int64(lst.shape[1]))
    return gathered, tile_ids


def sorted_pairs(pos_i, cov, mask, thresh, tlx, tly, ntx, nty, leftmost, topmost):
    """Canonical (tile, depth-bits, gaussian-index) order.  Returns keys[M] u64, ids[M] i32,
    accum[T+1] i32.  ``mask`` may be None (all rows are candidates)."""
    pos_i, cov = _f(pos_i), _f(cov)
    n = pos_i.shape[0]
    m = None if mask is None else np.ascontiguousarray(mask, np.int64)
    args = (_vp(pos_i), _vp(cov), _vp(m), C.c_int64(n), C.c_float(thresh), C.c_float(tlx),
            C.c_float(tly), C.c_int32(ntx), C.c_int32(nty), C.c_float(leftmost), C.c_float(topmost))
    M = lib().gso_sorted_pairs(*args, C.c_int64(0), None, None, None)
    keys = np.zeros(max(M, 1), np.uint64)
    ids = np.zeros(max(M, 1), np.int32)
    accum = np.zeros(ntx * nty + 1, np.int32)
    lib().gso_sorted_pairs(*args, C.c_int64(M), _vp(keys), _vp(ids), _vp(accum))
    return keys[:M], ids[:M], accum


def sort_float32_key(depth, tile_ids):
    """ref_compat ordering: splatter.py:608-613 float32 composite key, stable."""
    depth = _f(depth)
    tile_ids = np.ascontiguousarray(tile_ids, np.int32)
    perm = np.zeros(depth.shape[0], np.int64)
    lib().gso_sort_float32_key(_vp(depth), _vp(tile_ids), C.c_int64(depth.shape[0]), _vp(perm))
    return perm


def _sh_code(use_sh, rgb) -> int:
    """gs_oracle.c's `use_sh`: 0 = rgb logits, 9 = the reference's degree-2 basis, 16 = the degree-3 extension
    (recognised by the coefficient count: [.., 27] or [.., 48])."""
    if not use_sh:
        return 0
    d = int(np.asarray(rgb).shape[-1])
    if d not in (27, 48):
        raise ValueError(f"SH colour needs 27 or 48 coefficients per Gaussian, got {d}")
    return d // 3


def calc_sh(nb, dirs):
    """SH basis of gaussian.cu:405-426 for unit directions [K,3] -> [K,nb]; nb = 9 (reference) or 16 (degree-3
    extension, see gs_oracle.c)."""
    dirs = _f(dirs).reshape(-1, 3)
    out = np.zeros((dirs.shape[0], nb), np.float32)
    for k in range(dirs.shape[0]):
        lib().gso_calc_sh(C.c_int(nb), _vp(dirs[k]), _vp(out[k]))
    return out


def draw(pos, rgb, opa, cov, accum, padded_h, padded_w, focal_x, focal_y, weight_normalize=False,
         sigmoid=False, use_sh=False, fast=False, rays_o=None, lefttop=None, vdx=None, vdy=None):
    """K7 (gaussian.cu:806-970)."""
    pos, rgb, opa, cov = _f(pos), _f(rgb), _f(opa), _f(cov)
    accum = np.ascontiguousarray(accum, np.int32)
    res = np.zeros((padded_h, padded_w, 3), np.float32)
    rv = [_f(v) if v is not None else _Z3 for v in (rays_o, lefttop, vdx, vdy)]
    lib().gso_draw(_vp(pos), _vp(rgb), _vp(opa), _vp(cov), _vp(accum), _vp(res), C.c_int32(padded_h),
                   C.c_int32(padded_w), C.c_float(focal_x), C.c_float(focal_y),
                   C.c_int(bool(weight_normalize)), C.c_int(bool(sigmoid)), C.c_int(bool(fast)),
                   _vp(rv[0]), _vp(rv[1]), _vp(rv[2]), _vp(rv[3]), C.c_int(_sh_code(use_sh, rgb)))
    return res


def draw_ambiguous(pos, opa, cov, accum, padded_h, padded_w, focal_x, focal_y, band=2e-5):
    """[padded_h, padded_w] bool: pixels whose transmittance passes within ``band`` (relative) of the 1e-4 stop
    threshold of K7 / K8 -- where stopping one Gaussian earlier or later is a legitimate fp32 outcome (gs_oracle.c)."""
    pos, opa, cov = _f(pos), _f(opa), _f(cov)
    accum = np.ascontiguousarray(accum, np.int32)
    amb = np.zeros((padded_h, padded_w), np.uint8)
    lib().gso_draw_ambiguous(_vp(pos), _vp(opa), _vp(cov), _vp(accum), _vp(amb), C.c_int32(padded_h),
                             C.c_int32(padded_w), C.c_float(focal_x), C.c_float(focal_y), C.c_float(band))
    return amb.astype(bool)


def draw_backward(pos, rgb, opa, cov, accum, output, grad_output, focal_x, focal_y,
                  weight_normalize=False, sigmoid=False, use_sh=False, fast=False, rays_o=None,
                  lefttop=None, vdx=None, vdy=None, with_scale=False, scale_w=0.05, scale_w_exp=0.25):
    """K8 (gaussian.cu:440-803), intended semantics (see gs_oracle.c).  ``with_scale``: also return the
    conditioning scale of every output element (sum over pixels of |term| + scale_w x the term with every internal
    difference replaced by the magnitudes of its operands; gs_oracle.c), laid out like the four gradients."""
    pos, rgb, opa, cov = _f(pos), _f(rgb), _f(opa), _f(cov)
    output, grad_output = _f(output), _f(grad_output)
    accum = np.ascontiguousarray(accum, np.int32)
    h, w = output.shape[0], output.shape[1]
    gp, gr, go, gc = (np.zeros_like(pos), np.zeros_like(rgb), np.zeros_like(opa), np.zeros_like(cov))
    rv = [_f(v) if v is not None else _Z3 for v in (rays_o, lefttop, vdx, vdy)]
    if with_scale:
        cp, cr, co, cc = (np.zeros_like(pos), np.zeros_like(rgb), np.zeros_like(opa), np.zeros_like(cov))
        lib().gso_draw_backward_scaled(_vp(pos), _vp(rgb), _vp(opa), _vp(cov), _vp(accum), _vp(output),
                                       _vp(grad_output), _vp(gp), _vp(gr), _vp(go), _vp(gc), C.c_int32(h),
                                       C.c_int32(w), C.c_float(focal_x), C.c_float(focal_y), C.c_int(bool(sigmoid)),
                                       C.c_int(bool(fast)), _vp(rv[0]), _vp(rv[1]), _vp(rv[2]), _vp(rv[3]),
                                       C.c_int(_sh_code(use_sh, rgb)), _vp(cp), _vp(cr), _vp(co), _vp(cc),
                                       C.c_double(scale_w), C.c_double(scale_w_exp))
        return (gp, gr, go, gc), (cp, cr, co, cc)
    lib().gso_draw_backward(_vp(pos), _vp(rgb), _vp(opa), _vp(cov), _vp(accum), _vp(output),
                            _vp(grad_output), _vp(gp), _vp(gr), _vp(go), _vp(gc), C.c_int32(h),
                            C.c_int32(w), C.c_float(focal_x), C.c_float(focal_y),
                            C.c_int(bool(weight_normalize)), C.c_int(bool(sigmoid)),
                            C.c_int(bool(fast)), _vp(rv[0]), _vp(rv[1]), _vp(rv[2]), _vp(rv[3]),
                            C.c_int(_sh_code(use_sh, rgb)))
    return gp, gr, go, gc


def draw_backward_f64(pos, rgb, opa, cov, accum, grad_output, focal_x, focal_y, use_sh=False, rays_o=None,
                      lefttop=None, vdx=None, vdy=None):
    """K8 evaluated in double on the fp32 inputs (gs_oracle.c, gso_draw_backward_f64): the yardstick the fp32
    evaluations -- the reference's kernel, the oracle, the HIP kernels -- are measured against.  `fast` flavour,
    sigmoid off; the stop decisions are those of the fp32 transmittance chain.  -> four float64 arrays."""
    pos, rgb, opa, cov, grad_output = _f(pos), _f(rgb), _f(opa), _f(cov), _f(grad_output)
    accum = np.ascontiguousarray(accum, np.int32)
    h, w = grad_output.shape[0], grad_output.shape[1]
    gp, gr, go, gc = (np.zeros(a.shape, np.float64) for a in (pos, rgb, opa, cov))
    rv = [_f(v) if v is not None else _Z3 for v in (rays_o, lefttop, vdx, vdy)]
    lib().gso_draw_backward_f64(_vp(pos), _vp(rgb), _vp(opa), _vp(cov), _vp(accum), _vp(grad_output), _vp(gp),
                                _vp(gr), _vp(go), _vp(gc), C.c_int32(h), C.c_int32(w), C.c_float(focal_x),
                                C.c_float(focal_y), _vp(rv[0]), _vp(rv[1]), _vp(rv[2]), _vp(rv[3]),
                                C.c_int(_sh_code(use_sh, rgb)))
    return gp, gr, go, gc


def render_forward(pos, quat_raw, scale_raw, opa_raw, rgb_raw, rot, tran, near, W, H, fx, fy,
                   thresh, use_sh=False, rays_o=None, lefttop=None, vdx=None, vdy=None):
    """Whole forward frame (splatter.py:513-655, train.py defaults).  -> image[H,W,3], V, M."""
    pos, quat_raw, scale_raw, opa_raw, rgb_raw = map(_f, (pos, quat_raw, scale_raw, opa_raw, rgb_raw))
    rot, tran = _f(rot), _f(tran)
    img = np.zeros((H, W, 3), np.float32)
    V = C.c_int64(0)
    rv = [_f(v) if v is not None else _Z3 for v in (rays_o, lefttop, vdx, vdy)]
    M = lib().gso_render_forward(_vp(pos), _vp(quat_raw), _vp(scale_raw), _vp(opa_raw), _vp(rgb_raw),
                                 C.c_int64(pos.shape[0]), C.c_int(_sh_code(use_sh, rgb_raw)), _vp(rot), _vp(tran),
                                 C.c_float(near), C.c_int32(W), C.c_int32(H), C.c_float(fx),
                                 C.c_float(fy), C.c_float(thresh), _vp(rv[0]), _vp(rv[1]),
                                 _vp(rv[2]), _vp(rv[3]), _vp(img), C.byref(V))
    return img, int(V.value), int(M)
