"""ctypes binding of oracle/_ref/libgs_ref.so -- the REFERENCE's own kernels
(/root/reference/src/gaussian.cu) running on the CPU SIMT emulator (oracle/cuda_cpu_shim.h).

TEST INFRASTRUCTURE ONLY.  Used to pin oracle/gs_oracle.c and to generate tests/golden/.
Every function returns the outputs plus raises on an emulator deadlock; ``undefined_reads()``
counts shuffles that read a lane outside the active mask (behaviour CUDA leaves undefined).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libgs_ref.so")
_lib = None


def available() -> bool:
    return os.path.exists(_SO)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_SO)
        _lib.ref_undefined_reads.restype = C.c_long
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError(f"reference kernel {what}: SIMT emulator deadlock (divergent barrier/shuffle)")


def undefined_reads() -> int:
    return int(lib().ref_undefined_reads())


def reset_counters():
    lib().ref_reset_counters()


_Z3 = np.zeros(3, np.float32)


def world2camera(pos, rot, tran):
    pos, rot, tran = _f(pos), _f(rot), _f(tran)
    res = np.zeros_like(pos)
    _ok(lib().ref_world2camera(_p(pos), _p(rot), _p(tran), _p(res), C.c_uint32(pos.shape[0])), "world2camera")
    return res


def world2camera_backward(grad_out, rot):
    grad_out, rot = _f(grad_out), _f(rot)
    res = np.zeros_like(grad_out)
    _ok(lib().ref_world2camera_backward(_p(grad_out), _p(rot), _p(res), C.c_uint32(grad_out.shape[0])), "w2c_bwd")
    return res


def jacobian(pos_cam):
    pos_cam = _f(pos_cam)
    jac = np.zeros((pos_cam.shape[0], 3, 3), np.float32)
    _ok(lib().ref_jacobian(_p(pos_cam), _p(jac), C.c_uint32(pos_cam.shape[0])), "jacobian")
    return jac


def global_culling(pos, quat, scale, rot, tran, near, half_w, half_h):
    pos, quat, scale, rot, tran = map(_f, (pos, quat, scale, rot, tran))
    n = pos.shape[0]
    rp, rc, mk = np.zeros((n, 3), np.float32), np.zeros((n, 2, 2), np.float32), np.zeros(n, np.int64)
    _ok(lib().ref_global_culling(_p(pos), _p(quat), _p(scale), _p(rot), _p(tran), C.c_uint32(n), C.c_float(near),
                                 C.c_float(half_w), C.c_float(half_h), _p(rp), _p(rc), _p(mk)), "global_culling")
    return rp, rc, mk


def global_culling_backward(pos, quat, scale, rot, tran, gop, goc, mask):
    pos, quat, scale, rot, tran, gop, goc = map(_f, (pos, quat, scale, rot, tran, gop, goc))
    mask = np.ascontiguousarray(mask, np.int64)
    n = pos.shape[0]
    gp, gq, gs = np.zeros((n, 3), np.float32), np.zeros((n, 4), np.float32), np.zeros((n, 3), np.float32)
    _ok(lib().ref_global_culling_backward(_p(pos), _p(quat), _p(scale), _p(rot), _p(tran), C.c_uint32(n), _p(gop),
                                          _p(goc), _p(mask), _p(gp), _p(gq), _p(gs)), "global_culling_backward")
    return gp, gq, gs


def calc_tile_list(pos, cov, maxp, thresh, method, tlx, tly, ntx, nty, leftmost, topmost, top=None, bottom=None,
                   left=None, right=None):
    pos, cov = _f(pos), _f(cov)
    T = ntx * nty
    cnt = np.zeros(T, np.int32)
    lst = np.full((T, maxp), -1, np.int32)
    e = [_f(v) if v is not None else np.zeros(T, np.float32) for v in (top, bottom, left, right)]
    _ok(lib().ref_calc_tile_list(_p(pos), _p(cov), C.c_uint32(pos.shape[0]), _p(e[0]), _p(e[1]), _p(e[2]), _p(e[3]),
                                 C.c_uint32(T), _p(cnt), _p(lst), C.c_uint32(maxp), C.c_float(thresh), C.c_int(method),
                                 C.c_float(tlx), C.c_float(tly), C.c_uint32(ntx), C.c_uint32(nty),
                                 C.c_float(leftmost), C.c_float(topmost)), "calc_tile_list")
    return cnt, lst


def gather_gaussians(accum, lst, max_points_for_tile):
    accum, lst = np.ascontiguousarray(accum, np.int32), np.ascontiguousarray(lst, np.int32)
    M = int(accum[-1])
    g, t = np.zeros(M, np.int32), np.zeros(M, np.int32)
    _ok(lib().ref_gather_gaussians(_p(accum), _p(lst), _p(g), _p(t), C.c_int(accum.shape[0] - 1),
                                   C.c_int(max_points_for_tile), C.c_int(lst.shape[1])), "gather_gaussians")
    return g, t


def draw(pos, rgb, opa, cov, accum, h, w, fx, fy, weight_normalize=False, sigmoid=False, use_sh=False, fast=False,
         rays_o=None, lefttop=None, vdx=None, vdy=None):
    pos, rgb, opa, cov = map(_f, (pos, rgb, opa, cov))
    accum = np.ascontiguousarray(accum, np.int32)
    res = np.zeros((h, w, 3), np.float32)
    rv = [_f(v) if v is not None else _Z3.copy() for v in (rays_o, lefttop, vdx, vdy)]
    _ok(lib().ref_draw(_p(pos), _p(rgb), _p(opa), _p(cov), _p(accum), _p(res), C.c_uint32(h), C.c_uint32(w),
                       C.c_float(fx), C.c_float(fy), C.c_int(bool(weight_normalize)), C.c_int(bool(sigmoid)),
                       C.c_int(bool(fast)), _p(rv[0]), _p(rv[1]), _p(rv[2]), _p(rv[3]), C.c_int(bool(use_sh))), "draw")
    return res


def draw_backward(pos, rgb, opa, cov, accum, output, grad_output, fx, fy, weight_normalize=False, sigmoid=False,
                  use_sh=False, fast=False, rays_o=None, lefttop=None, vdx=None, vdy=None):
    pos, rgb, opa, cov, output, grad_output = map(_f, (pos, rgb, opa, cov, output, grad_output))
    accum = np.ascontiguousarray(accum, np.int32)
    h, w = output.shape[:2]
    gp, gr, go, gc = np.zeros_like(pos), np.zeros_like(rgb), np.zeros_like(opa), np.zeros_like(cov)
    rv = [_f(v) if v is not None else _Z3.copy() for v in (rays_o, lefttop, vdx, vdy)]
    _ok(lib().ref_draw_backward(_p(pos), _p(rgb), _p(opa), _p(cov), _p(accum), _p(output), _p(grad_output), _p(gp),
                                _p(gr), _p(go), _p(gc), C.c_uint32(h), C.c_uint32(w), C.c_float(fx), C.c_float(fy),
                                C.c_int(bool(weight_normalize)), C.c_int(bool(sigmoid)), C.c_int(bool(fast)),
                                _p(rv[0]), _p(rv[1]), _p(rv[2]), _p(rv[3]), C.c_int(bool(use_sh))), "draw_backward")
    return gp, gr, go, gc
