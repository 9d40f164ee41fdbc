"""Drop-in replacement for the reference's pybind11 extension module ``gaussian``.

Same names, same positional arguments, same "caller allocates, function fills, returns None"
convention as ``src/bindings.cpp:21-50`` -- so the reference's ``renderer.py`` / ``splatter.py``
run unchanged with this directory on ``sys.path`` -- but every function forwards raw device
pointers to the hand-written gfx950 kernels behind the C ABI of ``libgs_amd.so``
(``include/gs_abi.h``).  Launches go on ``torch.cuda.current_stream()`` (the reference uses the
legacy null stream).

Unlike the reference (whose ``CHECK_INPUT`` macros are never called, common.hpp:11-20), inputs
are validated: wrong dtype / device / contiguity raise ``RuntimeError`` -- the same exception
type ``data_ptr<float>()`` raises there on a dtype mismatch.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check as _check

__all__ = [
    "culling", "world2camera", "world2camera_backward", "jacobian", "Tiles", "Gaussian3ds",
    "calc_tile_list", "gather_gaussians", "draw", "draw_backward", "global_culling",
    "global_culling_backward",
]


class Tiles:
    """common.hpp:36-52: bag of four float tensors [T] (read by binning methods 0/1 only)."""

    def __init__(self):
        self.top = None
        self.bottom = None
        self.left = None
        self.right = None


class Gaussian3ds:
    """common.hpp:54-74: bag of tensors; calc_tile_list reads ``pos`` [V,3] and ``cov`` [V,2,2]."""

    def __init__(self):
        self.pos = None
        self.rgb = None
        self.opa = None
        self.quat = None
        self.scale = None
        self.cov = None


def _t(x, name, dtype=torch.float32):
    if not isinstance(x, torch.Tensor):
        raise RuntimeError(f"{name} must be a torch.Tensor, got {type(x).__name__}")
    if x.dtype != dtype:
        raise RuntimeError(f"expected scalar type {dtype} for {name} but found {x.dtype}")
    if not x.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA (HIP) tensor")
    if not x.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return x.data_ptr()


def _opt(x, name):
    return None if x is None else _t(x, name)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def culling(pos, rgb, quatenions, scales, w2c_quat, w2c_tran):
    """bindings.cpp:23 / gaussian.cu:6-8 -- a stub in the reference; nothing to compute."""
    _check(_lib.gs_culling(), "culling")


def world2camera(pos, rot, trans, res):
    _check(_lib.gs_world2camera(_t(pos, "pos"), _t(rot, "rot"), _t(trans, "trans"), _t(res, "res"),
                                pos.shape[0], _stream()), "world2camera")


def world2camera_backward(grad_out, rot, grad_inp):
    _check(_lib.gs_world2camera_backward(_t(grad_out, "grad_out"), _t(rot, "rot"), _t(grad_inp, "grad_inp"),
                                         grad_out.shape[0], _stream()), "world2camera_backward")


def jacobian(pos_camera_space, jacobian):
    _check(_lib.gs_jacobian(_t(pos_camera_space, "pos_camera_space"), _t(jacobian, "jacobian"),
                            pos_camera_space.shape[0], _stream()), "jacobian")


def global_culling(pos, quat, scale, current_rot, current_tran, res_pos, res_cov, culling_mask, near, half_width,
                   half_height):
    n = pos.shape[0]
    if quat.shape[0] != n or scale.shape[0] != n or res_pos.shape[0] != n or res_cov.shape[0] != n \
            or culling_mask.shape[0] != n:
        raise RuntimeError("global_culling: first dimensions disagree")
    _check(_lib.gs_global_culling(_t(pos, "pos"), _t(quat, "quat"), _t(scale, "scale"), _t(current_rot, "current_rot"),
                                  _t(current_tran, "current_tran"), n, float(near), float(half_width),
                                  float(half_height), _t(res_pos, "res_pos"), _t(res_cov, "res_cov"),
                                  _t(culling_mask, "culling_mask", torch.int64), _stream()), "global_culling")


def global_culling_backward(pos, quat, scale, current_rot, current_tran, gradout_pos, gradout_cov, culling_mask,
                            gradinput_pos, gradinput_quat, gradinput_scale):
    _check(_lib.gs_global_culling_backward(
        _t(pos, "pos"), _t(quat, "quat"), _t(scale, "scale"), _t(current_rot, "current_rot"),
        _t(current_tran, "current_tran"), pos.shape[0], _t(gradout_pos, "gradout_pos"), _t(gradout_cov, "gradout_cov"),
        _t(culling_mask, "culling_mask", torch.int64), _t(gradinput_pos, "gradinput_pos"),
        _t(gradinput_quat, "gradinput_quat"), _t(gradinput_scale, "gradinput_scale"), _stream()),
        "global_culling_backward")


def calc_tile_list(gaussians_image_space, tile_info, tile_n_point, tile_gaussian_list, thresh, method, tile_length_x,
                   tile_length_y, n_tiles_x, n_tiles_y, leftmost, topmost):
    g = gaussians_image_space
    method = int(method)
    if tile_gaussian_list.dim() != 2:
        raise RuntimeError("tile_gaussian_list must be [T, MAXP]")
    edges = [None] * 4
    if method != 2:
        edges = [_t(getattr(tile_info, k), f"tile_info.{k}") for k in ("top", "bottom", "left", "right")]
    _check(_lib.gs_calc_tile_list(_t(g.pos, "pos"), _t(g.cov, "cov"), g.pos.shape[0], *edges,
                                  _t(tile_n_point, "tile_n_point", torch.int32),
                                  _t(tile_gaussian_list, "tile_gaussian_list", torch.int32),
                                  tile_gaussian_list.shape[1], float(thresh), method, float(tile_length_x),
                                  float(tile_length_y), int(n_tiles_x), int(n_tiles_y), float(leftmost),
                                  float(topmost), _stream()), "calc_tile_list")


def gather_gaussians(tile_n_point_accum, tile_gaussian_list, gathered_list, tile_ids_for_points, max_points_for_tile):
    _check(_lib.gs_gather_gaussians(_t(tile_n_point_accum, "tile_n_point_accum", torch.int32),
                                    _t(tile_gaussian_list, "tile_gaussian_list", torch.int32),
                                    _t(gathered_list, "gathered_list", torch.int32),
                                    _t(tile_ids_for_points, "tile_ids_for_points", torch.int32),
                                    tile_n_point_accum.shape[0] - 1, int(max_points_for_tile),
                                    tile_gaussian_list.shape[1], _stream()), "gather_gaussians")


def draw(gaussian_pos, gaussian_rgb, gaussian_opa, gaussian_cov, tile_n_point_accum, res, focal_x, focal_y,
         weight_normalize, sigmoid, fast, rays_o, lefttop_pos, vec_dx, vec_dy, use_sh_coeff):
    h, w = int(res.shape[0]), int(res.shape[1])
    M = int(gaussian_pos.shape[0])
    D = 27 if use_sh_coeff else 3
    if gaussian_rgb.shape[0] != M or gaussian_rgb.numel() != M * D:
        raise RuntimeError(f"gaussian_rgb must be [M,{D}]")
    if tile_n_point_accum.shape[0] != (h // 16) * (w // 16) + 1:
        raise RuntimeError("tile_n_point_accum must have n_tiles + 1 entries")
    _check(_lib.gs_draw(_t(gaussian_pos, "gaussian_pos"), _t(gaussian_rgb, "gaussian_rgb"),
                        _t(gaussian_opa, "gaussian_opa"), _t(gaussian_cov, "gaussian_cov"),
                        _t(tile_n_point_accum, "tile_n_point_accum", torch.int32), _t(res, "res"), h, w, M,
                        float(focal_x), float(focal_y), int(bool(weight_normalize)), int(bool(sigmoid)),
                        int(bool(fast)), _opt(rays_o, "rays_o"), _opt(lefttop_pos, "lefttop_pos"),
                        _opt(vec_dx, "vec_dx"), _opt(vec_dy, "vec_dy"), int(bool(use_sh_coeff)), _stream()), "draw")


def draw_backward(gaussian_pos, gaussian_rgb, gaussian_opa, gaussian_cov, tile_n_point_accum, output, grad_output,
                  grad_pos, grad_rgb, grad_opa, grad_cov, focal_x, focal_y, weight_normalize, sigmoid, fast, rays_o,
                  lefttop_pos, vec_dx, vec_dy, use_sh_coeff):
    h, w = int(output.shape[0]), int(output.shape[1])
    M = int(gaussian_pos.shape[0])
    nbytes = _lib.gs_draw_backward_workspace_bytes(M, h, w)
    # scratch for the per-bucket pixel checkpoints; comes from torch's caching allocator
    ws = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=gaussian_pos.device)
    _check(_lib.gs_draw_backward(
        _t(gaussian_pos, "gaussian_pos"), _t(gaussian_rgb, "gaussian_rgb"), _t(gaussian_opa, "gaussian_opa"),
        _t(gaussian_cov, "gaussian_cov"), _t(tile_n_point_accum, "tile_n_point_accum", torch.int32),
        _t(output, "output"), _t(grad_output, "grad_output"), _t(grad_pos, "grad_pos"), _t(grad_rgb, "grad_rgb"),
        _t(grad_opa, "grad_opa"), _t(grad_cov, "grad_cov"), h, w, M, float(focal_x), float(focal_y),
        int(bool(weight_normalize)), int(bool(sigmoid)), int(bool(fast)), _opt(rays_o, "rays_o"),
        _opt(lefttop_pos, "lefttop_pos"), _opt(vec_dx, "vec_dx"), _opt(vec_dy, "vec_dy"), int(bool(use_sh_coeff)),
        ws.data_ptr(), int(nbytes), _stream()), "draw_backward")
