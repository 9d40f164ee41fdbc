"""Autograd boundary of the rasterizer -- same public API as the reference's ``renderer.py``.

``draw``, ``global_culling``, ``world2camera_func`` and ``trunc_exp`` keep the reference's
positional signatures, return values and gradient slots (renderer.py:6-158 there), so
``splatter.py`` can import them unchanged; the work happens in the gfx950 kernels behind the
``gaussian`` module.  Two deliberate differences: incoming gradients are made contiguous before
their pointers are handed to the kernels (the reference reads them with whatever strides autograd
produced), and saved tensors are not re-validated on the backward path.
"""
from __future__ import annotations

import torch

import gaussian

__all__ = ["draw", "global_culling", "world2camera_func", "trunc_exp"]


class _Drawer(torch.autograd.Function):
    """Tile rasterizer: sorted (pos, rgb, opa, cov) + per-tile prefix -> padded image [H,W,3]."""

    @staticmethod
    def forward(ctx, gaussians_pos, gaussians_rgb, gaussians_opa, gaussians_cov, tile_n_point_accum, padded_height,
                padded_width, focal_x, focal_y, render_weight_normalize=False, sigmoid=False, use_sh_coeff=False,
                fast=False, rays_o=None, lefttop_pos=None, vec_dx=None, vec_dy=None):
        image = torch.zeros(padded_height, padded_width, 3, device=gaussians_pos.device, dtype=torch.float32)
        gaussian.draw(gaussians_pos, gaussians_rgb, gaussians_opa, gaussians_cov, tile_n_point_accum, image, focal_x,
                      focal_y, render_weight_normalize, sigmoid, fast, rays_o, lefttop_pos, vec_dx, vec_dy,
                      use_sh_coeff)
        ctx.save_for_backward(gaussians_pos, gaussians_rgb, gaussians_opa, gaussians_cov, tile_n_point_accum, image,
                              rays_o, lefttop_pos, vec_dx, vec_dy)
        ctx.cfg = (focal_x, focal_y, render_weight_normalize, sigmoid, fast, use_sh_coeff)
        return image

    @staticmethod
    def backward(ctx, grad_output):
        pos, rgb, opa, cov, accum, image, rays_o, lefttop_pos, vec_dx, vec_dy = ctx.saved_tensors
        focal_x, focal_y, weight_normalize, sigmoid, fast, use_sh_coeff = ctx.cfg
        grads = [torch.zeros_like(t) for t in (pos, rgb, opa, cov)]
        gaussian.draw_backward(pos, rgb, opa, cov, accum, image, grad_output.contiguous(), *grads, focal_x, focal_y,
                               weight_normalize, sigmoid, fast, rays_o, lefttop_pos, vec_dx, vec_dy, use_sh_coeff)
        return (*grads, *([None] * 13))


draw = _Drawer.apply


class _trunc_exp(torch.autograd.Function):
    """exp with a clamped backward (reference renderer.py:91-100)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-1, 1))


trunc_exp = _trunc_exp.apply


class _world2camera(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, rot, tran):
        ctx.save_for_backward(rot)
        out = torch.zeros_like(pos)
        gaussian.world2camera(pos, rot, tran, out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (rot,) = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        grad_in = torch.zeros_like(grad_out)
        gaussian.world2camera_backward(grad_out, rot, grad_in)
        return grad_in, None, None


world2camera_func = _world2camera.apply


class _GlobalCulling(torch.autograd.Function):
    """Frustum culling + EWA projection: -> (pos_i [N,3], cov2d [N,2,2], mask [N] int64)."""

    @staticmethod
    def forward(ctx, pos, quat, scale, current_rot, current_tran, near, half_width, half_height):
        n = pos.shape[0]
        res_pos = torch.zeros_like(pos)
        res_cov = torch.zeros((n, 2, 2), device=pos.device, dtype=torch.float32)
        mask = torch.zeros(n, dtype=torch.long, device=pos.device)
        gaussian.global_culling(pos, quat, scale, current_rot, current_tran, res_pos, res_cov, mask, near, half_width,
                                half_height)
        ctx.save_for_backward(mask, pos, quat, scale, current_rot, current_tran)
        ctx.mark_non_differentiable(mask)
        return res_pos, res_cov, mask

    @staticmethod
    def backward(ctx, gradout_pos, gradout_cov, _grad_mask):
        mask, pos, quat, scale, rot, tran = ctx.saved_tensors
        n = pos.shape[0]
        g_pos = torch.zeros_like(pos)
        g_quat = torch.zeros((n, 4), device=pos.device, dtype=torch.float32)
        g_scale = torch.zeros((n, 3), device=pos.device, dtype=torch.float32)
        gaussian.global_culling_backward(pos, quat, scale, rot, tran, gradout_pos.contiguous(),
                                         gradout_cov.contiguous(), mask, g_pos, g_quat, g_scale)
        return g_pos, g_quat, g_scale, None, None, None, None, None


global_culling = _GlobalCulling.apply
