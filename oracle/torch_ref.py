"""Differentiable fp64 restatement of the rasterizer math in plain torch (CPU).

TEST INFRASTRUCTURE ONLY.  Used to cross-check the ANALYTIC backward formulas that
oracle/gs_oracle.c restates from gaussian.cu:582-772 and :1397-1575 against torch.autograd on
the forward formulas, i.e. an independent derivation.
"""
from __future__ import annotations

import torch


def quat_to_R(q):
    """gaussian.cu:1231-1245 (q already normalised, w,x,y,z)."""
    w, x, y, z = q.unbind(-1)
    return torch.stack([
        1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w,
        2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w,
        2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y,
    ], -1).reshape(*q.shape[:-1], 3, 3)


def project(pos, quat, scale, rot, tran, detach_jacobian=True):
    """A.4: returns pos_i [N,3] and cov2d [N,2,2] for every Gaussian (no culling)."""
    pc = pos @ rot.T + tran
    x, y, z = pc.unbind(-1)
    pos_i = torch.stack([x / z, y / z, pc.norm(dim=-1)], -1)
    zero = torch.zeros_like(z)
    J = torch.stack([1 / z, zero, -x / (z * z), zero, 1 / z, -y / (z * z)], -1).reshape(-1, 2, 3)
    if detach_jacobian:  # the reference backward ignores dJ/dp (gaussian.cu:1397-1421)
        J = J.detach()
    R = quat_to_R(quat)
    M = R * scale.unsqueeze(-2)  # R @ diag(s)
    cov3 = M @ M.transpose(-1, -2)
    JW = J @ rot
    return pos_i, JW @ cov3 @ JW.transpose(-1, -2)


def rasterize_tile(pix_x, pix_y, gx, gy, cov, opa, rgb):
    """A.7 for one tile: pix_* [P]; Gaussians already in order.  No early termination (callers
    keep transmittance above 1e-4).  Returns colour [P,3]."""
    a, b, c, d = cov[:, 0], cov[:, 1], cov[:, 2], cov[:, 3]
    det = a * d - b * c
    dx = pix_x[:, None] - gx[None, :]
    dy = pix_y[:, None] - gy[None, :]
    G = torch.exp(-(d * dx * dx - (b + c) * dx * dy + a * dy * dy) / (2 * det + 1e-14))
    alpha = G * opa[None, :]
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha[:, :-1]], 1), 1)
    w = alpha * T
    return w @ rgb
