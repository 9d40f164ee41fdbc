"""Densification: the reference's ``Gaussian3ds.adaptive_control`` / ``reset_opa`` (splatter.py:119-228).

``adaptive_control(params, grad, ...)`` takes the five parameter tensors (pos, quat, scale, opa, rgb) and the
accumulated position-gradient statistic, and returns the five tensors of the new Gaussian set in the
reference's order: kept Gaussians (pruned by opacity / size; the ones that are split get ``scale / 1.6``
and a fresh sample), then the clones, then the second split samples.  Two HIP launches classify and count,
one applies; the only host synchronisation is the read of the four counts between them (the reference
synchronises four times and launches ~40 torch kernels).  The optimizer state is dropped afterwards, as in
the reference, which re-creates ``torch.optim.Adam`` (train.py:169-179).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence, Tuple

import torch

from gaussian import _lib

SCALE_ACT = {"abs": 0, "exp": 1}
AGG = {"max": 0, "mean": 1}


def inverse_sigmoid(y: float) -> float:  # utils.py:350-351
    return -math.log(1 / y - 1)


def adaptive_control(params: Sequence[torch.Tensor], grad: torch.Tensor, taus: float, delete_thresh: float,
                     scale_activation: str = "abs", grad_thresh: float = 0.0002, grad_aggregation: str = "max",
                     use_clone: bool = True, use_split: bool = True, clone_dt: float = 0.01,
                     generator: Optional[torch.Generator] = None,
                     draws: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
                     ) -> Tuple[List[torch.Tensor], Tuple[int, int, int]]:
    """Returns ([pos, quat, scale, opa, rgb] of the new set, (kept, cloned, split)).

    ``draws`` = (eps1, eps2), two [>= n_split, 3] standard-normal tensors, replaces the internal
    ``torch.randn`` (tests replay the reference's draws with it)."""
    pos, quat, scale, opa, rgb = params
    n = int(pos.shape[0])
    dev = pos.device
    if dev.type != "cuda":
        raise RuntimeError("adaptive_control needs a HIP device; there is no CPU fallback")
    color_dim = int(rgb.shape[1])
    for name, t, shape in (("pos", pos, (n, 3)), ("quat", quat, (n, 4)), ("scale", scale, (n, 3)), ("opa", opa, (n,)),
                           ("rgb", rgb, (n, color_dim)), ("grad", grad, (n, 3))):
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or tuple(t.shape) != shape:
            raise RuntimeError(f"{name} must be a contiguous float32 HIP tensor of shape {shape}")
    opts = _lib.GsDensifyOpts(float(taus), float(delete_thresh), float(grad_thresh), float(clone_dt),
                              SCALE_ACT[scale_activation], AGG[grad_aggregation], int(bool(use_clone)),
                              int(bool(use_split)), color_dim)
    stream = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(int(_lib.gs_densify_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    counts = torch.zeros(4, dtype=torch.int64, device=dev)
    _lib.check(_lib.gs_densify_classify(scale.data_ptr(), opa.data_ptr(), grad.data_ptr(), n, C.byref(opts),
                                        counts.data_ptr(), ws.data_ptr(), ws.numel(), stream), "gs_densify_classify")
    kept, cloned, split, total = (int(v) for v in counts.cpu())  # the one host synchronisation
    if draws is None:  # MultivariateNormal.sample() twice (utils.py:396-401): two [n_split, 3] normal blocks
        draws = tuple(torch.randn(split, 3, device=dev, generator=generator) for _ in range(2))
    eps1, eps2 = (d.contiguous() for d in draws)
    if eps1.shape[0] < split or eps2.shape[0] < split:
        raise RuntimeError(f"need {split} normal draws per block, got {eps1.shape[0]}, {eps2.shape[0]}")
    out = [torch.empty((total,) + tuple(t.shape[1:]), dtype=torch.float32, device=dev) for t in params]
    _lib.check(_lib.gs_densify_apply(pos.data_ptr(), quat.data_ptr(), scale.data_ptr(), opa.data_ptr(), rgb.data_ptr(),
                                     grad.data_ptr(), n, C.byref(opts), eps1.data_ptr(), eps2.data_ptr(),
                                     min(int(eps1.shape[0]), int(eps2.shape[0])), *(t.data_ptr() for t in out), total,
                                     counts.data_ptr(), ws.data_ptr(), ws.numel(), stream), "gs_densify_apply")
    return out, (kept, cloned, split)


def reset_opa(opa: torch.Tensor) -> torch.Tensor:
    """splatter.py:119-120: every opacity logit back to logit(0.01)."""
    return opa.fill_(inverse_sigmoid(0.01))
