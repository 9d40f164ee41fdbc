"""View-parallel training support: one flat gradient bucket, one RCCL all-reduce per iteration.

The reference is single-GPU (no torch.distributed anywhere).  The natural
shard of its workload is the training VIEW: every iteration renders one camera (train.py:95-96),
views are independent, so rank r renders view r of an 8-view batch against a full replica of the
Gaussians and the parameter gradients are averaged.  Layout choices for MI355X / xGMI:

  * all five parameter tensors are views of ONE contiguous fp32 buffer (quat first, so its
    float4 accesses stay 16-byte aligned), and so are their gradients: ``gs_frame_backward``
    writes straight into the bucket, and the exchange is a single ``all_reduce`` of
    N*(4+3+3+1+C)*4 bytes (134 MB at 2.4 M Gaussians) -- one large collective instead of five
    small ones, which is what a point-to-point xGMI mesh wants (per-link bound, 7 links/GPU);
  * the reduction is the mean over views: ReduceOp.AVG inside the RCCL collective (SUM + an in-place scale
    by 1/world on backends without AVG, i.e. gloo in the CPU tests).

Works with any torch.distributed backend ("nccl" == RCCL on ROCm; "gloo" in CPU tests).
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist

ORDER = ("quat", "pos", "scale", "opa", "rgb")  # storage order inside the flat bucket


class FlatGaussianParams:
    """params / grads in the canonical (pos, quat, scale, opa, rgb) order, stored flat."""

    def __init__(self, params: Sequence[torch.Tensor], world_size: int = 1, force_collective: bool = False):
        pos, quat, scale, opa, rgb = params
        self.world_size = int(world_size)
        self.force_collective = bool(force_collective)  # issue the all-reduce even with one rank
        by_name = {"pos": pos, "quat": quat, "scale": scale, "opa": opa, "rgb": rgb}
        total = sum(by_name[k].numel() for k in ORDER)
        dev = pos.device
        self.flat_param = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        views_p, views_g, off = {}, {}, 0
        for k in ORDER:
            t = by_name[k]
            n = t.numel()
            views_p[k] = self.flat_param[off:off + n].view(t.shape)
            views_g[k] = self.flat_grad[off:off + n].view(t.shape)
            views_p[k].copy_(t)
            off += n
        names = ("pos", "quat", "scale", "opa", "rgb")
        self.params: List[torch.Tensor] = [views_p[k] for k in names]
        self.grads: List[torch.Tensor] = [views_g[k] for k in names]

    @property
    def bucket_bytes(self) -> int:
        return self.flat_grad.numel() * 4

    def all_reduce_grads(self, async_op: bool = False):
        """Mean of the per-view gradients over all ranks (no-op for a single process)."""
        if not dist.is_initialized() or (self.world_size <= 1 and not self.force_collective):
            return None
        # RCCL averages inside the collective (ReduceOp.AVG); gloo only sums, so the CPU tests scale afterwards
        self._avg_in_collective = dist.get_backend() == "nccl"
        op = dist.ReduceOp.AVG if self._avg_in_collective else dist.ReduceOp.SUM
        work = dist.all_reduce(self.flat_grad, op=op, async_op=async_op)
        if async_op:
            return work
        if not self._avg_in_collective:
            self.flat_grad.mul_(1.0 / self.world_size)
        return None

    def finish_all_reduce(self, work):
        if work is not None:
            work.wait()
            if not self._avg_in_collective:
                self.flat_grad.mul_(1.0 / self.world_size)

    def broadcast_params(self, src: int = 0):
        if self.world_size > 1 and dist.is_initialized():
            dist.broadcast(self.flat_param, src=src)
