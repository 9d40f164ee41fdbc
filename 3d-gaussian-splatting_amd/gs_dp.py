"""View-parallel training support: one flat gradient buffer, exchanged as one or two RCCL all-reduces per iteration.

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
    by 1/world on backends without AVG, i.e. gloo in the CPU tests);
  * the buffer is also two BUCKETS -- "geometry" = [quat | pos | scale] (10 N floats) and "color" = [opa | rgb]
    ((1 + C) N floats), each contiguous -- because every gradient of a frame only becomes final in the LAST kernel of
    the backward (the per-Gaussian sum of the per-pair rows, frame_project_backward_kernel): that kernel exists in a
    geometry-only and a colour-only flavour (gs_frame_backward_part), so the asynchronous all-reduce of the bucket
    written first runs underneath the kernel that writes the other one, and the fused Adam of the first bucket
    underneath the all-reduce of the second (``begin_bucket`` / ``finish_bucket``; gs_train.Trainer uses them when
    a process group is up).  That is all the overlap the step structure offers without applying stale gradients:
    what runs after the raster backward is short next to the exchange.

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
        n_geom = sum(by_name[k].numel() for k in ("quat", "pos", "scale"))
        self.bucket_ranges = {"geometry": (0, n_geom), "color": (n_geom, total)}  # element ranges of the flat buffers
        self._pending = {}

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

    # ---- bucketed, asynchronous exchange ---------------------------------------------------------------------
    def collective_active(self) -> bool:
        return dist.is_initialized() and (self.world_size > 1 or self.force_collective)

    def begin_bucket(self, name: str):
        """Start the mean all-reduce of one bucket ("geometry" or "color") without waiting for it.  The collective
        runs on the process group's own stream, which first waits for everything enqueued on the current stream so
        far -- i.e. for the kernel that wrote the bucket."""
        if not self.collective_active():
            return
        lo, hi = self.bucket_ranges[name]
        avg = dist.get_backend() == "nccl"
        work = dist.all_reduce(self.flat_grad[lo:hi], op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, async_op=True)
        self._pending[name] = (work, avg)

    def finish_bucket(self, name: str):
        """Make the current stream wait for the bucket's all-reduce (and scale it on backends without AVG)."""
        pend = self._pending.pop(name, None)
        if pend is None:
            return
        work, avg = pend
        work.wait()
        if not avg:
            lo, hi = self.bucket_ranges[name]
            self.flat_grad[lo:hi].mul_(1.0 / self.world_size)

    def broadcast_params(self, src: int = 0):
        if self.world_size > 1 and dist.is_initialized():
            dist.broadcast(self.flat_param, src=src)


class ViewParallelGradStat:
    """The densification statistic of train.py:145-154 when the views of a step are spread over ranks.

    The reference accumulates |pos.grad| of EACH VIEW (``accum_max_grad = max(accum, |grad|)``, or ``+= |grad|`` with
    a per-Gaussian visibility counter for "mean").  After the all-reduce every rank only holds the MEAN gradient of
    the step's views, and |mean| is not what the reference thresholds.  So each rank folds its own, pre-all-reduce
    gradient into a local statistic (one small HIP launch, ``gs_grad_stat_update``), and the local statistics are
    combined only where they are consumed -- at a densification boundary -- with ONE collective: elementwise MAX
    ("max") or SUM ("mean", statistic and counter packed into the same buffer).  Every rank then holds the same
    numbers and takes the same prune / clone / split decisions.
    """

    def __init__(self, n: int, device, mode: str = "max", world_size: int = 1):
        if mode not in ("max", "mean"):
            raise ValueError("mode must be 'max' or 'mean'")
        self.mode, self.world_size = mode, int(world_size)
        # [N,3] statistic + [N] visibility counter in one buffer: a single collective at the boundary
        self._buf = torch.zeros(int(n) * 4, dtype=torch.float32, device=device)
        self.accum = self._buf[: int(n) * 3].view(int(n), 3)
        self.counter = self._buf[int(n) * 3:]

    def clear(self):
        self._buf.zero_()

    def update(self, pos_grad: torch.Tensor, seen: torch.Tensor = None):
        """Fold this rank's view in.  ``pos_grad`` [N,3] is the LOCAL gradient (before the all-reduce); ``seen`` [N]
        is the view's culling mask as float (train.py:152-154), needed for "mean" only."""
        if pos_grad.device.type != "cuda":
            raise RuntimeError("ViewParallelGradStat.update needs a HIP device; there is no CPU fallback")
        if tuple(pos_grad.shape) != tuple(self.accum.shape) or not pos_grad.is_contiguous():
            raise RuntimeError(f"pos_grad must be contiguous {tuple(self.accum.shape)}")
        from gaussian import _lib

        _lib.check(_lib.gs_grad_stat_update(pos_grad.data_ptr(), self.accum.data_ptr(), self.accum.numel(),
                                            1 if self.mode == "max" else 2, torch.cuda.current_stream().cuda_stream),
                   "gs_grad_stat_update")
        if self.mode == "mean":
            if seen is None:
                raise RuntimeError("the 'mean' statistic needs the view's culling mask")
            self.counter.add_(seen)

    def reduce(self):
        """Combine the ranks' statistics in place (no-op for a single process); returns (accum [N,3], counter [N])."""
        if dist.is_initialized() and self.world_size > 1:
            dist.all_reduce(self._buf, op=dist.ReduceOp.MAX if self.mode == "max" else dist.ReduceOp.SUM)
        return self.accum, self.counter
