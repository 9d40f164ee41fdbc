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

  * two EXCHANGE modes per bucket (``exchange=``): "all_reduce" -- mean all-reduce, then every rank runs the fused Adam
    over the whole bucket (replicated optimizer) --, and "reduce_scatter" -- mean reduce-scatter of the bucket (rank r
    receives the mean of ITS 1/world slice, in place), fused Adam over that slice only with moments that exist for the
    slice only (1/world of the optimizer state and of its HBM traffic), then an all-gather of the updated PARAMETERS
    (in place).  The bytes on the xGMI links are the same (a ring all-reduce is a reduce-scatter + an all-gather);
    what changes is what can hide where: the all-reduce mode hides Adam under the other bucket's all-reduce, the
    reduce-scatter mode shortens Adam to 1/world but exposes the parameter all-gather in front of the next forward.
    Every bucket is padded to a multiple of 4 x world elements so that the slices are equal and float4-aligned (pad
    elements have zero gradient: Adam leaves them at zero).  Both modes give the same parameters bit for bit wherever
    the backend's reductions agree (2 gloo ranks: tests/test_host_logic.py).

Works with any torch.distributed backend ("nccl" == RCCL on ROCm; "gloo" in CPU tests).
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist

ORDER = ("quat", "pos", "scale", "opa", "rgb")  # storage order inside the flat bucket


EXCHANGES = ("all_reduce", "reduce_scatter")


class FlatGaussianParams:
    """params / grads in the canonical (pos, quat, scale, opa, rgb) order, stored flat."""

    def __init__(self, params: Sequence[torch.Tensor], world_size: int = 1, force_collective: bool = False,
                 exchange: str = "all_reduce", rank: int = None):
        pos, quat, scale, opa, rgb = params
        if exchange not in EXCHANGES:
            raise ValueError(f"exchange must be one of {EXCHANGES}")
        self.world_size = int(world_size)
        self.exchange = exchange
        self.force_collective = bool(force_collective)  # issue the collectives even with one rank
        if rank is None:
            rank = dist.get_rank() if (dist.is_initialized() and self.world_size > 1) else 0
        self.rank = int(rank)
        by_name = {"pos": pos, "quat": quat, "scale": scale, "opa": opa, "rgb": rgb}
        dev = pos.device
        # two buckets, each padded to a multiple of 4 * world elements (equal, float4-aligned slices per rank)
        quantum = 4 * max(self.world_size, 1)
        pad_to = lambda n: (n + quantum - 1) // quantum * quantum  # noqa: E731
        n_geom = pad_to(sum(by_name[k].numel() for k in ("quat", "pos", "scale")))
        n_col = pad_to(sum(by_name[k].numel() for k in ("opa", "rgb")))
        total = n_geom + n_col
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        views_p, views_g, self.offsets = {}, {}, {}
        off = 0
        for k in ORDER:
            if k == "opa":
                off = n_geom  # the colour bucket starts behind the geometry bucket's padding
            t = by_name[k]
            n = t.numel()
            self.offsets[k] = (off, off + n)
            views_p[k] = self.flat_param[off:off + n].view(t.shape)
            views_g[k] = self.flat_grad[off:off + n].view(t.shape)
            views_p[k].copy_(t)
            off += n
        names = ("pos", "quat", "scale", "opa", "rgb")
        self.params: List[torch.Tensor] = [views_p[k] for k in names]
        self.grads: List[torch.Tensor] = [views_g[k] for k in names]
        self.bucket_ranges = {"geometry": (0, n_geom), "color": (n_geom, total)}  # element ranges of the flat buffers
        # Adam group table in ORDER: a group ends where the next begins, so the padding of a bucket belongs to its
        # last group (zero gradient, zero moments: it never moves)
        self.group_ends = [self.offsets["quat"][1], self.offsets["pos"][1], n_geom, self.offsets["opa"][1], total]
        self._pending = {}
        self._pending_gather = {}

    @property
    def bucket_bytes(self) -> int:
        return self.flat_grad.numel() * 4

    def shard_range(self, name: str):
        """Element range of the flat buffers this rank owns inside bucket ``name`` (reduce-scatter mode)."""
        lo, hi = self.bucket_ranges[name]
        s = (hi - lo) // max(self.world_size, 1)
        return lo + self.rank * s, lo + (self.rank + 1) * s

    def optimizer_range(self, name: str):
        """What this rank's optimizer updates of bucket ``name``: all of it (replicated) or its slice (sharded)."""
        return self.shard_range(name) if self.exchange == "reduce_scatter" else self.bucket_ranges[name]

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
        """Start the mean all-reduce (or reduce-scatter) of one bucket ("geometry" or "color") without waiting for it.
        The collective runs on the process group's own stream, which first waits for everything enqueued on the
        current stream so far -- i.e. for the kernel that wrote the bucket."""
        if not self.collective_active():
            return
        lo, hi = self.bucket_ranges[name]
        avg = dist.get_backend() == "nccl"
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        if self.exchange == "reduce_scatter":
            slo, shi = self.shard_range(name)
            # in place: the output is this rank's slice of the input (RCCL's in-place reduce-scatter layout)
            work = dist.reduce_scatter_tensor(self.flat_grad[slo:shi], self.flat_grad[lo:hi], op=op, async_op=True)
            self._pending[name] = (work, avg, (slo, shi))
        else:
            work = dist.all_reduce(self.flat_grad[lo:hi], op=op, async_op=True)
            self._pending[name] = (work, avg, (lo, hi))

    def finish_bucket(self, name: str):
        """Make the current stream wait for the bucket's reduction (and scale it on backends without AVG)."""
        pend = self._pending.pop(name, None)
        if pend is None:
            return
        work, avg, (lo, hi) = pend
        work.wait()
        if not avg:
            self.flat_grad[lo:hi].mul_(1.0 / self.world_size)

    def begin_gather(self, name: str):
        """reduce-scatter mode: all-gather the bucket's updated PARAMETER slices (in place), without waiting."""
        if not self.collective_active() or self.exchange != "reduce_scatter":
            return
        lo, hi = self.bucket_ranges[name]
        slo, shi = self.shard_range(name)
        self._pending_gather[name] = dist.all_gather_into_tensor(self.flat_param[lo:hi], self.flat_param[slo:shi],
                                                                 async_op=True)

    def finish_gather(self, name: str = None):
        """Make the current stream wait for the parameter all-gather(s): before anything reads the parameters."""
        for k in ([name] if name else list(self._pending_gather)):
            work = self._pending_gather.pop(k, None)
            if work is not None:
                work.wait()

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
