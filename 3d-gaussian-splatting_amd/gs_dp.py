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

import math
from typing import List, Sequence

import torch
import torch.distributed as dist

ORDER = ("quat", "pos", "scale", "opa", "rgb")  # storage order inside the flat bucket
WIDTH = {"quat": 4, "pos": 3, "scale": 3, "opa": 1}  # floats per Gaussian ("rgb": its colour dimension)


EXCHANGES = ("all_reduce", "reduce_scatter")


def project_slice_size(n: int) -> int:
    """Gaussians per slice of the frame path's project stage (csrc/gs_frame_layout.h: gs_strip_plan_for): the array is
    cut into at most 256 slices of a multiple of 256 Gaussians.  The exchange slices below are made of whole project
    slices, so that the NEXT frame's project stage can be issued slice by slice behind the optimizer."""
    n = max(int(n), 1)
    return -(-(-(-n // 256)) // 256) * 256


class _WorkList:
    """Several works waited for as one (the fallback path of backends without grouped collectives on device tensors);
    `copies`: (destination, source) pairs to copy once the works are done (the emulated all-gather)."""

    def __init__(self, works, copies=()):
        self.works, self.copies = works, copies

    def wait(self):
        for w in self.works:
            w.wait()
        for dst, src in self.copies:
            dst.copy_(src)


class FlatGaussianParams:
    """params / grads in the canonical (pos, quat, scale, opa, rgb) order, stored flat.

    Layout (round 4): every tensor owns a region of ``Np * width`` floats, ``Np`` = N rounded up to a multiple of
    4 x world -- [quat | pos | scale | opa | rgb]; the pad rows have zero gradient and never move.  Two views of it:
      * BUCKETS (round 2): "geometry" = [quat | pos | scale], "color" = [opa | rgb], each one contiguous range;
      * SLICES (round 4): slice k = Gaussians [g_k, g_k+1) -- one range in each of the five regions.  The gradients of
        a frame only become final in the LAST kernel of the backward, the per-Gaussian sum of the gradient rows
        (frame_project_backward_kernel), and that kernel is independent per Gaussian: it is issued slice by slice, and a
        slice's exchange starts as soon as its sums are written -- underneath the sums of the following slices, the fused
        Adam of the preceding ones and (gs_train.Trainer) the project stage of the NEXT frame, which needs nothing but
        the updated parameters of its slice.  Slice boundaries are whole project slices (``project_slice_size``) and
        multiples of 4 x world Gaussians, so every range splits into equal, float4-aligned shards ("reduce_scatter").
    """

    def __init__(self, params: Sequence[torch.Tensor], world_size: int = 1, force_collective: bool = False,
                 exchange: str = "all_reduce", rank: int = None, n_slices: int = None):
        pos, quat, scale, opa, rgb = params
        if exchange not in EXCHANGES:
            raise ValueError(f"exchange must be one of {EXCHANGES}")
        self.world_size = int(world_size)
        self.exchange = exchange
        self.force_collective = bool(force_collective)  # issue the collectives even with one rank
        self.enable_collective = True  # False (measurement only, bench.py): the step without its gradient exchange
        # True (gs_train.Trainer): the exchange SUMS and the optimizer applies the 1 / world of the mean to the gradient
        # it reads (gs_adam_step_multi's grad_scale).  RCCL's ReduceOp.AVG is a pre-multiplied sum: it launches its
        # scaling kernel for every range even on ONE rank (oneRankReduce<FuncPreMulSum>: 40 launches, 0.28 ms per step
        # at 2.4 M Gaussians in the round-4 trace), a SUM over one rank is nothing at all.  False: after finish_exchange
        # the buffer holds the mean (AVG inside RCCL; SUM + a scaling pass on backends without AVG).
        self.mean_in_optimizer = False
        if rank is None:
            rank = dist.get_rank() if (dist.is_initialized() and self.world_size > 1) else 0
        self.rank = int(rank)
        by_name = {"pos": pos, "quat": quat, "scale": scale, "opa": opa, "rgb": rgb}
        dev = pos.device
        n = int(pos.shape[0])
        self.n = n
        quantum = 4 * max(self.world_size, 1)
        self.n_pad = (n + quantum - 1) // quantum * quantum  # rows of every region
        width = dict(WIDTH, rgb=int(rgb.numel() // max(n, 1)) if n else (rgb.shape[1] if rgb.dim() == 2 else 1))
        self.width = width
        total = self.n_pad * sum(width[k] for k in ORDER)
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        views_p, views_g, self.offsets, self.region = {}, {}, {}, {}
        off = 0
        for k in ORDER:
            t = by_name[k]
            cnt = t.numel()
            self.offsets[k] = (off, off + cnt)
            self.region[k] = off
            views_p[k] = self.flat_param[off:off + cnt].view(t.shape)
            views_g[k] = self.flat_grad[off:off + cnt].view(t.shape)
            views_p[k].copy_(t)
            off += self.n_pad * width[k]
        names = ("pos", "quat", "scale", "opa", "rgb")
        self.params: List[torch.Tensor] = [views_p[k] for k in names]
        self.grads: List[torch.Tensor] = [views_g[k] for k in names]
        n_geom = self.region["opa"]
        self.bucket_ranges = {"geometry": (0, n_geom), "color": (n_geom, total)}  # element ranges of the flat buffers
        # Adam group table in ORDER: a group ends where the next region begins (pad rows: zero gradient, zero moments)
        self.group_ends = [self.region["pos"], self.region["scale"], self.region["opa"], self.region["rgb"], total]
        # ---- slices of the Gaussian array (exchange units)
        self.project_slice = project_slice_size(n)
        self._pending = {}
        self._pending_gather = {}
        self._view_cache = {}
        self.set_slices(n_slices)

    def set_slices(self, n_slices: int = None):
        """(Re)cut the Gaussian array into exchange slices.  Allowed at any time between steps for a replicated optimizer
        (its moments mirror the flat buffer); a sharded optimizer packs its moments by slice and must be rebuilt."""
        if self._pending or self._pending_gather:
            raise RuntimeError("set_slices() with an exchange in flight")
        n = self.n
        quantum = 4 * max(self.world_size, 1)
        per = self.project_slice
        step = per * (quantum // math.gcd(per, quantum))  # whole project slices AND multiples of 4 x world Gaussians
        n_steps = max(-(-self.n_pad // step), 1)
        self._view_cache = {k: v for k, v in self._view_cache.items() if not (isinstance(k, tuple) and k[0] == "slice")}
        if n_slices is None:
            # One slice on a single rank (nothing travels: nothing to hide), two from a million Gaussians on when there
            # are peers.  Measured on one rank at 2.4 M Gaussians (round 4, profiles/r04_e_exchange_probe_*.jsonl): every
            # extra slice costs the step 70 - 90 us -- the per-slice kernels fill less of the chip (a project launch over
            # half the slices takes 55 us against 80 us for all of them) -- while the exchange time a further slice can
            # hide shrinks as 1 / K: DESIGN.md section 4 has the model.
            # Round 5: with peers a single slice leaves the whole exchange exposed (sums -> exchange -> wait
            # -> Adam), whatever the scene size; the 1 M threshold of round 4 came from single-GPU measurements only.  Two
            # slices whenever there are peers and the array can be cut (below 65,536 Gaussians every kernel of the step is
            # launch-latency bound and the exchange is a few tens of microseconds of wire latency: nothing to hide);
            # Trainer.tune_slices settles it by measurement on the real links.
            n_slices = 2 if (n >= 65_536 and self.world_size > 1) else 1
        k_slices = max(1, min(int(n_slices), n_steps))
        bounds = sorted({min(round(i * n_steps / k_slices) * step, self.n_pad) for i in range(k_slices)} | {self.n_pad})
        self.slice_bounds = [0] + [b for b in bounds if b > 0]

    # ---- geometry of the exchange units --------------------------------------------------------------------------
    @property
    def n_slices(self) -> int:
        return len(self.slice_bounds) - 1

    def slice_gaussians(self, k: int):
        """Gaussian range [g0, g1) of slice k, clipped to the real Gaussians (the pad rows belong to the last slice)."""
        return self.slice_bounds[k], min(self.slice_bounds[k + 1], self.n)

    def slice_ranges(self, k: int):
        """The five element ranges (ORDER) slice k owns in the flat buffers."""
        g0, g1 = self.slice_bounds[k], self.slice_bounds[k + 1]
        return [(self.region[t] + g0 * self.width[t], self.region[t] + g1 * self.width[t]) for t in ORDER]

    def owned(self, ranges):
        """What this rank's optimizer updates of ``ranges``: all of it (replicated) or its 1/world shard of every range
        (sharded; every range length is a multiple of 4 x world by construction)."""
        if self.exchange != "reduce_scatter":
            return list(ranges)
        w = max(self.world_size, 1)
        out = []
        for lo, hi in ranges:
            s = (hi - lo) // w
            out.append((lo + self.rank * s, lo + (self.rank + 1) * s))
        return out

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

    # ---- asynchronous exchange of one unit (a bucket or a slice) ---------------------------------------------------
    def collective_active(self) -> bool:
        return self.enable_collective and dist.is_initialized() and (self.world_size > 1 or self.force_collective)

    def _plain_collectives(self) -> bool:
        return dist.get_backend() != "nccl" and self.flat_grad.is_cuda

    # ---- which API the exchange goes through ------------------------------------------------------
    # "grouped": the process group's C++ entry points, one launch per unit, in place (each output view sits inside its
    #            input) -- fast on the host, but a private API whose in-place layout no test exercised with world > 1 on RCCL;
    # "public" : torch.distributed's documented calls, one per range (all_reduce / reduce_scatter_tensor /
    #            all_gather_into_tensor, async_op=True), same in-place views.
    # GS_DP_COLLECTIVES=public forces the second; otherwise the grouped path is used once `self_check()` has seen it
    # produce the closed-form result of a known pattern on THIS backend and world size (run on the first exchange with
    # peers); a mismatch or an exception switches to the public path (checked the same way) with a warning.
    def _api(self) -> str:
        api = getattr(self, "_collective_api", None)
        if api is None:
            api = self._collective_api = self._initial_api()
            if self.world_size > 1 and dist.is_initialized() and not self._plain_collectives():
                api = self.self_check()
        return api

    @staticmethod
    def _initial_api() -> str:
        import os

        return "public" if os.environ.get("GS_DP_COLLECTIVES", "").lower() == "public" else "grouped"

    def _issue_reduce(self, api, op, g, g_own):
        """-> work of the (all-)reduce of ranges `g` (reduce_scatter: result in `g_own`, a view inside each range)."""
        if api == "grouped":
            pg = dist.distributed_c10d._get_default_group()
            if self.exchange == "reduce_scatter":
                o = dist.ReduceScatterOptions()
                o.reduceOp = op
                return pg.reduce_scatter_tensor_coalesced(g_own, g, o)
            o = dist.AllreduceCoalescedOptions()
            o.reduceOp = op
            return pg.allreduce_coalesced(g, o)
        if self.exchange == "reduce_scatter":
            return _WorkList([dist.reduce_scatter_tensor(o_, i_, op=op, async_op=True) for o_, i_ in zip(g_own, g)])
        return _WorkList([dist.all_reduce(t, op=op, async_op=True) for t in g])

    def _issue_gather(self, api, p, p_own):
        if api == "grouped":
            return dist.distributed_c10d._get_default_group().allgather_into_tensor_coalesced(p, p_own)
        return _WorkList([dist.all_gather_into_tensor(o_, i_, async_op=True) for o_, i_ in zip(p, p_own)])

    def self_check(self) -> str:
        """One exchange of a known pattern through the selected API, compared with its closed form (exact: small integers
        in fp32).  Five ranges of different lengths, as a slice has; SUM; this rank's exchange mode; and the parameter
        all-gather of the reduce-scatter mode.  Costs one host synchronisation, once.  Returns the API in use."""
        w, r = self.world_size, self.rank
        dev = self.flat_grad.device
        lens = [4 * w * k for k in (4, 3, 3, 1, 27)]
        total = sum(lens)

        def pattern(rank_plus_1):
            return (torch.arange(total, device=dev, dtype=torch.float32) % 97 + 1) * rank_plus_1

        def run(api):
            buf, ranges, off = pattern(r + 1), [], 0
            for n_ in lens:
                ranges.append((off, off + n_))
                off += n_
            own = []
            for lo, hi in ranges:
                s_ = (hi - lo) // w
                own.append((lo + r * s_, lo + (r + 1) * s_))
            g = [buf[lo:hi] for lo, hi in ranges]
            g_own = [buf[lo:hi] for lo, hi in own]
            self._issue_reduce(api, dist.ReduceOp.SUM, g, g_own).wait()
            want = pattern(w * (w + 1) // 2)
            touched = own if self.exchange == "reduce_scatter" else ranges
            ok = all(bool(torch.equal(buf[lo:hi], want[lo:hi])) for lo, hi in touched)
            if self.exchange == "reduce_scatter":
                par = torch.zeros(total, device=dev)
                for lo, hi in own:
                    par[lo:hi] = pattern(r + 1)[lo:hi]
                self._issue_gather(api, [par[lo:hi] for lo, hi in ranges], [par[lo:hi] for lo, hi in own]).wait()
                want_p = torch.empty(total, device=dev)
                for lo, hi in ranges:
                    s_ = (hi - lo) // w
                    for q in range(w):
                        want_p[lo + q * s_:lo + (q + 1) * s_] = pattern(q + 1)[lo + q * s_:lo + (q + 1) * s_]
                ok = ok and bool(torch.equal(par, want_p))
            return ok

        def agreed(ok):  # every rank must take the same path: one MIN over the ranks' verdicts
            t = torch.tensor([1.0 if ok else 0.0], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item() > 0.5)

        api = getattr(self, "_collective_api", None) or self._initial_api()
        if api == "grouped":
            try:
                ok = run("grouped")
            except Exception as e:  # noqa: BLE001 -- a private API that moved or refuses the in-place views
                import warnings

                warnings.warn(f"gs_dp: grouped collectives unavailable ({e!r}); using torch.distributed's public API")
                ok = False
            if agreed(ok):
                self._collective_api = "grouped"
                return "grouped"
            import warnings

            warnings.warn("gs_dp: the grouped in-place collectives did not reproduce the known pattern on this backend; "
                          "falling back to torch.distributed's public API (one call per range)")
        if not agreed(run("public")):
            raise RuntimeError("gs_dp: the gradient exchange does not reproduce a known pattern on this backend "
                               f"({dist.get_backend()}, world {w}, {self.exchange}): refusing to train on it")
        self._collective_api = "public"
        return "public"

    # The collectives go straight to the process group's C++ entry points (allreduce_coalesced & co.: several ranges = ONE
    # grouped RCCL launch) with tensor views that are built once per unit: torch.distributed's Python wrappers cost
    # ~80 us per grouped call on the host (21 us this way, measured on gloo), and a step issues two or three per slice.
    def _views(self, key, ranges):
        v = self._view_cache.get(key)
        if v is None:
            own = self.owned(ranges)
            v = {"ranges": list(ranges), "own": own,
                 "g": [self.flat_grad[lo:hi] for lo, hi in ranges], "g_own": [self.flat_grad[lo:hi] for lo, hi in own],
                 "p": [self.flat_param[lo:hi] for lo, hi in ranges], "p_own": [self.flat_param[lo:hi] for lo, hi in own]}
            self._view_cache[key] = v
        return v

    def begin_exchange(self, key, ranges):
        """Start the mean all-reduce (or reduce-scatter) of the element ranges of one unit without waiting for it.  The
        collective runs on the process group's own stream, which first waits for everything enqueued on the current
        stream so far -- i.e. for the kernel that wrote the ranges."""
        if not self.collective_active():
            return
        # mean: inside the collective (RCCL: AVG), by a scaling pass in finish_exchange (gloo: SUM only), or left to the
        # optimizer (mean_in_optimizer: SUM, nothing else)
        avg = dist.get_backend() == "nccl" and not self.mean_in_optimizer
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        v = self._views(key, ranges)
        if self._plain_collectives():
            # a backend without grouped / scattering collectives on device tensors (gloo with HIP tensors: the two-rank
            # test on ONE GPU, tests/test_gpu_train.py): one all-reduce per range; the rank's shard of a reduce-scatter
            # is simply its part of the all-reduced range
            works = [dist.all_reduce(t, op=op, async_op=True) for t in v["g"]]
            self._pending[key] = (_WorkList(works), avg, v["g_own"] if self.exchange == "reduce_scatter" else v["g"])
            return
        # in place: a reduce-scatter's output is this rank's shard of its input (RCCL's in-place reduce-scatter layout)
        work = self._issue_reduce(self._api(), op, v["g"], v["g_own"])
        self._pending[key] = (work, avg, v["g_own"] if self.exchange == "reduce_scatter" else v["g"])

    def finish_exchange(self, key):
        """Make the current stream wait for the unit's reduction (and scale it on backends without AVG)."""
        pend = self._pending.pop(key, None)
        if pend is None:
            return
        work, avg, touched = pend
        work.wait()
        if not avg and not self.mean_in_optimizer:
            for t in touched:
                t.mul_(1.0 / self.world_size)

    def begin_param_gather(self, key, ranges):
        """reduce-scatter mode: all-gather the unit's updated PARAMETER shards (in place), without waiting."""
        if not self.collective_active() or self.exchange != "reduce_scatter":
            return
        v = self._views(key, ranges)
        if self._plain_collectives():
            # all-gather as a SUM all-reduce of a copy in which everything but this rank's shard is zero (exact: the
            # other ranks contribute zeros there)
            works, copies = [], []
            for full, own, (lo, hi), (slo, shi) in zip(v["p"], v["p_own"], v["ranges"], v["own"]):
                tmp = torch.zeros_like(full)
                tmp[slo - lo:shi - lo].copy_(own)
                works.append(dist.all_reduce(tmp, op=dist.ReduceOp.SUM, async_op=True))
                copies.append((full, tmp))
            self._pending_gather[key] = _WorkList(works, copies)
            return
        self._pending_gather[key] = self._issue_gather(self._api(), v["p"], v["p_own"])

    def finish_gather(self, name=None):
        """Make the current stream wait for the parameter all-gather(s): before anything reads the parameters."""
        for k in ([name] if name is not None else list(self._pending_gather)):
            work = self._pending_gather.pop(k, None)
            if work is not None:
                work.wait()

    # buckets (round 2 API: one contiguous range per unit)
    def begin_bucket(self, name: str):
        self.begin_exchange(name, [self.bucket_ranges[name]])

    def finish_bucket(self, name: str):
        self.finish_exchange(name)

    def begin_gather(self, name: str):
        self.begin_param_gather(name, [self.bucket_ranges[name]])

    # slices (round 4)
    def begin_slice(self, k: int):
        self.begin_exchange(("slice", k), self.slice_ranges(k))

    def finish_slice(self, k: int):
        self.finish_exchange(("slice", k))

    def begin_slice_gather(self, k: int):
        self.begin_param_gather(("slice", k), self.slice_ranges(k))

    def finish_slice_gather(self, k: int):
        self.finish_gather(("slice", k))

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

    def update_range(self, pos_grad: torch.Tensor, g0: int, g1: int):
        """``update`` for the Gaussians [g0, g1) only (the exchange slices of gs_train.Trainer; the visibility counter of
        the "mean" statistic is added once per view with ``add_seen``)."""
        if pos_grad.device.type != "cuda":
            raise RuntimeError("ViewParallelGradStat.update_range needs a HIP device; there is no CPU fallback")
        if tuple(pos_grad.shape) != tuple(self.accum.shape) or not pos_grad.is_contiguous():
            raise RuntimeError(f"pos_grad must be contiguous {tuple(self.accum.shape)}")
        if g1 <= g0:
            return
        from gaussian import _lib

        _lib.check(_lib.gs_grad_stat_update(pos_grad.data_ptr() + 12 * g0, self.accum.data_ptr() + 12 * g0, 3 * (g1 - g0),
                                            1 if self.mode == "max" else 2, torch.cuda.current_stream().cuda_stream),
                   "gs_grad_stat_update")

    def add_seen(self, seen: torch.Tensor):
        """"mean" statistic: the view's culling mask as float (train.py:152-154)."""
        if self.mode == "mean":
            self.counter.add_(seen)

    def reduce(self):
        """Combine the ranks' statistics in place (no-op for a single process); returns (accum [N,3], counter [N])."""
        if dist.is_initialized() and self.world_size > 1:
            dist.all_reduce(self._buf, op=dist.ReduceOp.MAX if self.mode == "max" else dist.ReduceOp.SUM)
        return self.accum, self.counter
