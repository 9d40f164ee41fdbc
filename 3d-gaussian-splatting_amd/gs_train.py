"""Training step around the frame path: image loss, fused Adam, the reference's LR schedule.

Mirrors ``Trainer.__init__`` / ``Trainer.train_step`` of the reference (train.py:16-67, 84-200) for the
part that runs every iteration:

    forward -> loss = (1-w) L1 + w (1-SSIM) -> backward -> [all-reduce] -> Adam step -> LR update

with three differences that matter on MI355X:
  * the loss and its gradient are one streaming HIP kernel + a one-block reduction (``gs_loss_l1_ssim``) instead of ~40 torch kernels
    over five padded copies of the image (torchmetrics SSIM + autograd);
  * Adam is ONE launch over the flat parameter bucket (``gs_adam_step``), which also accumulates the
    densification statistic ``accum_max_grad`` of train.py:145-154 while it reads the gradient;
  * nothing in the step synchronises with the host: the reference calls ``.item()`` three times per
    iteration (train.py:119-121); here the losses stay on the device until somebody reads them.
Densification (``Trainer(..., densify=True)``: ``gs_densify.adaptive_control`` / ``reset_opa`` on the reference's
schedule) is SURVEY 8f-2; data loading and the viewer stay out of scope (8f-3, 8f-4).
"""
from __future__ import annotations

import ctypes as C
import os
import math
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import torch

from gaussian import _lib
from gs_dp import ORDER, FlatGaussianParams, ViewParallelGradStat
from gs_frame import FrameRenderer

GROUPS = ("opa", "rgb", "pos", "scale", "quat")  # the reference's param-group order (train.py:59-65)


@dataclass
class TrainOptions:
    """Defaults of the reference's argparse (train.py:297-347) for the options the step uses."""
    lr: float = 0.003
    lr_factor_for_scale: float = 1.0
    lr_factor_for_rgb: float = 10.0
    lr_factor_for_opa: float = 10.0
    lr_factor_for_quat: float = 1.0
    lr_decay: str = "exp"  # "none" | "official" | "exp"
    n_iters: int = 7001
    n_iters_warmup: int = 300
    ssim_weight: float = 0.1
    scale_reg: float = 0.0  # train.py:108-109: + scale_reg * mean |scale|
    opa_reg: float = 0.0    # train.py:110-112: + opa_reg * mean sigma(opa) (1 - sigma(opa))
    grad_accum_method: str = "max"  # "max" | "mean"
    betas: tuple = (0.9, 0.99)
    eps: float = 1e-8
    # densification (train.py:302, 321-324, 340-347); off unless Trainer(..., densify=True)
    n_adaptive_control: int = 100
    adaptive_control_start_iter: int = 600  # the literal `i_iter > 600` of train.py:88-90
    adaptive_control_end_iter: int = 1000000000
    grad_accum_iters: int = 50
    n_opa_reset: int = 10000000
    reset_interval: int = 500
    split_thresh: float = 0.05
    delete_thresh: float = 1.5
    grad_thresh: float = 0.0002
    grad_aggregation: str = "max"
    use_clone: int = 0
    use_split: int = 1
    clone_dt: float = 0.01


def lr_lambdas(opt: TrainOptions) -> List[Callable[[int], float]]:
    """The five per-group multipliers of train.py:29-58, in GROUPS order."""
    w = opt.n_iters_warmup
    warm = lambda i: i / w  # noqa: E731  (i <= warmup)
    if opt.lr_decay == "none":
        f = lambda i: warm(i) if i <= w else 0.2 ** ((i - w) // 2000)  # noqa: E731
        return [f] * 5
    gamma = 0.01 ** (1 / (opt.n_iters - w))
    decay = lambda i: warm(i) if i <= w else gamma ** (i - w)  # noqa: E731
    flat = lambda i: warm(i) if i <= w else 1  # noqa: E731
    if opt.lr_decay == "official":
        return [decay, flat, decay, flat, flat]
    if opt.lr_decay != "exp":
        raise ValueError(f"lr_decay must be none|official|exp, got {opt.lr_decay!r}")
    return [decay] * 5


def base_lrs(opt: TrainOptions) -> List[float]:
    """train.py:20-25, in GROUPS order."""
    return [opt.lr * opt.lr_factor_for_opa, opt.lr * opt.lr_factor_for_rgb, opt.lr * 1,
            opt.lr * opt.lr_factor_for_scale, opt.lr * opt.lr_factor_for_quat]


class FusedAdam:
    """``torch.optim.Adam`` over a FlatGaussianParams buffer: ONE HIP launch per step on a single GPU, one launch per
    exchange slice (five element ranges each, ``gs_adam_step_multi``) under view parallelism.

    With ``flat.exchange == "reduce_scatter"`` the optimizer is SHARDED: this rank keeps the moments of its 1/world shard
    of every slice range only (densely packed, slice by slice) and updates those shards only (gs_dp.py); otherwise every
    rank updates everything and the moments mirror the flat buffer."""

    def __init__(self, flat: FlatGaussianParams, lrs: Sequence[float], betas=(0.9, 0.99), eps: float = 1e-8,
                 grad_stat: Optional[str] = None):
        if flat.flat_param.device.type != "cuda":
            raise RuntimeError("FusedAdam needs a HIP device; there is no CPU fallback")
        self.flat = flat
        self.betas, self.eps = (float(betas[0]), float(betas[1])), float(eps)
        self.sharded = flat.exchange == "reduce_scatter"
        dev = flat.flat_param.device
        size = self._build_units()
        self.exp_avg = torch.zeros(size, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        self.step_count = 0
        # buffer storage order (gs_dp.ORDER) -> group boundaries; lrs arrive in the reference's GROUPS order
        ends = list(flat.group_ends)
        self._ends = (C.c_int64 * len(ends))(*ends)
        self._lr = (C.c_float * len(ends))()
        self.set_lrs(lrs)
        self._stat_range = flat.offsets["pos"]
        self.stat_mode = {None: 0, "max": 1, "mean": 2}[grad_stat]
        self.accum_grad = torch.zeros_like(flat.params[0]) if self.stat_mode else None  # train.py:80-82
        self.skip_flag = None  # device address of a 64-bit counter: non-zero => the step is skipped (gs_abi.h)

    def _build_units(self) -> int:
        """Per exchange slice: the (up to five) element ranges this rank updates and where their moments start -> size of
        the moment arrays."""
        flat = self.flat
        self._units = []
        off = 0
        for k in range(flat.n_slices):
            own = [r for r in flat.owned(flat.slice_ranges(k)) if r[1] > r[0]]
            offs = []
            for lo, hi in own:
                offs.append(off if self.sharded else lo)  # replicated: the moments mirror the flat buffer
                off += hi - lo
            n_r = len(own)
            self._units.append((n_r, (C.c_int64 * n_r)(*[r[0] for r in own]), (C.c_int64 * n_r)(*[r[1] for r in own]),
                                (C.c_int64 * n_r)(*offs)))
        return off if self.sharded else flat.flat_param.numel()

    def reslice(self):
        """After ``flat.set_slices``: a replicated optimizer's moments mirror the flat buffer, so only the unit tables
        change; a sharded one packs its moments by slice and cannot follow."""
        if self.sharded:
            raise RuntimeError("a sharded optimizer cannot be re-sliced (its moments are packed slice by slice)")
        self._build_units()

    @property
    def state_bytes(self) -> int:
        """Optimizer state this rank keeps (both moments)."""
        return self.exp_avg.numel() * 8

    def set_lrs(self, lrs: Sequence[float]):
        by_group = dict(zip(GROUPS, lrs))
        for i, k in enumerate(ORDER):
            self._lr[i] = float(by_group[k])

    def clear_grad_stat(self):
        if self.accum_grad is not None:
            self.accum_grad.zero_()

    def step(self, advance: bool = True):
        """One Adam step over everything this rank owns (the step counter advances once per optimizer step).  Note: a
        step the device skips because its frame overflowed (``skip_flag``) still counts for the bias corrections --
        the host cannot know without synchronising."""
        if advance:
            self.step_count += 1
        if self.sharded:
            for k in range(len(self._units)):
                self.step_slice(k, advance=False)
            return
        f = self.flat
        b, e = self._stat_range
        n = f.flat_param.numel()
        _lib.check(_lib.gs_adam_step_sharded(
            f.flat_param.data_ptr(), f.flat_grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), n, 0, n,
            0, len(ORDER), self._ends, self._lr, self.betas[0], self.betas[1], self.eps, self.step_count,
            self.accum_grad.data_ptr() if self.accum_grad is not None else None, b, e, self.stat_mode, self.skip_flag,
            torch.cuda.current_stream().cuda_stream), "gs_adam_step")

    def fused_descriptor(self, advance: bool = True):
        """The ``gs_adam_fused`` argument of ``FrameRenderer.backward_adam`` for ONE optimizer step over everything
        (replicated optimizer: the moments mirror the flat buffer).  Advances the step counter like ``step``."""
        if self.sharded:
            raise RuntimeError("the fused backward + Adam step needs a replicated optimizer")
        if advance:
            self.step_count += 1
        f = self.flat
        d = _lib.GsAdamFused()
        lr_of = dict(zip(ORDER, self._lr))
        for i, name in enumerate(("pos", "quat", "scale", "opa", "rgb")):
            off = f.offsets[name][0] * 4
            d.exp_avg[i] = self.exp_avg.data_ptr() + off
            d.exp_avg_sq[i] = self.exp_avg_sq.data_ptr() + off
            d.lr[i] = lr_of[name]
        d.beta1, d.beta2, d.eps, d.step = self.betas[0], self.betas[1], self.eps, self.step_count
        d.grad_stat = self.accum_grad.data_ptr() if self.accum_grad is not None else None
        d.stat_mode = self.stat_mode
        d.skip_if_nonzero = self.skip_flag
        return d

    def step_slice(self, k: int, advance: bool = False, grad_scale: float = 1.0):
        """The same step for exchange slice ``k`` only (pass ``advance=True`` for the first slice of a step).
        ``grad_scale``: the gradient enters as grad * grad_scale (1 / world after a SUM exchange: gs_dp.py)."""
        if advance:
            self.step_count += 1
        n_r, lo, hi, off = self._units[k]
        if n_r == 0:
            return
        f = self.flat
        b, e = self._stat_range
        _lib.check(_lib.gs_adam_step_multi(
            f.flat_param.data_ptr(), f.flat_grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
            f.flat_param.numel(), n_r, lo, hi, off, len(ORDER), self._ends, self._lr, self.betas[0], self.betas[1],
            self.eps, self.step_count, self.accum_grad.data_ptr() if self.accum_grad is not None else None, b, e,
            self.stat_mode, self.skip_flag, float(grad_scale), torch.cuda.current_stream().cuda_stream),
            "gs_adam_step_multi")


class ImageLoss:
    """``(1-w) * L1 + w * (1 - SSIM)`` and its gradient w.r.t. the rendered image (train.py:99-107)."""

    def __init__(self, height: int, width: int, ssim_weight: float = 0.1, device="cuda"):
        self.H, self.W, self.w = int(height), int(width), float(ssim_weight)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("ImageLoss needs a HIP device; there is no CPU fallback")
        nbytes = _lib.gs_loss_workspace_bytes(self.H, self.W)
        self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        self.grad = torch.empty(self.H, self.W, 3, dtype=torch.float32, device=self.device)
        self.values = torch.zeros(3, dtype=torch.float32, device=self.device)  # (loss, l1, ssim), stays on device

    def __call__(self, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        for name, t in (("pred", pred), ("target", target)):
            if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or tuple(t.shape) != (self.H, self.W, 3):
                raise RuntimeError(f"{name} must be a contiguous float32 HIP tensor of shape [{self.H},{self.W},3]")
        _lib.check(_lib.gs_loss_l1_ssim(pred.data_ptr(), target.data_ptr(), self.H, self.W, self.w,
                                        self.grad.data_ptr(), self.values.data_ptr(), self._ws.data_ptr(),
                                        self._ws.numel(), torch.cuda.current_stream().cuda_stream), "gs_loss_l1_ssim")
        return self.grad


class Trainer:
    """One view per step on this rank; gradients are averaged over ranks when torch.distributed is up.

    NOTE for callers that read gradients: with ``fuse_adam`` active (the default on one rank, rgb or SH colours) the
    optimizer step runs inside the backward's last kernel and ``flat.grads`` is NOT written -- it holds whatever an
    earlier, unfused step left there.  Pass ``fuse_adam=False`` to get the gradients of every step."""

    def __init__(self, params: Sequence[torch.Tensor], cameras, targets: Sequence[torch.Tensor],
                 opt: Optional[TrainOptions] = None, world_size: int = 1, max_pairs: int = 1 << 20,
                 scale_activation: str = "abs", densify: bool = False, generator: Optional[torch.Generator] = None,
                 per_view_stat: Optional[bool] = None, exchange: str = "all_reduce", n_slices: Optional[int] = None,
                 bwd_rows: Optional[bool] = None, fuse_adam: Optional[bool] = None):
        self.opt = opt or TrainOptions()
        # `fuse_adam`: apply the Adam step inside the backward's last kernel (FrameRenderer.backward_adam: no gradient buffer is
        # written, bit-identical parameters) where the step allows it -- one rank, no regulariser that edits the
        # gradient, the densification statistic fused into the optimizer.  None: on (GS_TRAIN_FUSE_ADAM=0 turns it off for
        # A/B runs of bench.py); False: never (a caller that reads flat.grads behind a step).  Same-box A/B, round 5
        # (profiles/r05_r_*): 775 against 745 it/s on the moving 2.4 M-Gaussian scene, 933 against 894 fixed, 1786 against
        # 1738 at 376 k -- one 208 us kernel where the projection backward (113 us) and gs_adam_step (150 us) ran.  (The
        # first version -- every thread its own 14 parameters, 84 four-byte accesses at a 12-byte stride -- was 15 % SLOWER
        # than the two kernels, r05_q; the float4 walk through an LDS hand-over is what made it pay.)
        if fuse_adam is None:
            fuse_adam = os.environ.get("GS_TRAIN_FUSE_ADAM", "") != "0"
        self.fuse_adam = bool(fuse_adam)
        self.world_size = int(world_size)
        self.n_slices = n_slices  # exchange slices of the Gaussian array (gs_dp.py; None: by scene size)
        if exchange == "reduce_scatter" and self.world_size > 1:
            import torch.distributed as dist

            if not dist.is_initialized():
                # every rank would keep and update its own 1/world of the optimizer state while nobody exchanges
                # anything: (world - 1)/world of the parameters would silently never train
                raise RuntimeError("exchange='reduce_scatter' with world_size > 1 needs an initialised process group")
        # gradient exchange under view parallelism (gs_dp.py): "all_reduce" + replicated Adam, or "reduce_scatter" +
        # sharded Adam + all-gather of the parameters
        self.exchange = exchange
        # densification statistic: fused into the Adam launch on one GPU; with several ranks it must be taken from
        # each rank's own gradient before the all-reduce (gs_dp.ViewParallelGradStat).  `per_view_stat=True` forces
        # that path on a single rank too (it then gives bit-identical results; used by the tests).
        self.per_view_stat = (self.world_size > 1) if per_view_stat is None else bool(per_view_stat)
        self.scale_activation = scale_activation
        self.densify, self.generator = bool(densify), generator
        self.cameras, self.targets = list(cameras), list(targets)
        dev = params[0].device
        # "async": in steady state the pair capacity is checked from a pinned-memory copy one frame late -- no host
        # synchronisation in the training loop (the reference synchronises >= 8 times per forward).  The FIRST frame
        # of every view on every Gaussian set (start of training, after each densification) is checked synchronously
        # and redone in a larger workspace if it did not fit, so a step is never taken on an empty frame because the
        # capacity was simply unknown; after that a view's pair count only drifts, and the workspace grows ahead of
        # it (25 % head room).  Should a frame overflow all the same, `renderer.overflowed_frames` counts it and
        # train_step warns once the late counters arrive.
        # `bwd_rows`: which kernel composites the rgb backward (FrameRenderer; None: by the scene's share of saturated
        # buckets, asked for ONCE per Gaussian set -- behind its first step -- so that a run takes the same kernels, the
        # same bits, every time).
        self.renderer = FrameRenderer(dev, max_pairs=max_pairs, training=True, scale_activation=scale_activation,
                                      auto_grow="async", bwd_rows=bwd_rows)
        self._backward_choice_due = True
        self._views_checked = set()
        self._overflow_warned = 0
        self._lambdas, self._base = lr_lambdas(self.opt), base_lrs(self.opt)
        self._loss = {}
        self._bind(params, 0)

    def _bind(self, params: Sequence[torch.Tensor], i_iter: int):
        """(Re)creates the flat bucket and the optimizer for a (new) Gaussian set: train.py:59-67 / :169-179 --
        the reference also starts a fresh torch.optim.Adam after every adaptive_control."""
        force = getattr(getattr(self, "flat", None), "force_collective", False)
        self.flat = FlatGaussianParams(params, world_size=self.world_size, exchange=self.exchange,
                                       force_collective=force, n_slices=self.n_slices)
        self.flat.mean_in_optimizer = True  # the exchange sums, step_slice applies 1 / world (gs_dp.py)
        self.renderer.forward_abandon()  # a frame projected ahead belonged to the old Gaussian set
        split_stat = self.densify and self.per_view_stat
        self.optimizer = FusedAdam(self.flat, [b * f(i_iter) for b, f in zip(self._base, self._lambdas)],
                                   betas=self.opt.betas, eps=self.opt.eps,
                                   grad_stat=None if split_stat else self.opt.grad_accum_method)
        self.view_stat = (ViewParallelGradStat(self.flat.params[0].shape[0], self.flat.flat_param.device,
                                               self.opt.grad_accum_method, self.world_size) if split_stat else None)
        self.grad_counter = None  # "mean" accumulation only: per-Gaussian count of views that saw it (train.py:150)
        self._views_checked = set()  # a new Gaussian set: every view's first frame is capacity-checked again
        self._backward_choice_due = True  # ... and the rgb backward kernel is chosen again, behind its first step

    @property
    def n_gaussians(self) -> int:
        return int(self.flat.params[0].shape[0])

    def _loss_for(self, h: int, w: int) -> ImageLoss:
        key = (h, w)
        if key not in self._loss:
            self._loss[key] = ImageLoss(h, w, self.opt.ssim_weight, self.flat.flat_param.device)
        return self._loss[key]

    def _can_fuse_adam(self) -> bool:
        o, rgb = self.opt, self.flat.params[4]
        return (self.fuse_adam and not self.flat.collective_active() and not self.optimizer.sharded
                and self.view_stat is None and o.scale_reg == 0 and o.opa_reg == 0 and rgb.dim() == 2
                and rgb.shape[1] in (3, 27, 48))

    def _is_control_iteration(self, i_iter: int) -> bool:
        """Does train_step(i_iter) run adaptive_control (prune, or prune + densify)?  train.py:86-91."""
        o = self.opt
        return i_iter > o.adaptive_control_start_iter and i_iter % o.n_adaptive_control == 0

    def train_step(self, i_iter: int, camera_id: int, next_camera_id: Optional[int] = None) -> torch.Tensor:
        """Returns the device tensor (loss, l1, ssim) of this step (no host synchronisation).

        ``next_camera_id``: the view the NEXT step will render, if the caller knows it (a fixed view per rank, a
        pre-drawn camera schedule).  Under view parallelism the project stage of that frame is then issued slice by slice
        behind this step's optimizer, underneath the gradient exchange of the remaining slices (gs_dp.py); the next
        ``train_step`` picks the frame up if it is called with that camera, and renders from scratch otherwise."""
        o = self.opt
        # schedule flags, train.py:86-91
        in_reset = i_iter >= o.n_opa_reset and i_iter % o.n_opa_reset < o.reset_interval
        past = i_iter > o.adaptive_control_start_iter
        only_delete = past and i_iter % o.n_adaptive_control == 0
        control = only_delete and i_iter < o.adaptive_control_end_iter
        accum_start = past and (i_iter + o.grad_accum_iters - 1) % o.n_adaptive_control == 0
        cam, target = self.cameras[camera_id], self.targets[camera_id]
        flat, r = self.flat, self.renderer
        flat.finish_gather()  # reduce-scatter mode: parameter all-gathers of the previous step that nobody waited for yet
        if r.begun_frame_matches(*flat.params, cam):
            image, _ = r.forward_finish()  # projected behind the previous step's optimizer
        else:
            r.forward_abandon()
            if camera_id not in self._views_checked:  # first frame of this view on this Gaussian set: synchronous check
                self._views_checked.add(camera_id)
                r._checked_once = False
            image, _ = r.forward(*flat.params, cam)
        # a frame that overflowed its workspace all the same was rendered empty: on a single rank the optimizer step
        # is skipped ON THE DEVICE (the fused Adam looks at the frame's overflow counter; no host synchronisation).
        # With several ranks the step is taken -- the other ranks' views still carry gradient, and skipping on one rank
        # would let the replicas drift apart -- and the warning below reports it once the counters arrive.
        self.optimizer.skip_flag = r.overflow_flag() if self.world_size == 1 else None
        if r.overflowed_frames > self._overflow_warned:
            import warnings

            warnings.warn(f"{r.overflowed_frames - self._overflow_warned} training frame(s) exceeded the "
                          f"pair capacity and were rendered empty (single rank: their optimizer steps were skipped on "
                          f"the device; several ranks: this rank contributed zero gradients to them); the "
                          f"workspace has been enlarged to {r.max_pairs} pairs")
            self._overflow_warned = r.overflowed_frames
        loss = self._loss_for(image.shape[0], image.shape[1])
        grad_image = loss(image, target)
        if self.densify and accum_start:  # train.py:141-142 (before this step's gradient is accumulated)
            self.optimizer.clear_grad_stat()
            self.grad_counter = None
            if self.view_stat is not None:
                self.view_stat.clear()
        seen = None
        if self.densify and o.grad_accum_method == "mean":
            seen = r.culling_mask().to(torch.float32)

        def local_terms(g0, g1):
            """What needs this rank's OWN gradients of the Gaussians [g0, g1) -- regularisers, the per-view densification
            statistic -- before they are averaged over the ranks."""
            if o.scale_reg > 0:  # train.py:108-109
                sc = flat.params[2]
                flat.grads[2][g0:g1].add_(torch.sign(sc[g0:g1]), alpha=o.scale_reg / sc.numel())
            if self.view_stat is not None:
                self.view_stat.update_range(flat.grads[0], g0, g1)
            if o.opa_reg > 0:  # train.py:110-112
                sg = torch.sigmoid(flat.params[3][g0:g1])
                flat.grads[3][g0:g1].add_(sg * (1 - sg) * (1 - 2 * sg), alpha=o.opa_reg / flat.params[3].numel())

        def settle_backward_choice():
            # One host synchronisation per Gaussian set (the set's first frame was capacity-checked synchronously a moment
            # ago anyway): the backward that has just been issued left the share of its buckets that belong to saturated
            # tiles in the frame's counters; the renderer keeps the rgb backward kernel that share asks for until the next
            # Gaussian set (after a densification the scene is a different one).  Called while THIS frame's descriptor and
            # counters are still the workspace's current ones -- in the view-parallel path that is before the next frame's
            # project stage is issued ahead (behind it the counters may belong to the half-issued frame).
            if self._backward_choice_due:
                self._backward_choice_due = False
                if r.bwd_rows is None and self.flat.params[4].dim() == 2 and self.flat.params[4].shape[1] == 3:
                    r.stats()

        if flat.collective_active():
            # View parallelism.  Every gradient of the frame becomes final in the LAST kernel of the backward, the
            # per-Gaussian sum of the gradient rows; that kernel, the exchange, the optimizer and the NEXT frame's project
            # stage are all independent per Gaussian, so they run as a pipeline over K slices of the Gaussian array:
            #   main stream : rows | S_0 S_1 ... S_K-1 | A_0 P_0 | A_1 P_1 | ...   (S = sums, A = Adam, P = next project)
            #   RCCL stream :        X_0 X_1 ...                                    (X_k starts when S_k is done,
            #                                                                        A_k waits for X_k)
            # Nothing of a slice is applied before its exchange has finished: no stale gradients, the same numbers as
            # the blocking path bit for bit (tests/test_host_logic.py, tests/test_gpu_train.py).
            K = flat.n_slices
            r.backward(grad_image, out=flat.grads, part=_lib.GS_BWD_RASTER)
            settle_backward_choice()
            if self.view_stat is not None and seen is not None:
                self.view_stat.add_seen(seen)
            for k in range(K):
                g0, g1 = flat.slice_gaussians(k)
                r.backward_slice(flat.grads, g0, g1)
                local_terms(g0, g1)
                flat.begin_slice(k)
            ahead = self._can_project_ahead(i_iter, next_camera_id, control or only_delete)
            lag = 1 if flat.exchange == "reduce_scatter" else 0  # the project needs the GATHERED parameters of its slice
            for k in range(K + lag):
                if k < K:
                    flat.finish_slice(k)
                    self.optimizer.step_slice(k, advance=(k == 0), grad_scale=1.0 / max(self.world_size, 1))
                    flat.begin_slice_gather(k)  # reduce-scatter mode only: the updated shards travel underneath ...
                j = k - lag
                if ahead and j >= 0:
                    flat.finish_slice_gather(j)  # ... the next slice's reduce-scatter + Adam
                    ahead = self._project_ahead(j, next_camera_id)
            # (gathers nobody waited for are waited for where the parameters are read next: finish_gather)
        elif self._can_fuse_adam():
            # one kernel less and no gradient round trip through memory: the per-Gaussian sums, the projection / activation
            # backward and the Adam update of the Gaussian's 14 parameters (+ the |pos.grad| statistic) in one launch
            r.backward_adam(grad_image, self.optimizer.fused_descriptor())
        else:
            r.backward(grad_image, out=flat.grads)
            local_terms(0, flat.n)
            if self.view_stat is not None and seen is not None:
                self.view_stat.add_seen(seen)
            self.optimizer.step()  # also: accum_max_grad = max(|pos.grad|, accum) or += |pos.grad| (train.py:144-153)
        if seen is not None and self.view_stat is None:
            self.grad_counter = seen if self.grad_counter is None else self.grad_counter + seen
        settle_backward_choice()
        if self.densify and (control or only_delete):
            self.adaptive_control(i_iter, densify=control and not in_reset)
        # train.py:184-185: the learning rates of the NEXT step
        self.optimizer.set_lrs([f(i_iter) * b for f, b in zip(self._lambdas, self._base)])
        if self.densify and i_iter % o.n_opa_reset == 0 and i_iter > 0:  # train.py:189-190
            from gs_densify import reset_opa

            self.flat.finish_gather()
            self.renderer.forward_abandon()
            reset_opa(self.flat.params[3])
        return loss.values

    def tune_slices(self, i_iter: int, camera_id: int, candidates: Sequence[int] = (1, 2, 4), iters: int = 6,
                    next_camera_id: Optional[int] = None) -> int:
        """Pick the number of exchange slices by measurement: ``iters`` real training steps (starting at iteration
        ``i_iter``, all on view ``camera_id``) per candidate, timed with a device synchronisation around each group, the
        slowest rank's time decides (one MAX all-reduce per candidate, so every rank picks the same).  How much wire time
        a further slice hides against what it costs depends on the links, and the builder of this
        code never had more than one GPU: the first steps of a multi-GPU run can settle it themselves.  Replicated
        optimizer ("all_reduce") only.  Returns the chosen count; ``candidates * iters`` steps have been taken."""
        import time

        import torch.distributed as dist

        flat = self.flat
        if not flat.collective_active() or flat.exchange != "all_reduce":
            return flat.n_slices
        o = self.opt
        total = len(candidates) * iters
        if self.densify:
            # a step that runs adaptive_control rebuilds self.flat and the optimizer (_bind), and an opacity
            # reset changes what a step costs -- neither belongs inside a timing comparison.  Refuse a window that holds
            # a control point; the caller tunes between two of them (they are n_adaptive_control iterations apart).
            for it_ in range(i_iter, i_iter + total):
                if self._is_control_iteration(it_) or (it_ % o.n_opa_reset == 0 and it_ > 0):
                    raise RuntimeError(f"tune_slices({i_iter}): iteration {it_} of the {total} tuning steps is a densification "
                                       "/ opacity-reset boundary; start right after one")
        dev = flat.flat_param.device
        best, best_t, it = flat.n_slices, None, i_iter
        for ns in candidates:
            flat = self.flat  # (re-read: nothing below may act on a stale object)
            flat.finish_gather()
            self.renderer.forward_abandon()
            flat.set_slices(ns)
            self.optimizer.reslice()
            self.train_step(it, camera_id, next_camera_id=next_camera_id)  # (first step after a re-slice: views, caches)
            it += 1
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(iters - 1):
                self.train_step(it, camera_id, next_camera_id=next_camera_id)
                it += 1
            torch.cuda.synchronize(dev)
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            if self.world_size > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            t = float(t.item())
            if best_t is None or t < best_t:
                best, best_t = flat.n_slices, t
        flat = self.flat
        flat.finish_gather()
        self.renderer.forward_abandon()
        flat.set_slices(best)
        self.optimizer.reslice()
        self.n_slices = best  # a later _bind (densification) keeps the choice
        return best

    def _can_project_ahead(self, i_iter: int, next_camera_id: Optional[int], rebinding: bool) -> bool:
        """May the next frame's project stage be issued behind this step's optimizer?  Only for a view whose capacity
        has been checked on this Gaussian set, and not across a densification / opacity reset (the parameters the
        project stage read would no longer be the ones the frame is rendered with)."""
        o = self.opt
        if next_camera_id is None or next_camera_id not in self._views_checked:
            return False
        if self.densify and (rebinding or (i_iter % o.n_opa_reset == 0 and i_iter > 0)):
            return False
        return True

    def _project_ahead(self, k: int, next_camera_id: int) -> bool:
        """Project stage of the next frame for the Gaussians of exchange slice ``k`` (whole project slices)."""
        flat = self.flat
        per = flat.project_slice
        b0, b1 = flat.slice_bounds[k], min(flat.slice_bounds[k + 1], flat.n)
        if b1 <= b0:
            return True
        ok = self.renderer.forward_begin(*flat.params, self.cameras[next_camera_id], b0 // per, -(-b1 // per),
                                         expect_per_slice=per)
        return ok

    def adaptive_control(self, i_iter: int, densify: bool = True):
        """train.py:156-180: prune (+ clone / split when ``densify``), then a fresh optimizer."""
        from gs_densify import adaptive_control

        o = self.opt
        self.flat.finish_gather()
        if self.view_stat is not None:  # combine the ranks' per-view statistics: one collective per boundary
            accum, cnt = self.view_stat.reduce()
            counter = 1.0 if o.grad_accum_method == "max" else cnt
        else:
            accum = self.optimizer.accum_grad
            counter = 1.0 if o.grad_accum_method == "max" or self.grad_counter is None else self.grad_counter
        stat = accum / (counter + 1e-3 if isinstance(counter, float) else (counter + 1e-3).unsqueeze(-1))  # train.py:160
        new, counts = adaptive_control(self.flat.params, stat.contiguous(), taus=o.split_thresh,
                                       delete_thresh=o.delete_thresh, scale_activation=self.scale_activation,
                                       grad_thresh=o.grad_thresh, grad_aggregation=o.grad_aggregation,
                                       use_clone=bool(o.use_clone) and densify, use_split=bool(o.use_split) and densify,
                                       clone_dt=o.clone_dt, generator=self.generator)
        self._bind(new, i_iter)
        # the decisions are identical on every rank (same statistic, same parameters), the Gaussian draws of the
        # split are not unless every rank seeds `generator` alike: rank 0's new set is the one that counts
        self.flat.broadcast_params(0)
        return counts

    # ------------------------------------------------------------------ evaluation / viewer hook
    @torch.no_grad()
    def test(self, camera_id, extrinsics=None, intrinsics=None) -> dict:
        """``Trainer.test`` of the reference (train.py:256-281), which is also what its viser GUI calls per
        frame (visergui.py:137-149): ``test(None, extrinsics={"rot", "tran"}, intrinsics={"width", "height",
        "focal_x", "focal_y"})`` renders an arbitrary world->camera pose at an arbitrary resolution (the tile
        grid is rebuilt, sizes need not be multiples of 16); ``test(camera_id)`` renders a training / test
        camera and adds ``psnr``, ``ssim`` and ``render_time`` (seconds, device time) against its target."""
        import numpy as np

        from gs_scene import Camera

        self.flat.finish_gather()
        if extrinsics is not None and intrinsics is not None:
            to_np = lambda a: (a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)).astype(np.float32)  # noqa: E731
            cam = Camera(int(intrinsics["width"]), int(intrinsics["height"]), float(intrinsics["focal_x"]),
                         float(intrinsics["focal_y"]), to_np(extrinsics["rot"]).reshape(3, 3),
                         to_np(extrinsics["tran"]).reshape(3), near=getattr(self.cameras[0], "near", 0.3) if self.cameras else 0.3)
        elif camera_id is not None:
            cam = self.cameras[camera_id]
        else:
            raise RuntimeError("test() needs a camera_id or extrinsics + intrinsics")
        if getattr(self, "_eval_renderer", None) is None:
            self._eval_renderer = FrameRenderer(self.flat.flat_param.device, max_pairs=self.renderer.max_pairs,
                                                training=False, scale_activation=self.scale_activation,
                                                thresh=self.renderer.thresh)
            self._eval_renderer.tile_culling_method = self.renderer.tile_culling_method
            self._eval_renderer.tile_culling_dist_thresh = self.renderer.tile_culling_dist_thresh
        tic, toc = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tic.record()
        image, _ = self._eval_renderer.forward(*self.flat.params, cam)
        toc.record()
        out = {"image": image}
        if camera_id is not None:
            target = self.targets[camera_id]
            toc.synchronize()
            out["render_time"] = tic.elapsed_time(toc) / 1000
            out["psnr"] = self.psnr(image, target)
            h, w = image.shape[:2]
            if h > 10 and w > 10:
                probe = ImageLoss(h, w, 1.0, image.device)  # weight 1: values = (1 - ssim, l1, ssim)
                probe(image, target)
                out["ssim"] = float(probe.values[2])
        return out

    # ------------------------------------------------------------------ checkpoints (train.py:283-291, splatter.py:417-424)
    def save_checkpoint(self, path: str):
        """The reference's ``ckpt.pth``: a dict of the five raw parameter tensors."""
        self.flat.finish_gather()
        pos, quat, scale, opa, rgb = (t.detach().clone() for t in self.flat.params)
        torch.save({"pos": pos, "opa": opa, "rgb": rgb, "quat": quat, "scale": scale}, path)

    def load_checkpoint(self, path: str, i_iter: int = 0):
        ck = torch.load(path, map_location=self.flat.flat_param.device)
        params = [ck[k].detach().to(torch.float32).contiguous() for k in ("pos", "quat", "scale", "opa", "rgb")]
        self._bind(params, i_iter)

    @staticmethod
    def psnr(image: torch.Tensor, target: torch.Tensor) -> float:
        """torchmetrics PeakSignalNoiseRatio() with its default data_range = max(target) - min(target)."""
        mse = torch.mean((image - target) ** 2).item()
        rng = (target.max() - target.min()).item()
        return 10.0 * math.log10(rng * rng / mse) if mse > 0 else float("inf")
