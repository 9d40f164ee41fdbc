"""GPU parity tests of the training-step kernels (gs_adam_step, gs_loss_l1_ssim) and of the Trainer loop.
All calls go through the C ABI (ctypes); the oracle is oracle/train_ref.py."""
import os

import numpy as np
import pytest
import torch

from gs_testutil import to_torch
from oracle import train_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def make_flat(gpu, n, seed=0, color_dim=3):
    from gs_dp import FlatGaussianParams

    rng = np.random.default_rng(seed)
    shapes = [(n, 3), (n, 4), (n, 3), (n,), (n, color_dim)]
    params = [torch.from_numpy(rng.normal(size=s).astype(np.float32)).to(gpu) for s in shapes]
    return FlatGaussianParams(params), rng


@pytest.mark.parametrize("n,stat", [(1, None), (37, "max"), (1000, "mean"), (4099, "max")])
def test_adam_matches_oracle_and_torch(gpu, n, stat):
    """Five groups with different learning rates, 6 steps (group boundaries inside a float4: next test)."""
    from gs_train import FusedAdam, GROUPS

    flat, rng = make_flat(gpu, n, seed=n)
    lrs = dict(zip(GROUPS, [0.03, 0.02, 0.003, 0.004, 0.005]))
    opt = FusedAdam(flat, [lrs[k] for k in GROUPS], betas=(0.9, 0.99), eps=1e-8, grad_stat=stat)
    names = ("pos", "quat", "scale", "opa", "rgb")
    ref_p = [p.clone().requires_grad_(True) for p in flat.params]
    ref = torch.optim.Adam([{"params": [p], "lr": lrs[k]} for p, k in zip(ref_p, names)], betas=(0.9, 0.99),
                           foreach=False)
    ora = [(p.cpu().numpy().copy(), np.zeros(p.shape, np.float32), np.zeros(p.shape, np.float32)) for p in flat.params]
    stat_ref = np.zeros((n, 3), np.float32)
    for step in range(1, 7):
        grads = [(rng.normal(size=tuple(p.shape)) * 10.0 ** rng.integers(-3, 1)).astype(np.float32) for p in flat.params]
        for gview, p, g in zip(flat.grads, ref_p, grads):
            gview.copy_(torch.from_numpy(g))
            p.grad = torch.from_numpy(g).to(gpu)
        opt.step()
        ref.step()
        ora = [train_ref.adam_step(p0, g, m, v, lrs[k], 0.9, 0.99, 1e-8, step)
               for (p0, m, v), g, k in zip(ora, grads, names)]
        stat_ref = np.maximum(stat_ref, np.abs(grads[0])) if stat == "max" else stat_ref + np.abs(grads[0])
    for p, pr, (p0, _, _) in zip(flat.params, ref_p, ora):
        tol = 1e-6 * max(1.0, float(np.abs(p0).max()))
        assert np.abs(p.cpu().numpy() - p0).max() <= tol                      # vs the oracle
        assert float((p - pr.detach()).abs().max()) <= tol                    # vs torch.optim.Adam on the GPU
    if stat:
        assert np.allclose(opt.accum_grad.cpu().numpy(), stat_ref, rtol=1e-6, atol=0)
    assert opt.step_count == 6


def test_adam_group_boundaries_inside_a_float4(gpu):
    """gs_adam_step with group boundaries that are no multiples of 4 (FlatGaussianParams pads its regions to multiples
    of 4 rows since round 4, so the trainer no longer produces them; the C ABI still takes any ascending table): the
    float4 lanes that straddle a boundary must pick their own group's learning rate.  Also a sub-range launch
    (gs_adam_step_range) with ragged ends and the multi-range launch against the one-launch result."""
    import ctypes as C

    from gaussian import _lib

    n = 1003
    rng = np.random.default_rng(5)
    ends_l, lrs_l = [7, 310, 311, 640, 1003], [0.03, 0.02, 0.003, 0.004, 0.005]
    ends, lr = (C.c_int64 * 5)(*ends_l), (C.c_float * 5)(*lrs_l)
    p0 = rng.normal(size=n).astype(np.float32)
    stream = torch.cuda.current_stream().cuda_stream

    def run(kind):
        p, m, v = (torch.from_numpy(p0.copy()).to(gpu), torch.zeros(n, device=gpu), torch.zeros(n, device=gpu))
        g_rng = np.random.default_rng(6)
        for step in range(1, 5):
            g = torch.from_numpy(g_rng.normal(size=n).astype(np.float32)).to(gpu)
            a = (p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n)
            tail = (5, ends, lr, 0.9, 0.99, 1e-8, step, None, 0, 0, 0)
            if kind == "one":
                _lib.check(_lib.gs_adam_step(*a, *tail, stream), "gs_adam_step")
            elif kind == "ranges":
                for lo, hi in ((0, 5), (5, 309), (309, 1003)):
                    _lib.check(_lib.gs_adam_step_range(*a, lo, hi, *tail, stream), "gs_adam_step_range")
            else:
                lo_t, hi_t = (C.c_int64 * 3)(0, 308, 312), (C.c_int64 * 3)(308, 312, 1003)
                _lib.check(_lib.gs_adam_step_multi(*a, 3, lo_t, hi_t, lo_t, *tail, None, 1.0, stream), "gs_adam_step_multi")
        return p.cpu().numpy(), m.cpu().numpy(), v.cpu().numpy()

    one = run("one")
    for kind in ("ranges", "multi"):
        for x, y in zip(one, run(kind)):
            assert np.array_equal(x, y)
    # against the oracle, group by group
    want = p0.copy()
    m = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    g_rng = np.random.default_rng(6)
    for step in range(1, 5):
        g = g_rng.normal(size=n).astype(np.float32)
        lo = 0
        for hi, l in zip(ends_l, lrs_l):
            want[lo:hi], m[lo:hi], v[lo:hi] = train_ref.adam_step(want[lo:hi], g[lo:hi], m[lo:hi], v[lo:hi], l, 0.9, 0.99,
                                                                  1e-8, step)
            lo = hi
    assert np.abs(one[0] - want).max() <= 1e-6 * max(1.0, float(np.abs(want).max()))
    # the multi-range launch validates its tables
    bad = (C.c_int64 * 1)(2)
    rc = _lib.gs_adam_step_multi(0, 0, 0, 0, n, 1, bad, (C.c_int64 * 1)(8), bad, 5, ends, lr, 0.9, 0.99, 1e-8, 1, None, 0, 0,
                                 0, None, 1.0, stream)
    assert rc == -1


def test_adam_rejects_bad_arguments(gpu):
    import ctypes as C

    from gaussian import _lib

    t = torch.zeros(16, device=gpu)
    ends, lr = (C.c_int64 * 1)(15), (C.c_float * 1)(0.1)  # groups do not cover [0, n)
    rc = _lib.gs_adam_step(t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), 16, 1, ends, lr, 0.9, 0.99, 1e-8, 1,
                           None, 0, 0, 0, None)
    assert rc == -1 and b"cover" in _lib.gs_last_error()
    ends = (C.c_int64 * 1)(16)
    rc = _lib.gs_adam_step(t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), 16, 1, ends, lr, 0.9, 0.99, 1e-8, 0,
                           None, 0, 0, 0, None)
    assert rc == -1 and b"step" in _lib.gs_last_error()


@pytest.mark.parametrize("h,w,weight", [(11, 11, 0.1), (37, 53, 0.1), (64, 64, 1.0), (200, 300, 0.25), (40, 33, 0.0),
                                        (5, 7, 0.0),
                                        # the streaming kernel's strips are 52 rows x 482 flat columns (3 W floats per
                                        # row): ragged last strip / column block, exactly one, one more than one
                                        (53, 483, 0.3), (105, 161, 0.1), (52, 160, 0.5), (104, 322, 0.2), (12, 700, 1.0)])
def test_loss_matches_oracle(gpu, h, w, weight):
    from gs_train import ImageLoss

    rng = np.random.default_rng(h * 1000 + w)
    x = rng.uniform(0, 1, (h, w, 3)).astype(np.float32)
    y = np.clip(x + rng.normal(0, 0.1, x.shape), 0, 1).astype(np.float32)
    y[::7, ::5] = x[::7, ::5]  # exact ties: sign(0) = 0
    loss = ImageLoss(h, w, weight, gpu)
    g = loss(torch.from_numpy(x).to(gpu), torch.from_numpy(y).to(gpu))
    lo, l1, ssim, grad = train_ref.l1_ssim_loss(x, y, weight)
    vals = loss.values.cpu().numpy()
    assert abs(vals[1] - l1) < 1e-6 and abs(vals[2] - ssim) < 2e-6 and abs(vals[0] - lo) < 2e-6
    err = np.abs(g.cpu().numpy() - grad).max()
    assert err < 2e-5 * np.abs(grad).max(), (err, np.abs(grad).max())  # fp32 variance cancellation, E[x^2] - mu^2


def test_loss_full_hd(gpu):
    """1080p: the gradient of the loss against the fp64 oracle, and against finite differences of it."""
    from gs_train import ImageLoss

    rng = np.random.default_rng(3)
    base = rng.uniform(0, 1, (68, 120, 3))
    x = np.kron(base, np.ones((16, 16, 1)))[:1080, :1920].astype(np.float32)  # blocky image: flat + edges
    x = np.clip(x + rng.normal(0, 0.02, x.shape), 0, 1).astype(np.float32)
    y = np.clip(x + rng.normal(0, 0.05, x.shape), 0, 1).astype(np.float32)
    loss = ImageLoss(1080, 1920, 0.1, gpu)
    g = loss(torch.from_numpy(x).to(gpu), torch.from_numpy(y).to(gpu)).cpu().numpy()
    lo, l1, ssim, grad = train_ref.l1_ssim_loss(x, y, 0.1)
    vals = loss.values.cpu().numpy()
    assert abs(vals[0] - lo) < 2e-6 and abs(vals[1] - l1) < 1e-6 and abs(vals[2] - ssim) < 5e-6
    assert np.abs(g - grad).max() < 2e-4 * np.abs(grad).max()  # flat regions: E[x^2] - mu^2 cancels in fp32


def test_trainer_improves_psnr(gpu):
    """A perturbed copy of a small scene is fitted to renders of the original: loss falls, PSNR rises."""
    from gs_frame import FrameRenderer
    from gs_scene import make_camera, make_scene
    from gs_train import TrainOptions, Trainer

    W, H = 160, 128
    scene, cam = make_scene(3000, W, H, seed=9), make_camera(W, H)
    gt = to_torch(scene, gpu)
    target, _ = FrameRenderer(gpu, max_pairs=1 << 16).forward(*gt, cam)
    target = target.clone()
    rng = np.random.default_rng(1)
    start = [t.clone() for t in gt]
    start[4] = start[4] + torch.from_numpy(rng.normal(0, 1.0, tuple(start[4].shape)).astype(np.float32)).to(gpu)
    start[3] = start[3] + torch.from_numpy(rng.normal(0, 0.5, tuple(start[3].shape)).astype(np.float32)).to(gpu)
    opt = TrainOptions(n_iters=400, n_iters_warmup=10)
    tr = Trainer(start, [cam], [target], opt, max_pairs=1 << 16)
    img0, _ = tr.renderer.forward(*tr.flat.params, cam)
    psnr0 = Trainer.psnr(img0, target)
    losses = []
    for it in range(120):
        losses.append(tr.train_step(it, 0).clone())
    losses = torch.stack(losses).cpu().numpy()
    img1, _ = tr.renderer.forward(*tr.flat.params, cam)
    psnr1 = Trainer.psnr(img1, target)
    assert np.isfinite(losses).all()
    assert losses[0, 0] == pytest.approx(losses[1, 0])  # step 0 runs with lr = lambda(0) = 0 (train.py:59-65)
    assert losses[-1, 0] < 0.6 * losses[1, 0]
    assert psnr1 > psnr0 + 3.0, (psnr0, psnr1)


def test_trainer_fits_degree3_sh_colours(gpu):
    """The whole step (render fwd/bwd with 48 SH coefficients per Gaussian, loss, fused Adam) on the degree-3
    extension: the perturbed coefficients are pulled back towards the target renders."""
    from gs_frame import FrameRenderer
    from gs_scene import make_camera, make_scene
    from gs_train import TrainOptions, Trainer

    W, H = 160, 128
    scene, cam = make_scene(3000, W, H, seed=10, use_sh=True, sh_degree=3), make_camera(W, H)
    gt = to_torch(scene, gpu)
    target, _ = FrameRenderer(gpu, max_pairs=1 << 16).forward(*gt, cam)
    target = target.clone()
    start = [t.clone() for t in gt]
    noise = np.random.default_rng(2).normal(0, 0.5, tuple(start[4].shape)).astype(np.float32)
    start[4] = start[4] + torch.from_numpy(noise).to(gpu)
    tr = Trainer(start, [cam], [target], TrainOptions(n_iters=400, n_iters_warmup=10), max_pairs=1 << 16)
    assert tr.flat.params[4].shape[1] == 48
    img0, _ = tr.renderer.forward(*tr.flat.params, cam)
    psnr0 = Trainer.psnr(img0, target)
    losses = torch.stack([tr.train_step(it, 0).clone() for it in range(120)]).cpu().numpy()
    img1, _ = tr.renderer.forward(*tr.flat.params, cam)
    psnr1 = Trainer.psnr(img1, target)
    assert np.isfinite(losses).all()
    assert losses[-1, 0] < 0.8 * losses[1, 0]
    assert psnr1 > psnr0 + 1.5, (psnr0, psnr1)


def test_trainer_with_densification(gpu):
    """The reference's schedule (train.py:86-91, 141-190) on a small scene: the statistic is cleared
    grad_accum_iters before every adaptive_control, N changes there, Adam restarts, training stays finite."""
    from gs_frame import FrameRenderer
    from gs_scene import make_camera, make_scene
    from gs_train import TrainOptions, Trainer

    W, H = 128, 96
    scene, cam = make_scene(2500, W, H, seed=4), make_camera(W, H)
    gt = to_torch(scene, gpu)
    target = FrameRenderer(gpu, max_pairs=1 << 16).forward(*gt, cam)[0].clone()
    start = [t.clone() for t in gt]
    start[4] = start[4] + 0.8 * torch.randn(start[4].shape, device=gpu, generator=torch.Generator(gpu).manual_seed(0))
    opt = TrainOptions(n_iters=400, n_iters_warmup=5, adaptive_control_start_iter=20, n_adaptive_control=25,
                       grad_accum_iters=10, split_thresh=0.02, delete_thresh=1.5, grad_thresh=1e-7, use_clone=1)
    tr = Trainer(start, [cam], [target], opt, max_pairs=1 << 16, densify=True,
                 generator=torch.Generator(gpu).manual_seed(1), fuse_adam=False)  # (the test reads flat.grads behind a step)
    sizes, losses = [], []
    for it in range(80):
        if it == 41:  # cleared at (it + 10 - 1) % 25 == 0, i.e. it = 41 (and 66)
            assert float(tr.optimizer.accum_grad.abs().max()) > 0
        losses.append(tr.train_step(it, 0).clone())
        sizes.append(tr.n_gaussians)
        if it == 41:
            stat = tr.optimizer.accum_grad.clone()  # holds exactly this step's |grad_pos| after the clear
            assert torch.equal(stat, tr.flat.grads[0].abs())
    losses = torch.stack(losses).cpu().numpy()
    assert np.isfinite(losses).all()
    changes = [i for i in range(1, 80) if sizes[i] != sizes[i - 1]]
    assert changes and set(changes) <= {25, 50, 75}, (changes, sizes[::5])
    assert sizes[-1] > sizes[0]                       # grad_thresh ~ 0: everything seen is cloned or split
    assert tr.optimizer.step_count == 80 - 1 - 75     # a fresh Adam after the last adaptive_control at it = 75
    img, _ = tr.renderer.forward(*tr.flat.params, cam)
    assert torch.isfinite(img).all()


@pytest.mark.parametrize("mode", ["max", "mean"])
def test_per_view_statistic_path_equals_fused_path(gpu, mode):
    """View-parallel training takes the densification statistic from each rank's own gradient BEFORE the all-reduce
    (gs_grad_stat_update + gs_dp.ViewParallelGradStat) instead of inside the Adam launch.  On one rank both paths
    see the same gradient, so the whole schedule -- statistic, prune / clone / split, parameters -- must agree bit
    for bit."""
    from gs_frame import FrameRenderer
    from gs_scene import make_camera, make_scene
    from gs_train import TrainOptions, Trainer

    W, H = 128, 96
    scene, cam = make_scene(2500, W, H, seed=4), make_camera(W, H)
    gt = to_torch(scene, gpu)
    target = FrameRenderer(gpu, max_pairs=1 << 16).forward(*gt, cam)[0].clone()
    start = [t.clone() for t in gt]
    start[4] = start[4] + 0.8 * torch.randn(start[4].shape, device=gpu, generator=torch.Generator(gpu).manual_seed(0))
    opt = TrainOptions(n_iters=400, n_iters_warmup=5, adaptive_control_start_iter=20, n_adaptive_control=25,
                       grad_accum_iters=10, split_thresh=0.02, delete_thresh=1.5, grad_thresh=1e-7, use_clone=1,
                       grad_accum_method=mode)
    runs = []
    for per_view in (False, True):
        tr = Trainer([t.clone() for t in start], [cam], [target], opt, max_pairs=1 << 16, densify=True,
                     generator=torch.Generator(gpu).manual_seed(1), per_view_stat=per_view)
        assert (tr.view_stat is not None) == per_view and (tr.optimizer.accum_grad is None) == per_view
        for it in range(60):
            tr.train_step(it, 0)
            if it == 45:
                stat = tr.view_stat.accum if per_view else tr.optimizer.accum_grad
                runs.append(stat.clone())
        runs.append([p.clone() for p in tr.flat.params])
    assert float(runs[0].abs().max()) > 0 and torch.equal(runs[0], runs[2])
    assert runs[1][0].shape[0] != 2500  # the Gaussian set did change (adaptive_control at 25 and 50)
    for a, b in zip(runs[1], runs[3]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("use_sh", [False, True])
def test_backward_parts_and_bucketed_adam_equal_the_one_call_step(gpu, use_sh):
    """The view-parallel gradient exchange issues the backward as GS_BWD_RASTER + the per-Gaussian sums in pieces --
    GS_BWD_GEOMETRY / GS_BWD_COLOR over everything (round 2), or slice by slice of the Gaussian array
    (gs_frame_backward_slice, round 4: both buckets in one kernel, or one part at a time) -- and Adam slice by slice
    (gs_adam_step_multi), so that a slice's exchange runs under the sums of the next.  Nothing may change numerically:
    gradients, stepped parameters, moments and the densification statistic are bit-identical to gs_frame_backward + one
    gs_adam_step -- with an odd Gaussian count (padded regions) and a ragged last slice."""
    from gaussian import _lib
    from gs_dp import FlatGaussianParams
    from gs_frame import FrameRenderer
    from gs_scene import make_camera, make_scene
    from gs_train import GROUPS, FusedAdam

    W, H = 160, 112
    scene, cam = make_scene(7001, W, H, seed=9, use_sh=use_sh), make_camera(W, H, yaw_deg=1.0)
    params = to_torch(scene, gpu)
    g = torch.randn(H, W, 3, device=gpu)
    lrs = dict(zip(GROUPS, [0.03, 0.02, 0.003, 0.004, 0.005]))
    outs = []
    for mode in ("one_call", "parts_gc", "parts_cg", "slices", "slices_by_part"):
        flat = FlatGaussianParams([t.clone() for t in params], n_slices=3 if mode.startswith("slices") else 1)
        assert flat.n_pad == 7004 and flat.region["scale"] == 7 * 7004  # regions padded to a multiple of 4 rows
        assert flat.bucket_ranges["color"][0] % 4 == 0 and flat.bucket_ranges["color"][0] >= 10 * 7001  # padded
        if mode.startswith("slices"):
            assert flat.n_slices == 3 and flat.slice_gaussians(2)[1] == 7001 and flat.slice_bounds[1] % 256 == 0
        r = FrameRenderer(gpu, max_pairs=1 << 17, training=True, auto_grow=False)
        opt = FusedAdam(flat, [lrs[k] for k in GROUPS], grad_stat="max")
        for _ in range(3):
            r.forward(*flat.params, cam)
            for gview in flat.grads:  # every element must be written by exactly one piece (the regions' padding is
                gview.fill_(float("nan"))  # nobody's: it stays zero)
            if mode == "one_call":
                r.backward(g, out=flat.grads)
                opt.step()
            elif mode.startswith("parts"):
                part = {"g": _lib.GS_BWD_GEOMETRY, "c": _lib.GS_BWD_COLOR}
                r.backward(g, out=flat.grads, part=_lib.GS_BWD_RASTER)
                for name in mode[-2:]:
                    r.backward(None, out=flat.grads, part=part[name])
                opt.step()
            else:
                r.backward(g, out=flat.grads, part=_lib.GS_BWD_RASTER)
                for k in (2, 0, 1):  # any order
                    g0, g1 = flat.slice_gaussians(k)
                    if mode == "slices":
                        r.backward_slice(flat.grads, g0, g1)
                    else:
                        r.backward_slice(flat.grads, g0, g1, part=_lib.GS_BWD_COLOR)
                        r.backward_slice(flat.grads, g0, g1, part=_lib.GS_BWD_GEOMETRY)
                for i, k in enumerate((1, 2, 0)):
                    opt.step_slice(k, advance=(i == 0))
        assert opt.step_count == 3 and bool(torch.isfinite(flat.flat_grad).all())
        outs.append((flat.flat_grad.clone(), flat.flat_param.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(),
                     opt.accum_grad.clone()))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)
    assert float((outs[0][1] - FlatGaussianParams(params).flat_param).abs().max()) > 0
    with pytest.raises(RuntimeError):  # slices start on multiples of 256 Gaussians
        r.backward_slice(flat.grads, 100, 300)


def test_trainer_bucketed_exchange_equals_plain_step(gpu):
    """Trainer.train_step with a process group up (one rank, RCCL; the collective is forced) takes the bucketed,
    asynchronous path -- rows, first bucket, its all-reduce, second bucket underneath it, Adam per bucket -- and must
    reproduce the plain single-process step bit for bit, regularisers and densification statistic included."""
    import os

    import torch.distributed as dist

    from gs_frame import FrameRenderer
    from gs_scene import make_camera, make_scene
    from gs_train import TrainOptions, Trainer

    W, H = 128, 96
    scene, cam = make_scene(3001, W, H, seed=6), make_camera(W, H)
    gt = to_torch(scene, gpu)
    target = FrameRenderer(gpu, max_pairs=1 << 16).forward(*gt, cam)[0].clone()
    start = [t.clone() for t in gt]
    start[4] = start[4] + 0.5 * torch.randn(start[4].shape, device=gpu, generator=torch.Generator(gpu).manual_seed(2))
    opt = TrainOptions(n_iters=100, n_iters_warmup=3, scale_reg=0.01, opa_reg=0.02)
    runs = []
    # plain step; then through RCCL (one rank, collective forced): all-reduce + replicated Adam / reduce-scatter + sharded
    # Adam + parameter all-gather, as ONE exchange unit and as a pipeline over three slices of the Gaussian array with
    # the next frame's project stage issued slice by slice behind the optimizer (round 4: next_camera_id)
    for exchange, n_slices, ahead in ((None, 1, False), ("all_reduce", 1, False), ("reduce_scatter", 1, False),
                                      ("all_reduce", 3, True), ("reduce_scatter", 3, True), ("all_reduce", 3, False)):
        bucketed = exchange is not None
        if bucketed:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu)
        try:
            tr = Trainer([t.clone() for t in start], [cam], [target], opt, max_pairs=1 << 16,
                         exchange=exchange or "all_reduce", n_slices=n_slices)
            tr.flat.force_collective = bucketed
            assert tr.flat.n_slices == n_slices
            assert tr.flat.collective_active() == bucketed and tr.optimizer.sharded == (exchange == "reduce_scatter")
            finishes = 0
            vals = []
            for it in range(12):
                finishes += int(tr.renderer.begun_frame_matches(*tr.flat.params, cam))
                vals.append(tr.train_step(it, 0, next_camera_id=0 if ahead else None).clone())
            # every frame but the first (whose capacity is checked synchronously) was projected behind the previous step
            assert finishes == (11 if ahead else 0)
            tr.flat.finish_gather()
            runs.append(([p.clone() for p in tr.flat.params], torch.stack(vals), tr.optimizer.accum_grad.clone()))
        finally:
            if bucketed:
                dist.destroy_process_group()
    for other in runs[1:]:
        for a, b in zip(runs[0][0], other[0]):
            assert torch.equal(a, b)
        assert torch.equal(runs[0][1], other[1]) and torch.equal(runs[0][2], other[2])
    assert float((runs[0][0][0] - start[0]).abs().max()) > 0


def _two_rank_worker(rank, world, port, tmp, exchange):
    """One of two ranks that share ONE GPU (gloo moves the device tensors through the host): rank r trains on view r."""
    import os

    import torch.distributed as dist

    from gs_frame import FrameRenderer
    from gs_scene import make_camera, make_scene
    from gs_train import TrainOptions, Trainer

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    gpu = torch.device("cuda:0")
    torch.cuda.set_device(gpu)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    FrameRenderer.default_force_strips = True  # (tests/conftest.py does this in the parent: project stage in slices)
    W, H = 128, 96
    scene = make_scene(3001, W, H, seed=6)
    cams = [make_camera(W, H, yaw_deg=4.0 * r) for r in range(world)]  # (two ranks: 0 and 4 degrees)
    gt = to_torch(scene, gpu)
    targets = [FrameRenderer(gpu, max_pairs=1 << 16).forward(*gt, c)[0].clone() for c in cams]
    start = [t.clone() for t in gt]
    start[4] = start[4] + 0.5 * torch.randn(start[4].shape, device=gpu, generator=torch.Generator(gpu).manual_seed(2))
    tr = Trainer(start, cams, targets, TrainOptions(n_iters=100, n_iters_warmup=3), world_size=world, max_pairs=1 << 16,
                 exchange=exchange, n_slices=3)
    assert tr.flat.collective_active() and tr.flat.rank == rank and tr.flat.n_slices == 3
    assert tr.flat.n_pad % (4 * world) == 0 and tr.flat.self_check() in ("grouped", "public")
    ahead, n_steps = 0, (6 if world == 2 else 3)
    for it in range(n_steps):
        ahead += int(tr.renderer.begun_frame_matches(*tr.flat.params, cams[rank]))
        tr.train_step(it, rank, next_camera_id=rank)
    assert ahead == n_steps - 1  # every frame but the first was projected behind the previous step's optimizer
    tr.flat.finish_gather()
    out = {"param": tr.flat.flat_param.cpu(), "state_bytes": tr.optimizer.state_bytes}
    # the number of slices picked by measurement (replicated optimizer only): every rank must arrive at the same count,
    # and the steps taken while measuring are ordinary training steps
    out["tuned"] = tr.tune_slices(6, rank, candidates=(1, 2, 3), iters=2 if world == 2 else 1, next_camera_id=rank)
    out["slices_after"] = tr.flat.n_slices
    tr.train_step(20, rank, next_camera_id=rank)
    tr.flat.finish_gather()
    out["finite"] = bool(torch.isfinite(tr.flat.flat_param).all())
    out["param_after"] = tr.flat.flat_param.cpu()
    if world > 2:  # the per-view densification statistic through the same group (train.py:145-154 under view parallelism)
        from gs_dp import ViewParallelGradStat

        st = ViewParallelGradStat(1000, gpu, "max", world_size=world)
        st.accum.copy_(torch.rand(1000, 3, generator=torch.Generator().manual_seed(50 + rank)).to(gpu))
        local = st.accum.cpu().clone()
        acc, _ = st.reduce()
        out["stat_local"], out["stat_reduced"] = local, acc.cpu().clone()
    torch.save(out, os.path.join(tmp, f"two_rank_{exchange}_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["all_reduce", "reduce_scatter"])
def test_eight_ranks_on_one_gpu_equal_the_summed_gradient_step(gpu, tmp_path, exchange):
    """The same at the rank count of the target node: EIGHT ranks share this GPU through
    gloo, rank r trains on view r -- regions padded to multiples of 4 x 8 rows, eighth-shards of every slice range in
    the reduce-scatter mode, 1 / 8 inside the fused Adam, the grouped-collective self-check, Trainer.tune_slices and
    ViewParallelGradStat.reduce with eight peers.  Three steps; all eight replicas bit-identical to each other, and equal
    -- up to the order in which the backend adds eight buffers -- to one process that renders the eight views itself,
    adds the eight gradients and takes the same steps."""
    _ranks_on_one_gpu(gpu, tmp_path, exchange, 8)


@pytest.mark.parametrize("exchange", ["all_reduce", "reduce_scatter"])
def test_two_ranks_on_one_gpu_equal_the_summed_gradient_step(gpu, tmp_path, exchange):
    _ranks_on_one_gpu(gpu, tmp_path, exchange, 2)


def _ranks_on_one_gpu(gpu, tmp_path, exchange, world):
    """View parallelism with REAL kernels on two ranks: both ranks run on this one GPU and exchange through gloo (which
    stages device tensors through the host), rank r renders view r.  After six steps of the slice pipeline (three
    slices, SUM exchange, 1 / world inside the fused Adam, the next frame's project stage issued ahead; all-reduce +
    replicated Adam, and reduce-scatter + sharded Adam + parameter all-gather with a non-zero rank's shard offsets) both
    ranks hold the same parameters, and they are bit for bit those of a single process that renders both views itself,
    adds the two gradients and takes the same steps."""
    import torch.multiprocessing as mp

    from gs_dp import FlatGaussianParams
    from gs_frame import FrameRenderer
    from gs_scene import make_camera, make_scene
    from gs_train import FusedAdam, ImageLoss, TrainOptions, base_lrs, lr_lambdas

    port = 38500 + (os.getpid() % 1500) + (7 if exchange == "reduce_scatter" else 0) + 13 * world
    mp.spawn(_two_rank_worker, args=(world, port, str(tmp_path), exchange), nprocs=world, join=True)
    got = [torch.load(tmp_path / f"two_rank_{exchange}_{r}.pt") for r in range(world)]
    assert all(torch.equal(got[0]["param"], g["param"]) for g in got[1:])
    # ---- the same steps in ONE process: the sum of the views' gradients, Adam with grad_scale 1 / world
    W, H = 128, 96
    scene = make_scene(3001, W, H, seed=6)
    cams = [make_camera(W, H, yaw_deg=4.0 * r) for r in range(world)]
    gt = to_torch(scene, gpu)
    targets = [FrameRenderer(gpu, max_pairs=1 << 16).forward(*gt, c)[0].clone() for c in cams]
    start = [t.clone() for t in gt]
    start[4] = start[4] + 0.5 * torch.randn(start[4].shape, device=gpu, generator=torch.Generator(gpu).manual_seed(2))
    opt = TrainOptions(n_iters=100, n_iters_warmup=3)
    flat = FlatGaussianParams(start, world_size=world, rank=0, n_slices=3)  # same padded layout, no process group
    lam, base = lr_lambdas(opt), base_lrs(opt)
    adam = FusedAdam(flat, [b * f(0) for b, f in zip(base, lam)], betas=opt.betas, eps=opt.eps, grad_stat="max")
    r = FrameRenderer(gpu, max_pairs=1 << 16, training=True, auto_grow=False)
    loss = ImageLoss(H, W, opt.ssim_weight, gpu)
    other = [torch.empty_like(g) for g in flat.grads]
    total = [torch.empty_like(g) for g in flat.grads]
    for it in range(6 if world == 2 else 3):
        # the order in which the exchange adds the ranks' buffers is the backend's (gloo: not a fixed left-to-right sum
        # beyond two ranks), so with eight ranks the comparison below allows the last bits of a different summation
        # order; with two ranks (a + b is commutative in fp32) it is bit for bit
        for v in range(world - 1, -1, -1):
            img, _ = r.forward(*flat.params, cams[v])
            r.backward(loss(img, targets[v]), out=other if v else flat.grads)
            if v == world - 1:
                for a, b in zip(total, other):
                    a.copy_(b)
            elif v:
                for a, b in zip(total, other):
                    a.add_(b)
        for a, b in zip(flat.grads, total):
            a.add_(b)
        for k in range(flat.n_slices):
            adam.step_slice(k, advance=(k == 0), grad_scale=1.0 / world)
        adam.set_lrs([f(it) * b for f, b in zip(lam, base)])
    mine, theirs = flat.flat_param.cpu(), got[0]["param"]
    if world == 2:
        assert torch.equal(mine, theirs)
    else:
        # Adam divides by sqrt(v) + eps: a last-bit difference of a tiny gradient can move a parameter by a full step
        # (lr ~ 1e-3 x 10) in the first iterations, so the criterion is the share of parameters that agree closely
        close = (mine - theirs).abs() <= 1e-5 + 1e-4 * theirs.abs()
        assert float(close.float().mean()) > 0.995, float(close.float().mean())
    assert float((mine - FlatGaussianParams(start, world_size=world, rank=0).flat_param.cpu()).abs().max()) > 0
    # Trainer.tune_slices: the same choice on every rank, the replicas still agree afterwards
    assert len({g["tuned"] for g in got} | {g["slices_after"] for g in got}) == 1
    assert got[0]["tuned"] in ((1, 2, 3) if exchange == "all_reduce" else (3,))
    assert all(g["finite"] for g in got) and all(torch.equal(got[0]["param_after"], g["param_after"]) for g in got[1:])
    # the sharded optimizer keeps 1 / world of the state per rank
    full = 8 * flat.flat_param.numel()
    assert got[world - 1]["state_bytes"] == (full // world if exchange == "reduce_scatter" else full)
    if world > 2:
        want = torch.stack([g["stat_local"] for g in got]).amax(0)
        assert all(torch.equal(g["stat_reduced"], want) for g in got)


def test_grad_stat_update_kernel(gpu):
    from gaussian import _lib

    g = torch.randn(100_003, 3, device=gpu)
    for mode, ref in ((1, lambda s: torch.maximum(s, g.abs())), (2, lambda s: s + g.abs())):
        s = torch.rand(100_003, 3, device=gpu)
        want = ref(s)
        _lib.check(_lib.gs_grad_stat_update(g.data_ptr(), s.data_ptr(), s.numel(), mode,
                                            torch.cuda.current_stream().cuda_stream), "gs_grad_stat_update")
        assert torch.equal(s, want)
    with pytest.raises(RuntimeError):
        _lib.check(_lib.gs_grad_stat_update(g.data_ptr(), g.data_ptr(), 3, 0, None), "gs_grad_stat_update")


def test_viewer_hook_and_checkpoint(gpu, tmp_path):
    """Trainer.test(None, extrinsics, intrinsics) -- the call the reference's viser GUI makes per frame
    (visergui.py:137-149) -- at a size that is not a multiple of 16, against the oracle; test(camera_id)
    metrics; the reference's checkpoint dict round trip."""
    from gs_scene import make_camera, make_scene
    from gs_testutil import OracleFrame
    from gs_train import Trainer

    W, H = 333, 201
    scene, cam = make_scene(8000, W, H, seed=12), make_camera(W, H, yaw_deg=3.0)
    params = to_torch(scene, gpu)
    of = OracleFrame(scene, cam)
    target = torch.from_numpy(of.image).to(gpu)
    tr = Trainer(params, [cam], [target], max_pairs=1 << 17)
    out = tr.test(None, extrinsics={"rot": torch.from_numpy(cam.rot), "tran": cam.tran},
                  intrinsics={"width": W, "height": H, "focal_x": cam.focal_x, "focal_y": cam.focal_y})
    assert set(out) == {"image"} and tuple(out["image"].shape) == (H, W, 3)
    assert np.abs(out["image"].cpu().numpy() - of.image).max() < 5e-5
    m = tr.test(0)
    assert m["psnr"] > 60 and m["ssim"] > 0.9999 and m["render_time"] > 0
    path = str(tmp_path / "ckpt.pth")
    tr.save_checkpoint(path)
    ck = torch.load(path)
    assert set(ck) == {"pos", "opa", "rgb", "quat", "scale"}  # train.py:283-291
    tr2 = Trainer([torch.zeros_like(p) for p in params], [cam], [target], max_pairs=1 << 17)
    tr2.load_checkpoint(path)
    assert all(torch.equal(a, b) for a, b in zip(tr2.flat.params, tr.flat.params))
    assert torch.equal(tr2.test(0)["image"], m["image"])


def test_bench_multi_gpu_leg_runs_under_two_ranks(gpu, tmp_path):
    """bench.py's own N > 1 path -- `python -m torch.distributed.run --nproc-per-node 2 bench.py
    --gpus 2 ...`, the driver's SCALE command line -- had never executed with more than one rank anywhere: the builder's
    boxes have one GPU and RCCL refuses two ranks on a device.  With the file's test hooks (gloo as the backend, both
    ranks on this GPU) the very same code runs: rank-symmetric legs, barriers, max over ranks, one view per rank, and the
    multi_gpu leg with both exchange modes, the two-slice pipeline and the slice sweep.  Checked: ONE JSON line from rank
    0, n_gpus = 2, two ranks seen by a collective, every mode timed, no leg failed.  Not a measurement (--quick)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 40500 + (os.getpid() % 1500)
    env = dict(os.environ, GS_BENCH_BACKEND="gloo", GS_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
           "--config", "cfg1", "--legs", "headline,multi_gpu", "--quick"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # rank 0 alone prints, exactly one line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 2 and out["value"] > 0
    assert out["scaling"] == "weak" and "x2" in out["config"]["parallelism"]
    assert "leg_errors" not in out, out.get("leg_errors")
    mg = out["multi_gpu"]
    assert mg["ranks_seen"] == 2 and mg["backend"] == "gloo"
    scene = mg["scenes"]["cfg1"]
    for mode in ("all_reduce", "reduce_scatter"):
        m = scene["modes"][mode]
        # (busbw: half a megabyte through gloo and the host in ~40 ms rounds to 0.0 GB/s -- present and finite is the point)
        assert m["train_views_per_s"] > 0 and np.isfinite(m["exposed_ms"]) and m["exchange_ms"] > 0 and m["busbw_GBs"] >= 0
    assert scene["two_slice_pipeline"]["n_slices"] == 2
    assert set(scene["slices_sweep_all_reduce"]) >= {"1"}
    # sharded optimizer: half the state per rank
    assert scene["modes"]["reduce_scatter"]["optimizer_state_bytes_per_rank"] * 2 <= \
        scene["modes"]["all_reduce"]["optimizer_state_bytes_per_rank"] + 64


@pytest.mark.parametrize("n,W,H,stat,sh", [(6_000, 160, 112, "max", 0), (6_000, 160, 112, "mean", 0), (6_001, 160, 112, "max", 0),
                                           (2_400_000, 1920, 1080, "max", 0), (6_000, 160, 112, "max", 2),
                                           (6_001, 160, 112, "mean", 3), (724_312, 1920, 1080, "max", 2)])
def test_fused_backward_adam_equals_backward_then_adam(gpu, n, W, H, stat, sh):
    """Round 5: gs_frame_backward_adam (what a single-rank Trainer step uses unless fuse_adam=False)
    -- the Adam update applied inside the backward's last kernel, no gradient buffer -- against
    gs_frame_backward followed by gs_adam_step: parameters, both moments and the |pos.grad| statistic BIT FOR BIT over several
    steps of gs_train.Trainer (learning-rate warm-up included; Gaussians outside the frustum take their zero-gradient momentum
    step on both paths).  2.4 M Gaussians: the moments stream with non-temporal accesses there (the third kernel variant);
    6,001: the [N, 3] arrays end inside a float4 of the kernel's walk (its element-by-element tail).  Round 6: SH colours
    (``sh`` = the degree) -- the wave that sums a Gaussian's coefficient gradients steps its 27 / 48 coefficients in place;
    724,312 x SH2 is beyond the cache (the non-temporal variant), 6,001 ends inside a wave's run of Gaussians."""
    from gs_frame import FrameRenderer
    from gs_scene import make_camera, make_scene
    from gs_train import TrainOptions, Trainer

    scene = make_scene(n, W, H, seed=11, use_sh=bool(sh), sh_degree=sh) if sh else make_scene(n, W, H, seed=11)
    cam = make_camera(W, H, yaw_deg=3.0)
    gt = to_torch(scene, gpu)
    r0 = FrameRenderer(gpu, max_pairs=1 << 20, auto_grow=True)
    target = r0.forward(*gt, cam)[0].clone()
    pairs = r0.stats().pairs
    del r0
    start = [t.clone() for t in gt]
    start[4] = start[4] + 0.5 * torch.randn(start[4].shape, device=gpu, generator=torch.Generator(gpu).manual_seed(2))
    start[3] = start[3] - 0.3
    opt = TrainOptions(n_iters=100, n_iters_warmup=3, grad_accum_method=stat)
    steps = 3 if n > 1_000_000 else 7
    got = []
    for fuse in (True, False):
        tr = Trainer([t.clone() for t in start], [cam], [target], opt, max_pairs=int(pairs * 1.3) + 4096, fuse_adam=fuse)
        assert tr._can_fuse_adam() == fuse
        vals = [tr.train_step(i, 0).clone() for i in range(steps)]
        assert tr.optimizer.step_count == steps
        got.append((tr.flat.flat_param.clone(), tr.optimizer.exp_avg.clone(), tr.optimizer.exp_avg_sq.clone(),
                    tr.optimizer.accum_grad.clone(), torch.stack(vals)))
        if not fuse:
            culled = ~tr.renderer.culling_mask()
            assert int(culled.sum()) > 0  # (Gaussians outside the frustum: zero gradient, momentum step)
        del tr
        torch.cuda.empty_cache()
    for a, b, name in zip(got[0], got[1], ("parameters", "exp_avg", "exp_avg_sq", "grad statistic", "loss values")):
        assert torch.equal(a, b), name
    assert float(got[0][1].abs().max()) > 0 and float(got[0][3].abs().max()) > 0


@pytest.mark.parametrize("use_sh", [False, True])
def test_fused_backward_adam_skips_overflowed_frames(gpu, use_sh):
    """With the frame's overflow counter as skip flag an overflowed -- empty -- frame moves nothing: rgb logits and (round 6)
    SH coefficients alike, although momentum is there that WOULD move them."""
    from gs_frame import FrameRenderer
    from gs_scene import make_camera, make_scene
    from gs_train import TrainOptions, Trainer

    W, H = 128, 96
    cam = make_camera(W, H)
    sh = make_scene(1500, W, H, seed=3, use_sh=use_sh)
    gt = to_torch(sh, gpu)
    target = FrameRenderer(gpu, max_pairs=1 << 16).forward(*gt, cam)[0].clone()
    # a workspace that is too small: the frame overflows, the device-side flag skips the fused step
    rgb = make_scene(4000, W, H, seed=4, use_sh=use_sh)
    p = [t.clone() for t in to_torch(rgb, gpu)]
    tr2 = Trainer(p, [cam], [target], TrainOptions(n_iters_warmup=1), max_pairs=1 << 16, fuse_adam=True)
    assert tr2._can_fuse_adam()
    r = tr2.renderer
    r.auto_grow = False
    r.max_pairs = 64  # far too small
    before = tr2.flat.flat_param.clone()
    img, _ = r.forward(*tr2.flat.params, cam)
    assert r.stats().overflow > 0
    tr2.optimizer.skip_flag = r.overflow_flag()
    tr2.optimizer.exp_avg.fill_(0.5)  # momentum that WOULD move the parameters
    r.backward_adam(torch.ones(H, W, 3, device=gpu), tr2.optimizer.fused_descriptor())
    assert torch.equal(tr2.flat.flat_param, before)
