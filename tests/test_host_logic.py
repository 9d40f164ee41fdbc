"""CPU tier: host-side logic -- scene generator determinism, trunc_exp, and the view-parallel
gradient bucket over a 2-process gloo group (the RCCL path's CPU stand-in)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gs_geometry import TileGrid
from gs_scene import CONFIGS, make_camera, make_scene


def test_scene_generator_is_deterministic_and_matches_survey_statistics():
    a, b = make_scene(10_000, 256, 256), make_scene(10_000, 256, 256)
    for f in ("pos", "quat", "scale", "opa", "rgb"):
        assert np.array_equal(getattr(a, f), getattr(b, f))
    assert a.pos.dtype == np.float32 and a.rgb.shape == (10_000, 3)
    assert make_scene(100, 64, 64, use_sh=True).rgb.shape == (100, 27)
    import oracle
    from gs_testutil import OracleFrame

    of = OracleFrame(a, make_camera(256, 256))
    V, M = int(of.mask.sum()), len(of.ids)
    assert 7500 < V < 8200 and 27_000 < M < 31_000  # V ~ 7.9 k, M ~ 29 k
    assert set(CONFIGS) == {"cfg1", "cfg2", "cfg3", "cfg4", "cfg5", "cfg6"}  # cfg6: dense-scene stress, not a BASELINE config


def test_tile_grid_padding_and_crop():
    g = TileGrid(1920, 1080, 1440.0, 1440.0)
    assert (g.padded_width, g.padded_height, g.n_tile_x, g.n_tile_y, len(g)) == (1920, 1088, 120, 68, 8160)
    assert g.crop_offsets() == (4, 0)
    g = TileGrid(333, 201, 250.0, 250.0)
    assert (g.padded_width, g.padded_height) == (336, 208) and g.crop_offsets() == (3, 1)


def test_trunc_exp_matches_reference_definition():
    from renderer import trunc_exp

    x = torch.tensor([-3.0, -0.5, 0.0, 0.7, 2.5], requires_grad=True)
    y = trunc_exp(x)
    assert torch.allclose(y, torch.exp(x))
    y.backward(torch.ones_like(y))
    assert torch.allclose(x.grad, torch.exp(x.detach().clamp(-1, 1)))  # renderer.py:97-100


def _dp_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gs_dp import FlatGaussianParams
    from gs_testutil import OracleFrame

    scene = make_scene(300, 48, 32, seed=5)
    params = [torch.from_numpy(a.copy()) for a in (scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb)]
    flat = FlatGaussianParams(params, world_size=world)
    flat.broadcast_params(0)
    assert flat.flat_grad.data_ptr() % 16 == 0 and flat.grads[1].data_ptr() == flat.flat_grad.data_ptr()
    # each rank renders ITS view with the CPU oracle standing in for the HIP renderer
    cam = make_camera(48, 32, yaw_deg=5.0 * rank)
    of = OracleFrame(scene, cam)
    w = np.random.default_rng(100 + rank).normal(size=of.image.shape).astype(np.float32)
    g = of.backward(w)
    for dst, name in zip(flat.grads, ("pos", "quat", "scale", "opa", "rgb")):
        dst.copy_(torch.from_numpy(g[name]))
    np.save(os.path.join(tmp, f"local_{rank}.npy"), flat.flat_grad.numpy().copy())
    flat.all_reduce_grads()
    np.save(os.path.join(tmp, f"reduced_{rank}.npy"), flat.flat_grad.numpy().copy())
    dist.destroy_process_group()


def test_view_parallel_gradient_bucket_gloo(tmp_path):
    """2 ranks x 1 view == mean of the per-view gradients; one flat all-reduce."""
    world, port = 2, 29500 + (os.getpid() % 2000)
    mp.spawn(_dp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    local = [np.load(tmp_path / f"local_{r}.npy") for r in range(world)]
    red = [np.load(tmp_path / f"reduced_{r}.npy") for r in range(world)]
    expect = (local[0].astype(np.float64) + local[1]) / 2
    assert np.abs(local[0] - local[1]).max() > 0  # the two views really differ
    for r in range(world):
        assert np.allclose(red[r], expect, rtol=1e-6, atol=1e-9)
    assert np.array_equal(red[0], red[1])


def _bucket_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gs_dp import FlatGaussianParams

    rng = np.random.default_rng(40 + rank)
    n = 501  # odd: every tensor's region is padded to 504 rows (a multiple of 4 x world)
    shapes = [(n, 3), (n, 4), (n, 3), (n,), (n, 27)]
    params = [torch.from_numpy(rng.normal(size=s).astype(np.float32)) for s in shapes]
    out = {}
    for mode in ("blocking", "bucketed"):
        flat = FlatGaussianParams(params, world_size=world)
        g = np.random.default_rng(70 + rank).normal(size=flat.flat_grad.numel()).astype(np.float32)
        flat.flat_grad.copy_(torch.from_numpy(g))
        if mode == "blocking":
            flat.all_reduce_grads()
        else:
            assert flat.collective_active()
            # round 4 layout: every tensor owns a region of n_pad rows, n_pad = n rounded up to a multiple of 4 x world
            # (equal, float4-aligned shards of every bucket AND of every slice range); pad rows never move
            assert flat.n_pad == 504
            assert flat.bucket_ranges == {"geometry": (0, 5040), "color": (5040, 5040 + 28 * 504)}
            assert flat.offsets["opa"] == (5040, 5040 + n)
            assert flat.group_ends == [4 * 504, 7 * 504, 5040, 5040 + 504, 5040 + 28 * 504]
            flat.begin_bucket("color")      # the order gs_train.Trainer uses with SH colours
            flat.begin_bucket("geometry")
            flat.finish_bucket("color")
            flat.finish_bucket("geometry")
            flat.finish_bucket("geometry")  # idempotent
        out[mode] = flat.flat_grad.numpy().copy()
    np.save(os.path.join(tmp, f"bucket_{rank}.npy"), np.stack([g, out["blocking"], out["bucketed"]]))
    dist.destroy_process_group()


def test_bucketed_async_exchange_equals_blocking_all_reduce_gloo(tmp_path):
    """gs_dp: the two-bucket asynchronous exchange (colour bucket first, geometry second) gives every rank exactly
    what the single blocking all-reduce of the flat buffer gives: the mean of the ranks' local gradients."""
    world, port = 2, 33500 + (os.getpid() % 2000)
    mp.spawn(_bucket_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"bucket_{k}.npy") for k in range(world)]
    expect = ((r[0][0].astype(np.float64) + r[1][0]) / 2).astype(np.float32)
    for k in range(world):
        assert np.array_equal(r[k][1], r[k][2])          # bucketed == blocking, bit for bit
        assert np.allclose(r[k][1], expect, rtol=1e-6, atol=1e-9)
    assert np.array_equal(r[0][2], r[1][2])


def _exchange_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gs_dp import FlatGaussianParams

    n = 333  # 10 n and 4 n are no multiples of 8: both buckets are padded
    shapes = [(n, 3), (n, 4), (n, 3), (n,), (n, 3)]
    params = [torch.from_numpy(np.random.default_rng(5).normal(size=s).astype(np.float32)) for s in shapes]  # replicas
    out = {}
    for exchange in ("all_reduce", "reduce_scatter"):
        flat = FlatGaussianParams(params, world_size=world, exchange=exchange)
        assert flat.rank == rank
        for step in range(3):
            g = np.random.default_rng(1000 * step + rank).normal(size=flat.flat_grad.numel()).astype(np.float32)
            flat.flat_grad.copy_(torch.from_numpy(g))
            flat.finish_gather()  # what Trainer.train_step does before it reads the parameters
            for name in ("color", "geometry"):
                flat.begin_bucket(name)
            for name in ("color", "geometry"):
                flat.finish_bucket(name)
                lo, hi = flat.optimizer_range(name)  # the whole bucket, or this rank's slice of it
                lo_b, hi_b = flat.bucket_ranges[name]
                assert (lo, hi) == ((lo_b, hi_b) if exchange == "all_reduce" else flat.shard_range(name))
                assert (hi - lo) * (world if exchange == "reduce_scatter" else 1) == hi_b - lo_b and lo % 4 == 0
                # stand-in for the fused Adam (HIP only): an elementwise update of what this rank owns
                flat.flat_param[lo:hi].sub_(0.1 * flat.flat_grad[lo:hi] + 0.01 * torch.sign(flat.flat_param[lo:hi]))
                flat.begin_gather(name)
        flat.finish_gather()
        out[exchange] = flat.flat_param.numpy().copy()
    np.save(os.path.join(tmp, f"exchange_{rank}.npy"), np.stack([out["all_reduce"], out["reduce_scatter"]]))
    dist.destroy_process_group()


def test_reduce_scatter_sharded_update_equals_all_reduce_replicated_gloo(tmp_path):
    """gs_dp exchange modes: reduce-scatter -> update of the rank's own slice -> all-gather of the parameters leaves
    every rank with exactly the parameters that all-reduce -> update of everything gives (2 gloo ranks, 3 steps)."""
    world, port = 2, 35500 + (os.getpid() % 2000)
    mp.spawn(_exchange_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"exchange_{k}.npy") for k in range(world)]
    for k in range(world):
        assert np.array_equal(r[k][0], r[k][1])  # sharded == replicated, bit for bit
    assert np.array_equal(r[0][1], r[1][1])      # and the replicas agree
    assert np.abs(r[0][0]).max() > 0


def _slice_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gs_dp import ORDER, FlatGaussianParams, project_slice_size

    n = 1999  # project slices of 256 Gaussians -> 8 of them; three exchange slices, the last one ragged + 1 pad row
    shapes = [(n, 3), (n, 4), (n, 3), (n,), (n, 27)]
    params = [torch.from_numpy(np.random.default_rng(5).normal(size=s).astype(np.float32)) for s in shapes]  # replicas
    out = {}
    for exchange, sliced in (("all_reduce", False), ("all_reduce", True), ("reduce_scatter", True)):
        flat = FlatGaussianParams(params, world_size=world, exchange=exchange, n_slices=3 if sliced else 1)
        if sliced:
            assert project_slice_size(n) == 256 and flat.n_pad == 2000 and flat.n_slices == 3
            assert flat.slice_bounds == [0, 768, 1280, 2000]  # whole project slices, multiples of 4 x world
            # the slices tile every region exactly
            cover = sorted(rg for k in range(flat.n_slices) for rg in flat.slice_ranges(k))
            assert cover[0][0] == 0 and cover[-1][1] == flat.flat_grad.numel()
            assert all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
            assert [flat.slice_gaussians(k) for k in range(3)] == [(0, 768), (768, 1280), (1280, 1999)]
        for step in range(3):
            g = np.random.default_rng(1000 * step + rank).normal(size=flat.flat_grad.numel()).astype(np.float32)
            flat.flat_grad.copy_(torch.from_numpy(g))
            for t in ORDER:  # what the kernels guarantee: pad rows carry zero gradient
                lo = flat.offsets[t][1]
                hi = flat.region[t] + flat.n_pad * flat.width[t]
                flat.flat_grad[lo:hi].zero_()
            flat.finish_gather()
            # the order gs_train.Trainer uses: all exchanges started slice by slice, then per slice: wait, update what
            # this rank owns, send the updated shards on their way (reduce-scatter mode)
            for k in range(flat.n_slices):
                flat.begin_slice(k)
            for k in range(flat.n_slices):
                flat.finish_slice(k)
                for lo, hi in flat.owned(flat.slice_ranges(k)):
                    assert lo % 4 == 0 and (hi - lo) % 4 == 0
                    # stand-in for the fused Adam (HIP only): an elementwise update of what this rank owns
                    flat.flat_param[lo:hi].sub_(0.1 * flat.flat_grad[lo:hi] + 0.01 * torch.sign(flat.flat_param[lo:hi]))
                flat.begin_slice_gather(k)
                if k:
                    flat.finish_slice_gather(k - 1)  # where the next frame's project stage of slice k - 1 would start
        flat.finish_gather()
        out[(exchange, sliced)] = flat.flat_param.numpy().copy()
    np.save(os.path.join(tmp, f"slices_{rank}.npy"), np.stack([out[("all_reduce", False)], out[("all_reduce", True)],
                                                              out[("reduce_scatter", True)]]))
    dist.destroy_process_group()


def test_slice_pipeline_equals_one_blocking_exchange_gloo(tmp_path):
    """gs_dp round 4: the gradient exchange cut into slices of the Gaussian array (five element ranges per slice, started
    slice by slice, waited for slice by slice; all-reduce + replicated update, or reduce-scatter + sharded update +
    all-gather of the parameters) leaves every rank with exactly the parameters of the single-slice exchange."""
    world, port = 2, 37500 + (os.getpid() % 2000)
    mp.spawn(_slice_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"slices_{k}.npy") for k in range(world)]
    for k in range(world):
        assert np.array_equal(r[k][0], r[k][1])  # sliced == unsliced, bit for bit
        assert np.array_equal(r[k][0], r[k][2])  # sharded == replicated
    assert np.array_equal(r[0][0], r[1][0])      # and the replicas agree
    assert np.abs(r[0][0]).max() > 0


def _api_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import warnings

    from gs_dp import FlatGaussianParams

    n = 1999
    shapes = [(n, 3), (n, 4), (n, 3), (n,), (n, 27)]
    params = [torch.from_numpy(np.random.default_rng(5).normal(size=s).astype(np.float32)) for s in shapes]
    res = {}
    for exchange in ("all_reduce", "reduce_scatter"):
        outs = []
        for forced in ("", "public"):
            os.environ["GS_DP_COLLECTIVES"] = forced
            flat = FlatGaussianParams(params, world_size=world, exchange=exchange, n_slices=3)
            with warnings.catch_warnings(record=True) as wlist:
                warnings.simplefilter("always")
                api = flat.self_check()  # the closed-form pattern through the selected API; falls back if it must
            assert api in ("grouped", "public") and (forced != "public" or api == "public")
            res[f"{exchange}/{forced or 'auto'}"] = api + ("" if not wlist else " (fell back)")
            for step in range(2):  # and a real exchange through whichever API was settled on
                g = np.random.default_rng(77 * step + rank).normal(size=flat.flat_grad.numel()).astype(np.float32)
                flat.flat_grad.copy_(torch.from_numpy(g))
                for k in range(flat.n_slices):
                    flat.begin_slice(k)
                for k in range(flat.n_slices):
                    flat.finish_slice(k)
                    for lo, hi in flat.owned(flat.slice_ranges(k)):
                        flat.flat_param[lo:hi].sub_(0.1 * flat.flat_grad[lo:hi])
                    flat.begin_slice_gather(k)
                flat.finish_gather()
            assert flat._api() == api
            outs.append(flat.flat_param.numpy().copy())
        assert np.array_equal(outs[0], outs[1]), exchange  # grouped (or what it fell back to) == public, bit for bit
        np.save(os.path.join(tmp, f"api_{exchange}_{rank}.npy"), outs[0])
    os.environ.pop("GS_DP_COLLECTIVES", None)
    if rank == 0:
        import json

        json.dump(res, open(os.path.join(tmp, "api.json"), "w"))
    dist.destroy_process_group()


def test_exchange_api_self_check_and_public_fallback_gloo(tmp_path):
    """the exchange goes through private ProcessGroup entry points with in-place views.  gs_dp now
    (a) checks one grouped exchange of a known pattern against its closed form before it trusts that path with world > 1,
    (b) falls back to torch.distributed's public calls if the check (or the call) fails, and (c) can be forced onto the
    public calls (GS_DP_COLLECTIVES=public).  Two gloo ranks: the check passes on some API in both modes, and a sliced
    exchange + sharded update gives the same parameters bit for bit through either API and on both ranks."""
    world, port = 2, 39500 + (os.getpid() % 2000)
    mp.spawn(_api_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    import json

    res = json.load(open(tmp_path / "api.json"))
    assert set(res) == {"all_reduce/auto", "all_reduce/public", "reduce_scatter/auto", "reduce_scatter/public"}, res
    assert res["all_reduce/public"] == "public" and res["reduce_scatter/public"] == "public"
    for exchange in ("all_reduce", "reduce_scatter"):
        a, b = (np.load(tmp_path / f"api_{exchange}_{k}.npy") for k in range(world))
        assert np.array_equal(a, b)
    assert np.array_equal(np.load(tmp_path / "api_all_reduce_0.npy"), np.load(tmp_path / "api_reduce_scatter_0.npy"))


@pytest.mark.parametrize("n,world,n_slices", [(1, 1, None), (255, 1, 3), (1999, 2, 3), (7001, 1, 3), (100_000, 4, 5),
                                              (2_400_000, 8, None), (2_400_000, 1, None), (1_000_003, 3, 4)])
def test_exchange_slices_tile_the_flat_buffer(n, world, n_slices):
    """gs_dp.FlatGaussianParams (CPU tensors, no process group): regions padded to a multiple of 4 x world rows, slices
    made of whole project slices that are multiples of 4 x world Gaussians, the slices' element ranges tile the buffer
    exactly once, every range splits into equal float4-aligned shards; defaults: one slice on one rank, two with peers
    (from 65,536 Gaussians on: ADVICE round 4)."""
    from gs_dp import ORDER, FlatGaussianParams, project_slice_size

    shapes = [(n, 3), (n, 4), (n, 3), (n,), (n, 3)]
    flat = FlatGaussianParams([torch.zeros(s) for s in shapes], world_size=world, rank=world - 1,
                              exchange="reduce_scatter", n_slices=n_slices)
    q = 4 * world
    assert flat.n_pad % q == 0 and 0 <= flat.n_pad - n < q
    assert [tuple(p.shape) for p in flat.params] == shapes and all(p.is_contiguous() for p in flat.params)
    per = project_slice_size(n)
    assert per % 256 == 0 and -(-n // per) <= 256
    b = flat.slice_bounds
    assert b[0] == 0 and b[-1] == flat.n_pad and all(x < y for x, y in zip(b, b[1:]))
    assert all(x % per == 0 and x % q == 0 for x in b[:-1])
    if n_slices is None:
        assert flat.n_slices == (2 if (n >= 65_536 and world > 1) else 1)
    else:
        assert 1 <= flat.n_slices <= n_slices
    covered = torch.zeros(flat.flat_grad.numel(), dtype=torch.int32)
    for k in range(flat.n_slices):
        g0, g1 = flat.slice_gaussians(k)
        assert g0 % 256 == 0 and g0 <= g1 <= n
        for (lo, hi), (slo, shi), t in zip(flat.slice_ranges(k), flat.owned(flat.slice_ranges(k)), ORDER):
            covered[lo:hi] += 1
            assert lo % 4 == 0 and (hi - lo) % q == 0 and (shi - slo) * world == hi - lo and slo % 4 == 0
            assert flat.region[t] <= lo and (hi - flat.region[t]) % flat.width[t] == 0
    assert bool((covered == 1).all())
    assert flat.group_ends[-1] == flat.flat_grad.numel() and all(e % 4 == 0 for e in flat.group_ends)


def test_training_block_length_rule():
    """tools/train_timing.py: 25 iterations per timed block, or what ~30 ms of work take if that is more (capped)."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from train_timing import block_length

    assert block_length(1.34e-3) == 25 and block_length(4e-3) == 25
    assert block_length(0.56e-3) == 54 and block_length(0.2e-3) == 150 and block_length(1e-5) == 200


def _stat_worker(rank, world, port, tmp, mode):
    import torch.distributed as dist
    from gs_dp import ViewParallelGradStat

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    n = 1000
    st = ViewParallelGradStat(n, "cpu", mode, world_size=world)
    rng = np.random.default_rng(7 + rank)
    # what update() leaves behind on a GPU (the HIP launch itself is covered by tests/test_gpu_train.py)
    st.accum.copy_(torch.from_numpy(np.abs(rng.normal(size=(n, 3))).astype(np.float32)))
    st.counter.copy_(torch.from_numpy(rng.integers(0, 5, n).astype(np.float32)))
    np.save(os.path.join(tmp, f"stat_local_{rank}.npy"), st._buf.numpy().copy())
    acc, cnt = st.reduce()
    assert acc.data_ptr() == st.accum.data_ptr() and cnt.data_ptr() == st.counter.data_ptr()  # in place
    np.save(os.path.join(tmp, f"stat_red_{rank}.npy"), st._buf.numpy().copy())
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["max", "mean"])
def test_view_parallel_densification_statistic_gloo(tmp_path, mode):
    """train.py:145-154 under view parallelism: the ranks' per-view |grad| statistics are combined with ONE
    collective (elementwise max, or sum of statistic + visibility counter) and every rank ends up with the same
    numbers -- the precondition for identical prune / clone / split decisions."""
    world, port = 2, 31500 + (os.getpid() % 2000)
    mp.spawn(_stat_worker, args=(world, port, str(tmp_path), mode), nprocs=world, join=True)
    local = [np.load(tmp_path / f"stat_local_{r}.npy") for r in range(world)]
    red = [np.load(tmp_path / f"stat_red_{r}.npy") for r in range(world)]
    expect = np.maximum(local[0], local[1]) if mode == "max" else local[0] + local[1]
    assert np.array_equal(red[0], expect) and np.array_equal(red[1], expect)


def _world8_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gs_dp import ORDER, FlatGaussianParams, ViewParallelGradStat, project_slice_size

    n = 4099  # 17 project slices of 256; regions padded to 4,128 rows (a multiple of 4 x 8)
    shapes = [(n, 3), (n, 4), (n, 3), (n,), (n, 27)]
    params = [torch.from_numpy(np.random.default_rng(5).normal(size=s).astype(np.float32)) for s in shapes]  # replicas
    out, apis = {}, {}
    for exchange, n_slices in (("all_reduce", 1), ("all_reduce", 4), ("reduce_scatter", 4)):
        flat = FlatGaussianParams(params, world_size=world, exchange=exchange, n_slices=n_slices)
        assert flat.rank == rank and flat.n_pad == 4128 and project_slice_size(n) == 256
        if n_slices == 4:
            assert flat.n_slices == 4 and all(b % 32 == 0 and b % 256 == 0 for b in flat.slice_bounds[:-1])
            apis[exchange] = flat.self_check()  # the closed-form pattern over eight ranks before the path is trusted
        for step in range(2):
            # integer-valued gradients: every order of adding eight of them is exact, so all paths must agree bit for bit
            g = np.random.default_rng(1000 * step + rank).integers(-8, 9, flat.flat_grad.numel()).astype(np.float32)
            flat.flat_grad.copy_(torch.from_numpy(g))
            for t in ORDER:  # pad rows carry zero gradient
                flat.flat_grad[flat.offsets[t][1]:flat.region[t] + flat.n_pad * flat.width[t]].zero_()
            flat.finish_gather()
            for k in range(flat.n_slices):
                flat.begin_slice(k)
            for k in range(flat.n_slices):
                flat.finish_slice(k)
                for (lo, hi), (slo, shi) in zip(flat.slice_ranges(k), flat.owned(flat.slice_ranges(k))):
                    if exchange == "reduce_scatter":  # this rank's eighth of the range
                        assert (shi - slo) * world == hi - lo and slo == lo + rank * (shi - slo) and slo % 4 == 0
                    flat.flat_param[slo:shi].sub_(0.125 * flat.flat_grad[slo:shi])
                flat.begin_slice_gather(k)
                if k:
                    flat.finish_slice_gather(k - 1)
        flat.finish_gather()
        out[(exchange, n_slices)] = flat.flat_param.numpy().copy()
    stats = {}
    for mode in ("max", "mean"):
        st = ViewParallelGradStat(500, "cpu", mode, world_size=world)
        rng = np.random.default_rng(7 + rank)
        st.accum.copy_(torch.from_numpy(rng.integers(0, 64, (500, 3)).astype(np.float32)))
        st.counter.copy_(torch.from_numpy(rng.integers(0, 5, 500).astype(np.float32)))
        local = st._buf.numpy().copy()
        st.reduce()
        stats[mode] = (local, st._buf.numpy().copy())
    np.savez(os.path.join(tmp, f"w8_{rank}.npz"), unsliced=out[("all_reduce", 1)], sliced=out[("all_reduce", 4)],
             sharded=out[("reduce_scatter", 4)], max_local=stats["max"][0], max_red=stats["max"][1],
             mean_local=stats["mean"][0], mean_red=stats["mean"][1],
             apis=np.array([apis["all_reduce"], apis["reduce_scatter"]]))
    dist.destroy_process_group()


def test_world8_slice_pipeline_and_statistic_gloo(tmp_path):
    """The rank count of the target node: EIGHT gloo ranks run the slice
    pipeline -- Np = a multiple of 4 x 8 rows, four slices of whole project slices, eighth-shards of every range -- in
    both exchange modes after the grouped-collective self-check, and ViewParallelGradStat.reduce in both modes.  With
    integer-valued gradients every summation order is exact: sliced == unsliced == sharded bit for bit on every rank, all
    replicas agree, and the update equals the closed form (parameters - 0.125 x the mean of the ranks' gradients)."""
    world, port = 8, 41500 + (os.getpid() % 2000)
    mp.spawn(_world8_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"w8_{k}.npz") for k in range(world)]
    for k in range(world):
        assert np.array_equal(r[k]["unsliced"], r[k]["sliced"]) and np.array_equal(r[k]["unsliced"], r[k]["sharded"])
        assert np.array_equal(r[0]["unsliced"], r[k]["unsliced"])
        assert list(r[k]["apis"]) == list(r[0]["apis"])  # every rank settled on the same API
    assert set(r[0]["apis"]) <= {"grouped", "public"}
    from gs_dp import FlatGaussianParams

    n = 4099
    shapes = [(n, 3), (n, 4), (n, 3), (n,), (n, 27)]
    params = [torch.from_numpy(np.random.default_rng(5).normal(size=s).astype(np.float32)) for s in shapes]
    flat = FlatGaussianParams(params, world_size=world, rank=0)
    expect = flat.flat_param.numpy().copy()
    for step in range(2):
        tot = sum(np.random.default_rng(1000 * step + k).integers(-8, 9, expect.size).astype(np.float32) for k in range(world))
        expect = (expect - np.float32(0.125) * (tot / np.float32(world))).astype(np.float32)  # the exchange leaves the MEAN
    live = np.zeros(expect.size, bool)
    for t, (lo, hi) in flat.offsets.items():
        live[lo:hi] = True
    assert np.array_equal(r[0]["unsliced"][live], expect[live])
    assert np.array_equal(r[0]["unsliced"][~live], flat.flat_param.numpy()[~live])  # pad rows never move
    want_max = np.maximum.reduce([r[k]["max_local"] for k in range(world)])
    want_sum = np.add.reduce([r[k]["mean_local"] for k in range(world)])
    for k in range(world):
        assert np.array_equal(r[k]["max_red"], want_max) and np.array_equal(r[k]["mean_red"], want_sum)


def test_camera_shift_estimate_of_the_cull_policy():
    """FrameRenderer._camera_shift_px (host arithmetic only): how far image content moved, in pixels, between the pose the
    occlusion cuts were recorded under and the current one -- 0 for the identical pose, rotation angle x focal length for a
    small yaw (the first version's arccos returned 0 below 0.03 degree), + translation at unit depth, infinite when the image
    geometry differs.  The cull policy's thresholds (0.5 px: near factor; 8 px: no cull beyond) are in these units."""
    import numpy as np

    from gs_frame import FrameRenderer
    from gs_scene import make_camera

    def key(cam):
        return (int(cam.width), int(cam.height), float(cam.focal_x), float(cam.focal_y), float(cam.near),
                np.asarray(cam.rot, np.float32).tobytes(), np.asarray(cam.tran, np.float32).tobytes())

    r = object.__new__(FrameRenderer)  # (no device: only the two keys the method reads)
    base = make_camera(1920, 1080)
    f = float(base.focal_x)
    r._cut_ck = key(base)
    r._cur_ck = key(make_camera(1920, 1080))
    assert r._camera_shift_px(base) == 0.0
    for deg in (0.001, 0.01, 0.05, 0.2, 2.0):
        r._cur_ck = key(make_camera(1920, 1080, yaw_deg=deg))
        want = f * np.deg2rad(deg)
        got = r._camera_shift_px(base)
        assert abs(got - want) <= 0.02 * want + 1e-3, (deg, got, want)
    moved = make_camera(1920, 1080)
    moved.tran = np.asarray(moved.tran, np.float32) + np.array([0.002, 0.0, 0.0], np.float32)
    r._cur_ck = key(moved)
    assert abs(r._camera_shift_px(base) - 0.002 * f) < 0.05
    r._cur_ck = key(make_camera(1280, 720))
    assert r._camera_shift_px(base) == float("inf")
    r._cut_ck = None
    assert r._camera_shift_px(base) == float("inf")
    assert FrameRenderer.CULL_NEAR_SHIFT_PX < FrameRenderer.CULL_MAX_SHIFT_PX
