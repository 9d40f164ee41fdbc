#!/usr/bin/env python3
"""bench.py -- render FPS of the MI355X rasterizer path on BASELINE.json's configs.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python bench.py --gpus N ...            # no RANK in the environment: re-executes itself under torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE forward frame of the whole hot path (cull + project -> tile binning -> per-tile depth sort ->
16x16 compositing -> clamp + crop) on synthetic Gaussians already resident in HBM, ONE frame in flight (the
reference renders a frame and synchronises, train.py:256-281).  Headline workload (`--config cfg5`, the default):
the north-star target scene -- 2.4 M Gaussians at 1920x1080, forward render, the geometry of BASELINE.json
configs[3] / [4] (target >= 160 FPS on one MI355X).  With N > 1 every rank renders its own view (yaw k x 5 deg) of
a full replica of the scene: the path shards by view and a forward render has no exchange step, so there is no
data-path collective (weak scaling); the training step -- which does exchange gradients -- is reported per N under
"multi_gpu".

Rank 0 prints ONE JSON line (driver contract):
  value / ms_per_step  the K timed steps of the headline workload, one frame in flight.  The K-step block (barrier +
                       synchronize on both sides, max over ranks) is repeated R times ("repeats": R >= 15 and >= 0.3 s
                       of GPU work, so that an external sampler can see the run); value / ms_per_step are the MEDIAN
                       block, "ms_per_step_min" / "_max" its spread;
  latency_fps          the reference's protocol (train.py:259-266): events around ONE frame, synchronise, repeat;
  roofline             (`frac_executed`: the same fraction on the list steps the kernel executed)
                       dominant kernel of that workload (raster_forward_kernel), algorithmic bytes of SURVEY.md 8d S5
                       / its hipEvent-timed duration (events on the launch stream, inside the library); `traffic` = HBM-side
                       bytes per launch from the committed PMC passes (profiles/traffic.json) with `correction` = how
                       FETCH_SIZE was turned into bytes for this kernel's access pattern and `traffic_bounds`;
  stages               every stage of the frame: ms, algorithmic bytes, GB/s, fraction of HBM peak;
  cfg1                 BASELINE.json configs[0] (10 k Gaussians, 256 x 256, forward): GPU FPS next to the CPU oracle timed on
                       the SAME scene (BASELINE.md section 3 plans exactly this pair);
  cfg2                 BASELINE.json configs[1] (376,467 Gaussians, 1080p): FPS + the same roofline object;
  extra                training step (forward + L1/SSIM loss + backward + fused Adam; on one rank with rgb colours the Adam
                       step runs inside the backward's last kernel: `adam_fused_into_backward`) at cfg2 and at 2.4 M Gaussians -- timed
                       with tools/train_timing.py: the same iterations per block restored from a snapshot, 15 blocks.
                       `iters_per_s` is the MOVING scene (the reference's learning rates: what a training run sees; round 5,
                       ADVICE round 4), `fixed_scene` the same step with learning rate 0 (the full step runs, the parameters
                       stay put: the repeatability diagnostic, and the scene the render figures are quoted on), each with its
                       per-stage split and `moving_minus_fixed_ms` --, a 300-iteration fit of the cfg3 scene, cfg4
                       (2.4 M, SH) forward / backward stage times and the wall clock of the free-running loop -- its
                       `roofline_raster_backward_kernel` also carries `mfma`: the fp32 MFMA flops of the SH backward on
                       the matrix pipe, counted from the pixel-row steps the kernel EXECUTED (a device counter: rows whose
                       pixels had all stopped are left out) / its time against the 157.3 TFLOP/s matrix peak --, `soak`:
                       the densifying training run of tools/soak.py (376 k Gaussians growing, rgb / SH degree 2 / 3), rate
                       per block of 100 iterations --, `trained_state`: a deterministic translucent, heavy-tailed scene in the
                       state a trained model is in (gs_scene.make_trained_like_scene), rgb and SH degree 2: FPS, forward +
                       backward it/s, stage times, compositing against the roofline by EXECUTED steps, with the long-list
                       flag auto / off / on --, three frames in flight;
  multi_gpu            (under torchrun, or with --force-collective on one rank) ranks seen, gradient buffer bytes, and per
                       scene (rgb, SH) and exchange mode: training views/s over all ranks, the exchange alone, bus
                       bandwidth, and exposed_ms = the step with its exchange minus the same step without it in the same
                       process; the two-slice pipeline's figure; with peers also one and four slices;
  cpu_baseline         the C oracle (CPU port of the reference path, OpenMP over every host core) on the same scene.
`--legs` selects what runs (default: everything that fits the rank count); profiles/ holds rocprofv3 traces of
`--legs headline`.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "3d-gaussian-splatting_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(ROOT, "tools"))  # train_timing.py, compat_fps.py (measurement helpers, not product code)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALL_LEGS = ("headline", "cfg1", "cfg2", "train", "fit", "cfg4", "trained", "soak", "pipelined", "compat", "multi_gpu", "cpu")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def stage_table(n, V, M, W, H, C, stage_ms):
    """Per-stage algorithmic bytes and the HBM fraction they were moved at.

    project   = S1 cull/project (R 40 N, W 4 N + 28 V) + S1b activations (R and W (4 + 4 C) V)
    bin       = S2 key emission (R 28 V, W 12 M: 8-byte key + 4-byte id)
    tile_sort = S3 sort, single-pass lower bound (R 12 M + W 12 M) + S4 ranges (R 4 M, W 4 T)
    raster    = S5 (R (32 + 4 C) M, W 12 P)
    """
    P = (-(-W // 16) * 16) * (-(-H // 16) * 16)
    T = P // 256
    alg = {"project": 44 * n + 28 * V + 2 * (4 + 4 * C) * V, "bin": 28 * V + 12 * M,
           "tile_sort": 28 * M + 4 * T, "raster": (32 + 4 * C) * M + 12 * P}
    out = {}
    for k, b in alg.items():
        ms = stage_ms[k]
        gbs = b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        out[k] = {"ms": round(ms, 4), "algorithmic_bytes": int(b), "GBs": round(gbs, 1),
                  "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
    return out, alg, P, T


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg5", help="headline workload: cfg1..cfg5 of gs_scene.CONFIGS")
    ap.add_argument("--legs", default="all", help="comma list of " + ",".join(ALL_LEGS) + " (default: all)")
    ap.add_argument("--no-extra", action="store_true", help="= --legs headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1: still run the gradient all-reduce (through RCCL when launched under torchrun)")
    ap.add_argument("--quick", action="store_true",
                    help="test hook (tests/test_gpu_train.py runs this file's multi_gpu leg under two ranks): three blocks "
                         "per timing instead of 15, the multi_gpu leg on the headline scene only -- not a measurement")
    args = ap.parse_args()
    # test hooks: GS_BENCH_BACKEND=gloo + GS_BENCH_SHARE_GPU=1 run N ranks on ONE device (RCCL refuses two ranks on a
    # device; gloo stages device tensors through the host) so that the N > 1 code path of this file -- rank-symmetric
    # legs, barriers, max over ranks, the multi_gpu leg -- is executed on the one-GPU boxes the builder has
    backend = os.environ.get("GS_BENCH_BACKEND", "nccl")
    share_gpu = os.environ.get("GS_BENCH_SHARE_GPU", "") == "1"
    legs = set(ALL_LEGS if args.legs == "all" else args.legs.split(","))
    if args.no_extra:
        legs = {"headline"}
    if args.no_cpu_baseline:
        legs.discard("cpu")
    assert legs <= set(ALL_LEGS), f"unknown leg in {sorted(legs)}"

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (there is no CPU fallback)")
    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")
    if torch.cuda.device_count() < args.gpus and not share_gpu:
        sys.exit(f"bench.py --gpus {args.gpus} needs {args.gpus} visible devices, found {torch.cuda.device_count()}")
    if args.gpus > 1 and "RANK" not in os.environ:
        # launched blind (`python bench.py --gpus N`): one rank per GPU under torch.distributed.run, rank 0 prints the
        # JSON line on the inherited stdout
        import socket
        import subprocess

        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        log("[bench] no RANK in the environment: " + " ".join(cmd))
        sys.exit(subprocess.call(cmd, env=env))

    # stdout carries exactly ONE line, the JSON: whatever libraries print to file descriptor 1 while the benchmark runs
    # (RCCL's version banner, for one) is sent to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under torchrun always go through RCCL; a single process does when the collective is asked for (--force-collective)
    use_dist = world > 1 or "RANK" in os.environ or args.force_collective
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket

            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(so.getsockname()[1])
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from gs_frame import FrameRenderer
    from gs_scene import CONFIGS, make_camera, make_scene

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def load(cfg, sh_degree=2):
        n, W, H, use_sh = CONFIGS[cfg]
        scene = make_scene(n, W, H, seed=2023, use_sh=use_sh, sh_degree=sh_degree)
        cam = make_camera(W, H, yaw_deg=5.0 * rank)  # one view per GPU
        params = [torch.from_numpy(a).to(dev) for a in (scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb)]
        return scene, cam, params

    def sized_renderer(params, cam, training):
        r = FrameRenderer(dev, max_pairs=1 << 20, training=training, auto_grow=True)
        r.forward(*params, cam)  # grows the workspace until the frame fits
        st = r.stats()
        r.max_pairs = int(st.pairs * 1.1) + 4096
        r.auto_grow = False  # from here on: no host synchronisation inside a frame
        r.forward(*params, cam)
        # (V, M of the FRAME: the first forward of a workspace lists every pair; the inference frames after it are
        # occlusion-culled -- GS_FRAME_OCCLUSION_CULL -- and emit fewer: render_leg reports that count as well)
        return r, st

    def time_block(fn, steps):
        """EXACTLY `steps` steps between barrier + synchronize on both sides; max over ranks (seconds)."""
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        barrier()
        return max_over_ranks(dt)

    def time_frames(fn, steps, warmup, repeats=None, min_seconds=0.3):
        """`warmup` untimed steps, then the `steps`-step block R times: returns (median block seconds, all blocks).
        R = max(15, enough blocks for `min_seconds` of work), decided on rank 0's first block and shared."""
        for _ in range(warmup):
            fn()
        first = time_block(fn, steps)
        if repeats is None:
            repeats = 3 if args.quick else int(min(400, max(15, min_seconds / max(first, 1e-6) + 1)))
        blocks = [first] + [time_block(fn, steps) for _ in range(repeats - 1)]
        return statistics.median(blocks), blocks

    def settle(fn, seconds=0.4):
        """setup, not measurement: a few tenths of a second of work so that the clocks (DVFS) are at their steady
        state whatever --warmup the caller picked"""
        t = time.perf_counter()
        while time.perf_counter() - t < seconds:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()

    def latency_fps(frame, frames=60):
        """The reference's FPS protocol (train.py:259-266): an event in front of ONE frame, an event behind it, synchronise,
        repeat -- frames/s = 1 / the median per-frame span.  Every frame starts on an idle device: no launch of frame k + 1
        overlaps the tail of frame k, and the host's issue time of the first launch is inside the span."""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        spans = []
        for _ in range(frames + 5):
            torch.cuda.synchronize()
            e0.record()
            frame()
            e1.record()
            e1.synchronize()
            spans.append(e0.elapsed_time(e1))
        spans = spans[5:]
        med = statistics.median(spans)
        return {"fps": round(1e3 / med, 1), "ms_per_frame": round(med, 4), "ms_min": round(min(spans), 4),
                "ms_p90": round(sorted(spans)[int(0.9 * len(spans))], 4), "frames": len(spans),
                "protocol": "hipEvent before / after ONE forward, synchronise, repeat (the span of train.py:259-266); median"}

    def render_leg(cfg, steps, warmup):
        """One workload, one frame in flight: FPS over `steps` timed frames + roofline objects (rank 0)."""
        scene, cam, params = load(cfg)
        n, W, H, use_sh = CONFIGS[cfg]
        r, st = sized_renderer(params, cam, training=False)
        log(f"[rank {rank}] {cfg}: N={n} V={st.visible} M={st.pairs} {W}x{H} sh={use_sh}")
        frame = lambda: r.forward(*params, cam)  # noqa: E731
        settle(frame)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            frame()
        host_us = (time.perf_counter() - t0) / 50 * 1e6  # host cost of issuing one frame, GPU free-running
        torch.cuda.synchronize()
        dt, blocks = time_frames(frame, steps, warmup)
        lat = latency_fps(frame)
        st_run = r.stats()  # the steady-state frame: pairs EMITTED (after the occlusion cull), did it fall back
        st_run_culled = bool(r._frame.flags & 256)
        # the same workload with a camera that MOVES every frame (a viewer's pan: 0.01 degree per frame, a quarter pixel):
        # the host rebuilds the frame descriptor every frame, the occlusion cull works from the cuts of every tile's 3 x 3
        # neighbourhood (GS_FRAME_CULL_DILATE, the host-side rule for poses within 8 px of the recorded one)
        W_, H_ = CONFIGS[cfg][1], CONFIGS[cfg][2]
        pan = [make_camera(W_, H_, yaw_deg=5.0 * rank + 0.01 * i) for i in range(40)]
        mv = [0]

        def moving_frame():
            mv[0] += 1
            r.forward(*params, pan[mv[0] % len(pan)] if (mv[0] // len(pan)) % 2 == 0 else pan[-1 - mv[0] % len(pan)])

        dt_mv, _ = time_frames(moving_frame, steps, warmup, repeats=15)
        moving_culled, moving_dilated = bool(r._frame.flags & 256), bool(r._frame.flags & 512)
        r.forward(*params, cam)
        # ... and the camera at rest WITHOUT the occlusion cull (the same renderer, the feature switched off): what `value`
        # measured up to round 5 -- reported next to it so that nobody has to take the cull's share on trust
        fps_no_cull = stages_no_cull = None
        if st_run_culled:
            keep_cull = r.occlusion_cull
            r.occlusion_cull = False
            settle(frame, 0.2)
            dt_nc, _ = time_frames(frame, steps, warmup, repeats=15)
            fps_no_cull = round(world * steps / dt_nc, 2)
            if rank == 0:  # ... with its stage times (hipEvents, as `stages`): what every unculled and every training frame pays
                pnc = [r.profile_forward(*params, cam) for _ in range(12)][4:]
                stages_no_cull = {"project": round(statistics.median(x["project"] for x in pnc), 4),
                                  "bin": round(statistics.median(x["scan_emit"] + x["sort"] for x in pnc), 4),
                                  "tile_sort": round(statistics.median(x["ranges"] for x in pnc), 4),
                                  "raster": round(statistics.median(x["raster"] for x in pnc), 4)}
            r.occlusion_cull = keep_cull
            for _ in range(3):
                r.forward(*params, cam)
        res = {"fps": world * steps / dt, "ms": dt / steps * 1e3, "host_us": host_us, "stats": st, "latency": lat,
               "scene": scene, "cam": cam, "params": params, "renderer": r, "repeats": len(blocks),
               "occlusion_cull": {"active": st_run_culled, "pairs_emitted": st_run.pairs,
                                  "pairs_of_the_frame": st.pairs, "fell_back": st_run.cull_fallback,
                                  "fps_without_the_cull": fps_no_cull, "stage_ms_without_the_cull": stages_no_cull},
               "moving_camera": {"fps": round(world * steps / dt_mv, 2), "ms_per_frame": round(dt_mv / steps * 1e3, 4),
                                 "culled": moving_culled, "dilated_cuts": moving_dilated,
                                 "what": "the camera yaws 0.01 degree (0.25 px) per frame: a new frame descriptor per frame; "
                                         "occlusion cull from the cuts of every tile's 3 x 3 neighbourhood x 1.375 in depth"},
               "ms_min": min(blocks) / steps * 1e3, "ms_max": max(blocks) / steps * 1e3}
        if rank == 0:
            prof = [r.profile_forward(*params, cam) for _ in range(25)][5:]
            res["occlusion_cull"]["stage_times_of_culled_frames"] = bool(r._frame.flags & 256)
            med = {k: statistics.median(p[k] for p in prof) for k in prof[0]}
            # sort_mode 2: "scan_emit" = count + column scan + scatter (the tile binning), "ranges" = per-tile sort
            stage_ms = {"project": med["project"], "bin": med["scan_emit"] + med["sort"], "tile_sort": med["ranges"],
                        "raster": med["raster"]}
            C = 27 if use_sh else 3
            stages, alg, P, T = stage_table(n, st.visible, st.pairs, W, H, C, stage_ms)
            achieved = alg["raster"] / (stage_ms["raster"] * 1e-3) / 1e9
            # HBM bytes per launch and the SIMDs' VALU-issue occupancy from committed PMC passes (profiles/traffic.json),
            # if they belong to this workload
            traffic = issue_busy = correction = bounds = None
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[cfg]
                if tj["tile_pairs"] == st.pairs:
                    traffic = tj["raster_forward_kernel"]["traffic_bytes"]
                    issue_busy = tj["raster_forward_kernel"].get("issue_busy")
                    # how FETCH_SIZE was turned into bytes for THIS kernel's access pattern (round 4: calibrated per
                    # pattern, profiles/r04_fetch_calibration.json) and the bounds the truth lies between
                    correction = tj["raster_forward_kernel"].get("correction")
                    bounds = tj["raster_forward_kernel"].get("traffic_bytes_bounds")
            except (OSError, KeyError, ValueError):
                pass
            roof = {"bound": "hbm", "kernel": "raster_forward_kernel", "achieved": round(achieved, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "algorithmic_bytes": int(alg["raster"]),
                    # what rocprof's counters saw move, as a fraction of the HBM peak (tiles stop compositing once all
                    # their pixels are saturated, so this is BELOW frac on a scene with many hidden Gaussians)
                    "traffic_frac": None if traffic is None else round(
                        traffic / (stage_ms["raster"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "correction": correction, "traffic_bounds": bounds,
                    "issue_busy": issue_busy, "kernel_ms": round(stage_ms["raster"], 4), "workload": cfg}
            # the same fraction on the steps the kernel EXECUTED: `frac` credits every pair
            # of every list, but a tile stops reading its list once all its pixels have saturated -- composited steps x
            # the per-step bytes of SURVEY.md 8d S5 + the image written, over the same kernel time
            rt = FrameRenderer(dev, max_pairs=r.max_pairs, training=True, auto_grow=False)
            rt.forward(*params, cam)
            steps_done = rt.composited_steps()
            del rt
            exec_bytes = (32 + 4 * C) * steps_done + 12 * P
            roof["composited_steps"] = steps_done
            roof["executed_bytes"] = int(exec_bytes)
            roof["frac_executed"] = round(exec_bytes / (stage_ms["raster"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            roof["ns_per_executed_step"] = round(stage_ms["raster"] * 1e6 / max(steps_done, 1), 4)
            if not use_sh:
                # the compositing kernel is fp32-VALU bound, not HBM bound: 73 flops per Gaussian
                # per 4 pixels of a lane (5 shared + 34 per packed pixel pair) => 18.25 flops per (step, pixel), a step
                # being one Gaussian composited by one tile; tiles stop when all their pixels have (composited steps
                # <= pairs, counted by a training forward); peak = packed fp32 FMA, 256 CUs x 4 SIMD x 16 lanes, 2.4 GHz
                tf = 18.25 * 256 * steps_done / (stage_ms["raster"] * 1e-3) / 1e12
                roof["valu"] = {"achieved": round(tf, 1), "peak": 157.3, "unit": "TFLOP/s fp32 vector",
                                "frac": round(tf / 157.3, 3), "composited_steps": steps_done}
            b_fwd = 44 * n + (64 + 8 * C) * st.visible + (72 + 4 * C) * st.pairs + 12 * P + 4 * T
            res.update(roofline=roof, stages=stages, stage_total_ms=round(med["total"], 4),
                       frame_roofline={"algorithmic_bytes": int(b_fwd),
                                       "achieved_GBs": round(b_fwd / (res["ms"] * 1e-3) / 1e9, 1),
                                       "frac_of_hbm_peak": round(b_fwd / (res["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                       "note": "SURVEY.md 8d bytes of the WHOLE frame (all N Gaussians, all M pairs) over the "
                                               "measured frame time: an occlusion-culled frame moves fewer (pairs_emitted of M; "
                                               "only the projected Gaussians' records) -- an equivalent rate, not traffic"})
        return res

    def workload_name(cfg, st):
        n, W, H, use_sh = CONFIGS[cfg]
        what = {"cfg5": "north-star target scene (geometry of BASELINE configs[3]/[4])",
                "cfg2": "BASELINE configs[1]", "cfg3": "BASELINE configs[2] scene", "cfg4": "BASELINE configs[3]",
                "cfg1": "BASELINE configs[0]"}.get(cfg, "")
        return (f"{cfg}: {n} Gaussians, {W}x{H}, {'SH deg2 (27 coeff)' if use_sh else 'no SH'}, forward render, "
                f"one frame in flight, one view per GPU -- {what}")

    # ---------------------------------------------------------------- headline: render FPS, one frame in flight
    head = render_leg(args.config, args.steps, args.warmup)
    n, W, H, use_sh = CONFIGS[args.config]
    st = head["stats"]
    out = {
        "metric": "render_fps", "value": round(head["fps"], 2), "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(head["ms"], 4),
        "repeats": head["repeats"], "ms_per_step_min": round(head["ms_min"], 4),
        "ms_per_step_max": round(head["ms_max"], 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": workload_name(args.config, st), "n_gaussians": n, "visible": st.visible,
                   "tile_pairs": st.pairs, "width": W, "height": H,
                   "parallelism": f"view-sharded x{world} (no data-path collective)", "frames_in_flight": 1},
        "host_us_per_frame": round(head["host_us"], 1),
        # temporal occlusion cull (include/gs_abi.h, GS_FRAME_OCCLUSION_CULL): the steady-state frame emits / sorts only the
        # pairs in front of the depth at which the previous frame's tiles stopped; the image is bit-identical
        "occlusion_cull": head["occlusion_cull"],
        "moving_camera": head["moving_camera"],
        # the reference quotes FPS as one frame between two events with a synchronisation per frame (train.py:259-266);
        # `value` is throughput (K frames queued, one synchronisation) -- both, side by side
        "latency_fps": head["latency"]["fps"], "latency": head["latency"],
    }
    if rank == 0:
        out["roofline"], out["stages"] = head["roofline"], head["stages"]
        out["stage_total_ms"], out["frame_roofline"] = head["stage_total_ms"], head["frame_roofline"]

    extra = {}
    leg_errors = {}

    def guarded(name, fn):
        """Run one non-headline leg; an exception there must not cost the headline line (its message is reported).  Under
        several ranks the same exception is raised on every rank (the legs are rank-symmetric), so nobody is left
        waiting in a collective."""
        try:
            fn()
        except Exception as e:  # noqa: BLE001
            import traceback

            log(f"[rank {rank}] leg {name} failed: {e!r}\n{traceback.format_exc()}")
            leg_errors[name] = repr(e)[:300]
            torch.cuda.synchronize()
            torch.cuda.empty_cache()

    # ---------------------------------------------------------------- three frames in flight (throughput figure)
    if "pipelined" in legs:
        def _leg_pipelined():
            params, cam = head["params"], head["cam"]
            rs = [head["renderer"]] + [sized_renderer(params, cam, training=False)[0] for _ in range(2)]
            streams = [torch.cuda.Stream(device=dev) for _ in rs]
            counter = [0]
            for i in range(len(rs)):  # setup: the first launch on a fresh HIP stream creates its hardware queue (~ms)
                with torch.cuda.stream(streams[i]):
                    for _ in range(3):
                        rs[i].forward(*params, cam)
            torch.cuda.synchronize()

            def pipelined_frame():
                i = counter[0] % len(rs)
                counter[0] += 1
                with torch.cuda.stream(streams[i]):
                    rs[i].forward(*params, cam)

            k = max(args.steps, 50)
            dtp, _ = time_frames(pipelined_frame, k, 10)
            extra["three_frames_in_flight_fps"] = round(world * k / dtp, 2)
            del rs, streams

        guarded("pipelined", _leg_pipelined)
    head_scene, head_cam = head["scene"], head["cam"]
    head_params = head["params"]
    del head["renderer"]
    torch.cuda.empty_cache()

    # ---------------------------------------------------------------- BASELINE configs[0]: the reference's CPU-runnable case
    if "cfg1" in legs and args.config != "cfg1" and rank == 0 and world == 1:
        def _leg_cfg1():
            # 10 k Gaussians, 256 x 256, no SH, forward: the GPU path and the CPU oracle on the SAME scene (BASELINE.md
            # section 3: "CPU oracle (ours) on config 1" next to "MI355X, this repo, config 1").  A frame here is a chain of
            # five dependent launches of a few microseconds each: the figure is latency, not throughput.
            c1 = render_leg("cfg1", max(args.steps, 200), max(args.warmup, 20))
            n1, W1, H1, _ = CONFIGS["cfg1"]
            res = {"workload": workload_name("cfg1", c1["stats"]), "render_fps": round(c1["fps"], 2),
                   "ms_per_frame": round(c1["ms"], 4), "repeats": c1["repeats"],
                   "ms_per_frame_min": round(c1["ms_min"], 4), "ms_per_frame_max": round(c1["ms_max"], 4),
                   "visible": c1["stats"].visible, "tile_pairs": c1["stats"].pairs,
                   "host_us_per_frame": round(c1["host_us"], 1), "latency_fps": c1["latency"]["fps"],
                   "latency": c1["latency"], "binning_variant": c1["renderer"].binning_variant(),
                   "stages": c1["stages"], "roofline": c1["roofline"]}
            if "cpu" in legs:
                import oracle  # the checker, timed as the "port" baseline -- never on the product path
                from gs_geometry import RayBasis, TileGrid

                sc, cam = c1["scene"], c1["cam"]
                grid = TileGrid(W1, H1, cam.focal_x, cam.focal_y)
                rays = RayBasis.from_camera(cam.rot, cam.tran, grid.padded_height, grid.padded_width, cam.focal_x, cam.focal_y)
                t0, frames = time.perf_counter(), 0
                while time.perf_counter() - t0 < 3.0 or frames < 5:
                    oracle.render_forward(sc.pos, sc.quat, sc.scale, sc.opa, sc.rgb, cam.rot, cam.tran, cam.near, W1, H1,
                                          cam.focal_x, cam.focal_y, 0.05, use_sh=False, rays_o=rays.rays_o,
                                          lefttop=rays.lefttop, vdx=rays.dx, vdy=rays.dy)
                    frames += 1
                cpu_dt = time.perf_counter() - t0
                res["cpu_baseline"] = {"value": round(frames / cpu_dt, 3), "unit": "frames/s", "cores": oracle.num_threads(),
                                       "kind": "port", "sample": f"{frames} forward frames of cfg1 (oracle/gs_oracle.c)"}
                res["gpu_over_cpu"] = round(c1["fps"] / (frames / cpu_dt), 1)
            out["cfg1"] = res
            del c1
            torch.cuda.empty_cache()

        guarded("cfg1", _leg_cfg1)

    # ---------------------------------------------------------------- BASELINE configs[1]
    if "cfg2" in legs and args.config != "cfg2":
        def _leg_cfg2():
            c2 = render_leg("cfg2", max(args.steps, 50), max(args.warmup, 10))
            out["cfg2"] = {"workload": workload_name("cfg2", c2["stats"]), "render_fps": round(c2["fps"], 2),
                           "ms_per_frame": round(c2["ms"], 4), "repeats": c2["repeats"],
                           "ms_per_frame_min": round(c2["ms_min"], 4), "ms_per_frame_max": round(c2["ms_max"], 4),
                           "visible": c2["stats"].visible,
                           "tile_pairs": c2["stats"].pairs, "host_us_per_frame": round(c2["host_us"], 1),
                           "latency_fps": c2["latency"]["fps"], "latency": c2["latency"],
                           "occlusion_cull": c2["occlusion_cull"], "moving_camera": c2["moving_camera"]}
            if rank == 0:
                out["cfg2"].update(roofline=c2["roofline"], stages=c2["stages"], frame_roofline=c2["frame_roofline"])
            del c2
            torch.cuda.empty_cache()

        guarded("cfg2", _leg_cfg2)

    # ---------------------------------------------------------------- training step (train.py:84-185, no densification)
    def train_leg(cfg, params, cam, pairs, k, force=False, exchange="all_reduce", collective=True,
                  repeats=3 if args.quick else 15, n_slices=None, fixed_scene=True):
        """forward (checkpointing) -> L1 + 0.1 SSIM loss and gradient -> backward -> exchange of the flat gradient buffer
        (N > 1, or forced) -> fused Adam.  Timed with the snapshot / restore protocol of tools/train_timing.py: 30 warm-up
        iterations, then the SAME k iterations `repeats` times (everything the step changes is restored between blocks,
        outside the timed region).  `fixed_scene` (default): the learning rate is 0 -- every kernel of the step runs in full,
        Adam included, but the parameters stay where they are, so that the step is timed on the SAME synthetic scene the
        render figures are quoted on, whatever the block length; False: the reference's learning rates, i.e. the scene
        as 30 + k iterations against the noisy target have left it (lower opacities: tiles composite more Gaussians before
        they saturate; the step is 5 - 10 % slower at 2.4 M Gaussians and keeps slowing down).
        Returns (whole-job iterations/s, ms per iteration, detail dict on rank 0, trainer).
        `collective=False` (measurement only): the same step without its gradient exchange -- the reference the
        exchange's exposed time is taken against."""
        from gs_train import TrainOptions, Trainer
        from train_timing import time_training

        _, Wc, Hc, _ = CONFIGS[cfg]
        # target = the scene's own render + noise: a non-trivial loss gradient, but the scene stays where it is over the
        # timed iterations (a random target would drive it away and change the pair count under the fixed workspace)
        r0 = FrameRenderer(dev, max_pairs=int(pairs * 1.1) + 4096, auto_grow=False)
        target = (r0.forward(*params, cam)[0] + 0.05 * torch.randn(Hc, Wc, 3, device=dev)).clamp_(0, 1).contiguous()
        del r0
        tr = Trainer([t.clone() for t in params], [cam], [target], TrainOptions(lr=0.0) if fixed_scene else TrainOptions(),
                     world_size=world, max_pairs=int(pairs * 1.25) + 4096, exchange=exchange, n_slices=n_slices)
        tr.flat.force_collective = (use_dist or force) and dist.is_initialized()
        tr.flat.enable_collective = bool(collective)
        if args.quick:  # (test hook: a handful of steps -- under two gloo ranks on one GPU an exchange takes ~40 ms)
            k = 3
        dtt, blocks, k = time_training(tr, k, warm=10 if args.quick else 30, repeats=repeats, barrier=barrier,
                                       max_over_ranks=max_over_ranks)
        assert tr.renderer.overflowed_frames == 0 and not tr.renderer.last_frame_overflowed(wait=True)
        detail = {}
        if rank == 0:
            rt, flat = tr.renderer, tr.flat
            flat.finish_gather()
            rt.forward_abandon()
            img, _ = rt.forward(*flat.params, cam)
            lossk = tr._loss_for(Hc, Wc)
            pf = [rt.profile_forward(*flat.params, cam)["total"] for _ in range(8)][3:]
            pb = [rt.profile_backward(lossk(img, target)) for _ in range(8)][3:]
            detail["composited_steps"] = rt.composited_steps()  # (Gaussian, tile) steps of this scene state's frame
            detail["forward_ms"] = round(statistics.median(pf), 4)
            detail["backward_stage_ms"] = {key: round(statistics.median(p[key] for p in pb), 4) for key in pb[0]}

            def timed(fn, reps=20):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                fn()
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                e1.synchronize()
                return e0.elapsed_time(e1) / reps

            loss_ms, adam_ms = timed(lambda: lossk(img, target)), timed(tr.optimizer.step)
            n_par = flat.flat_param.numel()
            detail["loss_ms"], detail["adam_ms"] = round(loss_ms, 4), round(adam_ms, 4)
            # Adam is a pure stream: 16 B read + 12 B written per parameter
            detail["adam_roofline"] = {"bound": "hbm", "achieved": round(28 * n_par / (adam_ms * 1e-3) / 1e9, 1),
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(28 * n_par / (adam_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                       "parameters": n_par}
        detail["adam_fused_into_backward"] = bool(tr._can_fuse_adam())  # (one rank: gs_frame_backward_adam; GS_TRAIN_FUSE_ADAM=0 for A/B)
        detail.update(repeats=len(blocks), ms_per_iter_min=round(min(blocks) / k * 1e3, 4),
                      ms_per_iter_max=round(max(blocks) / k * 1e3, 4), iters_per_block=k,
                      scene="fixed (learning rate 0: the full step runs, the parameters stay put)" if fixed_scene else
                            "moving (the reference's learning rates, iterations 30 .. 30 + k against a noisy target)",
                      protocol="30 warm-up iterations, then the same k iterations per block: parameters, Adam moments "
                               "and step counter restored from a snapshot between blocks (tools/train_timing.py)")
        return world * k / dtt, dtt / k * 1e3, detail, tr

    def train_pair(cfg, params, cam, pairs, k, what):
        """The training step on the MOVING scene (headline: `iters_per_s`) and on the FIXED scene, with both stage splits."""
        ips_f, ms_f, det_f, tr = train_leg(cfg, params, cam, pairs, k)
        del tr
        ips_m, ms_m, det_m, tr = train_leg(cfg, params, cam, pairs, 25, fixed_scene=False)
        del tr
        res = {"iters_per_s": round(ips_m, 2), "ms_per_iter": round(ms_m, 4),
               "headline": "moving scene: the reference's learning rates, iterations 30 .. 55 against a noisy target -- what a "
                           "training run sees (ADVICE round 4); fixed_scene = the same step with learning rate 0",
               "iters_per_s_moving_scene": round(ips_m, 2), "iters_per_s_fixed_scene": round(ips_f, 2),
               **det_m, "fixed_scene": {"iters_per_s": round(ips_f, 2), "ms_per_iter": round(ms_f, 4), **det_f},
               "step": what}
        if rank == 0 and "backward_stage_ms" in det_m and "backward_stage_ms" in det_f:
            # where the moving scene's extra time goes (hipEvent stage times of one frame of each scene's final state)
            diff = {"forward": round(det_m["forward_ms"] - det_f["forward_ms"], 4),
                    "loss": round(det_m["loss_ms"] - det_f["loss_ms"], 4), "adam": round(det_m["adam_ms"] - det_f["adam_ms"], 4)}
            for key in det_m["backward_stage_ms"]:
                diff["backward." + key] = round(det_m["backward_stage_ms"][key] - det_f["backward_stage_ms"].get(key, 0.0), 4)
            res["moving_minus_fixed_ms"] = diff
            res["composited_steps"] = {"moving": det_m.get("composited_steps"), "fixed": det_f.get("composited_steps")}
        return res

    if "train" in legs:
        def _leg_train():
            k = None  # iterations per timed block: train_timing.block_length (25, or ~30 ms of work if that is more)
            _, cam2, params2 = (head_scene, head_cam, head_params) if args.config == "cfg2" else load("cfg2")
            r2, st2 = sized_renderer(params2, cam2, training=False)
            del r2
            extra["train_cfg2"] = train_pair("cfg2", params2, cam2, st2.pairs, k,
                                             "forward + L1/SSIM loss (w=0.1) + backward + grad all-reduce (N>1) + fused "
                                             "Adam, one view per GPU, 376,467 Gaussians, 1080p")
            del params2
            torch.cuda.empty_cache()
            if not CONFIGS[args.config][3]:
                extra["train_headline_scene"] = train_pair(args.config, head_params, head_cam, st.pairs, k,
                                                           f"the same step on the headline scene ({n} Gaussians)")
                torch.cuda.empty_cache()

        guarded("train", _leg_train)

    # ---------------------------------------------------------------- multi-GPU: the gradient exchange of a training step
    if "multi_gpu" in legs and not CONFIGS[args.config][3] and use_dist:
        def _leg_multi_gpu():
            # The training step of the view-parallel job: one view per rank, gradients exchanged once per iteration.
            # Both exchange modes of gs_dp.py -- (a) mean all-reduce + replicated fused Adam, (b) mean reduce-scatter +
            # Adam over the rank's shards + all-gather of the parameters -- as a pipeline over slices of the Gaussian
            # array (sums -> exchange -> Adam -> next frame's project stage, slice by slice), on the headline scene (rgb
            # colours) and on the cfg4 scene (degree-2 SH: 73 % of the exchanged bytes are coefficients).  Next to every
            # mode the SAME step without its exchange, in the same process: exposed_ms = what the exchange costs the step.
            k = None  # train_timing.block_length
            seen = torch.ones(1, device=dev)
            dist.all_reduce(seen)
            seen = int(seen.item())
            mg = {"ranks_seen": seen, "scenes": {}}

            def one_scene(tag, cfg_name, params_s, cam_s, pairs_s):
                res = {"modes": {}}
                _, plain_ms, _, tr = train_leg(cfg_name, params_s, cam_s, pairs_s, k, force=True, collective=False)
                res["plain_step_ms"] = round(plain_ms, 4)
                res["n_slices"] = tr.flat.n_slices
                del tr
                torch.cuda.empty_cache()
                for exchange in ("all_reduce", "reduce_scatter"):
                    views_per_s, ms, det, tr = train_leg(cfg_name, params_s, cam_s, pairs_s, k, force=True,
                                                         exchange=exchange)
                    flat = tr.flat
                    flat.finish_gather()
                    tr.renderer.forward_abandon()
                    res["bucket_bytes"] = flat.bucket_bytes
                    res["collective_api"] = "plain per-range all-reduces" if flat._plain_collectives() else flat._api()

                    def exchange_only():
                        # the collectives of one step, nothing else, as gs_train.Trainer.train_step issues them
                        for i in range(flat.n_slices):
                            flat.begin_slice(i)
                        for i in range(flat.n_slices):
                            flat.finish_slice(i)
                            flat.begin_slice_gather(i)
                        flat.finish_gather()

                    for _ in range(3):
                        exchange_only()
                    ex_ms = time_block(exchange_only, 10) / 10 * 1e3
                    res["modes"][exchange] = {
                        "train_views_per_s": round(views_per_s, 2), "train_ms_per_iter": round(ms, 4),
                        "train_ms_per_iter_min": det.get("ms_per_iter_min"), "train_ms_per_iter_max": det.get("ms_per_iter_max"),
                        "repeats": det.get("repeats"),
                        # what the exchange adds to the step of the same process without it
                        "exposed_ms": round(ms - plain_ms, 4),
                        "exchange_ms": round(ex_ms, 4),
                        "busbw_GBs": None if seen < 2 else round(2 * (seen - 1) / seen * flat.bucket_bytes / (ex_ms * 1e-3) / 1e9, 1),
                        "optimizer_state_bytes_per_rank": tr.optimizer.state_bytes}
                    res["modes"][exchange]["n_slices"] = flat.n_slices
                    del tr, flat
                    torch.cuda.empty_cache()
                # the same step as a TWO-slice pipeline whatever the rank count (one rank takes one slice by default:
                # nothing travels, nothing to hide): what the slicing itself costs -- per-slice kernels that fill less of
                # the chip, the extra launches and stream joins
                _, ms2, det2, tr = train_leg(cfg_name, params_s, cam_s, pairs_s, k, force=True, n_slices=2)
                res["two_slice_pipeline"] = {"n_slices": tr.flat.n_slices, "train_ms_per_iter": round(ms2, 4),
                                             "exposed_ms": round(ms2 - plain_ms, 4)}
                del tr
                torch.cuda.empty_cache()
                if seen >= 2:
                    # with peers the number of slices is a trade between hidden wire time and per-slice overhead that only
                    # a run on the real links can settle: one and four slices next to the default
                    res["slices_sweep_all_reduce"] = {}
                    for ns in (1, 4):
                        vps, msn, _, tr = train_leg(cfg_name, params_s, cam_s, pairs_s, k, force=True, n_slices=ns)
                        res["slices_sweep_all_reduce"][str(tr.flat.n_slices)] = {
                            "train_views_per_s": round(vps, 2), "train_ms_per_iter": round(msn, 4),
                            "exposed_ms": round(msn - plain_ms, 4)}
                        del tr
                        torch.cuda.empty_cache()
                mg["scenes"][tag] = res
                return res

            head_res = one_scene(args.config, args.config, head_params, head_cam, st.pairs)
            if not args.quick:
                _, cam4, p4 = load("cfg4")
                r4, st4 = sized_renderer(p4, cam4, training=False)
                del r4
                one_scene("cfg4_sh_deg2", "cfg4", p4, cam4, st4.pairs)
                del p4
                torch.cuda.empty_cache()
            best = max(head_res["modes"], key=lambda m: head_res["modes"][m]["train_views_per_s"])
            mg.update(modes=head_res["modes"], bucket_bytes=head_res["bucket_bytes"], best_mode=best,
                      train_views_per_s=head_res["modes"][best]["train_views_per_s"],
                      train_ms_per_iter=head_res["modes"][best]["train_ms_per_iter"],
                      exposed_ms=head_res["modes"][best]["exposed_ms"],
                      allreduce_ms=head_res["modes"]["all_reduce"]["exchange_ms"],
                      allreduce_busbw_GBs=head_res["modes"]["all_reduce"]["busbw_GBs"],
                      collective="per iteration, RCCL, one grouped launch per slice of the Gaussian array: all_reduce = "
                                 "mean all-reduce of the slice's five gradient ranges + replicated fused Adam; "
                                 "reduce_scatter = mean reduce-scatter + fused Adam over the rank's shards + all-gather "
                                 "of the parameters; the next frame's project stage follows the optimizer slice by slice",
                      backend=dist.get_backend(), collective_api=head_res.get("collective_api"),
                      note="1-GPU boxes only for the builder: no 2/4/8-GPU curve measured before the driver's SCALE run")
            out["multi_gpu"] = mg

        guarded("multi_gpu", _leg_multi_gpu)
    del head_params

    # ---------------------------------------------------------------- cfg3 in miniature: does the step train?
    if "fit" in legs and rank == 0 and world == 1:
        def _leg_fit():
            # BASELINE.json configs[2] in miniature: fit a perturbed copy of the cfg3 scene (506,627 Gaussians, 1080p)
            # to a render of the original with the reference's training step (train.py defaults: lr 0.003 x
            # (10, 10, 1, 1, 1), exp decay, L1 + 0.1 SSIM, Adam(0.9, 0.99)); there is no dataset here, so the
            # figure of merit is the PSNR against that synthetic ground truth before / after a short run
            from gs_train import TrainOptions, Trainer

            _, cam3, params3 = load("cfg3")
            _, W3, H3, _ = CONFIGS["cfg3"]
            r3, st3 = sized_renderer(params3, cam3, training=False)
            target3 = r3.forward(*params3, cam3)[0].clone()
            del r3
            g = torch.Generator(device=dev).manual_seed(7)
            start3 = [t.clone() for t in params3]
            start3[4] += 0.5 * torch.randn(start3[4].shape, device=dev, generator=g)   # colour logits
            start3[3] += 0.3 * torch.randn(start3[3].shape, device=dev, generator=g)   # opacity logits
            n_it = 300
            tr3 = Trainer(start3, [cam3], [target3], TrainOptions(n_iters=n_it + 1, n_iters_warmup=30),
                          max_pairs=int(st3.pairs * 1.2) + 4096)
            psnr0 = Trainer.psnr(tr3.renderer.forward(*tr3.flat.params, cam3)[0], target3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n_it):
                tr3.train_step(i, 0)
            torch.cuda.synchronize()
            dt3 = time.perf_counter() - t0
            psnr1 = Trainer.psnr(tr3.renderer.forward(*tr3.flat.params, cam3)[0], target3)
            extra["cfg3_fit"] = {"n_gaussians": CONFIGS["cfg3"][0], "tile_pairs": st3.pairs, "iters": n_it,
                                 "iters_per_s": round(n_it / dt3, 1), "psnr_before_dB": round(psnr0, 2),
                                 "psnr_after_dB": round(psnr1, 2),
                                 "final_loss": round(float(tr3._loss_for(H3, W3).values[0]), 5)}
            del tr3, target3, start3, params3
            torch.cuda.empty_cache()

        guarded("fit", _leg_fit)

    # ---------------------------------------------------------------- BASELINE configs[3]: 2.4 M Gaussians, SH, fwd + bwd
    if "cfg4" in legs and rank == 0 and world == 1:
        def _leg_cfg4():
            # hipEvent-timed stages.  Degree 2 (27 coefficients) is what the reference implements; degree 3 (48
            # coefficients, what configs[3] names) is the extension.  Roofline objects per SURVEY.md 8d:
            #   B_fwd = 44 N + (64 + 8 C) V + (72 + 4 C) M + 12 P + 4 T,  B_bwd = (32 + 4 C) M + 24 P + (100 + 12 C) V + (44 + 4 C) N
            # and for the dominant kernel of the backward (raster backward, S6): R (32 + 4 C) M + 24 P, W (28 + 4 C) V.
            cfg4 = {}
            try:
                traffic4 = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            except (OSError, ValueError):
                traffic4 = {}
            for deg in (2, 3):
                n4, W4, H4, _ = CONFIGS["cfg4"]
                _, cam4, p4 = load("cfg4", sh_degree=deg)
                r4, st4 = sized_renderer(p4, cam4, training=True)
                img4, _ = r4.forward(*p4, cam4)
                g4 = torch.sign(img4 - 0.5) / img4.numel()

                def fwd_bwd4():
                    r4.forward(*p4, cam4)
                    r4.backward(g4)

                settle(fwd_bwd4, 0.3)  # clocks at their steady state, as for the headline
                # wall clock of the free-running forward + backward loop (no events, no synchronisation inside a block),
                # next to the per-stage hipEvent profiles below (which synchronise per frame)
                wall_dt, wall_blocks = time_frames(fwd_bwd4, 10, 3, repeats=15)
                fw = [r4.profile_forward(*p4, cam4) for _ in range(8)][2:]
                bw = [r4.profile_backward(g4) for _ in range(8)][2:]
                f_ms = statistics.median(x["total"] for x in fw)
                b_ms = statistics.median(x["total"] for x in bw)
                rb_ms = statistics.median(x["raster_bwd"] for x in bw)
                C4 = 3 * (deg + 1) ** 2
                P4 = (-(-W4 // 16) * 16) * (-(-H4 // 16) * 16)
                V4, M4 = st4.visible, st4.pairs
                b_fwd = 44 * n4 + (64 + 8 * C4) * V4 + (72 + 4 * C4) * M4 + 12 * P4 + 4 * (P4 // 256)
                b_bwd = (32 + 4 * C4) * M4 + 24 * P4 + (100 + 12 * C4) * V4 + (44 + 4 * C4) * n4
                b_rbw = (32 + 4 * C4) * M4 + 24 * P4 + (28 + 4 * C4) * V4

                def roof(b, ms, **kw):
                    gbs = b / (ms * 1e-3) / 1e9
                    return {"bound": "hbm", "algorithmic_bytes": int(b), "ms": round(ms, 4), "achieved": round(gbs, 1),
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), **kw}

                tj = traffic4.get(f"cfg4_deg{deg}", {})
                # round 4: the SH backward of the frame path runs on the matrix pipe (raster_backward_mfma_sh_kernel);
                # a traffic.json recorded before that still carries the pixel-parallel kernel's entry
                bwd_kernel = "raster_backward_mfma_sh_kernel" if "raster_backward_mfma_sh_kernel" in tj \
                    else "raster_backward_pixel_sh_kernel"
                tk = tj.get(bwd_kernel, {}) if tj.get("tile_pairs") == M4 else {}
                tr_bw, busy_bw = tk.get("traffic_bytes"), tk.get("issue_busy")
                # its fp32 MFMAs: per 16 Gaussians x 16 pixels 3 x floor(NB / 4) (colour logits; degree 2's ninth basis
                # function is added on the VALU) + 12 (coefficient sums)
                # v_mfma_f32_16x16x4_f32 of 2,048 flops each, i.e. per composited (Gaussian, tile) step 1 / 16 of
                # that x 16 pixel rows; the matrix pipe's fp32 peak equals the vector peak (157.3 TFLOP/s) and on gfx950
                # an MFMA does not overlap the VALU stream of its SIMD (tools/ubench/mfma_valu_overlap.hip)
                steps4 = r4.composited_steps()
                r4.forward(*p4, cam4)
                r4.backward(g4)
                # pixel-row steps (16 Gaussians x 16 pixels) the kernel EXECUTED: a device counter per wave (round 5).  A
                # composited (Gaussian, tile) step is 16 pixel rows / 16 Gaussians = one row step if no row is skipped;
                # rows whose 16 pixels have all stopped are left out (the count by composited steps
                # overstated the flops by the skipped share)
                rows4 = r4.executed_row_steps()
                mfma_per_row = 3 * ((C4 // 3) // 4) + 12
                mfma_flops = rows4 * mfma_per_row * 2048
                cfg4[f"sh_degree_{deg}"] = {
                    "coefficients": C4, "visible": V4, "tile_pairs": M4,
                    "forward_ms": round(f_ms, 3), "backward_ms": round(b_ms, 3),
                    "raster_fwd_ms": round(statistics.median(x["raster"] for x in fw), 3),
                    "raster_bwd_ms": round(rb_ms, 3),
                    "project_bwd_ms": round(statistics.median(x["project_bwd"] for x in bw), 3),
                    "fwd_bwd_iters_per_s": round(1e3 / (f_ms + b_ms), 1),
                    "fwd_bwd_wall": {"ms_per_iter": round(wall_dt / 10 * 1e3, 4), "iters_per_s": round(10 / wall_dt, 1),
                                     "ms_per_iter_min": round(min(wall_blocks) / 10 * 1e3, 4),
                                     "ms_per_iter_max": round(max(wall_blocks) / 10 * 1e3, 4), "repeats": len(wall_blocks),
                                     "what": "free-running forward + backward loop, 10 iterations per block"},
                    "roofline_forward": roof(b_fwd, f_ms), "roofline_backward": roof(b_bwd, b_ms),
                    "roofline_fwd_bwd": roof(b_fwd + b_bwd, f_ms + b_ms),
                    "roofline_raster_backward_kernel": roof(
                        b_rbw, rb_ms, kernel=f"raster_backward_mfma_sh_kernel<{C4}>", traffic=tr_bw,
                        traffic_frac=None if tr_bw is None else round(tr_bw / (rb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        issue_busy=busy_bw, composited_steps=steps4,
                        traffic_source=tk.get("source"),
                        mfma={"bound": "mfma", "flops": int(mfma_flops), "executed_row_steps": rows4,
                              "row_steps_if_none_skipped": steps4,  # = composited (Gaussian, tile) steps (ragged groups aside)
                              "rows_skipped_frac": round(1.0 - rows4 / max(steps4, 1), 4),
                              "achieved": round(mfma_flops / (rb_ms * 1e-3) / 1e12, 2), "peak": 157.3, "unit": "TFLOP/s",
                              "frac": round(mfma_flops / (rb_ms * 1e-3) / 1e12 / 157.3, 4),
                              "what": "fp32 MFMA flops of the kernel / its time; MFMAs and VALU instructions of a SIMD "
                                      "do not overlap on gfx950, so this fraction and issue_busy share the same cycles"})}
                # inference render of the same SH scene, camera at rest: with the occlusion cull and without (same renderer)
                del r4
                ri, _ = sized_renderer(p4, cam4, training=False)
                fi = lambda: ri.forward(*p4, cam4)  # noqa: E731
                settle(fi, 0.3)
                dt_c, _ = time_frames(fi, 20, 5, repeats=10)
                culled4 = bool(ri._frame.flags & 256)
                ri.occlusion_cull = False
                settle(fi, 0.2)
                dt_n, _ = time_frames(fi, 20, 5, repeats=10)
                cfg4[f"sh_degree_{deg}"]["render"] = {"fps": round(20 / dt_c, 1), "occlusion_culled": culled4,
                                                      "fps_without_the_cull": round(20 / dt_n, 1),
                                                      "what": "inference frames of the SH scene, one in flight, camera at rest"}
                del ri, p4, img4, g4
                torch.cuda.empty_cache()
            extra["cfg4_2p4M_sh_fwd_bwd"] = cfg4

        guarded("cfg4", _leg_cfg4)

    # ---------------------------------------------------------------- the state a trained / densified model is in
    if "trained" in legs and rank == 0 and world == 1:
        def _leg_trained():
            # gs_scene.make_trained_like_scene: 724,312 translucent Gaussians with larger footprints and a heavy tail of
            # tile-list lengths (the end state of the densifying SH run of tools/soak.py, regenerated from a seed): every
            # pair of every list is composited, the longest lists are ten times the mean.  The reference's published
            # numbers are on trained models (README.md:34-48); the headline scene is the opaque generator of SURVEY.md 8d.
            # Per colour model: render FPS (throughput and the reference's synchronised-per-frame protocol), forward +
            # backward it/s (free running), stage times, and the compositing kernel against the HBM roofline on the steps it
            # EXECUTED -- with the long-list flags set by the renderer's own rule ("auto": GS_FRAME_LONG_SORT here), never, and
            # with GS_FRAME_LONG_LISTS (segmented compositing + the sort) forced on.
            from gs_scene import make_trained_like_scene

            Wt, Ht = 1920, 1080
            Pt = (-(-Wt // 16) * 16) * (-(-Ht // 16) * 16)
            ts = {}
            for tag, sh_t in (("rgb", False), ("sh_degree_2", True)):
                sc = make_trained_like_scene(width=Wt, height=Ht, use_sh=sh_t)
                cam_t = make_camera(Wt, Ht)
                p_t = [torch.from_numpy(a).to(dev) for a in (sc.pos, sc.quat, sc.scale, sc.opa, sc.rgb)]
                Ct = 27 if sh_t else 3
                res = {"n_gaussians": sc.n, "width": Wt, "height": Ht, "coefficients": Ct}
                for mode, ll in (("auto", None), ("flag_off", False), ("flag_on", True)):
                    r_i = FrameRenderer(dev, max_pairs=1 << 22, training=False, auto_grow=True, long_lists=ll)
                    r_i.forward(*p_t, cam_t)
                    st_i = r_i.stats()  # (auto: a longest list beyond LONG_LIST_FLAG_AT flags the following frames)
                    r_i.max_pairs, r_i.auto_grow = int(st_i.pairs * 1.1) + 4096, False
                    frame_i = lambda: r_i.forward(*p_t, cam_t)  # noqa: E731
                    settle(frame_i, 0.3)
                    dt_i, blocks_i = time_frames(frame_i, 20, 5)
                    pf = [r_i.profile_forward(*p_t, cam_t) for _ in range(10)][3:]
                    fwd = {k: round(statistics.median(x[k] for x in pf), 4) for k in pf[0]}
                    del r_i
                    # training renderer: checkpoints, forward + backward
                    r_t = FrameRenderer(dev, max_pairs=int(st_i.pairs * 1.1) + 4096, training=True, auto_grow=False,
                                        long_lists=ll)
                    r_t.forward(*p_t, cam_t)
                    st_t = r_t.stats()  # (auto: the renderer's rule flags the following frames from these counters)
                    img_t, _ = r_t.forward(*p_t, cam_t)
                    g_t = (torch.sign(img_t - 0.5) / img_t.numel()).contiguous()

                    def fwd_bwd_t():
                        r_t.forward(*p_t, cam_t)
                        r_t.backward(g_t)

                    settle(fwd_bwd_t, 0.3)
                    wall_dt, wall_blocks = time_frames(fwd_bwd_t, 10, 3, repeats=15)
                    r_t.forward(*p_t, cam_t)
                    steps_t = r_t.composited_steps()
                    tf = [r_t.profile_forward(*p_t, cam_t) for _ in range(8)][2:]
                    tb = [r_t.profile_backward(g_t) for _ in range(8)][2:]
                    tfwd = {k: round(statistics.median(x[k] for x in tf), 4) for k in tf[0]}
                    tbwd = {k: round(statistics.median(x[k] for x in tb), 4) for k in tb[0]}
                    flagged, flagged_sort = bool(r_t._frame.flags & 16), bool(r_t._frame.flags & (16 | 128))
                    del r_t, img_t, g_t
                    torch.cuda.empty_cache()
                    exec_b = (32 + 4 * Ct) * steps_t + 12 * Pt
                    res[mode] = {
                        "flagged_long_lists": flagged, "flagged_long_sort": flagged_sort, "visible": st_i.visible, "tile_pairs": st_i.pairs,
                        "longest_list": st_t.longest_list, "composited_steps": steps_t,
                        "render_fps": round(20 / dt_i, 1), "ms_per_frame": round(dt_i / 20 * 1e3, 4),
                        "ms_per_frame_min": round(min(blocks_i) / 20 * 1e3, 4),
                        "ms_per_frame_max": round(max(blocks_i) / 20 * 1e3, 4), "repeats": len(blocks_i),
                        "forward_stage_ms": fwd,
                        "fwd_bwd_iters_per_s": round(10 / wall_dt, 1), "fwd_bwd_ms_per_iter": round(wall_dt / 10 * 1e3, 4),
                        "fwd_bwd_ms_per_iter_min": round(min(wall_blocks) / 10 * 1e3, 4),
                        "fwd_bwd_ms_per_iter_max": round(max(wall_blocks) / 10 * 1e3, 4),
                        "training_forward_stage_ms": tfwd, "backward_stage_ms": tbwd,
                        # compositing on the executed steps: every list is walked to its end here, so this IS the
                        # algorithmic figure; ns per step next to the headline scene's
                        "raster_forward": {"bound": "hbm", "kernel_ms": fwd["raster"], "executed_bytes": int(exec_b),
                                           "achieved": round(exec_b / (fwd["raster"] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                           "unit": "GB/s", "frac_executed": round(exec_b / (fwd["raster"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                           "ns_per_executed_step": round(fwd["raster"] * 1e6 / max(steps_t, 1), 4),
                                           "training_ns_per_executed_step": round(tfwd["raster"] * 1e6 / max(steps_t, 1), 4)}}
                    if mode == "auto":
                        r_l = FrameRenderer(dev, max_pairs=int(st_i.pairs * 1.1) + 4096, auto_grow=False, long_lists=ll)
                        r_l.forward(*p_t, cam_t)
                        r_l.stats()  # (the renderer's own rule flags the following frames)
                        res[mode]["latency"] = latency_fps(lambda: r_l.forward(*p_t, cam_t), 30)
                        del r_l
                ts[tag] = res
                del p_t
                torch.cuda.empty_cache()
            extra["trained_state"] = ts

        guarded("trained", _leg_trained)

    # ---------------------------------------------------------------- training where it is slow: the densifying run
    if "soak" in legs and rank == 0 and world == 1:
        def _leg_soak():
            # tools/soak.py's training run: 376,467 Gaussians growing under the reference's densification schedule
            # (train.py:86-91, 141-190; clone + split every 100 iterations), nine views, a random one per iteration,
            # colours perturbed at the start -- opacities fall, Gaussians blow up, lists grow: the state a real run is in.
            # The rate is reported per block of 100 iterations (one device synchronisation per block).
            from soak import training_soak

            sk = {}
            for deg, iters in ((0, 3000), (2, 2000), (3, 2000)):  # (round 4 quoted these lengths: tools/soak.py)
                r_ = training_soak(dev, deg, iters)
                r_["iters_per_s_blocks"] = [round(x) for x in r_.get("iters_per_s_blocks", [])]  # (a stalled block shows)
                sk["rgb" if deg == 0 else f"sh_degree_{deg}"] = r_
                torch.cuda.empty_cache()
            extra["soak_densifying"] = sk

        guarded("soak", _leg_soak)

    # ---------------------------------------------------------------- zero-change integration mode
    if "compat" in legs and rank == 0 and world == 1:
        def _leg_compat():
            # the reference's own per-frame call sequence (splatter.py:562-641: T x MAXP table, cumsum, attribute gathers,
            # torch.sort, host syncs) over the drop-in gaussian / renderer modules, at BASELINE configs[1] and at the
            # north-star target scene; next to it the fused frame path the headline is measured on
            from compat_fps import measure as compat_measure

            extra["compat_mode"] = {c: compat_measure(c, dev, frames=10 if c == "cfg5" else 20) for c in ("cfg2", "cfg5")}
            torch.cuda.empty_cache()

        guarded("compat", _leg_compat)
    out["extra"] = extra
    if leg_errors:
        out["leg_errors"] = leg_errors

    # ---------------------------------------------------------------- CPU baseline (rank 0, N = 1 only)
    if "cpu" in legs and rank == 0 and world == 1:
        def _leg_cpu():
            import oracle  # the checker, timed here as the "port" baseline -- never on the product path
            from gs_geometry import RayBasis, TileGrid

            scene, cam = head_scene, head_cam
            grid = TileGrid(W, H, cam.focal_x, cam.focal_y)
            rays = RayBasis.from_camera(cam.rot, cam.tran, grid.padded_height, grid.padded_width, cam.focal_x, cam.focal_y)
            t0, frames = time.perf_counter(), 0
            while time.perf_counter() - t0 < 10.0 or frames < 2:
                oracle.render_forward(scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb, cam.rot, cam.tran,
                                      cam.near, W, H, cam.focal_x, cam.focal_y, 0.05, use_sh=use_sh, rays_o=rays.rays_o,
                                      lefttop=rays.lefttop, vdx=rays.dx, vdy=rays.dy)
                frames += 1
            cpu_dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": round(frames / cpu_dt, 4), "unit": "frames/s", "cores": oracle.num_threads(),
                                   "kind": "port",
                                   "sample": f"{frames} full forward frames of {args.config} (oracle/gs_oracle.c: cull + "
                                             f"project and the (tile, depth) sort on one core, compositing over "
                                             f"{oracle.num_threads()} OpenMP threads; {os.cpu_count()} host cores present)"}

        guarded("cpu", _leg_cpu)

    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
