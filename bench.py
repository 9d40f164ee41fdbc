#!/usr/bin/env python3
"""bench.py -- render FPS of the MI355X rasterizer path on BASELINE.json's configs.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one forward frame of the whole hot path (cull+project -> duplicate -> radix sort ->
tile ranges -> 16x16 compositing -> clamp+crop) on synthetic Gaussians already resident in
HBM.  Default workload: BASELINE.json configs[1] = 376,467 Gaussians at 1920x1080, no SH,
forward render.  With N > 1 every rank renders its own view (yaw k x 5 deg) of a full replica of
the scene -- the path shards by view and the forward render has no exchange step, so there is no
data-path collective (weak scaling).  The training leg reported under "extra" is the reference's
training step without densification (train.py:84-185: forward, L1 + 0.1 SSIM loss, backward, Adam)
and does have one: an RCCL all-reduce (mean) of the flat parameter-gradient bucket before the fused
Adam step.  "extra" also carries the loss / Adam kernel times, a 300-iteration fit of the cfg3 scene
(it/s, PSNR before/after against a synthetic ground truth), the cfg5 (2.4 M Gaussians) render FPS and the
cfg4 (2.4 M Gaussians with SH) forward / backward times.

Rank 0 prints ONE JSON line (driver contract) with "roofline" (dominant kernel =
raster_forward_kernel, timed live with hipEvents on its own stream inside the library) and
"cpu_baseline" (the C oracle = CPU port of the reference path, one host core).
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

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg2", help="cfg1..cfg5 of gs_scene.CONFIGS")
    ap.add_argument("--no-extra", action="store_true", help="skip the training / 2.4M legs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=3,
                    help="frames in flight: consecutive frames are independent, so frame i+1's latency-bound "
                         "binning/sort kernels overlap frame i's compositing on a second HIP stream")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ  # under torchrun always go through RCCL
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

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

    def load(cfg):
        n, W, H, use_sh = CONFIGS[cfg]
        scene = make_scene(n, W, H, seed=2023, use_sh=use_sh)
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
        return r, r.stats()

    def time_frames(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        barrier()
        return max_over_ranks(dt)

    # ---------------------------------------------------------------- headline: render FPS
    scene, cam, params = load(args.config)
    n, W, H, use_sh = CONFIGS[args.config]
    r, st = sized_renderer(params, cam, training=False)
    log(f"[rank {rank}] {args.config}: N={n} V={st.visible} M={st.pairs} {W}x{H} sh={use_sh}")
    # setup, not measurement: ~0.4 s of frames so that the clocks (DVFS) are at their steady state whatever
    # --warmup the caller picked; the W warm-up and K timed steps below follow the contract unchanged
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.4:
        for _ in range(50):
            r.forward(*params, cam)
        torch.cuda.synchronize()
    # host cost of issuing one frame (ctypes call + ~14 launches), GPU free-running
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        r.forward(*params, cam)
    host_us = (time.perf_counter() - t0) / 50 * 1e6
    torch.cuda.synchronize()
    dt1 = time_frames(lambda: r.forward(*params, cam), args.steps, args.warmup)
    single_stream_fps = world * args.steps / dt1
    if args.streams > 1:
        # one renderer (own workspace) per stream; frames alternate between them
        rs = [r] + [sized_renderer(params, cam, training=False)[0] for _ in range(args.streams - 1)]
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

        dt = time_frames(pipelined_frame, args.steps, args.warmup)
    else:
        dt = dt1
    ms_per_step = dt / args.steps * 1e3
    fps = world * args.steps / dt

    out = {
        "metric": "render_fps", "value": round(fps, 2), "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.config}: {n} Gaussians, {W}x{H}, "
                               f"{'SH deg2 (27 coeff)' if use_sh else 'no SH'}, forward render, one view per GPU",
                   "n_gaussians": n, "visible": st.visible, "tile_pairs": st.pairs, "width": W, "height": H,
                   "parallelism": f"view-sharded x{world} (no data-path collective)",
                   "frames_in_flight": args.streams},
        "single_stream_fps": round(single_stream_fps, 2), "host_us_per_frame": round(host_us, 1),
    }

    # ---------------------------------------------------------------- roofline of the dominant kernel
    if rank == 0:
        prof = [r.profile_forward(*params, cam) for _ in range(25)][5:]
        stage = {k: statistics.median(p[k] for p in prof) for k in prof[0]}
        C = 27 if use_sh else 3
        grid_px = (-(-W // 16) * 16) * (-(-H // 16) * 16)
        alg_bytes = (4 + 24 + 4 + 4 * C) * st.pairs + 12 * grid_px  # SURVEY.md 8d, stage S5 (raster)
        achieved = alg_bytes / (stage["raster"] * 1e-3) / 1e9
        traffic = None  # HBM bytes per launch from committed PMC passes (profiles/traffic.json), if they match
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[args.config]
            if tj["tile_pairs"] == st.pairs:
                traffic = tj["raster_forward_kernel"]["traffic_bytes"]
        except (OSError, KeyError, ValueError):
            pass
        out["roofline"] = {"bound": "hbm", "kernel": "raster_forward_kernel", "achieved": round(achieved, 1),
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                           "traffic": traffic, "algorithmic_bytes": alg_bytes,
                           "kernel_ms": round(stage["raster"], 4)}
        if not use_sh:
            # the kernel is fp32-VALU bound, not HBM bound: 73 flops per Gaussian per
            # 4 pixels of a lane (5 shared + 34 per packed pixel pair) => 18.25 flops per (pair, pixel);
            # peak = packed fp32 FMA on 256 CUs x 4 SIMD x 16 lanes at 2.4 GHz
            flops = 18.25 * 256 * st.pairs
            out["roofline"]["valu"] = {"achieved": round(flops / (stage["raster"] * 1e-3) / 1e12, 1), "peak": 157.3,
                                       "unit": "TFLOP/s fp32 vector",
                                       "frac": round(flops / (stage["raster"] * 1e-3) / 1e12 / 157.3, 3),
                                       "note": "upper bound on work: early-terminated tiles skip Gaussians"}
        out["stage_ms"] = {k: round(v, 4) for k, v in stage.items()}
        # whole-frame algorithmic bytes: 44N + (64+8C)V + (72+4C)M + 12P + 4T
        T = grid_px // 256
        b_fwd = 44 * n + (64 + 8 * C) * st.visible + (72 + 4 * C) * st.pairs + 12 * grid_px + 4 * T
        out["frame_roofline"] = {"algorithmic_bytes": b_fwd,
                                 "achieved_GBs": round(b_fwd / (ms_per_step * 1e-3) / 1e9, 1),
                                 "frac_of_hbm_peak": round(b_fwd / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    # ---------------------------------------------------------------- extra legs
    extra = {}
    if not args.no_extra:
        # training iteration of train.py:84-185 without densification: forward (checkpointing) -> L1 + SSIM loss
        # and its gradient -> backward -> RCCL all-reduce of the flat gradient bucket (N > 1) -> fused Adam
        from gs_train import TrainOptions, Trainer

        target = torch.rand(H, W, 3, device=dev)
        tr = Trainer(params, [cam], [target], TrainOptions(), world_size=world, max_pairs=int(st.pairs * 1.1) + 4096)
        tr.flat.force_collective = use_dist
        tr.renderer.auto_grow = False
        rt, flat = tr.renderer, tr.flat
        it = [0]

        def train_iter():
            tr.train_step(it[0], 0)
            it[0] += 1

        k = max(args.steps // 4, 10)
        dtt = time_frames(train_iter, k, max(args.warmup // 4, 3))
        extra["train_iters_per_s"] = round(world * k / dtt, 2)
        extra["train_ms_per_iter"] = round(dtt / k * 1e3, 4)
        extra["train_step"] = "forward + L1/SSIM loss (w=0.1) + backward + grad all-reduce + fused Adam, one view per GPU"
        if rank == 0:
            img, _ = rt.forward(*flat.params, cam)
            lossk = tr._loss_for(H, W)
            pb = [rt.profile_backward(lossk(img, target)) for _ in range(8)][3:]
            extra["backward_stage_ms"] = {key: round(statistics.median(p[key] for p in pb), 4) for key in pb[0]}

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
            extra["loss_ms"], extra["adam_ms"] = round(loss_ms, 4), round(adam_ms, 4)
            # Adam is a pure stream: 16 B read + 12 B written per parameter
            extra["adam_roofline"] = {"bound": "hbm", "achieved": round(28 * n_par / (adam_ms * 1e-3) / 1e9, 1),
                                      "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": round(28 * n_par / (adam_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                      "parameters": n_par}
        del tr
        if rank == 0 and world == 1 and args.config == "cfg2":
            # BASELINE.json configs[2] in miniature: fit a perturbed copy of the cfg3 scene (506,627 Gaussians, 1080p)
            # to a render of the original with the reference's training step (train.py defaults: lr 0.003 x
            # (10, 10, 1, 1, 1), exp decay, L1 + 0.1 SSIM, Adam(0.9, 0.99)); there is no dataset here, so the
            # figure of merit is the PSNR against that synthetic ground truth before / after a short run
            _, cam3, params3 = load("cfg3")
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
                                 "final_loss": round(float(tr3._loss_for(H, W).values[0]), 5)}
            del tr3, target3, start3, params3
        del rt, flat
        torch.cuda.empty_cache()
        if args.config != "cfg5":
            _, cam5, params5 = load("cfg5")  # 2.4M Gaussians, the north-star target (>= 160 FPS)
            r5, st5 = sized_renderer(params5, cam5, training=False)
            k5 = max(args.steps // 4, 10)
            dt5 = time_frames(lambda: r5.forward(*params5, cam5), k5, 5)
            extra["cfg5_2p4M_render_fps"] = round(world * k5 / dt5, 2)
            extra["cfg5_visible"], extra["cfg5_tile_pairs"] = st5.visible, st5.pairs
            del r5, params5
            torch.cuda.empty_cache()
        if rank == 0 and world == 1 and args.config == "cfg2":
            # BASELINE.json configs[3]: 2.4 M Gaussians, 1080p, SH, forward + backward (hipEvent-timed stages).
            # Degree 2 (27 coefficients) is what the reference implements; degree 3 (48) is the extension.
            n4, W4, H4, _ = CONFIGS["cfg4"]
            cam4 = make_camera(W4, H4)
            cfg4 = {}
            for deg in (2, 3):
                sc4 = make_scene(n4, W4, H4, seed=2023, use_sh=True, sh_degree=deg)
                p4 = [torch.from_numpy(a).to(dev) for a in (sc4.pos, sc4.quat, sc4.scale, sc4.opa, sc4.rgb)]
                r4, st4 = sized_renderer(p4, cam4, training=True)
                img4, _ = r4.forward(*p4, cam4)
                g4 = torch.sign(img4 - 0.5) / img4.numel()
                fw = [r4.profile_forward(*p4, cam4)["total"] for _ in range(6)][2:]
                bw = [r4.profile_backward(g4) for _ in range(6)][2:]
                f_ms = statistics.median(fw)
                b_ms = statistics.median(x["total"] for x in bw)
                cfg4[f"sh_degree_{deg}"] = {"coefficients": 3 * (deg + 1) ** 2, "tile_pairs": st4.pairs,
                                            "forward_ms": round(f_ms, 3), "backward_ms": round(b_ms, 3),
                                            "raster_bwd_ms": round(statistics.median(x["raster_bwd"] for x in bw), 3),
                                            "fwd_bwd_iters_per_s": round(1e3 / (f_ms + b_ms), 1)}
                del r4, p4, sc4, img4, g4
                torch.cuda.empty_cache()
            extra["cfg4_2p4M_sh_fwd_bwd"] = cfg4
    out["extra"] = extra

    # ---------------------------------------------------------------- CPU baseline (rank 0, N = 1 only)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle  # the checker, timed here as the "port" baseline -- never on the product path
        from gs_geometry import RayBasis, TileGrid

        grid = TileGrid(W, H, cam.focal_x, cam.focal_y)
        rays = RayBasis.from_camera(cam.rot, cam.tran, grid.padded_height, grid.padded_width, cam.focal_x, cam.focal_y)
        t0, frames = time.perf_counter(), 0
        while time.perf_counter() - t0 < 10.0 or frames < 2:
            oracle.render_forward(scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb, cam.rot, cam.tran,
                                  cam.near, W, H, cam.focal_x, cam.focal_y, 0.05, use_sh=use_sh, rays_o=rays.rays_o,
                                  lefttop=rays.lefttop, vdx=rays.dx, vdy=rays.dy)
            frames += 1
        cpu_dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(frames / cpu_dt, 4), "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": f"{frames} full forward frames of {args.config} "
                                         f"(oracle/gs_oracle.c, scalar C, {os.cpu_count()} host cores present)"}

    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
