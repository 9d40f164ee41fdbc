"""Soak test: a long render loop and a long training run with densification on the reference's schedule;
checks that results stay finite, memory does not creep and nothing stalls.  Prints one JSON object.

    python tools/soak.py [render frames] [training iterations] [SH degree: 0 (rgb logits, default) | 2 | 3]

`training_soak()` is also bench.py's `soak` leg (round 5): the densifying run's rate per block of 100 iterations."""
import json
import os
import statistics
import sys
import time

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (_ROOT, os.path.join(_ROOT, "3d-gaussian-splatting_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import torch  # noqa: E402

W, H = 1920, 1080


def soak_scene(dev, sh_degree):
    from gs_scene import make_camera, make_scene

    scene = make_scene(376_467, W, H, seed=2023, use_sh=sh_degree > 0, sh_degree=sh_degree if sh_degree else 2)
    params = [torch.from_numpy(a).to(dev) for a in (scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb)]
    cams = [make_camera(W, H, yaw_deg=float(y)) for y in np.linspace(-10, 10, 9)]
    return params, cams


def training_soak(dev, sh_degree, n_iters, params=None, cams=None, block=100, stats_every=500, keep=None):
    """Training with densification every 100 iterations (train.py's schedule) from a colour-perturbed copy of the 376,467-
    Gaussian scene against renders of the original from nine cameras, a random view per iteration.  The device is
    synchronised once per `block` iterations: `iters_per_s_blocks` = the rate of every block (the scene grows and its
    opacities fall while it trains: the rate is a trajectory, not a number), `iters_per_s` = all iterations / wall time."""
    from gs_frame import FrameRenderer
    from gs_train import TrainOptions, Trainer

    if params is None:
        params, cams = soak_scene(dev, sh_degree)
    r = FrameRenderer(dev, max_pairs=1 << 21)
    targets = [r.forward(*params, c)[0].clone() for c in cams]
    del r
    g = torch.Generator(device=dev).manual_seed(3)
    start = [t.clone() for t in params]
    start[4] += 0.5 * torch.randn(start[4].shape, device=dev, generator=g)
    opt = TrainOptions(n_iters=n_iters, use_clone=1, delete_thresh=1.5, grad_thresh=2e-4)
    tr = Trainer(start, cams, targets, opt, max_pairs=1 << 21, densify=True, generator=g)
    rng = np.random.default_rng(0)
    sizes, pairs, rates, flags = [tr.n_gaussians], [], [], []
    torch.cuda.synchronize()
    t0 = t_win = time.perf_counter()
    v = None
    for i in range(n_iters):
        v = tr.train_step(i, int(rng.integers(len(cams))))
        if (i + 1) % block == 0:
            torch.cuda.synchronize()
            now = time.perf_counter()
            rates.append(round(block / (now - t_win), 1))
            t_win = now
            # what the renderer has latched at the end of the block (host-side state, no synchronisation): frame flags
            # (16 segments, 64 row-layout rgb backward, 128 big-list sort), capacity, frames found overflowed
            f = tr.renderer._frame
            flags.append([int(f.flags) if f is not None else 0, int(tr.renderer.max_pairs), int(tr.renderer.overflowed_frames)])
        if i % stats_every == 0:
            sizes.append(tr.n_gaussians)
            pairs.append(tr.renderer.stats().pairs)  # synchronises: once per `stats_every` iterations
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    vals = v.cpu().numpy()
    if keep is not None:  # (diagnostics: the trainer in its end state)
        keep["trainer"], keep["cams"] = tr, cams
    return {"iters": n_iters, "iters_per_s": round(n_iters / dt, 1), "block": block, "repeats": len(rates),
            "iters_per_s_median_block": round(statistics.median(rates), 1) if rates else None,
            "iters_per_s_min_block": min(rates) if rates else None, "iters_per_s_max_block": max(rates) if rates else None,
            "iters_per_s_blocks": rates, "flags_capacity_overflows_per_block": flags, f"n_gaussians_every_{stats_every}": sizes, f"tile_pairs_every_{stats_every}": pairs,
            "final_loss": round(float(vals[0]), 5), "finite": bool(np.isfinite(vals).all()),
            "peak_memory_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}


def main():
    from gs_frame import FrameRenderer

    dev = torch.device('cuda:0')
    out = {}
    # ---- render: 3 frames in flight, changing cameras
    sh_degree = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    out["sh_degree"] = sh_degree
    params, cams = soak_scene(dev, sh_degree)
    rs = [FrameRenderer(dev, max_pairs=1 << 21, auto_grow="async") for _ in range(3)]
    streams = [torch.cuda.Stream(device=dev) for _ in rs]
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60_000
    torch.cuda.synchronize()
    m0 = torch.cuda.memory_allocated()
    t0 = time.perf_counter()
    last = None
    for k in range(n_frames):
        with torch.cuda.stream(streams[k % 3]):
            last = rs[k % 3].forward(*params, cams[k % len(cams)])[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if n_frames:
        out["render"] = {"frames": n_frames, "fps": round(n_frames / dt, 1), "finite": bool(torch.isfinite(last).all()),
                         "memory_growth_MiB": round((torch.cuda.memory_allocated() - m0) / 2**20, 1)}
    del rs, last
    n_iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
    out["train"] = training_soak(dev, sh_degree, n_iters, params, cams)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
