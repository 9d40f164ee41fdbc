#!/usr/bin/env python3
"""How much does the gradient exchange cost the training step on ONE rank (RCCL, collective forced)?

    python tools/exchange_probe.py [cfg5|cfg4] [--slices 1,2,4,8] [--modes all_reduce,reduce_scatter] [--k 25] [--trace]

For every (mode, number of slices): ms per training step (tools/train_timing.py protocol), the same step without its
exchange, their difference (exposed_ms), and the HOST time of issuing one step with the GPU free-running -- a step whose
host time approaches its GPU time is launch-bound, whatever the kernels do.  `--trace`: one short run of one setting,
for `rocprofv3 --kernel-trace --stats` (the RCCL kernels show up with their own durations)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-gaussian-splatting_amd"), os.path.join(ROOT, "tools")]
import torch
import torch.distributed as dist

from gs_frame import FrameRenderer
from gs_scene import CONFIGS, make_camera, make_scene
from gs_train import TrainOptions, Trainer
from train_timing import restore, snapshot, time_training

ap = argparse.ArgumentParser()
ap.add_argument("config", nargs="?", default="cfg5")
ap.add_argument("--slices", default="1,2,4,8")
ap.add_argument("--modes", default="all_reduce,reduce_scatter")
ap.add_argument("--k", type=int, default=25)
ap.add_argument("--repeats", type=int, default=7)
ap.add_argument("--trace", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n, W, H, use_sh = CONFIGS[a.config]
scene = make_scene(n, W, H, seed=2023, use_sh=use_sh)
cam = make_camera(W, H)
params = [torch.from_numpy(x).to(dev) for x in (scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb)]
r = FrameRenderer(dev, max_pairs=1 << 20, training=False, auto_grow=True)
img = r.forward(*params, cam)[0]
pairs = r.stats().pairs
target = (img + 0.05 * torch.randn(H, W, 3, device=dev)).clamp_(0, 1).contiguous()
del r


def run(mode, n_slices, collective):
    tr = Trainer([t.clone() for t in params], [cam], [target], TrainOptions(), max_pairs=int(pairs * 1.25) + 4096,
                 exchange=mode, n_slices=n_slices)
    tr.flat.force_collective = True
    tr.flat.enable_collective = collective
    dt, blocks, _ = time_training(tr, a.k, warm=30, repeats=a.repeats)
    # host time of issuing a step: restore, drain, then issue k steps without waiting for the GPU
    snap = snapshot(tr)
    restore(tr, snap)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(30, 30 + a.k):
        tr.train_step(i, 0, next_camera_id=0)
    host = (time.perf_counter() - t0) / a.k
    torch.cuda.synchronize()
    del tr
    torch.cuda.empty_cache()
    return dt / a.k * 1e3, min(blocks) / a.k * 1e3, max(blocks) / a.k * 1e3, host * 1e3


if a.trace:
    mode, ns = a.modes.split(",")[0], int(a.slices.split(",")[0])
    print(json.dumps({"trace": True, "mode": mode, "n_slices": ns, "ms": run(mode, ns, True)[0]}))
else:
    plain = run("all_reduce", 1, False)
    print(json.dumps({"config": a.config, "plain_step_ms": round(plain[0], 4), "plain_host_ms": round(plain[3], 4)}), flush=True)
    for mode in a.modes.split(","):
        for ns in (int(x) for x in a.slices.split(",")):
            ms, lo, hi, host = run(mode, ns, True)
            print(json.dumps({"config": a.config, "mode": mode, "n_slices": ns, "ms_per_iter": round(ms, 4),
                              "min": round(lo, 4), "max": round(hi, 4), "exposed_ms": round(ms - plain[0], 4),
                              "host_ms_per_iter": round(host, 4)}), flush=True)
dist.destroy_process_group()
