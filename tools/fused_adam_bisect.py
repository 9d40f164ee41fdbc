#!/usr/bin/env python3
"""Where do a fused-optimizer run and a two-kernel run of the SAME training problem part?

    python tools/fused_adam_bisect.py [iterations, default 7001] [check every, default 100]

tools/train_demo.py's problem (cfg3 scene, 17 views, the reference's schedule), two gs_train.Trainer objects stepped in lockstep
over the same view sequence: fuse_adam=True (gs_frame_backward_adam) and fuse_adam=False (gs_frame_backward + gs_adam_step).
Every `check` iterations the flat parameter buffers are compared bit for bit (one synchronisation); at the first difference the
script reports the window, then re-runs nothing -- it prints which arrays differ, how many elements, the largest difference, and
what the renderer knew about those Gaussians in the last frame (visible, tiles touched)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-gaussian-splatting_amd")]
import torch  # noqa: E402

from gs_frame import FrameRenderer  # noqa: E402
from gs_scene import CONFIGS, make_camera, make_scene  # noqa: E402
from gs_train import TrainOptions, Trainer  # noqa: E402

_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
n_iters = int(_pos[0]) if _pos else 7001
check = int(_pos[1]) if len(_pos) > 1 else 100
SEQUENTIAL = "--sequential" in sys.argv  # one trainer after the other (each at its own pace, as two stand-alone runs) instead of lockstep
MAX_PAIRS = 1 << 21 if "--demo-capacity" in sys.argv else 1 << 22  # (tools/train_demo.py starts from 1 << 21)
LOG = "--no-log" not in sys.argv         # sequential mode: per-iteration parameter checksum + loss of both runs
SAME_MODE = "--same-mode" in sys.argv    # both trainers with the two-kernel step: is a run reproducible inside one process?
SYNC_EVERY_STEP = "--sync" in sys.argv   # the host never runs ahead of the device (asynchronous counters arrive at once)
dev = torch.device("cuda:0")
n, W, H, _ = CONFIGS["cfg3"]
scene = make_scene(n, W, H, seed=2023)
gt = [torch.from_numpy(a).to(dev) for a in (scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb)]
cams = [make_camera(W, H, yaw_deg=float(y)) for y in np.linspace(-16, 16, 17)]
r = FrameRenderer(dev, max_pairs=1 << 21)
targets = [r.forward(*gt, c)[0].clone() for c in cams]
del r
g = torch.Generator(device=dev).manual_seed(11)
start = [t.clone() for t in gt]
start[4] += 0.5 * torch.randn(start[4].shape, device=dev, generator=g)
start[3] += 0.3 * torch.randn(start[3].shape, device=dev, generator=g)
start[0] += 0.002 * torch.randn(start[0].shape, device=dev, generator=g)
start[2] *= 1.0 + 0.1 * torch.randn(start[2].shape, device=dev, generator=g)
trs = [Trainer([t.clone() for t in start], cams, targets, TrainOptions(n_iters=n_iters), max_pairs=MAX_PAIRS, fuse_adam=f)
       for f in ((False, False) if SAME_MODE else (True, False))]
assert SAME_MODE or (trs[0]._can_fuse_adam() and not trs[1]._can_fuse_adam())
train_split = np.array(sorted(set(range(len(cams))) - set(np.arange(0, len(cams), 8))))
rng = np.random.default_rng(2023)
out = {"iterations": n_iters, "check_every": check, "first_difference_in": None, "sequential": SEQUENTIAL, "same_mode": SAME_MODE, "sync_every_step": SYNC_EVERY_STEP, "max_pairs_at_start": MAX_PAIRS}


def state(t):
    r_ = t.renderer
    return {"flags": int(r_._frame.flags), "max_pairs": int(r_.max_pairs), "overflowed_frames": int(r_.overflowed_frames),
            "long_lists_seen": bool(r_._long_lists_seen), "bwd_rows_seen": bool(r_._bwd_rows_seen)}


if SEQUENTIAL:
    views = [int(rng.choice(train_split)) for _ in range(n_iters)]
    logs = []
    for t in trs:
        sums = torch.zeros(n_iters, dtype=torch.int64, device=dev)   # bit-exact checksum of the parameters behind every step
        losses = torch.zeros(n_iters, dtype=torch.float32, device=dev)
        caps = []
        for i, v in enumerate(views):
            val = t.train_step(i, v)
            if LOG:  # (one more reduction per step on the stream: the host runs less far ahead with it)
                sums[i] = t.flat.flat_param.view(torch.int32).to(torch.int64).sum()
                losses[i] = val[0]
            caps.append(int(t.renderer.max_pairs))
            if SYNC_EVERY_STEP:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        logs.append((sums.cpu(), losses.cpu(), caps))
    diff = torch.nonzero(logs[0][0] != logs[1][0]).flatten() if LOG else []
    if len(diff):
        k = int(diff[0])
        out["first_differing_iteration"] = k
        lo, hi = max(k - 3, 0), min(k + 3, n_iters)
        out["around_it"] = {"iterations": list(range(lo, hi)), "views": views[lo:hi],
                            "loss_fused": [round(float(x), 6) for x in logs[0][1][lo:hi]],
                            "loss_two_kernels": [round(float(x), 6) for x in logs[1][1][lo:hi]],
                            "max_pairs_fused": logs[0][2][lo:hi], "max_pairs_two_kernels": logs[1][2][lo:hi]}
        out["largest_loss_fused"] = [float(logs[0][1].max()), int(logs[0][1].argmax())]
        out["largest_loss_two_kernels"] = [float(logs[1][1].max()), int(logs[1][1].argmax())]
        out["capacity_changes_fused"] = [(i, c) for i, c in enumerate(logs[0][2]) if i == 0 or c != logs[0][2][i - 1]][:12]
        out["capacity_changes_two_kernels"] = [(i, c) for i, c in enumerate(logs[1][2]) if i == 0 or c != logs[1][2][i - 1]][:12]
    out["equal_at_the_end"] = bool(torch.equal(trs[0].flat.flat_param, trs[1].flat.flat_param))
    out["max_abs_difference"] = float((trs[0].flat.flat_param - trs[1].flat.flat_param).abs().max())
    out["renderer_state"] = {"fused": state(trs[0]), "two_kernels": state(trs[1])}
    print(json.dumps(out))
    sys.exit(0)
for i in range(n_iters):
    v = int(rng.choice(train_split))
    for t in trs:
        t.train_step(i, v)
    if (i + 1) % check == 0 or i + 1 == n_iters:
        a, b = trs[0].flat, trs[1].flat
        if not torch.equal(a.flat_param, b.flat_param):
            out["first_difference_in"] = [i + 1 - check, i + 1]
            det = {}
            rects = trs[1].renderer._ws  # (not parsed here: the culling mask below is the renderer's own view of the frame)
            vis = trs[1].renderer.culling_mask()
            for name, pa, pb in zip(("pos", "quat", "scale", "opa", "rgb"), a.params, b.params):
                d = (pa != pb)
                if d.any():
                    rows = d.reshape(d.shape[0], -1).any(dim=1)
                    idx = torch.nonzero(rows).flatten()
                    det[name] = {"elements": int(d.sum()), "gaussians": int(rows.sum()),
                                 "max_abs_difference": float((pa - pb).abs().max()),
                                 "of_them_visible_in_last_frame": int(vis[idx].sum()),
                                 "first_indices": [int(x) for x in idx[:8]]}
            out["differing"] = det
            out["moments_equal"] = [bool(torch.equal(trs[0].optimizer.exp_avg, trs[1].optimizer.exp_avg)),
                                    bool(torch.equal(trs[0].optimizer.exp_avg_sq, trs[1].optimizer.exp_avg_sq))]
            out["overflowed_frames"] = [t.renderer.overflowed_frames for t in trs]
            out["max_pairs"] = [t.renderer.max_pairs for t in trs]
            out["flags"] = [int(t.renderer._frame.flags) for t in trs]
            break
out["renderer_state"] = {"fused": state(trs[0]), "two_kernels": state(trs[1])}
print(json.dumps(out))
