#!/usr/bin/env python3
"""One workload, nothing else: the process rocprofv3 is wrapped around for per-config kernel traces and PMC passes.

    python tools/prof_target.py cfg5 [--frames 60] [--backward] [--sh-degree 3] [--train]

Every kernel dispatched after the warm-up belongs to `frames` identical forward (and, with --backward, backward)
frames of that BASELINE.json config, so per-kernel averages / counter sums in the rocprofv3 output are per-config
numbers (bench.py mixes its legs).  Prints one JSON line with N, V, M and the hipEvent stage times of the same run.
"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-gaussian-splatting_amd")]
import torch

from gs_frame import FrameRenderer
from gs_scene import CONFIGS, make_camera, make_scene, make_trained_like_scene

ap = argparse.ArgumentParser()
ap.add_argument("config")
ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--backward", action="store_true")
ap.add_argument("--sh-degree", type=int, default=2)
ap.add_argument("--no-stage-times", action="store_true", help="skip the hipEvent-bracketed frames (PMC passes)")
ap.add_argument("--bwd-rows", type=int, default=-1,
                help="rgb backward kernel: 1 = row layout (GS_FRAME_BWD_ROWS), 0 = pixel-parallel, -1 = the renderer's own "
                     "choice from the saturated-bucket statistic (it needs a backward + stats() to have run: below)")
ap.add_argument("--long-lists", choices=("auto", "on", "off"), default="auto",
                help="GS_FRAME_LONG_LISTS: by the renderer's own rule, always, never")
ap.add_argument("--train", action="store_true",
                help="whole training steps of gs_train.Trainer instead of bare frames (learning rate 0: the scene stays the "
                     "config's): forward, loss, backward with the optimizer step fused in (rgb, one rank) or followed by it")
a = ap.parse_args()
dev = torch.device("cuda:0")
if a.config.startswith("trained"):  # "trained_rgb" / "trained_sh": gs_scene.make_trained_like_scene (bench.py's trained_state leg)
    W, H, use_sh = 1920, 1080, a.config.endswith("sh")
    scene = make_trained_like_scene(width=W, height=H, use_sh=use_sh, sh_degree=a.sh_degree)
    n = scene.n
else:
    n, W, H, use_sh = CONFIGS[a.config]
    scene = make_scene(n, W, H, seed=2023, use_sh=use_sh, sh_degree=a.sh_degree)
cam = make_camera(W, H)
params = [torch.from_numpy(x).to(dev) for x in (scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb)]
r = FrameRenderer(dev, max_pairs=1 << 20, training=a.backward, auto_grow=True,
                  bwd_rows=None if a.bwd_rows < 0 else bool(a.bwd_rows),
                  long_lists={"auto": None, "on": True, "off": False}[a.long_lists])
img, _ = r.forward(*params, cam)
st = r.stats()
if a.train:
    from gs_train import TrainOptions, Trainer

    target = (img + 0.05 * torch.randn(H, W, 3, device=dev)).clamp_(0, 1).contiguous()
    del r, img
    tr = Trainer(params, [cam], [target], TrainOptions(lr=0.0), max_pairs=int(st.pairs * 1.25) + 4096)
    for i in range(30 + a.frames):
        tr.train_step(i, 0)
    torch.cuda.synchronize()
    print(json.dumps({"config": a.config, "n": n, "visible": st.visible, "tile_pairs": st.pairs, "train_steps": a.frames,
                      "warm_up_steps": 30, "adam_fused_into_backward": bool(tr._can_fuse_adam()),
                      "bwd_rows_flag": bool(tr.renderer._frame.flags & 64)}), flush=True)
    sys.exit(0)
r.max_pairs = int(st.pairs * 1.1) + 4096
r.auto_grow = False
img, _ = r.forward(*params, cam)
g = (torch.sign(img - 0.5) / img.numel()).contiguous()
if a.backward:  # one backward + its counters: the renderer settles which rgb backward kernel this scene takes
    r.backward(g)
    r.stats()
    img, _ = r.forward(*params, cam)
for _ in range(a.frames):
    img, _ = r.forward(*params, cam)
    if a.backward:
        r.backward(g)
torch.cuda.synchronize()
out = {"config": a.config, "n": n, "visible": st.visible, "tile_pairs": st.pairs, "frames": a.frames,
       "backward": a.backward, "sh_degree": a.sh_degree if use_sh else None,
       "bwd_rows_flag": bool(r._frame.flags & 64), "long_lists_flag": bool(r._frame.flags & 16),
       "longest_list": st.longest_list}
if not a.no_stage_times:
    pf = [r.profile_forward(*params, cam) for _ in range(12)][4:]
    out["forward_stage_ms"] = {k: round(statistics.median(p[k] for p in pf), 4) for k in pf[0]}
    if a.backward:
        pb = [r.profile_backward(g) for _ in range(8)][3:]
        out["backward_stage_ms"] = {k: round(statistics.median(p[k] for p in pb), 4) for k in pb[0]}
print(json.dumps(out), flush=True)
