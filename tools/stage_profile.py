#!/usr/bin/env python3
"""hipEvent stage times of one forward + backward frame per config:  python tools/stage_profile.py [cfg2 cfg4 cfg5 cfg4_deg3 ...]

Suffixes: `_deg3` = degree-3 SH (48 coefficients), `_keys` = also write the sorted keys (GS_FRAME_EMIT_SORTED_KEYS),
`_fwd` = inference forward only (no checkpoints, no backward), `_ss` = slice-sorted binning variant, `_tb` = table
binning variant (default: the strip variant), `_rows` / `_pix` = the row-layout / pixel-parallel rgb backward kernel
(default: the renderer's choice from the saturated-bucket statistic of a first backward)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-gaussian-splatting_amd")]
import numpy as np
import torch

from gs_frame import FrameRenderer
from gs_scene import CONFIGS, make_camera, make_scene

dev = torch.device("cuda:0")
for cfg in (sys.argv[1:] or ("cfg2", "cfg4", "cfg5")):
    base = cfg
    flags = set()
    for suf in ("_deg3", "_keys", "_fwd", "_ss", "_tb", "_rows", "_pix"):
        if suf in base:
            base = base.replace(suf, "")
            flags.add(suf)
    deg = 3 if "_deg3" in flags else 2  # e.g. cfg4_deg3: BASELINE config 4 with true degree-3 SH (48 coefficients)
    n, W, H, use_sh = CONFIGS[base]
    scene = make_scene(n, W, H, seed=2023, use_sh=use_sh, sh_degree=deg)
    cam = make_camera(W, H)
    params = [torch.from_numpy(a).to(dev) for a in (scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb)]
    training = "_fwd" not in flags
    r = FrameRenderer(dev, max_pairs=1 << 20, training=training, auto_grow=True, emit_sorted_keys="_keys" in flags,
                      slice_sort="_ss" in flags, table_bin="_tb" in flags,
                      bwd_rows=True if "_rows" in flags else False if "_pix" in flags else None)
    r.forward(*params, cam)
    st = r.stats()
    r.max_pairs = int(st.pairs * 1.1) + 4096
    r.auto_grow = False
    img, _ = r.forward(*params, cam)
    for _ in range(30):  # clocks
        r.forward(*params, cam)
    prof = [r.profile_forward(*params, cam) for _ in range(16)][4:]
    fw = {k: round(float(np.median([p[k] for p in prof])), 4) for k in prof[0]}
    bw = {}
    if training:
        g = torch.sign(img - 0.5) / img.numel()
        r.backward(g)
        r.stats()  # (the renderer settles which rgb backward kernel this scene takes)
        r.forward(*params, cam)
        pb = [r.profile_backward(g) for _ in range(8)][3:]
        bw = {k: round(float(np.median([p[k] for p in pb])), 4) for k in pb[0]}
    print(cfg, "V", st.visible, "M", st.pairs, "fwd", fw, "bwd", bw,
          "rows" if (training and r._frame.flags & 64) else "", flush=True)
    del r, params
    torch.cuda.empty_cache()
