#!/usr/bin/env python3
"""What does the scene look like at the END of the densifying soak, and what do the backward's kernels cost there?

    python tools/soak_end_state.py [SH degree, default 2] [iterations, default 2000]

Runs tools/soak.py's training run, then renders one view of the end state and prints the list-length distribution of the
tiles (processed Gaussians per tile: mean, percentiles, maximum, tiles beyond 512 / 2048), pairs, composited steps, executed
row steps of the SH backward, the rectangle sizes of the largest Gaussians, and hipEvent stage times of forward / backward."""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-gaussian-splatting_amd"), os.path.join(ROOT, "tools")]
import ctypes as C

import numpy as np
import torch

from gaussian import _lib
from soak import training_soak

deg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
dev = torch.device("cuda:0")
keep = {}
res = training_soak(dev, deg, iters, keep=keep)
tr, cams = keep["trainer"], keep["cams"]
r, flat = tr.renderer, tr.flat
flat.finish_gather()
r.forward_abandon()
cam = cams[0]
img, _ = r.forward(*flat.params, cam)
g = (torch.sign(img - 0.5) / img.numel()).contiguous()
r.backward(g)
st = r.stats()
f = r._frame
ptr = C.c_void_p()
_lib.check(_lib.gs_frame_debug_tile_nproc(C.byref(f), C.byref(ptr)), "tile_nproc")
off = ptr.value - r._ws.data_ptr()
T = r._grid.n_tiles
nproc = r._ws[off:off + 4 * T].view(torch.int32).cpu().numpy().astype(np.int64)
v = r.debug_views()
ranges = v["tile_ranges"].cpu().numpy()
lens = (ranges[:, 1] - ranges[:, 0]).astype(np.int64)
rects = r._rects().cpu().numpy()
touched = np.sort(rects[:, 3].astype(np.int64))[::-1]
out = {"iters_per_s": res["iters_per_s"], "last_blocks": res["iters_per_s_blocks"][-3:], "n_gaussians": tr.n_gaussians,
       "visible": st.visible, "pairs": st.pairs, "longest_list": st.longest_list, "flags": int(f.flags),
       "buckets": st.buckets, "saturated_buckets": st.saturated_buckets,
       "list_len": {"mean": float(lens.mean()), "p50": float(np.percentile(lens, 50)), "p90": float(np.percentile(lens, 90)),
                    "p99": float(np.percentile(lens, 99)), "max": int(lens.max()), "tiles_gt_512": int((lens > 512).sum()),
                    "tiles_gt_2048": int((lens > 2048).sum())},
       "processed": {"sum": int(nproc.sum()), "mean": float(nproc.mean()), "p99": float(np.percentile(nproc, 99)),
                     "max": int(nproc.max()), "tiles_gt_512": int((nproc > 512).sum()), "tiles_gt_2048": int((nproc > 2048).sum())},
       "rows_per_gaussian": {"top5": touched[:5].tolist(), "gt_64": int((touched > 64).sum()), "gt_1000": int((touched > 1000).sum()),
                             "rows_of_gt_64": int(touched[touched > 64].sum())}}
if deg:
    out["executed_row_steps"] = r.executed_row_steps()
pf = [r.profile_forward(*flat.params, cam) for _ in range(6)][2:]
out["forward_stage_ms"] = {k: round(statistics.median(p[k] for p in pf), 4) for k in pf[0]}
pb = [r.profile_backward(g) for _ in range(6)][2:]
out["backward_stage_ms"] = {k: round(statistics.median(p[k] for p in pb), 4) for k in pb[0]}
# the same end state with the long-list flags forced: none / the sort alone / sort + segmented compositing (round 6: what the
# cost model of FrameRenderer._note_lists has to reproduce), forward and backward stage times, and the list statistics it sees
h = r._stats_host.tolist()
out["pairs_beyond_512_per_tile"] = int(np.clip(lens - 512, 0, None).sum())  # (the device counter [12] is retired: from the ranges)
out["steps_beyond_512_per_walk"] = int(h[14])
keep = (r.long_lists, r._long_sort_seen, r._long_lists_seen)
r.auto_grow = False  # (no asynchronous counters: nothing re-latches a flag underneath the forced modes)
for mode, (ll, srt) in (("none", (False, False)), ("sort_only", (None, True)), ("sort_and_segments", (True, True))):
    r.long_lists, r._long_sort_seen, r._long_lists_seen = ll, srt, False
    for _ in range(3):
        img, _ = r.forward(*flat.params, cam)
        r.backward(g)
    pf = [r.profile_forward(*flat.params, cam) for _ in range(6)][2:]
    pb = [r.profile_backward(g) for _ in range(6)][2:]
    out["mode_" + mode] = {"flags": int(r._frame.flags),
                           "forward_stage_ms": {k: round(statistics.median(p[k] for p in pf), 4) for k in pf[0]},
                           "backward_stage_ms": {k: round(statistics.median(p[k] for p in pb), 4) for k in pb[0]}}
r.long_lists, r._long_sort_seen, r._long_lists_seen = keep
lens_sorted = np.sort(lens)[::-1]
out["list_len"]["top10"] = lens_sorted[:10].tolist()
out["list_len"]["sum_of_top_64"] = int(lens_sorted[:64].sum())
print(json.dumps(out))
