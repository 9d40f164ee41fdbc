#!/usr/bin/env python3
"""HBM rate of the fused Adam launch as a function of the parameter count:  python tools/adam_bw.py [n_params ...]

28 bytes move per parameter (16 read + 12 written).  Below ~9 M parameters the four arrays (16 B per parameter) fit the
256 MiB Infinity Cache and the rate is the cache's, not HBM's; from ~20 M on it is a plain HBM stream."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-gaussian-splatting_amd")]
import torch

from gs_dp import FlatGaussianParams
from gs_train import FusedAdam

dev = torch.device("cuda:0")
for n_par in ([int(a) for a in sys.argv[1:]] or [5_270_538, 14_000_000, 33_600_000, 134_400_000]):
    n = n_par // 14
    params = [torch.randn(n, k, device=dev) if k else torch.randn(n, device=dev) for k in (3, 4, 3, 0, 3)]
    flat = FlatGaussianParams(params)
    flat.flat_grad.normal_()
    opt = FusedAdam(flat, [1e-3] * 5)
    for _ in range(5):
        opt.step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        opt.step()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    npar = flat.flat_param.numel()
    print(json.dumps({"parameters": npar, "working_set_MB": round(16 * npar / 1e6, 1), "ms": round(ms, 4),
                      "GBs": round(28 * npar / ms / 1e6, 1), "frac_of_hbm_peak": round(28 * npar / ms / 1e6 / 8000, 3)}),
          flush=True)
    del opt, flat, params
    torch.cuda.empty_cache()
