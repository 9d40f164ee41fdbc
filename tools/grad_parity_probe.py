"""Distribution of the gradient error of gs_frame_backward against the oracle, element by element.

    python tools/grad_parity_probe.py [cfg2 cfg3 cfg4 small ...]   (GPU box; writes gpurun_out/grad_parity_<cfg>.json)

For every parameter tensor: quantiles of |got - ref| relative to |ref|, to the plain sum of term magnitudes
(scale_w = 0) and to the conditioning scale the tests use (scale_w = 0.05); relative L2 error; worst elements.
This is how GRAD_RTOL / GRAD_KAPPA of tests/gs_testutil.py were chosen.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-gaussian-splatting_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch

from gs_frame import FrameRenderer
from gs_scene import CONFIGS, make_camera, make_scene
from gs_testutil import GRAD_KAPPA, GRAD_RTOL, OracleFrame, grad_close, to_torch

dev = torch.device("cuda:0")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
Q = (50, 90, 99, 99.9, 99.99, 100)


def probe(name, scene, cam, grad_kind="sign"):
    t0 = time.time()
    of = OracleFrame(scene, cam)
    t_fwd = time.time() - t0
    if grad_kind == "sign":  # dL/dimage of an L1 loss against a constant grey target: spatially coherent
        gimg = (np.sign(of.image - 0.5) / of.image.size).astype(np.float32)
    else:
        gimg = np.random.default_rng(4).normal(size=of.image.shape).astype(np.float32)
    gimg, n_amb = of.robust_grad_image(gimg)
    t0 = time.time()
    ref, s05 = of.backward(gimg, with_scale=True, scale_w=0.05)
    t_bwd = time.time() - t0
    _, s0 = of.backward(gimg, with_scale=True, scale_w=0.0)
    params = to_torch(scene, dev, requires_grad=True)
    r = FrameRenderer(dev, max_pairs=len(of.ids) + 64, training=True, auto_grow=False)
    img = r.render(*params, cam)
    img_err = float(np.abs(img.detach().cpu().numpy() - of.image).max())
    img.backward(torch.from_numpy(gimg).to(dev))
    out = {"config": name, "grad": grad_kind, "n": scene.n, "pairs": int(len(of.ids)), "image_err": img_err, "ambiguous_pixels_zeroed": n_amb,
           "oracle_fwd_s": round(t_fwd, 2), "oracle_bwd_s": round(t_bwd, 2), "tensors": {}}
    for t, key in zip(params, ("pos", "quat", "scale", "opa", "rgb")):
        got = t.grad.cpu().numpy().astype(np.float64)
        rf = ref[key].astype(np.float64)
        err = np.abs(got - rf)
        nz = np.abs(rf) > 0
        with np.errstate(divide="ignore", invalid="ignore"):
            rel = err[nz] / np.abs(rf[nz])
            r0 = err[s0[key] > 0] / s0[key][s0[key] > 0]
            r5 = err[s05[key] > 0] / s05[key][s05[key] > 0]
        ok, worst, where, pure = grad_close(got, rf, s05[key])
        tol = GRAD_RTOL * np.abs(rf) + GRAD_KAPPA * s05[key] + 1e-300
        top = np.argsort((err / tol).ravel())[-5:][::-1]
        worst5 = [{"index": [int(x) for x in np.unravel_index(i, rf.shape)], "got": float(got.flat[i]),
                   "ref": float(rf.flat[i]), "scale": float(s05[key].flat[i])} for i in top]
        out["tensors"][key] = {
            "max_abs_ref": float(np.abs(rf).max()), "rel_l2": float(np.linalg.norm(got - rf) / np.linalg.norm(rf)),
            "maxnorm": float(err.max() / np.abs(rf).max()),
            "err_over_ref_q": dict(zip(map(str, Q), (float(x) for x in np.percentile(rel, Q)))),
            "err_over_scale_w0_q": dict(zip(map(str, Q), (float(x) for x in np.percentile(r0, Q)))),
            "err_over_scale_w05_q": dict(zip(map(str, Q), (float(x) for x in np.percentile(r5, Q)))),
            "scale_over_ref_median": float(np.median(s05[key][nz] / np.abs(rf[nz]))),
            "grad_close": {"ok": ok, "worst_ratio": worst, "where": [int(i) for i in where], "rtol": GRAD_RTOL,
                           "kappa": GRAD_KAPPA, "frac_within_rtol_alone": pure},
            "nonzero_where_ref_zero": int(((rf == 0) & (got != 0)).sum()), "worst5_before_floor": worst5,
        }
    path = os.path.join(ROOT, "gpurun_out", f"grad_parity_{name}_{grad_kind}.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out), flush=True)


for cfg in (sys.argv[1:] or ["small", "cfg2"]):
    if cfg == "small":
        scene, cam = make_scene(20_000, 160, 112, seed=11), make_camera(160, 112, yaw_deg=2.0)
    elif cfg == "small_sh":
        scene, cam = make_scene(9_000, 160, 112, seed=11, use_sh=True), make_camera(160, 112, yaw_deg=2.0)
    else:
        n, W, H, use_sh = CONFIGS[cfg]
        scene, cam = make_scene(n, W, H, seed=2023, use_sh=use_sh), make_camera(W, H)
    for kind in ("sign", "normal"):
        probe(cfg, scene, cam, kind)
