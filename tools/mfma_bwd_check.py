#!/usr/bin/env python3
"""Gradients and backward stage times of a VARIANT build of libgs_amd.so against the in-tree build, scene by scene.

    python tools/mfma_bwd_check.py compare <variant[,variant...] under build/variants> [scene ...]   # on the GPU box
    python tools/mfma_bwd_check.py dump <out.npz> <scene>                                        # (one library, one scene)

Scenes: s2 / s3 = 30,000 Gaussians at 512 x 384 with degree-2 / degree-3 SH, d2 / d3 = a dense small scene (every tile
several buckets deep), c2 / c2_deg3 = 376,467 Gaussians at 1080p (two or three buckets per tile), cfg4 / cfg4_deg3 = the
2.4 M-Gaussian scene of BASELINE configs[3].  `compare` runs `dump` once per
library in fresh processes (GS_AMD_LIB selects the variant) and prints, per gradient tensor, the relative L2 distance,
the largest difference over the tensor's maximum and the number of non-finite entries, with both builds' stage times.
Built for the SH backward on the matrix pipe (raster_backward_mfma_sh_kernel); works for any backward variant."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-gaussian-splatting_amd")]

SCENES = {  # name: (Gaussians, W, H, SH degree, scale multiplier)
    "s2": (30_000, 512, 384, 2, 1.0),
    "s3": (30_000, 512, 384, 3, 1.0),
    "d2": (60_000, 256, 192, 2, 1.0),
    "d3": (60_000, 256, 192, 3, 1.0),
    "c2": (376_467, 1920, 1080, 2, 1.0),       # BASELINE configs[1]'s size with SH: ~135 pairs per tile, nothing saturates
    "c2_deg3": (376_467, 1920, 1080, 3, 1.0),
    "cfg4": (2_400_000, 1920, 1080, 2, 1.0),
    "cfg4_deg3": (2_400_000, 1920, 1080, 3, 1.0),
}
NAMES = ("pos", "quat", "scale", "opa", "rgb")


def dump(out, scene):
    import numpy as np
    import torch

    from gs_frame import FrameRenderer
    from gs_scene import make_camera, make_scene

    n, W, H, deg, _ = SCENES[scene]
    dev = torch.device("cuda:0")
    sc = make_scene(n, W, H, seed=2023, use_sh=True, sh_degree=deg)
    cam = make_camera(W, H)
    params = [torch.from_numpy(a).to(dev) for a in (sc.pos, sc.quat, sc.scale, sc.opa, sc.rgb)]
    r = FrameRenderer(dev, max_pairs=1 << 20, training=True, auto_grow=True)
    r.forward(*params, cam)
    st = r.stats()
    r.max_pairs = int(st.pairs * 1.1) + 4096
    r.auto_grow = False
    img, _ = r.forward(*params, cam)
    torch.manual_seed(7)
    g = (torch.sign(img - 0.5) * (0.5 + torch.rand_like(img))) / img.numel()
    grads = r.backward(g)
    torch.cuda.synchronize()
    for _ in range(3):
        r.forward(*params, cam)
    pb = [r.profile_backward(g) for _ in range(8)][3:]
    bw = {k: float(np.median([p[k] for p in pb])) for k in pb[0]}
    again = r.backward(g)  # (after profile_backward: same forward state)
    torch.cuda.synchronize()
    rep = all(torch.equal(a, b) for a, b in zip(grads, again))
    np.savez(out, **{k: t.cpu().numpy() for k, t in zip(NAMES, grads)}, raster_bwd=bw["raster_bwd"],
             project_bwd=bw["project_bwd"], total=bw["total"], pairs=st.pairs, steps=r.composited_steps(),
             repeatable=rep)


def compare(variants, scenes):
    import numpy as np

    variants = variants.split(",")
    libs = {v: os.path.join(ROOT, "build", "variants", v, "libgs_amd.so") for v in variants}
    for lib in libs.values():
        assert os.path.exists(lib), lib
    for scene in scenes:
        res = {}
        with tempfile.TemporaryDirectory() as td:
            for name in ["base"] + variants:
                env = dict(os.environ)
                if name != "base":
                    env["GS_AMD_LIB"] = libs[name]
                out = os.path.join(td, name + ".npz")
                try:
                    p = subprocess.run([sys.executable, os.path.abspath(__file__), "dump", out, scene], env=env,
                                       capture_output=True, text=True, timeout=600)
                except subprocess.TimeoutExpired:
                    print(f"[{scene}] {name} TIMED OUT", flush=True)
                    continue
                if p.returncode:
                    print(f"[{scene}] {name} FAILED rc={p.returncode}: {p.stderr[-1500:]}", flush=True)
                    continue
                res[name] = dict(np.load(out))
        if "base" not in res:
            continue
        a = res["base"]
        for variant in variants:
            if variant not in res:
                continue
            b = res[variant]
            print(f"[{scene}] pairs {int(a['pairs'])} steps {int(a['steps'])}  raster_bwd base {float(a['raster_bwd']):.4f} ms "
                  f"-> {variant} {float(b['raster_bwd']):.4f} ms   project_bwd {float(a['project_bwd']):.4f} -> "
                  f"{float(b['project_bwd']):.4f}   repeatable {bool(a['repeatable'])} / {bool(b['repeatable'])}", flush=True)
            for k in NAMES:
                x, y = a[k].astype(np.float64), b[k].astype(np.float64)
                bad = int((~np.isfinite(y)).sum())
                y = np.where(np.isfinite(y), y, 0.0)
                d = np.abs(x - y)
                i = int(np.argmax(d))
                print(f"[{scene}] {variant}  {k:5s} relL2 {np.linalg.norm(x - y) / max(np.linalg.norm(x), 1e-300):.3e}  "
                      f"max|d|/max|x| {d.max() / max(np.abs(x).max(), 1e-300):.3e}  nonfinite {bad}  "
                      f"|x|max {np.abs(x).max():.3e}  worst idx {i}: base {x.flat[i]:.6e} variant {y.flat[i]:.6e}  "
                      f"nonzero base/variant {int((x != 0).sum())}/{int((y != 0).sum())}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "dump":
        dump(sys.argv[2], sys.argv[3])
    elif len(sys.argv) >= 3 and sys.argv[1] == "compare":
        compare(sys.argv[2], sys.argv[3:] or ["s2", "s3", "d2", "d3"])
    else:
        raise SystemExit(__doc__)
