"""Writes a synthetic capture in the layout the reference's train.py reads (train.py:368-369):

    <out>/sparse/0/{cameras,images,points3D}.bin      COLMAP binaries (one PINHOLE camera, yawed poses)
    <out>/images_<k>/frame_XXX.png                    ground-truth renders at 1/k of the camera resolution

There is no real capture offline (no Garden dataset): the ground truth is a gs_scene.make_scene() scene rendered by
this package's own renderer from `n_views` cameras on a yaw arc, and the sparse point cloud is a random subset of the
scene's Gaussian centres with their colours -- i.e. what COLMAP would hand to Splatter.__init__.

    python tools/make_synthetic_colmap.py OUT [--n 20000] [--width 480] [--height 270] [--views 17] [--points 4000]
                                              [--downsample 1 4]
"""
import argparse
import os
import sys

import numpy as np

sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "3d-gaussian-splatting_amd")]
import torch  # noqa: E402
from PIL import Image  # noqa: E402

import gs_colmap as gc  # noqa: E402
from gs_frame import FrameRenderer  # noqa: E402
from gs_scene import Camera, make_camera, make_scene  # noqa: E402


def build(out, n=20000, width=480, height=270, views=17, points=4000, downsample=(1,), seed=2023, arc_deg=16.0,
          device="cuda:0"):
    dev = torch.device(device)
    scene = make_scene(n, width, height, seed=seed)
    gt = [torch.from_numpy(a).to(dev) for a in (scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb)]
    fx = 0.75 * width
    cams = [make_camera(width, height, yaw_deg=float(y)) for y in np.linspace(-arc_deg, arc_deg, views)]
    os.makedirs(os.path.join(out, "sparse", "0"), exist_ok=True)
    gc.write_cameras_binary(os.path.join(out, "sparse", "0", "cameras.bin"),
                            {1: gc.ColmapCamera(1, "PINHOLE", width, height,
                                                np.array([fx, fx, width / 2, height / 2]))})
    images = {}
    for i, c in enumerate(cams):
        images[i + 1] = gc.ColmapImage(i + 1, gc.rotmat2qvec(c.rot), np.asarray(c.tran, np.float64), 1,
                                       f"frame_{i:03d}.png", np.zeros((0, 2)), np.zeros(0, np.int64))
    gc.write_images_binary(os.path.join(out, "sparse", "0", "images.bin"), images)
    rng = np.random.default_rng(seed + 1)
    vis = np.nonzero((scene.pos[:, 2] > 0.5))[0]
    pick = rng.choice(vis, size=min(points, len(vis)), replace=False)
    rgb8 = np.clip(255.0 / (1.0 + np.exp(-scene.rgb[pick, :3])), 1, 254).astype(np.uint8)
    pts = {int(j): gc.ColmapPoint3D(int(j), scene.pos[j].astype(np.float64), rgb8[k], 0.5, np.array([1, 2], np.int32),
                                    np.array([0, 0], np.int32)) for k, j in enumerate(pick)}
    gc.write_points3d_binary(os.path.join(out, "sparse", "0", "points3D.bin"), pts)
    r = FrameRenderer(dev, max_pairs=1 << 20)
    for k in downsample:
        os.makedirs(os.path.join(out, f"images_{k}"), exist_ok=True)
        for i, c in enumerate(cams):
            ck = Camera(width // k, height // k, c.focal_x / k, c.focal_y / k, c.rot, c.tran)
            img = r.forward(*gt, ck)[0]
            arr = (img.clamp(0, 1) * 255 + 0.5).to(torch.uint8).cpu().numpy()
            Image.fromarray(arr, "RGB").save(os.path.join(out, f"images_{k}", f"frame_{i:03d}.png"))
    return scene, cams


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--n", type=int, default=20000)
    ap.add_argument("--width", type=int, default=480)
    ap.add_argument("--height", type=int, default=270)
    ap.add_argument("--views", type=int, default=17)
    ap.add_argument("--points", type=int, default=4000)
    ap.add_argument("--downsample", type=int, nargs="+", default=[1])
    a = ap.parse_args()
    build(a.out, a.n, a.width, a.height, a.views, a.points, tuple(a.downsample))
    print("wrote", a.out)
