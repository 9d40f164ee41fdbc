"""COLMAP sparse-model readers and the initial Gaussian set.

The reference loads ``cameras.bin`` / ``images.bin`` / ``points3D.bin`` (splatter.py:362-364, through the
COLMAP script functions vendored in its utils.py) and seeds one Gaussian per 3-D point (splatter.py:372-406):
position = the point, colour logit = logit(rgb / 255) (SH: DC coefficient via ``initialize_sh``), opacity
logit = logit(0.3), identity quaternion, isotropic scale = mean distance to the three nearest neighbours.
Host-side code (it runs once per scene); the readers parse each file with one pass over a bytes buffer.
Record layouts: COLMAP ``src/base/reconstruction.cc`` (Read*Binary).
"""
from __future__ import annotations

import math
import struct
from dataclasses import dataclass
from typing import Dict

import numpy as np

# model_id -> (name, number of parameters), COLMAP src/base/camera_models.h
CAMERA_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5),
                 4: ("OPENCV", 8), 5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5),
                 8: ("SIMPLE_RADIAL_FISHEYE", 4), 9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}


@dataclass(frozen=True)
class ColmapCamera:
    id: int
    model: str
    width: int
    height: int
    params: np.ndarray


@dataclass(frozen=True)
class ColmapImage:
    id: int
    qvec: np.ndarray  # w, x, y, z (world -> camera)
    tvec: np.ndarray
    camera_id: int
    name: str
    xys: np.ndarray  # [n, 2]
    point3D_ids: np.ndarray  # [n] int64, -1 = no 3-D point

    def qvec2rotmat(self) -> np.ndarray:
        return qvec2rotmat(self.qvec)


@dataclass(frozen=True)
class ColmapPoint3D:
    id: int
    xyz: np.ndarray
    rgb: np.ndarray  # uint8
    error: float
    image_ids: np.ndarray
    point2D_idxs: np.ndarray


def qvec2rotmat(q) -> np.ndarray:
    w, x, y, z = (float(v) for v in q)
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                     [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                     [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


def read_cameras_binary(path) -> Dict[int, ColmapCamera]:
    buf = memoryview(open(path, "rb").read())
    (count,), off, out = struct.unpack_from("<Q", buf, 0), 8, {}
    for _ in range(count):
        cam_id, model_id, width, height = struct.unpack_from("<iiQQ", buf, off)
        off += 24
        name, n_par = CAMERA_MODELS[model_id]
        params = np.frombuffer(buf, "<f8", n_par, off).copy()
        off += 8 * n_par
        out[cam_id] = ColmapCamera(cam_id, name, width, height, params)
    assert off == len(buf), "trailing bytes in cameras.bin"
    return out


def read_images_binary(path) -> Dict[int, ColmapImage]:
    raw = open(path, "rb").read()
    buf = memoryview(raw)
    (count,), off, out = struct.unpack_from("<Q", buf, 0), 8, {}
    obs = np.dtype([("xy", "<f8", 2), ("pid", "<i8")])
    for _ in range(count):
        image_id = struct.unpack_from("<i", buf, off)[0]
        pose = np.frombuffer(buf, "<f8", 7, off + 4)
        camera_id = struct.unpack_from("<i", buf, off + 60)[0]
        end = raw.index(b"\x00", off + 64)  # null-terminated name
        name = raw[off + 64:end].decode("utf-8")
        n2d = struct.unpack_from("<Q", buf, end + 1)[0]
        rec = np.frombuffer(buf, obs, n2d, end + 9)
        off = end + 9 + obs.itemsize * n2d
        out[image_id] = ColmapImage(image_id, pose[:4].copy(), pose[4:].copy(), camera_id, name, rec["xy"].copy(),
                                    rec["pid"].copy())
    assert off == len(buf), "trailing bytes in images.bin"
    return out


def read_points3d_binary(path) -> Dict[int, ColmapPoint3D]:
    buf = memoryview(open(path, "rb").read())
    (count,), off, out = struct.unpack_from("<Q", buf, 0), 8, {}
    for _ in range(count):
        pid = struct.unpack_from("<q", buf, off)[0]
        xyz = np.frombuffer(buf, "<f8", 3, off + 8).copy()
        rgb = np.frombuffer(buf, "u1", 3, off + 32).copy()
        error, track = struct.unpack_from("<dQ", buf, off + 35)
        tr = np.frombuffer(buf, "<i4", 2 * track, off + 51).reshape(track, 2)
        off += 51 + 8 * track
        out[pid] = ColmapPoint3D(pid, xyz, rgb, error, tr[:, 0].copy(), tr[:, 1].copy())
    assert off == len(buf), "trailing bytes in points3D.bin"
    return out


def inverse_sigmoid(y):  # utils.py:350-354
    return -np.log(1 / y - 1)


def initialize_sh(rgb_logits: np.ndarray) -> np.ndarray:
    """utils.py:345-348: 27 coefficients per Gaussian, the DC term of each channel carries the logit."""
    n = len(rgb_logits)
    sh = np.zeros((n, 3, 9), np.float32)
    sh[:, :, 0] = rgb_logits / 0.28209479177387814
    return sh.reshape(n, 27)


def initial_gaussians(points3d: Dict[int, ColmapPoint3D], scale_init_value: float = 1.0, opa_init_value: float = 0.3,
                      scale_activation: str = "abs", use_sh_coeff: bool = False):
    """(pos, quat, scale, opa, rgb) float32 arrays as Splatter.__init__ builds them (splatter.py:372-406).
    The three-nearest-neighbour distances come from an exact KD-tree query, as in the reference (pykdtree there,
    scipy's cKDTree here: both return exact Euclidean distances)."""
    from scipy.spatial import cKDTree

    pts = list(points3d.values())
    pos = np.stack([p.xyz for p in pts]).astype(np.float32)
    rgb = inverse_sigmoid(np.stack([p.rgb for p in pts]) / 255.0).astype(np.float32)
    if use_sh_coeff:
        rgb = initialize_sh(rgb)
    dist, _ = cKDTree(pos).query(pos, k=4)
    s = (dist[:, 1:].mean(axis=1).astype(np.float32) * np.float32(scale_init_value)).astype(np.float32)
    if scale_activation == "exp":
        s = np.log(s)
    n = len(pts)
    quat = np.tile(np.array([1, 0, 0, 0], np.float32), (n, 1))
    opa = np.full(n, -math.log(1 / opa_init_value - 1), np.float32)
    return pos, quat, np.repeat(s[:, None], 3, axis=1).astype(np.float32), opa, rgb


# ------------------------------------------------------------------------------------------------
# Writers (record layout of COLMAP's Reconstruction::Write*Binary): used to build synthetic datasets in the
# layout train.py expects -- <data>/sparse/0/{cameras,images,points3D}.bin + <data>/images_<downsample>/ --
# because no real capture is available offline (tools/make_synthetic_colmap.py, tests).
MODEL_IDS = {name: (mid, n_par) for mid, (name, n_par) in CAMERA_MODELS.items()}


def write_cameras_binary(path, cameras: Dict[int, ColmapCamera]):
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(cameras)))
        for cam in cameras.values():
            mid, n_par = MODEL_IDS[cam.model]
            assert len(cam.params) == n_par, f"{cam.model} takes {n_par} parameters"
            f.write(struct.pack("<iiQQ", cam.id, mid, cam.width, cam.height))
            f.write(struct.pack("<%dd" % n_par, *(float(v) for v in cam.params)))


def write_images_binary(path, images: Dict[int, ColmapImage]):
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(images)))
        for im in images.values():
            f.write(struct.pack("<i7di", im.id, *(float(v) for v in im.qvec), *(float(v) for v in im.tvec),
                                im.camera_id))
            f.write(im.name.encode("utf-8") + b"\x00")
            f.write(struct.pack("<Q", len(im.point3D_ids)))
            for xy, pid in zip(np.asarray(im.xys, np.float64).reshape(-1, 2), im.point3D_ids):
                f.write(struct.pack("<ddq", float(xy[0]), float(xy[1]), int(pid)))


def write_points3d_binary(path, points: Dict[int, ColmapPoint3D]):
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(points)))
        for p in points.values():
            f.write(struct.pack("<q3d3BdQ", p.id, *(float(v) for v in p.xyz), *(int(v) for v in p.rgb),
                                float(p.error), len(p.image_ids)))
            for a, b in zip(p.image_ids, p.point2D_idxs):
                f.write(struct.pack("<ii", int(a), int(b)))


def rotmat2qvec(R) -> np.ndarray:
    """Inverse of qvec2rotmat (COLMAP's read_write_model.rotmat2qvec: eigenvector of the symmetric K matrix)."""
    Rxx, Ryx, Rzx, Rxy, Ryy, Rzy, Rxz, Ryz, Rzz = np.asarray(R, np.float64).flat
    K = np.array([[Rxx - Ryy - Rzz, 0, 0, 0], [Ryx + Rxy, Ryy - Rxx - Rzz, 0, 0],
                  [Rzx + Rxz, Rzy + Ryz, Rzz - Rxx - Ryy, 0], [Ryz - Rzy, Rzx - Rxz, Rxy - Ryx, Rxx + Ryy + Rzz]]) / 3.0
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    return -q if q[0] < 0 else q


# ------------------------------------------------------------------------------------------------
@dataclass
class ColmapScene:
    """What Splatter.__init__ / parse_imgs / set_camera hold for a dataset (splatter.py:356-366, 429-452, 490-503)."""
    cameras: list          # gs_scene.Camera per image, sorted by COLMAP image id
    targets: list          # [H, W, 3] float32 tensors in [0, 1]
    names: list            # image file names
    points3d: Dict[int, ColmapPoint3D]


def load_scene(data_dir: str, render_downsample: int = 4, device="cpu", near: float = 0.3,
               image_dir: str = None) -> ColmapScene:
    """``<data_dir>/sparse/0/*.bin`` + ``<data_dir>/images_<render_downsample>/`` (train.py:368-369), as the
    reference reads them:

    * images sorted by COLMAP image id, files that do not exist are skipped (splatter.py:430-441);
    * pose: COLMAP's world->camera rotation and translation as they are (splatter.py:445-451);
    * intrinsics: focal = params[0:2] / render_downsample, size = the size of the LOADED image
      (splatter.py:497-501) -- the pre-downsampled ``images_<k>`` folders of the usual capture layout;
    * ground truth: uint8 -> float16 / 255 (splatter.py:443, 494), kept as float32 here because the loss runs in
      fp32 (the fp16 rounding of the reference's targets is applied, so the values are the same).

    The reference decodes with cv2.imread (absent in this image); Pillow decodes here -- PNG is lossless and both
    use libjpeg-turbo for JPEG."""
    import os

    import torch
    from PIL import Image

    from gs_scene import Camera

    sparse = os.path.join(data_dir, "sparse", "0")
    cams = read_cameras_binary(os.path.join(sparse, "cameras.bin"))
    imgs = read_images_binary(os.path.join(sparse, "images.bin"))
    pts = read_points3d_binary(os.path.join(sparse, "points3D.bin"))
    image_dir = image_dir or os.path.join(data_dir, f"images_{render_downsample}")
    out = ColmapScene([], [], [], pts)
    for image_id in sorted(imgs):
        info = imgs[image_id]
        fn = os.path.join(image_dir, info.name)
        if not os.path.exists(fn):
            continue
        rgb = np.asarray(Image.open(fn).convert("RGB"), dtype=np.uint8)
        target = (torch.from_numpy(rgb.copy()).to(device).to(torch.float16) / 255.0).to(torch.float32).contiguous()
        cam = cams[info.camera_id]
        if len(cam.params) < 2:
            raise RuntimeError(f"camera model {cam.model} has no focal parameters")
        fx, fy = float(cam.params[0]) / render_downsample, float(cam.params[1]) / render_downsample
        out.cameras.append(Camera(int(rgb.shape[1]), int(rgb.shape[0]), fx, fy,
                                  qvec2rotmat(info.qvec).astype(np.float32), np.asarray(info.tvec, np.float32),
                                  near=near))
        out.targets.append(target)
        out.names.append(info.name)
    if not out.cameras:
        raise RuntimeError(f"no image of {sparse}/images.bin found in {image_dir}")
    return out
