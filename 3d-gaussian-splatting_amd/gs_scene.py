"""Synthetic scenes and cameras for the rasterizer path (SURVEY.md section 8d contract).

The reference has no scene generator (it reads COLMAP Garden, splatter.py:324-454); there
is no dataset in this environment, so benchmarks and tests use this seeded generator.  The
distributions are chosen so that ~79 % of the Gaussians survive frustum culling and each
visible Gaussian touches ~3.6 tiles, which is the Garden-like load SURVEY.md section 8 sizes.
Everything is produced with NumPy ``default_rng(seed)`` so CPU-only tests and the GPU
box see identical bytes.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

C0 = 0.28209479177387814  # gaussian.cu:385


@dataclass
class Camera:
    """Pinhole camera as the reference passes it around (splatter.py:467-511)."""

    width: int
    height: int
    focal_x: float
    focal_y: float
    rot: np.ndarray  # [3,3] world->camera rotation (current_w2c_rot)
    tran: np.ndarray  # [3]   world->camera translation (current_w2c_tran)
    near: float = 0.3  # splatter.py:327


@dataclass
class Scene:
    pos: np.ndarray  # [N,3]
    quat: np.ndarray  # [N,4] raw (un-normalised) w,x,y,z
    scale: np.ndarray  # [N,3] raw; activation "abs": |s|+1e-4 (splatter.py:521)
    opa: np.ndarray  # [N]   logit
    rgb: np.ndarray  # [N,3] logit, or [N,27] SH coeffs (channel-major 3x9, utils.py:345-348); [N,48] = 3x16 (degree 3)

    @property
    def n(self) -> int:
        return self.pos.shape[0]

    @property
    def use_sh(self) -> bool:
        return self.rgb.shape[1] in (27, 48)


def make_camera(width: int, height: int, yaw_deg: float = 0.0) -> Camera:
    """Identity pose (optionally yawed about +y), fx = fy = 0.75 * W."""
    f = 0.75 * width
    a = math.radians(yaw_deg)
    rot = np.array([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]],
                   dtype=np.float32)
    return Camera(width, height, f, f, rot, np.zeros(3, np.float32))


def make_scene(n: int, width: int, height: int, seed: int = 2023, use_sh: bool = False,
               max_px_sigma: float = 16.0, sh_degree: int = 2) -> Scene:
    """SURVEY.md section 8d generator (camera = make_camera(width, height)).  ``sh_degree`` 2 is what the
    reference implements (9 functions per channel); 3 (16 per channel) is the extension BASELINE config 4 names."""
    rng = np.random.default_rng(seed)
    fx = 0.75 * width
    z = rng.uniform(-0.5, 10.0, n)
    u = rng.uniform(-1.0, 1.0, n)
    v = rng.uniform(-1.0, 1.0, n)
    x = u * 1.3 * (width / 2 / fx) * z
    y = v * 1.3 * (height / 2 / fx) * z
    pos = np.stack([x, y, z], 1).astype(np.float32)
    s_px = np.exp(rng.uniform(0.0, math.log(max_px_sigma), n))
    scale = (s_px * np.abs(z) / fx)[:, None] * rng.uniform(0.25, 1.0, (n, 3))
    quat = rng.normal(size=(n, 4))
    quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    quat *= rng.uniform(0.5, 2.0, (n, 1))  # raw parameters are not unit length
    opa = rng.normal(0.0, 2.0, n)
    if use_sh:
        if sh_degree not in (2, 3):
            raise ValueError("sh_degree must be 2 or 3")
        nb = (sh_degree + 1) ** 2
        rgb = rng.normal(0.0, 0.5, (n, 3, nb))
        rgb[:, :, 0] = rng.normal(0.0, 1.0, (n, 3)) / C0  # DC like initialize_sh
        rgb = rgb.reshape(n, 3 * nb)
    else:
        rgb = rng.normal(0.0, 1.0, (n, 3))
    return Scene(pos, quat.astype(np.float32), scale.astype(np.float32), opa.astype(np.float32),
                 rgb.astype(np.float32))


def make_trained_like_scene(n: int = 724_312, width: int = 1920, height: int = 1080, seed: int = 7, use_sh: bool = False,
                            sh_degree: int = 2, max_px_sigma: float = 26.0, clustered: float = 0.32, n_clusters: int = 30,
                            opa_shift: float = -3.5, cluster_opa_shift: float = -3.0) -> Scene:
    """A deterministic scene in the STATE a trained / densified model is in (missing item 2): translucent
    (opacity logits shifted by ``opa_shift``: no tile saturates, every pair of every list is composited), larger
    footprints (pixel sigma up to ``max_px_sigma``: ~6.5 tiles per visible Gaussian) and a heavy tail of tile-list
    lengths (a share ``clustered`` of the Gaussians sits in ``n_clusters`` screen-space clusters of 1 - 3 tiles' sigma).
    The defaults reproduce the end state of the densifying SH run of tools/soak.py (profiles/r05_m_soak_end_state.jsonl:
    724,312 Gaussians, 3.96 M pairs, list length mean 485 / median 317 / p99 2,391 / max 4,065, 2,277 tiles beyond 512):
    this generator gives 619,582 visible, 3,925,915 pairs, mean 481, median 312, p99 2,696, max 5,361, 1,673 tiles beyond
    512 at 1080p -- without 2,000 training iterations in front of the measurement."""
    rng = np.random.default_rng(seed + 1_000_003)
    sc = make_scene(n, width, height, seed=seed, use_sh=use_sh, max_px_sigma=max_px_sigma, sh_degree=sh_degree)
    fx = 0.75 * width
    m = int(clustered * n)
    idx = rng.choice(n, m, replace=False)
    cx = rng.uniform(-0.9, 0.9, n_clusters) * (width / 2 / fx)
    cy = rng.uniform(-0.9, 0.9, n_clusters) * (height / 2 / fx)
    cs = np.exp(rng.uniform(0.0, math.log(3.0), n_clusters)) * 16 / fx  # cluster sigma: 1 .. 3 tiles
    which = rng.integers(0, n_clusters, m)
    z = np.abs(sc.pos[idx, 2]) + 0.5
    sc.pos[idx, 2] = z
    sc.pos[idx, 0] = ((cx[which] + cs[which] * rng.normal(size=m)) * z).astype(np.float32)
    sc.pos[idx, 1] = ((cy[which] + cs[which] * rng.normal(size=m)) * z).astype(np.float32)
    sc.opa = (sc.opa + opa_shift).astype(np.float32)
    # the Gaussians of a cluster are fainter still: a trained model's long lists are walked to their end (the end states of
    # the densifying runs composite every pair: profiles/r06_g_*), whereas thousands of opa_shift Gaussians on top of each
    # other would stop the cluster tiles' pixels half-way -- and a walk that stops early hides the long-list problem
    sc.opa[idx] = (sc.opa[idx] + cluster_opa_shift).astype(np.float32)
    return sc


# BASELINE.json configs -> (n_gaussians, width, height, use_sh)
CONFIGS = {
    "cfg1": (10_000, 256, 256, False),
    "cfg2": (376_467, 1920, 1080, False),
    "cfg3": (506_627, 1920, 1080, False),
    "cfg4": (2_400_000, 1920, 1080, True),
    "cfg5": (2_400_000, 1920, 1080, False),
    "cfg6": (10_000_000, 1920, 1080, False),  # dense-scene stress (not a BASELINE config): ~3,500 pairs per tile
}
