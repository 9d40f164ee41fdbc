"""Shared helpers for the parity tests: oracle-side frame pipeline and comparisons."""
from __future__ import annotations

import numpy as np

import oracle
from gs_geometry import RayBasis, TileGrid
from gs_scene import Camera, Scene


def sigmoid32(x):
    x = np.asarray(x, np.float32)
    return (np.float32(1) / (np.float32(1) + np.exp(-x, dtype=np.float32))).astype(np.float32)


def activate(scene: Scene, scale_activation="abs"):
    """splatter.py:519-524 in fp32 (same expression order as the kernels)."""
    q = scene.quat.astype(np.float32)
    nr = np.sqrt(((q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1]) + q[:, 2] * q[:, 2]) + q[:, 3] * q[:, 3], dtype=np.float32)
    qn = (q / nr[:, None]).astype(np.float32)
    if scale_activation == "abs":
        sn = (np.abs(scene.scale) + np.float32(1e-4)).astype(np.float32)
    else:
        sn = np.exp(scene.scale, dtype=np.float32)
    return qn, sn


def frame_scalars(cam: Camera):
    grid = TileGrid(cam.width, cam.height, cam.focal_x, cam.focal_y)
    hw, hh = grid.frustum_half_extents()
    rays = RayBasis.from_camera(cam.rot, cam.tran, grid.padded_height, grid.padded_width, grid.focal_x, grid.focal_y)
    return grid, hw, hh, rays


class OracleFrame:
    """Oracle restatement of one forward frame on raw parameters, keeping every intermediate."""

    def __init__(self, scene: Scene, cam: Camera, thresh=0.05, scale_activation="abs", tile_culling_method="prob2"):
        self.scene, self.cam = scene, cam
        self.scale_activation = scale_activation
        grid, hw, hh, rays = frame_scalars(cam)
        self.grid, self.rays = grid, rays
        self.qn, self.sn = activate(scene, scale_activation)
        self.pos_i, self.cov, self.mask = oracle.global_culling(scene.pos, self.qn, self.sn, cam.rot, cam.tran,
                                                                cam.near, hw, hh)
        if tile_culling_method == "prob2":
            self.keys, self.ids, self.accum = oracle.sorted_pairs(
                self.pos_i, self.cov.reshape(-1, 4), self.mask, thresh, grid.tile_geo_length_x,
                grid.tile_geo_length_y, grid.n_tile_x, grid.n_tile_y, grid.leftmost, grid.topmost)
        else:
            self.keys, self.ids, self.accum = self._pairs_from_table(tile_culling_method, thresh)
        self.opa_act = sigmoid32(scene.opa)
        self.col_act = scene.rgb if scene.use_sh else sigmoid32(scene.rgb)
        ids = self.ids
        self.s_pos, self.s_cov = self.pos_i[ids], self.cov.reshape(-1, 4)[ids]
        self.s_opa, self.s_rgb = self.opa_act[ids], self.col_act[ids]
        self.padded = oracle.draw(self.s_pos, self.s_rgb, self.s_opa, self.s_cov, self.accum, grid.padded_height,
                                  grid.padded_width, grid.focal_x, grid.focal_y, use_sh=scene.use_sh, fast=True,
                                  rays_o=rays.rays_o, lefttop=rays.lefttop, vdx=rays.dx, vdy=rays.dy)
        self.image = grid.crop(np.clip(self.padded, 0, 1))

    def _pairs_from_table(self, method, thresh):
        """'prob' / 'dist' (splatter.py:571-578): the oracle's restatement of calc_tile_list methods 1 / 0 (pinned
        bit for bit against the reference kernels) on the visible Gaussians, uncapped, then the canonical
        (tile, depth bits, Gaussian index) order."""
        grid = self.grid
        vis = np.nonzero(self.mask)[0]
        top, bottom, left, right = grid.tile_edges()
        th = (grid.tile_geo_length_x / 0.5) ** 2 if method == "dist" else thresh  # splatter.py:577, dist_thresh 0.5
        cnt, lst = oracle.calc_tile_list(self.pos_i[vis], self.cov.reshape(-1, 4)[vis], len(vis), th,
                                         {"dist": 0, "prob": 1}[method], grid.tile_geo_length_x,
                                         grid.tile_geo_length_y, grid.n_tile_x, grid.n_tile_y, grid.leftmost,
                                         grid.topmost, top=top, bottom=bottom, left=left, right=right)
        tiles = np.repeat(np.arange(len(cnt)), cnt)
        ids = np.concatenate([vis[lst[t, :c]] for t, c in enumerate(cnt)]) if cnt.sum() else np.zeros(0, np.int64)
        dbits = self.pos_i[ids, 2].view(np.uint32).astype(np.uint64)
        keys = (tiles.astype(np.uint64) << np.uint64(32)) | dbits
        order = np.lexsort((ids, keys))
        accum = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
        return keys[order], ids[order].astype(np.int32), accum

    def backward(self, grad_image):
        """dL/d(image) -> dict of dL/d(raw parameter), the chain splatter.py's autograd runs."""
        sc, cam, grid, rays = self.scene, self.cam, self.grid, self.rays
        n = sc.n
        top, left = grid.crop_offsets()
        gpad = np.zeros_like(self.padded)
        inside = ((self.padded >= 0) & (self.padded <= 1)).astype(np.float32)
        gpad[top:top + grid.height, left:left + grid.width] = grad_image
        gpad *= inside
        gp, gr, go, gc = oracle.draw_backward(self.s_pos, self.s_rgb, self.s_opa, self.s_cov, self.accum, self.padded,
                                              gpad, grid.focal_x, grid.focal_y, use_sh=sc.use_sh, fast=True,
                                              rays_o=rays.rays_o, lefttop=rays.lefttop, vdx=rays.dx, vdy=rays.dy)
        self.pair_grads = (gp, gr, go, gc)
        # index backward (index_put accumulate) in double
        d_pos_i = np.zeros((n, 3), np.float64)
        d_cov = np.zeros((n, 4), np.float64)
        d_opa = np.zeros(n, np.float64)
        d_col = np.zeros((n, sc.rgb.shape[1]), np.float64)
        np.add.at(d_pos_i, self.ids, gp)
        np.add.at(d_cov, self.ids, gc)
        np.add.at(d_opa, self.ids, go)
        np.add.at(d_col, self.ids, gr)
        g_pos, g_qn, g_sn = oracle.global_culling_backward(sc.pos, self.qn, self.sn, cam.rot, cam.tran,
                                                           d_pos_i.astype(np.float32), d_cov.astype(np.float32),
                                                           self.mask)
        q = sc.quat.astype(np.float64)
        nr = np.linalg.norm(q, axis=1, keepdims=True)
        qh = q / nr
        g_q = (g_qn - qh * np.sum(qh * g_qn, axis=1, keepdims=True)) / nr
        if self.scale_activation == "abs":
            g_s = g_sn * np.sign(sc.scale)
        else:
            g_s = g_sn * np.exp(np.clip(sc.scale, -1, 1))
        o = self.opa_act.astype(np.float64)
        g_o = d_opa * o * (1 - o)
        if sc.use_sh:
            g_c = d_col
        else:
            c = self.col_act.astype(np.float64)
            g_c = d_col * c * (1 - c)
        return {"pos": g_pos.astype(np.float32), "quat": g_q.astype(np.float32), "scale": g_s.astype(np.float32),
                "opa": g_o.astype(np.float32), "rgb": g_c.astype(np.float32)}


def rel_err(a, b):
    """max |a-b| / (max |b| + tiny): scale-aware error for gradient tensors."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def to_torch(scene: Scene, device, requires_grad=False):
    import torch

    out = []
    for a in (scene.pos, scene.quat, scene.scale, scene.opa, scene.rgb):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(device)
        t.requires_grad_(requires_grad)
        out.append(t)
    return out
