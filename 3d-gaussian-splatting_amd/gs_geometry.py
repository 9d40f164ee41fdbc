"""Tile grid and per-pixel ray basis: host-side mirror of the reference's frame geometry.

``TileGrid`` restates ``splatter.Tiles`` (splatter.py:255-300) and ``RayBasis`` restates
``splatter.RayInfo`` (splatter.py:305-321).  Pure Python/NumPy: all quantities are a handful
of scalars that are computed once per camera in double precision (as the reference does in
Python) and handed to the kernels as fp32.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

TILE = 16


@dataclass
class TileGrid:
    width: int
    height: int
    focal_x: float
    focal_y: float

    def __post_init__(self):
        # splatter.py:259-264
        self.padded_width = int(math.ceil(self.width / TILE)) * TILE
        self.padded_height = int(math.ceil(self.height / TILE)) * TILE
        self.n_tile_x = self.padded_width // TILE
        self.n_tile_y = self.padded_height // TILE
        # splatter.py:279-282
        self.tile_geo_length_x = TILE / self.focal_x
        self.tile_geo_length_y = TILE / self.focal_y
        self.leftmost = -self.padded_width / 2 / self.focal_x
        self.topmost = -self.padded_height / 2 / self.focal_y

    def __len__(self) -> int:
        return self.n_tile_x * self.n_tile_y

    @property
    def n_tiles(self) -> int:
        return self.n_tile_x * self.n_tile_y

    def crop_offsets(self):
        """splatter.py:267-272: centred crop of the padded image."""
        top = int(self.padded_height - self.height) // 2
        left = int(self.padded_width - self.width) // 2
        return top, left

    def crop(self, image):
        top, left = self.crop_offsets()
        return image[top:top + int(self.height), left:left + int(self.width), :]

    def tile_edges(self):
        """top/bottom/left/right per tile in normalised image units (splatter.py:275-293),
        only consumed by the O(V*T) binning methods 0 and 1."""
        left = np.linspace(-self.padded_width / 2, self.padded_width / 2, self.n_tile_x + 1,
                           dtype=np.float32)[:-1]
        top = np.linspace(-self.padded_height / 2, self.padded_height / 2, self.n_tile_y + 1,
                          dtype=np.float32)[:-1]
        right, bottom = left + TILE, top + TILE
        left, right = left / np.float32(self.focal_x), right / np.float32(self.focal_x)
        top, bottom = top / np.float32(self.focal_y), bottom / np.float32(self.focal_y)
        tl = np.tile(left, self.n_tile_y)  # "b -> (c b)"
        tr = np.tile(right, self.n_tile_y)
        tt = np.repeat(top, self.n_tile_x)  # "b -> (b c)"
        tb = np.repeat(bottom, self.n_tile_x)
        return tt.astype(np.float32), tb.astype(np.float32), tl.astype(np.float32), tr.astype(np.float32)

    def frustum_half_extents(self):
        """splatter.py:532-533: the 1.2x guard band used by global_culling."""
        return (self.width * 1.2 / 2 / self.focal_x, self.height * 1.2 / 2 / self.focal_y)


@dataclass
class RayBasis:
    """rays_o / lefttop / dx / dy of splatter.RayInfo, as float32[3] arrays."""

    rays_o: np.ndarray
    lefttop: np.ndarray
    dx: np.ndarray
    dy: np.ndarray

    @staticmethod
    def from_camera(rot, tran, padded_h: int, padded_w: int, focal_x: float, focal_y: float) -> "RayBasis":
        w2c = np.asarray(rot, dtype=np.float32)
        t = np.asarray(tran, dtype=np.float32)
        c2w = np.linalg.inv(w2c).astype(np.float32)  # splatter.py:308
        rays_o = -(c2w @ t)  # :314
        lefttop_cam = np.array([(-padded_w / 2 + 0.5) / focal_x, (-padded_h / 2 + 0.5) / focal_y, 1.0],
                               dtype=np.float32)  # :316
        dx_cam = np.array([1.0 / focal_x, 0, 0], dtype=np.float32)
        dy_cam = np.array([0, 1.0 / focal_y, 0], dtype=np.float32)
        return RayBasis(rays_o.astype(np.float32), (c2w @ (lefttop_cam - t)).astype(np.float32),
                        (c2w @ dx_cam).astype(np.float32), (c2w @ dy_cam).astype(np.float32))
