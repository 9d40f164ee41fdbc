"""``Splatter`` / ``Gaussian3ds`` with the reference's class API (splatter.py:38-228, 323-655) on the fused path.

The boundary the rest of the reference sees: train.py and visergui.py only reach the renderer through
``Splatter.forward`` and ``gaussian_3ds``.  This module keeps that surface --

    gaussian_splatter = Splatter(colmap_path, image_path, render_downsample=4, use_sh_coeff=0, ...)
    rendered = gaussian_splatter(camera_id)                       # [H, W, 3], differentiable
    loss = (rendered - gaussian_splatter.ground_truth).abs().mean(); loss.backward()
    gaussian_splatter.gaussian_3ds.pos.grad ...                   # nn.Parameters, any torch optimizer
    gaussian_splatter.gaussian_3ds.adaptive_control(grad, taus=..., delete_thresh=..., ...)
    gaussian_splatter(None, extrinsics={"rot", "tran"}, intrinsics={"width", "height", "focal_x", "focal_y"})

-- and replaces what is behind it: one ``gs_frame_forward`` / ``gs_frame_backward`` call per direction instead of
``project_and_culling`` + ``render`` (~25 torch kernels, >= 8 host synchronisations, the T x MAXP table), and the
three HIP launches of ``gs_densify`` instead of ~40 torch kernels in ``adaptive_control``.  A training loop written
against the reference's Splatter (its own train.py included) runs unchanged; ``gs_train.Trainer`` / ``train.py`` of
this package additionally fuse the loss and the optimizer.

Differences, all deliberate: ``cudaculling`` / ``jacobian_calc`` / ``fast_drawing`` / ``debug`` /
``debug_align`` select debugging variants of the reference and are accepted and ignored; images are decoded with
Pillow instead of cv2; ``n_tile_gaussians`` is read from the device on access (one synchronisation) instead of on
every frame.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch
from torch import nn

import gs_colmap
from gs_densify import adaptive_control as _adaptive_control
from gs_densify import inverse_sigmoid
from gs_frame import FrameRenderer
from gs_geometry import TileGrid
from gs_scene import Camera


class Gaussian3ds(nn.Module):
    """splatter.py:38-228: the parameter bag.  With ``init_values`` the tensors become nn.Parameters."""

    def __init__(self, pos, rgb, opa, quat=None, scale=None, cov=None, init_values=False):
        super().__init__()
        self.init_values = init_values
        wrap = (lambda t: t if t is None else nn.Parameter(t)) if init_values else (lambda t: t)
        self.pos, self.rgb, self.opa = wrap(pos), wrap(rgb), wrap(opa)
        self.quat, self.scale, self.cov = wrap(quat), wrap(scale), wrap(cov)

    def reset_opa(self):
        """splatter.py:119-120."""
        torch.nn.init.uniform_(self.opa, a=inverse_sigmoid(0.01), b=inverse_sigmoid(0.01))

    def adaptive_control(self, grad, taus, delete_thresh, scale_activation="abs", grad_thresh=0.0002,
                         grad_aggregation="max", use_clone=True, use_split=True, clone_dt=0.01, generator=None):
        """splatter.py:122-228: prune, clone, split; the five tensors are replaced by new nn.Parameters (so the
        caller re-creates its optimizer, as train.py:169-179 does)."""
        with torch.no_grad():
            params = [t.detach().contiguous() for t in (self.pos, self.quat, self.scale, self.opa, self.rgb)]
            new, counts = _adaptive_control(params, grad.detach().to(torch.float32).contiguous(), taus=taus,
                                            delete_thresh=delete_thresh, scale_activation=scale_activation,
                                            grad_thresh=grad_thresh, grad_aggregation=grad_aggregation,
                                            use_clone=bool(use_clone), use_split=bool(use_split), clone_dt=clone_dt,
                                            generator=generator)
        self.pos, self.quat, self.scale, self.opa, self.rgb = (nn.Parameter(t) for t in new)
        return counts


class Splatter(nn.Module):
    def __init__(self, colmap_path, image_path, near=0.3, jacobian_calc="cuda", render_downsample=1,
                 use_sh_coeff=False, render_weight_normalize=False, opa_init_value=0.1, scale_init_value=0.02,
                 tile_culling_method="prob2", tile_culling_dist_thresh=0.5, tile_culling_prob_thresh=0.1, debug=1,
                 scale_activation="abs", cudaculling=0, load_ckpt=None, debug_align=False, fast_drawing=False,
                 test=False, max_pairs: int = 1 << 20):
        super().__init__()
        if not torch.cuda.is_available():
            raise RuntimeError("Splatter needs a HIP device; there is no CPU fallback")
        if tile_culling_method not in ("prob2", "prob", "dist"):
            raise ValueError("tile_culling_method must be 'dist', 'prob' or 'prob2' (splatter.py:571)")
        if render_weight_normalize:
            raise NotImplementedError("render_weight_normalize is a flag of the reference-API draw(); train.py "
                                      "passes False (splatter.py:627)")
        assert jacobian_calc in ("cuda", "torch")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.use_sh_coeff, self.near, self.render_downsample = bool(use_sh_coeff), near, render_downsample
        self.tile_culling_method, self.tile_culling_prob_thresh = tile_culling_method, tile_culling_prob_thresh
        self.tile_culling_dist_thresh, self.scale_activation = tile_culling_dist_thresh, scale_activation
        self.debug, self.cudaculling, self.fast_drawing, self.jacobian_calc = debug, cudaculling, fast_drawing, jacobian_calc
        # splatter.py:362-366
        self.points3d = gs_colmap.read_points3d_binary(os.path.join(colmap_path, "points3D.bin"))
        self.cameras = gs_colmap.read_cameras_binary(os.path.join(colmap_path, "cameras.bin"))
        self.images_info = gs_colmap.read_images_binary(os.path.join(colmap_path, "images.bin"))
        self.image_path, self.test = image_path, test
        if not self.test:
            self.parse_imgs()
        # splatter.py:372-424: one Gaussian per sparse point (or the checkpoint's set)
        pos, quat, scale, opa, rgb = gs_colmap.initial_gaussians(self.points3d, scale_init_value, opa_init_value,
                                                                 scale_activation, self.use_sh_coeff)
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)  # noqa: E731
        self.gaussian_3ds = Gaussian3ds(pos=to(pos), rgb=to(rgb), opa=to(opa), quat=to(quat), scale=to(scale),
                                        init_values=True)
        if load_ckpt is not None:
            ck = torch.load(load_ckpt, map_location=self.device)
            for k in ("pos", "opa", "rgb", "quat", "scale"):
                setattr(self.gaussian_3ds, k, nn.Parameter(ck[k].detach().to(torch.float32).contiguous()))
        # "async": the pair capacity is checked one frame late from pinned memory -- no host synchronisation per frame
        self._renderer = FrameRenderer(self.device, max_pairs=max_pairs, training=True, thresh=tile_culling_prob_thresh,
                                       scale_activation=scale_activation, auto_grow="async",
                                       tile_culling_method=tile_culling_method,
                                       tile_culling_dist_thresh=tile_culling_dist_thresh)
        self._views_checked = set()  # (view, Gaussian count) pairs whose first frame has been capacity-checked
        self.current_camera = None
        self.ground_truth = None
        if not self.test:
            self.set_camera(0)

    # ------------------------------------------------------------------ dataset (splatter.py:429-465)
    def parse_imgs(self):
        from PIL import Image

        self.w2c_quats, self.w2c_rots, self.w2c_trans, self.cam_ids, self.imgs = [], [], [], [], []
        self._host_poses = []  # (rot [3,3], tran [3]) as host float32 arrays: set_camera never reads the device back
        for img_id in sorted(im.id for im in self.images_info.values()):
            info = self.images_info[img_id]
            fn = os.path.join(self.image_path, info.name)
            if not os.path.exists(fn):
                continue
            rgb = np.asarray(Image.open(fn).convert("RGB"), dtype=np.uint8)  # cv2.imread + BGR2RGB
            self.imgs.append(torch.from_numpy(rgb.copy()).to(torch.uint8).to(self.device))
            self.w2c_quats.append(torch.from_numpy(np.asarray(info.qvec)).to(torch.float32).to(self.device))
            self.w2c_trans.append(torch.from_numpy(np.asarray(info.tvec)).to(torch.float32).to(self.device))
            rot = gs_colmap.qvec2rotmat(info.qvec)
            self.w2c_rots.append(torch.from_numpy(rot).to(torch.float32).to(self.device))
            self._host_poses.append((np.asarray(rot, np.float32).reshape(3, 3).copy(),
                                     np.asarray(info.tvec, np.float32).reshape(3).copy()))
            self.cam_ids.append(info.camera_id)

    def switch_resolution(self, downsample_factor):
        if downsample_factor == self.render_downsample:
            return
        self.image_path = self.image_path.replace(f"images_{self.render_downsample}", f"images_{downsample_factor}")
        self.render_downsample = downsample_factor
        self.parse_imgs()
        self.current_camera = None
        self.set_camera(0)

    def set_camera(self, idx, extrinsics=None, intrinsics=None):
        """splatter.py:467-511."""
        if idx is None:
            to = lambda a: (a if torch.is_tensor(a) else torch.from_numpy(np.asarray(a))).to(torch.float32).to(self.device)  # noqa: E731
            host = lambda a: (a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)).astype(np.float32)  # noqa: E731
            self.current_w2c_rot, self.current_w2c_tran = to(extrinsics["rot"]), to(extrinsics["tran"])
            rot_h, tran_h = host(extrinsics["rot"]).reshape(3, 3), host(extrinsics["tran"]).reshape(3)
            self.current_w2c_quat, self.ground_truth = None, None
            width, height = math.ceil(intrinsics["width"]), math.ceil(intrinsics["height"])
            fx, fy = float(intrinsics["focal_x"]), float(intrinsics["focal_y"])
            self.current_camera = None
        else:
            self.current_w2c_quat, self.current_w2c_tran = self.w2c_quats[idx], self.w2c_trans[idx]
            self.current_w2c_rot = self.w2c_rots[idx]
            rot_h, tran_h = self._host_poses[idx]
            self.ground_truth = self.imgs[idx].to(torch.float16) / 255.
            cam = self.cameras[self.cam_ids[idx]]
            self.current_camera = cam
            fx, fy = cam.params[0] / self.render_downsample, cam.params[1] / self.render_downsample
            width, height = int(self.ground_truth.shape[1]), int(self.ground_truth.shape[0])
        self.tile_info = TileGrid(width, height, fx, fy)
        # host copies of the pose (kept since parse_imgs): no device read-back, hence no synchronisation, per frame
        self._camera = Camera(width, height, float(fx), float(fy), rot_h, tran_h, near=self.near)

    # ------------------------------------------------------------------ the frame (splatter.py:513-655)
    @property
    def n_gaussians(self) -> int:
        return int(self.gaussian_3ds.pos.shape[0])

    @property
    def n_tile_gaussians(self) -> int:
        """Number of (tile, Gaussian) pairs of the last frame (one device read)."""
        return int(self._renderer.stats().pairs)

    @property
    def culling_mask(self) -> torch.Tensor:
        """[N] int64, 1 where the Gaussian passed the frustum test of the last frame (train.py:150 accumulates it
        for the "mean" statistic).  Computed on access from the frame's workspace."""
        return self._renderer.culling_mask().to(torch.int64)

    def forward(self, camera_id=None, extrinsics=None, intrinsics=None):
        self.set_camera(camera_id, extrinsics, intrinsics)
        g = self.gaussian_3ds
        key = (camera_id, int(g.pos.shape[0]))
        if camera_id is None or key not in self._views_checked:
            # first frame of this view on this Gaussian set: capacity checked synchronously (and the frame redone in
            # a larger workspace if needed); later frames of the view use the asynchronous counters
            self._views_checked.add(key)
            self._renderer._checked_once = False
        return self._renderer.render(g.pos, g.quat, g.scale, g.opa, g.rgb, self._camera)
