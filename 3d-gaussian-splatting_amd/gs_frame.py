"""Fused MI355X frame path: host-side mirror of ``Splatter.forward`` (splatter.py:513-655).

``FrameRenderer`` owns one workspace (a torch uint8 tensor living in HBM) and drives
``gs_frame_forward`` / ``gs_frame_backward`` of libgs_amd.so: cull + project + activations +
tile counting, scan + key emission, device radix sort on (tile, depth) keys, tile ranges and
the 16x16-tile compositing forward/backward -- one C call per direction, no host
synchronisation, no T x MAXP table, no sorted attribute copies.  ``render`` is the autograd
entry point (raw parameters in, clamped + cropped image out, exactly what
``Splatter.forward`` returns).

Semantics follow the reference with train.py's defaults (cudaculling=1,
tile_culling_method="prob2", fast_drawing=1, render_weight_normalize=False); the known
reference defects listed in SURVEY.md section 0 (per-tile cap MAXP, float32 sort key, racy
reductions) are not reproduced.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from gaussian import _lib
from gs_geometry import RayBasis, TileGrid

SCALE_ACT = {"abs": 0, "exp": 1}
TILE_CULLING = {"dist": 0, "prob": 1, "prob2": 2}  # the reference's `_method_config` (splatter.py:571)


@dataclass
class FrameStats:
    visible: int  # V: Gaussians surviving frustum culling
    pairs: int  # M: (tile, Gaussian) pairs after duplication (clamped to capacity)
    overflow: int  # 0, or the true M when it exceeded the workspace capacity
    buckets: int  # 64-Gaussian buckets in the last backward's work list
    longest_list: int = 0  # longest tile list of the frame (lists of up to 1024 pairs are reported as 0)
    saturated_buckets: int = 0  # ... of them in tiles whose compositing stopped before the end of their list
    cull_fallback: bool = False  # an occlusion-culled frame whose lists proved too short: rendered again from the full ones


_LOG_FLAGS = os.environ.get("GS_LOG_FLAGS", "") == "1"


class FrameRenderer:
    default_force_strips = False  # see `force_strips`

    def __init__(self, device="cuda", max_pairs: int = 1 << 20, training: bool = False,
                 thresh: float = 0.05, scale_activation: str = "abs", auto_grow: bool = True,
                 sort_mode: int = 2, tile_culling_method: str = "prob2", tile_culling_dist_thresh: float = 0.5,
                 emit_sorted_keys: bool = False, slice_sort: bool = False, table_bin: bool = False,
                 serial_long_lists: bool = False, long_lists: Optional[bool] = None, bwd_rows: Optional[bool] = None,
                 force_strips: Optional[bool] = None, occlusion_cull: Optional[bool] = None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("FrameRenderer needs a HIP device; there is no CPU fallback")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if tile_culling_method not in TILE_CULLING:
            raise ValueError(f"tile_culling_method must be one of {sorted(TILE_CULLING)}")
        # "prob2" (train.py's default), "prob" and "dist" (Splatter.__init__'s default): splatter.py:571-578.
        # For "prob" / "prob2" `thresh` is tile_culling_prob_thresh, for "dist" the radius is
        # tile_geo_length_x / tile_culling_dist_thresh (thresh is unused).
        self.tile_culling_method = TILE_CULLING[tile_culling_method]
        self.tile_culling_dist_thresh = float(tile_culling_dist_thresh)
        self.emit_sorted_keys = bool(emit_sorted_keys)
        self.slice_sort = bool(slice_sort)  # sort_mode 2: the slice-sorted binning variant (GS_FRAME_SLICE_SORT)
        self.table_bin = bool(table_bin)  # sort_mode 2: the table variant whatever the scene size
        # sort_mode 2: the strip variant whatever the scene size (GS_FRAME_STRIP_BIN).  Default (neither flag): the
        # library picks by N -- table variant below 131,072 Gaussians, strip variant from there on.  None = the class-wide
        # default below (the GPU test-suite sets it, so that its small scenes keep exercising the strip kernels).
        self.force_strips = FrameRenderer.default_force_strips if force_strips is None else bool(force_strips)
        self.serial_long_lists = bool(serial_long_lists)  # frames with long lists: no segmented compositing
        # GS_FRAME_LONG_LISTS (big-list sort + segmented compositing of long tile lists): True / False, or None = as soon
        # as an earlier frame of this renderer reported a tile list beyond LONG_LIST_FLAG_AT = 6,144 pairs (the counters that auto_grow reads
        # back anyway carry the longest list).  The workspace capacity plays no part in it.
        # GS_FRAME_OCCLUSION_CULL (include/gs_abi.h): inference frames drop, at emission, the pairs behind the depth at which
        # the PREVIOUS forward of this workspace saw all pixels of their tile stop -- exact (a frame whose trimmed lists
        # prove too short is rendered again from the full ones on the device, inside the same call).  None / True: on for
        # every frame that follows a forward of the same size in the same workspace; False: never.  `stats().pairs` of a
        # culled frame counts the pairs that were emitted.
        self.occlusion_cull = False if (occlusion_cull is None and os.environ.get("GS_NO_CULL", "") == "1") else occlusion_cull  # (GS_NO_CULL=1: A/B runs)
        self._cut_key = None  # (workspace address, width, height) of the inference forward that left the current cut table behind
        # does the cull pay on this scene?  (_cull_probe: asynchronous, tagged counter copies of an unculled and a culled frame)
        self._cull_off_until = 0      # frame serial from which the cull is allowed (again)
        self._cull_backoff = 256      # frames it is switched off for when it did not pay; doubles up to 4,096
        self._cull_full_pairs = None  # pairs of the last UNCULLED inference frame whose counters have arrived
        self._cull_settled = False    # a culled frame's counters have confirmed that the cull pays (until it is disturbed)
        self._cull_probe = None       # (event, pinned host buffer, serial, frame was culled)
        self.long_lists = long_lists
        self._long_lists_seen = False
        self._long_sort_seen = False  # GS_FRAME_LONG_SORT: a list beyond the per-tile sort's LDS window was seen (see _note_lists)
        # rgb training frames: which kernel composites the backward (GS_FRAME_BWD_ROWS, include/gs_abi.h).  None: by the
        # share of saturated buckets the last backward in front of a `stats()` call reported, with hysteresis -- the
        # choice moves at those (synchronising, caller-placed) calls only, never from asynchronously arriving counters,
        # so that a training run takes the same kernels -- the same bits -- every time (gs_train.Trainer asks once per
        # Gaussian set, after its first step); True / False: always / never.
        self.bwd_rows = bwd_rows
        self._bwd_rows_seen = False
        self.max_pairs = int(max_pairs)
        self.training = bool(training)
        self.thresh = float(thresh)
        self.scale_activation = SCALE_ACT[scale_activation]
        # True: check the pair count after every frame (one host synchronisation per frame, never a truncated
        # frame); "async": copy the counters to pinned memory after every frame and look at them when the NEXT
        # frame is issued, without waiting -- no synchronisation in steady state, the workspace grows with 25 %
        # head room as soon as the copy of an overflowed frame has landed (that one frame was rendered empty /
        # truncated); False: never check.
        self.auto_grow = auto_grow
        self._async_host = torch.zeros(_lib.GS_STATS_TAGGED_N, dtype=torch.int64).pin_memory()
        self._async_event: Optional[torch.cuda.Event] = None
        self.headroom = 1.25
        # 0: LSD radix on 64-bit keys, 1: tile-bit radix + per-tile LDS sort, 2: LDS counting sort by tile
        # + per-tile LDS sort (all three give the same list)
        self.sort_mode = int(sort_mode)
        self._ws: Optional[torch.Tensor] = None
        self._stats_host = torch.zeros(_lib.GS_STATS_TAGGED_N, dtype=torch.int64).pin_memory()
        self._frame: Optional[_lib.GsFrame] = None
        self._frame_serial = 0  # counts forwards: autograd checks that backward() belongs to the latest one
        self._cam_cache = {}
        self._keep = None
        # opt-in side stream of the library (include/gs_abi.h, gs_frame_async_*): owned by this renderer
        self._async = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.gs_frame_async_create(C.byref(self._async)), "gs_frame_async_create")
        self._checked_once = False  # auto_grow="async": the first frame is checked synchronously
        self.overflowed_frames = 0  # frames that were rendered empty / truncated and could not be redone (see forward)

    # ------------------------------------------------------------------ frame descriptor
    def _describe(self, pos, quat, scale, opa, rgb, camera, training) -> _lib.GsFrame:
        # A viewer or a benchmark renders the same tensors from the same camera object again and again: the descriptor
        # of the previous call is reused as long as nothing it was built from has changed (tensor identities and
        # storage, camera object, renderer settings) -- the validation below, ~40 ctypes stores and a library call cost
        # ~15 us of host time per frame otherwise, which is what bounds small scenes with several frames in flight.
        # (the camera enters by VALUE: a viewer may move one Camera object in place)
        ck = (int(camera.width), int(camera.height), float(camera.focal_x), float(camera.focal_y),
              float(camera.near), np.asarray(camera.rot, np.float32).tobytes(),
              np.asarray(camera.tran, np.float32).tobytes())
        # what a cache hit must not skip: a different tensor may reuse an address -- a freed fp32 tensor
        # replaced by an fp16 one, a strided view that shares its base's data_ptr.  dtype / contiguity are checked on every
        # call (~1 us), the shapes are part of the key.
        for name, t in (("pos", pos), ("quat", quat), ("scale", scale), ("opa", opa), ("rgb", rgb)):
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError(f"{name} must be a contiguous float32 HIP tensor")
        self._cur_ck = ck
        key = (ck, pos.data_ptr(), quat.data_ptr(), scale.data_ptr(), opa.data_ptr(), rgb.data_ptr(),
               tuple(pos.shape), tuple(quat.shape), tuple(scale.shape), tuple(opa.shape), tuple(rgb.shape),
               pos.shape[0], rgb.shape[-1] if rgb.dim() == 2 else 1, bool(training), self.max_pairs, self.sort_mode,
               self.tile_culling_method, self.tile_culling_dist_thresh, self.thresh, self.scale_activation,
               self.emit_sorted_keys, self.slice_sort, self.table_bin, self.force_strips, self.serial_long_lists,
               self.long_lists, self._long_lists_seen, self._long_sort_seen, self.bwd_rows, self._bwd_rows_seen,
               self.occlusion_cull, self._cut_key, getattr(self, "_cut_ck", None), self._frame_serial >= self._cull_off_until,
               self._ws.data_ptr() if self._ws is not None else 0)
        cached = getattr(self, "_desc_cache", None)
        if cached is not None and cached[0] == key:
            f = _lib.GsFrame()
            C.memmove(C.byref(f), C.byref(cached[1]), C.sizeof(_lib.GsFrame))
            self._grid = cached[3]
            return f
        f = self._describe_uncached(pos, quat, scale, opa, rgb, camera, training)
        # (the key is taken again: building the descriptor may have (re)allocated the workspace)
        key = key[:-1] + (self._ws.data_ptr(),)
        keep = _lib.GsFrame()
        C.memmove(C.byref(keep), C.byref(f), C.sizeof(_lib.GsFrame))
        self._desc_cache = (key, keep, None, self._grid)
        return f

    def _describe_uncached(self, pos, quat, scale, opa, rgb, camera, training) -> _lib.GsFrame:
        n = int(pos.shape[0])
        color_dim = int(rgb.shape[1]) if rgb.dim() == 2 else 1
        if color_dim not in (3, 27, 48):  # rgb logits, SH degree 2 (the reference's), SH degree 3 (extension)
            raise RuntimeError(f"rgb must be [N,3], [N,27] or [N,48], got {tuple(rgb.shape)}")
        for name, t, cols in (("pos", pos, 3), ("quat", quat, 4), ("scale", scale, 3), ("opa", opa, None),
                              ("rgb", rgb, color_dim)):
            if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
                raise RuntimeError(f"{name} must be a contiguous float32 HIP tensor")
            if t.shape[0] != n or (cols is not None and (t.dim() != 2 or t.shape[1] != cols)):
                raise RuntimeError(f"{name} has shape {tuple(t.shape)}, expected [{n},{cols}]")
        # per-camera constants (tile grid, frustum guard band, ray basis) are cached: a viewer or a
        # trainer cycles through a fixed set of cameras, and this host work (a 3x3 inverse, ~40
        # ctypes stores) would otherwise cost more than the launches themselves
        ck = (int(camera.width), int(camera.height), float(camera.focal_x), float(camera.focal_y),
              float(camera.near), np.asarray(camera.rot, np.float32).tobytes(),
              np.asarray(camera.tran, np.float32).tobytes())
        # (the ray basis -- a 3x3 inverse and four small matrix products in NumPy, ~100 us of host time -- only feeds the
        # SH colours: a moving camera over an rgb scene, where a frame is 0.14 - 0.3 ms of GPU time, does not pay for it)
        need_rays = color_dim != 3
        ck = ck + (need_rays,)
        cached = self._cam_cache.get(ck)
        if cached is None:
            grid = TileGrid(int(camera.width), int(camera.height), float(camera.focal_x), float(camera.focal_y))
            half_w, half_h = grid.frustum_half_extents()
            if need_rays:
                rays = RayBasis.from_camera(camera.rot, camera.tran, grid.padded_height, grid.padded_width,
                                            grid.focal_x, grid.focal_y)
            else:
                zero3 = np.zeros(3, np.float32)
                rays = RayBasis(zero3, zero3, zero3, zero3)
            c = _lib.GsFrame()
            c.rot = (C.c_float * 9)(*np.asarray(camera.rot, np.float32).reshape(9))
            c.tran = (C.c_float * 3)(*np.asarray(camera.tran, np.float32).reshape(3))
            c.near_plane, c.half_width, c.half_height = float(camera.near), half_w, half_h
            c.width, c.height = grid.width, grid.height
            c.focal_x, c.focal_y = grid.focal_x, grid.focal_y
            c.tile_length_x, c.tile_length_y = grid.tile_geo_length_x, grid.tile_geo_length_y
            c.leftmost, c.topmost = grid.leftmost, grid.topmost
            c.rays_o = (C.c_float * 3)(*rays.rays_o)
            c.lefttop = (C.c_float * 3)(*rays.lefttop)
            c.vec_dx = (C.c_float * 3)(*rays.dx)
            c.vec_dy = (C.c_float * 3)(*rays.dy)
            if len(self._cam_cache) > 64:
                self._cam_cache.clear()
            cached = self._cam_cache[ck] = (c, grid)
        proto, grid = cached
        f = _lib.GsFrame()
        C.memmove(C.byref(f), C.byref(proto), C.sizeof(_lib.GsFrame))
        f.N, f.color_dim, f.scale_activation = n, color_dim, self.scale_activation
        f.pos, f.quat, f.scale, f.opa, f.rgb = (t.data_ptr() for t in (pos, quat, scale, opa, rgb))
        # "dist": the squared distance threshold of splatter.py:577
        f.thresh = (grid.tile_geo_length_x / self.tile_culling_dist_thresh) ** 2 if self.tile_culling_method == 0 \
            else self.thresh
        f.max_pairs = self.max_pairs
        f.async_ = self._async
        f.flags = (_lib.GS_FRAME_EMIT_SORTED_KEYS if self.emit_sorted_keys else 0) | \
            (_lib.GS_FRAME_SLICE_SORT if self.slice_sort else 0) | (_lib.GS_FRAME_TABLE_BIN if self.table_bin else 0) | \
            (_lib.GS_FRAME_STRIP_BIN if (self.force_strips and not (self.slice_sort or self.table_bin)) else 0) | \
            (_lib.GS_FRAME_SERIAL_LONG_LISTS if self.serial_long_lists else 0) | \
            (_lib.GS_FRAME_LONG_LISTS if (self.long_lists or (self.long_lists is None and self._long_lists_seen)) else 0) | \
            (_lib.GS_FRAME_LONG_SORT if (self.long_lists is None and self._long_sort_seen) else 0) | \
            (_lib.GS_FRAME_BWD_ROWS if (self.bwd_rows or (self.bwd_rows is None and self._bwd_rows_seen)) else 0)
        if _LOG_FLAGS and (f.flags, self.max_pairs) != getattr(self, "_logged_flags", None):
            # GS_LOG_FLAGS=1: every change of the latched flags / the capacity, with the frame it happens at (diagnostic)
            print(f"[gs_frame {id(self) & 0xffff:04x}] frame {self._frame_serial} N={n} training={int(training)} "
                  f"flags={int(f.flags)} max_pairs={self.max_pairs}", file=sys.stderr, flush=True)
            self._logged_flags = (f.flags, self.max_pairs)
        f.training = int(training)
        f.sort_mode = self.sort_mode
        f.tile_culling_method = self.tile_culling_method
        need = _lib.gs_frame_workspace_bytes(n, self.max_pairs, grid.width, grid.height, color_dim, int(training))
        if self._ws is None or self._ws.numel() < need:
            self._release_workspace()  # side-stream work of an earlier training frame may still touch the old one
            self._ws = torch.empty(int(need) + 256, dtype=torch.uint8, device=self.device)
            self._cut_key = None  # a fresh allocation: no cut table
        base = self._ws.data_ptr()
        f.workspace = (base + 255) // 256 * 256
        f.workspace_bytes = self._ws.numel() - (f.workspace - base)
        if getattr(self, "_cull_scene_key", None) != (pos.data_ptr(), n):
            # other tensors / another Gaussian count: what the cull was judged against no longer applies
            self._cull_scene_key, self._cull_full_pairs, self._cull_settled = (pos.data_ptr(), n), None, False
            if self.long_lists is None:
                self._long_lists_seen = False  # (the segment decision holds for one Gaussian set: _note_lists)
        shift = self._camera_shift_px(camera)
        if shift > 0.0:
            self._cull_settled = False  # another pose: whether the cull still pays is looked at again
        if self.occlusion_cull is not False and not training and not self.emit_sorted_keys and \
                self._cut_key == (base, grid.width, grid.height) and self._frame_serial >= self._cull_off_until and \
                shift <= self.CULL_MAX_SHIFT_PX:
            f.flags |= _lib.GS_FRAME_OCCLUSION_CULL  # (the library ignores it where the cull does not apply)
            if shift > 0.0:
                f.flags |= _lib.GS_FRAME_CULL_DILATE  # a pose near the recorded one: every tile's cut from its 3 x 3 neighbourhood
                if shift <= self.CULL_NEAR_SHIFT_PX:
                    f.flags |= _lib.GS_FRAME_CULL_DILATE_NEAR  # ... pushed back by 1.125 instead of 1.375 in depth
        self._grid = grid
        return f

    def _stream(self):
        return torch.cuda.current_stream(self.device)

    def _release_workspace(self):
        """Before the workspace tensor goes back to torch's allocator: let the current stream wait for what the
        library's side stream may still be doing in it (include/gs_abi.h, gs_frame_async_wait)."""
        if self._ws is not None and self._async:
            _lib.check(_lib.gs_frame_async_wait(self._async, self._stream().cuda_stream), "gs_frame_async_wait")
            self._frame = None
        self._cut_key = None

    def __del__(self):
        try:
            self._release_workspace()
            if self._async:
                _lib.gs_frame_async_destroy(self._async)
                self._async = C.c_void_p()
        except Exception:  # interpreter shutdown: the library or torch may already be gone
            pass

    # ------------------------------------------------------------------ low-level API
    def forward(self, pos, quat, scale, opa, rgb, camera, training: Optional[bool] = None):
        """Raw parameters -> (image [H,W,3] clamped+cropped, padded raw image or None).

        Capacity (``max_pairs``) and ``auto_grow``: ``True`` reads the frame's counters after every frame (one host
        synchronisation) and redoes an overflowed frame in a larger workspace.  ``"async"`` does that for the FIRST
        frame and for every inference frame (``training=False``: an evaluation image is never returned truncated);
        training frames after the first copy their counters to pinned memory without waiting, and the next forward
        grows the workspace (25 % head room, and already when a frame comes within 25 % of the capacity) -- a training
        frame that overflowed all the same was rendered empty and is counted in ``overflowed_frames`` once its counters
        arrive; ``overflow_flag()`` is the device address of the frame's overflow counter, which gs_train.Trainer hands
        to the fused Adam so that such a step is skipped on the device.  ``False`` never checks."""
        training = self.training if training is None else training
        with torch.cuda.device(self.device):
            return self._forward(pos, quat, scale, opa, rgb, camera, training)

    # auto_grow="async": how many frames the host may issue beyond the frame whose counters are still on their way.  A
    # Python loop issues a training step in ~0.1 ms and the device takes ~1 ms for it: unbounded, the host runs hundreds of
    # frames ahead, the counters of frame k are looked at while frame k + 500 is being issued, and a scene whose pair count
    # grows (training lowers opacities; views differ) overflows its workspace -- frames rendered empty, steps skipped on the
    # device -- long before the growth that frame k asked for takes effect; with one copy in flight at a time most of those
    # frames are never even looked at.  (Found at the end of round 5: two 7,001-iteration fits that are bit-identical step by
    # step in lockstep ended 0.1 dB apart when run at their own pace, tools/fused_adam_bisect.py.)  Waiting for the copy of
    # frame k before frame k + 8 is issued costs nothing -- seven frames are queued behind it -- and bounds the lag.
    # Long tile lists: what the following frames are flagged with, from the counters every frame carries (longest list, M,
    # and X = the pairs that lie beyond the first 512 of their tile's list).
    #   GS_FRAME_LONG_SORT -- lists beyond the per-tile sort's LDS window go to big_list_sort_kernel -- as soon as a list beyond
    #     LONG_SORT_FLAG_AT = 2,048 (that window) was seen: the sorted list is the same either way, and in a trained /
    #     densified scene (3.9 M pairs, 165 lists beyond 2,048) the per-tile sort drops from 407 to 122 us whatever the
    #     colour model (profiles/r06_a_trained_state_flag_ab.txt);
    #   GS_FRAME_LONG_LISTS -- segmented compositing of every tile beyond 512 entries (two passes) -- by a COST MODEL per
    #     colour model (round 6; VERDICT round 5, item 3b: one threshold for all colour models made SH end states slower),
    #     from what the compositing kernel of an UNFLAGGED frame reports about itself: W = the longest walk a tile's wave made
    #     (not the longest LIST: a tile whose pixels stop early walks a fraction of it), X = the steps beyond the first 512 of
    #     every walk, and M = the frame's pairs:
    #       plain   = max(W x c_wave, M x c_dev)            the longest walk at the pace of a wave that shares its SIMD,
    #                                                       against the whole device's pace on all the work
    #       flagged = (M - X) x c_dev + X x k x c_dev + 50 us   the segment kernels cost k x the plain kernel per step they
    #                                                       take over (pass 1 + pass 2 + combine) plus their launches
    #     (c_dev, c_wave, k) measured on the end states of the densifying runs (tools/soak_end_state.py, profiles/r06_g_*:
    #     SH degree 2: 4.54 M pairs, W = 6,616, X = 1.66 M: plain 2.73 ms, flagged 1.75 ms; rgb: 3.48 M, 4,959, 0.70 M: 0.52
    #     against 0.48 ms) and on the trained-like scene (r06_a): rgb 0.066 ns / 105 ns / 5.0, SH degree 2 0.264 / 413 / 2.2,
    #     degree 3 0.37 / 580 / 2.2.  Flagged when the model says the segments save 10 %; the decision is taken on unflagged
    #     frames only and holds until the Gaussian set changes (a flagged frame's waves stop at 512 and report nothing).
    #     Walks up to LONG_LIST_FLAG_AT = 2,048 steps never flag.
    LONG_SORT_FLAG_AT = int(os.environ.get("GS_FRAME_LONG_SORT_FLAG_AT", "2048"))
    LONG_LIST_FLAG_AT = int(os.environ.get("GS_FRAME_LONG_LIST_FLAG_AT", "2048"))
    LONG_LIST_COST = {3: (0.066, 105.0, 5.0), 27: (0.264, 413.0, 2.2), 48: (0.37, 580.0, 2.2)}  # ns, ns, ratio

    # The occlusion cull is exact for ANY camera -- a frame whose trimmed lists prove too short is rendered again from the
    # full ones --, but that second pass costs 0.6 of a frame, and with 8,160 tiles SOME tile runs past its OWN cut in nearly
    # every frame of a moving camera (a pixel at the rim of an opaque Gaussian's footprint sees through to something twice
    # as deep: even a 0.25-pixel pan, 1,535 instead of 3,300 FPS, profiles/r06_g_*).  So:
    #   identical pose (a viewer at rest, a benchmark or an evaluation that renders one view repeatedly): the tiles' own cuts;
    #   a pose within CULL_MAX_SHIFT_PX of the recorded one (a viewer in motion): GS_FRAME_CULL_DILATE -- every tile's cut is
    #     the deepest of its 3 x 3 neighbourhood x 1.375 in depth.  Measured (tools/cull_moving.py, profiles/r06_x_*): no
    #     fallback over 119 frames of a 1.25-px/frame pan, 3 at 5 px/frame, 4 at 12 px/frame; 3,760 / 3,620 / 3,470 FPS
    #     against 3,090 / 3,060 / 3,270 unculled;
    #   beyond (a jump, another test camera: the reference's evaluation loop walks through DIFFERENT cameras, train.py:240-
    #     266): no cull, and nothing paid for the feature (no gated launches without the flag).
    # A frame that does fall back switches the cull off for a while through the probes below.
    CULL_MAX_SHIFT_PX = float(os.environ.get("GS_FRAME_CULL_MAX_SHIFT_PX", "8.0"))
    # within half a pixel (a viewer in motion at thousands of frames per second: 500 px/s are 0.125 px per frame) the depth factor
    # is 1.125: 2.5 M instead of 3.1 M pairs emitted, no fallback over a 0.25-px/frame pan (2 in 119 frames at 1.25 px/frame)
    CULL_NEAR_SHIFT_PX = float(os.environ.get("GS_FRAME_CULL_NEAR_SHIFT_PX", "0.5"))

    def _camera_shift_px(self, camera) -> float:
        """Upper estimate of how far image content moved, in pixels, between the camera the cut table was recorded under
        and ``camera``: rotation angle x focal length + focal length x camera-centre displacement / 1 (unit depth)."""
        prev, cur = getattr(self, "_cut_ck", None), getattr(self, "_cur_ck", None)
        if prev is None or cur is None or prev[:4] != cur[:4]:  # (width, height, focal lengths)
            return float("inf")
        if prev[5] == cur[5] and prev[6] == cur[6]:  # the same pose, byte for byte (the common case: no arithmetic)
            return 0.0
        r0, t0 = np.frombuffer(prev[5], np.float32).astype(np.float64).reshape(3, 3), np.frombuffer(prev[6], np.float32).astype(np.float64)
        r1, t1 = np.frombuffer(cur[5], np.float32).astype(np.float64).reshape(3, 3), np.frombuffer(cur[6], np.float32).astype(np.float64)
        # (|R1 - R0|_F / sqrt 2 = 2 sin(angle / 2): exact for small angles, where arccos((trace - 1) / 2) returns 0 for
        # anything below 0.03 degree -- cos rounds to 1 in fp32; first version of round 6, profiles/r06_h_cull_flag_trace.txt)
        ang = float(np.linalg.norm(r1 - r0) / np.sqrt(2.0))
        dc = float(np.linalg.norm(r1.T @ t1 - r0.T @ t0))
        return max(float(cur[2]), float(cur[3])) * (ang + dc)

    # Whether the cull PAYS is a property of the scene: on the opaque 2.4 M-Gaussian scene 74 % of the pairs are dropped and
    # the frame gains 5 %; on a scene whose tiles do not saturate (BASELINE configs[1], a trained model) nothing is dropped
    # and the gated launches + the second histogram cost 10 % (profiles/r06_d_*).  So the first unculled and the first culled
    # frame of a run of inference frames copy their counters to pinned memory (tagged, asynchronous: no synchronisation);
    # when they have landed the renderer keeps the cull if it emitted < CULL_MIN_GAIN of the frame's pairs without falling
    # back, else switches it off for `_cull_backoff` frames (256, doubling up to 4,096 while it keeps failing).
    CULL_MIN_GAIN = float(os.environ.get("GS_FRAME_CULL_MIN_GAIN", "0.65"))

    def _cull_probe_step(self, f, stream):
        if torch.cuda.is_current_stream_capturing():
            return  # (a frame being captured into a graph: no event queries, no copies to the host)
        p = self._cull_probe
        if p is not None and p[0].query():
            h = p[1].tolist()
            self._cull_probe = None
            if (int(h[11]) & 0xffffffff) == (p[2] & 0xffffffff):
                pairs, ran_past = int(h[1]), int(h[10])
                if int(h[2]) or pairs <= 0:
                    pass  # an overflowed (empty) frame says nothing about the scene
                elif not p[3]:
                    self._cull_full_pairs, self._cull_full_serial = pairs, p[2]
                elif self._cull_full_pairs:
                    if ran_past or pairs > self.CULL_MIN_GAIN * self._cull_full_pairs:
                        self._cull_off_until = self._frame_serial + self._cull_backoff
                        self._cull_backoff = min(2 * self._cull_backoff, 4096)
                        self._cull_settled = False
                    else:
                        self._cull_settled, self._cull_backoff = True, 256
        culled = bool(f.flags & _lib.GS_FRAME_OCCLUSION_CULL)
        # (judged from the SECOND culled frame of a run on: the first one may have been trimmed by a cut table that another
        # scene or another set of parameters left in the workspace -- all GS_NO_CUT, or cuts that make it fall back)
        self._cull_run = getattr(self, "_cull_run", 0) + 1 if culled else 0
        if not culled:
            self._cull_settled = False  # (camera moved, workspace changed, switched off: the next culled frames are looked at again)
        # what is worth a copy: a culled frame that has not been judged yet, and an unculled one when the frame's full pair
        # count is unknown or older than 64 frames (a moving camera renders unculled frame after unculled frame)
        fresh = self._cull_full_pairs is not None and self._frame_serial - getattr(self, "_cull_full_serial", -10**9) <= 64
        want = (not self._cull_settled and self._cull_full_pairs is not None and self._cull_run >= 2) if culled else not fresh
        if culled and self._cull_full_pairs is None and self._cull_probe is None:
            self._cull_off_until = self._frame_serial + 1  # nothing to compare with yet: one unculled frame, which is probed
        if self._cull_probe is None and want:
            host = getattr(self, "_cull_host", None)
            if host is None:
                host = self._cull_host = torch.zeros(_lib.GS_STATS_TAGGED_N, dtype=torch.int64).pin_memory()
            _lib.check(_lib.gs_frame_stats_tagged_async(C.byref(f), self._frame_serial & 0xffffffff, host.data_ptr(), stream),
                       "gs_frame_stats_tagged_async")
            ev = torch.cuda.Event()
            ev.record(self._stream())
            self._cull_probe = (ev, host, self._frame_serial, culled)

    def _note_cut_table(self, f):
        """Every inference frame's compositing launch leaves the per-tile occlusion cuts of ITS frame in the workspace: the
        next forward of the same size may use them (GS_FRAME_OCCLUSION_CULL).  A training forward does not write the table
        (and may use the workspace differently): no cull right behind one."""
        self._cut_key = None if f.training else (self._ws.data_ptr(), int(f.width), int(f.height))
        self._cut_ck = getattr(self, "_cur_ck", None)  # the camera (by value) the table was recorded under

    def _note_lists(self, longest: int, pairs: int, max_walk: int = 0, excess_walk: int = 0):
        self._long_sort_seen = self._long_sort_seen or longest > self.LONG_SORT_FLAG_AT
        if self._frame is None or (self._frame.flags & _lib.GS_FRAME_LONG_LISTS) or max_walk <= self.LONG_LIST_FLAG_AT:
            return
        c_dev, c_wave, k = self.LONG_LIST_COST.get(int(self._frame.color_dim), self.LONG_LIST_COST[3])
        x = min(max(int(excess_walk), 0), pairs)
        plain = max(max_walk * c_wave, pairs * c_dev)
        flagged = (pairs - x) * c_dev + x * k * c_dev + 50_000.0
        if flagged < 0.9 * plain:
            self._long_lists_seen = True

    ASYNC_COUNTER_LAG = int(os.environ.get("GS_FRAME_COUNTER_LAG", "8"))  # (the variable: A/B measurements)

    def _poll_async_counters(self):
        """Counters of an earlier frame that have landed in pinned memory (no waiting -- unless the host has run
        ``ASYNC_COUNTER_LAG`` frames ahead of them)."""
        if self._async_event is not None and self._frame_serial - self._async_serial >= self.ASYNC_COUNTER_LAG:
            self._async_event.synchronize()
        if self._async_event is not None and self._async_event.query():
            h = self._async_host.tolist()
            v, m, o, b, longest, tag, walk, xwalk = (int(h[k]) for k in (0, 1, 2, 3, 9, 11, 13, 14))
            self._async_event = None
            if (tag & 0xffffffff) != (self._async_serial & 0xffffffff):
                return  # (cannot happen in stream order; counters without their frame's tag are not acted upon)
            self._note_lists(longest, m, walk, xwalk)
            # (NOT the backward-kernel choice: counters that arrive asynchronously would make it -- and with it the
            # gradients' last bits -- depend on host timing; it moves at synchronous stats() calls only)
            if o:
                self.overflowed_frames += 1
                self._last_overflow_serial = self._async_serial
                self.max_pairs = max(self.max_pairs, int(o * self.headroom) + 1024)
            elif m * self.headroom > self.max_pairs:  # close to the limit: grow before it overflows
                self.max_pairs = int(m * self.headroom * self.headroom) + 1024

    def _note_buckets(self, b: int):
        """`b`: the `buckets` counter of a backward's preparation -- low half: buckets in the work list, high half: those of
        saturated tiles.  The row-layout rgb backward pays when most buckets are of that kind (it leaves dead pixel rows
        out) and costs ~10 % when none is: switch on above 60 %, off below 40 %."""
        total, sat = b & 0xffffffff, b >> 32
        if total > 0:
            share = sat / total
            if share > 0.6:
                self._bwd_rows_seen = True
            elif share < 0.4:
                self._bwd_rows_seen = False

    def last_frame_overflowed(self, wait: bool = False) -> bool:
        """auto_grow="async": did the most recent forward overflow its workspace?  Without ``wait`` only what has
        already reached the host is looked at (False if the counters are still in flight)."""
        if self._async_event is not None and wait:
            self._async_event.synchronize()
        self._poll_async_counters()
        return getattr(self, "_last_overflow_serial", -1) == self._frame_serial

    def _forward(self, pos, quat, scale, opa, rgb, camera, training):
        self._begun = None  # any frame opened by forward_begin lived in the workspace this frame is about to overwrite
        stream = self._stream().cuda_stream
        sync_check = self.auto_grow is True or (self.auto_grow == "async" and (not training or not self._checked_once))
        if self.auto_grow == "async":
            self._poll_async_counters()
        while True:
            f = self._describe(pos, quat, scale, opa, rgb, camera, training)
            g = self._grid
            image = torch.empty(g.height, g.width, 3, device=self.device, dtype=torch.float32)
            padded = torch.empty(g.padded_height, g.padded_width, 3, device=self.device,
                                 dtype=torch.float32) if training else None
            f.image = image.data_ptr()
            f.image_padded = padded.data_ptr() if padded is not None else None
            _lib.check(_lib.gs_frame_forward(C.byref(f), stream), "gs_frame_forward")
            self._frame = f
            self._frame_serial += 1
            self._note_cut_table(f)
            self._keep = (pos, quat, scale, opa, rgb, image, padded)
            if not training and self.occlusion_cull is not False and not self.emit_sorted_keys:
                self._cull_probe_step(f, stream)
            if sync_check:
                st = self.stats()
                if st.overflow:
                    self.max_pairs = int(st.overflow * self.headroom) + 1024  # grow and redo the frame
                    continue
                self._checked_once = True
                if self.auto_grow == "async" and st.pairs * self.headroom > self.max_pairs:
                    self.max_pairs = int(st.pairs * self.headroom * self.headroom) + 1024  # takes effect next frame
                break
            if self.auto_grow == "async" and self._async_event is None:  # one copy in flight at a time
                # tagged with the frame's serial number (include/gs_abi.h: the counters say which frame they belong to)
                _lib.check(_lib.gs_frame_stats_tagged_async(C.byref(f), self._frame_serial & 0xffffffff,
                                                            self._async_host.data_ptr(), stream),
                           "gs_frame_stats_tagged_async")
                self._async_event = torch.cuda.Event()
                self._async_event.record(self._stream())
                self._async_serial = self._frame_serial
            break
        return image, padded

    # ------------------------------------------------------------------ a frame in two phases (view-parallel trainer)
    def forward_begin(self, pos, quat, scale, opa, rgb, camera, slice_begin: int, slice_end: int,
                      expect_per_slice: int = None) -> bool:
        """Issue the PROJECT stage of the next training frame for the slices [slice_begin, slice_end) of the Gaussian
        array (``project_slices()`` of them, ``project_slice_size`` Gaussians each); the range that starts at slice 0
        opens the frame.  gs_train.Trainer calls this slice by slice behind the optimizer of the previous step, so that
        the next frame's cull + project + count runs underneath the gradient exchange of the remaining slices.
        Returns False -- nothing issued -- when this frame cannot be split (binning variant without the fused count, a
        pending workspace growth, a capacity check due): the caller then renders it with ``forward``.  The frame is
        completed by ``forward_finish``; no capacity check happens in between (steady-state training frames only).
        ``expect_per_slice``: the slice size the caller converted its Gaussian ranges with (gs_dp.project_slice_size
        restates the library's plan in Python); if the library's plan for this frame says otherwise, nothing is issued
        and the frame is rendered from scratch (the duplicated formula must not be able to project the
        wrong Gaussians silently)."""
        with torch.cuda.device(self.device):
            stream = self._stream().cuda_stream
            if slice_begin == 0:
                self._begun = None
                cur = self._frame
                if (cur is None or not cur.training or self.auto_grow is True or not self._checked_once
                        or cur.max_pairs != self.max_pairs or cur.N != pos.shape[0]):
                    return False
                f = self._describe(pos, quat, scale, opa, rgb, camera, True)
                if f.workspace != cur.workspace:
                    return False
                n_sl, per = C.c_int32(), C.c_int64()
                g = self._grid
                image = torch.empty(g.height, g.width, 3, device=self.device, dtype=torch.float32)
                padded = torch.empty(g.padded_height, g.padded_width, 3, device=self.device, dtype=torch.float32)
                f.image, f.image_padded = image.data_ptr(), padded.data_ptr()
                _lib.check(_lib.gs_frame_project_slices(C.byref(f), C.byref(n_sl), C.byref(per)), "gs_frame_project_slices")
                if n_sl.value == 0:
                    return False
                if expect_per_slice is not None and int(expect_per_slice) != per.value:
                    import warnings

                    warnings.warn(f"forward_begin: the caller cut the Gaussian array into project slices of "
                                  f"{expect_per_slice}, the library's plan says {per.value}; rendering from scratch")
                    return False
                self._begun = {"f": f, "image": image, "padded": padded, "keep": (pos, quat, scale, opa, rgb),
                               "camera": camera, "slices": n_sl.value, "per_slice": per.value, "done": 0}
            b = getattr(self, "_begun", None)
            if b is None:
                return False
            if expect_per_slice is not None and int(expect_per_slice) != b["per_slice"]:
                self._begun = None
                return False
            slice_end = min(int(slice_end), b["slices"])
            if slice_end <= slice_begin:
                return True
            _lib.check(_lib.gs_frame_forward_project(C.byref(b["f"]), int(slice_begin), slice_end, stream),
                       "gs_frame_forward_project")
            b["done"] += slice_end - slice_begin
            return True

    def begun_frame_matches(self, pos, quat, scale, opa, rgb, camera) -> bool:
        """Is a completely projected frame of exactly these tensors and this camera waiting for ``forward_finish``?"""
        b = getattr(self, "_begun", None)
        return (b is not None and b["done"] == b["slices"] and b["camera"] is camera
                and all(x is y for x, y in zip(b["keep"], (pos, quat, scale, opa, rgb))))

    def forward_abandon(self):
        """Drop a frame opened by ``forward_begin`` (its project stage only wrote per-frame scratch)."""
        self._begun = None

    def forward_finish(self):
        """Binning, per-tile sort and compositing of the frame opened by ``forward_begin`` -> (image, padded)."""
        b = getattr(self, "_begun", None)
        if b is None or b["done"] != b["slices"]:
            raise RuntimeError("forward_finish() needs a frame whose project stage was issued completely")
        self._begun = None
        with torch.cuda.device(self.device):
            stream = self._stream().cuda_stream
            if self.auto_grow == "async":
                self._poll_async_counters()
            f = b["f"]
            # the backward-kernel choice is a property of the BACKWARD: taken where the frame is completed, not where its
            # project stage was issued (a choice that moved in between must not depend on whether the frame was begun ahead)
            rows = self.bwd_rows or (self.bwd_rows is None and self._bwd_rows_seen)
            f.flags = (f.flags & ~_lib.GS_FRAME_BWD_ROWS) | (_lib.GS_FRAME_BWD_ROWS if rows else 0)
            _lib.check(_lib.gs_frame_forward_rest(C.byref(f), stream), "gs_frame_forward_rest")
            self._frame = f
            self._frame_serial += 1
            self._keep = (*b["keep"], b["image"], b["padded"])
            self._note_cut_table(f)
            if self.auto_grow == "async" and self._async_event is None:  # one copy in flight at a time
                _lib.check(_lib.gs_frame_stats_tagged_async(C.byref(f), self._frame_serial & 0xffffffff,
                                                            self._async_host.data_ptr(), stream),
                           "gs_frame_stats_tagged_async")
                self._async_event = torch.cuda.Event()
                self._async_event.record(self._stream())
                self._async_serial = self._frame_serial
        return b["image"], b["padded"]

    def backward_slice(self, out, g_begin: int, g_end: int, part: int = None):
        """After ``backward(part=GS_BWD_RASTER)``: the per-Gaussian sums (projection + activation backward) of the
        Gaussians [g_begin, g_end) -- g_begin a multiple of 256 -- into ``out``; ``part`` = GS_BWD_GEOMETRY, GS_BWD_COLOR
        or (default) both in one kernel.  Any partition of the Gaussians gives exactly what ``backward`` writes."""
        f = self._frame
        if f is None or not f.training:
            raise RuntimeError("backward_slice() needs a preceding forward(training=True)")
        if part is None:
            part = _lib.GS_BWD_GEOMETRY | _lib.GS_BWD_COLOR
        with torch.cuda.device(self.device):
            _lib.check(_lib.gs_frame_backward_slice(C.byref(f), *(t.data_ptr() for t in out), int(part), int(g_begin),
                                                    int(g_end), self._stream().cuda_stream), "gs_frame_backward_slice")
        return out

    def backward_adam(self, grad_image, adam):
        """``backward`` with the optimizer step fused into its last kernel (gs_frame_backward_adam, include/
        gs_abi.h): the frame's own parameter tensors are updated in place, no gradient is written.  ``adam``: a filled
        ``gaussian._lib.GsAdamFused`` (gs_train.FusedAdam.fused_descriptor)."""
        f = self._frame
        if f is None or not f.training:
            raise RuntimeError("backward_adam() needs a preceding forward(training=True)")
        # the step is applied to whatever the frame's pointers address: they must be the caller's LIVE parameter tensors
        # (a forward handed converted / non-contiguous tensors would step a temporary copy silently)
        for ptr, t in zip((f.pos, f.quat, f.scale, f.opa, f.rgb), self._keep[:5]):
            if int(ptr or 0) != t.data_ptr():
                raise RuntimeError("backward_adam(): the frame was rendered from tensors that are no longer alive")
        with torch.cuda.device(self.device):
            _lib.check(_lib.gs_frame_backward_adam(C.byref(f), grad_image.contiguous().data_ptr(), C.byref(adam),
                                                   self._stream().cuda_stream), "gs_frame_backward_adam")
        self._bwd_serial = self._frame_serial

    def backward(self, grad_image, out=None, part: int = 0):
        """dL/d(image) -> (grad_pos, grad_quat, grad_scale, grad_opa, grad_rgb).  ``out`` may
        supply the five destination tensors (e.g. views of one flat all-reduce bucket).

        ``part`` (view-parallel gradient exchange, gs_dp.py): 0 = everything; ``_lib.GS_BWD_RASTER`` = only the
        raster backward (per-pair rows), then ``GS_BWD_GEOMETRY`` (pos / quat / scale) and ``GS_BWD_COLOR`` (opa /
        rgb) fill their share of ``out`` in either order, bit-identical to the one-call backward; ``grad_image``
        is only read by part 0 / GS_BWD_RASTER."""
        f = self._frame
        if f is None or not f.training:
            raise RuntimeError("backward() needs a preceding forward(training=True)")
        pos, quat, scale, opa, rgb = self._keep[:5]
        if out is None:
            out = tuple(torch.empty_like(t) for t in (pos, quat, scale, opa, rgb))
        if part in (0, _lib.GS_BWD_RASTER):
            grad_image = grad_image.contiguous()
            if grad_image.dtype != torch.float32 or tuple(grad_image.shape) != (f.height, f.width, 3):
                raise RuntimeError("grad_image must be float32 [H,W,3]")
        for t, ref in zip(out, (pos, quat, scale, opa, rgb)):
            if t.shape != ref.shape or t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError("gradient destinations must match the parameters")
        with torch.cuda.device(self.device):
            if part == 0:
                _lib.check(_lib.gs_frame_backward(C.byref(f), grad_image.data_ptr(), *(t.data_ptr() for t in out),
                                                  self._stream().cuda_stream), "gs_frame_backward")
            else:
                _lib.check(_lib.gs_frame_backward_part(C.byref(f), grad_image.data_ptr() if grad_image is not None
                                                       else None, *(t.data_ptr() for t in out), int(part),
                                                       self._stream().cuda_stream), "gs_frame_backward_part")
        if part in (0, _lib.GS_BWD_RASTER):
            self._bwd_serial = self._frame_serial  # this frame's bucket counter is final once the stream gets here
        return out

    def profile_forward(self, pos, quat, scale, opa, rgb, camera, training: Optional[bool] = None):
        """One forward frame with every stage bracketed by hipEvents (synchronises).  Returns
        {stage: ms}; the raster stage is exactly one kernel launch."""
        training = self.training if training is None else training
        self._begun = None
        f = self._describe(pos, quat, scale, opa, rgb, camera, training)
        g = self._grid
        image = torch.empty(g.height, g.width, 3, device=self.device, dtype=torch.float32)
        padded = torch.empty(g.padded_height, g.padded_width, 3, device=self.device,
                             dtype=torch.float32) if training else None
        f.image = image.data_ptr()
        f.image_padded = padded.data_ptr() if padded is not None else None
        ms = (C.c_float * 6)()
        with torch.cuda.device(self.device):
            _lib.check(_lib.gs_frame_forward_profile(C.byref(f), ms, self._stream().cuda_stream),
                       "gs_frame_forward_profile")
        self._frame = f
        self._frame_serial += 1
        self._note_cut_table(f)
        self._keep = (pos, quat, scale, opa, rgb, image, padded)
        return dict(zip(("project", "scan_emit", "sort", "ranges", "raster", "total"), (float(x) for x in ms)))

    def profile_backward(self, grad_image):
        f = self._frame
        if f is None or not f.training:
            raise RuntimeError("profile_backward() needs a preceding training forward")
        pos, quat, scale, opa, rgb = self._keep[:5]
        out = tuple(torch.empty_like(t) for t in (pos, quat, scale, opa, rgb))
        ms = (C.c_float * 3)()
        with torch.cuda.device(self.device):
            _lib.check(_lib.gs_frame_backward_profile(C.byref(f), grad_image.contiguous().data_ptr(),
                                                      *(t.data_ptr() for t in out), ms, self._stream().cuda_stream),
                       "gs_frame_backward_profile")
        self._bwd_serial = self._frame_serial
        return dict(zip(("raster_bwd", "project_bwd", "total"), (float(x) for x in ms)))

    def stats(self) -> FrameStats:
        """Synchronises the current stream (one 32-byte D2H copy)."""
        if self._frame is None:
            raise RuntimeError("no frame rendered yet")
        stream = self._stream()
        tag = self._frame_serial & 0xffffffff
        _lib.check(_lib.gs_frame_stats_tagged_async(C.byref(self._frame), tag, self._stats_host.data_ptr(),
                                                    stream.cuda_stream), "gs_frame_stats_tagged_async")
        stream.synchronize()
        h = self._stats_host.tolist()
        v, m, o, b, longest, ran_past, walk, xwalk = (int(h[k]) for k in (0, 1, 2, 3, 9, 10, 13, 14))
        assert (int(h[11]) & 0xffffffff) == tag, "gs_frame_stats_tagged_async: the tag did not come back"
        self._note_lists(longest, m, walk, xwalk)
        # The bucket counter is written by the backward's preparation on the library's SIDE stream; the copy above is
        # ordered behind it only once a backward of this frame has been issued on this stream (it waits for the side
        # stream).  A stats() call between forward and backward may read 0, the previous frame's count or this one's:
        # it reports what it read, but the backward-kernel choice -- and with it the gradients' last bits -- moves only
        # on a count that is known to be this frame's.
        # (rgb frames only: the SH backward builds its own work list and leaves this counter alone -- whatever the
        # workspace memory held)
        bwd_done = getattr(self, "_bwd_serial", -1) == self._frame_serial and int(self._frame.color_dim) == 3
        if bwd_done:
            self._note_buckets(b)
        culled = bool(self._frame.flags & _lib.GS_FRAME_OCCLUSION_CULL)
        return FrameStats(v, m, o, (b & 0xffffffff) if bwd_done else 0, longest, (b >> 32) if bwd_done else 0,
                          bool(ran_past) and culled)

    def binning_variant(self) -> str:
        """Which binning / sort path the last frame took: "radix64", "radix_tile_bits", "table", "slice", "strip"."""
        if self._frame is None:
            raise RuntimeError("no frame rendered yet")
        v = _lib.gs_frame_binning_variant(C.byref(self._frame))
        if v < 0:
            raise RuntimeError(f"gs_frame_binning_variant failed (code {v})")
        return ("radix64", "radix_tile_bits", "table", "slice", "strip")[v]

    def overflow_flag(self) -> Optional[int]:
        """Device address of the last frame's 64-bit overflow counter (0 = the frame fitted), for consumers that must
        not act on an overflowed -- i.e. empty -- frame without asking the host (gs_adam_step_sharded)."""
        if self._frame is None:
            return None
        ptr = C.c_void_p()
        _lib.check(_lib.gs_frame_overflow_flag(C.byref(self._frame), C.byref(ptr)), "gs_frame_overflow_flag")
        return ptr.value

    def _rects(self, allow_culled: bool = False) -> torch.Tensor:
        """[N,4] int32 view of the workspace: (y0 | y1 << 16, x0 | x1 << 16, depth bits, tiles touched) per Gaussian.

        An occlusion-culled inference frame writes the records of the Gaussians it PROJECTED only (frustum-culled and
        occluded ones leave whatever an earlier frame wrote): ``debug_views`` refuses such a frame (render with
        ``occlusion_cull=False`` where it is wanted), ``culling_mask`` recomputes the frustum test."""
        f = self._frame
        if f is None:
            raise RuntimeError("no frame rendered yet")
        culled = C.c_int32(0)
        _lib.check(_lib.gs_frame_is_occlusion_culled(C.byref(f), C.byref(culled)), "gs_frame_is_occlusion_culled")
        if culled.value and not allow_culled:
            raise RuntimeError("the rectangle records of an occlusion-culled frame cover its projected Gaussians only: "
                               "render with FrameRenderer(occlusion_cull=False) for debug_views()")
        ptr = C.c_void_p()
        _lib.check(_lib.gs_frame_debug_rects(C.byref(f), C.byref(ptr)), "gs_frame_debug_rects")
        off = ptr.value - self._ws.data_ptr()
        return self._ws[off:off + 16 * f.N].view(torch.int32).reshape(f.N, 4)

    def culling_mask(self) -> torch.Tensor:
        """[N] bool: the Gaussians of the last forward that passed the frustum test (the reference's
        ``culling_mask``, renderer.py:123-132).  Computed from the workspace; no host synchronisation.

        An occlusion-culled inference frame wrote the records of its projected Gaussians only: for such a frame the
        reference's own ``global_culling`` operator (the same frustum test) is run on the frame's positions and pose."""
        f = self._frame
        if f is None:
            raise RuntimeError("no frame rendered yet")
        culled = C.c_int32(0)
        _lib.check(_lib.gs_frame_is_occlusion_culled(C.byref(f), C.byref(culled)), "gs_frame_is_occlusion_culled")
        if culled.value:
            import gaussian  # (the reference-API operators: gaussian/__init__.py)
            pos, quat, scale = self._keep[:3]
            n = pos.shape[0]
            rot = torch.tensor(list(f.rot), dtype=torch.float32, device=self.device).reshape(3, 3)
            tran = torch.tensor(list(f.tran), dtype=torch.float32, device=self.device)
            mask = torch.zeros(n, dtype=torch.int64, device=self.device)
            with torch.cuda.device(self.device):
                gaussian.global_culling(pos, quat, scale, rot, tran, torch.empty_like(pos),
                                        torch.empty((n, 2, 2), dtype=torch.float32, device=self.device), mask,
                                        float(f.near_plane), float(f.half_width), float(f.half_height))
            return mask != 0
        return self._rects()[:, 2] != 0  # depth bits: |p_c| > near > 0 for visible Gaussians, 0 for culled ones

    def composited_steps(self) -> int:
        """Sum over tiles of the Gaussians the last TRAINING forward composited before every pixel of the tile had
        stopped: the work the compositing kernels really did (the pair count is its upper bound).  Synchronises."""
        f = self._frame
        if f is None or not f.training:
            raise RuntimeError("composited_steps() needs a preceding training forward")
        ptr = C.c_void_p()
        _lib.check(_lib.gs_frame_debug_tile_nproc(C.byref(f), C.byref(ptr)), "gs_frame_debug_tile_nproc")
        off = ptr.value - self._ws.data_ptr()
        T = self._grid.n_tiles
        return int(self._ws[off:off + 4 * T].view(torch.int32).to(torch.int64).sum().item())

    def executed_row_steps(self) -> int:
        """SH frames: pixel-row steps (16 Gaussians x 16 pixels) the last backward's matrix-pipe kernel executed -- rows
        whose 16 pixels had all stopped are left out --, i.e. what its MFMA flops are counted from; 0 for frames that
        take another kernel (rgb colours, long-list tails).  Synchronises."""
        f = self._frame
        if f is None or not f.training:
            raise RuntimeError("executed_row_steps() needs a preceding training forward + backward")
        ptr, n = C.c_void_p(), C.c_int32()
        _lib.check(_lib.gs_frame_debug_bwd_exec_rows(C.byref(f), C.byref(ptr), C.byref(n)), "gs_frame_debug_bwd_exec_rows")
        if n.value == 0:
            return 0
        off = ptr.value - self._ws.data_ptr()
        return int(self._ws[off:off + 4 * n.value].view(torch.int32).to(torch.int64).sum().item())

    def debug_views(self):
        """Device tensors aliasing the workspace of the last forward (parity tests)."""
        f = self._frame
        ptrs = [C.c_void_p() for _ in range(7)]
        _lib.check(_lib.gs_frame_debug_views(C.byref(f), *[C.byref(p) for p in ptrs]), "gs_frame_debug_views")
        st = self.stats()
        n, m, T = f.N, st.pairs, self._grid.n_tiles

        def view(ptr, nbytes, dtype, shape):
            off = ptr.value - self._ws.data_ptr()
            return self._ws[off:off + nbytes].view(dtype).reshape(shape)

        ids = view(ptrs[1], 4 * m, torch.int32, (m,))
        ranges = view(ptrs[2], 8 * T, torch.int32, (T, 2))
        if ptrs[0].value:
            keys = view(ptrs[0], 8 * m, torch.int64, (m,))
        else:
            # sort_mode 2 without GS_FRAME_EMIT_SORTED_KEYS: the sorted keys are not materialised (the raster kernels
            # read the ids only).  They are (tile << 32 | depth bits) of the pairs in list order: tile from the
            # ranges, depth bits from the record of the sorted id.
            counts = (ranges[:, 1] - ranges[:, 0]).to(torch.int64)
            tiles = torch.repeat_interleave(torch.arange(T, device=self.device, dtype=torch.int64), counts)
            depth = view(ptrs[3], 64 * n, torch.int32, (n, 16))[:, 2].to(torch.int64) & 0xffffffff
            keys = (tiles << 32) | depth[ids.to(torch.int64)]
        # one 64-byte record per Gaussian: geom | cov | color | conic -- written for VISIBLE Gaussians only; here the
        # records of culled ones are shown as zeros (a copy: this is a test aid)
        visible = self.culling_mask()
        raw = view(ptrs[3], 64 * n, torch.float32, (n, 16))
        rec = torch.where(visible.unsqueeze(1), raw, torch.zeros_like(raw))
        out = {
            "sorted_keys": keys,
            "sorted_ids": ids,
            "tile_ranges": ranges,
            "visible": visible,
            "rec_geom": rec[:, 0:4],
            "rec_cov": rec[:, 4:8],
            "tiles_touched": self._rects()[:, 3],  # (the separate array is only written for sort_modes 0 / 1)
        }
        if f.color_dim == 3:
            out["rec_color"] = rec[:, 8:12]
        return out

    # ------------------------------------------------------------------ autograd entry point
    def render(self, pos, quat, scale, opa, rgb, camera):
        """Differentiable frame: the drop-in for ``Splatter.forward`` on explicit tensors."""
        if torch.is_grad_enabled() and any(t.requires_grad for t in (pos, quat, scale, opa, rgb)):
            return _FrameFunction.apply(pos, quat, scale, opa, rgb, self, camera)
        return self.forward(pos, quat, scale, opa, rgb, camera, training=False)[0]


class _FrameFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, quat, scale, opa, rgb, renderer, camera):
        image, _ = renderer.forward(pos.detach(), quat.detach(), scale.detach(), opa.detach(), rgb.detach(), camera,
                                    training=True)
        ctx.renderer = renderer
        ctx.frame_serial = renderer._frame_serial
        return image

    @staticmethod
    def backward(ctx, grad_image):
        r = ctx.renderer
        if r._frame_serial != ctx.frame_serial:
            raise RuntimeError("FrameRenderer workspace was reused by another forward before backward()")
        g = r.backward(grad_image)
        return (*g, None, None)
