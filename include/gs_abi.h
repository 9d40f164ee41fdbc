/*
 * gs_abi.h -- C ABI of libgs_amd.so, the MI355X (gfx950) rasterizer library.
 *
 * This is the drop-in boundary for the hot path of WangFeng18/3d-gaussian-splatting: the
 * entry points in section A are exactly what the reference's pybind11 module `gaussian`
 * (src/bindings.cpp:21-50) binds, with torch::Tensor arguments replaced by raw device
 * pointers + explicit sizes and an explicit HIP stream.  Section B is the fused,
 * MI355X-native frame path (cull+project -> binning by tile -> per-tile depth sort ->
 * raster fwd/bwd -> project bwd) that the reference spreads over 4 native launches and ~25
 * torch kernels per frame (splatter.py:513-641).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in `_host`;
 *   - all arrays are contiguous, row-major, fp32 unless stated;
 *   - the caller owns every buffer; the library never allocates device memory and keeps no
 *     pointer past the call; it is re-entrant across host threads (forward is called from
 *     the main thread, backward from PyTorch's autograd thread);
 *   - every function returns 0 on success, a negative GS_E_* validation code, or a positive
 *     hipError_t; gs_last_error() returns a thread-local message for the last failure;
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream, which is
 *     what the reference launches on).
 */
#ifndef GS_ABI_H
#define GS_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden; only this header's symbols are exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* 8 (round 6): gs_frame_is_occlusion_culled; an occlusion-culled frame's first pass projects only the Gaussians that are not
 * behind every cut they can reach and writes rectangle records for those only (workspace layout: + the survivor list).
 *   GS_FRAME_CULL_DILATE.
 * 7 (round 6): GS_FRAME_LONG_SORT, GS_FRAME_OCCLUSION_CULL + gs_frame_cull_fallback_async; gs_frame_stats_serial (the frame the
 * counters belong to).
 * 6 (round 5): gs_frame_backward_adam (the backward with the optimizer step fused in); gs_frame_debug_bwd_exec_rows; GS_FRAME_BWD_ROWS and the saturated-bucket count in the upper half of the
 * `buckets` counter; SH gradient rows in whole 64-byte lines without per-row flags (workspace
 * layout only: no signature changed).
 * 5 (round 4): the frame in pieces -- gs_frame_forward_project + gs_frame_forward_rest == gs_frame_forward,
 * gs_frame_backward_slice (the per-Gaussian sums of a range of Gaussians), gs_frame_project_slices --, gs_adam_step_multi /
 * gs_adam_step_range (up to eight element ranges per launch, grad_scale), gs_grad_stat_update, gs_frame_longest_list_async.
 * 4 (round 3): GS_FRAME_STRIP_BIN + the size-based choice of the binning variant, gs_frame_binning_variant,
 * gs_frame_debug_rects (records of culled Gaussians are no longer written), gs_frame_overflow_flag,
 * gs_adam_step_sharded, `fast = 0` of gs_draw / gs_draw_backward honoured, the long-list kernels follow GS_FRAME_LONG_LISTS alone (not the
 * workspace capacity).  3: gs_frame.async / flags. */
#define GS_ABI_VERSION 8

#define GS_E_INVALID (-1)   /* bad argument (null pointer, negative size, bad enum)   */
#define GS_E_UNSUPPORTED (-2) /* valid in the reference but not implemented here (none at present) */
#define GS_E_CAPACITY (-3)  /* workspace too small for this frame                     */

typedef void *gs_stream_t;

const char *gs_last_error(void);
int gs_abi_version(void);

/* =========================================================================================
 * A. The reference `gaussian` module surface (src/bindings.cpp:21-50)
 * ======================================================================================= */

/* bindings.cpp:23 `culling` -> gaussian.cu:6-8: a stub that prints "hellow". Kept as a no-op. */
int gs_culling(void);

/* bindings.cpp:24 `world2camera(pos, rot, trans, res)` -> gaussian.cu:49-76.
 * res[i] = rot * pos[i] + tran.   pos,res: [B,3]; rot: [3,3]; tran: [3]. */
int gs_world2camera(const float *pos, const float *rot, const float *tran, float *res, int64_t B,
                    gs_stream_t stream);

/* bindings.cpp:25 `world2camera_backward(grad_out, rot, grad_inp)` -> gaussian.cu:78-99. */
int gs_world2camera_backward(const float *grad_out, const float *rot, float *grad_inp, int64_t B,
                             gs_stream_t stream);

/* bindings.cpp:26 `jacobian(pos_camera_space, jacobian)` -> gaussian.cu:10-47.  jac: [B,3,3]. */
int gs_jacobian(const float *pos_cam, float *jac, int64_t B, gs_stream_t stream);

/* bindings.cpp:48 `global_culling` -> gaussian.cu:1182-1369.
 * pos [N,3], quat [N,4] (pre-normalised), scale [N,3] (pre-activated), rot [3,3], tran [3]
 * -> res_pos [N,3] = (x/z, y/z, |p_c|), res_cov [N,2,2], culling_mask [N] int64.
 * Rows of culled Gaussians are left untouched (the caller pre-zeroes, renderer.py:124-126). */
int gs_global_culling(const float *pos, const float *quat, const float *scale, const float *rot,
                      const float *tran, int64_t N, float near_plane, float half_width,
                      float half_height, float *res_pos, float *res_cov, int64_t *culling_mask,
                      gs_stream_t stream);

/* bindings.cpp:49 `global_culling_backward` -> gaussian.cu:1371-1609. */
int gs_global_culling_backward(const float *pos, const float *quat, const float *scale,
                               const float *rot, const float *tran, int64_t N,
                               const float *gradout_pos, const float *gradout_cov,
                               const int64_t *culling_mask, float *gradinput_pos,
                               float *gradinput_quat, float *gradinput_scale, gs_stream_t stream);

/* bindings.cpp:44 `calc_tile_list(Gaussian3ds&, Tiles&, tile_n_point, tile_gaussian_list,
 * thresh, method, tile_length_x, tile_length_y, n_tiles_x, n_tiles_y, leftmost, topmost)`
 * -> gaussian.cu:101-335.  The two structs are flattened: Gaussian3ds contributes pos [V,3]
 * and cov [V,2,2]; Tiles contributes top/bottom/left/right [T] (read by methods 0 and 1 only,
 * may be NULL for method 2).  tile_n_point [T] int32 must be zeroed by the caller;
 * tile_gaussian_list is [T, max_points_per_tile] int32.  Race-free: a tile keeps the first
 * `max_points_per_tile` arrivals, the counter keeps counting (the caller clamps it,
 * splatter.py:586). */
int gs_calc_tile_list(const float *pos, const float *cov, int64_t n_point, const float *tile_top,
                      const float *tile_bottom, const float *tile_left, const float *tile_right,
                      int32_t *tile_n_point, int32_t *tile_gaussian_list,
                      int64_t max_points_per_tile, float thresh, int method, float tile_length_x,
                      float tile_length_y, int32_t n_tiles_x, int32_t n_tiles_y, float leftmost,
                      float topmost, gs_stream_t stream);

/* bindings.cpp:45 `gather_gaussians(tile_n_point_accum, tile_gaussian_list, gathered_list,
 * tile_ids_for_points, max_points_for_tile)` -> gaussian.cu:337-381. */
int gs_gather_gaussians(const int32_t *tile_n_point_accum, const int32_t *tile_gaussian_list,
                        int32_t *gathered_list, int32_t *tile_ids_for_points, int64_t n_tiles,
                        int64_t max_points_for_tile, int64_t list_row_size, gs_stream_t stream);

/* bindings.cpp:46 `draw` -> gaussian.cu:806-1043.
 * pos [M,3] (z ignored), rgb [M,D] (D = 3, or 27 when use_sh_coeff), opa [M], cov [M,2,2],
 * tile_n_point_accum [T+1] int32, res [h,w,3] with h,w the PADDED size (multiples of 16).
 * rays_o/lefttop_pos/vec_dx/vec_dy: device float[3], only read when use_sh_coeff.
 * fast != 0: the Gaussian's value through the hardware exponential on a conic hoisted out of the pixel loop (the
 * reference's __expf flavour, gaussian.cu:920); fast == 0: the reference's exp() flavour (gaussian.cu:922-923): the
 * numerator as float products in its order, a double division by 2 det + 1e-14 and a double exp per pixel. */
int gs_draw(const float *pos, const float *rgb, const float *opa, const float *cov,
            const int32_t *tile_n_point_accum, float *res, int32_t h, int32_t w, int64_t M,
            float focal_x, float focal_y, int weight_normalize, int sigmoid, int fast,
            const float *rays_o, const float *lefttop_pos, const float *vec_dx,
            const float *vec_dy, int use_sh_coeff, gs_stream_t stream);

/* bindings.cpp:47 `draw_backward` -> gaussian.cu:440-803, 1045-1129.
 * grad_pos [M,3] (column 2 is never written), grad_rgb [M,D], grad_opa [M], grad_cov [M,2,2]:
 * one row per (tile, Gaussian) pair.  `workspace` holds the per-bucket pixel checkpoints of
 * the replayed forward pass: gs_draw_backward_workspace_bytes(M, h, w) bytes. */
size_t gs_draw_backward_workspace_bytes(int64_t M, int32_t h, int32_t w);
int gs_draw_backward(const float *pos, const float *rgb, const float *opa, const float *cov,
                     const int32_t *tile_n_point_accum, const float *output,
                     const float *grad_output, float *grad_pos, float *grad_rgb, float *grad_opa,
                     float *grad_cov, int32_t h, int32_t w, int64_t M, float focal_x,
                     float focal_y, int weight_normalize, int sigmoid, int fast,
                     const float *rays_o, const float *lefttop_pos, const float *vec_dx,
                     const float *vec_dy, int use_sh_coeff, void *workspace,
                     size_t workspace_bytes, gs_stream_t stream);

/* =========================================================================================
 * B. Fused frame path (replaces splatter.py:513-655 forward and its autograd backward)
 * ======================================================================================= */

/* Device radix sort of (key, value) pairs on key bits [0, end_bit), stable, ascending (LSD,
 * 8 bits per pass).  The element count is read from DEVICE memory (*d_count, clamped to
 * `capacity`) so that the frame needs no host synchronisation.  The input lives in buffer 0
 * (keys0/vals0, destroyed); buffer 1 is scratch; on return *sorted_in_buffer1 tells which
 * buffer holds the result (it only depends on end_bit).  tmp: gs_sort_pairs_tmp_bytes(capacity). */
size_t gs_sort_pairs_tmp_bytes(int64_t capacity);
int gs_sort_pairs(uint64_t *keys0, uint32_t *vals0, uint64_t *keys1, uint32_t *vals1,
                  const uint32_t *d_count, int64_t capacity, int end_bit, void *tmp,
                  size_t tmp_bytes, int *sorted_in_buffer1, gs_stream_t stream);
/* Same, restricted to key bits [begin_bit, end_bit): a stable partial sort (used by sort_mode 1
 * to group pairs by tile id only). */
int gs_sort_pairs_bits(uint64_t *keys0, uint32_t *vals0, uint64_t *keys1, uint32_t *vals1,
                       const uint32_t *d_count, int64_t capacity, int begin_bit, int end_bit,
                       void *tmp, size_t tmp_bytes, int *sorted_in_buffer1, gs_stream_t stream);

#define GS_FRAME_EMIT_SORTED_KEYS 1 /* sort_mode 2: also write the sorted (tile << 32 | depth bits) keys that
                                       gs_frame_debug_views returns (modes 0 / 1 always have them); the raster
                                       kernels only read the sorted ids, so the default path skips 8 bytes per pair */

#define GS_FRAME_SLICE_SORT 2       /* sort_mode 2: the slice-sorted binning variant (tile_bin.hip) -- every workgroup
                                       counting-sorts the pairs of its slice of the Gaussians by tile inside LDS and
                                       streams them out contiguously, the per-tile sort gathers from the slices.  No
                                       scattered global store, but one small (slice, tile) cell per gather step:
                                       measured slower than the table variant up to 2.4 M Gaussians at 1080p
                                       (DESIGN.md), so it is opt-in.  Same result either way. */

#define GS_FRAME_TABLE_BIN 4        /* sort_mode 2: the table variant (tile_bin.hip: count -> column scan -> one scattered
                                       8-byte store per pair) instead of the default two-level STRIP variant
                                       (strip_bin.hip).  Frames beyond the strip variant's limits (2^26 Gaussians, 8192
                                       strips of 8 tiles) take the table variant by themselves.  Same result. */

#define GS_FRAME_SERIAL_LONG_LISTS 8 /* dense frames (capacity above 1024 pairs per tile on average) cut the rest of a tile's
                                       list into segments composited by many waves when its pixels are still alive after
                                       4096 Gaussians (raster_fwd.hip); this flag keeps the one-wave-per-tile walk for
                                       such tiles (A/B and equivalence tests).  Same result up to the rounding of the
                                       transmittance entering a segment. */

#define GS_FRAME_LONG_LISTS 16       /* run the long-list kernels (big-list sort, segmented compositing): for dense scenes (10 M
                                       Gaussians at 1080p: ~3,500 pairs per tile) and for frames that hold a few very
                                       long lists (pile-ups of a degenerate densification run).  The library cannot know
                                       without a host synchronisation; the caller can: gs_frame_longest_list_async
                                       reports the longest list of an earlier frame.  A frame without the flag is
                                       correct whatever its lists are, only slower on long ones; the workspace CAPACITY
                                       never selects the path (up to ABI version 3 a capacity above 1024 pairs per tile
                                       did). */

#define GS_FRAME_STRIP_BIN 32        /* sort_mode 2: take the strip variant of the binning whatever the scene size.  Without
                                       this flag (and without GS_FRAME_TABLE_BIN / GS_FRAME_SLICE_SORT) the library picks
                                       by N: the table variant below GS_STRIP_AUTO_MIN_N (131,072) Gaussians, where its shorter
                                       chain of dependent kernels wins, the strip variant from there on.  Every variant
                                       produces the same lists. */
#define GS_FRAME_LONG_SORT 128        /* the SORT half of GS_FRAME_LONG_LISTS alone (round 6): tile lists beyond strip_sort_kernel's LDS
                                       window (2,048 pairs) are queued for big_list_sort_kernel, one workgroup per list, instead
                                       of being sorted by their strip's workgroup with the chunked bitonic network -- the tail of
                                       that kernel in a trained / densified scene (3.9 M pairs, 165 lists beyond 2,048: per-tile
                                       sort 407 -> 122 us, profiles/r06_a_*).  The sorted list is the same either way (the order is
                                       exact), so unlike the segmented compositing this half costs nothing but two launches in
                                       frames without such lists.  The caller sets it once gs_frame_longest_list_async reported
                                       a list beyond 2,048; GS_FRAME_LONG_LISTS implies it. */
#define GS_FRAME_OCCLUSION_CULL 256   /* temporal occlusion cull (round 6): allow the frame to drop, at emission, the (tile, Gaussian)
                                       pairs that lie behind the depth at which the PREVIOUS forward of this workspace saw all
                                       pixels of their tile stop -- 71 % of the pairs of the 2.4 M-Gaussian scene are emitted,
                                       scattered and sorted and never composited.  The image is exact (bit-identical to the
                                       unculled frame): the dropped pairs are deeper than every kept one, and a tile that
                                       reaches the end of a trimmed list with a live pixel makes the library render the frame
                                       again from the full lists, on the device, inside the same call (five gated launches:
                                       ~13 us when nothing ran past its cut; 0.6 of a frame when something did, which with thousands
                                       of tiles is nearly every frame of a MOVING camera: set the flag for a camera at rest).  The CALLER's promise: the previous
                                       forward of this workspace was an inference frame (training = 0: training forwards do
                                       not write the table) of the same width and height (the cut table is per tile).
                                       Ignored by training frames, frames with GS_FRAME_EMIT_SORTED_KEYS / GS_FRAME_LONG_LISTS /
                                       GS_FRAME_SERIAL_LONG_LISTS, the "dist" listing and the table / radix variants.
                                       gs_frame_stats_async then reports the pairs that were emitted (fewer than the frame
                                       lists); gs_frame_cull_fallback_async tells whether the frame was re-rendered. */
#define GS_FRAME_CULL_DILATE 512      /* with GS_FRAME_OCCLUSION_CULL: every tile's cut is the LARGEST cut of its 3 x 3 tile
                                       neighbourhood (no cut if one of the nine has none), pushed back by a factor 1.375 in depth
                                       -- for a camera that has MOVED by less than half a tile since the table was recorded.  On
                                       the 2.4 M-Gaussian scene no frame of a 1.25-pixel-per-frame pan falls back (3 of 119 at
                                       5 px per frame), 3.1 M of 6.95 M pairs are emitted (1.8 M with the tiles' own cuts, which
                                       fit the identical pose only): +20 % FPS.  The image is exact either way (the second pass).
                                       One more launch of ~2 us. */
#define GS_FRAME_CULL_DILATE_NEAR 1024 /* with GS_FRAME_CULL_DILATE: the pose is within half a pixel of the recorded one (what a viewer
                                       in motion produces at thousands of frames per second): depth factor 1.125 instead of
                                       1.375 -- 2.5 M instead of 3.1 M of 6.95 M pairs emitted, no fallback over a 0.25-px/frame pan */
#define GS_FRAME_BWD_ROWS 64         /* rgb training frames: composite the backward with the row-layout kernel (lanes = 16
                                       Gaussians x 4 pixel quads, pixel rows whose pixels have all stopped are left out)
                                       instead of the pixel-parallel one.  Worth it when most of the frame's buckets belong
                                       to SATURATED tiles -- tiles whose compositing stopped before the end of their list --
                                       and a loss otherwise (2.4 M Gaussians at 1080p: -9 %; 376 k Gaussians: +11 %, kernel
                                       traces profiles/r05_e_*).  The library cannot know without a host synchronisation;
                                       the caller can: the upper half of gs_frame_stats_async's `buckets` counter is the
                                       number of buckets of saturated tiles in the last backward's work list.  Either
                                       kernel is correct for any frame; the results differ in the last bits (another
                                       summation order).  Ignored by SH frames. */

/* Frame descriptor.  All scalars are per-camera constants computed on the host exactly as
 * splatter.py does (Tiles, RayInfo, frustum guard band); rot/tran are passed by value. */
typedef struct gs_frame {
    /* scene (raw parameters, activations are fused: |s|+1e-4 / exp, q/|q|, sigmoid) */
    int64_t N;
    int32_t color_dim;        /* 3 (sigmoid colour), 27 (degree-2 SH coefficients, the reference's basis_dim 9) or
                                 48 (degree-3 SH: extension with the C3 table of gaussian.cu:395-403) */
    int32_t scale_activation; /* 0 = abs (+1e-4), 1 = exp   (splatter.py:520-524)       */
    const float *pos;         /* [N,3]  */
    const float *quat;        /* [N,4]  */
    const float *scale;       /* [N,3]  */
    const float *opa;         /* [N]    */
    const float *rgb;         /* [N,color_dim] */
    /* camera */
    float rot[9];
    float tran[3];
    float near_plane, half_width, half_height;
    int32_t width, height;    /* un-padded output size */
    float focal_x, focal_y;
    /* Tiles.create_tiles scalars, computed by the host in double (splatter.py:279-282) */
    float tile_length_x, tile_length_y, leftmost, topmost;
    float thresh;             /* tile_culling_prob_thresh (train.py: 0.05) */
    float rays_o[3], lefttop[3], vec_dx[3], vec_dy[3];
    /* workspace (caller-allocated, sizes from gs_frame_workspace_layout) */
    int64_t max_pairs;        /* capacity for (tile, Gaussian) pairs                      */
    void *workspace;
    size_t workspace_bytes;
    /* outputs */
    float *image;             /* [height,width,3] clamped + cropped (may be NULL)         */
    float *image_padded;      /* [padH,padW,3] raw draw output (may be NULL if !training) */
    int32_t training;         /* 1: keep per-bucket checkpoints for gs_frame_backward      */
    int32_t sort_mode;        /* how the (tile, depth) order is produced -- same result either way:
                                 0 = LSD radix sort of 64-bit (tile<<32|depth) keys (gs_sort_pairs)
                                 1 = hybrid: stable LSD radix passes on the tile bits only (2 passes at
                                     1080p), then every tile's bucket sorted on (depth, id) in LDS
                                 2 = counting sort in LDS, no radix pass (default).  Strip variant: 8-byte
                                     entries per (Gaussian, strip of 8 tiles) counting-sorted by strip, expanded
                                     into the tile lists and depth-sorted inside LDS (up to 2^26 Gaussians and
                                     8192 strips; beyond that, or with GS_FRAME_TABLE_BIN: one LDS counter per
                                     tile, <= 32768 tiles, larger grids take mode 1).  On capacity overflow the
                                     frame is left empty (modes 0/1 keep the first max_pairs pairs); all modes
                                     report the true count in the stats. */
    int32_t tile_culling_method; /* which tiles a Gaussian is listed in (splatter.py:571-578, --tile_culling_method;
                                 the reference's own numbering, `_method_config` of splatter.py:571):
                                 2 = "prob2", the trainer's default (gaussian.cu:197-250: tile rectangle from the 2-D
                                     covariance's bounding box by index arithmetic);
                                 1 = "prob" (gaussian.cu:138-195: the same bounding box compared with the tiles'
                                     edges, Tiles.create_tiles of splatter.py:275-293 -- also a rectangle);
                                 0 = "dist" (gaussian.cu:101-136: every tile whose CENTRE lies closer than
                                     sqrt(thresh) to the Gaussian's centre, whatever its size).  `thresh` is then
                                     the squared distance (tile_length_x / tile_culling_dist_thresh)^2 of
                                     splatter.py:577; sort_mode 2 only (<= 32768 tiles).  The gradient rows are
                                     laid out over the bounding square of the disc, so max_pairs must cover the
                                     sum of those squares (reported as overflow otherwise). */
    /* ABI 3 */
    struct gs_frame_async *async; /* NULL, or a handle from gs_frame_async_create: training forwards then run the
                                 backward's preparation underneath the caller's loss (see gs_frame_forward) */
    int32_t flags;            /* GS_FRAME_* bits */
} gs_frame;

/* Bytes of workspace needed for N Gaussians, `max_pairs` pairs, a width x height image. */
size_t gs_frame_workspace_bytes(int64_t N, int64_t max_pairs, int32_t width, int32_t height,
                                int32_t color_dim, int32_t training);

/* Forward frame.  Launches everything on `stream`, never synchronises, and keeps no state of its own.
 * With f->training AND f->async the zero-fill of the per-pair gradient rows and the backward's bucket list are
 * issued on the handle's side stream, which waits for the frame's last kernel, so that they run underneath whatever
 * the caller enqueues between forward and backward (the loss); gs_frame_backward -- from any host thread -- and the
 * next gs_frame_forward with the same handle wait for that stream's event.  Without a handle (or while `stream` is
 * being captured into a graph) the backward prepares inline. */
int gs_frame_forward(const gs_frame *f, gs_stream_t stream);

/* The opt-in side stream: a caller-owned handle (one per workspace in use) holding a non-blocking stream and two
 * events on the current device.  The library keeps nothing else between calls.
 *   gs_frame_async_wait    : make `stream` wait for side-stream work that may still be writing into the workspace of
 *                            the handle's last training forward -- call it before freeing or re-purposing that
 *                            workspace if no gs_frame_backward / gs_frame_forward followed the frame;
 *   gs_frame_async_destroy : releases the handle (work already queued still completes). */
typedef struct gs_frame_async gs_frame_async;
int gs_frame_async_create(gs_frame_async **out);
int gs_frame_async_wait(gs_frame_async *a, gs_stream_t stream);
int gs_frame_async_destroy(gs_frame_async *a);

/* Same work as gs_frame_forward, but brackets every stage with hipEvents on `stream` and
 * returns the stage durations in milliseconds (synchronises; for bench.py / profiling only):
 * stage_ms_host[0..5] = project+count, scan+emit, radix sort, tile ranges, raster, total. */
#define GS_N_STAGES 6
int gs_frame_forward_profile(const gs_frame *f, float *stage_ms_host, gs_stream_t stream);
/* Likewise for gs_frame_backward: [0..2] = raster backward (incl. memset + bucket scan),
 * project backward, total. */
int gs_frame_backward_profile(const gs_frame *f, const float *grad_image, float *grad_pos,
                              float *grad_quat, float *grad_scale, float *grad_opa,
                              float *grad_rgb, float *stage_ms_host, gs_stream_t stream);

/* Counters of the last forward on this workspace, copied device->host asynchronously into
 * `stats_host` (4 x int64: V visible, M pairs, overflow flag, processed buckets).  The
 * caller synchronises the stream before reading.  `processed buckets` (training frames; written by the backward's
 * preparation, i.e. it may still be the previous frame's when read right behind a forward): low 32 bits = buckets of 64
 * Gaussians in the backward's work list, high 32 bits = how many of them belong to saturated tiles (GS_FRAME_BWD_ROWS). */
int gs_frame_stats_async(const gs_frame *f, int64_t *stats_host, gs_stream_t stream);

/* Length of the longest tile list of the last forward on this workspace (sort_mode 2, strip variant; lists of up to
 * 1024 pairs are reported as 0), copied device->host asynchronously into *longest_host.  A caller that sees a value
 * above 2,048 sets GS_FRAME_LONG_SORT on the following frames; GS_FRAME_LONG_LISTS follows a cost model over the longest list,
 * the pairs and the pairs beyond 512 per tile (gs_frame.py, FrameRenderer._note_lists). */
int gs_frame_longest_list_async(const gs_frame *f, int64_t *longest_host, gs_stream_t stream);

/* The ABI-level guard against LAGGING counters (round 6).  A client that copies the counters
 * asynchronously and looks at them frames later -- to grow its workspace, to set GS_FRAME_LONG_SORT / GS_FRAME_LONG_LISTS /
 * GS_FRAME_BWD_ROWS -- must know WHICH frame they belong to: flags latched from the counters of frame k while frame k + 500
 * is being issued made a training run timing-dependent in round 5.  This call enqueues, on `stream`: a fill of the counter
 * block's tag word with `tag` (the client's frame number), then ONE device-to-host copy of GS_STATS_TAGGED_N = 15 values:
 *   [0] visible, [1] emitted pairs, [2] overflow (0, or the pairs the frame needed), [3] buckets (low 32 bits: in the
 *   backward's work list, high: of saturated tiles), [4..8] internal, [9] longest tile list (0 up to 1,024), [10] non-zero iff
 *   an occlusion-culled frame was rendered again from its full lists, [11] the tag (both 32-bit halves), [12] 0 (retired in ABI 8: the pairs beyond
 *   the first 512 of every list, summed with one atomic per long tile -- 14 us on the 2.4 M scene; see [14]), [13] the longest walk a tile's wave actually made in the compositing
 *   (0 up to 1,024; a tile whose pixels stop early walks less than its list), [14] the steps beyond the first 512 of every walk
 *   (what a GS_FRAME_LONG_LISTS frame would composite in segments; both from UNFLAGGED frames: a flagged frame's waves stop at 512).
 * stats_host[11] == tag (low half) <=> the copy has landed and the values are those of the frame issued just before this
 * call on `stream`.  stats_host should be pinned memory (the copy is asynchronous only then).  Recommended policy
 * (gs_frame.py): one copy in flight, and never issue frame k + 8 before the counters of frame k have been looked at. */
#define GS_STATS_TAGGED_N 15
int gs_frame_stats_tagged_async(const gs_frame *f, uint32_t tag, int64_t *stats_host, gs_stream_t stream);

/* Non-zero in *ran_past_host (after `stream` has reached this copy) iff the frame's lists had been trimmed
 * (GS_FRAME_OCCLUSION_CULL) and a tile ran past its cut, i.e. the frame was rendered a second time from the full lists. */
int gs_frame_cull_fallback_async(const gs_frame *f, int64_t *ran_past_host, gs_stream_t stream);

/* *culled = 1 iff the library renders this frame description with the occlusion cull (the flag is set AND none of the
 * conditions under which it is ignored holds); host-side, no launch.  Such a frame's first pass tests every Gaussian
 * against the frustum and the cut pyramid on its position and scale alone and projects the survivors only: the rectangle
 * records (gs_frame_debug_rects) and the 64-byte records are then fresh for the PROJECTED Gaussians only -- the others keep
 * what an earlier frame left -- unless the frame fell back (second pass: everything is projected again). */
int gs_frame_is_occlusion_culled(const gs_frame *f, int32_t *culled);

/* (Validity, since ABI 4: `tiles_touched` is written by sort_modes 0 / 1 only -- sort_mode 2 keeps the count in
 * rects[i].w, gs_frame_debug_rects -- and the rec_* records are written for VISIBLE Gaussians only: the record of a
 * culled Gaussian holds whatever an earlier frame left there.  The culling mask is `rects[i].z != 0`, never
 * `rec_geom[i].z != 0`.)
 * Read-only views into the workspace of the last forward (for parity tests): sorted keys
 * (u64 [M]; NULL for a sort_mode-2 frame rendered without GS_FRAME_EMIT_SORTED_KEYS -- they are
 * (tile << 32 | depth bits of rec_geom[id].z) of the sorted ids), sorted Gaussian ids (u32 [M]), per-tile
 * ranges (int32 [T,2]), projected pos/cov/mask-equivalents.  Any out pointer may be NULL. */
int gs_frame_debug_views(const gs_frame *f, const uint64_t **sorted_keys,
                         const uint32_t **sorted_ids, const int32_t **tile_ranges,
                         const float **rec_geom, const float **rec_cov, const float **rec_color,
                         const uint32_t **tiles_touched);

/* Which binning / sort path gs_frame_forward takes for this frame description: 0 / 1 = sort_mode 0 / 1 (radix passes on
 * the 64-bit key / on its tile bits), 2 = sort_mode 2 with the table variant, 3 = slice-sorted variant, 4 = strip
 * variant (see GS_FRAME_STRIP_BIN for the size-based choice).  Negative: the description does not validate. */
int gs_frame_binning_variant(const gs_frame *f);

/* [N][4] uint32, written for EVERY Gaussian by every forward: (y0 | y1 << 16, x0 | x1 << 16, depth bits, tiles touched) --
 * the tile rectangle [y0, y1) x [x0, x1), the float bits of the depth |p_c|, the number of tiles listed.  The depth bits
 * are 0 exactly for the Gaussians the frustum test culled (the reference's culling_mask, renderer.py:123-132, is
 * `depth bits != 0`); the 64-byte record (rec_geom / rec_cov / rec_color above) of a culled Gaussian is NOT written and
 * holds whatever an earlier frame left there. */
int gs_frame_debug_rects(const gs_frame *f, const uint32_t **rects);

/* [T] uint32: how many Gaussians of its list each tile's forward actually composited before all of its pixels had
 * stopped (a multiple of 64 except for the last chunk), kept by TRAINING forwards for the backward.  Measurement
 * aid: the compositing work of a frame is sum(tile_nproc) steps, not the pair count. */
int gs_frame_debug_tile_nproc(const gs_frame *f, const uint32_t **tile_nproc);

/* SH training frames whose raster backward ran on the matrix pipe (color_dim 27 / 48, frame path): [*n_slots] uint32, the
 * pixel-row steps (16 Gaussians x 16 pixels: 3 floor(NB / 4) + 12 fp32 MFMAs each) every wave of the LAST backward executed --
 * rows whose 16 pixels had all stopped are left out, so this, not 16 x the composited steps / 16, is what the kernel's
 * MFMA flops are counted from (bench.py).  *n_slots = 0 for frames that take another kernel.  Measurement aid. */
int gs_frame_debug_bwd_exec_rows(const gs_frame *f, const uint32_t **exec_rows, int32_t *n_slots);

/* Backward frame: grad_image is dL/d(image) [height,width,3] (w.r.t. the clamped, cropped
 * output).  Writes dL/d(raw parameter) for every Gaussian (zeros for culled ones):
 * grad_pos [N,3], grad_quat [N,4], grad_scale [N,3], grad_opa [N], grad_rgb [N,color_dim].
 * Must follow a gs_frame_forward with training=1 on the same workspace. */
int gs_frame_backward(const gs_frame *f, const float *grad_image, float *grad_pos,
                      float *grad_quat, float *grad_scale, float *grad_opa, float *grad_rgb,
                      gs_stream_t stream);

/* The same backward in three calls, for the view-parallel gradient exchange (gs_dp.py): GS_BWD_RASTER runs the raster
 * backward (per-pair gradient rows in the workspace; needs grad_image only), after which GS_BWD_GEOMETRY writes
 * grad_pos / grad_quat / grad_scale and GS_BWD_COLOR writes grad_opa / grad_rgb -- in either order, each bit-identical
 * to what gs_frame_backward writes -- so that the all-reduce of the bucket written first runs underneath the kernel
 * that writes the other one.  Pointers a part does not write may be NULL. */
#define GS_BWD_RASTER 1
#define GS_BWD_GEOMETRY 2
#define GS_BWD_COLOR 4
int gs_frame_backward_part(const gs_frame *f, const float *grad_image, float *grad_pos, float *grad_quat,
                           float *grad_scale, float *grad_opa, float *grad_rgb, int32_t part, gs_stream_t stream);

/* ABI 5 -- the frame in pieces, for the view-parallel trainer (gs_dp.py, gs_train.py; no reference analogue: the
 * reference is single-GPU).  The gradients of a frame only become final in the last kernel of the backward -- the
 * per-Gaussian sum of the per-pair gradient rows -- and that kernel is independent per Gaussian:
 *   gs_frame_backward_slice : after gs_frame_backward_part(GS_BWD_RASTER), the sums of the Gaussians [g_begin, g_end)
 *                             only (g_begin a multiple of 256; `part` = GS_BWD_GEOMETRY, GS_BWD_COLOR or both OR-ed:
 *                             both in ONE kernel, the rows are read once).  Any partition of [0, N) into ranges writes
 *                             exactly what gs_frame_backward writes, so that the exchange of one range's gradients
 *                             runs underneath the sums of the next.
 * and the NEXT frame's project stage (cull + project + activations + level-1 count) needs nothing but the parameters of
 * its own Gaussians:
 *   gs_frame_project_slices : *slices = how many slices of the Gaussian array the project stage of this frame is made
 *                             of (0: it cannot be issued in pieces -- only sort_mode 2's strip variant can),
 *                             *per_slice = Gaussians per slice (a multiple of 256);
 *   gs_frame_forward_project: the project stage of slices [slice_begin, slice_end) -- call it for every slice exactly
 *                             once, the range that starts at slice 0 first;
 *   gs_frame_forward_rest   : binning, per-tile sort, compositing.  project(all slices) + rest == gs_frame_forward.
 * Nothing of the previous frame's workspace contents that its backward still needs is touched by the project stage
 * except the per-Gaussian records and rectangles, i.e. issue a slice's project only after that slice's
 * gs_frame_backward_slice (same stream). */
int gs_frame_backward_slice(const gs_frame *f, float *grad_pos, float *grad_quat, float *grad_scale, float *grad_opa,
                            float *grad_rgb, int32_t part, int64_t g_begin, int64_t g_end, gs_stream_t stream);
int gs_frame_project_slices(const gs_frame *f, int32_t *slices, int64_t *per_slice);
int gs_frame_forward_project(const gs_frame *f, int32_t slice_begin, int32_t slice_end, gs_stream_t stream);
int gs_frame_forward_rest(const gs_frame *f, gs_stream_t stream);

/* ===================================================================================
 * Section C -- the training step around the frame
 * =================================================================================== */

/* torch.optim.Adam(betas=(beta1, beta2), eps) .step() of the reference trainer (train.py:59-67, :113;
 * no weight decay, no amsgrad) on ONE flat fp32 bucket: param / grad / exp_avg / exp_avg_sq [n], all
 * 16-byte aligned.  Parameter group k is the index range [group_end[k-1], group_end[k]) (group_end[-1] = 0,
 * group_end[n_groups-1] = n, n_groups <= 8) with learning rate lr[k]; both tables are HOST arrays.
 * `step` is the 1-based step count (bias corrections are evaluated in double on the host, like torch).
 * Optional densification statistic of train.py:145-154 for the index range [stat_begin, stat_end) (the
 * `pos` group): stat_mode 1: grad_stat = max(grad_stat, |grad|); 2: grad_stat += |grad|; 0: off. */
int gs_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                 int32_t n_groups, const int64_t *group_end, const float *lr, float beta1, float beta2,
                 float eps, int64_t step, float *grad_stat, int64_t stat_begin, int64_t stat_end,
                 int32_t stat_mode, gs_stream_t stream);

/* The same step restricted to the elements [range_begin, range_end) of the same arrays (group table and statistic range
 * stay absolute): one bucket of the view-parallel gradient exchange at a time, so that the update of the bucket whose
 * all-reduce has finished runs underneath the all-reduce of the next one.  Every element's update is what gs_adam_step
 * computes for it.  The statistic is only touched where its range intersects [range_begin, range_end). */
int gs_adam_step_range(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                       int64_t range_begin, int64_t range_end, int32_t n_groups, const int64_t *group_end,
                       const float *lr, float beta1, float beta2, float eps, int64_t step, float *grad_stat,
                       int64_t stat_begin, int64_t stat_end, int32_t stat_mode, gs_stream_t stream);

/* View-parallel training with a SHARDED optimizer (reduce-scatter -> this step -> all-gather of the parameters): the
 * rank owns elements [range_begin, range_end) of the flat index space and keeps only THEIR moments: exp_avg_shard[0] /
 * exp_avg_sq_shard[0] belong to element `moment_base` (a multiple of 4, <= range_begin); param / grad / the group table
 * and the statistic range stay absolute.  `skip_if_nonzero` (may be NULL) is the device address of a 64-bit counter:
 * when it is non-zero the launch leaves everything untouched -- pass gs_frame_overflow_flag() so that a training frame
 * that overflowed its workspace (rendered empty, all-zero gradient) does not move the parameters by momentum, without
 * any host synchronisation.  Every element's update is what gs_adam_step computes for it. */
int gs_adam_step_sharded(float *param, const float *grad, float *exp_avg_shard, float *exp_avg_sq_shard, int64_t n,
                         int64_t range_begin, int64_t range_end, int64_t moment_base, int32_t n_groups,
                         const int64_t *group_end, const float *lr, float beta1, float beta2, float eps, int64_t step,
                         float *grad_stat, int64_t stat_begin, int64_t stat_end, int32_t stat_mode,
                         const void *skip_if_nonzero, gs_stream_t stream);

/* Up to 8 element ranges [range_begin[r], range_end[r]) of the same flat arrays in ONE launch (host tables;
 * range_begin a multiple of 4): one slice of the view-parallel exchange is a range of Gaussians in each of the five
 * parameter arrays.  The moments of range r start at exp_avg + moment_offset[r] (a multiple of 4): element i keeps its
 * moments at index moment_offset[r] + (i - range_begin[r]) -- moment_offset = range_begin for a replicated optimizer,
 * densely packed shards for a sharded one.  Group table, statistic and skip flag as in gs_adam_step_sharded.
 * grad_scale: the gradient enters as grad * grad_scale -- 1 / world when the exchange SUMS the ranks' gradients and
 * leaves the mean to the optimizer (RCCL's ReduceOp.AVG is a pre-multiplied sum that launches a scaling kernel even on
 * one rank; SUM does not), 1 otherwise: every element's update is then what gs_adam_step computes for it. */
int gs_adam_step_multi(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, int32_t n_ranges,
                       const int64_t *range_begin, const int64_t *range_end, const int64_t *moment_offset,
                       int32_t n_groups, const int64_t *group_end, const float *lr, float beta1, float beta2, float eps,
                       int64_t step, float *grad_stat, int64_t stat_begin, int64_t stat_end, int32_t stat_mode,
                       const void *skip_if_nonzero, float grad_scale, gs_stream_t stream);

/* Single-GPU training step (round 5: rgb colours; round 6: SH colours too): gs_frame_backward whose LAST kernel -- the per-Gaussian sum of the gradient
 * rows + projection / activation backward -- applies the Adam update to the Gaussian's 14 parameters on the spot instead of
 * writing their gradients.  The gradient never travels through memory (2 x 56 B per Gaussian) and the optimizer's stream of
 * parameters and moments runs underneath the projection backward's arithmetic (that kernel is VALU-bound, gs_adam_step
 * HBM-bound).  Every updated value is bit for bit what gs_frame_backward + gs_adam_step give (same expressions, same
 * rounding); culled Gaussians take their zero-gradient step (momentum) as they do there.
 *   The PARAMETERS updated in place are the frame's own f->pos, quat, scale, opa, rgb (the descriptor calls them const: they
 * are not, here); exp_avg / exp_avg_sq: their moments, same shapes, in the order pos, quat, scale, opa, rgb; lr likewise;
 * step counts from 1; grad_stat (may be NULL with stat_mode 0): [N,3], max (1) or sum (2) of |dL/dpos| folded in
 * (train.py:145-154); skip_if_nonzero (may be NULL): device address of a 64-bit counter -- non-zero: the step is skipped (the
 * frame's overflow counter, gs_frame_overflow_flag).  SH frames (color_dim 27 / 48): the wave that sums a Gaussian's
 * coefficient gradients steps the coefficients where it would have stored the gradients (+1 % on the SH training step: the
 * coefficient gradients' round trip and one launch; the optimizer's own stream stays).  No gradient buffer is written. */
typedef struct gs_adam_fused {
    float *exp_avg[5], *exp_avg_sq[5];
    float lr[5];
    float beta1, beta2, eps;
    int64_t step;
    float *grad_stat;
    int32_t stat_mode;
    const void *skip_if_nonzero;
} gs_adam_fused;
int gs_frame_backward_adam(const gs_frame *f, const float *grad_image, const gs_adam_fused *adam, gs_stream_t stream);

/* Device address of the frame's overflow counter (inside the caller's workspace; 64-bit, 0 = the last forward of this
 * frame description fitted its pair capacity, else the pair count it would have needed).  No launch, no copy. */
int gs_frame_overflow_flag(const gs_frame *f, const void **device_counter);

/* The same statistic without the Adam update: stat[i] = max(stat[i], |grad[i]|) (stat_mode 1) or stat[i] += |grad[i]|
 * (stat_mode 2) for i < n.  View-parallel training (one view per GPU) needs it: train.py:145-154 accumulates the
 * |gradient| of EACH VIEW, so every rank updates the statistic from its own gradient before the all-reduce averages
 * the gradients, and the statistics are combined (max / sum over ranks) at the densification boundaries. */
int gs_grad_stat_update(const float *grad, float *stat, int64_t n, int32_t stat_mode, gs_stream_t stream);

/* Image loss of train.py:99-107 and its gradient:
 *   loss = (1 - ssim_weight) * mean|pred - target| + ssim_weight * (1 - SSIM(pred, target)),
 * SSIM = torchmetrics StructuralSimilarityIndexMeasure(data_range=1.0) (11x11 Gaussian window, sigma 1.5,
 * evaluated on the pixels whose window lies inside the image).  pred, target, grad: [H, W, 3] fp32;
 * grad = dloss/dpred.  loss_out (device, may be NULL) receives (loss, l1 mean, ssim mean).
 * ssim_weight = 0 skips the SSIM passes (any image size); otherwise H, W must exceed 10. */
size_t gs_loss_workspace_bytes(int32_t H, int32_t W);
int gs_loss_l1_ssim(const float *pred, const float *target, int32_t H, int32_t W, float ssim_weight,
                    float *grad, float *loss_out, void *workspace, size_t workspace_bytes,
                    gs_stream_t stream);

/* ---- densification: Gaussian3ds.adaptive_control (splatter.py:122-228) ----
 * Two calls because the number of split Gaussians decides how many standard-normal draws the split
 * samples consume (the reference's MultivariateNormal.sample() draws them after its masks are known):
 *   gs_densify_classify: classes + counts_dev[4] = (kept, cloned, split, kept + cloned + split);
 *   gs_densify_apply   : writes the new Gaussian set in the reference's order -- kept ones (split ones
 *                        with scale / 1.6 (abs) or - log 1.6 (exp) and the first sample), clones
 *                        (pos - grad * clone_dt), second split samples.  eps1 / eps2: [n_eps, 3]
 *                        standard normals, n_eps >= split count; outputs hold `capacity` Gaussians.
 *                        Nothing is written if the total exceeds `capacity` or the split count n_eps.
 * A sample is pos + chol(R S^2 R^T) eps (utils.py:391-402).  grad: [N, 3] accumulated |grad_pos| statistic
 * (train.py:160).  All arrays fp32, rgb [N, color_dim]; the same workspace must be passed to both calls. */
typedef struct gs_densify_opts {
    float taus;              /* --split_thresh: ||act(scale)|| above which a Gaussian is split, else cloned */
    float delete_thresh;     /* --delete_thresh: Gaussians with ||act(scale)|| >= this are pruned            */
    float grad_thresh;       /* --grad_thresh                                                                 */
    float clone_dt;          /* --clone_dt                                                                    */
    int32_t scale_activation;/* 0 = abs, 1 = exp                                                              */
    int32_t grad_aggregation;/* 0 = max over xyz, 1 = mean                                                    */
    int32_t use_clone, use_split;
    int32_t color_dim;       /* 3, 27 or 48 (the colour row is copied through)                                */
} gs_densify_opts;
size_t gs_densify_workspace_bytes(int64_t N);
int gs_densify_classify(const float *scale, const float *opa, const float *grad, int64_t N,
                        const gs_densify_opts *opts, int64_t *counts_dev, void *workspace,
                        size_t workspace_bytes, gs_stream_t stream);
int gs_densify_apply(const float *pos, const float *quat, const float *scale, const float *opa,
                     const float *rgb, const float *grad, int64_t N, const gs_densify_opts *opts,
                     const float *eps1, const float *eps2, int64_t n_eps, float *out_pos, float *out_quat,
                     float *out_scale, float *out_opa, float *out_rgb, int64_t capacity,
                     const int64_t *counts_dev, void *workspace, size_t workspace_bytes, gs_stream_t stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* GS_ABI_H */
