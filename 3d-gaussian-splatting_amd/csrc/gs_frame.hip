// gs_frame.hip -- host orchestration of the fused frame path + error plumbing.
//
// gs_frame_forward issues, on ONE stream and with NO host synchronisation:
//   sort_mode 2 (default), strip variant: S1 project -> strip count -> column scan -> strip scatter (8-byte entries per
//       (Gaussian, strip of 8 tiles), counting-sorted by strip through LDS) -> strip sort (a workgroup per half strip
//       expands the entries into its four tile lists inside LDS and sorts them; [-> big-list sort in dense frames])
//       -> raster forward: six launches, no memset, no scattered global store.  Table / slice-sorted variants of
//       round 1 behind GS_FRAME_TABLE_BIN / GS_FRAME_SLICE_SORT (tile_bin.hip);
//   sort_modes 0 / 1: memset(counters, ranges) -> S1 -> scan block sums -> emit keys -> LSD radix passes
//       (3 launches per 8 bits of the key: all of it, or the tile bits only) -> tile ranges
//       [-> per-tile sort] -> raster forward.
// The number of (tile, Gaussian) pairs M lives in device memory only; every later stage is
// launched with a capacity-sized grid (or reads the tile ranges) and idles past M.  The reference
// needs >= 8 blocking host syncs for the same work.
#include <stdarg.h>
#include <string.h>

#include <mutex>
#include <new>

#include "gs_common.h"
#include "gs_frame_layout.h"

static thread_local char g_err[512] = "";

void gs_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *gs_last_error(void) { return g_err; }
extern "C" int gs_abi_version(void) { return GS_ABI_VERSION; }
extern "C" int gs_culling(void) { return 0; }  // gaussian.cu:6-8 prints "hellow"; nothing to compute

static int tile_bits(int n_tiles) {
    int b = 0;
    while ((1 << b) < n_tiles) ++b;
    return b < 1 ? 1 : b;
}

static bool use_strip_variant(const gs_frame *f);

static int validate(const gs_frame *f) {
    GS_CHECK_ARG(f != nullptr, "frame is null");
    GS_CHECK_ARG(f->N >= 0 && f->N < (1ll << 31), "N out of range");
    GS_CHECK_ARG(f->tile_culling_method >= 0 && f->tile_culling_method <= 2,
                 "tile_culling_method must be 0 (dist), 1 (prob) or 2 (prob2)");
    GS_CHECK_ARG(f->color_dim == 3 || f->color_dim == 27 || f->color_dim == 48,
                 "color_dim must be 3 (rgb logits), 27 (SH degree 2) or 48 (SH degree 3)");
    GS_CHECK_ARG(f->scale_activation == 0 || f->scale_activation == 1, "scale_activation must be 0 (abs) or 1 (exp)");
    GS_CHECK_ARG(f->width > 0 && f->height > 0, "empty image");
    GS_CHECK_ARG(f->width <= 65535 * 16 && f->height <= 65535 * 16, "image too large");
    GS_CHECK_ARG(f->max_pairs > 0 && f->max_pairs < (1ll << 30), "max_pairs out of range");
    GS_CHECK_ARG(f->N == 0 || (f->pos && f->quat && f->scale && f->opa && f->rgb), "null scene pointer");
    GS_CHECK_ARG(((uintptr_t)f->quat & 15) == 0, "quat must be 16-byte aligned");
    GS_CHECK_ARG(f->workspace != nullptr && ((uintptr_t)f->workspace & 255) == 0, "workspace null or not 256-byte aligned");
    if (f->tile_culling_method == 0) {
        GS_CHECK_ARG(f->thresh > 0.f && f->thresh < 3.0e38f, "dist: thresh is a squared distance, must be positive");
        GS_CHECK_ARG(f->sort_mode == 2 && (use_strip_variant(f) || gs_frame_geometry(f).n_tiles <= GS_BIN_MAX_TILES),
                     "tile_culling_method dist needs sort_mode 2 (table variant: at most 32768 tiles)");
    } else {
        GS_CHECK_ARG(f->thresh > 0.f && f->thresh < 1.f, "thresh must be in (0,1)");
    }
    GS_CHECK_ARG((f->flags & ~(GS_FRAME_EMIT_SORTED_KEYS | GS_FRAME_SLICE_SORT | GS_FRAME_TABLE_BIN |
                               GS_FRAME_SERIAL_LONG_LISTS | GS_FRAME_LONG_LISTS | GS_FRAME_STRIP_BIN |
                               GS_FRAME_BWD_ROWS | GS_FRAME_LONG_SORT | GS_FRAME_OCCLUSION_CULL | GS_FRAME_CULL_DILATE | GS_FRAME_CULL_DILATE_NEAR)) == 0,
                 "unknown flag bits");
    GS_CHECK_ARG(f->sort_mode >= 0 && f->sort_mode <= 2,
                 "sort_mode must be 0 (full LSD radix), 1 (tile-bit radix + per-tile LDS sort) or 2 (LDS counting sort "
                 "by tile + per-tile LDS sort)");
    const size_t need = gs_frame_workspace_bytes(f->N, f->max_pairs, f->width, f->height, f->color_dim, f->training);
    if (f->workspace_bytes < need) {
        gs_set_error("gs_frame: workspace too small (%zu < %zu bytes)", f->workspace_bytes, need);
        return GS_E_CAPACITY;
    }
    return 0;
}

extern "C" size_t gs_frame_workspace_bytes(int64_t N, int64_t max_pairs, int32_t width, int32_t height,
                                           int32_t color_dim, int32_t training) {
    if (N < 0 || max_pairs < 0 || width <= 0 || height <= 0) return 0;
    return gs_frame_carve(nullptr, N, max_pairs, width, height, color_dim, training).total_bytes;
}

// sort_mode 2: the strip variant needs one LDS counter per strip, the other variants one per tile; grids beyond that
// (> 8K x 4K pixels) take mode 1.
static int effective_sort_mode(const gs_frame *f) {
    if (f->sort_mode == 2 && !use_strip_variant(f) && gs_frame_geometry(f).n_tiles > GS_BIN_MAX_TILES) return 1;
    return f->sort_mode;
}

// sort_mode 2 runs the strip variant (strip_bin.hip) unless the caller asks for one of the others or the frame is
// outside its limits (2^26 Gaussians, GS_STRIP_MAX strips).
static bool use_strip_variant(const gs_frame *f) { return gs_frame_uses_strips(f); }

// Which double-buffer half holds the sorted (keys, ids) after the radix passes of this mode.
static int sort_passes(const gs_frame *f) {
    gs_frame_geom G = gs_frame_geometry(f);
    const int tb = tile_bits(G.n_tiles);
    return effective_sort_mode(f) == 1 ? (tb + 7) / 8 : (32 + tb + 7) / 8;
}
static void sorted_buffers(const gs_frame *f, const gs_frame_ws &ws, uint64_t **keys, uint32_t **ids,
                           uint64_t **other_keys) {
    if (effective_sort_mode(f) == 2) {  // packed pairs in keys_a -> sorted keys in keys_b, ids in vals_a
        *keys = ws.keys_b;
        *ids = ws.vals_a;
        *other_keys = ws.keys_a;
        return;
    }
    const bool in_b = sort_passes(f) & 1;
    *keys = in_b ? ws.keys_b : ws.keys_a;
    *ids = in_b ? ws.vals_b : ws.vals_a;
    *other_keys = in_b ? ws.keys_a : ws.keys_b;
}

struct StageTimer {
    hipEvent_t ev[GS_N_STAGES + 2];  // (+1: the gated second pass of an occlusion-culled frame counts towards the total only)
    int n = 0;
    bool on;
    hipStream_t s;
    bool ok = true;  // any event call failed: finish() reports it instead of returning garbage times
    StageTimer(bool enabled, hipStream_t stream) : on(enabled), s(stream) {
        for (auto &e : ev) e = nullptr;
        if (on)
            for (auto &e : ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    }
    ~StageTimer() {
        for (auto &e : ev)
            if (e) (void)hipEventDestroy(e);
    }
    void mark() {
        if (on && ok && n <= GS_N_STAGES + 1) ok = hipEventRecord(ev[n++], s) == hipSuccess;
    }
    // ms[i] = ev[i+1] - ev[i]; ms[last] = total
    int finish(float *ms, int n_stages) {
        if (!on) return 0;
        if (!ok || n < 2) {
            gs_set_error("stage timer: hipEvent create / record failed");
            return GS_E_INVALID;
        }
        GS_HIP(hipEventSynchronize(ev[n - 1]));
        for (int i = 0; i + 1 < n && i < n_stages - 1; ++i) GS_HIP(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));
        GS_HIP(hipEventElapsedTime(&ms[n_stages - 1], ev[0], ev[n - 1]));
        return 0;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// Backward preparation underneath the caller's loss (opt-in: gs_frame.async).  After a TRAINING forward the zero-fill
// of the gradient rows and the bucket work list (gs_stage_backward_prepare: ~25 us at 376 k Gaussians, on the critical
// path if done inside gs_frame_backward) are issued on the side stream of the CALLER-OWNED handle, which waits for the
// forward's last kernel; the backward call waits for the handle's event instead of doing the work.  The library itself
// keeps no state: the stream, the two events and the pending flag live in the handle (gs_frame_async_create /
// _destroy), and PyTorch may call backward from its autograd thread -- hence the mutex inside the handle.
// Skipped while the stream is being captured into a graph (the fork would never be joined inside the capture).
#ifndef GS_ASYNC_EVENT_FLAGS
// fork / join between two streams of ONE device: no system-scope release / acquire fence with the event (the host never waits on
// these events, and what the two streams hand each other lives in device memory)
// (hipEventDisableSystemFence is documented for timing-only events; relying on each kernel's own agent-scope
// release for cross-XCD visibility of the preparation's outputs is an undocumented runtime detail, and its gain was never
// measured above noise.  Default: the documented flag only; -DGS_ASYNC_EVENT_FLAGS=... to experiment.)
#define GS_ASYNC_EVENT_FLAGS hipEventDisableTiming
#endif
struct gs_frame_async {
    std::mutex mu;
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, done = nullptr;
    int device = -1;
    bool pending = false;
};

extern "C" int gs_frame_async_create(gs_frame_async **out) {
    GS_CHECK_ARG(out != nullptr, "out is null");
    gs_frame_async *a = new (std::nothrow) gs_frame_async();
    GS_CHECK_ARG(a != nullptr, "out of memory");
    if (hipGetDevice(&a->device) != hipSuccess ||
        hipStreamCreateWithFlags(&a->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&a->fork, GS_ASYNC_EVENT_FLAGS) != hipSuccess ||
        hipEventCreateWithFlags(&a->done, GS_ASYNC_EVENT_FLAGS) != hipSuccess) {
        if (a->fork) (void)hipEventDestroy(a->fork);
        if (a->done) (void)hipEventDestroy(a->done);
        if (a->side) (void)hipStreamDestroy(a->side);
        delete a;
        gs_set_error("gs_frame_async_create: could not create the side stream / events");
        return GS_E_INVALID;
    }
    *out = a;
    return 0;
}

extern "C" int gs_frame_async_wait(gs_frame_async *a, gs_stream_t stream) {
    if (!a) return 0;
    std::lock_guard<std::mutex> lock(a->mu);
    if (a->pending) GS_HIP(hipStreamWaitEvent((hipStream_t)stream, a->done, 0));
    a->pending = false;
    return 0;
}

extern "C" int gs_frame_async_destroy(gs_frame_async *a) {
    if (!a) return 0;
    {
        std::lock_guard<std::mutex> lock(a->mu);
        // queued work keeps running: the runtime releases a stream / event once what was enqueued on it has drained
        if (a->fork) (void)hipEventDestroy(a->fork);
        if (a->done) (void)hipEventDestroy(a->done);
        if (a->side) (void)hipStreamDestroy(a->side);
    }
    delete a;
    return 0;
}

static void prepare_on_side_stream(const gs_frame *f, const gs_frame_ws &ws, const uint32_t *sorted_ids, hipStream_t s) {
    gs_frame_async *a = (gs_frame_async *)f->async;
    if (!a) return;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != a->device) return;  // a handle belongs to the device it was made on
    std::lock_guard<std::mutex> lock(a->mu);
    a->pending = false;
    if (hipEventRecord(a->fork, s) != hipSuccess || hipStreamWaitEvent(a->side, a->fork, 0) != hipSuccess) return;
    if (gs_stage_backward_prepare(f, ws, sorted_ids, a->side) != 0) return;
    if (hipEventRecord(a->done, a->side) != hipSuccess) return;
    a->pending = true;
}

// true: the preparation of the handle's last forward is (being) done on the side stream and `s` now waits for it
static bool join_prepared(const gs_frame *f, hipStream_t s) {
    gs_frame_async *a = (gs_frame_async *)f->async;
    if (!a) return false;
    std::lock_guard<std::mutex> lock(a->mu);
    if (!a->pending) return false;
    a->pending = false;
    return hipStreamWaitEvent(s, a->done, 0) == hipSuccess;
}

// The forward in two phases: PROJECT (cull + project + activations + level-1 count; with the strip variant's fused count
// it may be issued as several ranges of slices of the Gaussian array, in any order, slice 0's range first) and REST
// (binning, per-tile sort, compositing).  gs_frame_forward = both; gs_frame_forward_project / _rest expose them to the
// view-parallel trainer, which projects the NEXT frame's Gaussians range by range as their parameters come out of the
// optimizer, underneath the gradient exchange of the remaining ranges (gs_dp.py).
// phases: 1 = project, 2 = rest, 3 = both
static int frame_forward_impl(const gs_frame *f, hipStream_t s, float *stage_ms, int phases = 3, int slice_begin = 0,
                              int slice_end = -1) {
    int rc = validate(f);
    if (rc) return rc;
    GS_CHECK_ARG(!f->training || f->image_padded, "training needs image_padded");
    GS_CHECK_ARG(f->image || f->image_padded, "no output image");
    gs_frame_ws ws = gs_frame_carve(f->workspace, f->N, f->max_pairs, f->width, f->height, f->color_dim, f->training);
    gs_frame_geom G = gs_frame_geometry(f);
    StageTimer tm(stage_ms != nullptr, s);
    if (phases & 1) {
        if (slice_begin == 0) {
            // a forward whose backward never came may still be zero-filling this workspace on the side stream
            (void)join_prepared(f, s);
        }
        tm.mark();
        // sort_mode 2 writes every counter and every tile range itself (tile_bin.hip, workgroup 0)
        if ((effective_sort_mode(f) != 2 || f->N == 0) && slice_begin == 0)
            GS_HIP(hipMemsetAsync(ws.counters, 0, ws.zero_bytes, s));
        if (gs_frame_occlusion_cull(f) && (f->flags & GS_FRAME_CULL_DILATE) && (rc = gs_stage_cut_dilate(f, ws, s))) return rc;
        if (f->N > 0 && (rc = gs_stage_project(f, ws, s, slice_begin, slice_end))) return rc;
        if (!(phases & 2)) return 0;
    } else {
        tm.mark();
    }
    tm.mark();
    uint64_t *skeys, *okeys;
    uint32_t *sids;
    sorted_buffers(f, ws, &skeys, &sids, &okeys);
    const int mode = effective_sort_mode(f);
    if (mode == 2) {
        // counting sort by tile in LDS (stages "scan_emit" and "sort" collapse into this one)
        const bool strips = use_strip_variant(f);
        if (f->N > 0 && (rc = strips ? gs_stage_strip_bin(f, ws, s) : gs_stage_tile_bin(f, ws, s))) return rc;
        tm.mark();
        tm.mark();
        if (f->N > 0) {
            uint64_t *keys_out = (f->flags & GS_FRAME_EMIT_SORTED_KEYS) ? skeys : nullptr;
            // strip variant: okeys holds the strip-ordered entries; half strips beyond the LDS window expand into skeys
            // and sort there.  Slice-sorted variant: okeys holds the S tile-ordered slice regions, the per-tile sort
            // gathers from them (buckets beyond its LDS window are gathered into skeys and sorted there); table variant:
            // okeys holds the pairs grouped by tile
            if (strips)
                rc = gs_stage_strip_sort(f, ws, okeys, skeys, keys_out, sids, s);
            else if (gs_bin_plan_for(f->N, f->max_pairs, G.n_tiles, (f->flags & GS_FRAME_SLICE_SORT) != 0).lds_sort)
                rc = gs_stage_tile_sort_gather(f, ws, okeys, skeys, keys_out, sids, s);
            else
                rc = gs_stage_tile_sort_packed(f, ws, okeys, keys_out, sids, s);
            if (rc) return rc;
        }
    } else {
        if (f->N > 0 && (rc = gs_stage_scan_emit(f, ws, s))) return rc;
        tm.mark();
        if (f->N > 0) {
            // mode 0: all 32 + tile_bits key bits; mode 1: the tile bits only (stable => grouped by tile,
            // Gaussian-index order inside a tile), the depth order is finished per tile in LDS below
            int in1 = 0;
            const int tb = tile_bits(G.n_tiles);
            rc = gs_sort_pairs_bits(ws.keys_a, ws.vals_a, ws.keys_b, ws.vals_b,
                                    (const uint32_t *)(ws.counters + GS_CNT_PAIRS), f->max_pairs, mode == 1 ? 32 : 0,
                                    32 + tb, ws.sort_tmp, ws.sort_tmp_bytes, &in1, s);
            if (rc) return rc;
        }
        tm.mark();
        if ((rc = gs_stage_tile_ranges(f, ws, skeys, s))) return rc;
        if (mode == 1 && f->N > 0 && (rc = gs_stage_tile_sort(f, ws, skeys, sids, okeys, s))) return rc;
    }
    tm.mark();  // stage "ranges" = tile ranges (+ the per-tile depth sort in mode 1)
    if ((rc = gs_stage_raster_forward(f, ws, sids, s))) return rc;
    tm.mark();
    static const bool no_second_pass = getenv("GS_CULL_NO_SECOND_PASS") != nullptr;  // TIMING ONLY: a frame that ran past a cut stays wrong
    if (mode == 2 && gs_frame_occlusion_cull(f) && !no_second_pass) {
        // The lists of this frame were trimmed by the occlusion cuts of the previous one (gs_frame_layout.h).  If a tile ran
        // past its cut, counters[GS_CNT_RANPAST] is set and the launches below render the frame again from the full
        // lists; otherwise each of them returns at its first instruction (five gated launches: project + count -- the
        // first pass did not project the Gaussians behind every cut they could reach --, column scan, scatter, per-tile
        // sort, compositing).
        if ((rc = gs_stage_project(f, ws, s, 0, -1, true))) return rc;
        if ((rc = gs_stage_strip_bin(f, ws, s, true))) return rc;
        uint64_t *keys_out = nullptr;  // (frames that export their sorted keys are never culled)
        if ((rc = gs_stage_strip_sort(f, ws, okeys, skeys, keys_out, sids, s, true))) return rc;
        if ((rc = gs_stage_raster_forward(f, ws, sids, s, true))) return rc;
        tm.mark();
    }
    if (f->training && f->N > 0) prepare_on_side_stream(f, ws, sids, s);
    return tm.finish(stage_ms, GS_N_STAGES);
}

extern "C" int gs_frame_forward(const gs_frame *f, gs_stream_t stream) {
    return frame_forward_impl(f, (hipStream_t)stream, nullptr);
}

// How the project phase of this frame may be cut: *slices = number of slices of the Gaussian array (0: the frame's
// project stage cannot be issued in ranges -- only the strip variant with the fused count can), *per_slice = Gaussians per
// slice (a multiple of 256; the last slice may be short).
extern "C" int gs_frame_project_slices(const gs_frame *f, int32_t *slices, int64_t *per_slice) {
    int rc = validate(f);
    if (rc) return rc;
    GS_CHECK_ARG(slices && per_slice, "null pointer");
    *slices = 0;
    *per_slice = 0;
    if (f->N > 0 && effective_sort_mode(f) == 2 && gs_frame_fused_count(f)) {
        gs_frame_geom G = gs_frame_geometry(f);
        const gs_strip_plan plan = gs_strip_plan_for(f->N, G.ntx, G.nty);
        *slices = (int32_t)plan.slices;
        *per_slice = plan.per_slice;
    }
    return 0;
}

extern "C" int gs_frame_forward_project(const gs_frame *f, int32_t slice_begin, int32_t slice_end, gs_stream_t stream) {
    GS_CHECK_ARG(slice_begin >= 0 && slice_end > slice_begin, "bad slice range");
    return frame_forward_impl(f, (hipStream_t)stream, nullptr, 1, slice_begin, slice_end);
}

extern "C" int gs_frame_forward_rest(const gs_frame *f, gs_stream_t stream) {
    return frame_forward_impl(f, (hipStream_t)stream, nullptr, 2);
}

extern "C" int gs_frame_forward_profile(const gs_frame *f, float *stage_ms_host, gs_stream_t stream) {
    GS_CHECK_ARG(stage_ms_host != nullptr, "stage_ms_host is null");
    return frame_forward_impl(f, (hipStream_t)stream, stage_ms_host);
}

// part: 0 = the whole backward; GS_BWD_RASTER (1) = the raster backward only (per-pair gradient rows), then any of
// GS_BWD_GEOMETRY (2) = grad_pos / quat / scale and GS_BWD_COLOR (4) = grad_opa / rgb from those rows, in any order.
static int frame_backward_impl(const gs_frame *f, const float *grad_image, float *grad_pos, float *grad_quat,
                               float *grad_scale, float *grad_opa, float *grad_rgb, int part, hipStream_t s,
                               float *stage_ms, int64_t g_begin = 0, int64_t g_end = -1) {
    int rc = validate(f);
    if (rc) return rc;
    if (g_end < 0) g_end = f->N;
    GS_CHECK_ARG(g_begin >= 0 && g_begin <= g_end && g_end <= f->N && (g_begin & 255) == 0,
                 "bad Gaussian range (g_begin must be a multiple of 256)");
    GS_CHECK_ARG(f->training && f->image_padded, "gs_frame_backward needs a training forward (image_padded kept)");
    // (part -1, internal: the projection backward alone, both buckets in one kernel -- gs_frame_backward_slice)
    GS_CHECK_ARG(part == -1 || part == 0 || part == GS_BWD_RASTER || part == GS_BWD_GEOMETRY || part == GS_BWD_COLOR,
                 "part must be 0, GS_BWD_RASTER, GS_BWD_GEOMETRY or GS_BWD_COLOR");
    GS_CHECK_ARG(part == -1 || part == GS_BWD_GEOMETRY || part == GS_BWD_COLOR || grad_image, "null pointer");
    GS_CHECK_ARG(part == GS_BWD_RASTER || part == GS_BWD_COLOR || (grad_pos && grad_quat && grad_scale), "null pointer");
    GS_CHECK_ARG(part == GS_BWD_RASTER || part == GS_BWD_GEOMETRY || (grad_opa && grad_rgb), "null pointer");
    GS_CHECK_ARG(((uintptr_t)grad_quat & 15) == 0, "grad_quat must be 16-byte aligned");
    if (f->N == 0) return 0;
    gs_frame_ws ws = gs_frame_carve(f->workspace, f->N, f->max_pairs, f->width, f->height, f->color_dim, 1);
    uint64_t *skeys, *okeys;
    uint32_t *sids;
    sorted_buffers(f, ws, &skeys, &sids, &okeys);
    StageTimer tm(stage_ms != nullptr, s);
    tm.mark();
    if (part == 0 || part == GS_BWD_RASTER) {
        GS_CHECK_ARG(g_begin == 0 && g_end == f->N, "the raster backward has no Gaussian range");
        const bool prepared = join_prepared(f, s);
        if ((rc = gs_stage_raster_backward(f, ws, sids, grad_image, s, prepared))) return rc;
    }
    tm.mark();
    if (part != GS_BWD_RASTER &&
        (rc = gs_stage_project_backward(f, ws, grad_pos, grad_quat, grad_scale, grad_opa, grad_rgb,
                                        part == GS_BWD_GEOMETRY ? 1 : part == GS_BWD_COLOR ? 2 : 0, g_begin, g_end, s)))
        return rc;
    tm.mark();
    return tm.finish(stage_ms, 3);
}

extern "C" int gs_frame_backward(const gs_frame *f, const float *grad_image, float *grad_pos, float *grad_quat,
                                 float *grad_scale, float *grad_opa, float *grad_rgb, gs_stream_t stream) {
    return frame_backward_impl(f, grad_image, grad_pos, grad_quat, grad_scale, grad_opa, grad_rgb, 0,
                               (hipStream_t)stream, nullptr);
}

extern "C" int gs_frame_backward_adam(const gs_frame *f, const float *grad_image, const gs_adam_fused *adam,
                                      gs_stream_t stream) {
    int rc = validate(f);
    if (rc) return rc;
    GS_CHECK_ARG(f->training && f->image_padded, "gs_frame_backward_adam needs a training forward (image_padded kept)");
    GS_CHECK_ARG(grad_image && adam, "null pointer");
    if ((rc = gs_validate_adam_fused(f, adam))) return rc;  // before anything is enqueued
    if (f->N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    gs_frame_ws ws = gs_frame_carve(f->workspace, f->N, f->max_pairs, f->width, f->height, f->color_dim, 1);
    uint64_t *skeys, *okeys;
    uint32_t *sids;
    sorted_buffers(f, ws, &skeys, &sids, &okeys);
    const bool prepared = join_prepared(f, s);
    if ((rc = gs_stage_raster_backward(f, ws, sids, grad_image, s, prepared))) return rc;
    return gs_stage_project_backward_adam(f, ws, adam, s);
}

extern "C" int gs_frame_backward_part(const gs_frame *f, const float *grad_image, float *grad_pos, float *grad_quat,
                                      float *grad_scale, float *grad_opa, float *grad_rgb, int32_t part,
                                      gs_stream_t stream) {
    GS_CHECK_ARG(part == GS_BWD_RASTER || part == GS_BWD_GEOMETRY || part == GS_BWD_COLOR,
                 "part must be GS_BWD_RASTER, GS_BWD_GEOMETRY or GS_BWD_COLOR");
    return frame_backward_impl(f, grad_image, grad_pos, grad_quat, grad_scale, grad_opa, grad_rgb, part,
                               (hipStream_t)stream, nullptr);
}

// The per-Gaussian sums (projection + activation backward) of Gaussians [g_begin, g_end) only; `part` = GS_BWD_GEOMETRY,
// GS_BWD_COLOR or both.  After gs_frame_backward_part(GS_BWD_RASTER); any partition of [0, N) into ranges gives exactly
// what gs_frame_backward writes.
extern "C" int gs_frame_backward_slice(const gs_frame *f, float *grad_pos, float *grad_quat, float *grad_scale,
                                       float *grad_opa, float *grad_rgb, int32_t part, int64_t g_begin, int64_t g_end,
                                       gs_stream_t stream) {
    GS_CHECK_ARG(part == GS_BWD_GEOMETRY || part == GS_BWD_COLOR || part == (GS_BWD_GEOMETRY | GS_BWD_COLOR),
                 "part must be GS_BWD_GEOMETRY, GS_BWD_COLOR or both");
    if (part == (GS_BWD_GEOMETRY | GS_BWD_COLOR)) {  // everything of the range in one kernel: the rows are read once
        GS_CHECK_ARG(grad_pos && grad_quat && grad_scale && grad_opa && grad_rgb, "null pointer");
        return frame_backward_impl(f, nullptr, grad_pos, grad_quat, grad_scale, grad_opa, grad_rgb, -1,
                                   (hipStream_t)stream, nullptr, g_begin, g_end);
    }
    return frame_backward_impl(f, nullptr, grad_pos, grad_quat, grad_scale, grad_opa, grad_rgb, part,
                               (hipStream_t)stream, nullptr, g_begin, g_end);
}

extern "C" int gs_frame_backward_profile(const gs_frame *f, const float *grad_image, float *grad_pos,
                                         float *grad_quat, float *grad_scale, float *grad_opa, float *grad_rgb,
                                         float *stage_ms_host, gs_stream_t stream) {
    GS_CHECK_ARG(stage_ms_host != nullptr, "stage_ms_host is null");
    return frame_backward_impl(f, grad_image, grad_pos, grad_quat, grad_scale, grad_opa, grad_rgb, 0,
                               (hipStream_t)stream, stage_ms_host);
}

extern "C" int gs_frame_stats_async(const gs_frame *f, int64_t *stats_host, gs_stream_t stream) {
    int rc = validate(f);
    if (rc) return rc;
    GS_CHECK_ARG(stats_host != nullptr, "stats_host is null");
    gs_frame_ws ws = gs_frame_carve(f->workspace, f->N, f->max_pairs, f->width, f->height, f->color_dim, f->training);
    GS_HIP(hipMemcpyAsync(stats_host, ws.counters, sizeof(int64_t) * 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return 0;
}

extern "C" int gs_frame_overflow_flag(const gs_frame *f, const void **device_counter) {
    int rc = validate(f);
    if (rc) return rc;
    GS_CHECK_ARG(device_counter != nullptr, "device_counter is null");
    gs_frame_ws ws = gs_frame_carve(f->workspace, f->N, f->max_pairs, f->width, f->height, f->color_dim, f->training);
    *device_counter = ws.counters + GS_CNT_OVERFLOW;
    return 0;
}

extern "C" int gs_frame_longest_list_async(const gs_frame *f, int64_t *longest_host, gs_stream_t stream) {
    int rc = validate(f);
    if (rc) return rc;
    GS_CHECK_ARG(longest_host != nullptr, "longest_host is null");
    gs_frame_ws ws = gs_frame_carve(f->workspace, f->N, f->max_pairs, f->width, f->height, f->color_dim, f->training);
    GS_HIP(hipMemcpyAsync(longest_host, ws.counters + GS_CNT_MAXLIST, sizeof(int64_t), hipMemcpyDeviceToHost,
                          (hipStream_t)stream));
    return 0;
}

// The counters of the last forward (+ the last backward's preparation) on this workspace WITH the caller's tag behind them:
// the tag is written into the counter block by a fill command on `stream` (no host memory involved) and travels in the
// same device-to-host copy, so stats_host[11] == tag tells the host -- without any bookkeeping of its own -- that the
// eleven values in front of it are those the stream held when it reached THIS call, i.e. of the frame issued just before.
extern "C" int gs_frame_stats_tagged_async(const gs_frame *f, uint32_t tag, int64_t *stats_host, gs_stream_t stream) {
    int rc = validate(f);
    if (rc) return rc;
    GS_CHECK_ARG(stats_host != nullptr, "stats_host is null");
    gs_frame_ws ws = gs_frame_carve(f->workspace, f->N, f->max_pairs, f->width, f->height, f->color_dim, f->training);
    GS_HIP(hipMemsetD32Async((hipDeviceptr_t)(ws.counters + GS_CNT_TAG), (int)tag, 2, (hipStream_t)stream));
    GS_HIP(hipMemcpyAsync(stats_host, ws.counters, sizeof(int64_t) * GS_STATS_TAGGED_N, hipMemcpyDeviceToHost,
                          (hipStream_t)stream));
    return 0;
}

extern "C" int gs_frame_cull_fallback_async(const gs_frame *f, int64_t *ran_past_host, gs_stream_t stream) {
    int rc = validate(f);
    if (rc) return rc;
    GS_CHECK_ARG(ran_past_host != nullptr, "ran_past_host is null");
    gs_frame_ws ws = gs_frame_carve(f->workspace, f->N, f->max_pairs, f->width, f->height, f->color_dim, f->training);
    GS_HIP(hipMemcpyAsync(ran_past_host, ws.counters + GS_CNT_RANPAST, sizeof(int64_t), hipMemcpyDeviceToHost,
                          (hipStream_t)stream));
    return 0;
}

extern "C" int gs_frame_is_occlusion_culled(const gs_frame *f, int32_t *culled) {
    GS_CHECK_ARG(f && culled, "null argument");
    *culled = (effective_sort_mode(f) == 2 && gs_frame_occlusion_cull(f)) ? 1 : 0;
    return 0;
}

extern "C" int gs_frame_debug_tile_nproc(const gs_frame *f, const uint32_t **tile_nproc) {
    int rc = validate(f);
    if (rc) return rc;
    GS_CHECK_ARG(tile_nproc != nullptr, "tile_nproc is null");
    GS_CHECK_ARG(f->training, "the per-tile processed counts are kept by training forwards only");
    gs_frame_ws ws = gs_frame_carve(f->workspace, f->N, f->max_pairs, f->width, f->height, f->color_dim, 1);
    *tile_nproc = ws.tile_nproc;
    return 0;
}

int gs_bwd_mfma_slots(int color_dim, int n_tiles, int64_t max_buckets);  // raster_bwd.hip: (workgroup, wave) slots of the SH backward on the matrix pipe

extern "C" int gs_frame_debug_bwd_exec_rows(const gs_frame *f, const uint32_t **exec_rows, int32_t *n_slots) {
    int rc = validate(f);
    if (rc) return rc;
    GS_CHECK_ARG(exec_rows != nullptr && n_slots != nullptr, "null pointer");
    GS_CHECK_ARG(f->training, "the executed-row counters are kept by training frames only");
    gs_frame_ws ws = gs_frame_carve(f->workspace, f->N, f->max_pairs, f->width, f->height, f->color_dim, 1);
    *exec_rows = ws.bwd_exec_rows;
    *n_slots = gs_bwd_mfma_slots(f->color_dim, gs_frame_geometry(f).n_tiles, ws.max_buckets);
    return 0;
}

// Which binning / sort path gs_frame_forward takes for this frame description: 0 / 1 = sort_mode 0 / 1 (radix passes),
// 2 = sort_mode 2 table variant, 3 = slice-sorted variant, 4 = strip variant; negative: the description is invalid.
extern "C" int gs_frame_binning_variant(const gs_frame *f) {
    int rc = validate(f);
    if (rc) return rc < 0 ? rc : -rc;
    const int mode = effective_sort_mode(f);
    if (mode != 2) return mode;
    if (use_strip_variant(f)) return 4;
    return (f->flags & GS_FRAME_SLICE_SORT) ? 3 : 2;
}

extern "C" int gs_frame_debug_rects(const gs_frame *f, const uint32_t **rects) {
    int rc = validate(f);
    if (rc) return rc;
    GS_CHECK_ARG(rects != nullptr, "rects is null");
    gs_frame_ws ws = gs_frame_carve(f->workspace, f->N, f->max_pairs, f->width, f->height, f->color_dim, f->training);
    *rects = reinterpret_cast<const uint32_t *>(ws.rects);
    return 0;
}

extern "C" int gs_frame_debug_views(const gs_frame *f, const uint64_t **sorted_keys, const uint32_t **sorted_ids,
                                    const int32_t **tile_ranges, const float **rec_geom, const float **rec_cov,
                                    const float **rec_color, const uint32_t **tiles_touched) {
    int rc = validate(f);
    if (rc) return rc;
    gs_frame_ws ws = gs_frame_carve(f->workspace, f->N, f->max_pairs, f->width, f->height, f->color_dim, f->training);
    uint64_t *skeys, *okeys;
    uint32_t *sids;
    sorted_buffers(f, ws, &skeys, &sids, &okeys);
    if (sorted_keys)
        *sorted_keys = (effective_sort_mode(f) == 2 && !(f->flags & GS_FRAME_EMIT_SORTED_KEYS)) ? nullptr : skeys;
    if (sorted_ids) *sorted_ids = sids;
    if (tile_ranges) *tile_ranges = ws.tile_ranges;
    if (rec_geom) *rec_geom = (const float *)ws.rec_geom;
    if (rec_cov) *rec_cov = (const float *)ws.rec_cov;
    if (rec_color) *rec_color = (const float *)ws.rec_color;
    if (tiles_touched) *tiles_touched = ws.tiles_touched;
    return 0;
}

#ifdef GS_DIAG_CKPT
// diagnostic builds only (tools/dead_pixel_stats.py with a variant library): the forward's checkpoints
extern "C" __attribute__((visibility("default"))) int gs_diag_ckpt(const gs_frame *f, const float **ckpt,
                                                                   int64_t *max_buckets) {
    int rc = validate(f);
    if (rc) return rc;
    gs_frame_ws ws = gs_frame_carve(f->workspace, f->N, f->max_pairs, f->width, f->height, f->color_dim, 1);
    *ckpt = reinterpret_cast<const float *>(ws.ckpt);
    *max_buckets = ws.max_buckets;
    return 0;
}
#endif
