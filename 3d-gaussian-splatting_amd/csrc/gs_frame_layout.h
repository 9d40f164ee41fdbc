// gs_frame_layout.h -- workspace carving for the fused frame path (host side, shared by stages).
#pragma once
#include "gs_common.h"

enum { GS_CNT_VISIBLE = 0, GS_CNT_PAIRS = 1, GS_CNT_OVERFLOW = 2, GS_CNT_BUCKETS = 3, GS_CNT_TICKET = 4, GS_CNT_ENTRIES = 5, GS_CNT_BIG = 6, GS_CNT_SEGS = 7, GS_CNT_GROUPS = 8, GS_CNT_MAXLIST = 9, GS_CNT_RANPAST = 10, GS_CNT_TAG = 11, GS_CNT_EXCESS = 12, GS_CNT_MAXWALK = 13, GS_CNT_EXCESS_WALK = 14, GS_CNT_N = 16 };

#define GS_BUCKET 64          // Gaussians per backward bucket (= wavefront size)
#define GS_SORT_TILE 2048     // keys per radix-sort workgroup (256 threads x 8)
#define GS_BIN_SLICES 256     // sort_mode 2, table variant: workgroups (= slices of the Gaussian array) of the counting sort
#define GS_BIN_MAX_TILES 32768  // sort_mode 2 needs a 4-byte LDS counter per tile (128 KiB of the CU's 160)
#define GS_BIN_MAX_SLICES 1024  // sort_mode 2, slice-sorted variant: at most this many slices
#define GS_BIN_LDS_BYTES 160000 // dynamic LDS of a slice-sort workgroup (the CU has 160 KiB = 163,840 B)

// sort_mode 2 comes in two variants (tile_bin.hip):
//   slice-sorted (lds_sort = 1; opt-in per frame with GS_FRAME_SLICE_SORT): every workgroup counting-sorts the pairs of ITS slice of the Gaussian array by tile
//     inside LDS (T counters + `cap` staged 8-byte pairs) and streams them out as one contiguous, tile-ordered
//     region; the per-tile sort gathers a tile's pairs from the S slices.  No scattered global stores at all.
//   table (lds_sort = 0): count -> column scan -> scattered 8-byte stores; kept for tile grids whose counters leave
//     no room for the staging buffer (beyond ~31 k tiles, i.e. 4K images).
struct gs_bin_plan {
    int lds_sort;        // 1: slice-sorted variant
    uint32_t slices;     // S
    uint32_t per_slice;  // Gaussians per slice (a multiple of 256, the project stage's block)
    uint32_t cap;        // staged pairs that fit LDS next to the T counters (slices with more take the direct path)
};
static inline gs_bin_plan gs_bin_plan_for(int64_t N, int64_t max_pairs, int n_tiles, int want_lds_sort = 1) {
    gs_bin_plan p;
    const int64_t room = (int64_t)GS_BIN_LDS_BYTES - 4 * (int64_t)n_tiles;
    p.cap = room > 0 ? (uint32_t)(room / 8 / 64 * 64) : 0;
    p.lds_sort = want_lds_sort && p.cap >= 4096;
    const int64_t n = N > 0 ? N : 1;
    int64_t S = GS_BIN_SLICES;
    if (p.lds_sort) {
        // A slice-sort workgroup owns a CU's LDS, so the slices run in rounds of 256 (one per CU): their number is a
        // multiple of 256, the smallest one that keeps the average slice within 95 % of the staging buffer --
        // estimated from the capacity, which is >= the frame's pair count (a slice that does not fit still works,
        // through the direct path)
        const int64_t want = gs_div_up(max_pairs > 0 ? max_pairs : 1, (int64_t)p.cap * 95 / 100);
        S = gs_div_up(want, GS_BIN_SLICES) * GS_BIN_SLICES;
        if (S > GS_BIN_MAX_SLICES) S = GS_BIN_MAX_SLICES;
    }
    const int64_t per = gs_div_up(gs_div_up(n, S), 256) * 256;
    p.per_slice = (uint32_t)per;
    p.slices = (uint32_t)gs_div_up(n, per);
    return p;
}

// sort_mode 2, STRIP variant (strip_bin.hip; the default): a strip is GS_STRIP_W consecutive tiles of one tile row.
#ifndef GS_STRIP_W
#define GS_STRIP_W 8          // tiles per strip: 8 or 4 (3 + 3 bits of an entry name the covered tiles)
#endif
#define GS_STRIP_ID_BITS 26   // an entry keeps the Gaussian index in 26 bits: scenes beyond 2^26 take the table variant
#define GS_STRIP_MAX 8192     // strips per frame the LDS histograms are sized for
// Frames with long tile lists launch the extra kernels for them: the big-list sort (tile_sort.hip) and the segmented
// compositing of tiles that are still alive after GS_LONG_MIN Gaussians (raster_fwd.hip); other frames save those
// launches (~25 us of idle grids at 1080p).  The caller asks for them with GS_FRAME_LONG_LISTS, normally once the
// longest-list statistic of an earlier frame (gs_frame_longest_list_async) exceeded the LDS window; a frame WITHOUT the
// flag is still correct whatever its lists are (strip_sort_kernel sorts a long list itself, the tile's own wave
// composites all of it), only slower on such lists.  The CAPACITY of the workspace plays no part: sizing it up never
// changes the kernel path or the rounding.
#define GS_LONGEST_MIN 1024   // the longest-list statistic only reports lists beyond this (shorter ones read as 0)
// Where the serial walk of a FLAGGED frame's tile ends and how long a segment is.  Round 2 measured a single 100,000-Gaussian
// pile (4096 / 2048: 1.02 ms, 2048 / 1024: 0.59 ms).  Round 5 looked at the end state of a densifying run instead
// (profiles/r05_m_*: 724 k Gaussians, 3.96 M pairs, nothing saturates, 2,277 tiles beyond 512 Gaussians, 160 beyond 2,048): the
// compositing kernel lasts as long as its longest serial walk -- 2,048 Gaussians x 0.25 us (rgb) / 0.8 us (SH) per step of a
// lone wave -- while the device idles: 0.60 ms (rgb) / 1.82 ms (SH degree 2) where the work is worth 0.21 / 1.0 ms.  Same-box
// A/B (profiles/r05_n_*), 2048 / 1024 -> 512 / 512: pile forward 0.56 -> 0.38 ms (rgb), 1.51 -> 0.86 ms (SH); the densifying
// soak's last block +10 ... +15 %; 512 / 256 and 1024 / 512 measured within 5 % of it.  Only flagged frames are concerned
// (GS_FRAME_LONG_LISTS: the caller's rule -- gs_frame.py's cost model over longest list / pairs / pairs beyond 512, round 6;
// unflagged frames walk lists of any length with one wave by design, and lists beyond 2,048 take the big-list SORT alone,
// GS_FRAME_LONG_SORT); a tile that saturates before 512 Gaussians never continues.
#ifndef GS_LONG_MIN
#define GS_LONG_MIN 512       // Gaussians a tile's own wave composites before the rest of its list is cut into segments
#endif
#ifndef GS_SEG_LEN
#define GS_SEG_LEN 512        // Gaussians per segment
#endif
static inline bool gs_frame_long_lists(const gs_frame *f, int n_tiles) {
    (void)n_tiles;
    return (f->flags & GS_FRAME_LONG_LISTS) != 0;
}
// ---------------------------------------------------------------- temporal occlusion cull (round 6; GS_FRAME_OCCLUSION_CULL)
// 71 % of the pairs of the 2.4 M-Gaussian scene lie behind the point where their tile's pixels have all stopped: they are
// emitted, scattered and sorted (32 + 70 us of a 300-us frame) and never read.  The compositing kernel records, per tile,
// the depth of the last Gaussian it composited when ALL the tile's pixels had stopped before the end of its list (x (1 +
// GS_CUT_MARGIN); GS_NO_CUT otherwise), and the NEXT frame of the same workspace -- when the caller allows it with
// GS_FRAME_OCCLUSION_CULL -- trims every level-1 entry at the ends whose tiles' cut lies in front of the Gaussian
// (strip_common.h, walk_strips): those pairs are never emitted.  Exactness: a tile's list ascends in depth, the dropped
// pairs are deeper than every kept one, so a tile whose pixels all stop inside its kept prefix composites exactly what it
// would have composited from the full list -- bit for bit.  A tile that reaches the end of a trimmed list with a live pixel
// "ran past the cut" (the camera moved, something in front went away): it raises counters[GS_CNT_RANPAST], and the same
// launch sequence that follows the compositing -- count (from the rectangles, untrimmed), column scan, scatter, per-tile
// sort, compositing, each gated on that counter (a few microseconds of empty grids otherwise) -- renders the frame again
// from the full lists.  No host synchronisation, the image is exact either way.  Not for training frames (the backward
// owns the full emission order), frames that export their sorted keys, the "dist" listing, segmented long lists, or
// the table / radix variants.
// LDS a culled frame's level-1 kernels add to their strip tables: the cuts, every tile row padded to whole strips --
static inline size_t gs_cull_lds_bytes(int64_t ns) { return (size_t)ns * 4 * 8; }
// -- and, in the project stage (frame_project_cull_count_kernel), three more levels of the cut pyramid behind them
static inline size_t gs_cull_pyramid_bytes(int ntx, int nty) {
    const int nsx = (ntx + 7) / 8;  // (GS_STRIP_W, defined below)
    size_t b = (size_t)nsx * 8 * nty * 4;
    int w = ntx, h = nty;
    for (int l = 1; l < 4; ++l) {
        w = (w + 1) / 2, h = (h + 1) / 2;
        b += (size_t)w * h * 4;
    }
    return b;
}
#define GS_NO_CUT 0xffffffffu
#define GS_CUT_MARGIN 0.0625f
// GS_FRAME_CULL_DILATE (a camera that moved a little since the table was recorded): a tile's cut is the deepest cut of its
// 3 x 3 neighbourhood x 1.375.  A tile's stop depth is set by its deepest-seeing pixel (a ray through a gap between opaque
// Gaussians) and jumps when content shifts by a pixel; measured on the 2.4 M scene, 119-frame pans (profiles/r06_x_*):
// neighbourhood alone: 45 frames of a 1.25-px/frame pan fall back; x 1.25: 1; x 1.375: 0 (5 px/frame: 3, 12 px/frame: 4),
// with 3.1 M instead of 1.8 M of 6.95 M pairs emitted.
#define GS_CUT_DILATE_SCALE 1.375f
// GS_FRAME_CULL_DILATE_NEAR (within half a pixel): x 1.125 -- no fallback over a 0.25-px/frame pan, 2 in 119 frames at 1.25 px
#define GS_CUT_DILATE_SCALE_NEAR 1.125f
static inline bool gs_frame_occlusion_cull(const gs_frame *f);

// the sort half alone (GS_FRAME_LONG_SORT, round 6): lists beyond the LDS window go to big_list_sort_kernel; the segmented
// compositing -- which changes the rounding of the transmittance and only pays for lists several times longer -- follows
// GS_FRAME_LONG_LISTS
static inline bool gs_frame_long_sort(const gs_frame *f) {
    return (f->flags & (GS_FRAME_LONG_LISTS | GS_FRAME_LONG_SORT)) != 0;
}
static inline int64_t gs_seg_items_cap(int64_t max_pairs, int n_tiles) { return max_pairs / GS_SEG_LEN + n_tiles; }
static inline int64_t gs_group_queue_cap(int64_t max_pairs, int n_tiles) { return max_pairs / 256 + 4 * (int64_t)n_tiles; }
#define GS_STRIP_SORT_CAP 2048  // pairs strip_sort_kernel's LDS window holds (a half strip's four lists, or one list at a time)
struct gs_strip_geom {
    uint32_t ntx, nty, nsx, NS;  // tile grid, strips per tile row, strips per frame
};
struct gs_strip_plan {
    int ok;              // 0: the frame takes the table variant
    gs_strip_geom geom;
    uint32_t slices;     // workgroups of the level-1 kernels (= slices of the Gaussian array), <= GS_BIN_SLICES
    uint32_t per_slice;  // Gaussians per slice (a multiple of 256, the project stage's block)
    uint32_t cap;        // entries the scatter's LDS staging buffer holds
};
static inline gs_strip_plan gs_strip_plan_for(int64_t N, int ntx, int nty) {
    gs_strip_plan p;
    p.geom.ntx = (uint32_t)ntx;
    p.geom.nty = (uint32_t)nty;
    p.geom.nsx = (uint32_t)((ntx + GS_STRIP_W - 1) / GS_STRIP_W);
    p.geom.NS = p.geom.nsx * (uint32_t)nty;
    p.ok = N < (1ll << GS_STRIP_ID_BITS) && p.geom.NS <= GS_STRIP_MAX;
    const int64_t n = N > 0 ? N : 1;
    const int64_t per = gs_div_up(gs_div_up(n, GS_BIN_SLICES), 256) * 256;
    p.per_slice = (uint32_t)per;
    p.slices = (uint32_t)gs_div_up(n, per);
    p.cap = p.ok ? (uint32_t)(((int64_t)GS_BIN_LDS_BYTES - 8 * (int64_t)p.geom.NS) / 8) : 0;
    return p;
}

// sort_mode 2 runs the strip variant (strip_bin.hip) unless the caller asks for one of the others or the frame is outside
// its limits (2^26 Gaussians, GS_STRIP_MAX strips)
// Below GS_STRIP_AUTO_MIN_N Gaussians the frame is bound by the latency of its chain of dependent kernels, not by their
// throughput, and the table variant's chain is the shorter one (tools/sweep_n.py --table / --strip, profiles/r03_b_sweep_
// table_vs_strip.jsonl: 20.2 k against 16.1 k FPS at 10 k Gaussians, equal at 100 k, 9.4 k against 9.7 k at 200 k);
// GS_FRAME_STRIP_BIN overrides the choice.
#ifndef GS_STRIP_AUTO_MIN_N
#define GS_STRIP_AUTO_MIN_N 131072
#endif
static inline bool gs_frame_uses_strips(const gs_frame *f) {
    if (f->sort_mode != 2 || (f->flags & (GS_FRAME_SLICE_SORT | GS_FRAME_TABLE_BIN))) return false;
    const int ntx = (f->width + GS_TILE - 1) / GS_TILE, nty = (f->height + GS_TILE - 1) / GS_TILE;
    if (!gs_strip_plan_for(f->N, ntx, nty).ok) return false;
    if (f->flags & GS_FRAME_STRIP_BIN) return true;
    // the kernels for long tile lists (big-list sort, segmented compositing) belong to the strip variant: a frame that
    // asks for them gets it whatever its size (round 4: below GS_STRIP_AUTO_MIN_N the flag used to be ignored silently)
    if (f->flags & (GS_FRAME_LONG_LISTS | GS_FRAME_LONG_SORT)) return true;
    // small scenes take the table variant where it exists (one LDS counter per tile)
    return f->N >= GS_STRIP_AUTO_MIN_N || ntx * nty > GS_BIN_MAX_TILES;
}

// Strip variant: the strip-entry count (level 1a) is taken inside the project stage (cull_project.hip,
// frame_project_count_kernel) instead of by strip_count_kernel re-reading the rectangles.
#ifndef GS_FUSED_PROJECT_COUNT
#define GS_FUSED_PROJECT_COUNT 1  // A/B switch (tools/ab_variants.py)
#endif
static inline bool gs_frame_fused_count(const gs_frame *f) { return GS_FUSED_PROJECT_COUNT && gs_frame_uses_strips(f); }
static inline bool gs_frame_occlusion_cull(const gs_frame *f) {
    if (!((f->flags & GS_FRAME_OCCLUSION_CULL) && !f->training && f->N > 0 && f->tile_culling_method != 0 &&
          !(f->flags & (GS_FRAME_EMIT_SORTED_KEYS | GS_FRAME_LONG_LISTS | GS_FRAME_SERIAL_LONG_LISTS)) &&
          gs_frame_uses_strips(f) && GS_FUSED_PROJECT_COUNT))
        return false;
    // the level-1 kernels keep the per-tile cuts in LDS next to their strip tables (project + count: histogram, cut pyramid
    // and a queue of up to 16,384 survivors; scatter: cursors + cuts + a staging buffer worth having): 1080p = 70 KiB of
    // 160; beyond ~1,900 strips (~15 k tiles: 4K images) the frame is not culled
    const int ntx = (f->width + GS_TILE - 1) / GS_TILE, nty = (f->height + GS_TILE - 1) / GS_TILE;
    const int64_t ns = (int64_t)((ntx + GS_STRIP_W - 1) / GS_STRIP_W) * nty;
    const int64_t room = (int64_t)GS_BIN_LDS_BYTES - 8 * 4096;
    return 8 * ns + (int64_t)gs_cull_pyramid_bytes(ntx, nty) + 2 * 16384 + 16 <= room &&
           8 * ns + (int64_t)gs_cull_lds_bytes(ns) + 8 * 4096 <= room;
}
// Table variant (small scenes: a frame is a chain of dependent launches of ~7 us each): the per-(slice, tile) count
// (bin_count_kernel) is taken inside the project stage as well -- five launches per frame instead of six.
static inline bool gs_frame_fused_table_count(const gs_frame *f) {
    const int ntx = (f->width + GS_TILE - 1) / GS_TILE, nty = (f->height + GS_TILE - 1) / GS_TILE;
    return GS_FUSED_PROJECT_COUNT && f->sort_mode == 2 && !gs_frame_uses_strips(f) && !(f->flags & GS_FRAME_SLICE_SORT) &&
           ntx * nty <= GS_BIN_MAX_TILES;
}

// floats per (tile, Gaussian) gradient row: 7 geometry/opacity sums + color_dim colour sums, in whole 64-byte lines (16 /
// 48 / 64 floats for color_dim 3 / 27 / 48).  rgb colours (10 floats): the row is ONE aligned 64-byte line (16 floats), which
// the raster backward writes with a single store instruction (four lanes x 16 bytes) -- a 48-byte row straddled two
// lines every other time and reached HBM as 32-byte partial writes (round 3, PMC: 134 B written per 48-byte row).
// SH rows (round 5): whole 64-byte lines, every line written completely by ONE store instruction (four lanes x 16 bytes, as
// the rgb rows since round 4).  Round 4 wrote a 144 / 224-byte row as per-lane scalars at a 28-byte offset into rows that
// straddle lines: PMC WRITE_SIZE 2.2 x / 1.32 x the row bytes, plus a flag byte (a 32-byte write) per row.  The raster
// backward on the matrix pipe holds a row spread over the four lanes (g, jq) of its Gaussian -- lane (g, jq) has the
// coefficient sums k = 4 jq .. 4 jq + 3 of every channel -- so the row is laid out by LINES of four 16-byte pieces, piece q
// of a line coming from lane (g, q):
//   degree 2 (27 coefficients, 9 per channel), 48 floats = 3 lines:  line ch = [coef(ch, 0..11: 9 sums + 3 zeros) | header piece ch]
//       header pieces: 0 = (dx, dy, da, db), 1 = (dc, dd, dopa, 0), 2 = zeros   (lane (g, 3) has no coefficient: it carries them)
//   degree 3 (48 coefficients, 16 per channel), 64 floats = 4 lines: line 0 = header (dx, dy, da, db | dc, dd, dopa, 0 | 0 | 0),
//       line 1 + ch = coef(ch, 0..15)
//   rgb: 16 floats = 1 line: (dx, dy, da, db, dc, dd, dopa, dr, dg, db, 0 ...)
// gs_row_geo / gs_row_col give the position of a row's floats; every writer and reader goes through them.
static constexpr int gs_row_floats(int color_dim) {
    return color_dim == 3 ? 16 : color_dim == 27 ? 48 : color_dim == 48 ? 64 : (7 + color_dim + 3) / 4 * 4;
}
// m = 0..6: dL/d(x, y, a, b, c, d, opacity) of the pair
static constexpr int gs_row_geo(int color_dim, int m) { return color_dim == 27 ? (m < 4 ? 12 + m : 28 + (m - 4)) : m; }
// colour float c (rgb: channel; SH: coefficient ch * NB + k, the parameter's own order)
static constexpr int gs_row_col(int color_dim, int c) {
    return color_dim == 27 ? 16 * (c / 9) + c % 9 : color_dim == 48 ? 16 + c : 7 + c;
}
// inverse: float f of a row -> its index in the compact order (dx, dy, da, db, dc, dd, dopa, colour 0 .. color_dim - 1), or -1
// for a padding float
static constexpr int gs_row_compact(int color_dim, int f) {
    if (color_dim == 27) {
        const int line = f / 16, o = f % 16;
        if (line > 2) return -1;
        if (o < 9) return 7 + 9 * line + o;
        if (o < 12) return -1;
        return line == 0 ? o - 12 : (line == 1 && o < 15) ? 4 + (o - 12) : -1;
    }
    if (color_dim == 48) return f < 7 ? f : f < 16 ? -1 : f < 64 ? 7 + (f - 16) : -1;
    return f < 7 + color_dim ? f : -1;
}
// executed pixel-row steps of the SH backward on the matrix pipe, one counter per (workgroup, wave): what its MFMA flops
// are counted from (bench.py); [GS_BWD_EXEC_SLOTS x T] u32
#define GS_BWD_EXEC_SLOTS 8
// SH backward on the matrix pipe: buckets per work item (raster_bwd.hip: mfma_items_kernel); the item list holds at most
// T + max_buckets / GS_MFMA_ITEM_BUCKETS items
#define GS_MFMA_ITEM_BUCKETS 8

struct gs_frame_geom {
    int padW, padH, ntx, nty, n_tiles, crop_top, crop_left;
    float tlx, tly, leftmost, topmost;
};

// Integer geometry of splatter.Tiles (splatter.py:259-272).
static inline gs_frame_geom gs_frame_geometry_i(int W, int H) {
    gs_frame_geom G;
    G.padW = (W + GS_TILE - 1) / GS_TILE * GS_TILE;
    G.padH = (H + GS_TILE - 1) / GS_TILE * GS_TILE;
    G.ntx = G.padW / GS_TILE;
    G.nty = G.padH / GS_TILE;
    G.n_tiles = G.ntx * G.nty;
    G.crop_top = (G.padH - H) / 2;
    G.crop_left = (G.padW - W) / 2;
    G.tlx = G.tly = G.leftmost = G.topmost = 0.f;
    return G;
}
static inline gs_frame_geom gs_frame_geometry(const gs_frame *f) {
    gs_frame_geom G = gs_frame_geometry_i(f->width, f->height);
    G.tlx = f->tile_length_x;
    G.tly = f->tile_length_y;
    G.leftmost = f->leftmost;
    G.topmost = f->topmost;
    return G;
}

struct gs_frame_ws {
    unsigned long long *counters;  // [GS_CNT_N]
    // one 64-byte record per Gaussian (GS_REC_STRIDE float4s, see gs_common.h); the four pointers
    // address the four float4 fields of record 0, index with [i * GS_REC_STRIDE]
    float4 *rec_geom;              // (x, y, depth, opacity)
    float4 *rec_cov;               // (a, b, c, d)
    float4 *rec_color;             // (r, g, b, -)
    float4 *rec_conic;             // (A, B, C, -)
    uint32_t *tiles_touched;       // [N]
    uint4 *rects;                  // [N] (y0 | y1 << 16, x0 | x1 << 16, depth bits, tiles touched): all the binning needs
    uint4 *surv;                   // [N] inference workspaces: compacted survivor rectangles of an occlusion-culled frame
    uint32_t *slice_nsurv;         // [slices] their count per slice
    uint32_t *block_sums;          // [ceil(N/256)] pairs emitted by each 256-Gaussian block
    uint32_t *block_vis;           // [ceil(N/256)] visible Gaussians of each block
    uint32_t *block_offsets;       // [ceil(N/256)]
    uint32_t *pair_offsets;        // [N] emission offset of each Gaussian's first pair (prefix sum of tiles_touched)
    uint64_t *keys_a, *keys_b;     // [max_pairs]
    uint32_t *vals_a, *vals_b;     // [max_pairs]
    void *sort_tmp;
    size_t sort_tmp_bytes;
    int32_t *tile_ranges;          // [T][2]
    uint32_t *cut;                 // [T] occlusion cut of every tile (GS_FRAME_OCCLUSION_CULL, below): depth bits behind which the
    uint32_t *cut_dilated;         // [T] GS_FRAME_CULL_DILATE: the largest cut of every tile's 3 x 3 neighbourhood (written per frame)
                                   // LAST forward of this workspace composited nothing in the tile, GS_NO_CUT if its pixels
                                   // had not all stopped; written by every INFERENCE frame-path compositing launch
    // sort_mode 2 (tile_bin.hip)
    uint32_t *bin_table;           // table variant: [GS_BIN_SLICES][T] pairs of (slice, tile), scanned in place over
                                   // slices; slice-sorted variant: [S][T + 1] offsets of a tile's pairs inside the
                                   // slice's region (last entry: the slice's pair count)
    uint32_t *tile_count;          // [T]
    uint32_t *slice_pairs, *slice_vis;  // [GS_BIN_MAX_SLICES]; slice-sorted variant: slice_pairs = start of the
                                        // slice's region in keys_a
    // sort_mode 2, strip variant (strip_bin.hip): packed (entries << 32 | pairs) per (slice, strip)
    uint64_t *strip_table;         // [2][GS_BIN_SLICES][NS]: raw counts, then their exclusive scan over the slices
    uint64_t *strip_tot;           // [NS] totals per strip
    uint64_t *strip_base;          // [NS] (first entry << 32 | first pair) of every strip
    uint32_t *big_tiles;           // [T] queue of the tiles whose list exceeds strip_sort_kernel's LDS window
    uint32_t *tile_cost;           // [T] Gaussian steps every tile composited in the LAST forward of this workspace
    uint32_t *tile_order;          // [T] tiles in descending order of that cost: the dispatch order of this forward
    uint4 *group_queue;            // dense frames: (tile, first slot, keys, -) of the groups big_list_sort_kernel cut
    // dense frames only (else NULL): segmented compositing of long tile lists (raster_fwd.hip)
    float4 *cont_state;            // [T][256] (T, C) of a tile's pixels after its first GS_LONG_MIN Gaussians
    uint32_t *cont_flag;           // [T] 1: the tile was still alive there and continues in segments
    uint32_t *seg_item_base;       // [T + 1] first segment item of every tile
    uint2 *seg_items;              // [items_cap] (tile, segment index)
    float *seg_P;                  // [items_cap][256] transmittance product of a segment
    float4 *seg_C;                 // [items_cap][256] (-, C) composited by a segment from its incoming transmittance
    uint32_t *seg_nproc;           // [items_cap] Gaussians a segment composited before all pixels had stopped
    // training only
    uint32_t *tile_nproc;          // [T] Gaussians processed by the forward (multiple of the chunk)
    uint32_t *bucket_offsets;      // [T+1] exclusive scan of ceil(nproc/64)
    uint4 *bucket_info;            // [max_buckets + 8] (tile, first Gaussian, count, list start) per bucket
    float4 *ckpt;                  // [max_buckets][256] (T, Cr, Cg, Cb) at bucket starts
    float *rows;                   // [max_pairs][GS_ROW(C)] per-pair gradient rows in EMISSION order
    uint4 *mfma_items;             // SH frames: [T + max_buckets / 8 + 8] work items (tile, first bucket, buckets <= 8, -) of the SH
    uint32_t *mfma_n_items;        //            backward on the matrix pipe, heavy tiles first, and their count (raster_bwd.hip)
    uint32_t *bwd_exec_rows;       // [GS_BWD_EXEC_SLOTS x T] SH backward on the matrix pipe: pixel-row steps (16 Gaussians x 16
                                   // pixels) every wave executed in the last backward (rows whose pixels had all stopped are
                                   // left out); slot = workgroup x waves + wave.  Diagnostic (bench.py's MFMA flop count).
    uint64_t *stop_keys;           // 2 x [T] u32: depth bits, then Gaussian index, of the LAST list entry the forward
                                   // processed in each tile (depth 0: none).  A tile's list ascends in exactly this key, so the pair
                                   // (tile, g) was processed -- its gradient row written -- iff key(g) <= stop_keys[tile]:
                                   // no reader needs a per-row flag (stop_key_kernel, raster_bwd.hip; rgb rows since round 4,
                                   // SH rows since round 5: the rows themselves are never zero-filled, unwritten ones are
                                   // never read)
    int64_t max_buckets;
    size_t zero_bytes;             // prefix of the workspace cleared at the start of every frame
    size_t total_bytes;
};

static inline int64_t gs_max_buckets(int64_t max_pairs, int n_tiles) {
    return max_pairs / GS_BUCKET + n_tiles + 1;
}

// Carves `base` (may be NULL to only compute the size).
static inline gs_frame_ws gs_frame_carve(void *base, int64_t N, int64_t max_pairs, int W, int H, int color_dim,
                                         int training) {
    gs_frame_ws ws;
    gs_frame_geom G = gs_frame_geometry_i(W, H);
    size_t off = 0;
    auto take = [&](size_t bytes) -> void * {
        void *p = base ? (void *)((char *)base + off) : nullptr;
        off += gs_align_up(bytes ? bytes : 1, 256);
        return p;
    };
    const int64_t nblk = gs_div_up(N > 0 ? N : 1, 256);
    // counters and tile_ranges are adjacent: ONE memset per frame clears both (empty tiles must read (0,0))
    ws.counters = (unsigned long long *)take(sizeof(unsigned long long) * GS_CNT_N);
    ws.tile_ranges = (int32_t *)take(sizeof(int32_t) * 2 * G.n_tiles);
    ws.zero_bytes = off;
    // (in front of everything whose size depends on N: the table outlives a change of the Gaussian count)
    ws.cut = (uint32_t *)take(sizeof(uint32_t) * G.n_tiles);
    ws.cut_dilated = (uint32_t *)take(sizeof(uint32_t) * G.n_tiles);
    ws.rec_geom = (float4 *)take(sizeof(float4) * GS_REC_STRIDE * N);
    ws.rec_cov = ws.rec_geom ? ws.rec_geom + 1 : nullptr;
    ws.rec_color = ws.rec_geom ? ws.rec_geom + 2 : nullptr;
    ws.rec_conic = ws.rec_geom ? ws.rec_geom + 3 : nullptr;
    ws.tiles_touched = (uint32_t *)take(sizeof(uint32_t) * N);
    ws.rects = (uint4 *)take(sizeof(uint4) * N);
    // occlusion-culled inference frames (first pass): the rectangles of the Gaussians that were projected AND touch a tile,
    // compacted per slice of the Gaussian array -- (y range, x range, depth bits, Gaussian) at [slice x per_slice, +
    // slice_nsurv[slice]) -- what the level-1 scatter of such a frame reads instead of all N rectangle records
    ws.surv = (uint4 *)take(training ? 0 : sizeof(uint4) * N);
    ws.slice_nsurv = (uint32_t *)take(sizeof(uint32_t) * GS_BIN_MAX_SLICES);
    ws.block_sums = (uint32_t *)take(sizeof(uint32_t) * nblk);
    ws.block_vis = (uint32_t *)take(sizeof(uint32_t) * nblk);
    ws.block_offsets = (uint32_t *)take(sizeof(uint32_t) * nblk);
    ws.pair_offsets = (uint32_t *)take(sizeof(uint32_t) * N);
    ws.keys_a = (uint64_t *)take(sizeof(uint64_t) * max_pairs);
    ws.keys_b = (uint64_t *)take(sizeof(uint64_t) * max_pairs);
    ws.vals_a = (uint32_t *)take(sizeof(uint32_t) * max_pairs);
    ws.vals_b = (uint32_t *)take(sizeof(uint32_t) * max_pairs);
    ws.sort_tmp_bytes = gs_sort_pairs_tmp_bytes(max_pairs);
    ws.sort_tmp = take(ws.sort_tmp_bytes);
    {
        // sized for whichever variant the frame selects (GS_FRAME_SLICE_SORT is a per-frame flag)
        const gs_bin_plan plan = gs_bin_plan_for(N, max_pairs, G.n_tiles);
        const size_t rows = plan.lds_sort && plan.slices > GS_BIN_SLICES ? plan.slices : GS_BIN_SLICES;
        ws.bin_table = (uint32_t *)take(sizeof(uint32_t) * rows * ((size_t)G.n_tiles + 1));
    }
    ws.tile_count = (uint32_t *)take(sizeof(uint32_t) * G.n_tiles);
    ws.slice_pairs = (uint32_t *)take(sizeof(uint32_t) * GS_BIN_MAX_SLICES);
    ws.slice_vis = (uint32_t *)take(sizeof(uint32_t) * GS_BIN_MAX_SLICES);
    {
        const gs_strip_plan sp = gs_strip_plan_for(N, G.ntx, G.nty);
        const size_t ns = sp.ok ? sp.geom.NS : 1;
        ws.strip_table = (uint64_t *)take(sizeof(uint64_t) * 2 * GS_BIN_SLICES * ns);
        ws.strip_tot = (uint64_t *)take(sizeof(uint64_t) * ns);
        ws.strip_base = (uint64_t *)take(sizeof(uint64_t) * ns);
        ws.big_tiles = (uint32_t *)take(sizeof(uint32_t) * (sp.ok ? (size_t)G.n_tiles : 1));
        ws.tile_cost = (uint32_t *)take(sizeof(uint32_t) * (sp.ok ? (size_t)G.n_tiles : 1));
        ws.tile_order = (uint32_t *)take(sizeof(uint32_t) * (sp.ok ? (size_t)G.n_tiles : 1));
    }
    if (gs_strip_plan_for(N, G.ntx, G.nty).ok) {  // the long-list kernels belong to the strip variant
        const size_t cap = (size_t)gs_seg_items_cap(max_pairs, G.n_tiles);
        ws.group_queue = (uint4 *)take(sizeof(uint4) * (size_t)gs_group_queue_cap(max_pairs, G.n_tiles));
        ws.cont_state = (float4 *)take(sizeof(float4) * 256 * (size_t)G.n_tiles);
        ws.cont_flag = (uint32_t *)take(sizeof(uint32_t) * G.n_tiles);
        ws.seg_item_base = (uint32_t *)take(sizeof(uint32_t) * ((size_t)G.n_tiles + 1));
        ws.seg_items = (uint2 *)take(sizeof(uint2) * cap);
        ws.seg_P = (float *)take(sizeof(float) * 256 * cap);
        ws.seg_C = (float4 *)take(sizeof(float4) * 256 * cap);
        ws.seg_nproc = (uint32_t *)take(sizeof(uint32_t) * cap);
    } else {
        ws.group_queue = nullptr;
        ws.cont_state = nullptr;
        ws.cont_flag = nullptr;
        ws.seg_item_base = nullptr;
        ws.seg_items = nullptr;
        ws.seg_P = nullptr;
        ws.seg_C = nullptr;
        ws.seg_nproc = nullptr;
    }
    ws.max_buckets = gs_max_buckets(max_pairs, G.n_tiles);
    if (training) {
        ws.tile_nproc = (uint32_t *)take(sizeof(uint32_t) * G.n_tiles);
        ws.bucket_offsets = (uint32_t *)take(sizeof(uint32_t) * (G.n_tiles + 1));
        ws.bucket_info = (uint4 *)take(sizeof(uint4) * (size_t)(ws.max_buckets + 8));
        ws.ckpt = (float4 *)take(sizeof(float4) * 256 * (size_t)ws.max_buckets);
        ws.rows = (float *)take(sizeof(float) * (size_t)gs_row_floats(color_dim) * max_pairs);
        ws.bwd_exec_rows = (uint32_t *)take(sizeof(uint32_t) * GS_BWD_EXEC_SLOTS * (size_t)G.n_tiles);
        ws.mfma_items = (uint4 *)take(sizeof(uint4) * (color_dim == 3 ? 1 : (size_t)(G.n_tiles + ws.max_buckets / GS_MFMA_ITEM_BUCKETS + 8)));
        ws.mfma_n_items = (uint32_t *)take(sizeof(uint32_t));
        ws.stop_keys = (uint64_t *)take(sizeof(uint64_t) * G.n_tiles);
    } else {
        ws.tile_nproc = nullptr;
        ws.bucket_offsets = nullptr;
        ws.bucket_info = nullptr;
        ws.ckpt = nullptr;
        ws.rows = nullptr;
        ws.bwd_exec_rows = nullptr;
        ws.mfma_items = nullptr;
        ws.mfma_n_items = nullptr;
        ws.stop_keys = nullptr;
    }
    ws.total_bytes = off;
    return ws;
}

// stage entry points (defined across the .hip files)
// the cut table a culled frame trims its lists by: the tiles' own cuts, or (GS_FRAME_CULL_DILATE) their neighbourhood maxima
static inline const uint32_t *gs_frame_cut_table(const gs_frame *f, const gs_frame_ws &ws) {
    return (f->flags & GS_FRAME_CULL_DILATE) ? ws.cut_dilated : ws.cut;
}
int gs_stage_cut_dilate(const gs_frame *f, const gs_frame_ws &ws, hipStream_t stream);
int gs_stage_project(const gs_frame *f, const gs_frame_ws &ws, hipStream_t stream, int slice_begin = 0, int slice_end = -1,
                     bool second_pass = false);
int gs_stage_project_backward(const gs_frame *f, const gs_frame_ws &ws, float *grad_pos, float *grad_quat,
                              float *grad_scale, float *grad_opa, float *grad_rgb, int part, int64_t g_begin,
                              int64_t g_end, hipStream_t stream);
int gs_stage_project_backward_adam(const gs_frame *f, const gs_frame_ws &ws, const gs_adam_fused *a, hipStream_t stream);
int gs_validate_adam_fused(const gs_frame *f, const gs_adam_fused *a);
int gs_stage_scan_emit(const gs_frame *f, const gs_frame_ws &ws, hipStream_t stream);
int gs_stage_tile_sort(const gs_frame *f, const gs_frame_ws &ws, uint64_t *keys, uint32_t *ids, uint64_t *scratch,
                       hipStream_t stream);
int gs_stage_tile_bin(const gs_frame *f, const gs_frame_ws &ws, hipStream_t stream);
int gs_stage_strip_bin(const gs_frame *f, const gs_frame_ws &ws, hipStream_t stream, bool second_pass = false);
int gs_stage_strip_sort(const gs_frame *f, const gs_frame_ws &ws, const uint64_t *entries, uint64_t *scratch,
                        uint64_t *keys_out, uint32_t *ids_out, hipStream_t stream, bool second_pass = false);
int gs_stage_tile_sort_packed(const gs_frame *f, const gs_frame_ws &ws, uint64_t *packed, uint64_t *keys_out,
                              uint32_t *ids_out, hipStream_t stream);
int gs_stage_tile_sort_gather(const gs_frame *f, const gs_frame_ws &ws, const uint64_t *slice_pairs_buf,
                              uint64_t *big_scratch, uint64_t *keys_out, uint32_t *ids_out, hipStream_t stream);
int gs_stage_tile_ranges(const gs_frame *f, const gs_frame_ws &ws, const uint64_t *sorted_keys, hipStream_t stream);
int gs_stage_raster_forward(const gs_frame *f, const gs_frame_ws &ws, const uint32_t *sorted_ids, hipStream_t stream,
                            bool second_pass = false);
int gs_stage_backward_prepare(const gs_frame *f, const gs_frame_ws &ws, const uint32_t *sorted_ids, hipStream_t stream);
int gs_stage_sh_big_rows(const gs_frame *f, const gs_frame_ws &ws, hipStream_t stream);
int gs_stage_raster_backward(const gs_frame *f, const gs_frame_ws &ws, const uint32_t *sorted_ids,
                             const float *grad_image, hipStream_t stream, bool prepared);
