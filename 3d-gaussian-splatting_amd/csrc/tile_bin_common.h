// tile_bin_common.h -- device helpers shared by the table variant's binning kernels (tile_bin.hip) and the fused
// project + tile-count kernel of small scenes (cull_project.hip).
#pragma once
#include "gs_common.h"
#include "gs_frame_layout.h"

namespace {

#ifndef BIN_THREADS
#define BIN_THREADS 1024
#endif
#define BIN_SOLO 16  // rectangles up to this many tiles are walked by their own lane

// Walks the rectangle of every Gaussian of this workgroup's slice and calls fn(tile, gaussian, depth_bits).
// Small rectangles are handled by their own lane, large ones by the whole wave (one screen-filling
// Gaussian must not serialise 64 lanes behind it).
// DIST ("dist" tile culling): the rectangle is only the bounding square of the disc of listed tiles; a tile is
// listed iff gs_dist_listed says so for the Gaussian's centre (cxy).
template <bool DIST, typename Fn>
__device__ __forceinline__ void walk_rect(const uint4 rc, int64_t g, uint32_t ntx, float2 cxy, const GsDistCull &D,
                                          Fn fn) {
    const int lane = threadIdx.x & 63;
    const uint32_t cnt = rc.w, dbits = rc.z;
    const uint32_t y0 = rc.x & 0xffff, x0 = rc.y & 0xffff, x1 = rc.y >> 16;
    const uint32_t wdt = x1 - x0;
    if (cnt && cnt <= BIN_SOLO) {
        uint32_t ix = x0, iy = y0;
        for (uint32_t k = 0; k < cnt; ++k) {
            if (!DIST || gs_dist_listed(cxy.x, cxy.y, ix, iy, D)) fn(ix + iy * ntx, (uint32_t)g, dbits);
            if (++ix == x1) {
                ix = x0;
                ++iy;
            }
        }
    }
    unsigned long long big = __ballot(cnt > BIN_SOLO);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const uint32_t c = __shfl(cnt, src, 64), d = __shfl(dbits, src, 64);
        const uint32_t sx0 = __shfl(x0, src, 64), sy0 = __shfl(y0, src, 64), sw = __shfl(wdt, src, 64);
        const float spx = DIST ? __shfl(cxy.x, src, 64) : 0.f, spy = DIST ? __shfl(cxy.y, src, 64) : 0.f;
        const uint32_t id = (uint32_t)(g - lane + src);
        for (uint32_t k = lane; k < c; k += 64) {
            const uint32_t ix = sx0 + k % sw, iy = sy0 + k / sw;
            if (!DIST || gs_dist_listed(spx, spy, ix, iy, D)) fn(ix + iy * ntx, id, d);
        }
    }
}

// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2), while the output region of a
// tile is laid out in slice order.  Giving XCD x a CONTIGUOUS range of slices makes the 8-byte pair
// stores that fill one 128-byte line come from one L2 instead of eight, so lines are merged in L2
// instead of being written back as eight partial sectors.
__device__ __forceinline__ uint32_t slice_of_block(uint32_t blk, uint32_t B) {
    const uint32_t xcd = blk & 7, idx = blk >> 3;
    uint32_t first = 0;
    for (uint32_t x = 0; x < xcd; ++x) first += (B - x + 7) >> 3;  // workgroups that landed on XCD x
    return first + idx;
}

}  // namespace

// Gaussians per slice of the table variant: a multiple of 256 (the project stage's block) with at most GS_BIN_SLICES slices.
static inline uint32_t bin_per_block(int64_t N) {
    const int64_t per = gs_div_up(gs_div_up(N > 0 ? N : 1, GS_BIN_SLICES), 256) * 256;
    return (uint32_t)per;
}
