// strip_common.h -- device helpers shared by the level-1 kernels of the strip variant (strip_bin.hip) and the fused
// project + strip-count kernel (cull_project.hip).
#pragma once
#include "gs_common.h"
#include "gs_frame_layout.h"

namespace {

#define STRIP_THREADS 1024
#ifndef STRIP_SOLO
#define STRIP_SOLO 16  // Gaussians with up to this many entries are walked by their own lane (4: +15 us, 8: +2 us at 2.4 M)
#endif
#ifndef STRIP_PF
#define STRIP_PF 2     // rectangles in flight per thread (4 left a quarter of the visits of a 9,472-Gaussian slice empty: +7 us)
#endif

// Calls fn(strip, lo32, depth_bits, pairs) for every strip entry of one Gaussian per lane.  lo32 = first covered tile of
// the strip << 29 | last covered tile << 26 | gaussian; pairs = listed tiles of the run.  DIST: a tile of the
// bounding square is listed iff gs_dist_listed says so; runs without a listed tile emit nothing (the same decision
// in the count and in the scatter pass, and in strip_sort_kernel).
// CUT (GS_FRAME_OCCLUSION_CULL, gs_frame_layout.h): `cut8` is the per-tile cut table staged in LDS with every tile row padded
// to whole strips (row stride nsx x GS_STRIP_W), so that the eight cuts of a strip are two aligned 16-byte reads.  The run
// is trimmed at both ends by the tiles whose cut depth lies in front of the Gaussian -- pairs the previous frame of this
// workspace proved to be behind the point where their tile's pixels have all stopped; a run that loses all its tiles emits
// nothing.  The same table in the count and in the scatter pass; strip_sort_kernel takes the run from the entry.
// (First version: a generic pointer and a data-dependent loop per end -- flat loads, +45 us in the scatter at 2.4 M Gaussians.)
template <bool DIST, bool CUT = false, typename Fn>
__device__ __forceinline__ void walk_strips(const uint4 rc, int64_t g, const gs_strip_geom SG, float2 cxy,
                                            const GsDistCull &D, Fn fn, const uint32_t *cut8 = nullptr) {
    static_assert(!CUT || GS_STRIP_W == 8, "the cut table is read eight tiles at a time");
    static_assert(!CUT || !DIST, "the occlusion cull is not combined with the \"dist\" listing");
    const int lane = threadIdx.x & 63;
    const uint32_t y0 = rc.x & 0xffff, y1 = rc.x >> 16, x0 = rc.y & 0xffff, x1 = rc.y >> 16, dbits = rc.z;
    const bool vis = rc.w != 0;
    const uint32_t sx0 = x0 / GS_STRIP_W, span = vis ? (x1 - 1) / GS_STRIP_W - sx0 + 1 : 0;
    const uint32_t ne = span * (y1 - y0);
    auto emit = [&](uint32_t sx, uint32_t iy, uint32_t ex0, uint32_t ex1, uint32_t id, uint32_t d, float px, float py) {
        const uint32_t t0 = sx * GS_STRIP_W;
        uint32_t lo = ex0 > t0 ? ex0 - t0 : 0, hi = (ex1 < t0 + GS_STRIP_W ? ex1 : t0 + GS_STRIP_W) - t0;  // [lo, hi)
        if (CUT) {
            const uint4 *c = reinterpret_cast<const uint4 *>(cut8 + (size_t)(iy * SG.nsx + sx) * GS_STRIP_W);
            const uint4 c0 = c[0], c1 = c[1];
            // bit j: tile j of the strip keeps the Gaussian (its depth does not lie behind the tile's cut)
            uint32_t keep = (d <= c0.x ? 1u : 0u) | (d <= c0.y ? 2u : 0u) | (d <= c0.z ? 4u : 0u) | (d <= c0.w ? 8u : 0u) |
                            (d <= c1.x ? 16u : 0u) | (d <= c1.y ? 32u : 0u) | (d <= c1.z ? 64u : 0u) | (d <= c1.w ? 128u : 0u);
            keep &= ((1u << hi) - 1u) & ~((1u << lo) - 1u);
            if (!keep) return;
            lo = (uint32_t)__ffs((int)keep) - 1u;
            hi = 32u - (uint32_t)__clz((int)keep);
        }
        uint32_t np = hi - lo;
        if (DIST) {
            np = 0;
            for (uint32_t x = lo; x < hi; ++x) np += gs_dist_listed(px, py, t0 + x, iy, D) ? 1u : 0u;
            if (!np) return;
        }
        fn(iy * SG.nsx + sx, (lo << 29) | ((hi - 1) << 26) | id, d, np);
    };
    if (ne && ne <= STRIP_SOLO) {
        uint32_t sx = sx0, iy = y0;
        for (uint32_t k = 0; k < ne; ++k) {
            emit(sx, iy, x0, x1, (uint32_t)g, dbits, cxy.x, cxy.y);
            if (++sx == sx0 + span) {
                sx = sx0;
                ++iy;
            }
        }
    }
    unsigned long long big = __ballot(ne > STRIP_SOLO);
    while (big) {  // a Gaussian that crosses many strips is walked by the whole wave
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const uint32_t c = __shfl(ne, src, 64), d = __shfl(dbits, src, 64), sp = __shfl(span, src, 64);
        const uint32_t bsx0 = __shfl(sx0, src, 64), by0 = __shfl(y0, src, 64);
        const uint32_t bx0 = __shfl(x0, src, 64), bx1 = __shfl(x1, src, 64);
        const float spx = DIST ? __shfl(cxy.x, src, 64) : 0.f, spy = DIST ? __shfl(cxy.y, src, 64) : 0.f;
        // (a culled frame's survivors are not consecutive Gaussians: their ids are shuffled; everybody else's follow from the lane
        // -- one ds_bpermute less per wave-walked Gaussian: the scatter of an unculled 2.4 M frame measured 43.2 against 41.6 us)
        const uint32_t id = CUT ? (uint32_t)__shfl((int)(uint32_t)g, src, 64) : (uint32_t)(g - lane + src);
        for (uint32_t k = lane; k < c; k += 64) emit(bsx0 + k % sp, by0 + k / sp, bx0, bx1, id, d, spx, spy);
    }
}

// same XCD-contiguous dealing of slices to workgroups as the table variant (tile_bin.hip): the runs of neighbouring
// slices are neighbours in memory, so the partial lines at run boundaries meet in one L2
__device__ __forceinline__ uint32_t strip_slice_of_block(uint32_t blk, uint32_t B) {
    const uint32_t xcd = blk & 7, idx = blk >> 3;
    uint32_t first = 0;
    for (uint32_t x = 0; x < xcd; ++x) first += (B - x + 7) >> 3;
    return first + idx;
}

// ---------------------------------------------------------------- compositing order of the tiles (longest first)
// One tile per wave is serial in the tile's list, and tiles differ by several times in the Gaussians they composite
// before all their pixels are saturated: dispatched in raster order, the long tiles that happen to start late leave most
// SIMDs idle at the end of raster_forward_kernel (PMC, round 2: 68 % of the issue cycles busy).  The forward kernel
// therefore records what every tile cost (Gaussian steps until it stopped) and the NEXT frame of the same workspace
// dispatches its tiles in descending order of that cost -- longest-processing-time-first list scheduling; a viewer's or
// a trainer's consecutive frames of one camera are nearly the same picture, and for an unrelated camera the order is
// merely arbitrary, as raster order is.  Whatever the costs hold (first frame: uninitialised memory), a counting sort
// of the tile indices yields a permutation, and the image does not depend on the order: tiles are independent.
// Runs as one EXTRA workgroup of strip_count_kernel's launch, i.e. underneath the binning, three kernels ahead of its
// consumer.  256 bins of GS_ORDER_QUANTUM steps; arrival order inside a bin is whatever the LDS atomics make it.
#define GS_ORDER_QUANTUM 8
#ifndef GS_ORDER_BLOCKS
#define GS_ORDER_BLOCKS 0  // A/B switch (tools/ab_variants.py): 1 = coarse cost classes, 2 x 2 tile blocks kept on one XCD
#endif
#ifndef GS_ORDER_CLASS
#define GS_ORDER_CLASS 64  // GS_ORDER_BLOCKS: Gaussian steps per cost class (16 classes)
#endif
// GS_ORDER_BLOCKS (round 4): the longest-first order above made neighbouring tiles run at
// unrelated times on unrelated XCDs, so a record shared by the tiles of one Gaussian's rectangle was fetched once per
// tile (FETCH_SIZE of the compositing kernel 73 -> 135 MB per launch, now about the bytes of the records it composites).
// Workgroup p of a launch runs on XCD p % 8 (observed, MI355X_MICROARCH.md), so the four tiles of an aligned 2 x 2 block
// share an L2 -- and run at about the same time -- when they sit at positions p, p + 8, p + 16, p + 24 of the order.
// Tiles are classed by cost (GS_ORDER_CLASS steps per class, longest class first: still longest-processing-time-first at
// that granularity); inside a class the blocks whose four tiles all belong to it are laid out eight per group of 32
// positions, block j of the group at positions j + 8 i; the other tiles follow in arrival order.  Any cost array
// (first frame: uninitialised memory) still yields a permutation.
__device__ __forceinline__ void tile_order_blocks_workgroup(const uint32_t *__restrict__ tile_cost, uint32_t T,
                                                            uint32_t *__restrict__ tile_order, uint32_t ntx, uint32_t nty) {
    constexpr uint32_t NC = 16;
    __shared__ uint32_t s_full[NC], s_loose[NC], s_groups[NC], s_base_full[NC], s_base_loose[NC], s_cf[NC], s_cl[NC];
    if (threadIdx.x < NC) s_full[threadIdx.x] = s_loose[threadIdx.x] = s_cf[threadIdx.x] = s_cl[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t nbx = (ntx + 1) / 2, nby = (nty + 1) / 2, NB = nbx * nby;
    auto cls = [](uint32_t c) {
        const uint32_t q = c / GS_ORDER_CLASS;
        return q < NC - 1 ? q : NC - 1;
    };
    auto block_tiles = [&](uint32_t b, uint32_t tile[4], uint32_t cl[4]) -> uint32_t {  // -> tiles of the block (1, 2 or 4)
        const uint32_t bx = b % nbx, by = b / nbx;
        uint32_t n = 0;
        for (uint32_t dy = 0; dy < 2; ++dy)
            for (uint32_t dx = 0; dx < 2; ++dx) {
                const uint32_t x = 2 * bx + dx, y = 2 * by + dy;
                if (x < ntx && y < nty && y * ntx + x < T) {
                    tile[n] = y * ntx + x;
                    cl[n] = cls(tile_cost[tile[n]]);
                    ++n;
                }
            }
        return n;
    };
    for (uint32_t b = threadIdx.x; b < NB; b += STRIP_THREADS) {
        uint32_t tile[4], cl[4];
        const uint32_t n = block_tiles(b, tile, cl);
        if (n == 4 && cl[0] == cl[1] && cl[0] == cl[2] && cl[0] == cl[3])
            atomicAdd(&s_full[cl[0]], 1u);
        else
            for (uint32_t i = 0; i < n; ++i) atomicAdd(&s_loose[cl[i]], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int c = (int)NC - 1; c >= 0; --c) {  // longest class first
            const uint32_t g = s_full[c] / 8, rem = s_full[c] % 8;
            s_groups[c] = g;
            s_base_full[c] = run;
            run += 32 * g;
            s_base_loose[c] = run;
            run += s_loose[c] + 4 * rem;
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < NB; b += STRIP_THREADS) {
        uint32_t tile[4], cl[4];
        const uint32_t n = block_tiles(b, tile, cl);
        if (n == 4 && cl[0] == cl[1] && cl[0] == cl[2] && cl[0] == cl[3]) {
            const uint32_t c = cl[0], j = atomicAdd(&s_cf[c], 1u);
            if (j < 8 * s_groups[c]) {
                const uint32_t at = s_base_full[c] + 32 * (j / 8) + (j % 8);
                for (uint32_t i = 0; i < 4; ++i) tile_order[at + 8 * i] = tile[i];
            } else {
                const uint32_t at = s_base_loose[c] + atomicAdd(&s_cl[c], 4u);
                for (uint32_t i = 0; i < 4; ++i) tile_order[at + i] = tile[i];
            }
        } else {
            for (uint32_t i = 0; i < n; ++i) tile_order[s_base_loose[cl[i]] + atomicAdd(&s_cl[cl[i]], 1u)] = tile[i];
        }
    }
}

__device__ __forceinline__ void tile_order_workgroup(const uint32_t *__restrict__ tile_cost, uint32_t T,
                                                     uint32_t *__restrict__ tile_order, uint32_t ntx = 0, uint32_t nty = 0) {
#if GS_ORDER_BLOCKS
    if (ntx && nty && (uint64_t)ntx * nty == T) {
        tile_order_blocks_workgroup(tile_cost, T, tile_order, ntx, nty);
        return;
    }
#endif
    // 256 bins x 16 sub-counters (lane % 16): in a sparse frame most tiles cost the same (empty tiles: 0), and 64 lanes
    // adding to ONE LDS word serialise (first version: +10 us on a 50-us frame of 10,000 Gaussians); with the
    // sub-counters a wave's add hits every word at most four times.  Slot order: bin-major, sub-counter-minor.
    constexpr uint32_t SUB = 16, NC = 256 * SUB, PER = NC / STRIP_THREADS;
    __shared__ uint32_t s_bin[NC];
    __shared__ uint32_t s_wsum[STRIP_THREADS / 64];
    for (uint32_t c = threadIdx.x; c < NC; c += STRIP_THREADS) s_bin[c] = 0;
    __syncthreads();
    const uint32_t sub = threadIdx.x & (SUB - 1);
    auto counter_of = [&](uint32_t c) {
        const uint32_t q = c / GS_ORDER_QUANTUM;
        return (255u - (q < 255u ? q : 255u)) * SUB + sub;  // descending cost
    };
    for (uint32_t t = threadIdx.x; t < T; t += STRIP_THREADS) atomicAdd(&s_bin[counter_of(tile_cost[t])], 1u);
    __syncthreads();
    // exclusive scan of the NC counters: thread t owns counters [t PER, t PER + PER)
    uint32_t c[PER], sum = 0;
#pragma unroll
    for (uint32_t j = 0; j < PER; ++j) {
        c[j] = sum;
        sum += s_bin[threadIdx.x * PER + j];
    }
    const uint32_t incl = gs_wave_incl_scan_u32(sum);
    if ((threadIdx.x & 63) == 63) s_wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t off = incl - sum;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) off += s_wsum[w];
#pragma unroll
    for (uint32_t j = 0; j < PER; ++j) s_bin[threadIdx.x * PER + j] = off + c[j];  // cursor = first slot of the counter
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < T; t += STRIP_THREADS) tile_order[atomicAdd(&s_bin[counter_of(tile_cost[t])], 1u)] = t;
}

}  // namespace
