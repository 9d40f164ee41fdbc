// tile_bin.hip -- sort_mode 2: one-pass counting sort of the (tile, Gaussian) pairs by tile.
//
// The reference duplicates every Gaussian into its tiles and runs a full device radix sort on
// 64-bit (tile, depth) keys (gaussian.cu:197-250 + torch.sort in renderer.py).  The tile part
// of that key has at most a few tens of thousands of distinct values, which is a histogram that
// fits the 160 KiB LDS of one CU.  So the pairs are never materialised in Gaussian order at all.
//
// SLICE-SORTED variant (default; gs_bin_plan.lds_sort): S workgroups, each owns a contiguous slice of the Gaussians
//   S1 slice_sort_kernel : histogram of the tiles of the slice's rectangles in LDS (ds_add) -> exclusive scan of
//                          the T counters inside the workgroup -> row [T + 1] of the offset table -> second walk
//                          places every pair (depth_bits << 32 | gaussian) at ds_add_rtn(cursor[tile], 1) in an LDS
//                          staging buffer -> the buffer is streamed to the slice's own contiguous region of the pair
//                          array with coalesced stores.  A slice with more pairs than the buffer holds stores
//                          straight to its region instead (slower, same result).
//   S2 bin_totals_kernel : per tile, the sum of the S slices' counts; the LAST workgroup to finish scans the T totals
//                          and writes tile_ranges and the pair counter.
//   tile_sort.hip then gathers a tile's pairs from the S regions (offset table) while it loads them for sorting.
// Why: the table variant below scatters 8-byte stores over T output streams per workgroup; at 2.4 M Gaussians that
// is 33 MB of open cache lines per XCD against 4 MB of L2, every line leaves L2 several times half-filled (PMC:
// WRITE_SIZE 255 MB for 55.6 MB of pairs) and the scatter alone took 109 us of a 480 us frame.  Here no global store
// is scattered: the pair array is written once, in full lines.
//
// TABLE variant (tile grids whose counters leave no LDS for staging: beyond ~31 k tiles):
//   B1 bin_count_kernel   : <= 256 workgroups, each owns a contiguous slice of the Gaussians and
//                           histograms the tiles of their rectangles in LDS (ds_add, no global
//                           atomics); row b of a [B][T] table <- the histogram.
//   B2 bin_colscan_kernel : exclusive scan of every table column over b (all loads of a column
//                           are issued up front: one memory round trip), column total -> tile_count.
//   B3 bin_scatter_kernel : every workgroup scans tile_count itself (T values, redundant but free
//                           of any grid-wide dependency), adds its table row -> the first output
//                           slot of (this slice, tile) in LDS, then walks its rectangles again
//                           and writes (depth_bits << 32 | gaussian) at ds_add_rtn(slot, 1).
//                           Workgroup 0 also writes tile_ranges and the frame counters.
//
// The order inside a (slice, tile) segment depends on LDS arbitration; the per-tile sort that
// follows (tile_sort.hip) orders each tile by the UNIQUE composite (depth_bits, gaussian), so the
// final list is the oracle's (tile, depth_bits, gaussian_index) order bit for bit, every run.
#include <atomic>
#include <mutex>

#include "gs_common.h"
#include "gs_frame_layout.h"
#include "tile_bin_common.h"

namespace {

// The Gaussians of a slice are visited BIN_THREADS at a time; the records of the next BIN_PF visits are requested
// before the current ones are walked.  (Without it every visit exposed a full memory round trip: ten dependent
// round trips per workgroup at 2.4 M Gaussians, with only 16 waves per CU to hide them.)
#ifndef BIN_PF
#define BIN_PF 4
#endif

// ---------------------------------------------------------------- B1
template <bool DIST>
__global__ void __launch_bounds__(BIN_THREADS) bin_count_kernel(
    const uint4 *__restrict__ rects, const float4 *__restrict__ rec_geom, GsDistCull D, int64_t n, uint32_t per_block,
    uint32_t T, uint32_t ntx, uint32_t *__restrict__ table, const uint32_t *__restrict__ block_sums,
    const uint32_t *__restrict__ block_vis, uint32_t *__restrict__ slice_pairs, uint32_t *__restrict__ slice_vis) {
    extern __shared__ uint32_t s_hist[];
    __shared__ uint32_t s_acc[2];
    const uint32_t slice = slice_of_block(blockIdx.x, gridDim.x);
    const int64_t g0 = (int64_t)slice * per_block;
    auto load_rect = [&](uint32_t base) {
        const uint32_t i = base + threadIdx.x;
        // one coalesced 16-byte load per Gaussian: rectangle, depth bits, tile count
        return (i < per_block && g0 + i < n) ? rects[g0 + i] : make_uint4(0, 0, 0, 0);
    };
    auto load_xy = [&](uint32_t base, const uint4 &rc) {
        const uint32_t i = base + threadIdx.x;
        if (!DIST || !rc.w) return make_float2(0.f, 0.f);
        const float4 ge = rec_geom[(g0 + i) * GS_REC_STRIDE];
        return make_float2(ge.x, ge.y);
    };
    uint4 rc[BIN_PF];
#pragma unroll
    for (int k = 0; k < BIN_PF; ++k) rc[k] = load_rect(k * BIN_THREADS);  // in flight while the histogram is cleared
    for (uint32_t t = threadIdx.x; t < T; t += BIN_THREADS) s_hist[t] = 0;
    if (threadIdx.x < 2) s_acc[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t base = 0; base < per_block; base += BIN_PF * BIN_THREADS) {
        uint4 cur[BIN_PF];
#pragma unroll
        for (int k = 0; k < BIN_PF; ++k) {
            cur[k] = rc[k];
            rc[k] = load_rect(base + (BIN_PF + k) * BIN_THREADS);
        }
#pragma unroll
        for (int k = 0; k < BIN_PF; ++k) {
            const uint32_t b = base + k * BIN_THREADS;
            if (b >= per_block) break;  // uniform
            walk_rect<DIST>(cur[k], g0 + b + threadIdx.x, ntx, load_xy(b, cur[k]), D,
                            [&](uint32_t tile, uint32_t, uint32_t) { atomicAdd(&s_hist[tile], 1u); });
        }
    }
    // pairs (rectangle areas: the gradient-row slots) / visible Gaussians of this slice: sums over the
    // 256-Gaussian blocks of the project stage
    const int64_t nblk = (n + 255) / 256;
    for (uint32_t k = threadIdx.x; k < per_block / 256; k += BIN_THREADS) {
        const int64_t pb = g0 / 256 + k;
        if (pb < nblk) {
            atomicAdd(&s_acc[0], block_sums[pb]);
            atomicAdd(&s_acc[1], block_vis[pb]);
        }
    }
    __syncthreads();
    uint32_t *row = table + (size_t)slice * T;
    for (uint32_t t = threadIdx.x; t < T; t += BIN_THREADS) row[t] = s_hist[t];
    if (threadIdx.x == 0) {
        slice_pairs[slice] = s_acc[0];
        slice_vis[slice] = s_acc[1];
    }
}

// ---------------------------------------------------------------- B2
// Workgroup = 64 consecutive tiles (lane = tile) x 4 waves; wave w owns slices [64 w, 64 w + 64).
__global__ void __launch_bounds__(256) bin_colscan_kernel(uint32_t *__restrict__ table, uint32_t B, uint32_t T,
                                                         uint32_t *__restrict__ tile_count) {
    __shared__ uint32_t s_tot[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t t = blockIdx.x * 64 + lane;
    const bool ok = t < T;
    uint32_t v[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const uint32_t b = wave * 64 + i;
        v[i] = (ok && b < B) ? table[(size_t)b * T + t] : 0;
    }
    uint32_t run = 0;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const uint32_t c = v[i];
        v[i] = run;
        run += c;
    }
    s_tot[wave][lane] = run;
    __syncthreads();
    uint32_t off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t x = s_tot[w][lane];
        off += w < wave ? x : 0;
        total += x;
    }
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const uint32_t b = wave * 64 + i;
        if (ok && b < B) table[(size_t)b * T + t] = v[i] + off;
    }
    if (ok && wave == 0) tile_count[t] = total;
}

// ---------------------------------------------------------------- B3
// Three 64-bit block sums -- and the exclusive prefix of the first -- behind ONE pair of barriers (round 2 took three
// sums and a scan with their own barriers: +5 us on a 45-us frame of a small scene).
__device__ __forceinline__ void block_sum3_u64(unsigned long long &a, unsigned long long &b, unsigned long long &c,
                                               unsigned long long *prefix_a = nullptr) {
    __shared__ unsigned long long s_w[3][BIN_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long incl = a;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long x = __shfl_up(incl, o, 64);
        if (lane >= o) incl += x;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        b += __shfl_xor(b, o, 64);
        c += __shfl_xor(c, o, 64);
    }
    __syncthreads();  // s_w may still be read from a previous call
    if (lane == 63) s_w[0][wave] = incl;
    if (lane == 0) {
        s_w[1][wave] = b;
        s_w[2][wave] = c;
    }
    __syncthreads();
    unsigned long long off = 0, ta = 0;
    b = c = 0;
#pragma unroll
    for (int w = 0; w < BIN_THREADS / 64; ++w) {
        const unsigned long long x = s_w[0][w];
        off += w < wave ? x : 0;
        ta += x;
        b += s_w[1][w];
        c += s_w[2][w];
    }
    if (prefix_a) *prefix_a = off + incl - a;
    a = ta;
}

__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_wave, uint32_t &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = gs_wave_incl_scan_u32(v);
    __syncthreads();  // s_wave may still be read from the previous call
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t off = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < BIN_THREADS / 64; ++w) {
        const uint32_t x = s_wave[w];
        off += w < wave ? x : 0;
        total += x;
    }
    return off + incl - v;
}

template <bool DIST>
__global__ void __launch_bounds__(BIN_THREADS) bin_scatter_kernel(
    const uint4 *__restrict__ rects, const float4 *__restrict__ rec_geom, GsDistCull D, int64_t n, uint32_t per_block,
    uint32_t T, uint32_t ntx, const uint32_t *__restrict__ table, const uint32_t *__restrict__ tile_count,
    const uint32_t *__restrict__ slice_pairs, const uint32_t *__restrict__ slice_vis, uint32_t B,
    uint64_t *__restrict__ out, uint64_t max_pairs, uint32_t *__restrict__ pair_offsets,
    int32_t *__restrict__ tile_ranges, unsigned long long *__restrict__ counters, uint32_t n_bands, uint32_t fused) {
    extern __shared__ uint32_t s_slot[];
    __shared__ uint32_t s_wave[BIN_THREADS / 64];
    const uint32_t slice = slice_of_block(blockIdx.x, gridDim.x);
    // `fused` (small scenes: B x T <= 65,536, round 6): `table` is the RAW count table and there was no column-scan launch --
    // every workgroup adds up the B rows itself (tile totals and the prefix of the slices in front of its own: B coalesced
    // loads per tile, 40 KB per workgroup at 10,000 Gaussians) into LDS behind s_slot.  A frame of 10,000 Gaussians is a
    // chain of dependent launches of ~7 us each: one launch fewer.
    uint32_t *s_tot = s_slot + T, *s_pre = s_tot + T;
    if (fused) {
        for (uint32_t t = threadIdx.x; t < T; t += BIN_THREADS) {
            uint32_t pre = 0, tot = 0;
#pragma unroll 8
            for (uint32_t b = 0; b < B; ++b) {
                const uint32_t v = table[(size_t)b * T + t];
                pre += b < slice ? v : 0u;
                tot += v;
            }
            s_tot[t] = tot;
            s_pre[t] = pre;
        }
        __syncthreads();
    }
    auto count_of = [&](uint32_t t) { return fused ? s_tot[t] : tile_count[t]; };
    const int64_t g0 = (int64_t)slice * per_block;
    auto load_rect = [&](uint32_t base) {
        const uint32_t i = base + threadIdx.x;
        return (i < per_block && g0 + i < n) ? rects[g0 + i] : make_uint4(0, 0, 0, 0);
    };
    auto load_xy = [&](uint32_t base, const uint4 &rc) {
        const uint32_t i = base + threadIdx.x;
        if (!DIST || !rc.w) return make_float2(0.f, 0.f);
        const float4 ge = rec_geom[(g0 + i) * GS_REC_STRIDE];
        return make_float2(ge.x, ge.y);
    };
    uint4 rc[BIN_PF];
#pragma unroll
    for (int k = 0; k < BIN_PF; ++k) rc[k] = load_rect(k * BIN_THREADS);  // in flight while the tile starts are scanned
    // 1. tile starts = exclusive scan of tile_count (every workgroup computes all of them)
    const uint32_t per = (T + BIN_THREADS - 1) / BIN_THREADS;
    const uint32_t t0 = threadIdx.x * per, t1 = t0 + per < T ? t0 + per : T;
    // 64-bit totals: a degenerate scene (hundreds of thousands of screen-filling Gaussians) can exceed 2^32 pairs,
    // and a wrapped total must not slip under the capacity test
    unsigned long long mine = 0;
    uint32_t longest = 0;  // the frame's longest tile list (slice 0 reports it: GS_CNT_MAXLIST)
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t c = count_of(t);
        mine += c;
        longest = c > longest ? c : longest;
    }
    __shared__ uint32_t s_longest;
    if (threadIdx.x == 0) s_longest = 0;
    // listed pairs M (sum over tiles) and gradient-row slots R (sum of the rectangle areas; R == M unless DIST)
    unsigned long long rp = 0, rv = 0;
    for (uint32_t b = threadIdx.x; b < B; b += BIN_THREADS) {
        rp += slice_pairs[b];
        rv += slice_vis[b];
    }
    unsigned long long M = mine, R = rp, V = rv, run64 = 0;
    block_sum3_u64(M, R, V, &run64);
    if (M > max_pairs || R > max_pairs) {  // not enough room: leave the frame empty and report the true count
        if (slice == 0) {
            for (uint32_t t = threadIdx.x; t < T; t += BIN_THREADS) reinterpret_cast<int2 *>(tile_ranges)[t] = make_int2(0, 0);
            if (threadIdx.x == 0) {
                counters[GS_CNT_PAIRS] = 0;
                counters[GS_CNT_OVERFLOW] = M > R ? M : R;
                counters[GS_CNT_VISIBLE] = V;
                // only the strip variant's kernels maintain these; whoever reads them back must not see stale values
                counters[GS_CNT_BIG] = counters[GS_CNT_GROUPS] = counters[GS_CNT_MAXLIST] = 0;
                counters[GS_CNT_EXCESS] = counters[GS_CNT_MAXWALK] = counters[GS_CNT_EXCESS_WALK] = 0;
                counters[GS_CNT_RANPAST] = 0;
            }
        }
        return;
    }
    uint32_t dummy;
    uint32_t run = (uint32_t)run64;  // the frame fits: every prefix is below max_pairs < 2^30
    for (uint32_t t = t0; t < t1; ++t) {
        s_slot[t] = run;
        run += count_of(t);
    }
    __syncthreads();
    const uint32_t *row = fused ? s_pre : table + (size_t)slice * T;
    if (slice == 0)
        for (uint32_t t = threadIdx.x; t < T; t += BIN_THREADS) {
            const uint32_t s = s_slot[t], c = count_of(t);  // row 0 of the scanned table is all zeros
            // every tile is written (empty ones as (0, 0)): this path needs no memset of the ranges
            reinterpret_cast<int2 *>(tile_ranges)[t] = c ? make_int2((int)s, (int)(s + c)) : make_int2(0, 0);
        }
    else
        for (uint32_t t = threadIdx.x; t < T; t += BIN_THREADS) s_slot[t] += row[t];
    // 2. slice totals: frame counters, and the emission offset of this slice (per-pair gradient rows)
    uint32_t before = 0;
    if (pair_offsets) {
        uint32_t p = 0;
        for (uint32_t b = threadIdx.x; b < B; b += BIN_THREADS) p += b < slice ? slice_pairs[b] : 0;
        block_excl_scan(p, s_wave, before);
    }
    // the longest-list statistic (lists beyond GS_LONGEST_MIN only, as strip_sort_kernel reports it): the table variant
    // has no kernels for long lists, but a renderer that reads the statistic back must learn that the frame has one --
    // it then asks for the strip variant and its long-list kernels (GS_FRAME_LONG_LISTS; ADVICE round 3: a small scene
    // with a pile-up used to stay on the serial path for ever because this counter read 0 here)
    if (slice == 0 && longest > (uint32_t)GS_LONGEST_MIN) atomicMax(&s_longest, longest);  // (s_longest was zeroed before
                                                                                           // the barriers of the block sums)
    if (slice == 0 && threadIdx.x == 0) {
        counters[GS_CNT_PAIRS] = M;
        counters[GS_CNT_OVERFLOW] = 0;
        counters[GS_CNT_VISIBLE] = V;
        counters[GS_CNT_BIG] = counters[GS_CNT_GROUPS] = 0;  // strip variant only
        // raster_forward_kernel's walk statistics (atomicMax / atomicAdd onto these): a workspace fresh from the allocator holds
        // whatever was there -- round 6 found the long-list flag of a small scene depending on which tests had run before it
        counters[GS_CNT_EXCESS] = counters[GS_CNT_MAXWALK] = counters[GS_CNT_EXCESS_WALK] = 0;
        counters[GS_CNT_RANPAST] = 0;
    }
    __syncthreads();
    if (slice == 0 && threadIdx.x == 0) counters[GS_CNT_MAXLIST] = s_longest;
    // 3. scatter, one horizontal BAND of the tile grid at a time.  A workgroup writes into T output streams (one per
    // tile); at 2.4 M Gaussians the lines those streams keep open add up to ~7 MB per XCD against 4 MB of L2, every
    // line left L2 several times half-filled (PMC: WRITE_SIZE 255 MB for 55.6 MB of pairs, 4.6 x) and the stores made
    // up half of the kernel's 109 us.  Walking the slice once per band -- rectangles clipped to the band's tile rows,
    // which costs a little integer work per Gaussian and band -- keeps the open lines of a pass inside L2.
    const uint32_t nty = (T + ntx - 1) / ntx, rows_per_band = (nty + n_bands - 1) / n_bands;
    for (uint32_t band = 0; band < n_bands; ++band) {
        const uint32_t yb0 = band * rows_per_band, yb1 = yb0 + rows_per_band;
        if (band) {
#pragma unroll
            for (int k = 0; k < BIN_PF; ++k) rc[k] = load_rect(k * BIN_THREADS);
        }
        for (uint32_t base = 0; base < per_block; base += BIN_PF * BIN_THREADS) {  // uniform trip counts (barriers inside)
            uint4 cur[BIN_PF];
#pragma unroll
            for (int k = 0; k < BIN_PF; ++k) {
                cur[k] = rc[k];
                rc[k] = load_rect(base + (BIN_PF + k) * BIN_THREADS);
            }
#pragma unroll
            for (int k = 0; k < BIN_PF; ++k) {
                const uint32_t b = base + k * BIN_THREADS;
                if (b >= per_block) break;  // uniform
                const uint32_t i = b + threadIdx.x;
                const int64_t g = g0 + i;
                if (pair_offsets && band == 0) {  // uniform: prefix sum of the rectangle areas in Gaussian order
                    const uint32_t ex = block_excl_scan(cur[k].w, s_wave, dummy);
                    if (i < per_block && g < n) pair_offsets[g] = before + ex;
                    before += dummy;
                }
                uint4 rcb = cur[k];
                if (n_bands > 1) {  // clip the rectangle to the band's tile rows
                    const uint32_t y0 = rcb.x & 0xffff, y1 = rcb.x >> 16, w = (rcb.y >> 16) - (rcb.y & 0xffff);
                    const uint32_t cy0 = y0 > yb0 ? y0 : yb0, cy1 = y1 < yb1 ? y1 : yb1;
                    rcb.x = cy0 | (cy1 << 16);
                    rcb.w = (rcb.w && cy1 > cy0) ? (cy1 - cy0) * w : 0;
                }
                walk_rect<DIST>(rcb, g, ntx, load_xy(b, rcb), D, [&](uint32_t tile, uint32_t id, uint32_t d) {
                    const uint32_t slot = atomicAdd(&s_slot[tile], 1u);
#ifdef GS_DIAG_SCATTER_SMALL  // timing experiment only (tools/ab_variants.py): every store lands in a 128 KiB window
                    out[slot & 0x3fff] = ((uint64_t)d << 32) | id;
#else
                    out[slot] = ((uint64_t)d << 32) | id;
#endif
                });
            }
        }
    }
}

// ================================================================= slice-sorted variant
// Dynamic LDS: s_cnt[T] counters / cursors, then the staging buffer of `cap` pairs (8-byte aligned).
template <bool DIST>
__global__ void __launch_bounds__(BIN_THREADS) slice_sort_kernel(
    const uint4 *__restrict__ rects, const float4 *__restrict__ rec_geom, GsDistCull D, int64_t n, uint32_t per_slice,
    uint32_t T, uint32_t ntx, uint32_t cap, const uint32_t *__restrict__ block_sums,
    const uint32_t *__restrict__ block_vis, uint32_t *__restrict__ table, uint32_t *__restrict__ slice_base,
    uint64_t *__restrict__ out, uint64_t max_pairs, uint32_t *__restrict__ pair_offsets,
    uint32_t *__restrict__ tile_count, unsigned long long *__restrict__ counters) {
    extern __shared__ uint32_t s_cnt[];
    uint64_t *s_stage = reinterpret_cast<uint64_t *>(s_cnt + ((T + 1) & ~1u));
    __shared__ uint32_t s_wave[BIN_THREADS / 64];
    const uint32_t slice = blockIdx.x;
    const int64_t g0 = (int64_t)slice * per_slice;
    auto load_rect = [&](uint32_t base) {
        const uint32_t i = base + threadIdx.x;
        return (i < per_slice && g0 + i < n) ? rects[g0 + i] : make_uint4(0, 0, 0, 0);
    };
    auto load_xy = [&](uint32_t base, const uint4 &rc) {
        const uint32_t i = base + threadIdx.x;
        if (!DIST || !rc.w) return make_float2(0.f, 0.f);
        const float4 ge = rec_geom[(g0 + i) * GS_REC_STRIDE];
        return make_float2(ge.x, ge.y);
    };
    uint4 rc[BIN_PF];
#pragma unroll
    for (int k = 0; k < BIN_PF; ++k) rc[k] = load_rect(k * BIN_THREADS);  // in flight during the set-up below
    for (uint32_t t = threadIdx.x; t < T; t += BIN_THREADS) s_cnt[t] = 0;
    if (slice == 0)  // bin_totals_kernel accumulates the per-tile totals with atomics
        for (uint32_t t = threadIdx.x; t < T; t += BIN_THREADS) tile_count[t] = 0;
    // frame totals from the project stage's per-block sums (rectangle areas = gradient-row slots; == pairs unless
    // DIST): R over all blocks, `base` over the blocks in front of this slice = start of its region
    const uint32_t nblk = (uint32_t)((n + 255) / 256), first_blk = slice * (per_slice / 256);
    unsigned long long r_all = 0, r_before = 0, v_all = 0;
    for (uint32_t b = threadIdx.x; b < nblk; b += BIN_THREADS) {
        const uint32_t c = block_sums[b];
        r_all += c;
        r_before += b < first_blk ? c : 0;
        v_all += block_vis[b];
    }
    unsigned long long R = r_all, base = r_before, V = v_all;
    block_sum3_u64(R, base, V);
    uint32_t *row = table + (size_t)slice * (T + 1);
    if (slice == 0 && threadIdx.x == 0) {
        counters[GS_CNT_VISIBLE] = V;
        counters[GS_CNT_OVERFLOW] = R > max_pairs ? R : 0;
        counters[GS_CNT_TICKET] = 0;  // bin_totals_kernel counts its finished workgroups here
        counters[GS_CNT_EXCESS] = 0;  // (the table variant does not count the pairs beyond GS_LONG_MIN per tile)
        counters[GS_CNT_MAXWALK] = 0;  // raster_forward_kernel's walk statistics
        counters[GS_CNT_EXCESS_WALK] = 0;
        counters[GS_CNT_BIG] = counters[GS_CNT_GROUPS] = counters[GS_CNT_MAXLIST] = 0;  // strip variant only
    }
    if (R > max_pairs) {  // not enough room: an all-zero table = an empty frame, the true count is reported
        for (uint32_t t = threadIdx.x; t <= T; t += BIN_THREADS) row[t] = 0;
        if (threadIdx.x == 0) slice_base[slice] = 0;
        return;
    }
    if (threadIdx.x == 0) slice_base[slice] = (uint32_t)base;
    __syncthreads();
    // ---- 1. count (and, for the backward, the emission offset of every Gaussian: prefix sum of the areas)
    uint32_t before = (uint32_t)base, dummy;
    for (uint32_t b0 = 0; b0 < per_slice; b0 += BIN_PF * BIN_THREADS) {
        uint4 cur[BIN_PF];
#pragma unroll
        for (int k = 0; k < BIN_PF; ++k) {
            cur[k] = rc[k];
            rc[k] = load_rect(b0 + (BIN_PF + k) * BIN_THREADS);
        }
#pragma unroll
        for (int k = 0; k < BIN_PF; ++k) {
            const uint32_t b = b0 + k * BIN_THREADS;
            if (b >= per_slice) break;  // uniform
            const uint32_t i = b + threadIdx.x;
            if (pair_offsets) {  // uniform
                const uint32_t ex = block_excl_scan(cur[k].w, s_wave, dummy);
                if (i < per_slice && g0 + i < n) pair_offsets[g0 + i] = before + ex;
                before += dummy;
            }
            walk_rect<DIST>(cur[k], g0 + i, ntx, load_xy(b, cur[k]), D,
                            [&](uint32_t tile, uint32_t, uint32_t) { atomicAdd(&s_cnt[tile], 1u); });
        }
    }
#pragma unroll
    for (int k = 0; k < BIN_PF; ++k) rc[k] = load_rect(k * BIN_THREADS);  // for the second walk, under the scan
    __syncthreads();
    // ---- 2. exclusive scan of the T counters in place -> offsets of the tiles inside this slice's region
    const uint32_t per = (T + BIN_THREADS - 1) / BIN_THREADS;
    const uint32_t t0 = threadIdx.x * per < T ? threadIdx.x * per : T, t1 = t0 + per < T ? t0 + per : T;
    uint32_t mine = 0;
    for (uint32_t t = t0; t < t1; ++t) mine += s_cnt[t];
    uint32_t L;
    uint32_t run = block_excl_scan(mine, s_wave, L);
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t c = s_cnt[t];
        s_cnt[t] = run;
        run += c;
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < T; t += BIN_THREADS) row[t] = s_cnt[t];
    if (threadIdx.x == 0) row[T] = L;
    __syncthreads();  // the offsets are on their way to the table before the cursors start moving
    // ---- 3. place: into the staging buffer, or straight into the region when the slice does not fit
    const bool fits = L <= cap;  // uniform
    uint64_t *region = out + base;
    auto place = [&](auto put) {
        for (uint32_t b0 = 0; b0 < per_slice; b0 += BIN_PF * BIN_THREADS) {
            uint4 cur[BIN_PF];
#pragma unroll
            for (int k = 0; k < BIN_PF; ++k) {
                cur[k] = rc[k];
                rc[k] = load_rect(b0 + (BIN_PF + k) * BIN_THREADS);
            }
#pragma unroll
            for (int k = 0; k < BIN_PF; ++k) {
                const uint32_t b = b0 + k * BIN_THREADS;
                if (b >= per_slice) break;  // uniform
                walk_rect<DIST>(cur[k], g0 + b + threadIdx.x, ntx, load_xy(b, cur[k]), D,
                                [&](uint32_t tile, uint32_t id, uint32_t d) {
                                    put(atomicAdd(&s_cnt[tile], 1u), ((uint64_t)d << 32) | id);
                                });
            }
        }
    };
    if (fits)  // uniform: LDS stores in the common case, global ones for a slice that does not fit
        place([&](uint32_t slot, uint64_t pair) { s_stage[slot] = pair; });
    else
        place([&](uint32_t slot, uint64_t pair) { region[slot] = pair; });
    if (!fits) return;
    __syncthreads();
    // ---- 4. stream the tile-ordered pairs out: consecutive lanes, consecutive addresses
    for (uint32_t i = threadIdx.x; i < L; i += BIN_THREADS) region[i] = s_stage[i];
}

// Per tile: number of pairs over all slices (grid.y cuts the slices into chunks that add their partial sums with one
// atomic per tile; tile_count was zeroed by slice 0 of slice_sort_kernel); the last workgroup to finish turns the
// totals into tile ranges.
#define TOT_CHUNK 32  // slices per workgroup
__global__ void __launch_bounds__(256) bin_totals_kernel(const uint32_t *__restrict__ table, uint32_t S, uint32_t T,
                                                        uint32_t *__restrict__ tile_count,
                                                        int32_t *__restrict__ tile_ranges,
                                                        unsigned long long *__restrict__ counters) {
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_last;
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const uint32_t s_begin = blockIdx.y * TOT_CHUNK, s_end = s_begin + TOT_CHUNK < S ? s_begin + TOT_CHUNK : S;
    if (t < T) {
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;  // four independent chains of loads
        const uint32_t *p = table + t;
        const size_t stride = (size_t)T + 1;
        uint32_t s = s_begin;
        for (; s + 4 <= s_end; s += 4) {
            a0 += p[(s + 0) * stride + 1] - p[(s + 0) * stride];
            a1 += p[(s + 1) * stride + 1] - p[(s + 1) * stride];
            a2 += p[(s + 2) * stride + 1] - p[(s + 2) * stride];
            a3 += p[(s + 3) * stride + 1] - p[(s + 3) * stride];
        }
        for (; s < s_end; ++s) a0 += p[s * stride + 1] - p[s * stride];
        const uint32_t part = (a0 + a1) + (a2 + a3);
        if (part) atomicAdd(&tile_count[t], part);
    }
    __threadfence();  // the totals are visible device-wide before this workgroup takes its ticket
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = atomicAdd(&counters[GS_CNT_TICKET], 1ull) == (unsigned long long)gridDim.x * gridDim.y - 1;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    // exclusive scan of the T totals by this (the last) workgroup: thread i owns a contiguous run of tiles
    const uint32_t per = (T + 255) / 256;
    const uint32_t t0 = threadIdx.x * per < T ? threadIdx.x * per : T, t1 = t0 + per < T ? t0 + per : T;
    // the totals were written by other workgroups' atomics (L2): read them with device-scope loads, eight in flight
    // (a `volatile` loop issued them one by one: 64 dependent round trips, 50 us)
    auto tc = [&](uint32_t i) { return __hip_atomic_load(tile_count + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    uint32_t mine = 0;
    for (uint32_t i = t0; i < t1; i += 8) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = i + k < t1 ? tc(i + k) : 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) mine += v[k];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = gs_wave_incl_scan_u32(mine);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        off += w < wave ? s_wave[w] : 0;
        total += s_wave[w];
    }
    uint32_t run = off + incl - mine;
    for (uint32_t i = t0; i < t1; i += 8) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = i + k < t1 ? tc(i + k) : 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (i + k >= t1) break;
            // every tile is written (empty ones as (0, 0)): no memset of the ranges
            reinterpret_cast<int2 *>(tile_ranges)[i + k] = v[k] ? make_int2((int)run, (int)(run + v[k])) : make_int2(0, 0);
            run += v[k];
        }
    }
    if (threadIdx.x == 0) counters[GS_CNT_PAIRS] = total;
}

}  // namespace

int gs_stage_tile_bin(const gs_frame *f, const gs_frame_ws &ws, hipStream_t stream) {
    gs_frame_geom G = gs_frame_geometry(f);
    const gs_bin_plan plan = gs_bin_plan_for(f->N, f->max_pairs, G.n_tiles, (f->flags & GS_FRAME_SLICE_SORT) != 0);
    const uint32_t T = (uint32_t)G.n_tiles;
    const bool dist = f->tile_culling_method == 0;
    GsDistCull D = {(float)(G.padW / 2), (float)(G.padH / 2), f->focal_x, f->focal_y, f->thresh};
    // dynamic LDS above 64 KiB needs the opt-in (gfx950: 160 KiB per workgroup): once per DEVICE (a function
    // attribute belongs to the device's code object), thread-safe
    static std::mutex attr_mu;
    static std::atomic<uint64_t> attr_done{0};
    int dev = 0;
    GS_HIP(hipGetDevice(&dev));
    if (dev < 64 && !((attr_done.load(std::memory_order_acquire) >> dev) & 1)) {
        std::lock_guard<std::mutex> lock(attr_mu);
        for (const void *fn : {(const void *)bin_count_kernel<false>, (const void *)bin_count_kernel<true>,
                               (const void *)bin_scatter_kernel<false>, (const void *)bin_scatter_kernel<true>})
            GS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, GS_BIN_MAX_TILES * 4));
        for (const void *fn : {(const void *)slice_sort_kernel<false>, (const void *)slice_sort_kernel<true>})
            GS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, GS_BIN_LDS_BYTES + 64));
        attr_done.fetch_or(1ull << dev, std::memory_order_release);
    }
    if (plan.lds_sort) {
        const size_t lds = sizeof(uint32_t) * ((T + 1) & ~1u) + sizeof(uint64_t) * plan.cap;
#define GS_LAUNCH_SLICE_SORT(DIST)                                                                                     \
    hipLaunchKernelGGL(slice_sort_kernel<DIST>, dim3(plan.slices), dim3(BIN_THREADS), lds, stream, ws.rects,           \
                       ws.rec_geom, D, f->N, plan.per_slice, T, (uint32_t)G.ntx, plan.cap, ws.block_sums,              \
                       ws.block_vis, ws.bin_table, ws.slice_pairs, ws.keys_a, (uint64_t)f->max_pairs,                  \
                       f->training ? ws.pair_offsets : nullptr, ws.tile_count, ws.counters)
        if (dist)
            GS_LAUNCH_SLICE_SORT(true);
        else
            GS_LAUNCH_SLICE_SORT(false);
#undef GS_LAUNCH_SLICE_SORT
        GS_CHECK_LAUNCH();
        hipLaunchKernelGGL(bin_totals_kernel, dim3((unsigned)gs_div_up(T, 256), (unsigned)gs_div_up(plan.slices, TOT_CHUNK)),
                           dim3(256), 0, stream, ws.bin_table, plan.slices, T, ws.tile_count, ws.tile_ranges,
                           ws.counters);
        GS_CHECK_LAUNCH();
        return 0;
    }
    const uint32_t per_block = bin_per_block(f->N);
    const uint32_t B = (uint32_t)gs_div_up(f->N, per_block);
    // small scenes: the scatter adds up the raw count table itself, no column-scan launch (bin_scatter_kernel)
    const bool fused_scan = (uint64_t)B * T <= 65536 && T <= GS_BIN_MAX_TILES / 3;
    const size_t lds = sizeof(uint32_t) * T, lds_scatter = fused_scan ? 3 * lds : lds;
    // scatter bands: the pairs of one band (8 B each, capacity as the estimate) should stay within ~2.5 MB per XCD
    uint32_t n_bands = (uint32_t)gs_div_up(f->max_pairs * 8, (int64_t)8 * 2560 * 1024);
#ifdef BIN_BANDS
    n_bands = BIN_BANDS;  // experiments (tools/ab_variants.py)
#endif
    if (n_bands < 1) n_bands = 1;
    if (n_bands > 8) n_bands = 8;
    if (n_bands > (uint32_t)G.nty) n_bands = (uint32_t)G.nty;
#define GS_LAUNCH_BIN(DIST)                                                                                            \
    do {                                                                                                               \
        if (!gs_frame_fused_table_count(f)) { /* else: counted by the project stage (frame_project_bin_count_kernel) */ \
            hipLaunchKernelGGL(bin_count_kernel<DIST>, dim3(B), dim3(BIN_THREADS), lds, stream, ws.rects, ws.rec_geom, \
                               D, f->N, per_block, T, (uint32_t)G.ntx, ws.bin_table, ws.block_sums, ws.block_vis,      \
                               ws.slice_pairs, ws.slice_vis);                                                          \
            GS_CHECK_LAUNCH();                                                                                         \
        }                                                                                                              \
        if (!fused_scan) {                                                                                             \
            hipLaunchKernelGGL(bin_colscan_kernel, dim3((unsigned)gs_div_up(T, 64)), dim3(256), 0, stream,             \
                               ws.bin_table, B, T, ws.tile_count);                                                     \
            GS_CHECK_LAUNCH();                                                                                         \
        }                                                                                                              \
        hipLaunchKernelGGL(bin_scatter_kernel<DIST>, dim3(B), dim3(BIN_THREADS), lds_scatter, stream, ws.rects,        \
                           ws.rec_geom, D, f->N, per_block, T, (uint32_t)G.ntx, ws.bin_table, ws.tile_count,           \
                           ws.slice_pairs, ws.slice_vis, B, ws.keys_a, (uint64_t)f->max_pairs,                         \
                           f->training ? ws.pair_offsets : nullptr, ws.tile_ranges, ws.counters, n_bands,              \
                           fused_scan ? 1u : 0u);                                                                      \
        GS_CHECK_LAUNCH();                                                                                             \
    } while (0)
    if (dist)
        GS_LAUNCH_BIN(true);
    else
        GS_LAUNCH_BIN(false);
#undef GS_LAUNCH_BIN
    return 0;
}
