// strip_bin.hip -- sort_mode 2, STRIP variant (default): two-level counting sort of the (tile, Gaussian) pairs.
//
// The table variant (tile_bin.hip) scatters one 8-byte pair per (tile, Gaussian) into T output streams per slice of
// the Gaussian array: at 2.4 M Gaussians / 1080p that is 254 x 8160 segments of 3.4 pairs on average, every store left
// L2 as its own 32-byte write (PMC, profiles/r02_a: WRITE_SIZE 233 MB for 55.6 MB of pairs, the scatter alone 101 us of
// a 450 us frame).  Here the T-way scatter is split into two levels with long runs on both:
//
//   level 1 (this file): a STRIP is GS_STRIP_W = 8 consecutive tiles of one tile row.  Every Gaussian emits ONE 8-byte
//     entry (depth_bits << 32 | first covered tile of the strip << 29 | last << 26 | gaussian) per strip its rectangle
//     crosses -- 2.1 entries instead of 3.7 pairs per Gaussian -- and the entries are counting-sorted by strip:
//       strip_count_kernel   <= 256 slices of the Gaussian array, one workgroup each: LDS histogram over the strips of
//                            (entries << 32 | pairs) in one 64-bit LDS atomic per entry -> row of a [S][NS] table;
//       strip_colscan_kernel exclusive scan of every column over the slices, strip totals;
//       strip_scatter_kernel every slice scans the strip totals itself (entry and pair base of every strip: redundant,
//                            but free of any grid-wide dependency -- a "last workgroup finishes" ticket costs a
//                            device-scope fence per workgroup, measured 13 us for this 2 MB scan; slice 0 writes the
//                            bases and the frame counters), places its entries by strip in an LDS staging buffer (ds_add_rtn cursors)
//                            and copies each (slice, strip) run to its final place with consecutive lanes: a run
//                            is ~15 entries = one 128-byte line, written by one instruction.
//   level 2 (tile_sort.hip, strip_sort_kernel): one workgroup per half strip (four tiles) reads the strip's entries
//     (contiguous), expands them into its tiles' pair lists INSIDE LDS, sorts every list by (depth_bits, gaussian) there
//     and writes only the sorted ids: the unsorted pairs never exist in HBM (table variant: 8 B written + 8 B read per
//     pair).  A half strip with more pairs than the LDS window expands into global memory and takes the per-tile sort
//     of the table variant.
//
// The result -- tile ranges, sorted ids, optional sorted keys, counters, pair offsets -- is bit-identical to the other
// variants: the final order is by the unique (tile, depth_bits, gaussian) whatever the arrival order.
#include <atomic>
#include <mutex>

#include "gs_common.h"
#include "gs_frame_layout.h"
#include "strip_common.h"

namespace {

struct SliceLoader {
    const uint4 *rects;
    const float4 *rec_geom;
    int64_t n, g0;
    uint32_t per_slice;
    __device__ __forceinline__ uint4 rect(uint32_t base) const {
        const uint32_t i = base + threadIdx.x;
        return (i < per_slice && g0 + i < n) ? rects[g0 + i] : make_uint4(0, 0, 0, 0);
    }
    template <bool DIST>
    __device__ __forceinline__ float2 xy(uint32_t base, const uint4 &rc) const {
        if (!DIST || !rc.w) return make_float2(0.f, 0.f);
        const float4 ge = rec_geom[(g0 + base + threadIdx.x) * GS_REC_STRIDE];
        return make_float2(ge.x, ge.y);
    }
};

// ---------------------------------------------------------------- L1a: count
template <bool DIST>
__global__ void __launch_bounds__(STRIP_THREADS) strip_count_kernel(
    const uint4 *__restrict__ rects, const float4 *__restrict__ rec_geom, GsDistCull D, int64_t n, uint32_t per_slice,
    gs_strip_geom SG, uint32_t S, unsigned long long *__restrict__ table, const uint32_t *__restrict__ block_sums,
    const uint32_t *__restrict__ block_vis, uint32_t *__restrict__ slice_pairs, uint32_t *__restrict__ slice_vis,
    const uint32_t *__restrict__ tile_cost, uint32_t n_tiles, uint32_t *__restrict__ tile_order,
    const unsigned long long *__restrict__ gate) {
    extern __shared__ unsigned long long s_hist[];  // [NS] entries << 32 | pairs of this slice
    __shared__ uint32_t s_acc[2];
    // `gate` (the second, untrimmed pass of a GS_FRAME_OCCLUSION_CULL frame, gs_frame_layout.h): nothing to do unless a tile
    // ran past its cut.  That pass recounts the entries only -- block_sums == NULL: the slices' rectangle areas and visible
    // counts are the first pass's
    if (gate && *gate == 0) return;
    if (blockIdx.x >= S) {  // the one extra workgroup of the launch (uniform)
        tile_order_workgroup(tile_cost, n_tiles, tile_order, SG.ntx, SG.nty);
        return;
    }
    const uint32_t slice = strip_slice_of_block(blockIdx.x, S);
    const SliceLoader L = {rects, rec_geom, n, (int64_t)slice * per_slice, per_slice};
    uint4 rc[STRIP_PF];
#pragma unroll
    for (int k = 0; k < STRIP_PF; ++k) rc[k] = L.rect(k * STRIP_THREADS);  // in flight while the histogram is cleared
    for (uint32_t t = threadIdx.x; t < SG.NS; t += STRIP_THREADS) s_hist[t] = 0;
    if (threadIdx.x < 2) s_acc[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t base = 0; base < per_slice; base += STRIP_PF * STRIP_THREADS) {
        uint4 cur[STRIP_PF];
#pragma unroll
        for (int k = 0; k < STRIP_PF; ++k) {
            cur[k] = rc[k];
            rc[k] = L.rect(base + (STRIP_PF + k) * STRIP_THREADS);
        }
#pragma unroll
        for (int k = 0; k < STRIP_PF; ++k) {
            const uint32_t b = base + k * STRIP_THREADS;
            if (b >= per_slice) break;  // uniform
            walk_strips<DIST>(cur[k], L.g0 + b + threadIdx.x, SG, L.xy<DIST>(b, cur[k]), D,
                              [&](uint32_t strip, uint32_t, uint32_t, uint32_t np) {
                                  atomicAdd(&s_hist[strip], (1ull << 32) | np);
                              });
        }
    }
    // rectangle areas (= gradient-row slots; == pairs unless DIST) and visible Gaussians of this slice, from the
    // project stage's per-block sums
    const int64_t nblk = (n + 255) / 256;
    for (uint32_t k = threadIdx.x; block_sums && k < per_slice / 256; k += STRIP_THREADS) {
        const int64_t pb = L.g0 / 256 + k;
        if (pb < nblk) {
            atomicAdd(&s_acc[0], block_sums[pb]);
            atomicAdd(&s_acc[1], block_vis[pb]);
        }
    }
    __syncthreads();
    unsigned long long *row = table + (size_t)slice * SG.NS;
    for (uint32_t t = threadIdx.x; t < SG.NS; t += STRIP_THREADS) row[t] = s_hist[t];
    if (threadIdx.x == 0 && block_sums) {
        slice_pairs[slice] = s_acc[0];
        slice_vis[slice] = s_acc[1];
    }
}

// ---------------------------------------------------------------- L1b: column scan + strip bases
// Workgroup = 16 consecutive strips x 16 groups of 16 slices (thread = (group, strip): a group's 16 loads of one
// slice row are one 128-byte line, and all 16 loads of a thread are in flight at once).  The packed
// (entries << 32 | pairs) sums never carry from the low half: a strip lists at most 8 N < 2^32 pairs.
__global__ void __launch_bounds__(256) strip_colscan_kernel(
    const unsigned long long *__restrict__ table, unsigned long long *__restrict__ scan, uint32_t S, uint32_t NS,
    unsigned long long *__restrict__ strip_tot, const unsigned long long *__restrict__ gate) {
    static_assert(GS_BIN_SLICES == 256, "16 groups of 16 slices");
    if (gate && *gate == 0) return;  // (see strip_count_kernel)
    __shared__ unsigned long long s_tot[16][17];
    const uint32_t col = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const uint32_t t = blockIdx.x * 16 + col;
    const bool ok = t < NS;
    unsigned long long v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t b = grp * 16 + i;
        v[i] = (ok && b < S) ? table[(size_t)b * NS + t] : 0;
    }
    unsigned long long run = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const unsigned long long c = v[i];
        v[i] = run;
        run += c;
    }
    s_tot[grp][col] = run;
    __syncthreads();
    unsigned long long off = 0, total = 0;
#pragma unroll
    for (uint32_t g = 0; g < 16; ++g) {
        const unsigned long long x = s_tot[g][col];
        off += g < grp ? x : 0;
        total += x;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t b = grp * 16 + i;
        if (ok && b < S) scan[(size_t)b * NS + t] = v[i] + off;
    }
    if (ok && grp == 0) strip_tot[t] = total;
}

// ---------------------------------------------------------------- L1c: scatter through an LDS staging buffer
__device__ __forceinline__ uint32_t strip_block_excl_scan(uint32_t v, uint32_t *s_wave, uint32_t &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = gs_wave_incl_scan_u32(v);
    __syncthreads();  // s_wave may still be read from the previous call
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t off = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < STRIP_THREADS / 64; ++w) {
        const uint32_t x = s_wave[w];
        off += w < wave ? x : 0;
        total += x;
    }
    return off + incl - v;
}

// Dynamic LDS: s_cur[NS] (staging cursor of every strip, ends up at the END of the strip's run), s_gd[NS] (final
// index of staging slot 0 of the strip's run, i.e. entry i of the staging buffer belongs at s_gd[strip] + i), then
// `cap` staged entries.  Entries beyond `cap` (a slice that does not fit) are stored straight to their final place.
template <bool DIST>
__global__ void __launch_bounds__(STRIP_THREADS) strip_scatter_kernel(
    const uint4 *__restrict__ rects, const float4 *__restrict__ rec_geom, GsDistCull D, int64_t n, uint32_t per_slice,
    gs_strip_geom SG, uint32_t S, uint32_t cap, const unsigned long long *__restrict__ scan,
    const unsigned long long *__restrict__ strip_tot, unsigned long long *__restrict__ strip_base,
    const uint32_t *__restrict__ slice_pairs, const uint32_t *__restrict__ slice_vis, uint64_t max_pairs,
    unsigned long long *__restrict__ out, uint32_t *__restrict__ pair_offsets,
    unsigned long long *__restrict__ counters, const uint32_t *__restrict__ cut, uint32_t n_tiles,
    const unsigned long long *__restrict__ gate, const uint4 *__restrict__ surv, const uint32_t *__restrict__ slice_nsurv) {
    extern __shared__ unsigned long long s_dyn[];
    if (gate && *gate == 0) return;  // (see strip_count_kernel)
    uint32_t *s_cur = reinterpret_cast<uint32_t *>(s_dyn), *s_gd = s_cur + SG.NS;
    unsigned long long *s_stage = s_dyn + SG.NS;  // 2 NS uint32 = NS uint64
    // occlusion cuts (the same table the count pass used): staged in LDS behind the `cap` staged entries, every tile row
    // padded to whole strips (walk_strips<.., true>)
    uint32_t *s_cut = reinterpret_cast<uint32_t *>(s_stage + cap);
    if (cut) {
        // (eight loads in flight per thread -- one at a time, each waited for, cost eight load latencies; barriers follow below)
        const uint32_t stride = SG.nsx * GS_STRIP_W;
        for (uint32_t t0 = threadIdx.x; t0 < stride * SG.nty; t0 += 8 * STRIP_THREADS) {
            uint32_t v[8];
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) {
                const uint32_t t = t0 + k * STRIP_THREADS, iy = t / stride, ix = t - iy * stride;
                const bool in = iy < SG.nty && ix < SG.ntx;
                v[k] = cut[in ? iy * SG.ntx + ix : 0u];
                v[k] = in ? v[k] : GS_NO_CUT;
            }
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k)
                if (t0 + k * STRIP_THREADS < stride * SG.nty) s_cut[t0 + k * STRIP_THREADS] = v[k];
        }
    }
    __shared__ uint32_t s_wave[STRIP_THREADS / 64];
    __shared__ unsigned long long s_wave64[4 * (STRIP_THREADS / 64)];
    const uint32_t slice = strip_slice_of_block(blockIdx.x, gridDim.x);
    // `surv` (first pass of an occlusion-culled frame): the slice's rectangles come from the compact list the project stage
    // left -- (y range, x range, depth, Gaussian) of the Gaussians it projected that touch a tile, slice_nsurv[slice] of them
    // at the slice's own offset -- instead of from all per_slice rectangle records, most of which were not even written
    const uint32_t count = surv ? slice_nsurv[slice] : per_slice;
    const SliceLoader L = {surv ? surv : rects, rec_geom, n, (int64_t)slice * per_slice, count};
    uint4 rc[STRIP_PF];
#pragma unroll
    for (int k = 0; k < STRIP_PF; ++k) rc[k] = L.rect(k * STRIP_THREADS);  // in flight during the set-up below
    // ---- strip totals -> entry / pair base of every strip, frame totals (every workgroup computes all of them)
    const uint32_t per = (SG.NS + STRIP_THREADS - 1) / STRIP_THREADS;
    const uint32_t t0 = threadIdx.x * per < SG.NS ? threadIdx.x * per : SG.NS, t1 = t0 + per < SG.NS ? t0 + per : SG.NS;
    unsigned long long me = 0, mp = 0, rp = 0, rv = 0;  // separate 64-bit sums: a degenerate scene can exceed 2^32 pairs
    const unsigned long long *row = scan + (size_t)slice * SG.NS, *next = row + SG.NS;
    const bool last = slice + 1 == S;
    // all loads of the set-up are issued before the first block scan; the common case (one strip per thread) keeps them
    // in registers
    const unsigned long long tot0 = t0 < t1 ? strip_tot[t0] : 0, a0 = t0 < t1 ? row[t0] : 0,
                             b0 = t0 < t1 ? (last ? tot0 : next[t0]) : 0;
    for (uint32_t t = t0; t < t1; ++t) {
        const unsigned long long x = t == t0 ? tot0 : strip_tot[t];
        me += x >> 32;
        mp += x & 0xffffffffull;
    }
    for (uint32_t b = threadIdx.x; b < S; b += STRIP_THREADS) {
        rp += slice_pairs[b];
        rv += slice_vis[b];
    }
    // one block scan for the four 64-bit values (entries and pairs need the prefix, R and V only the total)
    unsigned long long E, M, R, V, be, bp;
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        unsigned long long ie = me, ip = mp;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long x = __shfl_up(ie, o, 64), y = __shfl_up(ip, o, 64);
            if (lane >= o) {
                ie += x;
                ip += y;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            rp += __shfl_xor(rp, o, 64);
            rv += __shfl_xor(rv, o, 64);
        }
        if (lane == 63) {
            s_wave64[wave] = ie;
            s_wave64[16 + wave] = ip;
            s_wave64[32 + wave] = rp;
            s_wave64[48 + wave] = rv;
        }
        __syncthreads();
        unsigned long long oe = 0, op = 0;
        E = M = R = V = 0;
#pragma unroll
        for (int w = 0; w < STRIP_THREADS / 64; ++w) {
            const unsigned long long x = s_wave64[w], y = s_wave64[16 + w];
            oe += w < wave ? x : 0;
            op += w < wave ? y : 0;
            E += x;
            M += y;
            R += s_wave64[32 + w];  // listed pairs M; gradient-row slots R (rectangle areas; == M unless DIST)
            V += s_wave64[48 + w];
        }
        be = oe + ie - me;
        bp = op + ip - mp;
    }
    const bool overflow = M > max_pairs || R > max_pairs;  // not enough room: the frame is left empty, the true count reported
    if (slice == 0) {
        for (uint32_t t = t0; t < t1; ++t) {
            const unsigned long long x = t == t0 ? tot0 : strip_tot[t];
            strip_base[t] = overflow ? 0ull : (be << 32) | bp;  // both < 2^30 when the frame fits
            be += x >> 32;
            bp += x & 0xffffffffull;
        }
        be -= me;
        if (threadIdx.x == 0) {
            counters[GS_CNT_PAIRS] = overflow ? 0 : M;
            counters[GS_CNT_OVERFLOW] = overflow ? (M > R ? M : R) : 0;
            counters[GS_CNT_VISIBLE] = V;
            counters[GS_CNT_ENTRIES] = overflow ? 0 : E;
            counters[GS_CNT_BIG] = 0;  // strip_sort_kernel queues the tiles whose list exceeds its LDS window
            counters[GS_CNT_GROUPS] = 0;  // big_list_sort_kernel queues the groups it cut for group_sort_kernel
            counters[GS_CNT_MAXLIST] = 0;  // strip_sort_kernel: longest list above GS_LONGEST_MIN pairs
            counters[GS_CNT_EXCESS] = 0;   // ... and the pairs beyond the first GS_LONG_MIN of their tile's list
            counters[GS_CNT_MAXWALK] = 0;  // raster_forward_kernel: longest walk a tile's wave made, steps beyond GS_LONG_MIN
            counters[GS_CNT_EXCESS_WALK] = 0;
            if (!gate) counters[GS_CNT_RANPAST] = 0;  // raised by a tile that ran past its occlusion cut (raster_fwd.hip)
        }
    }
    if (overflow) return;  // uniform
    // ---- run lengths of this slice = next slice's scanned value - this one's -> local starts (exclusive scan over strips)
    uint32_t mine = 0;
    {
        uint32_t eb = (uint32_t)be;  // first entry of strip t0
        for (uint32_t t = t0; t < t1; ++t) {
            const unsigned long long tot = t == t0 ? tot0 : strip_tot[t];
            const unsigned long long a = t == t0 ? a0 : row[t], b = t == t0 ? b0 : (last ? tot : next[t]);
            const uint32_t len = (uint32_t)((b - a) >> 32);  // the low halves never borrow: pairs(b) >= pairs(a)
            s_gd[t] = eb + (uint32_t)(a >> 32);  // final index of the run's first entry
            s_cur[t] = len;
            mine += len;
            eb += (uint32_t)(tot >> 32);
        }
    }
    uint32_t Ls;
    uint32_t run = strip_block_excl_scan(mine, s_wave, Ls);
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t len = s_cur[t];
        s_cur[t] = run;
        s_gd[t] -= run;  // wraps; s_gd[t] + staging index is the final index
        run += len;
    }
    // emission offset of this slice's first pair (training: per-pair gradient rows in Gaussian order)
    uint32_t before = 0, dummy;
    if (pair_offsets) {
        uint32_t p = 0;
        for (uint32_t b = threadIdx.x; b < S; b += STRIP_THREADS) p += b < slice ? slice_pairs[b] : 0;
        strip_block_excl_scan(p, s_wave, before);
    }
    __syncthreads();
    // ---- place
    for (uint32_t base = 0; base < count; base += STRIP_PF * STRIP_THREADS) {  // uniform trip counts (barriers inside)
        uint4 cur[STRIP_PF];
#pragma unroll
        for (int k = 0; k < STRIP_PF; ++k) {
            cur[k] = rc[k];
            rc[k] = L.rect(base + (STRIP_PF + k) * STRIP_THREADS);
        }
#pragma unroll
        for (int k = 0; k < STRIP_PF; ++k) {
            const uint32_t b = base + k * STRIP_THREADS;
            if (b >= count) break;  // uniform
            const uint32_t i = b + threadIdx.x;
            if (pair_offsets) {  // uniform: prefix sum of the rectangle areas in Gaussian order
                const uint32_t ex = strip_block_excl_scan(cur[k].w, s_wave, dummy);
                if (i < per_slice && L.g0 + i < n) pair_offsets[L.g0 + i] = before + ex;
                before += dummy;
            }
            auto place = [&](uint32_t strip, uint32_t lo32, uint32_t d, uint32_t) {
                const uint32_t slot = atomicAdd(&s_cur[strip], 1u);
                const unsigned long long e = ((unsigned long long)d << 32) | lo32;
                if (slot < cap)
                    s_stage[slot] = e;
                else
                    out[s_gd[strip] + slot] = e;
            };
            if (cut) {  // (uniform; a "dist" frame is never culled)
                if constexpr (!DIST) {
                    uint4 e = cur[k];
                    int64_t gid = L.g0 + i;
                    if (surv) {  // (uniform) a list entry: the Gaussian rides in .w, the area follows from the ranges
                        gid = e.w;
                        e.w = ((e.x >> 16) - (e.x & 0xffff)) * ((e.y >> 16) - (e.y & 0xffff));
                    }
                    walk_strips<false, true>(e, gid, SG, make_float2(0.f, 0.f), D, place, s_cut);
                }
            } else {
                walk_strips<DIST>(cur[k], L.g0 + i, SG, L.xy<DIST>(b, cur[k]), D, place);
            }
        }
    }
    __syncthreads();
    // ---- flush: run of strip t = staging [end of strip t - 1, end of strip t); a 16-lane group per run
    const uint32_t sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
    for (uint32_t t = grp; t < SG.NS; t += STRIP_THREADS / 16) {
        const uint32_t lo = t ? s_cur[t - 1] : 0, hi = s_cur[t] < cap ? s_cur[t] : cap, gd = s_gd[t];
        for (uint32_t i = lo + sub; i < hi; i += 16) out[gd + i] = s_stage[i];
    }
}

__global__ void __launch_bounds__(256) cut_dilate_kernel(const uint32_t *__restrict__ cut, uint32_t *__restrict__ out,
                                                         uint32_t ntx, uint32_t nty, float scale, uint32_t radius) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= ntx * nty) return;
    const uint32_t y = t / ntx, x = t - y * ntx;
    const uint32_t x0 = x > radius ? x - radius : 0, x1 = x + radius < ntx ? x + radius : ntx - 1;
    const uint32_t y0 = y > radius ? y - radius : 0, y1 = y + radius < nty ? y + radius : nty - 1;
    uint32_t m = 0;
    for (uint32_t j = y0; j <= y1; ++j)
        for (uint32_t i = x0; i <= x1; ++i) {
            const uint32_t c = cut[j * ntx + i];
            m = c > m ? c : m;
        }
    // (depth bits of positive floats order like the floats; GS_NO_CUT stays GS_NO_CUT)
    out[t] = m == GS_NO_CUT ? m : __float_as_uint(__uint_as_float(m) * scale);
}

}  // namespace

// GS_FRAME_CULL_DILATE: out[tile] = the largest cut of the tile's 3 x 3 neighbourhood (GS_NO_CUT is the largest value there
// is), pushed back by GS_CUT_DILATE_SCALE in depth
int gs_stage_cut_dilate(const gs_frame *f, const gs_frame_ws &ws, hipStream_t stream) {
    gs_frame_geom G = gs_frame_geometry(f);
    // (the environment variables: A/B sweeps, tools/batches/gpu_r6w.sh / gpu_r6x.sh -> profiles/r06_x_cull_moving_camera.txt)
    static const float scale = getenv("GS_CULL_DILATE_SCALE") ? (float)atof(getenv("GS_CULL_DILATE_SCALE")) : GS_CUT_DILATE_SCALE;
    static const float scale_near = getenv("GS_CULL_DILATE_SCALE") ? scale : GS_CUT_DILATE_SCALE_NEAR;
    static const uint32_t radius = getenv("GS_CULL_DILATE_RADIUS") ? (uint32_t)atoi(getenv("GS_CULL_DILATE_RADIUS")) : 1u;
    hipLaunchKernelGGL(cut_dilate_kernel, dim3((unsigned)gs_div_up(G.n_tiles, 256)), dim3(256), 0, stream,
                       (const uint32_t *)ws.cut, ws.cut_dilated, (uint32_t)G.ntx, (uint32_t)G.nty,
                       (f->flags & GS_FRAME_CULL_DILATE_NEAR) ? scale_near : scale, radius);
    GS_CHECK_LAUNCH();
    return 0;
}

// LDS of the count kernel: 8 B per strip; of the scatter kernel: 8 B per strip + the staging buffer
// `second_pass`: the untrimmed re-run of a GS_FRAME_OCCLUSION_CULL frame, every kernel gated on counters[GS_CNT_RANPAST]
int gs_stage_strip_bin(const gs_frame *f, const gs_frame_ws &ws, hipStream_t stream, bool second_pass) {
    gs_frame_geom G = gs_frame_geometry(f);
    const gs_strip_plan plan = gs_strip_plan_for(f->N, G.ntx, G.nty);
    const gs_strip_geom SG = plan.geom;
    const bool dist = f->tile_culling_method == 0;
    GsDistCull D = {(float)(G.padW / 2), (float)(G.padH / 2), f->focal_x, f->focal_y, f->thresh};
    static std::mutex attr_mu;
    static std::atomic<uint64_t> attr_done{0};
    int dev = 0;
    GS_HIP(hipGetDevice(&dev));
    if (dev < 64 && !((attr_done.load(std::memory_order_acquire) >> dev) & 1)) {
        std::lock_guard<std::mutex> lock(attr_mu);
        for (const void *fn : {(const void *)strip_count_kernel<false>, (const void *)strip_count_kernel<true>})
            GS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, GS_STRIP_MAX * 8));
        for (const void *fn : {(const void *)strip_scatter_kernel<false>, (const void *)strip_scatter_kernel<true>})
            GS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, GS_BIN_LDS_BYTES));
        attr_done.fetch_or(1ull << dev, std::memory_order_release);
    }
    unsigned long long *table = (unsigned long long *)ws.strip_table, *scan = table + (size_t)GS_BIN_SLICES * SG.NS;
    const size_t lds_count = sizeof(unsigned long long) * SG.NS;
    const unsigned long long *gate = second_pass ? ws.counters + GS_CNT_RANPAST : nullptr;
    const uint32_t *cut = (!second_pass && gs_frame_occlusion_cull(f)) ? gs_frame_cut_table(f, ws) : nullptr;
    // a culled frame's scatter gives 4 staged entries per strip (32 B: the strip's eight cuts) to the cut table
    const uint32_t cap = cut ? plan.cap - 4 * SG.NS : plan.cap;
    const size_t lds_scatter = sizeof(unsigned long long) * ((size_t)SG.NS + cap) + (cut ? (size_t)32 * SG.NS : 0);
    // (second pass: the table has been rewritten, untrimmed, by the gated re-run of the project stage)
    const unsigned long long *raw = table;
#define GS_LAUNCH_STRIP(DIST)                                                                                          \
    do {                                                                                                               \
        if (!second_pass && !gs_frame_fused_count(f)) { /* else: counted by the project stage */                         \
            hipLaunchKernelGGL(strip_count_kernel<DIST>, dim3(plan.slices + 1), dim3(STRIP_THREADS), lds_count, stream,\
                               ws.rects, ws.rec_geom, D, f->N, plan.per_slice, SG, plan.slices, table, ws.block_sums,  \
                               ws.block_vis, ws.slice_pairs, ws.slice_vis, ws.tile_cost, (uint32_t)G.n_tiles,          \
                               ws.tile_order, (const unsigned long long *)nullptr);                                    \
            GS_CHECK_LAUNCH();                                                                                         \
        }                                                                                                              \
        hipLaunchKernelGGL(strip_colscan_kernel, dim3((unsigned)gs_div_up(SG.NS, 16)), dim3(256), 0, stream, raw,      \
                           scan, plan.slices, SG.NS, (unsigned long long *)ws.strip_tot, gate);                        \
        GS_CHECK_LAUNCH();                                                                                             \
        hipLaunchKernelGGL(strip_scatter_kernel<DIST>, dim3(plan.slices), dim3(STRIP_THREADS), lds_scatter, stream,    \
                           ws.rects, ws.rec_geom, D, f->N, plan.per_slice, SG, plan.slices, cap, scan,                 \
                           (const unsigned long long *)ws.strip_tot, (unsigned long long *)ws.strip_base,              \
                           ws.slice_pairs, ws.slice_vis, (uint64_t)f->max_pairs, (unsigned long long *)ws.keys_a,      \
                           f->training ? ws.pair_offsets : nullptr, ws.counters, cut, (uint32_t)G.n_tiles, gate,       \
                           cut ? (const uint4 *)ws.surv : (const uint4 *)nullptr, (const uint32_t *)ws.slice_nsurv);    \
        GS_CHECK_LAUNCH();                                                                                             \
    } while (0)
    if (dist)
        GS_LAUNCH_STRIP(true);
    else
        GS_LAUNCH_STRIP(false);
#undef GS_LAUNCH_STRIP
    return 0;
}
