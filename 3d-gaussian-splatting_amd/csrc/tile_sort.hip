// tile_sort.hip -- sort_modes 1 and 2: finish the (tile, depth) order inside each tile's bucket.
//
// The 64-bit key is (tile_id | depth_bits), ties broken by the Gaussian index.  After the pairs have
// been grouped by tile (sort_mode 1: stable LSD radix passes on the tile bits; sort_mode 2: the LDS
// counting sort of tile_bin.hip) every bucket is ordered by the UNIQUE composite
// (depth_bits << 32 | gaussian_id) with an all-ascending bitonic network (flip + disperse), so the
// result is exactly the oracle's (tile, depth_bits, gaussian_index) order whatever the arrival order.
//
// A first version ran every stage of the network through LDS (two 8-byte reads + two writes per
// comparator and stage): PMC showed 1.47 M LDS instructions per cfg2 frame -- LDS bandwidth bound.
// Now a wave holds a 128-key WINDOW in registers, two keys per lane (element e = lane and lane + 64):
//   * strides < 64 are lane exchanges: xor 1, 2, 3, 7, 15 are single DPP moves (quad_perm, row_half_mirror,
//     row_mirror), xor 4 and 8 two chained DPP moves, xor 16 / 31 / 32 / 63 go through ds_bpermute (the LDS
//     crossbar, no memory access); stride 64 is a compare-exchange between the lane's two registers;
//   * a bucket of <= 128 keys never touches LDS memory at all (global -> registers -> global);
//   * longer buckets use LDS only for the strides >= 128 of the merge levels (k = 256, 512, ...): one
//     pass per such stride plus one load/store of the window for the seven strides 64..1.
// At 1024 keys that is 10 LDS passes instead of 55.  Buckets <= 512 keys are sorted by ONE wave (four
// tiles per workgroup, no workgroup barrier), <= 2048 by the workgroup, longer ones in global memory.
#include "gs_common.h"
#include "gs_frame_layout.h"

namespace {

#define KEY_INF 0xffffffffffffffffull  // padding of a window beyond the bucket: sorts to the end

template <typename Mem>
__device__ __forceinline__ void cmpx(Mem a, uint32_t lo, uint32_t hi, uint32_t n) {
    if (hi < n) {  // indices >= n hold a virtual +inf: every comparator is ascending, so skip
        const uint64_t x = a[lo], y = a[hi];
        if (x > y) {
            a[lo] = y;
            a[hi] = x;
        }
    }
}

// ---- lane exchanges -----------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ uint32_t dpp32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ uint64_t dpp64(uint64_t v) {
    return ((uint64_t)dpp32<CTRL>((uint32_t)(v >> 32)) << 32) | dpp32<CTRL>((uint32_t)v);
}
// value of lane (lane ^ M)
template <int M>
__device__ __forceinline__ uint64_t xlane(uint64_t v, int lane) {
    if constexpr (M == 1) return dpp64<0xB1>(v);        // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return dpp64<0x4E>(v);   // quad_perm [2,3,0,1]
    else if constexpr (M == 3) return dpp64<0x1B>(v);   // quad_perm [3,2,1,0]
    else if constexpr (M == 7) return dpp64<0x141>(v);  // row_half_mirror
    else if constexpr (M == 15) return dpp64<0x140>(v); // row_mirror
    else if constexpr (M == 4) return dpp64<0x1B>(dpp64<0x141>(v));   // ^7 then ^3
    else if constexpr (M == 8) return dpp64<0x141>(dpp64<0x140>(v));  // ^15 then ^7
    else {
        const int src = (lane ^ M) << 2;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)v);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)(v >> 32));
        return ((uint64_t)hi << 32) | lo;
    }
}
constexpr int top_bit(int m) { return m >= 32 ? 32 : m >= 16 ? 16 : m >= 8 ? 8 : m >= 4 ? 4 : m >= 2 ? 2 : 1; }

// compare-exchange with lane ^ M: the lane whose index is lower keeps the smaller key
template <int M>
__device__ __forceinline__ void cx(uint64_t &a, int lane) {
    const uint64_t b = xlane<M>(a, lane);
    const bool low = (lane & top_bit(M)) == 0;
    a = ((a < b) == low) ? a : b;
}
__device__ __forceinline__ void cx_regs(uint64_t &a0, uint64_t &a1) {  // stride 64: inside the lane
    const uint64_t lo = a0 < a1 ? a0 : a1, hi = a0 < a1 ? a1 : a0;
    a0 = lo;
    a1 = hi;
}
template <int J>
__device__ __forceinline__ void disperse_lanes(uint64_t &a, int lane) {  // strides J, J/2, .., 1
    if constexpr (J >= 1) {
        cx<J>(a, lane);
        disperse_lanes<J / 2>(a, lane);
    }
}
// Windows are padded with KEY_INF beyond their `c` real keys (which occupy the lowest elements).  A
// block of the network that holds only padding, or a stride that can only pair a key with padding
// above it, changes nothing, so every routine below stops (wave-uniformly) as soon as c is covered.
//
// sorts the 64 keys held one per lane (the first c of them real)
__device__ __forceinline__ void sort64(uint64_t &a, int lane, uint32_t c) {
    if (c <= 1) return;
    cx<1>(a, lane);
    if (c <= 2) return;
    cx<3>(a, lane);
    disperse_lanes<1>(a, lane);
    if (c <= 4) return;
    cx<7>(a, lane);
    disperse_lanes<2>(a, lane);
    if (c <= 8) return;
    cx<15>(a, lane);
    disperse_lanes<4>(a, lane);
    if (c <= 16) return;
    cx<31>(a, lane);
    disperse_lanes<8>(a, lane);
    if (c <= 32) return;
    cx<63>(a, lane);
    disperse_lanes<16>(a, lane);
}
// strides 32 .. 1 over one register whose first c keys are real
__device__ __forceinline__ void disperse64(uint64_t &a, int lane, uint32_t c) {
    if (c > 32) disperse_lanes<32>(a, lane);
    else if (c > 16) disperse_lanes<16>(a, lane);
    else if (c > 8) disperse_lanes<8>(a, lane);
    else if (c > 4) disperse_lanes<4>(a, lane);
    else if (c > 2) disperse_lanes<2>(a, lane);
    else if (c > 1) disperse_lanes<1>(a, lane);
}
// strides 64 .. 1 of a 128-key window (element e = lane in a0, lane + 64 in a1; first c keys real)
__device__ __forceinline__ void disperse_window(uint64_t &a0, uint64_t &a1, int lane, uint32_t c) {
    if (c > 64) {
        cx_regs(a0, a1);
        disperse_lanes<32>(a0, lane);
        disperse64(a1, lane, c - 64);
    } else {
        disperse64(a0, lane, c);
    }
}
// full sort of a 128-key window
__device__ __forceinline__ void sort_window(uint64_t &a0, uint64_t &a1, int lane, uint32_t c) {
    sort64(a0, lane, c < 64 ? c : 64);
    if (c <= 64) return;
    sort64(a1, lane, c - 64);
    // flip of span 128: e <-> 127 - e, i.e. a0[lane] with a1[63 - lane]
    const uint64_t b0 = xlane<63>(a1, lane), b1 = xlane<63>(a0, lane);
    a0 = a0 < b0 ? a0 : b0;
    a1 = a1 < b1 ? b1 : a1;
    disperse_window(a0, a1, lane, c);
}

// ---- merge levels k >= 256 on an LDS array whose 128-key windows are sorted ----------------------
// `nthreads` threads with index `tid` cooperate (one wave, or the workgroup); `sync` orders the stages.
template <typename Sync>
__device__ __forceinline__ void merge_levels(uint64_t *a, uint32_t n, uint32_t P, uint32_t tid, uint32_t nthreads,
                                             Sync sync) {
    const int lane = tid & 63;
    const uint32_t wave = tid >> 6, nwaves = nthreads >> 6, nwin = (n + 127) / 128;
    for (uint32_t k = 256; k <= P; k <<= 1) {
        const uint32_t hk = k >> 1;
        for (uint32_t t = tid; t < (P >> 1); t += nthreads)  // flip
            cmpx(a, (t / hk) * k + (t % hk), (t / hk) * k + (k - 1) - (t % hk), n);
        sync();
        for (uint32_t j = hk >> 1; j >= 128; j >>= 1) {  // strides that cross windows
            for (uint32_t t = tid; t < (P >> 1); t += nthreads) {
                const uint32_t lo = (t / j) * 2 * j + (t % j);
                cmpx(a, lo, lo + j, n);
            }
            sync();
        }
        for (uint32_t w = wave; w < nwin; w += nwaves) {  // strides 64 .. 1 in registers
            const uint32_t e0 = w * 128 + lane, e1 = e0 + 64;
            uint64_t a0 = e0 < n ? a[e0] : KEY_INF, a1 = e1 < n ? a[e1] : KEY_INF;
            disperse_window(a0, a1, lane, n - w * 128 < 128 ? n - w * 128 : 128);
            if (e0 < n) a[e0] = a0;
            if (e1 < n) a[e1] = a1;
        }
        sync();
    }
}

// Strides top .. 128 through the array, then 64 .. 1 in registers: the tail of a merge level whose wider
// strides have already been applied (used per CAP-sized chunk of a bucket that does not fit LDS).
template <typename Sync>
__device__ __forceinline__ void disperse_levels(uint64_t *a, uint32_t n, uint32_t top, uint32_t tid, uint32_t nthreads,
                                                Sync sync) {
    const int lane = tid & 63;
    const uint32_t wave = tid >> 6, nwaves = nthreads >> 6, nwin = (n + 127) / 128;
    for (uint32_t j = top; j >= 128; j >>= 1) {
        // comparator t pairs lo(t) with lo(t) + j; both grow with t, so a thread can stop at its first miss
        for (uint32_t t = tid; (t / j) * 2 * j + (t % j) + j < n; t += nthreads) {
            const uint32_t lo = (t / j) * 2 * j + (t % j);
            cmpx(a, lo, lo + j, n);
        }
        sync();
    }
    for (uint32_t w = wave; w < nwin; w += nwaves) {
        const uint32_t e0 = w * 128 + lane, e1 = e0 + 64;
        uint64_t a0 = e0 < n ? a[e0] : KEY_INF, a1 = e1 < n ? a[e1] : KEY_INF;
        disperse_window(a0, a1, lane, n - w * 128 < 128 ? n - w * 128 : 128);
        if (e0 < n) a[e0] = a0;
        if (e1 < n) a[e1] = a1;
    }
    sync();
}

// ---------------------------------------------------------------------------------------------------------------
// Distribution sort of the list a[0 .. n) in LDS by the unique key (depth_bits << 32 | gaussian), result handed to
// store(position, key).  The bitonic network above costs ~55 compare-exchange stages of 64-bit keys per element at
// n = 1024 (PMC, profiles/r02_a: 43.5 M VALU wave-instructions for 6.95 M keys = 400 per key: the per-tile sort was
// VALU-bound at 99 us).  Depth bits of a tile's list are spread over a range, so ONE counting pass over B >= n
// buckets that are linear in the depth bits (monotone: float conversion, multiplication by a positive constant and
// truncation all are) leaves ~1 key per bucket:
//   keys -> registers, min / max of the depth bits -> bucket of every key, rank inside the bucket from ds_add_rtn ->
//   exclusive scan of the B counters -> keys written back in bucket order -> final position = bucket start + number
//   of smaller keys in the bucket (a loop over the bucket: buckets are tiny) -> store.
// Buckets with more than BUCKET_SMALL keys (depth clusters: many Gaussians on one surface; equal depths) are queued
// and sorted in place by the bitonic network, one wave per bucket (the workgroup for more than 512 keys): the worst
// case -- all keys in one bucket -- costs what the network cost before.
// `nthreads` threads with index `tid` cooperate (a wave: n <= 512, sync = wave barrier; the 256-thread workgroup:
// n <= 2048, sync = __syncthreads); cnt: nthreads x 8 + 1 words, wl: 1 + 2 x 128 words, red: 16 words of LDS.
#define BUCKET_SMALL 16u
// `nsplit` < n (round 3): a[0 .. nsplit) and a[nsplit .. n) are TWO lists (adjacent tiles) sorted by one call -- every list
// gets its own share of the buckets and its own depth range, all buckets of the first list come before those of the
// second, so the result is the first list sorted, then the second list sorted, at the same positions.  One call for
// two lists of ~850 keys instead of two calls halves the fixed cost (six workgroup barriers, the counters' clear and
// scan, eight-way unrolled loops that were less than half full).
template <typename Sync, typename Store>
__device__ __forceinline__ void bucket_sort_store(uint64_t *a, uint32_t n, uint32_t *cnt, uint32_t *wl, uint32_t *red,
                                                  uint32_t tid, uint32_t nthreads, Sync sync, Store store,
                                                  uint32_t nsplit = 0xffffffffu) {
    constexpr int R = 8;
    if (nsplit >= n || nsplit == 0) nsplit = n;  // a single list
    const int lane = tid & 63;
    const uint32_t wave = tid >> 6, nwaves = nthreads >> 6;
    const uint32_t per = (n + nthreads - 1) / nthreads;                 // <= 8
    const uint32_t CPT = per <= 1 ? 1 : per <= 2 ? 2 : per <= 4 ? 4 : 8;  // counters per thread
    const uint32_t B = nthreads * CPT;                                  // n <= B < 2 n (or B = nthreads)
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // ---- keys to registers, range of the depth bits
    uint64_t k[R];
    const bool two = nsplit < n;  // uniform
    uint32_t dmin = 0xffffffffu, dmax = 0, dmin1 = 0xffffffffu, dmax1 = 0;  // (.., ..1): the second list
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t i = tid + r * nthreads;
        k[r] = i < n ? a[i] : KEY_INF;
        if (i < n) {
            const uint32_t d = (uint32_t)(k[r] >> 32);
            if (i < nsplit) {
                dmin = d < dmin ? d : dmin;
                dmax = d > dmax ? d : dmax;
            } else {
                dmin1 = d < dmin1 ? d : dmin1;
                dmax1 = d > dmax1 ? d : dmax1;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t x = __shfl_xor(dmin, o, 64), y = __shfl_xor(dmax, o, 64);
        dmin = x < dmin ? x : dmin;
        dmax = y > dmax ? y : dmax;
        if (two) {
            const uint32_t x1 = __shfl_xor(dmin1, o, 64), y1 = __shfl_xor(dmax1, o, 64);
            dmin1 = x1 < dmin1 ? x1 : dmin1;
            dmax1 = y1 > dmax1 ? y1 : dmax1;
        }
    }
    for (uint32_t c = tid; c <= B; c += nthreads) cnt[c] = 0;
    if (tid == 0) wl[0] = 0;
    if (nwaves > 1) {
        if (lane == 0) {
            red[wave] = dmin;
            red[4 + wave] = dmax;
            red[8 + wave] = dmin1;
            red[12 + wave] = dmax1;
        }
        sync();
#pragma unroll
        for (uint32_t w = 0; w < 4; ++w) {
            if (w < nwaves) {
                dmin = red[w] < dmin ? red[w] : dmin;
                dmax = red[4 + w] > dmax ? red[4 + w] : dmax;
                dmin1 = red[8 + w] < dmin1 ? red[8 + w] : dmin1;
                dmax1 = red[12 + w] > dmax1 ? red[12 + w] : dmax1;
            }
        }
        // (red[8 ..] is written again by the scan below -- behind the barrier that follows the ranking loop)
    } else {
        sync();
    }
    // ---- bucket of every key, rank inside the bucket in arrival order.  Two lists: buckets [0, B0) belong to the
    // first, [B0, B) to the second, shared out by their lengths
    uint32_t B0 = B;
    if (two) {
        B0 = (uint32_t)(((uint64_t)B * nsplit) / n);
        B0 = B0 < 1 ? 1 : (B0 > B - 1 ? B - 1 : B0);
    }
    const float scale = (float)B0 / ((float)(dmax - dmin) + 1.0f);
    const float scale1 = two ? (float)(B - B0) / ((float)(dmax1 - dmin1) + 1.0f) : 0.f;
    uint32_t bk[R], rk[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        bk[r] = rk[r] = 0;
        const uint32_t i = tid + r * nthreads;
        if (i < n) {
            const uint32_t d = (uint32_t)(k[r] >> 32);
            if (i < nsplit) {
                const uint32_t b = (uint32_t)((float)(d - dmin) * scale);
                bk[r] = b < B0 - 1 ? b : B0 - 1;
            } else {
                const uint32_t b = B0 + (uint32_t)((float)(d - dmin1) * scale1);
                bk[r] = b < B - 1 ? b : B - 1;
            }
            rk[r] = atomicAdd(&cnt[bk[r]], 1u);
        }
    }
    sync();
    // ---- exclusive scan of the counters (thread t owns counters [t CPT, t CPT + CPT))
    {
        uint32_t c[8], sum = 0;
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
            c[j] = j < CPT ? cnt[tid * CPT + j] : 0;
            const uint32_t x = c[j];
            c[j] = sum;
            sum += x;
        }
        const uint32_t incl = gs_wave_incl_scan_u32(sum);
        uint32_t off = incl - sum;
        if (nwaves > 1) {
            if (lane == 63) red[8 + wave] = incl;
            sync();
#pragma unroll
            for (uint32_t w = 0; w < 4; ++w) off += w < wave ? red[8 + w] : 0;
        }
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j)
            if (j < CPT) cnt[tid * CPT + j] = off + c[j];
        if (tid == 0) cnt[B] = n;
    }
    sync();
    // ---- keys back in bucket order (in place: every key is in a register)
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (tid + r * nthreads < n) a[cnt[bk[r]] + rk[r]] = k[r];
    sync();
    // ---- final position inside the bucket; large buckets are queued
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (tid + r * nthreads < n) {
            const uint32_t bs = cnt[bk[r]], be = cnt[bk[r] + 1];
            if (be - bs <= BUCKET_SMALL) {
                uint32_t c = 0;
                for (uint32_t j = bs; j < be; ++j) c += a[j] < k[r] ? 1u : 0u;
                store(bs + c, k[r]);
            } else if (rk[r] == 0) {
                const uint32_t slot = atomicAdd(&wl[0], 1u);
                wl[1 + 2 * slot] = bs;
                wl[2 + 2 * slot] = be - bs;
            }
        }
    }
    sync();
    const uint32_t nwl = wl[0];
    if (nwl == 0) return;  // uniform
    auto sort_windows = [&](uint64_t *b, uint32_t m, uint32_t w0, uint32_t wstep) {
        const uint32_t nwin = (m + 127) / 128;
        for (uint32_t w = w0; w < nwin; w += wstep) {
            const uint32_t f0 = w * 128 + lane, f1 = f0 + 64;
            uint64_t a0 = f0 < m ? b[f0] : KEY_INF, a1 = f1 < m ? b[f1] : KEY_INF;
            sort_window(a0, a1, lane, m - w * 128 < 128 ? m - w * 128 : 128);
            if (f0 < m) b[f0] = a0;
            if (f1 < m) b[f1] = a1;
        }
    };
    for (uint32_t i = wave; i < nwl; i += nwaves) {  // one wave per queued bucket of up to 512 keys
        const uint32_t bs = wl[1 + 2 * i], m = wl[2 + 2 * i];
        if (m > 512) continue;
        uint64_t *b = a + bs;
        sort_windows(b, m, 0, 1);
        wave_sync();
        if (m > 128) {
            uint32_t P = 256;
            while (P < m) P <<= 1;
            merge_levels(b, m, P, (uint32_t)lane, 64u, wave_sync);
        }
        for (uint32_t j = lane; j < m; j += 64) store(bs + j, b[j]);
    }
    if (nwaves > 1) {
        for (uint32_t i = 0; i < nwl; ++i) {  // longer ones by the workgroup (uniform)
            const uint32_t bs = wl[1 + 2 * i], m = wl[2 + 2 * i];
            if (m <= 512) continue;
            uint64_t *b = a + bs;
            sync();
            sort_windows(b, m, wave, nwaves);
            sync();
            uint32_t P = 256;
            while (P < m) P <<= 1;
            merge_levels(b, m, P, tid, nthreads, sync);
            for (uint32_t j = tid; j < m; j += nthreads) store(bs + j, b[j]);
        }
    }
}

// Where a tile's unsorted pairs come from.
//   SRC_KEYS   (sort_mode 1): keys = (tile << 32 | depth_bits) grouped by tile, ids = Gaussian index; both are
//              overwritten in place with the depth-sorted order; scratch = idle half of the key buffer.
//   SRC_PACKED (sort_mode 2, table variant): scratch = (depth_bits << 32 | gaussian) grouped by tile in arbitrary
//              order; the sorted ids go to ids, the sorted (tile << 32 | depth_bits) to keys unless keys is NULL;
//              long buckets sort in place.
//   SRC_GATHER (sort_mode 2, slice-sorted variant): the pairs of tile t sit in S slice regions of `pairs`, region s
//              holding them at [slice_base[s] + table[s][t], slice_base[s] + table[s][t + 1]); they are gathered
//              into LDS while the tile is loaded (buckets beyond CAP: into scratch + start, then sorted there).
enum { SRC_KEYS = 0, SRC_PACKED = 1, SRC_GATHER = 2 };
struct GatherSrc {
    const uint64_t *pairs;
    const uint32_t *table;       // [S][T + 1]
    const uint32_t *slice_base;  // [S]
    uint32_t S, T;
};

// Sorts the (up to) four buckets `sel(q, tile, start, n)` names, q = 0 .. 3, with the 256 threads of the workgroup.
// s_a: CAP keys of LDS; s_scan: 4 words; s_off: the SRC_GATHER offset table (filled by the caller).
template <int CAP, int SRC, typename Sel>
__device__ __forceinline__ void tile_sort_body(uint64_t *__restrict__ keys, uint32_t *__restrict__ ids,
                                               uint64_t *__restrict__ scratch, Sel sel, const GatherSrc &GS,
                                               uint64_t *s_a, uint32_t *s_scan, const uint32_t *s_off,
                                               uint32_t *s_bcnt, uint32_t *s_wl, uint32_t *s_red) {
    static_assert(CAP == 2048, "the distribution sort holds eight keys per thread: 512 per wave, 2048 per workgroup");
    constexpr bool PACKED = SRC != SRC_KEYS;  // the unsorted element already is (depth_bits << 32 | gaussian)
    const int lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t tile, start, n;
    auto load = [&](uint32_t i) -> uint64_t {
        return PACKED ? scratch[start + i] : (keys[start + i] << 32) | ids[start + i];
    };
    // PACKED: the sorted keys are only written on request (GS_FRAME_EMIT_SORTED_KEYS): the raster kernels read the
    // sorted ids alone, and 8 of the 12 bytes this kernel would store per pair are the keys
    auto store = [&](uint32_t i, uint64_t v) {
#ifdef GS_DIAG_SCATTER_SMALL  // the scatter experiment leaves garbage pairs: keep the ids inside any scene >= 64 k
        v &= 0xffffffff0000ffffull;
#endif
        ids[start + i] = (uint32_t)v;
        if (!PACKED || keys) keys[start + i] = ((uint64_t)tile << 32) | (v >> 32);
    };
    uint32_t cur_q = 0;
    auto select = [&](uint32_t q) {
        cur_q = q;
        sel(q, tile, start, n);
    };
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // SRC_GATHER: copies the current tile's pairs from the S slice regions to dst[0 .. n) (any order: the sort that
    // follows orders by the unique (depth_bits, gaussian)).  `nthreads` threads with index `tid` cooperate (one wave
    // or the workgroup: `wg` selects the cross-wave prefix); the first pair of four slices per thread is in flight
    // at a time (most (slice, tile) cells hold one or two pairs).
    auto gather = [&](uint64_t *dst, uint32_t tid, uint32_t nthreads, bool wg) {
        uint32_t filled = 0;
        const uint32_t *o_lo = s_off + cur_q * GS.S, *o_hi = o_lo + GS.S;
        for (uint32_t s0 = 0; s0 < GS.S; s0 += 4 * nthreads) {
            uint32_t a[4], c[4], d[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t sl = s0 + q * nthreads + tid;
                a[q] = c[q] = 0;
                if (sl < GS.S) {
                    a[q] = o_lo[sl];
                    c[q] = o_hi[sl] - a[q];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t incl = gs_wave_incl_scan_u32(c[q]);
                uint32_t off = incl - c[q], total = __shfl(incl, 63, 64);
                if (wg) {  // uniform: prefix over the four waves
                    __syncthreads();
                    if (lane == 63) s_scan[wave] = incl;
                    __syncthreads();
                    total = 0;
#pragma unroll
                    for (uint32_t w = 0; w < 4; ++w) {
                        off += w < wave ? s_scan[w] : 0;
                        total += s_scan[w];
                    }
                }
                d[q] = filled + off;
                filled += total;
            }
            uint64_t first[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) first[q] = c[q] ? GS.pairs[a[q]] : 0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (c[q]) dst[d[q]] = first[q];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                for (uint32_t k = 1; k < c[q]; ++k) dst[d[q] + k] = GS.pairs[a[q] + k];
        }
    };
    // loads window w of the current bucket into registers, sorts it, hands it to `out(e, key)`
    auto sort_window_from = [&](auto in, uint32_t w, auto out) {
        const uint32_t e0 = w * 128 + lane, e1 = e0 + 64;
        uint64_t a0 = e0 < n ? in(e0) : KEY_INF, a1 = e1 < n ? in(e1) : KEY_INF;
        sort_window(a0, a1, lane, n - w * 128 < 128 ? n - w * 128 : 128);
        if (e0 < n) out(e0, a0);
        if (e1 < n) out(e1, a1);
    };

    // 1. one wave per short bucket (<= CAP/4 keys), four buckets per workgroup, no workgroup barrier
    select(wave);
    if (n >= (PACKED ? 1u : 2u) && n <= (uint32_t)CAP / 4) {
        uint64_t *a = s_a + wave * (CAP / 4);
        if (SRC == SRC_GATHER) {
            gather(a, (uint32_t)lane, 64u, false);
            wave_sync();
        }
        auto from_lds = [&](uint32_t e) -> uint64_t { return a[e]; };
        if (n <= 128) {  // registers only (SRC_GATHER: through the wave's LDS window)
            if (SRC == SRC_GATHER)
                sort_window_from(from_lds, 0, store);
            else
                sort_window_from(load, 0, store);
        } else {  // the distribution sort (bucket_sort_store) on the wave's LDS window
            if (SRC != SRC_GATHER) {
                for (uint32_t i = lane; i < n; i += 64) a[i] = load(i);
                wave_sync();
            }
            bucket_sort_store(a, n, s_bcnt + wave * 513, s_wl + wave * 257, s_red, (uint32_t)lane, 64u, wave_sync, store);
        }
    }
    // 2. long buckets, one after the other, by the whole workgroup (n is uniform => so are the barriers)
    for (uint32_t q = 0; q < 4; ++q) {
        select(q);
        if (n <= (uint32_t)CAP / 4) continue;
        __syncthreads();
        uint32_t P = 256;
        while (P < n) P <<= 1;
        if (n <= (uint32_t)CAP) {
            if (SRC == SRC_GATHER)
                gather(s_a, threadIdx.x, 256u, true);
            else
                for (uint32_t i = threadIdx.x; i < n; i += 256) s_a[i] = load(i);
            __syncthreads();
            bucket_sort_store(s_a, n, s_bcnt, s_wl, s_red, threadIdx.x, 256u, [] { __syncthreads(); }, store);
        } else {
            // The bucket does not fit the LDS window: sort it CAP keys at a time in LDS, then finish the merge levels
            // k = 2 CAP, 4 CAP, .. with their strides >= CAP through global memory (the bucket stays in L2) and
            // everything below per chunk in LDS / registers again.  (A first version ran the whole network through
            // global memory: 4.9 ms per frame at 3,500 pairs per tile; this path: see DESIGN.md.)
            uint64_t *a = scratch + start;  // PACKED: in place; else the idle half of the key buffer
            if (SRC == SRC_GATHER)
                gather(a, threadIdx.x, 256u, true);
            else if (!PACKED)
                for (uint32_t i = threadIdx.x; i < n; i += 256) a[i] = load(i);
            __syncthreads();
            auto block_sync = [] { __syncthreads(); };
            const uint32_t nchunk = (n + CAP - 1) / CAP;
            auto chunk_in = [&](uint32_t c) {
                const uint32_t cnt = n - c * CAP < (uint32_t)CAP ? n - c * CAP : (uint32_t)CAP;
                for (uint32_t i = threadIdx.x; i < cnt; i += 256) s_a[i] = a[c * CAP + i];
                __syncthreads();
                return cnt;
            };
            auto chunk_out = [&](uint32_t c, uint32_t cnt, bool final) {
                for (uint32_t i = threadIdx.x; i < cnt; i += 256) {
                    if (final)
                        store(c * CAP + i, s_a[i]);
                    else
                        a[c * CAP + i] = s_a[i];
                }
                __syncthreads();
            };
            for (uint32_t c = 0; c < nchunk; ++c) {  // every chunk sorted on its own
                const uint32_t cnt = chunk_in(c);
                const uint32_t nwin = (cnt + 127) / 128;
                for (uint32_t w = wave; w < nwin; w += 4) {
                    const uint32_t e0 = w * 128 + lane, e1 = e0 + 64;
                    uint64_t a0 = e0 < cnt ? s_a[e0] : KEY_INF, a1 = e1 < cnt ? s_a[e1] : KEY_INF;
                    sort_window(a0, a1, lane, cnt - w * 128 < 128 ? cnt - w * 128 : 128);
                    if (e0 < cnt) s_a[e0] = a0;
                    if (e1 < cnt) s_a[e1] = a1;
                }
                __syncthreads();
                uint32_t Pc = 256;
                while (Pc < cnt) Pc <<= 1;
                merge_levels(s_a, cnt, Pc, threadIdx.x, 256u, block_sync);
                chunk_out(c, cnt, false);
            }
            for (uint32_t k = 2 * CAP; k <= P; k <<= 1) {
                const uint32_t hk = k >> 1;
                for (uint32_t t = threadIdx.x; t < (P >> 1); t += 256)  // flip, through global memory
                    cmpx(a, (t / hk) * k + (t % hk), (t / hk) * k + (k - 1) - (t % hk), n);
                __syncthreads();
                for (uint32_t j = hk >> 1; j >= (uint32_t)CAP; j >>= 1) {  // strides that cross chunks
                    for (uint32_t t = threadIdx.x; t < (P >> 1); t += 256) {
                        const uint32_t lo = (t / j) * 2 * j + (t % j);
                        cmpx(a, lo, lo + j, n);
                    }
                    __syncthreads();
                }
                for (uint32_t c = 0; c < nchunk; ++c) {  // strides CAP/2 .. 1 inside every chunk
                    const uint32_t cnt = chunk_in(c);
                    disperse_levels(s_a, cnt, CAP / 2, threadIdx.x, 256u, block_sync);
                    chunk_out(c, cnt, k == P);
                }
            }
        }
    }
}

template <int CAP, int SRC>
__global__ void __launch_bounds__(256) tile_sort_kernel(uint64_t *__restrict__ keys, uint32_t *__restrict__ ids,
                                                       uint64_t *__restrict__ scratch,
                                                       const int32_t *__restrict__ ranges, uint32_t n_tiles,
                                                       GatherSrc GS) {
    __shared__ uint64_t s_a[CAP];
    __shared__ uint32_t s_scan[4];
    // SRC_GATHER: where the pairs of this workgroup's four tiles start in every slice region (absolute index into
    // `pairs`), tiles t0 .. t0 + 4 (the fifth column closes the fourth tile): row j holds S entries.  Loaded once per
    // workgroup -- thread = slice, five consecutive words of its table row: ONE cache line per slice for all four
    // tiles -- instead of two scattered 4-byte reads per slice and tile.
    __shared__ uint32_t s_off[SRC == SRC_GATHER ? 5 * GS_BIN_MAX_SLICES : 1];
    __shared__ uint32_t s_bcnt[4 * 513 + 4], s_wl[4 * 257], s_red[16];  // distribution sort (bucket_sort_store)
    if (SRC == SRC_GATHER) {
        const uint32_t t0 = blockIdx.x * 4;
        const size_t stride = (size_t)GS.T + 1;
        for (uint32_t sl = threadIdx.x; sl < GS.S; sl += 256) {
            const uint32_t *r = GS.table + sl * stride;
            const uint32_t b = GS.slice_base[sl];
#pragma unroll
            for (uint32_t j = 0; j < 5; ++j) s_off[j * GS.S + sl] = b + r[t0 + j < GS.T ? t0 + j : GS.T];
        }
        __syncthreads();
    }
    tile_sort_body<CAP, SRC>(
        keys, ids, scratch,
        [&](uint32_t q, uint32_t &tile, uint32_t &start, uint32_t &n) {
            tile = blockIdx.x * 4 + q;
            start = n = 0;
            if (tile < n_tiles) {
                start = (uint32_t)ranges[2 * tile];
                n = (uint32_t)ranges[2 * tile + 1] - start;
            }
        },
        GS, s_a, s_scan, s_off, s_bcnt, s_wl, s_red);

}

// ---------------------------------------------------------------------------------------------------------------
// STRIP variant of sort_mode 2 (level 2; level 1 is strip_bin.hip): one workgroup per HALF strip = four consecutive
// tiles of one tile row.  The strip's entries (depth_bits << 32 | first covered tile << 29 | last << 26 | gaussian),
// contiguous in `entries`, are read twice (the second time out of L2): once to count the pairs of every tile of the
// strip -- which gives this workgroup's tile ranges -- and once to place (depth_bits << 32 | gaussian) into the four
// tiles' lists.  A half strip of up to CAP pairs keeps its lists in LDS and sorts them there: the unsorted pairs never
// exist in global memory.  Beyond CAP the lists are placed in `scratch` (indexed like the final list: four streams
// per workgroup) and sorted by the per-tile code of the table variant.
// DIST: a tile is listed iff gs_dist_listed says so for the Gaussian's centre (the same test as in level 1).
template <int CAP, bool DIST>
#ifndef STRIP_SORT_WPE
#define STRIP_SORT_WPE 4  // waves per SIMD the register allocation aims at (A/B switch, tools/ab_variants.py)
#endif
__global__ void __launch_bounds__(256, STRIP_SORT_WPE) strip_sort_kernel(
    const uint64_t *__restrict__ entries, const uint64_t *__restrict__ strip_base,
    const uint64_t *__restrict__ strip_tot, const unsigned long long *__restrict__ counters,
    uint64_t *__restrict__ keys, uint32_t *__restrict__ ids, uint64_t *__restrict__ scratch,
    int32_t *__restrict__ ranges, gs_strip_geom SG, const float4 *__restrict__ rec_geom, GsDistCull D,
    uint32_t *__restrict__ big_queue, unsigned long long *__restrict__ big_count,
    unsigned long long *__restrict__ longest, const unsigned long long *__restrict__ gate) {
    constexpr uint32_t W = GS_STRIP_W, WAVE_MAX = 512, ID_MASK = (1u << GS_STRIP_ID_BITS) - 1;
    if (gate && *gate == 0) return;  // (see group_sort_kernel)
    static_assert(GS_STRIP_W == 8 || GS_STRIP_W == 4, "a workgroup owns four tiles: a strip or half a strip");
    __shared__ uint64_t s_a[CAP];
    __shared__ uint32_t s_scan[4];
    __shared__ uint32_t s_cnt[4][W];
    __shared__ uint32_t s_tile[4], s_start[4], s_n[4];
    // distribution sort (bucket_sort_store): counters of the workgroup (256 x 8 + 1) or of four waves (64 x 8 + 1 each),
    // queues of large buckets, cross-wave scratch
    __shared__ uint32_t s_bcnt[4 * 513 + 4], s_wl[4][1 + 2 * 128], s_red[16];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6;
    // the two halves of a strip are 8 workgroup ids apart: dealt to the same XCD, they share the entries in its L2
    const uint32_t strip = W == 8 ? (blockIdx.x >> 4) * 8 + (blockIdx.x & 7) : blockIdx.x;
    const uint32_t half = W == 8 ? (blockIdx.x >> 3) & 1 : 0;
    if (strip >= SG.NS) return;
    const uint32_t row = strip / SG.nsx, sx = strip - row * SG.nsx, tx0 = sx * W + half * 4;
    if (tx0 >= SG.ntx) return;  // the strip ends in its first half
    // the three loads are independent: issued together, one memory round trip in front of the entries instead of two
    const unsigned long long overflow = counters[GS_CNT_OVERFLOW];
    const uint64_t base = strip_base[strip], tot = strip_tot[strip];
    if (overflow) {  // the frame did not fit: every tile reads (0, 0)
        if (threadIdx.x < 4 && tx0 + threadIdx.x < SG.ntx)
            reinterpret_cast<int2 *>(ranges)[row * SG.ntx + tx0 + threadIdx.x] = make_int2(0, 0);
        return;
    }
    const uint32_t E = (uint32_t)(tot >> 32), e0 = (uint32_t)(base >> 32), p0 = (uint32_t)base;
    const uint64_t *ent = entries + e0;
    auto covers = [&](bool ok, uint32_t lo32, float2 xy, uint32_t j) {
        const uint32_t lx0 = lo32 >> 29, lx1 = (lo32 >> 26) & 7;  // [lx0, lx1]
        bool c = ok && j >= lx0 && j <= lx1;
        if (DIST) c = c && gs_dist_listed(xy.x, xy.y, sx * W + j, row, D);
        return c;
    };
    auto centre = [&](bool ok, uint32_t lo32) {
        if (!DIST || !ok) return make_float2(0.f, 0.f);
        const float4 ge = rec_geom[(size_t)(lo32 & ID_MASK) * GS_REC_STRIDE];
        return make_float2(ge.x, ge.y);
    };
    // Entries are handled in chunks of 16 x 256: all 16 loads of a thread are in flight at once (a loop of dependent
    // load -> ballot rounds exposed one memory round trip per 256 entries: 150 us for this kernel at 2.4 M
    // Gaussians), and a strip of up to 4096 entries -- the common case -- is read from memory only once.
    constexpr uint32_t EPT = 16, CHUNK = EPT * 256;
    const uint32_t nchunk = (E + CHUNK - 1) / CHUNK;
    uint64_t er[EPT];
    auto load_chunk = [&](uint32_t c) {
#pragma unroll
        for (uint32_t k = 0; k < EPT; ++k) {
            const uint32_t i = c * CHUNK + k * 256 + threadIdx.x;
            er[k] = i < E ? ent[i] : ~0ull;  // run [7, 7] of Gaussian 2^26 - 1 ... masked by `ok` below
        }
    };
    // ---- 1. pairs of every tile of the strip, per wave (wave-uniform counts from ballots)
    uint32_t c8[W];
#pragma unroll
    for (uint32_t j = 0; j < W; ++j) c8[j] = 0;
    for (uint32_t c = 0; c < nchunk; ++c) {
        load_chunk(c);
        if constexpr (DIST) {
#pragma unroll
            for (uint32_t k = 0; k < EPT; ++k) {
                if (c * CHUNK + k * 256 >= E) break;  // uniform
                const bool ok = c * CHUNK + k * 256 + threadIdx.x < E;
                const uint32_t lo32 = (uint32_t)er[k];
                const float2 xy = centre(ok, lo32);
#pragma unroll
                for (uint32_t j = 0; j < W; ++j) c8[j] += (uint32_t)__popcll(__ballot(covers(ok, lo32, xy, j)));
            }
        } else {
            // per lane: the run's 8-bit tile mask spread to one byte per tile (4 tiles per register: bit i of the
            // nibble times (1 + 2^7 + 2^14 + 2^21) lands on bit 8 i, the other partial products on bits that are masked
            // away), summed over the lane's <= 16 entries; then 16-bit fields summed over the wave (<= 1024)
            uint32_t acc_lo = 0, acc_hi = 0;
#pragma unroll
            for (uint32_t k = 0; k < EPT; ++k) {
                if (c * CHUNK + k * 256 >= E) break;  // uniform
                const bool ok = c * CHUNK + k * 256 + threadIdx.x < E;
                const uint32_t lo32 = (uint32_t)er[k];
                const uint32_t lx0 = lo32 >> 29, lx1 = (lo32 >> 26) & 7;
                const uint32_t m = ok ? ((2u << lx1) - 1u) & ~((1u << lx0) - 1u) : 0u;
                acc_lo += __umul24(m & 15u, 0x204081u) & 0x01010101u;
                acc_hi += __umul24(m >> 4, 0x204081u) & 0x01010101u;
            }
            const uint32_t w02 = gs_wave_sum_u32(acc_lo & 0x00ff00ffu), w13 = gs_wave_sum_u32((acc_lo >> 8) & 0x00ff00ffu);
            const uint32_t w46 = gs_wave_sum_u32(acc_hi & 0x00ff00ffu), w57 = gs_wave_sum_u32((acc_hi >> 8) & 0x00ff00ffu);
            c8[0] += w02 & 0xffff;
            c8[2] += w02 >> 16;
            c8[1] += w13 & 0xffff;
            c8[3] += w13 >> 16;
            if constexpr (W == 8) {
                c8[4] += w46 & 0xffff;
                c8[6] += w46 >> 16;
                c8[5] += w57 & 0xffff;
                c8[7] += w57 >> 16;
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (uint32_t j = 0; j < W; ++j) s_cnt[wave][j] = c8[j];
    }
    __syncthreads();
    // wave w places its pairs of tile q behind those of the waves before it: no atomics, and the same (wave, round)
    // order in both passes makes the placement deterministic
    uint32_t n4[4], wb4[4], st = p0;
#pragma unroll
    for (uint32_t j = 0; j < W; ++j) {
        const uint32_t c0 = s_cnt[0][j], c1 = s_cnt[1][j], c2 = s_cnt[2][j], c3 = s_cnt[3][j];
        const uint32_t nj = c0 + c1 + c2 + c3;
        if (j < half * 4) st += nj;
        if (j >= half * 4 && j < half * 4 + 4) {
            n4[j - half * 4] = nj;
            wb4[j - half * 4] = (wave > 0 ? c0 : 0) + (wave > 1 ? c1 : 0) + (wave > 2 ? c2 : 0);
        }
    }
    uint32_t start4[4], loff4[4], total4 = 0;
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) {
        start4[q] = st + total4;
        loff4[q] = total4;
        total4 += n4[q];
    }
    if (threadIdx.x < 4) {
        const uint32_t q = threadIdx.x;
        const bool valid = tx0 + q < SG.ntx;
        const uint32_t nq = q == 0 ? n4[0] : q == 1 ? n4[1] : q == 2 ? n4[2] : n4[3];
        const uint32_t sq = q == 0 ? start4[0] : q == 1 ? start4[1] : q == 2 ? start4[2] : start4[3];
        const uint32_t tile = row * SG.ntx + tx0 + q;
        // every tile is written (empty ones as (0, 0)): the frame needs no memset of the ranges
        if (valid) reinterpret_cast<int2 *>(ranges)[tile] = nq ? make_int2((int)sq, (int)(sq + nq)) : make_int2(0, 0);
        s_tile[q] = valid ? tile : 0;
        s_start[q] = sq;
        s_n[q] = valid ? nq : 0;
        // statistic for the caller (gs_frame_longest_list_async): rare, so the atomic costs nothing
        if (valid && nq > (uint32_t)GS_LONGEST_MIN) atomicMax(longest, (unsigned long long)nq);
        // (Round 6 also summed the pairs beyond the first GS_LONG_MIN of every list here, for a first version of the caller's
        // cost model: one 64-bit atomic per tile beyond 512 pairs onto ONE address -- 6,000 of them on the 2.4 M scene, and
        // this kernel went from 69.5 to 84 us.  The model reads the compositing kernel's walk statistics instead; counter
        // GS_CNT_EXCESS stays 0.)
    }
    if (total4 == 0) return;  // uniform
#ifdef GS_DIAG_STRIP_NO_PLACE  // timing experiments only (tools/ab_variants.py): wrong results
    return;
#endif
    // ---- 2. place (depth_bits << 32 | gaussian) and 3. sort.  A half strip of up to CAP pairs holds its four lists in
    // LDS side by side (short lists are then sorted by one wave each, concurrently); otherwise the lists take turns
    // in the LDS window, each placed from the entries in registers and sorted by the workgroup; a single list beyond
    // CAP is placed in `scratch` and sorted by the per-tile code of the table variant.  (CAP = 4096 for the first case
    // alone measured slower: 32 KiB of LDS leave four workgroups per CU to hide the barriers of the merge levels.)
    const unsigned long long lt = (1ull << lane) - 1ull;
    // tile q of the set `qmask` goes to dst[q] (its wave's share of it)
    auto place = [&](uint32_t qmask, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3, bool lds, bool reload) {
        uint32_t dst[4] = {d0 + wb4[0], d1 + wb4[1], d2 + wb4[2], d3 + wb4[3]};
        for (uint32_t c = 0; c < nchunk; ++c) {
            if (nchunk > 1 || reload) load_chunk(c);  // else: the single chunk is still in registers
#pragma unroll
            for (uint32_t k = 0; k < EPT; ++k) {
                if (c * CHUNK + k * 256 >= E) break;  // uniform
                const bool ok = c * CHUNK + k * 256 + threadIdx.x < E;
                const uint64_t e = er[k];
                const uint32_t lo32 = (uint32_t)e;
                const float2 xy = centre(ok, lo32);
                const uint64_t key = (e & 0xffffffff00000000ull) | (lo32 & ID_MASK);
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) {
                    if (!(qmask >> q & 1u)) continue;  // uniform
                    const bool cv = covers(ok, lo32, xy, half * 4 + q);
                    const unsigned long long b = __ballot(cv);
                    if (cv) {
                        const uint32_t pos = dst[q] + (uint32_t)__popcll(b & lt);
                        if (lds)
                            s_a[pos] = key;
                        else
                            scratch[pos] = key;
                    }
                    dst[q] += (uint32_t)__popcll(b);
                }
            }
        }
    };
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto block_sync = [] { __syncthreads(); };
    // a[0 .. n) is complete and visible to the workgroup: sorted and stored as list `start`, tile `tile`
    auto sort_store_by_workgroup = [&](uint64_t *a, uint32_t n, uint32_t start, uint64_t tile) {
        auto st = [&](uint32_t i, uint64_t v) {
            ids[start + i] = (uint32_t)v;
            if (keys) keys[start + i] = (tile << 32) | (v >> 32);
        };
        if (n <= 128) {  // registers of one wave
            if (wave == 0) {
                const uint32_t f0 = lane, f1 = lane + 64;
                uint64_t a0 = f0 < n ? a[f0] : KEY_INF, a1 = f1 < n ? a[f1] : KEY_INF;
                sort_window(a0, a1, lane, n);
                if (f0 < n) st(f0, a0);
                if (f1 < n) st(f1, a1);
            }
            return;
        }
        if (n > 2048) {  // beyond the distribution sort's eight keys per thread: the bitonic network, in place
            const uint32_t nwin = (n + 127) / 128;
            for (uint32_t w = wave; w < nwin; w += 4) {
                const uint32_t f0 = w * 128 + lane, f1 = f0 + 64;
                uint64_t a0 = f0 < n ? a[f0] : KEY_INF, a1 = f1 < n ? a[f1] : KEY_INF;
                sort_window(a0, a1, lane, n - w * 128 < 128 ? n - w * 128 : 128);
                if (f0 < n) a[f0] = a0;
                if (f1 < n) a[f1] = a1;
            }
            __syncthreads();
            uint32_t P = 256;
            while (P < n) P <<= 1;
            merge_levels(a, n, P, threadIdx.x, 256u, block_sync);
            for (uint32_t i = threadIdx.x; i < n; i += 256) st(i, a[i]);
            return;
        }
        bucket_sort_store(a, n, s_bcnt, s_wl[0], s_red, threadIdx.x, 256u, block_sync, st);
    };
    const uint32_t tile0 = row * SG.ntx + tx0;
    if (total4 <= (uint32_t)CAP) {  // uniform
        place(15u, loff4[0], loff4[1], loff4[2], loff4[3], true, false);
        __syncthreads();
        {  // short lists: one wave each, no workgroup barrier
            const uint32_t q = wave;
            const uint32_t n = q == 0 ? n4[0] : q == 1 ? n4[1] : q == 2 ? n4[2] : n4[3];
            const uint32_t start = q == 0 ? start4[0] : q == 1 ? start4[1] : q == 2 ? start4[2] : start4[3];
            const uint64_t tile = tile0 + q;
            uint64_t *a = s_a + (q == 0 ? loff4[0] : q == 1 ? loff4[1] : q == 2 ? loff4[2] : loff4[3]);
            auto st = [&](uint32_t i, uint64_t v) {
                ids[start + i] = (uint32_t)v;
                if (keys) keys[start + i] = (tile << 32) | (v >> 32);
            };
            if (n >= 1 && n <= 128) {  // registers only
                const uint32_t f0 = lane, f1 = lane + 64;
                uint64_t a0 = f0 < n ? a[f0] : KEY_INF, a1 = f1 < n ? a[f1] : KEY_INF;
                sort_window(a0, a1, lane, n);
                if (f0 < n) st(f0, a0);
                if (f1 < n) st(f1, a1);
            } else if (n <= WAVE_MAX) {
                bucket_sort_store(a, n, s_bcnt + wave * 513, s_wl[wave], s_red, (uint32_t)lane, 64u, wave_sync, st);
            }
        }
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {  // long lists: the workgroup, one after the other (n is uniform)
            const uint32_t n = n4[q];
            if (n <= WAVE_MAX) continue;
            __syncthreads();
            sort_store_by_workgroup(s_a + loff4[q], n, start4[q], tile0 + q);
        }
        return;
    }
    bool any_big = false;
    for (uint32_t q = 0; q < 4; ++q) {  // the lists take turns in the LDS window, two at a time where they fit (uniform)
        const uint32_t n = n4[q];
        if (n == 0) continue;
        if (n > (uint32_t)CAP) {
            place(1u << q, start4[0], start4[1], start4[2], start4[3], false, true);
            any_big = true;
            continue;
        }
        const uint32_t n2 = q + 1 < 4 ? n4[q + 1] : 0;
        const bool pair = n2 > 0 && n + n2 <= (uint32_t)CAP;  // one pass over the entries serves both lists
        __syncthreads();  // the previous list has been stored
        // entries re-read (L2): 32 registers less across the sorts
        // list q at the start of the window, list q + 1 (if paired) behind it
        place(pair ? 3u << q : 1u << q, 0, q == 0 ? n : 0, q == 1 ? n : 0, q == 2 ? n : 0, true, true);
        __syncthreads();
#ifdef GS_DIAG_STRIP_NO_SORT  // timing experiments only (tools/ab_variants.py): wrong results
        for (uint32_t i = threadIdx.x; i < n + (pair ? n2 : 0); i += 256) ids[start4[q] + i] = (uint32_t)s_a[i];
        if (pair) ++q;
        continue;
#endif
        if (pair && n + n2 > 256) {
            // both lists by ONE distribution sort (bucket_sort_store, nsplit): adjacent tiles, adjacent output ranges
            const uint32_t start = start4[q];
            const uint64_t tile = tile0 + q;
            bucket_sort_store(s_a, n + n2, s_bcnt, s_wl[0], s_red, threadIdx.x, 256u, block_sync,
                              [&](uint32_t i, uint64_t v) {
                                  ids[start + i] = (uint32_t)v;
                                  if (keys) keys[start + i] = ((tile + (i >= n ? 1u : 0u)) << 32) | (v >> 32);
                              },
                              n);
            ++q;
            continue;
        }
        sort_store_by_workgroup(s_a, n, start4[q], tile0 + q);
        if (pair) {
            __syncthreads();
            sort_store_by_workgroup(s_a + n, n2, start4[q + 1], tile0 + q + 1);
            ++q;
        }
    }
    if (any_big && big_queue) {  // dense frame: big_list_sort_kernel sorts the queued lists, one workgroup per list
        if (threadIdx.x < 4 && s_n[threadIdx.x] > (uint32_t)CAP)
            big_queue[atomicAdd(big_count, 1ull)] = s_tile[threadIdx.x];
        return;
    }
    if (any_big) {  // lists in global memory: the per-tile sort of the table variant (in place in `scratch`)
        __syncthreads();
        tile_sort_body<CAP, SRC_PACKED>(
            keys, ids, scratch,
            [&](uint32_t q, uint32_t &tile, uint32_t &start, uint32_t &n) {
                tile = s_tile[q];
                start = s_start[q];
                n = s_n[q] > (uint32_t)CAP ? s_n[q] : 0;
            },
            GatherSrc{}, s_a, s_scan, nullptr, s_bcnt, &s_wl[0][0], s_red);
    }
}

// Sorts and stores the m <= CAP keys of tile `tile` that wait as two 4-byte halves (depth bits in tmp_depth, Gaussian in
// ids) at slots [base, base + m) -- their final range.  256 threads; s_a: CAP keys, s_bcnt / s_wl / s_red: see
// bucket_sort_store.
template <int CAP>
__device__ __forceinline__ void sort_group_halves(uint32_t base, uint32_t m, uint32_t tile, uint64_t *__restrict__ keys,
                                                  uint32_t *__restrict__ ids, const uint32_t *__restrict__ tmp_depth,
                                                  uint64_t *s_a, uint32_t *s_bcnt, uint32_t *s_wl, uint32_t *s_red) {
    static_assert(CAP == 2048, "groups are loaded eight keys per thread");
    const int lane = threadIdx.x & 63;
    auto st = [&](uint32_t i, uint64_t v) {
        ids[base + i] = (uint32_t)v;
        if (keys) keys[base + i] = ((uint64_t)tile << 32) | (v >> 32);
    };
    {
        uint32_t hd[8], hi[8];  // CAP = 8 x 256: all sixteen loads in flight
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
            const uint32_t i = j * 256 + threadIdx.x;
            hd[j] = i < m ? tmp_depth[base + i] : 0;
            hi[j] = i < m ? ids[base + i] : 0;
        }
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
            const uint32_t i = j * 256 + threadIdx.x;
            if (i < m) s_a[i] = ((uint64_t)hd[j] << 32) | hi[j];
        }
    }
    __syncthreads();
    if (m <= 128) {
        if (threadIdx.x < 64) {
            const uint32_t f0 = lane, f1 = lane + 64;
            uint64_t a0 = f0 < m ? s_a[f0] : KEY_INF, a1 = f1 < m ? s_a[f1] : KEY_INF;
            sort_window(a0, a1, lane, m);
            if (f0 < m) st(f0, a0);
            if (f1 < m) st(f1, a1);
        }
    } else {
        bucket_sort_store(s_a, m, s_bcnt, s_wl, s_red, threadIdx.x, 256u, [] { __syncthreads(); }, st);
    }
    __syncthreads();
}

// The groups big_list_sort_kernel cut, sorted by the whole device instead of by the list's own workgroup.
template <int CAP>
__global__ void __launch_bounds__(256) group_sort_kernel(const uint4 *__restrict__ queue,
                                                        const unsigned long long *__restrict__ counters,
                                                        uint32_t queue_cap, uint64_t *__restrict__ keys,
                                                        uint32_t *__restrict__ ids,
                                                        const uint32_t *__restrict__ tmp_depth,
                                                        const unsigned long long *__restrict__ gate) {
    __shared__ uint64_t s_a[CAP];
    __shared__ uint32_t s_bcnt[256 * 8 + 4], s_wl[1 + 2 * 128], s_red[16];
    if (gate && *gate == 0) return;  // second pass of a GS_FRAME_OCCLUSION_CULL frame: only if a tile ran past its cut
    const unsigned long long total = counters[GS_CNT_GROUPS];
    const uint32_t ng = total < queue_cap ? (uint32_t)total : queue_cap;
    for (uint32_t w = blockIdx.x; w < ng; w += gridDim.x) {
        const uint4 g = queue[w];
        sort_group_halves<CAP>(g.y, g.z, g.x, keys, ids, tmp_depth, s_a, s_bcnt, s_wl, s_red);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Lists beyond strip_sort_kernel's LDS window (dense scenes: 10 M Gaussians at 1080p are ~3,500 pairs per tile), one
// workgroup per queued tile.  The list sits unsorted in `scratch` at its final range.  A range of keys is refined by
// one counting pass over 1024 bins that are linear in the 64-bit key (depth_bits << 32 | gaussian): the keys are
// scattered into bin order -- as two 4-byte halves in the range's own slots of the two id arrays (the sorted ids' final
// home and the idle second id buffer: no extra memory) --, runs of consecutive small bins form groups of up to CAP
// keys that are contiguous in the final order, and each group is loaded into LDS, sorted by the distribution sort and
// stored.  A bin with more than CAP / 2 keys (a depth cluster: a pile of clones at one depth behind a background that
// spreads the range) goes back to `scratch` and onto a stack of ranges: its own, >= 1024 times narrower key range is
// binned again (the range includes the id bits, so even equal depths separate).  Only launched in frames with
// GS_FRAME_LONG_LISTS (gs_stage_strip_sort); otherwise strip_sort_kernel sorts a long list itself.
template <int CAP>
__global__ void __launch_bounds__(256) big_list_sort_kernel(const uint32_t *__restrict__ queue,
                                                           const unsigned long long *__restrict__ counters,
                                                           const int32_t *__restrict__ ranges,
                                                           uint64_t *__restrict__ keys, uint32_t *__restrict__ ids,
                                                           uint64_t *__restrict__ scratch,
                                                           uint32_t *__restrict__ tmp_depth,
                                                           uint4 *__restrict__ group_queue, uint32_t queue_cap,
                                                           unsigned long long *__restrict__ group_count,
                                                           const unsigned long long *__restrict__ gate) {
    constexpr uint32_t NBIN = 1024, STACK = 192;
    if (gate && *gate == 0) return;  // (see group_sort_kernel)
    static_assert(CAP == 2048, "groups are loaded eight keys per thread");
    __shared__ uint64_t s_a[CAP];
    __shared__ uint32_t s_scan[4];
    __shared__ uint32_t s_bcnt[256 * 8 + 4], s_wl[4 * 257], s_red[16];
    __shared__ uint32_t s_bin[NBIN + 1];     // first key of every bin (relative to the range)
    __shared__ uint32_t s_gstart[NBIN + 2];  // first bin of every group
    __shared__ uint64_t s_mm[8];
    __shared__ uint2 s_stack[STACK];         // ranges (offset inside the list, count) still to be refined
    __shared__ uint32_t s_sp, s_ng, s_queued;
    uint32_t *s_cursor = reinterpret_cast<uint32_t *>(s_a);
    const int lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t nbig = (uint32_t)counters[GS_CNT_BIG];
    for (uint32_t w = blockIdx.x; w < nbig; w += gridDim.x) {
        const uint32_t tile = queue[w];
        const uint32_t start = (uint32_t)ranges[2 * tile], n_list = (uint32_t)ranges[2 * tile + 1] - start;
        // a group of m <= CAP keys waits as halves at [base, base + m): queued for group_sort_kernel (the device sorts
        // the groups of all lists in parallel), or sorted here when the queue is full
        // (only for lists of more than 16 k keys: the queue's counter is ONE address, 12 ns per atomic -- 32,000 groups of
        // a 10 M-Gaussian frame would spend 0.4 ms there, while a list's few groups are sorted in ~10 us each right here)
        const bool use_queue = group_queue != nullptr && n_list > 16384;
        auto sort_group = [&](uint32_t base, uint32_t m) {
            bool queued = false;
            if (use_queue) {  // uniform
                if (threadIdx.x == 0) {
                    const unsigned long long slot = atomicAdd(group_count, 1ull);
                    if (slot < queue_cap) group_queue[slot] = make_uint4(tile, base, m, 0);
                    s_queued = slot < queue_cap;
                }
                __syncthreads();
                queued = s_queued != 0;
                __syncthreads();
            }
            if (!queued) sort_group_halves<CAP>(base, m, tile, keys, ids, tmp_depth, s_a, s_bcnt, s_wl, s_red);
        };
        __syncthreads();  // LDS of the previous list
        if (threadIdx.x == 0) {
            s_stack[0] = make_uint2(0, n_list);
            s_sp = 1;
        }
        __syncthreads();
        while (s_sp > 0) {  // uniform
            const uint2 top = s_stack[s_sp - 1];
            __syncthreads();
            if (threadIdx.x == 0) s_sp = s_sp - 1;
            const uint32_t off = top.x, n = top.y, r0 = start + off;  // r0: the range's first slot
            uint64_t *a = scratch + r0;
            // every pass over the range keeps eight loads per thread in flight
            auto for_each_key = [&](auto fn) {
                for (uint32_t i0 = 0; i0 < n; i0 += 2048) {
                    uint64_t kk[8];
#pragma unroll
                    for (uint32_t j = 0; j < 8; ++j) {
                        const uint32_t i = i0 + j * 256 + threadIdx.x;
                        kk[j] = i < n ? a[i] : KEY_INF;
                    }
#pragma unroll
                    for (uint32_t j = 0; j < 8; ++j)
                        if (i0 + j * 256 + threadIdx.x < n) fn(kk[j]);
                }
            };
            // ---- range of the keys
            uint64_t kmin = KEY_INF, kmax = 0;
            for_each_key([&](uint64_t key) {
                kmin = key < kmin ? key : kmin;
                kmax = key > kmax ? key : kmax;
            });
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint64_t x = __shfl_xor(kmin, o, 64), y = __shfl_xor(kmax, o, 64);
                kmin = x < kmin ? x : kmin;
                kmax = y > kmax ? y : kmax;
            }
            if (lane == 0) {
                s_mm[wave] = kmin;
                s_mm[4 + wave] = kmax;
            }
            for (uint32_t c = threadIdx.x; c <= NBIN; c += 256) s_bin[c] = 0;
            __syncthreads();
#pragma unroll
            for (uint32_t x = 0; x < 4; ++x) {
                kmin = s_mm[x] < kmin ? s_mm[x] : kmin;
                kmax = s_mm[4 + x] > kmax ? s_mm[4 + x] : kmax;
            }
            const float scale = (float)NBIN / ((float)(kmax - kmin) + 1.0f);
            auto bin_of = [&](uint64_t key) {  // monotone in the key (u64 -> float, x positive constant, truncation)
                const uint32_t b = (uint32_t)((float)(key - kmin) * scale);
                return b < NBIN - 1 ? b : NBIN - 1;
            };
            for_each_key([&](uint64_t key) { atomicAdd(&s_bin[bin_of(key)], 1u); });
            __syncthreads();
            {  // thread t owns bins [4 t, 4 t + 4): exclusive scan of the counts; group boundaries
                const uint32_t HALF = (uint32_t)CAP / 2;
                uint32_t c[4], cs[4], sum = 0, ssum = 0;  // all counts / counts of the small bins only
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    c[j] = s_bin[4 * threadIdx.x + j];
                    cs[j] = c[j] > HALF ? 0 : c[j];
                    sum += c[j];
                    ssum += cs[j];
                }
                const uint32_t incl = gs_wave_incl_scan_u32(sum), sincl = gs_wave_incl_scan_u32(ssum);
                if (lane == 63) {
                    s_red[8 + wave] = incl;
                    s_red[12 + wave] = sincl;
                }
                // the last bin's count of the previous thread decides whether this thread's first bin follows a big bin
                const uint32_t prev_last = __shfl_up(c[3], 1, 64);
                if (lane == 63) s_red[wave] = c[3];
                __syncthreads();
                uint32_t o = incl - sum, so = sincl - ssum;
#pragma unroll
                for (uint32_t x = 0; x < 4; ++x) {
                    o += x < wave ? s_red[8 + x] : 0;
                    so += x < wave ? s_red[12 + x] : 0;
                }
                uint32_t before = lane > 0 ? prev_last : (wave > 0 ? s_red[wave - 1] : 0);  // count of bin 4 t - 1
                uint32_t flags = 0;  // bit j: bin 4 t + j starts a group
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    const uint32_t b = 4 * threadIdx.x + j;
                    // a new group starts at bin 0, at a big bin, behind a big bin, and where the running total of the
                    // small bins (in front of the bin) enters the next CAP / 2: a run of small bins whose totals in
                    // front lie in one window of CAP / 2 holds less than CAP / 2 + CAP / 2 keys
                    const uint32_t prev_so = so - (before > HALF ? 0 : before);
                    const bool boundary = b == 0 || c[j] > HALF || before > HALF || so / HALF != prev_so / HALF;
                    flags |= boundary ? 1u << j : 0u;
                    s_bin[b] = s_cursor[b] = o;
                    o += c[j];
                    so += cs[j];
                    before = c[j];
                }
                if (threadIdx.x == 0) s_bin[NBIN] = n;
                // compact the group starts
                const uint32_t nf = (uint32_t)__popc(flags);
                const uint32_t fincl = gs_wave_incl_scan_u32(nf);
                __syncthreads();
                if (lane == 63) s_red[8 + wave] = fincl;
                __syncthreads();
                uint32_t fo = fincl - nf;
#pragma unroll
                for (uint32_t x = 0; x < 4; ++x) fo += x < wave ? s_red[8 + x] : 0;
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j)
                    if (flags >> j & 1) s_gstart[fo++] = 4 * threadIdx.x + j;
                if (threadIdx.x == 255) {
                    s_ng = fo;
                    s_gstart[fo] = NBIN;
                }
            }
            __syncthreads();
            for_each_key([&](uint64_t key) {  // keys into bin order, as halves
                const uint32_t pos = atomicAdd(&s_cursor[bin_of(key)], 1u);
                tmp_depth[r0 + pos] = (uint32_t)(key >> 32);
                ids[r0 + pos] = (uint32_t)key;
            });
            __syncthreads();
            const uint32_t ng = s_ng;
            for (uint32_t g = 0; g < ng; ++g) {
                const uint32_t gs0 = s_bin[s_gstart[g]], m = s_bin[s_gstart[g + 1]] - gs0;  // uniform
                if (m == 0) continue;
                if (m <= (uint32_t)CAP) {
                    sort_group(r0 + gs0, m);
                    continue;
                }
                // a big bin: back to `scratch`, to be binned again over its own key range -- unless it cannot get
                // narrower (the whole range fell into it) or the stack is full: then the chunked bitonic sort, in place
                for (uint32_t i = threadIdx.x; i < m; i += 256)
                    a[gs0 + i] = ((uint64_t)tmp_depth[r0 + gs0 + i] << 32) | ids[r0 + gs0 + i];
                __syncthreads();
                const uint32_t sp = s_sp;  // every wave reads the depth BEFORE thread 0 may change it (uniform branch)
                __syncthreads();
                if (m < n && sp < STACK) {
                    if (threadIdx.x == 0) {
                        s_stack[sp] = make_uint2(off + gs0, m);
                        s_sp = sp + 1;
                    }
                    __syncthreads();
                } else {
                    tile_sort_body<CAP, SRC_PACKED>(
                        keys, ids, scratch,
                        [&](uint32_t qq, uint32_t &t_, uint32_t &s_, uint32_t &n_) {
                            t_ = tile;
                            s_ = r0 + gs0;
                            n_ = qq == 0 ? m : 0;
                        },
                        GatherSrc{}, s_a, s_scan, nullptr, s_bcnt, s_wl, s_red);
                    __syncthreads();
                }
            }
            __syncthreads();
        }
    }
}

}  // namespace

// CAP = 2048 keys (16 KiB of LDS per workgroup) cover Garden-scale tiles; longer buckets take the in-place
// global path of the same kernel.
int gs_stage_tile_sort(const gs_frame *f, const gs_frame_ws &ws, uint64_t *keys, uint32_t *ids, uint64_t *scratch,
                       hipStream_t stream) {
    gs_frame_geom G = gs_frame_geometry(f);
    hipLaunchKernelGGL((tile_sort_kernel<2048, SRC_KEYS>), dim3((G.n_tiles + 3) / 4), dim3(256), 0, stream, keys, ids,
                       scratch, ws.tile_ranges, (uint32_t)G.n_tiles, GatherSrc{});
    GS_CHECK_LAUNCH();
    return 0;
}

int gs_stage_tile_sort_packed(const gs_frame *f, const gs_frame_ws &ws, uint64_t *packed, uint64_t *keys_out,
                              uint32_t *ids_out, hipStream_t stream) {
    gs_frame_geom G = gs_frame_geometry(f);
    hipLaunchKernelGGL((tile_sort_kernel<2048, SRC_PACKED>), dim3((G.n_tiles + 3) / 4), dim3(256), 0, stream,
                       keys_out, ids_out, packed, ws.tile_ranges, (uint32_t)G.n_tiles, GatherSrc{});
    GS_CHECK_LAUNCH();
    return 0;
}

// slice-sorted variant of sort_mode 2: `slice_pairs_buf` holds the S tile-ordered slice regions, `big_scratch` is
// where buckets beyond the LDS window are gathered and sorted (indexed like the final list)
int gs_stage_tile_sort_gather(const gs_frame *f, const gs_frame_ws &ws, const uint64_t *slice_pairs_buf,
                              uint64_t *big_scratch, uint64_t *keys_out, uint32_t *ids_out, hipStream_t stream) {
    gs_frame_geom G = gs_frame_geometry(f);
    const gs_bin_plan plan = gs_bin_plan_for(f->N, f->max_pairs, G.n_tiles, (f->flags & GS_FRAME_SLICE_SORT) != 0);
    GatherSrc gsrc = {slice_pairs_buf, ws.bin_table, ws.slice_pairs, plan.slices, (uint32_t)G.n_tiles};
    hipLaunchKernelGGL((tile_sort_kernel<2048, SRC_GATHER>), dim3((G.n_tiles + 3) / 4), dim3(256), 0, stream,
                       keys_out, ids_out, big_scratch, ws.tile_ranges, (uint32_t)G.n_tiles, gsrc);
    GS_CHECK_LAUNCH();
    return 0;
}

#ifndef STRIP_SORT_CAP_
#define STRIP_SORT_CAP_ GS_STRIP_SORT_CAP  // A/B switch (tools/ab_variants.py)
#endif
// STRIP variant: entries (level 1, strip_bin.hip) -> tile ranges + sorted ids (+ sorted keys on request)
int gs_stage_strip_sort(const gs_frame *f, const gs_frame_ws &ws, const uint64_t *entries, uint64_t *scratch,
                        uint64_t *keys_out, uint32_t *ids_out, hipStream_t stream, bool second_pass) {
    const unsigned long long *gate = second_pass ? ws.counters + GS_CNT_RANPAST : nullptr;
    gs_frame_geom G = gs_frame_geometry(f);
    const gs_strip_plan plan = gs_strip_plan_for(f->N, G.ntx, G.nty);
    GsDistCull D = {(float)(G.padW / 2), (float)(G.padH / 2), f->focal_x, f->focal_y, f->thresh};
    const unsigned grid = GS_STRIP_W == 8 ? (unsigned)gs_div_up(plan.geom.NS, 8) * 16 : plan.geom.NS;
    // GS_FRAME_LONG_SORT / GS_FRAME_LONG_LISTS (the caller has seen a long list in an earlier frame): lists beyond the window are queued
    // for big_list_sort_kernel, one workgroup each; otherwise strip_sort_kernel sorts a long list itself and the
    // frame saves the launches
    const bool dense = gs_frame_long_sort(f) && ws.big_tiles != nullptr;
    uint32_t *queue = dense ? ws.big_tiles : nullptr;
    if (f->tile_culling_method == 0)
        hipLaunchKernelGGL((strip_sort_kernel<STRIP_SORT_CAP_, true>), dim3(grid), dim3(256), 0, stream, entries,
                           ws.strip_base, ws.strip_tot, ws.counters, keys_out, ids_out, scratch, ws.tile_ranges,
                           plan.geom, ws.rec_geom, D, queue, ws.counters + GS_CNT_BIG, ws.counters + GS_CNT_MAXLIST, gate);
    else
        hipLaunchKernelGGL((strip_sort_kernel<STRIP_SORT_CAP_, false>), dim3(grid), dim3(256), 0, stream, entries,
                           ws.strip_base, ws.strip_tot, ws.counters, keys_out, ids_out, scratch, ws.tile_ranges,
                           plan.geom, ws.rec_geom, D, queue, ws.counters + GS_CNT_BIG, ws.counters + GS_CNT_MAXLIST, gate);
    GS_CHECK_LAUNCH();
    if (dense) {
        const uint32_t qcap = (uint32_t)gs_group_queue_cap(f->max_pairs, G.n_tiles);
        hipLaunchKernelGGL((big_list_sort_kernel<STRIP_SORT_CAP_>), dim3((unsigned)G.n_tiles), dim3(256), 0, stream,
                           ws.big_tiles, ws.counters, ws.tile_ranges, keys_out, ids_out, scratch, ws.vals_b,
                           ws.group_queue, qcap, ws.counters + GS_CNT_GROUPS, gate);
        GS_CHECK_LAUNCH();
        hipLaunchKernelGGL((group_sort_kernel<STRIP_SORT_CAP_>), dim3(qcap < 8192u ? qcap : 8192u), dim3(256), 0, stream,
                           ws.group_queue, ws.counters, qcap, keys_out, ids_out, ws.vals_b, gate);
    }
    GS_CHECK_LAUNCH();
    return 0;
}
