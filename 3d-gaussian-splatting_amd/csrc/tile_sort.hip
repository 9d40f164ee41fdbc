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

template <int CAP, int SRC>
__global__ void __launch_bounds__(256) tile_sort_kernel(uint64_t *__restrict__ keys, uint32_t *__restrict__ ids,
                                                       uint64_t *__restrict__ scratch,
                                                       const int32_t *__restrict__ ranges, uint32_t n_tiles,
                                                       GatherSrc GS) {
    constexpr bool PACKED = SRC != SRC_KEYS;  // the unsorted element already is (depth_bits << 32 | gaussian)
    __shared__ uint64_t s_a[CAP];
    __shared__ uint32_t s_scan[4];
    // SRC_GATHER: where the pairs of this workgroup's four tiles start in every slice region (absolute index into
    // `pairs`), tiles t0 .. t0 + 4 (the fifth column closes the fourth tile): row j holds S entries.  Loaded once per
    // workgroup -- thread = slice, five consecutive words of its table row: ONE cache line per slice for all four
    // tiles -- instead of two scattered 4-byte reads per slice and tile.
    __shared__ uint32_t s_off[SRC == SRC_GATHER ? 5 * GS_BIN_MAX_SLICES : 1];
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
    auto select = [&](uint32_t t) {
        tile = t;
        start = n = 0;
        if (t < n_tiles) {
            start = (uint32_t)ranges[2 * t];
            n = (uint32_t)ranges[2 * t + 1] - start;
        }
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
        const uint32_t *o_lo = s_off + (tile - blockIdx.x * 4) * GS.S, *o_hi = o_lo + GS.S;
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
    select(blockIdx.x * 4 + wave);
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
        } else {
            const uint32_t nwin = (n + 127) / 128;
            for (uint32_t w = 0; w < nwin; ++w) {
                if (SRC == SRC_GATHER)
                    sort_window_from(from_lds, w, [&](uint32_t e, uint64_t v) { a[e] = v; });
                else
                    sort_window_from(load, w, [&](uint32_t e, uint64_t v) { a[e] = v; });
            }
            wave_sync();
            uint32_t P = 256;
            while (P < n) P <<= 1;
            merge_levels(a, n, P, (uint32_t)lane, 64u, wave_sync);
            for (uint32_t i = lane; i < n; i += 64) store(i, a[i]);
        }
    }
    // 2. long buckets, one after the other, by the whole workgroup (n is uniform => so are the barriers)
    for (uint32_t q = 0; q < 4; ++q) {
        select(blockIdx.x * 4 + q);
        if (n <= (uint32_t)CAP / 4) continue;
        __syncthreads();
        uint32_t P = 256;
        while (P < n) P <<= 1;
        if (n <= (uint32_t)CAP) {
            const uint32_t nwin = (n + 127) / 128;
            if (SRC == SRC_GATHER) {
                gather(s_a, threadIdx.x, 256u, true);
                __syncthreads();
                for (uint32_t w = wave; w < nwin; w += 4)
                    sort_window_from([&](uint32_t e) -> uint64_t { return s_a[e]; }, w,
                                     [&](uint32_t e, uint64_t v) { s_a[e] = v; });
            } else {
                for (uint32_t w = wave; w < nwin; w += 4)
                    sort_window_from(load, w, [&](uint32_t e, uint64_t v) { s_a[e] = v; });
            }
            __syncthreads();
            merge_levels(s_a, n, P, threadIdx.x, 256u, [] { __syncthreads(); });
            for (uint32_t i = threadIdx.x; i < n; i += 256) store(i, s_a[i]);
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
