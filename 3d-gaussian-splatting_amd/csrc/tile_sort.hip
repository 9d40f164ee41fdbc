// tile_sort.hip -- sort_mode 1: finish the (tile, depth) order inside each tile's bucket.
//
// The 64-bit key is (tile_id | depth_bits), ties broken by the Gaussian index.  sort_mode 0 runs
// six stable LSD radix passes over all M pairs (radix_sort.hip: 18 dependent launches, 192 B of
// traffic per pair).  sort_mode 1 runs the stable LSD passes on the TILE bits only (2 passes at
// 1080p) -- which groups the pairs by tile, i.e. performs the most-significant-digit split of
// the key -- and then orders every bucket by the unique composite (depth_bits << 32 | gaussian_id)
// with an all-ascending bitonic network:
//   * buckets <= 128 pairs : one wave, no workgroup barrier at all;
//   * buckets <= 2048 pairs: in LDS; comparator strides <= 64 stay inside one wave's 128-element
//     window, so those stages only need wave-level ordering -- a workgroup barrier is paid only
//     for the few stages with stride >= 128;
//   * larger buckets        : same network in place on a global scratch segment (rare).
// The composite key is unique, so the result is exactly the oracle's (tile, depth_bits,
// gaussian_index) order, bit-identical to sort_mode 0, at 64 + 36 B of traffic per pair.
#include "gs_common.h"
#include "gs_frame_layout.h"

namespace {

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

// Merge steps that stay inside 128-element windows (stride j <= 32, or a flip of span k <= 128):
// the wave that owns window `win` does them back to back with wave-level ordering only.
template <typename Mem>
__device__ __forceinline__ void local_disperse(Mem a, uint32_t n, uint32_t win, int lane, uint32_t j0) {
    const uint32_t W = win * 128;
    for (uint32_t j = j0; j >= 1; j >>= 1) {
        const uint32_t lo = W + (lane / j) * 2 * j + (lane % j);
        cmpx(a, lo, lo + j, n);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// Plain version (workgroup barrier after every stage) for buckets that live in global memory.
__device__ __forceinline__ void bitonic_sort_global(uint64_t *a, uint32_t n, uint32_t P) {
    for (uint32_t k = 2; k <= P; k <<= 1) {
        const uint32_t hk = k >> 1;
        for (uint32_t t = threadIdx.x; t < (P >> 1); t += blockDim.x)
            cmpx(a, (t / hk) * k + (t % hk), (t / hk) * k + (k - 1) - (t % hk), n);
        __syncthreads();
        for (uint32_t j = hk >> 1; j >= 1; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < (P >> 1); t += blockDim.x) {
                const uint32_t lo = (t / j) * 2 * j + (t % j);
                cmpx(a, lo, lo + j, n);
            }
            __syncthreads();
        }
    }
}

template <typename Mem, bool BLOCK>
__device__ __forceinline__ void bitonic_sort(Mem a, uint32_t n, uint32_t P) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nwin = (P + 127) / 128;
    const uint32_t wstep = BLOCK ? 4 : 1;
    // 1. every 128-window sorted independently (k = 2 .. 128)
    for (uint32_t win = wave; win < nwin; win += wstep) {
        const uint32_t W = win * 128;
        for (uint32_t k = 2; k <= 128 && k <= P; k <<= 1) {
            const uint32_t hk = k >> 1;
            const uint32_t base = W + (lane / hk) * k, r = lane % hk;
            cmpx(a, base + r, base + (k - 1) - r, n);  // flip
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (hk >= 2) local_disperse(a, n, win, lane, hk >> 1);
        }
    }
    if (P <= 128) return;
    if (BLOCK) __syncthreads();
    // 2. merges across windows (k = 256 .. P): wide strides with workgroup barriers, then local
    for (uint32_t k = 256; k <= P; k <<= 1) {
        const uint32_t hk = k >> 1;
        for (uint32_t t = threadIdx.x; t < (P >> 1); t += blockDim.x)
            cmpx(a, (t / hk) * k + (t % hk), (t / hk) * k + (k - 1) - (t % hk), n);
        __syncthreads();
        for (uint32_t j = hk >> 1; j >= 128; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < (P >> 1); t += blockDim.x) {
                const uint32_t lo = (t / j) * 2 * j + (t % j);
                cmpx(a, lo, lo + j, n);
            }
            __syncthreads();
        }
        for (uint32_t win = wave; win < nwin; win += 4) local_disperse(a, n, win, lane, 64);  // strides 64..1
        __syncthreads();
    }
}

// All-ascending bitonic network run by ONE wave on n <= 512 keys in its own LDS window: every stage is
// ordered by the wave's program order (the fence only stops the compiler from moving LDS accesses across
// it), so there is no workgroup barrier anywhere.  Comparators whose low index is >= n are skipped as a
// block (their partner is the virtual +inf).
__device__ __forceinline__ void bitonic_sort_wave(uint64_t *a, uint32_t n, uint32_t logP) {
    const uint32_t lane = threadIdx.x & 63;
    auto stage_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // comparators t = 0 .. with lo(t) = (t >> lj) << (lj + 1) | (t & (j - 1)) < n
    auto live = [&](uint32_t lj) {
        const uint32_t j = 1u << lj, r = n & (2 * j - 1);
        return ((n >> (lj + 1)) << lj) + (r < j ? r : j);
    };
    for (uint32_t lk = 1; lk <= logP; ++lk) {
        const uint32_t lhk = lk - 1, hk = 1u << lhk;
        for (uint32_t t = lane, te = live(lhk); t < te; t += 64) {  // flip
            const uint32_t base = (t >> lhk) << lk, r = t & (hk - 1);
            cmpx(a, base + r, base + (2 * hk - 1) - r, n);
        }
        stage_sync();
        for (uint32_t lj = lhk; lj-- > 0;) {  // disperse, strides hk/2 .. 1
            const uint32_t j = 1u << lj;
            for (uint32_t t = lane, te = live(lj); t < te; t += 64) {
                const uint32_t lo = ((t >> lj) << (lj + 1)) | (t & (j - 1));
                cmpx(a, lo, lo + j, n);
            }
            stage_sync();
        }
    }
}

// PACKED = false (sort_mode 1): keys = (tile << 32 | depth_bits) grouped by tile, ids = Gaussian index;
//   both are overwritten in place with the depth-sorted order; scratch = idle half of the key buffer.
// PACKED = true (sort_mode 2): scratch = (depth_bits << 32 | gaussian) grouped by tile in arbitrary
//   order; the sorted (tile << 32 | depth_bits) and ids go to keys / ids; long buckets sort in place.
// A workgroup owns four consecutive tiles.  Buckets up to CAP/4 keys are sorted by one wave each,
// concurrently; longer ones by the whole workgroup in the full CAP-key window (or in global memory).
template <int CAP, bool PACKED>
__global__ void __launch_bounds__(256) tile_sort_kernel(uint64_t *__restrict__ keys, uint32_t *__restrict__ ids,
                                                       uint64_t *__restrict__ scratch,
                                                       const int32_t *__restrict__ ranges, uint32_t n_tiles) {
    __shared__ uint64_t s_a[CAP];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t tile, start, n;
    auto load = [&](uint32_t i) -> uint64_t {
        return PACKED ? scratch[start + i] : (keys[start + i] << 32) | ids[start + i];
    };
    auto store = [&](uint32_t i, uint64_t v) {
        ids[start + i] = (uint32_t)v;
        keys[start + i] = ((uint64_t)tile << 32) | (v >> 32);
    };
    auto select = [&](uint32_t t) {
        tile = t;
        start = n = 0;
        if (t < n_tiles) {
            start = (uint32_t)ranges[2 * t];
            n = (uint32_t)ranges[2 * t + 1] - start;
        }
    };
    // 1. one wave per short bucket
    select(blockIdx.x * 4 + wave);
    if (n >= (PACKED ? 1u : 2u) && n <= (uint32_t)CAP / 4) {
        uint64_t *a = s_a + wave * (CAP / 4);
        for (uint32_t i = lane; i < n; i += 64) a[i] = load(i);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t logP = 0;
        while ((1u << logP) < n) ++logP;
        bitonic_sort_wave(a, n, logP);
        for (uint32_t i = lane; i < n; i += 64) store(i, a[i]);
    }
    // 2. long buckets, one after the other, by the whole workgroup (n is uniform => so are the barriers)
    for (uint32_t q = 0; q < 4; ++q) {
        select(blockIdx.x * 4 + q);
        if (n <= (uint32_t)CAP / 4) continue;
        __syncthreads();
        uint32_t P = 1;
        while (P < n) P <<= 1;
        uint64_t *a = n <= (uint32_t)CAP ? s_a : scratch + start;
        if (!PACKED || n <= (uint32_t)CAP)
            for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) a[i] = load(i);
        __syncthreads();
        if (n <= (uint32_t)CAP)
            bitonic_sort<uint64_t *, true>(s_a, n, P);
        else
            bitonic_sort_global(scratch + start, n, P);
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) store(i, a[i]);
    }
}

}  // namespace

// 2048 pairs (16 KiB of LDS, 8 workgroups per CU) cover Garden-scale tiles; longer buckets take the
// in-place global path of the same kernel.
int gs_stage_tile_sort(const gs_frame *f, const gs_frame_ws &ws, uint64_t *keys, uint32_t *ids, uint64_t *scratch,
                       hipStream_t stream) {
    gs_frame_geom G = gs_frame_geometry(f);
    hipLaunchKernelGGL((tile_sort_kernel<2048, false>), dim3((G.n_tiles + 3) / 4), dim3(256), 0, stream, keys, ids, scratch,
                       ws.tile_ranges, (uint32_t)G.n_tiles);
    GS_CHECK_LAUNCH();
    return 0;
}

int gs_stage_tile_sort_packed(const gs_frame *f, const gs_frame_ws &ws, uint64_t *packed, uint64_t *keys_out,
                              uint32_t *ids_out, hipStream_t stream) {
    gs_frame_geom G = gs_frame_geometry(f);
    hipLaunchKernelGGL((tile_sort_kernel<2048, true>), dim3((G.n_tiles + 3) / 4), dim3(256), 0, stream, keys_out, ids_out,
                       packed, ws.tile_ranges, (uint32_t)G.n_tiles);
    GS_CHECK_LAUNCH();
    return 0;
}
