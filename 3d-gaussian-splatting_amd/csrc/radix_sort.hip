// radix_sort.hip -- stable LSD radix sort of (u64 key, u32 value) pairs for wave64 / gfx950.
//
// Replaces `torch.sort` on the float32 composite key `depth + tile_id * (max_depth + 1)`
// (splatter.py:608-613): keys here are (tile_id << 32 | float_bits(depth)), so tile order and
// depth order are both exact, and stability makes ties fall back to emission (Gaussian-index)
// order -- the oracle's canonical (tile, depth_bits, gaussian_index) order.
//
// 8 bits per pass, three launches per pass:
//   digit_histogram_kernel : per-workgroup 256-bin histogram (LDS atomics) -> hist[digit][wg]
//   row_scan_kernel        : one workgroup per digit scans its row over workgroups
//   scatter_kernel         : wave-level match-any ranking (8 ballots), wave-private digit
//                            counters in LDS, cross-wave prefix, direct scatter
// The element count comes from device memory; grids are sized by capacity and idle
// workgroups exit at once, so the frame never synchronises with the host.
// Pure integer/byte work: HBM-bound (each pass reads 8+12 B and writes 12 B per pair).
#include "gs_common.h"
#include "gs_frame_layout.h"

namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_KPT = 8;
constexpr int RS_TILE = RS_THREADS * RS_KPT;  // 2048 == GS_SORT_TILE
constexpr int RS_BINS = 256;
static_assert(RS_TILE == GS_SORT_TILE, "tile size mismatch");

__device__ __forceinline__ uint32_t load_count(const uint32_t *d_count, int64_t capacity) {
    uint32_t n = *d_count;
    return n < (uint64_t)capacity ? n : (uint32_t)capacity;
}

__global__ void __launch_bounds__(RS_THREADS) digit_histogram_kernel(const uint64_t *__restrict__ keys,
                                                                    const uint32_t *__restrict__ d_count,
                                                                    int64_t capacity, int shift, int nwg_cap,
                                                                    uint32_t *__restrict__ hist) {
    const uint32_t n = load_count(d_count, capacity);
    const uint32_t base = blockIdx.x * RS_TILE;
    if (base >= n) return;
    __shared__ uint32_t s_hist[RS_BINS];
    s_hist[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_KPT; ++i) {
        const uint32_t idx = base + i * RS_THREADS + threadIdx.x;
        if (idx < n) atomicAdd(&s_hist[(uint32_t)(keys[idx] >> shift) & (RS_BINS - 1)], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nwg_cap + blockIdx.x] = s_hist[threadIdx.x];
}

// grid = 256 (one workgroup per digit): in-place exclusive scan of hist[digit][0..nwg) and the
// row total.
__global__ void __launch_bounds__(RS_THREADS) row_scan_kernel(uint32_t *__restrict__ hist,
                                                             uint32_t *__restrict__ digit_total,
                                                             const uint32_t *__restrict__ d_count,
                                                             int64_t capacity, int nwg_cap) {
    const uint32_t n = load_count(d_count, capacity);
    const int nwg = (int)((n + RS_TILE - 1) / RS_TILE);
    uint32_t *row = hist + (size_t)blockIdx.x * nwg_cap;
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int b0 = 0; b0 < nwg; b0 += RS_THREADS) {
        const int i = b0 + threadIdx.x;
        const uint32_t v = i < nwg ? row[i] : 0;
        const uint32_t incl = gs_wave_incl_scan_u32(v);
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) woff += w < wave ? s_wave[w] : 0;
        const uint32_t carry = s_carry;
        if (i < nwg) row[i] = carry + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == RS_THREADS - 1) s_carry = carry + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) digit_total[blockIdx.x] = s_carry;
}

__global__ void __launch_bounds__(RS_THREADS) scatter_kernel(
    const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, uint64_t *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out, const uint32_t *__restrict__ d_count, int64_t capacity, int shift,
    int nwg_cap, const uint32_t *__restrict__ hist, const uint32_t *__restrict__ digit_total) {
    const uint32_t n = load_count(d_count, capacity);
    const uint32_t base = blockIdx.x * RS_TILE;
    if (base >= n) return;
    __shared__ uint32_t s_wave_cnt[4][RS_BINS];  // per-wave digit counters, then per-wave bases
    __shared__ uint32_t s_scan[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d = threadIdx.x;

    // exclusive scan of the 256 digit totals (global digit base), one digit per thread
    const uint32_t tot = digit_total[d];
    const uint32_t incl = gs_wave_incl_scan_u32(tot);
    if (lane == 63) s_scan[wave] = incl;
#pragma unroll
    for (int w = 0; w < 4; ++w) s_wave_cnt[w][d] = 0;
    __syncthreads();
    uint32_t gbase = incl - tot + hist[(size_t)d * nwg_cap + blockIdx.x];
#pragma unroll
    for (int w = 0; w < 4; ++w) gbase += w < wave ? s_scan[w] : 0;

    // rank the wave's 512-key segment, 64 consecutive keys per round (coalesced loads)
    uint64_t key[RS_KPT];
    uint32_t val[RS_KPT], rank[RS_KPT];
    const uint32_t seg = base + wave * (RS_KPT * 64);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < RS_KPT; ++r) {
        const uint32_t idx = seg + r * 64 + lane;
        const bool valid = idx < n;
        key[r] = valid ? keys_in[idx] : ~0ull;
        val[r] = valid ? vals_in[idx] : 0u;
        const uint32_t dig = (uint32_t)(key[r] >> shift) & (RS_BINS - 1);
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (dig >> b) & 1;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t before = __popcll(peers & lt_mask);
        const uint32_t cnt = __popcll(peers);
        uint32_t c = 0;
        if (valid) c = s_wave_cnt[wave][dig];
        __builtin_amdgcn_wave_barrier();
        if (valid && before == 0) s_wave_cnt[wave][dig] = c + cnt;
        __builtin_amdgcn_wave_barrier();
        rank[r] = c + before;
    }
    __syncthreads();
    // per digit: exclusive prefix over the 4 waves, plus the global base
    {
        uint32_t run = gbase;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t c = s_wave_cnt[w][d];
            s_wave_cnt[w][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_KPT; ++r) {
        const uint32_t idx = seg + r * 64 + lane;
        if (idx < n) {
            const uint32_t dig = (uint32_t)(key[r] >> shift) & (RS_BINS - 1);
            const uint32_t pos = s_wave_cnt[wave][dig] + rank[r];
            keys_out[pos] = key[r];
            vals_out[pos] = val[r];
        }
    }
}

}  // namespace

extern "C" size_t gs_sort_pairs_tmp_bytes(int64_t capacity) {
    const int64_t nwg = gs_div_up(capacity > 0 ? capacity : 1, RS_TILE);
    return gs_align_up(sizeof(uint32_t) * RS_BINS * (size_t)nwg, 256) + gs_align_up(sizeof(uint32_t) * RS_BINS, 256);
}

extern "C" int gs_sort_pairs(uint64_t *keys0, uint32_t *vals0, uint64_t *keys1, uint32_t *vals1,
                             const uint32_t *d_count, int64_t capacity, int end_bit, void *tmp, size_t tmp_bytes,
                             int *sorted_in_buffer1, gs_stream_t stream) {
    return gs_sort_pairs_bits(keys0, vals0, keys1, vals1, d_count, capacity, 0, end_bit, tmp, tmp_bytes,
                              sorted_in_buffer1, stream);
}

extern "C" int gs_sort_pairs_bits(uint64_t *keys0, uint32_t *vals0, uint64_t *keys1, uint32_t *vals1,
                                  const uint32_t *d_count, int64_t capacity, int begin_bit, int end_bit, void *tmp,
                                  size_t tmp_bytes, int *sorted_in_buffer1, gs_stream_t stream) {
    GS_CHECK_ARG(capacity >= 0 && capacity < (1ll << 32), "capacity out of range");
    GS_CHECK_ARG(begin_bit >= 0 && begin_bit <= end_bit && end_bit <= 64, "need 0 <= begin_bit <= end_bit <= 64");
    const int npass = (end_bit - begin_bit + 7) / 8;
    if (sorted_in_buffer1) *sorted_in_buffer1 = npass & 1;
    if (capacity == 0 || npass == 0) return 0;
    GS_CHECK_ARG(keys0 && vals0 && keys1 && vals1 && d_count && tmp, "null pointer");
    GS_CHECK_ARG(tmp_bytes >= gs_sort_pairs_tmp_bytes(capacity), "tmp too small");
    const int nwg = (int)gs_div_up(capacity, RS_TILE);
    uint32_t *hist = (uint32_t *)tmp;
    uint32_t *digit_total = (uint32_t *)((char *)tmp + gs_align_up(sizeof(uint32_t) * RS_BINS * (size_t)nwg, 256));
    hipStream_t s = (hipStream_t)stream;
    uint64_t *kin = keys0, *kout = keys1;
    uint32_t *vin = vals0, *vout = vals1;
    for (int p = 0; p < npass; ++p) {
        const int shift = begin_bit + p * 8;
        hipLaunchKernelGGL(digit_histogram_kernel, dim3(nwg), dim3(RS_THREADS), 0, s, kin, d_count, capacity, shift,
                           nwg, hist);
        hipLaunchKernelGGL(row_scan_kernel, dim3(RS_BINS), dim3(RS_THREADS), 0, s, hist, digit_total, d_count,
                           capacity, nwg);
        hipLaunchKernelGGL(scatter_kernel, dim3(nwg), dim3(RS_THREADS), 0, s, kin, vin, kout, vout, d_count,
                           capacity, shift, nwg, hist, digit_total);
        uint64_t *tk = kin;
        kin = kout;
        kout = tk;
        uint32_t *tv = vin;
        vin = vout;
        vout = tv;
    }
    GS_CHECK_LAUNCH();
    return 0;
}
