// binning.hip -- per-tile duplication of the projected Gaussians.
//
// Reference-API kernels: calc_tile_list methods 0/1/2 (gaussian.cu:101-335) and
// gather_gaussians (gaussian.cu:337-381), which build and compact a fixed-capacity
// T x MAXP table with atomics.
//
// Frame path, sort_modes 0 and 1 (the default mode 2 bins in tile_bin.hip instead): the table is gone.
// Stage S1 (cull_project.hip) already counted the tiles each Gaussian touches and left one partial
// sum per 256-Gaussian block; here
//   S3  scan_block_sums_kernel : exclusive scan of the block sums, total M -> device counter
//   S4  emit_pairs_kernel      : block-local scan + emission of (tile<<32 | depth_bits, id)
//   S6  tile_ranges_kernel     : [start,end) of every tile in the sorted key array
// Emission order is deterministic (offsets are a prefix sum in Gaussian-index order), so the
// stable radix sort yields exactly the oracle's (tile, depth_bits, gaussian_index) order.
// Compiled with -ffp-contract=off (tile rectangles must match the oracle bit for bit).
#include "gs_common.h"
#include "gs_frame_layout.h"

namespace {

// ---------------------------------------------------------------- reference-API kernels
struct BinGeom {
    float tlx, tly, leftmost, topmost, tlog, thresh;
    uint32_t ntx, nty;
};

__device__ __forceinline__ bool bbox(const float *__restrict__ pos, const float4 *__restrict__ cov, int64_t pid,
                                     float tlog, float &l, float &r, float &t, float &b) {
    const float4 c = cov[pid];
    const float cx = pos[pid * 3], cy = pos[pid * 3 + 1];
    float det = (c.x * c.w - c.y * c.z);
    if (det <= 0) return false;
    float ai = (float)(c.w / (det + 1e-14));
    float di = (float)(c.x / (det + 1e-14));
    float shift_x = sqrtf(di * tlog * det);
    float shift_y = sqrtf(ai * tlog * det);
    r = cx + shift_x;
    l = cx - shift_x;
    t = cy - shift_y;
    b = cy + shift_y;
    return true;
}

// method 2 ("prob2", gaussian.cu:197-250): rectangle of tiles from the bounding box.
__global__ void __launch_bounds__(256) calc_tile_list_rect_kernel(const float *__restrict__ pos,
                                                                 const float4 *__restrict__ cov, int64_t n,
                                                                 BinGeom G, int32_t *__restrict__ tile_n_point,
                                                                 int32_t *__restrict__ list, int64_t maxp) {
    for (int64_t pid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pid < n;
         pid += (int64_t)gridDim.x * blockDim.x) {
        float l, r, t, b;
        if (!bbox(pos, cov, pid, G.tlog, l, r, t, b)) continue;
        uint32_t y0 = gs_f2u_sat(fmaxf((t - G.topmost) / G.tly, 0));
        uint32_t y1 = gs_f2u_sat((b - G.topmost) / G.tly + 1);
        uint32_t x0 = gs_f2u_sat(fmaxf((l - G.leftmost) / G.tlx, 0));
        uint32_t x1 = gs_f2u_sat((r - G.leftmost) / G.tlx + 1);
        if (y1 > G.nty) y1 = G.nty;
        if (x1 > G.ntx) x1 = G.ntx;
        for (uint32_t iy = y0; iy < y1; ++iy)
            for (uint32_t ix = x0; ix < x1; ++ix) {
                int64_t tid = ix + (int64_t)iy * G.ntx;
                int32_t old = atomicAdd(tile_n_point + tid, 1);
                if (old < maxp) list[maxp * tid + old] = (int32_t)pid;
            }
    }
}

// methods 0 ("dist", :101-136) and 1 ("prob", :138-195): O(V*T) tests against per-tile edges.
template <int METHOD>
__global__ void __launch_bounds__(256) calc_tile_list_brute_kernel(
    const float *__restrict__ pos, const float4 *__restrict__ cov, int64_t n, int64_t n_tiles,
    const float *__restrict__ top, const float *__restrict__ bottom, const float *__restrict__ left,
    const float *__restrict__ right, BinGeom G, int32_t *__restrict__ tile_n_point, int32_t *__restrict__ list,
    int64_t maxp) {
    const int64_t pid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= n) return;
    const float cx = pos[pid * 3], cy = pos[pid * 3 + 1];
    float l = 0, r = 0, t = 0, b = 0;
    if (METHOD == 1 && !bbox(pos, cov, pid, G.tlog, l, r, t, b)) return;
    for (int64_t tid = blockIdx.y; tid < n_tiles; tid += gridDim.y) {
        bool hit;
        if (METHOD == 0) {
            float center_y = (top[tid] + bottom[tid]) / 2;
            float center_x = (left[tid] + right[tid]) / 2;
            float d1 = cx - center_x, d2 = cy - center_y;
            hit = d1 * d1 + d2 * d2 < G.thresh;
        } else {
            hit = !(right[tid] < l || r < left[tid] || bottom[tid] < t || b < top[tid]);
        }
        if (hit) {
            int32_t old = atomicAdd(tile_n_point + tid, 1);
            if (old < maxp) list[maxp * tid + old] = (int32_t)pid;
        }
    }
}

__global__ void __launch_bounds__(256) gather_gaussians_kernel(const int32_t *__restrict__ accum,
                                                              const int32_t *__restrict__ list,
                                                              int32_t *__restrict__ gathered,
                                                              int32_t *__restrict__ tile_ids, int64_t n_tiles,
                                                              int64_t row) {
    for (int64_t tid = blockIdx.x; tid < n_tiles; tid += gridDim.x) {
        const int32_t s = accum[tid], cnt = accum[tid + 1] - s;
        for (int32_t p = threadIdx.x; p < cnt; p += blockDim.x) {
            gathered[s + p] = list[tid * row + p];
            tile_ids[s + p] = (int32_t)tid;
        }
    }
}

// ---------------------------------------------------------------- frame stage S3
// One workgroup scans the per-block pair sums (<= ~40k values for 10M Gaussians).
__global__ void __launch_bounds__(1024) scan_block_sums_kernel(const uint32_t *__restrict__ block_sums,
                                                              const uint32_t *__restrict__ block_vis,
                                                              uint32_t *__restrict__ block_offsets, int nblk,
                                                              unsigned long long *__restrict__ counters,
                                                              unsigned long long max_pairs) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry, s_vis;
    if (threadIdx.x == 0) s_carry = s_vis = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t vis = 0;
    for (int base = 0; base < nblk; base += 1024) {
        const int i = base + threadIdx.x;
        uint32_t v = i < nblk ? block_sums[i] : 0;
        vis += i < nblk ? block_vis[i] : 0;
        uint32_t incl = gs_wave_incl_scan_u32(v);
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t wave_off = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) wave_off += w < wave ? s_wave[w] : 0;
        const uint32_t carry = s_carry;
        if (i < nblk) block_offsets[i] = carry + wave_off + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + wave_off + incl;
        __syncthreads();
    }
    vis = gs_wave_sum_u32(vis);
    if (lane == 0 && vis) atomicAdd(&s_vis, vis);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long total = s_carry;
        counters[GS_CNT_PAIRS] = total < max_pairs ? total : max_pairs;
        counters[GS_CNT_OVERFLOW] = total > max_pairs ? total : 0ull;
        counters[GS_CNT_VISIBLE] = s_vis;
        // (the compositing kernel's walk statistics and the strip variant's counters: never stale for a reader)
        counters[GS_CNT_EXCESS] = counters[GS_CNT_MAXWALK] = counters[GS_CNT_EXCESS_WALK] = 0;
        counters[GS_CNT_RANPAST] = counters[GS_CNT_BIG] = counters[GS_CNT_GROUPS] = counters[GS_CNT_MAXLIST] = 0;
    }
}

// ---------------------------------------------------------------- frame stage S4
// Same 256-Gaussian blocks as S1.  Small rectangles are written by their own lane; large ones
// (> 16 tiles) are written cooperatively by the whole wave so that one screen-filling Gaussian
// does not serialise 64 lanes behind it.
#define GS_EMIT_SOLO 16
__global__ void __launch_bounds__(256) emit_pairs_kernel(const uint32_t *__restrict__ tiles_touched,
                                                        const uint4 *__restrict__ rects,
                                                        const uint32_t *__restrict__ block_offsets, int64_t n,
                                                        uint32_t ntx, uint64_t *__restrict__ keys,
                                                        uint32_t *__restrict__ vals, uint64_t max_pairs,
                                                        uint32_t *__restrict__ pair_offsets) {
    const int64_t pid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t cnt = pid < n ? tiles_touched[pid] : 0;
    __shared__ uint32_t s_wave[4];
    const uint32_t incl = gs_wave_incl_scan_u32(cnt);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t off = block_offsets[blockIdx.x] + incl - cnt;
#pragma unroll
    for (int w = 0; w < 4; ++w) off += w < wave ? s_wave[w] : 0;
    if (pid < n) pair_offsets[pid] = off;  // the backward pass addresses its per-pair rows with it

    uint4 rc = make_uint4(0, 0, 0, 0);
    if (cnt) rc = rects[pid];
    const uint32_t dbits = rc.z;
    const uint32_t y0 = rc.x & 0xffff, x0 = rc.y & 0xffff, x1 = rc.y >> 16;
    const uint32_t wdt = x1 - x0;
    if (cnt && cnt <= GS_EMIT_SOLO) {
        uint32_t ix = x0, iy = y0;
        for (uint32_t k = 0; k < cnt; ++k) {
            const uint64_t o = (uint64_t)off + k;
            if (o < max_pairs) {
                keys[o] = ((uint64_t)(ix + iy * ntx) << 32) | dbits;
                vals[o] = (uint32_t)pid;
            }
            if (++ix == x1) {
                ix = x0;
                ++iy;
            }
        }
    }
    unsigned long long big = __ballot(cnt > GS_EMIT_SOLO);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const uint32_t c = __shfl(cnt, src, 64), o0 = __shfl(off, src, 64), d = __shfl(dbits, src, 64);
        const uint32_t sx0 = __shfl(x0, src, 64), sy0 = __shfl(y0, src, 64), sw = __shfl(wdt, src, 64);
        const uint32_t id = (uint32_t)(pid - lane + src);
        for (uint32_t k = lane; k < c; k += 64) {
            const uint64_t o = (uint64_t)o0 + k;
            if (o < max_pairs) {
                const uint32_t iy = sy0 + k / sw, ix = sx0 + k % sw;
                keys[o] = ((uint64_t)(ix + iy * ntx) << 32) | d;
                vals[o] = id;
            }
        }
    }
}

// ---------------------------------------------------------------- frame stage S6
__global__ void __launch_bounds__(256) tile_ranges_kernel(const uint64_t *__restrict__ keys,
                                                         const unsigned long long *__restrict__ counters,
                                                         int32_t *__restrict__ ranges) {
    const int64_t M = (int64_t)counters[GS_CNT_PAIRS];
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < M; j += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t t = (uint32_t)(keys[j] >> 32);
        if (j == 0 || (uint32_t)(keys[j - 1] >> 32) != t) ranges[2 * t] = (int32_t)j;
        if (j == M - 1 || (uint32_t)(keys[j + 1] >> 32) != t) ranges[2 * t + 1] = (int32_t)(j + 1);
    }
}

}  // namespace

// ================================================================= C ABI (section A)
extern "C" int gs_calc_tile_list(const float *pos, const float *cov, int64_t n_point, const float *tile_top,
                                 const float *tile_bottom, const float *tile_left, const float *tile_right,
                                 int32_t *tile_n_point, int32_t *tile_gaussian_list, int64_t max_points_per_tile,
                                 float thresh, int method, float tile_length_x, float tile_length_y,
                                 int32_t n_tiles_x, int32_t n_tiles_y, float leftmost, float topmost,
                                 gs_stream_t stream) {
    GS_CHECK_ARG(n_point >= 0 && max_points_per_tile >= 0, "negative size");
    GS_CHECK_ARG(method >= 0 && method <= 2, "method must be 0 (dist), 1 (prob) or 2 (prob2)");
    GS_CHECK_ARG(n_tiles_x > 0 && n_tiles_y > 0, "empty tile grid");
    if (n_point == 0) return 0;
    GS_CHECK_ARG(pos && cov && tile_n_point && (tile_gaussian_list || max_points_per_tile == 0), "null pointer");
    GS_CHECK_ARG(((uintptr_t)cov & 15) == 0, "cov must be 16-byte aligned");
    BinGeom G;
    G.tlx = tile_length_x;
    G.tly = tile_length_y;
    G.leftmost = leftmost;
    G.topmost = topmost;
    G.thresh = thresh;
    G.tlog = -2 * logf(thresh);
    G.ntx = (uint32_t)n_tiles_x;
    G.nty = (uint32_t)n_tiles_y;
    const int64_t n_tiles = (int64_t)n_tiles_x * n_tiles_y;
    hipStream_t s = (hipStream_t)stream;
    if (method == 2) {
        int grid = (int)(gs_div_up(n_point, 256) < 8192 ? gs_div_up(n_point, 256) : 8192);
        hipLaunchKernelGGL(calc_tile_list_rect_kernel, dim3(grid), dim3(256), 0, s, pos, (const float4 *)cov,
                           n_point, G, tile_n_point, tile_gaussian_list, max_points_per_tile);
    } else {
        GS_CHECK_ARG(tile_top && tile_bottom && tile_left && tile_right, "methods 0/1 need the Tiles edges");
        dim3 grid((unsigned)gs_div_up(n_point, 256), (unsigned)(n_tiles < 1024 ? n_tiles : 1024));
        if (method == 0)
            hipLaunchKernelGGL(calc_tile_list_brute_kernel<0>, grid, dim3(256), 0, s, pos, (const float4 *)cov,
                               n_point, n_tiles, tile_top, tile_bottom, tile_left, tile_right, G, tile_n_point,
                               tile_gaussian_list, max_points_per_tile);
        else
            hipLaunchKernelGGL(calc_tile_list_brute_kernel<1>, grid, dim3(256), 0, s, pos, (const float4 *)cov,
                               n_point, n_tiles, tile_top, tile_bottom, tile_left, tile_right, G, tile_n_point,
                               tile_gaussian_list, max_points_per_tile);
    }
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_gather_gaussians(const int32_t *tile_n_point_accum, const int32_t *tile_gaussian_list,
                                   int32_t *gathered_list, int32_t *tile_ids_for_points, int64_t n_tiles,
                                   int64_t max_points_for_tile, int64_t list_row_size, gs_stream_t stream) {
    GS_CHECK_ARG(n_tiles >= 0 && list_row_size >= 0, "negative size");
    (void)max_points_for_tile;  // the reference sizes its grid with it; we walk each tile's own count
    if (n_tiles == 0) return 0;
    GS_CHECK_ARG(tile_n_point_accum && tile_gaussian_list && gathered_list && tile_ids_for_points, "null pointer");
    int grid = (int)(n_tiles < 16384 ? n_tiles : 16384);
    hipLaunchKernelGGL(gather_gaussians_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, tile_n_point_accum,
                       tile_gaussian_list, gathered_list, tile_ids_for_points, n_tiles, list_row_size);
    GS_CHECK_LAUNCH();
    return 0;
}

// ================================================================= frame stages (internal)
int gs_stage_scan_emit(const gs_frame *f, const gs_frame_ws &ws, hipStream_t stream) {
    const int nblk = (int)gs_div_up(f->N, 256);
    gs_frame_geom G = gs_frame_geometry(f);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, stream, ws.block_sums, ws.block_vis,
                       ws.block_offsets,
                       nblk, ws.counters, (unsigned long long)f->max_pairs);
    GS_CHECK_LAUNCH();
    hipLaunchKernelGGL(emit_pairs_kernel, dim3(nblk), dim3(256), 0, stream, ws.tiles_touched, ws.rects,
                       ws.block_offsets, f->N, (uint32_t)G.ntx, ws.keys_a, ws.vals_a,
                       (uint64_t)f->max_pairs, ws.pair_offsets);
    GS_CHECK_LAUNCH();
    return 0;
}

int gs_stage_tile_ranges(const gs_frame *f, const gs_frame_ws &ws, const uint64_t *sorted_keys, hipStream_t stream) {
    // ws.tile_ranges was cleared by the frame-start memset (it sits right behind the counters)
    int grid = (int)(gs_div_up(f->max_pairs, 256) < 4096 ? gs_div_up(f->max_pairs, 256) : 4096);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(tile_ranges_kernel, dim3(grid), dim3(256), 0, stream, sorted_keys, ws.counters,
                       ws.tile_ranges);
    GS_CHECK_LAUNCH();
    return 0;
}
