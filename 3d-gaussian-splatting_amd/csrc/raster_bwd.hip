// raster_bwd.hip -- tile rasterizer backward: one wave per 64-Gaussian bucket, two kernels.
//
// Replaces draw_backward_kernel (gaussian.cu:440-803).  The reference keeps one pixel per
// thread and, for EVERY Gaussian, reduces 10 (no SH) or 34 (SH) gradient terms across the
// warp with 5-step shuffle trees plus shared-memory atomics -- the reduction is most of its
// runtime -- and walks a tile's whole list in one workgroup.
//
// Here a tile's sorted list is cut into BUCKETS of 64 consecutive Gaussians.  Buckets are independent because
// the forward pass (or its replay for the reference-API entry point) checkpointed every pixel's (T, C_run) at
// each bucket boundary, so the grid is one wave per (tile, bucket): long tiles are spread over many CUs
// instead of serialising one workgroup.  Two kernels do the work inside a bucket:
//
//   * no SH -- raster_backward_pixel_kernel: lanes own PIXELS (four each), the bucket's Gaussians are applied
//     front to back, the ten per-Gaussian sums are reduced over the wave through LDS (see its header);
//   * SH (and the reference API's `sigmoid` flag) -- raster_backward_kernel, a 64-lane SYSTOLIC pipeline:
//     lane l keeps Gaussian l's parameters AND its gradient accumulators (10 + 27 / 48 SH sums) in registers,
//     the tile's 256 pixels stream through the lanes: at step t lane l handles pixel t - l.  A pixel's
//     running state (transmittance T, rho = dL/dC . (C_final - C_run), dL/dC, pixel centre: 7 registers) moves
//     from lane l to lane l+1 with one DPP `wave_shr:1` per register; lane 0 is fed from LDS.  No cross-lane
//     reduction and no atomics inside the loop: after 256+63 steps every lane holds the complete sums over
//     the tile's pixels for its Gaussian.
//
// Gradient identities used by both (A.8 of SURVEY.md):
//   s = dL/dalpha * alpha,  u = -ln G
//   dL/dx = ln2 (2A' Sx - B' Sy),            Sx = sum s dx, Sy = sum s dy
//   dL/da = (-Syy + 2 d Su)/Pn, dL/db = (Sxy - 2 c Su)/Pn, dL/dc = (Sxy - 2 b Su)/Pn,
//   dL/dd = (-Sxx + 2 a Su)/Pn,              Sxx = sum s dx^2, ..., Su = sum s u
// which is the reference's dP1_d{a,b,c,d} (gaussian.cu:610-621) with the per-Gaussian
// constants factored out of the pixel loop.
#include <stdlib.h>

#include "raster_common.h"

int gs_raster_forward_ref(const RasterSrc &S, const RasterGeom &G, const int32_t *accum, float *res, int use_sh,
                          int sigmoid, int weight_normalize, float4 *ckpt, uint32_t *tile_nproc,
                          hipStream_t stream, int exact);

namespace {

// exclusive scan of ceil(nproc/64) over tiles -> bucket_offsets[T+1], total -> *n_buckets (bucket_scan_kernel, one workgroup),
// then one record per bucket (bucket_fill_kernel, a wave per tile): (tile, first Gaussian of the bucket inside the tile's
// list, Gaussians in the bucket, start of the tile's list) -- a bucket kernel's wave learns everything about its work item
// from ONE load instead of a binary search over the offsets followed by three dependent loads.
// Round 5: with frame ranges the scan also counts the buckets of SATURATED tiles -- tiles whose compositing stopped before the
// end of their list because every pixel had saturated -- and returns the count in the upper half of *n_buckets: the share of
// such buckets is what decides between the two rgb backward kernels (GS_FRAME_BWD_ROWS, include/gs_abi.h), and the caller
// reads the counter back with the frame's other counters anyway.
// Two kernels since the end of round 5: until then the scan's ONE workgroup also stored the records -- 31 k x 16 bytes from one
// CU at 2.4 M Gaussians: 28 us alone, and 91 us where it actually runs, on the side stream underneath the caller's loss kernel
// (kernel trace of whole training steps, profiles/r05_z_*): longer than the loss it was meant to hide under; 62 k and more
// records in a densified scene.
__global__ void __launch_bounds__(1024) bucket_scan_kernel(const uint32_t *__restrict__ tile_nproc, int n_tiles,
                                                          uint32_t *__restrict__ bucket_offsets,
                                                          unsigned long long *__restrict__ n_buckets,
                                                          const int32_t *__restrict__ ranges, int frame_ranges) {
    __shared__ uint32_t s_wave[2][16];
    __shared__ uint32_t s_carry[2];
    if (threadIdx.x < 2) s_carry[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < n_tiles; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t np = i < n_tiles ? tile_nproc[i] : 0;
        const uint32_t v = (np + GS_BUCKET - 1) / GS_BUCKET;
        const bool sat = frame_ranges && i < n_tiles && np < (uint32_t)(ranges[2 * i + 1] - ranges[2 * i]);
        const uint32_t incl = gs_wave_incl_scan_u32(v), sat_sum = gs_wave_sum_u32(sat ? v : 0u);
        if (lane == 63) {
            s_wave[0][wave] = incl;
            s_wave[1][wave] = sat_sum;
        }
        __syncthreads();
        uint32_t woff = 0, sat_all = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            woff += w < wave ? s_wave[0][w] : 0;
            sat_all += s_wave[1][w];
        }
        const uint32_t carry = s_carry[0];
        if (i < n_tiles) bucket_offsets[i] = carry + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) {
            s_carry[0] = carry + woff + incl;
            s_carry[1] += sat_all;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        bucket_offsets[n_tiles] = s_carry[0];
        *n_buckets = (unsigned long long)s_carry[0] | ((unsigned long long)s_carry[1] << 32);
    }
}
// the records: a wave per tile, lane b takes the tile's buckets b, b + 64, ... (a pile of a densifying run has hundreds)
__global__ void __launch_bounds__(256) bucket_fill_kernel(const uint32_t *__restrict__ tile_nproc, int n_tiles,
                                                         const uint32_t *__restrict__ bucket_offsets,
                                                         uint4 *__restrict__ bucket_info,
                                                         const int32_t *__restrict__ ranges, int frame_ranges) {
    const int tile = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (tile >= n_tiles) return;
    const uint32_t np = tile_nproc[tile], v = (np + GS_BUCKET - 1) / GS_BUCKET;
    if (v == 0) return;
    const uint32_t first = bucket_offsets[tile], st = (uint32_t)(frame_ranges ? ranges[2 * tile] : ranges[tile]);
    for (uint32_t b = (uint32_t)lane; b < v; b += 64) {
        const uint32_t rem = np - b * GS_BUCKET;
        bucket_info[first + b] = make_uint4((uint32_t)tile, b * GS_BUCKET, rem < GS_BUCKET ? rem : GS_BUCKET, st);
    }
}
static inline void gs_launch_bucket_list(const uint32_t *tile_nproc, int n_tiles, uint32_t *bucket_offsets,
                                         unsigned long long *n_buckets, uint4 *bucket_info, const int32_t *ranges,
                                         int frame_ranges, hipStream_t stream) {
    hipLaunchKernelGGL(bucket_scan_kernel, dim3(1), dim3(1024), 0, stream, tile_nproc, n_tiles, bucket_offsets, n_buckets,
                       ranges, frame_ranges);
    hipLaunchKernelGGL(bucket_fill_kernel, dim3((unsigned)gs_div_up(n_tiles, 4)), dim3(256), 0, stream, tile_nproc, n_tiles,
                       bucket_offsets, bucket_info, ranges, frame_ranges);
}

// Frame path, rgb rows: the key (depth bits, Gaussian) of the last list entry the forward processed in every tile
// (depth 0: none), as two arrays of T 32-bit words.  A tile's list ascends in exactly this key and the key is unique, so "pair (tile, g) was processed --
// its gradient row written" <=> key(g) <= stop_keys[tile]: what the projection backward needs to know about a row
// without a flag per row (no scattered flag stores, no flag memset; gs_frame_layout.h).  One thread per tile, three
// dependent loads; runs with the bucket scan underneath the caller's loss.
__global__ void __launch_bounds__(256) stop_key_kernel(const uint32_t *__restrict__ tile_nproc, int n_tiles,
                                                       const int32_t *__restrict__ ranges,
                                                       const uint32_t *__restrict__ sorted_ids,
                                                       const uint4 *__restrict__ rects,
                                                       unsigned long long *__restrict__ stop_keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tiles) return;
    const uint32_t np = tile_nproc[i];
    uint32_t depth = 0, id = 0;
    if (np) {
        id = sorted_ids[(uint32_t)ranges[2 * i] + np - 1];
        depth = rects[id].z;
    }
    // two arrays of T words, depth bits first: the reader decides almost every pair from the 4-byte depth word alone (a
    // table of 32 KiB at 1080p instead of 64) and looks at the Gaussian index only where the depth bits are equal
    uint32_t *sk = reinterpret_cast<uint32_t *>(stop_keys);
    sk[i] = depth;
    sk[n_tiles + i] = id;
}

struct BwdOut {
    // FRAME: one row per pair, addressed in EMISSION order (pair_offsets[g] + index of the tile
    // inside g's rectangle), so the per-Gaussian sum is a contiguous, deterministic reduction
    float *rows;                   // [max_pairs][16 | 48 | 64]: gs_frame_layout.h (gs_row_floats, gs_row_geo, gs_row_col)
    uint32_t *exec_rows;           // SH backward on the matrix pipe: pixel-row steps executed per (workgroup, wave)
    const uint32_t *pair_offsets;  // [N]
    const uint4 *rects;            // [N]
    uint64_t max_pairs;
    // REF: one row per pair
    float *grad_pos, *grad_rgb, *grad_opa, *grad_cov;
};

struct BwdIn {
    const float *c_final;     // padded image [padH,padW,3]
    const float *grad;        // FRAME: dL/d(cropped, clamped image) [H,W,3]; REF: [padH,padW,3]
    const float4 *ckpt;
    const uint32_t *tile_nproc;
    const uint32_t *bucket_offsets;  // [T+1]
    const uint4 *bucket_info;        // [n_buckets] (tile, first Gaussian, count, start of the tile's list)
    const int32_t *ranges;           // FRAME: [T][2]; REF: accum [T+1]
    const uint32_t *tile_order;      // FRAME, optional: the tiles in descending order of their cost (one workgroup per tile)
    // Long lists (frames flagged GS_FRAME_LONG_LISTS): the per-tile SH kernel takes the first bucket_cap buckets of a tile
    // (0: all of them), the one-wave-per-bucket kernel the buckets from bucket_first on (see launch_bwd)
    uint32_t bucket_cap, bucket_first;
    uint32_t use_rows;  // rgb frames: GS_FRAME_BWD_ROWS -- the row-layout kernel instead of the pixel-parallel one (launch_bwd)
    // SH frames on the matrix pipe: the work items (tile, first bucket, buckets) of mfma_items_kernel and their count
    const uint4 *mfma_items;
    const uint32_t *mfma_n_items;
};

// Inputs of one pixel for the backward kernels: final colour, and dL/dC -- for the frame path the gradient of
// the clamped + cropped output (splatter.py:652-653: zero outside the crop and where the colour was clamped).
// The gradient is loaded unconditionally and masked afterwards: a load that depends on c_final's value would
// serialise two memory round trips.
template <bool FRAME>
__device__ __forceinline__ void load_pixel_inputs(const BwdIn &I, const RasterGeom &G, uint32_t id_x, uint32_t id_y,
                                                  float f[3], float g[3]) {
    const float *cf = I.c_final + ((size_t)id_y * G.padW + id_x) * 3;
    f[0] = cf[0];
    f[1] = cf[1];
    f[2] = cf[2];
    if (FRAME) {
        const int ox = (int)id_x - G.crop_left, oy = (int)id_y - G.crop_top;
        const bool in = ox >= 0 && ox < G.width && oy >= 0 && oy < G.height;
        const float *gp = I.grad + ((size_t)(in ? oy : 0) * G.width + (in ? ox : 0)) * 3;
        const float u0 = gp[0], u1 = gp[1], u2 = gp[2];
        g[0] = (in && f[0] >= 0.f && f[0] <= 1.f) ? u0 : 0.f;
        g[1] = (in && f[1] >= 0.f && f[1] <= 1.f) ? u1 : 0.f;
        g[2] = (in && f[2] >= 0.f && f[2] <= 1.f) ? u2 : 0.f;
    } else {
        const float *gp = I.grad + ((size_t)id_y * G.padW + id_x) * 3;
        g[0] = gp[0];
        g[1] = gp[1];
        g[2] = gp[2];
    }
}

// waves per workgroup of the systolic kernel: the degree-3 SH basis records need 20 KiB of LDS per wave, and
// single-wave workgroups pack the CU's 160 KiB best (the waves of a workgroup never synchronise anyway)
template <int CDIM>
struct BwdCfg {
    static constexpr int WPB = CDIM == 48 ? 1 : 4;
    // What is constant per pixel (dL/dC, pixel centre) either rides the DPP chain with the pixel's running state or
    // is read by every lane from an LDS table indexed by the pixel it currently holds.  A DPP move costs as much
    // SIMD time as three plain instructions (tools/ubench/pk_rate.hip), a per-lane LDS read costs no VALU slot but
    // LDS bandwidth and 4 KiB per wave.  Measured (cfg2 / cfg4, same run): the table is 9 % faster without SH
    // (issue-bound, 5 waves per SIMD) and 2 % slower with SH (register-limited to 2-3 waves, latency-bound, and
    // the SH basis records already load the LDS pipe).
    static constexpr bool TABLE = CDIM == 3;
    static constexpr int NB = CDIM > 3 ? CDIM / 3 : 1;        // SH basis functions per channel
    static constexpr int SHS = CDIM == 48 ? 20 : NB;          // LDS record stride (floats): 9 is conflict-free for
                                                              // scalar reads, 16 + 4 keeps float4 reads aligned
};

// SIG (reference API only): the `sigmoid` flag of draw / draw_backward -- alpha is squashed,
// alpha = 2 / (1 + e^-a) - 1 with a = p0 G opa, p0 = (pi/2) rsqrt(det + 1e-7) (gaussian.cu:593-594, 918, 930).
// p0 is folded into the lane's opacity; its own cov gradient (gaussian.cu:622-630) only needs sum(dL/da G), which
// is the opacity accumulator, so the loop pays two extra transcendentals and three plain instructions.
// EXACT (reference API, `fast = 0`, gaussian.cu:596-603): the Gaussian's value through a double-precision exp of the
// float argument, as the reference's backward evaluates it; v_exp_f32 otherwise.
template <int CDIM, bool FRAME, bool SIG = false, bool EXACT = false>
__global__ void __launch_bounds__(64 * BwdCfg<CDIM>::WPB) raster_backward_kernel(RasterSrc S, RasterGeom G, BwdIn I,
                                                                              BwdOut O) {
    static_assert(!(FRAME && SIG), "the frame path has no alpha squashing (splatter.py:627 passes sigmoid=False)");
    constexpr int WPB = BwdCfg<CDIM>::WPB, NB = BwdCfg<CDIM>::NB, SHS = BwdCfg<CDIM>::SHS;
    // Feed ring: the 256 pixel states of a bucket enter lane 0 in four segments of 64; while a
    // segment streams through the lanes, the next one is prefetched into registers and then
    // written to the other half of the ring.  ~6 KiB of LDS per wave keeps occupancy
    // register-limited (the dependent DPP chain needs >= 5 waves per SIMD to stay hidden).
    constexpr bool TABLE = BwdCfg<CDIM>::TABLE;
    constexpr int NF = TABLE ? 1 : 2;  // float4 feed records per pixel: (T, rho, -, -) | (T, rho, px, py), (dL/dC, -)
    __shared__ float4 s_feed[WPB][2][NF][64];
    __shared__ float4 s_pix[TABLE ? WPB : 1][TABLE ? 256 : 1];  // TABLE: (dL/dC rgb, pixel centre x) of tile pixel p
    __shared__ float s_py[TABLE ? WPB : 1][16];                 // TABLE: pixel centre y of tile row p >> 4
    // SH only: the 9 (16) basis values of a pixel never change while it travels, so they do not ride the DPP chain
    // (9 moves = 28 ns of SIMD time per step, tools/ubench/pk_rate.hip); they are staged once per bucket and
    // every lane reads the record of the pixel it currently holds (3-4 LDS reads, no VALU issue slots).
    __shared__ float s_sh[CDIM > 3 ? WPB : 1][CDIM > 3 ? 256 * SHS : 4] __attribute__((aligned(16)));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t n_tiles = (uint32_t)(G.ntx * G.nty);
    const uint32_t kb = blockIdx.x * WPB + wave;
    if (kb >= I.bucket_offsets[n_tiles]) return;  // whole wave exits; no block-level barrier below

    const uint4 info = I.bucket_info[kb];
    const uint32_t tile = info.x, b = info.y / GS_BUCKET;  // bucket -> (tile, local bucket)
    const uint32_t tx = tile % (uint32_t)G.ntx, ty = tile / (uint32_t)G.ntx;
    const uint32_t start = (uint32_t)(FRAME ? I.ranges[2 * tile] : I.ranges[tile]);
    const uint32_t nproc = I.tile_nproc[tile];
    const uint32_t jloc = b * GS_BUCKET + lane;
    const bool gvalid = jloc < nproc;
    const uint32_t j = start + jloc;
    const int nvalid = (int)((nproc - b * GS_BUCKET) < (uint32_t)GS_BUCKET ? (nproc - b * GS_BUCKET) : GS_BUCKET);

    // ---- per-segment staging of the pixel states (pixel p = 64 seg + lane)
    const float4 *ck = I.ckpt + raster_ckpt_slot(start, tile, b) * 256;
    const uint32_t id_x = tx * 16 + (lane & 15);
    const float my_px = raster_pixel_coord(id_x, G.padW, G.focal_x);
    float4 pc;            // prefetched checkpoint (T, C_run)
    float pf0, pf1, pf2;  // final colour
    float pg0, pg1, pg2;  // dL/dC
    auto load_segment = [&](int seg) {
        const int p = seg * 64 + lane;
        const uint32_t id_y = ty * 16 + (p >> 4);
        pc = b == 0 ? make_float4(1.f, 0.f, 0.f, 0.f) : ck[p];  // bucket 0 starts from the empty pixel (not stored)
        float f[3], g[3];
        load_pixel_inputs<FRAME>(I, G, id_x, id_y, f, g);
        pf0 = f[0];
        pf1 = f[1];
        pf2 = f[2];
        pg0 = g[0];
        pg1 = g[1];
        pg2 = g[2];
    };
    auto write_segment = [&](int buf, int seg) {
        const uint32_t id_y = ty * 16 + seg * 4 + (lane >> 4);
        const float py = raster_pixel_coord(id_y, G.padH, G.focal_y);
        // rho = dL/dC . (C_final - C_run): the only way the remaining colour enters dL/dalpha
        // (gaussian.cu:716-722), so one scalar travels instead of three colour channels
        const float rho = pg0 * (pf0 - pc.y) + pg1 * (pf1 - pc.z) + pg2 * (pf2 - pc.w);
        s_feed[wave][buf][0][lane] = make_float4(pc.x, rho, my_px, py);
        if (TABLE) {
            s_pix[wave][seg * 64 + lane] = make_float4(pg0, pg1, pg2, my_px);
            if ((lane & 15) == 0) s_py[wave][seg * 4 + (lane >> 4)] = py;
        } else {
            s_feed[wave][buf][NF - 1][lane] = make_float4(pg0, pg1, pg2, 0.f);
        }
        if (CDIM > 3) {
            float SH[NB];
            raster_pixel_sh<NB>(id_x, id_y, G, SH);
#pragma unroll
            for (int q = 0; q < NB; ++q) s_sh[wave][(seg * 64 + lane) * SHS + q] = SH[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    load_segment(0);
    if (TABLE) {
        // lanes that hold no pixel yet read the records of segment 3 before it is staged: their T is exactly zero,
        // but 0 x (whatever the previous kernel left in LDS, e.g. the tile sort's all-ones padding = NaN) must stay 0
        s_pix[wave][192 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < 4) s_py[wave][12 + lane] = 0.f;
    }
    if (CDIM > 3) {
        // lanes that hold no pixel yet read records of segments that are not staged yet: their weight is exactly
        // zero, but 0 x (whatever the previous kernel left in LDS, e.g. the tile sort's all-ones padding = NaN)
        // must not reach the accumulators
#pragma unroll
        for (int q = 0; q < 4 * SHS; ++q) s_sh[wave][q * 64 + lane] = 0.f;
    }

    // ---- this lane's Gaussian
    GaussianRec g = {0, 0, 0, 0, 0, 0, 0};
    float col0 = 0, col1 = 0, col2 = 0;
    float coef[CDIM > 3 ? CDIM : 1];
    uint32_t gid = 0;
    if (gvalid) {
        gid = raster_load<FRAME>(S, j, g);
        if (CDIM == 3) {
            raster_load_rgb<FRAME>(S, j, gid, col0, col1, col2);
        } else {
            const float *src = raster_sh_ptr<FRAME, CDIM>(S, j, gid);
#pragma unroll
            for (int q = 0; q < CDIM; ++q) coef[q] = src[q];
        }
    } else if (CDIM > 3) {
#pragma unroll
        for (int q = 0; q < CDIM; ++q) coef[q] = 0.f;
    }
    float cA = 0, cB = 0, cC = 0;
    if (gvalid) raster_conic(g, cA, cB, cC);
    float opa = gvalid ? g.opa : 0.f;  // opacity 0 => alpha 0 => state passes through unchanged
    float p0 = 1.0f;
    if (SIG) {  // the same folding as the forward kernel (raster_fwd.hip)
        p0 = 1.5707963268f * rsqrtf(raster_det(g.a, g.b, g.c, g.d) + 1e-7f);
        opa *= p0;
    }
    write_segment(0, 0);

    // gradient accumulators
    float Sx = 0, Sy = 0, Sxx = 0, Sxy = 0, Syy = 0, Sq = 0, Sopa = 0, Sc0 = 0, Sc1 = 0, Sc2 = 0;
    float Ssh[CDIM > 3 ? CDIM : 1];
    if (CDIM > 3) {
#pragma unroll
        for (int q = 0; q < CDIM; ++q) Ssh[q] = 0.f;
    }
    // outgoing state of the previous step (T = 0 means "no pixel here")
    float oT = 0, orho = 0, og0 = 0, og1 = 0, og2 = 0, opx = 0, opy = 0;

    for (int seg = 0; seg < 5; ++seg) {
        const int buf = seg & 1;
        if (seg + 1 < 4) load_segment(seg + 1);  // in flight while this segment streams
        const bool feeding = seg < 4;
        // drain: the last pixel leaves lane nvalid-1 after nvalid-1 more steps; rounded up to an even count (the
        // extra step moves an empty slot) so that the loop unrolls by two without a remainder loop, which the
        // convergent DPP moves forbid
        const int nsteps = feeding ? 64 : (nvalid & ~1);
        auto step = [&](const int t) {
            // feed for lane 0 (broadcast LDS reads; while draining feed T = 0)
            const float4 f0 = s_feed[wave][buf][0][t];
            const float fT = feeding ? f0.x : 0.f;
            // state entering this lane: lane l-1's output of the previous step; lane 0 takes the feed
            const float T = gs_wave_shr1(fT, oT);
            float rho = gs_wave_shr1(f0.y, orho);
            // the pixel in this lane entered lane 0 `lane` steps ago: p = 64 seg + t - lane (lanes that hold no
            // pixel yet / any more read a valid but irrelevant record: their T is 0)
            const int p = (seg * 64 + t - lane) & 255;
            float px, py, g0, g1, g2;
            if (TABLE) {
                const float4 pr = s_pix[wave][p];
                g0 = pr.x, g1 = pr.y, g2 = pr.z, px = pr.w, py = s_py[wave][p >> 4];
            } else {
                const float4 f1 = s_feed[wave][buf][NF - 1][t];
                px = gs_wave_shr1(f0.z, opx), py = gs_wave_shr1(f0.w, opy);
                g0 = gs_wave_shr1(f1.x, og0), g1 = gs_wave_shr1(f1.y, og1), g2 = gs_wave_shr1(f1.z, og2);
            }
            float sh[NB];
            if (CDIM > 3) {
                const float *rec = &s_sh[wave][p * SHS];
                if (NB % 4 == 0) {
#pragma unroll
                    for (int q = 0; q < NB / 4; ++q) {
                        const float4 v = reinterpret_cast<const float4 *>(__builtin_assume_aligned(rec, 16))[q];
                        sh[4 * q] = v.x;
                        sh[4 * q + 1] = v.y;
                        sh[4 * q + 2] = v.z;
                        sh[4 * q + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < NB; ++q) sh[q] = rec[q];
                }
            }

            const float dx = px - g.x, dy = py - g.y;
            const float q = fmaf(cC * dy, dy, dx * fmaf(-cB, dy, cA * dx));  // same evaluation order as the forward
            const float Gv = EXACT ? (float)exp(-(double)(q * GS_LN2)) : gs_exp2(-q);
            const bool live = T > GS_T_STOP;
            const float araw = live ? Gv * opa : 0.f;  // before squashing
            const float alpha = SIG ? gs_squash_alpha(araw) : araw;
            const float w = alpha * T;
            if (CDIM > 3) {
                float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    v0 = fmaf(sh[k], coef[k], v0);
                    v1 = fmaf(sh[k], coef[NB + k], v1);
                    v2 = fmaf(sh[k], coef[2 * NB + k], v2);
                }
                col0 = gs_rcp(1.0f + __expf(-v0));
                col1 = gs_rcp(1.0f + __expf(-v1));
                col2 = gs_rcp(1.0f + __expf(-v2));
            }
            // dL/dC . colour of this Gaussian, and rho AFTER it (the reference's cur_out - color, :719)
            const float gc = fmaf(g2, col2, fmaf(g1, col1, g0 * col0));
            rho = fmaf(-w, gc, rho);
            const float one_m = 1.0f - alpha;
            float d_alpha = fmaf(T, gc, -(rho * gs_rcp(one_m + 1e-7f)));
            d_alpha = live ? d_alpha : 0.f;
            if (SIG) d_alpha *= (alpha + 1.0f) - 0.5f * (alpha + 1.0f) * (alpha + 1.0f);  // d squash / da, :727
            if (CDIM > 3) {
                const float D0 = g0 * w * (col0 * (1.0f - col0)), D1 = g1 * w * (col1 * (1.0f - col1)),
                            D2 = g2 * w * (col2 * (1.0f - col2));
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    Ssh[k] = fmaf(D0, sh[k], Ssh[k]);
                    Ssh[NB + k] = fmaf(D1, sh[k], Ssh[NB + k]);
                    Ssh[2 * NB + k] = fmaf(D2, sh[k], Ssh[2 * NB + k]);
                }
            } else {
                Sc0 = fmaf(g0, w, Sc0);
                Sc1 = fmaf(g1, w, Sc1);
                Sc2 = fmaf(g2, w, Sc2);
            }
            Sopa = fmaf(d_alpha, Gv, Sopa);
            const float s = d_alpha * araw;
            const float sdx = s * dx, sdy = s * dy;
            Sx += sdx;
            Sy += sdy;
            Sxx = fmaf(sdx, dx, Sxx);
            Sxy = fmaf(sdx, dy, Sxy);
            Syy = fmaf(sdy, dy, Syy);
            Sq = fmaf(s, q, Sq);
            // outgoing state
            oT = fmaf(-alpha, T, T);  // identical to the forward's update
            orho = rho;
            og0 = g0;
            og1 = g1;
            og2 = g2;
            opx = px;
            opy = py;
        };
        for (int t = 0; t < nsteps; t += 2) {  // nsteps is even: two steps per iteration, no remainder loop
            step(t);
            step(t + 1);
        }
        if (seg + 1 < 4) write_segment(buf ^ 1, seg + 1);
    }

    if (!gvalid) return;
    const float det = raster_det(g.a, g.b, g.c, g.d);
    const float iPn = 1.0f / (2.0f * det + 1e-14f);
    const float Su = Sq * GS_LN2;  // u = -ln G = q ln 2
    const float gx = GS_LN2 * (2.0f * cA * Sx - cB * Sy);
    const float gy = GS_LN2 * (2.0f * cC * Sy - cB * Sx);
    float ga = iPn * (-Syy + 2.0f * g.d * Su);
    float gb = iPn * (Sxy - 2.0f * g.c * Su);
    float gc = iPn * (Sxy - 2.0f * g.b * Su);
    float gd = iPn * (-Sxx + 2.0f * g.a * Su);
    if (SIG) {
        // dp0/d{a,b,c,d} = k0 (-d, c, b, -a), k0 = p0^3 / (2 (pi/2)^2) (gaussian.cu:622-626), times
        // sum_pixels dL/da G opa_raw (:740-744); Sopa so far is sum dL/da G
        const float k = 0.5f * (p0 * p0 * p0) / (1.5707963268f * 1.5707963268f) * Sopa * g.opa;
        ga -= k * g.d;
        gb += k * g.c;
        gc += k * g.b;
        gd -= k * g.a;
        Sopa *= p0;  // dL/dopa = sum dL/da (p0 G)
    }
    if (FRAME) {
        const uint4 rc = O.rects[gid];
        const uint32_t y0 = rc.x & 0xffff, x0 = rc.y & 0xffff, x1 = rc.y >> 16;
        const uint64_t slot = (uint64_t)O.pair_offsets[gid] + (ty - y0) * (x1 - x0) + (tx - x0);
        if (slot < O.max_pairs) {
            // (the A/B fallback of the frame path -- GS_BWD_SH_PIXEL = 0 -- and no hot path: float by float through the row
            // layout of gs_frame_layout.h, every float of the row written)
            constexpr int RW = gs_row_floats(CDIM);
            float *row = O.rows + slot * RW;
#pragma unroll
            for (int m = 0; m < RW; ++m) row[m] = 0.f;
            const float geo[7] = {gx, gy, ga, gb, gc, gd, Sopa};
#pragma unroll
            for (int m = 0; m < 7; ++m) row[gs_row_geo(CDIM, m)] = geo[m];
            if (CDIM == 3) {
                row[gs_row_col(3, 0)] = Sc0;
                row[gs_row_col(3, 1)] = Sc1;
                row[gs_row_col(3, 2)] = Sc2;
            } else {
#pragma unroll
                for (int k = 0; k < CDIM; ++k) row[gs_row_col(CDIM, k)] = Ssh[k];
            }
        }
    } else {
        O.grad_pos[(size_t)j * 3 + 0] = gx;
        O.grad_pos[(size_t)j * 3 + 1] = gy;
        O.grad_opa[j] = Sopa;
        reinterpret_cast<float4 *>(O.grad_cov)[j] = make_float4(ga, gb, gc, gd);
        if (CDIM == 3) {
            O.grad_rgb[(size_t)j * 3 + 0] = Sc0;
            O.grad_rgb[(size_t)j * 3 + 1] = Sc1;
            O.grad_rgb[(size_t)j * 3 + 2] = Sc2;
        } else {
#pragma unroll
            for (int k = 0; k < CDIM; ++k) O.grad_rgb[(size_t)j * CDIM + k] = Ssh[k];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// No SH: the pixel-parallel bucket kernel.  One wave per (tile, bucket of 64 Gaussians), lanes own PIXELS (four
// each, the forward's layout), the bucket's Gaussians are applied one after the other from the bucket's checkpoint,
// and the ten per-Gaussian sums are reduced over the wave through LDS, two Gaussians at a time: 20 rows of 64
// partials, three lanes per row add a slice each (60 lanes busy), one lane per row adds the three slice sums, and
// at the end lane i finishes the algebra of Gaussian i and stores its row.
// Why not the systolic pipeline here: per (pixel, Gaussian) pair it issues 0.89 VALU instructions (every lane
// carries one pair per step: its own q, exp, rcp, DPP moves, table addressing; PMC), this layout 0.59 (dx and the
// Gaussian's constants are shared by the lane's four pixels, the reduction costs 20 of 146 instructions per
// Gaussian) -- and both are VALU-issue bound.  Measured, same run: cfg2 0.450 -> 0.362 ms, 2.4 M Gaussians
// 0.776 -> 0.611 ms.  It started as the handler of the ragged tails of the systolic kernel (a 5-Gaussian bucket
// costs the systolic pipeline as much as a full one) and took over the whole no-SH backward.
#ifndef GS_PP_TG
#define GS_PP_TG 2
#endif
#ifndef GS_PP_WPB
#define GS_PP_WPB 1
#endif
template <bool FRAME>
__global__ void __launch_bounds__(64 * GS_PP_WPB) raster_backward_pixel_kernel(RasterSrc S, RasterGeom G, BwdIn I,
                                                                                BwdOut O) {
    constexpr int TG = GS_PP_TG;          // Gaussians per reduction group (LDS: TG x 10 rows of 64 partials)
    constexpr int LPR = 60 / (10 * TG);   // lanes that share the sum of one row (two-level reduction)
    constexpr int CH = (64 + LPR - 1) / LPR;
    enum { FX, FY, FA, FB, FC, FOPA, FC0, FC1, FC2, NFLD };
    constexpr int WPB = GS_PP_WPB;  // waves per workgroup, one bucket each (they never synchronise); 1 measured best
    __shared__ float sw_g[WPB][NFLD][64];
    __shared__ float sw_red[WPB][TG * 10][65];
    __shared__ float sw_part[WPB][TG * 10][LPR + 1];
    __shared__ float sw_tot[WPB][64][10];
    const int wave = threadIdx.x >> 6;
    float (*s_g)[64] = sw_g[wave];
    float (*s_red)[65] = sw_red[wave];
    float (*s_part)[LPR + 1] = sw_part[wave];
    float (*s_tot)[10] = sw_tot[wave];
    // LDS traffic of ONE wave is processed in program order, so the stages below only need the compiler
    // not to move LDS accesses across them.  (A `fence acq_rel` would also drain the outstanding global
    // loads and stores -- measured: 80 us of a 124 us kernel when the row stores sat inside the loop.)
    auto lds_order = [] {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    };
    const int lane = threadIdx.x & 63;
    // ---- work item: ONE load; everything below is issued before anything is waited for (the loads of the wave's
    // short life -- bucket record, ids -> Gaussian records, checkpoints, pixel inputs -- used to be eight dependent
    // round trips, ~9 us in front of ~11 us of arithmetic)
    const uint32_t n_tiles = (uint32_t)(G.ntx * G.nty), kb = blockIdx.x * WPB + wave;
    const uint4 info = I.bucket_info[kb];  // in bounds for every launched wave (the table is padded)
    if (kb >= I.bucket_offsets[n_tiles]) return;
    const uint32_t tile = info.x, base = info.y, r = info.z, start = info.w;
    const uint32_t tx = tile % (uint32_t)G.ntx, ty = tile / (uint32_t)G.ntx;

    // ---- this lane's Gaussian (lanes >= r re-read the bucket's last one and are zeroed below: no branch around
    // the loads) and its four pixels (x, y0 + 4k)
    GaussianRec g;
    float c0, c1, c2;
    const uint32_t jl = start + base + ((uint32_t)lane < r ? (uint32_t)lane : r - 1);
    const uint32_t gid = raster_load<FRAME>(S, jl, g);
    raster_load_rgb<FRAME>(S, jl, gid, c0, c1, c2);
    const uint32_t id_x = tx * 16 + (lane & 15), id_y0 = ty * 16 + (lane >> 4);
    const float px = raster_pixel_coord(id_x, G.padW, G.focal_x);
    const float4 *ck = I.ckpt + raster_ckpt_slot(start, tile, base / GS_BUCKET) * 256;
    float4 c[4];
    float f[4][3], gr[4][3];
    bool inside[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        c[k] = ck[64 * k + lane];  // tile pixel index = 16 (y - ty 16) + (x - tx 16); stale for the first bucket
        // final colour and dL/dC of the pixel (see load_pixel_inputs): raw loads from clamped addresses here, the
        // crop / clamp masks after ALL loads are in flight
        const uint32_t id_y = id_y0 + 4 * k;
        const float *cf = I.c_final + ((size_t)id_y * G.padW + id_x) * 3;
        const int ox = FRAME ? (int)id_x - G.crop_left : (int)id_x, oy = FRAME ? (int)id_y - G.crop_top : (int)id_y;
        inside[k] = !FRAME || (ox >= 0 && ox < G.width && oy >= 0 && oy < G.height);
        const float *gp = I.grad + ((size_t)(inside[k] ? oy : 0) * (FRAME ? G.width : G.padW) + (inside[k] ? ox : 0)) * 3;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            f[k][e] = cf[e];
            gr[k][e] = gp[e];
        }
    }
    // the compiler would otherwise sink each gradient load behind its `inside` test and wait for it there: four
    // dependent round trips instead of one
#pragma unroll
    for (int k = 0; k < 4; ++k)
        asm volatile("" ::"v"(gr[k][0]), "v"(gr[k][1]), "v"(gr[k][2]), "v"(f[k][0]), "v"(f[k][1]), "v"(f[k][2]),
                     "v"(c[k].x), "v"(c[k].y), "v"(c[k].z), "v"(c[k].w));
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 3; ++e)
            gr[k][e] = (inside[k] && (!FRAME || (f[k][e] >= 0.f && f[k][e] <= 1.f))) ? gr[k][e] : 0.f;
    float cA = 0, cB = 0, cC = 0;
    {
        const bool valid = (uint32_t)lane < r;
        raster_conic(g, cA, cB, cC);
        // what the pixel loop reads goes to LDS; what only the final algebra of THIS lane's Gaussian needs
        // (covariance, conic, id) stays in its registers
        s_g[FX][lane] = g.x;
        s_g[FY][lane] = g.y;
        s_g[FA][lane] = cA;
        s_g[FB][lane] = cB;
        s_g[FC][lane] = cC;
        s_g[FOPA][lane] = valid ? g.opa : 0.f;  // opacity 0 => alpha 0: padded entries contribute exact zeros
        s_g[FC0][lane] = c0;
        s_g[FC1][lane] = c1;
        s_g[FC2][lane] = c2;
    }
    float py[4], T[4], rho[4], g0[4], g1[4], g2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        py[k] = raster_pixel_coord(id_y0 + 4 * k, G.padH, G.focal_y);
        // the tile's first bucket starts from the empty pixel (T, C) = (1, 0), which the forward does not store
        const float4 ci = base == 0 ? make_float4(1.f, 0.f, 0.f, 0.f) : c[k];
        g0[k] = gr[k][0];
        g1[k] = gr[k][1];
        g2[k] = gr[k][2];
        T[k] = ci.x;
        rho[k] = g0[k] * (f[k][0] - ci.y) + g1[k] * (f[k][1] - ci.z) + g2[k] * (f[k][2] - ci.w);
    }
    lds_order();

    for (uint32_t i0 = 0; i0 < r; i0 += TG) {
        // Step 1, everything that does not depend on the pixels' running state: the exponent, G = 2^-q and
        // 1 / (1 - alpha + 1e-7) for the TG x 4 (Gaussian, pixel) pairs -- 2 TG x 4 independent transcendentals
        // in flight (left inside the serial chain below they cost ~10 ns each instead of 3.4: ablation).  The
        // reciprocal is taken for the live case; a finished pixel has d_alpha forced to zero whatever it is.
        float dxs[TG], dys[TG][4], qs[TG][4], Gs[TG][4], rcs[TG][4], ops[TG], cc[TG][3];
#pragma unroll
        for (int u = 0; u < TG; ++u) {
            const uint32_t i = i0 + u;  // < 64: padded entries have opacity 0 and contribute exact zeros
            const float gx = s_g[FX][i], gy = s_g[FY][i], uA = s_g[FA][i], uB = s_g[FB][i], uC = s_g[FC][i];
            ops[u] = s_g[FOPA][i];
            cc[u][0] = s_g[FC0][i];
            cc[u][1] = s_g[FC1][i];
            cc[u][2] = s_g[FC2][i];
            const float dx = px - gx;
            const float bdx = uB * dx, adx2 = uA * dx * dx;
            dxs[u] = dx;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dy = py[k] - gy;
                const float q = fmaf(fmaf(uC, dy, -bdx), dy, adx2);
                const float Gv = gs_exp2(-q);
                dys[u][k] = dy;
                qs[u][k] = q;
                Gs[u][k] = Gv;
                // 1 / (1 - alpha + 1e-7) (gaussian.cu:722) as ONE fma + rcp: the constant is the fp32 neighbour of
                // 1 + 1e-7 (1 + 2^-23), 2e-8 away -- below the rounding of the two-step form it replaces
                rcs[u][k] = gs_rcp(fmaf(-Gv, ops[u], 1.00000011920928955f));
            }
        }
        // Step 2, front to back through the group: the chain through T and rho is plain arithmetic only
#pragma unroll
        for (int u = 0; u < TG; ++u) {
            const float dx = dxs[u], opa = ops[u], c0 = cc[u][0], c1 = cc[u][1], c2 = cc[u][2];
            // dx is the same for the lane's four pixels: sum(s), sum(s dy) are accumulated and multiplied by dx / dx^2
            // once per Gaussian (Sx, Sxx, Sxy need no per-pixel instruction)
            float S1 = 0, Sy = 0, Syy = 0, Sq = 0, Sopa = 0, Sc0 = 0, Sc1 = 0, Sc2 = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dy = dys[u][k], q = qs[u][k], Gv = Gs[u][k];
                const bool live = T[k] > GS_T_STOP;
                const float alpha = live ? Gv * opa : 0.f;
                const float w = alpha * T[k];
                const float gc = fmaf(g2[k], c2, fmaf(g1[k], c1, g0[k] * c0));
                rho[k] = fmaf(-w, gc, rho[k]);
                float d_alpha = fmaf(T[k], gc, -(rho[k] * rcs[u][k]));
                d_alpha = live ? d_alpha : 0.f;
                Sc0 = fmaf(g0[k], w, Sc0);
                Sc1 = fmaf(g1[k], w, Sc1);
                Sc2 = fmaf(g2[k], w, Sc2);
                Sopa = fmaf(d_alpha, Gv, Sopa);
                const float s = d_alpha * alpha;
                const float sdy = s * dy;
                S1 += s;
                Sy += sdy;
                Syy = fmaf(sdy, dy, Syy);
                Sq = fmaf(s, q, Sq);
                T[k] = T[k] - w;
            }
            const float Sx = S1 * dx, Sxx = Sx * dx, Sxy = Sy * dx;
            float *red = &s_red[u * 10][lane];
            red[0 * 65] = Sx;
            red[1 * 65] = Sy;
            red[2 * 65] = Sxx;
            red[3 * 65] = Sxy;
            red[4 * 65] = Syy;
            red[5 * 65] = Sq;
            red[6 * 65] = Sopa;
            red[7 * 65] = Sc0;
            red[8 * 65] = Sc1;
            red[9 * 65] = Sc2;
        }
        lds_order();
        // two-level sum of the TG x 10 rows: LPR lanes per row add a slice each (60 lanes busy), then one lane per
        // row adds the LPR slice sums.  Row 10 u + m = value m of Gaussian i0 + u.
        if (lane < TG * 10 * LPR) {
            const int row = lane / LPR, part = lane % LPR;
            float t0 = 0, t1 = 0;
#pragma unroll
            for (int l = 0; l < CH; l += 2) {
                const int a = part * CH + l;
                t0 += a < 64 ? s_red[row][a] : 0.f;
                t1 += a + 1 < 64 ? s_red[row][a + 1] : 0.f;
            }
            s_part[row][part] = t0 + t1;
        }
        lds_order();
        if (lane < TG * 10) {
            float t = s_part[lane][0];
#pragma unroll
            for (int l = 1; l < LPR; ++l) t += s_part[lane][l];
            s_tot[i0 + lane / 10][lane % 10] = t;
        }
        lds_order();
    }
    if ((uint32_t)lane < r) {  // one lane per Gaussian finishes the algebra and stores its row
        const uint32_t i = lane;
        const float *t = s_tot[lane];
        const float Sx = t[0], Sy = t[1], Sxx = t[2], Sxy = t[3], Syy = t[4], Sq = t[5];
        const float a = g.a, b = g.b, c = g.c, d = g.d;
        const float iPn = 1.0f / (2.0f * raster_det(a, b, c, d) + 1e-14f);
        const float Su = Sq * GS_LN2;
        const float ogx = GS_LN2 * (2.0f * cA * Sx - cB * Sy), ogy = GS_LN2 * (2.0f * cC * Sy - cB * Sx);
        const float ga = iPn * (-Syy + 2.0f * d * Su), gb = iPn * (Sxy - 2.0f * c * Su);
        const float gcc = iPn * (Sxy - 2.0f * b * Su), gd = iPn * (-Sxx + 2.0f * a * Su);
        if (FRAME) {
            const uint4 rc = O.rects[gid];
            const uint32_t y0 = rc.x & 0xffff, x0 = rc.y & 0xffff, x1 = rc.y >> 16;
            const uint64_t slot = (uint64_t)O.pair_offsets[gid] + (ty - y0) * (x1 - x0) + (tx - x0);
            if (slot < O.max_pairs) {
                float4 *row = reinterpret_cast<float4 *>(O.rows + slot * gs_row_floats(3));
                row[0] = make_float4(ogx, ogy, ga, gb);
                row[1] = make_float4(gcc, gd, t[6], t[7]);
                row[2] = make_float4(t[8], t[9], 0.f, 0.f);
                row[3] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            const size_t j = (size_t)start + base + i;
            O.grad_pos[j * 3 + 0] = ogx;
            O.grad_pos[j * 3 + 1] = ogy;
            O.grad_opa[j] = t[6];
            reinterpret_cast<float4 *>(O.grad_cov)[j] = make_float4(ga, gb, gcc, gd);
            O.grad_rgb[j * 3 + 0] = t[7];
            O.grad_rgb[j * 3 + 1] = t[8];
            O.grad_rgb[j * 3 + 2] = t[9];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// SH: the pixel-parallel bucket kernel with per-pixel colours.  Same layout as raster_backward_pixel_kernel -- one
// wave per (tile, bucket of 64 Gaussians), lanes own four PIXELS (x, y0 + 4k) as two packed pairs, the bucket's
// Gaussians are applied front to back from the bucket's checkpoint -- plus what SH adds per (pixel, Gaussian):
//   colour_c = sigmoid(sum_k sh_k(pixel) coef[c][k])   (27 / 48 FMAs forward, packed over the pixel pair)
//   dL/dcoef[c][k] += dL/dC_c w colour_c (1 - colour_c) sh_k(pixel)   (27 / 48 sums over the tile's pixels)
// The basis values of the lane's four pixels live in registers for the whole bucket (9 or 16 pairs x 2), the
// coefficients of the current Gaussian are wave-uniform (read through the scalar cache), and the 7 + 27 (48) sums of
// a Gaussian are reduced over the wave through LDS: every lane adds NROW consecutive entries of the [NROW][64] array
// of partials, the lanes that own a row add the two or three slice sums that fall into it.  The colour and opacity
// sums go straight to the Gaussian's gradient row; the six geometry sums wait in LDS for the closing algebra.
// Why it replaces the systolic kernel for the frame path: that one spends ~125 VALU instructions per (pixel,
// Gaussian) step of ONE pixel per lane plus 25 % pipeline fill; here the per-pixel math is packed two pixels per
// instruction (v_pk_fma_f32 runs at the rate of v_fma_f32 on this chip: tools/ubench/sort_ops.hip), dx and the
// Gaussian's constants are shared by the lane's four pixels and there is no fill: ~400 instructions per Gaussian
// per wave = 1.6 per (pixel, Gaussian) against 2.4.
template <int CDIM>
struct PixShCfg {
    static constexpr int NB = CDIM > 3 ? CDIM / 3 : 1;
    static constexpr int NROW = 7 + CDIM;  // Sx Sy Sxx Sxy Syy Sq Sopa + the colour(-coefficient) sums
};

#ifndef GS_BWD_SH48_WPE
#define GS_BWD_SH48_WPE 3
#endif
#ifndef GS_BWD_SH_SCALAR_ROWS
#define GS_BWD_SH_SCALAR_ROWS 1  // A/B switch (tools/ab_variants.py): see row_value below
#endif
#ifndef GS_BWD_SH_WPE
#define GS_BWD_SH_WPE 4  // waves per SIMD the register allocation aims at (A/B switch, tools/ab_variants.py)
#endif
#ifndef GS_BWD_PAIR_SKIP
#define GS_BWD_PAIR_SKIP 1
#endif
template <int CDIM, bool FRAME, bool EXACT = false>
__global__ void __launch_bounds__(64)
__attribute__((amdgpu_waves_per_eu(CDIM == 48 ? GS_BWD_SH48_WPE : CDIM == 3 ? 5 : GS_BWD_SH_WPE)))  // rgb: five (LDS: 7.9 KiB per wave)
raster_backward_pixel_sh_kernel(RasterSrc S, RasterGeom G, BwdIn I, BwdOut O) {
    static_assert(!(EXACT && FRAME), "the exact-exp flavour belongs to the reference API (gs_draw_backward, fast = 0)");
    constexpr int NB = PixShCfg<CDIM>::NB, NROW = PixShCfg<CDIM>::NROW;
    constexpr bool PRE = GS_SH_PRESCALE && CDIM == 27;  // SH basis pre-scaled by -log2(e): raster_common.h
    typedef float f2 __attribute__((ext_vector_type(2)));
    enum { FX, FY, FA, FB, FC, FOPA, FC0, FC1, FC2, NFLD };  // FC0..2: the Gaussian's colour (CDIM == 3 only)
    __shared__ float s_g[NFLD][64];
    __shared__ uint32_t s_id[64];          // FRAME: Gaussian id; else index of the pair in the sorted arrays
    // [row][lane] partial sums of the current Gaussian, 16 rows at a time (rows padded to 65).  rgb colours have only 10
    // rows: 2.6 instead of 4.2 KiB, 7.6 KiB of LDS per wave in all -- five resident waves per SIMD instead of four
    // (same-box A/B, round 3: 0.377 -> 0.359 ms at cfg2, 0.547 -> 0.515 ms at 2.4 M Gaussians)
    __shared__ float s_red[(NROW < 16 ? NROW : 16) * 65];
    __shared__ float s_part[NROW][4];      // quarter-row sums of the first reduction level
    // geometry sums per Gaussian (Sx Sy Sxx Sxy Syy Sq).  rgb frame rows: also the opacity and the three colour sums --
    // the complete 10-float row is put together here and leaves as ONE aligned 64-byte line at the end of the bucket
    // (four lanes x 16 bytes).  Round 3 stored floats 6..9 of a row from the Gaussian loop and the geometry part after
    // it: two or three partial 32-byte sectors per 48-byte row plus a flag byte = 134 bytes of HBM writes per row.
    constexpr bool STAGED = FRAME && CDIM == 3;
    constexpr int TOTW = STAGED ? 10 : 8;
    __shared__ float s_tot[64][TOTW];
    __shared__ uint32_t s_slot[64];        // FRAME: emission slot of Gaussian i's gradient row (GS_NO_SLOT: no row)
    constexpr uint32_t GS_NO_SLOT = 0xffffffffu;
    auto lds_order = [] {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    };
    const int lane = threadIdx.x & 63;
    // The work list is walked with the grid as the stride (round 5): the launch need not cover the list's CAPACITY
    // (max_pairs / 64 + T buckets: 128 k waves at 2.4 M Gaussians, of which 31 k have a bucket -- the empty ones cost the
    // dispatcher 22 us of the kernel's 455, kernel trace profiles/r05_d_*), any grid is correct, and a grid beyond the
    // list's length gives every wave at most one bucket, as before (launch_bwd picks it).
    const uint32_t n_tiles = (uint32_t)(G.ntx * G.nty), n_work = I.bucket_offsets[n_tiles];
    for (uint32_t kb = blockIdx.x; kb < n_work; kb += gridDim.x) {
    const uint4 info = I.bucket_info[kb];
    const uint32_t tile = info.x, base = info.y, r = info.z, start = info.w;
    if (base < I.bucket_first * GS_BUCKET) continue;  // (uniform) a bucket the per-tile SH kernel has taken: launch_bwd
    const uint32_t tx = tile % (uint32_t)G.ntx, ty = tile / (uint32_t)G.ntx;

    // ---- this lane's Gaussian (lanes >= r re-read the bucket's last one and are zeroed below)
    GaussianRec g;
    const uint32_t jl = start + base + ((uint32_t)lane < r ? (uint32_t)lane : r - 1);
    const uint32_t gid = raster_load<FRAME>(S, jl, g);
    const uint32_t id_x = tx * 16 + (lane & 15), id_y0 = ty * 16 + (lane >> 4);
    const float px = raster_pixel_coord(id_x, G.padW, G.focal_x);
    const float4 *ck = I.ckpt + raster_ckpt_slot(start, tile, base / GS_BUCKET) * 256;
    float4 c[4];
    float f[4][3], gr[4][3];
    bool inside[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        c[k] = ck[64 * k + lane];
        const uint32_t id_y = id_y0 + 4 * k;
        const float *cf = I.c_final + ((size_t)id_y * G.padW + id_x) * 3;
        const int ox = FRAME ? (int)id_x - G.crop_left : (int)id_x, oy = FRAME ? (int)id_y - G.crop_top : (int)id_y;
        inside[k] = !FRAME || (ox >= 0 && ox < G.width && oy >= 0 && oy < G.height);
        const float *gp = I.grad + ((size_t)(inside[k] ? oy : 0) * (FRAME ? G.width : G.padW) + (inside[k] ? ox : 0)) * 3;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            f[k][e] = cf[e];
            gr[k][e] = gp[e];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)  // keep every load in front of the first use (see raster_backward_pixel_kernel)
        asm volatile("" ::"v"(gr[k][0]), "v"(gr[k][1]), "v"(gr[k][2]), "v"(f[k][0]), "v"(f[k][1]), "v"(f[k][2]),
                     "v"(c[k].x), "v"(c[k].y), "v"(c[k].z), "v"(c[k].w));
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 3; ++e)
            gr[k][e] = (inside[k] && (!FRAME || (f[k][e] >= 0.f && f[k][e] <= 1.f))) ? gr[k][e] : 0.f;
    float cA = 0, cB = 0, cC = 0;
    {
        const bool valid = (uint32_t)lane < r;
        raster_conic(g, cA, cB, cC);
        s_g[FX][lane] = g.x;
        s_g[FY][lane] = g.y;
        s_g[FA][lane] = cA;
        s_g[FB][lane] = cB;
        s_g[FC][lane] = cC;
        s_g[FOPA][lane] = valid ? g.opa : 0.f;  // opacity 0 => alpha 0: padded entries contribute exact zeros
        if (CDIM == 3) {
            float r0, r1, r2;
            raster_load_rgb<FRAME>(S, jl, gid, r0, r1, r2);
            s_g[FC0][lane] = r0;
            s_g[FC1][lane] = r1;
            s_g[FC2][lane] = r2;
        }
        s_id[lane] = FRAME ? gid : jl;
        uint32_t myslot = GS_NO_SLOT;
        if (valid && FRAME) {
            const uint4 rc = O.rects[gid];
            const uint32_t y0 = rc.x & 0xffff, x0 = rc.y & 0xffff, x1 = rc.y >> 16;
            const uint64_t slot = (uint64_t)O.pair_offsets[gid] + (ty - y0) * (x1 - x0) + (tx - x0);
            // (no flag per row: the reader derives "written" from the tile's stop key, gs_frame_layout.h; SH rows: every
            // float of the row is written below -- coefficients, geometry, padding)
            if (slot < O.max_pairs) myslot = (uint32_t)slot;  // (max_pairs < 2^30: gs_frame validate)
        }
        s_slot[lane] = myslot;
    }
    // pixel pairs: h = 0: rows y0, y0 + 4 (k = 0, 1); h = 1: rows y0 + 8, y0 + 12 (k = 2, 3)
    f2 py2[2], T[2], rho[2], g0[2], g1[2], g2[2], SHB[2][NB];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if constexpr (CDIM > 3) {
            float sa[NB], sb[NB];
            raster_pixel_sh<NB>(id_x, id_y0 + 8 * h, G, sa);
            raster_pixel_sh<NB>(id_x, id_y0 + 8 * h + 4, G, sb);
            // pre-scaled by -log2(e) (raster_common.h): the colour's exponent needs no multiplication, and the same
            // registers serve the coefficient gradients, whose factor c (1 - c) is scaled by -ln 2 instead (below)
            // (not at degree 3: there the kernel sits at its register limit and the change costs four more spilled
            // registers -- measured 2.22 against 2.12 ms; degree 2: 1.60 against 1.62 ms)
            constexpr float KS = PRE ? -GS_LOG2E : 1.0f;
#pragma unroll
            for (int k = 0; k < NB; ++k) SHB[h][k] = f2{KS * sa[k], KS * sb[k]};
        }
        py2[h] = f2{raster_pixel_coord(id_y0 + 8 * h, G.padH, G.focal_y),
                    raster_pixel_coord(id_y0 + 8 * h + 4, G.padH, G.focal_y)};
        float Tk[2], rk[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int k = 2 * h + e;
            // the tile's first bucket starts from the empty pixel (T, C) = (1, 0), which the forward does not store
            const float4 ci = base == 0 ? make_float4(1.f, 0.f, 0.f, 0.f) : c[k];
            Tk[e] = ci.x;
            rk[e] = gr[k][0] * (f[k][0] - ci.y) + gr[k][1] * (f[k][1] - ci.z) + gr[k][2] * (f[k][2] - ci.w);
        }
        T[h] = f2{Tk[0], Tk[1]};
        rho[h] = f2{rk[0], rk[1]};
        g0[h] = f2{gr[2 * h][0], gr[2 * h + 1][0]};
        g1[h] = f2{gr[2 * h][1], gr[2 * h + 1][1]};
        g2[h] = f2{gr[2 * h][2], gr[2 * h + 1][2]};
    }
    lds_order();

    auto splat = [](float v) { return f2{v, v}; };
    auto pk_fma = [](f2 a, f2 b, f2 cc) { return __builtin_elementwise_fma(a, b, cc); };
    // first reduction level: in round rd this lane adds quarter `lane / 16` (16 consecutive entries) of row
    // 16 rd + lane % 16 -- with rows padded to 65 floats the 64 lanes of a read hit every LDS bank exactly twice
    constexpr int NROUND = (NROW + 15) / 16;
    const uint32_t red_row = (uint32_t)lane & 15, red_part = (uint32_t)lane >> 4;

    // the coefficients of Gaussian i sit at a wave-uniform address (one cache line broadcast to the wave); those of
    // Gaussian i + 1 are requested as soon as the forward evaluation of Gaussian i has consumed the registers, so
    // that the round trip hides behind the gradient half of the iteration
    float co[CDIM];
    auto load_coef = [&](uint32_t i) {
        const uint32_t id = __builtin_amdgcn_readfirstlane(s_id[i]);
        if constexpr (CDIM > 3) {
            const float *cf = (FRAME ? S.sh : S.rgb) + (size_t)id * CDIM;
#pragma unroll
            for (int k = 0; k < CDIM; ++k) co[k] = cf[k];
        }
        return id;
    };
    uint32_t id_next = load_coef(0);
#ifndef GS_BWD_PAIR_SKIP48
#define GS_BWD_PAIR_SKIP48 0  // A/B switch: the finished-half-tile skip for degree 3 as well (round 3: 18 spills at 3 waves)
#endif
    constexpr bool PAIR_SKIP = GS_BWD_PAIR_SKIP && (CDIM == 27 || (GS_BWD_PAIR_SKIP48 && CDIM == 48));
    bool pair_live[2] = {true, true};
    (void)pair_live;
    for (uint32_t i = 0; i < r; ++i) {
        const float gx = s_g[FX][i], gy = s_g[FY][i], uA = s_g[FA][i], uB = s_g[FB][i], uC = s_g[FC][i];
        const float opa = s_g[FOPA][i];
        const uint32_t id_i = id_next;
        const float dx = px - gx;
        const float bdx = uB * dx, adx2 = uA * dx * dx;
        f2 S1 = {0.f, 0.f}, Sy = S1, Syy = S1, Sq = S1, Sopa = S1;
        f2 D[2][3];
        // A pixel pair (= a half of the tile: rows 8 h .. 8 h + 7) whose 128 pixels are all finished adds exact zeros to
        // every sum below and its transmittance no longer changes: it is left out (wave-uniform branch, re-evaluated
        // every 8 Gaussians; the transmittance only falls, so a finished pair stays finished).  5 % of the pair
        // evaluations of the 2.4 M scene (oracle statistics).  Degree-2 SH only -- same-box A/B at 2.4 M
        // Gaussians: 1.40 -> 1.34 ms; degree 3 is at its register limit and loses (2.13 -> 2.42 ms: 18 spills), rgb
        // colours have too little work per pair to pay for the branches (0.63 = 0.63 ms, cfg2 +4 %).
        if constexpr (PAIR_SKIP) {
            if ((i & 7u) == 0) {
#pragma unroll
                for (int h = 0; h < 2; ++h) pair_live[h] = __ballot(T[h].x > GS_T_STOP || T[h].y > GS_T_STOP) != 0ull;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if constexpr (PAIR_SKIP) {
                if (!pair_live[h]) {  // uniform
                    D[h][0] = D[h][1] = D[h][2] = f2{0.f, 0.f};
                    continue;
                }
            }
            const f2 dy = py2[h] - splat(gy);
            const f2 q = pk_fma(pk_fma(splat(uC), dy, splat(-bdx)), dy, splat(adx2));
            // EXACT: a double-precision exp of the float argument, as the reference's backward does for fast = 0
            // (gaussian.cu:596-603)
            const f2 Gv = EXACT ? f2{(float)exp(-(double)(q.x * GS_LN2)), (float)exp(-(double)(q.y * GS_LN2))}
                                : f2{gs_exp2(-q.x), gs_exp2(-q.y)};
            // colours of the pixel pair
            f2 c0, c1, c2;
            if constexpr (CDIM > 3) {
                f2 v0 = {0.f, 0.f}, v1 = v0, v2 = v0;
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    v0 = pk_fma(SHB[h][k], splat(co[k]), v0);
                    v1 = pk_fma(SHB[h][k], splat(co[NB + k]), v1);
                    v2 = pk_fma(SHB[h][k], splat(co[2 * NB + k]), v2);
                }
                auto sg = [](float v) { return PRE ? gs_rcp(1.0f + gs_exp2(v)) : gs_rcp(1.0f + __expf(-v)); };
                c0 = f2{sg(v0.x), sg(v0.y)};
                c1 = f2{sg(v1.x), sg(v1.y)};
                c2 = f2{sg(v2.x), sg(v2.y)};
            } else {  // one colour per Gaussian
                c0 = splat(s_g[FC0][i]);
                c1 = splat(s_g[FC1][i]);
                c2 = splat(s_g[FC2][i]);
            }
            // 1 / (1 - alpha + 1e-7) as one fma + rcp (see raster_backward_pixel_kernel)
            const f2 den = pk_fma(-Gv, splat(opa), splat(1.00000011920928955f));
            const f2 rc = {gs_rcp(den.x), gs_rcp(den.y)};
            const bool l0 = T[h].x > GS_T_STOP, l1 = T[h].y > GS_T_STOP;
            const f2 araw = Gv * splat(opa);
            const f2 alpha = {l0 ? araw.x : 0.f, l1 ? araw.y : 0.f};
            const f2 w = alpha * T[h];
            const f2 gc = pk_fma(g2[h], c2, pk_fma(g1[h], c1, g0[h] * c0));
            rho[h] = pk_fma(-w, gc, rho[h]);
            f2 d_alpha = pk_fma(T[h], gc, -(rho[h] * rc));
            d_alpha = f2{l0 ? d_alpha.x : 0.f, l1 ? d_alpha.y : 0.f};
            if constexpr (CDIM > 3) {
                // c (1 - c) [x -ln 2 with the pre-scaled basis: D sh' = D sh] as c (K - K c): one packed FMA either way
                constexpr float K = PRE ? -GS_LN2 : 1.0f;
                const f2 kk = {K, K}, nk = {-K, -K};
                D[h][0] = g0[h] * w * (c0 * pk_fma(c0, nk, kk));
                D[h][1] = g1[h] * w * (c1 * pk_fma(c1, nk, kk));
                D[h][2] = g2[h] * w * (c2 * pk_fma(c2, nk, kk));
            } else {  // dL/dcolour_c = sum dL/dC_c w (the sigmoid's derivative is applied per Gaussian, later)
                D[h][0] = g0[h] * w;
                D[h][1] = g1[h] * w;
                D[h][2] = g2[h] * w;
            }
            Sopa = pk_fma(d_alpha, Gv, Sopa);
            const f2 s = d_alpha * alpha;
            const f2 sdy = s * dy;
            S1 += s;
            Sy += sdy;
            Syy = pk_fma(sdy, dy, Syy);
            Sq = pk_fma(s, q, Sq);
            T[h] = T[h] - w;
        }
        id_next = load_coef(i + 1 < r ? i + 1 : i);
        // row m of the [NROW][64] array of partial sums: 0..6 geometry / opacity, 7 + ch NB + k colour (coefficient)
        const float s1 = S1.x + S1.y, sy_ = Sy.x + Sy.y;
        const float sx_ = s1 * dx;
        auto row_value = [&](int m) -> float {
            if (m == 0) return sx_;                // Sx
            if (m == 1) return sy_;                // Sy
            if (m == 2) return sx_ * dx;           // Sxx
            if (m == 3) return sy_ * dx;           // Sxy
            if (m == 4) return Syy.x + Syy.y;      // Syy
            if (m == 5) return Sq.x + Sq.y;        // Sq
            if (m == 6) return Sopa.x + Sopa.y;
            const int ch = (m - 7) / NB, k = (m - 7) % NB;
            if constexpr (CDIM > 3) {
#if GS_BWD_SH_SCALAR_ROWS
                // four products of the lane's four pixels as ONE chain of scalar FMAs: a packed multiply + packed FMA +
                // the add of the two halves is three instructions but 5.2 ns of issue (packed fp32 has no rate
                // advantage on this chip: tools/ubench/pk_rate.hip), the chain is four instructions and 4.2 ns
                return fmaf(D[1][ch].y, SHB[1][k].y,
                            fmaf(D[1][ch].x, SHB[1][k].x, fmaf(D[0][ch].y, SHB[0][k].y, D[0][ch].x * SHB[0][k].x)));
#else
                const f2 p = pk_fma(D[1][ch], SHB[1][k], D[0][ch] * SHB[0][k]);
                return p.x + p.y;
#endif
            } else {
                const f2 p = D[0][ch] + D[1][ch];
                return p.x + p.y;
            }
        };
        // 16 rows at a time through a [16][65] window: the whole [NROW][65] array (8.8 / 14.3 KiB per wave) capped the
        // kernel at 2-3 waves per SIMD
#pragma unroll
        for (int rd = 0; rd < NROUND; ++rd) {
#pragma unroll
            for (int m = 16 * rd; m < 16 * rd + 16 && m < NROW; ++m) s_red[(m - 16 * rd) * 65 + lane] = row_value(m);
            lds_order();
            const uint32_t row = 16 * rd + red_row;
            if (16 * (rd + 1) <= NROW || row < (uint32_t)NROW) {
                const float *src = s_red + red_row * 65 + red_part * 16;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                for (int j = 0; j < 16; j += 4) {
                    a0 += src[j];
                    a1 += src[j + 1];
                    a2 += src[j + 2];
                    a3 += src[j + 3];
                }
                s_part[row][red_part] = (a0 + a1) + (a2 + a3);
            }
            lds_order();
        }
        if (lane < NROW) {
            const float t = (s_part[lane][0] + s_part[lane][1]) + (s_part[lane][2] + s_part[lane][3]);
            if (lane < 6 || STAGED) {
                s_tot[i][lane] = t;
            } else if (FRAME) {
                const uint32_t sl = s_slot[i];  // sum dL/dalpha G (opacity), then the coefficient sums (row layout: gs_frame_layout.h)
                if (sl != GS_NO_SLOT)
                    O.rows[(size_t)sl * gs_row_floats(CDIM) + (lane == 6 ? gs_row_geo(CDIM, 6) : gs_row_col(CDIM, lane - 7))] = t;
            } else {
                const size_t j = id_i;
                if (lane == 6)
                    O.grad_opa[j] = t;
                else
                    O.grad_rgb[j * CDIM + (lane - 7)] = t;
            }
        }
        lds_order();
    }
    if ((uint32_t)lane < r) {  // one lane per Gaussian finishes the geometry algebra
        const float *t = s_tot[lane];
        const float Sx = t[0], Sy = t[1], Sxx = t[2], Sxy = t[3], Syy = t[4], Sq = t[5];
        const float a = g.a, b = g.b, cc = g.c, d = g.d;
        const float iPn = 1.0f / (2.0f * raster_det(a, b, cc, d) + 1e-14f);
        const float Su = Sq * GS_LN2;
        const float ogx = GS_LN2 * (2.0f * cA * Sx - cB * Sy), ogy = GS_LN2 * (2.0f * cC * Sy - cB * Sx);
        const float ga = iPn * (-Syy + 2.0f * d * Su), gb = iPn * (Sxy - 2.0f * cc * Su);
        const float gcc = iPn * (Sxy - 2.0f * b * Su), gd = iPn * (-Sxx + 2.0f * a * Su);
        if (STAGED) {  // the finished geometry part replaces the sums it was computed from (this lane's own entries)
            float *t2 = s_tot[lane];
            t2[0] = ogx; t2[1] = ogy; t2[2] = ga; t2[3] = gb; t2[4] = gcc; t2[5] = gd;
        } else if (FRAME) {
            const uint32_t sl = s_slot[lane];
            if (sl != GS_NO_SLOT) {
                constexpr int RW = gs_row_floats(CDIM);
                float *row = O.rows + (size_t)sl * RW;
                reinterpret_cast<float4 *>(row + gs_row_geo(CDIM, 0))[0] = make_float4(ogx, ogy, ga, gb);
                reinterpret_cast<float2 *>(row + gs_row_geo(CDIM, 4))[0] = make_float2(gcc, gd);
                // the floats of the row that are neither a sum nor the geometry: zero (the reader adds whole rows)
#pragma unroll
                for (int m = 0; m < RW; ++m) {
                    bool used = false;
#pragma unroll
                    for (int e = 0; e < 7; ++e) used = used || m == gs_row_geo(CDIM, e);
#pragma unroll
                    for (int e = 0; e < CDIM; ++e) used = used || m == gs_row_col(CDIM, e);
                    if (!used) row[m] = 0.f;
                }
            }
        } else {
            const size_t j = (size_t)start + base + lane;
            O.grad_pos[j * 3 + 0] = ogx;
            O.grad_pos[j * 3 + 1] = ogy;
            reinterpret_cast<float4 *>(O.grad_cov)[j] = make_float4(ga, gb, gcc, gd);
        }
    }
    if constexpr (STAGED) {
        // rows leave as whole 64-byte lines: lane l stores quarter l % 4 of row 16 it + l / 4 -- sixteen rows per store
        // instruction, every line written completely by one instruction (floats 10..15 are zero padding)
        lds_order();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const uint32_t j = 16u * it + ((uint32_t)lane >> 2), qd = (uint32_t)lane & 3u;
            if (j >= r) break;  // (lanes of one row leave together; rows beyond r do not exist)
            const uint32_t sl = s_slot[j];
            if (sl == GS_NO_SLOT) continue;
            const float *t = s_tot[j];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qd == 0) v = make_float4(t[0], t[1], t[2], t[3]);
            else if (qd == 1) v = make_float4(t[4], t[5], t[6], t[7]);
            else if (qd == 2) v = make_float4(t[8], t[9], 0.f, 0.f);
            reinterpret_cast<float4 *>(O.rows + (size_t)sl * gs_row_floats(3))[qd] = v;
        }
    }
    lds_order();  // (the next bucket of this wave reuses the LDS arrays)
    }
}

// ---------------------------------------------------------------------------------------------
// SH, frame path: the two contractions of the SH backward on the matrix pipe (raster_backward_mfma_sh_kernel).
//
// Per bucket of 64 Gaussians and tile of 256 pixels the SH backward holds two true matrix products
//   logit[(g, ch)][p]   = sum_k  coef[g][ch][k] sh_k(p)          (64 x 3 rows, K = 9 | 16, 256 columns)
//   dcoef[(g, ch)][k]   = sum_p  D[g][ch][p]   sh_k(p)           (K = 256 pixels: the cross-pixel reduction itself)
// which the pixel-parallel kernel above evaluates with 2 x 27 | 48 FMAs per (pixel, Gaussian) on the VALU plus, for the
// second one, a 64 : 1 reduction of 27 | 48 rows through LDS per Gaussian -- together ~3/4 of its ~360 VALU
// instructions per Gaussian step.  An fp32 MFMA (v_mfma_f32_16x16x4_f32: exact fp32, a k-ordered fmaf chain) has the
// VALU's FMA rate but runs BESIDE the VALU and contracts across lanes for free.  The layouts that make both products
// MFMA-shaped at once: a wave works on 16 Gaussians x 16 pixels (one pixel row of the tile) per step,
//   lane l  <->  Gaussian g' = l & 15 of the current group of 16,  pixel quad j = l >> 4  (pixels x = 4 j + i, i = 0..3)
// * colour logits:   D[row = pixel x][col = g'] = sum_k A[x][k] B[k][g'],  A = sh_k(x, y) from an LDS table (lane: x = l & 15,
//   k = 4 kk + (l >> 4)), B = coef[g'][ch][k] (held in registers for the 16 pixel rows).  The result lands exactly where
//   the per-pixel arithmetic wants it: lane (g', j) gets the logits of its four pixels 4 j + reg.
// * coefficient sums: D[row = k][col = g'] += sum_x A'[k][x] B'[x][g'],  B' = D[g'][ch][pixel 4 j + i] -- register i of
//   lane (g', j), as computed --, A' = sh_k(x, y) from the same table (lane: k = l & 15, x = 4 (l >> 4) + i).  The MFMA's
//   K index contracts over the pixel quads, its accumulator over i and over the 16 pixel rows: after 16 steps lane (g', j)
//   holds dL/dcoef[g'][ch][4 j + reg] complete -- no reduction of the coefficient rows through LDS at all.
// What stays on the VALU per (pixel, Gaussian): the Gaussian's value (one v_exp), three sigmoids, the transmittance and
// rho recursions -- over the 16 Gaussians of a group they are a prefix product / prefix sum across the 16 lanes of a DPP
// row (row_shr 1, 2, 4, 8) -- and the seven geometry / opacity sums, which accumulate over the pixel rows in registers
// (dx of a lane's four pixel columns is constant over the rows) and are reduced 4 : 1 once per group.
// A workgroup = one tile (or half of its buckets: GS_BWD_MFMA_SPLIT), W waves; wave w takes buckets w, w + W, ...: the SH table (pre-scaled by -log2 e as in the
// kernels above), dL/dC and the final colours are staged once per tile; per wave only the pixels' (T, rho) live in LDS
// (read as broadcast float4, written back by the lanes of Gaussian 15 after every pixel row).
// Transmittance: T_before(g') = T_in prod_{h < g'} max(1 - alpha_h, 0) with UNMASKED alphas; a pixel is live while that
// is > 1e-4, exactly the reference's test (gaussian.cu:906) up to the rounding of a product tree against a chain; behind
// the stop every contribution is masked to zero and the unmasked product only ever falls, so it never revives a pixel.
#ifndef GS_BWD_SH_MFMA
#define GS_BWD_SH_MFMA 2  // 0: never, 1: degree 3 (48 coefficients) only, 2: degree 2 as well (A/B switch, tools/ab_variants.py)
#endif
#ifndef GS_BWD_MFMA_WAVES
#define GS_BWD_MFMA_WAVES 0  // waves per workgroup: 0 = by the size of the tile grid (below), 2 / 4 = fixed (A/B switch)
#endif
// Waves per workgroup, workgroups per tile.  A tile of the 376 k-Gaussian scene has two or three buckets, one of the 2.4 M
// scene four or five, a dense one dozens.  Four waves per tile left one or two of their register / LDS slots idle for the
// workgroup's whole life on the short tiles (376 k Gaussians, degree 2: 1.12 ms against 1.00 ms for the pixel-parallel
// kernel); TWO waves: 1.01 ms there, and no worse at 2.4 M Gaussians (1.44 = 1.44 ms at degree 2, 1.46 against 1.50 ms at
// degree 3; the kernel's first version had lost 5 % with two).  Only a small tile grid with long lists wants the four (192
// tiles, 900 pairs each: 0.120 against 0.187 ms): fewer workgroups than the device has slots.  Hence two waves from 1,024
// tiles on, four below.  Measured and dropped (profiles/r04_zz_*): TWO workgroups of two waves per tile (GS_BWD_MFMA_SPLIT =
// 2: workgroup h takes buckets 2 h + wave, + 4, ...; the second leaves at once on short tiles): 1.61 / 1.67 ms.
#ifndef GS_BWD_MFMA_SPLIT
#define GS_BWD_MFMA_SPLIT 1
#endif
#ifndef GS_BWD_MFMA_WPE
#define GS_BWD_MFMA_WPE 3
#endif

template <int CTRL>
__device__ __forceinline__ float gs_dpp(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                                 CTRL, 0xf, 0xf, false));
}
// inclusive product / sum over the 16 lanes of a DPP row (lanes whose source falls outside the row keep the identity)
__device__ __forceinline__ float gs_row_scan_mul(float v) {
    v *= gs_dpp<0x111>(1.0f, v);
    v *= gs_dpp<0x112>(1.0f, v);
    v *= gs_dpp<0x114>(1.0f, v);
    v *= gs_dpp<0x118>(1.0f, v);
    return v;
}
__device__ __forceinline__ float gs_row_scan_add(float v) {
    v += gs_dpp<0x111>(0.0f, v);
    v += gs_dpp<0x112>(0.0f, v);
    v += gs_dpp<0x114>(0.0f, v);
    v += gs_dpp<0x118>(0.0f, v);
    return v;
}

// The same scans for the lane's four pixels at once, every step ONE instruction per pixel (v_mul_f32_dpp v, v, v row_shr:n:
// lanes whose source falls outside the row are disabled by the DPP bound check and keep v -- the identity for free;
// through the builtin the compiler emits v_mov 1.0 + v_mov_dpp + v_mul for a product step).  The four chains are
// interleaved so that the write of a register and its DPP read in the next step are three instructions apart (the
// hardware wants two wait states there, and the compiler's hazard pass does not look into inline assembly: hence also
// the s_nop in front of the first and behind the last instruction of the block).
#ifndef GS_BWD_MFMA_ASM_SCAN
#define GS_BWD_MFMA_ASM_SCAN 1
#endif
#define GS_SCAN4(OP)                                                                                                   \
    asm volatile("s_nop 1\n\t" OP " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" OP                                          \
                    " %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" OP                                          \
                    " %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" OP                                          \
                    " %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" OP                                          \
                    " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t" OP                                          \
                    " %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t" OP                                          \
                    " %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n\t" OP                                          \
                    " %3, %3, %3 row_shr:2 row_mask:0xf bank_mask:0xf\n\t" OP                                          \
                    " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t" OP                                          \
                    " %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t" OP                                          \
                    " %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n\t" OP                                          \
                    " %3, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xf\n\t" OP                                          \
                    " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t" OP                                          \
                    " %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t" OP                                          \
                    " %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf\n\t" OP                                          \
                    " %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1"                                        \
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]))
__device__ __forceinline__ void gs_row_scan_mul4(float v[4]) {
#if GS_BWD_MFMA_ASM_SCAN
    GS_SCAN4("v_mul_f32_dpp");
#else
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = gs_row_scan_mul(v[i]);
#endif
}
__device__ __forceinline__ void gs_row_scan_add4(float v[4]) {
#if GS_BWD_MFMA_ASM_SCAN
    GS_SCAN4("v_add_f32_dpp");
#else
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = gs_row_scan_add(v[i]);
#endif
}

// Inclusive prefix sum over the 16 lanes of a DPP row for the lane's four pixels, OUT of place: the first step's
// bound_ctrl:0 reads 0 where a lane has no source -- the sum's identity --, so `in` survives (round 5: the unscanned w gc
// gives s = w gc - rho beta one instruction cheaper than dL/dalpha alpha).
__device__ __forceinline__ void gs_row_scan_add4_oop(float out[4], const float in[4]) {
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_add_f32_dpp %1, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_add_f32_dpp %2, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_add_f32_dpp %3, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %3, %3, %3 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %3, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf"
                 : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3])
                 : "v"(in[0]), "v"(in[1]), "v"(in[2]), "v"(in[3]));
    // (no s_nop behind the block: the hazard is VALU write -> DPP READ; what follows reads the results with plain VALU
    // instructions, and the kernels that use this helper contain no compiler-generated DPP instruction)
}
// T in front of the lane's Gaussian = T_in x the inclusive product scan of the lane to the left (row_shr:1); the row's first
// lane has no source: the DPP bound check disables it and it keeps T_in.  t: T_in on entry, the result on exit.
__device__ __forceinline__ void gs_row_excl_mul4(float t[4], const float scanned[4]) {
    asm volatile("s_nop 1\n\t"
                 "v_mul_f32_dpp %0, %4, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %1, %5, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %2, %6, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %3, %7, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3])
                 : "v"(scanned[0]), "v"(scanned[1]), "v"(scanned[2]), "v"(scanned[3]));
}
// Both at once (one asm block, no s_nop between the two: a register written by the scan's last step is read four
// instructions later): p <- inclusive product scan of p; t <- t x (scanned p of the lane to the left), lane 0 keeps t.
__device__ __forceinline__ void gs_row_scan_mul4_excl(float p[4], float t[4]) {
    asm volatile("s_nop 1\n\t"
                 "v_mul_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %4, %4, %4 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %5, %5, %5 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %6, %6, %6 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %7, %7, %7 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %4, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %5, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %6, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %7, %7, %7 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %4, %4, %4 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %5, %5, %5 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %6, %6, %6 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %7, %7, %7 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %0, %4, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %1, %5, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %2, %6, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %3, %7, %3 row_shr:1 row_mask:0xf bank_mask:0xf"
                 : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]));
}
// Round 5 instruction diet of the SH row loop (first used by raster_backward_rows_kernel below): the opacity in the
// exponent's constant term (no G x opacity product), T in front of the Gaussian as one DPP multiply (the builtin costs
// v_mov 1.0 + v_mov_dpp + v_mul), s = w gc - rho beta.  A/B switch (tools/ab_variants.py).
#ifndef GS_BWD_MFMA_DIET
#define GS_BWD_MFMA_DIET 1
#endif

#ifndef GS_BWD_MFMA_PF
#define GS_BWD_MFMA_PF 1
#endif
#ifndef GS_BWD_MFMA_PK
// 1: the per-pixel algebra of a row step between the two matrix products on PAIRS of the lane's four pixels (packed fp32, as
// raster_backward_rows_kernel's GS_BWD_ROWS_PK); the scans, the transcendentals and the stop-point selects stay per pixel
#define GS_BWD_MFMA_PK 1
#endif
static_assert(!GS_BWD_MFMA_PK || GS_BWD_MFMA_DIET, "the packed row step is the diet's");
#ifndef GS_BWD_MFMA_ORDER
#define GS_BWD_MFMA_ORDER 1
#endif
#ifndef GS_BWD_MFMA_ROW_SKIP
#define GS_BWD_MFMA_ROW_SKIP 1
#endif
#ifndef GS_BWD_MFMA_TILES
#define GS_BWD_MFMA_TILES 1  // tiles per workgroup (see the kernel's header: the waves of a workgroup share their buckets)
#endif
#ifndef GS_BWD_MFMA_CHUNK
#define GS_BWD_MFMA_CHUNK GS_MFMA_ITEM_BUCKETS  // buckets per work item (gs_frame_layout.h; 0: one workgroup per tile, round 4)
#endif
static_assert(GS_BWD_MFMA_CHUNK == 0 || GS_BWD_MFMA_CHUNK >= GS_MFMA_ITEM_BUCKETS, "the item list is sized for GS_MFMA_ITEM_BUCKETS");
constexpr bool GS_MFMA_ITEMS = GS_BWD_MFMA_CHUNK > 0 && GS_BWD_MFMA_TILES == 1 && GS_BWD_MFMA_SPLIT == 1;
// workgroups of the matrix-pipe launch: one per work item up to 2 T (the kernel walks the items with the grid as the stride;
// the executed-row counters have GS_BWD_EXEC_SLOTS x T = 8 T slots for grid x waves)
static inline unsigned gs_bwd_mfma_grid(int n_tiles, int64_t max_buckets) {
    if (!GS_MFMA_ITEMS)
        return (unsigned)(gs_div_up(n_tiles, GS_BWD_MFMA_TILES) * (GS_BWD_MFMA_TILES == 1 ? GS_BWD_MFMA_SPLIT : 1));
    const int64_t cap_items = n_tiles + max_buckets / (GS_BWD_MFMA_CHUNK ? GS_BWD_MFMA_CHUNK : 1);
    return (unsigned)(cap_items < 2ll * n_tiles ? cap_items : 2ll * n_tiles);
}
template <int CDIM, int W, int TPW>
__global__ void __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(GS_BWD_MFMA_WPE)))
raster_backward_mfma_sh_kernel(RasterSrc S, RasterGeom G, BwdIn I, BwdOut O) {
    static_assert(CDIM == 27 || CDIM == 48, "SH colours only");
    constexpr int NB = CDIM / 3;          // basis functions per channel: 9 | 16
    constexpr int KQ = NB / 4;            // k blocks of four of the colour product on the matrix pipe: 2 | 4
    // degree 2 has nine basis functions: the ninth would cost a third, three-quarters empty MFMA per channel (14 ns each);
    // it enters the logits as the MFMAs' initial accumulator instead, sh'_8(pixel) coef[ch][8]: 12 multiplications
    constexpr bool TAIL = NB % 4 != 0;
    static_assert(!TAIL || NB == 4 * KQ + 1, "one basis function beside the blocks of four");
    constexpr int P = 17;                 // floats per pixel in the SH table (16 + 1: both operand patterns nearly conflict-free)
    constexpr int RW = gs_row_floats(CDIM);
    constexpr uint32_t GS_NO_SLOT = 0xffffffffu;
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ float s_sh[TPW][256 * P];                                 // [pixel 16 y + x][k], entries k >= NB are zero
    __shared__ __attribute__((aligned(16))) float s_gr[TPW][3][256];    // dL/dC (masked: crop, clamp)
    __shared__ float s_py[TPW][16];
    __shared__ __attribute__((aligned(16))) float s_sh8[TAIL ? TPW : 1][TAIL ? 256 : 4];  // sh'_(NB - 1) of the pixel
    __shared__ __attribute__((aligned(16))) float s_T[W][256];          // per wave: transmittance / rho in front of the current group
    __shared__ __attribute__((aligned(16))) float s_rho[W][256];
    auto lds_order = [] {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Work items (round 5): (tile, first bucket, buckets) with at most GS_BWD_MFMA_CHUNK buckets, in the forward's dispatch
    // order of the tiles (heavy tiles first), built by mfma_items_kernel underneath the caller's loss.  One workgroup per TILE
    // (round 4) walks a tile's buckets W at a time, so the kernel lasts as long as its longest tile: on the 2.4 M scene
    // (3.8 buckets per tile on average, 14 at most) that is harmless, in a densifying run it is not -- a few tiles with 30 - 100
    // buckets kept one workgroup busy for milliseconds while the device idled (kernel trace of tools/soak.py, SH degree 2:
    // 1.78 ms per call on average, 5.4 ms at worst, at 376 k - 556 k Gaussians; 1.27 ms at 2.4 M Gaussians, profiles/r05_j_*).
    // A workgroup takes items blockIdx.x, + gridDim.x, ... and rebuilds the tile's tables per item (a few microseconds
    // against >= 56 us per bucket).  Without an item list (GS_BWD_MFMA_CHUNK = 0): one item per tile / tile part, as before.
    const uint32_t n_tiles = (uint32_t)(G.ntx * G.nty);
    constexpr uint32_t SPLIT = TPW == 1 ? GS_BWD_MFMA_SPLIT : 1;
    const uint32_t n_items = I.mfma_items ? *I.mfma_n_items : gridDim.x;
    uint32_t n_exec = 0;  // (wave-uniform) pixel-row steps this wave executed: what its MFMA flops are counted from
    constexpr float KS = -GS_LOG2E;  // the table holds sh'_k = -log2(e) sh_k (raster_common.h)
    const uint32_t gq = (uint32_t)lane & 15u, jq = (uint32_t)lane >> 4;  // Gaussian of the group, pixel quad
    float *sT = s_T[wave], *sR = s_rho[wave];
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
    uint32_t tile0, bk0 = 0, part = 0;
    // the workgroup's TPW consecutive tiles: their buckets form ONE work list that the W waves share (a tile has 4.3
    // buckets on average at 2.4 M Gaussians: alone it keeps four waves busy 68 % of the time, two tiles together 85 %)
    uint32_t nbk[TPW], nproc_t[TPW], total_bk = 0;
    if (TPW == 1 && I.mfma_items) {
        const uint4 it = I.mfma_items[item];
        tile0 = it.x;
        bk0 = it.y;
        nproc_t[0] = I.tile_nproc[tile0];
        nbk[0] = total_bk = it.z;
    } else {
        const uint32_t wg = item / SPLIT;
        part = item % SPLIT;  // workgroup `part` of SPLIT of this tile
        tile0 = (TPW == 1 && I.tile_order) ? I.tile_order[wg] : wg * TPW;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            nproc_t[t] = tile0 + t < n_tiles ? I.tile_nproc[tile0 + t] : 0;
            nbk[t] = (nproc_t[t] + GS_BUCKET - 1) / GS_BUCKET;
            if (I.bucket_cap && nbk[t] > I.bucket_cap) nbk[t] = I.bucket_cap;  // the rest: one wave per bucket (launch_bwd)
            total_bk += nbk[t];
        }
    }
    if (total_bk <= part * W) continue;  // uniform: nothing (left) of these tiles for this workgroup
    __syncthreads();  // (a wave of this workgroup may still be reading the previous item's tables)

    // ---- per tile: SH table, dL/dC, pixel-row centres
    for (int q = tid; q < 256 * TPW; q += 64 * W) {
        const int t = q >> 8, p = q & 255;
        if (nbk[t] == 0) continue;
        const uint32_t tile = tile0 + t, tx = tile % (uint32_t)G.ntx, ty = tile / (uint32_t)G.ntx;
        const uint32_t id_x = tx * 16 + (p & 15), id_y = ty * 16 + (p >> 4);
        float sh[NB];
        raster_pixel_sh<NB>(id_x, id_y, G, sh);
#pragma unroll
        for (int k = 0; k < 16; ++k) s_sh[t][p * P + k] = k < NB ? KS * sh[k < NB ? k : 0] : 0.f;
        if (TAIL) s_sh8[t][p] = KS * sh[NB - 1];
        float f[3], gr[3];
        load_pixel_inputs<true>(I, G, id_x, id_y, f, gr);
#pragma unroll
        for (int e = 0; e < 3; ++e) s_gr[t][e][p] = gr[e];
    }
    if (tid < 16 * TPW) {
        const uint32_t tile = tile0 + (tid >> 4);
        s_py[tid >> 4][tid & 15] = raster_pixel_coord((tile / (uint32_t)G.ntx) * 16 + (tid & 15), G.padH, G.focal_y);
    }
    __syncthreads();

    for (uint32_t u = part * W + wave; u < total_bk; u += W * SPLIT) {
        int t = 0;
        uint32_t b = bk0 + u;
#pragma unroll
        for (int k = 0; k + 1 < TPW; ++k)
            if (t == k && b >= nbk[k]) {
                b -= nbk[k];
                t = k + 1;
            }
        const uint32_t tile = tile0 + t, nproc = nproc_t[t];
        const uint32_t tx = tile % (uint32_t)G.ntx, ty = tile / (uint32_t)G.ntx;
        const uint32_t start = (uint32_t)I.ranges[2 * tile];
        const float *tab = s_sh[t], *gr0 = s_gr[t][0], *gr1 = s_gr[t][1], *gr2 = s_gr[t][2], *pyt = s_py[t];
        const float *tab8 = s_sh8[TAIL ? t : 0];
        const uint32_t base = b * GS_BUCKET, rem = nproc - base, r = rem < GS_BUCKET ? rem : GS_BUCKET;
        // the bucket's 64 Gaussian ids, one per lane: a group learns its ids from a lane exchange instead of a load that
        // everything else of the group's set-up would wait for
        const uint32_t id_lane = S.ids[start + base + ((uint32_t)lane < r ? (uint32_t)lane : r - 1)];
        // ---- pixel states at the bucket's boundary: the forward's checkpoint ((1, 0) in front of the tile's first bucket)
        {
            const float4 *ck = I.ckpt + raster_ckpt_slot(start, tile, b) * 256;
            float4 c[4];
            float f[4][3];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                c[k] = b == 0 ? make_float4(1.f, 0.f, 0.f, 0.f) : ck[64 * k + lane];
                // pixel (x = lane & 15, y = (lane >> 4) + 4 k) == index 64 k + lane of the tile
                const float *cf = I.c_final + ((size_t)(ty * 16 + (lane >> 4) + 4 * k) * G.padW + tx * 16 + (lane & 15)) * 3;
                f[k][0] = cf[0];
                f[k][1] = cf[1];
                f[k][2] = cf[2];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int p = 64 * k + lane;
                sT[p] = c[k].x;
                sR[p] = gr0[p] * (f[k][0] - c[k].y) + gr1[p] * (f[k][1] - c[k].z) + gr2[p] * (f[k][2] - c[k].w);
            }
        }
        lds_order();
        const uint32_t ngrp = (r + 15) / 16;
        for (uint32_t grp = 0; grp < ngrp; ++grp) {
            // ---- this lane's Gaussian (entries beyond r re-read the bucket's last one with opacity 0: alpha = 0)
            const uint32_t gi = grp * 16 + gq;
            const bool valid = gi < r;
            const uint32_t gid = (uint32_t)__shfl((int)id_lane, (int)(valid ? gi : r - 1), 64);
            GaussianRec g;
            {
                const float4 ge = S.geom[(size_t)gid * GS_REC_STRIDE], cv = S.cov4[(size_t)gid * GS_REC_STRIDE];
                g.x = ge.x;
                g.y = ge.y;
                g.opa = ge.w;
                g.a = cv.x;
                g.b = cv.y;
                g.c = cv.z;
                g.d = cv.w;
            }
            const float *cf = S.sh + (size_t)gid * CDIM;
            float cob[3][KQ];  // B operand of the colour product: coef[g'][ch][4 kk + jq]
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
#pragma unroll
                for (int kk = 0; kk < KQ; ++kk) {
                    const uint32_t k = 4 * kk + jq;
                    cob[ch][kk] = (4 * kk + 3 < NB || k < (uint32_t)NB) ? cf[ch * NB + (k < (uint32_t)NB ? k : 0)] : 0.f;
                }
            float ctail[3] = {0.f, 0.f, 0.f};
            if (TAIL) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) ctail[ch] = cf[ch * NB + NB - 1];
            }
            // the next group's coefficient rows (192 bytes = three lines per Gaussian at degree 3; 2.4 M of them do not
            // fit the Infinity Cache) and records are asked for now, one line per lane, and nothing waits for them: by the
            // time the next group sets up they sit in L2
            float warm = 0.f;
            if (grp + 1 < ngrp) {
                const uint32_t gn = (grp + 1) * 16 + gq;
                const uint32_t idn = (uint32_t)__shfl((int)id_lane, (int)(gn < r ? gn : r - 1), 64);
                warm = jq * 16 < (uint32_t)CDIM ? S.sh[(size_t)idn * CDIM + jq * 16]
                                                : reinterpret_cast<const float *>(S.geom + (size_t)idn * GS_REC_STRIDE)[0];
            }
            float cA, cB, cC;
            raster_conic(g, cA, cB, cC);
            const float lopa = fmaxf(__log2f(g.opa), -200.0f);
            uint32_t slot = GS_NO_SLOT;
            if (valid) {
                const uint4 rc = O.rects[gid];
                const uint32_t y0 = rc.x & 0xffff, x0 = rc.y & 0xffff, x1 = rc.y >> 16;
                const uint64_t sl = (uint64_t)O.pair_offsets[gid] + (ty - y0) * (x1 - x0) + (tx - x0);
                if (sl < O.max_pairs) slot = (uint32_t)sl;
            }
            // (what only the group's closing needs -- dx, the covariance, two of the conic's terms -- is recomputed /
            // re-read there instead of living in registers through the 16 pixel rows: the loop sits at the register
            // limit of three waves per SIMD)
            float bdx[4], adx2[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float dxi = raster_pixel_coord(tx * 16 + 4 * jq + i, G.padW, G.focal_x) - g.x;
                bdx[i] = cB * dxi;
                adx2[i] = cA * dxi * dxi;
#if GS_BWD_MFMA_DIET
                // q' = q - log2 sigma(opa): alpha = 2^-q'.  A padded entry: q' = 1e30 => alpha = 0 exactly, 0 x 1e30 = 0 in
                // the sum of s q'; an opacity that underflowed to 0: 2^-(q + 200) is flushed to 0
                adx2[i] = valid ? adx2[i] - lopa : 1e30f;
#endif
            }
            const float gy = g.y;
            f4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            float S1[4] = {0.f, 0.f, 0.f, 0.f}, Sy[4] = {0.f, 0.f, 0.f, 0.f}, Syy = 0.f, Sq = 0.f;
#if GS_BWD_MFMA_PK
            typedef float f2 __attribute__((ext_vector_type(2)));
            auto pfma = [](f2 x, f2 y, f2 z) { return __builtin_elementwise_fma(x, y, z); };
            f2 S1p[2] = {{0.f, 0.f}, {0.f, 0.f}}, Syp[2] = {{0.f, 0.f}, {0.f, 0.f}}, Sqp = {0.f, 0.f};
#endif
            // The LDS operands of a pixel row -- the A operands of both products, the row's pixel states and dL/dC -- are
            // requested one row ahead, behind the row's arithmetic and in front of its twelve coefficient MFMAs (registers
            // are free there, and the ~100 cycles of LDS latency pass under the MFMAs; asked for where they are used,
            // every row started and ended with an exposed wait).  GS_BWD_MFMA_PF = 0: loads where they are used.
            float a_n[KQ], tb_n[4];
            f4 Tin_n, Rin_n, G0_n, G1_n, G2_n, S8_n = {0.f, 0.f, 0.f, 0.f};
            auto row_loads = [&](int s) {
                const int prow = 16 * s;
#pragma unroll
                for (int kk = 0; kk < KQ; ++kk) a_n[kk] = tab[(prow + (int)gq) * P + 4 * kk + (int)jq];  // l & 15: pixel column
#pragma unroll
                for (int i = 0; i < 4; ++i) tb_n[i] = tab[(prow + 4 * (int)jq + i) * P + (int)gq];       // l & 15: basis function
                Tin_n = *reinterpret_cast<const f4 *>(sT + prow + 4 * jq);
                Rin_n = *reinterpret_cast<const f4 *>(sR + prow + 4 * jq);
                G0_n = *reinterpret_cast<const f4 *>(gr0 + prow + 4 * jq);
                G1_n = *reinterpret_cast<const f4 *>(gr1 + prow + 4 * jq);
                G2_n = *reinterpret_cast<const f4 *>(gr2 + prow + 4 * jq);
                if (TAIL) S8_n = *reinterpret_cast<const f4 *>(tab8 + prow + 4 * jq);
            };
            if (GS_BWD_MFMA_PF) row_loads(0);
            for (int s = 0; s < 16; ++s) {  // pixel row s of the tile
                const int prow = 16 * s;
                if (!GS_BWD_MFMA_PF) row_loads(s);
                float a[KQ], tb[4];
#pragma unroll
                for (int kk = 0; kk < KQ; ++kk) a[kk] = a_n[kk];
#pragma unroll
                for (int i = 0; i < 4; ++i) tb[i] = tb_n[i];
                const f4 Tin = Tin_n, Rin = Rin_n, G0 = G0_n, G1 = G1_n, G2 = G2_n, S8 = S8_n;
                // A pixel row whose 16 pixels have all stopped (T <= 1e-4 in front of the group: the transmittance only
                // falls) adds exact zeros to every sum of every Gaussian of the group and its states need not move: the
                // row is left out (wave-uniform).  The tile's list ends where its LAST pixel stops, so at 2.4 M Gaussians a
                // good part of the rows of a tile's later groups are of this kind.
                if (GS_BWD_MFMA_ROW_SKIP &&
                    __ballot(Tin[0] > GS_T_STOP || Tin[1] > GS_T_STOP || Tin[2] > GS_T_STOP || Tin[3] > GS_T_STOP) == 0ull) {
                    if (GS_BWD_MFMA_PF && s + 1 < 16) row_loads(s + 1);
                    continue;
                }
                ++n_exec;
                // colour logits of (pixel 4 jq + reg, Gaussian gq) on the matrix pipe
                f4 lg[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                if (TAIL) {
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) lg[ch] = S8 * ctail[ch];
                }
#pragma unroll
                for (int kk = 0; kk < KQ; ++kk)
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) lg[ch] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], cob[ch][kk], lg[ch], 0, 0, 0);
#if GS_BWD_MFMA_PK
                const float dy = pyt[s] - gy;
                const f2 dy2 = {dy, dy};
                f4 Tout, Rout;
                float dv[3][4];
                f2 q2[2];
                float araw[4], pin[4], Tb[4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    q2[h] = pfma(pfma(f2{cC, cC}, dy2, -f2{bdx[2 * h], bdx[2 * h + 1]}), dy2, f2{adx2[2 * h], adx2[2 * h + 1]});
                    araw[2 * h] = gs_exp2(-q2[h].x);  // (q is q' = q - log2 sigma(opa): 2^-q' is alpha itself)
                    araw[2 * h + 1] = gs_exp2(-q2[h].y);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pin[i] = fminf(fmaxf(1.0f - araw[i], 0.f), 1.0f);  // (the subtraction's clamp modifier; see below)
                    Tb[i] = Tin[i];
                }
                gs_row_scan_mul4_excl(pin, Tb);
                float alpha[4], wg[4], wsum[4];
                f2 w2[2], al2[2], cc2[3][2];
#pragma unroll
                for (int i = 0; i < 4; ++i) alpha[i] = Tb[i] > GS_T_STOP ? araw[i] : 0.f;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    al2[h] = f2{alpha[2 * h], alpha[2 * h + 1]};
                    w2[h] = al2[h] * f2{Tb[2 * h], Tb[2 * h + 1]};
                    // colours sigma(logit) = 1 / (1 + 2^(logit')) with the pre-scaled basis
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        const f2 den = f2{1.0f, 1.0f} + f2{gs_exp2(lg[ch][2 * h]), gs_exp2(lg[ch][2 * h + 1])};
                        cc2[ch][h] = f2{gs_rcp(den.x), gs_rcp(den.y)};
                    }
                    const f2 g0 = h ? G0.zw : G0.xy, g1 = h ? G1.zw : G1.xy, g2 = h ? G2.zw : G2.xy;
                    const f2 wgh = w2[h] * pfma(g2, cc2[2][h], pfma(g1, cc2[1][h], g0 * cc2[0][h]));
                    wg[2 * h] = wgh.x;
                    wg[2 * h + 1] = wgh.y;
                    const f2 to = (h ? Tin.zw : Tin.xy) * f2{pin[2 * h], pin[2 * h + 1]};  // (lanes of Gaussian 15: the group's product)
                    Tout[2 * h] = to.x;
                    Tout[2 * h + 1] = to.y;
                }
                gs_row_scan_add4_oop(wsum, wg);  // rho behind this Gaussian: rho_in minus the inclusive prefix sum of w gc
                f2 svs[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f2 rho = (h ? Rin.zw : Rin.xy) - f2{wsum[2 * h], wsum[2 * h + 1]};
                    const f2 den = f2{1.00000011920928955f, 1.00000011920928955f} - f2{araw[2 * h], araw[2 * h + 1]};
                    // s = dL/dalpha alpha = w gc - rho beta, beta = alpha / (1 - alpha + 1e-7): masked with alpha, like w
                    const f2 sv = pfma(-rho, al2[h] * f2{gs_rcp(den.x), gs_rcp(den.y)}, f2{wg[2 * h], wg[2 * h + 1]});
                    // D = dL/dC_ch w c (1 - c) [x -ln 2: D sh' = D' sh]
                    const f2 wk = w2[h] * f2{-GS_LN2, -GS_LN2};
                    const f2 g0 = h ? G0.zw : G0.xy, g1 = h ? G1.zw : G1.xy, g2 = h ? G2.zw : G2.xy;
                    const f2 d0 = (g0 * wk) * pfma(-cc2[0][h], cc2[0][h], cc2[0][h]);
                    const f2 d1 = (g1 * wk) * pfma(-cc2[1][h], cc2[1][h], cc2[1][h]);
                    const f2 d2 = (g2 * wk) * pfma(-cc2[2][h], cc2[2][h], cc2[2][h]);
                    dv[0][2 * h] = d0.x, dv[0][2 * h + 1] = d0.y;
                    dv[1][2 * h] = d1.x, dv[1][2 * h + 1] = d1.y;
                    dv[2][2 * h] = d2.x, dv[2][2 * h + 1] = d2.y;
                    S1p[h] += sv;
                    Syp[h] = pfma(sv, dy2, Syp[h]);
                    Sqp = pfma(sv, q2[h], Sqp);
                    svs[h] = sv;
                    Rout[2 * h] = rho.x;
                    Rout[2 * h + 1] = rho.y;
                }
                const f2 ss = svs[0] + svs[1];
                Syy = fmaf((ss.x + ss.y) * dy, dy, Syy);
#else
                const float dy = pyt[s] - gy;
                f4 Tout, Rout;
                float dv[3][4];
                float q[4], Gv[4], araw[4], pin[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    q[i] = fmaf(fmaf(cC, dy, -bdx[i]), dy, adx2[i]);
                    Gv[i] = gs_exp2(-q[i]);
                    araw[i] = GS_BWD_MFMA_DIET ? Gv[i] : Gv[i] * opa;  // (diet: q is q', 2^-q' is alpha itself)
                    // (0 <= 1 - alpha <= 1 holds anyway for sane records; written as a clamp to [0, 1] it is the clamp
                    // modifier of the subtraction, one instruction -- and a NaN / > 1 alpha of a broken record stops the pixel)
                    pin[i] = fminf(fmaxf(1.0f - araw[i], 0.f), 1.0f);
                }
                // transmittance in front of this Gaussian: T_in times the product over the group's earlier Gaussians
                float Tb[4], alpha[4], w[4], gc[4], wg[4], cc[3][4];
                bool live[4];
#if GS_BWD_MFMA_DIET
#pragma unroll
                for (int i = 0; i < 4; ++i) Tb[i] = Tin[i];
                gs_row_scan_mul4_excl(pin, Tb);
#else
                gs_row_scan_mul4(pin);
#endif
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (!GS_BWD_MFMA_DIET) Tb[i] = Tin[i] * gs_dpp<0x111>(1.0f, pin[i]);
                    live[i] = Tb[i] > GS_T_STOP;
                    alpha[i] = live[i] ? araw[i] : 0.f;
                    w[i] = alpha[i] * Tb[i];
                    // colours sigma(logit) = 1 / (1 + 2^(logit')) with the pre-scaled basis
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) cc[ch][i] = gs_rcp(1.0f + gs_exp2(lg[ch][i]));
                    gc[i] = fmaf(G2[i], cc[2][i], fmaf(G1[i], cc[1][i], G0[i] * cc[0][i]));
                    wg[i] = w[i] * gc[i];
                    Tout[i] = Tin[i] * pin[i];  // (meaningful in the lanes of Gaussian 15: the whole group's product)
                }
                // rho behind this Gaussian: rho_in minus the inclusive prefix sum of w gc
#if GS_BWD_MFMA_DIET
                float wsum[4];
                gs_row_scan_add4_oop(wsum, wg);
#else
                gs_row_scan_add4(wg);
                const float *wsum = wg;
#endif
                float ssum = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float rho = Rin[i] - wsum[i];
#if GS_BWD_MFMA_DIET
                    // s = dL/dalpha alpha = w gc - rho beta, beta = alpha / (1 - alpha + 1e-7): masked with alpha, like w
                    const float sv = fmaf(-rho, alpha[i] * gs_rcp(1.00000011920928955f - araw[i]), wg[i]);
#else
                    const float rc = gs_rcp(fmaf(-Gv[i], opa, 1.00000011920928955f));  // 1 / (1 - alpha + 1e-7)
                    float d_alpha = fmaf(Tb[i], gc[i], -(rho * rc));
                    d_alpha = live[i] ? d_alpha : 0.f;
                    const float sv = d_alpha * alpha[i];
#endif
                    // D = dL/dC_ch w c (1 - c) [x -ln 2: D sh' = D' sh]
                    const float wk = w[i] * -GS_LN2;
                    dv[0][i] = (G0[i] * wk) * fmaf(-cc[0][i], cc[0][i], cc[0][i]);
                    dv[1][i] = (G1[i] * wk) * fmaf(-cc[1][i], cc[1][i], cc[1][i]);
                    dv[2][i] = (G2[i] * wk) * fmaf(-cc[2][i], cc[2][i], cc[2][i]);
                    // (the opacity sum, sum of dL/dalpha G over the live pixels, is sum of s / opacity: at the group's end)
                    S1[i] += sv;
                    Sy[i] = fmaf(sv, dy, Sy[i]);
                    ssum += sv;
                    Sq = fmaf(sv, q[i], Sq);
                    Rout[i] = rho;
                }
                Syy = fmaf(ssum * dy, dy, Syy);
#endif
                if (GS_BWD_MFMA_PF && s + 1 < 16) {
                    row_loads(s + 1);
                    __builtin_amdgcn_sched_barrier(0);  // (the requests stay in front of the MFMAs below)
                }
                // coefficient sums: acc[ch][reg] += sum over the row's 16 pixels of D[ch] sh'_(4 jq + reg)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) acc[ch] = __builtin_amdgcn_mfma_f32_16x16x4f32(tb[i], dv[ch][i], acc[ch], 0, 0, 0);
                if (gq == 15) {  // the row's pixel states in front of the next group
                    *reinterpret_cast<f4 *>(sT + prow + 4 * jq) = Tout;
                    *reinterpret_cast<f4 *>(sR + prow + 4 * jq) = Rout;
                }
            }
            asm volatile("" ::"v"(warm));  // (the warming load must not be dropped; its value is not used)
#if GS_BWD_MFMA_PK
            S1[0] = S1p[0].x, S1[1] = S1p[0].y, S1[2] = S1p[1].x, S1[3] = S1p[1].y;
            Sy[0] = Syp[0].x, Sy[1] = Syp[0].y, Sy[2] = Syp[1].x, Sy[3] = Syp[1].y;
            Sq = Sqp.x + Sqp.y;
#endif
            // ---- close the group: the lane's four pixel columns, then the four pixel quads of the Gaussian
            const uint32_t gid2 = (uint32_t)__shfl((int)id_lane, (int)(valid ? gi : r - 1), 64);
            const float4 ge2 = S.geom[(size_t)gid2 * GS_REC_STRIDE], cv2 = S.cov4[(size_t)gid2 * GS_REC_STRIDE];
            float Sx = 0.f, Sxx = 0.f, Sxy = 0.f, Syt = 0.f;
            // s = dL/dalpha alpha and alpha = G sigma(opa) on every live pixel: sum dL/dalpha G = (sum s) / sigma(opa).  (An
            // opacity that underflowed to 0 composites nothing; its derivative sigma (1 - sigma) downstream is 0 as well.)
            float Sopa = ((S1[0] + S1[1]) + (S1[2] + S1[3])) * (ge2.w > 0.f ? gs_rcp(ge2.w) : 0.f);
            const float lopa2 = fmaxf(__log2f(ge2.w), -200.0f);
            (void)lopa2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float dxi = raster_pixel_coord(tx * 16 + 4 * jq + i, G.padW, G.focal_x) - ge2.x;
                const float sx = S1[i] * dxi;
                Sx += sx;
                Sxx = fmaf(sx, dxi, Sxx);
                Sxy = fmaf(Sy[i], dxi, Sxy);
                Syt += Sy[i];
            }
            auto quad_sum = [](float v) {
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                return v;
            };
            Sx = quad_sum(Sx);
            Syt = quad_sum(Syt);
            Sxx = quad_sum(Sxx);
            Sxy = quad_sum(Sxy);
            Syy = quad_sum(Syy);
            Sq = quad_sum(Sq);
            Sopa = quad_sum(Sopa);
            // ---- the group's 16 rows leave as whole 64-byte lines (round 5; row layout: gs_frame_layout.h).  A row is spread
            // over the four lanes (g, jq) of its Gaussian; line `it` of the row is put together from one 16-byte piece of each
            // of them -- destination lane 4 r + q takes the piece of lane (g = r, jq = q) through four ds_bpermute -- and
            // stored by ONE instruction, four lanes x 16 bytes per line, sixteen rows per instruction: every line reaches
            // the memory system complete (round 4: twelve scalar stores per lane at a 28-byte offset into rows that straddle
            // lines and a flag byte per row -- PMC WRITE_SIZE 2.2 x / 1.32 x the row bytes).  The geometry algebra runs in
            // all four lanes of a Gaussian (they hold the same sums after quad_sum): whichever lane carries a header piece
            // has it.
            float h0[4], h1[4];  // header pieces (dx, dy, da, db), (dc, dd, dopa, 0)
            {
                const float a = cv2.x, bb = cv2.y, cc = cv2.z, d = cv2.w;
                float cA, cB, cC;
                gs_conic(a, bb, cc, d, cA, cB, cC);
                const float iPn = 1.0f / (2.0f * raster_det(a, bb, cc, d) + 1e-14f);
                // (diet: the loop summed s q' with q' = q - log2 sigma(opa); Sopa x sigma(opa) is the sum of s)
                const float Su = (GS_BWD_MFMA_DIET ? fmaf(lopa2, Sopa * ge2.w, Sq) : Sq) * GS_LN2;
                h0[0] = GS_LN2 * (2.0f * cA * Sx - cB * Syt);
                h0[1] = GS_LN2 * (2.0f * cC * Syt - cB * Sx);
                h0[2] = iPn * (-Syy + 2.0f * d * Su);
                h0[3] = iPn * (Sxy - 2.0f * cc * Su);
                h1[0] = iPn * (Sxy - 2.0f * bb * Su);
                h1[1] = iPn * (-Sxx + 2.0f * a * Su);
                h1[2] = Sopa;
                h1[3] = 0.f;
            }
            {
                const int dr = lane >> 2, dq = lane & 3, src = dr + 16 * dq;  // destination row / piece, its source lane
                const uint32_t dslot = (uint32_t)__shfl((int)slot, dr, 64);
                float4 *drow = reinterpret_cast<float4 *>(O.rows + (size_t)(dslot != GS_NO_SLOT ? dslot : 0) * RW);
                auto put = [&](int line, const float v[4]) {  // this lane OFFERS v; lane 4 r + q stores what lane (r, q) offers
                    float4 o;
                    o.x = __shfl(v[0], src, 64);
                    o.y = __shfl(v[1], src, 64);
                    o.z = __shfl(v[2], src, 64);
                    o.w = __shfl(v[3], src, 64);
                    if (dslot != GS_NO_SLOT) drow[4 * line + dq] = o;
                };
                if constexpr (NB == 16) {
                    // line 0: the header, pieces 0 / 1 from lanes jq = 0 / 1, zeros from the others; line 1 + ch: coef(ch, 0..15)
                    float hv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) hv[e] = jq == 0 ? h0[e] : jq == 1 ? h1[e] : 0.f;
                    put(0, hv);
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        const float cv[4] = {acc[ch][0], acc[ch][1], acc[ch][2], acc[ch][3]};
                        put(1 + ch, cv);
                    }
                } else {
                    // line ch: coef(ch, 0..11) from lanes jq = 0..2 (k >= 9: the table's zero entries: exact zeros), header
                    // piece ch from lane jq = 3, which has no coefficient
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        float cv[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) cv[e] = jq == 3 ? (ch == 0 ? h0[e] : ch == 1 ? h1[e] : 0.f) : acc[ch][e];
                        put(ch, cv);
                    }
                }
            }
        }
        lds_order();  // (the next bucket's states overwrite this wave's arrays)
    }
    }  // work items
    if (lane == 0) O.exec_rows[(size_t)blockIdx.x * W + wave] = n_exec;
}

// The SH backward's work items: (tile, first bucket, buckets <= chunk) in the forward's dispatch order of the tiles (`order`:
// heavy tiles first; NULL: raster order); `cap`: buckets per tile the matrix-pipe kernel takes at most (long-list frames).
__global__ void __launch_bounds__(1024) mfma_items_kernel(const uint32_t *__restrict__ tile_nproc, int n_tiles,
                                                         const uint32_t *__restrict__ order, uint32_t cap, uint32_t chunk,
                                                         uint4 *__restrict__ items, uint32_t *__restrict__ n_items) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < n_tiles; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t tile = i < n_tiles ? (order ? order[i] : (uint32_t)i) : 0u;
        uint32_t nb = i < n_tiles ? (tile_nproc[tile] + GS_BUCKET - 1) / GS_BUCKET : 0u;
        if (cap && nb > cap) nb = cap;
        const uint32_t v = (nb + chunk - 1) / chunk;
        const uint32_t incl = gs_wave_incl_scan_u32(v);
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) woff += w < wave ? s_wave[w] : 0;
        const uint32_t carry = s_carry, first = carry + woff + incl - v;
        for (uint32_t c = 0; c < v; ++c) {
            const uint32_t b0 = c * chunk, left = nb - b0;
            items[first + c] = make_uint4(tile, b0, left < chunk ? left : chunk, 0u);
        }
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_items = s_carry;
}

// ---------------------------------------------------------------------------------------------
// rgb colours, frame path (round 5): the ROW layout of the kernel above without its matrix products.
//
// One wave per (tile, bucket of 64 Gaussians) -- the grid of raster_backward_pixel_sh_kernel<3>, which this kernel replaces on
// the frame path --, but inside the bucket lanes own GAUSSIANS, not pixels: lane l = (Gaussian l & 15 of the current group of
// 16, pixel quad l >> 4), and a step is one pixel ROW of the tile: 16 Gaussians x 16 pixels, four (pixel, Gaussian) pairs per
// lane.  What that buys for rgb colours, where there is no contraction to hand to the matrix pipe:
//   * the ten per-Gaussian sums accumulate in the lane's registers over the 16 pixel rows (a lane's four pixel columns have
//     constant dx) and are reduced 4 : 1 once per group -- the pixel-parallel kernel reduces ten rows of 64 partials through
//     LDS for EVERY Gaussian (PMC round 4: 42.8 M LDS + ~40 M VALU wave instructions of its 237 M per launch);
//   * the Gaussian's constants live in registers (no nine broadcast LDS reads per Gaussian);
//   * a pixel row whose 16 pixels have all stopped is left out, wave-uniformly: 14 % of the row steps of the 2.4 M scene
//     (the pixel layout could only skip half tiles, 5 %, and lost the gain to the branches).
// What it pays: the transmittance / rho recursions over the group's 16 Gaussians are DPP row scans (36 DPP instructions per
// step against eight plain ones).  Two more savings that the SH kernel does not have yet: the opacity is folded into the
// exponent (alpha = 2^(log2 sigma(opa) - q), as the forward does: no G x opacity product; sum dL/dalpha G is (sum s) /
// sigma(opa) and sum s q is corrected by log2 sigma(opa) sum s at the group's end), and the transmittance in front of a
// Gaussian is ONE v_mul_f32_dpp (row_shr:1 of the scanned products times T_in, lanes without a source keep T_in).
// Rows leave as one aligned 64-byte line each, sixteen rows per store instruction; no flags (stop keys).  No atomics; the
// sums run in a fixed order: bitwise repeatable.
#ifndef GS_BWD_RGB_ROWS
// 2: this kernel in frames flagged GS_FRAME_BWD_ROWS (include/gs_abi.h: the caller's statistic says most buckets belong to
//    saturated tiles), raster_backward_pixel_sh_kernel<3> in the others; 1: this kernel always; 0: never (A/B switch).
// Why not always: kernel traces on one box (profiles/r05_e_*) -- 2.4 M Gaussians, every tile saturates: 446 -> 408 us; 376 k
// Gaussians, none does and no row is ever left out: 276 -> 305 us (the DPP scans cost more than the LDS reduction they
// replace).  Why not per tile (built: two work lists, one kernel each): the second launch is serial behind the first on
// the stream and costs the tail of its slowest lone wave (33 us at 376 k Gaussians for the ~100 tiles that do saturate).
#define GS_BWD_RGB_ROWS 2
#endif
#ifndef GS_BWD_ROWS_PF
// 1: LDS operands of a pixel row requested one row ahead.  Measured equal (same box, 2.4 M Gaussians: 0.497 / 0.514 ms
// without against 0.508 / 0.498 ms with; profiles/r05_b_*): four resident waves per SIMD cover the LDS latency, and without
// the second register set the kernel needs 105 VGPRs and no scratch (128 and three spilled registers with it; with the
// packed row step, GS_BWD_ROWS_PK: 126 and no scratch)
#define GS_BWD_ROWS_PF 0
#endif
#ifndef GS_BWD_ROWS_WPE
#define GS_BWD_ROWS_WPE 4  // waves per SIMD the register allocation aims at
#endif
#ifndef GS_BWD_ROWS_PK
// 1: the per-pixel algebra of a row step on PAIRS of the lane's four pixels (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two
// fp32 results per issue slot; the kernel sits at the VALU issue limit).  The DPP scans, the exponentials / reciprocals and
// the stop-point selects stay per pixel.  0: one instruction per pixel (the first version; A/B switch)
#define GS_BWD_ROWS_PK 1
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GS_BWD_ROWS_WPE)))
raster_backward_rows_kernel(RasterSrc S, RasterGeom G, BwdIn I, BwdOut O) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr uint32_t GS_NO_SLOT = 0xffffffffu;
    constexpr int RW = gs_row_floats(3);
    __shared__ __attribute__((aligned(16))) float s_gr[3][256];  // dL/dC of the tile's pixels (masked: crop, clamp)
    __shared__ __attribute__((aligned(16))) float s_T[256];      // transmittance / rho in front of the current group
    __shared__ __attribute__((aligned(16))) float s_rho[256];
    __shared__ float s_py[16];
    auto lds_order = [] {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    };
    const int lane = threadIdx.x;
    const uint32_t gq = (uint32_t)lane & 15u, jq = (uint32_t)lane >> 4;  // Gaussian of the group, pixel quad
    // the work list with the grid as the stride: see raster_backward_pixel_sh_kernel
    const uint32_t n_tiles = (uint32_t)(G.ntx * G.nty), n_work = I.bucket_offsets[n_tiles];
    for (uint32_t kb = blockIdx.x; kb < n_work; kb += gridDim.x) {
    const uint4 info = I.bucket_info[kb];
    const uint32_t tile = info.x, base = info.y, r = info.z, start = info.w;
    const uint32_t tx = tile % (uint32_t)G.ntx, ty = tile / (uint32_t)G.ntx;
    // the bucket's 64 Gaussian ids, one per lane (a group learns its ids from a lane exchange)
    const uint32_t id_lane = S.ids[start + base + ((uint32_t)lane < r ? (uint32_t)lane : r - 1)];

    // ---- the tile's pixels at the bucket's boundary: pixel 64 k + lane = (x = lane & 15, y = (lane >> 4) + 4 k).  Every
    // load is issued before anything is waited for (see raster_backward_pixel_kernel).
    {
        const uint32_t id_x = tx * 16 + ((uint32_t)lane & 15), id_y0 = ty * 16 + ((uint32_t)lane >> 4);
        const float4 *ck = I.ckpt + raster_ckpt_slot(start, tile, base / GS_BUCKET) * 256;
        float4 c[4];
        float f[4][3], gr[4][3];
        bool inside[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            c[k] = ck[64 * k + lane];  // (stale for the tile's first bucket: replaced below)
            const uint32_t id_y = id_y0 + 4 * k;
            const float *cf = I.c_final + ((size_t)id_y * G.padW + id_x) * 3;
            const int ox = (int)id_x - G.crop_left, oy = (int)id_y - G.crop_top;
            inside[k] = ox >= 0 && ox < G.width && oy >= 0 && oy < G.height;
            const float *gp = I.grad + ((size_t)(inside[k] ? oy : 0) * G.width + (inside[k] ? ox : 0)) * 3;
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                f[k][e] = cf[e];
                gr[k][e] = gp[e];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            asm volatile("" ::"v"(gr[k][0]), "v"(gr[k][1]), "v"(gr[k][2]), "v"(f[k][0]), "v"(f[k][1]), "v"(f[k][2]),
                         "v"(c[k].x), "v"(c[k].y), "v"(c[k].z), "v"(c[k].w));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = 64 * k + lane;
            float g3[3];
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                g3[e] = (inside[k] && f[k][e] >= 0.f && f[k][e] <= 1.f) ? gr[k][e] : 0.f;
                s_gr[e][p] = g3[e];
            }
            // the tile's first bucket starts from the empty pixel (T, C) = (1, 0), which the forward does not store
            const float4 ci = base == 0 ? make_float4(1.f, 0.f, 0.f, 0.f) : c[k];
            s_T[p] = ci.x;
            // rho = dL/dC . (C_final - C_run): the only way the remaining colour enters dL/dalpha (gaussian.cu:716-722)
            s_rho[p] = g3[0] * (f[k][0] - ci.y) + g3[1] * (f[k][1] - ci.z) + g3[2] * (f[k][2] - ci.w);
        }
        if (lane < 16) s_py[lane] = raster_pixel_coord(ty * 16 + (uint32_t)lane, G.padH, G.focal_y);
    }
    float pxs[4];  // centres of this lane's four pixel columns
#pragma unroll
    for (int i = 0; i < 4; ++i) pxs[i] = raster_pixel_coord(tx * 16 + 4 * jq + i, G.padW, G.focal_x);
    lds_order();

    const uint32_t ngrp = (r + 15) / 16;
    for (uint32_t grp = 0; grp < ngrp; ++grp) {
        // ---- this lane's Gaussian (entries beyond r re-read the bucket's last one and are switched off: alpha = 0)
        const uint32_t gi = grp * 16 + gq;
        const bool valid = gi < r;
        const uint32_t gid = (uint32_t)__shfl((int)id_lane, (int)(valid ? gi : r - 1), 64);
        const float4 *rec = S.geom + (size_t)gid * GS_REC_STRIDE;  // ONE 64-byte line: geom | cov | colour | conic
        const float4 ge = rec[0], col = rec[2], cq = rec[3];
        uint32_t slot = GS_NO_SLOT;
        if (valid) {
            const uint4 rc = O.rects[gid];
            const uint32_t y0 = rc.x & 0xffff, x0 = rc.y & 0xffff, x1 = rc.y >> 16;
            const uint64_t sl = (uint64_t)O.pair_offsets[gid] + (ty - y0) * (x1 - x0) + (tx - x0);
            if (sl < O.max_pairs) slot = (uint32_t)sl;
        }
        // alpha = sigma(opa) 2^-q = 2^-(q - log2 sigma(opa)): the opacity rides in the exponent's constant term (an opacity
        // that underflowed to 0 composites nothing: the clamp keeps q' finite, 2^-(q + 200) is flushed to 0)
        const float lopa = fmaxf(__log2f(ge.w), -200.0f);
        const float cC = cq.z, gy = ge.y, c0 = col.x, c1 = col.y, c2 = col.z;
        float nbdx[4], adx2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float dxi = pxs[i] - ge.x;
            nbdx[i] = -(cq.y * dxi);
            // (a padded entry: q' = 1e30 => alpha = 0 exactly, and 0 x 1e30 = 0 in the sum of s q')
            adx2[i] = valid ? fmaf(cq.x * dxi, dxi, -lopa) : 1e30f;
        }
        float S1[4] = {0.f, 0.f, 0.f, 0.f}, Sy[4] = {0.f, 0.f, 0.f, 0.f}, Syy = 0.f, Sq = 0.f;
        float Sc0 = 0.f, Sc1 = 0.f, Sc2 = 0.f;
        // The LDS operands of a pixel row are requested one row ahead (GS_BWD_ROWS_PF); the loop runs two rows per trip so
        // that the two register sets take turns without moves.
        struct RowIn {
            f4 T, R, G0, G1, G2;
            float py;
        };
        auto row_loads = [&](RowIn &d, int s) {
            const int prow = 16 * s + 4 * (int)jq;
            d.T = *reinterpret_cast<const f4 *>(s_T + prow);
            d.R = *reinterpret_cast<const f4 *>(s_rho + prow);
            d.G0 = *reinterpret_cast<const f4 *>(s_gr[0] + prow);
            d.G1 = *reinterpret_cast<const f4 *>(s_gr[1] + prow);
            d.G2 = *reinterpret_cast<const f4 *>(s_gr[2] + prow);
            d.py = s_py[s];
        };
#if GS_BWD_ROWS_PK
        typedef float f2 __attribute__((ext_vector_type(2)));
        auto sp = [](float v) { return f2{v, v}; };
        auto pfma = [](f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); };
        f2 Scp[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}}, S1p[2] = {{0.f, 0.f}, {0.f, 0.f}}, Syp[2] = {{0.f, 0.f}, {0.f, 0.f}};
        f2 Sqp = {0.f, 0.f};
        const f2 nbdx2[2] = {{nbdx[0], nbdx[1]}, {nbdx[2], nbdx[3]}}, adx22[2] = {{adx2[0], adx2[1]}, {adx2[2], adx2[3]}};
        auto row_step = [&](RowIn &cur, RowIn &nxt, int s) {  // pixel row s of the tile
            if (!GS_BWD_ROWS_PF) row_loads(cur, s);
            const bool any_live = __ballot(cur.T[0] > GS_T_STOP || cur.T[1] > GS_T_STOP || cur.T[2] > GS_T_STOP ||
                                           cur.T[3] > GS_T_STOP) != 0ull;
            if (GS_BWD_ROWS_PF && s + 1 < 16) row_loads(nxt, s + 1);
            if (!any_live) return;  // (see the per-pixel version below)
            const float dy = cur.py - gy;
            const f2 dy2 = sp(dy);
            f2 q2[2];
            float araw[4], pin[4], Tb[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                q2[h] = pfma(pfma(sp(cC), dy2, nbdx2[h]), dy2, adx22[h]);  // q' = q - log2 sigma(opa); the forward's evaluation order
                araw[2 * h] = gs_exp2(-q2[h].x);
                araw[2 * h + 1] = gs_exp2(-q2[h].y);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pin[i] = fminf(fmaxf(1.0f - araw[i], 0.f), 1.0f);  // (the subtraction's clamp modifier)
                Tb[i] = cur.T[i];
            }
            gs_row_scan_mul4_excl(pin, Tb);
            float alpha[4], wg[4], ws[4];
            f2 w2[2], al2[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) alpha[i] = Tb[i] > GS_T_STOP ? araw[i] : 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                al2[h] = f2{alpha[2 * h], alpha[2 * h + 1]};
                w2[h] = al2[h] * f2{Tb[2 * h], Tb[2 * h + 1]};
                const f2 g0 = h ? cur.G0.zw : cur.G0.xy, g1 = h ? cur.G1.zw : cur.G1.xy, g2 = h ? cur.G2.zw : cur.G2.xy;
                const f2 wgh = w2[h] * pfma(g2, sp(c2), pfma(g1, sp(c1), g0 * sp(c0)));  // w (dL/dC . colour)
                wg[2 * h] = wgh.x;
                wg[2 * h + 1] = wgh.y;
            }
            gs_row_scan_add4_oop(ws, wg);
            f2 rho2[2], svs[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f2 R2 = h ? cur.R.zw : cur.R.xy;
                rho2[h] = R2 - f2{ws[2 * h], ws[2 * h + 1]};  // rho behind this Gaussian
                const f2 den = sp(1.00000011920928955f) - f2{araw[2 * h], araw[2 * h + 1]};
                const f2 beta = al2[h] * f2{gs_rcp(den.x), gs_rcp(den.y)};
                const f2 sv = pfma(-rho2[h], beta, f2{wg[2 * h], wg[2 * h + 1]});  // s = w gc - rho beta (see below)
                const f2 g0 = h ? cur.G0.zw : cur.G0.xy, g1 = h ? cur.G1.zw : cur.G1.xy, g2 = h ? cur.G2.zw : cur.G2.xy;
                Scp[0] = pfma(g0, w2[h], Scp[0]);
                Scp[1] = pfma(g1, w2[h], Scp[1]);
                Scp[2] = pfma(g2, w2[h], Scp[2]);
                S1p[h] += sv;
                Syp[h] = pfma(sv, dy2, Syp[h]);
                svs[h] = sv;
                Sqp = pfma(sv, q2[h], Sqp);
            }
            const f2 ss = svs[0] + svs[1];
            Syy = fmaf((ss.x + ss.y) * dy, dy, Syy);
            if (gq == 15) {  // the row's pixel states in front of the next group: T behind the group's last Gaussian
                f4 Tout, Rout;
#pragma unroll
                for (int i = 0; i < 4; ++i) Tout[i] = cur.T[i] * pin[i];
                Rout.xy = rho2[0];
                Rout.zw = rho2[1];
                *reinterpret_cast<f4 *>(s_T + 16 * s + 4 * jq) = Tout;
                *reinterpret_cast<f4 *>(s_rho + 16 * s + 4 * jq) = Rout;
            }
        };
#else
        auto row_step = [&](RowIn &cur, RowIn &nxt, int s) {  // pixel row s of the tile
            if (!GS_BWD_ROWS_PF) row_loads(cur, s);
            // a pixel row whose 16 pixels have all stopped adds exact zeros to every sum of every Gaussian of the group and
            // its states need not move: left out (wave-uniform; the transmittance only falls)
            const bool any_live = __ballot(cur.T[0] > GS_T_STOP || cur.T[1] > GS_T_STOP || cur.T[2] > GS_T_STOP ||
                                           cur.T[3] > GS_T_STOP) != 0ull;
            if (GS_BWD_ROWS_PF && s + 1 < 16) row_loads(nxt, s + 1);
            if (!any_live) return;
            const float dy = cur.py - gy;
            float q[4], araw[4], pin[4], Tb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                q[i] = fmaf(fmaf(cC, dy, nbdx[i]), dy, adx2[i]);  // q' = q - log2 sigma(opa); the forward's evaluation order
                araw[i] = gs_exp2(-q[i]);
                // (0 <= 1 - alpha <= 1 for sane records; as a clamp it is the subtraction's clamp modifier, and a NaN /
                // > 1 alpha of a broken record stops the pixel)
                pin[i] = fminf(fmaxf(1.0f - araw[i], 0.f), 1.0f);
                Tb[i] = cur.T[i];
            }
            // inclusive product over the group's Gaussians, in place; the transmittance in front of this Gaussian: T_in x the
            // product over the group's EARLIER Gaussians -- one DPP multiply: lane g >= 1 takes the scanned product of lane
            // g - 1, lane 0 has no source and keeps T_in
            gs_row_scan_mul4_excl(pin, Tb);
            float alpha[4], w[4], wg[4], ws[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                alpha[i] = Tb[i] > GS_T_STOP ? araw[i] : 0.f;
                w[i] = alpha[i] * Tb[i];
                wg[i] = w[i] * fmaf(cur.G2[i], c2, fmaf(cur.G1[i], c1, cur.G0[i] * c0));  // w (dL/dC . colour)
            }
            // inclusive prefix sum of w gc over the group's Gaussians, OUT of place: the unscanned wg stays for s below
            gs_row_scan_add4_oop(ws, wg);
            float ssum = 0.f;
            f4 Rout;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float rho = cur.R[i] - ws[i];  // rho behind this Gaussian
                // s = dL/dalpha alpha with dL/dalpha = T gc - rho / (1 - alpha + 1e-7) (gaussian.cu:716-722)
                //   = w gc - rho beta,  beta = alpha / (1 - alpha + 1e-7): masked with alpha, like w
                const float beta = alpha[i] * gs_rcp(1.00000011920928955f - araw[i]);
                const float sv = fmaf(-rho, beta, wg[i]);
                Sc0 = fmaf(cur.G0[i], w[i], Sc0);
                Sc1 = fmaf(cur.G1[i], w[i], Sc1);
                Sc2 = fmaf(cur.G2[i], w[i], Sc2);
                S1[i] += sv;
                Sy[i] = fmaf(sv, dy, Sy[i]);
                ssum += sv;
                Sq = fmaf(sv, q[i], Sq);
                Rout[i] = rho;
            }
            Syy = fmaf(ssum * dy, dy, Syy);
            if (gq == 15) {  // the row's pixel states in front of the next group: T behind the group's last Gaussian
                f4 Tout;
#pragma unroll
                for (int i = 0; i < 4; ++i) Tout[i] = cur.T[i] * pin[i];
                *reinterpret_cast<f4 *>(s_T + 16 * s + 4 * jq) = Tout;
                *reinterpret_cast<f4 *>(s_rho + 16 * s + 4 * jq) = Rout;
            }
        };
#endif
        RowIn ra, rb;
        if (GS_BWD_ROWS_PF) row_loads(ra, 0);
        for (int s = 0; s < 16; s += 2) {
            row_step(ra, rb, s);
            row_step(rb, ra, s + 1);
        }
#if GS_BWD_ROWS_PK
        S1[0] = S1p[0].x, S1[1] = S1p[0].y, S1[2] = S1p[1].x, S1[3] = S1p[1].y;
        Sy[0] = Syp[0].x, Sy[1] = Syp[0].y, Sy[2] = Syp[1].x, Sy[3] = Syp[1].y;
        Sc0 = Scp[0].x + Scp[0].y, Sc1 = Scp[1].x + Scp[1].y, Sc2 = Scp[2].x + Scp[2].y;
        Sq = Sqp.x + Sqp.y;
#endif
        // ---- close the group: the lane's four pixel columns, then the four pixel quads of the Gaussian
        const float4 cv = rec[1];
        float Sx = 0.f, Sxx = 0.f, Sxy = 0.f, Syt = 0.f, Stot = (S1[0] + S1[1]) + (S1[2] + S1[3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float dxi = pxs[i] - ge.x;
            const float sx = S1[i] * dxi;
            Sx += sx;
            Sxx = fmaf(sx, dxi, Sxx);
            Sxy = fmaf(Sy[i], dxi, Sxy);
            Syt += Sy[i];
        }
        auto quad_sum = [](float v) {
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            return v;
        };
        Sx = quad_sum(Sx);
        Syt = quad_sum(Syt);
        Sxx = quad_sum(Sxx);
        Sxy = quad_sum(Sxy);
        Syy = quad_sum(Syy);
        Sq = quad_sum(Sq);
        Stot = quad_sum(Stot);
        Sc0 = quad_sum(Sc0);
        Sc1 = quad_sum(Sc1);
        Sc2 = quad_sum(Sc2);
        // s = dL/dalpha alpha and alpha = G sigma(opa) on every live pixel: sum dL/dalpha G = (sum s) / sigma(opa); the
        // loop summed s q' with q' = q - log2 sigma(opa)
        const float Sopa = Stot * (ge.w > 0.f ? gs_rcp(ge.w) : 0.f);
        const float Su = fmaf(lopa, Stot, Sq) * GS_LN2;  // u = -ln G = q ln 2
        float piece[4];
        {
            const float a = cv.x, bb = cv.y, cc = cv.z, d = cv.w;
            const float iPn = 1.0f / (2.0f * raster_det(a, bb, cc, d) + 1e-14f);
            // every lane of the Gaussian holds the same sums: lane (g, jq) puts piece jq of the row together
            if (jq == 0) {
                piece[0] = GS_LN2 * (2.0f * cq.x * Sx - cq.y * Syt);
                piece[1] = GS_LN2 * (2.0f * cq.z * Syt - cq.y * Sx);
                piece[2] = iPn * (-Syy + 2.0f * d * Su);
                piece[3] = iPn * (Sxy - 2.0f * cc * Su);
            } else if (jq == 1) {
                piece[0] = iPn * (Sxy - 2.0f * bb * Su);
                piece[1] = iPn * (-Sxx + 2.0f * a * Su);
                piece[2] = Sopa;
                piece[3] = Sc0;
            } else {
                piece[0] = jq == 2 ? Sc1 : 0.f;
                piece[1] = jq == 2 ? Sc2 : 0.f;
                piece[2] = 0.f;
                piece[3] = 0.f;
            }
        }
        {
            // the group's 16 rows as ONE store instruction of whole 64-byte lines: destination lane 4 r + q takes piece q of
            // row r from lane (g = r, jq = q)
            const int dr = lane >> 2, dq = lane & 3, src = dr + 16 * dq;
            const uint32_t dslot = (uint32_t)__shfl((int)slot, dr, 64);
            float4 o;
            o.x = __shfl(piece[0], src, 64);
            o.y = __shfl(piece[1], src, 64);
            o.z = __shfl(piece[2], src, 64);
            o.w = __shfl(piece[3], src, 64);
            if (dslot != GS_NO_SLOT) reinterpret_cast<float4 *>(O.rows + (size_t)dslot * RW)[dq] = o;
        }
        lds_order();  // (the next group reads the states the lanes of Gaussian 15 wrote)
    }
    }  // work list
}

// sigmoid=True of the reference API: ceil buckets in the systolic kernel, no tail kernel (a rarely used flag)
template <int CDIM>
void launch_bwd_sig(const RasterSrc &S, const RasterGeom &G, const BwdIn &I, const BwdOut &O, int64_t max_buckets,
                    hipStream_t stream, int exact) {
    constexpr int WPB = BwdCfg<CDIM>::WPB;
    const int grid = (int)gs_div_up(max_buckets > 0 ? max_buckets : 1, WPB);
    if (exact)
        hipLaunchKernelGGL((raster_backward_kernel<CDIM, false, true, true>), dim3(grid), dim3(64 * WPB), 0, stream, S, G, I, O);
    else
        hipLaunchKernelGGL((raster_backward_kernel<CDIM, false, true>), dim3(grid), dim3(64 * WPB), 0, stream, S, G, I, O);
}

#ifndef GS_BWD_SH_PIXEL
#define GS_BWD_SH_PIXEL 1  // 0: the systolic kernel for SH as well (A/B switch for tools/ab_variants.py)
#endif
#ifndef GS_BWD_PACKED_RGB
#define GS_BWD_PACKED_RGB 1  // 0: the first, unpacked pixel-parallel kernel for rgb colours (A/B switch)
#endif
template <int CDIM, bool FRAME>
void launch_bwd(const RasterSrc &S, const RasterGeom &G, const BwdIn &I, const BwdOut &O, int64_t max_buckets,
                hipStream_t stream) {
    constexpr int WPB = BwdCfg<CDIM>::WPB;
    const int grid = (int)gs_div_up(max_buckets > 0 ? max_buckets : 1, WPB);
    // Grid of the one-wave-per-bucket kernels of the frame path: the work list's capacity, but at most GS_BWD_GRID_CAP waves.
    // The kernels walk the list with the grid as the stride, so the cap only decides how many waves come up empty (a frame
    // whose tiles saturate early fills a third of the capacity) or take a second bucket (a frame beyond the cap: the waves
    // are still 8 x the device's resident slots, i.e. fresh waves keep arriving while others are in their load phase -- what
    // the measured-and-dropped persistent grids of round 2 lacked).
#ifndef GS_BWD_GRID_CAP
#define GS_BWD_GRID_CAP 40960
#endif
    const unsigned fgrid = (unsigned)(max_buckets > 0 ? (max_buckets < GS_BWD_GRID_CAP ? max_buckets : GS_BWD_GRID_CAP) : 1);
    if constexpr (FRAME && ((CDIM == 48 && GS_BWD_SH_MFMA >= 1) || (CDIM == 27 && GS_BWD_SH_MFMA >= 2))) {
        // one workgroup per GS_BWD_MFMA_TILES tiles (tiles nothing was composited in leave at once)
        const unsigned mgrid = gs_bwd_mfma_grid(G.ntx * G.nty, max_buckets);
        const int mwaves = GS_BWD_MFMA_WAVES ? GS_BWD_MFMA_WAVES : (G.ntx * G.nty >= 1024 ? 2 : 4);
        static_assert((GS_BWD_MFMA_TILES == 1 ? GS_BWD_MFMA_SPLIT : 1) * 4 <= GS_BWD_EXEC_SLOTS, "executed-row counters");
        if (mwaves == 2)
            hipLaunchKernelGGL((raster_backward_mfma_sh_kernel<CDIM, 2, GS_BWD_MFMA_TILES>), dim3(mgrid), dim3(128), 0,
                               stream, S, G, I, O);
        else
            hipLaunchKernelGGL((raster_backward_mfma_sh_kernel<CDIM, 4, GS_BWD_MFMA_TILES>), dim3(mgrid), dim3(256), 0,
                               stream, S, G, I, O);
        // A workgroup walks its tile's buckets four at a time: a 100,000-Gaussian pile in one tile (a degenerate
        // densification run) would keep ONE workgroup busy for 26 ms.  Frames the caller has flagged for long lists
        // (GS_FRAME_LONG_LISTS, as for the forward's long-list kernels) leave a tile's buckets beyond the first 32 -- the
        // part of a list beyond the forward's own 2,048 -- to the one-wave-per-bucket kernel: spread over the device.
        if (I.bucket_first) {
            BwdIn I2 = I;
            I2.bucket_cap = 0;
            hipLaunchKernelGGL((raster_backward_pixel_sh_kernel<CDIM, FRAME>), dim3(fgrid), dim3(64), 0, stream, S, G, I2, O);
        }
        return;
    }
    if constexpr (FRAME && CDIM == 3 && GS_BWD_RGB_ROWS) {
        if (GS_BWD_RGB_ROWS == 1 || I.use_rows) {
            hipLaunchKernelGGL(raster_backward_rows_kernel, dim3(fgrid), dim3(64), 0, stream, S, G, I, O);
            return;
        }
    }
    if (CDIM == 3 && !GS_BWD_PACKED_RGB) {
        const int64_t blocks = gs_div_up(max_buckets > 0 ? max_buckets : 1, GS_PP_WPB);
        hipLaunchKernelGGL((raster_backward_pixel_kernel<FRAME>), dim3((unsigned)blocks), dim3(64 * GS_PP_WPB), 0, stream,
                           S, G, I, O);
    } else if (CDIM == 3 || GS_BWD_SH_PIXEL) {
        hipLaunchKernelGGL((raster_backward_pixel_sh_kernel<CDIM, FRAME>),
                           dim3(FRAME ? fgrid : (unsigned)(max_buckets > 0 ? max_buckets : 1)), dim3(64), 0, stream, S, G, I, O);
    } else {
        hipLaunchKernelGGL((raster_backward_kernel<CDIM, FRAME>), dim3(grid), dim3(64 * WPB), 0, stream, S, G, I, O);
    }
}

}  // namespace

// (workgroup, wave) slots the SH backward on the matrix pipe writes its executed-row counters to (0: this colour model
// takes another kernel): the launch geometry of launch_bwd above
int gs_bwd_mfma_slots(int color_dim, int n_tiles, int64_t max_buckets) {
    const bool mfma = (color_dim == 48 && GS_BWD_SH_MFMA >= 1) || (color_dim == 27 && GS_BWD_SH_MFMA >= 2);
    if (!mfma) return 0;
    const int mwaves = GS_BWD_MFMA_WAVES ? GS_BWD_MFMA_WAVES : (n_tiles >= 1024 ? 2 : 4);
    return (int)gs_bwd_mfma_grid(n_tiles, max_buckets) * mwaves;
}

namespace {

struct RefWs {
    uint32_t *tile_nproc, *bucket_offsets;
    uint4 *bucket_info;
    unsigned long long *n_buckets;
    float4 *ckpt;
    int64_t max_buckets;
    size_t bytes;
};
RefWs carve_ref(void *base, int64_t M, int32_t h, int32_t w) {
    RefWs r;
    const int n_tiles = (w / 16) * (h / 16);
    size_t off = 0;
    auto take = [&](size_t bytes) -> void * {
        void *p = base ? (void *)((char *)base + off) : nullptr;
        off += gs_align_up(bytes ? bytes : 1, 256);
        return p;
    };
    r.max_buckets = gs_max_buckets(M, n_tiles);
    r.tile_nproc = (uint32_t *)take(sizeof(uint32_t) * n_tiles);
    r.bucket_offsets = (uint32_t *)take(sizeof(uint32_t) * (n_tiles + 1));
    r.bucket_info = (uint4 *)take(sizeof(uint4) * (size_t)(r.max_buckets + 8));
    r.n_buckets = (unsigned long long *)take(sizeof(unsigned long long));
    r.ckpt = (float4 *)take(sizeof(float4) * 256 * (size_t)r.max_buckets);
    r.bytes = off;
    return r;
}

}  // namespace

extern "C" size_t gs_draw_backward_workspace_bytes(int64_t M, int32_t h, int32_t w) {
    if (M < 0 || h <= 0 || w <= 0) return 0;
    return carve_ref(nullptr, M, h, w).bytes;
}

extern "C" int gs_draw_backward(const float *pos, const float *rgb, const float *opa, const float *cov,
                                const int32_t *tile_n_point_accum, const float *output, const float *grad_output,
                                float *grad_pos, float *grad_rgb, float *grad_opa, float *grad_cov, int32_t h,
                                int32_t w, int64_t M, float focal_x, float focal_y, int weight_normalize,
                                int sigmoid, int fast, const float *rays_o, const float *lefttop_pos,
                                const float *vec_dx, const float *vec_dy, int use_sh_coeff, void *workspace,
                                size_t workspace_bytes, gs_stream_t stream) {
    const int exact = fast ? 0 : 1;  // fast = 0: the reference's exp() flavour (gaussian.cu:596-603, 922-923)
    (void)weight_normalize;  // the reference backward ignores it as well (gaussian.cu:440-803)
    GS_CHECK_ARG(h > 0 && w > 0 && (h % 16) == 0 && (w % 16) == 0, "h, w must be positive multiples of 16");
    GS_CHECK_ARG(M >= 0, "M < 0");
    if (M == 0) return 0;
    GS_CHECK_ARG(pos && rgb && opa && cov && tile_n_point_accum && output && grad_output && grad_pos && grad_rgb &&
                     grad_opa && grad_cov,
                 "null pointer");
    GS_CHECK_ARG(((uintptr_t)cov & 15) == 0 && ((uintptr_t)grad_cov & 15) == 0, "cov/grad_cov must be 16-byte aligned");
    GS_CHECK_ARG(workspace && workspace_bytes >= gs_draw_backward_workspace_bytes(M, h, w), "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    RefWs ws = carve_ref(workspace, M, h, w);
    RasterSrc S = {};
    S.pos = pos;
    S.rgb = rgb;
    S.opa = opa;
    S.cov = cov;
    RasterGeom G = {};
    G.padW = w;
    G.padH = h;
    G.ntx = w / 16;
    G.nty = h / 16;
    G.focal_x = focal_x;
    G.focal_y = focal_y;
    if (use_sh_coeff) {
        GS_CHECK_ARG(rays_o && lefttop_pos && vec_dx && vec_dy, "SH needs the ray basis");
        // the basis stays on the device: the kernels load it (raster_common.h), the call never synchronises
        G.dev_rays_o = rays_o;
        G.dev_lefttop = lefttop_pos;
        G.dev_vdx = vec_dx;
        G.dev_vdy = vec_dy;
    }
    // 1. replay the forward to checkpoint (T, C_run) at every bucket boundary
    int rc = gs_raster_forward_ref(S, G, tile_n_point_accum, nullptr, use_sh_coeff, sigmoid, 0, ws.ckpt,
                                   ws.tile_nproc, s, exact);
    if (rc) return rc;
    // 2. bucket work list
    gs_launch_bucket_list(ws.tile_nproc, G.ntx * G.nty, ws.bucket_offsets, ws.n_buckets, ws.bucket_info, tile_n_point_accum, 0, s);
    // 3. one wave per bucket, one output row per pair
    BwdIn I = {output, grad_output, ws.ckpt, ws.tile_nproc, ws.bucket_offsets, ws.bucket_info, tile_n_point_accum};
    BwdOut O = {nullptr, nullptr, nullptr, nullptr, 0, grad_pos, grad_rgb, grad_opa, grad_cov};  // (reference API: no rows)
    const unsigned nb = (unsigned)(ws.max_buckets > 0 ? ws.max_buckets : 1);
    if (sigmoid && use_sh_coeff)
        launch_bwd_sig<27>(S, G, I, O, ws.max_buckets, s, exact);
    else if (sigmoid)
        launch_bwd_sig<3>(S, G, I, O, ws.max_buckets, s, exact);
    else if (exact && use_sh_coeff)
        hipLaunchKernelGGL((raster_backward_pixel_sh_kernel<27, false, true>), dim3(nb), dim3(64), 0, s, S, G, I, O);
    else if (exact)
        hipLaunchKernelGGL((raster_backward_pixel_sh_kernel<3, false, true>), dim3(nb), dim3(64), 0, s, S, G, I, O);
    else if (use_sh_coeff)
        launch_bwd<27, false>(S, G, I, O, ws.max_buckets, s);
    else
        launch_bwd<3, false>(S, G, I, O, ws.max_buckets, s);
    GS_CHECK_LAUNCH();
    return 0;
}

// What the backward needs from the forward alone: which rows will exist -- rows of pairs behind a tile's
// early-termination point are never written, and the reader must skip them; the rows themselves, 64 to 224 bytes per
// pair, are not zero-filled (1.1 GB per frame at 2.4 M Gaussians with SH).  SH rows: one flag byte per pair, cleared
// here and set by the writer; rgb rows (round 4): nothing per pair at all, one "stop key" per TILE -- and the bucket
// work list.  gs_frame_forward runs it on a side stream underneath
// whatever the caller does between forward and backward (the loss); gs_frame_backward runs it inline otherwise.
int gs_stage_backward_prepare(const gs_frame *f, const gs_frame_ws &ws, const uint32_t *sorted_ids, hipStream_t stream) {
    gs_frame_geom FG = gs_frame_geometry(f);
    // no row carries a flag: the readers decide from the tiles' stop keys which rows exist (rgb: round 4; SH: round 5 --
    // until then a flag byte per pair, set by scattered one-byte stores = a 32-byte write each, cleared by a memset here)
    hipLaunchKernelGGL(stop_key_kernel, dim3((unsigned)gs_div_up(FG.n_tiles, 256)), dim3(256), 0, stream, ws.tile_nproc,
                       FG.n_tiles, ws.tile_ranges, sorted_ids, ws.rects, (unsigned long long *)ws.stop_keys);
    // the bucket work list is what the one-wave-per-bucket kernels read; the SH backward on the matrix pipe walks a
    // tile's buckets itself (one workgroup per tile) and needs none
    const bool per_tile = ((f->color_dim == 48 && GS_BWD_SH_MFMA >= 1) || (f->color_dim == 27 && GS_BWD_SH_MFMA >= 2)) &&
                          (GS_MFMA_ITEMS || !gs_frame_long_lists(f, FG.n_tiles));  // (no items + flagged frame: the buckets
                                                                                    // beyond a tile's first 32: hand-over)
    if (!per_tile)
        gs_launch_bucket_list(ws.tile_nproc, FG.n_tiles, ws.bucket_offsets, ws.counters + GS_CNT_BUCKETS, ws.bucket_info,
                              ws.tile_ranges, 1, stream);
    const bool mfma_frame = (f->color_dim == 48 && GS_BWD_SH_MFMA >= 1) || (f->color_dim == 27 && GS_BWD_SH_MFMA >= 2);
    if (mfma_frame && GS_MFMA_ITEMS) {  // the matrix-pipe kernel's work items, heavy tiles first where the frame has an order
        const uint32_t cap = 0u;  // (work items spread a long list over the device: no hand-over, see gs_stage_raster_backward)
        hipLaunchKernelGGL(mfma_items_kernel, dim3(1), dim3(1024), 0, stream, ws.tile_nproc, FG.n_tiles,
                           (gs_frame_uses_strips(f) && f->N > 0 && GS_BWD_MFMA_ORDER) ? ws.tile_order : nullptr, cap,
                           (uint32_t)GS_BWD_MFMA_CHUNK, ws.mfma_items, ws.mfma_n_items);
    }
    GS_CHECK_LAUNCH();
    return 0;
}

int gs_stage_raster_backward(const gs_frame *f, const gs_frame_ws &ws, const uint32_t *sorted_ids,
                             const float *grad_image, hipStream_t stream, bool prepared) {
    gs_frame_geom FG = gs_frame_geometry(f);
    RasterSrc S = {};
    S.ids = sorted_ids;
    S.geom = ws.rec_geom;
    S.cov4 = ws.rec_cov;
    S.color4 = ws.rec_color;
    S.conic4 = ws.rec_conic;
    S.sh = f->rgb;
    RasterGeom G = {};
    G.padW = FG.padW;
    G.padH = FG.padH;
    G.ntx = FG.ntx;
    G.nty = FG.nty;
    G.width = f->width;
    G.height = f->height;
    G.crop_top = FG.crop_top;
    G.crop_left = FG.crop_left;
    G.focal_x = f->focal_x;
    G.focal_y = f->focal_y;
    for (int i = 0; i < 3; ++i) {
        G.rays_o[i] = f->rays_o[i];
        G.lefttop[i] = f->lefttop[i];
        G.vdx[i] = f->vec_dx[i];
        G.vdy[i] = f->vec_dy[i];
    }
    if (!prepared) {
        const int rc = gs_stage_backward_prepare(f, ws, sorted_ids, stream);
        if (rc) return rc;
    }
    BwdIn I = {f->image_padded, grad_image, ws.ckpt, ws.tile_nproc, ws.bucket_offsets, ws.bucket_info, ws.tile_ranges,
               (gs_frame_uses_strips(f) && f->N > 0 && GS_BWD_MFMA_ORDER) ? ws.tile_order : nullptr, 0, 0,
               (f->flags & GS_FRAME_BWD_ROWS) ? 1u : 0u, GS_MFMA_ITEMS ? ws.mfma_items : nullptr,
               GS_MFMA_ITEMS ? ws.mfma_n_items : nullptr};
    // One workgroup per TILE (GS_BWD_MFMA_CHUNK = 0, round 4) walks a 100,000-Gaussian pile alone: frames flagged for long lists
    // then leave a tile's buckets beyond the first GS_BWD_SH_HANDOVER to the one-wave-per-bucket kernel.  With work items
    // (round 5) a long list is spread over the device anyway: no hand-over, the matrix-pipe kernel takes every bucket.
#ifndef GS_BWD_SH_HANDOVER
#define GS_BWD_SH_HANDOVER 32
#endif
    if (!GS_MFMA_ITEMS && f->color_dim != 3 && gs_frame_long_lists(f, FG.n_tiles))
        I.bucket_cap = I.bucket_first = GS_BWD_SH_HANDOVER;
    BwdOut O = {ws.rows, ws.bwd_exec_rows, ws.pair_offsets, ws.rects, (uint64_t)f->max_pairs, nullptr, nullptr, nullptr, nullptr};
    if (f->color_dim == 48)
        launch_bwd<48, true>(S, G, I, O, ws.max_buckets, stream);
    else if (f->color_dim == 27)
        launch_bwd<27, true>(S, G, I, O, ws.max_buckets, stream);
    else
        launch_bwd<3, true>(S, G, I, O, ws.max_buckets, stream);
    GS_CHECK_LAUNCH();
    // SH rows of Gaussians that cover hundreds of tiles are summed here, once, by whole workgroups (cull_project.hip)
    return gs_stage_sh_big_rows(f, ws, stream);
}
