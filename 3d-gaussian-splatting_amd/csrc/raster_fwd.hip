// raster_fwd.hip -- 16x16-tile front-to-back alpha compositing, forward.
//
// Replaces draw_kernel (gaussian.cu:806-970).  One wave64 per tile, four pixels per lane.  The tile's sorted
// Gaussian list is streamed through LDS in chunks with a two-deep ring: every thread gathers
// one Gaussian of the NEXT chunk from HBM/L2 into registers while the waves composite the
// current chunk out of LDS (broadcast ds_read_b128), so a chunk costs a single barrier.
// Differences from the reference kernel, all numerically neutral at fp32 tolerance:
//   * the per-pixel fp64 division by (2 det + 1e-14) is hoisted to once per (tile, Gaussian)
//     and the exponent is pre-scaled by log2(e) so the pixel loop issues one v_exp_f32;
//   * early termination is wave-uniform (ballot) with exact per-pixel masking, and the
//     inter-chunk shared-memory race of the reference (no barrier after its compute loop,
//     gaussian.cu:878-962) cannot occur: the ring buffer is only rewritten one barrier later;
//   * in training mode the kernel also stores the per-pixel state (T, C) at every 64-Gaussian
//     bucket boundary; the backward pass uses these checkpoints to process buckets
//     independently (raster_bwd.hip).
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "raster_common.h"

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int FWD_THREADS = 64;  // ONE wave64 per 16x16 tile, 4 pixels per lane
constexpr int NPP = 2;           // pixel pairs per lane
#ifndef GS_FWD_WAVES_PER_SIMD
#define GS_FWD_WAVES_PER_SIMD 8
#endif

template <int CDIM>
struct FwdSmem {  // SH: CDIM = 27 (degree 2) or 48 (degree 3) raw coefficients per Gaussian, channel-major
    static constexpr int CH = 64;
    static constexpr int NB = CDIM / 3;
    static constexpr int SHS = CDIM == 27 ? 28 : CDIM + 4;  // record stride: 16-byte rows, 4-way conflicts on the fill
    enum { X, Y, A, B, C, NLOP, DH, DL, NFIELD };  // DH, DL: exact-exp flavour only (see FwdSmem<3>)
    // ONE buffer: the single wave stages chunk k + 1 after it has composited chunk k (LDS operations of a wave execute in
    // order), and the coefficients are loaded in the iteration that consumes them anyway; a second buffer only halved
    // the waves per SIMD (17 / 30 KiB per wave: 2 / 1 waves per SIMD for degree 2 / 3)
    static constexpr int NBUF = 1;
    float f[NBUF][NFIELD][CH] __attribute__((aligned(16)));
    float sh[NBUF][CH][SHS] __attribute__((aligned(16)));
};
template <>
struct FwdSmem<3> {
    // structure of arrays: four consecutive Gaussians of one field are one ds_read_b128
    static constexpr int CH = 64;
    // NLOP = -log2(opacity) (frame path) or the opacity (reference API).  Exact-exp flavour of the reference API
    // (`fast = 0`, gaussian.cu:922-923): A, B, C hold the raw covariance entries a, b + c, d and DH + DL the double
    // 2 det + 1e-14 as two floats (48 of its 53 bits)
    enum { X, Y, A, B, C, NLOP, R, G, BL, DH, DL, NFIELD };
    static constexpr int NBUF = 2;
    float f[NBUF][NFIELD][CH] __attribute__((aligned(16)));
};

__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

#ifndef GS_FWD_LEGACY_MUL
#define GS_FWD_LEGACY_MUL 1  // 0: plain packed multiply (A/B switch for tools/ab_variants.py)
#endif
// a * b with DX9 zero rules: 0 x anything (NaN, infinity) = 0 (v_mul_legacy_f32).  Declared as the LLVM intrinsic, NOT
// as inline assembly: the operand is the result of a transcendental (v_exp_f32), which on gfx940/950 needs a wait state
// before a VALU instruction may read it -- the compiler inserts it for instructions it knows, not for opaque asm text
// (first version: stale alphas in a few pixels per frame).
extern "C" __device__ float gs_fmul_legacy(float, float) __asm("llvm.amdgcn.fmul.legacy");
__device__ __forceinline__ float mul_legacy(float a, float b) { return gs_fmul_legacy(a, b); }

// Exact per-pixel liveness in ONE packed instruction: m = clamp((t - 0.0001f) * 2^100) is 1.0 when
// t > 0.0001f and 0.0 otherwise (the fma is exact up to its final rounding, and the smallest positive
// t - 0.0001f is one ulp of 1e-4, which 2^100 lifts far above 1).  Replaces v_cmp + v_cndmask per pixel.
#define GS_LIVE_SCALE 1.2676506002282294e30f /* 2^100 */
__device__ __forceinline__ f2 live_mask(f2 t, f2 scale, f2 bias) {
    f2 m;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(m) : "v"(t), "v"(scale), "v"(bias));
    return m;
}

// Lane l owns the four pixels (x, y0 + 4k), k = 0..3, of the tile, x = l & 15, y0 = l >> 4; pixel
// index inside the tile p_k = 64 k + l.  The compositing loop is bound by fp32 VALU ISSUE (PMC: ~31 VALU
// wave-instructions per Gaussian step of 256 pixels, 28 of them arithmetic; the same instruction stream fed from
// registers without any memory access runs at the same speed, tools/ubench/raster_feed.hip), so the layout is chosen
// to share work between pixels: dx and the three dx-only terms of the exponent are computed once per lane for four
// pixels, the per-Gaussian operands are read from LDS once per lane (broadcast ds_read_b128, four Gaussians per
// read) and the fp32 math is packed over pixel PAIRS (v_pk_*_f32).  With a single wave per tile there is no workgroup
// barrier at all; the staging of the next chunk (one Gaussian per lane) overlaps the compositing of the current one
// through registers.
// The kernel is PERSISTENT: the grid is a few waves per SIMD and every wave walks tiles blockIdx.x,
// blockIdx.x + gridDim.x, ...  While a wave composites the LAST chunk of its tile it already gathers
// chunk 0 of its NEXT tile, so the dependent id -> record gather latency (and the per-tile range load)
// sits behind math instead of in front of it.  With one tile per wave all waves run their gather,
// math and store phases in lockstep and the phases add up instead of overlapping.
#ifndef GS_FWD_RGB_WPE
#define GS_FWD_RGB_WPE 4  // rgb colours: four waves per SIMD asked for.  Left to itself the allocator gave the training variant
                         // (checkpoint stores) 130 VGPRs = three waves; with the hint it fits 128 without scratch: training
                         // forward at 2.4 M Gaussians 136 -> 129 us, inference 126 VGPRs, +-1 % (same-box A/B, round 3)
#endif
#ifndef GS_FWD_SH27_WPE
#define GS_FWD_SH27_WPE 3  // waves per SIMD the register allocation of the SH kernels aims at (A/B switches; degree 2 at
                           // 3: 168 VGPRs and 22 spilled outside the loop, 0.535 -> 0.510 ms at 2.4 M Gaussians)
#endif
#ifndef GS_FWD_SH48_WPE
#define GS_FWD_SH48_WPE 2
#endif
template <int CDIM, bool FRAME, bool CKPT, bool SIG, bool WN, bool EXACT = false>
__global__ void __launch_bounds__(FWD_THREADS)
__attribute__((amdgpu_waves_per_eu(CDIM == 48 ? GS_FWD_SH48_WPE : CDIM == 27 ? GS_FWD_SH27_WPE
                                               : (SIG || WN || EXACT) ? 1 : GS_FWD_RGB_WPE)))  // (the rare flags would spill at 128)
raster_forward_kernel(RasterSrc S, RasterGeom G,
                                                                    const int32_t *__restrict__ ranges,
                                                                    float *__restrict__ out_padded,
                                                                    float *__restrict__ out_image,
                                                                    float4 *__restrict__ ckpt,
                                                                    uint32_t *__restrict__ tile_nproc,
                                                                    uint32_t n_tiles,
                                                                    float4 *__restrict__ cont_state,
                                                                    uint32_t *__restrict__ cont_flag,
                                                                    const uint32_t *__restrict__ tile_order,
                                                                    uint32_t *__restrict__ tile_cost,
                                                                    const uint32_t *cut_in, uint32_t *cut_out,
                                                                    unsigned long long *__restrict__ ranpast,
                                                                    const unsigned long long *__restrict__ gate) {
    static_assert(!(EXACT && FRAME), "the exact-exp flavour belongs to the reference API (gs_draw, fast = 0)");
    // frame path, temporal occlusion cull (gs_frame_layout.h): `cut_out` [T] receives, per tile, the depth behind which this
    // launch composited nothing (all its pixels had stopped) or GS_NO_CUT; `cut_in` (the same table, set when this frame's
    // lists were trimmed by it) makes a tile that reaches the end of its list with a live pixel raise *ranpast; `gate`: the
    // untrimmed second pass, which only runs if some tile did
    if (FRAME && gate && *gate == 0) return;
    using SM = FwdSmem<CDIM>;
    constexpr int CH = SM::CH;
#ifndef GS_FWD_GROUP
#define GS_FWD_GROUP 4
#endif
    // Gaussians per LDS read of a field (ds_read_b128: 4, ds_read_b64: 2 -- half the operand registers; A/B switch)
    constexpr uint32_t GROUP = CDIM == 3 ? GS_FWD_GROUP : 4;
    static_assert(GROUP == 4 || GROUP == 2, "GROUP");
    // the wave-uniform liveness test (3 VALU instructions + a branch) runs before every second group without SH;
    // with SH a Gaussian step is an order of magnitude longer and every group is tested
#ifndef GS_FWD_LIVE_EVERY
#define GS_FWD_LIVE_EVERY 8
#endif
    constexpr uint32_t LIVE_EVERY = CDIM == 3 ? GS_FWD_LIVE_EVERY : 4;
    __shared__ SM sm;
    const int lane = threadIdx.x;

    auto range_of = [&](uint32_t t, uint32_t &s0, uint32_t &cnt) {
        s0 = cnt = 0;
        if (t < n_tiles) {
            s0 = (uint32_t)(FRAME ? ranges[2 * t] : ranges[t]);
            cnt = (uint32_t)(FRAME ? ranges[2 * t + 1] : ranges[t + 1]) - s0;
        }
    };
    // Dispatch order (frame path, strip variant): wave k of the grid takes tile_order[k] -- the tiles in descending order
    // of what they cost in the previous frame of this workspace (strip_bin.hip, tile_order_workgroup), so that the long
    // tiles start first and the kernel's tail is made of short ones.  Measured at 2.4 M Gaussians (profiles/r03_a):
    // 146 -> 132 us.  A persistent grid (3 / 4 / 8 waves per SIMD) drawing positions of the same order from a device
    // ticket was also built and measured: 197 - 236 us -- the hardware's workgroup dispatcher refills a slot faster than
    // a resident wave can fetch a ticket, the order entry, the range and the first chunk one after the other.
    auto tile_at = [&](uint32_t li) {  // wave-uniform
        return li < n_tiles ? (tile_order ? (uint32_t)__builtin_amdgcn_readfirstlane(tile_order[li]) : li) : n_tiles;
    };
    auto next_index = [&](uint32_t li) { return li + gridDim.x; };
    uint32_t li = blockIdx.x, nli = next_index(li);
    uint32_t tile = tile_at(li), ntile = tile_at(nli), start, n, nstart, nn;
    range_of(tile, start, n);
    range_of(ntile, nstart, nn);

    // register stage for the next chunk (one Gaussian per lane)
    GaussianRec g;
    float r0 = 0, r1 = 0, r2 = 0;
    float4 cq = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t gid = 0, gj = 0;
    bool have = false;
    auto fetch = [&](uint32_t s0, uint32_t cnt, uint32_t base) {
        have = base + lane < cnt;
        if (have) {
            gj = s0 + base + lane;
            gid = raster_load<FRAME>(S, gj, g);
            if (FRAME) cq = S.conic4[(size_t)gid * GS_REC_STRIDE];
            if (CDIM == 3) raster_load_rgb<FRAME>(S, gj, gid, r0, r1, r2);
        }
    };
    fetch(start, n, 0);
    int k = 0;  // chunk counter across tiles: LDS ring slot

    for (; li < n_tiles; li = nli, nli = next_index(nli), tile = ntile, ntile = tile_at(nli), start = nstart, n = nn,
                         range_of(ntile, nstart, nn)) {
    const uint32_t tx = tile % (uint32_t)G.ntx, ty = tile / (uint32_t)G.ntx;
    const uint32_t id_x = tx * 16 + (lane & 15), id_y0 = ty * 16 + (lane >> 4);
    const float px = raster_pixel_coord(id_x, G.padW, G.focal_x);
    f2 py2[NPP];
#pragma unroll
    for (int h = 0; h < NPP; ++h)
        py2[h] = f2{raster_pixel_coord(id_y0 + 8 * h, G.padH, G.focal_y),
                    raster_pixel_coord(id_y0 + 8 * h + 4, G.padH, G.focal_y)};
    constexpr int NB = CDIM > 3 ? CDIM / 3 : 1;  // SH basis functions per channel
    f2 SH[NPP][NB];
    if (CDIM > 3) {
#pragma unroll
        for (int h = 0; h < NPP; ++h) {
            float a9[NB], b9[NB];
            raster_pixel_sh<NB>(id_x, id_y0 + 8 * h, G, a9);
            raster_pixel_sh<NB>(id_x, id_y0 + 8 * h + 4, G, b9);
            constexpr float KS = GS_SH_PRESCALE ? -GS_LOG2E : 1.0f;  // raster_common.h
#pragma unroll
            for (int k9 = 0; k9 < NB; ++k9) SH[h][k9] = f2{KS * a9[k9], KS * b9[k9]};
        }
    }

    // pair h holds rows y0 + 8h (x) and y0 + 8h + 4 (y): tile pixel indices 128h + lane, 128h + 64 + lane.
    // T is the MASKED transmittance: it drops to exactly 0 once the pixel's transmittance is <= 0.0001
    // (the reference's per-pixel `accum < 0.0001` stop, gaussian.cu:906), so a finished pixel adds nothing.
    f2 T[NPP], cr[NPP], cg[NPP], cb[NPP], accw[NPP];
#pragma unroll
    for (int h = 0; h < NPP; ++h) {
        T[h] = f2{1.0f, 1.0f};
        cr[h] = cg[h] = cb[h] = accw[h] = f2{0.f, 0.f};
    }
    f2 live_scale = splat(GS_LIVE_SCALE), live_bias = splat(-GS_T_STOP * GS_LIVE_SCALE);
    // opaque to the compiler: otherwise it keeps the two constants in SGPRs and copies them into VGPR pairs for the
    // inline v_pk_fma inside the hot loop (two v_mov_b64 per four Gaussians)
    asm volatile("" : "+v"(live_scale), "+v"(live_bias));
    uint32_t nproc = 0, steps = 0;  // steps: Gaussians composited before the wave stopped (the tile's cost)

    auto write_ckpt = [&](uint32_t idx_in_tile) {
        float4 *c = ckpt + raster_ckpt_slot(start, tile, idx_in_tile / GS_BUCKET) * 256;
#pragma unroll
        for (int h = 0; h < NPP; ++h) {
            c[128 * h + lane] = make_float4(T[h].x, cr[h].x, cg[h].x, cb[h].x);
            c[128 * h + 64 + lane] = make_float4(T[h].y, cr[h].y, cg[h].y, cb[h].y);
        }
    };
    auto any_live = [&]() {
        f2 sum = T[0];
#pragma unroll
        for (int h = 1; h < NPP; ++h) sum += T[h];
        return __ballot(sum.x + sum.y > 0.0f) != 0ull;
    };

    bool done = false, staged_next = false;  // staged_next: the register stage holds chunk 0 of the NEXT tile
    bool continues = false;  // dense frames: the rest of a long list goes to the segment kernels (below)
    for (uint32_t base = 0; base < n && !done; base += CH, ++k) {
        if (FRAME && cont_flag && base == (uint32_t)GS_LONG_MIN) {  // uniform
            if (any_live()) {
                // still alive after GS_LONG_MIN Gaussians (low-opacity pile-ups: a list of 100,000 would keep this
                // ONE wave busy for milliseconds): the pixels' state is saved and the rest of the list is composited in
                // segments by many waves (raster_segment_kernel); lists that saturate earlier never get here
                float4 *c = cont_state + (size_t)tile * 256;
#pragma unroll
                for (int h = 0; h < NPP; ++h) {
                    c[128 * h + lane] = make_float4(T[h].x, cr[h].x, cg[h].x, cb[h].x);
                    c[128 * h + 64 + lane] = make_float4(T[h].y, cr[h].y, cg[h].y, cb[h].y);
                }
                continues = true;
            }
            if (continues) break;
        }
        const int buf = k & (SM::NBUF - 1);  // no SH: two-deep ring (the wave is in program order, so writing buffer
                                             // k & 1 here cannot overtake its own reads of two chunks ago)
        // Can a Gaussian of this chunk yield a non-finite alpha (or colour) for some pixel?  Not if its record is
        // finite and its conic positive semi-definite and of ordinary size: then q >= -(rounding of a few hundred) and
        // alpha = 2^-(q + nlop) stays finite (|dx|, |dy| < 4: visible Gaussians lie inside the guard band).  Only chunks
        // with a Gaussian that fails the test need the DX9 multiply below (wave-uniform: one ballot per 64 Gaussians).
        bool suspicious = false;
        if (have) {
            float A, B, C;
            if (FRAME) {  // S1 stored the conic in the Gaussian's record (same line as geom / colour)
                A = cq.x;
                B = cq.y;
                C = cq.z;
            } else {
                raster_conic(g, A, B, C);
            }
            {
                const float chk = ((g.x + g.y) + (A + B + C)) + ((r0 + r1 + r2) + g.opa);  // NaN / inf propagate
                suspicious = !(fabsf(chk) < 3.0e38f) || !(A >= 0.f && A < 1.0e8f) || !(C >= 0.f && C < 1.0e8f) ||
                             !(4.f * A * C >= B * B) ||
                             !FRAME || CDIM != 3;
            }
            float opa = g.opa;
            if (SIG)  // gaussian.cu:918: (1.0/2*3.1415926536) * rsqrtf(det + 1e-7), folded into opacity
                opa *= 1.5707963268f * rsqrtf(raster_det(g.a, g.b, g.c, g.d) + 1e-7f);
            sm.f[buf][SM::X][lane] = g.x;
            sm.f[buf][SM::Y][lane] = g.y;
            if constexpr (EXACT) {
                // gaussian.cu:916-923: the exponent's argument is a float numerator over the DOUBLE 2 det + 1e-14
                const double den = (double)(2.0f * raster_det(g.a, g.b, g.c, g.d)) + 1e-14;
                const float dh = (float)den;
                sm.f[buf][SM::A][lane] = g.a;
                sm.f[buf][SM::B][lane] = g.b + g.c;
                sm.f[buf][SM::C][lane] = g.d;
                sm.f[buf][SM::DH][lane] = dh;
                sm.f[buf][SM::DL][lane] = (float)(den - (double)dh);
            } else {
                sm.f[buf][SM::A][lane] = A;
                sm.f[buf][SM::B][lane] = B;
                sm.f[buf][SM::C][lane] = C;
            }
            // frame path: the opacity (a sigmoid, > 0) rides in the exponent, alpha = 2^-(q + nlop); the
            // reference API may be handed any opacity (zero, negative), so it keeps the multiplication
            sm.f[buf][SM::NLOP][lane] = FRAME ? -__log2f(opa) : opa;
            if constexpr (CDIM == 3) {
                // a NaN colour must reach the LIVE pixels only (the reference never evaluates the Gaussian for a
                // finished one); 0 x NaN in the colour FMAs would poison every pixel, so the NaN is moved into the
                // opacity, where v_mul_legacy confines it to live pixels, and the colour becomes 0 (0 x NaN = NaN there)
                if (r0 != r0 || r1 != r1 || r2 != r2) {
                    sm.f[buf][SM::NLOP][lane] = __builtin_nanf("");
                    r0 = r1 = r2 = 0.f;
                }
                sm.f[buf][SM::R][lane] = r0;
                sm.f[buf][SM::G][lane] = r1;
                sm.f[buf][SM::BL][lane] = r2;
            } else {
                const float *src = raster_sh_ptr<FRAME, CDIM>(S, gj, gid);
#pragma unroll
                for (int q = 0; q < CDIM; ++q) sm.sh[buf][lane][q] = src[q];
            }
        } else if (base + lane < ((n + (GROUP - 1u)) & ~(GROUP - 1u))) {
            // pad the ragged tail to a multiple of GROUP with null Gaussians (opacity 0 => alpha 0)
#pragma unroll
            for (int q = 0; q < SM::NFIELD; ++q)  // (exact flavour: denominator 1, so that the null Gaussian's alpha is 0 x exp(0))
                sm.f[buf][q][lane] = (FRAME && q == SM::NLOP) ? 1e30f : (EXACT && q == SM::DH) ? 1.0f : 0.f;
            if constexpr (CDIM > 3) {
#pragma unroll
                for (int q = 0; q < CDIM; ++q) sm.sh[buf][lane][q] = 0.f;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // overlaps with the compositing below: the next chunk of this tile, or chunk 0 of the next tile
        staged_next = base + CH >= n;
        if (staged_next)
            fetch(nstart, nn, 0);
        else
            fetch(start, n, base + CH);
        const uint32_t cnt = (n - base) < (uint32_t)CH ? (n - base) : (uint32_t)CH;
        // CH == GS_BUCKET: one checkpoint per chunk; the state before the tile's first chunk is (T, C) = (1, 0) by
        // definition, the backward kernels synthesise it instead of reading 4 KiB per tile back
        if (CKPT && base > 0) write_ckpt(base);
        // groups of GROUP Gaussians: a wave-uniform liveness test every LIVE_EVERY Gaussians, per-pixel masking inside
        // (exactly the reference's per-pixel `accum < 0.0001` test, gaussian.cu:906)
        const bool chunk_legacy = __ballot(suspicious) != 0ull;
        auto composite_chunk = [&](auto legacy_tag) {
        constexpr bool LEGACY = decltype(legacy_tag)::value;
        uint32_t stop = cnt;  // Gaussians of this chunk composited before the wave stopped
#pragma unroll 1
        for (uint32_t i = 0; i < cnt; i += GROUP) {
            if ((i & (LIVE_EVERY - 1)) == 0 && !any_live()) {
                done = true;
                stop = i;
                break;
            }
            {
            constexpr uint32_t i4 = 0;
            struct Vg {
                float v[GROUP];
            };
            auto ld4 = [&](int q) {
                Vg r;
                if constexpr (GROUP == 4) {
                    const float4 t = *(const float4 *)__builtin_assume_aligned(&sm.f[buf][q][i + i4], 16);
                    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
                } else {
                    const float2 t = *(const float2 *)__builtin_assume_aligned(&sm.f[buf][q][i + i4], 8);
                    r.v[0] = t.x; r.v[1] = t.y;
                }
                return r;
            };
            const Vg X = ld4(SM::X), Y = ld4(SM::Y), A4 = ld4(SM::A), B4 = ld4(SM::B), C4 = ld4(SM::C);
            const Vg O4 = ld4(SM::NLOP);
            Vg DH4 = {}, DL4 = {};
            if constexpr (EXACT) {
                DH4 = ld4(SM::DH);
                DL4 = ld4(SM::DL);
            }
            Vg R4 = {}, G4 = {}, L4 = {};
            if constexpr (CDIM == 3) {
                R4 = ld4(SM::R);
                G4 = ld4(SM::G);
                L4 = ld4(SM::BL);
            }
#pragma unroll
            for (int u = 0; u < (int)GROUP; ++u) {
                const float gx = X.v[u], gy = Y.v[u], cA = A4.v[u], cB = B4.v[u], cC = C4.v[u], nlop = O4.v[u];
                // q + nlop = (C dy - B dx) dy + (A dx^2 + nlop): dx is shared by the lane's four pixels
                const float dx = px - gx;
                const float bdx = cB * dx;
                const float f = FRAME ? fmaf(cA * dx, dx, nlop) : cA * dx * dx;
#pragma unroll
                for (int h = 0; h < NPP; ++h) {
                    const f2 dy = py2[h] - splat(gy);
                    f2 al;
                    if constexpr (EXACT) {
                        // exp(double(-(d x x - (b + c) x y + a y y)) / (2 det + 1e-14)) rounded to float: the reference's
                        // `fast = 0` flavour (gaussian.cu:922-923), float products in its order (no contraction), then
                        // the double division and the double exponential it pays per pixel as well
                        const double den = (double)DH4.v[u] + (double)DL4.v[u];
                        auto exact_alpha = [&](float y) {
                            const float t1 = __fmul_rn(__fmul_rn(cC, dx), dx);   // d x x
                            const float t2 = __fmul_rn(__fmul_rn(cB, dx), y);    // (b + c) x y
                            const float t3 = __fmul_rn(__fmul_rn(cA, y), y);     // a y y
                            const float num = -__fadd_rn(__fsub_rn(t1, t2), t3);
                            return (float)exp((double)num / den);
                        };
                        al.x = exact_alpha(dy.x);
                        al.y = exact_alpha(dy.y);
                    } else {
                        const f2 q = pk_fma(pk_fma(splat(cC), dy, splat(-bdx)), dy, splat(f));
                        al.x = gs_exp2(-q.x);
                        al.y = gs_exp2(-q.y);
                    }
                    if (!FRAME) al = al * splat(nlop);
                    if (SIG) {  // gaussian.cu:930
                        al.x = gs_squash_alpha(al.x);
                        al.y = gs_squash_alpha(al.y);
                    }
                    // w = alpha T with v_mul_legacy_f32 (0 x anything = 0): a finished pixel (T == 0) stays
                    // untouched whatever alpha is -- NaN, infinite --, like the reference, which `break`s before it
                    // evaluates the Gaussian (gaussian.cu:906); a live pixel sees the NaN, as there
                    f2 w;
                    if constexpr (LEGACY && GS_FWD_LEGACY_MUL)
                        w = f2{mul_legacy(al.x, T[h].x), mul_legacy(al.y, T[h].y)};
                    else
                        w = al * T[h];
                    if constexpr (CDIM == 3) {
                        cr[h] = pk_fma(splat(R4.v[u]), w, cr[h]);
                        cg[h] = pk_fma(splat(G4.v[u]), w, cg[h]);
                        cb[h] = pk_fma(splat(L4.v[u]), w, cb[h]);
                    } else {
                        const float *co = sm.sh[buf][i + i4 + u];
                        f2 v0 = {0.f, 0.f}, v1 = {0.f, 0.f}, v2 = {0.f, 0.f};
#pragma unroll
                        for (int k9 = 0; k9 < NB; ++k9) {
                            v0 = pk_fma(SH[h][k9], splat(co[k9]), v0);
                            v1 = pk_fma(SH[h][k9], splat(co[NB + k9]), v1);
                            v2 = pk_fma(SH[h][k9], splat(co[2 * NB + k9]), v2);
                        }
                        auto sg = [](float v) { return GS_SH_PRESCALE ? gs_rcp(1.0f + gs_exp2(v)) : gs_rcp(1.0f + __expf(-v)); };
                        const f2 c0 = {sg(v0.x), sg(v0.y)}, c1 = {sg(v1.x), sg(v1.y)}, c2 = {sg(v2.x), sg(v2.y)};
                        cr[h] = pk_fma(w, c0, cr[h]);
                        cg[h] = pk_fma(w, c1, cg[h]);
                        cb[h] = pk_fma(w, c2, cb[h]);
                    }
                    if (WN) accw[h] += w;
                    const f2 t = T[h] - w;  // T * (1 - alpha)
                    T[h] = t * live_mask(t, live_scale, live_bias);
                }
            }
            }
        }
        return stop;
        };
        steps = base + (chunk_legacy ? composite_chunk(std::true_type{}) : composite_chunk(std::false_type{}));
        // a chunk whose checkpoint was written counts as processed even if the wave stopped inside it:
        // the backward pass masks finished pixels by their transmittance
        nproc = base + cnt;
    }
    if (!staged_next) fetch(nstart, nn, 0);  // empty tile, or the wave stopped before its last chunk
    if (FRAME && !CKPT && cut_out) {  // (uniform; inference frames only: the training variant sits at its 128-VGPR budget)
        // all pixels stopped inside the list (or exactly at its end): nothing behind the last composited Gaussian matters
        const bool sat = !continues && steps > 0 && (done || !any_live());
        if (lane == 0) {
            uint32_t c = GS_NO_CUT;
            if (sat) {
                const uint32_t last = S.ids[start + steps - 1];
                c = __float_as_uint(S.geom[(size_t)last * GS_REC_STRIDE].z * (1.0f + GS_CUT_MARGIN));
            }
            if (cut_in && !sat && cut_in[tile] != GS_NO_CUT) *ranpast = 1ull;  // this tile's list had been trimmed
            cut_out[tile] = c;
        }
    }
    if (tile_nproc && lane == 0) tile_nproc[tile] = nproc;
    if (FRAME && tile_cost && lane == 0) tile_cost[tile] = steps;
    // statistics for the caller's long-list cost model (gs_frame.py): the longest walk a wave actually made -- a list's
    // LENGTH says nothing about a tile whose pixels stop early -- and the steps beyond the first GS_LONG_MIN of every walk
    // (what the segmented compositing would take over).  Rare (tiles beyond 512 / 1,024 steps), so the atomics cost nothing.
    if (FRAME && ranpast && lane == 0 && steps > (uint32_t)GS_LONG_MIN) {
        atomicAdd(ranpast + (GS_CNT_EXCESS_WALK - GS_CNT_RANPAST), (unsigned long long)(steps - (uint32_t)GS_LONG_MIN));
        if (steps > (uint32_t)GS_LONGEST_MIN) atomicMax(ranpast + (GS_CNT_MAXWALK - GS_CNT_RANPAST), (unsigned long long)steps);
    }
    if (FRAME && cont_flag && lane == 0) cont_flag[tile] = continues ? 1u : 0u;

#pragma unroll
    for (int h = 0; h < NPP; ++h)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const uint32_t id_y = id_y0 + 8 * h + 4 * e;
            float aw = e ? accw[h].y : accw[h].x;
            if (!WN || aw < 0.01f) aw = 1.0f;  // gaussian.cu:964-969
            const float o0 = (e ? cr[h].y : cr[h].x) / aw, o1 = (e ? cg[h].y : cg[h].x) / aw,
                        o2 = (e ? cb[h].y : cb[h].x) / aw;
            if (out_padded) {
                float *o = out_padded + ((size_t)id_y * G.padW + id_x) * 3;
                o[0] = o0;
                o[1] = o1;
                o[2] = o2;
            }
            if (out_image) {  // clamp + centred crop (splatter.py:652-653, 267-272)
                const int ox = (int)id_x - G.crop_left, oy = (int)id_y - G.crop_top;
                if (ox >= 0 && ox < G.width && oy >= 0 && oy < G.height) {
                    float *o = out_image + ((size_t)oy * G.width + ox) * 3;
                    // torch.clamp semantics: NaN stays NaN (fminf / fmaxf would turn it into 0)
                    o[0] = o0 > 1.f ? 1.f : (o0 < 0.f ? 0.f : o0);
                    o[1] = o1 > 1.f ? 1.f : (o1 < 0.f ? 0.f : o1);
                    o[2] = o2 > 1.f ? 1.f : (o2 < 0.f ? 0.f : o2);
                }
            }
        }
    }  // tile loop
}

// =================================================================================================================
// Segmented compositing of long tile lists (dense frames only).  A tile whose pixels are still alive after
// GS_LONG_MIN Gaussians leaves its state in cont_state and raises cont_flag (raster_forward_kernel above); the rest of
// its list, [GS_LONG_MIN, n), is cut into segments of GS_SEG_LEN Gaussians, one wave each:
//   seg_scan_kernel            items (tile, segment) of the flagged tiles, first item of every tile;
//   raster_segment_kernel<1>   P = the segment's transmittance product prod(1 - alpha) per pixel (no colour, no stop);
//   raster_segment_kernel<2>   T_in = T(GS_LONG_MIN) x the products of the tile's earlier segments; pixels with
//                              T_in <= 1e-4 are finished (the transmittance only falls, so the reference's stop lies in
//                              an earlier segment exactly then); the segment is composited from T_in with the same
//                              per-pixel stop as everywhere, colours from zero; checkpoints with the LOCAL colour;
//   seg_combine_kernel         per flagged tile: colour = saved colour + the segments' colours in list order, final
//                              pixel written like the main kernel's; every checkpoint of segment s gets the colour in
//                              front of s added; tile_nproc = GS_LONG_MIN + the segments' processed counts.
// Compositing is associative in (C, T) -- segments combine as C1 + T1 C2, T1 T2 -- so the only difference from the serial
// walk is the rounding of T_in (a product of products instead of a chain): a few ulp.  A list of 100,000 low-opacity
// Gaussians costs two passes over 2048 Gaussians per wave instead of one pass over 100,000 by ONE wave.
__global__ void __launch_bounds__(1024) seg_scan_kernel(const uint32_t *__restrict__ cont_flag,
                                                        const int32_t *__restrict__ ranges, uint32_t n_tiles,
                                                        uint32_t items_cap, uint32_t *__restrict__ item_base,
                                                        uint2 *__restrict__ items,
                                                        unsigned long long *__restrict__ n_items) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < n_tiles; base += 1024) {
        const uint32_t t = base + threadIdx.x;
        uint32_t ns = 0;
        if (t < n_tiles && cont_flag[t]) {
            const uint32_t n = (uint32_t)(ranges[2 * t + 1] - ranges[2 * t]);
            ns = (n - GS_LONG_MIN + GS_SEG_LEN - 1) / GS_SEG_LEN;
        }
        const uint32_t incl = gs_wave_incl_scan_u32(ns);
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t off = s_carry, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            off += w < wave ? s_wave[w] : 0;
            tot += s_wave[w];
        }
        const uint32_t first = off + incl - ns;
        if (t < n_tiles) {
            item_base[t] = first;
            for (uint32_t q = 0; q < ns && first + q < items_cap; ++q) items[first + q] = make_uint2(t, q);
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        item_base[n_tiles] = s_carry;
        *n_items = s_carry < items_cap ? s_carry : items_cap;  // never more than the buffers hold (M / SEG + T bounds it)
    }
}

template <int CDIM, bool CKPT, int PHASE>
__global__ void __launch_bounds__(FWD_THREADS) raster_segment_kernel(
    RasterSrc S, RasterGeom G, const int32_t *__restrict__ ranges, const uint2 *__restrict__ items,
    const unsigned long long *__restrict__ n_items, const float4 *__restrict__ cont_state, float *__restrict__ seg_P,
    float4 *__restrict__ seg_C, uint32_t *__restrict__ seg_nproc, float4 *__restrict__ ckpt) {
    using SM = FwdSmem<CDIM>;
    constexpr int CH = SM::CH;
    constexpr int NB = CDIM > 3 ? CDIM / 3 : 1;
    __shared__ SM sm;
    const int lane = threadIdx.x;
    const uint32_t item = blockIdx.x;
    if (item >= (uint32_t)*n_items) return;
    const uint2 it = items[item];
    const uint32_t tile = it.x, seg = it.y;
    const uint32_t start = (uint32_t)ranges[2 * tile], n_all = (uint32_t)ranges[2 * tile + 1] - start;
    const uint32_t seg_begin = GS_LONG_MIN + seg * GS_SEG_LEN;
    const uint32_t n = n_all - seg_begin < (uint32_t)GS_SEG_LEN ? n_all - seg_begin : (uint32_t)GS_SEG_LEN;
    const uint32_t tx = tile % (uint32_t)G.ntx, ty = tile / (uint32_t)G.ntx;
    const uint32_t id_x = tx * 16 + (lane & 15), id_y0 = ty * 16 + (lane >> 4);
    const float px = raster_pixel_coord(id_x, G.padW, G.focal_x);
    f2 py2[NPP], SH[NPP][NB];
#pragma unroll
    for (int h = 0; h < NPP; ++h) {
        py2[h] = f2{raster_pixel_coord(id_y0 + 8 * h, G.padH, G.focal_y),
                    raster_pixel_coord(id_y0 + 8 * h + 4, G.padH, G.focal_y)};
        if (CDIM > 3 && PHASE == 2) {
            float a9[NB], b9[NB];
            raster_pixel_sh<NB>(id_x, id_y0 + 8 * h, G, a9);
            raster_pixel_sh<NB>(id_x, id_y0 + 8 * h + 4, G, b9);
            constexpr float KS = GS_SH_PRESCALE ? -GS_LOG2E : 1.0f;  // raster_common.h
#pragma unroll
            for (int k9 = 0; k9 < NB; ++k9) SH[h][k9] = f2{KS * a9[k9], KS * b9[k9]};
        }
    }
    // pixel slot of pair h, element e: 128 h + 64 e + lane (the checkpoint layout)
    f2 T[NPP], cr[NPP], cg[NPP], cb[NPP];
#pragma unroll
    for (int h = 0; h < NPP; ++h) {
        T[h] = f2{1.0f, 1.0f};
        cr[h] = cg[h] = cb[h] = f2{0.f, 0.f};
    }
    if (PHASE == 2) {
        const float4 *c0 = cont_state + (size_t)tile * 256;
        const float *Pp = seg_P + (size_t)(item - seg) * 256;  // the tile's items are consecutive
#pragma unroll
        for (int h = 0; h < NPP; ++h) {
            float t0 = c0[128 * h + lane].x, t1 = c0[128 * h + 64 + lane].x;
            for (uint32_t q = 0; q < seg; ++q) {
                t0 *= Pp[(size_t)q * 256 + 128 * h + lane];
                t1 *= Pp[(size_t)q * 256 + 128 * h + 64 + lane];
            }
            T[h] = f2{t0 > GS_T_STOP ? t0 : 0.f, t1 > GS_T_STOP ? t1 : 0.f};
        }
    }
    auto any_live = [&]() {
        f2 sum = T[0];
#pragma unroll
        for (int h = 1; h < NPP; ++h) sum += T[h];
        return __ballot(sum.x + sum.y > 0.0f) != 0ull;
    };
    f2 live_scale = splat(GS_LIVE_SCALE), live_bias = splat(-GS_T_STOP * GS_LIVE_SCALE);
    asm volatile("" : "+v"(live_scale), "+v"(live_bias));
    uint32_t nproc = 0;
    bool done = PHASE == 2 && !any_live();
    for (uint32_t base = 0; base < n && !done; base += CH) {
        {  // stage the chunk (one Gaussian per lane); a single buffer, the wave is in program order
            const bool have = base + lane < n;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (have) {
                GaussianRec g;
                const uint32_t gj = start + seg_begin + base + lane;
                const uint32_t gid = raster_load<true>(S, gj, g);
                const float4 cq = S.conic4[(size_t)gid * GS_REC_STRIDE];
                sm.f[0][SM::X][lane] = g.x;
                sm.f[0][SM::Y][lane] = g.y;
                sm.f[0][SM::A][lane] = cq.x;
                sm.f[0][SM::B][lane] = cq.y;
                sm.f[0][SM::C][lane] = cq.z;
                sm.f[0][SM::NLOP][lane] = -__log2f(g.opa);
                if constexpr (CDIM == 3) {
                    float r0, r1, r2;
                    raster_load_rgb<true>(S, gj, gid, r0, r1, r2);
                    if (r0 != r0 || r1 != r1 || r2 != r2) {  // see raster_forward_kernel
                        sm.f[0][SM::NLOP][lane] = __builtin_nanf("");
                        r0 = r1 = r2 = 0.f;
                    }
                    sm.f[0][SM::R][lane] = r0;
                    sm.f[0][SM::G][lane] = r1;
                    sm.f[0][SM::BL][lane] = r2;
                } else if (PHASE == 2) {
                    const float *src = raster_sh_ptr<true, CDIM>(S, gj, gid);
#pragma unroll
                    for (int q = 0; q < CDIM; ++q) sm.sh[0][lane][q] = src[q];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        const uint32_t cnt = (n - base) < (uint32_t)CH ? (n - base) : (uint32_t)CH;
        if (PHASE == 2 && CKPT) {  // every chunk start of a segment is a bucket boundary of the tile's list
            float4 *c = ckpt + raster_ckpt_slot(start, tile, (seg_begin + base) / GS_BUCKET) * 256;
#pragma unroll
            for (int h = 0; h < NPP; ++h) {
                c[128 * h + lane] = make_float4(T[h].x, cr[h].x, cg[h].x, cb[h].x);
                c[128 * h + 64 + lane] = make_float4(T[h].y, cr[h].y, cg[h].y, cb[h].y);
            }
        }
#pragma unroll 1
        for (uint32_t i = 0; i < cnt; ++i) {
            if (PHASE == 2 && (i & 7) == 0 && !any_live()) {
                done = true;
                break;
            }
            const float gx = sm.f[0][SM::X][i], gy = sm.f[0][SM::Y][i], cA = sm.f[0][SM::A][i], cB = sm.f[0][SM::B][i];
            const float cC = sm.f[0][SM::C][i], nlop = sm.f[0][SM::NLOP][i];
            // the same expressions as raster_forward_kernel
            const float dx = px - gx;
            const float bdx = cB * dx;
            const float f = fmaf(cA * dx, dx, nlop);
#pragma unroll
            for (int h = 0; h < NPP; ++h) {
                const f2 dy = py2[h] - splat(gy);
                const f2 q = pk_fma(pk_fma(splat(cC), dy, splat(-bdx)), dy, splat(f));
                const f2 al = {gs_exp2(-q.x), gs_exp2(-q.y)};
                const f2 w = {mul_legacy(al.x, T[h].x), mul_legacy(al.y, T[h].y)};
                if (PHASE == 2) {
                    if constexpr (CDIM == 3) {
                        cr[h] = pk_fma(splat(sm.f[0][SM::R][i]), w, cr[h]);
                        cg[h] = pk_fma(splat(sm.f[0][SM::G][i]), w, cg[h]);
                        cb[h] = pk_fma(splat(sm.f[0][SM::BL][i]), w, cb[h]);
                    } else {
                        const float *co = sm.sh[0][i];
                        f2 v0 = {0.f, 0.f}, v1 = {0.f, 0.f}, v2 = {0.f, 0.f};
#pragma unroll
                        for (int k9 = 0; k9 < NB; ++k9) {
                            v0 = pk_fma(SH[h][k9], splat(co[k9]), v0);
                            v1 = pk_fma(SH[h][k9], splat(co[NB + k9]), v1);
                            v2 = pk_fma(SH[h][k9], splat(co[2 * NB + k9]), v2);
                        }
                        auto sg = [](float v) { return GS_SH_PRESCALE ? gs_rcp(1.0f + gs_exp2(v)) : gs_rcp(1.0f + __expf(-v)); };
                        const f2 c0 = {sg(v0.x), sg(v0.y)}, c1 = {sg(v1.x), sg(v1.y)}, c2 = {sg(v2.x), sg(v2.y)};
                        cr[h] = pk_fma(w, c0, cr[h]);
                        cg[h] = pk_fma(w, c1, cg[h]);
                        cb[h] = pk_fma(w, c2, cb[h]);
                    }
                }
                const f2 t = T[h] - w;  // T (1 - alpha)
                T[h] = PHASE == 2 ? t * live_mask(t, live_scale, live_bias) : t;
            }
        }
        nproc = base + cnt;
    }
    if (PHASE == 1) {
        float *P = seg_P + (size_t)item * 256;
#pragma unroll
        for (int h = 0; h < NPP; ++h) {
            P[128 * h + lane] = T[h].x;
            P[128 * h + 64 + lane] = T[h].y;
        }
    } else {
        float4 *C = seg_C + (size_t)item * 256;
#pragma unroll
        for (int h = 0; h < NPP; ++h) {
            C[128 * h + lane] = make_float4(T[h].x, cr[h].x, cg[h].x, cb[h].x);
            C[128 * h + 64 + lane] = make_float4(T[h].y, cr[h].y, cg[h].y, cb[h].y);
        }
        if (lane == 0) seg_nproc[item] = nproc;
    }
}

// Workgroup (tile, y): thread p = pixel slot p of the checkpoint layout (128 h + 64 e + lane).  Workgroup y fixes the
// checkpoints of the tile's segments [8 y, 8 y + 8) -- it first adds up the colours in front of them --, workgroup 0
// also writes the final pixels and the processed count.  (One workgroup per tile doing all of it: 0.4 ms for the 1,500
// checkpoints of a 100,000-Gaussian list.)
template <bool CKPT>
__global__ void __launch_bounds__(256) seg_combine_kernel(RasterGeom G, const int32_t *__restrict__ ranges,
                                                          const uint32_t *__restrict__ cont_flag,
                                                          const uint32_t *__restrict__ item_base,
                                                          const float4 *__restrict__ cont_state,
                                                          const float4 *__restrict__ seg_C,
                                                          const uint32_t *__restrict__ seg_nproc,
                                                          float4 *__restrict__ ckpt, uint32_t *__restrict__ tile_nproc,
                                                          float *__restrict__ out_padded, float *__restrict__ out_image) {
    constexpr uint32_t SEGS_PER_BLOCK = 8;
    const uint32_t tile = blockIdx.x;
    if (!cont_flag[tile]) return;
    const uint32_t i0 = item_base[tile], i1 = item_base[tile + 1];
    const uint32_t nseg = i1 - i0;
    // this workgroup's segments: blocks of eight, dealt round-robin over gridDim.y
    if (blockIdx.y > 0 && (!CKPT || blockIdx.y * SEGS_PER_BLOCK >= nseg)) return;
    const uint32_t p = threadIdx.x, lane = p & 63, k = p >> 6;  // k = 2 h + e
    const uint32_t start = (uint32_t)ranges[2 * tile], n_all = (uint32_t)ranges[2 * tile + 1] - start;
    const float4 s0 = cont_state[(size_t)tile * 256 + p];
    float c0 = s0.y, c1 = s0.z, c2 = s0.w;
    uint32_t nproc = GS_LONG_MIN;
    for (uint32_t sidx = 0; sidx < nseg; ++sidx) {
        const uint32_t it = i0 + sidx;
        const uint32_t np = seg_nproc[it];  // uniform
        if (np == 0) break;
        const uint32_t seg_begin = GS_LONG_MIN + sidx * GS_SEG_LEN;
        if (CKPT && (sidx / SEGS_PER_BLOCK) % gridDim.y == blockIdx.y) {
            for (uint32_t b = 0; b * GS_BUCKET < np; ++b) {
                float4 *c = ckpt + raster_ckpt_slot(start, tile, seg_begin / GS_BUCKET + b) * 256 + p;
                float4 v = *c;
                v.y += c0;
                v.z += c1;
                v.w += c2;
                *c = v;
            }
        }
        const float4 sc = seg_C[(size_t)it * 256 + p];
        c0 += sc.y;
        c1 += sc.z;
        c2 += sc.w;
        nproc += np;
        // a segment that stopped inside ends the tile's processed range, whatever the (independently rounded) incoming
        // transmittance of the next one says: the backward walks the first tile_nproc Gaussians, contiguously
        const uint32_t seg_len = n_all - seg_begin < (uint32_t)GS_SEG_LEN ? n_all - seg_begin : (uint32_t)GS_SEG_LEN;
        if (np < seg_len) break;
    }
    if (blockIdx.y > 0) return;
    if (tile_nproc && p == 0) tile_nproc[tile] = nproc;
    const uint32_t tx = tile % (uint32_t)G.ntx, ty = tile / (uint32_t)G.ntx;
    // k = 2 h + e: row y0 + 8 h + 4 e
    const uint32_t id_x = tx * 16 + (lane & 15), id_y2 = ty * 16 + (lane >> 4) + 8 * (k >> 1) + 4 * (k & 1);
    if (out_padded) {
        float *o = out_padded + ((size_t)id_y2 * G.padW + id_x) * 3;
        o[0] = c0;
        o[1] = c1;
        o[2] = c2;
    }
    if (out_image) {
        const int ox = (int)id_x - G.crop_left, oy = (int)id_y2 - G.crop_top;
        if (ox >= 0 && ox < G.width && oy >= 0 && oy < G.height) {
            float *o = out_image + ((size_t)oy * G.width + ox) * 3;
            o[0] = c0 > 1.f ? 1.f : (c0 < 0.f ? 0.f : c0);
            o[1] = c1 > 1.f ? 1.f : (c1 < 0.f ? 0.f : c1);
            o[2] = c2 > 1.f ? 1.f : (c2 < 0.f ? 0.f : c2);
        }
    }
}

// Waves in the grid: every wave gets the same number of tiles (+-1) and there are at most GS_FWD_WAVES_PER_SIMD
// (default 8) waves per SIMD -- at 1080p that is one tile per wave, dispatched by the hardware in index order as slots
// free up (with tile_order: longest-first list scheduling); fewer, longer-lived waves were slower (122-154 us against
// 85 us at cfg2).  The slot count is cached from the first device seen (all GPUs of a node are the same part).
struct FwdPlan {
    int slots;  // wave slots of the device at GS_FWD_WAVES_PER_SIMD
    int order;  // use tile_order (default 1; GS_FWD_ORDER=0 in the environment switches it off: A/B measurements)
};
static const FwdPlan &fwd_plan() {
    static const FwdPlan plan = [] {
        FwdPlan p;
        hipDeviceProp_t prop;
        int dev = 0, wps = GS_FWD_WAVES_PER_SIMD, cus = 256;
        if (const char *e = getenv("GS_FWD_WAVES_PER_SIMD")) wps = atoi(e) > 0 ? atoi(e) : wps;  // tuning knob
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        p.slots = cus * 4 * wps;
        const char *o = getenv("GS_FWD_ORDER");
        p.order = o ? atoi(o) != 0 : 1;
        return p;
    }();
    return plan;
}
static uint32_t fwd_grid(uint32_t n_tiles) {
    const FwdPlan &p = fwd_plan();
    const uint32_t rounds = (n_tiles + p.slots - 1) / p.slots;
    return rounds ? (n_tiles + rounds - 1) / rounds : 1;
}

template <int CDIM, bool FRAME, bool CKPT, bool SIG, bool EXACT = false>
void launch_fwd(const RasterSrc &S, const RasterGeom &G, const int32_t *ranges, float *out_padded, float *out_image,
                float4 *ckpt, uint32_t *tile_nproc, int wn, hipStream_t stream, float4 *cont_state = nullptr,
                uint32_t *cont_flag = nullptr, const uint32_t *tile_order = nullptr, uint32_t *tile_cost = nullptr,
                const uint32_t *cut_in = nullptr, uint32_t *cut_out = nullptr, unsigned long long *ranpast = nullptr,
                const unsigned long long *gate = nullptr) {
    if (!fwd_plan().order) tile_order = nullptr;
    const uint32_t T = (uint32_t)(G.ntx * G.nty);
    // the gated second pass of a culled frame almost never runs: a small persistent grid (every wave walks several tiles)
    // returns in ~2 us where T workgroups take ~8 us to be dispatched and retired
    const uint32_t grid = gate ? (T < 1024u ? T : 1024u) : fwd_grid(T);
    if (wn)
        hipLaunchKernelGGL((raster_forward_kernel<CDIM, FRAME, CKPT, SIG, true, EXACT>), dim3(grid), dim3(FWD_THREADS),
                           0, stream, S, G, ranges, out_padded, out_image, ckpt, tile_nproc, T, cont_state, cont_flag,
                           tile_order, tile_cost, cut_in, cut_out, ranpast, gate);
    else
        hipLaunchKernelGGL((raster_forward_kernel<CDIM, FRAME, CKPT, SIG, false, EXACT>), dim3(grid), dim3(FWD_THREADS),
                           0, stream, S, G, ranges, out_padded, out_image, ckpt, tile_nproc, T, cont_state, cont_flag,
                           tile_order, tile_cost, cut_in, cut_out, ranpast, gate);
}

}  // namespace

// Shared with raster_bwd.hip (replay of the forward for the reference-API backward).  `exact`: the reference's
// `fast = 0` flavour of the exponential (double division + double exp per pixel, gaussian.cu:922-923).
template <int CDIM, bool CKPT>
static void fwd_ref_dispatch(const RasterSrc &S, const RasterGeom &G, const int32_t *accum, float *res, int sigmoid,
                             int weight_normalize, int exact, float4 *ckpt, uint32_t *tile_nproc, hipStream_t stream) {
    if (sigmoid) {
        if (exact)
            launch_fwd<CDIM, false, CKPT, true, true>(S, G, accum, res, nullptr, ckpt, tile_nproc, weight_normalize, stream);
        else
            launch_fwd<CDIM, false, CKPT, true, false>(S, G, accum, res, nullptr, ckpt, tile_nproc, weight_normalize, stream);
    } else {
        if (exact)
            launch_fwd<CDIM, false, CKPT, false, true>(S, G, accum, res, nullptr, ckpt, tile_nproc, weight_normalize, stream);
        else
            launch_fwd<CDIM, false, CKPT, false, false>(S, G, accum, res, nullptr, ckpt, tile_nproc, weight_normalize, stream);
    }
}
int gs_raster_forward_ref(const RasterSrc &S, const RasterGeom &G, const int32_t *accum, float *res, int use_sh,
                          int sigmoid, int weight_normalize, float4 *ckpt, uint32_t *tile_nproc,
                          hipStream_t stream, int exact) {
    if (ckpt) {
        if (use_sh)
            fwd_ref_dispatch<27, true>(S, G, accum, res, sigmoid, weight_normalize, exact, ckpt, tile_nproc, stream);
        else
            fwd_ref_dispatch<3, true>(S, G, accum, res, sigmoid, weight_normalize, exact, ckpt, tile_nproc, stream);
    } else {
        if (use_sh)
            fwd_ref_dispatch<27, false>(S, G, accum, res, sigmoid, weight_normalize, exact, nullptr, nullptr, stream);
        else
            fwd_ref_dispatch<3, false>(S, G, accum, res, sigmoid, weight_normalize, exact, nullptr, nullptr, stream);
    }
    return 0;
}

extern "C" int gs_draw(const float *pos, const float *rgb, const float *opa, const float *cov,
                       const int32_t *tile_n_point_accum, float *res, int32_t h, int32_t w, int64_t M,
                       float focal_x, float focal_y, int weight_normalize, int sigmoid, int fast,
                       const float *rays_o, const float *lefttop_pos, const float *vec_dx, const float *vec_dy,
                       int use_sh_coeff, gs_stream_t stream) {
    // fast != 0: v_exp_f32 on the hoisted conic (the reference's __expf); fast == 0: the exact flavour
    GS_CHECK_ARG(h > 0 && w > 0 && (h % 16) == 0 && (w % 16) == 0, "h, w must be positive multiples of 16");
    GS_CHECK_ARG(M >= 0, "M < 0");
    GS_CHECK_ARG(tile_n_point_accum && res, "null pointer");
    GS_CHECK_ARG(M == 0 || (pos && rgb && opa && cov), "null pointer");
    GS_CHECK_ARG(((uintptr_t)cov & 15) == 0, "cov must be 16-byte aligned");
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
    int rc = gs_raster_forward_ref(S, G, tile_n_point_accum, res, use_sh_coeff, sigmoid, weight_normalize, nullptr,
                                   nullptr, (hipStream_t)stream, fast ? 0 : 1);
    if (rc) {
        gs_set_error("gs_draw: unsupported flag combination");
        return rc;
    }
    GS_CHECK_LAUNCH();
    return 0;
}

int gs_stage_raster_forward(const gs_frame *f, const gs_frame_ws &ws, const uint32_t *sorted_ids,
                            hipStream_t stream, bool second_pass) {
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
    // dense frames: tiles still alive after GS_LONG_MIN Gaussians hand the rest of their list to the segment kernels
    const bool dense = gs_frame_long_lists(f, FG.n_tiles) && ws.cont_state != nullptr &&
                       !(f->flags & GS_FRAME_SERIAL_LONG_LISTS);
    float4 *cs = dense ? ws.cont_state : nullptr;
    uint32_t *cf = dense ? ws.cont_flag : nullptr;
    // longest-first dispatch: the order is written by the strip variant's project + count launch and the cost is recorded
    // by strip-variant frames only (frames of the other variants neither read the order nor record a cost)
    const bool ordered = gs_frame_uses_strips(f) && f->N > 0 && ws.tile_order != nullptr;
    const uint32_t *order = ordered ? ws.tile_order : nullptr;
    uint32_t *cost = ordered ? ws.tile_cost : nullptr;
    // occlusion cut (gs_frame_layout.h): every INFERENCE frame-path launch leaves the table behind for the next frame of this
    // workspace (a training forward leaves it untouched: the caller must not allow the cull right behind one); a frame
    // whose lists were trimmed by it checks them, and its gated second pass composites the full lists
    const bool culled = gs_frame_occlusion_cull(f);
    const uint32_t *cut_in = culled && !second_pass ? gs_frame_cut_table(f, ws) : nullptr;
    unsigned long long *ranpast = ws.counters + GS_CNT_RANPAST;
    const unsigned long long *gate = second_pass ? ranpast : nullptr;
#define GS_LAUNCH_FRAME_FWD(CD)                                                                                        \
    do {                                                                                                               \
        if (f->training)                                                                                               \
            launch_fwd<CD, true, true, false>(S, G, ws.tile_ranges, f->image_padded, f->image, ws.ckpt, ws.tile_nproc, \
                                              0, stream, cs, cf, order, cost, nullptr, nullptr, ranpast, nullptr);     \
        else                                                                                                           \
            launch_fwd<CD, true, false, false>(S, G, ws.tile_ranges, f->image_padded, f->image, nullptr, nullptr, 0,   \
                                               stream, cs, cf, order, cost, cut_in, ws.cut, ranpast, gate);            \
    } while (0)
    if (f->color_dim == 48)
        GS_LAUNCH_FRAME_FWD(48);
    else if (f->color_dim == 27)
        GS_LAUNCH_FRAME_FWD(27);
    else
        GS_LAUNCH_FRAME_FWD(3);
#undef GS_LAUNCH_FRAME_FWD
    GS_CHECK_LAUNCH();
    if (dense) {
        const uint32_t T = (uint32_t)FG.n_tiles, cap = (uint32_t)gs_seg_items_cap(f->max_pairs, FG.n_tiles);
        unsigned long long *n_items = ws.counters + GS_CNT_SEGS;
        hipLaunchKernelGGL(seg_scan_kernel, dim3(1), dim3(1024), 0, stream, ws.cont_flag, ws.tile_ranges, T, cap,
                           ws.seg_item_base, ws.seg_items, n_items);
        GS_CHECK_LAUNCH();
#define GS_LAUNCH_SEG(CD, CK)                                                                                          \
    do {                                                                                                               \
        hipLaunchKernelGGL((raster_segment_kernel<CD, CK, 1>), dim3(cap), dim3(FWD_THREADS), 0, stream, S, G,          \
                           ws.tile_ranges, ws.seg_items, n_items, ws.cont_state, ws.seg_P, ws.seg_C, ws.seg_nproc,    \
                           ws.ckpt);                                                                                   \
        hipLaunchKernelGGL((raster_segment_kernel<CD, CK, 2>), dim3(cap), dim3(FWD_THREADS), 0, stream, S, G,          \
                           ws.tile_ranges, ws.seg_items, n_items, ws.cont_state, ws.seg_P, ws.seg_C, ws.seg_nproc,    \
                           ws.ckpt);                                                                                   \
        hipLaunchKernelGGL((seg_combine_kernel<CK>), dim3(T, CK ? 16 : 1), dim3(256), 0, stream, G, ws.tile_ranges,     \
                           ws.cont_flag,                                                                               \
                           ws.seg_item_base, ws.cont_state, ws.seg_C, ws.seg_nproc, ws.ckpt, ws.tile_nproc,           \
                           f->image_padded, f->image);                                                                 \
    } while (0)
        if (f->training) {
            if (f->color_dim == 48)
                GS_LAUNCH_SEG(48, true);
            else if (f->color_dim == 27)
                GS_LAUNCH_SEG(27, true);
            else
                GS_LAUNCH_SEG(3, true);
        } else {
            if (f->color_dim == 48)
                GS_LAUNCH_SEG(48, false);
            else if (f->color_dim == 27)
                GS_LAUNCH_SEG(27, false);
            else
                GS_LAUNCH_SEG(3, false);
        }
#undef GS_LAUNCH_SEG
    }
    GS_CHECK_LAUNCH();
    return 0;
}
