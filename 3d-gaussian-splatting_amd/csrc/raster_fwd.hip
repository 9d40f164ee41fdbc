// raster_fwd.hip -- 16x16-tile front-to-back alpha compositing, forward.
//
// Replaces draw_kernel (gaussian.cu:806-970).  One 256-thread workgroup (4 wave64) per tile,
// one pixel per lane (wave w owns pixel rows 4w..4w+3 of the tile).  The tile's sorted
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
#include "raster_common.h"

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

template <int CDIM>
struct FwdSmem;
template <>
struct FwdSmem<3> {
    // structure of arrays: four consecutive Gaussians of one field are one ds_read_b128, and two
    // consecutive ones sit in an even/odd register pair -- the operand shape of v_pk_*_f32
    static constexpr int CH = 256;
    enum { X, Y, A, B, C, OPA, R, G, BL, NFIELD };
    float f[2][NFIELD][CH] __attribute__((aligned(16)));
    int done[2][4];
};
template <>
struct FwdSmem<27> {
    static constexpr int CH = 64;
    float4 a[2][CH];
    float4 b[2][CH];  // C, opacity, -, -
    float sh[2][CH][28];
    int done[2][4];
};

template <int CDIM, bool FRAME, bool CKPT, bool SIG, bool WN>
__global__ void __launch_bounds__(256) raster_forward_kernel(RasterSrc S, RasterGeom G,
                                                            const int32_t *__restrict__ ranges,
                                                            float *__restrict__ out_padded,
                                                            float *__restrict__ out_image,
                                                            float4 *__restrict__ ckpt,
                                                            uint32_t *__restrict__ tile_nproc) {
    using SM = FwdSmem<CDIM>;
    constexpr int CH = SM::CH;
    __shared__ SM sm;

    const uint32_t tile = blockIdx.x;
    const uint32_t tx = tile % (uint32_t)G.ntx, ty = tile / (uint32_t)G.ntx;
    const uint32_t start = (uint32_t)(FRAME ? ranges[2 * tile] : ranges[tile]);
    const uint32_t end = (uint32_t)(FRAME ? ranges[2 * tile + 1] : ranges[tile + 1]);
    const uint32_t n = end - start;
    const int p = threadIdx.x, wave = p >> 6, lane = p & 63;
    const uint32_t id_x = tx * 16 + (p & 15), id_y = ty * 16 + (p >> 4);
    const float px = raster_pixel_coord(id_x, G.padW, G.focal_x);
    const float py = raster_pixel_coord(id_y, G.padH, G.focal_y);
    float SH[9];
    if (CDIM == 27) raster_pixel_sh(id_x, id_y, G, SH);

    float T = 1.0f, cr = 0.f, cg = 0.f, cb = 0.f, accw = 0.f;
    f2 cr2 = {0.f, 0.f}, cg2 = {0.f, 0.f}, cb2 = {0.f, 0.f};  // even/odd partial colours of the packed path
    const f2 px2 = {px, px}, py2 = {py, py};
    bool wave_done = false;
    uint32_t nproc = 0;

    // register stage for the next chunk
    GaussianRec g;
    float r0 = 0, r1 = 0, r2 = 0;
    uint32_t gid = 0;
    bool have = false;
    auto fetch = [&](uint32_t base) {
        have = (uint32_t)p < (uint32_t)CH && base + p < n;
        if (have) {
            gid = raster_load<FRAME>(S, start + base + p, g);
            if (CDIM == 3) raster_load_rgb<FRAME>(S, start + base + p, gid, r0, r1, r2);
        }
    };
    auto write_ckpt = [&](uint32_t idx_in_tile) {
        ckpt[raster_ckpt_slot(start, tile, idx_in_tile / GS_BUCKET) * 256 + p] =
            make_float4(T, cr + cr2.x + cr2.y, cg + cg2.x + cg2.y, cb + cb2.x + cb2.y);
    };

    fetch(0);
    int k = 0;
    for (uint32_t base = 0; base < n; base += CH, ++k) {
        const int buf = k & 1;
        if (have) {
            float A, B, C;
            raster_conic(g, A, B, C);
            float opa = g.opa;
            if (SIG)  // gaussian.cu:918: (1.0/2*3.1415926536) * rsqrtf(det + 1e-7), folded into opacity
                opa *= 1.5707963268f * rsqrtf(raster_det(g.a, g.b, g.c, g.d) + 1e-7f);
            if constexpr (CDIM == 3) {
                sm.f[buf][SM::X][p] = g.x;
                sm.f[buf][SM::Y][p] = g.y;
                sm.f[buf][SM::A][p] = A;
                sm.f[buf][SM::B][p] = B;
                sm.f[buf][SM::C][p] = C;
                sm.f[buf][SM::OPA][p] = opa;
                sm.f[buf][SM::R][p] = r0;
                sm.f[buf][SM::G][p] = r1;
                sm.f[buf][SM::BL][p] = r2;
            } else {
                sm.a[buf][p] = make_float4(g.x, g.y, A, B);
                sm.b[buf][p] = make_float4(C, opa, 0.f, 0.f);
                const float *src = raster_sh_ptr<FRAME>(S, start + base + p, gid);
#pragma unroll
                for (int q = 0; q < 27; ++q) sm.sh[buf][p][q] = src[q];
            }
        } else if ((uint32_t)p < (uint32_t)CH && base + p < ((n + 3u) & ~3u)) {
            // pad the ragged tail to a multiple of 4 with null Gaussians (opacity 0 => alpha 0)
            if constexpr (CDIM == 3) {
#pragma unroll
                for (int k = 0; k < SM::NFIELD; ++k) sm.f[buf][k][p] = 0.f;
            } else {
                sm.a[buf][p] = make_float4(0.f, 0.f, 0.f, 0.f);
                sm.b[buf][p] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int q = 0; q < 27; ++q) sm.sh[buf][p][q] = 0.f;
            }
        }
        if (lane == 0) sm.done[buf][wave] = wave_done ? 1 : 0;
        __syncthreads();
        if (sm.done[buf][0] & sm.done[buf][1] & sm.done[buf][2] & sm.done[buf][3]) break;
        fetch(base + CH);  // overlaps with the compositing below
        const uint32_t cnt = (n - base) < (uint32_t)CH ? (n - base) : (uint32_t)CH;
        for (uint32_t sub = 0; sub < cnt; sub += GS_BUCKET) {
            if (CKPT) write_ckpt(base + sub);
            if (wave_done) continue;
            const uint32_t lim = (sub + GS_BUCKET) < cnt ? (sub + GS_BUCKET) : cnt;
            // groups of 4 Gaussians: one wave-uniform liveness test per group, per-pixel masking
            // inside (exactly the reference's per-pixel `accum < 0.0001` test, gaussian.cu:906)
            for (uint32_t i = sub; i < lim; i += 4) {
                if (__ballot(T > GS_T_STOP) == 0ull) {
                    wave_done = true;
                    break;
                }
                if constexpr (CDIM == 3) {
                    // two Gaussians per packed instruction (v_pk_add/mul/fma_f32); the transmittance
                    // recurrence and the per-pixel liveness test stay scalar and in order
                    auto ld4 = [&](int k) {
                        return *(const float4 *)__builtin_assume_aligned(&sm.f[buf][k][i], 16);
                    };
                    const float4 X = ld4(SM::X), Y = ld4(SM::Y), A4 = ld4(SM::A), B4 = ld4(SM::B), C4 = ld4(SM::C);
                    const float4 O4 = ld4(SM::OPA), R4 = ld4(SM::R), G4 = ld4(SM::G), L4 = ld4(SM::BL);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f2 gx = h ? f2{X.z, X.w} : f2{X.x, X.y}, gy = h ? f2{Y.z, Y.w} : f2{Y.x, Y.y};
                        const f2 cA = h ? f2{A4.z, A4.w} : f2{A4.x, A4.y}, cB = h ? f2{B4.z, B4.w} : f2{B4.x, B4.y};
                        const f2 cC = h ? f2{C4.z, C4.w} : f2{C4.x, C4.y}, op = h ? f2{O4.z, O4.w} : f2{O4.x, O4.y};
                        const f2 dx = px2 - gx, dy = py2 - gy;
                        f2 t = __builtin_elementwise_fma(-cB, dy, cA * dx);  // A dx - B dy
                        f2 q = __builtin_elementwise_fma(cC * dy, dy, dx * t);  // + C dy^2
                        f2 al;
                        al.x = gs_exp2(-q.x);
                        al.y = gs_exp2(-q.y);
                        al = al * op;
                        if (SIG) {  // gaussian.cu:930
                            al.x = 2.0f / (__expf(-al.x) + 1.0f) - 1.0f;
                            al.y = 2.0f / (__expf(-al.y) + 1.0f) - 1.0f;
                        }
                        f2 w;
                        const float a0 = (T > GS_T_STOP) ? al.x : 0.0f;
                        w.x = a0 * T;
                        T = fmaf(-a0, T, T);  // T * (1 - alpha)
                        const float a1 = (T > GS_T_STOP) ? al.y : 0.0f;
                        w.y = a1 * T;
                        T = fmaf(-a1, T, T);
                        cr2 = __builtin_elementwise_fma(h ? f2{R4.z, R4.w} : f2{R4.x, R4.y}, w, cr2);
                        cg2 = __builtin_elementwise_fma(h ? f2{G4.z, G4.w} : f2{G4.x, G4.y}, w, cg2);
                        cb2 = __builtin_elementwise_fma(h ? f2{L4.z, L4.w} : f2{L4.x, L4.y}, w, cb2);
                        if (WN) accw += w.x + w.y;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float4 ga = sm.a[buf][i + u], gb = sm.b[buf][i + u];
                        const float dx = px - ga.x, dy = py - ga.y;
                        float t = ga.z * dx;
                        t = fmaf(-ga.w, dy, t);  // A dx - B dy
                        float q = dx * t;
                        q = fmaf(gb.x * dy, dy, q);  // + C dy^2
                        float alpha = gs_exp2(-q) * gb.y;
                        if (SIG) alpha = 2.0f / (__expf(-alpha) + 1.0f) - 1.0f;  // gaussian.cu:930
                        alpha = (T > GS_T_STOP) ? alpha : 0.0f;
                        const float wgt = alpha * T;
                        const float *co = sm.sh[buf][i + u];
                        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
                        for (int k9 = 0; k9 < 9; ++k9) {
                            v0 = fmaf(SH[k9], co[k9], v0);
                            v1 = fmaf(SH[k9], co[9 + k9], v1);
                            v2 = fmaf(SH[k9], co[18 + k9], v2);
                        }
                        cr = fmaf(wgt, gs_rcp(1.0f + __expf(-v0)), cr);
                        cg = fmaf(wgt, gs_rcp(1.0f + __expf(-v1)), cg);
                        cb = fmaf(wgt, gs_rcp(1.0f + __expf(-v2)), cb);
                        if (WN) accw += wgt;
                        T = fmaf(-alpha, T, T);  // T * (1 - alpha)
                    }
                }
            }
        }
        nproc = base + cnt;
    }
    if (tile_nproc && p == 0) tile_nproc[tile] = nproc;

    if (!WN || accw < 0.01f) accw = 1.0f;  // gaussian.cu:964-969
    cr += cr2.x + cr2.y;
    cg += cg2.x + cg2.y;
    cb += cb2.x + cb2.y;
    const float o0 = cr / accw, o1 = cg / accw, o2 = cb / accw;
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
            o[0] = fminf(fmaxf(o0, 0.f), 1.f);
            o[1] = fminf(fmaxf(o1, 0.f), 1.f);
            o[2] = fminf(fmaxf(o2, 0.f), 1.f);
        }
    }
}

template <int CDIM, bool FRAME, bool CKPT, bool SIG>
void launch_fwd(const RasterSrc &S, const RasterGeom &G, const int32_t *ranges, float *out_padded, float *out_image,
                float4 *ckpt, uint32_t *tile_nproc, int wn, hipStream_t stream) {
    if (wn)
        hipLaunchKernelGGL((raster_forward_kernel<CDIM, FRAME, CKPT, SIG, true>), dim3(G.ntx * G.nty), dim3(256), 0,
                           stream, S, G, ranges, out_padded, out_image, ckpt, tile_nproc);
    else
        hipLaunchKernelGGL((raster_forward_kernel<CDIM, FRAME, CKPT, SIG, false>), dim3(G.ntx * G.nty), dim3(256), 0,
                           stream, S, G, ranges, out_padded, out_image, ckpt, tile_nproc);
}

}  // namespace

// Shared with raster_bwd.hip (replay of the forward for the reference-API backward).
int gs_raster_forward_ref(const RasterSrc &S, const RasterGeom &G, const int32_t *accum, float *res, int use_sh,
                          int sigmoid, int weight_normalize, float4 *ckpt, uint32_t *tile_nproc,
                          hipStream_t stream) {
    if (ckpt) {
        if (sigmoid) return GS_E_UNSUPPORTED;
        if (use_sh)
            launch_fwd<27, false, true, false>(S, G, accum, res, nullptr, ckpt, tile_nproc, weight_normalize, stream);
        else
            launch_fwd<3, false, true, false>(S, G, accum, res, nullptr, ckpt, tile_nproc, weight_normalize, stream);
    } else if (use_sh) {
        if (sigmoid)
            launch_fwd<27, false, false, true>(S, G, accum, res, nullptr, nullptr, nullptr, weight_normalize, stream);
        else
            launch_fwd<27, false, false, false>(S, G, accum, res, nullptr, nullptr, nullptr, weight_normalize, stream);
    } else {
        if (sigmoid)
            launch_fwd<3, false, false, true>(S, G, accum, res, nullptr, nullptr, nullptr, weight_normalize, stream);
        else
            launch_fwd<3, false, false, false>(S, G, accum, res, nullptr, nullptr, nullptr, weight_normalize, stream);
    }
    return 0;
}

extern "C" int gs_draw(const float *pos, const float *rgb, const float *opa, const float *cov,
                       const int32_t *tile_n_point_accum, float *res, int32_t h, int32_t w, int64_t M,
                       float focal_x, float focal_y, int weight_normalize, int sigmoid, int fast,
                       const float *rays_o, const float *lefttop_pos, const float *vec_dx, const float *vec_dy,
                       int use_sh_coeff, gs_stream_t stream) {
    (void)fast;  // both exp flavours map to v_exp_f32 here (accuracy ~1 ulp)
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
        float hb[12];
        hipStream_t s = (hipStream_t)stream;
        GS_HIP(hipMemcpyAsync(hb + 0, rays_o, 12, hipMemcpyDeviceToHost, s));
        GS_HIP(hipMemcpyAsync(hb + 3, lefttop_pos, 12, hipMemcpyDeviceToHost, s));
        GS_HIP(hipMemcpyAsync(hb + 6, vec_dx, 12, hipMemcpyDeviceToHost, s));
        GS_HIP(hipMemcpyAsync(hb + 9, vec_dy, 12, hipMemcpyDeviceToHost, s));
        GS_HIP(hipStreamSynchronize(s));
        for (int i = 0; i < 3; ++i) {
            G.rays_o[i] = hb[i];
            G.lefttop[i] = hb[3 + i];
            G.vdx[i] = hb[6 + i];
            G.vdy[i] = hb[9 + i];
        }
    }
    int rc = gs_raster_forward_ref(S, G, tile_n_point_accum, res, use_sh_coeff, sigmoid, weight_normalize, nullptr,
                                   nullptr, (hipStream_t)stream);
    if (rc) {
        gs_set_error("gs_draw: unsupported flag combination");
        return rc;
    }
    GS_CHECK_LAUNCH();
    return 0;
}

int gs_stage_raster_forward(const gs_frame *f, const gs_frame_ws &ws, const uint32_t *sorted_ids,
                            hipStream_t stream) {
    gs_frame_geom FG = gs_frame_geometry(f);
    RasterSrc S = {};
    S.ids = sorted_ids;
    S.geom = ws.rec_geom;
    S.cov4 = ws.rec_cov;
    S.color4 = ws.rec_color;
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
    if (f->color_dim == 27) {
        if (f->training)
            launch_fwd<27, true, true, false>(S, G, ws.tile_ranges, f->image_padded, f->image, ws.ckpt, ws.tile_nproc,
                                              0, stream);
        else
            launch_fwd<27, true, false, false>(S, G, ws.tile_ranges, f->image_padded, f->image, nullptr, nullptr, 0,
                                               stream);
    } else {
        if (f->training)
            launch_fwd<3, true, true, false>(S, G, ws.tile_ranges, f->image_padded, f->image, ws.ckpt, ws.tile_nproc,
                                             0, stream);
        else
            launch_fwd<3, true, false, false>(S, G, ws.tile_ranges, f->image_padded, f->image, nullptr, nullptr, 0,
                                              stream);
    }
    GS_CHECK_LAUNCH();
    return 0;
}
