// raster_common.h -- shared pieces of the tile rasterizer forward/backward kernels.
#pragma once
// SH colours sigma(sum_k sh_k(pixel) coef_k) are evaluated as 1 / (1 + 2^(sum_k sh'_k coef_k)) with the pixel's basis
// values pre-scaled once per wave, sh'_k = -log2(e) sh_k: the exponential's argument needs no multiplication per
// (pixel, Gaussian, channel) -- 12 VALU instructions per Gaussian step of a wave in the SH forward and backward
// kernels (8 % of the forward's).  A/B switch for tools/ab_variants.py.
#ifndef GS_SH_PRESCALE
#define GS_SH_PRESCALE 1
#endif
#include "gs_common.h"
#include "gs_frame_layout.h"

// Where the Gaussians of a tile's sorted list live.
//  FRAME = true : sorted Gaussian ids + N-indexed records written by stage S1 (no sorted
//                 attribute copies exist; gathers hit L2 / Infinity Cache).
//  FRAME = false: the reference API -- attributes already gathered in sorted order
//                 (pos[M,3], rgb[M,D], opa[M], cov[M,2,2]; gaussian.cu:806-813).
struct RasterSrc {
    const uint32_t *ids;
    const float4 *geom;    // (x, y, depth, opacity)   } interleaved, stride GS_REC_STRIDE float4
    const float4 *cov4;    // (a, b, c, d)             } (one 64-byte record per Gaussian)
    const float4 *color4;  // (r, g, b, -)             }
    const float4 *conic4;  // (A, B, C, -)             }
    const float *sh;       // raw rgb parameter [N,27 | 48] (SH coefficients, channel-major)
    const float *pos, *rgb, *opa, *cov;
};

struct RasterGeom {
    int32_t padW, padH, ntx, nty;
    int32_t width, height, crop_top, crop_left;  // un-padded output (frame path)
    float focal_x, focal_y;
    float rays_o[3], lefttop[3], vdx[3], vdy[3];
    // reference API (gs_draw / gs_draw_backward): the ray basis arrives as four DEVICE tensors of three floats; the
    // kernels read them themselves (uniform loads in the prologue) instead of the host copying them back and
    // synchronising the stream.  All NULL on the frame path, which passes the basis by value above.
    const float *dev_rays_o, *dev_lefttop, *dev_vdx, *dev_vdy;
};

struct GaussianRec {
    float x, y, a, b, c, d, opa;
};

template <bool FRAME>
__device__ __forceinline__ uint32_t raster_load(const RasterSrc &S, uint32_t j, GaussianRec &g) {
    if (FRAME) {
        const uint32_t id = S.ids[j];
        const float4 ge = S.geom[(size_t)id * GS_REC_STRIDE], cv = S.cov4[(size_t)id * GS_REC_STRIDE];
        g.x = ge.x;
        g.y = ge.y;
        g.opa = ge.w;
        g.a = cv.x;
        g.b = cv.y;
        g.c = cv.z;
        g.d = cv.w;
        return id;
    } else {
        g.x = S.pos[(size_t)j * 3 + 0];
        g.y = S.pos[(size_t)j * 3 + 1];
        g.opa = S.opa[j];
        const float4 cv = reinterpret_cast<const float4 *>(S.cov)[j];
        g.a = cv.x;
        g.b = cv.y;
        g.c = cv.z;
        g.d = cv.w;
        return j;
    }
}

template <bool FRAME>
__device__ __forceinline__ void raster_load_rgb(const RasterSrc &S, uint32_t j, uint32_t id, float &r, float &g,
                                                float &b) {
    if (FRAME) {
        const float4 c = S.color4[(size_t)id * GS_REC_STRIDE];
        r = c.x;
        g = c.y;
        b = c.z;
    } else {
        r = S.rgb[(size_t)j * 3 + 0];
        g = S.rgb[(size_t)j * 3 + 1];
        b = S.rgb[(size_t)j * 3 + 2];
    }
}

template <bool FRAME, int CDIM>
__device__ __forceinline__ const float *raster_sh_ptr(const RasterSrc &S, uint32_t j, uint32_t id) {
    return FRAME ? S.sh + (size_t)id * CDIM : S.rgb + (size_t)j * CDIM;
}

__device__ __forceinline__ float raster_det(float a, float b, float c, float d) { return gs_det(a, b, c, d); }
__device__ __forceinline__ void raster_conic(const GaussianRec &g, float &A, float &B, float &C) {
    gs_conic(g.a, g.b, g.c, g.d, A, B, C);
}

// Pixel centre in normalised image units.  The reference evaluates (id + 0.5 - w/2) / focal in double
// and rounds to float (gaussian.cu:839-840).  The numerator is a half-integer (exact in fp32), so one
// correctly-rounded fp32 division gives the same float except in the ~2^-29 of cases where the double
// quotient sits within half a double-ulp of an fp32 rounding boundary (then 1 ulp): two fp64 divisions
// per lane were ~15 % of this kernel's per-tile prologue.
__device__ __forceinline__ float raster_pixel_coord(uint32_t id, int32_t padded, float focal) {
    return ((float)id + 0.5f - (float)((uint32_t)padded / 2)) / focal;
}

// Per-pixel SH basis (gaussian.cu:849-861, 405-426), same promotions as the reference.  NB = 9 is the reference's
// degree 2.  NB = 16 adds the degree-3 band in the same (svox2) sign convention with the C3 table the reference
// declares (gaussian.cu:395-403) but never reads -- BASELINE config 4 names "SH degree 3"; it is an extension.
template <int NB>
__device__ __forceinline__ void raster_pixel_sh(uint32_t id_x, uint32_t id_y, const RasterGeom &G, float *SH) {
    float dir[3], nrm = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const bool dev = G.dev_rays_o != nullptr;  // uniform
        const float lt = dev ? G.dev_lefttop[i] : G.lefttop[i], vx = dev ? G.dev_vdx[i] : G.vdx[i];
        const float vy = dev ? G.dev_vdy[i] : G.vdy[i], ro = dev ? G.dev_rays_o[i] : G.rays_o[i];
        dir[i] = lt + id_x * vx + id_y * vy - ro;
        nrm += dir[i] * dir[i];
    }
    nrm = sqrtf(nrm);
#pragma unroll
    for (int i = 0; i < 3; ++i) dir[i] = (float)(dir[i] / (nrm + 1e-7));
    const float x = dir[0], y = dir[1], z = dir[2];
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    SH[0] = 0.28209479177387814f;
    SH[1] = -0.4886025119029199f * y;
    SH[2] = 0.4886025119029199f * z;
    SH[3] = -0.4886025119029199f * x;
    SH[4] = 1.0925484305920792f * xy;
    SH[5] = -1.0925484305920792f * yz;
    SH[6] = (float)(0.31539156525252005f * (2.0 * zz - xx - yy));
    SH[7] = -1.0925484305920792f * xz;
    SH[8] = 0.5462742152960396f * (xx - yy);
    if (NB == 16) {
        SH[9] = -0.5900435899266435f * y * (3.0f * xx - yy);
        SH[10] = 2.890611442640554f * xy * z;
        SH[11] = -0.4570457994644658f * y * (4.0f * zz - xx - yy);
        SH[12] = 0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
        SH[13] = -0.4570457994644658f * x * (4.0f * zz - xx - yy);
        SH[14] = 1.445305721320277f * z * (xx - yy);
        SH[15] = -0.5900435899266435f * x * (xx - 3.0f * yy);
    }
}

// Checkpoint slot of local bucket b of a tile whose sorted list starts at `start`:
// floor(start/64) + tile + b.  Slots of different tiles never overlap and the total is
// <= M/64 + T, so no scan is needed before the forward pass.
__device__ __forceinline__ size_t raster_ckpt_slot(uint32_t start, uint32_t tile, uint32_t b) {
    return (size_t)(start / GS_BUCKET) + tile + b;
}
