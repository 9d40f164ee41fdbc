// gs_common.h -- shared host/device helpers for libgs_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gs_abi.h"

#define GS_WAVE 64
#define GS_TILE 16

// ---- error plumbing ------------------------------------------------------------------
void gs_set_error(const char *fmt, ...);

#define GS_CHECK_ARG(cond, what)                                   \
    do {                                                           \
        if (!(cond)) {                                             \
            gs_set_error("%s: invalid argument: %s", __func__, what); \
            return GS_E_INVALID;                                   \
        }                                                          \
    } while (0)

#define GS_CHECK_LAUNCH()                                                         \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            gs_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
            return (int)e_;                                                       \
        }                                                                         \
    } while (0)

#define GS_HIP(call)                                                              \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess) {                                                   \
            gs_set_error("%s: %s failed: %s", __func__, #call, hipGetErrorString(e_)); \
            return (int)e_;                                                       \
        }                                                                         \
    } while (0)

static inline int64_t gs_div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t gs_align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// ---- device helpers ---------------------------------------------------------------------
#ifdef __HIPCC__

// float -> uint32 with the saturating semantics of the reference's `(uint32_t)float`
// conversions in calc_tile_info_kernel3 (gaussian.cu:241-242): NaN/negative -> 0.
__device__ __forceinline__ uint32_t gs_f2u_sat(float v) {
    if (!(v > 0.0f)) return 0u;
    if (v >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)v;
}

// wave64 helpers
__device__ __forceinline__ int gs_lane() { return (int)__lane_id(); }

__device__ __forceinline__ float gs_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t gs_wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ uint32_t gs_wave_incl_scan_u32(uint32_t v) {
    const int lane = gs_lane();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// DPP full-wave shift right by one lane (gfx9 `wave_shr:1`): lane l receives src[l-1];
// lane 0 keeps `old`.
__device__ __forceinline__ float gs_wave_shr1(float old, float src) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                           0x138, 0xf, 0xf, false));
}

// 2^x on the transcendental unit (v_exp_f32)
// The `sigmoid` flag's squashing of alpha (reference API only; gaussian.cu:930, :622): 2. / (exp(-alpha) + 1) - 1.  The
// reference evaluates it in DOUBLE (its literals are doubles), and it has to: in float the result carries an ABSOLUTE error
// of ~6e-8 whatever alpha is (2 / (2 - a) - 1 cancels), i.e. percents of a small alpha -- the row gradients of faint
// Gaussians then come out with errors of 30 % of their term sum (round 6: the element-wise parity test of the flag
// found it; a tensor-max criterion had hidden it).  The flag's kernels are no hot path.
__device__ __forceinline__ float gs_squash_alpha(float a) { return (float)(2.0 / (exp(-(double)a) + 1.0) - 1.0); }
__device__ __forceinline__ float gs_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float gs_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float gs_rsq(float x) { return __builtin_amdgcn_rsqf(x); }

#define GS_LOG2E 1.4426950408889634f
#define GS_LN2 0.6931471805599453f

// Early-stop test of the reference: `if (accum < 0.0001) break;` compares the float
// promoted to double against the double literal (gaussian.cu:906); 0.0001f is the largest
// float below 0.0001, so the test is exactly `accum <= 0.0001f`.
#define GS_T_STOP 0.0001f

// Per-Gaussian record of the frame path: 4 float4 = 64 B, one cache-line-aligned slot per Gaussian so
// that a tile's gather of a Gaussian touches ONE line instead of three:
//   [0] geom  (x/z, y/z, |p_c|, sigmoid(opacity))     [1] cov   (a, b, c, d)
//   [2] color (sigmoid r, g, b, -)                    [3] conic (A, B, C, -) in log2 units
#define GS_REC_STRIDE 4

// det = a*d - b*c without FMA contraction: the reference (and the oracle) round both products
// before subtracting; for needle-like footprints (a*d ~ b*c) a contracted det differs by many
// ulps and that difference is amplified into the exponent.
__device__ __forceinline__ float gs_det(float a, float b, float c, float d) {
#pragma clang fp contract(off)
    return a * d - b * c;
}

// "dist" tile culling (calc_tile_info_kernel, gaussian.cu:101-136; splatter.py:571-578): is tile (ix, iy) listed for
// a Gaussian centred at (px, py)?  The tile edges are Tiles.create_tiles' (splatter.py:275-293): left(i) =
// (16 i - pad/2) / focal with an exact integer-valued numerator, right(i) = left(i + 1) bit for bit; the centre is
// (left + right) / 2 and the test d1 d1 + d2 d2 < thresh, all in fp32 without contraction, as the reference kernel.
struct GsDistCull {
    float half_padw, half_padh, fx, fy, thresh;
};
__device__ __forceinline__ bool gs_dist_listed(float px, float py, uint32_t ix, uint32_t iy, const GsDistCull &D) {
#pragma clang fp contract(off)
    const float left = (16.0f * (float)ix - D.half_padw) / D.fx, right = (16.0f * (float)(ix + 1) - D.half_padw) / D.fx;
    const float top = (16.0f * (float)iy - D.half_padh) / D.fy, bottom = (16.0f * (float)(iy + 1) - D.half_padh) / D.fy;
    const float center_y = (top + bottom) / 2, center_x = (left + right) / 2;
    const float d1 = px - center_x, d2 = py - center_y;
    return d1 * d1 + d2 * d2 < D.thresh;
}

// Conic in log2 units: G = 2^-(A dx^2 - B dx dy + C dy^2) == exp(-(d dx^2-(b+c)dx dy+a dy^2)/(2det+1e-14))
// (gaussian.cu:916-923).  Hoists the reference's per-pixel fp64 division to once per Gaussian.
__device__ __forceinline__ void gs_conic(float a, float b, float c, float d, float &A, float &B, float &C) {
#pragma clang fp contract(off)
    const float det = a * d - b * c;
    const float k = GS_LOG2E / (2.0f * det + 1e-14f);
    A = d * k;
    B = (b + c) * k;
    C = a * k;
}

// One element of torch's _single_tensor_adam (adam.hip; also the fused projection-backward + Adam kernel of cull_project.hip --
// both translation units are compiled with -ffp-contract=off, so the two paths round alike):
//   m <- m + (g - m)(1 - b1);  v <- v b2 + (1 - b2) g g;  p <- p - step_size m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__device__ __forceinline__ void gs_adam_one(float &p, float g, float &m, float &v, float step_size, float one_m_b1,
                                            float b2, float one_m_b2, float inv_bc2_sqrt, float eps) {
#pragma clang fp contract(off)
    m = m + (g - m) * one_m_b1;
    v = v * b2 + one_m_b2 * (g * g);
    const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

#endif  // __HIPCC__
