// loss.hip -- the reference trainer's image loss and its gradient, fused.
//
// train.py:99-107:  loss = (1 - w) mean|x - y| + w (1 - SSIM(x, y)),  w = --ssim_weight (0.1),
// SSIM = torchmetrics.StructuralSimilarityIndexMeasure(data_range=1.0) (train.py:71; unpinned third-party
// dependency, absent here; algorithm restated from its published source, functional/image/ssim.py):
// 11x11 Gaussian window (sigma 1.5, separable, normalised), k1 = 0.01, k2 = 0.03; inputs are reflect-
// padded by 5, filtered with a VALID convolution and the result cropped by 5 again -- i.e. the SSIM map
// is evaluated exactly on the interior pixels whose window lies inside the image, and averaged over
// them and the 3 channels.  Per pixel, with mu = g*x, nu = g*y, Exx = g*x^2, Eyy = g*y^2, Exy = g*xy:
//   A1 = 2 mu nu + c1, A2 = 2 (Exy - mu nu) + c2, B1 = mu^2 + nu^2 + c1,
//   B2 = max(Exx - mu^2, 0) + max(Eyy - nu^2, 0) + c2,   S = A1 A2 / (B1 B2).
// torch autograd would run ~40 elementwise/conv kernels over 5 padded copies of the image for this.  With
//   D_mu = dS/dmu, D_xx = dS/dExx, D_xy = dS/dExy  (zero outside the interior),
//   dL/dx(q) = a sign(x - y) - b [ (g*D_mu)(q) + 2 x(q) (g*D_xx)(q) + y(q) (g*D_xy)(q) ],
//   a = (1 - w) / (3 H W), b = w / (3 (H - 10)(W - 10))  (g is symmetric)
// the whole thing is FOUR separable filter stages with a pointwise step in the middle, and it runs as ONE streaming
// kernel (loss_fused_kernel below; rounds 1 - 3 used two tiled passes with the nine adjoint planes in HBM between
// them, 143 us at 1080p) plus a one-block reduction.
// Images are [H, W, 3] fp32, the layout the rasterizer writes.
#include <type_traits>

#include "gs_common.h"

namespace {

constexpr int R = 5, K = 11;   // window radius / size
struct Window {
    float g[K];
};

struct LossGeom {
    int32_t H, W;
    float c1, c2, a, b;
};

template <int WAVES = 4>
__device__ __forceinline__ float block_sum(float v, float *s_red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) tot += s_red[w];
    return tot;
}

// ---------------------------------------------------------------------------------------------------------------------
// The fused kernel.  The image is a matrix of H rows x F = 3 W floats (channels interleaved), so a horizontal tap of
// one channel is a stride-3 tap of the flat row and every load / store of a wave is contiguous.  A workgroup of FT
// threads owns a vertical strip of SH gradient rows: thread t <-> flat column c0 - 15 + t.  It walks DOWN the strip one
// image row per step and keeps everything vertical in registers:
//   row of x, y (FT + 30 columns) -> LDS
//   H1  h[5]   = horizontal filter of (x, y, xx, yy, xy) at the row                              all FT columns
//   V1  ring of 11 x 5 partial vertical sums; the one that is complete after row i is the moment vector of row i - 5
//   P   moments -> S (summed over the strip's own rows) and the three adjoints, zero outside the interior -> LDS row
//   H2  g[3]   = horizontal filter of the adjoint row                                      FT - 30 output columns
//   V2  ring of 11 x 3 partial sums; complete after row i: gradient row i - 10 (+ the L1 term), stored
// The loop is unrolled over the 11 phases of the rings, so every ring slot is a fixed register (133 VGPRs, no scratch).
// Costs against the two-pass version: the strip's first 20 rows and 30 of its FT columns are recomputed halo (SH = 52:
// 1.37 x on H1 / V1, 1.19 x on H2 / V2; FT = 512: 1.06 x), nothing else is -- no adjoint planes (75 MB written,
// ~165 MB read back), no second read of the image beyond the two values of the output pixel, 22 + 33 LDS reads per
// column and row.  Measured at 1080p (profiles/r03_ab_*): 0.146 -> 0.096 ms per loss evaluation; the kernel issues
// 35.3 M VALU wave instructions (255 per column-row step: 77 H1 + 55 V1 + ~30 P + 33 H2 + 33 V2 + ~25 addressing and
// masks) = 57 us of issue time on 1024 SIMDs, and runs 91 - 95 us: 12 x 21 = 252 workgroups, one per CU with two
// waves per SIMD.  Measured alternatives (same box): FT = 256 with two workgroups per CU 0.100 ms, FT = 256 / SH = 38
// (three per CU, 23 % more halo work) 0.098, FT = 768 0.102, FT = 1024 0.116 - 0.150, the three products staged in LDS
// next to x, y (5 reads per tap instead of 2 reads + 2 multiplies) 0.111, every LDS read of a step issued before
// the first use 0.105, two barriers per step with single row buffers 0.097, IEEE divisions in P 0.101.
#ifndef GS_LOSS_SH
#define GS_LOSS_SH 52
#endif
#ifndef GS_LOSS_FT
#define GS_LOSS_FT 512
#endif
#ifndef GS_LOSS_RCP
#define GS_LOSS_RCP 1  // the three quotients of the pointwise step through v_rcp_f32 (1 ulp) instead of IEEE divisions
#endif
constexpr int FT = GS_LOSS_FT;      // threads = columns of the H1 / V1 / P stages
constexpr int FHALO = 3 * R;        // flat halo of one horizontal filter
constexpr int FCW = FT - 2 * FHALO; // output columns of a workgroup
constexpr int SH = GS_LOSS_SH;      // output rows of a workgroup

__global__ void __launch_bounds__(FT) loss_fused_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                        LossGeom L, Window Wd, float *__restrict__ grad,
                                                        float *__restrict__ p_ssim, float *__restrict__ p_l1) {
    constexpr int LW = FT + 2 * FHALO;
    // every LDS row is double-buffered by the parity of the step, so that ONE barrier per step suffices: step n reads
    // the image row n and the adjoint row n - 1 and writes the image row n + 1 and the adjoint row n into the other halves
    // (round 5: x and y interleaved, and the first two adjoints: one 8-byte read per tap feeds a packed fp32 instruction)
    typedef float f2 __attribute__((ext_vector_type(2)));
    __shared__ f2 s_xy[2][LW];
    __shared__ f2 s_d01[2][LW];
    __shared__ float s_d2[2][LW];
    __shared__ float s_red[FT / 64];
    const int t = threadIdx.x;
    const int F = 3 * L.W;
    const int c0 = blockIdx.x * FCW, r0 = blockIdx.y * SH;
    const int cv = c0 - FHALO + t;                     // this thread's column (may lie outside the image)
    const int pxv = cv >= 0 ? cv / 3 : -1;
    const bool col_in = cv < F && pxv >= R && pxv < L.W - R;          // the SSIM window fits horizontally
    const bool out_col = t >= FHALO && t < FT - FHALO && cv < F;      // this thread stores a gradient column
    // columns of the two LDS slots this thread fills (clamped: values outside the image only reach masked results)
    const int ca = min(max(c0 - 2 * FHALO + t, 0), F - 1), cb = min(max(c0 - 2 * FHALO + FT + t, 0), F - 1);
    const int co = min(max(cv, 0), F - 1);
    auto row_ptr = [&](const float *base, int i) { return base + (size_t)min(max(i, 0), L.H - 1) * F; };
    float xa, ya, xb = 0.f, yb = 0.f;
    auto fetch = [&](int i) {
        const float *px = row_ptr(x, i), *py = row_ptr(y, i);
        xa = px[ca];
        ya = py[ca];
        if (t < 2 * FHALO) {
            xb = px[cb];
            yb = py[cb];
        }
    };
    auto stage = [&](int buf) {
        s_xy[buf][t] = f2{xa, ya};
        if (t < 2 * FHALO) s_xy[buf][FT + t] = f2{xb, yb};
    };
    if (t < FHALO) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {  // never written again, read by idle columns
            s_d01[q][t] = s_d01[q][FT + FHALO + t] = f2{0.f, 0.f};
            s_d2[q][t] = s_d2[q][FT + FHALO + t] = 0.f;
        }
    }
    // the rings, in pairs for the packed instructions: (mu, nu), (Exx, Eyy), Exy; (D_mu, D_xx), D_xy
    f2 a1p[K][2], a2p[K];
    float a1s[K], a2s[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        a1p[k][0] = a1p[k][1] = a2p[k] = f2{0.f, 0.f};
        a1s[k] = a2s[k] = 0.f;
    }
    auto pfma = [](f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); };
    float ssim_sum = 0.f, l1_sum = 0.f;
    const int rows = min(SH, L.H - r0), n_end = rows + 4 * R;
    fetch(r0 - 2 * R);
    stage(0);
    fetch(r0 - 2 * R + 1);
    __syncthreads();

    // Step n, ring phase P = n mod 11: the image row i = r0 - 10 + n enters (H1, V1, P -> adjoint row i - 5), and the
    // adjoint row of step n - 1 goes through H2, V2 -> gradient row i - 11.  One step past the last image row drains it.
    auto step = [&](auto phase, int n) {
        constexpr int P = decltype(phase)::value;
        const int i = r0 - 2 * R + n, buf = n & 1;
        const bool head = n < n_end, tail = n > 2 * R;  // uniform: is there an image row / an adjoint row for this step?
        const int o = i - 2 * R - 1;                    // the gradient row this step completes (n > 20)
        const bool emit = n > 4 * R && out_col;
        float xo = 0.f, yo = 0.f;
        if (emit) {  // in flight across the whole step
            xo = row_ptr(x, o)[co];
            yo = row_ptr(y, o)[co];
        }
        if (head) {
            stage(buf ^ 1);  // the image row of step n + 1 (fetched during step n - 1)
            fetch(i + 2);
        }
        f2 h01 = {0.f, 0.f}, h23 = {0.f, 0.f}, g01 = {0.f, 0.f};
        float h4 = 0.f, g2 = 0.f;
        if (head) {  // H1
            const f2 *sxy = &s_xy[buf][t];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const f2 ab = sxy[3 * k];
                const f2 w = f2{Wd.g[k], Wd.g[k]} * ab;
                h01 += w;
                h23 = pfma(w, ab, h23);
                h4 = fmaf(w.x, ab.y, h4);
            }
        }
        if (tail) {  // H2 of the adjoint row written by step n - 1
            const f2 *sd01 = &s_d01[buf ^ 1][t];
            const float *sd2 = &s_d2[buf ^ 1][t];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                g01 = pfma(f2{Wd.g[k], Wd.g[k]}, sd01[3 * k], g01);
                g2 = fmaf(Wd.g[k], sd2[3 * k], g2);
            }
        }
        if (head) {
            // V1: image row n adds tap k to the partial sum of moment row n - k (slot (n - k) mod 11)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int slot = (P - k + K) % K;
                const f2 gk = {Wd.g[k], Wd.g[k]};
                a1p[slot][0] = k == 0 ? gk * h01 : pfma(gk, h01, a1p[slot][0]);
                a1p[slot][1] = k == 0 ? gk * h23 : pfma(gk, h23, a1p[slot][1]);
                a1s[slot] = k == 0 ? Wd.g[0] * h4 : fmaf(Wd.g[k], h4, a1s[slot]);
            }
            if (n >= 2 * R) {  // uniform: moment row rho = i - 5 is complete
                constexpr int E = (P + 1) % K;
                const int rho = i - R;
                const float mu = a1p[E][0].x, nu = a1p[E][0].y, exx = a1p[E][1].x, eyy = a1p[E][1].y, exy = a1s[E];
                const float vx_raw = exx - mu * mu, vy_raw = eyy - nu * nu;
                const float vx = fmaxf(vx_raw, 0.f), vy = fmaxf(vy_raw, 0.f);
                const float A1 = 2.f * mu * nu + L.c1, A2 = 2.f * (exy - mu * nu) + L.c2;
                const float B1 = mu * mu + nu * nu + L.c1, B2 = vx + vy + L.c2;
#if GS_LOSS_RCP
                const float i1 = gs_rcp(B1), i2 = gs_rcp(B2), inv = i1 * i2;
                const float S = A1 * A2 * inv, S_B2 = S * i2, S_B1 = S * i1;
#else
                const float inv = 1.0f / (B1 * B2);
                const float S = A1 * A2 * inv, S_B2 = S / B2, S_B1 = S / B1;
#endif
                // dS/dExx = -S/B2 (zero where the variance clamp is active); the same factor enters dS/dmu
                const float dExx = vx_raw > 0.f ? -S_B2 : 0.f;
                const float dmu = 2.f * nu * (A2 - A1) * inv - 2.f * mu * S_B1 - 2.f * mu * dExx;
                const float dExy = 2.f * A1 * inv;
                const bool valid = col_in && rho >= R && rho < L.H - R;  // the window lies inside the image
                if (valid && out_col && rho >= r0 && rho < r0 + SH) ssim_sum += S;
                s_d01[buf][t + FHALO] = f2{valid ? dmu : 0.f, valid ? dExx : 0.f};
                s_d2[buf][t + FHALO] = valid ? dExy : 0.f;
            }
        }
        if (tail) {
            // V2: the adjoint row of step n - 1 was that step's phase P - 1, i.e. phase P of the second ring
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int slot = (P - k + K) % K;
                const f2 gk = {Wd.g[k], Wd.g[k]};
                a2p[slot] = k == 0 ? gk * g01 : pfma(gk, g01, a2p[slot]);
                a2s[slot] = k == 0 ? Wd.g[0] * g2 : fmaf(Wd.g[k], g2, a2s[slot]);
            }
            if (emit) {
                constexpr int E2 = (P + 1) % K;
                const float d = xo - yo;
                l1_sum += fabsf(d);
                float g = L.a * (d > 0.f ? 1.f : d < 0.f ? -1.f : 0.f);  // torch: sign(0) = 0
                g -= L.b * (a2p[E2].x + 2.f * xo * a2p[E2].y + yo * a2s[E2]);
                grad[(size_t)o * F + cv] = g;
            }
        }
        __syncthreads();
    };
    for (int nb = 0; nb <= n_end; nb += K) {
#define GS_LOSS_STEP(PH) \
    if (nb + PH <= n_end) step(std::integral_constant<int, PH>{}, nb + PH);
        GS_LOSS_STEP(0)
        GS_LOSS_STEP(1)
        GS_LOSS_STEP(2)
        GS_LOSS_STEP(3)
        GS_LOSS_STEP(4)
        GS_LOSS_STEP(5)
        GS_LOSS_STEP(6)
        GS_LOSS_STEP(7)
        GS_LOSS_STEP(8)
        GS_LOSS_STEP(9)
        GS_LOSS_STEP(10)
#undef GS_LOSS_STEP
    }
    __syncthreads();
    const float tot_s = block_sum<FT / 64>(ssim_sum, s_red);
    __syncthreads();
    const float tot_l = block_sum<FT / 64>(l1_sum, s_red);
    if (t == 0) {
        const int blk = blockIdx.y * gridDim.x + blockIdx.x;
        p_ssim[blk] = tot_s;
        p_l1[blk] = tot_l;
    }
}

// ssim_weight = 0: the L1 term alone (any image size)
constexpr int L1_BLOCKS = 1024;
__global__ void __launch_bounds__(256) l1_only_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                      int64_t n, float a, float *__restrict__ grad,
                                                      float *__restrict__ p_ssim, float *__restrict__ p_l1) {
    __shared__ float s_red[4];
    float l1_sum = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = x[i] - y[i];
        l1_sum += fabsf(d);
        grad[i] = a * (d > 0.f ? 1.f : d < 0.f ? -1.f : 0.f);
    }
    const float tot = block_sum(l1_sum, s_red);
    if (threadIdx.x == 0) {
        p_ssim[blockIdx.x] = 0.f;
        p_l1[blockIdx.x] = tot;
    }
}

// loss_out = (loss, l1 mean, ssim mean)
__global__ void __launch_bounds__(256) loss_finish_kernel(const float *__restrict__ p_ssim,
                                                          const float *__restrict__ p_l1, int n, float inv_l1,
                                                          float inv_ssim, float w, float *__restrict__ loss_out) {
    __shared__ float s_red[4];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        a += p_ssim[i];
        b += p_l1[i];
    }
    const float ssim = block_sum(a, s_red) * inv_ssim;
    __syncthreads();
    const float l1 = block_sum(b, s_red) * inv_l1;
    if (threadIdx.x == 0) {
        loss_out[0] = (1.f - w) * l1 + (w > 0.f ? w * (1.f - ssim) : 0.f);
        loss_out[1] = l1;
        loss_out[2] = w > 0.f ? ssim : 0.f;
    }
}

}  // namespace

static inline int64_t loss_blocks(int32_t H, int32_t W) {
    const int64_t fused = gs_div_up(3 * (int64_t)W, FCW) * gs_div_up(H, SH);
    return fused > L1_BLOCKS ? fused : L1_BLOCKS;
}

extern "C" size_t gs_loss_workspace_bytes(int32_t H, int32_t W) {
    if (H <= 0 || W <= 0) return 0;
    return 2 * gs_align_up(sizeof(float) * loss_blocks(H, W), 256);
}

extern "C" int gs_loss_l1_ssim(const float *pred, const float *target, int32_t H, int32_t W, float ssim_weight,
                               float *grad, float *loss_out, void *workspace, size_t workspace_bytes,
                               gs_stream_t stream) {
    GS_CHECK_ARG(H > 0 && W > 0, "empty image");
    GS_CHECK_ARG(pred && target && grad, "null pointer");
    GS_CHECK_ARG(ssim_weight >= 0.f && ssim_weight <= 1.f, "ssim_weight must be in [0, 1]");
    GS_CHECK_ARG(ssim_weight == 0.f || (H > 2 * R && W > 2 * R), "SSIM needs an image larger than its 11x11 window");
    GS_CHECK_ARG(workspace && workspace_bytes >= gs_loss_workspace_bytes(H, W), "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float *p_ssim = (float *)workspace;
    float *p_l1 = (float *)((char *)p_ssim + gs_align_up(sizeof(float) * loss_blocks(H, W), 256));
    Window Wd;
    {  // torchmetrics _gaussian: exp(-(d / sigma)^2 / 2) in fp32, normalised by its fp32 sum
        float sum = 0.f;
        for (int k = 0; k < K; ++k) {
            const float d = (float)(k - R) / 1.5f;
            Wd.g[k] = expf(-(d * d) / 2.f);
            sum += Wd.g[k];
        }
        for (int k = 0; k < K; ++k) Wd.g[k] /= sum;
    }
    LossGeom L;
    L.H = H;
    L.W = W;
    L.c1 = 0.01f * 0.01f;
    L.c2 = 0.03f * 0.03f;
    const double n_l1 = 3.0 * H * W, n_ss = ssim_weight > 0.f ? 3.0 * (H - 2 * R) * (double)(W - 2 * R) : 1.0;
    L.a = (float)((1.0 - ssim_weight) / n_l1);
    L.b = (float)(ssim_weight / n_ss);
    int64_t nb;
    if (ssim_weight > 0.f) {
        const dim3 grid((unsigned)gs_div_up(3 * (int64_t)W, FCW), (unsigned)gs_div_up(H, SH));
        nb = (int64_t)grid.x * grid.y;
        hipLaunchKernelGGL(loss_fused_kernel, grid, dim3(FT), 0, s, pred, target, L, Wd, grad, p_ssim, p_l1);
    } else {
        const int64_t n = 3 * (int64_t)H * W;
        nb = gs_div_up(n, 256 * 4) < L1_BLOCKS ? gs_div_up(n, 256 * 4) : L1_BLOCKS;
        hipLaunchKernelGGL(l1_only_kernel, dim3((unsigned)nb), dim3(256), 0, s, pred, target, n, L.a, grad, p_ssim, p_l1);
    }
    GS_CHECK_LAUNCH();
    if (loss_out) {
        hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, s, p_ssim, p_l1, (int)nb, (float)(1.0 / n_l1),
                           (float)(1.0 / n_ss), ssim_weight, loss_out);
        GS_CHECK_LAUNCH();
    }
    return 0;
}
