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
// torch autograd would run ~40 elementwise/conv kernels over 5 padded copies of the image for this;
// here it is two tiled passes (separable filter in LDS) plus a one-block reduction:
//   pass 1: the five filtered moments -> S (summed) and the three adjoint maps
//           D_mu = dS/dmu, D_xx = dS/dExx, D_xy = dS/dExy on the interior (planar scratch);
//   pass 2: dL/dx(q) = a sign(x - y) - b [ (g*D_mu)(q) + 2 x(q) (g*D_xx)(q) + y(q) (g*D_xy)(q) ],
//           a = (1 - w) / (3 H W), b = w / (3 (H - 10)(W - 10))  (g is symmetric).
// Images are [H, W, 3] fp32, the layout the rasterizer writes.
#include "gs_common.h"

namespace {

constexpr int R = 5, K = 11;   // window radius / size
#ifndef GS_LOSS_TH
#define GS_LOSS_TH 32
#endif
#ifndef GS_LOSS_NBV
#define GS_LOSS_NBV 4
#endif
constexpr int TW = 32, TH = GS_LOSS_TH;  // output tile of a 256-thread workgroup
constexpr int NB = 4;             // horizontal outputs per thread: each LDS value feeds up to NB outputs
constexpr int NBV = GS_LOSS_NBV;  // vertical outputs per thread
constexpr int RW = TW + 2 * R, RH = TH + 2 * R;

struct Window {
    float g[K];
};

struct LossGeom {
    int32_t H, W;
    float c1, c2, a, b;
};

__device__ __forceinline__ float block_sum(float v, float *s_red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// pass 1: moments -> SSIM sum + adjoint maps D[c][3][H][W] (only interior pixels are written / read)
__global__ void __launch_bounds__(256) ssim_moments_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                           LossGeom L, Window Wd, float *__restrict__ D,
                                                           float *__restrict__ partial) {
    __shared__ float s_x[RH][RW + 1], s_y[RH][RW + 1];
    __shared__ float s_h[5][RH][TW + 1];
    __shared__ float s_red[4];
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;  // tile origin (output coordinates)
    const size_t plane = (size_t)L.H * L.W;
    float ssim_sum = 0.f;
    // the (TH+10) x (TW+10) input region of one channel, NLD elements per thread; the NEXT channel's
    // region is fetched into registers while the current one is filtered (loads are unconditional with
    // clamped coordinates: a predicated load would put a branch and a wait in front of every element)
    constexpr int NLD = (RH * RW + 255) / 256;
    float rx_[NLD], ry_[NLD];
    auto fetch = [&](int c) {
#pragma unroll
        for (int e = 0; e < NLD; ++e) {
            const int i = threadIdx.x + e * 256;
            const int ry = i / RW, rx = i % RW;
            const int gy = min(max(y0 + ry - R, 0), L.H - 1), gx = min(max(x0 + rx - R, 0), L.W - 1);
            const size_t o = ((size_t)gy * L.W + gx) * 3 + c;
            rx_[e] = x[o];
            ry_[e] = y[o];
        }
    };
    auto stage = [&]() {  // values outside the image are never used by an interior output; no masking needed
#pragma unroll
        for (int e = 0; e < NLD; ++e) {
            const int i = threadIdx.x + e * 256;
            if (i < RH * RW) {
                s_x[i / RW][i % RW] = rx_[e];
                s_y[i / RW][i % RW] = ry_[e];
            }
        }
    };
    fetch(0);
    for (int c = 0; c < 3; ++c) {
        stage();
        __syncthreads();
        if (c + 1 < 3) fetch(c + 1);
        for (int i = threadIdx.x; i < RH * (TW / NB); i += 256) {  // horizontal filter of the five moments
            const int ry = i / (TW / NB), tx0 = (i % (TW / NB)) * NB;
            float h[5][NB];
#pragma unroll
            for (int m = 0; m < 5; ++m)
#pragma unroll
                for (int e = 0; e < NB; ++e) h[m][e] = 0.f;
#pragma unroll
            for (int j = 0; j < K + NB - 1; ++j) {  // input column tx0 + j feeds output e with tap j - e
                const float a = s_x[ry][tx0 + j], b = s_y[ry][tx0 + j];
                const float aa = a * a, bb = b * b, ab = a * b;
#pragma unroll
                for (int e = 0; e < NB; ++e) {
                    if (j - e < 0 || j - e >= K) continue;
                    const float w = Wd.g[j - e];
                    h[0][e] = fmaf(w, a, h[0][e]);
                    h[1][e] = fmaf(w, b, h[1][e]);
                    h[2][e] = fmaf(w, aa, h[2][e]);
                    h[3][e] = fmaf(w, bb, h[3][e]);
                    h[4][e] = fmaf(w, ab, h[4][e]);
                }
            }
#pragma unroll
            for (int m = 0; m < 5; ++m)
#pragma unroll
                for (int e = 0; e < NB; ++e) s_h[m][ry][tx0 + e] = h[m][e];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < (TH / NBV) * TW; i += 256) {  // vertical filter + SSIM + adjoints
            const int ty0 = (i / TW) * NBV, tx = i % TW;
            float v[5][NBV];
#pragma unroll
            for (int m = 0; m < 5; ++m)
#pragma unroll
                for (int e = 0; e < NBV; ++e) v[m][e] = 0.f;
#pragma unroll
            for (int j = 0; j < K + NBV - 1; ++j) {
#pragma unroll
                for (int m = 0; m < 5; ++m) {
                    const float a = s_h[m][ty0 + j][tx];
#pragma unroll
                    for (int e = 0; e < NBV; ++e)
                        if (j - e >= 0 && j - e < K) v[m][e] = fmaf(Wd.g[j - e], a, v[m][e]);
                }
            }
#pragma unroll
            for (int e = 0; e < NBV; ++e) {
                const int gy = y0 + ty0 + e, gx = x0 + tx;
                if (gy < R || gy >= L.H - R || gx < R || gx >= L.W - R) continue;  // window must lie inside the image
                const float mu = v[0][e], nu = v[1][e], exx = v[2][e], eyy = v[3][e], exy = v[4][e];
                const float vx_raw = exx - mu * mu, vy_raw = eyy - nu * nu;
                const float vx = fmaxf(vx_raw, 0.f), vy = fmaxf(vy_raw, 0.f);
                const float A1 = 2.f * mu * nu + L.c1, A2 = 2.f * (exy - mu * nu) + L.c2;
                const float B1 = mu * mu + nu * nu + L.c1, B2 = vx + vy + L.c2;
                const float inv = 1.0f / (B1 * B2);
                const float S = A1 * A2 * inv;
                ssim_sum += S;
                // dS/dExx = -S/B2 (zero where the variance clamp is active); the same factor enters dS/dmu
                const float dExx = vx_raw > 0.f ? -S / B2 : 0.f;
                const float dmu = 2.f * nu * (A2 - A1) * inv - 2.f * mu * S / B1 - 2.f * mu * dExx;
                const float dExy = 2.f * A1 * inv;
                const size_t o = (size_t)gy * L.W + gx;
                D[(size_t)(c * 3 + 0) * plane + o] = dmu;
                D[(size_t)(c * 3 + 1) * plane + o] = dExx;
                D[(size_t)(c * 3 + 2) * plane + o] = dExy;
            }
        }
        __syncthreads();
    }
    const float tot = block_sum(ssim_sum, s_red);
    if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = tot;
}

// pass 2: gradient = L1 term + filtered adjoints
__global__ void __launch_bounds__(256) loss_grad_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                        LossGeom L, Window Wd, const float *__restrict__ D,
                                                        float *__restrict__ grad, float *__restrict__ partial,
                                                        int use_ssim) {
    __shared__ float s_d[3][RH][RW + 1];
    __shared__ float s_h[3][RH][TW + 1];
    __shared__ float s_red[4];
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const size_t plane = (size_t)L.H * L.W;
    float l1_sum = 0.f;
    constexpr int NLD = (RH * RW + 255) / 256;
    float rd[3][NLD];
    auto fetch = [&](int c) {
#pragma unroll
        for (int e = 0; e < NLD; ++e) {
            const int i = threadIdx.x + e * 256;
            const int ry = i / RW, rx = i % RW;
            const int gy = y0 + ry - R, gx = x0 + rx - R;
            const bool in = gy >= R && gy < L.H - R && gx >= R && gx < L.W - R;  // adjoints live on the interior
            const size_t o = (size_t)min(max(gy, 0), L.H - 1) * L.W + min(max(gx, 0), L.W - 1);
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const float v = D[(size_t)(c * 3 + m) * plane + o];  // unconditional load, masked afterwards
                rd[m][e] = in ? v : 0.f;
            }
        }
    };
    if (use_ssim) fetch(0);
    for (int c = 0; c < 3; ++c) {
        if (use_ssim) {
#pragma unroll
            for (int e = 0; e < NLD; ++e) {
                const int i = threadIdx.x + e * 256;
                if (i < RH * RW) {
#pragma unroll
                    for (int m = 0; m < 3; ++m) s_d[m][i / RW][i % RW] = rd[m][e];
                }
            }
            __syncthreads();
            if (c + 1 < 3) fetch(c + 1);
            for (int i = threadIdx.x; i < RH * (TW / NB); i += 256) {
                const int ry = i / (TW / NB), tx0 = (i % (TW / NB)) * NB;
                float h[3][NB];
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int e = 0; e < NB; ++e) h[m][e] = 0.f;
#pragma unroll
                for (int j = 0; j < K + NB - 1; ++j)
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        const float a = s_d[m][ry][tx0 + j];
#pragma unroll
                        for (int e = 0; e < NB; ++e)
                            if (j - e >= 0 && j - e < K) h[m][e] = fmaf(Wd.g[j - e], a, h[m][e]);
                    }
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int e = 0; e < NB; ++e) s_h[m][ry][tx0 + e] = h[m][e];
            }
            __syncthreads();
        }
        for (int i = threadIdx.x; i < (TH / NBV) * TW; i += 256) {
            const int ty0 = (i / TW) * NBV, tx = i % TW;
            float gsum[3][NBV];
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int e = 0; e < NBV; ++e) gsum[m][e] = 0.f;
            if (use_ssim) {
#pragma unroll
                for (int j = 0; j < K + NBV - 1; ++j)
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        const float a = s_h[m][ty0 + j][tx];
#pragma unroll
                        for (int e = 0; e < NBV; ++e)
                            if (j - e >= 0 && j - e < K) gsum[m][e] = fmaf(Wd.g[j - e], a, gsum[m][e]);
                    }
            }
#pragma unroll
            for (int e = 0; e < NBV; ++e) {
                const int gy = y0 + ty0 + e, gx = x0 + tx;
                if (gy >= L.H || gx >= L.W) continue;
                const size_t o = ((size_t)gy * L.W + gx) * 3 + c;
                const float xv = x[o], yv = y[o], d = xv - yv;
                l1_sum += fabsf(d);
                float g = L.a * (d > 0.f ? 1.f : d < 0.f ? -1.f : 0.f);  // torch: sign(0) = 0
                if (use_ssim) g -= L.b * (gsum[0][e] + 2.f * xv * gsum[1][e] + yv * gsum[2][e]);
                grad[o] = g;
            }
        }
        __syncthreads();
    }
    const float tot = block_sum(l1_sum, s_red);
    if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = tot;
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

static inline int64_t loss_blocks(int32_t H, int32_t W) { return gs_div_up(W, TW) * gs_div_up(H, TH); }

extern "C" size_t gs_loss_workspace_bytes(int32_t H, int32_t W) {
    if (H <= 0 || W <= 0) return 0;
    return gs_align_up(sizeof(float) * 9 * (size_t)H * W, 256) + 2 * gs_align_up(sizeof(float) * loss_blocks(H, W), 256);
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
    const int64_t nb = loss_blocks(H, W);
    float *D = (float *)workspace;
    float *p_ssim = (float *)((char *)workspace + gs_align_up(sizeof(float) * 9 * (size_t)H * W, 256));
    float *p_l1 = (float *)((char *)p_ssim + gs_align_up(sizeof(float) * nb, 256));
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
    const dim3 grid((unsigned)gs_div_up(W, TW), (unsigned)gs_div_up(H, TH));
    if (ssim_weight > 0.f) {
        hipLaunchKernelGGL(ssim_moments_kernel, grid, dim3(256), 0, s, pred, target, L, Wd, D, p_ssim);
        GS_CHECK_LAUNCH();
    } else if (loss_out) {
        GS_HIP(hipMemsetAsync(p_ssim, 0, sizeof(float) * nb, s));
    }
    hipLaunchKernelGGL(loss_grad_kernel, grid, dim3(256), 0, s, pred, target, L, Wd, D, grad, p_l1,
                       ssim_weight > 0.f ? 1 : 0);
    GS_CHECK_LAUNCH();
    if (loss_out) {
        hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, s, p_ssim, p_l1, (int)nb, (float)(1.0 / n_l1),
                           (float)(1.0 / n_ss), ssim_weight, loss_out);
        GS_CHECK_LAUNCH();
    }
    return 0;
}
