// densify.hip -- Gaussian3ds.adaptive_control of the reference (splatter.py:122-228) as three launches
// (SURVEY.md section 8f-2).
//
// The reference prunes with four boolean-mask gathers, builds two more masks, clones with five masked
// gathers, splits with a batched covariance (q2r, two bmm), a batched Cholesky and two
// MultivariateNormal samples, and concatenates fifteen tensors -- some forty torch kernels and four host
// synchronisations (`.any()`, `print(...sum())`) every --n_adaptive_control iterations.  Here:
//   D1 densify_classify_kernel : class of every Gaussian (deleted / kept / kept+cloned / kept+split) and
//                                per-workgroup counts;
//   D2 densify_scan_kernel     : exclusive scan of the counts, totals -> device counters;
//   D3 densify_apply_kernel    : every Gaussian writes itself to its kept rank (split ones with the new
//                                scale and the first sample), its clone to K + clone rank, its second
//                                sample to K + C + split rank -- the reference's output order.
// The two standard-normal blocks of the split samples are inputs (the caller draws them once the
// counts are known, exactly when the reference's MultivariateNormal.sample() would); a sample is
// pos + chol(R S^2 R^T) eps, which is what torch's MultivariateNormal computes (utils.py:391-402).
// Compiled with -ffp-contract=off: the masks compare fp32 norms against thresholds like torch does.
#include "gs_common.h"

namespace {

enum { CLS_DELETE = 0, CLS_KEEP = 1, CLS_CLONE = 2, CLS_SPLIT = 3 };

struct DensifyParams {
    float taus, delete_thresh, grad_thresh, clone_dt, opa_thresh, split_div, split_sub;
    int scale_act, agg_mean, use_clone, use_split, color_dim;
};

__device__ __forceinline__ float act_norm(const float *s, int scale_act) {
    const float a = scale_act == 0 ? fabsf(s[0]) : expf(s[0]);
    const float b = scale_act == 0 ? fabsf(s[1]) : expf(s[1]);
    const float c = scale_act == 0 ? fabsf(s[2]) : expf(s[2]);
    return sqrtf(a * a + b * b + c * c);
}

__device__ __forceinline__ int classify(const float *scale, const float *opa, const float *grad, int64_t i,
                                        const DensifyParams &P) {
    const float s[3] = {scale[i * 3], scale[i * 3 + 1], scale[i * 3 + 2]};
    const float norm = act_norm(s, P.scale_act);
    if (!(opa[i] > P.opa_thresh && norm < P.delete_thresh)) return CLS_DELETE;  // splatter.py:143-145
    const float g0 = fabsf(grad[i * 3]), g1 = fabsf(grad[i * 3 + 1]), g2 = fabsf(grad[i * 3 + 2]);
    const float agg = P.agg_mean ? (g0 + g1 + g2) / 3.0f : fmaxf(g0, fmaxf(g1, g2));
    if (!(agg > P.grad_thresh)) return CLS_KEEP;  // :158-162
    if (norm > P.taus) return P.use_split ? CLS_SPLIT : CLS_KEEP;  // :170-174
    return P.use_clone ? CLS_CLONE : CLS_KEEP;
}

// block-wide exclusive scan of three 0/1 flags; returns this thread's three ranks and the block totals
__device__ __forceinline__ void block_scan3(uint32_t f0, uint32_t f1, uint32_t f2, uint32_t r[3], uint32_t tot[3],
                                            uint32_t (*s_w)[4]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t i0 = gs_wave_incl_scan_u32(f0), i1 = gs_wave_incl_scan_u32(f1), i2 = gs_wave_incl_scan_u32(f2);
    if (lane == 63) {
        s_w[0][wave] = i0;
        s_w[1][wave] = i1;
        s_w[2][wave] = i2;
    }
    __syncthreads();
    const uint32_t incl[3] = {i0, i1, i2}, f[3] = {f0, f1, f2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        uint32_t off = 0, t = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            off += w < wave ? s_w[k][w] : 0;
            t += s_w[k][w];
        }
        r[k] = off + incl[k] - f[k];
        tot[k] = t;
    }
}

__global__ void __launch_bounds__(256) densify_classify_kernel(const float *__restrict__ scale,
                                                               const float *__restrict__ opa,
                                                               const float *__restrict__ grad, int64_t n,
                                                               DensifyParams P, uint32_t *__restrict__ cls,
                                                               uint32_t *__restrict__ block_counts) {
    __shared__ uint32_t s_w[3][4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c = i < n ? classify(scale, opa, grad, i, P) : CLS_DELETE;
    if (i < n) cls[i] = (uint32_t)c;
    uint32_t r[3], tot[3];
    block_scan3(c != CLS_DELETE, c == CLS_CLONE, c == CLS_SPLIT, r, tot, s_w);
    if (threadIdx.x < 3) block_counts[(size_t)blockIdx.x * 3 + threadIdx.x] = tot[threadIdx.x];
}

// one workgroup: exclusive scan of block_counts[nblk][3] in place; counts = (kept, cloned, split, total)
__global__ void __launch_bounds__(1024) densify_scan_kernel(uint32_t *__restrict__ block_counts, int nblk,
                                                            long long *__restrict__ counts) {
    __shared__ uint32_t s_wave[3][16];
    __shared__ uint32_t s_carry[3];
    if (threadIdx.x < 3) s_carry[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < nblk; base += 1024) {
        const int i = base + threadIdx.x;
        uint32_t v[3], incl[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            v[k] = i < nblk ? block_counts[(size_t)i * 3 + k] : 0;
            incl[k] = gs_wave_incl_scan_u32(v[k]);
            if (lane == 63) s_wave[k][wave] = incl[k];
        }
        __syncthreads();
        uint32_t next[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            uint32_t off = 0, t = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                off += w < wave ? s_wave[k][w] : 0;
                t += s_wave[k][w];
            }
            if (i < nblk) block_counts[(size_t)i * 3 + k] = s_carry[k] + off + incl[k] - v[k];
            next[k] = s_carry[k] + t;
        }
        __syncthreads();
        if (threadIdx.x < 3) s_carry[threadIdx.x] = next[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        counts[0] = s_carry[0];
        counts[1] = s_carry[1];
        counts[2] = s_carry[2];
        counts[3] = (long long)s_carry[0] + s_carry[1] + s_carry[2];
    }
}

// lower Cholesky factor of Sigma = (R S)(R S)^T, R = q2r(quat) (utils.py:318-333), S = act(scale) (+1e-4 for abs)
__device__ __forceinline__ void chol_cov(const float *q_raw, const float *s_raw, int scale_act, float L[6]) {
    const float nq = sqrtf(q_raw[0] * q_raw[0] + q_raw[1] * q_raw[1] + q_raw[2] * q_raw[2] + q_raw[3] * q_raw[3]);
    const float w = q_raw[0] / nq, x = q_raw[1] / nq, y = q_raw[2] / nq, z = q_raw[3] / nq;
    const float R[3][3] = {{1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y},
                           {2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x},
                           {2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y}};
    float s[3], M[3][3], C[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) s[k] = scale_act == 0 ? fabsf(s_raw[k]) + 1e-4f : expf(s_raw[k]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) M[i][k] = R[i][k] * s[k];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[i][j] = M[i][0] * M[j][0] + M[i][1] * M[j][1] + M[i][2] * M[j][2];
    L[0] = sqrtf(C[0][0]);
    L[1] = C[1][0] / L[0];
    L[2] = sqrtf(C[1][1] - L[1] * L[1]);
    L[3] = C[2][0] / L[0];
    L[4] = (C[2][1] - L[3] * L[1]) / L[2];
    L[5] = sqrtf(C[2][2] - L[3] * L[3] - L[4] * L[4]);
}

struct DensifyIO {
    const float *pos, *quat, *scale, *opa, *rgb, *grad, *eps1, *eps2;
    float *o_pos, *o_quat, *o_scale, *o_opa, *o_rgb;
};

__device__ __forceinline__ void write_gaussian(const DensifyIO &IO, int64_t dst, const float p[3], const float q[4],
                                               const float s[3], float o, int64_t src, int color_dim) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        IO.o_pos[dst * 3 + k] = p[k];
        IO.o_scale[dst * 3 + k] = s[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) IO.o_quat[dst * 4 + k] = q[k];
    IO.o_opa[dst] = o;
    for (int k = 0; k < color_dim; ++k) IO.o_rgb[dst * color_dim + k] = IO.rgb[src * color_dim + k];
}

__global__ void __launch_bounds__(256) densify_apply_kernel(DensifyIO IO, int64_t n, DensifyParams P,
                                                            const uint32_t *__restrict__ cls,
                                                            const uint32_t *__restrict__ block_offsets,
                                                            const long long *__restrict__ counts, int64_t capacity,
                                                            int64_t n_eps) {
    __shared__ uint32_t s_w[3][4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c = i < n ? (int)cls[i] : CLS_DELETE;
    uint32_t r[3], tot[3];
    block_scan3(c != CLS_DELETE, c == CLS_CLONE, c == CLS_SPLIT, r, tot, s_w);
    if (counts[3] > capacity || counts[2] > n_eps || c == CLS_DELETE) return;  // the host checks the same counts
    const int64_t K = counts[0], Cn = counts[1];
    const int64_t kept = block_offsets[(size_t)blockIdx.x * 3 + 0] + r[0];
    float p[3], q[4], s[3], g[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        p[k] = IO.pos[i * 3 + k];
        s[k] = IO.scale[i * 3 + k];
        g[k] = IO.grad[i * 3 + k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = IO.quat[i * 4 + k];
    const float o = IO.opa[i];
    if (c == CLS_SPLIT) {
        const int64_t sr = block_offsets[(size_t)blockIdx.x * 3 + 2] + r[2];
        float L[6];
        chol_cov(q, s, P.scale_act, L);  // from the ORIGINAL scale (splatter.py:202)
        float s2[3], p1[3], p2[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) s2[k] = P.scale_act == 0 ? s[k] / P.split_div : s[k] - P.split_sub;
        const float *e1 = IO.eps1 + sr * 3, *e2 = IO.eps2 + sr * 3;
        p1[0] = p[0] + L[0] * e1[0];
        p1[1] = p[1] + (L[1] * e1[0] + L[2] * e1[1]);
        p1[2] = p[2] + (L[3] * e1[0] + L[4] * e1[1] + L[5] * e1[2]);
        p2[0] = p[0] + L[0] * e2[0];
        p2[1] = p[1] + (L[1] * e2[0] + L[2] * e2[1]);
        p2[2] = p[2] + (L[3] * e2[0] + L[4] * e2[1] + L[5] * e2[2]);
        write_gaussian(IO, kept, p1, q, s2, o, i, P.color_dim);
        write_gaussian(IO, K + Cn + sr, p2, q, s2, o, i, P.color_dim);
    } else {
        write_gaussian(IO, kept, p, q, s, o, i, P.color_dim);
        if (c == CLS_CLONE) {
            const int64_t cr = block_offsets[(size_t)blockIdx.x * 3 + 1] + r[1];
            float pc[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) pc[k] = p[k] - g[k] * P.clone_dt;  // splatter.py:178
            write_gaussian(IO, K + cr, pc, q, s, o, i, P.color_dim);
        }
    }
}

int fill_params(const gs_densify_opts *o, DensifyParams &P) {
    GS_CHECK_ARG(o != nullptr, "opts is null");
    GS_CHECK_ARG(o->scale_activation == 0 || o->scale_activation == 1, "scale_activation must be 0 (abs) or 1 (exp)");
    GS_CHECK_ARG(o->grad_aggregation == 0 || o->grad_aggregation == 1, "grad_aggregation must be 0 (max) or 1 (mean)");
    GS_CHECK_ARG(o->color_dim == 3 || o->color_dim == 27 || o->color_dim == 48, "color_dim must be 3, 27 or 48");
    P.taus = o->taus;
    P.delete_thresh = o->delete_thresh;
    P.grad_thresh = o->grad_thresh;
    P.clone_dt = o->clone_dt;
    P.opa_thresh = (float)(-log(1.0 / 0.02 - 1.0));  // inverse_sigmoid(0.02), utils.py:350-351
    P.split_div = 1.6f;
    P.split_sub = (float)log(1.6);
    P.scale_act = o->scale_activation;
    P.agg_mean = o->grad_aggregation;
    P.use_clone = o->use_clone != 0;
    P.use_split = o->use_split != 0;
    P.color_dim = o->color_dim;
    return 0;
}

struct DensifyWs {
    uint32_t *cls, *block_counts;
    size_t bytes;
};
DensifyWs carve(void *base, int64_t n) {
    DensifyWs w;
    const int64_t nblk = gs_div_up(n > 0 ? n : 1, 256);
    w.cls = (uint32_t *)base;
    const size_t a = gs_align_up(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1), 256);
    w.block_counts = base ? (uint32_t *)((char *)base + a) : nullptr;
    w.bytes = a + gs_align_up(sizeof(uint32_t) * 3 * nblk, 256);
    return w;
}

}  // namespace

extern "C" size_t gs_densify_workspace_bytes(int64_t N) { return N < 0 ? 0 : carve(nullptr, N).bytes; }

extern "C" int gs_densify_classify(const float *scale, const float *opa, const float *grad, int64_t N,
                                   const gs_densify_opts *opts, int64_t *counts_dev, void *workspace,
                                   size_t workspace_bytes, gs_stream_t stream) {
    DensifyParams P;
    int rc = fill_params(opts, P);
    if (rc) return rc;
    GS_CHECK_ARG(N >= 0 && N < (1ll << 31), "N out of range");
    GS_CHECK_ARG(counts_dev != nullptr, "counts_dev is null");
    GS_CHECK_ARG(workspace && workspace_bytes >= gs_densify_workspace_bytes(N), "workspace too small");
    GS_CHECK_ARG(N == 0 || (scale && opa && grad), "null pointer");
    hipStream_t s = (hipStream_t)stream;
    DensifyWs w = carve(workspace, N);
    const int nblk = (int)gs_div_up(N > 0 ? N : 1, 256);
    hipLaunchKernelGGL(densify_classify_kernel, dim3(nblk), dim3(256), 0, s, scale, opa, grad, N, P, w.cls,
                       w.block_counts);
    GS_CHECK_LAUNCH();
    hipLaunchKernelGGL(densify_scan_kernel, dim3(1), dim3(1024), 0, s, w.block_counts, nblk, (long long *)counts_dev);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_densify_apply(const float *pos, const float *quat, const float *scale, const float *opa,
                                const float *rgb, const float *grad, int64_t N, const gs_densify_opts *opts,
                                const float *eps1, const float *eps2, int64_t n_eps, float *out_pos,
                                float *out_quat, float *out_scale, float *out_opa, float *out_rgb,
                                int64_t capacity, const int64_t *counts_dev, void *workspace,
                                size_t workspace_bytes, gs_stream_t stream) {
    DensifyParams P;
    int rc = fill_params(opts, P);
    if (rc) return rc;
    GS_CHECK_ARG(N >= 0 && N < (1ll << 31) && capacity >= 0 && n_eps >= 0, "size out of range");
    GS_CHECK_ARG(counts_dev != nullptr, "counts_dev is null");
    GS_CHECK_ARG(workspace && workspace_bytes >= gs_densify_workspace_bytes(N), "workspace too small");
    if (N == 0) return 0;
    GS_CHECK_ARG(pos && quat && scale && opa && rgb && grad, "null input");
    if (capacity == 0) return 0;  // nothing can be written (the caller saw total == 0)
    GS_CHECK_ARG(out_pos && out_quat && out_scale && out_opa && out_rgb, "null output");
    GS_CHECK_ARG(n_eps == 0 || (eps1 && eps2), "null normal draws");
    DensifyWs w = carve(workspace, N);
    DensifyIO IO = {pos, quat, scale, opa, rgb, grad, eps1, eps2, out_pos, out_quat, out_scale, out_opa, out_rgb};
    hipLaunchKernelGGL(densify_apply_kernel, dim3((unsigned)gs_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, IO, N,
                       P, w.cls, w.block_counts, (const long long *)counts_dev, capacity, n_eps);
    GS_CHECK_LAUNCH();
    return 0;
}
