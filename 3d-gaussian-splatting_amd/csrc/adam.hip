// adam.hip -- fused Adam step over the flat parameter bucket.
//
// Replaces `torch.optim.Adam([...5 groups...], betas=(0.9, 0.99))` + `optimizer.step()` of the
// reference trainer (train.py:59-67, :113): default eps = 1e-8, no weight decay, no amsgrad.  The
// reference pays one multi-tensor launch chain per group and keeps its five parameter tensors
// separate; here all parameters, gradients and both moment buffers are four flat fp32 arrays
// (gs_dp.FlatGaussianParams), a group is a contiguous index range with its own learning rate, and
// ONE launch updates everything: 16 B read + 12 B written per parameter -- a pure HBM stream.
// The densification statistic of train.py:145-154 (`accum_max_grad = max(|pos.grad|, accum)` or
// `+= |pos.grad|`) rides along for the index range of `pos`, so the gradient is read once.
// Arithmetic follows torch's _single_tensor_adam:
//   m <- m + (g - m)(1 - b1);  v <- v b2 + (1 - b2) g g
//   p <- p - (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// with the bias corrections evaluated on the host in double, as torch does.
#include "gs_common.h"

namespace {

#define GS_ADAM_MAX_GROUPS 8
// Gradient and moments are touched once per step.  Beyond the 256 MiB Infinity Cache (16 bytes of arrays per parameter:
// from ~17 M parameters on) they only evict each other on their way through it: non-temporal loads / stores there
// (measured, tools/adam_bw.py: 33.6 M parameters 0.188 -> 0.145 ms = 63 % -> 81 % of the HBM peak; at 14 M parameters,
// where the four arrays still fit the cache, the plain accesses are faster: 0.060 against 0.065 ms).
#define GS_ADAM_NT_BYTES (300ull << 20)

struct AdamGroups {
    int64_t end[GS_ADAM_MAX_GROUPS];  // group k covers [end[k-1], end[k])
    float step_size[GS_ADAM_MAX_GROUPS];  // lr_k / (1 - b1^t)
    int32_t n;
};

__device__ __forceinline__ float group_step(const AdamGroups &G, int64_t i) {
    float s = G.step_size[0];
#pragma unroll
    for (int k = 1; k < GS_ADAM_MAX_GROUPS; ++k)
        if (k < G.n && i >= G.end[k - 1]) s = G.step_size[k];
    return s;
}

// (the per-element arithmetic lives in gs_common.h: gs_adam_one -- shared with the fused backward + Adam kernel)
__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float step_size, float one_m_b1,
                                         float b2, float one_m_b2, float inv_bc2_sqrt, float eps) {
    gs_adam_one(p, g, m, v, step_size, one_m_b1, b2, one_m_b2, inv_bc2_sqrt, eps);
}

template <bool NT>
__device__ __forceinline__ void adam_range(float *__restrict__ param, const float *__restrict__ grad,
                                           float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq,
                                           int64_t lo, int64_t hi, const AdamGroups &G, float one_m_b1, float b2,
                                           float one_m_b2, float inv_bc2_sqrt, float eps,
                                           float *__restrict__ stat, int64_t stat_begin, int64_t stat_end,
                                           int stat_mode, int64_t moment_base, float grad_scale = 1.0f) {
    // grad_scale: the gradient is used as grad * grad_scale (x * 1.0f == x: the plain step is unchanged bit for bit).
    // The view-parallel exchange SUMS the ranks' gradients and leaves the 1 / world of the mean to this kernel.
    // the moments may be a SHARD that starts at element moment_base (a multiple of 4) of the flat index space
    exp_avg -= moment_base;
    exp_avg_sq -= moment_base;
    // elements [lo, hi) of the (16-byte aligned) arrays: whole float4s [q0, q1) + up to three scalars at either end
    const int64_t q0 = (lo + 3) >> 2, q1 = hi >> 2;
    for (int64_t q = q0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < q1; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = q << 2;
        typedef float nt4 __attribute__((ext_vector_type(4)));
        float4 p = reinterpret_cast<float4 *>(param)[q];
        float4 g, m, v;
        if constexpr (NT) {  // streaming (non-temporal) accesses for what is touched once per step
            const nt4 g_ = __builtin_nontemporal_load(reinterpret_cast<const nt4 *>(grad) + q);
            const nt4 m_ = __builtin_nontemporal_load(reinterpret_cast<nt4 *>(exp_avg) + q);
            const nt4 v_ = __builtin_nontemporal_load(reinterpret_cast<nt4 *>(exp_avg_sq) + q);
            g = make_float4(g_.x, g_.y, g_.z, g_.w);
            m = make_float4(m_.x, m_.y, m_.z, m_.w);
            v = make_float4(v_.x, v_.y, v_.z, v_.w);
        } else {
            g = reinterpret_cast<const float4 *>(grad)[q];
            m = reinterpret_cast<float4 *>(exp_avg)[q];
            v = reinterpret_cast<float4 *>(exp_avg_sq)[q];
        }
        g.x *= grad_scale; g.y *= grad_scale; g.z *= grad_scale; g.w *= grad_scale;
        const float s0 = group_step(G, i), s3 = group_step(G, i + 3);
        const bool uniform = s0 == s3;  // a float4 straddles a group boundary at most five times per launch
        adam_one(p.x, g.x, m.x, v.x, s0, one_m_b1, b2, one_m_b2, inv_bc2_sqrt, eps);
        adam_one(p.y, g.y, m.y, v.y, uniform ? s0 : group_step(G, i + 1), one_m_b1, b2, one_m_b2, inv_bc2_sqrt, eps);
        adam_one(p.z, g.z, m.z, v.z, uniform ? s0 : group_step(G, i + 2), one_m_b1, b2, one_m_b2, inv_bc2_sqrt, eps);
        adam_one(p.w, g.w, m.w, v.w, s3, one_m_b1, b2, one_m_b2, inv_bc2_sqrt, eps);
        reinterpret_cast<float4 *>(param)[q] = p;
        if constexpr (NT) {
            __builtin_nontemporal_store(nt4{m.x, m.y, m.z, m.w}, reinterpret_cast<nt4 *>(exp_avg) + q);
            __builtin_nontemporal_store(nt4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt4 *>(exp_avg_sq) + q);
        } else {
            reinterpret_cast<float4 *>(exp_avg)[q] = m;
            reinterpret_cast<float4 *>(exp_avg_sq)[q] = v;
        }
        if (stat_mode && i + 3 >= stat_begin && i < stat_end) {
            const float ga[4] = {fabsf(g.x), fabsf(g.y), fabsf(g.z), fabsf(g.w)};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t k = i + e;
                if (k >= stat_begin && k < stat_end) {
                    float *d = stat + (k - stat_begin);
                    *d = stat_mode == 1 ? fmaxf(*d, ga[e]) : *d + ga[e];
                }
            }
        }
    }
    // ragged ends (every element's update is independent of how the range is cut)
    const int64_t head_end = (q0 << 2) < hi ? (q0 << 2) : hi;
    const int64_t tail_begin = (q1 << 2) > head_end ? (q1 << 2) : head_end;
    const int64_t gt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t t = -1;
    if (lo + gt < head_end)
        t = lo + gt;
    else if (gt >= 4 && tail_begin + (gt - 4) < hi)
        t = tail_begin + (gt - 4);
    if (t >= 0) {
        float p = param[t], m = exp_avg[t], v = exp_avg_sq[t];
        const float g = grad[t] * grad_scale;
        adam_one(p, g, m, v, group_step(G, t), one_m_b1, b2, one_m_b2, inv_bc2_sqrt, eps);
        param[t] = p;
        exp_avg[t] = m;
        exp_avg_sq[t] = v;
        if (stat_mode && t >= stat_begin && t < stat_end) {
            float *d = stat + (t - stat_begin);
            *d = stat_mode == 1 ? fmaxf(*d, fabsf(g)) : *d + fabsf(g);
        }
    }
}

template <bool NT>
__global__ void __launch_bounds__(256) adam_kernel(float *__restrict__ param, const float *__restrict__ grad,
                                                   float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq,
                                                   int64_t lo, int64_t hi, AdamGroups G, float one_m_b1, float b2,
                                                   float one_m_b2, float inv_bc2_sqrt, float eps,
                                                   float *__restrict__ stat, int64_t stat_begin, int64_t stat_end,
                                                   int stat_mode, int64_t moment_base,
                                                   const unsigned long long *__restrict__ skip_if_nonzero) {
    // a frame that overflowed its workspace was rendered empty: its all-zero gradient must not move the parameters by
    // momentum (the flag is the frame's device-side overflow counter: no host synchronisation; uniform branch)
    if (skip_if_nonzero && *skip_if_nonzero) return;
    adam_range<NT>(param, grad, exp_avg, exp_avg_sq, lo, hi, G, one_m_b1, b2, one_m_b2, inv_bc2_sqrt, eps, stat,
                   stat_begin, stat_end, stat_mode, moment_base);
}

// Several element ranges in ONE launch (blockIdx.y = range): a slice of the view-parallel exchange is one range of
// Gaussians in each of the five parameter arrays (gs_dp.py) -- five launches of ~2 us of work each would cost more in
// dependent-launch latency than they compute.
#define GS_ADAM_MAX_RANGES 8
struct AdamRanges {
    int64_t lo[GS_ADAM_MAX_RANGES], hi[GS_ADAM_MAX_RANGES], moment_base[GS_ADAM_MAX_RANGES];
};
template <bool NT>
__global__ void __launch_bounds__(256) adam_multi_kernel(float *__restrict__ param, const float *__restrict__ grad,
                                                         float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq,
                                                         AdamRanges R, AdamGroups G, float one_m_b1, float b2,
                                                         float one_m_b2, float inv_bc2_sqrt, float eps,
                                                         float *__restrict__ stat, int64_t stat_begin, int64_t stat_end,
                                                         int stat_mode,
                                                         const unsigned long long *__restrict__ skip_if_nonzero,
                                                         float grad_scale) {
    if (skip_if_nonzero && *skip_if_nonzero) return;
    const int r = blockIdx.y;
    adam_range<NT>(param, grad, exp_avg, exp_avg_sq, R.lo[r], R.hi[r], G, one_m_b1, b2, one_m_b2, inv_bc2_sqrt, eps,
                   stat, stat_begin, stat_end, stat_mode, R.moment_base[r], grad_scale);
}

// the statistic alone (view-parallel training: it must see this rank's OWN gradient, before the all-reduce)
__global__ void __launch_bounds__(256) grad_stat_kernel(const float *__restrict__ grad, float *__restrict__ stat,
                                                        int64_t n, int mode) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float g = fabsf(grad[i]);
        stat[i] = mode == 1 ? fmaxf(stat[i], g) : stat[i] + g;
    }
}

}  // namespace

extern "C" int gs_grad_stat_update(const float *grad, float *stat, int64_t n, int32_t stat_mode, gs_stream_t stream) {
    GS_CHECK_ARG(n >= 0, "n < 0");
    GS_CHECK_ARG(stat_mode == 1 || stat_mode == 2, "stat_mode must be 1 (max) or 2 (sum)");
    if (n == 0) return 0;
    GS_CHECK_ARG(grad && stat, "null pointer");
    int64_t blocks = gs_div_up(n, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(grad_stat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, grad, stat, n,
                       (int)stat_mode);
    GS_CHECK_LAUNCH();
    return 0;
}

static int adam_step_impl(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                          int64_t range_begin, int64_t range_end, int32_t n_groups, const int64_t *group_end,
                          const float *lr, float beta1, float beta2, float eps, int64_t step, float *grad_stat,
                          int64_t stat_begin, int64_t stat_end, int32_t stat_mode, gs_stream_t stream,
                          int64_t moment_base = 0, const unsigned long long *skip_if_nonzero = nullptr) {
    GS_CHECK_ARG(n >= 0, "n < 0");
    GS_CHECK_ARG(range_begin >= 0 && range_begin <= range_end && range_end <= n, "bad element range");
    GS_CHECK_ARG(moment_base >= 0 && (moment_base & 3) == 0 && moment_base <= range_begin,
                 "moment_base must be a multiple of 4 and <= range_begin");
    GS_CHECK_ARG(n_groups >= 1 && n_groups <= GS_ADAM_MAX_GROUPS, "n_groups must be in [1, 8]");
    GS_CHECK_ARG(group_end && lr, "null group table");
    GS_CHECK_ARG(step >= 1, "step counts from 1 (torch.optim.Adam increments before the update)");
    GS_CHECK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "bad hyper-parameters");
    GS_CHECK_ARG(stat_mode >= 0 && stat_mode <= 2, "stat_mode must be 0 (off), 1 (max) or 2 (sum)");
    if (range_end == range_begin) return 0;
    GS_CHECK_ARG(param && grad && exp_avg && exp_avg_sq, "null pointer");
    GS_CHECK_ARG((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                 "buffers must be 16-byte aligned");
    GS_CHECK_ARG(!stat_mode || (grad_stat && stat_begin >= 0 && stat_begin <= stat_end && stat_end <= n),
                 "bad statistic range");
    AdamGroups G;
    int64_t prev = 0;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    for (int k = 0; k < GS_ADAM_MAX_GROUPS; ++k) {
        const int kk = k < n_groups ? k : n_groups - 1;
        GS_CHECK_ARG(group_end[kk] >= prev && group_end[kk] <= n, "group_end must be ascending and <= n");
        prev = group_end[kk];
        G.end[k] = group_end[kk];
        G.step_size[k] = (float)((double)lr[kk] / bc1);
    }
    GS_CHECK_ARG(group_end[n_groups - 1] == n, "the groups must cover [0, n)");
    G.n = n_groups;
    const int64_t len = range_end - range_begin;
    int64_t blocks = gs_div_up(gs_div_up(len, 4) > 0 ? gs_div_up(len, 4) : 1, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride beyond 16 workgroups per CU
    if (blocks < 1) blocks = 1;
    if ((unsigned long long)len * 16ull > GS_ADAM_NT_BYTES)
        hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad,
                           exp_avg, exp_avg_sq, range_begin, range_end, G, 1.0f - beta1, beta2, 1.0f - beta2,
                           (float)(1.0 / sqrt(bc2)), eps, grad_stat, stat_begin, stat_end, (int)stat_mode, moment_base,
                           skip_if_nonzero);
    else
        hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad,
                           exp_avg, exp_avg_sq, range_begin, range_end, G, 1.0f - beta1, beta2, 1.0f - beta2,
                           (float)(1.0 / sqrt(bc2)), eps, grad_stat, stat_begin, stat_end, (int)stat_mode, moment_base,
                           skip_if_nonzero);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                            int32_t n_groups, const int64_t *group_end, const float *lr, float beta1, float beta2,
                            float eps, int64_t step, float *grad_stat, int64_t stat_begin, int64_t stat_end,
                            int32_t stat_mode, gs_stream_t stream) {
    return adam_step_impl(param, grad, exp_avg, exp_avg_sq, n, 0, n, n_groups, group_end, lr, beta1, beta2, eps, step,
                          grad_stat, stat_begin, stat_end, stat_mode, stream);
}

extern "C" int gs_adam_step_range(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                                  int64_t range_begin, int64_t range_end, int32_t n_groups, const int64_t *group_end,
                                  const float *lr, float beta1, float beta2, float eps, int64_t step,
                                  float *grad_stat, int64_t stat_begin, int64_t stat_end, int32_t stat_mode,
                                  gs_stream_t stream) {
    return adam_step_impl(param, grad, exp_avg, exp_avg_sq, n, range_begin, range_end, n_groups, group_end, lr, beta1,
                          beta2, eps, step, grad_stat, stat_begin, stat_end, stat_mode, stream);
}

// View-parallel training with a SHARDED optimizer (gs_dp.py, exchange = "reduce_scatter"): this rank owns elements
// [range_begin, range_end) of the flat index space and keeps only THEIR moments -- exp_avg[0] / exp_avg_sq[0] belong to
// element moment_base (a multiple of 4, <= range_begin) -- so the optimizer state and its traffic shrink by the world
// size.  `skip_if_nonzero` (may be NULL): device address of a counter; when it is non-zero the launch does nothing (a
// frame that overflowed its workspace was rendered empty: its zero gradient must not move parameters by momentum).
extern "C" int gs_adam_step_sharded(float *param, const float *grad, float *exp_avg_shard, float *exp_avg_sq_shard,
                                    int64_t n, int64_t range_begin, int64_t range_end, int64_t moment_base,
                                    int32_t n_groups, const int64_t *group_end, const float *lr, float beta1,
                                    float beta2, float eps, int64_t step, float *grad_stat, int64_t stat_begin,
                                    int64_t stat_end, int32_t stat_mode, const void *skip_if_nonzero,
                                    gs_stream_t stream) {
    return adam_step_impl(param, grad, exp_avg_shard, exp_avg_sq_shard, n, range_begin, range_end, n_groups, group_end,
                          lr, beta1, beta2, eps, step, grad_stat, stat_begin, stat_end, stat_mode, stream, moment_base,
                          (const unsigned long long *)skip_if_nonzero);
}

// Up to 8 element ranges [range_begin[r], range_end[r]) of the same flat arrays in one launch (host arrays).  The moments
// of range r start at exp_avg + moment_offset[r] (a multiple of 4): element i of range r keeps its moments at index
// moment_offset[r] + (i - range_begin[r]) -- identity (moment_offset = range_begin) for a replicated optimizer, densely
// packed shards for a sharded one.  range_begin[r] must be a multiple of 4.  grad_scale: the gradient is used as
// grad * grad_scale (1 / world after a SUM all-reduce; 1: every element's update is what gs_adam_step computes for it).
extern "C" int gs_adam_step_multi(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                                  int32_t n_ranges, const int64_t *range_begin, const int64_t *range_end,
                                  const int64_t *moment_offset, int32_t n_groups, const int64_t *group_end,
                                  const float *lr, float beta1, float beta2, float eps, int64_t step, float *grad_stat,
                                  int64_t stat_begin, int64_t stat_end, int32_t stat_mode, const void *skip_if_nonzero,
                                  float grad_scale, gs_stream_t stream) {
    GS_CHECK_ARG(n >= 0, "n < 0");
    GS_CHECK_ARG(grad_scale > 0.f && grad_scale < 3.0e38f, "grad_scale must be positive and finite");
    GS_CHECK_ARG(n_ranges >= 1 && n_ranges <= GS_ADAM_MAX_RANGES, "n_ranges must be in [1, 8]");
    GS_CHECK_ARG(range_begin && range_end && moment_offset, "null range table");
    GS_CHECK_ARG(n_groups >= 1 && n_groups <= GS_ADAM_MAX_GROUPS, "n_groups must be in [1, 8]");
    GS_CHECK_ARG(group_end && lr, "null group table");
    GS_CHECK_ARG(step >= 1, "step counts from 1 (torch.optim.Adam increments before the update)");
    GS_CHECK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "bad hyper-parameters");
    GS_CHECK_ARG(stat_mode >= 0 && stat_mode <= 2, "stat_mode must be 0 (off), 1 (max) or 2 (sum)");
    GS_CHECK_ARG(param && grad && exp_avg && exp_avg_sq, "null pointer");
    GS_CHECK_ARG((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                 "buffers must be 16-byte aligned");
    GS_CHECK_ARG(!stat_mode || (grad_stat && stat_begin >= 0 && stat_begin <= stat_end && stat_end <= n),
                 "bad statistic range");
    AdamRanges R;
    int64_t longest = 0;
    for (int r = 0; r < GS_ADAM_MAX_RANGES; ++r) {
        const int rr = r < n_ranges ? r : 0;
        const int64_t lo = range_begin[rr], hi = r < n_ranges ? range_end[rr] : range_begin[rr];
        GS_CHECK_ARG(lo >= 0 && lo <= hi && hi <= n && (lo & 3) == 0, "bad element range (begin must be a multiple of 4)");
        GS_CHECK_ARG(moment_offset[rr] >= 0 && (moment_offset[rr] & 3) == 0, "moment_offset must be a multiple of 4");
        R.lo[r] = lo;
        R.hi[r] = hi;
        R.moment_base[r] = lo - moment_offset[rr];  // the kernel indexes the moments with (element - moment_base)
        if (hi - lo > longest) longest = hi - lo;
    }
    if (longest == 0) return 0;
    AdamGroups G;
    int64_t prev = 0;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    for (int k = 0; k < GS_ADAM_MAX_GROUPS; ++k) {
        const int kk = k < n_groups ? k : n_groups - 1;
        GS_CHECK_ARG(group_end[kk] >= prev && group_end[kk] <= n, "group_end must be ascending and <= n");
        prev = group_end[kk];
        G.end[k] = group_end[kk];
        G.step_size[k] = (float)((double)lr[kk] / bc1);
    }
    GS_CHECK_ARG(group_end[n_groups - 1] == n, "the groups must cover [0, n)");
    G.n = n_groups;
    int64_t blocks = gs_div_up(gs_div_up(longest, 4), 256);
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    const dim3 grid((unsigned)blocks, (unsigned)n_ranges);
    // streaming accesses once the four arrays as a whole exceed the Infinity Cache (the ranges of one step add up to them)
    if ((unsigned long long)n * 16ull > GS_ADAM_NT_BYTES)
        hipLaunchKernelGGL(adam_multi_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                           exp_avg_sq, R, G, 1.0f - beta1, beta2, 1.0f - beta2, (float)(1.0 / sqrt(bc2)), eps, grad_stat,
                           stat_begin, stat_end, (int)stat_mode, (const unsigned long long *)skip_if_nonzero, grad_scale);
    else
        hipLaunchKernelGGL(adam_multi_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                           exp_avg_sq, R, G, 1.0f - beta1, beta2, 1.0f - beta2, (float)(1.0 / sqrt(bc2)), eps, grad_stat,
                           stat_begin, stat_end, (int)stat_mode, (const unsigned long long *)skip_if_nonzero, grad_scale);
    GS_CHECK_LAUNCH();
    return 0;
}
