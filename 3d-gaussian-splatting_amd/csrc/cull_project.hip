// cull_project.hip -- frustum culling + 3D->2D EWA covariance projection, forward/backward.
//
// Replaces global_culling_kernel (gaussian.cu:1182-1336), global_culling_backward_kernel
// (:1371-1576), world2camera (:49-99) and jacobian (:10-47) of the reference, and provides the
// fused first/last stages of the frame path (gs_frame_forward / gs_frame_backward): projection
// + activations + tile-rectangle count + per-block pair sums in ONE pass over the Gaussians.
//
// This file is compiled with -ffp-contract=off and evaluates every expression in the
// reference's source order, so that depth bits and tile rectangles (the integer inputs of the
// sort) are bit-identical to oracle/gs_oracle.c.  The stage is HBM-bound (40 B in, <=60 B out,
// ~300 flops per Gaussian); losing FMA contraction costs nothing measurable.
#include <atomic>
#include <mutex>

#include "gs_common.h"
#include "gs_frame_layout.h"
#include "strip_common.h"
#include "tile_bin_common.h"

namespace {

struct Cam {
    float rot[9];
    float tran[3];
};

__device__ __forceinline__ void world_to_camera(const float p[3], const Cam &cam, float pc[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
        pc[i] = cam.rot[i * 3 + 0] * p[0] + cam.rot[i * 3 + 1] * p[1] + cam.rot[i * 3 + 2] * p[2] +
                cam.tran[i];
}

__device__ __forceinline__ void quat_to_R(float w, float x, float y, float z, float R[9]) {
    R[0] = 1 - 2 * y * y - 2 * z * z;
    R[1] = 2 * x * y - 2 * z * w;
    R[2] = 2 * x * z + 2 * y * w;
    R[3] = 2 * x * y + 2 * z * w;
    R[4] = 1 - 2 * x * x - 2 * z * z;
    R[5] = 2 * y * z - 2 * x * w;
    R[6] = 2 * x * z - 2 * y * w;
    R[7] = 2 * y * z + 2 * x * w;
    R[8] = 1 - 2 * x * x - 2 * y * y;
}

__device__ __forceinline__ void mm3(const float A[9], const float B[9], float C[9]) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) s += A[r * 3 + k] * B[k * 3 + c];
            C[r * 3 + c] = s;
        }
}
__device__ __forceinline__ void mm3_nt(const float A[9], const float B[9], float C[9]) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) s += A[r * 3 + k] * B[c * 3 + k];
            C[r * 3 + c] = s;
        }
}

// Rows 0,1 of J*W (row 2 of the Jacobian never reaches the 2x2 covariance).  JW is 3x3 with
// row 2 left zero so the 3x3 products below have the reference's shape; the compiler drops
// the dead row.
__device__ __forceinline__ void jacobian_rows(const float pc[3], float J[9]) {
    float u0 = pc[0], u1 = pc[1], u2 = pc[2];
    J[0] = 1 / u2;
    J[1] = 0;
    J[2] = -u0 / (u2 * u2);
    J[3] = 0;
    J[4] = 1 / u2;
    J[5] = -u1 / (u2 * u2);
    J[6] = 0;
    J[7] = 0;
    J[8] = 0;
}

// Returns false when culled.  pos_i = (x/z, y/z, |p_c|), cov = (S00, S01, S10, S11).
// `project` in two halves:
// project_cull = camera transform + near plane + frustum test -> pc, pos_i[0..1]; project_cov = depth + covariance.
__device__ __forceinline__ bool project_cull(const float p[3], const Cam &cam, float near_plane, float half_w,
                                             float half_h, float pc[3], float pos_i[3]) {
    world_to_camera(p, cam, pc);
    if (pc[2] <= near_plane) return false;
    pos_i[0] = pc[0] / pc[2];
    pos_i[1] = pc[1] / pc[2];
    return !(fabsf(pos_i[0]) >= half_w || fabsf(pos_i[1]) >= half_h);
}
__device__ __forceinline__ void project_cov(const float pc[3], const float q[4], const float s[3], const Cam &cam,
                                            float pos_i[3], float cov[4]) {
    pos_i[2] = sqrtf(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]);
    float R[9], S[9] = {s[0], 0, 0, 0, s[1], 0, 0, 0, s[2]}, RS[9], RSSR[9], J[9], JW[9], JWC[9], JWCWJ[9];
    quat_to_R(q[0], q[1], q[2], q[3], R);
    mm3(R, S, RS);
    mm3_nt(RS, RS, RSSR);
    jacobian_rows(pc, J);
    mm3(J, cam.rot, JW);
    mm3(JW, RSSR, JWC);
    mm3_nt(JWC, JW, JWCWJ);
    cov[0] = JWCWJ[0];
    cov[1] = JWCWJ[1];
    cov[2] = JWCWJ[3];
    cov[3] = JWCWJ[4];
}
__device__ __forceinline__ bool project(const float p[3], const float q[4], const float s[3],
                                        const Cam &cam, float near_plane, float half_w, float half_h,
                                        float pos_i[3], float cov[4]) {
    float pc[3];
    if (!project_cull(p, cam, near_plane, half_w, half_h, pc, pos_i)) return false;
    project_cov(pc, q, s, cam, pos_i, cov);
    return true;
}

// Backward of `project` w.r.t. (p, q_hat, s_hat) given dL/dpos_i and dL/dcov
// (gaussian.cu:1393-1575; the dependence of J on p is dropped exactly as there).
// Unlike the forward (depth bits and tile rectangles must equal the oracle's bit for bit: -ffp-contract=off for the
// file), gradients are compared within a tolerance: everything between the two pragmas contracts to FMAs, and the
// backward uses v_rcp_f32 / v_rsq_f32 (1 ulp) for its reciprocals -- seven correctly rounded divisions, a square root and
// ~150 separate multiplies and adds were a third of the projection backward's instructions (round 4: the kernel turned
// out to be VALU-bound, not HBM-bound, once it stopped fetching rows nobody wrote).
#pragma clang fp contract(fast)
__device__ __forceinline__ void project_backward(const float p[3], const float q[4], const float s[3],
                                                 const Cam &cam, const float gi[3], const float g2[4],
                                                 float gp[3], float gq[4], float gs[3]) {
    float pc[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) pc[r] = cam.rot[r * 3 + 0] * p[0] + cam.rot[r * 3 + 1] * p[1] + cam.rot[r * 3 + 2] * p[2] + cam.tran[r];
    const float ir_ = gs_rsq(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]), iz = gs_rcp(pc[2]);
    float gc[3];
    gc[0] = gi[0] * iz + gi[2] * pc[0] * ir_;
    gc[1] = gi[1] * iz + gi[2] * pc[1] * ir_;
    gc[2] = -(gi[0] * pc[0] + gi[1] * pc[1]) * (iz * iz) + gi[2] * pc[2] * ir_;
#pragma unroll
    for (int ir = 0; ir < 3; ++ir) {
        float a = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) a += cam.rot[k * 3 + ir] * gc[k];
        gp[ir] = a;
    }
    // rows 0, 1 of J W (J = [[1/z, 0, -x/z^2], [0, 1/z, -y/z^2]])
    float JW[6];
    const float jx = -pc[0] * (iz * iz), jy = -pc[1] * (iz * iz);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        JW[c] = iz * cam.rot[c] + jx * cam.rot[6 + c];
        JW[3 + c] = iz * cam.rot[3 + c] + jy * cam.rot[6 + c];
    }
    float g3[9];
#pragma unroll
    for (int ir = 0; ir < 3; ++ir)
#pragma unroll
        for (int ic = 0; ic < 3; ++ic) {
            float a = 0;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int ij = 0; ij < 2; ++ij) a += g2[ii * 2 + ij] * JW[ii * 3 + ir] * JW[ij * 3 + ic];
            g3[ir * 3 + ic] = a;
        }
    float R[9], RS[9], gRS[9];
    quat_to_R(q[0], q[1], q[2], q[3], R);
#pragma unroll
    for (int i = 0; i < 9; ++i) RS[i] = R[i] * s[i % 3];
#pragma unroll
    for (int ir = 0; ir < 3; ++ir)
#pragma unroll
        for (int ic = 0; ic < 3; ++ic) {
            float a = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) a += (g3[k * 3 + ir] + g3[ir * 3 + k]) * RS[k * 3 + ic];
            gRS[ir * 3 + ic] = a;
        }
#pragma unroll
    for (int i = 0; i < 3; ++i)
        gs[i] = gRS[0 * 3 + i] * R[0 * 3 + i] + gRS[1 * 3 + i] * R[1 * 3 + i] + gRS[2 * 3 + i] * R[2 * 3 + i];
    const float sx = s[0], sy = s[1], sz = s[2];
    const float qr = q[0], qi = q[1], qj = q[2], qk = q[3];
    const float c_qr[9] = {0, -2 * sy * qk, 2 * sz * qj, 2 * sx * qk, 0, -2 * sz * qi, -2 * sx * qj, 2 * sy * qi, 0};
    const float c_qi[9] = {0, 2 * sy * qj, 2 * sz * qk, 2 * sx * qj, -4 * sy * qi, -2 * sz * qr,
                           2 * sx * qk, 2 * sy * qr, -4 * sz * qi};
    const float c_qj[9] = {-4 * sx * qj, 2 * sy * qi, 2 * sz * qr, 2 * sx * qi, 0, 2 * sz * qk,
                           -2 * sx * qr, 2 * sy * qk, -4 * sz * qj};
    const float c_qk[9] = {-4 * sx * qk, -2 * sy * qr, 2 * sz * qi, 2 * sx * qr, -4 * sy * qk, 2 * sz * qj,
                           2 * sx * qi, 2 * sy * qj, 0};
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    {
        // NOT contracted: for an isotropic Gaussian the rotation has no effect and these sums are +p - p with both
        // products rounded alike, i.e. exactly zero (COLMAP-initialised scenes are all isotropic); an fma would leave the
        // rounding error of one product as a "gradient" (tests/test_gpu_splatter.py caught 5e-8 against an exact 0)
#pragma clang fp contract(off)
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            a0 += c_qr[i] * gRS[i];
            a1 += c_qi[i] * gRS[i];
            a2 += c_qj[i] * gRS[i];
            a3 += c_qk[i] * gRS[i];
        }
    }
    gq[0] = a0;
    gq[1] = a1;
    gq[2] = a2;
    gq[3] = a3;
}
#pragma clang fp contract(off)

__device__ __forceinline__ void load3(const float *base, int64_t i, float v[3]) {
    v[0] = base[i * 3 + 0];
    v[1] = base[i * 3 + 1];
    v[2] = base[i * 3 + 2];
}

// ---------------------------------------------------------------- reference-API kernels
__global__ void __launch_bounds__(256) global_culling_kernel(
    const float *__restrict__ pos, const float4 *__restrict__ quat, const float *__restrict__ scale,
    const float *__restrict__ rot, const float *__restrict__ tran, int64_t n, float near_plane,
    float half_w, float half_h, float *__restrict__ res_pos, float4 *__restrict__ res_cov,
    int64_t *__restrict__ mask) {
    Cam cam;
#pragma unroll
    for (int i = 0; i < 9; ++i) cam.rot[i] = rot[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) cam.tran[i] = tran[i];
    for (int64_t pid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pid < n;
         pid += (int64_t)gridDim.x * blockDim.x) {
        float p[3], s[3], pi[3], cv[4];
        load3(pos, pid, p);
        load3(scale, pid, s);
        float4 q4 = quat[pid];
        float q[4] = {q4.x, q4.y, q4.z, q4.w};
        if (!project(p, q, s, cam, near_plane, half_w, half_h, pi, cv)) continue;
        mask[pid] = 1;
        res_pos[pid * 3 + 0] = pi[0];
        res_pos[pid * 3 + 1] = pi[1];
        res_pos[pid * 3 + 2] = pi[2];
        res_cov[pid] = make_float4(cv[0], cv[1], cv[2], cv[3]);
    }
}

__global__ void __launch_bounds__(256) global_culling_backward_kernel(
    const float *__restrict__ pos, const float4 *__restrict__ quat, const float *__restrict__ scale,
    const float *__restrict__ rot, const float *__restrict__ tran, int64_t n,
    const float *__restrict__ gradout_pos, const float4 *__restrict__ gradout_cov,
    const int64_t *__restrict__ mask, float *__restrict__ gin_pos, float4 *__restrict__ gin_quat,
    float *__restrict__ gin_scale) {
    Cam cam;
#pragma unroll
    for (int i = 0; i < 9; ++i) cam.rot[i] = rot[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) cam.tran[i] = tran[i];
    for (int64_t pid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pid < n;
         pid += (int64_t)gridDim.x * blockDim.x) {
        if (mask[pid] == 0) continue;
        float p[3], s[3], gi[3], gp[3], gq[4], gs[3];
        load3(pos, pid, p);
        load3(scale, pid, s);
        load3(gradout_pos, pid, gi);
        float4 q4 = quat[pid], c4 = gradout_cov[pid];
        float q[4] = {q4.x, q4.y, q4.z, q4.w}, g2[4] = {c4.x, c4.y, c4.z, c4.w};
        project_backward(p, q, s, cam, gi, g2, gp, gq, gs);
        gin_pos[pid * 3 + 0] = gp[0];
        gin_pos[pid * 3 + 1] = gp[1];
        gin_pos[pid * 3 + 2] = gp[2];
        gin_quat[pid] = make_float4(gq[0], gq[1], gq[2], gq[3]);
        gin_scale[pid * 3 + 0] = gs[0];
        gin_scale[pid * 3 + 1] = gs[1];
        gin_scale[pid * 3 + 2] = gs[2];
    }
}

__global__ void __launch_bounds__(256) world2camera_kernel(const float *__restrict__ pos,
                                                          const float *__restrict__ rot,
                                                          const float *__restrict__ tran,
                                                          float *__restrict__ res, int64_t B) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        float p[3];
        load3(pos, i, p);
        res[i * 3 + 0] = p[0] * rot[0] + p[1] * rot[1] + p[2] * rot[2] + tran[0];
        res[i * 3 + 1] = p[0] * rot[3] + p[1] * rot[4] + p[2] * rot[5] + tran[1];
        res[i * 3 + 2] = p[0] * rot[6] + p[1] * rot[7] + p[2] * rot[8] + tran[2];
    }
}

__global__ void __launch_bounds__(256) world2camera_backward_kernel(const float *__restrict__ grad_out,
                                                                   const float *__restrict__ rot,
                                                                   float *__restrict__ grad_inp, int64_t B) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        float g[3];
        load3(grad_out, i, g);
        grad_inp[i * 3 + 0] = g[0] * rot[0] + g[1] * rot[3] + g[2] * rot[6];
        grad_inp[i * 3 + 1] = g[0] * rot[1] + g[1] * rot[4] + g[2] * rot[7];
        grad_inp[i * 3 + 2] = g[0] * rot[2] + g[1] * rot[5] + g[2] * rot[8];
    }
}

__global__ void __launch_bounds__(256) jacobian_kernel(const float *__restrict__ pos_cam,
                                                      float *__restrict__ jac, int64_t B) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        float u[3];
        load3(pos_cam, i, u);
        float *J = jac + i * 9;
        J[0] = 1 / u[2];
        J[1] = 0;
        J[2] = -u[0] / (u[2] * u[2]);
        J[3] = 0;
        J[4] = 1 / u[2];
        J[5] = -u[1] / (u[2] * u[2]);
        float rs = 1.0f / sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        J[6] = rs * u[0];
        J[7] = rs * u[1];
        J[8] = rs * u[2];
    }
}

// ---------------------------------------------------------------- fused frame stage S1
// One thread per Gaussian: activations -> project -> tile rectangle -> per-Gaussian record.
// Per-block sum of tiles_touched goes to block_sums[blockIdx.x] (input of the scan stage).
struct ProjectParams {
    Cam cam;
    float near_plane, half_w, half_h;
    float tlog;  // -2*logf(thresh), computed on the host
    float tlx, tly, leftmost, topmost;
    uint32_t ntx, nty;
    int32_t scale_act;
    int32_t color_dim;
    int32_t cull_method;        // 0: "dist" (tile centres), 1: "prob" (tile edges), 2: "prob2" (index arithmetic)
    float dist_thresh, dist_radius;  // "dist": squared distance threshold (splatter.py:577) and its square root
    float half_padw, half_padh;  // padded size / 2, in pixels (exact in fp32)
    float fx, fy;
    // occlusion test of frame_project_cull_count_kernel (conservative, never compared bit for bit): 1 / tlx, 1 / tly and
    // 1.02 x sqrt(tlog) x the largest singular value of the camera rotation (1 for a rotation; the caller's matrix is not trusted)
    float inv_tlx, inv_tly, occ_k;
};

// sigmoid on the transcendental unit (v_exp_f32 + v_rcp_f32, ~2 ulp): the opacity / colour activations feed the
// compositing only (image tolerance 5e-5), not the integer side of the pipeline; expf + an IEEE division cost
// ~25 instructions each, four times per Gaussian, in a kernel that is VALU-issue bound
__device__ __forceinline__ float sigmoid_f(float x) { return gs_rcp(1.0f + gs_exp2(-GS_LOG2E * x)); }

// "prob" (calc_tile_info_kernel2, gaussian.cu:138-195): tile i of an axis is listed unless
// `edge(i+1) < lo || hi < edge(i)`, with the tile edges of Tiles.create_tiles (splatter.py:275-293):
// edge(i) = (-pad/2 + 16 i) / focal, evaluated in fp32 exactly as torch does (an exact integer-valued float divided by
// the focal length; right(i) = left(i) + 16 before the division, so right(i) == left(i+1) bit for bit).  Both
// conditions are monotone in i, so the listed tiles are the range [first i with !(edge(i+1) < lo), last i with
// !(hi < edge(i))]: two binary searches with the reference's own comparisons (NaN bounds list every tile, as there).
__device__ __forceinline__ void edge_range(float lo, float hi, float half_pad, float focal, uint32_t n,
                                           uint32_t &i0, uint32_t &i1) {
    auto edge = [&](uint32_t i) { return (16.0f * (float)i - half_pad) / focal; };
    uint32_t a = 0, b = n;  // first i in [0, n] with !(edge(i+1) < lo)
    while (a < b) {
        const uint32_t m = (a + b) >> 1;
        if (edge(m + 1) < lo) a = m + 1; else b = m;
    }
    i0 = a;
    a = 0, b = n;  // first i in [0, n] with (hi < edge(i)): one past the last listed tile
    while (a < b) {
        const uint32_t m = (a + b) >> 1;
        if (hi < edge(m)) b = m; else a = m + 1;
    }
    i1 = a;
    if (i0 > i1) i0 = i1;
}

// Tile rectangle of calc_tile_info_kernel3 (gaussian.cu:226-242); count = 0 if det <= 0.
__device__ __forceinline__ uint32_t tile_rect(float cx, float cy, const float cv[4], const ProjectParams &P,
                                              uint32_t &y0, uint32_t &y1, uint32_t &x0, uint32_t &x1) {
    float det = (cv[0] * cv[3] - cv[1] * cv[2]);
    y0 = y1 = x0 = x1 = 0;
    if (det <= 0) return 0;
    float ai = (float)(cv[3] / (det + 1e-14));
    float di = (float)(cv[0] / (det + 1e-14));
    float shift_x = sqrtf(di * P.tlog * det);
    float shift_y = sqrtf(ai * P.tlog * det);
    float bbx_right = cx + shift_x, bbx_left = cx - shift_x;
    float bbx_top = cy - shift_y, bbx_bottom = cy + shift_y;
    if (P.cull_method == 1) {
        edge_range(bbx_left, bbx_right, P.half_padw, P.fx, P.ntx, x0, x1);
        edge_range(bbx_top, bbx_bottom, P.half_padh, P.fy, P.nty, y0, y1);
        return (y1 - y0) * (x1 - x0);
    }
    y0 = gs_f2u_sat(fmaxf((bbx_top - P.topmost) / P.tly, 0));
    y1 = gs_f2u_sat((bbx_bottom - P.topmost) / P.tly + 1);
    x0 = gs_f2u_sat(fmaxf((bbx_left - P.leftmost) / P.tlx, 0));
    x1 = gs_f2u_sat((bbx_right - P.leftmost) / P.tlx + 1);
    if (y1 > P.nty) y1 = P.nty;
    if (x1 > P.ntx) x1 = P.ntx;
    if (y0 > y1) y0 = y1;
    if (x0 > x1) x0 = x1;
    return (y1 - y0) * (x1 - x0);
}

// "dist" (calc_tile_info_kernel, gaussian.cu:101-136): a Gaussian is listed in every tile whose centre is closer than
// sqrt(thresh) to its own centre.  The listed tiles are decided per tile by gs_dist_listed (gs_common.h, the
// reference's own fp32 comparison); this is only the bounding square of the disc, with one tile of slack on every side
// (the index arithmetic below is not the reference's edge arithmetic; it is off by rounding only), over which the
// binning walks and the gradient rows are laid out.  NaN / infinite centres list nothing, as there.
__device__ __forceinline__ uint32_t dist_rect(float cx, float cy, const ProjectParams &P, uint32_t &y0, uint32_t &y1,
                                              uint32_t &x0, uint32_t &x1) {
    y0 = y1 = x0 = x1 = 0;
    if (!(fabsf(cx) < 3.0e38f) || !(fabsf(cy) < 3.0e38f) || !(P.dist_radius >= 0.f)) return 0;
    // tile centre i sits at (16 i + 8 - pad/2) / focal: i = ((c -+ r) focal + pad/2 - 8) / 16
    const float lx = ((cx - P.dist_radius) * P.fx + P.half_padw - 8.0f) * 0.0625f;
    const float hx = ((cx + P.dist_radius) * P.fx + P.half_padw - 8.0f) * 0.0625f;
    const float ly = ((cy - P.dist_radius) * P.fy + P.half_padh - 8.0f) * 0.0625f;
    const float hy = ((cy + P.dist_radius) * P.fy + P.half_padh - 8.0f) * 0.0625f;
    if (hx < -1.0f || hy < -1.0f || lx > (float)P.ntx || ly > (float)P.nty) return 0;
    // centres i with lx < i < hx can be listed: [ceil(lx), floor(hx)] -> one more on either side
    x0 = gs_f2u_sat(floorf(lx));
    y0 = gs_f2u_sat(floorf(ly));
    x1 = gs_f2u_sat(ceilf(hx) + 1.0f);
    y1 = gs_f2u_sat(ceilf(hy) + 1.0f);
    if (x1 > P.ntx) x1 = P.ntx;
    if (y1 > P.nty) y1 = P.nty;
    if (x0 > x1) x0 = x1;
    if (y0 > y1) y0 = y1;
    return (y1 - y0) * (x1 - x0);
}

__device__ __forceinline__ void activate(const float qraw[4], const float sraw[3], int scale_act,
                                         float q[4], float s[3]) {
    float nr = sqrtf(qraw[0] * qraw[0] + qraw[1] * qraw[1] + qraw[2] * qraw[2] + qraw[3] * qraw[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = qraw[k] / nr;
#pragma unroll
    for (int k = 0; k < 3; ++k) s[k] = scale_act == 0 ? fabsf(sraw[k]) + 1e-4f : expf(sraw[k]);
}

// Raw parameters of one Gaussian (what S1 reads: 56 bytes with rgb logits, 44 with SH)
struct RawGaussian {
    float p[3], sraw[3], qraw[4], opa, rgb[3];
};
__device__ __forceinline__ RawGaussian load_raw(const float *__restrict__ pos, const float4 *__restrict__ quat,
                                                const float *__restrict__ scale, const float *__restrict__ opa,
                                                const float *__restrict__ rgb, int64_t pid, int color_dim) {
    RawGaussian r;
    load3(pos, pid, r.p);
    load3(scale, pid, r.sraw);
    const float4 q4 = quat[pid];
    r.qraw[0] = q4.x; r.qraw[1] = q4.y; r.qraw[2] = q4.z; r.qraw[3] = q4.w;
    r.opa = opa[pid];
    r.rgb[0] = r.rgb[1] = r.rgb[2] = 0.f;
    if (color_dim == 3) load3(rgb, pid, r.rgb);
    return r;
}

// The raw parameters of the FIRST round are waited for before the loop is entered.  Without this the loop header merges
// "the prologue's loads are in flight" with the back edge's clean state, and the waitcnt pass, conservative across the
// merge, puts an `s_waitcnt vmcnt(1)` in front of the first use of the current round's quaternion -- right behind the
// five loads of the NEXT round, which are thereby waited for as well: the prefetch hid nothing.
#ifndef GS_PROJECT_PROLOGUE_WAIT
#define GS_PROJECT_PROLOGUE_WAIT 1
#endif
__device__ __forceinline__ void settle(const RawGaussian &r) {
#if GS_PROJECT_PROLOGUE_WAIT
    asm volatile("" ::"v"(r.p[0]), "v"(r.p[1]), "v"(r.p[2]), "v"(r.sraw[0]), "v"(r.sraw[1]), "v"(r.sraw[2]), "v"(r.qraw[0]),
                 "v"(r.qraw[1]), "v"(r.qraw[2]), "v"(r.qraw[3]), "v"(r.opa), "v"(r.rgb[0]), "v"(r.rgb[1]), "v"(r.rgb[2]));
#endif
}

// S1 for one Gaussian: activations -> project -> tile rectangle -> 64-byte record (visible Gaussians only) + the
// 16-byte rectangle record (every Gaussian).  Returns the rectangle record; `vis` = passed the frustum test; `cxy` = the
// projected centre (the "dist" listing test of the binning needs it).
__device__ __forceinline__ uint4 project_one(const RawGaussian &in, int64_t pid, const ProjectParams &P,
                                             float4 *__restrict__ rec_geom, uint32_t *__restrict__ tiles_touched,
                                             uint4 *__restrict__ rects, uint32_t &vis, float2 &cxy) {
    float q[4], s[3], pi[3], cv[4];
    activate(in.qraw, in.sraw, P.scale_act, q, s);
    uint32_t cnt = 0;
    vis = 0;
    cxy = make_float2(0.f, 0.f);
    uint2 rc = make_uint2(0, 0);
    float depth = 0.f;
    // A culled Gaussian leaves 16 (20) bytes -- its all-zero rectangle, which is what every later stage looks at first
    // (rects[i].z, the depth bits, is 0 exactly for culled Gaussians: visible ones lie beyond the near plane) --
    // and NOT its 64-byte record: nothing reads the record of a Gaussian that is in no tile's list (21 % of the
    // Gaussians of the 2.4 M scene: 33 of this stage's 337 MB).  The record of a culled Gaussian is unspecified.
    if (project(in.p, q, s, P.cam, P.near_plane, P.half_w, P.half_h, pi, cv)) {
        vis = 1;
        uint32_t y0, y1, x0, x1;
        cnt = P.cull_method == 0 ? dist_rect(pi[0], pi[1], P, y0, y1, x0, x1)
                                 : tile_rect(pi[0], pi[1], cv, P, y0, y1, x0, x1);
        rc = make_uint2(y0 | (y1 << 16), x0 | (x1 << 16));
        depth = pi[2];
        cxy = make_float2(pi[0], pi[1]);
        float4 col = make_float4(0, 0, 0, 0);
        if (P.color_dim == 3)
            col = make_float4(sigmoid_f(in.rgb[0]), sigmoid_f(in.rgb[1]), sigmoid_f(in.rgb[2]), 0.0f);
        float cA = 0.f, cB = 0.f, cC = 0.f;
        gs_conic(cv[0], cv[1], cv[2], cv[3], cA, cB, cC);
        float4 *rec = rec_geom + pid * GS_REC_STRIDE;  // one 64-byte record per Gaussian
        rec[0] = make_float4(pi[0], pi[1], pi[2], sigmoid_f(in.opa));
        rec[1] = make_float4(cv[0], cv[1], cv[2], cv[3]);
        rec[2] = col;
        rec[3] = make_float4(cA, cB, cC, 0.f);
    }
    if (tiles_touched) tiles_touched[pid] = cnt;  // read by the radix paths (sort_modes 0 / 1) only
    const uint4 out = make_uint4(rc.x, rc.y, __float_as_uint(depth), cnt);
    rects[pid] = out;
    return out;
}

__global__ void __launch_bounds__(256) frame_project_kernel(
    const float *__restrict__ pos, const float4 *__restrict__ quat, const float *__restrict__ scale,
    const float *__restrict__ opa, const float *__restrict__ rgb, int64_t n, ProjectParams P,
    float4 *__restrict__ rec_geom,
    uint32_t *__restrict__ tiles_touched, uint4 *__restrict__ rects, uint32_t *__restrict__ block_sums,
    uint32_t *__restrict__ block_vis) {
    const int64_t pid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t cnt = 0, vis = 0;
    if (pid < n) {
        float2 cxy;
        cnt = project_one(load_raw(pos, quat, scale, opa, rgb, pid, P.color_dim), pid, P, rec_geom, tiles_touched, rects,
                          vis, cxy).w;
    }
    // block sums of cnt and of the visible flag -> two plain stores per block (a same-address
    // atomic per block would serialise at ~12 ns each: 112 us for 2.4 M Gaussians)
    __shared__ uint32_t s_cnt[4], s_vis[4];
    uint32_t wsum = gs_wave_sum_u32(cnt), wvis = gs_wave_sum_u32(vis);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_cnt[wave] = wsum;
        s_vis[wave] = wvis;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        block_sums[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        block_vis[blockIdx.x] = s_vis[0] + s_vis[1] + s_vis[2] + s_vis[3];
    }
}

// ---------------------------------------------------------------- S1 + L1a fused (strip variant of sort_mode 2)
// frame_project_kernel writes a 16-byte rectangle per Gaussian and strip_count_kernel (strip_bin.hip) reads all of them
// back to histogram the strip entries: 38 MB and a 16-us launch at 2.4 M Gaussians for information that was in
// registers.  Here ONE workgroup per slice of the Gaussian array (the slices of the level-1 kernels: <= 256, dealt to
// the XCDs in contiguous runs) projects its Gaussians, 1024 at a time, and counts their strip entries on the spot -- one
// 64-bit LDS atomic per entry, as there.  The raw parameters of the NEXT round are requested before the current one
// is projected (16 waves per CU hide the rest).  Same outputs as the two kernels: records, rectangles, the [S][NS] table
// row, the slice's pair / visible counts; the one extra workgroup of the launch writes the tile dispatch order.
// Measured and dropped (round 3, same-box A/B): a COMPACTING variant of this kernel -- every wave tests 64 Gaussians
// against the frustum (~40 instructions), queues the survivors' raw parameters in a 128-entry LDS ring and runs the
// long half (~750 instructions) on full waves of survivors only, so that the 21 % culled Gaussians of the 2.4 M scene
// stop paying for it lane-masked.  Bit-identical outputs, 21 % fewer long-half wave passes -- and 89 - 92 us against
// 81 - 82 us for the kernel below (parameters loaded by index in the long half instead of queued: 93 us).
template <bool DIST>
__global__ void __launch_bounds__(STRIP_THREADS) frame_project_count_kernel(
    const float *__restrict__ pos, const float4 *__restrict__ quat, const float *__restrict__ scale,
    const float *__restrict__ opa, const float *__restrict__ rgb, int64_t n, ProjectParams P,
    float4 *__restrict__ rec_geom, uint32_t *__restrict__ tiles_touched, uint4 *__restrict__ rects, GsDistCull D,
    uint32_t per_slice, gs_strip_geom SG, uint32_t S, uint32_t slice0, unsigned long long *__restrict__ table,
    uint32_t *__restrict__ slice_pairs, uint32_t *__restrict__ slice_vis, const uint32_t *__restrict__ tile_cost,
    uint32_t n_tiles, uint32_t *__restrict__ tile_order, const unsigned long long *__restrict__ gate) {
    extern __shared__ unsigned long long s_hist[];  // [NS] entries << 32 | pairs of this slice
    __shared__ uint32_t s_acc[2];
    // `gate`: the second, unculled pass of a GS_FRAME_OCCLUSION_CULL frame -- nothing to do unless a tile ran past its cut
    if (gate && *gate == 0) return;
    if (blockIdx.x >= S) {  // the one extra workgroup of the launch (uniform)
        tile_order_workgroup(tile_cost, n_tiles, tile_order, SG.ntx, SG.nty);
        return;
    }
    // this launch covers the slices [slice0, slice0 + S): all of them, or one range of a frame whose project stage is
    // issued range by range (gs_frame_forward_project: the view-parallel trainer projects a range of Gaussians as soon
    // as their parameters have been updated)
    const uint32_t slice = slice0 + strip_slice_of_block(blockIdx.x, S);
    const int64_t g0 = (int64_t)slice * per_slice;
    auto in_range = [&](uint32_t i) { return i < per_slice && g0 + i < n; };
    RawGaussian cur = {}, nxt = {};
    if (in_range(threadIdx.x)) cur = load_raw(pos, quat, scale, opa, rgb, g0 + threadIdx.x, P.color_dim);
    for (uint32_t t = threadIdx.x; t < SG.NS; t += STRIP_THREADS) s_hist[t] = 0;
    if (threadIdx.x < 2) s_acc[threadIdx.x] = 0;
    __syncthreads();
    settle(cur);
    uint32_t acc_cnt = 0, acc_vis = 0;
    for (uint32_t base = 0; base < per_slice; base += STRIP_THREADS) {  // uniform trip count
        const uint32_t i = base + threadIdx.x;
        if (in_range(i + STRIP_THREADS)) nxt = load_raw(pos, quat, scale, opa, rgb, g0 + i + STRIP_THREADS, P.color_dim);
        uint4 rc = make_uint4(0, 0, 0, 0);
        uint32_t vis = 0;
        float2 cxy = make_float2(0.f, 0.f);
        if (in_range(i)) rc = project_one(cur, g0 + i, P, rec_geom, tiles_touched, rects, vis, cxy);
        acc_cnt += rc.w;
        acc_vis += vis;
        walk_strips<DIST>(rc, g0 + i, SG, cxy, D,
                          [&](uint32_t strip, uint32_t, uint32_t, uint32_t np) { atomicAdd(&s_hist[strip], (1ull << 32) | np); });
        cur = nxt;
    }
    // rectangle areas (= gradient-row slots; == pairs unless DIST) and visible Gaussians of this slice
    acc_cnt = gs_wave_sum_u32(acc_cnt);
    acc_vis = gs_wave_sum_u32(acc_vis);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&s_acc[0], acc_cnt);
        atomicAdd(&s_acc[1], acc_vis);
    }
    __syncthreads();
    unsigned long long *row = table + (size_t)slice * SG.NS;
    for (uint32_t t = threadIdx.x; t < SG.NS; t += STRIP_THREADS) row[t] = s_hist[t];
    if (threadIdx.x == 0) {
        slice_pairs[slice] = s_acc[0];
        slice_vis[slice] = s_acc[1];
    }
}

// ---------------------------------------------------------------- S1 + L1a of an occlusion-culled frame (first pass)
// GS_FRAME_OCCLUSION_CULL (gs_frame_layout.h): the previous frame of this workspace left, per tile, the depth behind which
// nothing was composited (`cut`).  74 % of the pairs of the opaque 2.4 M-Gaussian scene lie behind their tile's cut, and
// more than half of the visible Gaussians have NO pair in front of one -- yet every one of them paid the ~750 instructions
// of the projection.  Lane-masking them out gains nothing (a wave skips only what all of its 64 lanes skip, and Gaussians
// arrive in no spatial order: measured, 92.7 against 79 us), so this kernel COMPACTS:
//   phase A, every Gaussian of the slice: position and scale only (24 of 56 bytes), the exact frustum test, and a
//     conservative occlusion test (occluded_everywhere) -- ~120 instructions; the survivors' indices are queued in LDS;
//   phase B, full waves of survivors: the unchanged project_one + the trimmed strip count of the culled frame.
// A Gaussian that fails the occlusion test would have lost every pair to the trimming of walk_strips<.., true>: the
// emitted lists are exactly those of the lane-masked kernel, and the frame's second pass (a tile ran past its cut) starts
// from a gated re-run of frame_project_count_kernel, which rewrites records, rectangles and the strip table untrimmed.
// The cut pyramid: level 0 = the per-tile cuts (rows padded to whole strips, what walk_strips reads), level l = the LARGEST
// cut of each block of 2^l x 2^l tiles (GS_NO_CUT -- a tile that did not saturate -- is the largest value there is).
#define GS_OCC_QCAP 16384u   // Gaussians per chunk of a slice: their in-chunk indices are queued as 16-bit words
#define GS_OCC_LEVELS 4      // pyramid levels 0..3: rectangles of up to 16 x 16 tiles are tested with <= 3 x 3 look-ups
struct OccPyramid {
    const uint32_t *t0;           // level 0
    uint32_t o1, o2, o3;          // words from level l - 1's table to level l's
    uint32_t st0, d1, d2, d3;     // row stride of level 0; stride of level l minus stride of level l - 1 (mod 2^32)
};
// Is the Gaussian behind the cut of every tile its rectangle can reach?  Conservative by construction: the rectangle's
// half extents are sqrt(tlog S00) and sqrt(tlog S11) (tile_rect; its +1e-14 only shrinks them), S00 = r0 (R S S R^T) r0^T
// with r0 the first row of J W, so S00 <= s_max^2 |r0|^2 = s_max^2 (1 + (x/z)^2) / z^2 (W and R orthonormal: P.occ_k
// carries W's largest singular value, sqrt(tlog) and 2 % for the roundings -- fp32's are 1e-6), likewise S11; the tile
// range follows tile_rect's "prob2" arithmetic, monotone in the extents, with 1e-3 tile of slack.  Anything not finite,
// larger than 16 tiles, or another listing method: not tested (false).
__device__ __forceinline__ bool occluded_everywhere(const float pi[3], float z, float smax, uint32_t dbits,
                                                    const ProjectParams &P, const OccPyramid &Y) {
    if (P.cull_method != 2) return false;
    const float k = P.occ_k * smax * gs_rcp(z);
    const float ax = 1.0f + pi[0] * pi[0], ay = 1.0f + pi[1] * pi[1];
    const float rx = k * (ax * gs_rsq(ax)) * 1.0001f, ry = k * (ay * gs_rsq(ay)) * 1.0001f;
    const float fx0 = (pi[0] - rx - P.leftmost) * P.inv_tlx - 1e-3f, fx1 = (pi[0] + rx - P.leftmost) * P.inv_tlx + 1e-3f;
    const float fy0 = (pi[1] - ry - P.topmost) * P.inv_tly - 1e-3f, fy1 = (pi[1] + ry - P.topmost) * P.inv_tly + 1e-3f;
    if (!(fx1 - fx0 < 16.0f) || !(fy1 - fy0 < 16.0f)) return false;  // large or not finite: projected
    // beside the grid (23 % of the Gaussians inside the frustum test of the 2.4 M scene: its margin is wider than the image):
    // tile_rect's range is empty -- x1 = floor(v + 1) = 0 for v < 0, x0 >= ntx >= x1 on the other side -- no tile, skipped
    if (fx1 < 0.f || fy1 < 0.f) return true;
    uint32_t x0 = gs_f2u_sat(fx0), y0 = gs_f2u_sat(fy0), x1 = gs_f2u_sat(fx1), y1 = gs_f2u_sat(fy1);  // tiles [x0, x1] x [y0, y1]
    if (x0 >= P.ntx || y0 >= P.nty) return true;
    x1 = x1 < P.ntx ? x1 : P.ntx - 1;
    y1 = y1 < P.nty ? y1 : P.nty - 1;
    const uint32_t e = (x1 - x0) > (y1 - y0) ? (x1 - x0) : (y1 - y0);
    const uint32_t L = e <= 1 ? 0u : (e <= 3 ? 1u : (e <= 7 ? 2u : 3u));  // (x1 >> L) - (x0 >> L) <= 2 then
    // (level L's table starts off[L] words behind level 0's; selected arithmetically: a select between the struct's fields
    // became a load from a scratch copy of it)
    const uint32_t m1 = L >= 1 ? ~0u : 0u, m2 = L >= 2 ? ~0u : 0u, m3 = L >= 3 ? ~0u : 0u;
    const uint32_t *tab = Y.t0 + ((Y.o1 & m1) + (Y.o2 & m2) + (Y.o3 & m3));
    const uint32_t st = Y.st0 + ((Y.d1 & m1) + (Y.d2 & m2) + (Y.d3 & m3));
    const uint32_t cx0 = x0 >> L, cy0 = y0 >> L, w = (x1 >> L) - cx0, h = (y1 >> L) - cy0;  // w, h in 0..2
    uint32_t m = 0;
#pragma unroll
    for (uint32_t j = 0; j < 3; ++j)
#pragma unroll
        for (uint32_t i = 0; i < 3; ++i) {
            const uint32_t c = tab[(cy0 + (j < h ? j : h)) * st + cx0 + (i < w ? i : w)];
            m = c > m ? c : m;
        }
    return dbits > m;
}

__global__ void __launch_bounds__(STRIP_THREADS) frame_project_cull_count_kernel(
    const float *__restrict__ pos, const float4 *__restrict__ quat, const float *__restrict__ scale,
    const float *__restrict__ opa, const float *__restrict__ rgb, int64_t n, ProjectParams P,
    float4 *__restrict__ rec_geom, uint4 *__restrict__ rects, uint32_t per_slice, gs_strip_geom SG, uint32_t S,
    unsigned long long *__restrict__ table, uint32_t *__restrict__ slice_pairs, uint32_t *__restrict__ slice_vis,
    const uint32_t *__restrict__ tile_cost, uint32_t n_tiles, uint32_t *__restrict__ tile_order,
    const uint32_t *__restrict__ cut, uint32_t qcap, uint32_t stash_cap, uint4 *__restrict__ surv,
    uint32_t *__restrict__ slice_nsurv, uint32_t diag) {
    extern __shared__ unsigned long long s_hist[];  // [NS] entries << 32 | pairs of this slice, then the pyramid, then the queue
    __shared__ uint32_t s_acc[2], s_qn, s_ns;
    if (blockIdx.x >= S) {  // the one extra workgroup of the launch (uniform)
        if (diag & 16) return;
        tile_order_workgroup(tile_cost, n_tiles, tile_order, SG.ntx, SG.nty);
        return;
    }
    const uint32_t slice = strip_slice_of_block(blockIdx.x, S);
    const int64_t g0 = (int64_t)slice * per_slice;
    const int lane = threadIdx.x & 63;
    // ---- LDS: histogram | cut pyramid | survivor queue
    uint32_t *s_cut = reinterpret_cast<uint32_t *>(s_hist + SG.NS);
    const uint32_t st0 = SG.nsx * GS_STRIP_W;
    const uint32_t w1 = (SG.ntx + 1) / 2, h1 = (SG.nty + 1) / 2, w2 = (w1 + 1) / 2, h2 = (h1 + 1) / 2, w3 = (w2 + 1) / 2,
                   h3 = (h2 + 1) / 2;
    uint32_t *s_l1 = s_cut + st0 * SG.nty, *s_l2 = s_l1 + w1 * h1, *s_l3 = s_l2 + w2 * h2;
    uint16_t *s_q = reinterpret_cast<uint16_t *>(s_l3 + w3 * h3);
    // the first `stash_cap` survivors of a chunk keep their position and scale in LDS (six planes of stash_cap floats): phase B
    // then gathers only the 32 bytes phase A did not read -- 17 % of the array still touches 62 % of its 64-byte lines
    float *s_stash = reinterpret_cast<float *>(s_q + ((qcap + 1) & ~1u));
    const OccPyramid Y = {s_cut, st0 * SG.nty, w1 * h1, w2 * h2, st0, w1 - st0, w2 - w1, w3 - w2};
    // the first two rounds of positions and scales are requested before the set-up (their latency runs underneath it)
    auto fetch_at = [&](float (&pp)[3], float (&ss)[3], uint32_t i) {
        int64_t g = g0 + i;
        g = g < n ? g : n - 1;
        load3(pos, g, pp);
        load3(scale, g, ss);
    };
    float pa[3], sa[3], pb[3], sb[3];
    fetch_at(pa, sa, threadIdx.x);
    fetch_at(pb, sb, threadIdx.x + STRIP_THREADS);
    for (uint32_t t = threadIdx.x; t < SG.NS; t += STRIP_THREADS) s_hist[t] = 0;
    // (eight loads of the cut table in flight per thread: one at a time, each waited for, was 8 x the latency of a load)
    for (uint32_t t0 = threadIdx.x; t0 < st0 * SG.nty; t0 += 8 * STRIP_THREADS) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            const uint32_t t = t0 + k * STRIP_THREADS, iy = t / st0, ix = t - iy * st0;
            const bool in = iy < SG.nty && ix < SG.ntx;
            v[k] = cut[in ? iy * SG.ntx + ix : 0u];
            v[k] = in ? v[k] : GS_NO_CUT;
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k)
            if (t0 + k * STRIP_THREADS < st0 * SG.nty) s_cut[t0 + k * STRIP_THREADS] = v[k];
    }
    if (threadIdx.x < 2) s_acc[threadIdx.x] = 0;
    if (threadIdx.x == 2) s_ns = 0;
    auto build = [&](uint32_t *dst, uint32_t dw, uint32_t dh, const uint32_t *src, uint32_t sst, uint32_t sw, uint32_t sh) {
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < dw * dh; t += STRIP_THREADS) {
            const uint32_t cy = t / dw, cx = t - cy * dw, x = 2 * cx, y = 2 * cy;
            uint32_t m = src[y * sst + x];
            if (x + 1 < sw) m = max(m, src[y * sst + x + 1]);
            if (y + 1 < sh) {
                m = max(m, src[(y + 1) * sst + x]);
                if (x + 1 < sw) m = max(m, src[(y + 1) * sst + x + 1]);
            }
            dst[t] = m;
        }
    };
    build(s_l1, w1, h1, s_cut, st0, SG.ntx, SG.nty);
    build(s_l2, w2, h2, s_l1, w1, w1, h1);
    build(s_l3, w3, h3, s_l2, w2, w2, h2);
    uint32_t acc_cnt = 0, acc_vis = 0;
    for (uint32_t c0 = 0; c0 < per_slice; c0 += qcap) {  // uniform: chunks of the slice (one, unless the scene is huge)
        const uint32_t cn = (diag & 8) ? 0u : (per_slice - c0 < qcap ? per_slice - c0 : qcap);
        if (threadIdx.x == 0) s_qn = 0;
        __syncthreads();  // (also: the pyramid is complete; the previous chunk's queue has been drained)
        // ---- phase A: frustum + occlusion test of every Gaussian of the chunk; survivors are queued
        // The phase is a 58 MB stream (24 B in per Gaussian, nothing out) at one workgroup per CU: three rounds of positions
        // and scales are kept in flight.  That takes THREE NAMED register sets and a loop unrolled by three -- rotating one
        // set into the next at the end of a round (`cur = nxt`) makes the move wait for the newest load (first version:
        // s_waitcnt vmcnt(0) in every round, 30 us for the bare stream) -- and loads whose control flow is uniform (a lane
        // beyond the chunk loads the array's last Gaussian and drops it), so that the waitcnt pass counts them exactly.
        auto in_chunk = [&](uint32_t i) { return i < cn && g0 + c0 + i < n; };
        auto fetch = [&](float (&pp)[3], float (&ss)[3], uint32_t i) { fetch_at(pp, ss, c0 + i); };
        auto test = [&](const float (&pp)[3], const float (&ss)[3], uint32_t i) {
            bool surv = false;
            if (in_chunk(i)) {
                float pc[3], pi[3];
                if (project_cull(pp, P.cam, P.near_plane, P.half_w, P.half_h, pc, pi)) {
                    const float dep = sqrtf(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]);  // == project_cov's pos_i[2]
                    float s[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) s[k] = P.scale_act == 0 ? fabsf(ss[k]) + 1e-4f : gs_exp2(GS_LOG2E * ss[k]);
                    const float smax = fmaxf(s[0], fmaxf(s[1], s[2]));
                    // (a NaN scale slips through fmaxf: s0 + s1 + s2 is NaN then, and the Gaussian is projected)
                    // (`diag`, GS_OCC_DIAG: TIMING-ONLY switches, the frames they render are wrong -- 1: nobody is tested, 2: nobody is
                    // projected, 4: everybody is skipped, 8: no phase A either, 16: no tile order, 32: phase B loads contiguous
                    // Gaussians instead of the queued ones, 64: no strip count; tools/batches/gpu_r6n.sh, profiles/r06_n_*)
                    if ((diag & 4) || (!(diag & 1) && (s[0] + s[1] + s[2] < 3.0e38f) &&
                                       occluded_everywhere(pi, pc[2], smax, __float_as_uint(dep), P, Y))) {
                        // behind every cut it can reach, or beside the grid: visible, no tile.  NOTHING is written for it (nor
                        // for a Gaussian outside the frustum): the scatter of this pass reads the survivor list, the second
                        // pass re-projects everything -- rects[] of a culled frame is only fresh for the survivors
                        acc_vis += 1;
                    } else {
                        surv = true;
                    }
                }
            }
            const unsigned long long b = __ballot(surv);
            if (b) {  // (uniform per wave)
                uint32_t wbase = 0;
                if (lane == 0) wbase = atomicAdd(&s_qn, (uint32_t)__popcll(b));
                wbase = __shfl(wbase, 0, 64);
                if (surv) {
                    const uint32_t slot = wbase + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
                    s_q[slot] = (uint16_t)i;
                    if (slot < stash_cap) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            s_stash[c * stash_cap + slot] = pp[c];
                            s_stash[(3 + c) * stash_cap + slot] = ss[c];
                        }
                    }
                }
            }
        };
        float pq[3], sq[3];
        if (c0 != 0) {  // (the first chunk's were requested in front of the set-up)
            fetch(pa, sa, threadIdx.x);
            fetch(pb, sb, threadIdx.x + STRIP_THREADS);
        }
        for (uint32_t base = 0; base < cn; base += 3 * STRIP_THREADS) {  // uniform trip count, uniform exits
            const uint32_t i = base + threadIdx.x;
            fetch(pq, sq, i + 2 * STRIP_THREADS);
            test(pa, sa, i);
            if (base + STRIP_THREADS >= cn) break;
            fetch(pa, sa, i + 3 * STRIP_THREADS);
            test(pb, sb, i + STRIP_THREADS);
            if (base + 2 * STRIP_THREADS >= cn) break;
            fetch(pb, sb, i + 4 * STRIP_THREADS);
            test(pq, sq, i + 2 * STRIP_THREADS);
        }
        __syncthreads();
        // ---- phase B: the survivors, 1,024 at a time on full waves
        const uint32_t nq = (diag & 2) ? 0u : s_qn;
        RawGaussian cur = {}, nxt = {};
        uint32_t qi = 0, qn_ = 0;
        auto load_survivor = [&](uint32_t k, uint32_t q) {
            RawGaussian r;
            const int64_t pid = g0 + c0 + q;
            if (k < stash_cap) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    r.p[c] = s_stash[c * stash_cap + k];
                    r.sraw[c] = s_stash[(3 + c) * stash_cap + k];
                }
            } else {
                load3(pos, pid, r.p);
                load3(scale, pid, r.sraw);
            }
            const float4 q4 = quat[pid];
            r.qraw[0] = q4.x; r.qraw[1] = q4.y; r.qraw[2] = q4.z; r.qraw[3] = q4.w;
            r.opa = opa[pid];
            r.rgb[0] = r.rgb[1] = r.rgb[2] = 0.f;
            if (P.color_dim == 3) load3(rgb, pid, r.rgb);
            return r;
        };
        if (threadIdx.x < nq) {
            qi = (diag & 32) ? threadIdx.x : s_q[threadIdx.x];
            cur = load_survivor(threadIdx.x, qi);
        }
        settle(cur);
        for (uint32_t base = 0; base < nq; base += STRIP_THREADS) {  // uniform trip count
            const uint32_t k = base + threadIdx.x;
            if (k + STRIP_THREADS < nq) {
                qn_ = (diag & 32) ? k + STRIP_THREADS : s_q[k + STRIP_THREADS];
                nxt = load_survivor(k + STRIP_THREADS, qn_);
            }
            uint4 rc = make_uint4(0, 0, 0, 0);
            uint32_t vis = 0;
            float2 cxy = make_float2(0.f, 0.f);
            const int64_t pid = g0 + c0 + qi;
            if (k < nq) rc = project_one(cur, pid, P, rec_geom, nullptr, rects, vis, cxy);
            acc_cnt += rc.w;
            acc_vis += vis;
            {   // survivors that touch a tile: (rectangle, depth, Gaussian) appended to the slice's compact list
                const bool has = rc.w != 0;
                const unsigned long long hb = __ballot(has);
                if (hb) {  // (uniform per wave)
                    uint32_t wbase = 0;
                    if (lane == 0) wbase = atomicAdd(&s_ns, (uint32_t)__popcll(hb));
                    wbase = __shfl(wbase, 0, 64);
                    if (has)
                        surv[g0 + wbase + (uint32_t)__popcll(hb & ((1ull << lane) - 1ull))] =
                            make_uint4(rc.x, rc.y, rc.z, (uint32_t)pid);
                }
            }
            if (!(diag & 64)) walk_strips<false, true>(rc, pid, SG, cxy, GsDistCull{},
                                           [&](uint32_t strip, uint32_t, uint32_t, uint32_t np) {
                                               atomicAdd(&s_hist[strip], (1ull << 32) | np);
                                           },
                                           s_cut);
            cur = nxt;
            qi = qn_;
        }
    }
    acc_cnt = gs_wave_sum_u32(acc_cnt);
    acc_vis = gs_wave_sum_u32(acc_vis);
    if (lane == 0) {
        atomicAdd(&s_acc[0], acc_cnt);
        atomicAdd(&s_acc[1], acc_vis);
    }
    __syncthreads();
    unsigned long long *row = table + (size_t)slice * SG.NS;
    for (uint32_t t = threadIdx.x; t < SG.NS; t += STRIP_THREADS) row[t] = s_hist[t];
    if (threadIdx.x == 0) {
        slice_pairs[slice] = s_acc[0];
        slice_vis[slice] = s_acc[1];
        slice_nsurv[slice] = s_ns;
    }
}


// ---------------------------------------------------------------- S1 + B1 fused (table variant of sort_mode 2: small scenes)
// The same fusion for the table variant, which small scenes take: a frame of 10,000 Gaussians is six dependent launches
// of ~7 us each, so the launch that re-reads the rectangles to histogram them per (slice, tile) is worth removing for
// its latency alone.  One workgroup per slice (tile_bin.hip's slices: a multiple of 256 Gaussians), one LDS counter per
// tile; same outputs as frame_project_kernel + bin_count_kernel.
template <bool DIST>
__global__ void __launch_bounds__(BIN_THREADS) frame_project_bin_count_kernel(
    const float *__restrict__ pos, const float4 *__restrict__ quat, const float *__restrict__ scale,
    const float *__restrict__ opa, const float *__restrict__ rgb, int64_t n, ProjectParams P,
    float4 *__restrict__ rec_geom, uint4 *__restrict__ rects, GsDistCull D, uint32_t per_block, uint32_t T,
    uint32_t *__restrict__ table, uint32_t *__restrict__ slice_pairs, uint32_t *__restrict__ slice_vis) {
    extern __shared__ uint32_t s_tile_hist[];  // [T]
    __shared__ uint32_t s_acc[2];
    const uint32_t slice = slice_of_block(blockIdx.x, gridDim.x);
    const int64_t g0 = (int64_t)slice * per_block;
    auto in_range = [&](uint32_t i) { return i < per_block && g0 + i < n; };
    RawGaussian cur = {}, nxt = {};
    if (in_range(threadIdx.x)) cur = load_raw(pos, quat, scale, opa, rgb, g0 + threadIdx.x, P.color_dim);
    for (uint32_t t = threadIdx.x; t < T; t += BIN_THREADS) s_tile_hist[t] = 0;
    if (threadIdx.x < 2) s_acc[threadIdx.x] = 0;
    __syncthreads();
    settle(cur);
    uint32_t acc_cnt = 0, acc_vis = 0;
    for (uint32_t base = 0; base < per_block; base += BIN_THREADS) {  // uniform trip count
        const uint32_t i = base + threadIdx.x;
        if (in_range(i + BIN_THREADS)) nxt = load_raw(pos, quat, scale, opa, rgb, g0 + i + BIN_THREADS, P.color_dim);
        uint4 rc = make_uint4(0, 0, 0, 0);
        uint32_t vis = 0;
        float2 cxy = make_float2(0.f, 0.f);
        if (in_range(i)) rc = project_one(cur, g0 + i, P, rec_geom, nullptr, rects, vis, cxy);
        acc_cnt += rc.w;
        acc_vis += vis;
        walk_rect<DIST>(rc, g0 + i, P.ntx, cxy, D, [&](uint32_t tile, uint32_t, uint32_t) { atomicAdd(&s_tile_hist[tile], 1u); });
        cur = nxt;
    }
    acc_cnt = gs_wave_sum_u32(acc_cnt);
    acc_vis = gs_wave_sum_u32(acc_vis);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&s_acc[0], acc_cnt);
        atomicAdd(&s_acc[1], acc_vis);
    }
    __syncthreads();
    uint32_t *row = table + (size_t)slice * T;
    for (uint32_t t = threadIdx.x; t < T; t += BIN_THREADS) row[t] = s_tile_hist[t];
    if (threadIdx.x == 0) {
        slice_pairs[slice] = s_acc[0];
        slice_vis[slice] = s_acc[1];
    }
}

// ---------------------------------------------------------------- fused frame stage B2
// rows[pair][12|36|56] = (dx, dy, da, db, dc, dd, dopa, colour grads..) written by the raster backward
// in emission order: Gaussian g owns rows [pair_offsets[g], +tiles_touched[g]).  They are summed
// here in a fixed order (the reference's index_put_(accumulate=True), but deterministic) and pushed
// through the projection + activation backward.  Culled Gaussians get zeros.  Only rows whose flag is set were written
// by the raster backward (pairs behind a tile's early-termination point are not): the others are skipped, never read
// as numbers -- they are uninitialised memory.
// PART: 0 = everything; 1 = only the projection / activation backward (grad_pos, grad_quat, grad_scale -- the
// "geometry" bucket of the view-parallel gradient exchange); 2 = only grad_opa and grad_rgb (the "colour" bucket).
// Parts 1 and 2 read the same rows and add them in the same order as part 0: their outputs are bit-identical to it.
// They exist so that the all-reduce of the first bucket can run underneath the second kernel (gs_dp.py).
// SH rows of Gaussians that cover hundreds of tiles.  The projection backward's wave walks the rows of its 64 Gaussians one
// row per load instruction; a Gaussian with thousands of rows (a blown-up scale, a background blob: every densifying run
// has a few) kept ONE wave walking for a millisecond while the rest of the device had finished -- 1.04 of the 2.9 ms of a
// training iteration in the SH soak of round 4 (376 k Gaussians, profiles/r04_zn_*).  This kernel runs once per backward,
// right behind the raster backward: a workgroup scans its slice of the rectangles, and for every Gaussian with more than
// GS_PB_SH_BIG rows its sixteen waves add a sixteenth of the Gaussian's existing rows each (ascending, lane c = float c),
// the partial sums are added in a fixed order and the TOTAL replaces the first row of the Gaussian's region.  The
// projection backward then treats such a Gaussian as having that one row (whether or not its own pair was processed), in
// every part and slice.  Deterministic: fixed partition, fixed order.  Which rows exist: the tiles' stop keys (round 5; a
// flag byte per row until then), exactly the test of the rgb reader below.
#ifndef GS_PB_SH_BIG
#define GS_PB_SH_BIG 64
#endif
template <int CDIM>
__global__ void __launch_bounds__(1024) sh_big_rows_kernel(const uint4 *__restrict__ rects,
                                                          const uint32_t *__restrict__ pair_offsets,
                                                          float *__restrict__ rows,
                                                          const unsigned long long *__restrict__ stop_keys,
                                                          const float4 *__restrict__ rec_geom, uint32_t ntx, uint32_t n_tiles,
                                                          int cull_method, GsDistCull D,
                                                          int64_t n, uint64_t max_pairs, int64_t per_block) {
    constexpr int RWF = gs_row_floats(CDIM), WAVES = 16;
    static_assert(RWF <= 64, "a row is read by one wave instruction");
    __shared__ uint8_t s_big[1024];
    __shared__ float s_part[WAVES][RWF];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t *stop_depth = reinterpret_cast<const uint32_t *>(stop_keys), *stop_id = stop_depth + n_tiles;
    const int64_t g_begin = (int64_t)blockIdx.x * per_block;
    const int64_t g_end = g_begin + per_block < n ? g_begin + per_block : n;
    for (int64_t b0 = g_begin; b0 < g_end; b0 += 1024) {  // (uniform trip count)
        const int64_t i = b0 + threadIdx.x;
        const uint4 rc = i < g_end ? rects[i] : make_uint4(0, 0, 0, 0);
        const bool big = rc.z != 0 && rc.w > (uint32_t)GS_PB_SH_BIG;
        if (!__syncthreads_or(big)) continue;  // nearly every batch
        s_big[threadIdx.x] = big ? 1 : 0;
        __syncthreads();
        for (int j = 0; j < 1024; ++j) {
            if (!s_big[j]) continue;  // uniform
            const int64_t g = b0 + j;
            const uint64_t off = pair_offsets[g];
            const uint4 grc = rects[g];  // (uniform)
            const uint32_t gy0 = grc.x & 0xffff, gx0 = grc.y & 0xffff, gw = (grc.y >> 16) - (grc.y & 0xffff);
            float gcx = 0.f, gcy = 0.f;
            if (cull_method == 0) {  // "dist": not every tile of the bounding square is listed
                const float4 gg = rec_geom[g * GS_REC_STRIDE];
                gcx = gg.x;
                gcy = gg.y;
            }
            uint64_t cnt = grc.w;
            if (off >= max_pairs) cnt = 0;
            else if (off + cnt > max_pairs) cnt = max_pairs - off;
            const uint32_t chunk = (uint32_t)((cnt + WAVES - 1) / WAVES);
            const uint32_t k1 = (uint32_t)((uint64_t)(wv + 1) * chunk < cnt ? (uint64_t)(wv + 1) * chunk : cnt);
            float acc = 0.f;
            for (uint32_t k = (uint32_t)wv * chunk; k < k1; k += 64) {
                const uint32_t kk = k + (uint32_t)lane;
                bool ex = false;
                if (kk < k1) {  // row kk = tile (gx0 + kk % gw, gy0 + kk / gw): processed iff key(g) <= the tile's stop key
                    const uint32_t iy = gy0 + kk / gw, ix = gx0 + kk % gw, t = iy * ntx + ix;
                    const uint32_t sd = stop_depth[t];
                    ex = grc.z < sd || (grc.z == sd && (uint32_t)g <= stop_id[t]);
                    if (cull_method == 0 && !gs_dist_listed(gcx, gcy, ix, iy, D)) ex = false;
                }
                unsigned long long m = __ballot(ex);
                while (m) {  // the window's existing rows in ascending order, eight loads in flight
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        v[u] = 0.f;
                        if (m) {
                            const uint32_t r = (uint32_t)__ffsll((long long)m) - 1;
                            m &= m - 1;
                            if (lane < RWF) v[u] = rows[(off + k + r) * RWF + lane];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc += v[u];
                }
            }
            if (lane < RWF) s_part[wv][lane] = acc;
            __syncthreads();  // every wave has read its rows (the first row among them) and left its partial sums
            if (threadIdx.x < (unsigned)RWF && cnt) {
                const int c = threadIdx.x;
                float t[WAVES];
#pragma unroll
                for (int w = 0; w < WAVES; ++w) t[w] = s_part[w][c];
#pragma unroll
                for (int st = 1; st < WAVES; st <<= 1)  // fixed pairwise tree
#pragma unroll
                    for (int w = 0; w < WAVES; w += 2 * st) t[w] += t[w + st];
                rows[off * RWF + c] = t[0];
            }
            __syncthreads();
        }
    }
}

#ifndef GS_PB_DIRECT
#define GS_PB_DIRECT 2  // A/B switch (tools/ab_variants.py): how the rgb rows are fetched, see below
#endif
#ifndef GS_PB_SH_PASSES
#define GS_PB_SH_PASSES 2  // A/B switch: the SH row walk in this many passes over 64 / PASSES owners each (LDS per wave)
#endif
#ifndef GS_PB_SH_U
#define GS_PB_SH_U 8  // A/B switch: row loads the SH walk keeps in flight
#endif
// The optimizer step fused into the kernel's epilogue (gs_frame_backward_adam, include/gs_abi.h; rgb colours, PART 0): instead of
// storing the Gaussian's 14 gradients the thread applies gs_adam_one to its 14 parameters -- the five raw parameter arrays
// are then written through `p_*` (the same memory the kernel read them from: every thread reads its own Gaussian's parameters
// before it overwrites them, nobody else's), the moments through m_* / v_*.  ADAM: 0 = off, 1 = plain accesses, 2 = moments
// with non-temporal loads / stores (beyond the Infinity Cache they only evict each other on their way through: adam.hip).
struct AdamFusedDev {
    float *p_pos, *p_quat, *p_scale, *p_opa, *p_rgb;
    float *m_pos, *m_quat, *m_scale, *m_opa, *m_rgb;
    float *v_pos, *v_quat, *v_scale, *v_opa, *v_rgb;
    float step_pos, step_quat, step_scale, step_opa, step_rgb;  // lr_k / (1 - b1^t)
    float one_m_b1, b2, one_m_b2, inv_bc2_sqrt, eps;
    float *stat;  // [N,3] or NULL
    int stat_mode;
    const unsigned long long *skip_if_nonzero;
};
template <int CDIM, int PART = 0, int BLOCK = (CDIM == 3 ? 256 : 128), int ADAM = 0>
__global__ void __launch_bounds__(BLOCK) frame_project_backward_kernel(
    const float *pos, const float4 *quat, const float *scale,
    int64_t n, ProjectParams P, const float4 *__restrict__ rec_geom,
    const float4 *__restrict__ rec_color, const float4 *__restrict__ rows,
    const unsigned long long *__restrict__ stop_keys, const float *opa_raw,
    const float *rgb_raw, GsDistCull D,
    const uint32_t *__restrict__ pair_offsets, const uint4 *__restrict__ rects, uint64_t max_pairs, int64_t g_first,
    float *__restrict__ grad_pos,
    float4 *__restrict__ grad_quat, float *__restrict__ grad_scale, float *__restrict__ grad_opa,
    float *__restrict__ grad_rgb, AdamFusedDev A = AdamFusedDev{}) {
    static_assert(ADAM == 0 || PART == 0, "the fused optimizer step: everything in one kernel");
    // rgb rows (round 4): only rows that EXIST are fetched.  71 % of the pairs of the 2.4 M scene lie behind their
    // tile's stop point and their rows are uninitialised memory; round 3 streamed all of them through LDS and looked at
    // the flags afterwards (PMC: 916 MB of traffic against 316 MB algorithmic).  Whether the row of pair (tile, g) was
    // written follows from one number per TILE -- the key of the last list entry the forward processed there (stop_keys,
    // raster_bwd.hip: stop_key_kernel): the tile's list ascends in (depth bits, Gaussian), so the row exists iff
    // key(g) <= stop key.  Every thread first turns its rectangle into a bit mask of existing rows (stop-key loads eight
    // at a time: a loop with one dependent load per row costs a memory round trip per row -- the first version of this
    // kernel: 278 us against round 3's 175), then adds the rows up in ascending order, as before: bitwise unchanged.
    //   GS_PB_DIRECT 2: the wave fetches the existing rows of its 64 Gaussians together, 16 rows per load instruction;
    //   GS_PB_DIRECT 1: every thread fetches its own existing rows (one aligned 64-byte line each, ~1 per visible
    //                   Gaussian at 2.4 M Gaussians), two rows in flight;
    //   GS_PB_DIRECT 0: the workgroup's contiguous row range goes through LDS chunk by chunk, four lanes per existing
    //                   row, and every thread adds its rows out of LDS.
    constexpr int CHUNK_ROWS = 512;  // staged variant: 24 KiB of LDS, three float4s (the 10 floats in use) per row
    __shared__ float4 s_rows[(CDIM == 3 && GS_PB_DIRECT == 0) ? CHUNK_ROWS * 3 : 1];
    __shared__ uint8_t s_flag[(CDIM == 3 && GS_PB_DIRECT == 0) ? CHUNK_ROWS : 1];
    const int64_t pid0 = (int64_t)blockIdx.x * blockDim.x + g_first, pid = pid0 + threadIdx.x;
    const int64_t pid_last = (pid0 + blockDim.x < n ? pid0 + blockDim.x : n) - 1;
    const bool valid = pid < n;
    (void)pid_last;
    (void)s_rows;
    (void)s_flag;
    // (y0 | y1 << 16, x0 | x1 << 16, depth bits, tiles touched); depth bits != 0 <=> visible (depth > near > 0).  The
    // record of a culled Gaussian is unspecified (frame_project_kernel does not write it): not read.
    const uint4 rc = valid ? rects[pid] : make_uint4(0, 0, 0, 0);
    const bool vis = rc.z != 0;
    // rgb colours: the record is not needed -- sigma(opa) and the sigma(colour)s are recomputed from the raw parameters
    // (two coalesced streams, the same instructions as project_one: the same bits) instead of gathering one 64-byte
    // line per visible Gaussian for 16 + 12 of its bytes; only the "dist" listing test needs the projected centre
    const bool need_rec = CDIM > 3 || P.cull_method == 0;
    const float4 g = (vis && need_rec) ? rec_geom[pid * GS_REC_STRIDE] : make_float4(0, 0, 0, 0);
    float gp[3] = {0, 0, 0}, gqr[4] = {0, 0, 0, 0}, gsr[3] = {0, 0, 0}, gopa = 0, gcol[3] = {0, 0, 0};
    constexpr int RW4 = gs_row_floats(CDIM) / 4;  // float4s per row
    float4 d0 = make_float4(0, 0, 0, 0), d1 = d0, d2 = d0;
    const uint64_t off = vis ? pair_offsets[pid] : 0, cnt = rc.w;
    // SH: the column sums of every Gaussian's rows, [Gaussian of the workgroup][sum] with an odd stride, the sums in the
    // COMPACT order (dx, dy, da, db, dc, dd, dopa, coefficient 0 ..): the row's padding floats (gs_frame_layout.h) are
    // neither loaded nor kept
    constexpr int RWF = 4 * RW4, RS = ((7 + CDIM + 3) / 4) * 4 + 1;
    __shared__ float s_sum[CDIM > 3 ? BLOCK / GS_PB_SH_PASSES * RS : 1];
    __shared__ uint32_t s_brow[CDIM > 3 ? BLOCK / 64 : 1][64], s_bown[CDIM > 3 ? BLOCK / 64 : 1][64];

    // the stop keys as two arrays of T words: depth bits, Gaussian index (stop_key_kernel)
    const uint32_t *stop_depth = reinterpret_cast<const uint32_t *>(stop_keys);
    const uint32_t n_tiles_pb = P.ntx * P.nty;
    const uint32_t *stop_id = stop_depth + n_tiles_pb;
    // rgb: does the row of tile t = (ix, iy) of the Gaussian (depth bits dz, index id) exist?  key(g) <= stop key(t)
    auto row_exists = [&](uint32_t t, uint32_t dz, uint32_t id, uint32_t ix, uint32_t iy, float cx, float cy) {
        if (P.cull_method == 0 && !gs_dist_listed(cx, cy, ix, iy, D)) return false;  // "dist": holes in the square
        const uint32_t sd = stop_depth[t];
        return dz < sd || (dz == sd && id <= stop_id[t]);
    };
    const uint32_t my_y0 = rc.x & 0xffff, my_x0 = rc.y & 0xffff, my_x1 = rc.y >> 16;

    // A Gaussian that covers hundreds of tiles (early in training from a sparse cloud; a scale that blew up) would
    // keep ONE thread adding its rows while 255 wait: 195 us instead of 40 us for this kernel in a 500 k-Gaussian fit.
    // Such Gaussians are summed by the whole workgroup first -- thread t takes rows t, t + 256, ... straight from
    // global memory (consecutive threads, consecutive rows), a fixed shuffle tree and a fixed wave order give the
    // total to the owning thread: deterministic -- and are skipped by the per-thread loops below.
#ifndef GS_PB_BIG
#define GS_PB_BIG 64  // rows beyond which the whole workgroup sums an rgb Gaussian (A/B switch; at most 256: the row masks)
#endif
    constexpr uint32_t BIG = GS_PB_BIG;
    constexpr int NA = 12;  // floats of an rgb row that are summed (10 in use)
    constexpr int NBIG = CDIM == 3 ? 256 : 1;  // (SH rows are summed by the whole wave anyway: below)
    __shared__ uint32_t s_nbig, s_big_owner[NBIG];
    __shared__ uint64_t s_big_off[NBIG];
    __shared__ uint32_t s_big_cnt[NBIG];
    __shared__ float s_big_part[4][CDIM == 3 ? NA : 1];
    __shared__ __attribute__((aligned(16))) float s_adam_tr[ADAM ? BLOCK * 3 : 1];  // the fused optimizer step's hand-over (below)
    (void)s_adam_tr;
    const bool big = CDIM == 3 && cnt > BIG;
    if (threadIdx.x == 0) s_nbig = 0;
    __syncthreads();
    if (big) {
        const uint32_t slot = atomicAdd(&s_nbig, 1u);
        s_big_owner[slot] = threadIdx.x;
        s_big_off[slot] = off;
        s_big_cnt[slot] = (uint32_t)cnt;
    }
    __syncthreads();
    const uint32_t nbig = CDIM == 3 ? s_nbig : 0;
    for (uint32_t b = 0; b < nbig; ++b) {
        const uint64_t boff = s_big_off[b];
        const uint32_t bcnt = s_big_cnt[b];
        // the owner's rectangle and key (read back from its rectangle record: uniform over the workgroup)
        const int64_t bpid = pid0 + s_big_owner[b];
        const uint4 brc = rects[bpid];
        const uint32_t by0 = brc.x & 0xffff, bx0 = brc.y & 0xffff, bw = (brc.y >> 16) - (brc.y & 0xffff);
        float bcx = 0.f, bcy = 0.f;
        if (P.cull_method == 0) {
            const float4 bg = rec_geom[bpid * GS_REC_STRIDE];
            bcx = bg.x;
            bcy = bg.y;
        }
        float acc[NA];
#pragma unroll
        for (int e = 0; e < NA; ++e) acc[e] = 0.f;
        for (uint32_t k = threadIdx.x; k < bcnt && boff + k < max_pairs; k += 256) {
            const uint32_t iy = by0 + k / bw, ix = bx0 + k % bw;
            if (!row_exists(iy * P.ntx + ix, brc.z, (uint32_t)bpid, ix, iy, bcx, bcy)) continue;
            const float4 *row = rows + (boff + k) * RW4;
#pragma unroll
            for (int m = 0; m < NA / 4; ++m) {
                const float4 r = row[m];
                acc[4 * m] += r.x; acc[4 * m + 1] += r.y; acc[4 * m + 2] += r.z; acc[4 * m + 3] += r.w;
            }
        }
#pragma unroll
        for (int e = 0; e < NA; ++e) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc[e] += __shfl_xor(acc[e], o, 64);
            if ((threadIdx.x & 63) == 0) s_big_part[threadIdx.x >> 6][e] = acc[e];
        }
        __syncthreads();
        if (threadIdx.x == s_big_owner[b]) {
            auto tot = [&](int e) {
                return (s_big_part[0][e] + s_big_part[1][e]) + (s_big_part[2][e] + s_big_part[3][e]);
            };
            d0 = make_float4(tot(0), tot(1), tot(2), tot(3));
            d1 = make_float4(tot(4), tot(5), tot(6), tot(7));
            d2 = make_float4(tot(8), tot(9), 0.f, 0.f);
        }
        __syncthreads();
    }
    if (CDIM == 3) {
        // ---- which of this Gaussian's (at most 256) rows exist: four 64-bit words, stop keys loaded eight at a time
        unsigned long long wmask[4] = {0ull, 0ull, 0ull, 0ull};
#ifndef GS_PB_DIAG
#define GS_PB_DIAG 0  // timing-only builds (tools/ab_variants.py): 1 = no mask, no rows; 2 = mask but no row loads
#endif
#if GS_PB_DIRECT == 2
        constexpr int WAVES = BLOCK / 64;
        __shared__ uint32_t s_list[WAVES][64];   // row (relative to `rows`) of entry e of the current batch
        __shared__ float4 s_win[WAVES][64 * 3];  // the 12 leading floats of the batch's rows
        auto wave_sync = [] {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        // (Measured and dropped, round 4: looking the stop keys up wave-cooperatively as well -- the lanes list the tiles of
        // their rectangles in owner order, the wave fetches 256 stop keys in one round trip, every owner compares its own
        // -- 0.125 - 0.133 ms against 0.129 - 0.131 ms for the per-thread walk below at 2.4 M Gaussians, 0.044 against
        // 0.043 ms at cfg2: the LDS hand-overs cost what the shorter dependency chain saves.)
#endif
        if (GS_PB_DIAG != 1 && vis && !big && cnt) {
            uint32_t ix = my_x0, iy = my_y0;  // tile of row k, advanced row by row (no division)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if ((uint32_t)w * 64u >= cnt) break;
                unsigned long long m = 0;
                for (uint32_t k0 = (uint32_t)w * 64u; k0 < (uint32_t)w * 64u + 64u && k0 < cnt; k0 += 8) {
                    uint32_t sd[8], tx8[8], ty8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const bool in = k0 + j < cnt;
                        tx8[j] = ix;
                        ty8[j] = iy;
                        sd[j] = in ? stop_depth[iy * P.ntx + ix] : 0u;  // 0: nothing processed / not a pair of mine
                        if (in && ++ix == my_x1) {
                            ix = my_x0;
                            ++iy;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const bool in = k0 + j < cnt && off + k0 + j < max_pairs;
                        bool yes = in && rc.z < sd[j];
                        // equal depth bits (the tile's stop entry itself, exact copies): the Gaussian index decides
                        if (in && rc.z == sd[j]) yes = (uint32_t)pid <= stop_id[ty8[j] * P.ntx + tx8[j]];
                        if (yes) m |= 1ull << ((k0 + j) & 63u);
                    }
                    if (P.cull_method == 0) {  // "dist": not every tile of the bounding square is listed (uniform branch)
                        for (int j = 0; j < 8; ++j)
                            if (k0 + j < cnt && !gs_dist_listed(g.x, g.y, tx8[j], ty8[j], D)) m &= ~(1ull << ((k0 + j) & 63u));
                    }
                }
                wmask[w] = m;
            }
        }
#if GS_PB_DIRECT == 2
        // ---- the WAVE fetches the existing rows of its 64 Gaussians together.  A thread fetching its own rows keeps the
        // wave in the loop for as long as its busiest lane has rows (a Gaussian in front of a dense region: 9+ rows, the
        // average: 1.06), with a memory round trip per pair of rows -- 74 of the kernel's 162 us in a timing-only build
        // (profiles/r04_g_project_backward_time_split_diag.txt).  Instead the lanes' existing rows are listed in owner
        // order (LDS), the wave loads them 16 per instruction -- four lanes per aligned 64-byte line, every lane busy --
        // into an LDS window, and every owner adds ITS rows out of LDS in ascending order: the same sums, bit for bit.
        {
            const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
            for (int w = 0; w < 4; ++w) {  // rows 64 w .. 64 w + 63 of every Gaussian (beyond the first word: rare)
                const unsigned long long wm = wmask[w];
                if (__ballot(wm != 0ull) == 0ull) continue;  // uniform
                const uint32_t mine = (uint32_t)__popcll(wm);
                const uint32_t incl = gs_wave_incl_scan_u32(mine), first = incl - mine;
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                unsigned long long cm = wm;  // this lane's rows not yet added
                uint32_t done = 0;
                for (uint32_t e0 = 0; e0 < total; e0 += 64) {  // uniform trip count
                    // 1. list the entries [e0, e0 + 64) in owner order: this lane's are first + done .. first + mine - 1
                    {
                        unsigned long long lm = cm;
                        for (uint32_t e = first + done; e < e0 + 64 && lm; ++e) {
                            if (e >= e0) s_list[wv][e - e0] = (uint32_t)(off + (uint32_t)w * 64u + (uint32_t)__ffsll((long long)lm) - 1u);
                            lm &= lm - 1;
                        }
                    }
                    wave_sync();
                    // 2. the wave loads the batch: lane l = quarter l % 4 of entry 16 it + l / 4 (quarter 3 is padding)
                    const uint32_t nb = total - e0 < 64 ? total - e0 : 64u;
                    float4 v[4];
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const uint32_t j = 16u * it + ((uint32_t)lane >> 2), q = (uint32_t)lane & 3u;
                        v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (j < nb && q < 3) v[it] = rows[(size_t)s_list[wv][j] * RW4 + q];
                    }
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const uint32_t j = 16u * it + ((uint32_t)lane >> 2), q = (uint32_t)lane & 3u;
                        if (j < nb && q < 3) s_win[wv][j * 3 + q] = v[it];
                    }
                    wave_sync();
                    // 3. every owner adds its rows of this batch, ascending
                    const uint32_t lo_e = first + done > e0 ? first + done : e0;
                    const uint32_t hi_e = first + mine < e0 + 64 ? first + mine : e0 + 64;
                    for (uint32_t x = lo_e; x < hi_e; ++x) {
                        const float4 *row = &s_win[wv][(x - e0) * 3];
                        const float4 r0 = row[0], r1 = row[1], r2 = row[2];
                        d0.x += r0.x; d0.y += r0.y; d0.z += r0.z; d0.w += r0.w;
                        d1.x += r1.x; d1.y += r1.y; d1.z += r1.z; d1.w += r1.w;
                        d2.x += r2.x; d2.y += r2.y;
                        cm &= cm - 1;  // consumed
                        ++done;
                    }
                    wave_sync();  // the batch arrays are rewritten next
                }
            }
        }
#elif GS_PB_DIRECT
        // ---- every thread adds its existing rows in ascending order, two rows (six loads) in flight
        const float4 *myrows = rows + off * RW4;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            unsigned long long m = wmask[w];
            if (GS_PB_DIAG == 2) {  // keep the mask alive, fetch nothing
                d0.x += (float)__popcll(m);
                m = 0;
            }
            while (m) {
                const uint32_t ka = (uint32_t)__ffsll((long long)m) - 1;
                m &= m - 1;
                const bool two = m != 0;
                const uint32_t kb = two ? (uint32_t)__ffsll((long long)m) - 1 : ka;
                if (two) m &= m - 1;
                const float4 *ra = myrows + (size_t)(w * 64 + ka) * RW4, *rb = myrows + (size_t)(w * 64 + kb) * RW4;
                const float4 a0 = ra[0], a1 = ra[1], a2 = ra[2], b0 = rb[0], b1 = rb[1], b2 = rb[2];
                d0.x += a0.x; d0.y += a0.y; d0.z += a0.z; d0.w += a0.w;
                d1.x += a1.x; d1.y += a1.y; d1.z += a1.z; d1.w += a1.w;
                d2.x += a2.x; d2.y += a2.y;
                if (two) {
                    d0.x += b0.x; d0.y += b0.y; d0.z += b0.z; d0.w += b0.w;
                    d1.x += b1.x; d1.y += b1.y; d1.z += b1.z; d1.w += b1.w;
                    d2.x += b2.x; d2.y += b2.y;
                }
            }
        }
#else
        auto mask_bit = [&](uint32_t k) -> bool {  // k < 256
            const unsigned long long m = k < 64 ? wmask[0] : k < 128 ? wmask[1] : k < 192 ? wmask[2] : wmask[3];
            return (m >> (k & 63u)) & 1ull;
        };
        uint64_t row_begin = pair_offsets[pid0];
        uint64_t row_end = (uint64_t)pair_offsets[pid_last] + rects[pid_last].w;
        if (row_end > max_pairs) row_end = max_pairs;
        for (uint64_t base = row_begin; base < row_end; base += CHUNK_ROWS) {
            const uint32_t nrows = row_end - base < CHUNK_ROWS ? (uint32_t)(row_end - base) : (uint32_t)CHUNK_ROWS;
            const uint64_t lo = off > base ? off : base, hi = off + cnt < base + nrows ? off + cnt : base + nrows;
            // 1. every thread marks which of ITS rows inside this chunk exist (a row belongs to exactly one Gaussian;
            //    the rows of "big" Gaussians were summed above and are marked absent)
            for (uint64_t k = lo; k < hi; ++k) s_flag[k - base] = !big && mask_bit((uint32_t)(k - off));
            __syncthreads();
            // 2. existing rows -> LDS, four lanes per 64-byte row (the fourth quarter is padding: not fetched); all
            //    loads of the chunk are issued before the first one is stored
            const float4 *src = rows + base * RW4;
            constexpr int PER = CHUNK_ROWS * 4 / 256;
            float4 v[PER];
            bool take[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const uint32_t i = threadIdx.x + 256u * u, r = i >> 2, q = i & 3;
                take[u] = i < nrows * 4 && q < 3 && s_flag[r];
                if (take[u]) v[u] = src[i];
            }
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const uint32_t i = threadIdx.x + 256u * u, r = i >> 2, q = i & 3;
                if (take[u]) s_rows[r * 3 + q] = v[u];
            }
            __syncthreads();
            // 3. every thread adds its existing rows in ascending order
            for (uint64_t k = lo; k < hi; ++k) {
                if (!s_flag[k - base]) continue;
                const float4 *row = s_rows + (k - base) * 3;
                const float4 r0 = row[0], r1 = row[1], r2 = row[2];
                d0.x += r0.x; d0.y += r0.y; d0.z += r0.z; d0.w += r0.w;
                d1.x += r1.x; d1.y += r1.y; d1.z += r1.z; d1.w += r1.w;
                d2.x += r2.x; d2.y += r2.y;
            }
            __syncthreads();
        }
#endif
    } else {
        // SH rows are 144 (224) contiguous bytes.  A thread walking its own rows issues, per row, nine (fourteen) loads
        // whose 64 lanes touch 64 different rows: the texture-address unit serialises them lane by lane -- PMC, round 2:
        // 157 such loads per wave, 0.48 ms for this kernel, with VALU and HBM both far from busy.  Instead every WAVE
        // walks the written rows of its 64 Gaussians one row per load instruction, lane c reading float c of the row
        // (one or two cache lines per instruction), and keeps the running column sums of the current Gaussian in a
        // register per lane:
        //   1. every lane (as the owner of a Gaussian) turns the one-byte flags of its next 64 rows into a bit mask --
        //      4-byte loads, four in flight;
        //   2. the set bits of all 64 owners are laid out in owner order, 64 entries (row, owner) at a time, in LDS;
        //   3. the wave takes the entries in order, eight row loads in flight; when the owner changes, the finished
        //      sums go to s_sum[owner][c] and the next owner's partial sums (zero, or what an earlier window left) come
        //      back.  A Gaussian's rows are added in ascending order from zero, exactly as its own thread did: the
        //      results are bitwise what they were, for any PART.
        // Gaussians with thousands of rows need no special path any more: the wave works through them at one row per
        // instruction.
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        // The walk runs GS_PB_SH_PASSES times, each over the rows of 64 / PASSES owners (lanes [OWN pass, OWN (pass + 1))): the
        // column sums need [OWN][RS] floats of LDS per wave instead of [64][RS] -- 30 KiB per workgroup at degree 3 left 10 of
        // the CU's 32 wave slots filled, and the walk is bound by the latency of its row loads, i.e. by how many waves wait
        // at once.  Every owner's rows are still added in ascending order from zero: bitwise the same sums.
        constexpr int OWN = 64 / GS_PB_SH_PASSES;
        float *wsum = s_sum + (size_t)wv * OWN * RS;
        // (a Gaussian beyond GS_PB_SH_BIG rows: its first row holds the total of all of them, sh_big_rows_kernel)
        const bool big_sh = cnt > (uint64_t)GS_PB_SH_BIG;
        const uint64_t nrow_all = off + cnt < max_pairs ? cnt : (max_pairs > off ? max_pairs - off : 0);
        const uint64_t nrow = big_sh ? (nrow_all ? 1 : 0) : nrow_all;
        uint32_t maxrows = (uint32_t)(nrow < 0xffffffffull ? nrow : 0xffffffffull);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t x = __shfl_xor(maxrows, o, 64);
            maxrows = x > maxrows ? x : maxrows;
        }
        const float *rowf = reinterpret_cast<const float *>(rows);
        const int cidx = lane < RWF ? gs_row_compact(CDIM, lane) : -1;  // this lane's float of a row, in the compact order
        auto wave_sync = [] {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        // Which of this lane's rows exist (at most 64: a Gaussian beyond GS_PB_SH_BIG = 64 rows presents one), once for both
        // passes: key(g) <= the stop key of the row's tile (round 5; until then a flag byte per row, written by the raster
        // backward and cleared by a memset per frame).  Stop keys eight at a time, the tile advanced row by row, as in the
        // rgb branch above.
        unsigned long long written_all = 0;
        if (big_sh) {
            written_all = nrow ? 1ull : 0ull;
        } else if (nrow) {
            const uint32_t m = nrow < 64 ? (uint32_t)nrow : 64u;  // (nrow <= GS_PB_SH_BIG = 64 here)
            uint32_t iy = my_y0, ix = my_x0;
            for (uint32_t j0 = 0; j0 < m; j0 += 8) {
                uint32_t sd[8], tx8[8], ty8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool in = j0 + j < m;
                    tx8[j] = ix;
                    ty8[j] = iy;
                    sd[j] = in ? stop_depth[iy * P.ntx + ix] : 0u;
                    if (in && ++ix == my_x1) {
                        ix = my_x0;
                        ++iy;
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool in = j0 + j < m;
                    bool yes = in && rc.z < sd[j];
                    if (in && rc.z == sd[j]) yes = (uint32_t)pid <= stop_id[ty8[j] * P.ntx + tx8[j]];
                    if (yes && P.cull_method == 0 && !gs_dist_listed(g.x, g.y, tx8[j], ty8[j], D)) yes = false;
                    if (yes) written_all |= 1ull << (j0 + j);
                }
            }
        }
        static_assert(GS_PB_SH_BIG <= 64, "one 64-bit row mask per Gaussian");
#ifndef GS_PB_SH_DIAG
#define GS_PB_SH_DIAG 0  // timing-only builds (tools/ab_variants.py): 2 = the existence mask is built but no row is walked
#endif
        if (GS_PB_SH_DIAG == 2) {
            asm volatile("" ::"v"((uint32_t)written_all), "v"((uint32_t)(written_all >> 32)));  // (the mask stays alive)
            written_all = 0;
            maxrows = 0;
        }
        for (int pass = 0; pass < GS_PB_SH_PASSES; ++pass) {
        const int own0 = pass * OWN;
        const bool mine_pass = lane >= own0 && lane < own0 + OWN;
        for (int i = lane; i < OWN * RS; i += 64) wsum[i] = 0.f;
        wave_sync();
        float acc = 0.f;
        int cur = -1;  // owner whose sums `acc` holds (wave-uniform)
        for (uint32_t k0 = 0; k0 < maxrows; k0 += 64) {  // windows of 64 rows per owner (uniform trip count)
            const unsigned long long written = (k0 == 0 && mine_pass) ? written_all : 0ull;
            const uint32_t mine = (uint32_t)__popcll(written);
            const uint32_t incl = gs_wave_incl_scan_u32(mine), first = incl - mine;
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            for (uint32_t e0 = 0; e0 < total; e0 += 64) {  // batches of 64 entries, in owner order (uniform)
                {
                    unsigned long long mm = written;
                    uint32_t e = first;
                    while (mm && e < e0 + 64) {
                        const uint32_t k = (uint32_t)__ffsll((long long)mm) - 1;
                        mm &= mm - 1;
                        if (e >= e0) {
                            s_brow[wv][e - e0] = (uint32_t)(off + k0 + k);  // < max_pairs < 2^30
                            s_bown[wv][e - e0] = (uint32_t)lane;
                        }
                        ++e;
                    }
                }
                wave_sync();
                const uint32_t nb = total - e0 < 64 ? total - e0 : 64u;
                constexpr uint32_t U = GS_PB_SH_U;  // row loads in flight per wave
                for (uint32_t e = 0; e < nb; e += U) {
                    float v[U];
                    uint32_t own[U];
#pragma unroll
                    for (uint32_t u = 0; u < U; ++u) {
                        const bool ok = e + u < nb;
                        const uint32_t row = __builtin_amdgcn_readfirstlane(s_brow[wv][ok ? e + u : 0]);
                        own[u] = __builtin_amdgcn_readfirstlane(s_bown[wv][ok ? e + u : 0]);
                        v[u] = (ok && cidx >= 0) ? rowf[(size_t)row * RWF + lane] : 0.f;
                    }
#pragma unroll
                    for (uint32_t u = 0; u < U; ++u) {
                        if (e + u >= nb) break;  // uniform
                        if ((int)own[u] != cur) {  // uniform
                            if (cur >= 0 && cidx >= 0) wsum[(cur - own0) * RS + cidx] = acc;
                            cur = (int)own[u];
                            acc = cidx >= 0 ? wsum[(cur - own0) * RS + cidx] : 0.f;
                        }
                        acc += v[u];
                    }
                }
                wave_sync();  // the batch arrays are rewritten next
            }
        }
        if (cur >= 0 && cidx >= 0) wsum[(cur - own0) * RS + cidx] = acc;
        wave_sync();
        if (mine_pass) {
            const float *t = wsum + (lane - own0) * RS;  // this thread's Gaussian: (dx, dy, da, db | dc, dd, dopa, coefficient 0 | ...)
            d0 = make_float4(t[0], t[1], t[2], t[3]);
            d1 = make_float4(t[4], t[5], t[6], t[7]);
        }
        if (PART != 1) {
            // coefficient gradients: CDIM consecutive floats per Gaussian in grad_rgb, the pass's OWN Gaussians -- written
            // by the wave as one contiguous run (a culled Gaussian's sums are the zeros the array started with)
            const int64_t g0w = pid0 + (int64_t)wv * 64 + own0;
            const int ng = n - g0w < OWN ? (int)(n - g0w) : OWN;  // Gaussians of this pass inside the array (may be <= 0)
            if constexpr (ADAM != 0) {
                // the fused optimizer step of the run's coefficients (round 6: SH colours too): the wave walks its contiguous
                // run of the coefficient array and of the two moments -- a kilobyte per instruction -- and applies gs_adam_one
                // with the gradients it would have stored.  Nothing else reads the raw coefficients
                // in this kernel, and a Gaussian's coefficients belong to this wave alone.
                if (!(A.skip_if_nonzero && *A.skip_if_nonzero)) {  // (uniform: an overflowed frame takes no step)
                    float *pp = A.p_rgb + g0w * CDIM, *mm = A.m_rgb + g0w * CDIM, *vv = A.v_rgb + g0w * CDIM;
                    const int ne = ng > 0 ? ng * CDIM : 0, ne4 = ne & ~3;
                    auto one = [&](float &pe, float &me, float &ve, int e) {
                        const int gl = e / CDIM, c = e - gl * CDIM;
                        gs_adam_one(pe, wsum[gl * RS + 7 + c], me, ve, A.step_rgb, A.one_m_b1, A.b2, A.one_m_b2, A.inv_bc2_sqrt,
                                    A.eps);
                    };
                    // float4 by float4 (the run starts at a multiple of 32 Gaussians: 16-byte aligned with the arrays) ...
                    for (int e = lane * 4; e < ne4; e += 256) {
                        typedef float nt4v __attribute__((ext_vector_type(4)));
                        float4 pv = *reinterpret_cast<const float4 *>(pp + e), mv, vw;
                        if (ADAM == 2) {
                            const nt4v a = __builtin_nontemporal_load(reinterpret_cast<const nt4v *>(mm + e));
                            const nt4v b = __builtin_nontemporal_load(reinterpret_cast<const nt4v *>(vv + e));
                            mv = make_float4(a.x, a.y, a.z, a.w), vw = make_float4(b.x, b.y, b.z, b.w);
                        } else {
                            mv = *reinterpret_cast<const float4 *>(mm + e);
                            vw = *reinterpret_cast<const float4 *>(vv + e);
                        }
                        one(pv.x, mv.x, vw.x, e);
                        one(pv.y, mv.y, vw.y, e + 1);
                        one(pv.z, mv.z, vw.z, e + 2);
                        one(pv.w, mv.w, vw.w, e + 3);
                        *reinterpret_cast<float4 *>(pp + e) = pv;
                        if (ADAM == 2) {
                            __builtin_nontemporal_store(nt4v{mv.x, mv.y, mv.z, mv.w}, reinterpret_cast<nt4v *>(mm + e));
                            __builtin_nontemporal_store(nt4v{vw.x, vw.y, vw.z, vw.w}, reinterpret_cast<nt4v *>(vv + e));
                        } else {
                            *reinterpret_cast<float4 *>(mm + e) = mv;
                            *reinterpret_cast<float4 *>(vv + e) = vw;
                        }
                    }
                    // ... and the up to three elements an array that ends inside the run leaves over
                    if (const int e = ne4 + lane; e < ne) {
                        float pe = pp[e], me = mm[e], ve = vv[e];
                        one(pe, me, ve, e);
                        pp[e] = pe, mm[e] = me, vv[e] = ve;
                    }
                }
            } else {
                float *dst = grad_rgb + g0w * CDIM;
                for (int e = lane; e < ng * CDIM; e += 64) {
                    const int gl = e / CDIM, c = e - gl * CDIM;
                    dst[e] = wsum[gl * RS + 7 + c];
                }
            }
        }
        wave_sync();  // (the next pass clears the sums)
        }
    }
    if (ADAM == 0 && !valid) return;  // (ADAM: the epilogue's LDS hand-over is the whole wave's; an invalid thread is culled: zeros)
    if (vis && PART != 2) {
        float p[3], sraw[3], q[4], s[3];
        load3(pos, pid, p);
        load3(scale, pid, sraw);
        float4 q4 = quat[pid];
        float qraw[4] = {q4.x, q4.y, q4.z, q4.w};
        activate(qraw, sraw, P.scale_act, q, s);
        float gi[3] = {d0.x, d0.y, 0.0f}, g2[4] = {d0.z, d0.w, d1.x, d1.y}, gq[4], gs[3];
        project_backward(p, q, s, P.cam, gi, g2, gp, gq, gs);
        // q_hat = q / |q|  ->  dq = (dq_hat - q_hat (q_hat . dq_hat)) / |q|
        const float inr = gs_rsq(qraw[0] * qraw[0] + qraw[1] * qraw[1] + qraw[2] * qraw[2] + qraw[3] * qraw[3]);
        float dt = q[0] * gq[0] + q[1] * gq[1] + q[2] * gq[2] + q[3] * gq[3];
#pragma unroll
        for (int k = 0; k < 4; ++k) gqr[k] = (gq[k] - q[k] * dt) * inr;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (P.scale_act == 0)  // |s| + 1e-4 : d/ds = sign(s)
                gsr[k] = sraw[k] > 0 ? gs[k] : (sraw[k] < 0 ? -gs[k] : 0.0f);
            else  // trunc_exp backward (renderer.py:97-100): g * exp(clamp(x, -1, 1))
                gsr[k] = gs[k] * expf(fminf(fmaxf(sraw[k], -1.0f), 1.0f));
        }
    }
    if (vis && PART != 1) {
        if (CDIM == 3) {
            // sigma(opa), sigma(colour): the expressions of project_one, which wrote them into the record
            const float so = sigmoid_f(opa_raw[pid]);
            const float c0 = sigmoid_f(rgb_raw[pid * 3 + 0]), c1 = sigmoid_f(rgb_raw[pid * 3 + 1]);
            const float c2 = sigmoid_f(rgb_raw[pid * 3 + 2]);
            gopa = d1.z * so * (1.0f - so);
            gcol[0] = d1.w * c0 * (1.0f - c0);
            gcol[1] = d2.x * c1 * (1.0f - c1);
            gcol[2] = d2.y * c2 * (1.0f - c2);
        } else {
            gopa = d1.z * g.w * (1.0f - g.w);
        }
    }
    if constexpr (ADAM != 0) {
        // ---- the optimizer step of the wave's 64 x 14 parameters (gs_adam_one: torch's _single_tensor_adam); a culled
        // Gaussian takes its zero-gradient step (momentum), as gs_adam_step gives it.  Every global access is a whole
        // float4 per lane over a contiguous run of the wave: a thread owns a GAUSSIAN, but the three [N, 3] arrays (and
        // their moments) are walked by ELEMENT -- the wave's 192 gradients go through LDS once ([Gaussian][3] in, float4
        // by float4 out, 48 lanes) and lane l updates elements 4 l .. 4 l + 3 of the wave's run.  (First version, r05_q:
        // every thread its own 14 parameters, 84 four-byte accesses at a 12-byte stride -- 0.19 ms SLOWER than
        // backward + gs_adam_step at 2.4 M Gaussians.)  The lanes that write a Gaussian's parameters are lanes of the
        // wave that read them (above, in program order): nobody else's.  (Measured and dropped, r5s: every load of the step
        // issued first -- one round trip per wave instead of five, 86 VGPRs and 9 KiB more LDS: 1,750 against 1,785 it/s
        // at 376 k Gaussians, 942 against 950 at 2.4 M; the other waves of the CU already cover the round trips.)
        if (A.skip_if_nonzero && *A.skip_if_nonzero) return;  // the frame overflowed and was rendered empty: no step (uniform)
        typedef float nt4 __attribute__((ext_vector_type(4)));
        auto ld4 = [](const float *q) {
            if (ADAM == 2) {
                const nt4 x = __builtin_nontemporal_load(reinterpret_cast<const nt4 *>(q));
                return make_float4(x.x, x.y, x.z, x.w);
            }
            return *reinterpret_cast<const float4 *>(q);
        };
        auto st4 = [](float *q, float4 x) {
            if (ADAM == 2)
                __builtin_nontemporal_store(nt4{x.x, x.y, x.z, x.w}, reinterpret_cast<nt4 *>(q));
            else
                *reinterpret_cast<float4 *>(q) = x;
        };
        auto one4 = [&](float4 &pv, float4 gv, float4 &mv, float4 &vv, float step) {
            gs_adam_one(pv.x, gv.x, mv.x, vv.x, step, A.one_m_b1, A.b2, A.one_m_b2, A.inv_bc2_sqrt, A.eps);
            gs_adam_one(pv.y, gv.y, mv.y, vv.y, step, A.one_m_b1, A.b2, A.one_m_b2, A.inv_bc2_sqrt, A.eps);
            gs_adam_one(pv.z, gv.z, mv.z, vv.z, step, A.one_m_b1, A.b2, A.one_m_b2, A.inv_bc2_sqrt, A.eps);
            gs_adam_one(pv.w, gv.w, mv.w, vv.w, step, A.one_m_b1, A.b2, A.one_m_b2, A.inv_bc2_sqrt, A.eps);
        };
        const int a_lane = threadIdx.x & 63, a_wv = threadIdx.x >> 6;
        float *tr = s_adam_tr + a_wv * 192;
        const int64_t wbase = pid0 + (int64_t)a_wv * 64, left = n - wbase;  // the wave's first Gaussian; how many lie inside
        const int ne = left >= 64 ? 192 : (left > 0 ? (int)left * 3 : 0);  // elements of the wave's run of an [N, 3] array
        const int e0 = a_lane * 4;
        auto step3 = [&](float *P3, float *M3, float *V3, const float (&g3)[3], float step, float *stat) {
            tr[a_lane * 3 + 0] = g3[0];
            tr[a_lane * 3 + 1] = g3[1];
            tr[a_lane * 3 + 2] = g3[2];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (e0 + 4 <= ne) {
                const int64_t b = wbase * 3 + e0;
                const float4 gv = *reinterpret_cast<const float4 *>(tr + e0);
                float4 pv = *reinterpret_cast<const float4 *>(P3 + b), mv = ld4(M3 + b), vv = ld4(V3 + b);
                one4(pv, gv, mv, vv, step);
                *reinterpret_cast<float4 *>(P3 + b) = pv;
                st4(M3 + b, mv);
                st4(V3 + b, vv);
                if (stat) {
                    float4 sv = *reinterpret_cast<const float4 *>(stat + b);
                    if (A.stat_mode == 1)
                        sv = make_float4(fmaxf(sv.x, fabsf(gv.x)), fmaxf(sv.y, fabsf(gv.y)), fmaxf(sv.z, fabsf(gv.z)),
                                         fmaxf(sv.w, fabsf(gv.w)));
                    else
                        sv = make_float4(sv.x + fabsf(gv.x), sv.y + fabsf(gv.y), sv.z + fabsf(gv.z), sv.w + fabsf(gv.w));
                    *reinterpret_cast<float4 *>(stat + b) = sv;
                }
            } else {
                for (int e = e0; e < ne; ++e) {  // the ragged end of the array (N not a multiple of 4): element by element
                    const int64_t b = wbase * 3 + e;
                    const float ge = tr[e];
                    float pe = P3[b], me = M3[b], ve = V3[b];
                    gs_adam_one(pe, ge, me, ve, step, A.one_m_b1, A.b2, A.one_m_b2, A.inv_bc2_sqrt, A.eps);
                    P3[b] = pe;
                    M3[b] = me;
                    V3[b] = ve;
                    if (stat) stat[b] = A.stat_mode == 1 ? fmaxf(stat[b], fabsf(ge)) : stat[b] + fabsf(ge);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();  // (the next array's gradients overwrite tr)
        };
        step3(A.p_pos, A.m_pos, A.v_pos, gp, A.step_pos, A.stat_mode ? A.stat : nullptr);
        step3(A.p_scale, A.m_scale, A.v_scale, gsr, A.step_scale, nullptr);
        if constexpr (CDIM == 3) step3(A.p_rgb, A.m_rgb, A.v_rgb, gcol, A.step_rgb, nullptr);  // (SH: stepped by the wave, above)
        if (!valid) return;
        {  // the quaternion: the thread's own float4s
            float4 pv = *reinterpret_cast<const float4 *>(A.p_quat + pid * 4), mv = ld4(A.m_quat + pid * 4);
            float4 vv = ld4(A.v_quat + pid * 4);
            one4(pv, make_float4(gqr[0], gqr[1], gqr[2], gqr[3]), mv, vv, A.step_quat);
            *reinterpret_cast<float4 *>(A.p_quat + pid * 4) = pv;
            st4(A.m_quat + pid * 4, mv);
            st4(A.v_quat + pid * 4, vv);
        }
        float po = A.p_opa[pid], mo = A.m_opa[pid], vo = A.v_opa[pid];
        gs_adam_one(po, gopa, mo, vo, A.step_opa, A.one_m_b1, A.b2, A.one_m_b2, A.inv_bc2_sqrt, A.eps);
        A.p_opa[pid] = po;
        A.m_opa[pid] = mo;
        A.v_opa[pid] = vo;
        return;
    }
    if (PART != 2) {
        grad_pos[pid * 3 + 0] = gp[0];
        grad_pos[pid * 3 + 1] = gp[1];
        grad_pos[pid * 3 + 2] = gp[2];
        grad_quat[pid] = make_float4(gqr[0], gqr[1], gqr[2], gqr[3]);
        grad_scale[pid * 3 + 0] = gsr[0];
        grad_scale[pid * 3 + 1] = gsr[1];
        grad_scale[pid * 3 + 2] = gsr[2];
    }
    if (PART == 1) return;
    grad_opa[pid] = gopa;
    if constexpr (CDIM == 3) {
        grad_rgb[pid * 3 + 0] = gcol[0];
        grad_rgb[pid * 3 + 1] = gcol[1];
        grad_rgb[pid * 3 + 2] = gcol[2];
    }  // (SH: the coefficient gradients were written by the wave, above)
}

inline int grid_for(int64_t n, int block) {
    int64_t g = gs_div_up(n, block);
    if (g > 8192) g = 8192;  // 256 CUs x 8 blocks x 4: grid-stride beyond that
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

// ================================================================= C ABI (section A)
extern "C" int gs_world2camera(const float *pos, const float *rot, const float *tran, float *res, int64_t B,
                               gs_stream_t stream) {
    GS_CHECK_ARG(B >= 0, "B < 0");
    if (B == 0) return 0;
    GS_CHECK_ARG(pos && rot && tran && res, "null pointer");
    hipLaunchKernelGGL(world2camera_kernel, dim3(grid_for(B, 256)), dim3(256), 0, (hipStream_t)stream, pos, rot,
                       tran, res, B);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_world2camera_backward(const float *grad_out, const float *rot, float *grad_inp, int64_t B,
                                        gs_stream_t stream) {
    GS_CHECK_ARG(B >= 0, "B < 0");
    if (B == 0) return 0;
    GS_CHECK_ARG(grad_out && rot && grad_inp, "null pointer");
    hipLaunchKernelGGL(world2camera_backward_kernel, dim3(grid_for(B, 256)), dim3(256), 0, (hipStream_t)stream,
                       grad_out, rot, grad_inp, B);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_jacobian(const float *pos_cam, float *jac, int64_t B, gs_stream_t stream) {
    GS_CHECK_ARG(B >= 0, "B < 0");
    if (B == 0) return 0;
    GS_CHECK_ARG(pos_cam && jac, "null pointer");
    hipLaunchKernelGGL(jacobian_kernel, dim3(grid_for(B, 256)), dim3(256), 0, (hipStream_t)stream, pos_cam, jac, B);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_global_culling(const float *pos, const float *quat, const float *scale, const float *rot,
                                 const float *tran, int64_t N, float near_plane, float half_width,
                                 float half_height, float *res_pos, float *res_cov, int64_t *culling_mask,
                                 gs_stream_t stream) {
    GS_CHECK_ARG(N >= 0, "N < 0");
    if (N == 0) return 0;
    GS_CHECK_ARG(pos && quat && scale && rot && tran && res_pos && res_cov && culling_mask, "null pointer");
    GS_CHECK_ARG(((uintptr_t)quat & 15) == 0 && ((uintptr_t)res_cov & 15) == 0, "quat/res_cov must be 16-byte aligned");
    hipLaunchKernelGGL(global_culling_kernel, dim3(grid_for(N, 256)), dim3(256), 0, (hipStream_t)stream, pos,
                       (const float4 *)quat, scale, rot, tran, N, near_plane, half_width, half_height, res_pos,
                       (float4 *)res_cov, culling_mask);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_global_culling_backward(const float *pos, const float *quat, const float *scale,
                                          const float *rot, const float *tran, int64_t N,
                                          const float *gradout_pos, const float *gradout_cov,
                                          const int64_t *culling_mask, float *gradinput_pos,
                                          float *gradinput_quat, float *gradinput_scale, gs_stream_t stream) {
    GS_CHECK_ARG(N >= 0, "N < 0");
    if (N == 0) return 0;
    GS_CHECK_ARG(pos && quat && scale && rot && tran && gradout_pos && gradout_cov && culling_mask &&
                     gradinput_pos && gradinput_quat && gradinput_scale,
                 "null pointer");
    GS_CHECK_ARG(((uintptr_t)quat & 15) == 0 && ((uintptr_t)gradout_cov & 15) == 0 &&
                     ((uintptr_t)gradinput_quat & 15) == 0,
                 "quat/gradout_cov/gradinput_quat must be 16-byte aligned");
    hipLaunchKernelGGL(global_culling_backward_kernel, dim3(grid_for(N, 256)), dim3(256), 0, (hipStream_t)stream,
                       pos, (const float4 *)quat, scale, rot, tran, N, gradout_pos, (const float4 *)gradout_cov,
                       culling_mask, gradinput_pos, (float4 *)gradinput_quat, gradinput_scale);
    GS_CHECK_LAUNCH();
    return 0;
}

// ================================================================= frame stages (internal)
static ProjectParams make_params(const gs_frame *f) {
    ProjectParams P;
    for (int i = 0; i < 9; ++i) P.cam.rot[i] = f->rot[i];
    for (int i = 0; i < 3; ++i) P.cam.tran[i] = f->tran[i];
    P.near_plane = f->near_plane;
    P.half_w = f->half_width;
    P.half_h = f->half_height;
    P.tlog = -2 * logf(f->thresh);
    P.dist_thresh = f->thresh;
    P.dist_radius = sqrtf(f->thresh);
    gs_frame_geom G = gs_frame_geometry(f);
    P.tlx = G.tlx;
    P.tly = G.tly;
    P.leftmost = G.leftmost;
    P.topmost = G.topmost;
    P.ntx = (uint32_t)G.ntx;
    P.nty = (uint32_t)G.nty;
    P.scale_act = f->scale_activation;
    P.color_dim = f->color_dim;
    P.cull_method = f->tile_culling_method;
    P.half_padw = (float)(G.padW / 2);
    P.half_padh = (float)(G.padH / 2);
    P.fx = f->focal_x;
    P.fy = f->focal_y;
    P.inv_tlx = 1.0f / G.tlx;
    P.inv_tly = 1.0f / G.tly;
    {   // largest singular value of the camera rotation as given (power iteration on W^T W; 1 for a rotation)
        double A[9], v[3] = {0.6, 0.5, 0.62}, lam = 1.0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                A[i * 3 + j] = 0;
                for (int k = 0; k < 3; ++k) A[i * 3 + j] += (double)f->rot[k * 3 + i] * (double)f->rot[k * 3 + j];
            }
        for (int it = 0; it < 48; ++it) {
            double w[3];
            for (int i = 0; i < 3; ++i) w[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
            lam = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
            if (!(lam > 0)) break;
            for (int i = 0; i < 3; ++i) v[i] = w[i] / lam;
        }
        // (power iteration approaches the largest eigenvalue from below: 1 % on top; trace as the fail-safe upper bound)
        const double tr = A[0] + A[4] + A[8];
        double sig = sqrt(lam) * 1.01;
        if (!(sig > 0) || !(sig <= sqrt(tr) * 1.01)) sig = sqrt(tr) * 1.01;
        P.occ_k = (float)(1.02 * sqrt((double)P.tlog) * sig);
    }
    return P;
}

// slice_begin / slice_end: the slices of the Gaussian array to project (strip variant with the fused count only:
// gs_frame_project_slices; every other path projects everything at once: 0, -1)
// `second_pass`: the unculled re-run of a GS_FRAME_OCCLUSION_CULL frame's project stage, gated on counters[GS_CNT_RANPAST]
int gs_stage_project(const gs_frame *f, const gs_frame_ws &ws, hipStream_t stream, int slice_begin, int slice_end,
                     bool second_pass) {
    ProjectParams P = make_params(f);
    // sort_modes 0 / 1 read tiles_touched (emit_pairs_kernel); sort_mode 2 reads the rectangle records only
    uint32_t *touched = f->sort_mode == 2 && gs_frame_geometry(f).n_tiles <= GS_BIN_MAX_TILES ? nullptr : ws.tiles_touched;
    if (gs_frame_uses_strips(f)) touched = nullptr;
    if (gs_frame_fused_count(f)) {
        gs_frame_geom G = gs_frame_geometry(f);
        const gs_strip_plan plan = gs_strip_plan_for(f->N, G.ntx, G.nty);
        const gs_strip_geom SG = plan.geom;
        GsDistCull D = {(float)(G.padW / 2), (float)(G.padH / 2), f->focal_x, f->focal_y, f->thresh};
        static std::mutex attr_mu;
        static std::atomic<uint64_t> attr_done{0};
        int dev = 0;
        GS_HIP(hipGetDevice(&dev));
        if (dev < 64 && !((attr_done.load(std::memory_order_acquire) >> dev) & 1)) {
            std::lock_guard<std::mutex> lock(attr_mu);
            for (const void *fn : {(const void *)frame_project_count_kernel<false>, (const void *)frame_project_count_kernel<true>,
                                   (const void *)frame_project_cull_count_kernel})
                // (the kernels also hold ~17 KiB of static LDS -- the tile-order workgroup's bins --: the strip histogram, and the
                // cut pyramid + survivor queue behind it in a culled frame, get what gs_frame_occlusion_cull's room rule allows)
                GS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, GS_BIN_LDS_BYTES - 8 * 4096));
            attr_done.fetch_or(1ull << dev, std::memory_order_release);
        }
        unsigned long long *table = (unsigned long long *)ws.strip_table;
        if (slice_end < 0) slice_end = (int)plan.slices;
        GS_CHECK_ARG(slice_begin >= 0 && slice_begin < slice_end && slice_end <= (int)plan.slices, "bad slice range");
        const uint32_t nsl = (uint32_t)(slice_end - slice_begin);
        if (gs_frame_occlusion_cull(f) && !second_pass) {
            // GS_FRAME_OCCLUSION_CULL, first pass: Gaussians behind every cut they can reach are not projected, the level-1
            // entries of the others are trimmed by the cut table the previous frame of this workspace left
            GS_CHECK_ARG(slice_begin == 0 && nsl == plan.slices, "an occlusion-culled frame is projected in one piece");
            // (GS_OCC_QCAP / GS_OCC_STASH: smaller chunks / stash, read per frame -- the parity tests walk the several-chunks and
            // beyond-the-stash paths at sizes a test can afford; a slice has more than 16,384 Gaussians only beyond 4.2 M)
            uint32_t qcap_max = GS_OCC_QCAP;
            if (const char *e = getenv("GS_OCC_QCAP")) {
                const long v = atol(e);
                if (v >= 64 && v <= (long)GS_OCC_QCAP) qcap_max = (uint32_t)v;
            }
            const uint32_t qcap = plan.per_slice < qcap_max ? plan.per_slice : qcap_max;
            static const uint32_t diag = getenv("GS_OCC_DIAG") ? (uint32_t)atoi(getenv("GS_OCC_DIAG")) : 0u;  // timing-only builds of the kernel's phases
            size_t lds = sizeof(unsigned long long) * SG.NS + gs_cull_pyramid_bytes(G.ntx, G.nty) + 2 * (size_t)qcap + 16;
            // what is left of the kernel's LDS room holds positions and scales of the chunk's first survivors (24 B each)
            const size_t room = (size_t)GS_BIN_LDS_BYTES - 8 * 4096;
            uint32_t stash_cap = room > lds ? (uint32_t)((room - lds) / 24) & ~63u : 0u;
            if (stash_cap > qcap) stash_cap = (qcap + 63u) & ~63u;
            if (const char *e = getenv("GS_OCC_STASH")) {
                const long v = atol(e);
                if (v >= 0 && (uint32_t)v < stash_cap) stash_cap = (uint32_t)v & ~63u;
            }
            lds += (size_t)stash_cap * 24;
            hipLaunchKernelGGL(frame_project_cull_count_kernel, dim3(nsl + 1), dim3(STRIP_THREADS), lds, stream, f->pos,
                               (const float4 *)f->quat, f->scale, f->opa, f->rgb, f->N, P, ws.rec_geom, ws.rects,
                               plan.per_slice, SG, nsl, table, ws.slice_pairs, ws.slice_vis, ws.tile_cost,
                               (uint32_t)G.n_tiles, ws.tile_order, gs_frame_cut_table(f, ws), qcap, stash_cap, ws.surv,
                               ws.slice_nsurv, diag);
            GS_CHECK_LAUNCH();
            return 0;
        }
        const size_t lds = sizeof(unsigned long long) * SG.NS;
        const uint32_t extra = (slice_begin == 0 && !second_pass) ? 1u : 0u;  // the tile-order workgroup rides with the first range
        const unsigned long long *gate = second_pass ? ws.counters + GS_CNT_RANPAST : nullptr;
#define GS_LAUNCH_PROJECT_COUNT(DIST)                                                                                  \
    hipLaunchKernelGGL(frame_project_count_kernel<DIST>, dim3(nsl + extra), dim3(STRIP_THREADS), lds, stream,          \
                       f->pos, (const float4 *)f->quat, f->scale, f->opa, f->rgb, f->N, P, ws.rec_geom, touched,       \
                       ws.rects, D, plan.per_slice, SG, nsl, (uint32_t)slice_begin, table, ws.slice_pairs,             \
                       ws.slice_vis, ws.tile_cost, (uint32_t)G.n_tiles, ws.tile_order, gate)
        if (f->tile_culling_method == 0)
            GS_LAUNCH_PROJECT_COUNT(true);
        else
            GS_LAUNCH_PROJECT_COUNT(false);
#undef GS_LAUNCH_PROJECT_COUNT
        GS_CHECK_LAUNCH();
        return 0;
    }
    GS_CHECK_ARG(!second_pass, "the second pass of an occlusion-culled frame belongs to the fused project + count stage");
    GS_CHECK_ARG(slice_begin == 0 && slice_end < 0, "this frame's project stage cannot be issued in ranges");
    if (gs_frame_fused_table_count(f)) {
        gs_frame_geom G = gs_frame_geometry(f);
        GsDistCull D = {(float)(G.padW / 2), (float)(G.padH / 2), f->focal_x, f->focal_y, f->thresh};
        static std::mutex attr_mu2;
        static std::atomic<uint64_t> attr_done2{0};
        int dev = 0;
        GS_HIP(hipGetDevice(&dev));
        if (dev < 64 && !((attr_done2.load(std::memory_order_acquire) >> dev) & 1)) {
            std::lock_guard<std::mutex> lock(attr_mu2);
            for (const void *fn : {(const void *)frame_project_bin_count_kernel<false>, (const void *)frame_project_bin_count_kernel<true>})
                GS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, GS_BIN_MAX_TILES * 4));
            attr_done2.fetch_or(1ull << dev, std::memory_order_release);
        }
        const uint32_t per_block = bin_per_block(f->N), B = (uint32_t)gs_div_up(f->N, per_block);
        const uint32_t T = (uint32_t)G.n_tiles;
#define GS_LAUNCH_PROJECT_BIN(DIST)                                                                                    \
    hipLaunchKernelGGL(frame_project_bin_count_kernel<DIST>, dim3(B), dim3(BIN_THREADS), sizeof(uint32_t) * T, stream, \
                       f->pos, (const float4 *)f->quat, f->scale, f->opa, f->rgb, f->N, P, ws.rec_geom, ws.rects, D,   \
                       per_block, T, ws.bin_table, ws.slice_pairs, ws.slice_vis)
        if (f->tile_culling_method == 0)
            GS_LAUNCH_PROJECT_BIN(true);
        else
            GS_LAUNCH_PROJECT_BIN(false);
#undef GS_LAUNCH_PROJECT_BIN
        GS_CHECK_LAUNCH();
        return 0;
    }
    int nblk = (int)gs_div_up(f->N, 256);
    hipLaunchKernelGGL(frame_project_kernel, dim3(nblk), dim3(256), 0, stream, f->pos, (const float4 *)f->quat,
                       f->scale, f->opa, f->rgb, f->N, P, ws.rec_geom,
                       touched, ws.rects, ws.block_sums, ws.block_vis);
    GS_CHECK_LAUNCH();
    return 0;
}

// Gaussians [g_begin, g_end) only (g_begin a multiple of 256; the whole array: 0, N): the view-parallel exchange sums
// the rows slice by slice, so that a slice's gradients travel while the next slice is summed (gs_dp.py).
// SH frames, once per backward, behind the raster backward: the rows of Gaussians beyond GS_PB_SH_BIG rows are summed
// by whole workgroups (sh_big_rows_kernel)
int gs_stage_sh_big_rows(const gs_frame *f, const gs_frame_ws &ws, hipStream_t stream) {
    if (f->color_dim == 3 || f->N <= 0) return 0;
    const int blocks = 512;
    const int64_t per_block = gs_div_up(f->N, blocks);  // (the kernel walks its slice 1,024 Gaussians at a time)
    gs_frame_geom G = gs_frame_geometry(f);
    GsDistCull D = {(float)(G.padW / 2), (float)(G.padH / 2), f->focal_x, f->focal_y, f->thresh};
#define GS_LAUNCH_BIG_ROWS(CD)                                                                                          \
    hipLaunchKernelGGL(sh_big_rows_kernel<CD>, dim3(blocks), dim3(1024), 0, stream, ws.rects, ws.pair_offsets, ws.rows, \
                       (const unsigned long long *)ws.stop_keys, ws.rec_geom, (uint32_t)G.ntx, (uint32_t)G.n_tiles,     \
                       f->tile_culling_method, D, f->N, (uint64_t)f->max_pairs, per_block)
    if (f->color_dim == 48)
        GS_LAUNCH_BIG_ROWS(48);
    else
        GS_LAUNCH_BIG_ROWS(27);
#undef GS_LAUNCH_BIG_ROWS
    GS_CHECK_LAUNCH();
    return 0;
}

// The projection backward with the Adam step in its epilogue (gs_frame_backward_adam): rgb colours, all Gaussians, one launch
// Everything gs_frame_backward_adam can reject about its optimizer argument, checked BEFORE anything is enqueued (ADVICE
// round 5: a call rejected behind the raster backward left the caller's step counter ahead of the moments).
int gs_validate_adam_fused(const gs_frame *f, const gs_adam_fused *a) {
    GS_CHECK_ARG(f->color_dim == 3 || f->color_dim == 27 || f->color_dim == 48, "color_dim must be 3, 27 or 48");
    GS_CHECK_ARG(a->step >= 1, "step counts from 1 (torch.optim.Adam increments before the update)");
    GS_CHECK_ARG(a->beta1 >= 0.f && a->beta1 < 1.f && a->beta2 >= 0.f && a->beta2 < 1.f && a->eps >= 0.f, "bad hyper-parameters");
    GS_CHECK_ARG(a->stat_mode >= 0 && a->stat_mode <= 2 && (!a->stat_mode || a->grad_stat), "bad statistic");
    for (int k = 0; k < 5; ++k) GS_CHECK_ARG(a->exp_avg[k] && a->exp_avg_sq[k], "null moment pointer");
    {  // the kernel walks the arrays float4 by float4
        const void *al[] = {f->pos, f->quat, f->scale, f->rgb, a->exp_avg[0], a->exp_avg[1], a->exp_avg[2], a->exp_avg[4],
                            a->exp_avg_sq[0], a->exp_avg_sq[1], a->exp_avg_sq[2], a->exp_avg_sq[4], a->grad_stat};
        for (const void *q : al) GS_CHECK_ARG(((uintptr_t)q & 15) == 0, "parameters, moments and statistic must be 16-byte aligned");
    }
    return 0;
}

int gs_stage_project_backward_adam(const gs_frame *f, const gs_frame_ws &ws, const gs_adam_fused *a, hipStream_t stream) {
    int vrc = gs_validate_adam_fused(f, a);
    if (vrc) return vrc;
    if (f->N <= 0) return 0;
    ProjectParams P = make_params(f);
    gs_frame_geom Gf = gs_frame_geometry(f);
    GsDistCull Dc = {(float)(Gf.padW / 2), (float)(Gf.padH / 2), f->focal_x, f->focal_y, f->thresh};
    // bias corrections on the host in double, as torch does (adam.hip: adam_step_impl)
    const double bc1 = 1.0 - pow((double)a->beta1, (double)a->step), bc2 = 1.0 - pow((double)a->beta2, (double)a->step);
    AdamFusedDev A;
    A.p_pos = const_cast<float *>(f->pos);
    A.p_quat = const_cast<float *>(f->quat);
    A.p_scale = const_cast<float *>(f->scale);
    A.p_opa = const_cast<float *>(f->opa);
    A.p_rgb = const_cast<float *>(f->rgb);
    A.m_pos = a->exp_avg[0]; A.m_quat = a->exp_avg[1]; A.m_scale = a->exp_avg[2]; A.m_opa = a->exp_avg[3]; A.m_rgb = a->exp_avg[4];
    A.v_pos = a->exp_avg_sq[0]; A.v_quat = a->exp_avg_sq[1]; A.v_scale = a->exp_avg_sq[2]; A.v_opa = a->exp_avg_sq[3]; A.v_rgb = a->exp_avg_sq[4];
    A.step_pos = (float)((double)a->lr[0] / bc1);
    A.step_quat = (float)((double)a->lr[1] / bc1);
    A.step_scale = (float)((double)a->lr[2] / bc1);
    A.step_opa = (float)((double)a->lr[3] / bc1);
    A.step_rgb = (float)((double)a->lr[4] / bc1);
    A.one_m_b1 = 1.0f - a->beta1;
    A.b2 = a->beta2;
    A.one_m_b2 = 1.0f - a->beta2;
    A.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    A.eps = a->eps;
    A.stat = a->grad_stat;
    A.stat_mode = a->stat_mode;
    A.skip_if_nonzero = (const unsigned long long *)a->skip_if_nonzero;
    // (11 + C) parameters x 16 bytes of arrays: beyond the Infinity Cache the moments stream with non-temporal accesses (adam.hip)
    const bool nt = (unsigned long long)f->N * (unsigned long long)(11 + f->color_dim) * 16ull > (300ull << 20);
#define GS_LAUNCH_PB_ADAM(CD, BLK, MODE)                                                                                \
    hipLaunchKernelGGL((frame_project_backward_kernel<CD, 0, BLK, MODE>), dim3((unsigned)gs_div_up(f->N, BLK)),         \
                       dim3(BLK), 0, stream, f->pos,                                                                   \
                       (const float4 *)f->quat, f->scale, f->N, P, ws.rec_geom, ws.rec_color, (const float4 *)ws.rows, \
                       (const unsigned long long *)ws.stop_keys, f->opa, f->rgb, Dc, ws.pair_offsets, ws.rects,        \
                       (uint64_t)f->max_pairs, (int64_t)0, (float *)nullptr, (float4 *)nullptr, (float *)nullptr,      \
                       (float *)nullptr, (float *)nullptr, A)
    if (f->color_dim == 48) {
        if (nt) GS_LAUNCH_PB_ADAM(48, 128, 2); else GS_LAUNCH_PB_ADAM(48, 128, 1);
    } else if (f->color_dim == 27) {
        if (nt) GS_LAUNCH_PB_ADAM(27, 128, 2); else GS_LAUNCH_PB_ADAM(27, 128, 1);
    } else {
        if (nt) GS_LAUNCH_PB_ADAM(3, 256, 2); else GS_LAUNCH_PB_ADAM(3, 256, 1);
    }
#undef GS_LAUNCH_PB_ADAM
    GS_CHECK_LAUNCH();
    return 0;
}

int gs_stage_project_backward(const gs_frame *f, const gs_frame_ws &ws, float *grad_pos, float *grad_quat,
                              float *grad_scale, float *grad_opa, float *grad_rgb, int part, int64_t g_begin,
                              int64_t g_end, hipStream_t stream) {
    if (g_end <= g_begin) return 0;
    ProjectParams P = make_params(f);
    gs_frame_geom Gf = gs_frame_geometry(f);
    GsDistCull Dc = {(float)(Gf.padW / 2), (float)(Gf.padH / 2), f->focal_x, f->focal_y, f->thresh};
#define GS_LAUNCH_PROJECT_BWD(CD, PT)                                                                              \
    hipLaunchKernelGGL((frame_project_backward_kernel<CD, PT>),                                                   \
                       dim3((unsigned)gs_div_up(g_end - g_begin, CD == 3 ? 256 : 128)),                           \
                       dim3(CD == 3 ? 256 : 128), 0, stream, f->pos,                                              \
                       (const float4 *)f->quat, f->scale, g_end, P, ws.rec_geom, ws.rec_color,                    \
                       (const float4 *)ws.rows, (const unsigned long long *)ws.stop_keys, f->opa,                 \
                       f->rgb, Dc, ws.pair_offsets, ws.rects, (uint64_t)f->max_pairs, g_begin,                    \
                       grad_pos, (float4 *)grad_quat, grad_scale, grad_opa, grad_rgb)
#define GS_LAUNCH_PROJECT_BWD_PARTS(CD)  \
    do {                                 \
        if (part == 1)                   \
            GS_LAUNCH_PROJECT_BWD(CD, 1); \
        else if (part == 2)              \
            GS_LAUNCH_PROJECT_BWD(CD, 2); \
        else                             \
            GS_LAUNCH_PROJECT_BWD(CD, 0); \
    } while (0)
    if (f->color_dim == 48)
        GS_LAUNCH_PROJECT_BWD_PARTS(48);
    else if (f->color_dim == 27)
        GS_LAUNCH_PROJECT_BWD_PARTS(27);
    else
        GS_LAUNCH_PROJECT_BWD_PARTS(3);
#undef GS_LAUNCH_PROJECT_BWD_PARTS
#undef GS_LAUNCH_PROJECT_BWD
    GS_CHECK_LAUNCH();
    return 0;
}
