/*
 * gs_oracle.c -- CPU restatement of the reference rasterizer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the package
 * `3d-gaussian-splatting_amd/`) may import, link or call this file.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * Every function restates one piece of /root/reference (WangFeng18/3d-gaussian-splatting)
 * and cites the file:line it follows.  The arithmetic keeps the reference's C type
 * promotions (float vs double literals) so that the fp32 build is the closest thing to
 * "what the CUDA kernel computes" that a CPU can produce; the reference's nvcc build would
 * additionally contract a*b+c into FMAs, which is unknowable from the source, so the
 * canonical oracle is IEEE fp32 in SOURCE ORDER with NO contraction (build with
 * -ffp-contract=off).
 *
 * Pinning status: the reference ships no tests/golden vectors.  The
 * oracle is pinned against the reference's own kernels compiled for the CPU through the SIMT
 * emulator in oracle/ref_harness.cpp (outputs in oracle/_ref/, fixtures in tests/golden/),
 * against closed-form known answers, and against torch.autograd on oracle/torch_ref.py.
 *
 * Semantics deliberately NOT reproduced (reference defects):
 *   - backward shared-memory gradient slots not re-zeroed between chunks (gaussian.cu:508-522),
 *   - forward inter-chunk shared-memory race (gaussian.cu:878-962),
 *   - partial-mask warp shuffles (gaussian.cu:675-687),
 *   - racy check-then-atomicAdd cap (gaussian.cu:244-247): the oracle applies the cap
 *     serially in Gaussian-index order when `maxp` > 0.
 *
 * Threading: the per-pixel loop of K7 and the per-tile loop of K8 are independent work items and
 * carry `#pragma omp parallel for` (build with -fopenmp).  Nothing is summed across threads, so
 * the results are bit-identical for any thread count (OMP_NUM_THREADS=1 is the scalar port).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GSO_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------------------------------
 * legacy helpers: world2camera (gaussian.cu:49-69), backward (:78-93), jacobian (:10-39)
 * ------------------------------------------------------------------------------------- */
GSO_API void gso_world2camera(const float *pos, const float *rot, const float *tran,
                              float *res, int64_t B) {
    for (int64_t i = 0; i < B; ++i) {
        const float *p = pos + 3 * i;
        float *r = res + 3 * i;
        /* gaussian.cu:66-68 */
        r[0] = p[0] * rot[0] + p[1] * rot[1] + p[2] * rot[2] + tran[0];
        r[1] = p[0] * rot[3] + p[1] * rot[4] + p[2] * rot[5] + tran[1];
        r[2] = p[0] * rot[6] + p[1] * rot[7] + p[2] * rot[8] + tran[2];
    }
}

GSO_API void gso_world2camera_backward(const float *grad_out, const float *rot,
                                       float *grad_inp, int64_t B) {
    for (int64_t i = 0; i < B; ++i) {
        const float *g = grad_out + 3 * i;
        float *o = grad_inp + 3 * i;
        /* gaussian.cu:90-92 */
        o[0] = g[0] * rot[0] + g[1] * rot[3] + g[2] * rot[6];
        o[1] = g[0] * rot[1] + g[1] * rot[4] + g[2] * rot[7];
        o[2] = g[0] * rot[2] + g[1] * rot[5] + g[2] * rot[8];
    }
}

/* rsqrtf on the device is an approximate op; the oracle uses 1/sqrtf (correctly rounded
 * twice).  Only row 2 of the Jacobian (unused by the 2x2 covariance) depends on it. */
static void calc_jacobian(const float *u, float *J) {
    /* gaussian.cu:1156-1180 (== :10-39) */
    float u0 = u[0], u1 = u[1], u2 = u[2];
    J[0] = 1 / u2;
    J[1] = 0;
    J[2] = -u0 / (u2 * u2);
    J[3] = 0;
    J[4] = 1 / u2;
    J[5] = -u1 / (u2 * u2);
    float rs = 1.0f / sqrtf(u0 * u0 + u1 * u1 + u2 * u2);
    J[6] = rs * u0;
    J[7] = rs * u1;
    J[8] = rs * u2;
}

GSO_API void gso_jacobian(const float *pos_cam, float *jac, int64_t B) {
    for (int64_t i = 0; i < B; ++i) calc_jacobian(pos_cam + 3 * i, jac + 9 * i);
}

/* ---------------------------------------------------------------------------------------
 * K1: frustum cull + EWA projection            gaussian.cu:1131-1154, 1182-1336
 * ------------------------------------------------------------------------------------- */
static void world_to_camera(const float *p, const float *rot, const float *tran, float *pc) {
    /* gaussian.cu:1150-1153 */
    for (int i = 0; i < 3; ++i)
        pc[i] = rot[i * 3 + 0] * p[0] + rot[i * 3 + 1] * p[1] + rot[i * 3 + 2] * p[2] + tran[i];
}

static void quat_to_R(const float *q, float *R) {
    /* gaussian.cu:1231-1245 */
    float w = q[0], x = q[1], y = q[2], z = q[3];
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

/* C = A * B, 3x3, accumulation order of the reference loops (start at 0, k ascending). */
static void mm3(const float *A, const float *B, float *C) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            float s = 0;
            for (int k = 0; k < 3; ++k) s += A[r * 3 + k] * B[k * 3 + c];
            C[r * 3 + c] = s;
        }
}
/* C = A * B^T */
static void mm3_nt(const float *A, const float *B, float *C) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            float s = 0;
            for (int k = 0; k < 3; ++k) s += A[r * 3 + k] * B[c * 3 + k];
            C[r * 3 + c] = s;
        }
}

GSO_API void gso_global_culling(const float *pos, const float *quat, const float *scale,
                                const float *rot, const float *tran, int64_t n, float near_,
                                float half_w, float half_h, float *res_pos, float *res_cov,
                                int64_t *mask) {
    for (int64_t pid = 0; pid < n; ++pid) {
        float pc[3];
        world_to_camera(pos + 3 * pid, rot, tran, pc);
        if (pc[2] <= near_) continue; /* :1208 */
        float pi[3];
        pi[0] = pc[0] / pc[2];
        pi[1] = pc[1] / pc[2];
        pi[2] = sqrtf(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]); /* :1215-1217 */
        if (fabsf(pi[0]) >= half_w || fabsf(pi[1]) >= half_h) continue; /* :1220 */
        mask[pid] = 1;
        res_pos[pid * 3 + 0] = pi[0];
        res_pos[pid * 3 + 1] = pi[1];
        res_pos[pid * 3 + 2] = pi[2];

        float R[9], S[9] = {0}, RS[9], RSSR[9], J[9], JW[9], JWC[9], JWCWJ[9];
        quat_to_R(quat + 4 * pid, R);
        S[0] = scale[pid * 3 + 0];
        S[4] = scale[pid * 3 + 1];
        S[8] = scale[pid * 3 + 2];
        mm3(R, S, RS);        /* :1259-1270 */
        mm3_nt(RS, RS, RSSR); /* :1272-1283 */
        calc_jacobian(pc, J);
        mm3(J, rot, JW);        /* :1292-1303 */
        mm3(JW, RSSR, JWC);     /* :1305-1316 */
        mm3_nt(JWC, JW, JWCWJ); /* :1318-1329 */
        res_cov[pid * 4 + 0] = JWCWJ[0];
        res_cov[pid * 4 + 1] = JWCWJ[1];
        res_cov[pid * 4 + 2] = JWCWJ[3];
        res_cov[pid * 4 + 3] = JWCWJ[4];
    }
}

/* ---------------------------------------------------------------------------------------
 * K2: cull/project backward                    gaussian.cu:1371-1576
 * ------------------------------------------------------------------------------------- */
GSO_API void gso_global_culling_backward(const float *pos, const float *quat, const float *scale,
                                         const float *rot, const float *tran, int64_t n,
                                         const float *gradout_pos, const float *gradout_cov,
                                         const int64_t *mask, float *gin_pos, float *gin_quat,
                                         float *gin_scale) {
    for (int64_t pid = 0; pid < n; ++pid) {
        if (mask[pid] == 0) continue; /* :1389 */
        float pc[3];
        world_to_camera(pos + 3 * pid, rot, tran, pc);
        float r = sqrtf(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]);
        float gi[3] = {gradout_pos[pid * 3], gradout_pos[pid * 3 + 1], gradout_pos[pid * 3 + 2]};
        float gc[3];
        /* :1404-1406 */
        gc[0] = gi[0] / pc[2] + gi[2] * pc[0] / r;
        gc[1] = gi[1] / pc[2] + gi[2] * pc[1] / r;
        gc[2] = -gi[0] * pc[0] / (pc[2] * pc[2]) - gi[1] * pc[1] / (pc[2] * pc[2]) +
                gi[2] * pc[2] / r;
        for (int ir = 0; ir < 3; ++ir) { /* :1411-1417 */
            float s = 0;
            for (int k = 0; k < 3; ++k) s += rot[k * 3 + ir] * gc[k];
            gin_pos[pid * 3 + ir] = s;
        }
        float J[9], JW[9];
        calc_jacobian(pc, J);
        mm3(J, rot, JW);
        float g2[4] = {gradout_cov[pid * 4], gradout_cov[pid * 4 + 1], gradout_cov[pid * 4 + 2],
                       gradout_cov[pid * 4 + 3]};
        float g3[9];
        for (int ir = 0; ir < 3; ++ir) /* :1449-1461 */
            for (int ic = 0; ic < 3; ++ic) {
                float s = 0;
                for (int ii = 0; ii < 2; ++ii)
                    for (int ij = 0; ij < 2; ++ij)
                        s += g2[ii * 2 + ij] * JW[ii * 3 + ir] * JW[ij * 3 + ic];
                g3[ir * 3 + ic] = s;
            }
        float R[9], S[9] = {0}, RS[9], gRS[9];
        quat_to_R(quat + 4 * pid, R);
        S[0] = scale[pid * 3 + 0];
        S[4] = scale[pid * 3 + 1];
        S[8] = scale[pid * 3 + 2];
        mm3(R, S, RS);
        for (int ir = 0; ir < 3; ++ir) /* :1506-1519 */
            for (int ic = 0; ic < 3; ++ic) {
                float s = 0;
                for (int k = 0; k < 3; ++k)
                    s += (g3[k * 3 + ir] + g3[ir * 3 + k]) * RS[k * 3 + ic];
                gRS[ir * 3 + ic] = s;
            }
        for (int i = 0; i < 3; ++i) /* :1522-1526 */
            gin_scale[pid * 3 + i] =
                gRS[0 * 3 + i] * R[0 * 3 + i] + gRS[1 * 3 + i] * R[1 * 3 + i] + gRS[2 * 3 + i] * R[2 * 3 + i];
        float sx = S[0], sy = S[4], sz = S[8];
        float qr = quat[4 * pid], qi = quat[4 * pid + 1], qj = quat[4 * pid + 2],
              qk = quat[4 * pid + 3];
        /* :1535-1554 */
        float c_qr[9] = {0, -2 * sy * qk, 2 * sz * qj, 2 * sx * qk, 0, -2 * sz * qi,
                         -2 * sx * qj, 2 * sy * qi, 0};
        float c_qi[9] = {0, 2 * sy * qj, 2 * sz * qk, 2 * sx * qj, -4 * sy * qi, -2 * sz * qr,
                         2 * sx * qk, 2 * sy * qr, -4 * sz * qi};
        float c_qj[9] = {-4 * sx * qj, 2 * sy * qi, 2 * sz * qr, 2 * sx * qi, 0, 2 * sz * qk,
                         -2 * sx * qr, 2 * sy * qk, -4 * sz * qj};
        float c_qk[9] = {-4 * sx * qk, -2 * sy * qr, 2 * sz * qi, 2 * sx * qr, -4 * sy * qk,
                         2 * sz * qj, 2 * sx * qi, 2 * sy * qj, 0};
        float gqr = 0, gqi = 0, gqj = 0, gqk = 0;
        for (int i = 0; i < 9; ++i) { /* :1560-1566 */
            gqr += c_qr[i] * gRS[i];
            gqi += c_qi[i] * gRS[i];
            gqj += c_qj[i] * gRS[i];
            gqk += c_qk[i] * gRS[i];
        }
        gin_quat[pid * 4 + 0] = gqr;
        gin_quat[pid * 4 + 1] = gqi;
        gin_quat[pid * 4 + 2] = gqj;
        gin_quat[pid * 4 + 3] = gqk;
    }
}

/* ---------------------------------------------------------------------------------------
 * K3/K4/K5: tile binning                       gaussian.cu:101-250, splatter.py:255-300
 * ------------------------------------------------------------------------------------- */
/* CUDA float->uint32 conversion (cvt.rzi.u32.f32): NaN -> 0, negative -> 0, saturating. */
static uint32_t f2u_sat(float v) {
    if (!(v > 0.0f)) return 0u;
    if (v >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)v;
}

/* Rectangle of tiles covered by one Gaussian under method 2 ("prob2", gaussian.cu:226-242).
 * Returns 0 if the Gaussian is dropped (det <= 0).  rect = {y0, y1, x0, x1}, half-open. */
GSO_API int gso_tile_rect(float cx, float cy, float a, float b, float c, float d, float thresh,
                          float tlx, float tly, uint32_t ntx, uint32_t nty, float leftmost,
                          float topmost, uint32_t *rect) {
    float det = (a * d - b * c);
    if (det <= 0) return 0;
    float ai = (float)(d / (det + 1e-14)); /* float / double -> double -> float, :229 */
    float di = (float)(a / (det + 1e-14)); /* :232 */
    float tlog = -2 * logf(thresh);        /* :233 */
    float shift_x = sqrtf(di * tlog * det);
    float shift_y = sqrtf(ai * tlog * det);
    float bbx_right = cx + shift_x, bbx_left = cx - shift_x;
    float bbx_top = cy - shift_y, bbx_bottom = cy + shift_y;
    /* :241-242.  `uint32_t i = fmaxf(.., 0)` then `i < (uint32_t)(.. + 1) && i < n` */
    uint32_t y0 = f2u_sat(fmaxf((bbx_top - topmost) / tly, 0));
    uint32_t y1 = f2u_sat((bbx_bottom - topmost) / tly + 1);
    uint32_t x0 = f2u_sat(fmaxf((bbx_left - leftmost) / tlx, 0));
    uint32_t x1 = f2u_sat((bbx_right - leftmost) / tlx + 1);
    if (y1 > nty) y1 = nty;
    if (x1 > ntx) x1 = ntx;
    if (y0 > y1) y0 = y1;
    if (x0 > x1) x0 = x1;
    rect[0] = y0;
    rect[1] = y1;
    rect[2] = x0;
    rect[3] = x1;
    return 1;
}

/* calc_tile_list (gaussian.cu:254-335).  tile_n_point[T] must be zeroed by the caller (as the
 * reference's caller does, splatter.py:567).  The list keeps at most maxp entries per tile
 * (serial, Gaussian-index order); the counter keeps counting past the cap for methods 0 and
 * is capped for methods 1/2 exactly as the reference's `if(n<max){atomicAdd}` does when run
 * serially.  top/bottom/left/right are only read by methods 0 and 1. */
GSO_API void gso_calc_tile_list(const float *pos, const float *cov, int64_t n_point,
                                const float *top, const float *bottom, const float *left,
                                const float *right, int32_t *tile_n_point, int32_t *list,
                                int64_t maxp, float thresh, int method, float tlx, float tly,
                                int32_t ntx, int32_t nty, float leftmost, float topmost) {
    int64_t n_tiles = (int64_t)ntx * nty;
    if (method == 2) {
        for (int64_t pid = 0; pid < n_point; ++pid) {
            uint32_t rc[4];
            if (!gso_tile_rect(pos[pid * 3], pos[pid * 3 + 1], cov[pid * 4], cov[pid * 4 + 1],
                               cov[pid * 4 + 2], cov[pid * 4 + 3], thresh, tlx, tly, ntx, nty,
                               leftmost, topmost, rc))
                continue;
            for (uint32_t iy = rc[0]; iy < rc[1]; ++iy)
                for (uint32_t ix = rc[2]; ix < rc[3]; ++ix) {
                    int64_t tid = ix + (int64_t)iy * ntx;
                    if (tile_n_point[tid] < maxp) { /* :244-247 */
                        int32_t old = tile_n_point[tid]++;
                        list[maxp * tid + old] = (int32_t)pid;
                    }
                }
        }
        return;
    }
    for (int64_t pid = 0; pid < n_point; ++pid) {
        float cx = pos[pid * 3], cy = pos[pid * 3 + 1];
        if (method == 0) {
            for (int64_t tid = 0; tid < n_tiles; ++tid) { /* :124-135 */
                float center_y = (top[tid] + bottom[tid]) / 2;
                float center_x = (left[tid] + right[tid]) / 2;
                float d1 = cx - center_x, d2 = cy - center_y;
                if (d1 * d1 + d2 * d2 < thresh) {
                    int32_t old = tile_n_point[tid]++;
                    if (old < maxp) list[maxp * tid + old] = (int32_t)pid;
                }
            }
        } else { /* method 1, :163-194 */
            float a = cov[pid * 4], b = cov[pid * 4 + 1], c = cov[pid * 4 + 2], d = cov[pid * 4 + 3];
            float det = (a * d - b * c);
            if (det <= 0) continue;
            float ai = (float)(d / (det + 1e-14));
            float di = (float)(a / (det + 1e-14));
            float tlog = -2 * logf(thresh);
            float shift_x = sqrtf(di * tlog * det);
            float shift_y = sqrtf(ai * tlog * det);
            float br = cx + shift_x, bl = cx - shift_x, bt = cy - shift_y, bb = cy + shift_y;
            for (int64_t tid = 0; tid < n_tiles; ++tid) {
                if (!(right[tid] < bl || br < left[tid] || bottom[tid] < bt || bb < top[tid])) {
                    if (tile_n_point[tid] < maxp) {
                        int32_t old = tile_n_point[tid]++;
                        list[maxp * tid + old] = (int32_t)pid;
                    }
                }
            }
        }
    }
}

/* gather_gaussians (gaussian.cu:337-381): compact the T x MAXP table with the exclusive scan. */
GSO_API void gso_gather_gaussians(const int32_t *accum, const int32_t *list, int32_t *gathered,
                                  int32_t *tile_ids, int64_t n_tiles, int64_t list_size) {
    for (int64_t tid = 0; tid < n_tiles; ++tid) {
        int32_t s = accum[tid], cnt = accum[tid + 1] - s;
        for (int32_t p = 0; p < cnt; ++p) {
            gathered[s + p] = list[tid * list_size + p];
            tile_ids[s + p] = (int32_t)tid;
        }
    }
}

/* ---------------------------------------------------------------------------------------
 * Conditioning scale of K2 (test infrastructure, see draw_backward_impl): the same formulas as
 * gso_global_culling_backward with every product taken between magnitudes and every sum / difference
 * replaced by the sum of the magnitudes of its operands (double).  s_pos[N,3] / s_cov[N,4] are the scales
 * of dL/dpos_i and dL/dcov; out_*: the scales of dL/dpos, dL/dquat_hat, dL/dscale_hat.  Any fp32
 * evaluation of K2 differs from another one by a modest number of ulp of these.
 * ------------------------------------------------------------------------------------- */
GSO_API void gso_global_culling_backward_scale(const float *pos, const float *quat, const float *scale,
                                               const float *rot, const float *tran, int64_t n, const double *s_pos,
                                               const double *s_cov, const int64_t *mask, double *out_pos,
                                               double *out_quat, double *out_scale) {
    for (int64_t pid = 0; pid < n; ++pid) {
        if (mask[pid] == 0) continue;
        float pcf[3];
        world_to_camera(pos + 3 * pid, rot, tran, pcf);
        const double pc[3] = {fabs((double)pcf[0]), fabs((double)pcf[1]), fabs((double)pcf[2])};
        const double r = sqrt(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]);
        const double gi[3] = {s_pos[pid * 3], s_pos[pid * 3 + 1], s_pos[pid * 3 + 2]};
        double gc[3];
        gc[0] = gi[0] / pc[2] + gi[2] * pc[0] / r;
        gc[1] = gi[1] / pc[2] + gi[2] * pc[1] / r;
        gc[2] = gi[0] * pc[0] / (pc[2] * pc[2]) + gi[1] * pc[1] / (pc[2] * pc[2]) + gi[2] * pc[2] / r;
        for (int ir = 0; ir < 3; ++ir) {
            double a = 0;
            for (int k = 0; k < 3; ++k) a += fabs((double)rot[k * 3 + ir]) * gc[k];
            out_pos[pid * 3 + ir] = a;
        }
        /* |J| (rows 0, 1) and |J| |W| */
        const double J[9] = {1 / pc[2], 0, pc[0] / (pc[2] * pc[2]), 0, 1 / pc[2], pc[1] / (pc[2] * pc[2]), 0, 0, 0};
        double JW[9];
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) {
                double v = 0;
                for (int k = 0; k < 3; ++k) v += J[a * 3 + k] * fabs((double)rot[k * 3 + c]);
                JW[a * 3 + c] = v;
            }
        const double g2[4] = {s_cov[pid * 4], s_cov[pid * 4 + 1], s_cov[pid * 4 + 2], s_cov[pid * 4 + 3]};
        double g3[9];
        for (int ir = 0; ir < 3; ++ir)
            for (int ic = 0; ic < 3; ++ic) {
                double v = 0;
                for (int ii = 0; ii < 2; ++ii)
                    for (int ij = 0; ij < 2; ++ij) v += g2[ii * 2 + ij] * JW[ii * 3 + ir] * JW[ij * 3 + ic];
                g3[ir * 3 + ic] = v;
            }
        const double w = fabs((double)quat[4 * pid]), x = fabs((double)quat[4 * pid + 1]),
                     y = fabs((double)quat[4 * pid + 2]), z = fabs((double)quat[4 * pid + 3]);
        const double R[9] = {1 + 2 * y * y + 2 * z * z, 2 * x * y + 2 * z * w, 2 * x * z + 2 * y * w,
                             2 * x * y + 2 * z * w, 1 + 2 * x * x + 2 * z * z, 2 * y * z + 2 * x * w,
                             2 * x * z + 2 * y * w, 2 * y * z + 2 * x * w, 1 + 2 * x * x + 2 * y * y};
        const double S[3] = {fabs((double)scale[pid * 3]), fabs((double)scale[pid * 3 + 1]),
                             fabs((double)scale[pid * 3 + 2])};
        double RS[9], gRS[9];
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) RS[a * 3 + c] = R[a * 3 + c] * S[c];
        for (int ir = 0; ir < 3; ++ir)
            for (int ic = 0; ic < 3; ++ic) {
                double v = 0;
                for (int k = 0; k < 3; ++k) v += (g3[k * 3 + ir] + g3[ir * 3 + k]) * RS[k * 3 + ic];
                gRS[ir * 3 + ic] = v;
            }
        for (int i = 0; i < 3; ++i)
            out_scale[pid * 3 + i] = gRS[0 * 3 + i] * R[0 * 3 + i] + gRS[1 * 3 + i] * R[1 * 3 + i] + gRS[2 * 3 + i] * R[2 * 3 + i];
        const double sx = S[0], sy = S[1], sz = S[2], qr = w, qi = x, qj = y, qk = z;
        const double c_qr[9] = {0, 2 * sy * qk, 2 * sz * qj, 2 * sx * qk, 0, 2 * sz * qi, 2 * sx * qj, 2 * sy * qi, 0};
        const double c_qi[9] = {0, 2 * sy * qj, 2 * sz * qk, 2 * sx * qj, 4 * sy * qi, 2 * sz * qr,
                                2 * sx * qk, 2 * sy * qr, 4 * sz * qi};
        const double c_qj[9] = {4 * sx * qj, 2 * sy * qi, 2 * sz * qr, 2 * sx * qi, 0, 2 * sz * qk,
                                2 * sx * qr, 2 * sy * qk, 4 * sz * qj};
        const double c_qk[9] = {4 * sx * qk, 2 * sy * qr, 2 * sz * qi, 2 * sx * qr, 4 * sy * qk, 2 * sz * qj,
                                2 * sx * qi, 2 * sy * qj, 0};
        double gq[4] = {0, 0, 0, 0};
        for (int i = 0; i < 9; ++i) {
            gq[0] += c_qr[i] * gRS[i];
            gq[1] += c_qi[i] * gRS[i];
            gq[2] += c_qj[i] * gRS[i];
            gq[3] += c_qk[i] * gRS[i];
        }
        for (int k = 0; k < 4; ++k) out_quat[pid * 4 + k] = gq[k];
    }
}

/* Canonical (tile, depth-bits, gaussian-index) pair list -- the integer-exact specification
 * of what splatter.py:567-613 intends (bin -> scan -> gather -> sort by (tile, depth)).
 * Pass 1 (out arrays NULL) returns M.  accum has T+1 entries. */
typedef struct {
    uint64_t key;
    uint32_t id;
} gso_pair_t;

static int pair_cmp(const void *pa, const void *pb) {
    const gso_pair_t *a = (const gso_pair_t *)pa, *b = (const gso_pair_t *)pb;
    if (a->key != b->key) return a->key < b->key ? -1 : 1;
    if (a->id != b->id) return a->id < b->id ? -1 : 1;
    return 0;
}

GSO_API int64_t gso_sorted_pairs(const float *pos, const float *cov, const int64_t *mask,
                                 int64_t n_point, float thresh, float tlx, float tly, int32_t ntx,
                                 int32_t nty, float leftmost, float topmost, int64_t capacity,
                                 uint64_t *keys_out, int32_t *ids_out, int32_t *accum_out) {
    int64_t n_tiles = (int64_t)ntx * nty, M = 0;
    gso_pair_t *pairs = NULL;
    if (keys_out || ids_out) pairs = (gso_pair_t *)malloc(sizeof(gso_pair_t) * (size_t)(capacity > 0 ? capacity : 1));
    for (int64_t pid = 0; pid < n_point; ++pid) {
        if (mask && mask[pid] == 0) continue;
        uint32_t rc[4];
        if (!gso_tile_rect(pos[pid * 3], pos[pid * 3 + 1], cov[pid * 4], cov[pid * 4 + 1],
                           cov[pid * 4 + 2], cov[pid * 4 + 3], thresh, tlx, tly, ntx, nty, leftmost,
                           topmost, rc))
            continue;
        uint32_t dbits;
        memcpy(&dbits, &pos[pid * 3 + 2], 4);
        for (uint32_t iy = rc[0]; iy < rc[1]; ++iy)
            for (uint32_t ix = rc[2]; ix < rc[3]; ++ix) {
                if (pairs && M < capacity) {
                    pairs[M].key = ((uint64_t)(ix + (uint64_t)iy * ntx) << 32) | dbits;
                    pairs[M].id = (uint32_t)pid;
                }
                ++M;
            }
    }
    if (pairs) {
        int64_t m = M < capacity ? M : capacity;
        qsort(pairs, (size_t)m, sizeof(gso_pair_t), pair_cmp);
        if (accum_out) memset(accum_out, 0, sizeof(int32_t) * (size_t)(n_tiles + 1));
        for (int64_t j = 0; j < m; ++j) {
            if (keys_out) keys_out[j] = pairs[j].key;
            if (ids_out) ids_out[j] = (int32_t)pairs[j].id;
            if (accum_out) accum_out[(pairs[j].key >> 32) + 1]++;
        }
        if (accum_out)
            for (int64_t t = 0; t < n_tiles; ++t) accum_out[t + 1] += accum_out[t];
        free(pairs);
    }
    return M;
}

/* ref_compat order (splatter.py:608-613): float32 composite key depth + tile*(max_depth+1),
 * stable ascending (torch.sort is unstable; stability is the oracle's tie-break). */
typedef struct {
    float key;
    int64_t idx;
} gso_fkey_t;
static int fkey_cmp(const void *pa, const void *pb) {
    const gso_fkey_t *a = (const gso_fkey_t *)pa, *b = (const gso_fkey_t *)pb;
    if (a->key != b->key) return a->key < b->key ? -1 : 1;
    return a->idx < b->idx ? -1 : (a->idx > b->idx);
}
GSO_API void gso_sort_float32_key(const float *depth, const int32_t *tile_ids, int64_t M,
                                  int64_t *perm_out) {
    if (M <= 0) return;
    float base = depth[0];
    for (int64_t j = 1; j < M; ++j) base = depth[j] > base ? depth[j] : base;
    gso_fkey_t *k = (gso_fkey_t *)malloc(sizeof(gso_fkey_t) * (size_t)M);
    for (int64_t j = 0; j < M; ++j) {
        k[j].key = depth[j] + (float)tile_ids[j] * (base + 1); /* splatter.py:611 */
        k[j].idx = j;
    }
    qsort(k, (size_t)M, sizeof(gso_fkey_t), fkey_cmp);
    for (int64_t j = 0; j < M; ++j) perm_out[j] = k[j].idx;
    free(k);
}

/* ---------------------------------------------------------------------------------------
 * SH basis                                     gaussian.cu:385-426
 * ------------------------------------------------------------------------------------- */
static const float C0 = 0.28209479177387814;
static const float C1 = 0.4886025119029199;
static const float C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
                            -1.0925484305920792, 0.5462742152960396};

/* Degree 3 is an EXTENSION (BASELINE config 4 names "SH degree 3"): the reference declares C3
 * (gaussian.cu:395-403) but calc_sh (:405-426) only has the basis_dim 9 and 4 cases.  The seven extra
 * functions below are the degree-3 band of the same real-SH convention (svox2 / PlenOctrees) using that
 * table; tests/test_oracle_kat.py pins them against scipy's spherical harmonics. */
static const float C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
                            -0.4570457994644658, 1.445305721320277, -0.5900435899266435};

/* `use_sh` as passed around below: 0 = rgb logits, 1 or 9 = the reference's 9 basis functions (its boolean
 * use_sh_coeff), 16 = the degree-3 extension.  Coefficient layout [channel][nb] (gaussian.cu:644, 942). */
static int sh_nb(int use_sh) { return use_sh == 0 ? 0 : (use_sh == 16 ? 16 : 9); }

static void calc_sh(int nb, const float *dir, float *out) {
    out[0] = C0;
    const float x = dir[0], y = dir[1], z = dir[2];
    const float xx = x * x, yy = y * y, zz = z * z;
    const float xy = x * y, yz = y * z, xz = x * z;
    out[4] = C2[0] * xy;
    out[5] = C2[1] * yz;
    out[6] = (float)(C2[2] * (2.0 * zz - xx - yy)); /* double literal, :417 */
    out[7] = C2[3] * xz;
    out[8] = C2[4] * (xx - yy);
    out[1] = -C1 * y;
    out[2] = C1 * z;
    out[3] = -C1 * x;
    if (nb == 16) {
        out[9] = C3[0] * y * (3.0f * xx - yy);
        out[10] = C3[1] * xy * z;
        out[11] = C3[2] * y * (4.0f * zz - xx - yy);
        out[12] = C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
        out[13] = C3[4] * x * (4.0f * zz - xx - yy);
        out[14] = C3[5] * z * (xx - yy);
        out[15] = C3[6] * x * (xx - 3.0f * yy);
    }
}

/* Conditioning of the basis values (test infrastructure, for the gradient tolerance's scale): every monomial of calc_sh with
 * its magnitude, every difference replaced by the sum of its operands' magnitudes.  b6 = C (2 zz - xx - yy), b8 = C (xx - yy)
 * and the degree-3 functions cancel internally: along |x| = |y| b8 is a few ulp of xx + yy with either sign, whatever
 * its own size (round 6: one SH-coefficient row element of 188 M, in a tile on that diagonal, sat at 2.4 x its tolerance
 * while the scale only carried |b8|). */
static void calc_sh_abs(int nb, const float *dir, double *out) {
    const double x = fabs((double)dir[0]), y = fabs((double)dir[1]), z = fabs((double)dir[2]);
    const double xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    out[0] = fabs((double)C0);
    out[1] = fabs((double)C1) * y;
    out[2] = fabs((double)C1) * z;
    out[3] = fabs((double)C1) * x;
    out[4] = fabs((double)C2[0]) * xy;
    out[5] = fabs((double)C2[1]) * yz;
    out[6] = fabs((double)C2[2]) * (2.0 * zz + xx + yy);
    out[7] = fabs((double)C2[3]) * xz;
    out[8] = fabs((double)C2[4]) * (xx + yy);
    if (nb == 16) {
        out[9] = fabs((double)C3[0]) * y * (3.0 * xx + yy);
        out[10] = fabs((double)C3[1]) * xy * z;
        out[11] = fabs((double)C3[2]) * y * (4.0 * zz + xx + yy);
        out[12] = fabs((double)C3[3]) * z * (2.0 * zz + 3.0 * xx + 3.0 * yy);
        out[13] = fabs((double)C3[4]) * x * (4.0 * zz + xx + yy);
        out[14] = fabs((double)C3[5]) * z * (xx + yy);
        out[15] = fabs((double)C3[6]) * x * (xx + 3.0 * yy);
    }
}

static void pixel_sh(int nb, uint32_t id_x, uint32_t id_y, const float *rays_o, const float *lefttop,
                     const float *vdx, const float *vdy, float *SH) {
    /* gaussian.cu:849-860 */
    float dir[3], nrm = 0.0f;
    for (int i = 0; i < 3; ++i) {
        dir[i] = lefttop[i] + id_x * vdx[i] + id_y * vdy[i] - rays_o[i];
        nrm += dir[i] * dir[i];
    }
    nrm = sqrtf(nrm);
    for (int i = 0; i < 3; ++i) dir[i] = (float)(dir[i] / (nrm + 1e-7));
    calc_sh(nb, dir, SH);
}

static void pixel_sh_abs(int nb, uint32_t id_x, uint32_t id_y, const float *rays_o, const float *lefttop,
                         const float *vdx, const float *vdy, double *SHabs) {
    float dir[3], nrm = 0.0f;
    for (int i = 0; i < 3; ++i) {
        dir[i] = lefttop[i] + id_x * vdx[i] + id_y * vdy[i] - rays_o[i];
        nrm += dir[i] * dir[i];
    }
    nrm = sqrtf(nrm);
    for (int i = 0; i < 3; ++i) dir[i] = (float)(dir[i] / (nrm + 1e-7));
    calc_sh_abs(nb, dir, SHabs);
}

GSO_API void gso_pixel_sh(uint32_t id_x, uint32_t id_y, const float *rays_o, const float *lefttop,
                          const float *vdx, const float *vdy, float *SH) {
    pixel_sh(9, id_x, id_y, rays_o, lefttop, vdx, vdy, SH);
}

GSO_API void gso_pixel_sh16(uint32_t id_x, uint32_t id_y, const float *rays_o, const float *lefttop,
                            const float *vdx, const float *vdy, float *SH) {
    pixel_sh(16, id_x, id_y, rays_o, lefttop, vdx, vdy, SH);
}

GSO_API void gso_calc_sh(int nb, const float *dir, float *out) { calc_sh(nb, dir, out); }

/* ---------------------------------------------------------------------------------------
 * K7: tile rasterizer forward                  gaussian.cu:806-970
 * pos[M,3] (z ignored), rgb[M,D] (D = 3 or 27), opa[M], cov[M,4], accum[T+1], res[h,w,3].
 * h, w are the PADDED sizes (multiples of 16).  fast!=0 -> __expf path (float exp).
 * ------------------------------------------------------------------------------------- */
GSO_API void gso_draw(const float *pos, const float *rgb, const float *opa, const float *cov,
                      const int32_t *accum_idx, float *res, int32_t h, int32_t w, float focal_x,
                      float focal_y, int weight_normalize, int sigmoid, int fast,
                      const float *rays_o, const float *lefttop, const float *vdx,
                      const float *vdy, int use_sh) {
    const uint32_t ntx = (uint32_t)(w + 15) / 16;
    const int nb = sh_nb(use_sh), D = use_sh ? 3 * nb : 3;
#pragma omp parallel for schedule(dynamic, 4)
    for (uint32_t id_y = 0; id_y < (uint32_t)h; ++id_y)
        for (uint32_t id_x = 0; id_x < (uint32_t)w; ++id_x) {
            uint32_t id_tile = id_x / 16 + (id_y / 16) * ntx; /* :832 */
            uint32_t start = (uint32_t)accum_idx[id_tile], end = (uint32_t)accum_idx[id_tile + 1];
            float pixel_x = (float)((id_x + 0.5 - (uint32_t)w / 2) / focal_x); /* :839-840 */
            float pixel_y = (float)((id_y + 0.5 - (uint32_t)h / 2) / focal_y);
            float color[3] = {0, 0, 0}, accum = 1.0f, accum_weight = 0.0f, SH[16];
            if (use_sh) pixel_sh(nb, id_x, id_y, rays_o, lefttop, vdx, vdy, SH);
            for (uint32_t g = start; g < end; ++g) {
                if (accum < 0.0001) break; /* :906, float promoted to double */
                float a = cov[g * 4], b = cov[g * 4 + 1], c = cov[g * 4 + 2], d = cov[g * 4 + 3];
                float x = pixel_x - pos[g * 3], y = pixel_y - pos[g * 3 + 1];
                float det = (a * d - b * c);
                /* :918 -- note operator precedence: 1.0/2*3.14.. == pi/2 */
                float prob = sigmoid ? (float)(1.0 / 2 * 3.1415926536 * (1.0f / sqrtf((float)(det + 1e-7)))) : 1;
                double q = -(d * x * x - (b + c) * x * y + a * y * y) / (2 * det + 1e-14);
                if (fast)
                    prob *= expf((float)q); /* :920 __expf(float) */
                else
                    prob = (float)(prob * exp(q)); /* :923 exp(double) */
                float alpha = prob * opa[g];
                if (sigmoid) alpha = (float)(2. / (exp(-alpha) + 1) - 1); /* :930 */
                float weight = alpha * accum;
                if (use_sh) { /* :936-952 */
                    for (int ch = 0; ch < 3; ++ch) {
                        float v = 0.0f;
                        for (int s = 0; s < nb; ++s) v += SH[s] * rgb[(size_t)g * D + ch * nb + s];
                        v = (float)(1. / (1 + expf(-v)));
                        color[ch] += v * weight;
                    }
                } else {
                    color[0] += rgb[(size_t)g * 3 + 0] * weight;
                    color[1] += rgb[(size_t)g * 3 + 1] * weight;
                    color[2] += rgb[(size_t)g * 3 + 2] * weight;
                }
                accum_weight += weight;
                accum *= (1 - alpha);
            }
            (void)D;
            if (accum_weight < 0.01 || !weight_normalize) accum_weight = 1; /* :964 */
            float *o = res + ((size_t)id_x + (size_t)id_y * w) * 3;
            o[0] = color[0] / accum_weight;
            o[1] = color[1] / accum_weight;
            o[2] = color[2] / accum_weight;
        }
}

/* ---------------------------------------------------------------------------------------
 * Pixels whose early-stop decision is not robust in fp32 (test infrastructure for the gradient parity tests).
 * K7 / K8 stop a pixel at the first Gaussian that finds its transmittance below 1e-4 (gaussian.cu:906, :578).
 * The transmittance is a product of up to hundreds of rounded factors (1 - alpha): two correct fp32 evaluations
 * (other expression order, 1-ulp exp) agree on it to ~1e-5 relative at best -- worse behind nearly opaque Gaussians,
 * see `eps` below -- so when it passes within that uncertainty of the threshold, stopping one Gaussian earlier or
 * later is a legitimate outcome, and changes WHICH terms exist for that pixel.  amb[h,w] <- 1 for those pixels; the gradient parity tests feed a dL/dimage that is
 * zero there, so that every remaining term is comparable.  Same loop as gso_draw (fast exp path), alpha only.
 * ------------------------------------------------------------------------------------- */
GSO_API void gso_draw_ambiguous(const float *pos, const float *opa, const float *cov, const int32_t *accum_idx,
                                uint8_t *amb, int32_t h, int32_t w, float focal_x, float focal_y, float band) {
    const uint32_t ntx = (uint32_t)(w + 15) / 16;
#pragma omp parallel for schedule(dynamic, 4)
    for (uint32_t id_y = 0; id_y < (uint32_t)h; ++id_y)
        for (uint32_t id_x = 0; id_x < (uint32_t)w; ++id_x) {
            uint32_t id_tile = id_x / 16 + (id_y / 16) * ntx;
            uint32_t start = (uint32_t)accum_idx[id_tile], end = (uint32_t)accum_idx[id_tile + 1];
            float pixel_x = (float)((id_x + 0.5 - (uint32_t)w / 2) / focal_x);
            float pixel_y = (float)((id_y + 0.5 - (uint32_t)h / 2) / focal_y);
            float accum = 1.0f;
            /* relative uncertainty of `accum`: `band` + what the factors (1 - alpha) contribute -- a relative error
             * e of alpha (exponent rounding: a few 1e-7) is an error e alpha / (1 - alpha) of the factor, which is
             * large exactly where it matters, behind nearly opaque Gaussians */
            double eps = band;
            uint8_t flag = 0;
            for (uint32_t g = start; g < end; ++g) {
                const double lo = 0.0001 * (1.0 - eps), hi = 0.0001 * (1.0 + eps);
                if (accum >= lo && accum <= hi) flag = 1; /* this test could go either way */
                if (accum < lo) break;                    /* well below: every evaluation has stopped */
                float a = cov[g * 4], b = cov[g * 4 + 1], c = cov[g * 4 + 2], d = cov[g * 4 + 3];
                float x = pixel_x - pos[g * 3], y = pixel_y - pos[g * 3 + 1];
                float det = (a * d - b * c);
                double q = -(d * x * x - (b + c) * x * y + a * y * y) / (2 * det + 1e-14);
                float alpha = expf((float)q) * opa[g];
                eps += 2e-6 * fabs((double)alpha) / fabs(1.0 - (double)alpha + 1e-7) + 2e-7;
                if (!(eps < 0.5)) eps = 0.5;
                accum *= (1 - alpha);
            }
            amb[(size_t)id_x + (size_t)id_y * w] = flag;
        }
}

/* ---------------------------------------------------------------------------------------
 * K8: tile rasterizer backward                 gaussian.cu:440-803
 * Intended semantics: each (tile, Gaussian) row = sum over the tile's 256 pixels of the
 * per-pixel contribution, for pixels whose transmittance is still >= 1e-4 before that
 * Gaussian.  Per-pixel terms are evaluated in the reference's fp32 expression order; the
 * 256-term sums are accumulated in double and rounded once (the reference's shuffle/atomic
 * order is not defined).  grad_pos[:,2] is never written (stays the caller's zero).
 *
 * With non-NULL `cs_*` pointers the same pass also returns, per row element, its CONDITIONING
 * SCALE: sum over the tile's pixels of |term| (1 + cs_w_exp x the cancellation inside exp's argument)
 * + cs_w x (the term with every other internal difference replaced by the sum of its operands' magnitudes).  Two fp32 evaluations of the same
 * formulas in different orders (and with 1-ulp exp / rcp) agree to (a modest number of ulp) x
 * that scale, however small the signed sum turns out; the element-wise gradient tolerance of
 * the parity tests is stated in these units (tests/gs_testutil.py).
 * ------------------------------------------------------------------------------------- */
static void draw_backward_impl(const float *pos, const float *rgb, const float *opa,
                               const float *cov, const int32_t *accum_idx, const float *output,
                               const float *grad_output, float *grad_pos, float *grad_rgb,
                               float *grad_opa, float *grad_cov, int32_t h, int32_t w,
                               float focal_x, float focal_y, int sigmoid, int fast,
                               const float *rays_o, const float *lefttop, const float *vdx,
                               const float *vdy, int use_sh, float *cs_pos, float *cs_rgb,
                               float *cs_opa, float *cs_cov, double cs_w, double cs_w_exp) {
    const uint32_t ntx = (uint32_t)(w + 15) / 16, nty = (uint32_t)(h + 15) / 16;
    const int nb = sh_nb(use_sh), D = use_sh ? 3 * nb : 3;
    const int NV = 2 + D + 1 + 4;
    const int want_cs = cs_pos != NULL;
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t id_tile = 0; id_tile < ntx * nty; ++id_tile) {
            const uint32_t tx = id_tile % ntx, ty = id_tile / ntx;
            uint32_t start = (uint32_t)accum_idx[id_tile], end = (uint32_t)accum_idx[id_tile + 1];
            uint32_t len = end - start;
            if (len == 0) continue;
            double *acc = (double *)calloc((size_t)len * NV, sizeof(double));
            double *cs = want_cs ? (double *)calloc((size_t)len * NV, sizeof(double)) : NULL;
            for (uint32_t ly = 0; ly < 16; ++ly)
                for (uint32_t lx = 0; lx < 16; ++lx) {
                    uint32_t id_x = tx * 16 + lx, id_y = ty * 16 + ly;
                    if (id_x >= (uint32_t)w || id_y >= (uint32_t)h) continue;
                    float pixel_x = (float)((id_x + 0.5 - (uint32_t)w / 2) / focal_x);
                    float pixel_y = (float)((id_y + 0.5 - (uint32_t)h / 2) / focal_y);
                    float SH[16];
                    double SHabs[16];
                    if (use_sh) pixel_sh(nb, id_x, id_y, rays_o, lefttop, vdx, vdy, SH);
                    if (use_sh && want_cs) pixel_sh_abs(nb, id_x, id_y, rays_o, lefttop, vdx, vdy, SHabs);
                    const float *go = grad_output + ((size_t)id_x + (size_t)id_y * w) * 3;
                    const float *co = output + ((size_t)id_x + (size_t)id_y * w) * 3;
                    float color[3] = {0, 0, 0}, accum = 1.0f;
                    double ampT = 0; /* conditioning of the transmittance: sum of alpha / (1 - alpha) so far */
                    for (uint32_t i = 0; i < len; ++i) {
                        uint32_t g = start + i;
                        if (accum < 0.0001) break; /* :578 */
                        float _a = cov[g * 4], _b = cov[g * 4 + 1], _c = cov[g * 4 + 2],
                              _d = cov[g * 4 + 3];
                        float _x = pixel_x - pos[g * 3], _y = pixel_y - pos[g * 3 + 1];
                        float det = (_a * _d - _b * _c);
                        float Pm = -(_d * _x * _x - (_b + _c) * _x * _y + _a * _y * _y); /* :590 */
                        float Pn = (float)(2 * det + 1e-14);                              /* :591 */
                        float p_c0 = sigmoid ? (float)(1.0 / 2 * 3.1415926536) : 1.0f;    /* :593 */
                        float p0 = sigmoid ? p_c0 * (1.0f / sqrtf((float)(det + 1e-7))) : 1.0f;   /* :594 */
                        float p1 = fast ? expf(Pm / Pn) : (float)exp(Pm / Pn);            /* :595-600 */
                        float prob = p0 * p1;
                        float alpha = prob * opa[g];
                        if (sigmoid) alpha = (float)(2. / (exp(-alpha) + 1) - 1);
                        /* :610-634 */
                        float dPm_da = -(_y * _y), dPm_db = _x * _y, dPm_dc = _x * _y,
                              dPm_dd = -(_x * _x);
                        float dPn_da = 2 * _d, dPn_db = -2 * _c, dPn_dc = -2 * _b, dPn_dd = 2 * _a;
                        float dP1_da = p1 * (dPm_da * Pn - dPn_da * Pm) / (Pn * Pn);
                        float dP1_db = p1 * (dPm_db * Pn - dPn_db * Pm) / (Pn * Pn);
                        float dP1_dc = p1 * (dPm_dc * Pn - dPn_dc * Pm) / (Pn * Pn);
                        float dP1_dd = p1 * (dPm_dd * Pn - dPn_dd * Pm) / (Pn * Pn);
                        float k0 = sigmoid ? (float)(0.5 * (p0 * p0 * p0) / (p_c0 * p_c0)) : 0.0f;
                        float dP0_da = -k0 * _d, dP0_db = k0 * _c, dP0_dc = k0 * _b, dP0_dd = -k0 * _a;
                        float dP_da = p0 * dP1_da + p1 * dP0_da;
                        float dP_db = p0 * dP1_db + p1 * dP0_db;
                        float dP_dc = p0 * dP1_dc + p1 * dP0_dc;
                        float dP_dd = p0 * dP1_dd + p1 * dP0_dd;
                        float dP_dx = prob / Pn * (2 * _d * _x - _b * _y - _c * _y);
                        float dP_dy = prob / Pn * (2 * _a * _y - _b * _x - _c * _x);
                        float weight = alpha * accum;
                        float cpc[3] = {0, 0, 0};
                        if (use_sh) { /* :639-652 */
                            for (int ch = 0; ch < 3; ++ch) {
                                for (int s = 0; s < nb; ++s)
                                    cpc[ch] += SH[s] * rgb[(size_t)g * D + ch * nb + s];
                                cpc[ch] = (float)(1. / (1 + expf(-cpc[ch])));
                            }
                        } else {
                            cpc[0] = rgb[(size_t)g * 3];
                            cpc[1] = rgb[(size_t)g * 3 + 1];
                            cpc[2] = rgb[(size_t)g * 3 + 2];
                        }
                        color[0] += cpc[0] * weight;
                        color[1] += cpc[1] * weight;
                        color[2] += cpc[2] * weight;
                        double *row = acc + (size_t)i * NV;
                        if (use_sh) { /* :665-688 */
                            for (int ch = 0; ch < 3; ++ch) {
                                float Dk = go[ch] * weight * (cpc[ch] * (1 - cpc[ch]));
                                for (int s = 0; s < nb; ++s) row[2 + ch * nb + s] += (double)(Dk * SH[s]);
                            }
                        } else { /* :691-706 */
                            row[2 + 0] += (double)(go[0] * weight);
                            row[2 + 1] += (double)(go[1] * weight);
                            row[2 + 2] += (double)(go[2] * weight);
                        }
                        /* :710-726 */
                        float d_alpha = 0;
                        for (int m = 0; m < 3; ++m) d_alpha += go[m] * cpc[m];
                        d_alpha *= accum;
                        float dacc = 0;
                        for (int m = 0; m < 3; ++m) dacc += go[m] * (co[m] - color[m]);
                        dacc = (float)(dacc / (1 - alpha + 1e-7));
                        d_alpha -= dacc;
                        float dsq = 1.0f;
                        if (sigmoid) {
                            dsq = (float)(alpha + 1 - 0.5 * (alpha + 1) * (alpha + 1));
                            d_alpha = (float)(d_alpha * (alpha + 1 - 0.5 * (alpha + 1) * (alpha + 1)));
                        }
                        row[2 + D] += (double)(float)(d_alpha * prob); /* :729 */
                        float d_prob = d_alpha * opa[g];                /* :740 */
                        row[0] += (double)(float)(d_prob * dP_dx);
                        row[1] += (double)(float)(d_prob * dP_dy);
                        row[2 + D + 1] += (double)(float)(d_prob * dP_da);
                        row[2 + D + 2] += (double)(float)(d_prob * dP_db);
                        row[2 + D + 3] += (double)(float)(d_prob * dP_dc);
                        row[2 + D + 4] += (double)(float)(d_prob * dP_dd);
                        if (want_cs) {
                            /* scale = |term| + cs_w x (the same term with every difference replaced by the sum of
                             * the magnitudes of its operands): the first part carries the errors that are relative
                             * to the term (exp, rcp, products, the running T), the second the cancellations inside it
                             * -- exp's argument d x^2 - (b+c) x y + a y^2, g.(C_final - C_run), dPm Pn - dPn Pm,
                             * 2 d x - b y - c y --, each worth a few ulp of its operands. */
                            double *cr = cs + (size_t)i * NV;
                            const double W = cs_w;
                            const double PmAbs = fabs((double)_d * _x * _x) + fabs((double)(_b + _c) * _x * _y) +
                                                 fabs((double)_a * _y * _y);
                            const double pn = fabs((double)Pn);
                            /* every term carries alpha (exponent rounding: relative error ~ PmAbs / Pn ulp) and the
                             * transmittance, a product of factors 1 - alpha_j whose relative error is that of alpha_j
                             * times alpha_j / (1 - alpha_j): behind a nearly opaque Gaussian it is far from 1 ulp */
                            ampT += fabs((double)alpha) / fabs(1.0 - (double)alpha + 1e-7);
                            const double rel = 1.0 + cs_w_exp * PmAbs / pn + W * ampT;
                            double gcabs = 0, dcabs = 0;
                            for (int m = 0; m < 3; ++m) {
                                gcabs += fabs((double)go[m] * cpc[m]);
                                dcabs += fabs((double)go[m]) * (fabs((double)co[m]) + fabs((double)color[m]));
                            }
                            const double da1 = fabs((double)d_alpha);
                            const double da2 = ((double)accum * gcabs + dcabs / fabs(1 - (double)alpha + 1e-7)) *
                                               fabs((double)dsq);
                            if (use_sh) {
                                for (int ch = 0; ch < 3; ++ch) {
                                    /* c (1 - c): for a saturated colour (c -> 1) the difference 1 - c only has the
                                     * absolute accuracy of c itself, a few ulp of 1 */
                                    double Dk = fabs((double)go[ch] * weight * (cpc[ch] * (1 - cpc[ch]))) * rel +
                                                W * fabs((double)go[ch] * weight) * fabs((double)cpc[ch]) *
                                                    (1.0 + fabs((double)cpc[ch]));
                                    /* ... times the basis value, whose own internal differences count like every other */
                                    for (int s = 0; s < nb; ++s)
                                        cr[2 + ch * nb + s] += Dk * (fabs((double)SH[s]) + W * SHabs[s]);
                                }
                            } else {
                                for (int m = 0; m < 3; ++m) cr[2 + m] += fabs((double)go[m] * weight) * rel;
                            }
                            const double po = fabs((double)opa[g]), pq = fabs((double)prob) / pn;
                            cr[2 + D] += (da1 * rel + W * da2) * fabs((double)prob);
                            const double ex = fabs(2.0 * _d * _x) + fabs((double)_b * _y) + fabs((double)_c * _y);
                            const double ey = fabs(2.0 * _a * _y) + fabs((double)_b * _x) + fabs((double)_c * _x);
                            cr[0] += da1 * po * fabs((double)dP_dx) * rel + W * da2 * po * pq * ex;
                            cr[1] += da1 * po * fabs((double)dP_dy) * rel + W * da2 * po * pq * ey;
                            const double p1n = fabs((double)p0 * p1) / (pn * pn), p10 = fabs((double)p1);
                            const double ea = p1n * (fabs((double)dPm_da) * pn + fabs((double)dPn_da) * PmAbs) + p10 * fabs((double)dP0_da);
                            const double eb = p1n * (fabs((double)dPm_db) * pn + fabs((double)dPn_db) * PmAbs) + p10 * fabs((double)dP0_db);
                            const double ec = p1n * (fabs((double)dPm_dc) * pn + fabs((double)dPn_dc) * PmAbs) + p10 * fabs((double)dP0_dc);
                            const double ed = p1n * (fabs((double)dPm_dd) * pn + fabs((double)dPn_dd) * PmAbs) + p10 * fabs((double)dP0_dd);
                            cr[2 + D + 1] += da1 * po * fabs((double)dP_da) * rel + W * da2 * po * ea;
                            cr[2 + D + 2] += da1 * po * fabs((double)dP_db) * rel + W * da2 * po * eb;
                            cr[2 + D + 3] += da1 * po * fabs((double)dP_dc) * rel + W * da2 * po * ec;
                            cr[2 + D + 4] += da1 * po * fabs((double)dP_dd) * rel + W * da2 * po * ed;
                        }
                        accum *= (1 - alpha); /* :774 */
                    }
                }
            for (uint32_t i = 0; i < len; ++i) { /* :779-799 */
                uint32_t g = start + i;
                const double *row = acc + (size_t)i * NV;
                grad_pos[(size_t)g * 3 + 0] = (float)row[0];
                grad_pos[(size_t)g * 3 + 1] = (float)row[1];
                for (int k = 0; k < D; ++k) grad_rgb[(size_t)g * D + k] = (float)row[2 + k];
                grad_opa[g] = (float)row[2 + D];
                for (int k = 0; k < 4; ++k) grad_cov[(size_t)g * 4 + k] = (float)row[2 + D + 1 + k];
                if (want_cs) {
                    const double *cr = cs + (size_t)i * NV;
                    cs_pos[(size_t)g * 3 + 0] = (float)cr[0];
                    cs_pos[(size_t)g * 3 + 1] = (float)cr[1];
                    for (int k = 0; k < D; ++k) cs_rgb[(size_t)g * D + k] = (float)cr[2 + k];
                    cs_opa[g] = (float)cr[2 + D];
                    for (int k = 0; k < 4; ++k) cs_cov[(size_t)g * 4 + k] = (float)cr[2 + D + 1 + k];
                }
            }
            free(acc);
            free(cs);
    }
}

GSO_API void gso_draw_backward(const float *pos, const float *rgb, const float *opa,
                               const float *cov, const int32_t *accum_idx, const float *output,
                               const float *grad_output, float *grad_pos, float *grad_rgb,
                               float *grad_opa, float *grad_cov, int32_t h, int32_t w,
                               float focal_x, float focal_y, int weight_normalize, int sigmoid,
                               int fast, const float *rays_o, const float *lefttop,
                               const float *vdx, const float *vdy, int use_sh) {
    (void)weight_normalize; /* the reference backward ignores it too */
    draw_backward_impl(pos, rgb, opa, cov, accum_idx, output, grad_output, grad_pos, grad_rgb, grad_opa, grad_cov,
                       h, w, focal_x, focal_y, sigmoid, fast, rays_o, lefttop, vdx, vdy, use_sh, NULL, NULL, NULL,
                       NULL, 0.0, 0.0);
}

/* the same rows + the conditioning scale of every row element (see above); cs_* are laid out like grad_* */
GSO_API void gso_draw_backward_scaled(const float *pos, const float *rgb, const float *opa,
                                      const float *cov, const int32_t *accum_idx, const float *output,
                                      const float *grad_output, float *grad_pos, float *grad_rgb,
                                      float *grad_opa, float *grad_cov, int32_t h, int32_t w,
                                      float focal_x, float focal_y, int sigmoid, int fast,
                                      const float *rays_o, const float *lefttop, const float *vdx,
                                      const float *vdy, int use_sh, float *cs_pos, float *cs_rgb,
                                      float *cs_opa, float *cs_cov, double cs_w, double cs_w_exp) {
    draw_backward_impl(pos, rgb, opa, cov, accum_idx, output, grad_output, grad_pos, grad_rgb, grad_opa, grad_cov,
                       h, w, focal_x, focal_y, sigmoid, fast, rays_o, lefttop, vdx, vdy, use_sh, cs_pos, cs_rgb,
                       cs_opa, cs_cov, cs_w, cs_w_exp);
}

/* ---------------------------------------------------------------------------------------
 * K8 in DOUBLE: the yardstick the fp32 evaluations are measured against (tests/test_grad_calibration*.py,
 * tools/grad_calibration.py).  The same formulas as draw_backward_impl above (gaussian.cu:574-775; `fast` flavour,
 * sigmoid = 0), with every operation carried out in double on the fp32 inputs -- pixel coordinates, ray direction
 * and SH basis, exp, the colour sigmoid, the running and the FINAL colour (recomputed here in double rather than
 * taken from the fp32 forward), all sums.  The reference's constants 1e-14 (:591) and 1e-7 (:720) are part of the
 * formula and stay.  The only fp32 quantity is the early-stop DECISION: a pixel stops where the fp32 transmittance
 * chain of K7 / K8 (:906, :578) finds T < 1e-4, so that truth, oracle and kernels sum the same set of terms (the
 * parity tests zero dL/dimage on the pixels where that decision is not robust, gso_draw_ambiguous).
 * Outputs are double arrays laid out like the gradients: gp[M,3] (z untouched), gr[M,D], go[M], gc[M,4].
 * ------------------------------------------------------------------------------------- */
static void calc_sh_f64(int nb, const double *dir, double *out) {
    const double x = dir[0], y = dir[1], z = dir[2];
    const double xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    out[0] = (double)C0;
    out[1] = -(double)C1 * y;
    out[2] = (double)C1 * z;
    out[3] = -(double)C1 * x;
    out[4] = (double)C2[0] * xy;
    out[5] = (double)C2[1] * yz;
    out[6] = (double)C2[2] * (2.0 * zz - xx - yy);
    out[7] = (double)C2[3] * xz;
    out[8] = (double)C2[4] * (xx - yy);
    if (nb == 16) {
        out[9] = (double)C3[0] * y * (3.0 * xx - yy);
        out[10] = (double)C3[1] * xy * z;
        out[11] = (double)C3[2] * y * (4.0 * zz - xx - yy);
        out[12] = (double)C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy);
        out[13] = (double)C3[4] * x * (4.0 * zz - xx - yy);
        out[14] = (double)C3[5] * z * (xx - yy);
        out[15] = (double)C3[6] * x * (xx - 3.0 * yy);
    }
}

GSO_API void gso_draw_backward_f64(const float *pos, const float *rgb, const float *opa, const float *cov,
                                   const int32_t *accum_idx, const float *grad_output, double *gp, double *gr,
                                   double *go_, double *gc, int32_t h, int32_t w, float focal_x, float focal_y,
                                   const float *rays_o, const float *lefttop, const float *vdx, const float *vdy,
                                   int use_sh) {
    const uint32_t ntx = (uint32_t)(w + 15) / 16, nty = (uint32_t)(h + 15) / 16;
    const int nb = sh_nb(use_sh), D = use_sh ? 3 * nb : 3;
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t id_tile = 0; id_tile < ntx * nty; ++id_tile) {
        const uint32_t tx = id_tile % ntx, ty = id_tile / ntx;
        const uint32_t start = (uint32_t)accum_idx[id_tile], end = (uint32_t)accum_idx[id_tile + 1];
        const uint32_t len = end - start;
        if (len == 0) continue;
        double *cpc = (double *)malloc(sizeof(double) * 3 * (size_t)len); /* colours of this pixel's Gaussians */
        for (uint32_t ly = 0; ly < 16; ++ly)
            for (uint32_t lx = 0; lx < 16; ++lx) {
                const uint32_t id_x = tx * 16 + lx, id_y = ty * 16 + ly;
                if (id_x >= (uint32_t)w || id_y >= (uint32_t)h) continue;
                const double px = (id_x + 0.5 - (uint32_t)w / 2) / (double)focal_x;
                const double py = (id_y + 0.5 - (uint32_t)h / 2) / (double)focal_y;
                const float pxf = (float)px, pyf = (float)py; /* the fp32 chain that takes the stop decision */
                double SH[16];
                if (use_sh) {
                    double dir[3], nrm = 0.0;
                    for (int i = 0; i < 3; ++i) {
                        dir[i] = (double)lefttop[i] + id_x * (double)vdx[i] + id_y * (double)vdy[i] - (double)rays_o[i];
                        nrm += dir[i] * dir[i];
                    }
                    nrm = sqrt(nrm);
                    for (int i = 0; i < 3; ++i) dir[i] = dir[i] / (nrm + 1e-7);
                    calc_sh_f64(nb, dir, SH);
                }
                const float *g3 = grad_output + ((size_t)id_x + (size_t)id_y * w) * 3;
                const double gO[3] = {g3[0], g3[1], g3[2]};
                /* pass 1: how many Gaussians this pixel composites (fp32 decision) and its final colour in double */
                uint32_t n_live = 0;
                double Cf[3] = {0, 0, 0}, T = 1.0;
                float accum = 1.0f;
                for (uint32_t i = 0; i < len; ++i) {
                    const uint32_t g = start + i;
                    if (accum < 0.0001) break;
                    const float af = cov[g * 4], bf = cov[g * 4 + 1], cf = cov[g * 4 + 2], df = cov[g * 4 + 3];
                    { /* fp32 alpha exactly as K7 / K8's fast path (gso_draw above) */
                        const float x = pxf - pos[g * 3], y = pyf - pos[g * 3 + 1];
                        const float det = af * df - bf * cf;
                        const double q = -(df * x * x - (bf + cf) * x * y + af * y * y) / (2 * det + 1e-14);
                        const float alpha = expf((float)q) * opa[g];
                        accum *= (1 - alpha);
                    }
                    const double a = af, b = bf, c = cf, d = df;
                    const double x = px - (double)pos[g * 3], y = py - (double)pos[g * 3 + 1];
                    const double det = a * d - b * c;
                    const double alpha = exp(-(d * x * x - (b + c) * x * y + a * y * y) / (2 * det + 1e-14)) *
                                         (double)opa[g];
                    for (int ch = 0; ch < 3; ++ch) {
                        double v;
                        if (use_sh) {
                            v = 0.0;
                            for (int s = 0; s < nb; ++s) v += SH[s] * (double)rgb[(size_t)g * D + ch * nb + s];
                            v = 1.0 / (1.0 + exp(-v));
                        } else {
                            v = (double)rgb[(size_t)g * 3 + ch];
                        }
                        cpc[3 * (size_t)i + ch] = v;
                        Cf[ch] += v * alpha * T;
                    }
                    T *= (1.0 - alpha);
                    n_live = i + 1;
                }
                /* pass 2: the terms */
                double Crun[3] = {0, 0, 0};
                T = 1.0;
                for (uint32_t i = 0; i < n_live; ++i) {
                    const uint32_t g = start + i;
                    const double a = cov[g * 4], b = cov[g * 4 + 1], c = cov[g * 4 + 2], d = cov[g * 4 + 3];
                    const double x = px - (double)pos[g * 3], y = py - (double)pos[g * 3 + 1];
                    const double det = a * d - b * c;
                    const double Pm = -(d * x * x - (b + c) * x * y + a * y * y), Pn = 2 * det + 1e-14;
                    const double p1 = exp(Pm / Pn);
                    const double o = (double)opa[g], alpha = p1 * o, weight = alpha * T;
                    const double *cc = cpc + 3 * (size_t)i;
                    double d_alpha = 0, dacc = 0;
                    for (int m = 0; m < 3; ++m) {
                        Crun[m] += cc[m] * weight;
                        d_alpha += gO[m] * cc[m];
                    }
                    d_alpha *= T;
                    for (int m = 0; m < 3; ++m) dacc += gO[m] * (Cf[m] - Crun[m]);
                    d_alpha -= dacc / (1 - alpha + 1e-7);
                    if (use_sh) {
                        for (int ch = 0; ch < 3; ++ch) {
                            const double Dk = gO[ch] * weight * (cc[ch] * (1 - cc[ch]));
                            for (int s = 0; s < nb; ++s) gr[(size_t)g * D + ch * nb + s] += Dk * SH[s];
                        }
                    } else {
                        for (int m = 0; m < 3; ++m) gr[(size_t)g * 3 + m] += gO[m] * weight;
                    }
                    go_[g] += d_alpha * p1;
                    const double d_prob = d_alpha * o;
                    gp[(size_t)g * 3 + 0] += d_prob * p1 / Pn * (2 * d * x - b * y - c * y);
                    gp[(size_t)g * 3 + 1] += d_prob * p1 / Pn * (2 * a * y - b * x - c * x);
                    const double k = d_prob * p1 / (Pn * Pn);
                    gc[(size_t)g * 4 + 0] += k * (-(y * y) * Pn - 2 * d * Pm);
                    gc[(size_t)g * 4 + 1] += k * ((x * y) * Pn + 2 * c * Pm);
                    gc[(size_t)g * 4 + 2] += k * ((x * y) * Pn + 2 * b * Pm);
                    gc[(size_t)g * 4 + 3] += k * (-(x * x) * Pn - 2 * a * Pm);
                    T *= (1.0 - alpha);
                }
            }
        free(cpc);
    }
}

/* ---------------------------------------------------------------------------------------
 * Whole forward frame on raw parameters (splatter.py:513-655 with the train.py defaults:
 * cudaculling=1, scale_activation="abs", tile_culling_method="prob2", fast_drawing=1),
 * canonical order, no MAXP cap.  Used as bench.py's cpu_baseline ("port") and by smoke().
 * image_out is [H, W, 3] (clamped and cropped).  Returns M; *V_out = visible count.
 * ------------------------------------------------------------------------------------- */
GSO_API int64_t gso_render_forward(const float *pos, const float *quat_raw, const float *scale_raw,
                                   const float *opa_raw, const float *rgb_raw, int64_t n,
                                   int use_sh, const float *rot, const float *tran, float near_,
                                   int32_t W, int32_t H, float fx, float fy, float thresh,
                                   const float *rays_o, const float *lefttop, const float *vdx,
                                   const float *vdy, float *image_out, int64_t *V_out) {
    const int D = use_sh ? 3 * sh_nb(use_sh) : 3;
    int32_t padW = ((W + 15) / 16) * 16, padH = ((H + 15) / 16) * 16; /* splatter.py:259-260 */
    int32_t ntx = padW / 16, nty = padH / 16;
    float tlx = (float)(16 / (double)fx), tly = (float)(16 / (double)fy); /* :279-280 */
    float leftmost = (float)(-padW / 2.0 / fx), topmost = (float)(-padH / 2.0 / fy);
    float half_w = (float)(W * 1.2 / 2 / fx), half_h = (float)(H * 1.2 / 2 / fy); /* :532-533 */
    float *nq = (float *)malloc(sizeof(float) * 4 * (size_t)n);
    float *ns = (float *)malloc(sizeof(float) * 3 * (size_t)n);
    for (int64_t i = 0; i < n; ++i) { /* splatter.py:519-521 */
        const float *q = quat_raw + 4 * i;
        float nr = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        for (int k = 0; k < 4; ++k) nq[4 * i + k] = q[k] / nr;
        for (int k = 0; k < 3; ++k) ns[3 * i + k] = fabsf(scale_raw[3 * i + k]) + 1e-4f;
    }
    float *rp = (float *)calloc((size_t)n * 3, sizeof(float));
    float *rc = (float *)calloc((size_t)n * 4, sizeof(float));
    int64_t *mask = (int64_t *)calloc((size_t)n, sizeof(int64_t));
    gso_global_culling(pos, nq, ns, rot, tran, n, near_, half_w, half_h, rp, rc, mask);
    int64_t V = 0;
    for (int64_t i = 0; i < n; ++i) V += mask[i];
    if (V_out) *V_out = V;
    int64_t M = gso_sorted_pairs(rp, rc, mask, n, thresh, tlx, tly, ntx, nty, leftmost, topmost, 0,
                                 NULL, NULL, NULL);
    int32_t *ids = (int32_t *)malloc(sizeof(int32_t) * (size_t)(M > 0 ? M : 1));
    int32_t *accum = (int32_t *)calloc((size_t)ntx * nty + 1, sizeof(int32_t));
    uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(M > 0 ? M : 1));
    gso_sorted_pairs(rp, rc, mask, n, thresh, tlx, tly, ntx, nty, leftmost, topmost, M, keys, ids, accum);
    /* gather #1/#2 + activations (splatter.py:536-541, 604, 613) */
    float *spos = (float *)malloc(sizeof(float) * 3 * (size_t)(M > 0 ? M : 1));
    float *scov = (float *)malloc(sizeof(float) * 4 * (size_t)(M > 0 ? M : 1));
    float *sopa = (float *)malloc(sizeof(float) * (size_t)(M > 0 ? M : 1));
    float *srgb = (float *)malloc(sizeof(float) * (size_t)D * (size_t)(M > 0 ? M : 1));
    for (int64_t j = 0; j < M; ++j) {
        int64_t g = ids[j];
        memcpy(spos + 3 * j, rp + 3 * g, 12);
        memcpy(scov + 4 * j, rc + 4 * g, 16);
        sopa[j] = 1.0f / (1.0f + expf(-opa_raw[g]));
        for (int k = 0; k < D; ++k)
            srgb[(size_t)j * D + k] = use_sh ? rgb_raw[(size_t)g * D + k]
                                             : 1.0f / (1.0f + expf(-rgb_raw[(size_t)g * D + k]));
    }
    float *padded = (float *)calloc((size_t)padH * padW * 3, sizeof(float));
    gso_draw(spos, srgb, sopa, scov, accum, padded, padH, padW, fx, fy, 0, 0, 1, rays_o, lefttop,
             vdx, vdy, use_sh);
    int32_t top = (padH - H) / 2, left = (padW - W) / 2; /* splatter.py:267-272, 652-653 */
    for (int32_t y = 0; y < H; ++y)
        for (int32_t x = 0; x < W; ++x)
            for (int k = 0; k < 3; ++k) {
                float v = padded[((size_t)(y + top) * padW + (x + left)) * 3 + k];
                image_out[((size_t)y * W + x) * 3 + k] = v < 0 ? 0 : (v > 1 ? 1 : v);
            }
    free(nq); free(ns); free(rp); free(rc); free(mask); free(ids); free(accum); free(keys);
    free(spos); free(scov); free(sopa); free(srgb); free(padded);
    return M;
}
