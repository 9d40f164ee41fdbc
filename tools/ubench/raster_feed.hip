// raster_feed.hip -- how should the compositing loop be FED?  Same math as raster_forward_kernel
// (4 pixels per lane, packed over pixel pairs); the per-Gaussian operands (x, y, A, B, C, opacity,
// r, g, b -- identical for every lane of the wave) come from
//   MODE 0: registers (no memory at all; lower bound),
//   MODE 1: LDS, nine broadcast ds_read_b128 per group of four Gaussians (what raster_fwd.hip does),
//   MODE 2: SGPRs, one s_load_dwordx16 per Gaussian through the scalar cache, gathered by a sorted id
//           list (scalar loads, prefetched one group ahead),
//   MODE 3: as 2 but reading records sequentially (no id indirection).
// One wave per workgroup so that every address is provably wave-uniform.  Reports wall time scaled to
// the cfg2 frame (1,088,150 (tile, Gaussian) steps spread over 1024 SIMDs).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
#define N_IT 512  // Gaussians per wave (multiple of 4)
#define N_REC (1 << 18)

struct Rec {  // 64 B
    float x, y, opa, A, B, C, r, g, b, pad[7];
};

struct Px {
    float px;
    f2 py2[2], T[2], cr[2], cg[2], cb[2];
};

__device__ __forceinline__ void step(Px &P, float gx, float gy, float cA, float cB, float cC, float op, float r,
                                     float g, float b) {
    const float dx = P.px - gx, adx = cA * dx;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f2 dy = P.py2[h] - splat(gy);
        const f2 t = pk_fma(splat(-cB), dy, splat(adx));
        const f2 q = pk_fma(splat(cC) * dy, dy, splat(dx) * t);
        f2 al;
        al.x = __builtin_amdgcn_exp2f(-q.x);
        al.y = __builtin_amdgcn_exp2f(-q.y);
        al = al * splat(op);
        al.x = (P.T[h].x > 1e-4f) ? al.x : 0.f;
        al.y = (P.T[h].y > 1e-4f) ? al.y : 0.f;
        const f2 w = al * P.T[h];
        P.cr[h] = pk_fma(splat(r), w, P.cr[h]);
        P.cg[h] = pk_fma(splat(g), w, P.cg[h]);
        P.cb[h] = pk_fma(splat(b), w, P.cb[h]);
        P.T[h] = pk_fma(-al, P.T[h], P.T[h]);
    }
}

template <int MODE>
__global__ void __launch_bounds__(64) k(float *__restrict__ out, const Rec *__restrict__ rec,
                                        const uint32_t *__restrict__ ids) {
    const int lane = threadIdx.x;
    Px P;
    P.px = lane * 0.01f;
    P.py2[0] = f2{0.1f * lane, 0.2f};
    P.py2[1] = f2{0.3f, 0.4f * lane};
    for (int h = 0; h < 2; ++h) {
        P.T[h] = f2{1, 1};
        P.cr[h] = P.cg[h] = P.cb[h] = f2{0, 0};
    }
    const uint32_t base = (blockIdx.x * N_IT) & (N_REC - 1);
    if (MODE == 0) {
        Rec R = rec[ids[base]];
        float gx = R.x, gy = R.y;
        for (int i = 0; i < N_IT; ++i) {
            gx += 1e-6f;
            gy -= 1e-6f;
            step(P, gx, gy, R.A, R.B, R.C, R.opa, R.r, R.g, R.b);
        }
    } else if (MODE == 1) {
        __shared__ float f[2][9][64] __attribute__((aligned(16)));
        // stage one chunk of 64 per buffer as the real kernel does (gather by id, one Gaussian per lane)
        Rec R = rec[ids[base + lane]];
        for (int c = 0; c < N_IT / 64; ++c) {
            const int buf = c & 1;
            f[buf][0][lane] = R.x; f[buf][1][lane] = R.y; f[buf][2][lane] = R.A; f[buf][3][lane] = R.B;
            f[buf][4][lane] = R.C; f[buf][5][lane] = R.opa; f[buf][6][lane] = R.r; f[buf][7][lane] = R.g;
            f[buf][8][lane] = R.b;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (c + 1 < N_IT / 64) R = rec[ids[base + (c + 1) * 64 + lane]];
            for (int i = 0; i < 64; i += 4) {
                bool l = P.T[0].x > 1e-4f || P.T[0].y > 1e-4f || P.T[1].x > 1e-4f || P.T[1].y > 1e-4f;
                if (__ballot(l) == 0ull) break;
                auto ld4 = [&](int q) { return *(const float4 *)__builtin_assume_aligned(&f[buf][q][i], 16); };
                const float4 X = ld4(0), Y = ld4(1), A = ld4(2), B = ld4(3), C = ld4(4), O = ld4(5), Rr = ld4(6),
                             G = ld4(7), Bl = ld4(8);
                step(P, X.x, Y.x, A.x, B.x, C.x, O.x, Rr.x, G.x, Bl.x);
                step(P, X.y, Y.y, A.y, B.y, C.y, O.y, Rr.y, G.y, Bl.y);
                step(P, X.z, Y.z, A.z, B.z, C.z, O.z, Rr.z, G.z, Bl.z);
                step(P, X.w, Y.w, A.w, B.w, C.w, O.w, Rr.w, G.w, Bl.w);
            }
        }
    } else {
        // scalar feed: ids / records addressed with wave-uniform indices => s_load
        auto rec_of = [&](uint32_t j) -> const Rec & { return rec[MODE == 2 ? ids[j] : (j & (N_REC - 1))]; };
        Rec c0 = rec_of(base + 0), c1 = rec_of(base + 1);
        for (int i = 0; i < N_IT; i += 2) {
            const Rec n0 = rec_of(base + ((i + 2) & (N_IT - 1))), n1 = rec_of(base + ((i + 3) & (N_IT - 1)));
            bool l = P.T[0].x > 1e-4f || P.T[0].y > 1e-4f || P.T[1].x > 1e-4f || P.T[1].y > 1e-4f;
            if (__ballot(l) == 0ull) break;
            step(P, c0.x, c0.y, c0.A, c0.B, c0.C, c0.opa, c0.r, c0.g, c0.b);
            step(P, c1.x, c1.y, c1.A, c1.B, c1.C, c1.opa, c1.r, c1.g, c1.b);
            c0 = n0;
            c1 = n1;
        }
    }
    out[blockIdx.x * 64 + lane] = P.T[0].x + P.T[0].y + P.T[1].x + P.T[1].y + P.cr[0].x + P.cr[1].y + P.cg[0].x +
                                  P.cg[1].y + P.cb[0].y + P.cb[1].x;
}

template <int MODE>
void run(const char *name, int wps, float *out, Rec *rec, uint32_t *ids) {
    const int blocks = 1024 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, rec, ids);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, rec, ids);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us_per_kernel = ms * 1e3 / 5;
    const double steps_per_simd = (double)N_IT * wps;
    printf("%-22s waves/SIMD=%d  %.1f ns per step per SIMD  => cfg2 frame (1062 steps/SIMD): %.1f us\n", name, wps,
           us_per_kernel * 1e3 / steps_per_simd, us_per_kernel / steps_per_simd * 1062.6);
}

int main() {
    float *out;
    Rec *rec;
    uint32_t *ids;
    hipMalloc(&out, 1024 * 8 * 64 * 4);
    hipMalloc(&rec, sizeof(Rec) * N_REC);
    hipMalloc(&ids, 4 * N_REC);
    std::vector<Rec> h(N_REC);
    std::vector<uint32_t> hi(N_REC);
    srand(1);
    for (int i = 0; i < N_REC; ++i) {
        h[i] = Rec{0.5f + 1e-6f * i, 0.5f, 0.01f, 3.f, 0.5f, 2.f, 0.3f, 0.6f, 0.9f, {0}};
        hi[i] = (uint32_t)(((unsigned)rand() * 2654435761u) & (N_REC - 1));
    }
    hipMemcpy(rec, h.data(), sizeof(Rec) * N_REC, hipMemcpyHostToDevice);
    hipMemcpy(ids, hi.data(), 4 * N_REC, hipMemcpyHostToDevice);
    for (int w : {2, 4, 5, 8}) {
        run<0>("registers", w, out, rec, ids);
        run<1>("LDS ring (current)", w, out, rec, ids);
        run<2>("SGPR, gather by id", w, out, rec, ids);
        run<3>("SGPR, sequential", w, out, rec, ids);
    }
    return 0;
}
