// raster_loop.hip -- cost of the compositing inner loop WITHOUT memory: same instruction sequence as
// raster_forward_kernel (4 pixels per lane, packed over pixel pairs), operands from registers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
#define N_IT 2048
template <int MODE>
__global__ void k(float *out, unsigned long long *cyc, const float *in) {
    const int lane = threadIdx.x & 63;
    float px = lane * 0.01f;
    f2 py2[2] = {f2{0.1f * lane, 0.2f}, f2{0.3f, 0.4f * lane}};
    f2 T[2] = {f2{1, 1}, f2{1, 1}}, cr[2] = {f2{0, 0}, f2{0, 0}}, cg[2] = {f2{0, 0}, f2{0, 0}}, cb[2] = {f2{0, 0}, f2{0, 0}};
    float gx = in[0], gy = in[1], cA = in[2], cB = in[3], cC = in[4], op = in[5], r = in[6], g = in[7], b = in[8];
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N_IT; ++i) {
        // perturb the "Gaussian" so the compiler cannot hoist
        gx += 1e-6f; gy -= 1e-6f;
        const float dx = px - gx, adx = cA * dx;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f2 dy = py2[h] - splat(gy);
            f2 al;
            if (MODE == 0) {  // packed
                const f2 t = pk_fma(splat(-cB), dy, splat(adx));
                const f2 q = pk_fma(splat(cC) * dy, dy, splat(dx) * t);
                al.x = __builtin_amdgcn_exp2f(-q.x); al.y = __builtin_amdgcn_exp2f(-q.y);
                al = al * splat(op);
                al.x = (T[h].x > 1e-4f) ? al.x : 0.f; al.y = (T[h].y > 1e-4f) ? al.y : 0.f;
                const f2 w = al * T[h];
                cr[h] = pk_fma(splat(r), w, cr[h]); cg[h] = pk_fma(splat(g), w, cg[h]); cb[h] = pk_fma(splat(b), w, cb[h]);
                T[h] = pk_fma(-al, T[h], T[h]);
            } else {  // same math, scalar per pixel
                float a2[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float dye = e ? dy.y : dy.x;
                    const float t = fmaf(-cB, dye, adx);
                    const float q = fmaf(cC * dye, dye, dx * t);
                    float a = __builtin_amdgcn_exp2f(-q) * op;
                    float Te = e ? T[h].y : T[h].x;
                    a = Te > 1e-4f ? a : 0.f;
                    const float w = a * Te;
                    if (e) { cr[h].y = fmaf(r, w, cr[h].y); cg[h].y = fmaf(g, w, cg[h].y); cb[h].y = fmaf(b, w, cb[h].y); T[h].y = fmaf(-a, Te, Te); }
                    else   { cr[h].x = fmaf(r, w, cr[h].x); cg[h].x = fmaf(g, w, cg[h].x); cb[h].x = fmaf(b, w, cb[h].x); T[h].x = fmaf(-a, Te, Te); }
                    a2[e] = a;
                }
                (void)a2;
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = T[0].x + T[0].y + T[1].x + T[1].y + cr[0].x + cr[1].y + cg[0].x + cg[1].y + cb[0].y + cb[1].x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char *name, int wps) {
    float *out, *in; unsigned long long *cyc;
    int blocks = 256 * wps;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8); hipMalloc(&in, 64);
    float h[9] = {0.5f, 0.5f, 3.f, 0.5f, 2.f, 0.01f, 0.3f, 0.6f, 0.9f};
    hipMemcpy(in, h, 36, hipMemcpyHostToDevice);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, in);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(c.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : c) s += v;
    printf("%-10s waves/SIMD=%d  cycles per Gaussian-iteration (4 px/lane): per wave %.1f, per SIMD %.1f\n", name, wps,
           s / blocks / N_IT, s / blocks / N_IT / wps);
}
int main() {
    for (int w : {1, 2, 4, 6}) { run<0>("packed", w); run<1>("scalar", w); }
    return 0;
}
