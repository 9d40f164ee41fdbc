// valu_cost.hip -- cycles per wave-instruction on gfx950 for the op mix of the raster kernels.
// build: hipcc --offload-arch=gfx950 -O3 valu_cost.hip -o valu_cost ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float float2_t __attribute__((ext_vector_type(2)));
#define N_IT 4096
template <int OP>
__global__ void k(float *out, unsigned long long *cyc, float a, float b) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float2_t p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N_IT; ++i) {
        if (OP == 0) {  // 8 independent v_fma_f32
            x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
            x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
        } else if (OP == 1) {  // 4 independent v_pk_fma_f32 (= 8 FMAs per lane)
            p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb);
            p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb);
        } else if (OP == 2) {  // 8 independent v_exp_f32
            x0 = __builtin_amdgcn_exp2f(x0); x1 = __builtin_amdgcn_exp2f(x1); x2 = __builtin_amdgcn_exp2f(x2); x3 = __builtin_amdgcn_exp2f(x3);
            x4 = __builtin_amdgcn_exp2f(x4); x5 = __builtin_amdgcn_exp2f(x5); x6 = __builtin_amdgcn_exp2f(x6); x7 = __builtin_amdgcn_exp2f(x7);
        } else if (OP == 3) {  // 8 independent DPP wave_shr:1 moves
#define D(v) v = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false))
            D(x0); D(x1); D(x2); D(x3); D(x4); D(x5); D(x6); D(x7);
        } else if (OP == 4) {  // dependent chain of 8 v_fma_f32
            x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b);
            x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b);
        } else if (OP == 5) {  // 8 v_cndmask
            x0 = x0 > a ? x0 : b; x1 = x1 > a ? x1 : b; x2 = x2 > a ? x2 : b; x3 = x3 > a ? x3 : b;
            x4 = x4 > a ? x4 : b; x5 = x5 > a ? x5 : b; x6 = x6 > a ? x6 : b; x7 = x7 > a ? x7 : b;
        } else if (OP == 6) {  // 8 v_rcp_f32
            x0 = __builtin_amdgcn_rcpf(x0); x1 = __builtin_amdgcn_rcpf(x1); x2 = __builtin_amdgcn_rcpf(x2); x3 = __builtin_amdgcn_rcpf(x3);
            x4 = __builtin_amdgcn_rcpf(x4); x5 = __builtin_amdgcn_rcpf(x5); x6 = __builtin_amdgcn_rcpf(x6); x7 = __builtin_amdgcn_rcpf(x7);
        } else if (OP == 7) {  // DPP row_shr:1 (within 16 lanes)
#define R(v) v = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false))
            R(x0); R(x1); R(x2); R(x3); R(x4); R(x5); R(x6); R(x7);
        } else if (OP == 8) {  // 8 independent v_mul_f32
            x0 *= a; x1 *= a; x2 *= a; x3 *= a; x4 *= a; x5 *= a; x6 *= a; x7 *= a;
        } else if (OP == 9) {  // 4 independent v_pk_mul_f32
            p0 *= pa; p1 *= pa; p2 *= pa; p3 *= pa;
        }
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP>
void run(const char *name, int insts, int waves_per_simd) {
    float *out; unsigned long long *cyc;
    int blocks = 256 * waves_per_simd;  // 256-thread blocks = 4 waves = 1 per SIMD of a CU
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f, 0.5f);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    // s_memtime-style counter runs at a fixed 100 MHz? report raw ticks per instruction and per-SIMD cost
    printf("%-28s waves/SIMD=%d  ticks/inst/wave=%.3f  => SIMD cost per inst=%.3f ticks\n", name, waves_per_simd,
           s / blocks / N_IT / insts, s / blocks / N_IT / insts / waves_per_simd);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32 x8 indep", 8, w); run<1>("v_pk_fma_f32 x4 indep", 4, w); run<8>("v_mul_f32 x8", 8, w);
        run<9>("v_pk_mul_f32 x4", 4, w); run<2>("v_exp_f32 x8", 8, w); run<6>("v_rcp_f32 x8", 8, w);
        run<3>("v_mov_dpp wave_shr:1 x8", 8, w); run<7>("v_mov_dpp row_shr:1 x8", 8, w); run<5>("v_cmp+v_cndmask x8", 8, w);
        run<4>("v_fma_f32 x8 dependent", 8, w);
    }
    return 0;
}
