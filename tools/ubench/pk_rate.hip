// pk_rate.hip -- wall-clock issue rate of v_fma_f32, v_pk_fma_f32, v_pk_mul_f32, v_exp_f32 and DPP moves:
// 16 independent accumulator chains per lane, 8 waves per SIMD, whole chip busy.  Prints ns per
// wave-instruction per SIMD (= cycles / clock) and the implied TFLOP/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define N_IT 4096
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, float seed) {
    float a[16];
    f2 p[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed * i + threadIdx.x; p[i] = f2{a[i], a[i] + 1.f}; }
    const float m = 1.0000001f, c = 1e-9f;
    const f2 m2 = {m, m}, c2 = {c, c};
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) a[i] = fmaf(a[i], m, c);
            if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], m2, c2);
            if (MODE == 2) p[i] = p[i] * m2;
            if (MODE == 3) a[i] = __builtin_amdgcn_exp2f(a[i]);
            if (MODE == 4) a[i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0x138, 0xf, 0xf, false));
            if (MODE == 5) a[i] = __builtin_amdgcn_rcpf(a[i]);
            if (MODE == 6) a[i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0x111, 0xf, 0xf, false));  // row_shr:1
            if (MODE == 7) a[i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0xB1, 0xf, 0xf, false));   // quad_perm
            if (MODE == 8) a[i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0x140, 0xf, 0xf, false));  // row_mirror
            if (MODE == 9) a[i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0x142, 0xa, 0xf, false));  // row_bcast15
            if (MODE == 10) a[i] = a[i] > c ? a[i] * m : c;                                   // cmp + cndmask + mul
            if (MODE == 11) a[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((threadIdx.x ^ 32) << 2, __builtin_bit_cast(int, a[i])));
            if (MODE == 12) a[i] = a[i] + c;
            if (MODE == 13) { float t = a[i] * m; asm volatile("v_add_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(t), "v"(a[i])); }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, float *out, double flops_per_instr_lane) {
    const int blocks = 256 * 8;  // 8 waves per SIMD
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 0.5f);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 0.5f);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double t = ms * 1e-3 / 3;
    const double instr_per_simd = (double)N_IT * 16 * 8;  // per SIMD: 8 waves
    printf("%-14s %.3f ns per wave-instruction per SIMD   %.1f TFLOP/s\n", name, t / instr_per_simd * 1e9,
           (double)blocks * 256 * N_IT * 16 * flops_per_instr_lane / t / 1e12);
}
int main() {
    float *out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int rep = 0; rep < 1; ++rep) {
        run<0>("v_fma_f32", out, 2);
        run<1>("v_pk_fma_f32", out, 4);
        run<2>("v_pk_mul_f32", out, 2);
        run<3>("v_exp_f32", out, 1);
        run<4>("v_mov_dpp", out, 0);
        run<5>("v_rcp_f32", out, 1);
        run<6>("dpp row_shr:1", out, 0);
        run<7>("dpp quad_perm", out, 0);
        run<8>("dpp row_mirror", out, 0);
        run<9>("dpp row_bcast15", out, 0);
        run<10>("cmp+cnd+mul", out, 1);
        run<11>("ds_bpermute", out, 0);
        run<12>("v_add_f32", out, 1);
        run<13>("mul + add_dpp wave_shr", out, 2);
    }
    return 0;
}
