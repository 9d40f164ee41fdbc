// mfma_valu_overlap.hip -- does an fp32 MFMA (v_mfma_f32_16x16x4_f32, 32 cycles of the matrix pipe) run BESIDE the
// VALU stream on gfx950, inside one wave and across the waves of a SIMD?  And what do v_exp / v_rcp / DPP steps cost?
//
// The SH backward on the matrix pipe (raster_backward_mfma_sh_kernel) issues 24 MFMAs and ~280 VALU instructions per
// step of a wave; its first measurement (2.4 M Gaussians, degree 3: 1.88 ms against 0.7 ms of pure MFMA time and ~0.9 ms
// of estimated VALU time) looked like the two were added, not overlapped.  This program times, per iteration and SIMD:
//   mfma      8 independent MFMAs
//   fma       128 v_fma_f32 (8 chains)
//   seq       8 MFMAs, then 128 v_fma_f32            (one after the other in program order)
//   mix       8 x (1 MFMA + 16 v_fma_f32)            (interleaved in program order)
//   exp, rcp, dpp   32 v_exp_f32 / 32 v_rcp_f32 / 32 v_add_f32_dpp row_shr
//   bf16      8 independent v_mfma_f32_16x16x32_bf16;  bfmix / bfseq: the same with 128 v_fma_f32 interleaved / behind
// at 1, 2, 3 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int ITERS = 4096;

#define MF(i) "v_mfma_f32_16x16x4_f32 %" #i ", %8, %9, %" #i "\n\t"
#define F8(a) "v_fma_f32 %" #a ", %" #a ", %10, %11\n\t"
#define FMA8 F8(12) F8(13) F8(14) F8(15) F8(16) F8(17) F8(18) F8(19)
#define FMA16 FMA8 FMA8
#define MB(i) "v_mfma_f32_16x16x32_bf16 %" #i ", %20, %21, %" #i "\n\t"
#define OPS                                                                                                       \
    : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])              \
    : "v"(a), "v"(b), "v"(m), "v"(c), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(ha), "v"(hb)

template <int MODE>
__global__ void __launch_bounds__(64) k(float *out, float s) {
    const int lane = threadIdx.x;
    f4 d[8];
    float x[8];
    for (int i = 0; i < 8; ++i) {
        d[i] = f4{s, s, s, s};
        x[i] = s * (lane + i);
    }
    float a = s * lane, b = s + lane, m = 0.999f, c = 1e-7f * s;
    typedef short h8 __attribute__((ext_vector_type(8)));
    h8 ha, hb;
    for (int i = 0; i < 8; ++i) {
        ha[i] = (short)(0x3f80 + lane + i);  // bf16 bit patterns near 1.0
        hb[i] = (short)(0x3c00 + i);
    }
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) asm volatile(MF(0) MF(1) MF(2) MF(3) MF(4) MF(5) MF(6) MF(7) OPS);
        if (MODE == 1) asm volatile(FMA16 FMA16 FMA16 FMA16 FMA16 FMA16 FMA16 FMA16 OPS);
        if (MODE == 2) asm volatile(MF(0) MF(1) MF(2) MF(3) MF(4) MF(5) MF(6) MF(7) FMA16 FMA16 FMA16 FMA16 FMA16 FMA16 FMA16 FMA16 OPS);
        if (MODE == 3)
            asm volatile(MF(0) FMA16 MF(1) FMA16 MF(2) FMA16 MF(3) FMA16 MF(4) FMA16 MF(5) FMA16 MF(6) FMA16 MF(7) FMA16 OPS);
        if (MODE == 7) asm volatile(MB(0) MB(1) MB(2) MB(3) MB(4) MB(5) MB(6) MB(7) OPS);
        if (MODE == 8)
            asm volatile(MB(0) FMA16 MB(1) FMA16 MB(2) FMA16 MB(3) FMA16 MB(4) FMA16 MB(5) FMA16 MB(6) FMA16 MB(7) FMA16 OPS);
        if (MODE == 9)
            asm volatile(MB(0) MB(1) MB(2) MB(3) MB(4) MB(5) MB(6) MB(7) FMA16 FMA16 FMA16 FMA16 FMA16 FMA16 FMA16 FMA16 OPS);
        if (MODE == 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]);
        }
        if (MODE == 5) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_rcpf(x[i]);
        }
        if (MODE == 6) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %5, %5, %5 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %6, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %7, %7, %7 row_shr:8 row_mask:0xf bank_mask:0xf"
                             : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
        }
    }
    float acc = 0;
    for (int i = 0; i < 8; ++i) acc += d[i][0] + d[i][1] + d[i][2] + d[i][3] + x[i];
    out[blockIdx.x * 64 + lane] = acc;
}

template <int MODE>
void run(const char *name, float *out) {
    printf("%-6s", name);
    for (int wps : {1, 2, 3, 4}) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        float ms = 0;
        for (int r = 0; r < 3; ++r) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k<MODE>, dim3(1024 * wps), dim3(64), 0, 0, out, 0.5f);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        printf("  %d wave%s/SIMD: %7.1f ns per iteration per SIMD", wps, wps > 1 ? "s" : " ", ms * 1e6 / ((double)ITERS * wps));
    }
    printf("\n");
}

int main() {
    float *out;
    (void)hipMalloc(&out, 4096 * 64 * 4);
    run<0>("mfma", out);
    run<1>("fma", out);
    run<2>("seq", out);
    run<3>("mix", out);
    run<4>("exp", out);
    run<5>("rcp", out);
    run<6>("dpp", out);
    run<7>("bf16", out);
    run<8>("bfmix", out);
    run<9>("bfseq", out);
    return 0;
}
