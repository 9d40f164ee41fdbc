// mfma_4x4.hip -- operand/result layout of v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products):
// checks that register i of lane 4b + j holds A[4b + i] * B[4b + j] and times it next to 3 v_pk_fma_f32.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void layout(float *out) {
    const int lane = threadIdx.x;
    f4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)lane, 100.f + lane, d, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = d[i];
}
template <int MODE>
__global__ void __launch_bounds__(256) rate(float *out, float s) {
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f2 c[6] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
    float w = s * threadIdx.x, col = s + (threadIdx.x & 3);
    for (int it = 0; it < 4096; ++it) {
        w += 1e-7f;
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(w + k, col, acc[k], 0, 0, 0);
        } else {
            const f2 w2 = {w, w + 1.f};
#pragma unroll
            for (int k = 0; k < 6; ++k) c[k] = __builtin_elementwise_fma(f2{col + k, col + k}, w2, c[k]);
        }
    }
    float r = 0;
    for (int k = 0; k < 4; ++k) r += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    for (int k = 0; k < 6; ++k) r += c[k].x + c[k].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
int main() {
    float *out, h[256];
    (void)hipMalloc(&out, 2048 * 256 * 4);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, out);
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int i = 0; i < 4; ++i) {
            const int b = lane / 4, j = lane % 4;
            const float want = (float)(4 * b + i) * (100.f + 4 * b + j);
            if (h[lane * 4 + i] != want) ++bad;
        }
    printf("layout D[reg i][lane 4b+j] = A[4b+i] * B[4b+j]: %s (%d mismatches); lane 5: %g %g %g %g\n", bad ? "NO" : "yes", bad,
           h[20], h[21], h[22], h[23]);
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int r = 0; r < 2; ++r) {
            (void)hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(2048), dim3(256), 0, 0, out, 0.5f);
            else hipLaunchKernelGGL(rate<1>, dim3(2048), dim3(256), 0, 0, out, 0.5f);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.2f ns per Gaussian step (4 px/lane colour accumulation) per SIMD\n", mode == 0 ? "4 x mfma_4x4x1 " : "6 x v_pk_fma_f32",
               ms * 1e6 / (4096.0 * 8));
    }
    return 0;
}
