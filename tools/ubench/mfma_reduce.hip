// mfma_reduce.hip -- can the matrix pipe take over the cross-lane reduction of raster_backward_pixel_sh_kernel?
//
// Per Gaussian that kernel reduces NROW = 7 + 27 (degree-2 SH) per-lane partial sums over the 64 lanes of the wave
// (the sum over the tile's 256 pixels of dL/dC w c (1 - c) sh_k(pixel), ...): the one true contraction on the path.
// The contraction index is the LANE (pixel); an MFMA contracts over K, and of K only the part l / 16 (16x16x4) or l / 32
// (32x32x2) lies across lanes.  The most a single fp32 MFMA can do for one register of partials is therefore
//     out[i][j] = sum_{k < 4} v[16 k + i] * 1        (v_mfma_f32_16x16x4_f32, A = the partials, B = ones)
// a 4 : 1 reduction that leaves 16 sums (replicated over j) per row -- the remaining 16 : 1 still needs DPP or LDS.
// This program times, per Gaussian step and per SIMD (4 resident waves, like the kernel):
//   lds    the kernel's reduction as it is: 34 ds_write_b32, three rounds of quarter-row sums, final adds;
//   mfma   34 x v_mfma_f32_16x16x4_f32 alone (the 4 : 1 step for every row; a LOWER bound of any MFMA scheme);
// and checks the MFMA result against the plain sum of the four lane groups.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int NROW = 34, ITERS = 2048;

__global__ void check(float *out) {
    const int lane = threadIdx.x;
    f4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_16x16x4f32((float)(lane * lane % 97), 1.0f, d, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = d[i];  // lane l, register r: out[i = 4 (l / 16) + r][j = l % 16]
}

template <int MODE>
__global__ void __launch_bounds__(64) reduce_rate(float *out, float s) {
    __shared__ float s_red[16 * 65];
    __shared__ float s_part[NROW][4];
    const int lane = threadIdx.x;
    float v[NROW];
#pragma unroll
    for (int m = 0; m < NROW; ++m) v[m] = s * (lane + m);
    float acc = 0.f;
    f4 d[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const unsigned red_row = lane & 15, red_part = lane >> 4;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int m = 0; m < NROW; ++m) v[m] += 1e-7f;  // (a stand-in for the step's arithmetic: keeps the values live)
        if (MODE == 0) {
#pragma unroll
            for (int rd = 0; rd < (NROW + 15) / 16; ++rd) {
#pragma unroll
                for (int m = 16 * rd; m < 16 * rd + 16 && m < NROW; ++m) s_red[(m - 16 * rd) * 65 + lane] = v[m];
                __builtin_amdgcn_wave_barrier();
                const unsigned row = 16 * rd + red_row;
                if (row < (unsigned)NROW) {
                    const float *src = s_red + red_row * 65 + red_part * 16;
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                    for (int j = 0; j < 16; j += 4) {
                        a0 += src[j];
                        a1 += src[j + 1];
                        a2 += src[j + 2];
                        a3 += src[j + 3];
                    }
                    s_part[row][red_part] = (a0 + a1) + (a2 + a3);
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (lane < NROW) acc += (s_part[lane][0] + s_part[lane][1]) + (s_part[lane][2] + s_part[lane][3]);
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int m = 0; m < NROW; ++m) d[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[m], 1.0f, d[m & 3], 0, 0, 0);
        }
    }
    for (int k = 0; k < 4; ++k) acc += d[k][0] + d[k][1] + d[k][2] + d[k][3];
    out[blockIdx.x * 64 + lane] = acc;
}

int main() {
    float *out, h[256];
    (void)hipMalloc(&out, 4096 * 64 * 4);
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, out);
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * (l / 16) + r;
            float want = 0;
            for (int k = 0; k < 4; ++k) want += (float)((16 * k + i) * (16 * k + i) % 97);
            if (h[l * 4 + r] != want) ++bad;
        }
    printf("v_mfma_f32_16x16x4_f32 with B = 1: out[i][*] = v[i] + v[16+i] + v[32+i] + v[48+i]: %s (%d mismatches)\n",
           bad ? "NO" : "yes", bad);
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        float ms = 0;
        for (int r = 0; r < 2; ++r) {  // 4096 waves = 4 per SIMD on 256 CUs
            (void)hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(reduce_rate<0>, dim3(4096), dim3(64), 0, 0, out, 0.5f);
            else hipLaunchKernelGGL(reduce_rate<1>, dim3(4096), dim3(64), 0, 0, out, 0.5f);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%s: %.1f ns per Gaussian step per SIMD (34 rows, 4 waves per SIMD)\n",
               mode == 0 ? "LDS reduction of the kernel (64 : 1, complete)    " : "34 x v_mfma_f32_16x16x4_f32 (4 : 1 only, no 16 : 1)",
               ms * 1e6 / ((double)ITERS * 4));
    }
    return 0;
}
