// sort_ops.hip -- wall-clock issue cost (ns per wave-instruction per SIMD) of the primitives the per-tile sort and the
// compositing loop are made of, on a full chip (256 CUs x 4 SIMDs x W waves).  hipEvent-timed, not s_memtime.
// build: hipcc --offload-arch=gfx950 -O3 sort_ops.hip -o sort_ops ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define N_IT 2048
template <int CTRL>
__device__ __forceinline__ uint32_t dpp32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t a, uint32_t b) {
    const int lane = threadIdx.x & 63;
    uint32_t x[8];
    uint64_t y[4];
    float f[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        x[i] = threadIdx.x * 2654435761u + i * a;
        f[i] = 1.0f + (threadIdx.x + i) * 1e-3f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        y[i] = ((uint64_t)x[i] << 32) | x[i + 4];
        p[i] = f2{f[i], f[i + 4]};
    }
    const float fa = __uint_as_float(a | 0x3f800000u) , fb = 0.5f;
    for (int it = 0; it < N_IT; ++it) {
        if (OP == 0) {  // 8 v_fma_f32
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = fmaf(f[i], fa, fb);
        } else if (OP == 1) {  // 4 v_pk_fma_f32
#pragma unroll
            for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(p[i], f2{fa, fa}, f2{fb, fb});
        } else if (OP == 2) {  // 8 v_mul_legacy_f32
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mul_legacy_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fa));
        } else if (OP == 3) {  // 8 DPP quad_perm moves
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = dpp32<0xB1>(x[i]);
        } else if (OP == 4) {  // 8 DPP row_mirror moves
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = dpp32<0x140>(x[i]);
        } else if (OP == 5) {  // 8 ds_bpermute
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, (int)x[i]);
        } else if (OP == 6) {  // 4 x (v_cmp_lt_u64 + 2 v_cndmask): the select of a 64-bit compare-exchange
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint64_t o = y[(i + 1) & 3] + b;
                y[i] = y[i] < o ? y[i] : o;
            }
        } else if (OP == 7) {  // 8 v_min_u32
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = min(x[i], x[(i + 1) & 7] + b);
        } else if (OP == 8) {  // 4 complete 64-bit compare-exchanges with lane ^ 1 (2 DPP + cmp + 2 cndmask), as tile_sort.hip
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint64_t o = ((uint64_t)dpp32<0xB1>((uint32_t)(y[i] >> 32)) << 32) | dpp32<0xB1>((uint32_t)y[i]);
                const bool low = (lane & 1) == 0;
                y[i] = ((y[i] < o) == low) ? y[i] : o;
            }
        } else if (OP == 9) {  // 4 complete 64-bit compare-exchanges with lane ^ 32 (2 ds_bpermute + cmp + 2 cndmask)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int src = (lane ^ 32) << 2;
                const uint64_t o = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(y[i] >> 32)) << 32) |
                                   (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)y[i]);
                const bool low = (lane & 32) == 0;
                y[i] = ((y[i] < o) == low) ? y[i] : o;
            }
        } else if (OP == 10) {  // 8 32-bit key-only compare-exchanges with lane ^ 1 (DPP + min + max + cndmask)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t o = dpp32<0xB1>(x[i]);
                x[i] = (lane & 1) == 0 ? min(x[i], o) : max(x[i], o);
            }
        } else if (OP == 11) {  // 8 v_exp_f32
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = __builtin_amdgcn_exp2f(f[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(x[i]), "+v"(f[i]));
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(y[i]), "+v"(p[i]));
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += x[i] + __float_as_uint(f[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) r += (uint32_t)y[i] + (uint32_t)(y[i] >> 32) + __float_as_uint(p[i].x + p[i].y);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP>
void run(const char *name, int units, int waves_per_simd) {
    uint32_t *out;
    const int blocks = 256 * waves_per_simd;  // 256-thread blocks = 4 waves = 1 per SIMD of a CU
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 3u, 1u);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 3u, 1u);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // one SIMD executes waves_per_simd waves x N_IT iterations x `units` of the op per launch
    printf("%-44s waves/SIMD=%d  %.3f ns per unit per SIMD\n", name, waves_per_simd,
           ms / 5 * 1e6 / ((double)waves_per_simd * N_IT * units));
    hipFree(out);
}
int main() {
    for (int w : {2, 5, 8}) {
        run<0>("v_fma_f32", 8, w);
        run<1>("v_pk_fma_f32", 4, w);
        run<2>("v_mul_legacy_f32", 8, w);
        run<11>("v_exp_f32", 8, w);
        run<3>("v_mov_dpp quad_perm", 8, w);
        run<4>("v_mov_dpp row_mirror", 8, w);
        run<5>("ds_bpermute_b32", 8, w);
        run<6>("64-bit min (cmp_u64 + 2 cndmask)", 4, w);
        run<7>("v_min_u32", 8, w);
        run<8>("64-bit compare-exchange, DPP (lane^1)", 4, w);
        run<9>("64-bit compare-exchange, bpermute (lane^32)", 4, w);
        run<10>("32-bit key-only compare-exchange, DPP", 8, w);
    }
    return 0;
}
