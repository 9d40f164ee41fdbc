// fetch_gather.hip -- what do rocprofv3's FETCH_SIZE / WRITE_SIZE report for the access patterns of this library?
//
// MI355X_MICROARCH.md calibrates FETCH_SIZE for ONE pattern only: a wide coalesced streaming read is reported at
// exactly half its bytes.  The compositing kernels gather 64-byte records by sorted Gaussian id, the backward kernels
// gather 16-byte rectangles and 4-byte offsets and scatter gradient rows -- for those the factor was unknown, and
// round 3's traffic figures applied the x2 blanket.  Every kernel here moves a KNOWN
// number of bytes in one of those patterns over an array far beyond the 256 MiB Infinity Cache, with indices computed
// in registers (an odd multiplier over a power-of-two range is a permutation: every record is touched exactly once, no
// index traffic).  Run under `rocprofv3 --pmc FETCH_SIZE` and, separately, `--pmc WRITE_SIZE`
// (tools/batches/gpu_r4b.sh); tools/fetch_calibration.py divides the known bytes by the counter.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/fetch_gather tools/ubench/fetch_gather.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

static constexpr uint32_t MULT = 2654435761u;  // odd: i -> i * MULT mod 2^k is a permutation of [0, 2^k)

__device__ __forceinline__ void sink(float4 v, float *out) {
    if (v.x + v.y + v.z + v.w == 1.2345e-30f) *out = v.x;  // never true for the data used; keeps the loads alive
}

// coalesced 16 bytes per lane
__global__ void k_stream16(const float4 *__restrict__ a, uint64_t n16, float *out) {
    float4 s = make_float4(0, 0, 0, 0);
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        const float4 v = a[i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    sink(s, out);
}
// coalesced 4 bytes per lane
__global__ void k_stream4(const float *__restrict__ a, uint64_t n4, float *out) {
    float s = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) s += a[i];
    sink(make_float4(s, 0, 0, 0), out);
}
// 64-byte records in pseudo-random order, four lanes x 16 bytes per record (raster kernels: id -> record)
__global__ void k_gather64_quad(const float4 *__restrict__ a, uint32_t mask, uint64_t nrec, float *out) {
    float4 s = make_float4(0, 0, 0, 0);
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < nrec * 4; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = ((uint32_t)(t >> 2) * MULT) & mask;
        const float4 v = a[(uint64_t)r * 4 + (t & 3)];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    sink(s, out);
}
// 64-byte records in pseudo-random order, ONE lane reads all four quarters (a lane gathering its own Gaussian)
__global__ void k_gather64_lane(const float4 *__restrict__ a, uint32_t mask, uint64_t nrec, float *out) {
    float4 s = make_float4(0, 0, 0, 0);
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < nrec; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = ((uint32_t)t * MULT) & mask;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = a[(uint64_t)r * 4 + q];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    sink(s, out);
}
// 16 bytes out of every 64-byte record, one lane per record (rec_geom alone; a rectangle out of a 16-byte-stride
// array behaves the same per line touched)
__global__ void k_gather16_of64(const float4 *__restrict__ a, uint32_t mask, uint64_t nrec, float *out) {
    float4 s = make_float4(0, 0, 0, 0);
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < nrec; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = ((uint32_t)t * MULT) & mask;
        const float4 v = a[(uint64_t)r * 4];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    sink(s, out);
}
// 4 bytes per lane at pseudo-random 4-byte positions (pair_offsets[id])
__global__ void k_gather4(const float *__restrict__ a, uint32_t mask4, uint64_t n, float *out) {
    float s = 0;
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < n; t += (uint64_t)gridDim.x * blockDim.x)
        s += a[((uint32_t)t * MULT) & mask4];
    sink(make_float4(s, 0, 0, 0), out);
}
// 48-byte rows (stride 48) in pseudo-random order, three lanes x 16 bytes (round 3's gradient rows, read side)
__global__ void k_gather48_rows(const float4 *__restrict__ a, uint32_t mask, uint64_t nrow, float *out) {
    float4 s = make_float4(0, 0, 0, 0);
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < nrow * 3; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = ((uint32_t)(t / 3) * MULT) & mask;
        const float4 v = a[(uint64_t)r * 3 + t % 3];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    sink(s, out);
}
// ---- writes
__global__ void k_wstream16(float4 *__restrict__ a, uint64_t n16) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x)
        a[i] = make_float4((float)i, 1.f, 2.f, 3.f);
}
// whole 64-byte lines at pseudo-random places, four lanes x 16 bytes in ONE instruction (round 4's rgb rows)
__global__ void k_wscatter64_quad(float4 *__restrict__ a, uint32_t mask, uint64_t nrec) {
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < nrec * 4; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = ((uint32_t)(t >> 2) * MULT) & mask;
        a[(uint64_t)r * 4 + (t & 3)] = make_float4((float)t, 1.f, 2.f, 3.f);
    }
}
// 48-byte rows at stride 48, one lane per row: float4 + float2 + two scalars + (elsewhere in time) a 16-byte piece --
// here simply three 16-byte stores by one lane (round 3's rgb rows, write side)
__global__ void k_wscatter48_lane(float4 *__restrict__ a, uint32_t mask, uint64_t nrow) {
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < nrow; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = ((uint32_t)t * MULT) & mask;
#pragma unroll
        for (int q = 0; q < 3; ++q) a[(uint64_t)r * 3 + q] = make_float4((float)t, 1.f, 2.f, (float)q);
    }
}
// one byte per lane at pseudo-random places (row flags)
__global__ void k_wscatter1(uint8_t *__restrict__ a, uint32_t mask1, uint64_t n) {
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < n; t += (uint64_t)gridDim.x * blockDim.x)
        a[((uint32_t)t * MULT) & mask1] = 1;
}
// eight bytes per lane at pseudo-random places (table variant's pair scatter)
__global__ void k_wscatter8(uint64_t *__restrict__ a, uint32_t mask8, uint64_t n) {
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < n; t += (uint64_t)gridDim.x * blockDim.x)
        a[((uint32_t)t * MULT) & mask8] = t;
}

int main() {
    const uint64_t BYTES = 1ull << 30;            // 1 GiB array: four times the Infinity Cache
    const uint64_t NREC = BYTES / 64;             // 2^24 records of 64 bytes
    const uint32_t RMASK = (uint32_t)(NREC - 1);
    void *buf = nullptr;
    float *out = nullptr;
    CHECK(hipMalloc(&buf, BYTES));
    CHECK(hipMalloc(&out, 256));
    CHECK(hipMemset(buf, 0, BYTES));
    CHECK(hipDeviceSynchronize());
    const dim3 grid(256 * 16), block(256);
    const uint64_t NG = NREC / 2;                 // records / rows touched by the gather kernels: 2^23 (512 MiB of 64-byte records)
    const uint64_t NROW48 = 1ull << 23;           // 48-byte rows addressed: 2^23 slots x 48 B = 384 MiB region
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto timed = [&](const char *name, double bytes, auto launch) -> int {
        launch();  // warm-up (TLB, clocks); the profiler sees both launches: per-kernel AVERAGES are what is compared
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"kernel\": \"%s\", \"useful_bytes\": %.0f, \"ms\": %.4f, \"GBs\": %.1f}\n", name, bytes, ms,
               bytes / (ms * 1e-3) / 1e9);
        return 0;
    };
    int rc = 0;
    rc |= timed("k_stream16", (double)BYTES, [&] { hipLaunchKernelGGL(k_stream16, grid, block, 0, 0, (const float4 *)buf, BYTES / 16, out); });
    rc |= timed("k_stream4", (double)BYTES, [&] { hipLaunchKernelGGL(k_stream4, grid, block, 0, 0, (const float *)buf, BYTES / 4, out); });
    rc |= timed("k_gather64_quad", 64.0 * NG, [&] { hipLaunchKernelGGL(k_gather64_quad, grid, block, 0, 0, (const float4 *)buf, RMASK, NG, out); });
    rc |= timed("k_gather64_lane", 64.0 * NG, [&] { hipLaunchKernelGGL(k_gather64_lane, grid, block, 0, 0, (const float4 *)buf, RMASK, NG, out); });
    rc |= timed("k_gather16_of64", 16.0 * NG, [&] { hipLaunchKernelGGL(k_gather16_of64, grid, block, 0, 0, (const float4 *)buf, RMASK, NG, out); });
    rc |= timed("k_gather4", 4.0 * NG, [&] { hipLaunchKernelGGL(k_gather4, grid, block, 0, 0, (const float *)buf, (uint32_t)(BYTES / 4 - 1), NG, out); });
    rc |= timed("k_gather48_rows", 48.0 * NROW48 / 2, [&] { hipLaunchKernelGGL(k_gather48_rows, grid, block, 0, 0, (const float4 *)buf, (uint32_t)(NROW48 - 1), NROW48 / 2, out); });
    rc |= timed("k_wstream16", (double)BYTES, [&] { hipLaunchKernelGGL(k_wstream16, grid, block, 0, 0, (float4 *)buf, BYTES / 16); });
    rc |= timed("k_wscatter64_quad", 64.0 * NG, [&] { hipLaunchKernelGGL(k_wscatter64_quad, grid, block, 0, 0, (float4 *)buf, RMASK, NG); });
    rc |= timed("k_wscatter48_lane", 48.0 * NROW48 / 2, [&] { hipLaunchKernelGGL(k_wscatter48_lane, grid, block, 0, 0, (float4 *)buf, (uint32_t)(NROW48 - 1), NROW48 / 2); });
    rc |= timed("k_wscatter1", 1.0 * NG, [&] { hipLaunchKernelGGL(k_wscatter1, grid, block, 0, 0, (uint8_t *)buf, (uint32_t)(BYTES - 1), NG); });
    rc |= timed("k_wscatter8", 8.0 * NG, [&] { hipLaunchKernelGGL(k_wscatter8, grid, block, 0, 0, (uint64_t *)buf, (uint32_t)(BYTES / 8 - 1), NG); });
    CHECK(hipDeviceSynchronize());
    return rc;
}
