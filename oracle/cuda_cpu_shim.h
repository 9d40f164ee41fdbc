// cuda_cpu_shim.h -- a minimal SIMT emulator so that the REFERENCE's own CUDA kernels
// (/root/reference/src/gaussian.cu, compiled from where they lie, see oracle/build_ref.py)
// can run on the CPU and pin the oracle.  TEST INFRASTRUCTURE ONLY.
//
// Every CUDA thread of a block is a ucontext fiber; the scheduler runs each fiber until it
// blocks in __syncthreads / __shfl_down_sync / __activemask (or returns) and releases a group
// when all of its members have arrived.  Execution is deterministic (lane order), atomics are
// plain read-modify-writes.  Semantics that CUDA leaves undefined are DETECTED, not guessed:
//   * a shuffle that reads a lane outside the participating mask counts one "undefined read"
//     (simt::undefined_reads) and returns the caller's own value;
//   * a barrier that can never complete returns a deadlock status from the launch.
// __activemask() returns the lanes of the warp that reach the call together, i.e. those not
// parked at a barrier and not exited once every lane has run to its next blocking point --
// the behaviour the reference relies on (gaussian.cu:675-676).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <cmath>
#include <cstdlib>
#include <functional>
#include <vector>

using std::abs;

#define __global__
#define __device__
#define __constant__
#define __inline__ inline
#define __shared__ static
#define DIV_ROUND_UP(X, Y) ((X) + (Y)-1) / (Y) /* src/include/common.hpp:34 */

struct uint3 {
    unsigned x, y, z;
};
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
static uint3 threadIdx, blockIdx;
static dim3 blockDim, gridDim;

static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
#define __expf(x) expf(x) /* fast-math intrinsic -> libm (glibc reserves the name __expf) */
static inline int atomicAdd(int *p, int v) {
    int o = *p;
    *p = o + v;
    return o;
}
static inline float atomicAdd(float *p, float v) {
    float o = *p;
    *p = o + v;
    return o;
}

namespace simt {
enum St { RUN, SYNC, SHFL, AMASK, DONE };
struct Fiber {
    ucontext_t ctx;
    St st;
    uint3 tid;
    float sh_val, sh_res;
    int sh_off;
    unsigned sh_mask, am_res;
};
static std::vector<Fiber> fibers;
static std::vector<char> stacks;
static ucontext_t sched_ctx;
static int cur = -1;
static std::function<void()> body;
static long undefined_reads = 0;
static const size_t STACK = 96 * 1024;

static void trampoline() {
    body();
    fibers[cur].st = DONE;
    swapcontext(&fibers[cur].ctx, &sched_ctx);
}
static inline void yield_to_scheduler() { swapcontext(&fibers[cur].ctx, &sched_ctx); }

// Runs one block; returns 0, or 1 on deadlock.
static int run_block(dim3 bd) {
    const int nt = (int)(bd.x * bd.y * bd.z);
    fibers.assign(nt, Fiber());
    if (stacks.size() < (size_t)nt * STACK) stacks.resize((size_t)nt * STACK);
    for (int i = 0; i < nt; ++i) {
        Fiber &f = fibers[i];
        f.st = RUN;
        f.tid.x = i % bd.x;
        f.tid.y = (i / bd.x) % bd.y;
        f.tid.z = i / (bd.x * bd.y);
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = stacks.data() + (size_t)i * STACK;
        f.ctx.uc_stack.ss_size = STACK;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())trampoline, 0);
    }
    const int nwarp = (nt + 31) / 32;
    for (;;) {
        for (int i = 0; i < nt; ++i)
            if (fibers[i].st == RUN) {
                cur = i;
                threadIdx = fibers[i].tid;
                swapcontext(&sched_ctx, &fibers[i].ctx);
            }
        bool released = false, all_done = true;
        for (int w = 0; w < nwarp; ++w) {
            const int lo = w * 32, hi = (lo + 32 < nt) ? lo + 32 : nt;
            unsigned need = 0, at_shfl = 0, at_amask = 0;
            for (int i = lo; i < hi; ++i) {
                if (fibers[i].st == SHFL) {
                    at_shfl |= 1u << (i - lo);
                    need |= fibers[i].sh_mask;
                }
                if (fibers[i].st == AMASK) at_amask |= 1u << (i - lo);
            }
            if (at_shfl) {
                bool ok = true;
                for (int i = lo; i < hi; ++i)
                    if ((need >> (i - lo) & 1) && fibers[i].st != SHFL && fibers[i].st != DONE) ok = false;
                if (ok) {
                    for (int i = lo; i < hi; ++i) {
                        Fiber &f = fibers[i];
                        if (f.st != SHFL) continue;
                        const int src = (i - lo) + f.sh_off;
                        if (src < 32 && lo + src < hi && (at_shfl >> src & 1)) {
                            f.sh_res = fibers[lo + src].sh_val;
                        } else {
                            f.sh_res = f.sh_val;  // out of range: own value (CUDA); inactive: undefined
                            if (src < 32) ++undefined_reads;
                        }
                    }
                    for (int i = lo; i < hi; ++i)
                        if (fibers[i].st == SHFL) fibers[i].st = RUN;
                    released = true;
                }
            } else if (at_amask) {
                for (int i = lo; i < hi; ++i)
                    if (fibers[i].st == AMASK) {
                        fibers[i].am_res = at_amask;
                        fibers[i].st = RUN;
                    }
                released = true;
            }
        }
        if (!released) {
            bool any_sync = false, all_sync = true;
            for (int i = 0; i < nt; ++i) {
                if (fibers[i].st == SYNC) any_sync = true;
                else if (fibers[i].st != DONE) all_sync = false;
            }
            if (any_sync && all_sync) {
                for (int i = 0; i < nt; ++i)
                    if (fibers[i].st == SYNC) fibers[i].st = RUN;
                released = true;
            }
        }
        for (int i = 0; i < nt; ++i)
            if (fibers[i].st != DONE) all_done = false;
        if (all_done) return 0;
        if (!released) return 1;
    }
}

// <<<grid, block>>> : blocks run one after another.
static int launch(dim3 grid, dim3 block, std::function<void()> kernel_call) {
    body = kernel_call;
    gridDim = grid;
    blockDim = block;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx.x = bx;
                blockIdx.y = by;
                blockIdx.z = bz;
                if (run_block(block)) return 1;
            }
    return 0;
}
}  // namespace simt

static inline void __syncthreads() {
    simt::fibers[simt::cur].st = simt::SYNC;
    simt::yield_to_scheduler();
}
static inline unsigned __activemask() {
    simt::fibers[simt::cur].st = simt::AMASK;
    simt::yield_to_scheduler();
    return simt::fibers[simt::cur].am_res;
}
static inline float __shfl_down_sync(unsigned mask, float v, int offset) {
    simt::Fiber &f = simt::fibers[simt::cur];
    f.sh_val = v;
    f.sh_off = offset;
    f.sh_mask = mask;
    f.st = simt::SHFL;
    simt::yield_to_scheduler();
    return simt::fibers[simt::cur].sh_res;
}
