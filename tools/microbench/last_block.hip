// Does "the last workgroup to finish folds the group's results" beat a second launch on MI355X (8 XCDs, an L2 each)?
// 256 workgroups x 1024 threads write one double per thread and segment (16 groups x 16 segments, as k_score_c at C3);
// variant A: a second kernel sums the 16 segment values per particle and reduces 256-particle blocks (the partials launch);
// variant B: every workgroup releases (__threadfence), takes a ticket, and the last of its group's 16 does that work.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/last_block.hip -o /tmp/lb && /tmp/lb
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#ifndef NSEG
#define NSEG 16      // -DNSEG=32 -DNGRP=1: the shape of a small filter (C2: 30 segment workgroups, one particle group)
#endif
#ifndef NGRP
#define NGRP 16
#endif
#define N (NGRP * 1024)

__device__ __forceinline__ double busy(double v, int iters) {
    for (int i = 0; i < iters; i++) v = v * 1.0000001 + 1e-9;
    return v;
}

template <bool COHERENT>
__device__ __forceinline__ void fold(const double *part, double *out, int grp) {
    const int p = grp * 1024 + threadIdx.x;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NSEG; k++) s += COHERENT ? __hip_atomic_load(&part[(size_t)k * N + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : part[(size_t)k * N + p];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ double sw[16];
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if ((threadIdx.x & 255) == 0) { const int w = threadIdx.x >> 6; out[grp * 4 + (threadIdx.x >> 8)] = ((sw[w] + sw[w + 1]) + sw[w + 2]) + sw[w + 3]; }
}

__global__ void __launch_bounds__(1024) k_work(double *part, int iters, unsigned *ticket, double *out, int tail) {
    const int i = blockIdx.x, seg = (i & 7) * (NSEG / 8) + (i >> 3) % (NSEG / 8), grp = (i >> 3) / (NSEG / 8);
    const int p = grp * 1024 + threadIdx.x;
    const double v = busy(1.0 + p * 1e-9 + seg, iters);
    if (tail >= 3) __hip_atomic_store(&part[(size_t)seg * N + p], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through (sc1): nothing dirty stays in this XCD's L2
    else part[(size_t)seg * N + p] = v;
    if (!tail) return;
    __shared__ int s_last;
    if (tail == 1) {                       // every thread fences (the textbook form)
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned t = atomicAdd(&ticket[grp], 1u);
            s_last = t == NSEG - 1;
            if (s_last) ticket[grp] = 0u;
        }
        __syncthreads();
        if (!s_last) return;
        __threadfence();
    } else if (tail == 4) {                // hardware protocol: write-through stores, wait for their acknowledgement, relaxed ticket, coherent loads
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned t = __hip_atomic_fetch_add(&ticket[grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = t == NSEG - 1;
            if (s_last) __hip_atomic_store(&ticket[grp], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!s_last) return;
        fold<true>(part, out, grp);
        return;
    } else {                               // one thread releases / acquires for the workgroup; the barriers order the rest
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned t = __hip_atomic_fetch_add(&ticket[grp], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            s_last = t == NSEG - 1;
            if (s_last) __hip_atomic_store(&ticket[grp], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!s_last) return;
    }
    fold<false>(part, out, grp);
}

__global__ void __launch_bounds__(1024) k_fold(const double *part, double *out) { fold<false>(part, out, blockIdx.x); }

int main() {
    double *part, *out, *out2; unsigned *ticket;
    CHECK(hipMalloc(&part, (size_t)NSEG * N * 8)); CHECK(hipMalloc(&out, 64 * 8)); CHECK(hipMalloc(&out2, 64 * 8));
    CHECK(hipMalloc(&ticket, NGRP * 4)); CHECK(hipMemset(ticket, 0, NGRP * 4));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int iters : {200, 1500}) {
        for (int variant = 0; variant < 6; variant++) {
            const int reps = 200;
            float ms = 0;
            for (int pass = 0; pass < 2; pass++) {
                CHECK(hipEventRecord(a));
                for (int r = 0; r < reps; r++) {
                    if (variant == 0) { k_work<<<NSEG * NGRP, 1024>>>(part, iters, ticket, out, 0); }
                    else if (variant == 1) { k_work<<<NSEG * NGRP, 1024>>>(part, iters, ticket, out, 0); k_fold<<<NGRP, 1024>>>(part, out); }
                    else { k_work<<<NSEG * NGRP, 1024>>>(part, iters, ticket, out2, variant - 1); }
                }
                CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b));
            }
            printf("iters %4d  %-44s %7.2f us per step\n", iters, variant == 0 ? "work kernel alone" : variant == 1 ? "work kernel + fold kernel" : variant == 2 ? "last workgroup folds, every thread fences" : variant == 3 ? "last workgroup folds, one thread fences" : variant == 4 ? "... write-through stores, one thread fences" : "... write-through stores, s_waitcnt, coherent loads", ms / reps * 1e3);
        }
        double h1[64], h2[64];
        CHECK(hipMemcpy(h1, out, 64 * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h2, out2, 64 * 8, hipMemcpyDeviceToHost));
        int bad = 0; for (int k = 0; k < 64; k++) bad += h1[k] != h2[k];
        printf("           results differ in %d of 64 blocks\n", bad);
    }
    return 0;
}
