// Ceiling measurement for the scoring kernel's access pattern: independent 8-byte gathers, one per lane, from a
// 32 MiB table (L2 / Infinity-Cache resident), full occupancy, nothing else in the loop.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/gather8.hip -o /tmp/gather8 && /tmp/gather8
// Prints lanes/s for (a) uniformly random cells, (b) cells confined to a 64x64 patch per workgroup (L1-friendly),
// (c) the same with 4-byte elements, to separate the address-processing rate from line fills.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename T, int U>
__global__ void __launch_bounds__(1024)
k_gather(const T *__restrict__ table, uint32_t cells, uint32_t W, int iters, int patch, T *__restrict__ out) {
    uint32_t s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    const uint32_t bx = (blockIdx.x * 97u) % (W - 64), by = (blockIdx.x * 61u) % (W - 64);
    T acc = (T)1;
    for (int i = 0; i < iters; i += U) {
        uint32_t idx[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            s = s * 1664525u + 1013904223u;
            const uint32_t r = s >> 8;
            idx[u] = patch ? (by + ((r >> 6) & 63)) * W + bx + (r & 63) : r % cells;
        }
        T v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = table[idx[u]];
#pragma unroll
        for (int u = 0; u < U; u++) acc *= v[u];
    }
    out[blockIdx.x * 1024u + threadIdx.x] = acc;
}

template <typename T>
static void run(const char *name, int patch) {
    const uint32_t W = 2048, cells = W * W;
    T *table, *out;
    CHECK(hipMalloc(&table, (size_t)cells * sizeof(T)));
    const int blocks = 256, iters = 704;             // 256 x 1024 lanes x 704 = the C3 scoring volume x 16
    CHECK(hipMalloc(&out, (size_t)blocks * 1024 * sizeof(T)));
    T *h = (T *)malloc((size_t)cells * sizeof(T));
    for (uint32_t i = 0; i < cells; i++) h[i] = (T)1;
    CHECK(hipMemcpy(table, h, (size_t)cells * sizeof(T), hipMemcpyHostToDevice));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; rep++) k_gather<T, 2><<<blocks, 1024>>>(table, cells, W, iters, patch, out);
    CHECK(hipEventRecord(a));
    const int reps = 20;
    for (int rep = 0; rep < reps; rep++) k_gather<T, 2><<<blocks, 1024>>>(table, cells, W, iters, patch, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double lanes = (double)blocks * 1024 * iters * reps;
    printf("%-44s %7.1f us/launch  %7.1f G lanes/s  %5.2f lanes/clk/CU @2.4GHz\n", name, ms / reps * 1e3, lanes / (ms * 1e-3) / 1e9,
           lanes / (ms * 1e-3) / 256 / 2.4e9);
    hipFree(table); hipFree(out); free(h);
}

int main() {
    run<double>("8-byte gather, random over 32 MiB", 0);
    run<double>("8-byte gather, 64x64 patch per workgroup", 1);
    run<float>("4-byte gather, random over 16 MiB", 0);
    run<float>("4-byte gather, 64x64 patch per workgroup", 1);
    return 0;
}
