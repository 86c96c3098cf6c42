// What does the texture-address pipe charge for an 8-byte gather: one slot per DISTINCT 128-byte line in the
// wavefront, or one per lane unless NEIGHBOURING lanes share a line?  Decides whether re-ordering the particles (or
// tiling the table) could lift the scoring kernel's ceiling.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/gather_coalesce.hip -o /tmp/gc && /tmp/gc
// (pw, ph, g are powers of two: the index arithmetic must stay far below the address pipe's cost)
// mode 0: every lane its own random line inside a 64x64 patch (the gather8 ceiling case)
// mode 1: lanes draw random cells from a SMALL patch (pw x ph cells), so a wavefront touches few distinct lines,
//         but which lanes share a line is random
// mode 2: groups of g neighbouring lanes read neighbouring cells of one line; groups are random in the 64x64 patch
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(1024)
k_gather(const double *__restrict__ table, uint32_t W, int iters, int mode, int pw, int ph, int g, double *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t sg = (blockIdx.x * 1024u + (threadIdx.x & ~(uint32_t)(g - 1))) * 2654435761u + 999u;   // shared by a lane group
    const uint32_t bx = ((blockIdx.x * 97u) % (W - 64)) & ~15u, by = (blockIdx.x * 61u) % (W - 64);
    double acc = 1.0;
    for (int i = 0; i < iters; i += 2) {
        uint32_t idx[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            s = s * 1664525u + 1013904223u;
            sg = sg * 1664525u + 1013904223u;
            const uint32_t r = s >> 8, rg = sg >> 8;
            if (mode == 0) idx[u] = (by + ((r >> 6) & 63)) * W + bx + (r & 63);
            else if (mode == 1) idx[u] = (by + ((r >> 6) & (uint32_t)(ph - 1))) * W + bx + (r & (uint32_t)(pw - 1));
            else idx[u] = (by + ((rg >> 6) & 63)) * W + bx + ((rg & 3) << 4) + (lane & (uint32_t)(g - 1));
        }
        double v[2];
#pragma unroll
        for (int u = 0; u < 2; u++) v[u] = table[idx[u]];
#pragma unroll
        for (int u = 0; u < 2; u++) acc *= v[u];
    }
    out[blockIdx.x * 1024u + threadIdx.x] = acc;
}

static double *table, *out;
static void run(const char *name, int mode, int pw, int ph, int g) {
    const uint32_t W = 2048;
    const int blocks = 256, iters = 704;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; rep++) k_gather<<<blocks, 1024>>>(table, W, iters, mode, pw, ph, g, out);
    CHECK(hipEventRecord(a));
    const int reps = 20;
    for (int rep = 0; rep < reps; rep++) k_gather<<<blocks, 1024>>>(table, W, iters, mode, pw, ph, g, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double lanes = (double)blocks * 1024 * iters * reps;
    printf("%-58s %7.1f us  %7.1f G lanes/s  %5.2f lanes/clk/CU  (%.1f clk per wave instruction)\n", name, ms / reps * 1e3,
           lanes / (ms * 1e-3) / 1e9, lanes / (ms * 1e-3) / 256 / 2.4e9, 64.0 / (lanes / (ms * 1e-3) / 256 / 2.4e9));
}

int main() {
    const uint32_t W = 2048, cells = W * W;
    CHECK(hipMalloc(&table, (size_t)cells * 8));
    CHECK(hipMalloc(&out, (size_t)256 * 1024 * 8));
    double *h = (double *)malloc((size_t)cells * 8);
    for (uint32_t i = 0; i < cells; i++) h[i] = 1.0;
    CHECK(hipMemcpy(table, h, (size_t)cells * 8, hipMemcpyHostToDevice));
    run("mode 0: 64 lanes, 64x64 patch (~64 lines)", 0, 0, 0, 1);
    run("mode 1: random lanes in 32x32 cells (~45 lines)", 1, 32, 32, 1);
    run("mode 1: random lanes in 16x32 cells (32 lines)", 1, 16, 32, 1);
    run("mode 1: random lanes in 16x16 cells (16 lines)", 1, 16, 16, 1);
    run("mode 1: random lanes in 16x8 cells (8 lines)", 1, 16, 8, 1);
    run("mode 1: random lanes in 16x2 cells (2 lines)", 1, 16, 2, 1);
    run("mode 1: random lanes in 16x1 cells (1 line)", 1, 16, 1, 1);
    run("mode 1: random lanes in 4x1 cells (one 32-byte sector)", 1, 4, 1, 1);
    run("mode 1: all lanes one cell", 1, 1, 1, 1);
    run("mode 2: pairs of neighbour lanes share a line (32 lines)", 2, 0, 0, 2);
    run("mode 2: quads of neighbour lanes share a line (16 lines)", 2, 0, 0, 4);
    run("mode 2: 8 neighbour lanes share a line (8 lines)", 2, 0, 0, 8);
    run("mode 2: 16 neighbour lanes share a line (4 lines)", 2, 0, 0, 16);
    return 0;
}
