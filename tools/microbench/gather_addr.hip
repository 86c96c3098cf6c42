// Does the FORM of the gather instruction change what the texture-address pipe charges for 64 independent 8-byte look-ups?
//   form 0  global_load_dwordx2 v, v[addr64], off            (64-bit address per lane: what the compiler emits for table[idx])
//   form 1  global_load_dwordx2 v, v_off32, s[base]          (scalar base + 32-bit byte offset per lane)
//   form 2  buffer_load_dwordx2 v, v_off32, s[rsrc], 0 offen (buffer resource + 32-bit byte offset per lane)
//   form 3  form 2 with a 4-byte element (buffer_load_dword): half the bytes returned
// Access pattern: every lane its own random cell inside a 64x64 patch of the workgroup (the scoring kernel's case), and the
// same with quads of neighbouring lanes on one line.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/gather_addr.hip -o /tmp/ga && /tmp/ga
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int FORM>
__global__ void __launch_bounds__(1024)
k_gather(const double *__restrict__ table, uint32_t W, int iters, int g, double *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t sg = (blockIdx.x * 1024u + (threadIdx.x & ~(uint32_t)(g - 1))) * 2654435761u + 999u;   // shared by a lane group
    const uint32_t bx = ((blockIdx.x * 97u) % (W - 64)) & ~15u, by = (blockIdx.x * 61u) % (W - 64);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(table), 0, (int)0x7fffffff, 0x00020000);
    double acc = 1.0;
    for (int i = 0; i < iters; i += 2) {
        uint32_t idx[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            sg = sg * 1664525u + 1013904223u;
            const uint32_t rg = sg >> 8;
            idx[u] = g == 1 ? (by + ((rg >> 6) & 63)) * W + bx + (rg & 63)
                            : (by + ((rg >> 6) & 63)) * W + bx + ((rg & 3) << 4) + (lane & (uint32_t)(g - 1));
        }
        double v[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (FORM == 0) {
                const double *p = table + idx[u];
                asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v[u]) : "v"(p) : "memory");
            } else if (FORM == 1) {
                const uint32_t off = idx[u] * 8u;
                asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v[u]) : "v"(off), "s"(table) : "memory");
            } else if (FORM == 2) {
                const auto r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(idx[u] * 8u), 0, 0);
                v[u] = __builtin_bit_cast(double, r);
            } else {
                const uint32_t r = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(idx[u] * 8u), 0, 0);
                v[u] = (double)__uint_as_float(r);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 2; u++) acc *= v[u];
    }
    out[blockIdx.x * 1024u + threadIdx.x] = acc;
}

static double *table, *out;
template <int FORM>
static void run(const char *name, int g) {
    const uint32_t W = 2048;
    const int blocks = 256, iters = 704;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; rep++) k_gather<FORM><<<blocks, 1024>>>(table, W, iters, g, out);
    CHECK(hipEventRecord(a));
    const int reps = 20;
    for (int rep = 0; rep < reps; rep++) k_gather<FORM><<<blocks, 1024>>>(table, W, iters, g, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double lanes = (double)blocks * 1024 * iters * reps;
    CHECK(hipDeviceSynchronize());
    printf("%-64s g=%2d %7.1f us  %7.1f G lanes/s  (%.1f clk per wave instruction)\n", name, g, ms / reps * 1e3,
           lanes / (ms * 1e-3) / 1e9, 64.0 / (lanes / (ms * 1e-3) / 256 / 2.4e9));
    fflush(stdout);
}

int main() {
    const uint32_t W = 2048, cells = W * W;
    CHECK(hipMalloc(&table, (size_t)cells * 8));
    CHECK(hipMalloc(&out, (size_t)256 * 1024 * 8));
    double *h = (double *)malloc((size_t)cells * 8);
    for (uint32_t i = 0; i < cells; i++) h[i] = 1.0;
    CHECK(hipMemcpy(table, h, (size_t)cells * 8, hipMemcpyHostToDevice));
    for (int g = 1; g <= 4; g *= 4) {
        run<0>("form 0: global_load_dwordx2, 64-bit address per lane", g);
        run<1>("form 1: global_load_dwordx2, scalar base + 32-bit offset", g);
        run<2>("form 2: buffer_load_dwordx2 offen", g);
        run<3>("form 3: buffer_load_dword offen (4 bytes)", g);
    }
    return 0;
}
