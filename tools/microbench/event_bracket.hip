// What does timing a kernel inside a busy in-order stream cost, and what does it read?
//   (a) no events; (b) hipEventRecord before and after every 2nd launch; (c) the same launches through hipExtLaunchKernelGGL
//   with a start and a stop event attached to the dispatch itself (no packets of their own).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/event_bracket.hip -o /tmp/eb && /tmp/eb
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(1024) k_busy(double *out, int iters) {
    double v = 1.0 + threadIdx.x * 1e-9;
    for (int i = 0; i < iters; i++) v = v * 1.0000001 + 1e-9;
    out[blockIdx.x * 1024 + threadIdx.x] = v;
}

int main() {
    double *out; CHECK(hipMalloc(&out, 256 * 1024 * 8));
    hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int steps = 200, per_step = 4, iters[4] = {1300, 900, 250, 300};      // ~20, 14, 5, 6 us: the shape of a C3 scan step
    std::vector<hipEvent_t> ev(2 * steps);
    for (auto &evt : ev) CHECK(hipEventCreate(&evt));
    for (int mode = 0; mode < 3; mode++) {
        for (int pass = 0; pass < 2; pass++) {
            CHECK(hipStreamSynchronize(st));
            auto t0 = std::chrono::steady_clock::now();
            for (int s = 0; s < steps; s++)
                for (int k = 0; k < per_step; k++) {
                    const bool timed = k == 0 && (s & 1) == 0 && mode > 0;
                    if (timed && mode == 1) CHECK(hipEventRecord(ev[2 * s], st));
                    if (timed && mode == 2) hipExtLaunchKernelGGL(k_busy, dim3(256), dim3(1024), 0, st, ev[2 * s], ev[2 * s + 1], 0, out, iters[k]);
                    else hipLaunchKernelGGL(k_busy, dim3(256), dim3(1024), 0, st, out, iters[k]);
                    if (timed && mode == 1) CHECK(hipEventRecord(ev[2 * s + 1], st));
                }
            CHECK(hipStreamSynchronize(st));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / steps;
            if (pass == 1) {
                double sum = 0; int n = 0;
                if (mode > 0)
                    for (int s = 0; s < steps; s += 2) { float ms; CHECK(hipEventElapsedTime(&ms, ev[2 * s], ev[2 * s + 1])); sum += ms * 1e3; n++; }
                printf("%-56s %7.2f us per step", mode == 0 ? "no events" : mode == 1 ? "hipEventRecord around every 2nd first kernel" : "hipExtLaunchKernelGGL events on every 2nd first kernel", us);
                if (n) printf("   timed kernel reads %6.2f us (n = %d)", sum / n, n);
                printf("\n");
            }
        }
    }
    return 0;
}
