// Issue cost of the double-precision instructions the likelihood passes are made of (development tool): clocks per wavefront
// instruction on one SIMD with 1, 2 and 4 wavefronts resident, eight independent chains per lane.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o fp64_rate fp64_rate.hip && ./fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 256
template <int OP>
__global__ void k(double *out, long long *clk, double a, double b) {
    double x[8];
    for (int i = 0; i < 8; i++) x[i] = a + i + threadIdx.x * 1e-9;
    uint32_t u = threadIdx.x;
    __syncthreads();
    const long long t0 = clock64();
#pragma unroll 1
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) x[i] = x[i] * b;
            if (OP == 1) x[i] = x[i] + b;
            if (OP == 2) x[i] = __builtin_fma(x[i], b, a);
            if (OP == 3) { asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(x[i]) : "v"(u + i)); }
            if (OP == 4) { float f = (float)x[i]; asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f) : "v"((float)b)); x[i] = f; }
            if (OP == 5) { x[i] = x[i] * b; x[i] = x[i] + a; }
        }
    }
    const long long t1 = clock64();
    double s = 0; for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
int main() {
    double *out; long long *clk;
    hipMalloc(&out, 4096 * sizeof(double)); hipMalloc(&clk, 64 * sizeof(long long));
    const char *names[] = {"v_mul_f64", "v_add_f64", "v_fma_f64", "v_cvt_f64_u32", "f32 mul (+2 cvt)", "mul then add f64"};
    for (int op = 0; op < 6; op++) {
        for (int threads : {256, 512, 1024}) {
            long long c = 0;
            for (int rep = 0; rep < 3; rep++) {
                switch (op) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, out, clk, 1.0000001, 0.9999999); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, out, clk, 1.0000001, 0.9999999); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(1), dim3(threads), 0, 0, out, clk, 1.0000001, 0.9999999); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(1), dim3(threads), 0, 0, out, clk, 1.0000001, 0.9999999); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(1), dim3(threads), 0, 0, out, clk, 1.0000001, 0.9999999); break;
                    case 5: hipLaunchKernelGGL(k<5>, dim3(1), dim3(threads), 0, 0, out, clk, 1.0000001, 0.9999999); break;
                }
                hipDeviceSynchronize();
                hipMemcpy(&c, clk, sizeof(c), hipMemcpyDeviceToHost);
            }
            const int waves_per_simd = threads / 256;
            const double per = (double)c / (REP * 8.0 * (op == 5 ? 2 : 1)) / waves_per_simd;
            printf("%-18s %4d threads (%d wavefront(s) per SIMD): %7lld ticks of clock64 for %d x 8 ops per wavefront -> %.2f ticks per wavefront instruction and SIMD\n",
                   names[op], threads, waves_per_simd, c, REP, per);
        }
    }
    return 0;
}
