// Clocks per step of RayIterator.next's float recurrence (RayIterator.java:117-123) for one wavefront per SIMD, by instruction order.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/dda_chain.hip -o /tmp/dda && /tmp/dda
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define STEPS 640
#define S0 "v_cmp_lt_f32_e32 vcc, 0, %0\n\tv_cndmask_b32_e32 %2, %3, %4, vcc\n\tv_add_f32_e32 %0, %0, %2\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc\n\t"
#define S1 "v_cmp_lt_f32_e32 vcc, 0, %0\n\tv_cndmask_b32_e32 %2, %3, %4, vcc\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc\n\tv_add_f32_e32 %0, %0, %2\n\t"
#define R4(x) x x x x
#define R32(x) R4(x) R4(x) R4(x) R4(x) R4(x) R4(x) R4(x) R4(x)

template <int V>
__global__ void __launch_bounds__(256) k(float *out, uint32_t *words, uint64_t *clk, float dyv, float ndxv) {
    float err = 0.25f + threadIdx.x * 1e-3f, t, t2;
    float dy = dyv, ndx = ndxv;
    asm volatile("" : "+v"(ndx), "+v"(dy));
    uint32_t word = 0, acc = 0;
    const bool producer = (threadIdx.x & 63) < 4;              // four lanes per wavefront, as in the library; every wavefront of the workgroup runs it (one per SIMD)
    const uint64_t w0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    if (producer) {
        for (int w = 0; w < STEPS / 32; w++) {
            if (V == 6) { asm volatile(R32(S0) : "+v"(err), "+v"(word), "=&v"(t) : "v"(dy), "v"(ndx) : "vcc"); acc += __brev(word); continue; }
            if (V == 7) { asm volatile(R32(S1) : "+v"(err), "+v"(word), "=&v"(t) : "v"(dy), "v"(ndx) : "vcc"); acc += __brev(word); continue; }
#pragma unroll
            for (int j = 0; j < 32; j++) {
                if (V == 0)        // the library's order
                    asm volatile("v_cmp_lt_f32_e32 vcc, 0, %0\n\tv_cndmask_b32_e32 %2, %3, %4, vcc\n\tv_add_f32_e32 %0, %0, %2\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc"
                                 : "+v"(err), "+v"(word), "=&v"(t) : "v"(dy), "v"(ndx) : "vcc");
                else if (V == 1)   // the decision bit shifted in between the select and the add
                    asm volatile("v_cmp_lt_f32_e32 vcc, 0, %0\n\tv_cndmask_b32_e32 %2, %3, %4, vcc\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc\n\tv_add_f32_e32 %0, %0, %2"
                                 : "+v"(err), "+v"(word), "=&v"(t) : "v"(dy), "v"(ndx) : "vcc");
                else if (V == 2)   // ... between the compare and the select (carry-out to a scratch SGPR pair)
                    asm volatile("v_cmp_lt_f32_e32 vcc, 0, %0\n\tv_addc_co_u32_e64 %1, s[20:21], %1, %1, vcc\n\tv_cndmask_b32_e32 %2, %3, %4, vcc\n\tv_add_f32_e32 %0, %0, %2"
                                 : "+v"(err), "+v"(word), "=&v"(t) : "v"(dy), "v"(ndx) : "vcc", "s20", "s21");
                else if (V == 3)   // no decision word (wrong): three instructions
                    asm volatile("v_cmp_lt_f32_e32 vcc, 0, %0\n\tv_cndmask_b32_e32 %2, %3, %4, vcc\n\tv_add_f32_e32 %0, %0, %2"
                                 : "+v"(err), "+v"(word), "=&v"(t) : "v"(dy), "v"(ndx) : "vcc");
                else if (V == 4)   // both successors, one select
                    asm volatile("v_cmp_lt_f32_e32 vcc, 0, %0\n\tv_add_f32_e32 %2, %0, %5\n\tv_add_f32_e32 %3, %0, %4\n\tv_cndmask_b32_e32 %0, %3, %2, vcc\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc"
                                 : "+v"(err), "+v"(word), "=&v"(t), "=&v"(t2) : "v"(dy), "v"(ndx) : "vcc");
                else if (V == 5)   // two rays' recurrences interleaved in one lane (independent chains fill each other's bubbles): per PAIR of steps
                    asm volatile("v_cmp_lt_f32_e32 vcc, 0, %0\n\tv_cndmask_b32_e32 %2, %4, %5, vcc\n\tv_add_f32_e32 %0, %0, %2\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc\n\t"
                                 "v_cmp_lt_f32_e32 vcc, 0, %3\n\tv_cndmask_b32_e32 %2, %4, %5, vcc\n\tv_add_f32_e32 %3, %3, %2\n\tv_addc_co_u32_e32 %6, vcc, %6, %6, vcc"
                                 : "+v"(err), "+v"(word), "=&v"(t), "+v"(t2) : "v"(dy), "v"(ndx), "v"(acc) : "vcc");
            }
            acc += __brev(word);
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    const uint64_t w1 = wall_clock64();
    if (producer) { out[blockIdx.x * 256 + threadIdx.x] = err; words[blockIdx.x * 256 + threadIdx.x] = acc; }
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

__global__ void k_warm(float *out, int iters) {
    float v = threadIdx.x;
    for (int i = 0; i < iters; i++) v = v * 1.000001f + 0.5f;
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

int main() {
    float *out; uint32_t *words; uint64_t *clk, h[360];
    CHECK(hipMalloc(&out, 4 * 1024 * 1024)); CHECK(hipMalloc(&words, 4 * 256 * 256)); CHECK(hipMalloc(&clk, 16 * 180));
    const char *nm[8] = { "cmp, cndmask, add, addc (library)", "cmp, cndmask, addc, add", "cmp, addc(e64), cndmask, add", "cmp, cndmask, add (no word: wrong)",
                          "cmp, add, add, cndmask, addc", "two interleaved chains (per step pair)", "library order, 32 steps in ONE asm statement", "cmp, cndmask, addc, add; 32 steps in one asm" };
    hipLaunchKernelGGL(k_warm, dim3(1024), dim3(1024), 0, 0, out, 400000);       // bring the clocks up first
    CHECK(hipDeviceSynchronize());
    double best_t[8], best_w[8];
    for (int v = 0; v < 8; v++) best_t[v] = best_w[v] = 1e30;
    for (int round = 0; round < 6; round++) {
        for (int v = 0; v < 8; v++) {
            switch (v) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(180), dim3(256), 0, 0, out, words, clk, 0.37f, -1.21f); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(180), dim3(256), 0, 0, out, words, clk, 0.37f, -1.21f); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(180), dim3(256), 0, 0, out, words, clk, 0.37f, -1.21f); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(180), dim3(256), 0, 0, out, words, clk, 0.37f, -1.21f); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(180), dim3(256), 0, 0, out, words, clk, 0.37f, -1.21f); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(180), dim3(256), 0, 0, out, words, clk, 0.37f, -1.21f); break;
                case 6: hipLaunchKernelGGL(k<6>, dim3(180), dim3(256), 0, 0, out, words, clk, 0.37f, -1.21f); break;
                default: hipLaunchKernelGGL(k<7>, dim3(180), dim3(256), 0, 0, out, words, clk, 0.37f, -1.21f); break;
            }
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(h, clk, 16 * 180, hipMemcpyDeviceToHost));
            for (int b = 0; b < 180; b++) {          // the fastest workgroup: one that had its compute unit to itself
                if ((double)h[2 * b] < best_t[v]) best_t[v] = (double)h[2 * b];
                if ((double)h[2 * b + 1] < best_w[v]) best_w[v] = (double)h[2 * b + 1];
            }
        }
    }
    {   // the one-statement forms must walk exactly like the library's: same decision words, same final error
        static float e0[180 * 256], e1[180 * 256];
        static uint32_t w0[180 * 256], w1[180 * 256];
        hipLaunchKernelGGL(k<0>, dim3(180), dim3(256), 0, 0, out, words, clk, 0.37f, -1.21f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(e0, out, sizeof(e0), hipMemcpyDeviceToHost)); CHECK(hipMemcpy(w0, words, sizeof(w0), hipMemcpyDeviceToHost));
        for (int v = 6; v < 8; v++) {
            if (v == 6) hipLaunchKernelGGL(k<6>, dim3(180), dim3(256), 0, 0, out, words, clk, 0.37f, -1.21f);
            else hipLaunchKernelGGL(k<7>, dim3(180), dim3(256), 0, 0, out, words, clk, 0.37f, -1.21f);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(e1, out, sizeof(e1), hipMemcpyDeviceToHost)); CHECK(hipMemcpy(w1, words, sizeof(w1), hipMemcpyDeviceToHost));
            int bad = 0;
            for (int i = 0; i < 180 * 256; i++) if ((i & 63) < 4 && (w0[i] != w1[i] || e0[i] != e1[i])) bad++;
            printf("variant %d vs the library's form: %d of %d lanes differ\n", v, bad, 180 * 4 * 4);
        }
    }
    for (int v = 0; v < 8; v++)
        printf("%-44s %6.2f s_memtime ticks per step, %6.2f ns per step (100 MHz wall clock over %d steps)\n", nm[v], best_t[v] / STEPS, best_w[v] * 10.0 / STEPS, STEPS);
    return 0;
}
