// One launch per scan step, or four?  A timing model of the C3 scan step on MI355X (8 XCDs, an L2 each).
//
// The step's dependency chain is  score -> block partials -> fold -> [ray cast | normalise] -> [likelihood | resample];
// the library runs it as four launches.  Here every phase's BODY is a calibrated spin (the time the real kernel's body
// takes once the launch floor is taken off) plus the real hand-off traffic, and the step is timed three ways:
//   L  four launches on one stream (what the library does)
//   P  ONE launch of 256 persistent workgroups x 1024 threads that pull the phases' items from ordered tickets;
//      a workgroup only ever waits for items that a RUNNING workgroup has already claimed, so the launch completes
//      whatever the residency (no grid barrier, no co-residency assumption); hand-offs by the guide's hardware
//      protocol: write-through (sc1) payload stores, s_waitcnt vmcnt(0), workgroup barrier, one relaxed agent-scope
//      ticket or flag, sc1 loads on the consuming side
//   P2 the same with the likelihood | resample phase cut off into a second launch
// Command line: [variant (-1 all, 0 L, 1 P, 2 P2)] [body times in us: S P F R N A L X].
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/persistent_step.hip -o /tmp/ps && /tmp/ps
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define NSEG 16
#define NGRP 16
#define NP (NGRP * 1024)
#define NBLK (NP / 256)
#define N_RAY_ITEMS 180
#define N_NORM_ITEMS 16
#define N_APPLY_ITEMS 64
#define N_LIK_ITEMS 216
#define N_RES_ITEMS 16
#ifndef POLL_SLEEP
#define POLL_SLEEP 4        // s_sleep argument between two polls of a flag (units of 64 clocks)
#endif
#ifndef SPIN_LIMIT
#define SPIN_LIMIT (1u << 15)
#endif

struct Body { float s, p, f, r, n, a, l, x; };

struct Ctl {                       // every contended word on a line of its own
    unsigned head_s[8][32];        // per-XCD heads of the score items
    unsigned head2[32];            // ordered head of everything after the scores
    unsigned grp_done[NGRP][32];
    unsigned fold_cnt[32];
    unsigned f_flag[32];
    unsigned r_done[32];
    unsigned n_done[32];
    unsigned a_done[32];
    unsigned tile_head[32];
    unsigned fault[32];
    double stats[16];
};

__device__ __forceinline__ void spin_us(float us) {
    const uint64_t t0 = wall_clock64();
    const uint64_t ticks = (uint64_t)(us * 100.0f);
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}
__device__ __forceinline__ unsigned ld_u32(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_u32(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_f64(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_f64(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned add_u32(unsigned *p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Every branch around a barrier in these loops is WAVE-UNIFORM (readfirstlane).  A lane-dependent `if (threadIdx.x == 0)`
// in front of a barrier inside a loop is legal HIP, but the compiler may linearise it so that wavefront 0 runs the loop body
// once for lanes 1..63 and once more for lane 0: its s_barrier count then differs from the other wavefronts' and the launch
// hangs (seen with the first version of this file: the claim at the top of the item loop became an outer loop that lane 0
// could never reach).  So wavefront 0 does the single-lane work with ALL its lanes: the same address, the increment in
// lane 0 only (the compiler's atomic optimiser turns that into one atomic), the same value stored by every lane.
__device__ unsigned g_fault;
__device__ unsigned *g_dbg;     // host-pinned progress markers (debug mode), one per workgroup
__device__ unsigned long long *g_ts;   // [256][16] wall-clock stamps (10 ns units) of one launch (mode 8)
#define STAMP(k) do { if (g_ts && threadIdx.x == 0) g_ts[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
__device__ __forceinline__ bool wave0() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0; }
#define MARK(code) do { if (g_dbg && wave0()) __hip_atomic_store(&g_dbg[blockIdx.x], (unsigned)(code), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (0)
// One lane's returning atomic add with the other lanes switched off INSIDE the asm statement (exec is restored before it ends):
// the compiler sees a straight-line statement executed by the whole wavefront.  (With `lane == 0 ? inc : 0` as a per-lane operand it
// issued all 64 lanes' atomics on the one address, ~11 ns each, or serialised them in a 64-iteration scan loop.)
__device__ __forceinline__ unsigned lane0_fetch_add(unsigned *p, unsigned inc) {
    unsigned old;
    unsigned long long saved;
    asm volatile("s_mov_b64 %1, exec\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "global_atomic_add %0, %2, %3, off sc0\n\t"
                 "s_waitcnt vmcnt(0)\n\t"
                 "s_mov_b64 exec, %1"
                 : "=&v"(old), "=&s"(saved) : "v"(p), "v"(inc) : "memory");
    return __builtin_amdgcn_readfirstlane(old);
}
__device__ __forceinline__ void lane0_add(unsigned *p, unsigned inc) {
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "global_atomic_add %1, %2, off\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(saved) : "v"(p), "v"(inc) : "memory");
}
// fetch-and-add by the workgroup: returns the old value to every thread (two barriers)
__device__ __forceinline__ unsigned wg_ticket(unsigned *word, unsigned inc, unsigned *s_slot) {
    if (wave0()) *s_slot = lane0_fetch_add(word, inc);            // (every lane of wavefront 0 writes the same value)
    __syncthreads();
    const unsigned t = __builtin_amdgcn_readfirstlane(*s_slot);
    __syncthreads();
    return t;
}
// one more arrival on a counter that others poll (no value needed)
__device__ __forceinline__ void wg_signal(unsigned *word) {
    if (wave0()) lane0_add(word, 1u);
}
// wavefront 0 polls, the workgroup waits at the barrier behind it
__device__ __forceinline__ void wait_ge(unsigned *word, unsigned want, Ctl *c) {
    if (wave0()) {
        unsigned spins = 0;
        while (__builtin_amdgcn_readfirstlane(ld_u32(word)) < want) {
            __builtin_amdgcn_s_sleep(POLL_SLEEP);
            if (++spins > SPIN_LIMIT) { st_u32(&g_fault, 1u); break; }
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------- four launches
__global__ void __launch_bounds__(1024) kL_score(double *part, Body b) {
    const int i = blockIdx.x, seg = (i & 7) * (NSEG / 8) + (i >> 3) % (NSEG / 8), grp = (i >> 3) / (NSEG / 8);
    spin_us(b.s);
    part[(size_t)seg * NP + grp * 1024 + threadIdx.x] = 1.0 + seg;
}
__global__ void __launch_bounds__(256) kL_partials(const double *part, double *partials, Body b) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NSEG; k++) s += part[(size_t)k * NP + p];
    spin_us(b.p);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ double sw[4];
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x < 9) partials[blockIdx.x * 9 + threadIdx.x] = ((sw[0] + sw[1]) + sw[2]) + sw[3];
}
__global__ void __launch_bounds__(256) kL_norm_raycast(const double *partials, double *out, unsigned *cnt, Body b) {
    double s = 0.0;
    for (int k = threadIdx.x; k < NBLK * 9; k += 256) s += partials[k];        // every workgroup folds the partials itself
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ double sw[4];
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    s = ((sw[0] + sw[1]) + sw[2]) + sw[3];
    if (blockIdx.x < N_RAY_ITEMS) { spin_us(b.r); atomicAdd(&cnt[(blockIdx.x * 256 + threadIdx.x) * 16], 1u); }
    else if (blockIdx.x < N_RAY_ITEMS + 64) { spin_us(b.n * 0.5f); out[(blockIdx.x - N_RAY_ITEMS) * 256 + threadIdx.x] = s; }
    else spin_us(b.a);
}
__global__ void __launch_bounds__(256) kL_lik_resample(const unsigned *cnt, const double *in, double *out, Body b) {
    if (blockIdx.x < 64) { const double v = in[blockIdx.x * 256 + threadIdx.x]; spin_us(b.x); out[NP + blockIdx.x * 256 + threadIdx.x] = v; }
    else { const unsigned v = cnt[((blockIdx.x - 64) * 256 + threadIdx.x) * 16]; spin_us(b.l); out[2 * NP + (blockIdx.x - 64) * 256 + threadIdx.x] = (double)v; }
}

// ---------------------------------------------------------------------------------------------- one launch
// phases: 1 = everything; 2 = stop after the ray cast / normalise (the tail is kL_lik_resample)
__global__ void __launch_bounds__(1024) kP_step(double *part, double *partials, double *out, unsigned *cnt, Ctl *c, Ctl *c_next, Body b,
                                                int phases) {
    __shared__ unsigned s_item;
    __shared__ double sw[16];
    if (blockIdx.x == 0)                                 // the control block of the NEXT launch (nobody uses it during this one)
        for (unsigned i = threadIdx.x; i < sizeof(Ctl) / 4; i += 1024) reinterpret_cast<unsigned *>(c_next)[i] = 0u;
    STAMP(0);
    const unsigned xcd = xcc_id();
    MARK(0x100 + xcd);
    // ---- score items: this XCD's queue first (item i is on XCD i & 7 in the library's mapping); once it is empty ONE look at all
    // eight heads says whether anything is left elsewhere (a failing claim per queue would be eight dependent round trips)
    unsigned q = xcd;
    for (;;) {
        const unsigned t = wg_ticket(&c->head_s[q][0], 1u, &s_item);
        if (t >= NSEG * NGRP / 8) {
            if (wave0()) {
                unsigned h[8], found = 8u;
                for (unsigned k = 1; k < 8; k++) h[k] = ld_u32(&c->head_s[(xcd + k) & 7u][0]);      // seven loads in flight together
                for (unsigned k = 7; k >= 1; k--) if (h[k] < NSEG * NGRP / 8) found = (xcd + k) & 7u;
                s_item = found;
            }
            __syncthreads();
            q = __builtin_amdgcn_readfirstlane(s_item);
            __syncthreads();
            if (q == 8u) break;
            continue;
        }
        const unsigned i = q + 8u * t, seg = (i & 7) * (NSEG / 8) + (i >> 3) % (NSEG / 8), grp = (i >> 3) / (NSEG / 8);
        MARK(0x1000 + i);
        STAMP(1);
        spin_us(b.s);
        STAMP(2);
        st_f64(&part[(size_t)seg * NP + grp * 1024 + threadIdx.x], 1.0 + seg);       // sc1: write-through
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const bool last_of_group = wg_ticket(&c->grp_done[grp][0], 1u, &s_item) == NSEG - 1;
        STAMP(3);
        if (last_of_group) {
            // ---- the group's partials, by the last of its 16 segment workgroups
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < NSEG; k++) s += ld_f64(&part[(size_t)k * NP + grp * 1024 + threadIdx.x]);
            STAMP(4);
            spin_us(b.p);
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
            __syncthreads();
            if ((threadIdx.x & 255) < 9) {
                const int w = (threadIdx.x >> 8) * 4;
                st_f64(&partials[(grp * 4 + (threadIdx.x >> 8)) * 9 + (threadIdx.x & 255)], ((sw[w] + sw[w + 1]) + sw[w + 2]) + sw[w + 3]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            STAMP(5);
            const bool last_group = wg_ticket(&c->fold_cnt[0], 1u, &s_item) == NGRP - 1;
            STAMP(6);
            if (last_group) {
                // ---- the fold, by the last group
                double f = 0.0;
                for (int k = threadIdx.x; k < NBLK * 9; k += 1024) f += ld_f64(&partials[k]);
                for (int o = 32; o > 0; o >>= 1) f += __shfl_xor(f, o, 64);
                __syncthreads();
                if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = f;
                __syncthreads();
                STAMP(7);
                spin_us(b.f);
                if (wave0()) {                              // every lane of wavefront 0 stores the same values
                    double tot = 0.0;
                    for (int w = 0; w < 16; w++) tot += sw[w];
                    st_f64(&c->stats[0], tot); st_f64(&c->stats[1], tot * 0.5);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    st_u32(&c->f_flag[0], 1u);
                }
                STAMP(8);
            }
        }
        if (q != xcd) q = xcd;            // one stolen item at a time
    }
    // ---- everything after the scores, in dependency order: apply (none) | ray cast, normalise (the fold) | resample (normalise)
    STAMP(9);
    const unsigned n2 = N_APPLY_ITEMS + N_RAY_ITEMS + N_NORM_ITEMS + (phases == 1 ? N_RES_ITEMS : 0);
    for (;;) {
        unsigned t = wg_ticket(&c->head2[0], 1u, &s_item);
        if (t >= n2) break;
        MARK(0x5000 + t);
        if (t < N_APPLY_ITEMS) { spin_us(b.a); wg_signal(&c->a_done[0]); continue; }
        t -= N_APPLY_ITEMS;
        if (t < N_RAY_ITEMS + N_NORM_ITEMS) {
            STAMP(10);
            wait_ge(&c->f_flag[0], 1u, c);
            STAMP(11);
            const double sum = ld_f64(&c->stats[0]);
            if (t < N_RAY_ITEMS) {
                spin_us(b.r);
                if (threadIdx.x < 256) atomicAdd(&cnt[(t * 256 + threadIdx.x) * 16], 1u);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                wg_signal(&c->r_done[0]);
                STAMP(12);
            } else {
                spin_us(b.n);
                st_f64(&out[(t - N_RAY_ITEMS) * 1024 + threadIdx.x], sum);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                wg_signal(&c->n_done[0]);
            }
            continue;
        }
        t -= N_RAY_ITEMS + N_NORM_ITEMS;
        wait_ge(&c->n_done[0], N_NORM_ITEMS, c);
        const double v = ld_f64(&out[t * 1024 + threadIdx.x]);
        spin_us(b.x);
        out[NP + t * 1024 + threadIdx.x] = v;
    }
    MARK(0x6000);
    if (phases != 1) return;
    // ---- likelihood tiles: four per workgroup round (256 threads each), after every ray item and the apply pass
    STAMP(13);
    wait_ge(&c->r_done[0], N_RAY_ITEMS, c);
    wait_ge(&c->a_done[0], N_APPLY_ITEMS, c);
    STAMP(14);
    for (;;) {
        const unsigned base = wg_ticket(&c->tile_head[0], 4u, &s_item), t = base + (threadIdx.x >> 8);
        if (base >= N_LIK_ITEMS) break;
        if (t < N_LIK_ITEMS) {
            const unsigned v = __hip_atomic_load(&cnt[(t * 256 + (threadIdx.x & 255)) * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            spin_us(b.l);
            out[2 * NP + t * 256 + (threadIdx.x & 255)] = (double)v;
        }
    }
    STAMP(15);
}

int main(int argc, char **argv) {
    Body b = { 15.4f, 1.0f, 0.5f, 11.0f, 2.0f, 3.0f, 4.5f, 3.0f };
    float *bf = &b.s;
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int only = argc > 1 ? atoi(argv[1]) : -1;          // first argument: -1 all variants, 0 / 1 / 2 one of them
    for (int i = 2; i < argc && i <= 9; i++) bf[i - 2] = (float)atof(argv[i]);
    printf("bodies (us): score %.1f partials %.1f fold %.1f raycast %.1f normalise %.1f apply %.1f likelihood %.1f resample %.1f\n", b.s, b.p, b.f,
           b.r, b.n, b.a, b.l, b.x);
    double *part, *partials, *out;
    unsigned *cnt;
    Ctl *ctl;
    CHECK(hipMalloc(&part, sizeof(double) * NSEG * NP));
    CHECK(hipMalloc(&partials, sizeof(double) * NBLK * 9));
    CHECK(hipMalloc(&out, sizeof(double) * (3 * NP + N_LIK_ITEMS * 256)));
    CHECK(hipMalloc(&cnt, sizeof(unsigned) * 16 * 256 * 256));
    CHECK(hipMalloc(&ctl, sizeof(Ctl) * 2));
    CHECK(hipMemset(cnt, 0, sizeof(unsigned) * 16 * 256 * 256));
    CHECK(hipMemset(ctl, 0, sizeof(Ctl) * 2));
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    if (only == 8) {                                        // time stamps of the hand-offs inside one launch (the third of three)
        unsigned long long *dts = nullptr, *hts = (unsigned long long *)calloc(256 * 16, 8);
        CHECK(hipMalloc(&dts, 256 * 16 * 8));
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipMemset(dts, 0, 256 * 16 * 8));
            CHECK(hipMemset(ctl, 0, sizeof(Ctl) * 2));
            CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_ts), &dts, sizeof(dts)));
            CHECK(hipDeviceSynchronize());
            hipLaunchKernelGGL(kP_step, dim3(256), dim3(1024), 0, st, part, partials, out, cnt, ctl, ctl + 1, b, 1);
            CHECK(hipStreamSynchronize(st));
        }
        CHECK(hipMemcpy(hts, dts, 256 * 16 * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < 256; w++) if (hts[w * 16] && hts[w * 16] < t0) t0 = hts[w * 16];
        const char *nm[16] = { "kernel entered", "score item claimed", "score body done", "group ticket back", "segment products loaded (last of group)",
                               "partials stored + drained", "fold ticket back", "partials loaded (last group)", "stats + flag stored",
                               "left the score loop", "ray/normalise item claimed", "flag seen", "ray item done + signalled", "likelihood phase entered",
                               "all rays + apply seen", "kernel left" };
        for (int k = 0; k < 16; k++) {
            unsigned long long lo = ~0ull, hi = 0; int n = 0;
            for (int w = 0; w < 256; w++) { const unsigned long long v = hts[w * 16 + k]; if (v) { n++; if (v < lo) lo = v; if (v > hi) hi = v; } }
            if (n) printf("%-44s n=%3d  first %7.2f us  last %7.2f us\n", nm[k], n, (lo - t0) * 0.01, (hi - t0) * 0.01);
        }
        return 0;
    }
    if (only == 9) {                                        // debug: one launch, then the control block
        unsigned *hdbg = nullptr;
        CHECK(hipHostMalloc(&hdbg, 256 * 4, hipHostMallocMapped));
        for (int i = 0; i < 256; i++) hdbg[i] = 0;
        CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &hdbg, sizeof(hdbg)));
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(kP_step, dim3(256), dim3(1024), 0, st, part, partials, out, cnt, ctl, ctl + 1, b, 2);
        for (int w = 0; w < 30 && hipStreamQuery(st) != hipSuccess; w++) usleep(100000);
        if (hipStreamQuery(st) != hipSuccess) {
            printf("kernel still running after 3 s; markers per workgroup:\n");
            for (int i = 0; i < 256; i++) printf("%x%c", hdbg[i], (i & 15) == 15 ? '\n' : ' ');
            fflush(stdout);
            _exit(3);
        }
        Ctl h;
        CHECK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
        unsigned fault = 0;
        CHECK(hipMemcpyFromSymbol(&fault, HIP_SYMBOL(g_fault), 4));
        printf("head_s:"); for (int i = 0; i < 8; i++) printf(" %u", h.head_s[i][0]);
        printf("\nhead2 %u fold_cnt %u f_flag %u r_done %u n_done %u a_done %u fault %u stats %g\ngrp_done:", h.head2[0], h.fold_cnt[0], h.f_flag[0],
               h.r_done[0], h.n_done[0], h.a_done[0], fault, h.stats[0]);
        for (int i = 0; i < NGRP; i++) printf(" %u", h.grp_done[i][0]);
        printf("\n");
        return 0;
    }
    const int steps = 200;
    for (int variant = 0; variant < 3; variant++) {
        if (only >= 0 && variant != only) continue;
        for (int rep = 0; rep < 3; rep++) {
            int k = 0;
            for (int i = 0; i < steps + 20; i++) {
                if (i == 20) CHECK(hipEventRecord(e0, st));
                if (variant == 0) {
                    hipLaunchKernelGGL(kL_score, dim3(256), dim3(1024), 0, st, part, b);
                    hipLaunchKernelGGL(kL_partials, dim3(NBLK), dim3(256), 0, st, part, partials, b);
                    hipLaunchKernelGGL(kL_norm_raycast, dim3(N_RAY_ITEMS + 64 + 64), dim3(256), 0, st, partials, out, cnt, b);
                    hipLaunchKernelGGL(kL_lik_resample, dim3(64 + N_LIK_ITEMS), dim3(256), 0, st, cnt, out, out, b);
                } else {
                    hipLaunchKernelGGL(kP_step, dim3(256), dim3(1024), 0, st, part, partials, out, cnt, ctl + (k & 1), ctl + ((k + 1) & 1), b,
                                       variant == 1 ? 1 : 2);
                    k++;
                    if (variant == 2) hipLaunchKernelGGL(kL_lik_resample, dim3(64 + N_LIK_ITEMS), dim3(256), 0, st, cnt, out, out, b);
                }
            }
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned fault = 0;
            CHECK(hipMemcpyFromSymbol(&fault, HIP_SYMBOL(g_fault), 4));
            printf("%s  %.2f us per step%s\n", variant == 0 ? "L  four launches      " : (variant == 1 ? "P  one launch         " : "P2 two launches       "),
                   ms * 1e3 / steps, fault ? "  (SPIN LIMIT HIT)" : "");
        }
    }
    const float chain = b.s + b.p + b.f + b.r + b.l;
    printf("critical chain of the bodies alone: %.1f us\n", chain);
    return 0;
}
