// gms_pf_kernels.hip -- gfx950 kernels for the particle scan matcher.
//
//   k_pf_prep / k_compact_beams  per-particle float-rounded trig (J/math/Transform.java:15-16) and
//                                the list of beams with wasHit (GridMap.java:269)
//   k_score        GridMap.probabilityOf for every particle (J/slam/GridMap.java:261-294): one
//                  wavefront per particle, lanes stride the beams, beam table in LDS, likelihood
//                  gathers from L2 / Infinity Cache, product reduced with a wave64 xor-butterfly.
//   k_partials ... SLAM.update's weight bookkeeping (J/slam/SLAM.java:87-129), calculateNeff
//                  (:180-190) and getWeightedPose (:165-178) as fixed-shape blocked reductions:
//                  blocks of GMS_BLOCK particles by GLOBAL index, so the result does not depend on
//                  how many GPUs the particles are sharded over.
//   k_scan_* / k_resample  SLAM.resample (:133-153): chunked sequential cumulative sums + one
//                  search per output slot.
//
// HBM layout: poses SoA x[n], y[n], theta[n] (float); weights double[n]; all [n_maps][n].
#include "gms_device.h"

#define SCAN_CHUNK 64

// ---------------------------------------------------------------------------------------------
__global__ void k_pf_init(float *x, float *y, float *th, double *w, double *logw, int64_t total, double w0) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) { x[i] = 0.0f; y[i] = 0.0f; th[i] = 0.0f; w[i] = w0; logw[i] = 0.0; }
}

__global__ void k_pf_prep(const float *__restrict__ th, float *__restrict__ cs, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        float c, s;
        pose_trig(th[i], c, s);
        cs[2 * i] = c;
        cs[2 * i + 1] = s;
    }
}

// one wave per map: order-preserving compaction of the beams with wasHit
__global__ void __launch_bounds__(64)
k_compact_beams(const gms_beam *__restrict__ beams, int32_t B, int32_t beam_stride, int32_t out_stride,
                double *__restrict__ hitbeams, int32_t *__restrict__ nhit) {
    const int32_t mi = blockIdx.x;
    const int32_t lane = threadIdx.x;
    const gms_beam *mb = beams + (size_t)mi * beam_stride;
    double *out = hitbeams + (size_t)mi * out_stride * 2;
    int32_t base = 0;
    for (int32_t b0 = 0; b0 < B; b0 += 64) {
        const int32_t b = b0 + lane;
        const bool hit = b < B && mb[b].hit != 0;
        const unsigned long long mask = __ballot(hit);
        if (hit) {
            const int32_t pos = base + __popcll(mask & ((1ull << lane) - 1ull));
            out[2 * pos] = mb[b].local_x;
            out[2 * pos + 1] = mb[b].local_y;
        }
        base += __popcll(mask);
    }
    if (lane == 0) nhit[mi] = base;
}

// ---------------------------------------------------------------------------------------------
// probabilityOf: factor of one beam end point (GridMap.java:273-288)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double beam_factor(const GridDev &g, const double *__restrict__ lik, const XformDev &t,
                                              double lx, double ly) {
    const int32_t gx = j_d2i((xform_x(t, lx, ly) - g.posx) / g.res);   // :273
    const int32_t gy = j_d2i((xform_y(t, lx, ly) - g.posy) / g.res);   // :274
    if (!(gx < 0 || gy < 0 || gx >= g.W || gy >= g.H)) {               // :276
        const double val = lik[(size_t)gy * g.W + gx];                 // :277
        return val == 0.5 ? g.inv_max : g.z_hit * val + g.c_rand;      // :285-288
    }
    return 1.0;   // beam skipped: the product is left alone
}

// multiply two (mantissa, exponent) products; scaling by powers of two is exact, so the mantissa
// path rounds exactly like the plain product while it stays clear of the denormal range
__device__ __forceinline__ void mx_mul(double &m, int32_t &e, double m2, int32_t e2) {
    m *= m2;
    e += e2;
    int de;
    m = frexp(m, &de);
    e += de;
}

__global__ void __launch_bounds__(256)
k_score(GridDev g, const double *__restrict__ lik_all, const double *__restrict__ hitbeams,
        const int32_t *__restrict__ nhit, int32_t beam_stride, const float *__restrict__ px,
        const float *__restrict__ py, const float *__restrict__ cs, int32_t n, double *__restrict__ w,
        double *__restrict__ logw) {
    extern __shared__ __align__(16) unsigned char smem[];
    double2 *sb = reinterpret_cast<double2 *>(smem);
    const int32_t mi = blockIdx.y;
    const int32_t nb = nhit[mi];
    const double2 *hb = reinterpret_cast<const double2 *>(hitbeams + (size_t)mi * beam_stride * 2);
    for (int32_t i = threadIdx.x; i < nb; i += blockDim.x) sb[i] = hb[i];
    __syncthreads();

    const double *lik = lik_all + (size_t)mi * g.cells;
    const int32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    for (int32_t p = blockIdx.x * wpb + wave; p < n; p += gridDim.x * wpb) {
        const size_t gi = (size_t)mi * n + p;
        XformDev t;
        t.c = (double)cs[2 * gi]; t.s = (double)cs[2 * gi + 1];       // Transform.java:15-16
        t.px = (double)px[gi];    t.py = (double)py[gi];
        double prod = 1.0;                                             // GridMap.java:262
#pragma unroll 4
        for (int32_t j = lane; j < nb; j += 64) {
            const double2 bm = sb[j];
            prod *= beam_factor(g, lik, t, bm.x, bm.y);
        }
        int e;
        double mnt = frexp(prod, &e);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double m2 = __shfl_xor(mnt, o, GMS_WAVE);
            const int32_t e2 = __shfl_xor(e, o, GMS_WAVE);
            mx_mul(mnt, e, m2, e2);
        }
        if (lane == 0) {
            w[gi] = ldexp(mnt, e);
            logw[gi] = log(mnt) + (double)e * 0.6931471805599453;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// blocked reductions.  Block sum shape (identical everywhere): 64-lane xor butterfly per wave,
// then ((w0 + w1) + w2) + w3 over the four waves.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_256(double v, double *lds4) {
    v = wave_sum_f64(v);
    const int32_t wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds4[wave] = v;
    __syncthreads();
    return ((lds4[0] + lds4[1]) + lds4[2]) + lds4[3];
}

// (value, index) max with "first maximum wins" (SLAM.java:110-115: strict >); NaN never wins
__device__ __forceinline__ void argmax_merge(double &v, double &i, double v2, double i2) {
    if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
}

__device__ __forceinline__ void block_argmax_256(double &v, double &i, double *ldsv, double *ldsi) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double v2 = __shfl_xor(v, o, GMS_WAVE), i2 = __shfl_xor(i, o, GMS_WAVE);
        argmax_merge(v, i, v2, i2);
    }
    const int32_t wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { ldsv[wave] = v; ldsi[wave] = i; }
    __syncthreads();
    v = ldsv[0]; i = ldsi[0];
    for (int k = 1; k < 4; k++) argmax_merge(v, i, ldsv[k], ldsi[k]);
}

// phase 1: this shard's block partials {sum, max, argmax, n_zero, max_logw} at their global slots
__global__ void __launch_bounds__(256)
k_partials(const double *__restrict__ w, const double *__restrict__ logw, int32_t n, int64_t offset,
           int64_t nblk_global, double *__restrict__ partials) {
    __shared__ double l4[4], lv[4], li[4];
    const int32_t mi = blockIdx.y;
    const int32_t i = blockIdx.x * GMS_BLOCK + threadIdx.x;
    const bool in = i < n;
    const double v = in ? w[(size_t)mi * n + i] : 0.0;
    const double lw = in ? logw[(size_t)mi * n + i] : -INFINITY;
    const double s = block_sum_256(v, l4);
    double mv = (in && v == v) ? v : -INFINITY;
    double mx = (double)(offset + i);
    block_argmax_256(mv, mx, lv, li);
    const double nz = block_sum_256((in && v == 0.0) ? 1.0 : 0.0, l4);
    double ml = (lw == lw) ? lw : -INFINITY, mli = 0.0;
    block_argmax_256(ml, mli, lv, li);
    if (threadIdx.x == 0) {
        const int64_t gb = offset / GMS_BLOCK + blockIdx.x;
        double *p = partials + ((size_t)mi * nblk_global + gb) * GMS_PARTIAL_STRIDE;
        p[0] = s; p[1] = mv; p[2] = mx; p[3] = nz; p[4] = ml;
    }
}

// every block folds the (all-reduced) partials in block order: thread t takes partials t, t+256, ...
// sequentially, then the block shape above.  Deterministic for a given nblk_global.
__device__ __forceinline__ void fold_partials(const double *__restrict__ p, int64_t nblk, int stride,
                                              int ncols, double *out, double *l4) {
    for (int c = 0; c < ncols; c++) {
        double acc = 0.0;
        for (int64_t b = threadIdx.x; b < nblk; b += GMS_BLOCK) acc += p[b * stride + c];
        out[c] = block_sum_256(acc, l4);
    }
}

// phase 2: weightSum, strongest; weight /= weightSum; pack {w,x,y,theta} for the all-gather
__global__ void __launch_bounds__(256)
k_normalize_pack(const double *__restrict__ partials_all, int64_t nblk_global, double *__restrict__ w,
                 const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ th, int32_t n,
                 int64_t packed_stride, PackedParticle *__restrict__ packed, PfStatsDev *__restrict__ stats) {
    __shared__ double l4[4], lv[4], li[4];
    const int32_t mi = blockIdx.y;
    const double *p = partials_all + (size_t)mi * nblk_global * GMS_PARTIAL_STRIDE;
    double sum;
    fold_partials(p, nblk_global, GMS_PARTIAL_STRIDE, 1, &sum, l4);
    if (blockIdx.x == 0) {
        double nz;
        {
            double acc = 0.0;
            for (int64_t b = threadIdx.x; b < nblk_global; b += GMS_BLOCK) acc += p[b * GMS_PARTIAL_STRIDE + 3];
            nz = block_sum_256(acc, l4);
        }
        double mv = -INFINITY, mx = 0.0;
        bool first = true;
        for (int64_t b = threadIdx.x; b < nblk_global; b += GMS_BLOCK) {
            const double v2 = p[b * GMS_PARTIAL_STRIDE + 1], i2 = p[b * GMS_PARTIAL_STRIDE + 2];
            if (first) { mv = v2; mx = i2; first = false; } else argmax_merge(mv, mx, v2, i2);
        }
        if (first) { mv = -INFINITY; mx = 9.0e15; }
        block_argmax_256(mv, mx, lv, li);
        double ml = -INFINITY, mli = 0.0;
        for (int64_t b = threadIdx.x; b < nblk_global; b += GMS_BLOCK) {
            const double v2 = p[b * GMS_PARTIAL_STRIDE + 4];
            if (v2 > ml) ml = v2;
        }
        block_argmax_256(ml, mli, lv, li);
        if (threadIdx.x == 0) {
            PfStatsDev *s = stats + mi;
            s->weight_sum = sum;
            s->max_w = mv;
            s->strongest = (int32_t)mx;
            s->n_zero = (int32_t)nz;
            s->max_logw = ml;
        }
    }
    const int32_t i = blockIdx.x * GMS_BLOCK + threadIdx.x;
    if (i < n) {
        const size_t gi = (size_t)mi * n + i;
        const double wn = w[gi] / sum;                                 // SLAM.java:120-121
        w[gi] = wn;
        PackedParticle pp;
        pp.w = wn; pp.x = x[gi]; pp.y = y[gi]; pp.theta = th[gi]; pp.pad = 0u;
        packed[(size_t)mi * packed_stride + i] = pp;
    }
}

// pack without normalising (stand-alone resample / getWeightedPose on the current particles)
__global__ void k_pack(const double *__restrict__ w, const float *__restrict__ x, const float *__restrict__ y,
                       const float *__restrict__ th, int32_t n, int64_t stride, PackedParticle *__restrict__ packed) {
    const int32_t mi = blockIdx.y;
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t gi = (size_t)mi * n + i;
    PackedParticle pp;
    pp.w = w[gi]; pp.x = x[gi]; pp.y = y[gi]; pp.theta = th[gi]; pp.pad = 0u;
    packed[(size_t)mi * stride + i] = pp;
}

// phase 3a: over the GLOBAL normalised population: block partials {sum w, sum x*w, sum y*w, sum th*w}
__global__ void __launch_bounds__(256)
k_global_partials(const PackedParticle *__restrict__ glob, int64_t n_global, int64_t nblk_global,
                  double *__restrict__ partials2) {
    __shared__ double l4[4];
    const int32_t mi = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * GMS_BLOCK + threadIdx.x;
    double w = 0.0, xw = 0.0, yw = 0.0, tw = 0.0;
    if (i < n_global) {
        const PackedParticle pp = glob[(size_t)mi * n_global + i];
        w = pp.w;
        xw = (double)pp.x * pp.w;                                      // SLAM.java:170
        yw = (double)pp.y * pp.w;                                      // :171
        tw = angle_constrain((double)pp.theta) * pp.w;                 // :172
    }
    const double s0 = block_sum_256(w, l4), s1 = block_sum_256(xw, l4), s2 = block_sum_256(yw, l4),
                 s3 = block_sum_256(tw, l4);
    if (threadIdx.x == 0) {
        double *p = partials2 + ((size_t)mi * nblk_global + blockIdx.x) * 4;
        p[0] = s0; p[1] = s1; p[2] = s2; p[3] = s3;
    }
}

// phase 3b: fold 3a (every block), then block partials of (w/sum)^2 (SLAM.java:185-187)
__global__ void __launch_bounds__(256)
k_global_sq(const PackedParticle *__restrict__ glob, int64_t n_global, int64_t nblk_global,
            const double *__restrict__ partials2, double *__restrict__ partials3, PfStatsDev *__restrict__ stats) {
    __shared__ double l4[4];
    const int32_t mi = blockIdx.y;
    double f[4];
    fold_partials(partials2 + (size_t)mi * nblk_global * 4, nblk_global, 4, 4, f, l4);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        PfStatsDev *s = stats + mi;
        s->norm_sum = f[0]; s->xs = f[1]; s->ys = f[2]; s->ts = f[3];
        s->wpose[0] = (float)(f[1] / f[0]);                            // SLAM.java:176
        s->wpose[1] = (float)(f[2] / f[0]);
        s->wpose[2] = (float)(f[3] / f[0]);
        int64_t st = s->strongest;
        if (st < 0) st = 0;
        if (st >= n_global) st = n_global - 1;
        const PackedParticle pp = glob[(size_t)mi * n_global + st];
        s->spose[0] = pp.x; s->spose[1] = pp.y; s->spose[2] = pp.theta;
    }
    const int64_t i = (int64_t)blockIdx.x * GMS_BLOCK + threadIdx.x;
    double q = 0.0;
    if (i < n_global) {
        const double w = glob[(size_t)mi * n_global + i].w;
        q = (w / f[0]) * (w / f[0]);
    }
    q = block_sum_256(q, l4);
    if (threadIdx.x == 0) partials3[(size_t)mi * nblk_global + blockIdx.x] = q;
}

// phase 3c: fold the squared partials -> sq_sum (Neff = 1 / sq_sum)
__global__ void __launch_bounds__(256)
k_fold_sq(const double *__restrict__ partials3, int64_t nblk_global, PfStatsDev *__restrict__ stats) {
    __shared__ double l4[4];
    const int32_t mi = blockIdx.x;
    double q;
    fold_partials(partials3 + (size_t)mi * nblk_global, nblk_global, 1, 1, &q, l4);
    if (threadIdx.x == 0) stats[mi].sq_sum = q;
}

// ---------------------------------------------------------------------------------------------
// resampling
// ---------------------------------------------------------------------------------------------
// one lane per chunk of SCAN_CHUNK particles: sequential inclusive sums (the reference's order inside
// the chunk)
__global__ void k_scan_chunks(const PackedParticle *__restrict__ glob, int64_t n_global, int64_t nchunks,
                              double *__restrict__ cum, double *__restrict__ chunk_tot) {
    const int32_t mi = blockIdx.y;
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    const PackedParticle *g = glob + (size_t)mi * n_global;
    double *cm = cum + (size_t)mi * n_global;
    double acc = 0.0;
    const int64_t i0 = c * SCAN_CHUNK;
    for (int64_t i = i0; i < i0 + SCAN_CHUNK && i < n_global; i++) {
        acc = (i == i0) ? g[i].w : acc + g[i].w;
        cm[i] = acc;
    }
    chunk_tot[(size_t)mi * (nchunks + 1) + c] = acc;
}

// one lane per map: chunk totals -> exclusive offsets, in order; slot [nchunks] = grand total
__global__ void k_scan_offsets(double *__restrict__ chunk_tot, int64_t nchunks, PfStatsDev *__restrict__ stats) {
    const int32_t mi = blockIdx.x;
    if (threadIdx.x != 0) return;
    stats[mi].n_ambiguous = 0;
    double *t = chunk_tot + (size_t)mi * (nchunks + 1);
    double acc = 0.0;
    for (int64_t c = 0; c < nchunks; c++) {
        const double v = t[c];
        t[c] = acc;
        acc = (c == 0) ? v : acc + v;
    }
    t[nchunks] = acc;
}

// one lane per output slot (SLAM.java:140-149)
__global__ void __launch_bounds__(256)
k_resample(const PackedParticle *__restrict__ glob, int64_t n_global, int64_t nchunks,
           const double *__restrict__ cum, const double *__restrict__ chunk_off, const double *__restrict__ r01,
           double fraction, int32_t n, int64_t offset, float *__restrict__ x2, float *__restrict__ y2,
           float *__restrict__ th2, double *__restrict__ w2, int32_t *__restrict__ idx_out,
           PfStatsDev *__restrict__ stats) {
    const int32_t mi = blockIdx.y;
    const int32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const PfStatsDev *st = stats + mi;
    const bool go = fraction < 0.0 || (1.0 / st->sq_sum) < fraction * (double)n_global;   // GridMapApp.java:185
    const PackedParticle *g = glob + (size_t)mi * n_global;
    const int64_t m0 = offset + t;              // m - 1
    int64_t src = m0;
    if (go) {
        const double *off = chunk_off + (size_t)mi * (nchunks + 1);
        const double *cm = cum + (size_t)mi * n_global;
        const double N = (double)n_global;
        const double r = r01[mi] * 1.0 / N;                             // SLAM.java:136
        const double U = r + (double)m0 * 1.0 / N;                      // :141
        // first chunk whose end value stops the `while (U > c)` loop: !(U > off[c+1])
        int64_t lo = 0, hi = nchunks;           // answer in [0, nchunks]; nchunks = none
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (U > off[mid + 1]) lo = mid + 1; else hi = mid;
        }
        if (lo >= nchunks) {
            src = n_global - 1;                 // Java would run off the list: clamp
        } else {
            const double base = off[lo];
            const int64_t i0 = lo * SCAN_CHUNK;
            int64_t a = 0, b = min((int64_t)SCAN_CHUNK, n_global - i0);   // first j with !(U > c_j)
            const int64_t len = b;
            while (a < b) {
                const int64_t mid = (a + b) >> 1;
                const double c = (lo == 0) ? cm[i0 + mid] : base + cm[i0 + mid];
                if (U > c) a = mid + 1; else b = mid;
            }
            src = i0 + (a < len ? a : len - 1);
            // boundary within rounding distance of U: a sequential scan may choose a neighbour
            const double tot = off[nchunks];
            const double tol = N * 4.5e-16 * fabs(tot);
            const double c_hit = (lo == 0) ? cm[src] : base + cm[src];
            bool amb = fabs(U - c_hit) <= tol;
            if (src > 0) {
                const int64_t pc = (src - 1) / SCAN_CHUNK;
                const double c_prev = (pc == 0) ? cm[src - 1] : off[pc] + cm[src - 1];
                amb = amb || fabs(U - c_prev) <= tol;
            }
            if (amb) atomicAdd(&stats[mi].n_ambiguous, 1);
        }
    }
    const PackedParticle pp = g[src];
    const size_t o = (size_t)mi * n + t;
    x2[o] = pp.x; y2[o] = pp.y; th2[o] = pp.theta; w2[o] = pp.w;       // copies keep their weight (SLAM.java:42)
    if (idx_out) idx_out[o] = (int32_t)src;
    if (t == 0) stats[mi].did_resample = go ? 1 : 0;
}

__global__ void k_pose_from_stats(const PfStatsDev *__restrict__ stats, int32_t which, float *__restrict__ poses) {
    const int32_t mi = blockIdx.x * blockDim.x + threadIdx.x;
    if (mi >= (int32_t)gridDim.x * (int32_t)blockDim.x) return;
    const float *src = which == 0 ? stats[mi].wpose : stats[mi].spose;
    poses[3 * mi] = src[0]; poses[3 * mi + 1] = src[1]; poses[3 * mi + 2] = src[2];
}

// ---------------------------------------------------------------------------------------------
// GridMap.findBestPose (J/slam/GridMap.java:319-346): lattice search around every particle.
// One workgroup per particle; waves take lattice poses round-robin; lanes stride the beams.
// ---------------------------------------------------------------------------------------------
#define REFINE_MAX_STEPS 16
__global__ void __launch_bounds__(256)
k_refine(GridDev g, const double *__restrict__ lik_all, const double *__restrict__ hitbeams,
         const int32_t *__restrict__ nhit, int32_t beam_stride, float *__restrict__ px, float *__restrict__ py,
         float *__restrict__ pth, int32_t n) {
    extern __shared__ __align__(16) unsigned char smem[];
    double2 *sb = reinterpret_cast<double2 *>(smem);
    __shared__ float s_dx[REFINE_MAX_STEPS], s_dy[REFINE_MAX_STEPS], s_dt[REFINE_MAX_STEPS];
    __shared__ int32_t s_n[3];
    __shared__ double s_best[4];
    __shared__ int32_t s_besti[4];
    const int32_t mi = blockIdx.y, p = blockIdx.x;
    const int32_t nb = nhit[mi];
    const double2 *hb = reinterpret_cast<const double2 *>(hitbeams + (size_t)mi * beam_stride * 2);
    for (int32_t i = threadIdx.x; i < nb; i += blockDim.x) sb[i] = hb[i];
    if (threadIdx.x == 0) {
        // the reference's float loop counters (GridMap.java:324-330)
        const float xSpan = 0.20f, ySpan = 0.20f, thetaSpan = (float)(15 * (3.141592653589793 / 180.0));
        const float transStep = 0.04f, thetaStep = thetaSpan / 5;
        int32_t c = 0;
        for (float d = -xSpan; d < xSpan && c < REFINE_MAX_STEPS; d += transStep) s_dx[c++] = d;
        s_n[0] = c; c = 0;
        for (float d = -ySpan; d < ySpan && c < REFINE_MAX_STEPS; d += transStep) s_dy[c++] = d;
        s_n[1] = c; c = 0;
        for (float d = -thetaSpan; d < thetaSpan && c < REFINE_MAX_STEPS; d += thetaStep) s_dt[c++] = d;
        s_n[2] = c;
    }
    __syncthreads();
    const size_t gi = (size_t)mi * n + p;
    const float x0 = px[gi], y0 = py[gi], t0 = pth[gi];
    const double *lik = lik_all + (size_t)mi * g.cells;
    const int32_t nx = s_n[0], ny = s_n[1], nt = s_n[2], total = nx * ny * nt;
    const int32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double best = 0.0;            // maxProb = 0 (:321)
    int32_t besti = -1;           // -1 = keep the start pose (:320)
    for (int32_t q = wave; q < total; q += 4) {
        const int32_t it = q % nt, iy = (q / nt) % ny, ix = q / (nt * ny);
        const float cx = x0 + s_dx[ix], cy = y0 + s_dy[iy], ct = t0 + s_dt[it];       // :332
        float c, s;
        pose_trig(ct, c, s);
        XformDev t;
        t.c = (double)c; t.s = (double)s; t.px = (double)cx; t.py = (double)cy;
        double prod = 1.0;
        for (int32_t j = lane; j < nb; j += 64) {
            const double2 bm = sb[j];
            prod *= beam_factor(g, lik, t, bm.x, bm.y);
        }
        int e;
        double mnt = frexp(prod, &e);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double m2 = __shfl_xor(mnt, o, GMS_WAVE);
            const int32_t e2 = __shfl_xor(e, o, GMS_WAVE);
            mx_mul(mnt, e, m2, e2);
        }
        const double prob = ldexp(mnt, e);
        if (prob > best) { best = prob; besti = q; }                                   // :334 (q ascending per wave)
    }
    if (lane == 0) { s_best[wave] = best; s_besti[wave] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double b = 0.0; int32_t bi = -1;
        for (int k = 0; k < 4; k++) {
            const double v = s_best[k]; const int32_t vi = s_besti[k];
            if (vi >= 0 && (v > b || (v == b && bi >= 0 && vi < bi))) { b = v; bi = vi; }
        }
        if (bi >= 0) {
            const int32_t it = bi % nt, iy = (bi / nt) % ny, ix = bi / (nt * ny);
            px[gi] = x0 + s_dx[ix]; py[gi] = y0 + s_dy[iy]; pth[gi] = t0 + s_dt[it];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline int64_t nblk_global_of(const gms_pf *pf) { return (pf->n_global + GMS_BLOCK - 1) / GMS_BLOCK; }
static inline int64_t nchunks_of(const gms_pf *pf) { return (pf->n_global + SCAN_CHUNK - 1) / SCAN_CHUNK; }

void gms_launch_pf_init(gms_pf *pf) {
    const int64_t total = (int64_t)pf->n_maps * pf->n;
    hipLaunchKernelGGL(k_pf_init, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, pf->map->stream, pf->d_x,
                       pf->d_y, pf->d_th, pf->d_w, pf->d_logw, total, 1.0 / (double)pf->n_global);
}

__global__ void k_set_poses_aos(const float *__restrict__ aos, float *__restrict__ x, float *__restrict__ y,
                                float *__restrict__ th, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) { x[i] = aos[3 * i]; y[i] = aos[3 * i + 1]; th[i] = aos[3 * i + 2]; }
}

void gms_launch_pf_set_poses_aos(gms_pf *pf, const float *d_xytheta) {
    const int64_t total = (int64_t)pf->n_maps * pf->n;
    hipLaunchKernelGGL(k_set_poses_aos, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, pf->map->stream, d_xytheta,
                       pf->d_x, pf->d_y, pf->d_th, total);
}

void gms_launch_pf_prep(gms_pf *pf, const gms_beam *d_beams, int32_t B, int32_t beam_stride) {
    gms_map *m = pf->map;
    const int64_t total = (int64_t)pf->n_maps * pf->n;
    hipLaunchKernelGGL(k_pf_prep, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, m->stream, pf->d_th, pf->d_cs,
                       total);
    hipLaunchKernelGGL(k_compact_beams, dim3(pf->n_maps), dim3(64), 0, m->stream, d_beams, B, beam_stride,
                       m->max_beams, pf->d_hitbeams, pf->d_nhit);
}

void gms_launch_pf_score(gms_pf *pf, int32_t B) {
    gms_map *m = pf->map;
    ProfScope ps(m, GMS_K_SCORE);
    const int32_t wpb = 4;
    int64_t blocks = ((int64_t)pf->n + wpb - 1) / wpb;
    // enough waves to fill the chip, few enough that the beam table is staged a bounded number of times
    const int64_t cap = 2048 / (pf->n_maps > 8 ? 8 : pf->n_maps);
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    const size_t smem = (size_t)B * sizeof(double2);
    if (smem > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_score), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem);
    hipLaunchKernelGGL(k_score, dim3((unsigned)blocks, pf->n_maps), dim3(64 * wpb), smem, m->stream, m->gd, m->d_lik,
                       pf->d_hitbeams, pf->d_nhit, m->max_beams, pf->d_x, pf->d_y, pf->d_cs, pf->n, pf->d_w,
                       pf->d_logw);
}

void gms_launch_pf_partials(gms_pf *pf, double *d_partials) {
    gms_map *m = pf->map;
    ProfScope ps(m, GMS_K_REDUCE);
    const int64_t nblk = nblk_global_of(pf);
    hipMemsetAsync(d_partials, 0, sizeof(double) * (size_t)pf->n_maps * nblk * GMS_PARTIAL_STRIDE, m->stream);
    hipLaunchKernelGGL(k_partials, dim3((pf->n + GMS_BLOCK - 1) / GMS_BLOCK, pf->n_maps), dim3(GMS_BLOCK), 0,
                       m->stream, pf->d_w, pf->d_logw, pf->n, pf->offset, nblk, d_partials);
}

void gms_launch_pf_apply_partials(gms_pf *pf, const double *d_partials, PackedParticle *d_packed_local) {
    gms_map *m = pf->map;
    ProfScope ps(m, GMS_K_REDUCE);
    const int64_t nblk = nblk_global_of(pf);
    // packed_stride: a stand-alone filter packs straight into its global population
    const int64_t stride = (d_packed_local == pf->d_global) ? pf->n_global : pf->n;
    hipLaunchKernelGGL(k_normalize_pack, dim3((pf->n + GMS_BLOCK - 1) / GMS_BLOCK, pf->n_maps), dim3(GMS_BLOCK), 0,
                       m->stream, d_partials, nblk, pf->d_w, pf->d_x, pf->d_y, pf->d_th, pf->n, stride,
                       d_packed_local, pf->d_stats);
}

void gms_launch_pf_pack(gms_pf *pf, PackedParticle *d_packed, int64_t stride) {
    gms_map *m = pf->map;
    hipLaunchKernelGGL(k_pack, dim3((pf->n + 255) / 256, pf->n_maps), dim3(256), 0, m->stream, pf->d_w, pf->d_x, pf->d_y,
                       pf->d_th, pf->n, stride, d_packed);
}

void gms_launch_pf_global_stats(gms_pf *pf) {
    gms_map *m = pf->map;
    ProfScope ps(m, GMS_K_REDUCE);
    const int64_t nblk = nblk_global_of(pf);
    double *p2 = pf->d_partials2;
    double *p3 = pf->d_partials2 + (size_t)pf->n_maps * nblk * 4;
    hipLaunchKernelGGL(k_global_partials, dim3((unsigned)nblk, pf->n_maps), dim3(GMS_BLOCK), 0, m->stream,
                       pf->d_global, pf->n_global, nblk, p2);
    hipLaunchKernelGGL(k_global_sq, dim3((unsigned)nblk, pf->n_maps), dim3(GMS_BLOCK), 0, m->stream, pf->d_global,
                       pf->n_global, nblk, p2, p3, pf->d_stats);
    hipLaunchKernelGGL(k_fold_sq, dim3(pf->n_maps), dim3(GMS_BLOCK), 0, m->stream, p3, nblk, pf->d_stats);
}

void gms_launch_pf_resample(gms_pf *pf, double fraction) {
    gms_map *m = pf->map;
    ProfScope ps(m, GMS_K_RESAMPLE);
    const int64_t nch = nchunks_of(pf);
    hipLaunchKernelGGL(k_scan_chunks, dim3((unsigned)((nch + 63) / 64), pf->n_maps), dim3(64), 0, m->stream,
                       pf->d_global, pf->n_global, nch, pf->d_cum, pf->d_chunk_tot);
    hipLaunchKernelGGL(k_scan_offsets, dim3(pf->n_maps), dim3(64), 0, m->stream, pf->d_chunk_tot, nch, pf->d_stats);
    hipLaunchKernelGGL(k_resample, dim3((pf->n + 255) / 256, pf->n_maps), dim3(256), 0, m->stream, pf->d_global,
                       pf->n_global, nch, pf->d_cum, pf->d_chunk_tot, pf->d_r01, fraction, pf->n, pf->offset,
                       pf->d_x2, pf->d_y2, pf->d_th2, pf->d_w2, pf->d_idx, pf->d_stats);
}

void gms_launch_pf_refine(gms_pf *pf, int32_t B) {
    gms_map *m = pf->map;
    ProfScope ps(m, GMS_K_REFINE);
    const size_t smem = (size_t)B * sizeof(double2);
    if (smem > 32 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_refine), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem);
    hipLaunchKernelGGL(k_refine, dim3(pf->n, pf->n_maps), dim3(256), smem, m->stream, m->gd, m->d_lik,
                       pf->d_hitbeams, pf->d_nhit, m->max_beams, pf->d_x, pf->d_y, pf->d_th, pf->n);
}

void gms_launch_pose_from_pf(gms_map *m, gms_pf *pf, int32_t which, float *d_poses) {
    hipLaunchKernelGGL(k_pose_from_stats, dim3(1), dim3(m->n_maps), 0, m->stream, pf->d_stats, which, d_poses);
}
