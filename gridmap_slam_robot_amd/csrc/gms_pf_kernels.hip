// gms_pf_kernels.hip -- gfx950 kernels for the particle scan matcher.
//
//   k_pf_prep / k_compact_beams  per-particle float-rounded trig (J/math/Transform.java:15-16) and
//                                the list of beams with wasHit (GridMap.java:269)
//   k_score_c      GridMap.probabilityOf for every particle (J/slam/GridMap.java:261-294): lane = particle,
//                  workgroup = up to 1024 particles x one beam segment; factor gathers from the L1-resident
//                  patch of the map that the workgroup's beams hit.
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
#define RES_SUB_MAX_CHUNKS 1024     // resampling: populations of up to 65536 keep the octet boundaries of every chunk in LDS (64 KiB at the limit)

// ---------------------------------------------------------------------------------------------
__global__ void k_pf_init(float *pose, float *cs, double *w, double *logw, int64_t total, double w0) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        pose[3 * i] = 0.0f; pose[3 * i + 1] = 0.0f; pose[3 * i + 2] = 0.0f;
        cs[2 * i] = 1.0f; cs[2 * i + 1] = 0.0f;           // (float)cos(0), (float)sin(0)
        w[i] = w0; logw[i] = 0.0;
    }
}

// Poses enter the filter through this kernel: copy (src may equal dst) + the per-particle float-rounded
// trig of Transform.fromRobotToWorld (Transform.java:15-16).  Invariant: cs[] always matches pose[].
__global__ void __launch_bounds__(256)
k_pose_trig(const float *__restrict__ src, float *__restrict__ dst, float *__restrict__ cs, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float x = src[3 * i], y = src[3 * i + 1], th = src[3 * i + 2];
    dst[3 * i] = x; dst[3 * i + 1] = y; dst[3 * i + 2] = th;
    float c, s;
    pose_trig(th, c, s);
    cs[2 * i] = c; cs[2 * i + 1] = s;
}

// Odometry.apply for every particle (J/slam/Odometry.java:60-96): d ~ N(dCenter, dCenterSD), theta ~ N(dTheta,
// dThetaSD); heading first, then the step along the new heading (float pose, double arithmetic).  The
// reference's stream (commons-math Well1024a, unseeded: Odometry.java:28-30) cannot be reproduced; the variates
// come from Philox4x32-10 keyed by `seed` with counter {GLOBAL particle index, sequence} -- the same numbers
// whatever the sharding -- through Box-Muller on two 53-bit uniforms.  cs[] is refreshed in the same pass.
__device__ __forceinline__ void philox_round(uint32_t c[4], const uint32_t k[2]) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1], n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

// Odometry.apply for one particle, on values (J/slam/Odometry.java:77-96): (x, y, th) move on, (fc, fs) is the new heading's trig.
// index: the particle's global index (+ the map's counter offset); the variates are Philox4x32-10 on (index, sequence) keyed by seed.
struct MotionArgs {
    int32_t on;
    double d_center, d_theta, d_center_sd, d_theta_sd;
    uint64_t seed, sequence;
    int64_t index0;             // global index of the launch's particle 0 (a shard's offset): the variates are keyed by the GLOBAL index
};
__device__ __forceinline__ void motion_apply(float &x, float &y, float &th, float &fc, float &fs, uint64_t index, double d_center,
                                             double d_theta, double d_center_sd, double d_theta_sd, uint64_t seed, uint64_t sequence) {
    uint32_t c[4] = { (uint32_t)index, (uint32_t)(index >> 32), (uint32_t)sequence, (uint32_t)(sequence >> 32) };
    uint32_t k[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
#pragma unroll
    for (int r = 0; r < 10; r++) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
    const double u1 = ((double)(((uint64_t)(c[0] >> 5) << 26) | (uint64_t)(c[1] >> 6)) + 0.5) * (1.0 / 9007199254740992.0);
    const double u2 = ((double)(((uint64_t)(c[2] >> 5) << 26) | (uint64_t)(c[3] >> 6)) + 0.5) * (1.0 / 9007199254740992.0);
    const double rad = sqrt(-2.0 * log(u1));
    const double z0 = rad * cos(2.0 * 3.141592653589793 * u2), z1 = rad * sin(2.0 * 3.141592653589793 * u2);
    const double d = d_center + d_center_sd * z0;                                // ndCenter.sample()  :80
    const double theta = d_theta + d_theta_sd * z1;                              // ndTheta.sample()   :81
    th = (float)angle_constrain((double)th + theta);                             // :92
    pose_trig(th, fc, fs);                                                       // MathUtil.cos(float) :93
    x = (float)((double)x + (double)fc * d);                                     // :93  p.x += cos * d
    y = (float)((double)y + (double)fs * d);                                     // :94
}
__device__ __forceinline__ void
motion_body(float *__restrict__ pose, float *__restrict__ cs, int32_t n, int64_t offset, double d_center, double d_theta,
            double d_center_sd, double d_theta_sd, uint64_t seed, uint64_t sequence, int32_t mi, int32_t i) {
    if (i >= n) return;
    const uint64_t index = (uint64_t)(offset + i) + ((uint64_t)mi << 40);      // maps draw from disjoint counters
    const size_t gi = (size_t)mi * n + i;
    float x = pose[3 * gi], y = pose[3 * gi + 1], th = pose[3 * gi + 2], fc, fs;
    motion_apply(x, y, th, fc, fs, index, d_center, d_theta, d_center_sd, d_theta_sd, seed, sequence);
    pose[3 * gi + 2] = th;
    pose[3 * gi] = x;
    pose[3 * gi + 1] = y;
    cs[2 * gi] = fc; cs[2 * gi + 1] = fs;
}
__global__ void __launch_bounds__(256)
k_motion(float *__restrict__ pose, float *__restrict__ cs, int32_t n, int64_t offset, double d_center, double d_theta,
         double d_center_sd, double d_theta_sd, uint64_t seed, uint64_t sequence) {
    motion_body(pose, cs, n, offset, d_center, d_theta, d_center_sd, d_theta_sd, seed, sequence, (int32_t)blockIdx.y,
                (int32_t)(blockIdx.x * blockDim.x + threadIdx.x));
}

// One wavefront per map: order-preserving compaction of the beams with wasHit (GridMap.java:269); used
// by the lattice search (k_refine).
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
// probabilityOf: one beam end point (GridMap.java:273-288).  The factor f(likelihoodData[cell]) is
// read from the map's factor table, which the likelihood kernel keeps in step with likelihoodData
// (same multiply-then-add, same `== 0.5` test: GridMap.java:285-288); a beam whose end point falls
// outside the map (:276) reads the neutral entry [g.fneutral] = 1.0, so the loop has no branch.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t cell_or_neutral(const GridDev &g, int32_t gx, int32_t gy) {
    return fac_index(g, gx, gy);                                                     // :276 -> the table's neutral border
}
// fast form (no division, no branch); `guard` accumulates "an exact quotient is needed"
__device__ __forceinline__ uint32_t beam_cell_fast(const GridDev &g, const XformDev &t, double lx, double ly, bool &guard) {
    const int32_t gx = j_cell_fast(xform_x(t, lx, ly) - g.posx, g.rinv, guard);     // :273
    const int32_t gy = j_cell_fast(xform_y(t, lx, ly) - g.posy, g.rinv, guard);     // :274
    return cell_or_neutral(g, gx, gy);
}
// the reference's expression, used for the whole batch when any guard tripped (rare)
__device__ __forceinline__ uint32_t beam_cell(const GridDev &g, const XformDev &t, double lx, double ly) {
    const int32_t gx = j_cell_exact(xform_x(t, lx, ly) - g.posx, g.res);
    const int32_t gy = j_cell_exact(xform_y(t, lx, ly) - g.posy, g.res);
    return cell_or_neutral(g, gx, gy);
}

// ---------------------------------------------------------------------------------------------
// Locality order of the particles for a large scoring launch.  The texture-address pipe serves an 8-byte gather of 64
// lanes in ~48 clocks when every lane has a line of its own and in ~25 when NEIGHBOURING lanes share lines (it merges
// adjacent lanes only: tools/microbench/gather_coalesce.hip), and the caller's particles come in no particular order.
// The particles of a map are bucketed by (theta, y, x) -- 64 x 16 x 4 bins, theta-major, at least 0.5 degrees and one cell
// wide, wider when the cloud is -- by a counting sort in LDS, and {x, y, cos, sin} and the particle index are stored in
// bucket order: neighbouring lanes of k_score_c then hold neighbouring poses, whose beam end points fall into the same
// cells' lines (C5: the scoring kernel 328 -> 250 us).  Which lane forms which particle's product does not enter the
// arithmetic: weights are bit-identical with or without the order (and whatever the arrival order of the atomics
// inside a bucket).
// grid = (map, slice of 1024 particles).  Every workgroup histograms ALL particles of its map (12 bytes each, from L2)
// -- so it knows every bucket's start and how many particles of EARLIER slices precede its own in each bucket -- and
// then places only its own slice: position = bucket start + earlier slices' count + arrival rank inside the slice.
// No workgroup waits for another, the double-precision sincos of the poses that enter through this launch (pose_src;
// what k_score_c / k_pose_trig do otherwise) is one per thread instead of n / 1024, and the work spreads over
// n_maps x n / 1024 CUs (one workgroup per map: 15.6 us at C5, 5.3 of them trig on 64 CUs).
// ---------------------------------------------------------------------------------------------
// theta x y x x buckets and their smallest widths.  Measured at C5 (scoring kernel, us): 64 x 8 x 8 at >= 2 cells 250;
// 64 x 16 x 4 at >= 2 cells 245; 64 x 16 x 4 at >= 1 cell 234 (kept); 64 x 64 x 1 236; 64 x 32 x 2 238; 128 x 16 x 2 239; 32 x 32 x 4 240;
// 64 x 8 x 8 at >= 1 cell 249; >= 0.25 / 1 degree instead of 0.5: 239 / 241.  The table's rows run along x, and the pipe merges
// neighbouring lanes on neighbouring ADDRESSES: fine buckets in y (same row), coarse in x (a run inside the row).
#define ORD_TBITS 6
#define ORD_YBITS 4
#define ORD_XBITS 2
#define ORD_MIN_CELLS 1.0f
#define ORD_MIN_RAD 0.0087266463f
#define ORD_BINS (1 << (ORD_TBITS + ORD_XBITS + ORD_YBITS))
#define ORD_THREADS 1024
__global__ void __launch_bounds__(ORD_THREADS)
k_order(GridDev g, const float *__restrict__ pose_src, const float *__restrict__ pose, const float *__restrict__ cs, int32_t n,
        float4 *__restrict__ ord, int32_t *__restrict__ perm, float *__restrict__ pose_dst, float *__restrict__ cs_dst) {
    __shared__ uint32_t s_all[ORD_BINS];       // particles per bucket, then the buckets' starts
    __shared__ uint32_t s_before[ORD_BINS];    // ... of the slices before this workgroup's
    __shared__ uint32_t s_own[ORD_BINS];       // arrival counter of this workgroup's slice
    __shared__ float s_lo[3][ORD_THREADS / 64], s_hi[3][ORD_THREADS / 64];
    __shared__ uint32_t s_wsum[ORD_THREADS / 64];
    const int32_t mi = blockIdx.x, slice = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *src = (pose_src ? pose_src : pose) + (size_t)mi * n * 3;
    const int32_t own = slice * ORD_THREADS + tid;                    // this thread's particle
    // bounds of the cloud (NaN coordinates are ignored by fminf / fmaxf and land in bin 0 below)
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll 4
    for (int32_t i = tid; i < n; i += ORD_THREADS)
#pragma unroll
        for (int d = 0; d < 3; d++) { const float v = src[3 * (size_t)i + d]; lo[d] = fminf(lo[d], v); hi[d] = fmaxf(hi[d], v); }
#pragma unroll
    for (int d = 0; d < 3; d++) {
#define GMS_STEP_(O) { lo[d] = fminf(lo[d], wave_xor<O>(lo[d])); hi[d] = fmaxf(hi[d], wave_xor<O>(hi[d])); }
        GMS_BUTTERFLY(GMS_STEP_)
#undef GMS_STEP_
        if (lane == 0) { s_lo[d][wave] = lo[d]; s_hi[d][wave] = hi[d]; }
    }
    for (int32_t b = tid; b < ORD_BINS; b += ORD_THREADS) { s_all[b] = 0u; s_before[b] = 0u; s_own[b] = 0u; }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 3; d++) {
        lo[d] = s_lo[d][0]; hi[d] = s_hi[d][0];
        for (int w = 1; w < ORD_THREADS / 64; w++) { lo[d] = fminf(lo[d], s_lo[d][w]); hi[d] = fmaxf(hi[d], s_hi[d][w]); }
    }
    const float nx = (float)(1 << ORD_XBITS), ny = (float)(1 << ORD_YBITS), nth = (float)(1 << ORD_TBITS);
    const float inv_x = nx / fmaxf(hi[0] - lo[0], nx * ORD_MIN_CELLS * g.resf), inv_y = ny / fmaxf(hi[1] - lo[1], ny * ORD_MIN_CELLS * g.resf);
    const float inv_t = nth / fmaxf(hi[2] - lo[2], nth * ORD_MIN_RAD);
    auto key_of = [&](float x, float y, float th) -> uint32_t {
        const uint32_t ix = (uint32_t)fminf(fmaxf((x - lo[0]) * inv_x, 0.0f), nx - 1.0f);
        const uint32_t iy = (uint32_t)fminf(fmaxf((y - lo[1]) * inv_y, 0.0f), ny - 1.0f);
        const uint32_t it = (uint32_t)fminf(fmaxf((th - lo[2]) * inv_t, 0.0f), nth - 1.0f);
        return (it << (ORD_XBITS + ORD_YBITS)) | (iy << ORD_XBITS) | ix;
    };
#pragma unroll 4
    for (int32_t i = tid; i < n; i += ORD_THREADS) {
        const uint32_t key = key_of(src[3 * (size_t)i], src[3 * (size_t)i + 1], src[3 * (size_t)i + 2]);
        atomicAdd(&s_all[key], 1u);
        if (i < slice * ORD_THREADS) atomicAdd(&s_before[key], 1u);   // (uniform per iteration: i / 1024 < slice)
    }
    __syncthreads();
    {   // exclusive prefix over the buckets: thread t owns buckets 4t .. 4t+3
        constexpr int PER = ORD_BINS / ORD_THREADS;
        uint32_t c[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) { c[k] = s_all[tid * PER + k]; sum += c[k]; }
        uint32_t incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o, GMS_WAVE);
            if (lane >= o) incl += up;
        }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t base = incl - sum;
        for (int w = 0; w < wave; w++) base += s_wsum[w];
#pragma unroll
        for (int k = 0; k < PER; k++) { s_all[tid * PER + k] = base; base += c[k]; }
    }
    __syncthreads();
    if (own < n) {
        const size_t gi = (size_t)mi * n + own;
        const float x = src[3 * (size_t)own], y = src[3 * (size_t)own + 1], th = src[3 * (size_t)own + 2];
        float c, sn;
        if (pose_src) {
            pose_trig(th, c, sn);                                      // Transform.java:15-16
            pose_dst[3 * gi] = x; pose_dst[3 * gi + 1] = y; pose_dst[3 * gi + 2] = th;
            cs_dst[2 * gi] = c; cs_dst[2 * gi + 1] = sn;
        } else {
            c = cs[2 * gi]; sn = cs[2 * gi + 1];
        }
        const uint32_t key = key_of(x, y, th);
        const uint32_t pos = s_all[key] + s_before[key] + atomicAdd(&s_own[key], 1u);
        ord[(size_t)mi * n + pos] = make_float4(x, y, c, sn);
        perm[(size_t)mi * n + pos] = own;
    }
}

// ---------------------------------------------------------------------------------------------
// probabilityOf, cache-blocked form (the default).  PMC counters on MI355X showed the first two forms (a wavefront per
// particle; a wavefront per beam segment of 64 particles: removed after round 2, 40 and 25+ us at C3) bound by L2->L1 line fills (TCP_PENDING_STALL, TA_ADDR_STALLED_BY_TC: every 8-byte look-up drags a
// 128-byte line into a 32 KiB L1 that 16 wavefronts with 16 different access patches keep evicting).
// Here a workgroup is up to 1024 particles (lane = particle, 16 wavefronts) x ONE beam segment: all
// wavefronts of a CU walk the same few dozen beams, so their look-ups fall into one patch of the map
// (the beam's end point seen from a cloud of neighbouring poses) that stays L1-resident and is
// reused by every wavefront and by the following beams.  grid = (particle groups, beam segments).
// Each lane multiplies its factors sequentially in beam order (the reference's order,
// GridMap.java:267-288); with more than one segment the per-segment products are stored and
// multiplied in segment order by k_score_combine.  One segment => the reference's product exactly.
// ---------------------------------------------------------------------------------------------
// U: look-ups in flight per lane beside the software pipeline.  1 for particles in the caller's order (C3: 1 / 2 / 3 / 4 -> 20.6 / 21.2 /
// 24.9 / 21.0 us: more gathers in flight from 64 unrelated lanes thrash the L1 patch); 3 for particles in k_order's locality order, whose
// neighbouring lanes share lines (C5: 1 / 2 / 3 / 4 -> 240 / 233 / 223 / 237 us; 3 divides the 45-beam segment: no padded batch).
template <int U>
__global__ void __launch_bounds__(1024)
k_score_c(GridDev g, const double *__restrict__ fac_all, int64_t fac_stride, const gms_beam *__restrict__ beams,
          int32_t B, int32_t beam_stride, const float *__restrict__ pose, const float *__restrict__ cs, int32_t n,
          int32_t nseg, double *__restrict__ part, double *__restrict__ w, double *__restrict__ logw,
          const float *__restrict__ pose_src, float *__restrict__ pose_dst, float *__restrict__ cs_dst,
          const float4 *__restrict__ ord, const int32_t *__restrict__ perm, MotionArgs mo, int64_t offset, int32_t spread) {
    __shared__ double2 s_beam[128 + U];        // this segment's beams with wasHit, in order
    __shared__ int32_t s_nb;
    // Workgroup -> (beam segment, particle group), XCD-aware: consecutive workgroup ids go round-robin over the 8 XCDs
    // (each with an L2 of its own), so id & 7 picks the XCD and every XCD gets nseg / 8 ADJACENT segments for all particle
    // groups: its L2 then holds the windows of a few neighbouring beams instead of the whole scan's (C3: 23.8 -> 22.3 us
    // for segment-major order, -> see DESIGN.md for the adjacent-segment form; C5: 335 -> 326 us).
    GMS_STAMP(GMS_STAMP_ROW(0, blockIdx.x), 0);
    const int32_t mi = blockIdx.z;
    int32_t seg, grp;
    {
        const int32_t i = blockIdx.x;
        if ((nseg & 7) == 0) {
            // an XCD's workgroups in dispatch order (j) alternate between its segments.  (Round 4 measured the other order -- all
            // particle groups of the XCD's first segment, then of its next, so that two workgroups sharing a CU walk the same L1
            // patch: 65 536 particles 87.5 -> 104.3 us, 512-lane workgroups at C3 20 -> 30.6 us: thirty-two CUs gathering from one
            // segment's few hundred L2 lines at once queue on those lines' channels.)
            const int32_t spx = nseg >> 3, j = i >> 3;
            seg = (i & 7) * spx + j % spx;
            grp = j / spx;
            const int32_t ngrp = gridDim.x / nseg;
            if (spread && (ngrp & 1) == 0) {
                // launches of several workgroups per CU (config 4 on one or two GPUs): an XCD takes the first half of the particle
                // groups of its own nseg / 8 segments and the second half of the next XCD's, i.e. twice the segments with half the
                // CUs on each -- fewer CUs queue on one patch's L2 lines, at the price of every patch being fetched into two L2s
                // (65 536 particles: 86.2 -> 79.8 us; at one workgroup per CU, config 3, it costs a microsecond: 19.35 -> 20.3)
                const int32_t k = j % (2 * spx), h = j / (2 * spx);
                seg = ((i & 7) * spx + k) % nseg;
                grp = h + (k >= spx ? ngrp / 2 : 0);
            }
        } else {
            seg = i % nseg;
            grp = i / nseg;
        }
    }
    const int32_t L = (B + nseg - 1) / nseg;          // <= 128 (launcher)
    const int32_t j0 = seg * L, j1 = min(B, j0 + L);
    const gms_beam *mb = beams + (size_t)mi * beam_stride;
    const double *fac = fac_all + (size_t)mi * fac_stride;
    if (threadIdx.x < 64) {                            // first wavefront: order-preserving compaction (GridMap.java:269)
        int32_t base = 0;
        for (int32_t b0 = j0; b0 < j1; b0 += 64) {
            const int32_t b = b0 + (int32_t)threadIdx.x;
            const bool hit = b < j1 && mb[b].hit != 0;
            const unsigned long long mask = __ballot(hit);
            if (hit) s_beam[base + __popcll(mask & ((1ull << threadIdx.x) - 1ull))] = make_double2(mb[b].local_x, mb[b].local_y);
            base += __popcll(mask);
        }
        if (threadIdx.x < U) s_beam[base + threadIdx.x] = make_double2(0.0, 0.0);   // padding of the last batch
        if (threadIdx.x == 0) s_nb = base;
    }
    // pose_src: the poses enter the filter through this launch (SLAM.java:90): every segment's workgroup takes its
    // particles' trig itself -- the arithmetic hides under the first wavefront's beam compaction -- and segment 0
    // stores pose and trig where the other kernels expect them (what k_pose_trig does in a launch of its own)
    // ord: the lanes take the particles in k_order's locality order; op is the particle whose product this lane forms
    const int32_t p = grp * blockDim.x + threadIdx.x;
    const size_t li = (size_t)mi * n + (p < n ? p : 0);
    const int32_t op = ord ? perm[li] : (p < n ? p : 0);
    const size_t gi = (size_t)mi * n + op;
    XformDev t;
    if (ord) {
        const float4 o = ord[li];
        t.c = (double)o.z; t.s = (double)o.w; t.px = (double)o.x; t.py = (double)o.y;
    } else if (pose_src) {
        float x = pose_src[3 * gi], y = pose_src[3 * gi + 1], th = pose_src[3 * gi + 2];
        float c, sn;
#ifdef GMS_STAMPS
        if (th != th) x = 0.0f;                 // (the stamp below waits for the pose)
        GMS_STAMP(GMS_STAMP_ROW(0, blockIdx.x), 3);
#endif
        if (mo.on)                                                     // (uniform) the motion-model sample happens here: SLAM.java:90
            motion_apply(x, y, th, c, sn, (uint64_t)(offset + op) + ((uint64_t)mi << 40), mo.d_center, mo.d_theta, mo.d_center_sd,
                         mo.d_theta_sd, mo.seed, mo.sequence);
        else
            pose_trig(th, c, sn);                                      // Transform.java:15-16
        t.c = (double)c; t.s = (double)sn; t.px = (double)x; t.py = (double)y;
#ifdef GMS_STAMPS
        if (c != c) x = 0.0f;
        GMS_STAMP(GMS_STAMP_ROW(0, blockIdx.x), 4);
#endif
        if (seg == 0 && p < n) {
            pose_dst[3 * gi] = x; pose_dst[3 * gi + 1] = y; pose_dst[3 * gi + 2] = th;
            cs_dst[2 * gi] = c; cs_dst[2 * gi + 1] = sn;
        }
    } else {
        t.c = (double)cs[2 * gi]; t.s = (double)cs[2 * gi + 1];
        t.px = (double)pose[3 * gi]; t.py = (double)pose[3 * gi + 1];
    }
    __syncthreads();
    GMS_STAMP(GMS_STAMP_ROW(0, blockIdx.x), 1);
    const int32_t nb = s_nb;
    if (p >= n) return;
    double prod = 1.0;                                                 // GridMap.java:262
    // Software pipeline: the cell indices of batch b+1 are computed while the look-ups of batch b are in
    // flight, so the vector ALU work hides under the gather latency inside each wavefront (the gathers,
    // not the arithmetic, are the scarcer resource: tools/microbench/gather8.hip).
    auto cells_of = [&](int32_t base, uint32_t cell[U]) {
        double2 bm[U];
        bool guard = false;
#pragma unroll
        for (int u = 0; u < U; u++) {
            bm[u] = s_beam[base + u];                                  // same address in every lane: LDS broadcast
            const uint32_t c = beam_cell_fast(g, t, bm[u].x, bm[u].y, guard);
            cell[u] = base + u < nb ? c : g.fneutral;
        }
        if (__builtin_expect(guard, 0)) {                              // ~4e-6 of the end points
            asm volatile("; exact quotients for this batch" ::: "memory");
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t c = beam_cell(g, t, bm[u].x, bm[u].y);
                cell[u] = base + u < nb ? c : g.fneutral;
            }
        }
    };
    uint32_t cell[U];
    if (nb > 0) cells_of(0, cell);
    for (int32_t base = 0; base < nb; base += U) {
        double f[U];
#pragma unroll
        for (int u = 0; u < U; u++) f[u] = fac[cell[u]];         // issue the look-ups of this batch
        if (base + U < nb) cells_of(base + U, cell);       // ... and compute the next batch's cells meanwhile
#pragma unroll
        for (int u = 0; u < U; u++) prod *= f[u];                // beam order: batch by batch, u ascending
    }
    if (nseg == 1) {
        int e;
        const double mnt = frexp(prod, &e);
        w[gi] = prod;
        logw[gi] = log(mnt) + (double)e * 0.6931471805599453;
    } else {
        part[((size_t)mi * nseg + seg) * n + op] = prod;              // <= 128 factors >= 0.01: no underflow
    }
    GMS_STAMP(GMS_STAMP_ROW(0, blockIdx.x), 2);
}

// product of the per-segment products, in segment order, with an exact exponent (no underflow on the
// way); the loads of all segments are issued together.
// Every segment product is split into mantissa and exponent first (independent of one another); the mantissas, all in
// [0.5, 1), are multiplied in segment order WITHOUT renormalising in between -- thirty-two of them stay above 2^-32, far from
// the denormal range, and scaling by a power of two does not change how a product rounds, so the bits are those of the
// renormalised chain (a frexp after every factor: eight dependent double-precision operations per segment on a wavefront
// that has its SIMD to itself; this form has one) -- and the exponents are added up as integers.
__device__ __forceinline__ void combine_segments(const double *__restrict__ part, int32_t mi, int32_t n, int32_t nseg,
                                                 int64_t p, double &wv, double &lwv) {
    const double *q = part + ((size_t)mi * nseg) * n + p;
    double M = 1.0;
    int32_t e = 0;
    // every segment's load is issued before the first product is used: one round trip for up to GMS_SCORE_MAXSEG = 32 segments
    // (a loop over chunks of sixteen was two dependent round trips for the thirty 12-beam segments of a short scan: C2's
    // k_partials 7.1 -> 6.4 us)
    double v[GMS_SCORE_MAXSEG];
    const bool wide = nseg > 16;                                       // (uniform) the second sixteen only when there are any
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = q[(size_t)min(k, nseg - 1) * n];            // (clamped: no load behind a branch)
    if (wide) {
#pragma unroll
        for (int k = 16; k < GMS_SCORE_MAXSEG; k++) v[k] = q[(size_t)min(k, nseg - 1) * n];
    }
#define GMS_FACTOR_(k) { int e2; double m2 = frexp(v[k], &e2);                                                     \
                         if ((k) >= nseg) { m2 = 1.0; e2 = 0; }    /* (uniform) x * 1.0 == x: a missing segment changes nothing */ \
                         M *= m2; e += e2; }                        /* 1.0 * m == m: the first factor enters as it is */
#pragma unroll
    for (int k = 0; k < 16; k++) GMS_FACTOR_(k)
    if (wide) {
#pragma unroll
        for (int k = 16; k < GMS_SCORE_MAXSEG; k++) GMS_FACTOR_(k)
    }
#undef GMS_FACTOR_
    int de;
    const double mnt = frexp(M, &de);
    e += de;
    wv = ldexp(mnt, e);
    lwv = log(mnt) + (double)e * 0.6931471805599453;
}

__global__ void __launch_bounds__(256)
k_score_combine(const double *__restrict__ part, int32_t n, int32_t nseg, double *__restrict__ w,
                double *__restrict__ logw) {
    const int32_t mi = blockIdx.y;
    const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    double wv, lwv;
    combine_segments(part, mi, n, nseg, p, wv, lwv);
    w[(size_t)mi * n + p] = wv;
    logw[(size_t)mi * n + p] = lwv;
}

// ---------------------------------------------------------------------------------------------
// Blocked reductions.  A reduction block is GMS_BLOCK = 256 consecutive particles (by GLOBAL index)
// and is reduced by a "group" of 256 consecutive threads (4 waves), EPT = GMS_BLOCK/256 particles per thread.
// Shape, identical in every kernel that uses it:
//   thread t folds its EPT particles t, t+256, ... of the block sequentially (EPT = 1: just its own),
//   64-lane xor butterfly per wave, then ((w0 + w1) + w2) + w3 over the group's four waves;
//   blocks are folded in block order (thread t takes blocks t, t+256, ... then the group shape).
// Kernels run 1 or 4 groups per workgroup; every thread of the workgroup must make the calls (they
// contain barriers).  Because the shape depends only on global indices, the results are the same for
// any number of GPUs.
// ---------------------------------------------------------------------------------------------
#define MAX_WAVES 16
#define GRP 256
#define EPT (GMS_BLOCK / GRP)
#define EPB (EPT < 4 ? EPT : 4)      // particles in flight per thread

__device__ __forceinline__ double group_sum(double v, double *lds /* [MAX_WAVES] */) {
    v = wave_sum_f64(v);
    const int32_t wave = threadIdx.x >> 6, g4 = (wave >> 2) << 2;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds[wave] = v;
    __syncthreads();
    return ((lds[g4] + lds[g4 + 1]) + lds[g4 + 2]) + lds[g4 + 3];
}

// (value, index) max with "first maximum wins" (SLAM.java:110-115: strict >); NaN never wins
__device__ __forceinline__ void argmax_merge(double &v, double &i, double v2, double i2) {
    if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
}

__device__ __forceinline__ void group_argmax(double &v, double &i, double *ldsv, double *ldsi) {
#define GMS_STEP_(O) { const double v2 = wave_xor<O>(v), i2 = wave_xor<O>(i); argmax_merge(v, i, v2, i2); }
    GMS_BUTTERFLY(GMS_STEP_)
#undef GMS_STEP_
    const int32_t wave = threadIdx.x >> 6, g4 = (wave >> 2) << 2;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { ldsv[wave] = v; ldsi[wave] = i; }
    __syncthreads();
    v = ldsv[g4]; i = ldsi[g4];
    for (int k = 1; k < 4; k++) argmax_merge(v, i, ldsv[g4 + k], ldsi[g4 + k]);
}

struct RedLds {
    double a[MAX_WAVES], b[MAX_WAVES];
    double m[GMS_PARTIAL_STRIDE][MAX_WAVES];
};

// Partial columns of one block of GMS_BLOCK particles (GMS_PARTIAL_STRIDE = 9):
//   0 sum w            (SLAM.java:100)          5 sum w*w          (-> calculateNeff, SLAM.java:180-190)
//   1 max w            (SLAM.java:110-115)      6 sum x*w          (-> getWeightedPose, SLAM.java:165-178)
//   2 first argmax (global index)               7 sum y*w
//   3 count of w == 0                           8 sum angleConstrain(theta)*w
//   4 max log-weight
// One pass over the raw weights yields the weight sum, the strongest particle and the weighted pose:
// with S = sum w, normalised weight = w / S and pose = (sum x*w) / S.  The reference computes the pose
// from the already-normalised weights; algebraically identical, and the rounding difference is ~1e-15
// relative, ten orders of magnitude inside the 1e-5 parity bar (DESIGN.md "Bookkeeping").
// Neff is NOT derived from column 5: raw weights are products of hundreds of factors (1e-150 is
// typical at 720 beams) and their squares underflow.  Like the reference (SLAM.java:180-190) it is
// computed from the normalised weights: level 0 of the cumulative-weight scan (k_normalize_pack /
// k_chunk_sums) leaves per-block {sum wn, sum wn^2}, fold_neff() folds them where Neff is consumed.
#define COL_SUM 0
#define COL_MAX 1
#define COL_ARG 2
#define COL_NZ 3
#define COL_MLW 4
#define COL_SQ 5
#define COL_XW 6
#define COL_YW 7
#define COL_TW 8

// part != nullptr: the weights are still per-segment products from k_score_c: combine them here (and
// store weight and log-weight) instead of in a launch of their own.
// lognorm (gms_pf_set_log_normalize): the underflow-free normalisation SURVEY 9.6 asks for beside the reference's plain product,
// weight = exp(logw - M) / sum, M the largest log-weight of the map's population, WITHOUT a pass of its own in front: this block's
// columns are taken relative to the block's OWN largest log-weight m_b (column COL_MLW) -- v = exp(logw - m_b) -- and whoever folds
// the partial vector rescales every block by exp(m_b - M) (fold_sums_scaled: the online-softmax rearrangement; the sums differ from
// sum exp(logw - M) by a few ulps).  The weights themselves are formed from the log-weights where they are normalised
// (normalize_pack_body), exp(logw - M) / sum: nothing block-relative is ever stored.  COL_MAX / COL_ARG then hold the largest
// LOG-weight and its first index (the strongest particle is the first maximum of the log-weights: two different log-weights may
// round to the same weight).  (Round 4 took M from a launch of its own, k_logmax, in front of this one: +4 us per step.)
static_assert(EPT == 1, "the log-normalising form of block_partials takes one particle per thread");
__device__ __forceinline__ double lognorm_ref(double m) { return m > -INFINITY ? m : 0.0; }      // a block (or a population) whose products are all 0 or NaN: nothing to rescale by
__device__ __forceinline__ void block_partials(double *__restrict__ w, double *__restrict__ logw,
                                               const float *__restrict__ pose, int64_t cnt, int64_t base,
                                               double out[GMS_PARTIAL_STRIDE], RedLds &L,
                                               const double *__restrict__ part = nullptr, int32_t part_mi = 0,
                                               int32_t part_n = 0, int32_t part_nseg = 0, int64_t part_p0 = 0,
                                               bool lognorm = false) {
    const int32_t tl = threadIdx.x & (GRP - 1);
    double s = 0.0, nz = 0.0, mv = -INFINITY, mx = 9.0e15, ml = -INFINITY, sq = 0.0, xw = 0.0, yw = 0.0, tw = 0.0;
#pragma unroll
    for (int e0 = 0; e0 < EPT; e0 += EPB) {
        double v[EPB], lw[EPB];
        float px[EPB], py[EPB], pt[EPB];
#pragma unroll
        for (int e = 0; e < EPB; e++) {
            const int64_t i = tl + (e0 + e) * GRP;
            const bool in = i < cnt;
            if (part) {
                v[e] = 0.0; lw[e] = -INFINITY;
                if (in) {
                    combine_segments(part, part_mi, part_n, part_nseg, part_p0 + i, v[e], lw[e]);
                    w[i] = v[e]; logw[i] = lw[e];
                }
            } else {
                v[e] = in ? w[i] : 0.0;
                lw[e] = in ? logw[i] : -INFINITY;
            }
            px[e] = in ? pose[3 * i] : 0.0f; py[e] = in ? pose[3 * i + 1] : 0.0f; pt[e] = in ? pose[3 * i + 2] : 0.0f;
        }
        if (lognorm) {                                                     // (uniform; one particle per thread)
            double mloc = lw[0] == lw[0] ? lw[0] : -INFINITY;              // a NaN product does not set the scale
#define GMS_STEP_(O) { const double v2 = wave_xor<O>(mloc); if (v2 > mloc) mloc = v2; }
            GMS_BUTTERFLY(GMS_STEP_)
#undef GMS_STEP_
            const int32_t wv = threadIdx.x >> 6, g4 = (wv >> 2) << 2;
            __syncthreads();
            if ((threadIdx.x & 63) == 0) L.b[wv] = mloc;
            __syncthreads();
            double mb = L.b[g4];
            for (int k = 1; k < 4; k++) if (L.b[g4 + k] > mb) mb = L.b[g4 + k];
            if (tl < cnt) v[0] = exp(lw[0] - lognorm_ref(mb));
        }
#pragma unroll
        for (int e = 0; e < EPB; e++) {
            const int64_t i = tl + (e0 + e) * GRP;
            if (i < cnt) {
                s += v[e];
                sq += v[e] * v[e];
                xw += (double)px[e] * v[e];                               // SLAM.java:170
                yw += (double)py[e] * v[e];                               // :171
                tw += angle_constrain((double)pt[e]) * v[e];              // :172
                if (v[e] == 0.0) nz += 1.0;
                const double key = lognorm ? lw[e] : v[e];                // (log-normalising: the strongest is the first maximum of the log-weights)
                if (key > mv) { mv = key; mx = (double)(base + i); }      // ascending index: strict > keeps the first
                else if (mx > 8.0e15 && key == key) { mv = key; mx = (double)(base + i); }
                if (lw[e] > ml) ml = lw[e];
            }
        }
    }
    GMS_STAMP(GMS_STAMP_ROW(1, blockIdx.x), 2);
    // one barrier pair for all nine columns: wave butterflies first, then the four waves of the group
    // are combined in wave order by every thread (same shape as group_sum / group_argmax)
    double sums[6] = { s, nz, sq, xw, yw, tw };
#pragma unroll
    for (int c = 0; c < 6; c++) sums[c] = wave_sum_f64(sums[c]);
    double mli = 0.0;
#define GMS_STEP_(O) { const double v2 = wave_xor<O>(mv), i2 = wave_xor<O>(mx); argmax_merge(mv, mx, v2, i2); \
                       const double l2 = wave_xor<O>(ml); if (l2 > ml) ml = l2; }
    GMS_BUTTERFLY(GMS_STEP_)
#undef GMS_STEP_
    const int32_t wave = threadIdx.x >> 6, g4 = (wave >> 2) << 2;
    GMS_STAMP(GMS_STAMP_ROW(1, blockIdx.x), 3);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 6; c++) L.m[c][wave] = sums[c];
        L.m[6][wave] = mv; L.m[7][wave] = mx; L.m[8][wave] = ml;
    }
    __syncthreads();
    GMS_STAMP(GMS_STAMP_ROW(1, blockIdx.x), 4);
#pragma unroll
    for (int c = 0; c < 6; c++) sums[c] = ((L.m[c][g4] + L.m[c][g4 + 1]) + L.m[c][g4 + 2]) + L.m[c][g4 + 3];
    mv = L.m[6][g4]; mx = L.m[7][g4]; ml = L.m[8][g4];
    for (int k = 1; k < 4; k++) {
        argmax_merge(mv, mx, L.m[6][g4 + k], L.m[7][g4 + k]);
        if (L.m[8][g4 + k] > ml) ml = L.m[8][g4 + k];
    }
    (void)mli;
    out[COL_SUM] = sums[0]; out[COL_NZ] = sums[1]; out[COL_SQ] = sums[2];
    out[COL_XW] = sums[3]; out[COL_YW] = sums[4]; out[COL_TW] = sums[5];
    out[COL_MAX] = mv; out[COL_ARG] = mx; out[COL_MLW] = ml;
}

// fold per-block partials in block order
__device__ __forceinline__ double fold_sum(const double *__restrict__ p, int64_t nblk, int col, double *lds) {
    double acc = 0.0;
    for (int64_t b = threadIdx.x & (GRP - 1); b < nblk; b += GRP) acc += p[b * GMS_PARTIAL_STRIDE + col];
    return group_sum(acc, lds);
}

// NC columns at once: the loads of all columns are in flight together and the reductions share ONE barrier pair
// (a fold_sum per column is a dependent memory round trip + two barriers each: 2-3 us of pure latency for the four
// columns the ray-cast workgroups need).  Per column the arithmetic is fold_sum's, operation for operation: the
// per-thread strided sum, the wave butterfly, ((w0 + w1) + w2) + w3 over the group's waves -- same bits.
template <int NC>
__device__ __forceinline__ void fold_sums(const double *__restrict__ p, int64_t nblk, const int (&col)[NC], double (&out)[NC],
                                          RedLds &L) {
    static_assert(NC <= GMS_PARTIAL_STRIDE, "RedLds::m has GMS_PARTIAL_STRIDE rows");
    double acc[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) acc[c] = 0.0;
    for (int64_t b = threadIdx.x & (GRP - 1); b < nblk; b += GRP) {
        double v[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) v[c] = p[b * GMS_PARTIAL_STRIDE + col[c]];
#pragma unroll
        for (int c = 0; c < NC; c++) acc[c] += v[c];
    }
#pragma unroll
    for (int c = 0; c < NC; c++) acc[c] = wave_sum_f64(acc[c]);
    const int32_t wave = threadIdx.x >> 6, g4 = (wave >> 2) << 2;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < NC; c++) L.m[c][wave] = acc[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NC; c++) out[c] = ((L.m[c][g4] + L.m[c][g4 + 1]) + L.m[c][g4 + 2]) + L.m[c][g4 + 3];
}

// The same fold over block-relative columns of a log-normalising pass (block_partials): block b's entries are multiplied by
// exp(m_b - M), m_b its own reference (COL_MLW), M the population's.  Same shape, same order.
template <int NC>
__device__ __forceinline__ void fold_sums_scaled(const double *__restrict__ p, int64_t nblk, const int (&col)[NC], double (&out)[NC],
                                                 RedLds &L, double M) {
    static_assert(NC <= GMS_PARTIAL_STRIDE, "RedLds::m has GMS_PARTIAL_STRIDE rows");
    const double R = lognorm_ref(M);
    double acc[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) acc[c] = 0.0;
    for (int64_t b = threadIdx.x & (GRP - 1); b < nblk; b += GRP) {
        double v[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) v[c] = p[b * GMS_PARTIAL_STRIDE + col[c]];
        const double sc = exp(lognorm_ref(p[b * GMS_PARTIAL_STRIDE + COL_MLW]) - R);
#pragma unroll
        for (int c = 0; c < NC; c++) acc[c] += v[c] * sc;
    }
#pragma unroll
    for (int c = 0; c < NC; c++) acc[c] = wave_sum_f64(acc[c]);
    const int32_t wave = threadIdx.x >> 6, g4 = (wave >> 2) << 2;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < NC; c++) L.m[c][wave] = acc[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NC; c++) out[c] = ((L.m[c][g4] + L.m[c][g4 + 1]) + L.m[c][g4 + 2]) + L.m[c][g4 + 3];
}

__device__ __forceinline__ void fold_argmax(const double *__restrict__ p, int64_t nblk, int colv, int coli, double &mv,
                                            double &mx, RedLds &L);
// The log-normalising fold in one piece: the population's reference M = max_b m_b, then the rescaled sums.  Up to GRP blocks
// (65 536 particles) every thread holds its block's entries from ONE round trip to memory -- this fold sits in front of the ray
// cast's pose, on the step's critical path --; more blocks take the two passes.  Every caller gets the same bits (one function).
template <int NC>
__device__ __forceinline__ double fold_lognorm(const double *__restrict__ p, int64_t nblk, const int (&col)[NC], double (&out)[NC], RedLds &L) {
    if (nblk > GRP) {
        double ml, mli;
        fold_argmax(p, nblk, COL_MLW, -1, ml, mli, L);
        fold_sums_scaled<NC>(p, nblk, col, out, L, ml);
        return ml;
    }
    const int64_t b = threadIdx.x & (GRP - 1);
    const bool in = b < nblk;
    double v[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) v[c] = in ? p[b * GMS_PARTIAL_STRIDE + col[c]] : 0.0;
    const double mb = in ? p[b * GMS_PARTIAL_STRIDE + COL_MLW] : -INFINITY;
    double mloc = mb == mb ? mb : -INFINITY;
#define GMS_STEP_(O) { const double v2 = wave_xor<O>(mloc); if (v2 > mloc) mloc = v2; }
    GMS_BUTTERFLY(GMS_STEP_)
#undef GMS_STEP_
    const int32_t wave = threadIdx.x >> 6, g4 = (wave >> 2) << 2;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) L.b[wave] = mloc;
    __syncthreads();
    double M = L.b[g4];
    for (int k = 1; k < 4; k++) if (L.b[g4 + k] > M) M = L.b[g4 + k];
    const double sc = in ? exp(lognorm_ref(mb) - lognorm_ref(M)) : 0.0;
    double acc[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) acc[c] = wave_sum_f64(0.0 + v[c] * sc);       // (0.0 + x: the strided accumulation of fold_sums_scaled, one term)
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < NC; c++) L.m[c][wave] = acc[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NC; c++) out[c] = ((L.m[c][g4] + L.m[c][g4 + 1]) + L.m[c][g4 + 2]) + L.m[c][g4 + 3];
    return M;
}

__device__ __forceinline__ void fold_argmax(const double *__restrict__ p, int64_t nblk, int colv, int coli, double &mv,
                                            double &mx, RedLds &L) {
    mv = -INFINITY; mx = 9.0e15;
    bool first = true;
    for (int64_t b = threadIdx.x & (GRP - 1); b < nblk; b += GRP) {
        const double v2 = p[b * GMS_PARTIAL_STRIDE + colv], i2 = coli >= 0 ? p[b * GMS_PARTIAL_STRIDE + coli] : (double)b;
        if (first) { mv = v2; mx = i2; first = false; } else argmax_merge(mv, mx, v2, i2);
    }
    group_argmax(mv, mx, L.a, L.b);
}

// all statistics from the (all-reduced) partial vector; every thread of the workgroup must call.  lognorm: the columns are
// block-relative (block_partials): *ref_out receives M, the population's largest log-weight, the returned sum is sum exp(logw - M).
__device__ __forceinline__ double fold_stats(const double *__restrict__ p, int64_t nblk, PfStatsDev *s, bool write,
                                             const float *__restrict__ pose, int64_t pose_base, int64_t pose_n, RedLds &L,
                                             const PackedParticle *__restrict__ glob = nullptr, bool lognorm = false, double *ref_out = nullptr) {
    double ml = 0.0, mli;
    if (!write) {
        if (!lognorm) return fold_sum(p, nblk, COL_SUM, L.a);
        const int c1[1] = { COL_SUM };
        double f1[1];
        ml = fold_lognorm<1>(p, nblk, c1, f1, L);
        if (ref_out) *ref_out = ml;
        return f1[0];
    }
    const int cols[5] = { COL_SUM, COL_NZ, COL_XW, COL_YW, COL_TW };
    double f[5];
    if (lognorm) { ml = fold_lognorm<5>(p, nblk, cols, f, L); if (ref_out) *ref_out = ml; }
    else fold_sums<5>(p, nblk, cols, f, L);
    const double sum = f[0], nz = f[1], xw = f[2], yw = f[3], tw = f[4];
    double mv, mx;
    fold_argmax(p, nblk, COL_MAX, COL_ARG, mv, mx, L);
    if (!lognorm) fold_argmax(p, nblk, COL_MLW, -1, ml, mli, L);
    if (threadIdx.x == 0) {
        s->n_ambiguous = 0;
        s->weight_sum = sum;
        s->max_w = lognorm ? exp(ml - lognorm_ref(ml)) : mv;            // (the largest rescaled weight is 1)
        s->strongest = mx < 8.0e15 ? (int32_t)mx : 0;
        if (!lognorm) s->n_zero = (int32_t)nz;                          // (log-normalising: counted where the weights are formed, normalize_pack_body)
        s->max_logw = ml;
        s->xs = xw; s->ys = yw; s->ts = tw;
        s->wpose[0] = (float)(xw / sum);                               // SLAM.java:176
        s->wpose[1] = (float)(yw / sum);
        s->wpose[2] = (float)(tw / sum);
        // strongest particle's pose, when this shard holds it
        const int64_t st = (int64_t)s->strongest - pose_base;
        if (glob) {                                                    // the gathered population is at hand
            const PackedParticle pp = glob[s->strongest];
            s->spose[0] = pp.x; s->spose[1] = pp.y; s->spose[2] = pp.theta;
        } else if (pose && st >= 0 && st < pose_n) {
            s->spose[0] = pose[3 * st]; s->spose[1] = pose[3 * st + 1]; s->spose[2] = pose[3 * st + 2];
        }
    }
    return sum;
}

__device__ __forceinline__ int64_t block_count(int64_t n, int64_t blk) {
    const int64_t left = n - blk * GMS_BLOCK;
    return left < 0 ? 0 : (left > GMS_BLOCK ? GMS_BLOCK : left);
}

// calculateNeff (SLAM.java:180-190) from the per-block {sum wn, sum wn^2} of the normalised population:
// sum = fold(sum wn); sq_sum = fold(sum wn^2) / sum^2; Neff = 1 / sq_sum.  Every thread of the workgroup calls.
__device__ __forceinline__ void fold_neff(const double *__restrict__ p2, int64_t nblk, double &norm_sum, double &sq_sum,
                                          double *lds) {
    double a = 0.0, q = 0.0;
    for (int64_t b = threadIdx.x & (GRP - 1); b < nblk; b += GRP) { a += p2[2 * b]; q += p2[2 * b + 1]; }
    norm_sum = group_sum(a, lds);
    const double qq = group_sum(q, lds);
    sq_sum = qq / (norm_sum * norm_sum);
}

__global__ void __launch_bounds__(256)
k_fold_neff(const double *__restrict__ p2_all, int64_t nblk, PfStatsDev *__restrict__ stats) {
    __shared__ RedLds L;
    const int32_t mi = blockIdx.x;
    double ns, sq;
    fold_neff(p2_all + (size_t)mi * nblk * 2, nblk, ns, sq, L.a);
    if (threadIdx.x == 0) { stats[mi].norm_sum = ns; stats[mi].sq_sum = sq; }
}

// phase 1: this shard's block partials at their global slots; blocks of other shards are zeroed so
// that an all-reduce(SUM) assembles the full vector exactly.  grid = (nblk_global, n_maps).
__device__ __forceinline__ void
partials_body(double *__restrict__ w, double *__restrict__ logw, const float *__restrict__ pose, int32_t n,
              int64_t offset, int64_t nblk_global, double *__restrict__ partials, const double *__restrict__ part,
              int32_t part_nseg, uint32_t bx, uint32_t by, PfStatsDev *__restrict__ lognorm_stats = nullptr) {
    // lognorm_stats != nullptr: the log-normalising form (block_partials); the zero census of this pass is counted where the
    // weights are formed (normalize_pack_body adds to it): cleared here, one launch ahead
    __shared__ RedLds L;
    const int32_t mi = (int32_t)by;
    if (lognorm_stats && bx == 0 && threadIdx.x == 0) lognorm_stats[mi].n_zero = 0;
    const int64_t gb = bx;
    const int64_t lb = gb - offset / GMS_BLOCK;                       // block index inside the shard
    const int64_t nlb = ((int64_t)n + GMS_BLOCK - 1) / GMS_BLOCK;
    double *p = partials + ((size_t)mi * nblk_global + gb) * GMS_PARTIAL_STRIDE;
    if (lb < 0 || lb >= nlb) {                                        // uniform per workgroup
        if (threadIdx.x < GMS_PARTIAL_STRIDE) p[threadIdx.x] = 0.0;
        return;
    }
    double out[GMS_PARTIAL_STRIDE];
    const size_t o = (size_t)mi * n + lb * GMS_BLOCK;
    block_partials(w + o, logw + o, pose + 3 * o, block_count(n, lb), offset + lb * GMS_BLOCK, out, L, part, mi, n,
                   part_nseg, lb * GMS_BLOCK, lognorm_stats != nullptr);
    if (threadIdx.x == 0)
        for (int k = 0; k < GMS_PARTIAL_STRIDE; k++) p[k] = out[k];
}

__global__ void __launch_bounds__(256)
k_partials(double *__restrict__ w, double *__restrict__ logw, const float *__restrict__ pose, int32_t n,
           int64_t offset, int64_t nblk_global, double *__restrict__ partials, const double *__restrict__ part,
           int32_t part_nseg, PfStatsDev *__restrict__ lognorm_stats) {
    GMS_STAMP(GMS_STAMP_ROW(1, blockIdx.x), 0);
    partials_body(w, logw, pose, n, offset, nblk_global, partials, part, part_nseg, blockIdx.x, blockIdx.y, lognorm_stats);
    GMS_STAMP(GMS_STAMP_ROW(1, blockIdx.x), 1);
}

// Level 0 of the cumulative weights: one wavefront = one 64-particle chunk, lane = particle, inclusive scan by
// six shuffle steps (a fixed shape, identical wherever it runs: every rank of a sharded filter, the normalise pass
// and the pass over a gathered population all produce the same bits); a 256-thread workgroup = one reduction
// block = four chunks, whose {sum wn, sum wn^2} are combined in chunk order for calculateNeff (SLAM.java:180-190).
// Every thread of the workgroup calls; v = normalised weight of particle i of this map (0 beyond the population).
// The in-chunk cumulative weight at the end of every octet (lanes 7, 15, ... 63) also goes to sub[chunk][8], compactly, for the first
// 8-way step of the resampling search (lanes beyond the population add 0.0: their value is the chunk's last one, which is what a
// search clamped to the chunk's length reads).  sub follows the chunk totals of all maps in the same allocation (chunk_sub_of).
__device__ __forceinline__ double *chunk_sub_of(double *chunk_tot_all, int64_t nchunks, int32_t n_maps, int32_t mi) {
    return chunk_tot_all + (size_t)n_maps * (nchunks + 1) + (size_t)mi * nchunks * 8;
}
__device__ __forceinline__ void block_chunk_scan(double v, int64_t i, int64_t n_pop, int64_t blk, int64_t nchunks,
                                                 double *__restrict__ cum, double *__restrict__ chunk_tot,
                                                 double *__restrict__ p2_blk, double *__restrict__ sub) {
    __shared__ double s_ct[4][2];
    const int32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double up = __shfl_up(inc, o, GMS_WAVE);
        if (lane >= o) inc += up;
    }
    const double sq = wave_sum_f64(v * v);
    const double tot = __shfl(inc, 63, GMS_WAVE);
    if (i < n_pop) cum[i] = inc;
    if ((lane & 7) == 7 && blk * 4 + wave < nchunks) sub[(blk * 4 + wave) * 8 + (lane >> 3)] = inc;
    if (lane == 0) {
        const int64_t c = blk * 4 + wave;
        if (c < nchunks) chunk_tot[c] = tot;
        s_ct[wave][0] = tot; s_ct[wave][1] = sq;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        p2_blk[0] = ((s_ct[0][0] + s_ct[1][0]) + s_ct[2][0]) + s_ct[3][0];
        p2_blk[1] = ((s_ct[0][1] + s_ct[1][1]) + s_ct[2][1]) + s_ct[3][1];
    }
}
static_assert(4 * SCAN_CHUNK == GMS_BLOCK, "a reduction block is four scan chunks");

// phase 2: every workgroup folds the (all-reduced) partials; weight /= weightSum (SLAM.java:120-121);
// packs {w,x,y,theta} (the all-gather payload / the resampling source).  A stand-alone filter also
// gets the dense weight copy and level 0 of the cumulative weights here (cum != nullptr).
__device__ __forceinline__ void
normalize_pack_body(const double *__restrict__ partials_all, int64_t nblk_global, double *__restrict__ w,
                    const float *__restrict__ pose, int32_t n, int64_t offset, PackedParticle *__restrict__ packed,
                    double *__restrict__ cum, double *__restrict__ chunk_tot, int64_t nchunks,
                    double *__restrict__ p2_all, PfStatsDev *__restrict__ stats, uint32_t bx, uint32_t by,
                    const double *__restrict__ logw_lognorm = nullptr) {
    // logw_lognorm != nullptr (gms_pf_set_log_normalize): the partial vector is block-relative (block_partials) and the weights are
    // formed here from the log-weights, exp(logw - M) / sum
    __shared__ RedLds L;
    const int32_t mi = (int32_t)by;
    const double *p = partials_all + (size_t)mi * nblk_global * GMS_PARTIAL_STRIDE;
    double M = 0.0;
    const double sum = fold_stats(p, nblk_global, stats + mi, bx == 0, pose + (size_t)mi * n * 3, offset, n, L, nullptr, logw_lognorm != nullptr, &M);
    const int32_t i = (int32_t)bx * 256 + threadIdx.x;
    double wn = 0.0;
    if (logw_lognorm) {                                                    // (uniform)
        const double v = i < n ? exp(logw_lognorm[(size_t)mi * n + i] - lognorm_ref(M)) : 1.0;
        const int32_t zeros = __popcll(__ballot(v == 0.0));
        if ((threadIdx.x & 63) == 0 && zeros) atomicAdd(&stats[mi].n_zero, zeros);      // (cleared by the partials launch)
        if (i < n) w[(size_t)mi * n + i] = v;
    }
    if (i < n) {
        const size_t gi = (size_t)mi * n + i;
        wn = w[gi] / sum;
        w[gi] = wn;
        PackedParticle pp;
        pp.w = wn; pp.x = pose[3 * gi]; pp.y = pose[3 * gi + 1]; pp.theta = pose[3 * gi + 2]; pp.pad = 0u;
        packed[gi] = pp;
    }
    if (cum)                                          // uniform
        block_chunk_scan(wn, i, n, bx, nchunks, cum + (size_t)mi * n, chunk_tot + (size_t)mi * (nchunks + 1),
                         p2_all + ((size_t)mi * nblk_global + bx) * 2, chunk_sub_of(chunk_tot, nchunks, (int32_t)gridDim.y, mi));
}

__global__ void __launch_bounds__(256)
k_normalize_pack(const double *__restrict__ partials_all, int64_t nblk_global, double *__restrict__ w,
                 const float *__restrict__ pose, int32_t n, int64_t offset, PackedParticle *__restrict__ packed,
                 double *__restrict__ cum, double *__restrict__ chunk_tot, int64_t nchunks,
                 double *__restrict__ p2_all, PfStatsDev *__restrict__ stats, const double *__restrict__ logw_lognorm) {
    normalize_pack_body(partials_all, nblk_global, w, pose, n, offset, packed, cum, chunk_tot, nchunks, p2_all, stats, blockIdx.x,
                        blockIdx.y, logw_lognorm);
}

// Level 0 as a kernel of its own (sharded filters after the all-gather; resample without normalise): the same
// scan over the GLOBAL population, plus the strongest particle's pose, which only the owning rank knew.
// 256 threads per workgroup, one particle per thread.
__device__ __forceinline__ void
chunk_sums_body(const PackedParticle *__restrict__ glob_all, int64_t n_global, int64_t nchunks, double *__restrict__ cum_all,
                double *__restrict__ chunk_tot, double *__restrict__ p2_all, int64_t nblk_global, PfStatsDev *__restrict__ stats,
                uint32_t bx, uint32_t by) {
    const int32_t mi = (int32_t)by;
    const PackedParticle *g = glob_all + (size_t)mi * n_global;
    if (bx == 0 && threadIdx.x == 0) {
        stats[mi].n_ambiguous = 0;
        int64_t st = stats[mi].strongest;
        if (st < 0) st = 0;
        if (st >= n_global) st = n_global - 1;
        const PackedParticle pp = g[st];
        stats[mi].spose[0] = pp.x; stats[mi].spose[1] = pp.y; stats[mi].spose[2] = pp.theta;
    }
    const int64_t i = (int64_t)bx * 256 + threadIdx.x;
    const double v = i < n_global ? g[i].w : 0.0;
    block_chunk_scan(v, i, n_global, bx, nchunks, cum_all + (size_t)mi * n_global, chunk_tot + (size_t)mi * (nchunks + 1),
                     p2_all + ((size_t)mi * nblk_global + bx) * 2, chunk_sub_of(chunk_tot, nchunks, (int32_t)gridDim.y, mi));
}

__global__ void __launch_bounds__(256)
k_chunk_sums(const PackedParticle *__restrict__ glob_all, int64_t n_global, int64_t nchunks, double *__restrict__ cum_all,
             double *__restrict__ chunk_tot, double *__restrict__ p2_all, int64_t nblk_global, PfStatsDev *__restrict__ stats) {
    chunk_sums_body(glob_all, n_global, nchunks, cum_all, chunk_tot, p2_all, nblk_global, stats, blockIdx.x, blockIdx.y);
}

// ---- sharded filters whose ranks exchange RAW weights (one all-gather per scan; gms_slam_update_sharded_*) ----
// After the gather every rank holds all raw weights, poses and block partials.  This body is `weight /= weightSum`
// (SLAM.java:120-121) for the rank's own particles plus the statistics; nothing is packed (the raw pack travelled).
__device__ __forceinline__ void
normalize_own_body(const double *__restrict__ partials, int64_t nblk_global, double *__restrict__ w, const float *__restrict__ pose,
                   int32_t n, int64_t offset, const PackedParticle *__restrict__ glob_raw, PfStatsDev *__restrict__ stats,
                   uint32_t bx) {
    __shared__ RedLds L;
    const double sum = fold_stats(partials, nblk_global, stats, bx == 0, pose, offset, n, L, glob_raw);
    const int32_t i = (int32_t)bx * 256 + threadIdx.x;
    if (i < n) w[i] = w[i] / sum;
}

// Level 0 of the cumulative NORMALISED weights of the gathered raw population: the division the owner performs
// (same operands, same result), then the scan of block_chunk_scan.  Every workgroup folds the weight sum itself.
__device__ __forceinline__ void
chunk_sums_raw_body(const PackedParticle *__restrict__ glob_raw, int64_t n_global, int64_t nchunks, double *__restrict__ cum,
                    double *__restrict__ chunk_tot, double *__restrict__ p2, int64_t nblk_global,
                    const double *__restrict__ partials, uint32_t bx) {
    __shared__ RedLds L;
    const double sum = fold_sum(partials, nblk_global, COL_SUM, L.a);
    const int64_t i = (int64_t)bx * 256 + threadIdx.x;
    const double v = i < n_global ? glob_raw[i].w / sum : 0.0;
    block_chunk_scan(v, i, n_global, bx, nchunks, cum, chunk_tot, p2 + (size_t)bx * 2, chunk_sub_of(chunk_tot, nchunks, 1, 0));
}

// raw pack of one reduction block (the all-gather payload): weight as scored, pose
__device__ __forceinline__ void pack_raw_block(const double *__restrict__ w, const float *__restrict__ pose, int32_t n, int64_t lb,
                                               PackedParticle *__restrict__ packed_local) {
    const int64_t i = lb * GMS_BLOCK + threadIdx.x;
    if (i < n) {
        PackedParticle pp;
        pp.w = w[i]; pp.x = pose[3 * i]; pp.y = pose[3 * i + 1]; pp.theta = pose[3 * i + 2]; pp.pad = 0u;
        packed_local[i] = pp;
    }
}

// statistics only (getWeightedPose / calculateNeff on the current particles, nothing rewritten)
__global__ void __launch_bounds__(256)
k_stats_only(const double *__restrict__ partials_all, int64_t nblk_global, const float *__restrict__ pose, int32_t n,
             int64_t offset, PfStatsDev *__restrict__ stats) {
    __shared__ RedLds L;
    const int32_t mi = blockIdx.x;
    fold_stats(partials_all + (size_t)mi * nblk_global * GMS_PARTIAL_STRIDE, nblk_global, stats + mi, true,
               pose + (size_t)mi * n * 3, offset, n, L);
}

// pack without normalising
__global__ void k_pack(const double *__restrict__ w, const float *__restrict__ pose, int32_t n,
                       PackedParticle *__restrict__ packed) {
    const int32_t mi = blockIdx.y;
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t gi = (size_t)mi * n + i;
    PackedParticle pp;
    pp.w = w[gi]; pp.x = pose[3 * gi]; pp.y = pose[3 * gi + 1]; pp.theta = pose[3 * gi + 2]; pp.pad = 0u;
    packed[gi] = pp;
}

// ---------------------------------------------------------------------------------------------
// resampling.  Cumulative weights, fixed shape (depends on n_global only, so every rank of a sharded
// filter computes the same values):
//   level 0  chunk of SCAN_CHUNK = 64 weights: one wavefront's inclusive shuffle scan (block_chunk_scan);
//   level 1  super-chunk of 64 chunks: one wavefront's shuffle scan of the chunk totals;
//   level 2  wavefront 0 scans the super-chunk totals, 64 per pass, passes chained in order.
// offset[c] = level2[c / 64] + level1[c]; cumulative weight of particle i = offset[i / 64] + cum[i].
// Levels 1 and 2 are a few hundred additions: every workgroup of k_resample redoes them in LDS.
// ---------------------------------------------------------------------------------------------
// one lane per output slot (SLAM.java:140-149); the chunk offsets are staged in LDS for the first search level
__device__ __forceinline__ void
resample_body(const PackedParticle *__restrict__ glob_all, int64_t n_global, int64_t nchunks,
              const double *__restrict__ cum_all, const double *__restrict__ chunk_off, const double *__restrict__ r01,
              double r01_scalar, double fraction, int32_t n, int64_t offset, float *__restrict__ pose2, float *__restrict__ cs2,
              double *__restrict__ w2, int32_t *__restrict__ idx_out, const double *__restrict__ p2_all, int64_t nblk_global,
              PfStatsDev *__restrict__ stats, uint32_t bx, uint32_t by, unsigned char *smem, bool raw_weights = false) {
    double *off = reinterpret_cast<double *>(smem);                    // [nchunks + 1]
    const int32_t mi = (int32_t)by;
    __shared__ RedLds L;
    // The chunk totals are loaded BEFORE the Neff fold decides whether the resample runs (they sit in registers meanwhile):
    // one global round trip fewer on the critical path of the paired launch.  (More than 8 * blockDim chunks: the rest in
    // the loop below.)
    const double *tot = chunk_off + (size_t)mi * (nchunks + 1);
    double tv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { const int64_t c = (int64_t)threadIdx.x + k * (int64_t)blockDim.x; tv[k] = c < nchunks ? tot[c] : 0.0; }
    // Populations of at most RES_SUB_MAX_CHUNKS chunks: the in-chunk cumulative weight at the end of every octet, eight per chunk
    // (a compact table the scan's level 0 leaves behind the chunk totals), goes to LDS as well (loaded here, beside the chunk totals: the
    // loads fly during the Neff fold).  The search below then takes
    // its first 8-way step from LDS instead of from memory: one dependent round trip fewer per output slot.
    const bool use_sub = nchunks <= RES_SUB_MAX_CHUNKS;
    double *sub = reinterpret_cast<double *>(smem) + (nchunks + 1) + ((nchunks + 63) / 64 + 1);        // [nchunks][8]
    const double *sub_g = chunk_sub_of(const_cast<double *>(chunk_off), nchunks, (int32_t)gridDim.y, mi);       // written with the chunk totals (block_chunk_scan)
    if (use_sub) {
        for (int64_t e0 = (int64_t)threadIdx.x; e0 < nchunks * 8; e0 += 8 * (int64_t)blockDim.x) {      // eight coalesced loads in flight per thread
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { const int64_t e = e0 + k * (int64_t)blockDim.x; v[k] = e < nchunks * 8 ? sub_g[e] : 0.0; }
#pragma unroll
            for (int k = 0; k < 8; k++) { const int64_t e = e0 + k * (int64_t)blockDim.x; if (e < nchunks * 8) sub[e] = v[k]; }
        }
    }
    double norm_sum, sq_sum;
    fold_neff(p2_all + (size_t)mi * nblk_global * 2, nblk_global, norm_sum, sq_sum, L.a);   // calculateNeff (SLAM.java:180-190)
    if (bx == 0 && threadIdx.x == 0) { stats[mi].norm_sum = norm_sum; stats[mi].sq_sum = sq_sum; }
    GMS_STAMP(GMS_STAMP_ROW(3, blockIdx.x), 5);                                              // (development builds) Neff folded
    const bool go = fraction < 0.0 || (1.0 / sq_sum) < fraction * (double)n_global;         // GridMapApp.java:185
    const int64_t nsuper = (nchunks + 63) / 64;
    double *sup = off + nchunks + 1;                                   // [nsuper + 1]
    if (go) {
#pragma unroll
        for (int k = 0; k < 8; k++) { const int64_t c = (int64_t)threadIdx.x + k * (int64_t)blockDim.x; if (c < nchunks) off[c] = tv[k]; }
        for (int64_t c0 = (int64_t)threadIdx.x + 8 * (int64_t)blockDim.x; c0 < nchunks; c0 += 8 * (int64_t)blockDim.x) {  // eight loads in flight per thread
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { const int64_t c = c0 + k * (int64_t)blockDim.x; v[k] = c < nchunks ? tot[c] : 0.0; }
#pragma unroll
            for (int k = 0; k < 8; k++) { const int64_t c = c0 + k * (int64_t)blockDim.x; if (c < nchunks) off[c] = v[k]; }
        }
        __syncthreads();
        // level 1: one wavefront per super-chunk of 64 chunk totals, exclusive scan by shuffles (a fixed shape)
        for (int64_t sidx = threadIdx.x >> 6; sidx < nsuper; sidx += blockDim.x >> 6) {
            const int32_t lane = threadIdx.x & 63;
            const int64_t c = sidx * 64 + lane;
            const double v = c < nchunks ? off[c] : 0.0;
            double inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const double up = __shfl_up(inc, o, GMS_WAVE);
                if (lane >= o) inc += up;
            }
            const double excl = __shfl_up(inc, 1, GMS_WAVE);
            if (c < nchunks) off[c] = lane == 0 ? 0.0 : excl;
            if (lane == 63) sup[sidx] = inc;
        }
        __syncthreads();
        if (threadIdx.x < 64) {                                                   // level 2: wavefront 0, 64 totals per pass
            const int32_t lane = threadIdx.x;
            double carry = 0.0;
            for (int64_t s0 = 0; s0 < nsuper; s0 += 64) {
                const int64_t sidx = s0 + lane;
                const double v = sidx < nsuper ? sup[sidx] : 0.0;
                double inc = v;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const double up = __shfl_up(inc, o, GMS_WAVE);
                    if (lane >= o) inc += up;
                }
                const double excl = __shfl_up(inc, 1, GMS_WAVE);
                if (sidx < nsuper) sup[sidx] = s0 == 0 ? (lane == 0 ? 0.0 : excl) : carry + (lane == 0 ? 0.0 : excl);
                const double tot = __shfl(inc, 63, GMS_WAVE);
                carry = s0 == 0 ? tot : carry + tot;
            }
            if (lane == 0) off[nchunks] = carry;                                  // grand total
        }
        __syncthreads();
        for (int64_t c = 64 + threadIdx.x; c < nchunks; c += blockDim.x) off[c] = sup[c >> 6] + off[c];   // super-chunk 0 adds nothing
        __syncthreads();
    }
    GMS_STAMP(GMS_STAMP_ROW(3, blockIdx.x), 6);                                              // chunk offsets scanned
    const int32_t t = (int32_t)bx * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const PackedParticle *g = glob_all + (size_t)mi * n_global;
    const int64_t m0 = offset + t;              // m - 1
    int64_t src = m0;
    if (go) {
        const double *cm = cum_all + (size_t)mi * n_global;
        const double N = (double)n_global;
        const double r = (r01 ? r01[mi] : r01_scalar) * 1.0 / N;        // SLAM.java:136
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
            // first j in the chunk with !(U > c_j): two 8-way steps, the eight candidates of a step loaded together
            // (two dependent round trips to memory instead of six for a bisection; same answer on a sorted chunk)
            const int64_t len = min((int64_t)SCAN_CHUNK, n_global - i0);
            double cv[8];
#pragma unroll
            for (int k = 0; k < 8; k++) cv[k] = use_sub ? sub[lo * 8 + k] : cm[i0 + min((int64_t)(8 * k + 7), len - 1)];
            int32_t oct = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) oct += (8 * k + 7 < len && U > base + cv[k]) ? 1 : 0;
            // U can exceed the chunk's last in-chunk value by a rounding (off[] comes from another association of the same sums): the
            // octet is clamped to the last one that holds a particle -- also in a final partial chunk whose length is a multiple of
            // 8 -- so that src's own octet is the one loaded below and c_prev is cm[src - 1]
            const int32_t oct_max = (int32_t)((len - 1) >> 3);
            if (oct > oct_max) oct = oct_max;
            const double before_oct = oct > 0 ? cv[oct - 1] : 0.0;      // in-chunk cumulative weight just before this octet (oct > 0)
#pragma unroll
            for (int k = 0; k < 8; k++) cv[k] = cm[i0 + min((int64_t)(8 * oct + k), len - 1)];
            int64_t a = 8 * oct;
#pragma unroll
            for (int k = 0; k < 8; k++) a += (8 * oct + k < len && U > base + cv[k]) ? 1 : 0;
            src = i0 + (a < len ? a : len - 1);
            // boundary within rounding distance of U: a sequential scan may choose a neighbour.  cm[src] and cm[src - 1] are
            // among the values already at hand (the octet's eight, the octet boundary before it, the previous chunk's end):
            // the same operands as loading them again, without the round trip.
            const double tol = N * 4.5e-16 * fabs(off[nchunks]);
            const int32_t ki = (int32_t)(src - i0) - 8 * oct;           // 0..7: where cm[src] sits in cv[] (src lies in octet `oct`: see the clamp above)
            double c_src = cv[0], c_prev = before_oct;
#pragma unroll
            for (int k = 1; k < 8; k++) if (ki == k) { c_src = cv[k]; c_prev = cv[k - 1]; }
            bool amb = fabs(U - (base + c_src)) <= tol;
            if (src > 0) {
                if (ki > 0 || oct > 0) amb = amb || fabs(U - (base + c_prev)) <= tol;                  // src - 1 lies in this chunk
                else amb = amb || fabs(U - (off[lo - 1] + (use_sub ? sub[(lo - 1) * 8 + 7] : cm[src - 1]))) <= tol;   // ... ends the chunk before
            }
            if (amb) atomicAdd(&stats[mi].n_ambiguous, 1);
        }
    }
    GMS_STAMP(GMS_STAMP_ROW(3, blockIdx.x), 7);                                              // source found
    const PackedParticle pp = g[src];
    const size_t o = (size_t)mi * n + t;
    pose2[3 * o] = pp.x; pose2[3 * o + 1] = pp.y; pose2[3 * o + 2] = pp.theta;
    float tc, ts;
    pose_trig(pp.theta, tc, ts);                                        // keep cs[] in step with pose[]
    cs2[2 * o] = tc; cs2[2 * o + 1] = ts;
    // copies keep their (normalised) weight (SLAM.java:42); a gathered RAW population is normalised here, by the
    // division its owner performs
    w2[o] = raw_weights ? pp.w / stats[mi].weight_sum : pp.w;
    if (idx_out) idx_out[o] = (int32_t)src;
    if (t == 0) stats[mi].did_resample = go ? 1 : 0;
}

__global__ void __launch_bounds__(256)
k_resample(const PackedParticle *__restrict__ glob_all, int64_t n_global, int64_t nchunks,
           const double *__restrict__ cum_all, const double *__restrict__ chunk_off, const double *__restrict__ r01,
           double r01_scalar, double fraction, int32_t n, int64_t offset, float *__restrict__ pose2, float *__restrict__ cs2,
           double *__restrict__ w2, int32_t *__restrict__ idx_out, const double *__restrict__ p2_all, int64_t nblk_global,
           PfStatsDev *__restrict__ stats, int32_t raw_weights, int32_t *__restrict__ epoch2) {
    extern __shared__ __align__(16) unsigned char smem[];
    resample_body(glob_all, n_global, nchunks, cum_all, chunk_off, r01, r01_scalar, fraction, n, offset, pose2, cs2, w2, idx_out,
                  p2_all, nblk_global, stats, blockIdx.x, blockIdx.y, smem, raw_weights != 0);
    // a filter whose particles own maps (gms_slam): which generation of the maps is current is a device-side fact -- epoch2[0] counts the
    // draws that ran (the generation is its parity), epoch2[1] says whether this one did; the thread that published did_resample adds it up
    if (epoch2 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        const int32_t did = stats[0].did_resample;
        epoch2[1] = did;
        if (did) epoch2[0] = epoch2[0] + 1;
    }
}

// ---------------------------------------------------------------------------------------------
// GridMap.findBestPose (J/slam/GridMap.java:319-346): lattice search around every particle.
// One workgroup per particle, ONE LANE PER LATTICE POSE (1210 poses over 256 lanes): every lane walks the hit beams in
// order and multiplies its factors sequentially -- the reference's own order and its plain double product
// (GridMap.java:262-288), with no cross-lane reduction per pose -- then one argmax over the lattice with the reference's
// rule: strict `>` against maxProb = 0 in loop order, i.e. the first maximum wins (:334).  The 64 end points of a
// wave-instruction are one beam seen from 64 neighbouring lattice poses: a compact, L1-resident patch.
// (Round 1 gave a whole wavefront to each lattice pose: the trig of the pose was computed 64 times over and every pose
// paid a 6-step shuffle product: 1.26 ms for 1024 particles x 360 beams against 0.62 ms for this form, 7e11 beam-evals/s.)
// ---------------------------------------------------------------------------------------------
#define REFINE_MAX_STEPS 16
__global__ void __launch_bounds__(256)
k_refine(GridDev g, const double *__restrict__ fac_all, int64_t fac_stride, const double *__restrict__ hitbeams,
         const int32_t *__restrict__ nhit, int32_t beam_stride, float *__restrict__ pose, float *__restrict__ cs,
         int32_t n) {
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
    const float x0 = pose[3 * gi], y0 = pose[3 * gi + 1], t0 = pose[3 * gi + 2];
    const double *fac = fac_all + (size_t)mi * fac_stride;
    const int32_t nx = s_n[0], ny = s_n[1], nt = s_n[2], total = nx * ny * nt;
    const int32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double best = 0.0;            // maxProb = 0 (:321)
    int32_t besti = -1;           // -1 = keep the start pose (:320)
    for (int32_t q = threadIdx.x; q < total; q += 256) {               // q = loop order of the reference: theta fastest (:328-330)
        const int32_t it = q % nt, iy = (q / nt) % ny, ix = q / (nt * ny);
        const float cx = x0 + s_dx[ix], cy = y0 + s_dy[iy], ct = t0 + s_dt[it];       // :332
        float c, sn;
        pose_trig(ct, c, sn);
        XformDev t;
        t.c = (double)c; t.s = (double)sn; t.px = (double)cx; t.py = (double)cy;
        double prod = 1.0;                                             // GridMap.java:262
        for (int32_t j = 0; j < nb; j++) {                             // beams in order, sequential product (:267-288)
            const double2 bm = sb[j];                                  // same address in every lane: LDS broadcast
            bool guard = false;
            uint32_t cell = beam_cell_fast(g, t, bm.x, bm.y, guard);   // no division in the common case (see j_cell_fast)
            if (__builtin_expect(guard, 0)) cell = beam_cell(g, t, bm.x, bm.y);
            prod *= fac[cell];
        }
        if (prod > best) { best = prod; besti = q; }                   // :334 (q ascending per lane)
    }
    // first maximum over the lattice: larger probability wins, equal probabilities the smaller q
#define GMS_STEP_(O) { const double v2 = wave_xor<O>(best); const int32_t i2 = wave_xor<O>(besti); \
                       if (i2 >= 0 && (besti < 0 || v2 > best || (v2 == best && i2 < besti))) { best = v2; besti = i2; } }
    GMS_BUTTERFLY(GMS_STEP_)
#undef GMS_STEP_
    if (lane == 0) { s_best[wave] = best; s_besti[wave] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double b = 0.0; int32_t bi = -1;
        for (int k = 0; k < 4; k++) {
            const double v = s_best[k]; const int32_t vi = s_besti[k];
            if (vi >= 0 && (bi < 0 || v > b || (v == b && vi < bi))) { b = v; bi = vi; }
        }
        if (bi >= 0) {
            const int32_t it = bi % nt, iy = (bi / nt) % ny, ix = bi / (nt * ny);
            pose[3 * gi] = x0 + s_dx[ix]; pose[3 * gi + 1] = y0 + s_dy[iy]; pose[3 * gi + 2] = t0 + s_dt[it];
            float tc, ts;
            pose_trig(t0 + s_dt[it], tc, ts);
            cs[2 * gi] = tc; cs[2 * gi + 1] = ts;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Reference-order audit (gms_pf_set_reference_order; tests, slow on purpose).  The default kernels re-associate three chains of the
// reference: the product of a scan's factors (16 segment products, combine_segments), weightSum (blocked sums) and the cumulative
// weights of resample() (a three-level scan).  Here each is ONE chain in the reference's own order -- so that a test can demand
// equality with the oracle, bit for bit, zeros and denormals included, and then compare the default path with THIS path: whatever
// differs between the two is association and nothing else.
// ---------------------------------------------------------------------------------------------
// probabilityOf (GridMap.java:261-294): lane = particle, the product of ALL hit beams' factors in one register, in beam order.  The
// beams pass through the LDS table 128 at a time; log-weight = sum over those chunks of log(chunk product) (every chunk product is
// a normal double: >= 0.01^128).
__global__ void __launch_bounds__(256)
k_score_seq(GridDev g, const double *__restrict__ fac_all, int64_t fac_stride, const gms_beam *__restrict__ beams, int32_t B,
            int32_t beam_stride, const float *__restrict__ pose, const float *__restrict__ cs, int32_t n, double *__restrict__ w,
            double *__restrict__ logw) {
    __shared__ double2 s_beam[128];
    __shared__ int32_t s_nb;
    const int32_t mi = blockIdx.y;
    const int32_t p = blockIdx.x * 256 + threadIdx.x;
    const size_t gi = (size_t)mi * n + (p < n ? p : 0);
    const gms_beam *mb = beams + (size_t)mi * beam_stride;
    const double *fac = fac_all + (size_t)mi * fac_stride;
    XformDev t;
    t.c = (double)cs[2 * gi]; t.s = (double)cs[2 * gi + 1]; t.px = (double)pose[3 * gi]; t.py = (double)pose[3 * gi + 1];
    double prod = 1.0, lsum = 0.0;                                     // GridMap.java:262
    for (int32_t j0 = 0; j0 < B; j0 += 128) {
        __syncthreads();                                               // the previous chunk has been consumed
        if (threadIdx.x < 64) {                                        // order-preserving compaction of the chunk's hit beams (:269)
            const int32_t j1 = min(B, j0 + 128);
            int32_t base = 0;
            for (int32_t b0 = j0; b0 < j1; b0 += 64) {
                const int32_t b = b0 + (int32_t)threadIdx.x;
                const bool hit = b < j1 && mb[b].hit != 0;
                const unsigned long long mask = __ballot(hit);
                if (hit) s_beam[base + __popcll(mask & ((1ull << threadIdx.x) - 1ull))] = make_double2(mb[b].local_x, mb[b].local_y);
                base += __popcll(mask);
            }
            if (threadIdx.x == 0) s_nb = base;
        }
        __syncthreads();
        const int32_t nb = s_nb;
        double cp = 1.0;
        for (int32_t k = 0; k < nb; k++) {
            const double2 bm = s_beam[k];
            const double f = fac[beam_cell(g, t, bm.x, bm.y)];         // the reference's own quotients (:273-288)
            prod *= f;                                                 // product *= ... (:286-288)
            cp *= f;
        }
        int e;
        const double mnt = frexp(cp, &e);
        lsum += log(mnt) + (double)e * 0.6931471805599453;
    }
    if (p < n) { w[gi] = prod; logw[gi] = lsum; }
}

// SLAM.update's bookkeeping as the reference's own loops (SLAM.java:87-124, calculateNeff :180-190, getWeightedPose :165-178): ONE
// lane adds in particle order.  The weights pass through LDS 2048 at a time (every thread loads, lane 0 adds).  normalise = 0: the
// statistics of the weights as they stand (nothing rewritten).  One workgroup per map.
#define SEQ_CHUNK 2048
__global__ void __launch_bounds__(256)
k_normalize_seq(double *__restrict__ w_all, const double *__restrict__ logw_all, const float *__restrict__ pose_all, int32_t n,
                PfStatsDev *__restrict__ stats_all, int32_t normalise) {
    __shared__ double s_w[SEQ_CHUNK];
    __shared__ double s_l[SEQ_CHUNK];
    __shared__ float s_p[SEQ_CHUNK * 3];
    __shared__ double s_sum;
    const int32_t mi = blockIdx.x;
    double *w = w_all + (size_t)mi * n;
    const double *lw = logw_all + (size_t)mi * n;
    const float *pose = pose_all + (size_t)mi * n * 3;
    PfStatsDev *st = stats_all + mi;
    double weight_sum = 0.0, max_w = 0.0, max_l = -INFINITY;
    int32_t strongest = -1, nz = 0;
    for (int32_t c0 = 0; c0 < n; c0 += SEQ_CHUNK) {                    // :88-117
        const int32_t len = min(SEQ_CHUNK, n - c0);
        __syncthreads();
        for (int32_t i = threadIdx.x; i < len; i += 256) { s_w[i] = w[c0 + i]; s_l[i] = lw[c0 + i]; }
        __syncthreads();
        if (threadIdx.x == 0)
            for (int32_t i = 0; i < len; i++) {
                const double v = s_w[i];
                weight_sum += v;                                       // :100
                if (strongest < 0) { strongest = c0 + i; max_w = v; }  // :110-111
                else if (v > max_w) { strongest = c0 + i; max_w = v; } // :113-114 (strict)
                if (v == 0.0) nz++;
                if (s_l[i] > max_l) max_l = s_l[i];
            }
    }
    if (threadIdx.x == 0) s_sum = weight_sum;
    __syncthreads();
    const double sum = s_sum;
    if (normalise)
        for (int32_t i = threadIdx.x; i < n; i += 256) w[i] = w[i] / sum;          // :120-121
    __syncthreads();
    // calculateNeff's `sum` (:181-183) and getWeightedPose's four sums (:167-174), then Neff's squares (:185-187)
    double norm_sum = 0.0, xs = 0.0, ys = 0.0, ts = 0.0, sq = 0.0;
    for (int pass = 0; pass < 2; pass++)
        for (int32_t c0 = 0; c0 < n; c0 += SEQ_CHUNK) {
            const int32_t len = min(SEQ_CHUNK, n - c0);
            __syncthreads();
            for (int32_t i = threadIdx.x; i < len; i += 256) s_w[i] = w[c0 + i];
            if (pass == 0) for (int32_t i = threadIdx.x; i < 3 * len; i += 256) s_p[i] = pose[3 * (size_t)c0 + i];
            __syncthreads();
            if (threadIdx.x == 0)
                for (int32_t i = 0; i < len; i++) {
                    const double v = s_w[i];
                    if (pass == 0) {
                        xs += (double)s_p[3 * i] * v;                                  // :170
                        ys += (double)s_p[3 * i + 1] * v;                              // :171
                        ts += angle_constrain((double)s_p[3 * i + 2]) * v;             // :172
                        norm_sum += v;                                                 // :173 == :183
                    } else {
                        sq += (v / norm_sum) * (v / norm_sum);                         // :187
                    }
                }
        }
    if (threadIdx.x == 0) {
        st->weight_sum = weight_sum; st->max_w = max_w; st->max_logw = max_l; st->strongest = strongest < 0 ? 0 : strongest; st->n_zero = nz;
        st->norm_sum = norm_sum; st->sq_sum = sq; st->xs = xs; st->ys = ys; st->ts = ts;
        st->wpose[0] = (float)(xs / norm_sum); st->wpose[1] = (float)(ys / norm_sum); st->wpose[2] = (float)(ts / norm_sum);   // :176
        const int32_t sp = strongest < 0 ? 0 : strongest;
        st->spose[0] = pose[3 * (size_t)sp]; st->spose[1] = pose[3 * (size_t)sp + 1]; st->spose[2] = pose[3 * (size_t)sp + 2];
        st->n_ambiguous = 0;
    }
}

// resample() (SLAM.java:133-153) as the reference's own loop: ONE lane, c += weight in particle order.  One thread per map.
__global__ void k_resample_seq_idx(const double *__restrict__ w_all, int32_t n, const double *__restrict__ r01_maps, double r01_scalar,
                                   double fraction, int32_t *__restrict__ idx_all, PfStatsDev *__restrict__ stats_all, int32_t *__restrict__ epoch2) {
    const int32_t mi = blockIdx.x;
    if (threadIdx.x != 0) return;
    const double *w = w_all + (size_t)mi * n;
    int32_t *idx = idx_all + (size_t)mi * n;
    PfStatsDev *st = stats_all + mi;
    const bool go = fraction < 0.0 || (1.0 / st->sq_sum) < fraction * (double)n;        // GridMapApp.java:185
    st->did_resample = go ? 1 : 0;
    st->n_ambiguous = 0;
    if (epoch2 && mi == 0) { epoch2[1] = go ? 1 : 0; if (go) epoch2[0] = epoch2[0] + 1; }   // (see k_resample)
    if (!go) { for (int32_t m = 0; m < n; m++) idx[m] = m; return; }
    const double N = (double)n;
    const double r = (r01_maps ? r01_maps[mi] : r01_scalar) * 1.0 / N;                  // :136
    double c = w[0];                                                                    // :137
    int32_t i = 0;
    for (int32_t m = 1; m <= n; m++) {                                                  // :140
        const double U = r + (double)(m - 1) * 1.0 / N;                                 // :141
        while (U > c) {                                                                 // :142
            if (i >= n - 1) break;                                                      // (Java: IndexOutOfBoundsException; clamped as the oracle does)
            i++;
            c += w[i];                                                                  // :144
        }
        idx[m - 1] = i;                                                                 // :147
    }
}
// ... and the copies (:147 -> :41-43): pose, trig, weight of slot m from particle idx[m]
__global__ void __launch_bounds__(256)
k_resample_seq_gather(const int32_t *__restrict__ idx_all, const float *__restrict__ pose, const float *__restrict__ cs,
                      const double *__restrict__ w, int32_t n, float *__restrict__ pose2, float *__restrict__ cs2, double *__restrict__ w2) {
    const int32_t mi = blockIdx.y;
    const int32_t m = blockIdx.x * 256 + threadIdx.x;
    if (m >= n) return;
    const size_t o = (size_t)mi * n + m, sidx = (size_t)mi * n + idx_all[o];
    pose2[3 * o] = pose[3 * sidx]; pose2[3 * o + 1] = pose[3 * sidx + 1]; pose2[3 * o + 2] = pose[3 * sidx + 2];
    cs2[2 * o] = cs[2 * sidx]; cs2[2 * o + 1] = cs[2 * sidx + 1];
    w2[o] = w[sidx];
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline int64_t nblk_global_of(const gms_pf *pf) { return (pf->n_global + GMS_BLOCK - 1) / GMS_BLOCK; }
static inline int64_t nchunks_of(const gms_pf *pf) { return (pf->n_global + SCAN_CHUNK - 1) / SCAN_CHUNK; }
// dynamic LDS of a resampling workgroup: chunk offsets, super-chunk offsets, and (small populations) the octet boundaries
static inline size_t gms_resample_lds_bytes(int64_t nch) {
    return (size_t)(nch + 1 + (nch + 63) / 64 + 1 + (nch <= RES_SUB_MAX_CHUNKS ? nch * 8 : 0)) * sizeof(double);
}

void gms_launch_pf_init(gms_pf *pf) {
    const int64_t total = (int64_t)pf->n_maps * pf->n;
    hipLaunchKernelGGL(k_pf_init, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, pf->map->stream, pf->d_pose,
                       pf->d_cs, pf->d_w, pf->d_logw, total, 1.0 / (double)pf->n_global);
}

void gms_launch_pf_pose_trig(gms_pf *pf, const float *d_src) {
    const int64_t total = (int64_t)pf->n_maps * pf->n;
    hipLaunchKernelGGL(k_pose_trig, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, pf->map->stream, d_src, pf->d_pose,
                       pf->d_cs, total);
}

void gms_launch_pf_motion(gms_pf *pf, double d_center, double d_theta, uint64_t seed, uint64_t sequence) {
    const double d_center_sd = (0.01 + fabs(d_center) * 0.05) / 2;               // Odometry.java:63
    const double d_theta_sd = 5 * (3.141592653589793 / 180.0) + 0.1 * fabs(d_theta);   // :64
    hipLaunchKernelGGL(k_motion, dim3((pf->n + 255) / 256, pf->n_maps), dim3(256), 0, pf->map->stream, pf->d_pose, pf->d_cs,
                       pf->n, pf->offset, d_center, d_theta, d_center_sd, d_theta_sd, seed, sequence);
}

static void launch_compact(gms_pf *pf, const gms_beam *d_beams, int32_t B, int32_t beam_stride) {
    gms_map *m = pf->map;
    hipLaunchKernelGGL(k_compact_beams, dim3(pf->n_maps), dim3(64), 0, m->stream, d_beams, B, beam_stride, m->max_beams,
                       pf->d_hitbeams, pf->d_nhit);
}

// Beam segments of the scoring kernel: a function of B ONLY.  The segmentation decides how the product of the factors is
// associated (in order inside a segment, segment products in segment order), so it must not depend on how many particles a
// handle holds: a shard of a sharded filter and the stand-alone filter of the whole population have to round alike
// (tests/test_gpu_configs.py caught 8 x 8192 vs 65536 differing in the last bit when it did).
// 45 beams per segment = the 16 segments measured best at C3 (720 beams x 16 particle groups: one 16-wavefront workgroup
// per CU; 23 / 30 / 36 / 60 / 90 beams per segment: 21.6 / 24.8 / 27.7 / 25.3 / 34.5 us against 21.2) and at C5 (1080 beams:
// 332 us against 343-346 for 23 or 12).  Short scans (<= 384 beams: the reference's own 360, C2) come with small filters,
// where the launch is a handful of workgroups and the time is one workgroup's walk through its segment: 12 beams per
// segment (C2: 18.4 -> 8.4 us).  <= 128 beams per segment (the LDS beam table; also keeps a segment product >= 0.01^128, a
// normal double).
// Long scans (> 768 beams: config 5's 1080) take 90 beams per segment: half the segment products per particle to store and to
// combine (C5, 64 maps x 4096 x 1080: 45 / 68 / 90 / 120 beams per segment -> 402 / 398 / 392 / 386 us per batched step; 90 keeps
// a single-map filter of 16384 particles at 192 workgroups, 120 would leave it 144 for 256 CUs).
// Batched handles (n_maps > 1) are never sharded -- the "function of the beam count only" rule is what makes a shard round like
// the stand-alone filter -- and bring their workgroups by the thousand: their long scans take 120 beams per segment.
static int64_t score_segments(int32_t B, bool batched) {
    const int32_t seglen = B <= GMS_SCORE_SHORT_SCAN ? GMS_SCORE_SEGLEN_SHORT
                         : (B > GMS_SCORE_LONG_SCAN ? (batched ? GMS_SCORE_SEGLEN_LONG_BATCHED : GMS_SCORE_SEGLEN_LONG) : GMS_SCORE_SEGLEN);
    int64_t nseg = ((int64_t)B + seglen - 1) / seglen;
    const int64_t min_seg = ((int64_t)B + 127) / 128;
    if (nseg < min_seg) nseg = min_seg;
    if (nseg < 1) nseg = 1;
    if (nseg > GMS_SCORE_MAXSEG) nseg = GMS_SCORE_MAXSEG;         // B <= GMS_MAX_BEAMS = 128 * GMS_SCORE_MAXSEG
    return nseg;
}

// motion (may be NULL): the particles take a motion-model sample on their way into the scoring kernel (SLAM.java:90 inside the launch
// of :99): every segment's workgroup computes its particles' sample itself -- the same Philox counters, the same bits as k_motion --
// reads the old poses from d_pose and segment 0 stores the new ones into d_pose2 / d_cs2, which then change places with d_pose / d_cs
// (in place the other segments' workgroups would read poses that segment 0 had already moved).
void gms_launch_pf_score(gms_pf *pf, const gms_beam *d_beams, int32_t B, int32_t beam_stride, const float *d_pose_src,
                         const MotionModel *motion) {
    gms_map *m = pf->map;
    if (pf->reference_order) {
        // the audit path: poses first (launches of their own), then the whole scan's product in one register per particle
        if (d_pose_src) gms_launch_pf_pose_trig(pf, d_pose_src);
        if (motion) gms_launch_pf_motion(pf, motion->d_center, motion->d_theta, motion->seed, motion->sequence);
        ProfScope ps(m, GMS_K_SCORE);
        hipLaunchKernelGGL(k_score_seq, dim3((unsigned)((pf->n + 255) / 256), pf->n_maps), dim3(256), 0, m->stream, m->gd, m->d_fac, m->fac_stride,
                           d_beams, B, beam_stride, pf->d_pose, pf->d_cs, pf->n, pf->d_w, pf->d_logw);
        pf->pending_nseg = 0;
        pf->score_fresh = 1;
        return;
    }
    MotionArgs mo;
    mo.on = 0; mo.d_center = mo.d_theta = mo.d_center_sd = mo.d_theta_sd = 0.0; mo.seed = mo.sequence = 0; mo.index0 = 0;
    const int64_t nseg = score_segments(B, pf->n_maps > 1);
    // The locality order (k_order) costs a launch of its own, ~8 us for 4096 particles per map, and takes a fifth to a
    // quarter off the scoring kernel: it pays once the scoring launch is several rounds of workgroups deep (C5: 6144
    // workgroups, 328 -> 247 us), not at 256 workgroups (C3: -3 us for +5.5).  Results are the same either way.
    const int64_t wgs = nseg * (((int64_t)pf->n + 1023) / 1024) * pf->n_maps;
    // (batched handles order from one workgroup per CU on: 8 / 16 maps of config 5's shape, one GPU's share of it on an 8- / 4-GPU
    // node: k_order 9 us, k_score_c 79 -> 54 / 110 -> 78 us: round 5, tools/c5_small_batch_tune.sh)
    const bool ordered = pf->order_mode > 0 || (pf->order_mode < 0 && wgs >= (pf->n_maps > 1 ? (int64_t)m->n_cus : 2048) && pf->n <= 32768);
    if (motion && ordered) {
        // the locality order is built from the poses the particles score at: the sample gets a launch of its own in front of it
        gms_launch_pf_motion(pf, motion->d_center, motion->d_theta, motion->seed, motion->sequence);
        motion = nullptr;
        d_pose_src = nullptr;
    }
    if (ordered) {
        ProfScope po(m, GMS_K_ORDER);
        hipLaunchKernelGGL(k_order, dim3(pf->n_maps, (unsigned)((pf->n + ORD_THREADS - 1) / ORD_THREADS)), dim3(ORD_THREADS), 0,
                           m->stream, m->gd, d_pose_src, pf->d_pose, pf->d_cs, pf->n, pf->d_ord, pf->d_perm, pf->d_pose, pf->d_cs);
        d_pose_src = nullptr;                                     // stored by k_order
    }
    float *pose_dst = pf->d_pose, *cs_dst = pf->d_cs;
    if (motion) {
        mo.on = 1; mo.d_center = motion->d_center; mo.d_theta = motion->d_theta; mo.seed = motion->seed; mo.sequence = motion->sequence;
        mo.d_center_sd = (0.01 + fabs(motion->d_center) * 0.05) / 2;             // Odometry.java:63
        mo.d_theta_sd = 5 * (3.141592653589793 / 180.0) + 0.1 * fabs(motion->d_theta);   // :64
        d_pose_src = pf->d_pose; pose_dst = pf->d_pose2; cs_dst = pf->d_cs2;
    }
    ProfScope ps(m, GMS_K_SCORE);
    pf->pending_nseg = 0;
    // Lanes per workgroup: 1024 (sixteen wavefronts sharing one segment's L1 patch) while that still gives every CU a workgroup;
    // a smaller population -- one shard of eight of config 4's 65 536 particles, config 2 -- takes 512 or 256 lanes per workgroup
    // instead, so that the look-ups spread over all the CUs' address pipes rather than queueing on half or an eighth of them
    // (8192 particles x 720 beams: 128 workgroups of 1024 lanes on 256 CUs).  Which lanes share a workgroup does not enter the arithmetic.
    // (At C3 512-lane workgroups measure the same as 1024; 768: +40 %.)
    int32_t threads = pf->n >= 1024 ? 1024 : ((pf->n + 63) / 64) * 64;
    if (pf->score_threads >= 64 && pf->score_threads <= 1024) threads = (pf->score_threads / 64) * 64;
    else {
        // (batched, ordered launches want eight workgroups per CU before they stop halving: 8 maps 54 -> 44 -> 39 us at 1024 / 512 / 256
        // lanes, 16 maps 78 -> 69 -> 68; 64 maps bring 2304 workgroups of 1024 lanes and keep them)
        const int64_t want = ordered && pf->n_maps > 1 ? 8 * (int64_t)m->n_cus : (int64_t)m->n_cus;
        while (threads > 256 && nseg * (((int64_t)pf->n + threads - 1) / threads) * pf->n_maps < want) threads >>= 1;
        threads = ((threads + 63) / 64) * 64;                     // whole wavefronts (700 particles: 704 -> 352 -> 384 lanes, not 352)
    }
    const int64_t groups = ((int64_t)pf->n + threads - 1) / threads;
    // two or more workgroups per CU (see k_score_c's workgroup map; 24 576 particles, 1.5 per CU: 35.5 us without, 37.5 with; 32 768: 48.7 / 43.3)
    int32_t spread = nseg * groups >= 2 * (int64_t)m->n_cus ? 1 : 0;
    if (pf->score_spread >= 0) spread = pf->score_spread;
    if (ordered)
        hipLaunchKernelGGL(k_score_c<3>, dim3((unsigned)(nseg * groups), 1, pf->n_maps), dim3(threads), 0, m->stream, m->gd,
                           m->d_fac, m->fac_stride, d_beams, B, beam_stride, pf->d_pose, pf->d_cs, pf->n, (int32_t)nseg,
                           pf->d_part, pf->d_w, pf->d_logw, d_pose_src, pose_dst, cs_dst, pf->d_ord, pf->d_perm, mo, pf->offset, 0);
    else
        hipLaunchKernelGGL(k_score_c<1>, dim3((unsigned)(nseg * groups), 1, pf->n_maps), dim3(threads), 0, m->stream, m->gd,
                           m->d_fac, m->fac_stride, d_beams, B, beam_stride, pf->d_pose, pf->d_cs, pf->n, (int32_t)nseg,
                           pf->d_part, pf->d_w, pf->d_logw, d_pose_src, pose_dst, cs_dst, (const float4 *)nullptr,
                           (const int32_t *)nullptr, mo, pf->offset, spread);
    if (motion) { std::swap(pf->d_pose, pf->d_pose2); std::swap(pf->d_cs, pf->d_cs2); }
    pf->score_fresh = 1;
    if (nseg > 1) pf->pending_nseg = (int32_t)nseg;               // combined by the next consumer of the weights
}

// materialise weight / log-weight from the per-segment products if nobody has yet
void gms_launch_pf_combine(gms_pf *pf) {
    if (!pf->pending_nseg) return;
    gms_map *m = pf->map;
    ProfScope ps(m, GMS_K_SCORE);
    hipLaunchKernelGGL(k_score_combine, dim3((pf->n + 255) / 256, pf->n_maps), dim3(256), 0, m->stream, pf->d_part, pf->n,
                       pf->pending_nseg, pf->d_w, pf->d_logw);
    pf->pending_nseg = 0;
}

// whether the next partials / normalise pair takes the log-normalising form: the option is on and the weights are those of a scoring
// pass nobody has consumed (weights the caller set, or the copies a resampling left, are taken as they are)
bool gms_pf_lognorm_now(const gms_pf *pf) { return pf->log_norm && pf->score_fresh; }

void gms_launch_pf_partials(gms_pf *pf, double *d_partials) {
    gms_map *m = pf->map;
    ProfScope ps(m, GMS_K_REDUCE);
    const int64_t nblk = nblk_global_of(pf);
    hipLaunchKernelGGL(k_partials, dim3((unsigned)nblk, pf->n_maps), dim3(256), 0, m->stream, pf->d_w, pf->d_logw,
                       pf->d_pose, pf->n, pf->offset, nblk, d_partials,
                       pf->pending_nseg ? (const double *)pf->d_part : (const double *)nullptr, pf->pending_nseg,
                       gms_pf_lognorm_now(pf) ? pf->d_stats : (PfStatsDev *)nullptr);
    pf->pending_nseg = 0;                                             // k_partials stored the combined weights
}

void gms_launch_pf_apply_partials(gms_pf *pf, const double *d_partials, PackedParticle *d_packed_local, bool own) {
    gms_map *m = pf->map;
    ProfScope ps(m, GMS_K_REDUCE);
    const int64_t nblk = nblk_global_of(pf);
    // a stand-alone filter packs straight into its own population and gets level 0 of the scan with it
    if (own) { pf->d_global = pf->d_global_own; pf->global_raw = 0; }
    hipLaunchKernelGGL(k_normalize_pack, dim3((pf->n + 255) / 256, pf->n_maps), dim3(256), 0, m->stream, d_partials, nblk,
                       pf->d_w, pf->d_pose, pf->n, pf->offset, d_packed_local, own ? pf->d_cum : (double *)nullptr,
                       own ? pf->d_chunk_tot : (double *)nullptr, nchunks_of(pf), own ? pf->d_p2 : (double *)nullptr, pf->d_stats,
                       gms_pf_lognorm_now(pf) ? (const double *)pf->d_logw : (const double *)nullptr);
    pf->score_fresh = 0;                                              // the scoring pass has been consumed
    pf->chunks_ready = own ? 1 : 0;
    pf->neff_folded = 0;
}

void gms_launch_pf_stats_only(gms_pf *pf, const double *d_partials, PfStatsDev *d_stats_out) {
    gms_map *m = pf->map;
    ProfScope ps(m, GMS_K_REDUCE);
    hipLaunchKernelGGL(k_stats_only, dim3(pf->n_maps), dim3(256), 0, m->stream, d_partials, nblk_global_of(pf), pf->d_pose,
                       pf->n, pf->offset, d_stats_out);
}

void gms_launch_pf_pack(gms_pf *pf, PackedParticle *d_packed) {
    gms_map *m = pf->map;
    hipLaunchKernelGGL(k_pack, dim3((pf->n + 255) / 256, pf->n_maps), dim3(256), 0, m->stream, pf->d_w, pf->d_pose,
                       pf->n, d_packed);
    if (d_packed == pf->d_global) { pf->chunks_ready = 0; pf->neff_folded = 0; pf->global_raw = 0; }
}

// level 0 of the scan + {sum wn, sum wn^2} + strongest pose from d_global (paths that did not come through a
// stand-alone normalise: after the all-gather of a sharded filter, or a resample without a normalise)
void gms_launch_pf_chunk_sums(gms_pf *pf) {
    if (pf->chunks_ready) return;
    gms_map *m = pf->map;
    ProfScope ps(m, GMS_K_RESAMPLE);
    const int64_t nch = nchunks_of(pf);
    hipLaunchKernelGGL(k_chunk_sums, dim3((unsigned)nblk_global_of(pf), pf->n_maps), dim3(256), 0, m->stream, pf->d_global,
                       pf->n_global, nch, pf->d_cum, pf->d_chunk_tot, pf->d_p2, nblk_global_of(pf), pf->d_stats);
    pf->chunks_ready = 1;
}

void gms_launch_pf_fold_neff(gms_pf *pf) {
    if (pf->neff_folded) return;
    gms_map *m = pf->map;
    gms_launch_pf_chunk_sums(pf);
    hipLaunchKernelGGL(k_fold_neff, dim3(pf->n_maps), dim3(256), 0, m->stream, pf->d_p2, nblk_global_of(pf), pf->d_stats);
    pf->neff_folded = 1;
}

void gms_launch_pf_after_gather(gms_pf *pf) {
    pf->global_raw = 0;
    pf->chunks_ready = 0;
    pf->neff_folded = 0;
    gms_launch_pf_chunk_sums(pf);          // eagerly: it also publishes the strongest particle's pose
}

void gms_launch_pf_resample(gms_pf *pf, double fraction) {
    gms_map *m = pf->map;
    gms_launch_pf_chunk_sums(pf);
    ProfScope ps(m, GMS_K_RESAMPLE);
    const int64_t nch = nchunks_of(pf);
    const size_t smem = gms_resample_lds_bytes(nch);
    if (smem > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_resample), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem);
    hipLaunchKernelGGL(k_resample, dim3((pf->n + 255) / 256, pf->n_maps), dim3(256), smem, m->stream, pf->d_global,
                       pf->n_global, nch, pf->d_cum, pf->d_chunk_tot, pf->n_maps == 1 ? (const double *)nullptr : pf->d_r01_src,
                       pf->r01_scalar, fraction, pf->n, pf->offset,
                       pf->d_pose2, pf->d_cs2, pf->d_w2, pf->d_idx, pf->d_p2, nblk_global_of(pf), pf->d_stats, pf->global_raw, pf->d_epoch2);
    pf->neff_folded = 1;
}

void gms_launch_pf_refine(gms_pf *pf, const gms_beam *d_beams, int32_t B, int32_t beam_stride) {
    gms_map *m = pf->map;
    launch_compact(pf, d_beams, B, beam_stride);
    ProfScope ps(m, GMS_K_REFINE);
    const size_t smem = (size_t)B * sizeof(double2);
    if (smem > 32 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_refine), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem);
    hipLaunchKernelGGL(k_refine, dim3(pf->n, pf->n_maps), dim3(256), smem, m->stream, m->gd, m->d_fac, m->fac_stride,
                       pf->d_hitbeams, pf->d_nhit, m->max_beams, pf->d_pose, pf->d_cs, pf->n);
}

// ---- the reference-order audit path (gms_pf_set_reference_order) ----
void gms_launch_pf_normalize_seq(gms_pf *pf, PfStatsDev *d_stats_out, bool normalise) {
    gms_map *m = pf->map;
    ProfScope ps(m, GMS_K_REDUCE);
    hipLaunchKernelGGL(k_normalize_seq, dim3(pf->n_maps), dim3(256), 0, m->stream, pf->d_w, pf->d_logw, pf->d_pose, pf->n, d_stats_out, normalise ? 1 : 0);
}
void gms_launch_pf_resample_seq(gms_pf *pf, double fraction) {
    gms_map *m = pf->map;
    ProfScope ps(m, GMS_K_RESAMPLE);
    hipLaunchKernelGGL(k_resample_seq_idx, dim3(pf->n_maps), dim3(64), 0, m->stream, pf->d_w, pf->n,
                       pf->n_maps == 1 ? (const double *)nullptr : pf->d_r01_src, pf->r01_scalar, fraction, pf->d_idx, pf->d_stats, pf->d_epoch2);
    hipLaunchKernelGGL(k_resample_seq_gather, dim3((unsigned)((pf->n + 255) / 256), pf->n_maps), dim3(256), 0, m->stream, pf->d_idx, pf->d_pose,
                       pf->d_cs, pf->d_w, pf->n, pf->d_pose2, pf->d_cs2, pf->d_w2);
}
