// gms_slam_kernels.hip -- the reference's own filter shape: SLAM (J/slam/SLAM.java), N particles, each with ITS OWN GridMapData.
//
// In the reference every Particle owns a map (SLAM.java:30-47): update() rebuilds the particle's likelihood field, scores the
// scan against it and integrates the scan into the particle's map at the particle's pose (:88-107); resample() deep-copies both
// arrays of the surviving particle's map (:41-45 -> GridMap.createMapData(other), J/slam/GridMap.java:106-124).  The shared-map
// filter of gms_pf_kernels.hip is what BASELINE's configurations need (16 384 poses against one 2048^2 map); this file is what
// SLAM.update / SLAM.resample literally do, at the reference's own operating point (500 particles x 120^2 cells, SLAM.java:50,57)
// and beyond (thousands of particles x 256^2).
//
// HBM layout: logData / likelihoodData of all particles as two arrays [N][H][W] of doubles (a particle's map is GridMapData's two
// arrays as they are, row-major x + y * W), double-buffered for the resampling copy.  Poses, weights and every statistic live in a
// gms_pf of N particles (one "map" of particles): the normalisation, Neff, weighted pose and the systematic resampling's index search
// are the kernels of gms_pf_kernels.hip, unchanged.
//
//   k_slam_likelihood   computeLikelihoodMap(p.m) for every particle (SLAM.java:93): likelihood_body over [tiles][N] workgroups,
//                       likelihoodData only (16 bytes per cell: the streaming kernel of this mode)
//   k_slam_particle     one workgroup per particle: the motion-model sample (:90), probabilityOf(p.m, z, p.pose) with the product
//                       taken by ONE lane in beam order -- the reference's own association, bit for bit, underflow included (:99,
//                       GridMap.java:262-288) -- and integrateObservation(p.m, z, p.pose) (:105): producer wavefronts walk the scan's
//                       rays (RayIterator's float recurrence, ray_phase_a), every wavefront counts their cells in an LDS tile of the
//                       scan's bounding box (the sensor class from two thresholds per ray, no square root per cell), and
//                       `logData[c] += ...` (GridMap.java:223) is applied from that tile straight to the particle's rows: no count grid
//                       in memory, no global atomic, touched cells read and written once
//   k_slam_gather_maps  resample()'s deep copies: map[m] <- map[idx[m]] for both arrays, a pure HBM stream (32 bytes per cell)
#include "gms_device.h"

// a generation's arrays out of SlamBufs by SELECTION, never by a run-time index into the by-value kernel argument (that moves the
// struct -- and whatever else the compiler then keeps addressable -- into scratch memory: k_slam_gather_one ran at a quarter of its speed)
__device__ __forceinline__ double *sb_log(const SlamBufs &sb, int32_t g) { return g ? sb.log[1] : sb.log[0]; }
__device__ __forceinline__ double *sb_lik(const SlamBufs &sb, int32_t g) { return g ? sb.lik[1] : sb.lik[0]; }
__device__ __forceinline__ uint32_t *sb_code(const SlamBufs &sb, int32_t g) { return g ? sb.code[1] : sb.code[0]; }

// likelihoodData of every particle's map from its logData (mode 1 of likelihood_body: no factor table, no tile states, every tile)
template <int KH>
__global__ void __launch_bounds__(256)
k_slam_likelihood(GridDev g, SlamBufs sb, const double *__restrict__ taps_g, int32_t tiles_x, int32_t tiles_y) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int32_t cur = sb.epoch[0] & 1;               // the current generation (SlamBufs)
    const double *__restrict__ logd = sb_log(sb, cur);
    double *__restrict__ lik = sb_lik(sb, cur);
    // Eight workgroups walk a map's tiles (the launcher's usual shape): they are given ids that differ by 8, i.e. ONE XCD, so that
    // the tiles' halos are read from that XCD's L2 instead of once per XCD from memory.  Workgroups are dispatched to the XCDs round
    // robin by their linear id: of 64 consecutive ones, id & 7 picks the map of a group of eight and (id >> 3) & 7 the walker.
    uint32_t bx = blockIdx.x, by = blockIdx.y;
    if (gridDim.x == 8u) {
        const uint32_t L = blockIdx.x + 8u * blockIdx.y, grp = L >> 6;
        if (grp * 8u + 8u <= gridDim.y) { by = grp * 8u + (L & 7u); bx = (L >> 3) & 7u; }
    }
    likelihood_body<KH>(g, logd, lik, lik, 0, taps_g, nullptr, 0, tiles_x, tiles_y, bx, by, gridDim.x, smem, nullptr, nullptr, 1);
}

// ---- the class plane of a particle's map ------------------------------------------------------------------------------------
// computeLikelihoodMap reads of logData only its sign (GridMap.java:239-244), and probabilityOf reads of the field it produces only
// the cells under the scan's end points (:273-277).  So every particle keeps, beside logData, a packed plane of 2 bits per cell --
// 0 logData == 0 (or NaN), 1 logData < 0, 2 logData > 0; 1/32 of logData's bytes, all zero for a fresh map -- kept in step with
// logData by everything that writes it (the apply pass of k_slam_particle, uploads, the resampling copy), in two copies per particle:
// plane 0 as logData stands, plane 1 as it stood at the START of the last update, which is the logData the particle's likelihoodData
// is the field of (SLAM.java:93 runs before :105).  update() then needs no k_slam_likelihood launch: k_slam_particle stages plane 0 in
// LDS and blurs just the end points' neighbourhoods ("on_demand" there); likelihoodData is written, from plane 1, only when somebody asks for
// it (k_slam_likelihood_codes: a download, an upload of a field, the pose refinement).
#ifndef PS_PRECLEAR
#define PS_PRECLEAR 1                   // the first band's count tile cleared while the class plane's loads fly (0: in the band loop, as every other band's)
#endif
#define PS_CODE_MAX_KHALF 7             // on-demand evaluation: up to 15 taps (a row's window of 2-bit classes is one 64-bit read; 16 lanes per end point)
__host__ __device__ inline int64_t slam_code_words(int64_t cells) { return ((cells + 15) / 16 + 1 + 3) & ~(int64_t)3; }   // per plane: one spare word, 16-byte multiples

// likelihoodData of every particle from a plane of its class planes (mode 1 of likelihood_body, as k_slam_likelihood): plane 1 -- the
// field the last update saw, written late --, or plane 0: the field of logData as it stands, in front of the pose refinement (which
// looks up most of a field; 1/32 of k_slam_likelihood's reads)
template <int KH>
__global__ void __launch_bounds__(256)
k_slam_likelihood_codes(GridDev g, SlamBufs sb, int64_t code_words, int32_t plane, const double *__restrict__ taps_g, int32_t tiles_x, int32_t tiles_y) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int32_t cur = sb.epoch[0] & 1;
    const uint32_t *__restrict__ planes = sb_code(sb, cur) + (plane ? code_words : 0);           // that plane of every particle
    const int64_t code_stride = 2 * code_words;
    double *__restrict__ lik = sb_lik(sb, cur);
    uint32_t bx = blockIdx.x, by = blockIdx.y;
    if (gridDim.x == 8u) {
        const uint32_t L = blockIdx.x + 8u * blockIdx.y, grp = L >> 6;
        if (grp * 8u + 8u <= gridDim.y) { by = grp * 8u + (L & 7u); bx = (L >> 3) & 7u; }
    }
    likelihood_body<KH, 1, true>(g, reinterpret_cast<const double *>(planes), lik, lik, 0, taps_g, nullptr, 0, tiles_x, tiles_y, bx, by, gridDim.x, smem,
                                 nullptr, nullptr, 1, code_stride);
}

// plane 0 of `count` particles' class planes from their logData (uploads): a thread per word of 16 cells
__global__ void __launch_bounds__(256)
k_slam_codes_from_log(SlamBufs sb, int64_t cells, int32_t first, int64_t code_words, int64_t words) {
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= words) return;
    const int32_t cur = sb.epoch[0] & 1;
    const double *__restrict__ logd = sb_log(sb, cur) + (size_t)first * (size_t)cells;
    uint32_t *__restrict__ planes = sb_code(sb, cur) + (size_t)first * 2 * (size_t)code_words;
    const int64_t code_stride = 2 * code_words;
    const double *ml = logd + (size_t)blockIdx.y * (size_t)cells;
    uint32_t word = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int64_t c = 16 * w + k;
        const double v = c < cells ? ml[c] : 0.0;
        word |= (v > 0.0 ? 2u : (v < 0.0 ? 1u : 0u)) << (2 * k);
    }
    planes[(size_t)blockIdx.y * (size_t)code_stride + w] = word;
}

// GridMap.integrateObservation's per-beam locals (GridMap.java:175-188) from a transform that is already at hand (make_ray takes
// the pose and its trig)
__device__ __forceinline__ RayIn ps_make_ray(const GridDev &g, const XformDev &t, const gms_beam &m) {
    RayIn r;
    r.sx = (float)((xform_x(t, 0.0, 0.0) - g.posx) / g.res);                  // :178
    r.sy = (float)((xform_y(t, 0.0, 0.0) - g.posy) / g.res);                  // :179
    r.ex = (float)((xform_x(t, m.local_x, m.local_y) - g.posx) / g.res);      // :185
    r.ey = (float)((xform_y(t, m.local_x, m.local_y) - g.posy) / g.res);      // :186
    r.measured = (float)m.distance / g.resf;                                  // :188
    r.hit = m.hit != 0;
    return r;
}

#include <type_traits>

#ifndef PS_APPLY
#define PS_APPLY 12                     // cells per thread and pass of k_slam_particle's read-modify-write of logData
#endif
#define PS_WORDS 8                      // decision words per ray and round of k_slam_particle (256 steps of the walk)
struct PsRay {                          // what a consumer needs of a ray
    int32_t x0, y0, x_inc, y_inc, n_eff, hit;
    float sx, sy, s_free, s_prior;
};

// phase B of the ray cast (ray_phase_b) for an LDS tile of 32-bit cells n_free | n_occ << 16 -- a whole scan's visits of one cell fit
// (gms_map_create: (1 + extra) * beams < 65536) -- or, NARROW, of 16-bit cells n_free | n_occ << 8, covering [tx0, tx0 + tw) x [ty0, ty0 + th); cells outside it belong to another band
// one lane's cell of a ray: step k of the walk, from the decision word `sl` that covers it (its y steps before the word in the high half)
// step k of a ray's walk: its cell, rebuilt from the decision word that covers it, and the sensor class of that cell (what
// k_slam_particle counts and gms_slam_trace_scan lists: one function, so the trace IS the counting kernel's walk and classifier)
struct PsCell { int32_t cx, cy, cls; };
__device__ __forceinline__ PsCell ps_cell_of(const PsRay &mt, uint64_t sl, int32_t k, int32_t lane) {
    const int32_t ny = (int32_t)(((uint32_t)(sl >> 32) & ~RC_VALID) + __popc((uint32_t)sl & ((1u << (lane & 31)) - 1u)));
    const int32_t nx = k - ny;
    PsCell c;
    c.cx = mt.x0 + __mul24(mt.x_inc, nx); c.cy = mt.y0 + __mul24(mt.y_inc, ny);              // (|n| <= W + H + 1 < 2^23: gms_map_create)
    const float dX = mt.sx - ((float)c.cx + 0.5f), dY = mt.sy - ((float)c.cy + 0.5f);        // GridMap.java:215-216
    c.cls = sensor_class_sq(dX * dX + dY * dY, RayThr{mt.s_free, mt.s_prior}, mt.hit);       // :217, :223
    return c;
}
template <bool NARROW>
__device__ __forceinline__ void ps_count_cell(const GridDev &g, const PsRay &mt, uint64_t sl, int32_t k, int32_t lane, uint32_t *__restrict__ tile,
                                              int32_t tx0, int32_t ty0, int32_t tw, int32_t th) {
    const PsCell pc = ps_cell_of(mt, sl, k, lane);
    const int32_t cx = pc.cx, cy = pc.cy, cls = pc.cls;
    const uint32_t ux = (uint32_t)(cx - tx0), uy = (uint32_t)(cy - ty0);
    // inside the map (RayIterator.java:108) and of this band -- the tile is the scan's box, which lies inside the map (ray_meta clamps a
    // ray's box into it), and a walk is monotonic in x and in y: a cell inside the tile is a cell the reference's loop reaches --, and
    // not `+= logOdds(0.5)` = 0.0
    if (k < mt.n_eff && ux < (uint32_t)tw && uy < (uint32_t)th && cls != 1) {
        uint32_t cell;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(cell) : "v"(uy), "v"((uint32_t)tw), "v"(ux));      // (one instruction; the compiler spread the byte scaling over both terms)
        if (NARROW)     // 16-bit cells n_free | n_occ << 8, two to a word (k_slam_particle decides: no field of this scan can pass 255)
            __hip_atomic_fetch_add((gms_lds_u32 *)(tile) + (cell >> 1), (cls == 0 ? 1u : 0x100u) << ((cell & 1u) << 4), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
        else
            __hip_atomic_fetch_add((gms_lds_u32 *)(tile) + cell, cls == 0 ? 1u : 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
__device__ __forceinline__ void ps_wait_words(const uint64_t *__restrict__ slots, int32_t o0, int32_t o1, uint64_t &a, uint64_t &c) {
    for (;;) {                                         // wave-uniform: every lane reads the same two slots
        lds_poll_2xu64(&slots[o0], &slots[o1], a, c);
        if (((a & c) >> 63) != 0u) break;
        __builtin_amdgcn_s_sleep(2);
    }
}

// phase B of the ray cast (ray_phase_b) for an LDS tile of 32-bit cells n_free | n_occ << 16 -- a whole scan's visits of one cell fit
// (gms_map_create: (1 + extra) * beams < 65536) -- or, NARROW, of 16-bit cells n_free | n_occ << 8, covering [tx0, tx0 + tw) x [ty0, ty0 + th); cells outside it belong to another band.
// NB = 1: the 64 cells of block blk; NB = 2: blocks blk and blk + 1, two cells per lane (the second block exists: the caller checks).
template <int NB, bool NARROW>
__device__ __forceinline__ void ps_phase_b(const GridDev &g, const PsRay &mt, const uint64_t *__restrict__ slots, int32_t stride, int32_t slot,
                                           int32_t blk, int32_t lane, uint32_t *__restrict__ tile, int32_t tx0, int32_t ty0, int32_t tw,
                                           int32_t th, int32_t w_base) {
    const int32_t nwords = (mt.n_eff + 31) >> 5;
    uint64_t a[NB], c[NB];
#pragma unroll
    for (int j = 0; j < NB; j++) {
        const int32_t w0 = 2 * (blk + j) - w_base, w1 = min(2 * (blk + j) + 1, nwords - 1) - w_base;
        ps_wait_words(slots, w0 * stride + slot, w1 * stride + slot, a[j], c[j]);
    }
#pragma unroll
    for (int j = 0; j < NB; j++)
        ps_count_cell<NARROW>(g, mt, lane < 32 ? a[j] : c[j], (blk + j) * 64 + lane, lane, tile, tx0, ty0, tw, th);
}

// One workgroup per particle.  NT threads; wavefronts 0 .. NP-1 walk 64 rays each (RayIterator's float recurrence, ray_phase_a), then
// join the others in the cell work.  Dynamic LDS: factors [Bpad] f64 | thresholds [Bpad] 2 x f32 | decision slots [PS_WORDS][64 NP] u64 | ray records [64 NP] |
// count tile [tile_bytes].
// (A lane per ray running the reference's loop as it stands -- walk, distance, class, count -- was built and measured: 80 instructions
// per step on one or two wavefronts per SIMD, which issue one instruction per ~4.6 clocks: 29 us of walking for 90 rays against 17 for
// this split form, whose cell work spreads over every wavefront of the workgroup.)
template <int NT, int NP, bool NA, bool CODES>      // NA: 16-bit count cells are on offer (the map does not fit as 32-bit cells); CODES: the class planes are kept
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4)))       // (2 x 512 or 1024 lanes per CU: 128 registers)
k_slam_particle(GridDev g, const gms_beam *__restrict__ beams, int32_t B, int32_t Bpad, SlamBufs sb, int32_t field_in_memory,
                float *__restrict__ pose, float *__restrict__ cs, double *__restrict__ w,
                double *__restrict__ logw, MotionArgs mo, int32_t integrate, int32_t tile_bytes, int32_t code_words,
                const double *__restrict__ taps_g, int32_t taps_plain) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int32_t cur = sb.epoch[0] & 1;               // the current generation of the particles' maps (SlamBufs)
    double *__restrict__ log_all = sb_log(sb, cur);
    const double *__restrict__ lik_all = field_in_memory ? sb_lik(sb, cur) : nullptr;
    uint32_t *__restrict__ code_all = sb_code(sb, cur);
    constexpr int NW = NT / 64, GR = 64 * NP;
    static_assert(NW > NP, "at least one wavefront that only consumes");
    double *s_fac = reinterpret_cast<double *>(smem);                          // [Bpad]
    RayThr *s_thr = reinterpret_cast<RayThr *>(s_fac + Bpad);                  // [Bpad] every ray's squared class thresholds
    uint64_t *s_slots = reinterpret_cast<uint64_t *>(s_thr + Bpad);            // [PS_WORDS][GR]
    PsRay *s_ray = reinterpret_cast<PsRay *>(s_slots + PS_WORDS * GR);         // [GR]
    uint32_t *s_tile = reinterpret_cast<uint32_t *>(s_ray + GR);               // [tile_bytes / 4] words: 32-bit cells, or 16-bit cells two to a word
    uint32_t *s_plane = s_tile + (tile_bytes >> 2);                            // [code_words] the particle's class plane 0 (CODES)
    __shared__ float s_pose[5];                                                // x, y, theta, (float)cos, (float)sin
    __shared__ int32_t s_box[4];                                               // the scan's box x0, y0, x1, y1 (inclusive)
    __shared__ int32_t s_grp_words[NP];
    __shared__ uint16_t s_work[GR];                                            // a round's cell work: the rays that have cells in it
    __shared__ int32_t s_nwork, s_next;
    __shared__ int32_t s_nzero;                                                // rays of zero length: the only ones that visit a cell more than once
    __shared__ double s_taps[CODES ? 2 * PS_CODE_MAX_KHALF + 2 : 2];           // the blur kernel (CODES)
    __shared__ int32_t s_changed;                                              // a cell of this particle changed its class (CODES): plane 0 is written back
    const int32_t p = blockIdx.x;
    const int32_t lane = threadIdx.x & 63;
    const int32_t wave = __builtin_amdgcn_readfirstlane((int32_t)(threadIdx.x >> 6));
    GMS_STAMP(GMS_STAMP_ROW(0, blockIdx.x), 0);
    uint32_t *s_cell = s_plane + code_words;                                   // [Bpad] the end point's cell of every beam, for the on-demand field (CODES)
    uint32_t *gplane = CODES ? code_all + (size_t)p * 2 * (size_t)code_words : nullptr;

    // (the first beam of this thread: on its way while the pose is drawn)
    double tap_first = 0.0;                                                     // (the blur kernel's taps for the literal form of the on-demand field: on their way as well)
    if (CODES && threadIdx.x >= 128 && (int32_t)threadIdx.x - 128 < g.ktaps) tap_first = taps_g[threadIdx.x - 128];
    gms_beam bm_first{}, bm_thr{};
    if (B > 0) {
        bm_first = beams[min((int32_t)threadIdx.x, B - 1)];
        bm_thr = beams[min(NT - 1 - (int32_t)threadIdx.x, B - 1)];              // (the beam whose thresholds this thread finds: below)
    }
    // ---- the particle's pose: sampleMotionModel (SLAM.java:90, Odometry.java:77-96) or the pose as it stands
    if (CODES && wave != 0) {
        // the class plane as logData stands now -- the start of this update -- into LDS, by the other wavefronts while wavefront 0 draws
        // the pose; twelve loads in flight per lane (a plane is at most 24 KiB: gms_slam_create).  While the first twelve fly, the
        // count tile of the first band of rows is cleared.
        constexpr int PLB = 12;
        bool first = true;
        for (int32_t i0 = (int32_t)threadIdx.x - 64; first || i0 < code_words; i0 += PLB * (NT - 64)) {
            uint32_t plw[PLB];
#pragma unroll
            for (int u = 0; u < PLB; u++) plw[u] = gplane[min(i0 + u * (NT - 64), code_words - 1)];
            if (PS_PRECLEAR && first && integrate)
                for (int32_t i = (int32_t)threadIdx.x - 64; i < (tile_bytes >> 2); i += NT - 64) s_tile[i] = 0u;
            first = false;
#pragma unroll
            for (int u = 0; u < PLB; u++)
                if (i0 + u * (NT - 64) < code_words) s_plane[i0 + u * (NT - 64)] = plw[u];
        }
    }
    if (wave == 0) {
        float x = pose[3 * (size_t)p], y = pose[3 * (size_t)p + 1], th = pose[3 * (size_t)p + 2], c, sn;
        if (mo.on) {                                                           // (uniform)
            motion_apply(x, y, th, c, sn, (uint64_t)(mo.index0 + p), mo.d_center, mo.d_theta, mo.d_center_sd, mo.d_theta_sd, mo.seed, mo.sequence);
        } else {
            c = cs[2 * (size_t)p]; sn = cs[2 * (size_t)p + 1];                 // (cs[] always matches pose[]: k_pose_trig)
        }
        if (lane == 0) {
            s_pose[0] = x; s_pose[1] = y; s_pose[2] = th; s_pose[3] = c; s_pose[4] = sn;
            if (mo.on) {
                pose[3 * (size_t)p] = x; pose[3 * (size_t)p + 1] = y; pose[3 * (size_t)p + 2] = th;
                cs[2 * (size_t)p] = c; cs[2 * (size_t)p + 1] = sn;
            }
        }
    }
    if (threadIdx.x >= 64 && threadIdx.x < 68) s_box[threadIdx.x - 64] = threadIdx.x < 66 ? INT32_MAX : INT32_MIN;
    if (threadIdx.x == 68) s_nzero = 0;
    if (CODES && threadIdx.x >= 128 && (int32_t)threadIdx.x - 128 < g.ktaps) s_taps[threadIdx.x - 128] = tap_first;
    if (CODES && threadIdx.x == 69) s_changed = 0;
    __syncthreads();
    if (CODES)                                         // ... and into plane 1, which defines the particle's likelihoodData from here on (SLAM.java:93);
        for (int32_t i = threadIdx.x; i < code_words; i += NT) gplane[code_words + i] = s_plane[i];    // after the barrier, which would wait for the stores (it fences memory)
    GMS_STAMP(GMS_STAMP_ROW(0, blockIdx.x), 1);
    XformDev t;
    t.px = (double)s_pose[0]; t.py = (double)s_pose[1]; t.c = (double)s_pose[3]; t.s = (double)s_pose[4];      // Transform.java:13-21

    // ---- probabilityOf(p.m, z, p.pose): the factors in parallel, the product by one lane in beam order (GridMap.java:262-288);
    //      beside it every beam's ray box (GridMap.java:175-188 + ray_meta) for the count tile
    const double *lik = lik_all + (size_t)p * (size_t)g.cells;
    const bool on_demand = CODES && lik_all == nullptr;                        // no field in memory: the end points' cells are evaluated from the class plane
    int32_t bx0 = INT32_MAX, by0 = INT32_MAX, bx1 = INT32_MIN, by1 = INT32_MIN, nzero = 0;
    // the ray of beam threadIdx.x, kept: the producer lane of that beam in the first group of rays below is this very thread
    RayDev r_first;
    RayMeta mt_first;
    r_first.dx = r_first.dy = r_first.error = 0.0f; r_first.x = r_first.y = r_first.x_inc = r_first.y_inc = r_first.n = 0;
    mt_first = RayMeta{};
    for (int32_t b = (int32_t)threadIdx.x; b < B; b += NT) {
        const gms_beam bm = b == (int32_t)threadIdx.x ? bm_first : beams[b];
        double f = 1.0;                                                        // a beam that is skipped leaves the product as it is: x * 1.0 == x
        uint32_t cell = 0xffffffffu;
        if (bm.hit) {                                                          // :269
            const int32_t gx = j_cell_exact(xform_x(t, bm.local_x, bm.local_y) - g.posx, g.res);      // :273
            const int32_t gy = j_cell_exact(xform_y(t, bm.local_x, bm.local_y) - g.posy, g.res);      // :274
            if (!(gx < 0 || gy < 0 || gx >= g.W || gy >= g.H)) {                                      // :276
                if (on_demand) cell = (uint32_t)gx | ((uint32_t)gy << 16);                            // (its factor: the pass below)
                else f = lik_factor(g, lik[(size_t)gy * g.W + gx]);                                   // :277-288
            }
        }
        s_fac[b] = f;
        if (CODES) s_cell[b] = cell;
        if (integrate) {
            RayDev r;
            const RayMeta mt = ray_meta(g, ps_make_ray(g, t, bm), r);
            if (b == (int32_t)threadIdx.x) { r_first = r; mt_first = mt; }
            if (mt.n_eff > 0) {
                nzero += (mt.x_inc == 0 && mt.y_inc == 0) ? 1 : 0;
                bx0 = min(bx0, min(mt.x0, mt.hx)); bx1 = max(bx1, max(mt.x0, mt.hx));
                by0 = min(by0, min(mt.y0, mt.hy)); by1 = max(by1, max(mt.y0, mt.hy));
            }
        }
    }
    if (integrate) {
        // every ray's squared class thresholds (ray_thresholds: a few correctly rounded square roots each), beam b by thread NT - 1 - b:
        // the LAST wavefronts' work while the first ones set up the rays -- on the producers it was 2.3 us in front of the first walk
        for (int32_t b = NT - 1 - (int32_t)threadIdx.x; b < B; b += NT) {
            const gms_beam bm = b == NT - 1 - (int32_t)threadIdx.x ? bm_thr : beams[b];
            s_thr[b] = ray_thresholds((float)bm.distance / g.resf, bm.hit != 0, g.half_tol);        // GridMap.java:188; SensorModel.java:31-41
        }
#define GMS_STEP_(O) { bx0 = min(bx0, wave_xor<O>(bx0)); by0 = min(by0, wave_xor<O>(by0)); bx1 = max(bx1, wave_xor<O>(bx1)); by1 = max(by1, wave_xor<O>(by1)); nzero += wave_xor<O>(nzero); }
        GMS_BUTTERFLY(GMS_STEP_)
#undef GMS_STEP_
        if (lane == 0 && bx1 >= bx0) {
            atomicMin(&s_box[0], bx0); atomicMin(&s_box[1], by0); atomicMax(&s_box[2], bx1); atomicMax(&s_box[3], by1);
            if (nzero) atomicAdd(&s_nzero, nzero);
        }
    }
    __syncthreads();
    if (on_demand) {
        // likelihoodData[gx + gy * W] under the end points as computeLikelihoodMap would have written it (GridMap.java:239-249 ->
        // Util.doGaussianBlurdSeparable, Util.java:378-426): for every row gy + i inside the map (:418) the horizontal sum of that row at
        // column gx (`total = 0; total += kernel[j + k] * in[...]` over the columns inside the map, :391-401), then the vertical sum of
        // those (:413-422) -- the reference's operations in the reference's order, for these cells only.  Spread over the workgroup:
        // L lanes per beam (L = 8 / 16 for up to 7 / 15 taps), lane r the horizontal sum of row gy - k + r; then every lane of
        // the group adds the rows' sums in row order and the first stores the factor.
        // (One lane per beam doing all (2 k + 1)^2 taps took 7 us on three wavefronts while the others waited: profiles/r06.)
        const int32_t kh = g.khalf, ntaps = 2 * kh + 1;
        auto field_pass = [&](auto width) {            // (the group width at compile time: both loops unroll, their LDS reads and shuffles fly together)
            constexpr int L = decltype(width)::value, SH = L == 8 ? 3 : 4;
            const int32_t r = lane & (L - 1), grp = lane & ~(L - 1);
            for (int32_t base = 0; base < (B << SH); base += NT) {             // (every lane stays in the loop: the shuffles read all of a group)
                const int32_t item = base + (int32_t)threadIdx.x;
                const int32_t b = min(item >> SH, B - 1);
                const uint32_t cell = (item >> SH) < B ? s_cell[b] : 0xffffffffu;
                const int32_t gx = (int32_t)(cell & 0xffffu), gy = (int32_t)(cell >> 16);
                const int32_t y = gy - kh + r;
                double h = 0.0;
                if (cell != 0xffffffffu && r < ntaps && y >= 0 && y < g.H) {   // Util.java:418
                    const int32_t jlo = max(-kh, -gx), jhi = min(kh, g.W - 1 - gx);     // columns inside the map (:396)
                    const int32_t c0 = y * g.W + gx + jlo;
                    const uint32_t w0 = s_plane[c0 >> 4], w1 = s_plane[(c0 >> 4) + 1];  // (a spare word follows the plane)
                    const uint64_t win = (((uint64_t)w1 << 32) | w0) >> (2 * (c0 & 15));
#pragma unroll
                    for (int t = 0; t < (L < 2 * PS_CODE_MAX_KHALF + 1 ? L : 2 * PS_CODE_MAX_KHALF + 1); t++) {
                        const int32_t j = jlo + t;
                        const uint32_t e = (uint32_t)(win >> (2 * t)) & 3u;
                        const double term = s_taps[min(j + kh, ntaps - 1)] * (e == 0u ? 0.5 : (e == 2u ? 1.0 : 0.0));   // GridMap.java:239-244; Util.java:399
                        if (j <= jhi) h += term;
                    }
                }
                double total = 0.0;
#pragma unroll
                for (int rr = 0; rr < (L < 2 * PS_CODE_MAX_KHALF + 1 ? L : 2 * PS_CODE_MAX_KHALF + 1); rr++) {
                    const double hh = __shfl(h, grp + rr);
                    const int32_t yy = gy - kh + rr;
                    if (rr < ntaps && yy >= 0 && yy < g.H) total += s_taps[min(rr, ntaps - 1)] * hh;   // :418-420
                }
                if (r == 0 && cell != 0xffffffffu) s_fac[b] = lik_factor(g, total);       // GridMap.java:277-288
            }
        };
        // The same sums where every tap is +0.0 or a normal number well inside the range (gms_map::taps_plain; any Gaussian kernel):
        // a likelihood value is 0.5 c with c in {0, 1, 2}, and tap * c, their running sum and its half are exact images of the
        // reference's tap * value sums (scaling by two commutes with rounding where nothing is subnormal or overflows); a column or
        // row outside the map enters as `total + tap * 0.0`, which leaves a total that is never -0.0 as it is.  So: the row's window
        // of classes turned into the c's at once (bit-parallel), the outside columns masked to c = 0, the taps wave-uniform from the
        // scalar cache, and four instructions per tap where the literal form needs sixteen.
        auto field_pass_plain = [&](auto width) {
            constexpr int L = decltype(width)::value, SH = L == 8 ? 3 : 4, NTP = L < 2 * PS_CODE_MAX_KHALF + 1 ? L : 2 * PS_CODE_MAX_KHALF + 1;
            const int32_t r = lane & (L - 1), grp = lane & ~(L - 1);
            double tp[NTP];
#pragma unroll
            for (int t = 0; t < NTP; t++) tp[t] = t < ntaps ? taps_g[t] : 0.0;
            for (int32_t base = 0; base + wave * 64 < (B << SH); base += NT) {  // (a wavefront's lanes stay together: the shuffles read all of a group)
                const int32_t item = base + (int32_t)threadIdx.x;
                const int32_t b = min(item >> SH, B - 1);
                const uint32_t cell = (item >> SH) < B ? s_cell[b] : 0xffffffffu;
                const int32_t gx = (int32_t)(cell & 0xffffu), gy = (int32_t)(cell >> 16);
                const int32_t y = gy - kh + r;
                double h = 0.0;
                if (cell != 0xffffffffu && r < ntaps && y >= 0 && y < g.H) {   // Util.java:418
                    const int32_t c0 = y * g.W + gx - kh;                      // (>= -kh: word -1 is the count tile's last, its fields are masked)
                    const uint32_t w0 = s_plane[c0 >> 4], w1 = s_plane[(c0 >> 4) + 1];  // (a spare word follows the plane)
                    const uint32_t wv = (uint32_t)((((uint64_t)w1 << 32) | w0) >> (2 * (c0 & 15)));
                    const int32_t tlo = max(0, kh - gx), thi = min(2 * kh, g.W - 1 - gx + kh);   // columns inside the map (:396)
                    const uint32_t inside = (uint32_t)((1ull << (2 * thi + 2)) - 1ull) & ~((1u << (2 * tlo)) - 1u);
                    // class 0 (logData == 0: 0.5) -> 1, class 1 (< 0: 0.0) -> 0, class 2 (> 0: 1.0) -> 2 (GridMap.java:239-244)
                    const uint32_t cw = (((~(wv | (wv >> 1))) & 0x55555555u) | (wv & 0xaaaaaaaau)) & inside;
#pragma unroll
                    for (int t = 0; t < NTP; t++) h += tp[t] * (double)((cw >> (2 * t)) & 3u);    // Util.java:399, twice over
                    h *= 0.5;
                }
                double total = 0.0;
#pragma unroll
                for (int rr = 0; rr < NTP; rr++) total += tp[rr] * __shfl(h, grp + rr);          // :418-420 (a row outside the map: + tap * 0.0)
                if (r == 0 && cell != 0xffffffffu) s_fac[b] = lik_factor(g, total);               // GridMap.java:277-288
            }
        };
        if (taps_plain) {
            if (ntaps <= 8) field_pass_plain(std::integral_constant<int, 8>{});
            else field_pass_plain(std::integral_constant<int, 16>{});
        } else {
            if (ntaps <= 8) field_pass(std::integral_constant<int, 8>{});
            else field_pass(std::integral_constant<int, 16>{});
        }
        __syncthreads();
    }
    GMS_STAMP(GMS_STAMP_ROW(0, blockIdx.x), 2);
    if (threadIdx.x == NT - 64) {
        // the last wavefront's first lane, while the others set up the ray cast: product *= factor, beam by beam (:262, :286-288)
        double prod = 1.0;
        int32_t b = 0;
        for (; b + 8 <= B; b += 8) {
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = s_fac[b + k];
#pragma unroll
            for (int k = 0; k < 8; k++) prod *= v[k];
        }
        for (; b < B; b++) prod *= s_fac[b];
        w[p] = prod;                                                           // p.weight (SLAM.java:99)
        GMS_STAMP_T(NT - 64, GMS_STAMP_ROW(0, blockIdx.x), 3);
    }
    if (wave == NW - 2) {
        // the sum of the log factors, the product's underflow-free companion (gms_pf_set_log_normalize): one wavefront's worth of
        // double-precision logarithms, beside the ray set-up
        double ls = 0.0;
        for (int32_t b = lane; b < B; b += 64) ls += log(s_fac[b]);
        ls = wave_sum_f64(ls);
        if (lane == 0) logw[p] = ls;
        GMS_STAMP_T(NT - 128, GMS_STAMP_ROW(0, blockIdx.x), 13);
    }
    if (!integrate) return;                                                    // skipUpdate (SLAM.java:82,102)

    // ---- integrateObservation(p.m, z, p.pose) (SLAM.java:105, GridMap.java:173-228)
    const int32_t X0 = s_box[0], Y0 = s_box[1], X1 = s_box[2], Y1 = s_box[3];
    if (X1 < X0) return;                                                       // no ray touches the map
    const int32_t tw = X1 - X0 + 1, th_all = Y1 - Y0 + 1;
    // 16-bit cells -- twice the box per band, and a band is a walk of every ray that crosses it -- where the launcher allows them (the map
    // does not fit as 32-bit cells) and no count of this scan can pass 255: a ray visits a cell once, except a ray of zero length,
    // which emits its one cell 1 + extra times (RayIterator.java:75)
    const bool narrow = NA && B + g.extra * s_nzero <= 255;
    const int32_t tile_cap = narrow ? tile_bytes / 2 : tile_bytes / 4;
    const int32_t band_rows = max(1, min(th_all, tile_cap / tw));              // (the launcher guarantees a row at least)
    double *mlog = log_all + (size_t)p * (size_t)g.cells;
    for (int32_t ty0 = Y0; ty0 <= Y1; ty0 += band_rows) {
        const int32_t th = min(band_rows, Y1 - ty0 + 1);
        if (!(PS_PRECLEAR && CODES) || ty0 != Y0)
            for (int32_t i = threadIdx.x; i < (narrow ? (tw * th + 1) >> 1 : tw * th); i += NT) s_tile[i] = 0u;
        for (int32_t g0 = 0; g0 < B; g0 += GR) {
            // this group's rays: a producer lane per ray
            RayDev r;
            r.dx = r.dy = r.error = 0.0f; r.x = r.y = r.x_inc = r.y_inc = r.n = 0;
            int32_t my_nwords = 0;
            if (wave < NP) {
                const int32_t ri = g0 + wave * 64 + lane;
                PsRay pr;
                pr.n_eff = 0; pr.x0 = pr.y0 = pr.x_inc = pr.y_inc = pr.hit = 0; pr.sx = pr.sy = pr.s_free = pr.s_prior = 0.0f;
                if (ri < B) {
                    RayMeta mt;
                    if (g0 == 0) { mt = mt_first; r = r_first; }               // (ri == threadIdx.x)
                    else mt = ray_meta(g, ps_make_ray(g, t, beams[ri]), r);    // the same arithmetic as the box pass above: same values
                    // a ray that never enters this band's rows is not walked for it (its box says so)
                    const bool in_band = mt.n_eff > 0 && !(max(mt.y0, mt.hy) < ty0 || min(mt.y0, mt.hy) >= ty0 + th);
                    const RayThr thr = s_thr[ri];
                    pr.x0 = mt.x0; pr.y0 = mt.y0; pr.x_inc = mt.x_inc; pr.y_inc = mt.y_inc; pr.n_eff = in_band ? mt.n_eff : 0; pr.hit = mt.hit;
                    pr.sx = mt.sx; pr.sy = mt.sy; pr.s_free = thr.s_free; pr.s_prior = thr.s_prior;
                }
                s_ray[wave * 64 + lane] = pr;
                my_nwords = (pr.n_eff + 31) >> 5;
                int32_t nwm = my_nwords;
#define GMS_STEP_(O) nwm = max(nwm, wave_xor<O>(nwm));
                GMS_BUTTERFLY(GMS_STEP_)
#undef GMS_STEP_
                if (lane == 0) s_grp_words[wave] = nwm;
                if (g0 == 0 && ty0 == Y0) GMS_STAMP(GMS_STAMP_ROW(0, blockIdx.x), 14);
            }
            if (g0 == 0 && ty0 == Y0) GMS_STAMP_T(128, GMS_STAMP_ROW(0, blockIdx.x), 15);
            __syncthreads();
            if (g0 == 0 && ty0 == Y0) GMS_STAMP(GMS_STAMP_ROW(0, blockIdx.x), 4);
            int32_t nwords_max = 0;
#pragma unroll
            for (int k = 0; k < NP; k++) nwords_max = max(nwords_max, s_grp_words[k]);
            RayWalk wk = ray_walk_begin(r);
            for (int32_t wb = 0; wb < nwords_max; wb += PS_WORDS) {            // one round = up to 256 steps of every ray of the group
                for (int32_t i = threadIdx.x; i < PS_WORDS * GR; i += NT) s_slots[i] = 0ull;
                // The round's cell work as a list of the rays that have cells in it, built by the last wavefront while the others clear
                // the slots; a wavefront takes a ray off the list and counts its 64-cell blocks of this round one after the other --
                // the ray's record, the list entry and the ticket are fetched once per ray, not once per block.  (Walking all
                // PS_WORDS / 2 x GR (block, ray) pairs and skipping the empty ones -- rays shorter than the round, slots beyond the
                // scan's beams -- cost an LDS round trip per skipped pair.)
                const int32_t blk0 = wb >> 1, nblk = min(PS_WORDS / 2, (nwords_max - wb + 1) >> 1);
                if (wave == NW - 1) {
                    int32_t cnt = 0;
#pragma unroll
                    for (int h = 0; h < NP; h++) {
                        const int32_t slot = h * 64 + lane;
                        const bool on = blk0 * 64 < s_ray[slot].n_eff;
                        const uint64_t mask = __ballot(on);
                        if (on) s_work[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)slot;
                        cnt += __popcll(mask);
                    }
                    if (lane == 0) { s_nwork = cnt; s_next = 0; }
                }
                __syncthreads();                                               // (also: the tile is cleared, the previous round consumed)
                if (wave < NP) {
                    if (wb < my_nwords) ray_phase_a(wk, wb, min(my_nwords, wb + PS_WORDS), s_slots, GR, wave * 64 + lane);
                    if (g0 == 0 && ty0 == Y0 && wb == 0) GMS_STAMP(GMS_STAMP_ROW(0, blockIdx.x), 5);
                }
                // the cell work: every wavefront (the producers once their words are out), with the next ray's record on the way
                const int32_t nwork = s_nwork;
                auto grab = [&]() -> int32_t {
                    int32_t w = 0;
                    if (lane == 0) w = __hip_atomic_fetch_add(&s_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    return __builtin_amdgcn_readfirstlane(w);
                };
                if (g0 == 0 && ty0 == Y0 && wb == 0) { GMS_STAMP_T(NT - 64, GMS_STAMP_ROW(0, blockIdx.x), 8); GMS_STAMP_T(64 * NP, GMS_STAMP_ROW(0, blockIdx.x), 12); }
                auto consume = [&](auto narrow_c) {
                    constexpr bool NR = decltype(narrow_c)::value;
                    int32_t w = grab();
                    int32_t slot = w < nwork ? (int32_t)s_work[w] : 0;
                    PsRay ray = s_ray[slot];
                    while (w < nwork) {
                        const int32_t w2 = grab();
                        const int32_t slot2 = w2 < nwork ? (int32_t)s_work[w2] : 0;
                        const PsRay ray2 = s_ray[slot2];
                        const int32_t nb = min(nblk, ((ray.n_eff + 63) >> 6) - blk0);
                        int32_t b = 0;
                        for (; b + 2 <= nb; b += 2) ps_phase_b<2, NR>(g, ray, s_slots, GR, slot, blk0 + b, lane, s_tile, X0, ty0, tw, th, wb);
                        if (b < nb) ps_phase_b<1, NR>(g, ray, s_slots, GR, slot, blk0 + b, lane, s_tile, X0, ty0, tw, th, wb);
                        w = w2; slot = slot2; ray = ray2;
                    }
                };
                if (narrow) consume(std::true_type{}); else consume(std::false_type{});
                if (g0 == 0 && ty0 == Y0 && wb == 0) {      // the last wavefront's, the first consumer-only wavefront's and a producer's end of the cell work
                    GMS_STAMP_T(NT - 64, GMS_STAMP_ROW(0, blockIdx.x), 9); GMS_STAMP_T(64 * NP, GMS_STAMP_ROW(0, blockIdx.x), 10); GMS_STAMP_T(0, GMS_STAMP_ROW(0, blockIdx.x), 11);
                }
                __syncthreads();
            }
            if (nwords_max == 0) __syncthreads();                              // (s_ray / s_grp_words are rewritten by the next group)
        }
        // logData[c] += n_free * logOdds(P_FREE) + n_occ * logOdds(P_OCC): the expression of apply_body (GridMap.java:223).  PS_APPLY
        // cells per thread and pass (cell i = base + u NT + thread, row-major through the box), every load of a pass issued before its
        // first store, so a pass is one memory round trip; the cell's row and column advance by a recurrence (no division per cell).
        // (A read-modify-write per loop iteration was 14 round trips in a row on a 100 x 100 box.  Beyond 8 cells per pass the width
        // does not matter -- 8 / 12 / 16: 70.8 / 71.8 / 71.1 us per update at 500 x 120 x 120, 2.08 / 2.05 / 2.09 ms at 4096 x 256 x 256;
        // 24 spills: 83 us -- the pass is bound by the touched lines' read-modify-write, not by its round trips.  Loading the box BEFORE
        // the counting, to fly during it, was measured too: the registers it holds through the counting spill there, 73 -> 82 us.)
        const int32_t ncell = tw * th;
        const int32_t pq = NT / tw, pr = NT - pq * tw;                         // cell i + NT is pq rows and pr columns further on
        if (ty0 == Y0) GMS_STAMP(GMS_STAMP_ROW(0, blockIdx.x), 6);
        for (int32_t base = 0; base < ncell; base += NT * PS_APPLY) {
            const int32_t i0 = base + (int32_t)threadIdx.x;
            const int32_t y0 = i0 / tw, x0 = i0 - y0 * tw;
            uint32_t c[PS_APPLY];
            double v[PS_APPLY];
#pragma unroll
            for (int u = 0; u < PS_APPLY; u++) {
                const int32_t i = i0 + u * NT;
                if (narrow) {                                  // (uniform) n_free | n_occ << 8 -> n_free | n_occ << 16
                    const uint32_t h = i < ncell ? (s_tile[i >> 1] >> ((i & 1) << 4)) & 0xffffu : 0u;
                    c[u] = (h & 0xffu) | ((h >> 8) << 16);
                } else {
                    c[u] = i < ncell ? s_tile[i] : 0u;
                }
            }
            {
                int32_t y = y0, x = x0;
#pragma unroll
                for (int u = 0; u < PS_APPLY; u++) {
                    v[u] = c[u] ? mlog[(size_t)(ty0 + y) * g.W + X0 + x] : 0.0;
                    x += pr; y += pq;
                    if (x >= tw) { x -= tw; y++; }
                }
            }
            {
                int32_t y = y0, x = x0;
                asm volatile("" : "+v"(y), "+v"(x));           // (the addresses are formed again, not kept in 2 PS_APPLY registers)
#pragma unroll
                for (int u = 0; u < PS_APPLY; u++) {
                    if (c[u]) {
                        const double nv = v[u] + ((double)(c[u] & 0xffffu) * g.l_free + (double)(c[u] >> 16) * g.l_occ);
                        mlog[(size_t)(ty0 + y) * g.W + X0 + x] = nv;
                        if (CODES && (((v[u] > 0.0) != (nv > 0.0)) | ((v[u] < 0.0) != (nv < 0.0)))) {      // the cell's class changes: the old one is known, one exclusive-or
                            const uint32_t e0 = v[u] > 0.0 ? 2u : (v[u] < 0.0 ? 1u : 0u), e1 = nv > 0.0 ? 2u : (nv < 0.0 ? 1u : 0u);
                            {
                                s_changed = 1;
                                const uint32_t ci = (uint32_t)((ty0 + y) * g.W + X0 + x);
                                __hip_atomic_fetch_xor((gms_lds_u32 *)(s_plane) + (ci >> 4), (e0 ^ e1) << (2u * (ci & 15u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                    }
                    x += pr; y += pq;
                    if (x >= tw) { x -= tw; y++; }
                }
            }
        }
        __syncthreads();
        if (ty0 == Y0) GMS_STAMP(GMS_STAMP_ROW(0, blockIdx.x), 7);
    }
    if (CODES && s_changed)                                                    // plane 0 follows logData
        for (int32_t i = threadIdx.x; i < code_words; i += NT) gplane[i] = s_plane[i];
}

// The cell walk of k_slam_particle for ONE particle without touching its map (tests): the same ray set-up (ps_make_ray, ray_meta,
// ray_thresholds), the same recurrence (ray_phase_a) and the same per-step function (ps_cell_of) as the counting kernel, but every
// emitted step -- prior-class visits included -- is written out in walk order: for beam b, counts[b] cells (x, y) and their classes,
// [B][cap] entries (RayIterator.java:107-130: the walk stops at the first cell outside the map).  One workgroup; wavefront 0 walks 64
// rays per group, all four list their cells.
__global__ void __launch_bounds__(256)
k_slam_trace(GridDev g, const gms_beam *__restrict__ beams, int32_t B, const float *__restrict__ pose, const float *__restrict__ cs, int32_t p,
             int32_t *__restrict__ t_cells, uint8_t *__restrict__ t_cls, int32_t cap, int32_t *__restrict__ t_counts) {
    __shared__ uint64_t s_slots[PS_WORDS * 64];
    __shared__ PsRay s_ray[64];
    __shared__ int32_t s_count[64];
    __shared__ int32_t s_nwm;
    const int32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    XformDev t;
    t.px = (double)pose[3 * (size_t)p]; t.py = (double)pose[3 * (size_t)p + 1]; t.c = (double)cs[2 * (size_t)p]; t.s = (double)cs[2 * (size_t)p + 1];
    for (int32_t g0 = 0; g0 < B; g0 += 64) {
        RayDev r;
        r.dx = r.dy = r.error = 0.0f; r.x = r.y = r.x_inc = r.y_inc = r.n = 0;
        int32_t my_nwords = 0;
        if (wave == 0) {
            const int32_t ri = g0 + lane;
            PsRay pr;
            pr.n_eff = 0; pr.x0 = pr.y0 = pr.x_inc = pr.y_inc = pr.hit = 0; pr.sx = pr.sy = pr.s_free = pr.s_prior = 0.0f;
            if (ri < B) {
                const RayMeta mt = ray_meta(g, ps_make_ray(g, t, beams[ri]), r);
                const RayThr thr = ray_thresholds(mt.measured, mt.hit, g.half_tol);
                pr.x0 = mt.x0; pr.y0 = mt.y0; pr.x_inc = mt.x_inc; pr.y_inc = mt.y_inc; pr.n_eff = mt.n_eff; pr.hit = mt.hit;
                pr.sx = mt.sx; pr.sy = mt.sy; pr.s_free = thr.s_free; pr.s_prior = thr.s_prior;
            }
            s_ray[lane] = pr;
            s_count[lane] = 0;
            my_nwords = (pr.n_eff + 31) >> 5;
            int32_t nwm = my_nwords;
#define GMS_STEP_(O) nwm = max(nwm, wave_xor<O>(nwm));
            GMS_BUTTERFLY(GMS_STEP_)
#undef GMS_STEP_
            if (lane == 0) s_nwm = nwm;
        }
        __syncthreads();
        const int32_t nwords_max = s_nwm;
        RayWalk wk = ray_walk_begin(r);
        for (int32_t wb = 0; wb < nwords_max; wb += PS_WORDS) {
            for (int32_t i = threadIdx.x; i < PS_WORDS * 64; i += 256) s_slots[i] = 0ull;
            __syncthreads();
            if (wave == 0 && wb < my_nwords) ray_phase_a(wk, wb, min(my_nwords, wb + PS_WORDS), s_slots, 64, lane);
            __syncthreads();
            const int32_t blk0 = wb >> 1;
            for (int32_t pair = wave; pair < 64 * (PS_WORDS / 2); pair += 4) {
                const int32_t slot = pair & 63, blk = blk0 + (pair >> 6);
                const PsRay ray = s_ray[slot];
                if (blk * 64 >= ray.n_eff) continue;                           // (uniform per wavefront)
                const int32_t nwords = (ray.n_eff + 31) >> 5;
                uint64_t a, c;
                ps_wait_words(s_slots, (2 * blk - wb) * 64 + slot, (min(2 * blk + 1, nwords - 1) - wb) * 64 + slot, a, c);
                const int32_t k = blk * 64 + lane;
                const PsCell pc = ps_cell_of(ray, lane < 32 ? a : c, k, lane);
                const bool inside = k < ray.n_eff && (uint32_t)pc.cx < (uint32_t)g.W && (uint32_t)pc.cy < (uint32_t)g.H;      // RayIterator.java:108
                if (inside && k < cap) {
                    const size_t o = (size_t)(g0 + slot) * cap + k;
                    t_cells[2 * o] = pc.cx; t_cells[2 * o + 1] = pc.cy; t_cls[o] = (uint8_t)pc.cls;
                }
                const int32_t n = __popcll(__ballot(inside));
                if (lane == 0 && n) atomicAdd(&s_count[slot], n);
            }
            __syncthreads();
        }
        if (wave == 0 && g0 + lane < B) t_counts[g0 + lane] = s_count[lane];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// GridMap.findBestPose inside SLAM.update (SLAM.java:96 -> GridMap.java:319-346), every particle against ITS OWN field.
// One workgroup per particle.  The lattice is the reference's: float loop counters `d += step` (:328-330), 11 x 11 x 10 poses, each
// scored by probabilityOf (:261-294): 1210 x (hit beams) look-ups per particle.
//   * The particle's field is 115 KB at the reference's 120 x 120 cells: it is staged ONCE into the CU's LDS as the per-cell FACTOR of
//     probabilityOf (:285-288) with a border of the neutral factor 1.0 (a beam that ends outside the map leaves the product alone,
//     :276).  (k_refine, the shared-map form, gathers from memory and is bound by the L1 address pipe.)
//   * A look-up's cell is (gx, gy) = ((int)((x c - y s + px - posx) / res), (int)((x s + y c + py - posy) / res)): gx depends on
//     the lattice pose through (theta step, dx step) only and gy through (theta step, dy step) only.  Both are therefore computed
//     once per (theta, dx | dy, beam) -- 10 x 22 x beams coordinates instead of 1210 x 2 x beams, the reference's expressions in
//     the reference's order (Transform.java:23,28; GridMap.java:273-274) -- into LDS tables of 16-bit entries: min(gx, W) and
//     min(gy, H) x pitch, so that a look-up is  factor[tx[j] + ty[j]]  and the border does the bounds test.  A lane (= one lattice
//     pose) reads its two table rows eight beams at a time (they are contiguous in j) and multiplies its factors beam by beam in
//     beam order: the reference's plain double product (:262-288).
//   * wave task = (theta step, 64 of the (dx, dy) pairs).  argmax with the reference's rule: strict `>` against maxProb = 0 in loop
//     order, i.e. the first maximum wins (:334).
// (First built with the cell arithmetic per look-up -- 23 vector instructions each, the busiest SIMD issuing 6 tasks x 86 beams of
// them: 67 us per launch at 500 x 120 x 120 x 90, of which the LDS gather itself was 6: profiles/r06/refine_notes.md.)
// LDSF = false: the field stays in memory (maps too large for the LDS: 256 x 256 cells are 512 KB), factor formed at the look-up.
// The motion-model sample (SLAM.java:90) is drawn here when refinement is on: it precedes findBestPose (:90 -> :96).
// ---------------------------------------------------------------------------------------------
#define SR_NT 640                       // the field in memory: ten wavefronts (the lattice is 10 theta steps x 2 wavefronts of (dx, dy) pairs), three workgroups per CU
#define SR_NT_LDS 1024                  // the field in LDS, one workgroup per CU: SIXTEEN wavefronts, four on every SIMD.  Ten sat 3 + 3 + 2 + 2 and the SIMDs that
                                        // held three set every phase's pace (the twenty look-up tasks fell 6 + 6 + 4 + 4: the youngest wavefront left the look-ups
                                        // 2.9 us after the oldest); with sixteen, wavefront w takes tasks w and w + 16: 5 + 5 + 5 + 5, the field's passes are two
                                        // rounds instead of four and three, the tables two instead of three -- 25.9 -> 22.2 us per workgroup (twelve: 23.5)
#define SR_NT_OF(LDSF, KHF) ((LDSF) ? ((KHF) == 5 ? 768 : SR_NT_LDS) : SR_NT)      // (the 11-tap field passes hold 130+ registers: twelve wavefronts, 170 each)
#define SR_MAXSTEPS 16
// the reference's float loop `for (float d = -span; d < span; d += step)` (GridMap.java:328-330): the offsets and how many
__host__ __device__ inline int32_t slam_lattice_steps(float span, float step, float *out) {
    int32_t c = 0;
    for (float d = -span; d < span && c < SR_MAXSTEPS; d += step) { if (out) out[c] = d; c++; }
    return c;
}
#define SR_XSPAN 0.20f
#define SR_TRANS_STEP 0.04f
#define SR_THETA_SPAN ((float)(15 * (3.141592653589793 / 180.0)))       // GridMap.java:324
#define SR_THETA_STEP (SR_THETA_SPAN / 5)                               // :325

// KHF = 3 / 5 (with LDSF): the field is not read from memory -- no k_slam_likelihood launch writes it -- but COMPUTED in the workgroup's
// LDS from the particle's class plane (3.6 KB at 120 x 120 instead of 115 KB), with the arithmetic of likelihood_body's compile-time
// kernels (the classes' doubles {0, 1, 2} for the values {0, 0.5, 1}, the halved sum: exact for the taps gms_map::lik_kh admits) and
// no halo to recompute, the whole map being one workgroup's:
//   pass 1  a thread takes a strip of 8 (4) cells of a row: its 14 classes are one 64-bit window of the plane (read from memory: the
//           plane is L1-resident and no barrier has to wait for the pose draw), turned into doubles once and shared by the strip's
//           horizontal sums (Util.java:391-401), written where the factors will be;
//   pass 2  a thread takes a column and a band of rows and marches down it with a ring of the 2 k + 1 horizontal sums (the rows it
//           needs of the neighbouring bands are read before anybody overwrites anything), replacing them by the factor of the
//           vertical sum (:413-422, GridMap.java:285-288): lane = column, so the LDS accesses are contiguous.
template <bool LDSF, int KHF>
__global__ void __launch_bounds__(SR_NT_OF(LDSF, KHF)) __attribute__((amdgpu_waves_per_eu(LDSF ? (KHF == 5 ? 3 : 4) : 8)))      // (the field in LDS: one workgroup of sixteen wavefronts per CU; in memory: three of ten, 64 registers)
k_slam_refine(GridDev g, const gms_beam *__restrict__ beams, int32_t B, int32_t Bpad, SlamBufs sb,
              float *__restrict__ pose, float *__restrict__ cs, MotionArgs mo, int32_t fp, int32_t nt_batch, int32_t code_words,
              const double *__restrict__ taps_g) {
    static_assert(KHF == 0 || LDSF, "the field is computed into LDS");
    extern __shared__ __align__(16) unsigned char smem[];
    const double *__restrict__ lik_all = sb_lik(sb, sb.epoch[0] & 1);
    constexpr int NT = SR_NT_OF(LDSF, KHF), NW = NT / 64;
    double *s_f = reinterpret_cast<double *>(smem);                            // [H + 1][fp] factors, column W and row H neutral (LDSF)
    // (every carve offset a multiple of 16: a 16-byte LDS access off its alignment is replayed at 64 cycles -- 121 x 121 doubles are not)
    double2 *s_hb = reinterpret_cast<double2 *>(s_f + (LDSF ? (((size_t)(g.H + 1) * fp + 1) & ~(size_t)1) : 0));    // [Bpad] the hit beams' (localX, localY), in beam order
    uint16_t *s_tab = reinterpret_cast<uint16_t *>(s_hb + Bpad);               // [nt_batch][nx + ny][Bpad] cell coordinates (see above)
    __shared__ float s_dx[SR_MAXSTEPS], s_dy[SR_MAXSTEPS], s_dt[SR_MAXSTEPS], s_c[SR_MAXSTEPS], s_s[SR_MAXSTEPS];
    __shared__ int32_t s_n[3], s_nhit;
    __shared__ float s_pose[3];
    __shared__ double s_best[NW];
    __shared__ int32_t s_bestq[NW];
    const int32_t p = blockIdx.x;
    const int32_t lane = threadIdx.x & 63;
    const int32_t wave = __builtin_amdgcn_readfirstlane((int32_t)(threadIdx.x >> 6));
    GMS_STAMP(GMS_STAMP_ROW(3, blockIdx.x), 0);

    if (wave == 0) {                                   // sampleMotionModel (SLAM.java:90), as k_slam_particle draws it; the theta steps' trig
        float x = pose[3 * (size_t)p], y = pose[3 * (size_t)p + 1], th = pose[3 * (size_t)p + 2], c, sn;
        if (mo.on) {
            motion_apply(x, y, th, c, sn, (uint64_t)(mo.index0 + p), mo.d_center, mo.d_theta, mo.d_center_sd, mo.d_theta_sd, mo.seed, mo.sequence);
            if (lane == 0) {
                pose[3 * (size_t)p] = x; pose[3 * (size_t)p + 1] = y; pose[3 * (size_t)p + 2] = th;
                cs[2 * (size_t)p] = c; cs[2 * (size_t)p + 1] = sn;
            }
        }
        if (lane == 0) { s_pose[0] = x; s_pose[1] = y; s_pose[2] = th; }
        // lane l: the l-th value of the reference's float counter dTheta (:330) and Transform.fromRobotToWorld's float trig of
        // theta + dTheta (Transform.java:15-16)
        float d = -SR_THETA_SPAN;
        int32_t cnt = 0;
        bool mine = false;
        for (int32_t k = 0; k < SR_MAXSTEPS; k++) {
            if (!(d < SR_THETA_SPAN)) break;
            cnt++;
            if (k == lane) { mine = true; break; }
            d += SR_THETA_STEP;
        }
        if (mine) {
            float tc, ts;
            pose_trig(th + d, tc, ts);
            s_dt[lane] = d; s_c[lane] = tc; s_s[lane] = ts;
        }
        if (lane == SR_MAXSTEPS) s_n[2] = cnt;         // (a lane beyond the last step has counted them all)
        GMS_STAMP(GMS_STAMP_ROW(3, blockIdx.x), 1);
    } else if (wave == 1) {                            // the lattice's translations (GridMap.java:324-329)
        if (lane == 0) s_n[0] = slam_lattice_steps(SR_XSPAN, SR_TRANS_STEP, s_dx);
        if (lane == 1) s_n[1] = slam_lattice_steps(SR_XSPAN, SR_TRANS_STEP, s_dy);
    } else if (wave == 2) {                            // `if (!m.wasHit) continue` (:269): the hit beams, order kept
        int32_t base = 0;
        for (int32_t b0 = 0; b0 < B; b0 += 64) {
            const int32_t b = b0 + lane;
            const bool hit = b < B && beams[b].hit != 0;
            const unsigned long long mask = __ballot(hit);
            if (hit) s_hb[base + __popcll(mask & ((1ull << lane) - 1ull))] = make_double2(beams[b].local_x, beams[b].local_y);
            base += __popcll(mask);
        }
        if (lane == 0) s_nhit = base;
    }
    const double *lik = lik_all + (size_t)p * (size_t)g.cells;
    if (KHF > 0) {
        // ---- pass 1: horizontal sums.  16 (the power of two that holds a row's strips) threads per row, thread = strip
        constexpr int K = KHF > 0 ? KHF : 1, NTAP = 2 * K + 1, STRIP = K <= 3 ? 8 : 4, NV = STRIP + 2 * K;      // NV <= 14 classes: 28 bits
        const uint32_t *__restrict__ gpl = sb_code(sb, sb.epoch[0] & 1) + (size_t)p * 2 * (size_t)code_words;    // plane 0: logData as it stands (the start of the update)
        double tp[NTAP];
#pragma unroll
        for (int i = 0; i < NTAP; i++) tp[i] = taps_g[i];
        const int32_t nstrips = (g.W + STRIP - 1) / STRIP;
        const int32_t sh = nstrips > 1 ? 32 - __clz(nstrips - 1) : 0;          // (at most NT threads per row: the launcher checks)
        // (wavefront 0 takes none: its pose and trig end 2.6 us in, where the others' rows are nearly done -- unless a row needs more
        //  threads than nine wavefronts have)
        const int32_t skip = (NT - 64) >> sh > 0 ? 64 : 0;
        const int32_t st = ((int32_t)threadIdx.x - skip) & ((1 << sh) - 1), rows_it = (NT - skip) >> sh;
        const int32_t c0 = st * STRIP, cw0 = c0 - K;                           // the strip's first cell, its window's first column
        const int32_t tlo = max(0, -cw0), thi = min(NV - 1, g.W - 1 - cw0);    // columns inside the map (:396)
        const uint32_t inside = (uint32_t)((1ull << (2 * max(thi, 0) + 2)) - 1ull) & ~((1u << (2 * tlo)) - 1u);
        constexpr int PF = 2;                          // rows whose windows are on their way together (a window per row and round trip left the loop waiting on memory)
        for (int32_t y0 = ((int32_t)threadIdx.x - skip) >> sh; (int32_t)threadIdx.x >= skip && y0 < g.H && st < nstrips; y0 += PF * rows_it) {
            uint32_t w0[PF], w1[PF];
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const int32_t y = min(y0 + u * rows_it, g.H - 1);
                const int32_t ci = y * g.W + cw0, ci2 = max(ci, 0);            // (row 0's window starts left of the plane: shifted in below, those columns are masked)
                w0[u] = gpl[ci2 >> 4]; w1[u] = gpl[(ci2 >> 4) + 1];            // (a spare word follows the plane)
            }
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const int32_t y = y0 + u * rows_it;
                if (y < g.H) {
                    const int32_t ci = y * g.W + cw0, d = max(0, -ci), ci2 = ci + d;
                    const uint32_t wv = (uint32_t)((((uint64_t)w1[u] << 32) | w0[u]) >> (2 * (ci2 & 15))) << (2 * d);
                    // class 0 (logData == 0: 0.5) -> 1, class 1 (< 0: 0.0) -> 0, class 2 (> 0: 1.0) -> 2 (GridMap.java:239-244); outside: 0
                    const uint32_t cw = (((~(wv | (wv >> 1))) & 0x55555555u) | (wv & 0xaaaaaaaau)) & inside;
                    double v[NV];
#pragma unroll
                    for (int t = 0; t < NV; t++) v[t] = (double)((cw >> (2 * t)) & 3u);
                    double *row = s_f + (size_t)y * fp + c0;
                    double tot[STRIP];                 // (all sums first, in one block: the strip's chains of dependent additions interleave)
#pragma unroll
                    for (int o = 0; o < STRIP; o++) {
                        double total = tp[0] * v[o];                           // (== 0.0 + the product: no tap is negative)
#pragma unroll
                        for (int i = 1; i < NTAP; i++) total += tp[i] * v[o + i];      // Util.java:399, twice over: a column outside adds tap * 0.0
                        tot[o] = 0.5 * total;
                    }
#pragma unroll
                    for (int o = 0; o < STRIP; o++)
                        if (c0 + o < g.W) row[o] = tot[o];
                }
            }
        }
    } else if (LDSF) {
        // the factor of every cell (:285-288): wavefront = rows, lane = a pair of columns, a row's loads issued together; then the border
        // (wavefront 0 joins when its pose and trig are done -- 2.6 us, which its share of the rows would otherwise follow: it takes none)
        if ((g.W & 1) == 0) {
            constexpr int NS = NW - 1;
            const int32_t w2 = g.W >> 1;
            for (int32_t y0r = wave - 1; wave > 0 && y0r < g.H; y0r += 4 * NS) {
                double2 v[4];
                for (int32_t x2 = lane; x2 < w2; x2 += 64) {
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] = reinterpret_cast<const double2 *>(lik + (size_t)min(y0r + u * NS, g.H - 1) * g.W)[x2];
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (y0r + u * NS < g.H) {
                            double *row = s_f + (size_t)(y0r + u * NS) * fp + 2 * x2;
                            row[0] = lik_factor(g, v[u].x); row[1] = lik_factor(g, v[u].y);
                        }
                }
            }
        } else {
            for (int32_t i = threadIdx.x; i < g.W * g.H; i += NT) {
                const int32_t y = i / g.W, x = i - y * g.W;
                s_f[y * fp + x] = lik_factor(g, lik[i]);
            }
        }
        for (int32_t i = threadIdx.x; i < g.H; i += NT) s_f[i * fp + g.W] = 1.0;
        for (int32_t i = threadIdx.x; i <= g.W; i += NT) s_f[g.H * fp + i] = 1.0;
    }
    GMS_STAMP_T(NT - 64, GMS_STAMP_ROW(3, blockIdx.x), 2);
    __syncthreads();
    GMS_STAMP(GMS_STAMP_ROW(3, blockIdx.x), 10);
    if (KHF > 0) {
        // ---- pass 2: vertical sums and factors in place
        constexpr int K = KHF > 0 ? KHF : 1, NTAP = 2 * K + 1;
        double tp[NTAP];
#pragma unroll
        for (int i = 0; i < NTAP; i++) tp[i] = taps_g[i];
        const int32_t nb = max(1, min(NT / g.W, g.H / K));                  // (a band has at least K rows: its ring is primed from its own rows)
        const int32_t rows_per = (g.H + nb - 1) / nb;
        const int32_t band = (int32_t)threadIdx.x / g.W, x = (int32_t)threadIdx.x - band * g.W;
        const int32_t r0 = band * rows_per, r1 = min(g.H, r0 + rows_per);
        const bool work = band < nb && r0 < g.H;
        double top[K], bot[K];
#pragma unroll
        for (int i = 0; i < K; i++) {
            const int32_t yt = r0 - K + i, yb = r1 + i;
            top[i] = work && yt >= 0 ? s_f[(size_t)yt * fp + x] : 0.0;         // (a row outside the map: + tap * 0.0 below, Util.java:418)
            bot[i] = work && yb < g.H ? s_f[(size_t)yb * fp + x] : 0.0;
        }
        __syncthreads();                               // every band's neighbours are in registers: the rows may be overwritten
        GMS_STAMP(GMS_STAMP_ROW(3, blockIdx.x), 11);
        if (work) {
            // the horizontal sums of rows y - K .. y + K in a ring of RING = NTAP + 1 registers: logical entry i of step o is P[(o + i) % RING];
            // a chunk of RING rows per pass of the loop makes those indices compile-time (24 rows per band at 120 x 120: three chunks of 8),
            // and the NEXT chunk's incoming rows are read before this chunk's factors are stored: none of them is a row this chunk writes
            constexpr int RING = NTAP + 1;
            double P[RING];
#pragma unroll
            for (int i = 0; i < K; i++) P[i] = top[i];
#pragma unroll
            for (int i = 0; i < K; i++) P[K + i] = r0 + i < r1 ? s_f[(size_t)(r0 + i) * fp + x] : 0.0;    // (short of K rows only at the map's last rows: nothing below)
            auto incoming = [&](int32_t y, double (&hin)[RING]) {
#pragma unroll
                for (int o = 0; o < RING; o++) {
                    const int32_t yin = y + o + K;
                    double v = 0.0;
                    if (yin < r1) v = s_f[(size_t)yin * fp + x];
                    else {
#pragma unroll
                        for (int i = 0; i < K; i++) if (yin - r1 == i) v = bot[i];
                    }
                    hin[o] = v;
                }
            };
            double hnext[RING];
            incoming(r0, hnext);
            for (int32_t y = r0; y < r1; y += RING) {
                double hcur[RING];
#pragma unroll
                for (int o = 0; o < RING; o++) hcur[o] = hnext[o];
                if (y + RING < r1) incoming(y + RING, hnext);
                double fac[RING];                      // (all sums first: the chunk's chains of dependent additions interleave)
#pragma unroll
                for (int o = 0; o < RING; o++) {
                    P[(o + 2 * K) % RING] = hcur[o];
                    double total = tp[0] * P[o % RING];
#pragma unroll
                    for (int i = 1; i < NTAP; i++) total += tp[i] * P[(o + i) % RING];    // Util.java:413-422
                    fac[o] = lik_factor(g, total);                                       // GridMap.java:285-288
                }
#pragma unroll
                for (int o = 0; o < RING; o++)
                    if (y + o < r1) s_f[(size_t)(y + o) * fp + x] = fac[o];
            }
        }
        GMS_STAMP(GMS_STAMP_ROW(3, blockIdx.x), 12);
        for (int32_t i = threadIdx.x; i < g.H; i += NT) s_f[i * fp + g.W] = 1.0;
        for (int32_t i = threadIdx.x; i <= g.W; i += NT) s_f[g.H * fp + i] = 1.0;
    }
    GMS_STAMP(GMS_STAMP_ROW(3, blockIdx.x), 3);
#if defined(SR_EXP) && SR_EXP == 1            // experiment: staging only
    if (s_nhit >= 0) return;
#endif
    const int32_t nx = s_n[0], ny = s_n[1], nt = s_n[2], nhit = s_nhit;
    const uint32_t base8 = (uint32_t)(uintptr_t)(const gms_lds_f64 *)s_f >> 3;                        // the field's LDS offset in doubles (16-byte aligned)
    if (LDSF && base8 > 512u) __builtin_trap();        // (the launcher's 16-bit bound on the table sums allows for 4 KB of static LDS in front of the field: this kernel has 0.7)
    const float x0 = s_pose[0], y0 = s_pose[1], t0 = s_pose[2];
    const int32_t nc = nx + ny;
    const int32_t nxy = nx * ny, nhalf = (nxy + 63) >> 6;
    double best = 0.0;                                 // maxProb = 0 (:321)
    int32_t bestq = INT32_MAX;                         // none yet: the start pose stays (:320)
    for (int32_t it0 = 0; it0 < nt; it0 += nt_batch) {         // (all theta steps at once where the tables fit beside the field)
        const int32_t ntb = min(nt_batch, nt - it0);
        if (it0 > 0) __syncthreads();                  // the previous batch's tables have been read
        // ---- the tables: entry (theta step, coordinate c, beam j), j fastest.  A work item is one (theta step, x | y, beam): the
        //      rotated beam once, then the nx (ny) lattice offsets along that axis.  One expression for both coordinates:
        //      x s + y c == x s - y (-c) bit for bit (negation is exact), so gy is gx's arithmetic with (s, -c) for (c, s).
        for (int32_t i = threadIdx.x; i < ntb * 2 * nhit; i += NT) {
            const int32_t j = i % nhit, r2 = i / nhit, itl = r2 >> 1;
            const bool isx = (r2 & 1) == 0;
            const float cf = s_c[it0 + itl], sf = s_s[it0 + itl];
            const double A = (double)(isx ? cf : sf), Bn = (double)(isx ? sf : -cf);
            const double2 bm = s_hb[j];
            const double rot = bm.x * A - bm.y * Bn;                                       // Transform.java:23,28 before `+ px`
            const double pos = isx ? g.posx : g.posy;
            const uint32_t lim = (uint32_t)(isx ? g.W : g.H), mul = LDSF && !isx ? (uint32_t)fp : 1u;
            const uint32_t off = LDSF && !isx ? base8 : 0u;                                // (the y entries carry the field's LDS offset / 8: the look-ups)
            const float base = isx ? x0 : y0;
            const float *dd = isx ? s_dx : s_dy;
            const int32_t nk = isx ? nx : ny;
            uint16_t *out = s_tab + ((size_t)(itl * nc + (isx ? 0 : nx)) * Bpad + j);
            bool guard = false;                            // (one for the work item)
            constexpr int NKC = 11;                        // the translation offsets of the reference's lattice (slam_lattice_steps(0.20f, 0.04f): the float counter stops after eleven)
            if (LDSF && nxy == NKC * NKC) {                // (uniform) the usual count at compile time: no loop, the offsets' LDS reads issued together
#pragma unroll
                for (int k = 0; k < NKC; k++) {
                    const double w = rot + (double)(base + dd[k]) - pos;                   // :332 (a float sum) ... - position (:273-274)
                    out[(size_t)k * Bpad] = (uint16_t)(__umul24(min((uint32_t)j_cell_fast(w, g.rinv, guard), lim), mul) + off);
                }
            } else
            for (int32_t k = 0; k < nk; k++) {
                const double w = rot + (double)(base + dd[k]) - pos;                       // :332 (a float sum) ... - position (:273-274)
                out[(size_t)k * Bpad] = (uint16_t)(__umul24(min((uint32_t)j_cell_fast(w, g.rinv, guard), lim), mul) + off);      // (both below 2^24: one v_mad_u32_u24)
            }
            if (__builtin_expect(guard, 0))                // a quotient within 2^-19 of an integer: the reference's division decides
                for (int32_t k = 0; k < nk; k++) {
                    const double w = rot + (double)(base + dd[k]) - pos;
                    out[(size_t)k * Bpad] = (uint16_t)(min((uint32_t)j_cell_exact(w, g.res), lim) * mul + off);
                }
        }
        GMS_STAMP_T(NT - 64, GMS_STAMP_ROW(3, blockIdx.x), 4);
        __syncthreads();
        GMS_STAMP(GMS_STAMP_ROW(3, blockIdx.x), 5);
#if defined(SR_EXP) && SR_EXP == 2            // experiment: staging and tables
        continue;
#endif
        // ---- the look-ups
        for (int32_t task = wave; task < ntb * nhalf; task += NW) {
            const int32_t itl = task / nhalf, e = (task - itl * nhalf) * 64 + lane;
            const bool live = e < nxy;
            const int32_t ec = live ? e : 0;
            const int32_t ix = ec / ny, iy = ec - ix * ny;
            const uint16_t *tx = s_tab + (size_t)(itl * nc + ix) * Bpad, *ty = s_tab + (size_t)(itl * nc + nx + iy) * Bpad;
            auto factor_at = [&](uint32_t ex, uint32_t ey) -> double {
                if (LDSF) {
                    return *reinterpret_cast<const gms_lds_f64 *>((uintptr_t)((ex + ey) << 3));        // (ey carries the field's LDS offset: the tables)
                } else {
                    const bool in = ex < (uint32_t)g.W && ey < (uint32_t)g.H;                      // :276
                    const double v = lik[in ? (size_t)ey * g.W + ex : 0];
                    return in ? lik_factor(g, v) : 1.0;
                }
            };
            double prod = 1.0;                                                             // :262
            // Eight beams per pass, software-pipelined: with two or three wavefronts per SIMD a pass that read its table rows,
            // waited, read its factors, waited, and multiplied took 470 clocks per wavefront (three LDS latencies in a row).  Now
            // pass i multiplies while pass i + 1's factors and pass i + 2's table rows are in flight (LDS returns in order).
            const int32_t npass = nhit >> 3;               // (Bpad is a multiple of 8: the rows are 16-byte aligned)
            uint4 ta = make_uint4(0, 0, 0, 0), tb = ta;
            double fc[8], fn[8];
            auto issue = [&](double (&f)[8]) {
                const uint32_t ax[4] = {ta.x, ta.y, ta.z, ta.w}, bx[4] = {tb.x, tb.y, tb.z, tb.w};
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (LDSF) {
                        // tx + ty of two beams in ONE addition (the sums stay below 2^16: the launcher checks, no carry crosses), the y entries
                        // carry the field's LDS offset / 8: a look-up is a word extracted and shifted, and the read
                        const uint32_t sm = ax[u] + bx[u];
                        f[2 * u] = *reinterpret_cast<const gms_lds_f64 *>((uintptr_t)((sm & 0xffffu) << 3));
                        f[2 * u + 1] = *reinterpret_cast<const gms_lds_f64 *>((uintptr_t)((sm >> 16) << 3));
                    } else {
                        f[2 * u] = factor_at(ax[u] & 0xffffu, bx[u] & 0xffffu);
                        f[2 * u + 1] = factor_at(ax[u] >> 16, bx[u] >> 16);
                    }
                }
            };
            if (npass > 0) {
                ta = *reinterpret_cast<const uint4 *>(tx); tb = *reinterpret_cast<const uint4 *>(ty);
                issue(fc);
                if (npass > 1) { ta = *reinterpret_cast<const uint4 *>(tx + 8); tb = *reinterpret_cast<const uint4 *>(ty + 8); }
            }
            for (int32_t ps = 0; ps < npass; ps += 2) {                                   // (two passes per turn: the two sets of factors swap roles, no copies)
                if (ps + 1 < npass) {
                    issue(fn);
                    if (ps + 2 < npass) { ta = *reinterpret_cast<const uint4 *>(tx + 8 * (ps + 2)); tb = *reinterpret_cast<const uint4 *>(ty + 8 * (ps + 2)); }
                }
#pragma unroll
                for (int u = 0; u < 8; u++) prod *= fc[u];                                 // beams in order (:267-288)
                if (ps + 1 < npass) {
                    if (ps + 2 < npass) {
                        issue(fc);
                        if (ps + 3 < npass) { ta = *reinterpret_cast<const uint4 *>(tx + 8 * (ps + 3)); tb = *reinterpret_cast<const uint4 *>(ty + 8 * (ps + 3)); }
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) prod *= fn[u];
                }
            }
            int32_t j = npass << 3;
            for (; j < nhit; j++) prod *= factor_at(tx[j], ty[j]);
            const int32_t q = (ix * ny + iy) * nt + it0 + itl;                             // the reference's loop order: theta fastest
            if (live && (prod > best || (prod == best && prod > 0.0 && q < bestq))) { best = prod; bestq = q; }   // :334
        }
    }
    GMS_STAMP(GMS_STAMP_ROW(3, blockIdx.x), 6); GMS_STAMP_T(NT - 64, GMS_STAMP_ROW(3, blockIdx.x), 7); GMS_STAMP_T(128, GMS_STAMP_ROW(3, blockIdx.x), 8);
    // the first maximum over the lattice: the larger probability wins, of equal ones the earlier pose
#define GMS_STEP_(O) { const double v2 = wave_xor<O>(best); const int32_t q2 = wave_xor<O>(bestq); \
                       if (v2 > best || (v2 == best && q2 < bestq)) { best = v2; bestq = q2; } }
    GMS_BUTTERFLY(GMS_STEP_)
#undef GMS_STEP_
    if (lane == 0) { s_best[wave] = best; s_bestq[wave] = bestq; }
    __syncthreads();
    if (wave == 0) {
        // the wavefronts' maxima by one more butterfly (NW <= 16 of them, a lane each: a loop over them by one thread was sixteen
        // dependent LDS round trips in front of the kernel's end)
        double b = lane < NW ? s_best[lane] : 0.0;
        int32_t bq = lane < NW ? s_bestq[lane] : INT32_MAX;
#define GMS_STEP16_(O) { const double v2 = wave_xor<O>(b); const int32_t q2 = wave_xor<O>(bq); \
                         if (v2 > b || (v2 == b && q2 < bq)) { b = v2; bq = q2; } }
        GMS_STEP16_(1) GMS_STEP16_(2) GMS_STEP16_(4) GMS_STEP16_(8)
#undef GMS_STEP16_
        static_assert(NW <= 16, "one lane per wavefront, four butterfly steps");
        if (lane == 0 && !(b > 0.0)) bq = INT32_MAX;   // maxProb = 0 is never exceeded (:321, :334): the start pose stays
        if (lane == 0 && bq != INT32_MAX) {
            const int32_t it = bq % nt, iy = (bq / nt) % ny, ix = bq / (nt * ny);
            pose[3 * (size_t)p] = x0 + s_dx[ix]; pose[3 * (size_t)p + 1] = y0 + s_dy[iy]; pose[3 * (size_t)p + 2] = t0 + s_dt[it];
            cs[2 * (size_t)p] = s_c[it]; cs[2 * (size_t)p + 1] = s_s[it];
        }
    }
    GMS_STAMP(GMS_STAMP_ROW(3, blockIdx.x), 9);
}

// resample()'s deep copies (SLAM.java:147 -> :41-45 -> GridMap.java:106-124): slot m of the new generation receives both arrays of
// particle idx[m]'s map.  grid = (chunks, N); a workgroup streams its chunk of both arrays, 16 bytes per lane, four loads in flight.
__global__ void __launch_bounds__(256)
k_slam_gather_maps(SlamBufs sb, const int32_t *__restrict__ idx, int64_t cells) {
    const int32_t m = blockIdx.y;
    const int32_t did = sb.epoch[1], epoch = sb.epoch[0], i = idx[m];
    if (!did) return;                                  // the rule said no (GridMapApp.java:185): nothing was drawn, nothing moves
    const int32_t cur = epoch & 1;                     // the generation the draw has just made current receives the copies
    const double *__restrict__ src_log = sb_log(sb, cur ^ 1), *__restrict__ src_lik = sb_lik(sb, cur ^ 1);
    double *__restrict__ dst_log = sb_log(sb, cur), *__restrict__ dst_lik = sb_lik(sb, cur);
    const size_t so = (size_t)i * (size_t)cells, dof = (size_t)m * (size_t)cells;
    if ((cells & 1) == 0) {                            // every map starts on a 16-byte boundary
        const int64_t n2 = cells >> 1;
        const double2 *sl = reinterpret_cast<const double2 *>(src_log + so), *sk = reinterpret_cast<const double2 *>(src_lik + so);
        double2 *dl = reinterpret_cast<double2 *>(dst_log + dof), *dk = reinterpret_cast<double2 *>(dst_lik + dof);
        const int64_t stride = (int64_t)gridDim.x * 256;
        for (int64_t e0 = (int64_t)blockIdx.x * 256 + threadIdx.x; e0 < n2; e0 += 2 * stride) {
            const int64_t e1 = e0 + stride;
            const bool two = e1 < n2;
            const double2 a0 = sl[e0], b0 = sk[e0];
            double2 a1 = a0, b1 = b0;
            if (two) { a1 = sl[e1]; b1 = sk[e1]; }
            dl[e0] = a0; dk[e0] = b0;
            if (two) { dl[e1] = a1; dk[e1] = b1; }
        }
    } else {
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < cells; e += (int64_t)gridDim.x * 256) {
            dst_log[dof + e] = src_log[so + e];
            dst_lik[dof + e] = src_lik[so + e];
        }
    }
}

// ... and one array only: resample() with likelihoodData's copies deferred (gms_slam::lik_behind) moves logData at once and
// likelihoodData if and when somebody reads it before the next update's computeLikelihoodMap has overwritten it
#ifndef GATHER_U
#define GATHER_U 8
#endif
// one array of one map: dst[dof ..) <- src[so ..), GATHER_U 16-byte loads in flight per lane.  (A function of its own with restrict
// PARAMETERS: with the pointers picked inside the kernel the compiler kept the loads' registers addressable and parked them in
// scratch memory between the loads and the stores -- the copy ran at a quarter of its speed.)
__device__ __forceinline__ void gather_copy_array(const double *__restrict__ src, double *__restrict__ dst, size_t so, size_t dof, int64_t cells) {
    if ((cells & 1) == 0) {
        const int64_t n2 = cells >> 1;
        const double2 *sp = reinterpret_cast<const double2 *>(src + so);
        double2 *dp = reinterpret_cast<double2 *>(dst + dof);
        const int64_t stride = (int64_t)gridDim.x * 256;
        // GATHER_U 16-byte loads in flight per lane: at 500 x 120 x 120 the copy is bound by its workgroups' round trips (the source
        // index, then the data), not by bytes -- the maps sit in the Infinity Cache --, so a workgroup moves 32 KiB and the whole
        // copy is one residency of workgroups
        for (int64_t e0 = (int64_t)blockIdx.x * 256 + threadIdx.x; e0 < n2; e0 += GATHER_U * stride) {
            double2 v[GATHER_U];
#pragma unroll
            for (int u = 0; u < GATHER_U; u++) v[u] = sp[min(e0 + u * stride, n2 - 1)];              // (clamped: no load behind a branch)
#pragma unroll
            for (int u = 0; u < GATHER_U; u++) if (e0 + u * stride < n2) dp[e0 + u * stride] = v[u];
        }
    } else {
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < cells; e += (int64_t)gridDim.x * 256) dst[dof + e] = src[so + e];
    }
}
__global__ void __launch_bounds__(256)
k_slam_gather_one(SlamBufs sb, int32_t lik_array, const int32_t *__restrict__ idx, int64_t cells, int32_t *__restrict__ idx_keep, int64_t code_words2) {
    const int32_t m = blockIdx.y;
    const int32_t did = sb.epoch[1], epoch = sb.epoch[0], i = idx[m];          // (three independent scalar loads: one round trip in front of the data's)
    if (!did) return;                                  // the rule said no (GridMapApp.java:185): nothing was drawn, nothing moves
    const int32_t cur = epoch & 1;                     // the generation the draw has just made current receives the copies
    if (idx_keep && blockIdx.x == 0 && threadIdx.x == 0) idx_keep[m] = i;      // for the copy that is still owed (gms_slam::d_idx_lik)
    if (!lik_array && code_words2 && blockIdx.x == gridDim.x - 1) {            // the particle's two class planes travel with its logData (16-byte multiples)
        const uint4 *sc = reinterpret_cast<const uint4 *>(sb_code(sb, cur ^ 1) + (size_t)i * (size_t)code_words2);
        uint4 *dc = reinterpret_cast<uint4 *>(sb_code(sb, cur) + (size_t)m * (size_t)code_words2);
        for (int64_t e = threadIdx.x; e < code_words2 / 4; e += 256) dc[e] = sc[e];
    }
    gather_copy_array(lik_array ? sb_lik(sb, cur ^ 1) : sb_log(sb, cur ^ 1), lik_array ? sb_lik(sb, cur) : sb_log(sb, cur), (size_t)i * (size_t)cells,
                      (size_t)m * (size_t)cells, cells);
}

// the class planes alone (the resampling copy that moves both arrays at once, k_slam_gather_maps, does not carry them)
__global__ void __launch_bounds__(256)
k_slam_gather_codes(SlamBufs sb, const int32_t *__restrict__ idx, int64_t code_words2) {
    if (!sb.epoch[1]) return;
    const int32_t cur = sb.epoch[0] & 1;
    const uint32_t *__restrict__ src_code = sb_code(sb, cur ^ 1);
    uint32_t *__restrict__ dst_code = sb_code(sb, cur);
    const int32_t m = blockIdx.x;
    const uint4 *sc = reinterpret_cast<const uint4 *>(src_code + (size_t)idx[m] * (size_t)code_words2);
    uint4 *dc = reinterpret_cast<uint4 *>(dst_code + (size_t)m * (size_t)code_words2);
    for (int64_t e = threadIdx.x; e < code_words2 / 4; e += 256) dc[e] = sc[e];
}

// ---- a sharded filter's resample(): the maps whose source particle lives on another rank travel as RECORDS -------------------
// record of a particle = logData [cells] doubles | class planes [2][code_words] words (likelihoodData is the field of plane 1: it
// needs no bytes of its own).  rec_doubles = cells + code_words.
// export: records of the listed local particles, read from the PREVIOUS generation (the one the last draw left behind)
__global__ void __launch_bounds__(256)
k_slam_export_records(SlamBufs sb, const int32_t *__restrict__ list, int64_t cells, int64_t code_words, double *__restrict__ dst) {
    const int32_t prev = (sb.epoch[0] & 1) ^ 1;
    const int32_t i = list[blockIdx.y];
    const int64_t rec = cells + code_words;
    const double *log = sb_log(sb, prev) + (size_t)i * (size_t)cells;
    const double *code = reinterpret_cast<const double *>(sb_code(sb, prev) + (size_t)i * 2 * (size_t)code_words);      // (2 code_words words = code_words doubles; 16-byte multiples)
    double *out = dst + (size_t)blockIdx.y * (size_t)rec;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < rec; e += (int64_t)gridDim.x * 256) out[e] = e < cells ? log[e] : code[e - cells];
}
// the copies of a sharded resample(): slot m of the CURRENT generation <- local particle src_local[m] of the previous one, or (src_local[m]
// < 0) record recv_pos[m] of the buffer received from the other ranks
__global__ void __launch_bounds__(256)
k_slam_shard_gather(SlamBufs sb, const int32_t *__restrict__ src_local, const int32_t *__restrict__ recv_pos, const double *__restrict__ recv,
                    int64_t cells, int64_t code_words) {
    const int32_t cur = sb.epoch[0] & 1, prev = cur ^ 1;
    const int32_t m = blockIdx.y;
    const int32_t i = src_local[m];
    const int64_t rec = cells + code_words;
    const double *log, *code;
    if (i >= 0) {
        log = sb_log(sb, prev) + (size_t)i * (size_t)cells;
        code = reinterpret_cast<const double *>(sb_code(sb, prev) + (size_t)i * 2 * (size_t)code_words);
    } else {
        log = recv + (size_t)recv_pos[m] * (size_t)rec;
        code = log + cells;
    }
    double *dlog = sb_log(sb, cur) + (size_t)m * (size_t)cells;
    double *dcode = reinterpret_cast<double *>(sb_code(sb, cur) + (size_t)m * 2 * (size_t)code_words);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < rec; e += (int64_t)gridDim.x * 256) {
        if (e < cells) dlog[e] = log[e];
        else dcode[e - cells] = code[e - cells];
    }
}

// createMapData(null) for every particle (SLAM.reset, SLAM.java:65-77): logData = logOdds(0.5) = 0.0, likelihoodData = 0.0
// (hipMemsetAsync does it: both are all-zero bit patterns)

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
void gms_launch_slam_likelihood(gms_map *m, const SlamBufs &sb, int32_t n) {
    ProfScope ps(m, GMS_K_LIKELIHOOD);
    const int32_t k = m->lik_kh;
    const int32_t tiles_x = (m->gd.W + LK_TW - 1) / LK_TW, tiles_y = (m->gd.H + LK_TH - 1) / LK_TH;
    const size_t smem = gms_likelihood_lds_bytes(m->gd.khalf, k != 0);
    // a workgroup per tile while that stays a few rounds of the chip; beyond it persistent workgroups walk a map's tiles
    int32_t blocks = tiles_x * tiles_y;
    int32_t per_cu = (int32_t)((size_t)m->lds_per_cu / (smem + 256));
    if (per_cu > GMS_LIK_WG_PER_CU) per_cu = GMS_LIK_WG_PER_CU;
    if (per_cu < 1) per_cu = 1;
    const int64_t resident = (int64_t)per_cu * m->n_cus;
    // (4096 maps of 32 tiles: 32 / 16 / 8 / 4 workgroups per map 889 / 886 / 850 / 916 us; 500 maps of 8 tiles: 8 / 4 / 2 / 1: 25.3 / 26.8 / 27.4 / 32.1 us)
    while (blocks > 1 && (int64_t)blocks * n > 32 * resident) blocks = (blocks + 1) / 2;
    dim3 grid((unsigned)blocks, (unsigned)n);
#define SLK_LAUNCH(KH)                                                                                                          \
    do {                                                                                                                          \
        if (smem > 48 * 1024)                                                                                                     \
            hipFuncSetAttribute(reinterpret_cast<const void *>(&k_slam_likelihood<KH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
        hipLaunchKernelGGL((k_slam_likelihood<KH>), grid, dim3(256), smem, m->stream, m->gd, sb, m->d_taps, tiles_x, tiles_y);          \
    } while (0)
    if (k == 3) SLK_LAUNCH(3);
    else if (k == 5) SLK_LAUNCH(5);
    else SLK_LAUNCH(0);
#undef SLK_LAUNCH
}

int64_t gms_slam_code_words(int64_t cells) { return slam_code_words(cells); }
// likelihoodData of all n particles from plane `plane` of their class planes (d_code [n][2][code_words])
void gms_launch_slam_likelihood_codes(gms_map *m, const SlamBufs &sb, int64_t code_words, int32_t n, int32_t plane) {
    ProfScope ps(m, GMS_K_LIKELIHOOD);
    const int32_t k = m->lik_kh;
    const int32_t tiles_x = (m->gd.W + LK_TW - 1) / LK_TW, tiles_y = (m->gd.H + LK_TH - 1) / LK_TH;
    const size_t smem = gms_likelihood_lds_bytes(m->gd.khalf, k != 0);
    int32_t blocks = tiles_x * tiles_y;
    int32_t per_cu = (int32_t)((size_t)m->lds_per_cu / (smem + 256));
    if (per_cu > GMS_LIK_WG_PER_CU) per_cu = GMS_LIK_WG_PER_CU;
    if (per_cu < 1) per_cu = 1;
    const int64_t resident = (int64_t)per_cu * m->n_cus;
    while (blocks > 1 && (int64_t)blocks * n > 32 * resident) blocks = (blocks + 1) / 2;
    dim3 grid((unsigned)blocks, (unsigned)n);
#define SLK_LAUNCH(KH)                                                                                                          \
    do {                                                                                                                          \
        if (smem > 48 * 1024)                                                                                                     \
            hipFuncSetAttribute(reinterpret_cast<const void *>(&k_slam_likelihood_codes<KH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
        hipLaunchKernelGGL((k_slam_likelihood_codes<KH>), grid, dim3(256), smem, m->stream, m->gd, sb, code_words, plane, m->d_taps, tiles_x, tiles_y); \
    } while (0)
    if (k == 3) SLK_LAUNCH(3);
    else if (k == 5) SLK_LAUNCH(5);
    else SLK_LAUNCH(0);
#undef SLK_LAUNCH
}

// plane 0 of particles first .. first + count - 1 from their logData
void gms_launch_slam_codes_from_log(gms_map *m, const SlamBufs &sb, int32_t first, int32_t count, int64_t code_words) {
    const int64_t words = (m->gd.cells + 15) / 16;
    hipLaunchKernelGGL(k_slam_codes_from_log, dim3((unsigned)((words + 255) / 256), (unsigned)count), dim3(256), 0, m->stream, sb, m->gd.cells, first,
                       code_words, words);
}

// dynamic LDS of k_slam_particle<NT, NP> without its count tile
static inline size_t slam_particle_fixed_lds(int32_t Bpad, int np) {
    return (size_t)Bpad * (sizeof(double) + sizeof(RayThr)) + (size_t)PS_WORDS * 64 * np * sizeof(uint64_t) + (size_t)64 * np * sizeof(PsRay);
}

// SLAM.update's per-particle body for all n particles of pf (one map each, d_log / d_lik [n][cells]); motion may be NULL
// sb.code (may be NULL): the particles' class planes [n][2][code_words], kept in step with logData; with them field_in_memory may be
// false: the field is then evaluated at the scan's end points from the planes instead of read from sb.lik
void gms_launch_slam_particle(gms_pf *pf, const gms_beam *d_beams, int32_t B, const SlamBufs &sb, bool field_in_memory, const MotionModel *motion,
                              int32_t integrate, int64_t code_words) {
    const uint32_t *d_code = sb.code[0];
    gms_map *m = pf->map;
    MotionArgs mo;
    mo.on = 0; mo.d_center = mo.d_theta = mo.d_center_sd = mo.d_theta_sd = 0.0; mo.seed = mo.sequence = 0; mo.index0 = pf->offset;
    if (motion) {
        mo.on = 1; mo.d_center = motion->d_center; mo.d_theta = motion->d_theta; mo.seed = motion->seed; mo.sequence = motion->sequence;
        mo.d_center_sd = (0.01 + fabs(motion->d_center) * 0.05) / 2;             // Odometry.java:63
        mo.d_theta_sd = 5 * (3.141592653589793 / 180.0) + 0.1 * fabs(motion->d_theta);   // :64
    }
    ProfScope ps(m, GMS_K_SCORE);
    const int32_t Bpad = (B + 7) & ~7;
    constexpr int NP = 2;
    const size_t fixed = slam_particle_fixed_lds(Bpad, NP) + (d_code ? (size_t)code_words * 4 + (size_t)Bpad * 4 : 0);
    // the count tile: the whole map when two workgroups then still share a CU's LDS, else whatever one workgroup can have (the kernel
    // walks the scan's box in bands of rows when it is larger)
    const size_t lds_wg = (size_t)m->lds_per_cu - 4096;                         // (static LDS of the kernel -- 1.2 KiB per workgroup --, allocation granularity)
    size_t cells = (size_t)m->gd.cells;
    size_t tile = cells;
    if (fixed + tile * 4 > lds_wg / 2) {
        const size_t room = lds_wg > fixed ? (lds_wg - fixed) / 4 : 0;
        if (tile > room) tile = room;
    }
    if (m->slam_tile_cells > 0 && tile > (size_t)m->slam_tile_cells) tile = (size_t)m->slam_tile_cells;
    if (tile < (size_t)m->gd.W) tile = (size_t)m->gd.W;                         // one row at least (refused at creation if even that cannot fit)
    // a map that does not fit as 32-bit cells: 16-bit cells where a workgroup finds that its scan allows them (k_slam_particle)
    const int32_t narrow_allowed = tile < cells && B <= 255 ? 1 : 0;
    const size_t smem = fixed + tile * 4;
    // Workgroup shape: 512 lanes when two workgroups then share a CU (their latency chains -- pose, factors, the product, the walks,
    // the read-modify-write of the touched cells -- overlap), 1024 when the tile leaves room for one only.  GMS_SLAM_THREADS forces one.
    int32_t threads = smem <= lds_wg / 2 ? 512 : 1024;
    if (m->slam_threads == 512 || m->slam_threads == 1024) threads = m->slam_threads;
#define PS_LAUNCH(NT, NA, CD)                                                                                                           \
    do {                                                                                                                                \
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_slam_particle<NT, NP, NA, CD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
        hipLaunchKernelGGL((k_slam_particle<NT, NP, NA, CD>), dim3((unsigned)pf->n), dim3(NT), smem, m->stream, m->gd, d_beams, B, Bpad, sb,           \
                           field_in_memory ? 1 : 0, pf->d_pose, pf->d_cs, pf->d_w, pf->d_logw, mo, integrate, (int32_t)(tile * 4), (int32_t)code_words, \
                           m->d_taps, m->taps_plain);                                                                                   \
    } while (0)
#define PS_LAUNCH2(NT, NA) do { if (d_code) PS_LAUNCH(NT, NA, true); else PS_LAUNCH(NT, NA, false); } while (0)
    if (threads == 512) { if (narrow_allowed) PS_LAUNCH2(512, true); else PS_LAUNCH2(512, false); }
    else { if (narrow_allowed) PS_LAUNCH2(1024, true); else PS_LAUNCH2(1024, false); }
#undef PS_LAUNCH2
#undef PS_LAUNCH
    pf->pending_nseg = 0;
    pf->score_fresh = 1;
}

void gms_launch_slam_trace(gms_pf *pf, const gms_beam *d_beams, int32_t B, int32_t particle, int32_t *d_cells, uint8_t *d_cls, int32_t cap, int32_t *d_counts) {
    gms_map *m = pf->map;
    hipLaunchKernelGGL(k_slam_trace, dim3(1), dim3(256), 0, m->stream, m->gd, d_beams, B, pf->d_pose, pf->d_cs, particle, d_cells, d_cls, cap, d_counts);
}

struct RefinePlan {
    bool ok, ldsf;
    int32_t khf, Bpad, fp, nt_batch;               // khf: 0 the field is read from memory (staged or looked up), 3 / 5 computed in LDS from the class plane
    size_t smem;
};
static RefinePlan slam_refine_plan(const gms_map *m, int32_t B, int32_t field_in_lds, int64_t code_words) {
    RefinePlan r{};
    if (m->gd.W > 65535 || m->gd.H > 65535) return r;
    r.Bpad = B < 8 ? 8 : (B + 7) & ~7;
    const int32_t nt = slam_lattice_steps(SR_THETA_SPAN, SR_THETA_STEP, nullptr);
    const int32_t nc = 2 * slam_lattice_steps(SR_XSPAN, SR_TRANS_STEP, nullptr);
    r.fp = m->gd.W + 1;
    const size_t beams_b = (size_t)r.Bpad * 16, tab1_b = (size_t)nc * r.Bpad * 2, field_b = (((size_t)(m->gd.H + 1) * r.fp + 1) & ~(size_t)1) * 8;
    const size_t room = (size_t)m->lds_per_cu - 2048;                            // (static LDS: 0.5 KB; allocation granularity)
    if (beams_b + tab1_b > room) return r;
    // the field in LDS: its 16-bit table entries must hold (H + 1) * pitch, and one theta step's tables must fit beside it
    r.ldsf = field_in_lds != 0 && (size_t)(m->gd.H + 1) * r.fp + 512 <= 65535 && field_b + beams_b + tab1_b <= room;    // (+ the field's own LDS offset / 8, which the y entries carry: the static LDS is below 4 KB)
    // (the field in memory: 48 KB of tables per workgroup, so that three workgroups -- 64 registers, 30 wavefronts -- share a CU and
    //  hide each other's misses: 4096 x 256^2 x 180 beams 2.33 ms with 96 KB and one workgroup per CU, 1.69 with two, 1.61 with three)
    static const size_t tab_budget = []() { const char *e = getenv("GMS_SLAM_REFINE_TAB_KB"); return (size_t)(e && atoi(e) > 0 ? atoi(e) : 48) * 1024; }();
    const size_t left = (r.ldsf ? room - field_b : std::max(tab_budget, beams_b + tab1_b)) - beams_b;
    r.nt_batch = (int32_t)(left / tab1_b);
    if (r.nt_batch > nt) r.nt_batch = nt;
    if (r.nt_batch < 1) r.nt_batch = 1;
    r.nt_batch = (nt + (nt + r.nt_batch - 1) / r.nt_batch - 1) / ((nt + r.nt_batch - 1) / r.nt_batch);  // batches of equal size (6 + 4 theta steps keep ten wavefronts busy for two rounds each)
    r.smem = (r.ldsf ? field_b : 0) + beams_b + (size_t)r.nt_batch * tab1_b;
    // ... computed there from the class plane: the compile-time blur kernels' conditions (gms_map::lik_kh: 7 or 11 plain taps), a thread
    // per column in the vertical pass and per strip in the horizontal one.  field_in_lds 2: not this form (tests of the staged one)
    if (r.ldsf && field_in_lds != 2 && code_words > 0 && m->lik_kh != 0 && m->gd.W <= SR_NT_OF(true, m->lik_kh) && m->gd.H >= m->lik_kh) r.khf = m->lik_kh;
    r.ok = true;
    return r;
}
// whether the refinement of a scan of B beams computes its field itself (no k_slam_likelihood launch in front of it)
bool gms_slam_refine_from_planes(const gms_map *m, int32_t B, int32_t field_in_lds, int64_t code_words) {
    const RefinePlan r = slam_refine_plan(m, B, field_in_lds, code_words);
    return r.ok && r.khf != 0;
}

// findBestPose for every particle of pf against its own field (SLAM.java:96): sb.lik [n][cells], or -- code_words != 0 and
// gms_slam_refine_from_planes -- the field computed from the particle's class plane inside the workgroup.  motion (may be NULL): the
// motion-model sample of SLAM.java:90 is drawn first.  field_in_lds: -1 the launcher decides (whenever it fits), 0 never, 2 staged
// from memory wherever it fits (tests of the other forms).  Returns false (nothing launched) where a theta step's tables do not fit
// a workgroup's LDS (scans of more than ~2600 beams) or a map side does not fit their 16-bit entries.
bool gms_launch_slam_refine(gms_pf *pf, const gms_beam *d_beams, int32_t B, const SlamBufs &sb, const MotionModel *motion, int32_t field_in_lds,
                            int64_t code_words) {
    gms_map *m = pf->map;
    const RefinePlan r = slam_refine_plan(m, B, field_in_lds, code_words);
    if (!r.ok) return false;
    MotionArgs mo;
    mo.on = 0; mo.d_center = mo.d_theta = mo.d_center_sd = mo.d_theta_sd = 0.0; mo.seed = mo.sequence = 0; mo.index0 = pf->offset;
    if (motion) {
        mo.on = 1; mo.d_center = motion->d_center; mo.d_theta = motion->d_theta; mo.seed = motion->seed; mo.sequence = motion->sequence;
        mo.d_center_sd = (0.01 + fabs(motion->d_center) * 0.05) / 2;             // Odometry.java:63
        mo.d_theta_sd = 5 * (3.141592653589793 / 180.0) + 0.1 * fabs(motion->d_theta);   // :64
    }
    ProfScope ps(m, GMS_K_REFINE);
#define SR_LAUNCH(LF, KF)                                                                                                               \
    do {                                                                                                                                \
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_slam_refine<LF, KF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)r.smem); \
        hipLaunchKernelGGL((k_slam_refine<LF, KF>), dim3((unsigned)pf->n), dim3(SR_NT_OF(LF, KF)), r.smem, m->stream, m->gd, d_beams, B, r.Bpad, sb, \
                           pf->d_pose, pf->d_cs, mo, r.fp, r.nt_batch, (int32_t)code_words, m->d_taps);                                \
    } while (0)
    if (r.khf == 3) SR_LAUNCH(true, 3);
    else if (r.khf == 5) SR_LAUNCH(true, 5);
    else if (r.ldsf) SR_LAUNCH(true, 0);
    else SR_LAUNCH(false, 0);
#undef SR_LAUNCH
    return true;
}

void gms_launch_slam_gather(gms_pf *pf, const SlamBufs &sb, int32_t what, const int32_t *d_idx, int32_t *d_idx_keep, int64_t code_words) {
    gms_map *m = pf->map;
    ProfScope ps(m, GMS_K_MAPCOPY);
    const int64_t cells = m->gd.cells;
    if (what == 3) {
        // both arrays at once: 256 lanes x 16 bytes x 2 in flight per array = 8 KiB of each array per workgroup pass
        int64_t chunks = (cells / 2 + 511) / 512;
        if (chunks < 1) chunks = 1;
        while (chunks > 1 && chunks * pf->n > 65536) chunks = (chunks + 1) / 2;
        hipLaunchKernelGGL(k_slam_gather_maps, dim3((unsigned)chunks, (unsigned)pf->n), dim3(256), 0, m->stream, sb, d_idx, cells);
        if (sb.code[0]) hipLaunchKernelGGL(k_slam_gather_codes, dim3((unsigned)pf->n), dim3(256), 0, m->stream, sb, d_idx, 2 * code_words);
        return;
    }
    // one array: 256 lanes x 16 bytes x GATHER_U in flight = 32 KiB of the array per workgroup pass
    // ... while that is a residency or two of workgroups (500 maps of 120 x 120: 2000); a copy that streams from memory does better with
    // a quarter of that per workgroup and pass (4096 x 256 x 256: 443 us at 32 KiB, 409 at 16, 367 at 8, 496 at 4; 1024 x 256 x 256: 147 / 122)
    int64_t per = 256 * GATHER_U;
    const bool streams = ((cells / 2 + per - 1) / per) * pf->n > 4096;
    if (streams) per = 512;
    int64_t chunks = (cells / 2 + per - 1) / per;
    if (chunks < 1) chunks = 1;
    while (chunks > 1 && chunks * pf->n > 262144) chunks = (chunks + 1) / 2;
    // the class planes ride in the copy's last workgroup of every map while the launch is one residency of workgroups (500 x 120 x 120:
    // resample() 21.6 us against 24.1 with a launch of their own); where the copy streams from memory that workgroup's nine round
    // trips in a row hold the launch's tail open (4096 x 256 x 256: 0.60 ms against 0.38): there they get a launch of their own
    const bool separate = streams;
    hipLaunchKernelGGL(k_slam_gather_one, dim3((unsigned)chunks, (unsigned)pf->n), dim3(256), 0, m->stream, sb, what == 2 ? 1 : 0, d_idx, cells, d_idx_keep,
                       what == 1 && sb.code[0] && !separate ? 2 * code_words : (int64_t)0);
    if (what == 1 && sb.code[0] && separate) hipLaunchKernelGGL(k_slam_gather_codes, dim3((unsigned)pf->n), dim3(256), 0, m->stream, sb, d_idx, 2 * code_words);
}

// GridMapApp.calculateCombined over the particles' maps (J/app/GridMapApp.java:439-458) into a single map's logData
void gms_launch_slam_export_records(gms_pf *pf, const SlamBufs &sb, const int32_t *d_list, int32_t count, int64_t code_words, double *d_dst) {
    gms_map *m = pf->map;
    ProfScope ps(m, GMS_K_MAPCOPY);
    const int64_t rec = m->gd.cells + code_words;
    int64_t chunks = (rec + 256 * 8 - 1) / (256 * 8);
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(k_slam_export_records, dim3((unsigned)chunks, (unsigned)count), dim3(256), 0, m->stream, sb, d_list, m->gd.cells, code_words, d_dst);
}
void gms_launch_slam_shard_gather(gms_pf *pf, const SlamBufs &sb, const int32_t *d_src_local, const int32_t *d_recv_pos, const double *d_recv, int64_t code_words) {
    gms_map *m = pf->map;
    ProfScope ps(m, GMS_K_MAPCOPY);
    const int64_t rec = m->gd.cells + code_words;
    int64_t chunks = (rec + 256 * 8 - 1) / (256 * 8);
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(k_slam_shard_gather, dim3((unsigned)chunks, (unsigned)pf->n), dim3(256), 0, m->stream, sb, d_src_local, d_recv_pos, d_recv, m->gd.cells,
                       code_words);
}

__global__ void k_slam_combine(SlamBufs sb, int32_t n, int64_t cells, double *__restrict__ out) { combine_body(sb_log(sb, sb.epoch[0] & 1), n, cells, out); }
void gms_launch_slam_combine(gms_map *dst, const SlamBufs &sb, int32_t n) {
    hipLaunchKernelGGL(k_slam_combine, dim3(2048), dim3(256), 0, dst->stream, sb, n, dst->gd.cells, dst->d_log);
}
