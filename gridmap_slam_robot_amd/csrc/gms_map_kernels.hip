// gms_map_kernels.hip -- gfx950 kernels for the log-odds map update and the likelihood field.
//
//   k_raycast     GridMap.integrateObservation / applyMeasurement + RayIterator + SensorModel
//                 (J/slam/GridMap.java:173-228, J/slam/RayIterator.java:65-130,
//                  J/slam/SensorModel.java:31-41): a producer wavefront runs the serial float recurrence of the
//                 4-connected DDA, one lane per ray; consumer wavefronts turn the published decision words into
//                 per-cell (n_free, n_occ) counts with 32-bit atomics meanwhile.
//   k_raycast_tile  the same for batched maps: 64 consecutive beams per workgroup, counts accumulated in an LDS
//                 tile of the wedge's bounding box and flushed row by row.
//   k_apply       log += n_free*l_free + n_occ*l_occ over the touched bounding box (the three
//                 possible increments of GridMap.java:223 are constants: J/app/Util.java:35-37).
//   k_likelihood  GridMap.computeLikelihoodMap (GridMap.java:233-250) + Util.doGaussianBlurdSeparable
//                 (J/app/Util.java:378-426): threshold, horizontal and vertical pass fused through
//                 an LDS tile; every sum runs in the reference's tap order, without FMA.
//
// HBM layout: log/lik [n_maps][H][W] doubles, row-major x + y*W as GridMapData's arrays;
// cnt [n_maps][H][W] uint32 = n_free | n_occ << 16, all zero between calls.
#include "gms_device.h"

// ---------------------------------------------------------------------------------------------
// bbox encoding: {max(W-1-x), max(H-1-y), max(x+1), max(y+1)}; all-zero == empty, so a memset
// resets it and every update is an atomicMax.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bbox_decode(const int32_t *bb, int32_t W, int32_t H, int32_t &x0, int32_t &y0,
                                            int32_t &x1, int32_t &y1) {
    x0 = W - 1 - bb[0]; y0 = H - 1 - bb[1]; x1 = bb[2]; y1 = bb[3];   // [x0,x1) x [y0,y1)
}

// ---------------------------------------------------------------------------------------------
// Ray cast in two phases inside one workgroup of RC_RAYS rays x 64 lanes.
//
// The only truly serial part of RayIterator.next (J/slam/RayIterator.java:117-123) is the float
// `error` recurrence that decides "y step or x step"; everything else (cell coordinates, distance,
// sensor class, map update) is a function of how many y steps precede step k.
//   phase A  one lane per ray runs just that recurrence (compare, select, add: the exact float
//            operations of the reference, in order) and publishes the decisions as bit words in LDS,
//            with the running count of y steps per word, every 32 steps.
//   phase B  a wavefront per 64-step block: lane j owns step k = 64 i + j, rebuilds (x_k, y_k) from a
//            popcount, and does the per-cell work of GridMap.applyMeasurement (GridMap.java:215-223)
//            in parallel -- while phase A is still walking the rest of the ray.
// (Tried and measured slower at C3: gathering the decisions by wave ballot and packing them on the scalar
// unit -- 3 VALU per step instead of 5, but the scalar packing does not overlap the recurrence: 19.3-20.6 us
// against 18.1 us for the per-lane words below.)
// hasNext (:108) stops at the first out-of-bounds cell; x and y move monotonically, so step k is
// emitted iff cell 0 and cell k are both inside, and no walk is longer than W + H + 1 steps.
// ---------------------------------------------------------------------------------------------
// Rays per workgroup of the fused kernel.  One scan (720 rays) is latency-bound: 4 rays per workgroup spread it
// over the most CUs (measured at C3: 4 -> 18 us, 8 -> 20 us, 16 -> 24 us).  Batched maps (tens of thousands of
// rays) are throughput-bound and use 16 (a split phase-A / phase-B pair of kernels was measured slower at
// config 5: 0.30 vs 0.22 ms; the u32 atomics, ~150 G/s scattered, are the floor there).

struct RayMeta {
    int32_t x0, y0, x_inc, y_inc, n_eff, hit;
    float sx, sy, measured;
    int32_t hx, hy;            // far corner of a box that contains every cell of the walk (ray_meta), inside the map
};

// Decision words travel from the phase-A wavefront to the phase-B wavefronts through LDS slots of 8 bytes,
// (RC_VALID | number of y steps before the word) << 32 | decision word: one ds_write_b64, so a reader sees a slot either
// empty (cleared at kernel start) or complete.  Slot of word w of ray `slot`: slots[w * stride + slot].
#define RC_VALID 0x80000000u
#define RC_BOX_MARGIN 5           // cells beyond the end point's cell that a walk's box allows for (ray_meta)

// The slots are written and polled with explicit DS instructions: through a generic pointer the compiler emits flat
// accesses, and marking them volatile adds a full wait after every one (the producer would stall on each publish).
// LDS executes a wavefront's DS instructions in order, a 64-bit access is a single one, and the "memory" clobber
// keeps the compiler from caching or reordering around them.
typedef __attribute__((address_space(3))) uint64_t gms_lds_u64;
typedef __attribute__((address_space(3))) uint32_t gms_lds_u32;
typedef __attribute__((address_space(3))) double gms_lds_f64;
__device__ __forceinline__ uint32_t lds_offset(const uint64_t *p) { return (uint32_t)(uintptr_t)(const gms_lds_u64 *)p; }
__device__ __forceinline__ void lds_publish_u64(uint64_t *p, uint64_t v) {
    asm volatile("ds_write_b64 %0, %1" : : "v"(lds_offset(p)), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_poll_2xu64(const uint64_t *p0, const uint64_t *p1, uint64_t &a, uint64_t &c) {
    asm volatile("ds_read_b64 %0, %2\n\t"
                 "ds_read_b64 %1, %3\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(c) : "v"(lds_offset(p0)), "v"(lds_offset(p1)) : "memory");
}

// ray_init + the parts of the walk that do not depend on the recurrence
__device__ __forceinline__ RayMeta ray_meta(const GridDev &g, const RayIn &ray, RayDev &r) {
    RayMeta mt;
    ray_init(r, ray.sx + 0.5f, ray.sy + 0.5f, ray.ex + 0.5f, ray.ey + 0.5f, g.extra);   // GridMap.java:210
    mt.x0 = r.x; mt.y0 = r.y; mt.x_inc = r.x_inc; mt.y_inc = r.y_inc;
    mt.sx = ray.sx; mt.sy = ray.sy; mt.measured = ray.measured; mt.hit = ray.hit;
    const bool inb0 = !(r.x < 0 || r.x >= g.W || r.y < 0 || r.y >= g.H);
    mt.n_eff = (inb0 && r.n > 0) ? min(r.n, g.W + g.H + 1) : 0;
    // Where the walk can go.  It is monotonic in x and in y and makes n - 1 moves; the recurrence keeps `error` inside (-dx, dy], so
    // after k moves the number of x moves is within (max(dx, dy) + drift) / (dx + dy) < 1.3 of k dx / (dx + dy) (the float drift of
    // `error` over a walk is below a quarter of a cell), and n - 1 = extra + |floor x1 - x| + |floor y1 - y| (RayIterator.java:75-100):
    // at most |floor x1 - x| + extra + 4.3 moves along x, likewise along y.  RC_BOX_MARGIN = 5 makes the box [x0, hx] x [y0, hy] a
    // superset of the visited cells; the scan's dirty box is raised from these boxes BEFORE the walk (no reduction and no barrier
    // at the kernel's end), and a counted cell that were ever outside its ray's box would still raise the dirty box by itself
    // (ray_phase_b).
    const int64_t ax = llabs((int64_t)j_d2i(floor((double)(ray.ex + 0.5f))) - (int64_t)r.x) + g.extra + RC_BOX_MARGIN;
    const int64_t ay = llabs((int64_t)j_d2i(floor((double)(ray.ey + 0.5f))) - (int64_t)r.y) + g.extra + RC_BOX_MARGIN;
    const int32_t moves = mt.n_eff > 0 ? mt.n_eff - 1 : 0;
    mt.hx = min(max(r.x + r.x_inc * (int32_t)min((int64_t)moves, ax), 0), g.W - 1);
    mt.hy = min(max(r.y + r.y_inc * (int32_t)min((int64_t)moves, ay), 0), g.H - 1);
    return mt;
}
// encoded box of one ray's walk (all zero: the ray visits nothing)
__device__ __forceinline__ void ray_box(const GridDev &g, const RayMeta &mt, int32_t bb[4]) {
    if (mt.n_eff <= 0) { bb[0] = bb[1] = bb[2] = bb[3] = 0; return; }
    bb[0] = g.W - 1 - min(mt.x0, mt.hx); bb[1] = g.H - 1 - min(mt.y0, mt.hy);
    bb[2] = max(mt.x0, mt.hx) + 1;       bb[3] = max(mt.y0, mt.hy) + 1;
}
// raise the scan's dirty box by the union of the lanes' boxes: wave butterfly, then four atomics by lane 0, not waited for
__device__ __forceinline__ void bbox_raise_wave(int32_t bb[4], int32_t lane, int32_t *__restrict__ bbox_map) {
#define GMS_STEP_(O) { _Pragma("unroll") for (int q = 0; q < 4; q++) bb[q] = max(bb[q], wave_xor<O>(bb[q])); }
    GMS_BUTTERFLY(GMS_STEP_)
#undef GMS_STEP_
    if (lane == 0 && bb[2] > 0)
        for (int q = 0; q < 4; q++) atomicMax(&bbox_map[q], bb[q]);
}

// phase A for one ray: the float `error` recurrence, 32 decisions per published slot.  Resumable: words [w0, w1) are
// produced (slot index w - w0), `err` and `ycount` carry the state from one call to the next.
struct RayWalk {
    float err, ndx, dy;
    uint32_t ycount;
};
__device__ __forceinline__ RayWalk ray_walk_begin(const RayDev &r) {
    RayWalk k;
    k.err = r.error; k.ndx = -r.dx; k.dy = r.dy; k.ycount = 0;
    // Plain registers (no |x| / -x source modifiers folded in): v_cndmask then keeps its e32 form with the implicit vcc,
    // which needs no wait states after v_cmp (the e64 form the compiler picked cost an s_nop 1 per step).
    asm volatile("" : "+v"(k.ndx), "+v"(k.dy));
    return k;
}
__device__ __forceinline__ void ray_phase_a(RayWalk &k, int32_t w0, int32_t w1, uint64_t *__restrict__ slots, int32_t stride, int32_t slot) {
    float err = k.err;
    const float ndx = k.ndx, dy = k.dy;
    uint32_t ycount = k.ycount;
    for (int32_t w = w0; w < w1; ++w) {
        uint32_t word = 0;
        float t;
        // one step of RayIterator.next (RayIterator.java:117-123) in four VALU instructions, the wavefront's issue rate being
        // the limit (one lane per ray, one phase-A wavefront per SIMD: 4.6 clocks per instruction):
        //   vcc  = 0 < err                       c = error > 0                               (:117)
        //   t    = vcc ? -dx : dy
        //   err  = err + t                       error -= dx  /  error += dy                 (:119 / :122; a - b == a + (-b))
        //   word = word + word + vcc             shift the decision in (bit 31 - j after 32 steps)
        // All 32 steps of a word are ONE asm statement: between two statements the compiler pads a hazard it cannot rule out
        // with an s_nop, a fifth issue slot per step (tools/microbench/dda_chain.hip: 24.2 -> 18.5 clocks per step).
#define GMS_DDA_STEP "v_cmp_lt_f32_e32 vcc, 0, %0\n\tv_cndmask_b32_e32 %2, %3, %4, vcc\n\tv_add_f32_e32 %0, %0, %2\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc\n\t"
#define GMS_DDA_STEP4 GMS_DDA_STEP GMS_DDA_STEP GMS_DDA_STEP GMS_DDA_STEP
        asm volatile(GMS_DDA_STEP4 GMS_DDA_STEP4 GMS_DDA_STEP4 GMS_DDA_STEP4 GMS_DDA_STEP4 GMS_DDA_STEP4 GMS_DDA_STEP4 GMS_DDA_STEP4
                     : "+v"(err), "+v"(word), "=&v"(t)
                     : "v"(dy), "v"(ndx)
                     : "vcc");
#undef GMS_DDA_STEP4
#undef GMS_DDA_STEP
        word = __brev(word);                           // decision j at bit j
        lds_publish_u64(&slots[(w - w0) * stride + slot], ((uint64_t)(RC_VALID | ycount) << 32) | (uint64_t)word);
        ycount += __popc(word);
    }
    k.err = err; k.ycount = ycount;
}

// phase B for 64 consecutive steps [64 blk, 64 blk + 64) of one ray, executed by one wavefront: lane j owns step
// k = 64 blk + j, rebuilds (x_k, y_k) from a popcount, and does the per-cell work of GridMap.applyMeasurement
// (GridMap.java:215-223).  Waits (LDS polling) until phase A has published the two words it needs.  Returns the
// number of cells of the block that are inside the map.
struct CountTile {         // per-workgroup accumulation tile in LDS (batched maps): cells [x0, x0 + w) x [y0, y0 + h); w == 0: none.
    uint32_t *cells;       // a cell is 16 bits, n_free | n_occ << 8 (at most 64 rays x (1 + extra) visits per workgroup), two per word
    int32_t x0, y0, w, h;
};

template <bool TRACE>
__device__ __forceinline__ int32_t ray_phase_b(const GridDev &g, const RayMeta &mt, const uint64_t *__restrict__ slots, int32_t stride,
                                               int32_t slot, int32_t blk, int32_t lane, uint32_t *__restrict__ mcnt, int32_t *__restrict__ bbox_map,
                                               int32_t b, int32_t *__restrict__ t_cells, uint8_t *__restrict__ t_cls, int32_t cap,
                                               const CountTile tile = CountTile{nullptr, 0, 0, 0, 0}, int32_t w_base = 0) {
    const int32_t nwords = (mt.n_eff + 31) >> 5;
    const int32_t w0 = 2 * blk - w_base, w1 = min(2 * blk + 1, nwords - 1) - w_base;
    uint64_t a, c;
    for (;;) {                                         // wave-uniform: every lane reads the same two slots
        lds_poll_2xu64(&slots[w0 * stride + slot], &slots[w1 * stride + slot], a, c);
        if (((a & c) >> 63) != 0u) break;
        __builtin_amdgcn_s_sleep(2);
    }
    const int32_t k = blk * 64 + lane;
    bool valid = false;
    if (k < mt.n_eff) {
        const uint64_t sl = lane < 32 ? a : c;
        const int32_t j = lane & 31;
        const int32_t ny = (int32_t)(((uint32_t)(sl >> 32) & ~RC_VALID) + __popc((uint32_t)sl & ((1u << j) - 1u)));
        const int32_t nx = k - ny;
        const int32_t cx = mt.x0 + mt.x_inc * nx, cy = mt.y0 + mt.y_inc * ny;
        valid = !(cx < 0 || cx >= g.W || cy < 0 || cy >= g.H);                            // RayIterator.java:108
        if (valid) {
            const float d = cell_distance(mt.sx, mt.sy, cx, cy);                          // GridMap.java:215-217
            const int32_t cls = sensor_class(d, mt.measured, mt.hit, g.half_tol);         // :223
            if (TRACE) {
                if (k < cap) {
                    const size_t o = (size_t)b * cap + k;
                    if (t_cells) { t_cells[2 * o] = cx; t_cells[2 * o + 1] = cy; }
                    if (t_cls) t_cls[o] = (uint8_t)cls;
                }
            } else if (cls != 1) {
                const uint32_t inc = cls == 0 ? 1u : 0x10000u;
                const uint32_t ux = (uint32_t)(cx - tile.x0), uy = (uint32_t)(cy - tile.y0);
                if (ux < (uint32_t)tile.w && uy < (uint32_t)tile.h) {                         // (w == 0: no tile)
                    const uint32_t ci = uy * (uint32_t)tile.w + ux;                           // LDS; flushed row by row afterwards
                    __hip_atomic_fetch_add((gms_lds_u32 *)(tile.cells) + (ci >> 1), (cls == 0 ? 1u : 0x100u) << ((ci & 1u) << 4),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    atomicAdd(&mcnt[(size_t)cy * g.W + cx], inc);
                    // the dirty box was raised from the rays' boxes before the walk (ray_meta); a cell outside its ray's box --
                    // there should be none -- raises it here
                    if (__builtin_expect(cx < min(mt.x0, mt.hx) || cx > max(mt.x0, mt.hx) || cy < min(mt.y0, mt.hy) || cy > max(mt.y0, mt.hy), 0)) {
                        atomicMax(&bbox_map[0], g.W - 1 - cx); atomicMax(&bbox_map[1], g.H - 1 - cy);
                        atomicMax(&bbox_map[2], cx + 1);       atomicMax(&bbox_map[3], cy + 1);
                    }
                }
            }
        }
    }
    return __popcll(__ballot(valid));
}

// One workgroup = RC_RAYS rays, NWAVES wavefronts.  Wavefront 0 is the producer: lane r runs phase A of ray r and
// publishes a slot every 32 steps.  The other NWAVES - 1 wavefronts are consumers: they take the 64-step blocks of
// all rays round-robin (block-major, the order in which the producer publishes them) and run phase B as soon as a
// block's words are there, so the cell work -- distance, sensor class, count atomics -- hides under the recurrence
// instead of following it (measured at C3: 17.4 -> see DESIGN.md).  The producer never waits for a consumer.
// (bx, by) = workgroup / map index and `smem` = the dynamic LDS: the body is shared by k_raycast and by the
// launch that runs the ray cast beside the weight normalisation (gms_fused_kernels.hip); pose_lds, when given,
// replaces poses[] (a pose the workgroup has just folded itself).
template <bool TRACE, int RC_RAYS, int NWAVES = RC_RAYS>
__device__ __forceinline__ void
raycast_body(const GridDev &g, const gms_beam *__restrict__ beams, int32_t B, int32_t beam_stride,
             const float *__restrict__ poses, int32_t pose_stride, const RayIn *__restrict__ single,
             uint32_t *__restrict__ cnt, int32_t *__restrict__ bbox, int32_t *__restrict__ t_cells,
             uint8_t *__restrict__ t_cls, int32_t cap, int32_t *__restrict__ t_counts, int32_t nw_max,
             uint32_t bx, uint32_t by, unsigned char *smem, const float *pose_lds, int32_t first_blk = 0) {
    // first_blk = 1: the rays' first 64 steps (block 0) are counted by the near-field workgroups (raycast_near_body)
    uint64_t *s_slots = reinterpret_cast<uint64_t *>(smem);            // [nw_max][RC_RAYS]
    __shared__ RayMeta s_meta[RC_RAYS];
    __shared__ int32_t s_count[RC_RAYS];

    const int32_t mi = (int32_t)by;
    const int32_t lane = threadIdx.x & 63;
    const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int32_t i = threadIdx.x; i < nw_max * RC_RAYS; i += NWAVES * 64) s_slots[i] = 0ull;
    RayDev r;
    r.dx = r.dy = r.error = 0.0f; r.x = r.y = r.x_inc = r.y_inc = r.n = 0;
    int32_t my_n_eff = 0;
    if (wave == 0 && lane < RC_RAYS) {
        const int32_t b = (int32_t)bx * RC_RAYS + lane;
        RayMeta mt;
        mt.n_eff = 0; mt.x0 = mt.y0 = mt.x_inc = mt.y_inc = mt.hit = 0; mt.sx = mt.sy = mt.measured = 0.0f;
        if (b < B) {
            RayIn ray;
            if (single) ray = *single;
            else ray = make_ray(g, beams[(size_t)mi * beam_stride + b], pose_lds ? pose_lds : poses + (size_t)pose_stride * mi);
            mt = ray_meta(g, ray, r);
        }
        my_n_eff = mt.n_eff;
        s_meta[lane] = mt;
        s_count[lane] = 0;
    }
    __syncthreads();
    GMS_STAMP(GMS_STAMP_ROW(2, blockIdx.x), 2);
    if (!TRACE && first_blk == 0 && wave == (NWAVES > 1 ? 1 : 0)) {
        // The scan's dirty box from the rays' boxes, up front (with near-field workgroups in the launch, they raise it for their
        // wedges).  On a CONSUMER wavefront, behind the set-up barrier: a few hundred workgroups raise the same four words, the
        // requests queue up on their way to the memory-side atomic unit, and the wavefront that issues them is held up with them
        // (the barrier itself waits for LDS only).  Issued by the producer's wavefront in front of the barrier they delayed the
        // walk: measured 0.7 us of k_norm_raycast on a 360-beam scan (c2_step_timeline.txt of that build: rays set up 2.7-6 us
        // after the pose).
        int32_t hb[4] = { 0, 0, 0, 0 };
        if (lane < RC_RAYS) ray_box(g, s_meta[lane], hb);
        bbox_raise_wave(hb, lane, bbox + 4 * mi);
    }
    if (wave == 0) {
        if (lane < RC_RAYS) {
            RayWalk wk = ray_walk_begin(r);
            ray_phase_a(wk, 0, (my_n_eff + 31) >> 5, s_slots, RC_RAYS, lane);
        }
        GMS_STAMP(GMS_STAMP_ROW(2, blockIdx.x), 3);
    } else {
        int32_t nblk_max = 0;
#pragma unroll
        for (int q = 0; q < RC_RAYS; q++) nblk_max = max(nblk_max, (s_meta[q].n_eff + 63) >> 6);
        constexpr int32_t NC = NWAVES - 1;
        for (int32_t q = wave - 1 + first_blk * RC_RAYS; q < nblk_max * RC_RAYS; q += NC) {
            const int32_t blk = q / RC_RAYS, ray = q - blk * RC_RAYS;
            const RayMeta mt = s_meta[ray];
            if (blk * 64 >= mt.n_eff) continue;
            const int32_t n = ray_phase_b<TRACE>(g, mt, s_slots, RC_RAYS, ray, blk, lane, TRACE ? nullptr : cnt + (size_t)mi * g.cells, TRACE ? nullptr : bbox + 4 * mi,
                                                 (int32_t)bx * RC_RAYS + ray, t_cells, t_cls, cap);
            if (TRACE && lane == 0) atomicAdd(&s_count[ray], n);
        }
        GMS_STAMP_T(64, GMS_STAMP_ROW(2, blockIdx.x), 4);
    }
    if (TRACE) {
        __syncthreads();
        const int32_t b = (int32_t)bx * RC_RAYS + (int32_t)threadIdx.x;
        if (threadIdx.x < RC_RAYS && b < B && t_counts) t_counts[b] = s_count[threadIdx.x];
    } else {
        GMS_STAMP(GMS_STAMP_ROW(2, blockIdx.x), 5);
    }
}

// ---------------------------------------------------------------------------------------------
// The near field of a scan.  Every ray of a scan leaves the same cell, and the cells around it are crossed by dozens to
// hundreds of rays: the start cell alone takes one count atomic per ray, all on one address, and same-address atomics
// serialise at ~10 ns each -- with the direct atomics of raycast_body the 720 rays of a C3 scan kept the memory-side atomic
// unit busy for 3.7-4.4 us after the last ray had been walked (measured by leaving the first 16 / 64 cells of every ray
// uncounted: k_raycast 17.1 -> 13.1 / 12.6 us, the scan step 54.0 -> 52.5 / 52.1 us).
// So the first 64 steps of every ray (block 0 of phase B) belong to NEAR-FIELD workgroups: one takes RCN_RAYS = 16
// CONSECUTIVE beams (a wedge), walks their first 64 steps itself (two decision words per ray: 64 steps of the recurrence, a
// fraction of a microsecond), counts them in an LDS tile of the wedge's near box (16-bit cells, ds_add) and flushes the tile
// row by row: one global atomic per touched cell and workgroup, consecutive lanes on consecutive cells -- 45 atomics on the
// start cell instead of 720.  The far-field workgroups skip block 0 (raycast_body, first_blk = 1).  The counts in d_cnt are the
// same (integer adds commute); a cell outside the tile, or a wedge whose box does not fit, takes the direct atomics.
// ---------------------------------------------------------------------------------------------
#ifndef RCN_RAYS
#define RCN_RAYS 16                     // beams per near-field workgroup (C3 scan step, us: 8 / 16 / 32 / 64 -> 53.1 / 52.1 / 52.0 / 55.0; without the near field 53.4)
#endif
#define RCN_TILE_CELLS 12288            // 24 KiB of 16-bit cells: the 64-step walks of a wedge that spans a half plane are at most 128 x 66 cells
#define RCN_LDS_BYTES (2 * RCN_RAYS * 8 + RCN_TILE_CELLS * 2)
static_assert(RCN_RAYS >= 1 && RCN_RAYS <= 64, "the near-field producer is one wavefront, a lane per ray");

__device__ __forceinline__ void
raycast_near_body(const GridDev &g, const gms_beam *__restrict__ beams, int32_t B, int32_t beam_stride,
                  const float *__restrict__ poses, int32_t pose_stride, uint32_t *__restrict__ cnt, int32_t *__restrict__ bbox,
                  uint32_t bx, uint32_t by, unsigned char *smem, const float *pose_lds) {
    uint64_t *s_slots = reinterpret_cast<uint64_t *>(smem);                                   // [2][RCN_RAYS]
    uint32_t *s_tile = reinterpret_cast<uint32_t *>(s_slots + 2 * RCN_RAYS);                  // [RCN_TILE_CELLS / 2]
    __shared__ RayMeta s_nmeta[RCN_RAYS];
    __shared__ int32_t s_nbox[4];                                                             // wedge box x0, y0, x1, y1 (inclusive)
    const int32_t mi = (int32_t)by;
    const int32_t lane = threadIdx.x & 63;
    const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int32_t nwaves = __builtin_amdgcn_readfirstlane((int32_t)(blockDim.x >> 6));
    for (int32_t i = threadIdx.x; i < 2 * RCN_RAYS; i += blockDim.x) s_slots[i] = 0ull;
    RayDev r;
    r.dx = r.dy = r.error = 0.0f; r.x = r.y = r.x_inc = r.y_inc = r.n = 0;
    int32_t my_nwords = 0;
    if (wave == 0) {
        const int32_t b = (int32_t)bx * RCN_RAYS + lane;
        RayMeta mt;
        mt.n_eff = 0; mt.x0 = mt.y0 = mt.x_inc = mt.y_inc = mt.hit = 0; mt.sx = mt.sy = mt.measured = 0.0f;
        int32_t bx0 = INT32_MAX, by0 = INT32_MAX, bx1 = INT32_MIN, by1 = INT32_MIN;
        if (lane < RCN_RAYS && b < B) {
            const RayIn ray = make_ray(g, beams[(size_t)mi * beam_stride + b], pose_lds ? pose_lds : poses + (size_t)pose_stride * mi);
            mt = ray_meta(g, ray, r);
            if (mt.n_eff > 0) {
                // the first 64 steps: at most 63 moves from the start cell, split between x and y roughly as dx : dy (+2 for
                // the rounding of the recurrence); a cell outside the box takes the direct-atomic path: the box is a hint
                const int32_t steps = min(mt.n_eff, 64) - 1;
                const float tot = r.dx + r.dy;
                const int32_t mx = tot > 0.0f ? min(steps, (int32_t)((float)steps * (r.dx / tot)) + 2) : 0;
                const int32_t my = tot > 0.0f ? min(steps, (int32_t)((float)steps * (r.dy / tot)) + 2) : 0;
                const int32_t ex = mt.x0 + mt.x_inc * mx, ey = mt.y0 + mt.y_inc * my;
                bx0 = max(min(mt.x0, ex), 0); bx1 = min(max(mt.x0, ex), g.W - 1);
                by0 = max(min(mt.y0, ey), 0); by1 = min(max(mt.y0, ey), g.H - 1);
            }
        }
        my_nwords = min((mt.n_eff + 31) >> 5, 2);
        if (lane < RCN_RAYS) s_nmeta[lane] = mt;
#define GMS_STEP_(O) { bx0 = min(bx0, wave_xor<O>(bx0)); by0 = min(by0, wave_xor<O>(by0)); bx1 = max(bx1, wave_xor<O>(bx1)); by1 = max(by1, wave_xor<O>(by1)); }
        GMS_BUTTERFLY(GMS_STEP_)
#undef GMS_STEP_
        if (lane == 0) { s_nbox[0] = bx0; s_nbox[1] = by0; s_nbox[2] = bx1; s_nbox[3] = by1; }
    }
    __syncthreads();
    CountTile tile;
    tile.cells = s_tile; tile.x0 = s_nbox[0]; tile.y0 = s_nbox[1];
    tile.w = s_nbox[2] >= s_nbox[0] ? s_nbox[2] - s_nbox[0] + 1 : 0;
    tile.h = s_nbox[3] >= s_nbox[1] ? s_nbox[3] - s_nbox[1] + 1 : 0;
    // 8 bits per count: RCN_RAYS * (1 + extra) visits of one cell at most
    if ((int64_t)tile.w * tile.h > RCN_TILE_CELLS || RCN_RAYS * (1 + g.extra) > 255) tile.w = tile.h = 0;       // direct atomics
    const int32_t tcells = tile.w * tile.h;
    for (int32_t i = threadIdx.x; i < (tcells + 1) / 2; i += blockDim.x) s_tile[i] = 0u;
    __syncthreads();
    GMS_STAMP(GMS_STAMP_ROW(2, blockIdx.x), 9);
    if (wave == 0) {
        RayWalk wk = ray_walk_begin(r);
        if (lane < RCN_RAYS && my_nwords > 0) ray_phase_a(wk, 0, my_nwords, s_slots, RCN_RAYS, lane);
    } else {
        if (wave == 1) {
            // the scan's dirty box from the WHOLE walks of this wedge's rays (the far-field workgroups of these rays leave it to this
            // one), by a consumer wavefront behind the barriers: see raycast_body
            int32_t hb[4] = { 0, 0, 0, 0 };
            if (lane < RCN_RAYS) ray_box(g, s_nmeta[lane], hb);
            bbox_raise_wave(hb, lane, bbox + 4 * mi);
        }
        for (int32_t ray = wave - 1; ray < RCN_RAYS; ray += nwaves - 1) {
            RayMeta mt = s_nmeta[ray];
            if (mt.n_eff <= 0) continue;
            mt.n_eff = min(mt.n_eff, 64);                      // block 0 only: the far-field workgroups count the rest
            ray_phase_b<false>(g, mt, s_slots, RCN_RAYS, ray, 0, lane, cnt + (size_t)mi * g.cells, bbox + 4 * mi, 0, nullptr, nullptr, 0, tile, 0);
        }
    }
    __syncthreads();
    GMS_STAMP(GMS_STAMP_ROW(2, blockIdx.x), 10);
    // flush: one atomic per touched cell, lanes on consecutive cells of a row
    uint32_t *mcnt = cnt + (size_t)mi * g.cells;
    for (int32_t ry = wave; ry < tile.h; ry += nwaves) {
        const int32_t cy = tile.y0 + ry, rbase = ry * tile.w;
        for (int32_t rx = lane; rx < tile.w; rx += 64) {
            const int32_t i = rbase + rx;
            const uint32_t v = (s_tile[i >> 1] >> ((i & 1) << 4)) & 0xffffu;
            if (v) atomicAdd(&mcnt[(size_t)cy * g.W + tile.x0 + rx], (v & 0xffu) | ((v >> 8) << 16));      // (inside the box raised above)
        }
    }
    GMS_STAMP(GMS_STAMP_ROW(2, blockIdx.x), 11);
}
// Whether a scan of B beams gets near-field workgroups, and how many.  They pay where many rays' atomics collide on the cells
// around the robot: 720 beams at 2 cm (C3) -1.3 us per step; at 360 beams (C2, the recording) the tile's clear / walk / flush chain
// ends 3 us after the far-field workgroups and costs the launch 0.8 us.  raycast_near: 0 never, 1 from GMS_RAYCAST_NEAR_MIN_BEAMS
// beams on, 2 always (GMS_RAYCAST_NEAR=1: tests/test_gpu_near_field.py runs every beam count through them).
#define GMS_RAYCAST_NEAR_MIN_BEAMS 512
static inline uint32_t rc_near_blocks(const gms_map *m, int32_t B) {
    if (!m->raycast_near || B < (m->raycast_near == 2 ? 32 : GMS_RAYCAST_NEAR_MIN_BEAMS)) return 0u;
    return (uint32_t)((B + RCN_RAYS - 1) / RCN_RAYS);
}

template <bool TRACE, int RC_RAYS>
__global__ void __launch_bounds__(RC_RAYS * 64)
k_raycast(GridDev g, const gms_beam *__restrict__ beams, int32_t B, int32_t beam_stride,
          const float *__restrict__ poses, int32_t pose_stride, const RayIn *__restrict__ single,
          uint32_t *__restrict__ cnt, int32_t *__restrict__ bbox, int32_t *__restrict__ t_cells,
          uint8_t *__restrict__ t_cls, int32_t cap, int32_t *__restrict__ t_counts, int32_t nw_max, uint32_t n_near_blocks) {
    extern __shared__ __align__(16) unsigned char smem[];
    // grid.x = ray blocks + near-field blocks (the latter only for the count form with four rays per workgroup)
    if (!TRACE && RC_RAYS == 4 && n_near_blocks && blockIdx.x >= gridDim.x - n_near_blocks) {
        raycast_near_body(g, beams, B, beam_stride, poses, pose_stride, cnt, bbox, blockIdx.x - (gridDim.x - n_near_blocks), blockIdx.y, smem,
                          nullptr);
        return;
    }
    raycast_body<TRACE, RC_RAYS>(g, beams, B, beam_stride, poses, pose_stride, single, cnt, bbox, t_cells, t_cls, cap, t_counts,
                                 nw_max, blockIdx.x, blockIdx.y, smem, nullptr, n_near_blocks ? 1 : 0);
}

// ---------------------------------------------------------------------------------------------
// Batched maps (config 5: tens of thousands of rays per launch): the scattered u32 atomics on the count grid are the
// floor of k_raycast there -- device-scope atomics execute at the memory side, and a wave-instruction whose 64 lanes
// land in 64 different 64-byte segments runs ~17x below the rate of 256 contiguous bytes (MI355X_MICROARCH.md).
// Here a workgroup takes 64 CONSECUTIVE beams of one map (a wedge of the scan: all rays leave the same cell), keeps
// the counts of the wedge's bounding box in an LDS tile (ds_add_u32), and flushes the tile row by row afterwards:
// one global atomic per touched cell and workgroup instead of one per visit, consecutive lanes on consecutive cells.
// Same counts in d_cnt as k_raycast (integer adds commute); a wedge whose box does not fit the tile, or a cell
// outside it, falls back to the direct atomics.
// ---------------------------------------------------------------------------------------------
#define RCT_RAYS 64
#define RCT_THREADS 1024                // 1 producer + 15 consumer wavefronts: phase B (~70 VALU per 64 cells) is the longer side here (256 threads: 125 us, 512: 73, 1024: 57 at C5)
#define RCT_WORDS 16                    // decision words per ray and round (512 steps); longer walks take more rounds
#define RCT_TILE_CELLS 32768            // 64 KiB of 16-bit cells
#define RCT_LDS_BYTES (RCT_WORDS * RCT_RAYS * 8 + RCT_TILE_CELLS * 2)

__device__ void apply_slices(const GridDev &g, double *__restrict__ logd, uint32_t *__restrict__ cnt, const int32_t *__restrict__ bbox,
                             uint32_t first_block);

__global__ void __launch_bounds__(RCT_THREADS)
k_raycast_tile(GridDev g, const gms_beam *__restrict__ beams, int32_t B, int32_t beam_stride, const float *__restrict__ poses,
               int32_t pose_stride, uint32_t *__restrict__ cnt, int32_t *__restrict__ bbox, uint32_t n_ray_blocks,
               double *__restrict__ logd, uint32_t *__restrict__ cnt_pend, const int32_t *__restrict__ bbox_pend) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (blockIdx.x >= n_ray_blocks) {
        // the PREVIOUS scan's deferred `logData[c] += ...` (GridMap.java:223), from the other count grid, as four 256-thread
        // slices per workgroup: bandwidth-bound work beside this kernel's LDS-bound work (defined below this kernel)
        apply_slices(g, logd, cnt_pend, bbox_pend, n_ray_blocks);
        return;
    }
    uint64_t *s_slots = reinterpret_cast<uint64_t *>(smem);            // [RCT_WORDS][RCT_RAYS]
    uint32_t *s_tile = reinterpret_cast<uint32_t *>(s_slots + RCT_WORDS * RCT_RAYS);         // [RCT_TILE_CELLS / 2]
    __shared__ RayMeta s_meta[RCT_RAYS];
    __shared__ int32_t s_box[5];                                        // wedge box x0, y0, x1, y1 (inclusive); longest walk in words

    const int32_t mi = blockIdx.y;
    const int32_t lane = threadIdx.x & 63;
    const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    RayDev r;
    r.dx = r.dy = r.error = 0.0f; r.x = r.y = r.x_inc = r.y_inc = r.n = 0;
    int32_t my_nwords = 0;
    if (wave == 0) {
        const int32_t b = (int32_t)blockIdx.x * RCT_RAYS + lane;
        RayMeta mt;
        mt.n_eff = 0; mt.x0 = mt.y0 = mt.x_inc = mt.y_inc = mt.hit = 0; mt.sx = mt.sy = mt.measured = 0.0f;
        int32_t bx0 = INT32_MAX, by0 = INT32_MAX, bx1 = INT32_MIN, by1 = INT32_MIN;
        if (b < B) {
            const RayIn ray = make_ray(g, beams[(size_t)mi * beam_stride + b], poses + (size_t)pose_stride * mi);
            mt = ray_meta(g, ray, r);
            if (mt.n_eff > 0) {
                // The walk is monotonic in x and in y and ends n - 1 steps from the start cell; the end point's cell plus
                // the additional steps (RayIterator.java:75,83,96; one more for a near-tie decided by float rounding)
                // bounds it tighter.  A cell outside the box would take the direct-atomic path: the box is a hint.
                const int32_t xe = mt.x0 + mt.x_inc * (mt.n_eff - 1), ye = mt.y0 + mt.y_inc * (mt.n_eff - 1);
                const int32_t xt = (int32_t)floorf(ray.ex + 0.5f), yt = (int32_t)floorf(ray.ey + 0.5f);
                const int32_t ex = mt.x_inc > 0 ? min(xe, xt + g.extra + 1) : (mt.x_inc < 0 ? max(xe, xt - g.extra - 1) : mt.x0);
                const int32_t ey = mt.y_inc > 0 ? min(ye, yt + g.extra + 1) : (mt.y_inc < 0 ? max(ye, yt - g.extra - 1) : mt.y0);
                bx0 = max(min(mt.x0, ex), 0); bx1 = min(max(mt.x0, ex), g.W - 1);
                by0 = max(min(mt.y0, ey), 0); by1 = min(max(mt.y0, ey), g.H - 1);
            }
        }
        my_nwords = (mt.n_eff + 31) >> 5;
        s_meta[lane] = mt;
        int32_t nwm = my_nwords;
#define GMS_STEP_(O) { bx0 = min(bx0, wave_xor<O>(bx0)); by0 = min(by0, wave_xor<O>(by0)); bx1 = max(bx1, wave_xor<O>(bx1)); by1 = max(by1, wave_xor<O>(by1)); \
                       nwm = max(nwm, wave_xor<O>(nwm)); }
        GMS_BUTTERFLY(GMS_STEP_)
#undef GMS_STEP_
        if (lane == 0) { s_box[0] = bx0; s_box[1] = by0; s_box[2] = bx1; s_box[3] = by1; s_box[4] = nwm; }
    }
    __syncthreads();
    CountTile tile;
    tile.cells = s_tile; tile.x0 = s_box[0]; tile.y0 = s_box[1];
    tile.w = s_box[2] >= s_box[0] ? s_box[2] - s_box[0] + 1 : 0;
    tile.h = s_box[3] >= s_box[1] ? s_box[3] - s_box[1] + 1 : 0;
    // 8 bits per count: RCT_RAYS * (1 + extra) visits of one cell at most
    if ((int64_t)tile.w * tile.h > RCT_TILE_CELLS || RCT_RAYS * (1 + g.extra) > 255) tile.w = tile.h = 0;     // direct atomics
    const int32_t tcells = tile.w * tile.h;
    const int32_t nwords_max = s_box[4];
    for (int32_t i = threadIdx.x; i < (tcells + 1) / 2; i += RCT_THREADS) s_tile[i] = 0u;
    RayWalk wk = ray_walk_begin(r);
    for (int32_t wb = 0; wb < nwords_max; wb += RCT_WORDS) {           // one round = up to 512 steps of every ray
        for (int32_t i = threadIdx.x; i < RCT_WORDS * RCT_RAYS; i += RCT_THREADS) s_slots[i] = 0ull;
        __syncthreads();                                                // (also: the tile is cleared, the previous round consumed)
        if (wave == 0) {
            if (wb < my_nwords) ray_phase_a(wk, wb, min(my_nwords, wb + RCT_WORDS), s_slots, RCT_RAYS, lane);
        } else {
            constexpr int32_t NC = RCT_THREADS / 64 - 1;
            const int32_t blk0 = wb >> 1, nblk = min(RCT_WORDS / 2, (nwords_max - wb + 1) >> 1);
            for (int32_t q = wave - 1; q < nblk * RCT_RAYS; q += NC) {
                const int32_t blk = blk0 + q / RCT_RAYS, ray = q % RCT_RAYS;
                const RayMeta mt = s_meta[ray];
                if (blk * 64 >= mt.n_eff) continue;
                ray_phase_b<false>(g, mt, s_slots, RCT_RAYS, ray, blk, lane, cnt + (size_t)mi * g.cells, bbox + 4 * mi, 0, nullptr, nullptr, 0,
                                   tile, wb);
            }
        }
        __syncthreads();
    }
    if (nwords_max == 0) __syncthreads();
    if (wave == 1) {
        // the map's dirty box from the rays' boxes (ray_meta): by a consumer wavefront at the very end, where nobody is held up by
        // atomics that every workgroup of the map sends to the same four words (see raycast_body; C5: 65.0 -> 62.3 us)
        int32_t hb[4] = { 0, 0, 0, 0 };
        if (lane < RCT_RAYS) ray_box(g, s_meta[lane], hb);
        bbox_raise_wave(hb, lane, bbox + 4 * mi);
    }
    // flush: one atomic per touched cell, lanes on consecutive cells of a row
    uint32_t *mcnt = cnt + (size_t)mi * g.cells;
    for (int32_t ry = wave; ry < tile.h; ry += RCT_THREADS / 64) {     // a wavefront per tile row: no division per cell
        const int32_t cy = tile.y0 + ry, rbase = ry * tile.w;
        for (int32_t rx = lane; rx < tile.w; rx += 64) {
            const int32_t i = rbase + rx;
            const uint32_t v = (s_tile[i >> 1] >> ((i & 1) << 4)) & 0xffffu;
            if (v) atomicAdd(&mcnt[(size_t)cy * g.W + tile.x0 + rx], (v & 0xffu) | ((v >> 8) << 16));      // (inside the box raised above)
        }
    }
}

// plain RayIterator walk (gms_map_trace_ray)
__global__ void k_trace_ray(int32_t W, int32_t H, float x0, float y0, float x1, float y1, int32_t extra,
                            int32_t *__restrict__ cells, int32_t cap, int32_t *__restrict__ count_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    RayDev r;
    ray_init(r, x0, y0, x1, y1, extra);
    int32_t count = 0;
    while (ray_has_next(r, W, H)) {
        if (count < cap) { cells[2 * count] = r.x; cells[2 * count + 1] = r.y; }
        ray_step(r);
        count++;
    }
    *count_out = count;
}

// log += n_free*l_free + n_occ*l_occ on the touched box; counts cleared.  Persistent workgroups
// enumerate only the 256 x 4 cell tiles that intersect the box; a lane owns 4 consecutive cells.
#ifndef GMS_APPLY_BLOCKS
#define GMS_APPLY_BLOCKS 2048   // persistent workgroups of the apply pass (per map)
#endif
#define APPLY_TW 256
#define APPLY_TH 4
__device__ __forceinline__ void
apply_body(const GridDev &g, double *__restrict__ logd, uint32_t *__restrict__ cnt, const int32_t *__restrict__ bbox,
           int32_t *__restrict__ bbox_idle, uint32_t bx, uint32_t by, uint32_t gdx, uint32_t tid = threadIdx.x) {
    // (bx, gdx, tid) name a 256-thread slice: the workgroup itself, or a quarter of a 1024-thread one (k_raycast_tile).  No barriers here.
    const int32_t mi = (int32_t)by;
    if (bbox_idle && bx == 0 && tid < 4) bbox_idle[4 * mi + tid] = 0;     // (nullptr: the other half is in use, see k_raycast_apply)
    int32_t x0, y0, x1, y1;
    bbox_decode(bbox + 4 * mi, g.W, g.H, x0, y0, x1, y1);
    if (x1 <= 0) return;
    const int32_t qx0 = x0 / APPLY_TW, qy0 = y0 / APPLY_TH;
    const int32_t qnx = (x1 - 1) / APPLY_TW - qx0 + 1, qny = (y1 - 1) / APPLY_TH - qy0 + 1;
    const bool vec = (g.W & 3) == 0;
    const bool lazy_log = gridDim.y > 1;                 // batched maps (uniform)
    for (int32_t t = (int32_t)bx; t < qnx * qny; t += (int32_t)gdx) {
        const int32_t tx0 = (qx0 + t % qnx) * APPLY_TW, ty0 = (qy0 + t / qnx) * APPLY_TH;
        const int32_t y = ty0 + (int32_t)(tid >> 6);
        const int32_t xb = tx0 + (int32_t)(tid & 63u) * 4;
        if (y >= g.H || xb >= g.W) continue;
        const size_t o = (size_t)mi * g.cells + (size_t)y * g.W + xb;
        if (vec) {                                       // rows are 16-byte aligned: one 16-byte count load
            // the log-odds travel with the counts (two round trips per tile instead of three; untouched cells
            // are read for nothing, which costs bandwidth this kernel does not use)
            // Batched handles (many maps' boxes: ~200 MB at config 5) are bandwidth-bound instead: there the log-odds of
            // an untouched quad are not read (the extra dependent round trip is hidden by the other tiles in flight).
            const uint4 c = *reinterpret_cast<const uint4 *>(cnt + o);
            double2 la, lb;
            if (!lazy_log) { la = *reinterpret_cast<const double2 *>(logd + o); lb = *reinterpret_cast<const double2 *>(logd + o + 2); }
            if ((c.x | c.y | c.z | c.w) == 0u) continue;
            if (lazy_log) { la = *reinterpret_cast<const double2 *>(logd + o); lb = *reinterpret_cast<const double2 *>(logd + o + 2); }
            const uint32_t cc[4] = { c.x, c.y, c.z, c.w };
            const double lv[4] = { la.x, la.y, lb.x, lb.y };
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (cc[i]) {
                    const double nv = lv[i] + ((double)(cc[i] & 0xffffu) * g.l_free + (double)(cc[i] >> 16) * g.l_occ);
                    // Single maps: written through to memory at once.  The pass hides under the ray cast it shares a launch with, and
                    // ~7 MB of dirty lines left in the L2s would be written back at the kernel's END, in front of the next launch
                    // (C3 step 49.5 -> 48.2 us).  Batched maps are bandwidth-bound here and keep the L2's write combining (C5: the
                    // tiled ray cast 65 -> 73 us with write-through).
                    if (lazy_log) logd[o + i] = nv; else store_through(&logd[o + i], nv);
                }
            if (lazy_log) {
                *reinterpret_cast<uint4 *>(cnt + o) = make_uint4(0u, 0u, 0u, 0u);
            } else {
                store_through(reinterpret_cast<uint64_t *>(cnt + o), (uint64_t)0);
                store_through(reinterpret_cast<uint64_t *>(cnt + o + 2), (uint64_t)0);
            }
        } else {
            for (int i = 0; i < 4 && xb + i < g.W; i++) {
                const uint32_t c = cnt[o + i];
                if (c) {
                    logd[o + i] = logd[o + i] + ((double)(c & 0xffffu) * g.l_free + (double)(c >> 16) * g.l_occ);
                    cnt[o + i] = 0u;
                }
            }
        }
    }
}

__device__ void apply_slices(const GridDev &g, double *__restrict__ logd, uint32_t *__restrict__ cnt, const int32_t *__restrict__ bbox,
                             uint32_t first_block) {
    const uint32_t q = blockDim.x >> 8;
    apply_body(g, logd, cnt, bbox, nullptr, (blockIdx.x - first_block) * q + (threadIdx.x >> 8), blockIdx.y, (gridDim.x - first_block) * q,
               threadIdx.x & 255u);
}

__global__ void __launch_bounds__(256)
k_apply(GridDev g, double *__restrict__ logd, uint32_t *__restrict__ cnt, const int32_t *__restrict__ bbox,
        int32_t *__restrict__ bbox_idle) {
    apply_body(g, logd, cnt, bbox, bbox_idle, blockIdx.x, blockIdx.y, gridDim.x);
}

// ---------------------------------------------------------------------------------------------
// Likelihood field (GridMap.computeLikelihoodMap, GridMap.java:233-250 + Util.doGaussianBlurdSeparable,
// Util.java:378-426).  Persistent workgroups walk LK_TW x LK_TH output tiles:
//   phase 1  threshold the tile + k-halo into LDS as doubles {0, 0.5, 1} (GridMap.java:239-244).
//            Cells outside the map are staged as 0.0: the reference skips those taps
//            (Util.java:396,418), and `total + tap*0.0` leaves a non-negative total unchanged, so the
//            sums stay bit-identical without a branch per tap.
//   uniform  if every staged cell holds the same value c and the halo is inside the map, every output
//            of the tile is the same constant (the in-order tap sums of c): written without blurring.
//            Free space and unexplored space -- most of a map -- take this path.
//   phase 2  horizontal sums, 8 outputs per thread from a register window, taps in the reference's
//            order, multiply then add (no FMA)                                  (Util.java:387-404)
//   phase 3  vertical sums the same way, coalesced store of likelihoodData      (Util.java:410-425)
// dirty_only: tiles that cannot have changed (outside the touched box dilated by k) are skipped.
// ---------------------------------------------------------------------------------------------
#define LK_TW 64
// registers: five workgroups of four wavefronts per CU = five wavefronts per SIMD = at most 96 vector registers (the variant that
// also carries a scan's pending counts would spill at that: four)
#ifndef GMS_LIK_WAVES_PER_EU
#define GMS_LIK_WAVES_PER_EU 5
#endif
#define GMS_LIK_WAVES __attribute__((amdgpu_waves_per_eu(PENDING && KH == 5 ? 4 : GMS_LIK_WAVES_PER_EU)))
#define LK_TH 32
#define LK_STRIP 8

// the factor table's stores go through to memory at once: nobody in the launch reads them again, and dirty lines would be written
// back at the kernel's end, in front of the next scan's scoring launch (C3 step -0.4 us, C5 -5 us)
#define FAC_STORE(p, v) store_through((p), (v))
// ... and so do likelihoodData's: a full rebuild of a 2048 x 2048 map leaves 33 MB of it, more than the L2s hold, and what is still dirty
// at the kernel's end is written back there with nothing to hide behind (dense map: 24.6 -> 21.8 us per rebuild, bracketed)
#ifndef GMS_LIK_STORE_PLAIN
#define LIK_STORE(p, v) store_through((p), (v))
#else
#define LIK_STORE(p, v) (*(p) = (v))
#endif
// A cell's class from a packed 2-bit plane (the per-particle maps keep one beside logData, gms_slam_kernels.hip): 0 logData == 0 (or
// NaN), 1 logData < 0, 2 logData > 0 -- returned as a stand-in log-odds value of that sign, which is all computeLikelihoodMap reads
// of logData (GridMap.java:239-244).
__device__ __forceinline__ double plane_log_sign(const uint32_t *__restrict__ plane, size_t cell) {
    const uint32_t e = (plane[cell >> 4] >> (2u * ((uint32_t)cell & 15u))) & 3u;
    return e == 2u ? 1.0 : (e == 1u ? -1.0 : 0.0);
}
template <int KH, int SPLIT = 1, bool CODES = false>   // KH > 0: compile-time half width; KH == 0: runtime g.khalf (generic, slower).  SPLIT = 2 (KH > 0, dirty
                                  // tiles only): a tile's horizontal and vertical sums are shared by TWO workgroups, see "split" below.
                                  // CODES: `logd` is a packed 2-bit class plane per map (code_stride 32-bit words apart) instead of logData
__device__ __forceinline__ void
likelihood_body(const GridDev &g, const double *__restrict__ logd, double *__restrict__ lik, double *__restrict__ fac,
                int64_t fac_stride, const double *__restrict__ taps_g, const int32_t *__restrict__ bbox, int32_t dirty_only,
                int32_t tiles_x, int32_t tiles_y, uint32_t bx, uint32_t by, uint32_t gdx, unsigned char *smem,
                const uint32_t *__restrict__ cnt_pending = nullptr, uint8_t *__restrict__ tile_state = nullptr, int32_t mode = 3,
                int64_t code_stride = 0) {
    // mode: bit 0 = write likelihoodData, bit 1 = write the factor table (and keep tile_state, which describes the factor table).
    // The scan steps' dirty-tile rebuilds write the factor table only (mode 2): nothing on the hot path reads likelihoodData
    // (GridMap.java:150-156,371-388 are its only readers: getLikelihood and the renderer), so it is brought up to date on demand by a
    // pass of mode 1 (gms_ensure_lik) -- half the stores of a rebuild, and half the dirty lines at the kernel's end.
#if defined(GMS_LIK_EXP) && GMS_LIK_EXP == 4       // experiment: a likelihood pass that costs nothing (what the paired launch's other half then takes)
    if (dirty_only) return;
#endif
    const bool wr_lik = (mode & 1) != 0, wr_fac = (mode & 2) != 0;
    // bit 2: a tile none of whose staged cells changes its code under this scan's counts is left alone.  Its stored field is the
    // blur of exactly these codes already -- the launcher sets the bit only when the factor table is current (every change of
    // logData since the last rebuild went through that rebuild as pending counts: gms_map::fac_current) -- so only the staging and
    // the comparison are spent on it.  A map that is explored changes codes along its frontier; a map that is revisited hardly at all.
    const bool skip_unchanged = (mode & 4) != 0 && dirty_only && cnt_pending != nullptr;
    const int32_t k = KH > 0 ? KH : g.khalf;
    const int32_t ntaps = 2 * k + 1;
    const int32_t RW = LK_TW + 2 * k, RH = LK_TH + 2 * k;     // staged columns / rows
    // LDS: the thresholded cells, then the horizontal sums.  KH > 0 stages the cells as their CODES, one byte each (0, 1, 2
    // for 0, 0.5, 1; outside the map 0: Util.java:396 skips those taps, `total + tap * 0.0` leaves a non-negative total unchanged)
    // and widens them when the horizontal pass reads them: 25 KiB per workgroup instead of 47 (KH = 5), i.e. twice the
    // workgroups per CU, and this kernel lives on resident workgroups (likelihood_lds_bytes).  The generic path keeps doubles.
    const int32_t PIN = RW + 1, PHS = LK_TW + 1;              // LDS pitches (doubles)
    constexpr int32_t PINB = (LK_TW + 2 * (KH > 0 ? KH : 1) + 7) & ~7;      // byte pitch of the code rows: a multiple of 8
    double *in_s = reinterpret_cast<double *>(smem);          // [RH][PIN] (generic path)
    uint8_t *in_b = reinterpret_cast<uint8_t *>(smem);        // [RH][PINB] (KH > 0)
    double *hs = KH > 0 ? reinterpret_cast<double *>(smem + (((size_t)RH * PINB + 15) & ~(size_t)15)) : in_s + (size_t)RH * PIN;   // [RH][PHS]
    double *taps_s = hs + (size_t)RH * PHS;                   // [ntaps] (generic path)
    __shared__ int32_t s_mask[3];
    int32_t tile_iter = 0;
    // measurement (gms_map_tile_stats): what became of the tiles this workgroup walked, kept in thread 0's registers and added to
    // one of 64 counter rows at the end (no store, no atomic per tile)
    uint32_t ts_left = 0, ts_const_kept = 0, ts_const_written = 0, ts_blurred = 0;

    const int32_t mi = (int32_t)by;
    const double *mlog = CODES ? logd : logd + (size_t)mi * g.cells;
    const uint32_t *mcode = reinterpret_cast<const uint32_t *>(logd) + (CODES ? (size_t)mi * (size_t)code_stride : 0);
    // cnt_pending: the scan's counts have not been added to logData yet (the apply pass runs later, beside another
    // kernel); a staged cell is logData + its increment, the expression of apply_body (GridMap.java:223), not stored
    const uint32_t *mcnt = cnt_pending ? cnt_pending + (size_t)mi * g.cells : nullptr;
    double *mlik = lik + (size_t)mi * g.cells;
    double *mfac = fac + (size_t)mi * fac_stride;
    // tile rectangle to process: the whole map, or the tiles that intersect the touched box dilated
    // by k (enumerated directly, so the persistent workgroups share the dirty tiles evenly)
    int32_t qx0 = 0, qy0 = 0, qnx = tiles_x, qny = tiles_y;
    if (dirty_only) {
        int32_t bx0, by0, bx1, by1;
        bbox_decode(bbox + 4 * mi, g.W, g.H, bx0, by0, bx1, by1);
        if (bx1 <= 0) return;
        const int32_t x_lo = max(bx0 - k, 0), y_lo = max(by0 - k, 0);
        const int32_t x_hi = min(bx1 + k, g.W) - 1, y_hi = min(by1 + k, g.H) - 1;     // inclusive
        qx0 = x_lo / LK_TW; qy0 = y_lo / LK_TH;
        qnx = x_hi / LK_TW - qx0 + 1; qny = y_hi / LK_TH - qy0 + 1;
    }
    const int32_t ntiles = qnx * qny;
    if (KH == 0)
        for (int32_t i = threadIdx.x; i < ntaps; i += blockDim.x) taps_s[i] = taps_g[i];

    // XCD-aware order (full rebuild): workgroups b and b+8 share an XCD, so each XCD walks a
    // contiguous band of tiles and the halo re-reads hit its own L2
    auto tile_of = [&](int32_t t) { return (!dirty_only && (ntiles & 7) == 0) ? (t & 7) * (ntiles >> 3) + (t >> 3) : t; };
    // The staged rectangle of a tile: every load is issued before the first one is consumed (addresses clamped into
    // the map, so no load sits behind a branch: 13 dependent round trips otherwise), and the NEXT tile's loads are
    // issued as soon as this tile's values are in LDS, so they fly during the barriers and the two blur passes.
    constexpr int32_t CRW = LK_TW + 2 * (KH > 0 ? KH : 1);
    // Which staged cells a thread takes.  Elements 0 .. QM-1: the rectangle's first 64 columns, lane = column, row = wavefront + 4 q:
    // the row is the same for a whole wavefront, so its clamp, its bounds test and its base address are scalar work and a load is
    // `row base (scalar) + column offset (one register for all rows)`: no vector instruction per load.  Elements QM .. P1-1: the
    // SW = 2 KH columns on the right, (row, column) per lane, computed once per workgroup.  (Rounds 1-4 numbered the rectangle's
    // cells row-major through the workgroup: a division, two clamps, a 64-bit multiply-add and four bounds tests under a branch per
    // cell were 2300 of this kernel's 4800 vector instructions per tile: profiles/r05/dense_likelihood_counters.json.)
    // (SPLIT = 2: an item is half a tile's rows and stages the CRS = LK_TH / 2 + 2 KH rows its own sums need, from row R0 = 16 part of
    // the tile's rectangle on: a 64 x 16 tile with its own class decision and its own tile state)
    constexpr int32_t CRS = LK_TH / SPLIT + 2 * (KH > 0 ? KH : 1);
    constexpr int32_t SW = CRW - LK_TW, QM = (CRS + 3) / 4, QS = (CRS * SW + 255) / 256;
    constexpr int32_t P1 = QM + QS;
    static_assert(LK_TW == 64, "lane = column of the staged rectangle's first 64");
    const int32_t e_lane = (int32_t)(threadIdx.x & 63);
    const int32_t e_wave = __builtin_amdgcn_readfirstlane((int32_t)(threadIdx.x >> 6));
    int32_t s_r[QS], s_c[QS];
    bool s_have[QS];
#pragma unroll
    for (int j = 0; j < QS; j++) {
        const int32_t i2 = (int32_t)threadIdx.x + j * 256;
        s_r[j] = i2 / SW; s_c[j] = LK_TW + (i2 - s_r[j] * SW);
        s_have[j] = i2 < CRS * SW;
        if (!s_have[j]) { s_r[j] = 0; s_c[j] = 0; }         // (loads a duplicate of the rectangle's first cell; not staged)
    }
    double lv[P1];
    uint32_t cv[P1];
    auto issue_loads = [&](int32_t t, int32_t lr0) {          // tile index, first staged row of the tile's rectangle (0, or 16 part)
        const int32_t tile = tile_of(t);
        const int32_t ltx0 = (qx0 + tile % qnx) * LK_TW, lty0 = (qy0 + tile / qnx) * LK_TH + lr0;
        // addresses clamped into the map (a cell outside it is given its code from the coordinates, below): no load behind a branch
        const uint32_t gxm = (uint32_t)min(max(ltx0 - KH + e_lane, 0), g.W - 1);
#pragma unroll
        for (int q = 0; q < QM; q++) {
            const int32_t gy = min(max(lty0 - KH + e_wave + 4 * q, 0), g.H - 1);
            const size_t row = (size_t)gy * (size_t)g.W;
            lv[q] = CODES ? plane_log_sign(mcode, row + gxm) : (mlog + row)[gxm];
            cv[q] = mcnt ? (mcnt + row)[gxm] : 0u;
        }
#pragma unroll
        for (int j = 0; j < QS; j++) {
            const int32_t gy = min(max(lty0 - KH + s_r[j], 0), g.H - 1), gx = min(max(ltx0 - KH + s_c[j], 0), g.W - 1);
            lv[QM + j] = CODES ? plane_log_sign(mcode, (size_t)gy * g.W + gx) : mlog[(size_t)gy * g.W + gx];
            cv[QM + j] = mcnt ? mcnt[(size_t)gy * g.W + gx] : 0u;
        }
    };
    // work items (see "split" at the tile loop): item w -> tile item_tile(w), part (w >> 3) & 1
    const int32_t nitems = SPLIT == 1 ? ntiles : ((ntiles + 7) >> 3) << 4;
    auto item_tile = [&](int32_t w) { return SPLIT == 1 ? w : ((w >> 4) << 3) + (w & 7); };
    auto item_row0 = [&](int32_t w) { return SPLIT == 1 ? 0 : ((w >> 3) & 1) * (LK_TH / SPLIT); };
    auto next_item = [&](int32_t w) {                     // the next item from w on that names a tile (the last group of eight may not be full)
        if (SPLIT != 1)
            while (w < nitems && item_tile(w) >= ntiles) w += (int32_t)gdx;
        return w;
    };
    const int32_t first_w = next_item((int32_t)bx);
    if (KH > 0) {
        if (threadIdx.x < 3) s_mask[threadIdx.x] = 0;
        __syncthreads();
        if (first_w < nitems) issue_loads(item_tile(first_w), item_row0(first_w));
    }

    // split (SPLIT = 2): a scan step's dirty box is a few hundred tiles of which a few dozen are blurred, on a chip that holds 1280
    // workgroups: the blurred ones decide when the launch ends (8.7 us each at C3: 4.7 staging, 2.2 + 1.7 for the two passes).  Work item
    // w = (tile, part) is a 64 x 16 tile of its own: it stages and classifies the 16 + 2 KH rows of the tile's rectangle from row 16 part on
    // (its own uniform / unchanged / blur decision, its own tile state), takes their horizontal sums and the vertical sums and stores of its
    // 16 rows.  Of sixteen consecutive items the first eight are part 0 of eight tiles and the last eight part 1 of the same: one XCD.
    static_assert(SPLIT == 1 || (SPLIT == 2 && KH > 0), "split tiles: compile-time half widths only");
    int32_t w_next = nitems;
    for (int32_t w = first_w; w < nitems; w = w_next) {
        w_next = next_item(w + (int32_t)gdx);
        const int32_t t = item_tile(w);
        const int32_t part = SPLIT == 1 ? 0 : (w >> 3) & 1;
        const int32_t R0 = part * (LK_TH / SPLIT);                                      // the item's first row of the tile (and of its rectangle)
        const int32_t tile = tile_of(t);
        const int32_t tx0 = (qx0 + tile % qnx) * LK_TW, ty0 = (qy0 + tile / qnx) * LK_TH;
        // what this tile of likelihoodData / the factor table holds: 0 unknown, 1..3 the constants of a uniform tile
        // (two entries per tile, one per half of its rows: a split tile's two workgroups each keep their own -- one of them must not find
        // the other's fresh entry and skip its own half's stores --, an unsplit tile's workgroup keeps both alike)
        uint8_t *tstate = tile_state ? tile_state + 2 * ((size_t)mi * tiles_x * tiles_y + (size_t)(qy0 + tile / qnx) * tiles_x + (qx0 + tile % qnx)) +
                                           (SPLIT == 1 ? 0 : part)
                                     : nullptr;
        // Read by EVERY thread here, before this tile's first barrier: thread 0 rewrites the state after that barrier, and a
        // wavefront that read it later could see the new value and skip its share of a uniform tile's stores (round 1 read
        // it at the point of use: one wavefront's 8 rows of a far-away tile were left stale about once in forty full-size
        // multi-map runs; found by tests/test_gpu_configs.py).
        uint8_t tstate_old = tstate ? *reinterpret_cast<volatile uint8_t *>(tstate) : (uint8_t)0;
        if (SPLIT == 1 && tstate && *reinterpret_cast<volatile uint8_t *>(tstate + 1) != tstate_old) tstate_old = 0;

        // ---- phase 1
        int32_t seen = 0;                                      // bit c: a cell of code c; bit 3: outside the map; bit 4: a code changes
        uint32_t codes = 0;                                    // KH > 0: 2 bits per staged cell of this thread
        bool a_chg = false;
        int32_t *smask = &s_mask[0];
        if (KH > 0) {
            // The codes come out of registers; LDS is written only if the tile turns out not to be uniform (most of a
            // map is), and the mask word rotates through three slots so that a uniform tile costs one barrier.
            smask = &s_mask[tile_iter % 3];
#pragma unroll
            for (int q = 0; q < P1; q++)
                if (cv[q]) {
                    const double nv = lv[q] + ((double)(cv[q] & 0xffffu) * g.l_free + (double)(cv[q] >> 16) * g.l_occ);
                    // (a clamped duplicate of a border cell counts too: conservative)
                    if (skip_unchanged && ((nv > 0.0) != (lv[q] > 0.0) || (nv < 0.0) != (lv[q] < 0.0))) a_chg = true;
                    lv[q] = nv;
                }
            // The class of every staged cell, two bits each in `codes` (3 = outside the map); which classes occur at all is read off
            // the packed word afterwards -- a handful of instructions per thread instead of per cell -- and "does any lane of the
            // wavefront hold one" is a scalar test of a comparison's lane mask: no butterfly.
            auto classify = [&](int q, bool in) {
                const double v = lv[q];
                const uint32_t code = v > 0.0 ? 2u : (v < 0.0 ? 0u : 1u);               // GridMap.java:239-244
                codes |= (in ? code : 3u) << (2 * q);
            };
            const bool x_in = (uint32_t)(tx0 - KH + e_lane) < (uint32_t)g.W;
            uint32_t have_bits = 0;                                                     // bit 2q: element q is a cell of the rectangle
#pragma unroll
            for (int q = 0; q < QM; q++) {
                const int32_t r = e_wave + 4 * q;                                       // (a row of the item's CRS staged rows)
                classify(q, x_in && (uint32_t)(ty0 + R0 - KH + r) < (uint32_t)g.H);
                if (r < CRS) have_bits |= 1u << (2 * q);
            }
#pragma unroll
            for (int j = 0; j < QS; j++) {
                classify(QM + j, (uint32_t)(tx0 - KH + s_c[j]) < (uint32_t)g.W && (uint32_t)(ty0 + R0 - KH + s_r[j]) < (uint32_t)g.H);
                if (s_have[j]) have_bits |= 1u << (2 * (QM + j));
            }
            {
                const uint32_t lo = codes, hi = codes >> 1;
                const uint32_t is3 = lo & hi & have_bits, is2 = hi & ~lo & have_bits, is1 = lo & ~hi & have_bits, is0 = ~(lo | hi) & have_bits;
                seen = (__builtin_amdgcn_ballot_w64(is0 != 0u) ? 1 : 0) | (__builtin_amdgcn_ballot_w64(is1 != 0u) ? 2 : 0) |
                       (__builtin_amdgcn_ballot_w64(is2 != 0u) ? 4 : 0) | (__builtin_amdgcn_ballot_w64(is3 != 0u) ? 8 : 0) |
                       (__builtin_amdgcn_ballot_w64(a_chg) ? 16 : 0);
                codes &= ~(is3 | (is3 << 1));                                           // what LDS holds: outside the map 0
            }
        } else {
            __syncthreads();                                   // previous tile's LDS reads are done
            if (threadIdx.x == 0) *smask = 0;
            __syncthreads();
#pragma unroll 4
            for (int32_t idx = threadIdx.x; idx < RH * RW; idx += blockDim.x) {
                const int32_t r = idx / RW, c = idx - r * RW;
                const int32_t gy = ty0 - k + r, gx = tx0 - k + c;
                double val = 0.0;
                if (gx >= 0 && gx < g.W && gy >= 0 && gy < g.H) {
                    double v = CODES ? plane_log_sign(mcode, (size_t)gy * g.W + gx) : mlog[(size_t)gy * g.W + gx];
                    const uint32_t cc = mcnt ? mcnt[(size_t)gy * g.W + gx] : 0u;
                    if (cc) v = v + ((double)(cc & 0xffffu) * g.l_free + (double)(cc >> 16) * g.l_occ);
                    const int32_t code = v > 0.0 ? 2 : (v < 0.0 ? 0 : 1);             // GridMap.java:239-244
                    val = 0.5 * (double)code;
                    seen |= 1 << code;
                } else {
                    seen |= 8;
                }
                in_s[r * PIN + c] = val;
            }
        }
        if (KH == 0) {
#define GMS_STEP_(O) seen |= wave_xor<O>(seen);
            GMS_BUTTERFLY(GMS_STEP_)
#undef GMS_STEP_
        }
        if ((threadIdx.x & 63) == 0) atomicOr(smask, seen);
        __syncthreads();
        if (tile_iter == 0) GMS_STAMP(GMS_STAMP_ROW(3, blockIdx.x), 3);      // first tile: loads arrived, codes known
        const int32_t mask = *smask;
        if (KH > 0) {
            // Re-arm the slot of the PREVIOUS tile (every thread read it before arriving at this barrier); it is used
            // again two tiles from now, after the next barrier.  (Re-arming the next tile's slot here would race with
            // a wavefront that is already OR-ing into it.)
            if (threadIdx.x == 0) s_mask[(tile_iter + 2) % 3] = 0;
            tile_iter++;
            const int32_t mk = mask & 15;
            const bool uniform = mk == 1 || mk == 2 || mk == 4, unchanged = skip_unchanged && (mask & 16) == 0;
            if (!uniform && !unchanged) {
                // stage {0, 0.5, 1} (outside the map: 0.0).  Every thread passed the barrier above, so the previous
                // tile's reads of in_s and hs are over.
                uint8_t *wb = in_b + (R0 + e_wave) * PINB + e_lane;
#pragma unroll
                for (int q = 0; q < QM; q++)
                    if (e_wave + 4 * q < CRS) wb[q * 4 * PINB] = (uint8_t)((codes >> (2 * q)) & 3u);
#pragma unroll
                for (int j = 0; j < QS; j++)
                    if (s_have[j]) in_b[(R0 + s_r[j]) * PINB + s_c[j]] = (uint8_t)((codes >> (2 * (QM + j))) & 3u);
            }
            if (w_next < nitems) issue_loads(item_tile(w_next), item_row0(w_next));      // in flight during the rest of this tile
            if (unchanged) { if (part == 0) ts_left++; continue; }             // (one barrier, like a uniform tile)
            if (!uniform) __syncthreads();
        }

        const int32_t mk = mask & 15;
        if (mk == 1 || mk == 2 || mk == 4) {
            // ---- uniform tile: every in-order sum sees the same inputs
            const uint8_t want = mk == 1 ? 1 : (mk == 2 ? 2 : 3);
            if (wr_fac && tstate && tstate_old == want) { if (part == 0) ts_const_kept++; continue; }    // the tile already holds exactly these constants: no store (mode 3: both
                                                                    // arrays do -- a full rebuild invalidates the states first when likelihoodData is behind)
            const double cval = mk == 1 ? 0.0 : (mk == 2 ? 0.5 : 1.0);
            double hc = 0.0;
            for (int32_t i = 0; i < ntaps; i++) hc += taps_g[i] * cval;           // Util.java:393-401
            double vc = 0.0;
            for (int32_t i = 0; i < ntaps; i++) vc += taps_g[i] * hc;             // Util.java:415-422
            const double fc = lik_factor(g, vc);
            for (int32_t idx = threadIdx.x; idx < (LK_TH / SPLIT) * LK_TW; idx += blockDim.x) {
                const int32_t r = part * (LK_TH / SPLIT) + idx / LK_TW, c = idx % LK_TW;
                const size_t o = (size_t)(ty0 + r) * g.W + tx0 + c;               // tile is inside the map (bit 3 clear)
                if (wr_lik) LIK_STORE(&mlik[o], vc);
                if (wr_fac) FAC_STORE(&mfac[(size_t)(ty0 + r) * g.fpitch + tx0 + c], fc);
            }
            if (wr_fac && tstate && threadIdx.x == 0) { *tstate = want; if (SPLIT == 1) tstate[1] = want; }
            if (part == 0) ts_const_written++;
            continue;
        }
        if (wr_fac && tstate && threadIdx.x == 0) { *tstate = 0; if (SPLIT == 1) tstate[1] = 0; }
        if (part == 0) ts_blurred++;
#if defined(GMS_LIK_EXP) && GMS_LIK_EXP == 1       // experiment: the staging alone (tools/lik_phases.sh)
        continue;
#endif

        if (KH > 0) {
            // ---- phase 2: strips of LK_STRIP outputs along x
            // (RH * 8 strips are 5.25 (KH = 5) wavefronts' worth: the wavefronts that take a second pass rotate from tile to tile,
            // so that over the tiles a workgroup walks no SIMD carries more of them than another)
            constexpr int32_t RHP = CRS;                                        // rows of horizontal sums this workgroup needs: the rows it staged
            for (int32_t sidx = (int32_t)((threadIdx.x + 64u * (uint32_t)tile_iter) & 255u); sidx < RHP * (LK_TW / LK_STRIP); sidx += 256) {
                const int32_t r = R0 + sidx / (LK_TW / LK_STRIP), c0 = (sidx % (LK_TW / LK_STRIP)) * LK_STRIP;
                double v[LK_STRIP + 2 * (KH > 0 ? KH : 1)];
                // c0 and PINB are multiples of 8: the strip's LK_STRIP + 2 KH codes come as 8-byte words
                const uint64_t *row = reinterpret_cast<const uint64_t *>(in_b + r * PINB + c0);
                constexpr int NQ = (LK_STRIP + 2 * KH + 7) / 8;
                uint64_t q[NQ];
#pragma unroll
                for (int j = 0; j < NQ; j++) q[j] = row[j];
#pragma unroll
                // Twice the sums: the codes themselves (0, 1, 2) stand in for the cells' {0, 0.5, 1}.  Doubling every term of an
                // in-order sum of products doubles every intermediate exactly (scaling by a power of two commutes with rounding while
                // nothing is subnormal: the launchers check the taps, gms_map::lik_kh), so 0.5 * total below IS Util.java:393-401's
                // total, bit for bit -- and a conversion is one instruction less per staged value.
                for (int j = 0; j < LK_STRIP + 2 * KH; j++) v[j] = (double)(uint32_t)((q[j >> 3] >> (8 * (j & 7))) & 0xffu);
#pragma unroll
                for (int o = 0; o < LK_STRIP; o++) {
                    double total = taps_g[0] * v[o];            // (== 0.0 + the product: no tap is negative, gms_map::lik_kh)
#pragma unroll
                    for (int i = 1; i < 2 * KH + 1; i++) total += taps_g[i] * v[o + i];
                    hs[r * PHS + c0 + o] = 0.5 * total;         // see "twice the sums" above
                }
            }
            __syncthreads();
            if (tile_iter == 1) GMS_STAMP(GMS_STAMP_ROW(3, blockIdx.x), 4);  // first tile (non-uniform): horizontal sums done
            // ---- phase 3: strips of LK_STRIP outputs along y
#if defined(GMS_LIK_EXP) && GMS_LIK_EXP == 2       // experiment: staging and horizontal sums
            if (hs[threadIdx.x] == 123.0) mlik[0] = 1.0;
            continue;
#endif
            {
                // lane = column, wavefront = eight rows: the rows' bounds tests and base addresses are scalar, a store is
                // `row base (scalar) + column offset (one register)`
                constexpr int VS = LK_STRIP / SPLIT;                               // outputs per lane: four wavefronts x VS rows = this workgroup's rows
                const int32_t c = e_lane, r0 = R0 + e_wave * VS;
                double v[VS + 2 * (KH > 0 ? KH : 1)];
                const double *hcol = hs + r0 * PHS + c;
#pragma unroll
                for (int j = 0; j < VS + 2 * KH; j++) v[j] = hcol[j * PHS];
                const uint32_t gx = (uint32_t)(tx0 + c);
                const bool x_ok = gx < (uint32_t)g.W;
#pragma unroll
                for (int o = 0; o < VS; o++) {
                    double total = taps_g[0] * v[o];
#pragma unroll
                    for (int i = 1; i < 2 * KH + 1; i++) total += taps_g[i] * v[o + i];
                    const int32_t gy = ty0 + r0 + o;
#if defined(GMS_LIK_EXP) && GMS_LIK_EXP == 3       // experiment: everything but the stores
                    if (total == 123.0) mlik[0] = total;
                    continue;
#endif
                    if (gy < g.H && x_ok) {
                        if (wr_lik) LIK_STORE(mlik + (size_t)gy * (size_t)g.W + gx, total);
                        if (wr_fac) FAC_STORE(mfac + (size_t)gy * (size_t)g.fpitch + gx, lik_factor(g, total));
                    }
                }
            }
        } else {
            // ---- generic half width: one output per thread per step, taps from LDS
            for (int32_t idx = threadIdx.x; idx < RH * LK_TW; idx += blockDim.x) {
                const int32_t r = idx / LK_TW, c = idx - r * LK_TW;
                double total = 0.0;
                for (int32_t i = 0; i < ntaps; i++) total += taps_s[i] * in_s[r * PIN + c + i];
                hs[r * PHS + c] = total;
            }
            __syncthreads();
            for (int32_t idx = threadIdx.x; idx < LK_TH * LK_TW; idx += blockDim.x) {
                const int32_t r = idx / LK_TW, c = idx - r * LK_TW;
                const int32_t gy = ty0 + r, gx = tx0 + c;
                if (gx < g.W && gy < g.H) {
                    double total = 0.0;
                    for (int32_t i = 0; i < ntaps; i++) total += taps_s[i] * hs[(r + i) * PHS + c];
                    if (wr_lik) mlik[(size_t)gy * g.W + gx] = total;
                    if (wr_fac) FAC_STORE(&mfac[(size_t)gy * g.fpitch + gx], lik_factor(g, total));
                }
            }
        }
    }
    if (g.tile_stats && threadIdx.x == 0) {
        uint32_t *row = g.tile_stats + 4 * (bx & 63u);
        if (ts_left) atomicAdd(row + 0, ts_left);
        if (ts_const_kept) atomicAdd(row + 1, ts_const_kept);
        if (ts_const_written) atomicAdd(row + 2, ts_const_written);
        if (ts_blurred) atomicAdd(row + 3, ts_blurred);
    }
}

template <int KH, bool PENDING, int SPLIT = 1>      // PENDING = false: no count grid is read (the code for it is not generated: 1.3 us of a 21 us full rebuild)
__global__ void __launch_bounds__(256) GMS_LIK_WAVES
k_likelihood(GridDev g, const double *__restrict__ logd, double *__restrict__ lik, double *__restrict__ fac,
             int64_t fac_stride, const double *__restrict__ taps_g, const int32_t *__restrict__ bbox, int32_t dirty_only,
             int32_t tiles_x, int32_t tiles_y, uint8_t *__restrict__ tile_state, const uint32_t *__restrict__ cnt_pending,
             int32_t *__restrict__ bbox_clear, int32_t mode) {
    extern __shared__ __align__(16) unsigned char smem[];
    // bbox_clear: the box half the NEXT ray cast will raise (it shares its launch with this scan's deferred apply pass,
    // which therefore cannot clear it: k_raycast_apply); nobody reads it during this launch
    if (bbox_clear && blockIdx.x == 0 && threadIdx.x < 4) bbox_clear[4 * blockIdx.y + threadIdx.x] = 0;
    likelihood_body<KH, SPLIT>(g, logd, lik, fac, fac_stride, taps_g, bbox, dirty_only, tiles_x, tiles_y, blockIdx.x, blockIdx.y,
                               gridDim.x, smem, PENDING ? cnt_pending : (const uint32_t *)nullptr, tile_state, mode);
}

// The stand-alone map update (GridMap.integrateObservation + computeLikelihoodMap as an entry point of its own) in two
// launches instead of three: this scan's ray cast (into the idle count grid and the idle box half) beside the PREVIOUS
// scan's `logData[c] += ...` (GridMap.java:223, from the other count grid); the likelihood pass that follows adds this
// scan's counts on the fly, as it does in the paired scan step.  grid.x = ray blocks + apply blocks, one map.
__global__ void __launch_bounds__(256)
k_raycast_apply(GridDev g, const gms_beam *__restrict__ beams, int32_t B, int32_t beam_stride, const float *__restrict__ poses,
                int32_t pose_stride, uint32_t *__restrict__ cnt, int32_t *__restrict__ bbox_next, int32_t nw_max, uint32_t n_ray_blocks,
                uint32_t n_near_blocks, double *__restrict__ logd, uint32_t *__restrict__ cnt_pend, const int32_t *__restrict__ bbox_pend) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (blockIdx.x < n_ray_blocks)
        raycast_body<false, 4>(g, beams, B, beam_stride, poses, pose_stride, nullptr, cnt, bbox_next, nullptr, nullptr, 0, nullptr, nw_max,
                               blockIdx.x, 0, smem, nullptr, n_near_blocks ? 1 : 0);
    else if (blockIdx.x < n_ray_blocks + n_near_blocks)
        raycast_near_body(g, beams, B, beam_stride, poses, pose_stride, cnt, bbox_next, blockIdx.x - n_ray_blocks, 0, smem, nullptr);
    else
        apply_body(g, logd, cnt_pend, bbox_pend, nullptr, blockIdx.x - n_ray_blocks - n_near_blocks, 0, gridDim.x - n_ray_blocks - n_near_blocks);
}

// scoring factors from an existing likelihood field (upload / copy), and the table's neutral border (fac_index)
__global__ void k_factors(GridDev g, const double *__restrict__ lik, double *__restrict__ fac, int64_t fac_stride) {
    const int32_t mi = blockIdx.y;
    double *mfac = fac + (size_t)mi * fac_stride;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < fac_stride; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t gy = (int32_t)(i / g.fpitch), gx = (int32_t)(i - (int64_t)gy * g.fpitch);
        mfac[i] = gx < g.W && gy < g.H ? lik_factor(g, lik[(size_t)mi * g.cells + (size_t)gy * g.W + gx]) : 1.0;
    }
}

// GridMapApp.calculateCombined (J/app/GridMapApp.java:439-458): one combined map out of the n_maps of a batch,
// log-odds of 1 - prod(1 - p_m), the maps multiplied in index order (Util.invLogOdds / logOdds: Util.java:35-48).
__device__ __forceinline__ void combine_body(const double *__restrict__ logs, int32_t n_maps, int64_t cells, double *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (int64_t)gridDim.x * blockDim.x) {
        double product = 1.0;
        for (int32_t m = 0; m < n_maps; m++) {
            const double l = logs[(size_t)m * cells + i];
            product *= 1.0 - ((double)1.0f - (double)1.0f / (1.0 + exp(l)));
        }
        const double odds = 1.0 - product;
        out[i] = log(odds / ((double)1.0f - odds));
    }
}
__global__ void k_combine(const double *__restrict__ logs, int32_t n_maps, int64_t cells, double *__restrict__ out) { combine_body(logs, n_maps, cells, out); }

// The de-skew loop of GridMapApp.onHandleData (J/app/GridMapApp.java:143-175) + Measurement(x, y, wasHit, dummy)
// (J/slam/Observation.java:69-76): raw polar measurements {angle, distance, hit} -> beams, on the device.
__device__ __forceinline__ void deskew_body(const double *__restrict__ angle, const double *__restrict__ distance,
                                            const uint8_t *__restrict__ hit, int32_t length, double d_center, double d_theta,
                                            gms_beam *__restrict__ out, int32_t i) {
    if (i >= length) return;
    const double d_i = -(double)(length - i) / (double)length;                    // :150
    const double delta_theta = d_theta * d_i, delta_x = d_center * d_i;           // :157-158
    const double a = angle[i] + delta_theta;
    const double x_a = distance[i] * cos(a) + delta_x;                            // :166
    const double y_a = distance[i] * sin(a);                                      // :167
    gms_beam b;
    b.local_x = x_a; b.local_y = y_a;
    b.distance = sqrt(x_a * x_a + y_a * y_a);                                     // Observation.java:71
    b.hit = hit[i] ? 1 : 0;
    for (int k = 0; k < 7; k++) b.pad_[k] = 0;
    out[i] = b;
}
__global__ void k_deskew(const double *__restrict__ angle, const double *__restrict__ distance,
                         const uint8_t *__restrict__ hit, int32_t length, double d_center, double d_theta,
                         gms_beam *__restrict__ out) {
    deskew_body(angle, distance, hit, length, d_center, d_theta, out, (int32_t)(blockIdx.x * blockDim.x + threadIdx.x));
}

__global__ void k_fill(double *__restrict__ d, double v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = v;
}

// getRawAt / getProbAt (GridMap.java:134-140, Util.java:46-48)
__global__ void k_get_raw(GridDev g, const double *__restrict__ logd, int32_t mi, int32_t x, int32_t y, double *out2) {
    const double l = logd[(size_t)mi * g.cells + (size_t)y * g.W + x];
    out2[0] = l;
    out2[1] = (double)1.0f - (double)1.0f / (1.0 + exp(l));
}

// diagnostics: the float-rounded primitives the parity contract leans on
__global__ void k_debug_f32(int32_t op, const float *__restrict__ a, float *__restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float c, s;
        if (op == 0) out[i] = j_sqrtf(a[i]);
        else if (op == 3) {
            // wave_xor<O> against __shfl_xor, both readings of the swap order / rotation direction: bit k (k = 0..5 for O = 32..1) set
            // when the product form differs from __shfl_xor in this lane, bit 6 + k when the FLIP form does.  (n is a multiple of 64.)
            const uint32_t v = __float_as_uint(a[i]);
            uint32_t code = 0;
#define GMS_STEP_(O) { const uint32_t want = (uint32_t)__shfl_xor((int)v, O, GMS_WAVE);                              \
                       code |= (wave_xor_u32<O, false>(v) != want ? 1u : 0u) << (5 - __builtin_ctz(O));            \
                       code |= (wave_xor_u32<O, true>(v) != want ? 1u : 0u) << (11 - __builtin_ctz(O)); }
            GMS_BUTTERFLY(GMS_STEP_)
#undef GMS_STEP_
            out[i] = (float)code;
        }
        else if (op == 4) out[i] = sq_lower(a[i]);
        else if (op == 5) out[i] = sq_upper(a[i]);
        else { pose_trig(a[i], c, s); out[i] = op == 1 ? c : s; }
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline int32_t rc_nw_max(const gms_map *m) { return (m->gd.W + m->gd.H + 1 + 31) / 32; }
static inline size_t rc_smem(const gms_map *m, int rays, uint32_t n_near = 0) {
    const size_t a = (size_t)rc_nw_max(m) * rays * sizeof(uint64_t);
    return n_near && a < RCN_LDS_BYTES ? (size_t)RCN_LDS_BYTES : a;              // near-field blocks carry their count tile
}

template <bool TRACE, int RAYS>
static void rc_launch(gms_map *m, dim3 grid, const gms_beam *d_beams, int32_t B, int32_t beam_stride, const float *d_poses,
                      int32_t pose_stride, const RayIn *single, uint32_t *cnt, int32_t *bbox, int32_t *t_cells, uint8_t *t_cls,
                      int32_t cap, int32_t *t_counts, uint32_t n_near = 0) {
    const size_t smem = rc_smem(m, RAYS, n_near);
    grid.x += n_near;                                       // near-field workgroups behind the ray blocks (raycast_near_body)
    if (smem > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_raycast<TRACE, RAYS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem);
    hipLaunchKernelGGL((k_raycast<TRACE, RAYS>), grid, dim3(RAYS * 64), smem, m->stream, m->gd, d_beams, B, beam_stride, d_poses,
                       pose_stride, single, cnt, bbox, t_cells, t_cls, cap, t_counts, rc_nw_max(m), n_near);
}

// batched ray casts of more than raycast_tile_min rays in all go through LDS tiles (k_raycast_tile)
bool gms_raycast_tiled(const gms_map *m, int32_t B) { return (int64_t)B * m->n_maps > m->raycast_tile_min && m->raycast_tile; }

void gms_launch_raycast(gms_map *m, const gms_beam *d_beams, int32_t B, int32_t beam_stride, const float *d_poses,
                        int32_t pose_stride, bool take_pending_apply) {
    const bool tile = gms_raycast_tiled(m, B);
    // take_pending_apply: a deferred apply pass rides in this launch (tiled form only) instead of a launch of its own: the ray
    // cast then raises the other box half (cleared by the previous likelihood launch) and fills the other count grid
    const bool riding = take_pending_apply && tile && m->apply_pending;
    if (!riding) gms_flush_apply(m);
    ProfScope ps(m, GMS_K_RAYCAST);
    int32_t *bb = m->d_bbox + (size_t)m->bbox_cur * m->n_maps * 4;
    if (tile) {
        // batched maps: throughput-bound; 64 consecutive beams per workgroup, counts accumulated in an LDS tile
        const size_t smem = RCT_LDS_BYTES;
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_raycast_tile), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        const uint32_t n_ray = (uint32_t)((B + RCT_RAYS - 1) / RCT_RAYS);
        uint32_t n_apply = 0;
        int32_t *next = bb;
        if (riding) {
            const int32_t all = ((m->gd.W + APPLY_TW - 1) / APPLY_TW) * ((m->gd.H + APPLY_TH - 1) / APPLY_TH);
#ifndef GMS_APPLY_RIDING_TILE
#define GMS_APPLY_RIDING_TILE 4        // workgroups of four 256-thread slices per map (C5 step, us: 1 / 2 / 3 / 4 / 6 / 8 / 16 / 32: 462 / 427 / 421 / 419 / 422 / 422 / 427 / 427; the pass beside the partials: 431)
#endif
            n_apply = (uint32_t)((all < GMS_APPLY_BLOCKS ? all : GMS_APPLY_BLOCKS) + 3) / 4;
            if (n_apply > GMS_APPLY_RIDING_TILE) n_apply = GMS_APPLY_RIDING_TILE;
            if (n_apply < 1) n_apply = 1;
            next = m->d_bbox + (size_t)(1 - m->bbox_cur) * m->n_maps * 4;
        }
        hipLaunchKernelGGL(k_raycast_tile, dim3(n_ray + n_apply, m->n_maps), dim3(RCT_THREADS), smem, m->stream, m->gd,
                           d_beams, B, beam_stride, d_poses, pose_stride, m->d_cnt, next, n_ray, m->d_log, m->d_cnt_pend, bb);
        if (riding) gms_apply_done(m);
    } else if ((int64_t)B * m->n_maps > m->raycast_tile_min)       // (GMS_RAYCAST_TILE=0, or not enough LDS: 16 rays per workgroup, direct atomics)
        rc_launch<false, 16>(m, dim3((B + 15) / 16, m->n_maps), d_beams, B, beam_stride, d_poses, pose_stride, nullptr, m->d_cnt, bb,
                             nullptr, nullptr, 0, nullptr);
    else
        rc_launch<false, 4>(m, dim3((B + 3) / 4, m->n_maps), d_beams, B, beam_stride, d_poses, pose_stride, nullptr, m->d_cnt, bb,
                            nullptr, nullptr, 0, nullptr, rc_near_blocks(m, B));
}

void gms_launch_trace_scan(gms_map *m, const gms_beam *d_beams, int32_t B, const float *d_pose, int32_t *d_cells,
                           uint8_t *d_cls, int32_t cap, int32_t *d_counts) {
    rc_launch<true, 4>(m, dim3((B + 3) / 4, 1), d_beams, B, m->max_beams, d_pose, 3, nullptr, nullptr, nullptr, d_cells, d_cls, cap,
                       d_counts);
}

void gms_launch_trace_ray(gms_map *m, float x0, float y0, float x1, float y1, int32_t extra, int32_t *d_cells,
                          int32_t cap, int32_t *d_count) {
    hipLaunchKernelGGL(k_trace_ray, dim3(1), dim3(64), 0, m->stream, m->gd.W, m->gd.H, x0, y0, x1, y1, extra, d_cells,
                       cap, d_count);
}

__global__ void k_store_ray(RayIn *dst, RayIn r) { *dst = r; }

void gms_launch_apply_ray(gms_map *m, RayIn ray) {
    gms_flush_apply(m);
    // the single ray travels through the beam staging buffer
    RayIn *d_ray = reinterpret_cast<RayIn *>(m->d_beams);
    hipLaunchKernelGGL(k_store_ray, dim3(1), dim3(1), 0, m->stream, d_ray, ray);
    ProfScope ps(m, GMS_K_RAYCAST);
    rc_launch<false, 4>(m, dim3(1, 1), nullptr, 1, m->max_beams, nullptr, 3, d_ray, m->d_cnt,
                        m->d_bbox + (size_t)m->bbox_cur * m->n_maps * 4, nullptr, nullptr, 0, nullptr);
}

// The paired scan step leaves its counts un-applied (the likelihood pass adds them on the fly) so that the apply pass
// can run beside the next step's weight reduction.  Anything else that reads or writes logData, the counts or the box
// first brings the map to the state the immediate protocol would have left: counts applied and cleared, the idle half
// of the box cleared, the box consumed.
static void apply_launch(gms_map *m) {
    ProfScope ps(m, GMS_K_APPLY);
    const int32_t all = ((m->gd.W + APPLY_TW - 1) / APPLY_TW) * ((m->gd.H + APPLY_TH - 1) / APPLY_TH);
    dim3 grid(all < GMS_APPLY_BLOCKS ? all : GMS_APPLY_BLOCKS, m->n_maps);
    int32_t *cur = m->d_bbox + (size_t)m->bbox_cur * m->n_maps * 4, *idle = m->d_bbox + (size_t)(1 - m->bbox_cur) * m->n_maps * 4;
    // a deferred pass applies the grid that was set aside (gms_defer_apply); the immediate one the grid just cast into
    hipLaunchKernelGGL(k_apply, grid, dim3(256), 0, m->stream, m->gd, m->d_log, m->apply_pending ? m->d_cnt_pend : m->d_cnt, cur, idle);
}
void gms_apply_done(gms_map *m) {          // host bookkeeping after a deferred apply pass has been enqueued
    m->bbox_cur = 1 - m->bbox_cur;
    m->bbox_dirty = 0;
    m->apply_pending = 0;
}
void gms_defer_apply(gms_map *m) {         // the scan just cast (and already in the likelihood field) keeps its counts for a later launch
    uint32_t *t = m->d_cnt; m->d_cnt = m->d_cnt_pend; m->d_cnt_pend = t;       // the other grid is all zero: the next ray cast's
    m->apply_pending = 1;
    m->bbox_dirty = 0;
}

// this scan's ray cast beside the previous scan's deferred apply pass (k_raycast_apply); single maps, scans of <= 4096 beams
void gms_launch_raycast_apply(gms_map *m, const gms_beam *d_beams, int32_t B, int32_t beam_stride, const float *d_poses,
                              int32_t pose_stride) {
    ProfScope ps(m, GMS_K_RAYCAST);
    const uint32_t n_ray = (uint32_t)((B + 3) / 4);
    const int32_t all = ((m->gd.W + APPLY_TW - 1) / APPLY_TW) * ((m->gd.H + APPLY_TH - 1) / APPLY_TH);
    const uint32_t n_apply = (uint32_t)(all < GMS_APPLY_BLOCKS ? all : GMS_APPLY_BLOCKS);
    int32_t *pend = m->d_bbox + (size_t)m->bbox_cur * 4, *next = m->d_bbox + (size_t)(1 - m->bbox_cur) * 4;
    const uint32_t n_near = rc_near_blocks(m, B);
    const size_t smem = rc_smem(m, 4, n_near);
    if (smem > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_raycast_apply), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k_raycast_apply, dim3(n_ray + n_near + n_apply), dim3(256), smem, m->stream, m->gd, d_beams, B, beam_stride, d_poses,
                       pose_stride, m->d_cnt, next, rc_nw_max(m), n_ray, n_near, m->d_log, m->d_cnt_pend, pend);
    gms_apply_done(m);                      // the box of the scan just cast is the current half now
}
void gms_flush_apply(gms_map *m) {
    if (!m->apply_pending) return;
    hipSetDevice(m->device);
    apply_launch(m);
    gms_apply_done(m);
}

void gms_launch_apply_counts(gms_map *m) {
    gms_flush_apply(m);
    apply_launch(m);
    m->bbox_dirty = 1;
    m->fac_current = 0;            // logData moved on without a rebuild having seen these counts: the next rebuild leaves no tile alone
}

// dynamic LDS of a likelihood workgroup (likelihood_body's layout) and how many of them to launch per map
size_t gms_likelihood_lds_bytes(int32_t k, bool coded) {
    const size_t RH = LK_TH + 2 * k, RW = LK_TW + 2 * k;
    if (coded && (k == 3 || k == 5)) return ((RH * ((RW + 7) & ~(size_t)7) + 15) & ~(size_t)15) + RH * (LK_TW + 1) * sizeof(double);
    return (RH * (RW + 1) + RH * (LK_TW + 1) + (2 * (size_t)k + 1)) * sizeof(double);
}
int32_t gms_likelihood_blocks_cap(const gms_map *m, size_t smem) {
    // persistent workgroups: as many as stay resident (LDS per CU and CU count of the map's device, read once at gms_map_create:
    // 160 KiB and 256 on MI355X; registers allow five)
#ifndef GMS_LIK_WG_PER_CU
#define GMS_LIK_WG_PER_CU 5
#endif
    const int32_t n_maps = m->n_maps;
    int32_t per_cu = (int32_t)((size_t)m->lds_per_cu / (smem + 256));
    if (per_cu > GMS_LIK_WG_PER_CU) per_cu = GMS_LIK_WG_PER_CU;
    if (per_cu < 1) per_cu = 1;
    const int32_t resident = per_cu * m->n_cus;
    if (n_maps <= 4) return resident / n_maps;
    // many maps: two residencies' worth of workgroups over all maps, a quarter of one at most per map.  A map's dirty box is a few
    // dozen tiles; workgroups beyond that only cost their dispatch (C5, 64 maps: 320 / 160 / 80 / 40 / 20 / 10 per map ->
    // 50.3 / 48.0 / 48.2 / 43.9 / 47.8 / 65.8 us for the likelihood | resample launch)
    int32_t per_map = 2 * resident / n_maps;
    if (per_map > resident / 4) per_map = resident / 4;
    if (per_map < 8) per_map = 8;
    return per_map;
}

// Whether a dirty-tile rebuild launched with `blocks` workgroups per map may split its tiles in two: one map, and twice the tiles of the
// largest box a scan can touch (both ends of a ray of maximal range, its walk's margin, the blur's reach) are not more than `blocks`
bool gms_likelihood_split(const gms_map *m, int32_t blocks) {
    if (m->n_maps != 1 || m->lik_kh == 0 || m->lik_split == 0) return false;
    const double side = 2.0 / (m->gd.inv_max * m->gd.res) + 2.0 * m->gd.khalf + 2.0 * RC_BOX_MARGIN + 8.0;       // cells
    const int64_t tx = (int64_t)(side / LK_TW) + 2, ty = (int64_t)(side / LK_TH) + 2;
    const int64_t tiles_x = (m->gd.W + LK_TW - 1) / LK_TW, tiles_y = (m->gd.H + LK_TH - 1) / LK_TH;
    const int64_t most = (tx < tiles_x ? tx : tiles_x) * (ty < tiles_y ? ty : tiles_y);
    return 2 * (((most + 7) >> 3) << 3) <= blocks;
}

void gms_launch_likelihood(gms_map *m, int32_t dirty_only, bool counts_pending, bool materialize) {
    // counts_pending: the scan just cast is not in logData yet; its counts (m->d_cnt) are added on the fly and its apply pass
    // is deferred by the caller (gms_defer_apply); the other box half is cleared for the next ray cast
    // materialize: bring likelihoodData up to date everywhere (mode 1) from logData and -- when an apply pass is still deferred --
    // the counts that pass will add: exactly what the last rebuild of the factor table saw (gms_ensure_lik)
    if (!counts_pending && !materialize) gms_flush_apply(m);
    int32_t mode = 3;
    if (materialize) mode = 1;
    else if (dirty_only && m->lik_lazy && (counts_pending || m->lik_stale)) { mode = 2; m->lik_stale = 1; }   // the hot path: factor table only
    else if (!dirty_only) {
        if (m->lik_stale) gms_invalidate_tile_state(m);       // the tile states speak for the factor table only: trust none of them for likelihoodData
        m->lik_stale = 0;
    }
    if (!materialize) {
        if (dirty_only && counts_pending && m->fac_current && m->lik_skip) mode |= 4;     // unchanged tiles are left alone
        m->fac_current = 1;                                   // after this launch the factor table is the field of logData + pending counts
    }
    ProfScope ps(m, GMS_K_LIKELIHOOD);
    const int32_t k = m->lik_kh;
    const int32_t tiles_x = (m->gd.W + LK_TW - 1) / LK_TW, tiles_y = (m->gd.H + LK_TH - 1) / LK_TH;
    const size_t smem = gms_likelihood_lds_bytes(m->gd.khalf, k != 0);
    // persistent workgroups, each walks tiles blockIdx.x, += gridDim.x
    int32_t blocks = tiles_x * tiles_y;
    const int32_t cap = gms_likelihood_blocks_cap(m, smem);
    if (blocks > cap) blocks = cap;
    blocks = (blocks + 7) & ~7;                      // keep the XCD round-robin aligned
    dim3 grid(blocks, m->n_maps);
    const int32_t *bb = m->d_bbox + (size_t)m->bbox_cur * m->n_maps * 4;
    const uint32_t *pend = counts_pending ? m->d_cnt : (const uint32_t *)nullptr;
    int32_t *bb_clear = counts_pending ? m->d_bbox + (size_t)(1 - m->bbox_cur) * m->n_maps * 4 : (int32_t *)nullptr;
    uint8_t *tstate = m->d_tile_state;
    if (materialize) {
        dirty_only = 0; bb_clear = nullptr; tstate = nullptr;
        counts_pending = m->apply_pending != 0;
        pend = counts_pending ? m->d_cnt_pend : (const uint32_t *)nullptr;       // (gms_defer_apply has swapped the grids)
    }
    // a dirty-tile rebuild of a single map splits its tiles over two workgroups each when the largest box a scan can touch still leaves
    // every (tile, part) a workgroup of its own (likelihood_body, "split")
    const bool split = k != 0 && dirty_only && counts_pending && gms_likelihood_split(m, blocks);
#define LK_LAUNCH(KH)                                                                                         \
    do {                                                                                                      \
        if (split) LK_LAUNCH2(KH, true, 2); else if (counts_pending) LK_LAUNCH2(KH, true, 1); else LK_LAUNCH2(KH, false, 1); \
    } while (0)
#define LK_LAUNCH2(KH, PEND, SP)                                                                              \
    do {                                                                                                      \
        if (smem > 48 * 1024)                                                                                 \
            hipFuncSetAttribute(reinterpret_cast<const void *>(&k_likelihood<KH, PEND, SP>),                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                       \
        hipLaunchKernelGGL((k_likelihood<KH, PEND, SP>), grid, dim3(256), smem, m->stream, m->gd, m->d_log, m->d_lik, \
                           m->d_fac, m->fac_stride, m->d_taps, bb, dirty_only, tiles_x, tiles_y, tstate, pend, bb_clear, mode);                         \
    } while (0)
    if (k == 3) LK_LAUNCH(3);
    else if (k == 5) LK_LAUNCH(5);
    else { if (counts_pending) LK_LAUNCH2(0, true, 1); else LK_LAUNCH2(0, false, 1); }
#undef LK_LAUNCH
#undef LK_LAUNCH2
}

// likelihoodData as the reference would have it: the field of the last computeLikelihoodMap / scan step, everywhere
void gms_ensure_lik(gms_map *m) {
    if (!m->lik_stale) return;
    hipSetDevice(m->device);
    gms_launch_likelihood(m, 0, false, true);
    m->lik_stale = 0;
}

// pinned host memory -> device memory, 16 bytes per lane (nbytes rounded up by the caller's buffers).  A kernel rather
// than hipMemcpyAsync: enqueueing an asynchronous host-to-device copy costs ~45 us of HOST time on this stack
// (tools/host_path_probe.py), a launch ~5 us.
__global__ void k_copy16(uint4 *__restrict__ dst, const uint4 *__restrict__ src, int64_t n16) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
void gms_launch_copy(gms_map *m, void *dst, const void *src, size_t nbytes) {
    const int64_t n16 = (int64_t)((nbytes + 15) / 16);
    if (n16 == 0) return;
    const int64_t blocks = (n16 + 255) / 256;
    hipLaunchKernelGGL(k_copy16, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, m->stream,
                       reinterpret_cast<uint4 *>(dst), reinterpret_cast<const uint4 *>(src), n16);
}

bool gms_set_stamp_buffer(gms_map *m, void *dev_buffer) {
#ifdef GMS_STAMPS
    hipStreamSynchronize(m->stream);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_gms_stamps), &dev_buffer, sizeof(dev_buffer)) == hipSuccess;
#else
    (void)m; (void)dev_buffer;
    return false;
#endif
}

__global__ void k_noop() {}
// holds the stream for `us` microseconds (wall_clock64: the 100 MHz constant clock); calibration only
__global__ void k_spin(double us) {
    const uint64_t t0 = wall_clock64();
    const uint64_t ticks = (uint64_t)(us * 100.0);
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
void gms_launch_spin(gms_map *m, double us) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, m->stream, us); }
void gms_launch_noop(gms_map *m) { hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, m->stream); }

// likelihoodData / the factor table were written by something other than the likelihood kernel: nothing is known
// about their tiles any more
void gms_invalidate_tile_state(gms_map *m) {
    const size_t tiles = (size_t)((m->gd.W + LK_TW - 1) / LK_TW) * ((m->gd.H + LK_TH - 1) / LK_TH);
    hipMemsetAsync(m->d_tile_state, 0, 2 * tiles * m->n_maps, m->stream);     // (two entries per tile: likelihood_body)
}

void gms_launch_factors(gms_map *m) {
    gms_invalidate_tile_state(m);
    hipLaunchKernelGGL(k_factors, dim3(1024, m->n_maps), dim3(256), 0, m->stream, m->gd, m->d_lik, m->d_fac, m->fac_stride);
}

void gms_launch_combine(gms_map *src, gms_map *dst) {   // the caller has flushed both maps and ordered dst's stream behind src's
    hipLaunchKernelGGL(k_combine, dim3(2048), dim3(256), 0, dst->stream, src->d_log, src->n_maps, src->gd.cells, dst->d_log);
}

void gms_launch_deskew(gms_map *m, const double *d_angle, const double *d_distance, const uint8_t *d_hit, int32_t length,
                       double d_center, double d_theta, gms_beam *d_out) {
    hipLaunchKernelGGL(k_deskew, dim3((length + 255) / 256), dim3(256), 0, m->stream, d_angle, d_distance, d_hit, length,
                       d_center, d_theta, d_out);
}

void gms_launch_fill(gms_map *m, double *d, double v, int64_t n) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)blocks), dim3(256), 0, m->stream, d, v, n);
}

void gms_launch_get_raw(gms_map *m, int32_t mi, int32_t x, int32_t y, double *d_out2) {
    hipLaunchKernelGGL(k_get_raw, dim3(1), dim3(1), 0, m->stream, m->gd, m->d_log, mi, x, y, d_out2);
}

void gms_launch_debug_f32(gms_map *m, int32_t op, const float *d_a, float *d_out, int64_t n) {
    hipLaunchKernelGGL(k_debug_f32, dim3(1024), dim3(256), 0, m->stream, op, d_a, d_out, n);
}
