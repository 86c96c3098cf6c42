// gms_fused_kernels.hip -- launches that run two independent kernels of a scan step side by side.
//
// After the weights are reduced a scan step has two independent branches (SLAM.java:100-124 vs :93,:102-105):
//     weights:  normalise + pack  ->  [all-gather]  ->  cumulative sums  ->  resample
//     map:      ray cast          ->  likelihood rebuild (counts added on the fly)  ...  apply counts (next step)
// Every one of those kernels is a few microseconds of latency on a small part of the GPU (the ray cast keeps 180
// workgroups busy, the normalise 64).  On one in-order stream they run back to back; on two streams the event
// fork/join costs more than it hides (measured: +18 us per step).  So the branches are paired inside single
// launches instead: workgroups [0, n_first) run one kernel's body, the rest run the other's.  The bodies are
// the same device functions the stand-alone kernels wrap, so results are bit-identical to the separate calls
// (tests/test_gpu_parity.py::test_fused_scan_step_equals_the_separate_calls).
//
// This file is the device translation unit of the library: it includes the two kernel files.
#include "gms_map_kernels.hip"
#include "gms_pf_kernels.hip"
#include "gms_slam_kernels.hip"

// 256-thread workgroups of a deferred apply pass that rides beside a ray cast (128 to 2048 measure alike at C3)
#ifndef GMS_APPLY_BLOCKS_RIDING
#define GMS_APPLY_BLOCKS_RIDING 512
#endif

// rays per 256-thread workgroup of the paired launches (one producer wavefront, three consumers)
#ifndef RCF_RAYS
#define RCF_RAYS 4
#endif

// ---- A: normalise + pack  |  ray cast at the weighted pose --------------------------------------------------
// The ray-cast workgroups fold the weighted pose from the partial vector themselves (the arithmetic of
// fold_stats: SLAM.java:165-178), because the workgroup that publishes it runs beside them.
template <bool LOGNORM>   // (the log-normalising form is a separate instantiation: the default kernel is exactly what it was -- as one kernel
                          // with a uniform branch its ray blocks ran 1.1 us longer, measured A/B on one box)
__global__ void __launch_bounds__(256)
k_norm_raycast(GridDev g, const gms_beam *__restrict__ beams, int32_t B, uint32_t *__restrict__ cnt, int32_t *__restrict__ bbox,
               int32_t nw_max, uint32_t n_ray_blocks,
               const double *__restrict__ partials, int64_t nblk_global, double *__restrict__ w, const float *__restrict__ pose,
               int32_t n, int64_t offset, PackedParticle *__restrict__ packed, double *__restrict__ cum,
               double *__restrict__ chunk_tot, int64_t nchunks, double *__restrict__ p2, PfStatsDev *__restrict__ stats,
               uint32_t n_norm_blocks, double *__restrict__ logd, uint32_t *__restrict__ cnt_pend, const int32_t *__restrict__ bbox_pend,
               uint32_t n_near_blocks, const double *__restrict__ logw_lognorm) {
    extern __shared__ __align__(16) unsigned char smem[];
    // workgroups: [far-field ray blocks | near-field ray blocks) = n_ray_blocks, then normalise, then the riding apply pass
    // logw_lognorm != nullptr: the partial vector is block-relative (gms_pf_set_log_normalize, block_partials)
    GMS_STAMP(GMS_STAMP_ROW(2, blockIdx.x), 0);
    if (blockIdx.x >= n_ray_blocks + n_norm_blocks) {
        // The PREVIOUS scan's `logData[c] += ...` (GridMap.java:223) from the other count grid: it needs nothing of this launch
        // and only has to be done before this scan's likelihood pass.  Its ~5 us hide under the ray cast's 17 us latency chain
        // on CUs that launch leaves idle (beside the block partials, round 1's place, it cost that launch 2.3 us).
        apply_body(g, logd, cnt_pend, bbox_pend, nullptr, blockIdx.x - n_ray_blocks - n_norm_blocks, 0,
                   gridDim.x - n_ray_blocks - n_norm_blocks);
        GMS_STAMP(GMS_STAMP_ROW(2, blockIdx.x), 15);
        return;
    }
    if (blockIdx.x < n_ray_blocks) {
        __shared__ RedLds L;
        __shared__ float s_pose[3];
        const int cols[4] = { COL_SUM, COL_XW, COL_YW, COL_TW };
        double f[4];
        if (LOGNORM) {                                                 // the population's reference, then the rescaled sums
            fold_lognorm<4>(partials, nblk_global, cols, f, L);
        } else {
            fold_sums<4>(partials, nblk_global, cols, f, L);           // one round trip, one barrier pair
        }
        if (threadIdx.x == 0) {
            s_pose[0] = (float)(f[1] / f[0]);                          // SLAM.java:176
            s_pose[1] = (float)(f[2] / f[0]);
            s_pose[2] = (float)(f[3] / f[0]);
        }
        __syncthreads();
        GMS_STAMP(GMS_STAMP_ROW(2, blockIdx.x), 1);
        if (blockIdx.x >= n_ray_blocks - n_near_blocks)
            raycast_near_body(g, beams, B, B, nullptr, 0, cnt, bbox, blockIdx.x - (n_ray_blocks - n_near_blocks), 0, smem, s_pose);
        else
            raycast_body<false, RCF_RAYS, 4>(g, beams, B, B, nullptr, 0, nullptr, cnt, bbox, nullptr, nullptr, 0, nullptr, nw_max, blockIdx.x, 0,
                                             smem, s_pose, n_near_blocks ? 1 : 0);
    } else {
        normalize_pack_body(partials, nblk_global, w, pose, n, offset, packed, cum, chunk_tot, nchunks, p2, stats,
                            blockIdx.x - n_ray_blocks, 0, LOGNORM ? logw_lognorm : (const double *)nullptr);
        GMS_STAMP(GMS_STAMP_ROW(2, blockIdx.x), 14);
    }
}

// ---- B: block partials of the weights  |  the PREVIOUS scan's apply pass -------------------------------------
// The likelihood pass of a paired step adds the scan's counts on the fly, so `logData += ...` (GridMap.java:223)
// is off the critical path: it runs here, beside the next scan's weight reduction, before that scan's ray cast.
__global__ void __launch_bounds__(256)
k_partials_apply(double *__restrict__ w, double *__restrict__ logw, const float *__restrict__ pose, int32_t n, int64_t offset,
                 int64_t nblk_global, double *__restrict__ partials, const double *__restrict__ part, int32_t part_nseg,
                 GridDev g, double *__restrict__ logd, uint32_t *__restrict__ cnt, const int32_t *__restrict__ bbox,
                 int32_t *__restrict__ bbox_idle, PfStatsDev *__restrict__ lognorm_stats) {
    if (blockIdx.x < (uint32_t)nblk_global)
        partials_body(w, logw, pose, n, offset, nblk_global, partials, part, part_nseg, blockIdx.x, blockIdx.y, lognorm_stats);
    else
        apply_body(g, logd, cnt, bbox, bbox_idle, blockIdx.x - (uint32_t)nblk_global, blockIdx.y, gridDim.x - (uint32_t)nblk_global);
}

// ---- sharded filters, one all-gather per scan (gms_slam_update_sharded_*) ---------------------------------------
// B': this shard's block partials and the raw pack of its particles (the two all-gather payloads)  |  previous apply
__global__ void __launch_bounds__(256)
k_partials_pack_apply(double *__restrict__ w, double *__restrict__ logw, const float *__restrict__ pose, int32_t n, int64_t offset,
                      int64_t nblk_global, double *__restrict__ partials, const double *__restrict__ part, int32_t part_nseg,
                      PackedParticle *__restrict__ packed_local, uint32_t n_local_blocks,
                      GridDev g, double *__restrict__ logd, uint32_t *__restrict__ cnt, const int32_t *__restrict__ bbox,
                      int32_t *__restrict__ bbox_idle) {
    if (blockIdx.x < n_local_blocks) {
        partials_body(w, logw, pose, n, offset, nblk_global, partials, part, part_nseg, (uint32_t)(offset / GMS_BLOCK) + blockIdx.x, 0);
        pack_raw_block(w, pose, n, blockIdx.x, packed_local);         // each thread re-reads the weight it stored itself
    } else {
        apply_body(g, logd, cnt, bbox, bbox_idle, blockIdx.x - n_local_blocks, 0, gridDim.x - n_local_blocks);
    }
}

// A': ray cast at the weighted pose  |  normalise this shard's weights + statistics  |  level 0 of the cumulative
// normalised weights of the gathered raw population.  All three only need the gathered partial vector.
__global__ void __launch_bounds__(256)
k_raycast_norm_chunks(GridDev g, const gms_beam *__restrict__ beams, int32_t B, uint32_t *__restrict__ cnt, int32_t *__restrict__ bbox,
                      int32_t nw_max, uint32_t n_ray_blocks, uint32_t n_norm_blocks,
                      const double *__restrict__ partials, int64_t nblk_global, double *__restrict__ w, const float *__restrict__ pose,
                      int32_t n, int64_t offset, const PackedParticle *__restrict__ glob_raw, int64_t n_global, int64_t nchunks,
                      double *__restrict__ cum, double *__restrict__ chunk_tot, double *__restrict__ p2,
                      PfStatsDev *__restrict__ stats, uint32_t n_chunk_blocks, double *__restrict__ logd,
                      uint32_t *__restrict__ cnt_pend, const int32_t *__restrict__ bbox_pend, uint32_t n_near_blocks) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (blockIdx.x >= n_ray_blocks + n_norm_blocks + n_chunk_blocks) {           // the previous scan's deferred apply pass (see k_norm_raycast)
        const uint32_t first = n_ray_blocks + n_norm_blocks + n_chunk_blocks;
        apply_body(g, logd, cnt_pend, bbox_pend, nullptr, blockIdx.x - first, 0, gridDim.x - first);
        return;
    }
    if (blockIdx.x < n_ray_blocks) {
        __shared__ RedLds L;
        __shared__ float s_pose[3];
        const int cols[4] = { COL_SUM, COL_XW, COL_YW, COL_TW };
        double f[4];
        fold_sums<4>(partials, nblk_global, cols, f, L);               // one round trip, one barrier pair
        if (threadIdx.x == 0) {
            s_pose[0] = (float)(f[1] / f[0]);                          // SLAM.java:176
            s_pose[1] = (float)(f[2] / f[0]);
            s_pose[2] = (float)(f[3] / f[0]);
        }
        __syncthreads();
        if (blockIdx.x >= n_ray_blocks - n_near_blocks)
            raycast_near_body(g, beams, B, B, nullptr, 0, cnt, bbox, blockIdx.x - (n_ray_blocks - n_near_blocks), 0, smem, s_pose);
        else
            raycast_body<false, RCF_RAYS, 4>(g, beams, B, B, nullptr, 0, nullptr, cnt, bbox, nullptr, nullptr, 0, nullptr, nw_max, blockIdx.x, 0,
                                             smem, s_pose, n_near_blocks ? 1 : 0);
    } else if (blockIdx.x < n_ray_blocks + n_norm_blocks) {
        normalize_own_body(partials, nblk_global, w, pose, n, offset, glob_raw, stats, blockIdx.x - n_ray_blocks);
    } else {
        chunk_sums_raw_body(glob_raw, n_global, nchunks, cum, chunk_tot, p2, nblk_global, partials,
                            blockIdx.x - n_ray_blocks - n_norm_blocks);
    }
}

// ---- C: likelihood rebuild (dirty tiles)  |  resample ---------------------------------------------------------
template <int KH, int SPLIT>
__global__ void __launch_bounds__(256)
k_lik_resample(GridDev g, const double *__restrict__ logd, double *__restrict__ lik, double *__restrict__ fac, int64_t fac_stride,
               const double *__restrict__ taps_g, const int32_t *__restrict__ bbox, int32_t tiles_x, int32_t tiles_y,
               const uint32_t *__restrict__ cnt_pending, uint8_t *__restrict__ tile_state, uint32_t n_res_blocks,
               const PackedParticle *__restrict__ glob, int64_t n_global, int64_t nchunks, const double *__restrict__ cum,
               const double *__restrict__ chunk_off, const double *__restrict__ r01_maps, double r01, double fraction,
               int32_t n, int64_t offset,
               float *__restrict__ pose2, float *__restrict__ cs2, double *__restrict__ w2, int32_t *__restrict__ idx_out,
               const double *__restrict__ p2, int64_t nblk_global, PfStatsDev *__restrict__ stats, int32_t raw_weights,
               int32_t *__restrict__ bbox_clear, int32_t lik_mode) {
    extern __shared__ __align__(16) unsigned char smem[];
    // the box half the next ray cast will raise: cleared here because that ray cast may share its launch with this scan's
    // deferred apply pass (k_raycast_apply), which otherwise does the clearing
    GMS_STAMP(GMS_STAMP_ROW(3, blockIdx.x), 0);
    if (blockIdx.x == 0 && threadIdx.x < 4) bbox_clear[4 * blockIdx.y + threadIdx.x] = 0;
    if (blockIdx.x < n_res_blocks)                      // a multiple of 8 keeps the likelihood tiles' XCD round-robin aligned
        resample_body(glob, n_global, nchunks, cum, chunk_off, r01_maps, r01, fraction, n, offset, pose2, cs2, w2, idx_out, p2,
                      nblk_global, stats, blockIdx.x, blockIdx.y, smem, raw_weights != 0);
    else
        likelihood_body<KH, SPLIT>(g, logd, lik, fac, fac_stride, taps_g, bbox, 1, tiles_x, tiles_y, blockIdx.x - n_res_blocks,
                                   blockIdx.y, gridDim.x - n_res_blocks, smem, cnt_pending, tile_state, lik_mode);
    GMS_STAMP(GMS_STAMP_ROW(3, blockIdx.x), blockIdx.x < n_res_blocks ? 1 : 2);
}

// ---- a recorded revolution's two preparations in one launch: de-skew of the raw scan  |  motion-model sample per particle ----
// (GridMapApp.java:143-175 | SLAM.java:90, Odometry.java:77-96: independent of one another)
__global__ void __launch_bounds__(256)
k_deskew_motion(const double *__restrict__ angle, const double *__restrict__ distance, const uint8_t *__restrict__ hit, int32_t length,
                double d_center, double d_theta, gms_beam *__restrict__ beams_out, uint32_t n_deskew_blocks,
                float *__restrict__ pose, float *__restrict__ cs, int32_t n, int64_t offset, double d_center_sd, double d_theta_sd,
                uint64_t seed, uint64_t sequence) {
    if (blockIdx.x < n_deskew_blocks)
        deskew_body(angle, distance, hit, length, d_center, d_theta, beams_out, (int32_t)(blockIdx.x * 256u + threadIdx.x));
    else
        motion_body(pose, cs, n, offset, d_center, d_theta, d_center_sd, d_theta_sd, seed, sequence, 0,
                    (int32_t)((blockIdx.x - n_deskew_blocks) * 256u + threadIdx.x));
}

// ---------------------------------------------------------------------------------------------
// launchers (single-map handles; the callers in gms_host.hip check the preconditions)
// ---------------------------------------------------------------------------------------------
bool gms_can_pair_launches(const gms_pf *pf, int32_t B) {
    const gms_map *m = pf->map;
    return pf->n_maps == 1 && B > 0 && B <= GMS_MAX_BEAMS && !m->need_full_build && m->pair_launches;
}

// normalise (SLAM.java:120-124) beside integrateObservation at the weighted pose (:93, GridMap.java:173-191)
void gms_launch_norm_raycast(gms_pf *pf, const double *d_partials, PackedParticle *d_packed_local, bool own,
                             const gms_beam *d_beams, int32_t B) {
    gms_map *m = pf->map;
    ProfScope ps(m, GMS_K_RAYCAST);
    if (own) { pf->d_global = pf->d_global_own; pf->global_raw = 0; }     // normalised weights are packed (as apply_partials does)
    const uint32_t n_near = rc_near_blocks(m, B);
    const uint32_t n_ray = (uint32_t)((B + RCF_RAYS - 1) / RCF_RAYS) + n_near, n_norm = (uint32_t)((pf->n + 255) / 256);
    const size_t smem = rc_smem(m, RCF_RAYS, n_near);
    // a deferred apply pass rides along: the ray cast then raises the OTHER box half (cleared by the previous likelihood launch)
    // while the pass reads the pending scan's half; afterwards that other half is the current one (gms_apply_done)
    uint32_t n_apply = 0;
    if (m->apply_pending) {
        const int32_t all = ((m->gd.W + APPLY_TW - 1) / APPLY_TW) * ((m->gd.H + APPLY_TH - 1) / APPLY_TH);
        n_apply = (uint32_t)(all < GMS_APPLY_BLOCKS_RIDING ? all : GMS_APPLY_BLOCKS_RIDING);
    }
    int32_t *pend = m->d_bbox + (size_t)m->bbox_cur * 4;
    int32_t *bb = n_apply ? m->d_bbox + (size_t)(1 - m->bbox_cur) * 4 : pend;
    const bool lognorm = gms_pf_lognorm_now(pf);
#define NR_LAUNCH(LN)                                                                                                                 \
    do {                                                                                                                              \
        if (smem > 48 * 1024)                                                                                                         \
            hipFuncSetAttribute(reinterpret_cast<const void *>(&k_norm_raycast<LN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
        hipLaunchKernelGGL(k_norm_raycast<LN>, dim3(n_ray + n_norm + n_apply), dim3(256), smem, m->stream, m->gd, d_beams, B, m->d_cnt, bb, \
                           rc_nw_max(m), n_ray, d_partials, nblk_global_of(pf), pf->d_w, pf->d_pose, pf->n, pf->offset, d_packed_local, \
                           own ? pf->d_cum : (double *)nullptr, own ? pf->d_chunk_tot : (double *)nullptr, nchunks_of(pf),          \
                           own ? pf->d_p2 : (double *)nullptr, pf->d_stats, n_norm, m->d_log, m->d_cnt_pend, pend, n_near,          \
                           LN ? (const double *)pf->d_logw : (const double *)nullptr);                                              \
    } while (0)
    if (lognorm) NR_LAUNCH(true); else NR_LAUNCH(false);
#undef NR_LAUNCH
    pf->score_fresh = 0;                                              // the scoring pass has been consumed
    if (n_apply) gms_apply_done(m);
    pf->chunks_ready = own ? 1 : 0;
    pf->neff_folded = 0;
}

// block partials (SLAM.java:100-115) beside the apply pass the previous paired step left pending
void gms_launch_partials_apply(gms_pf *pf, double *d_partials, bool apply_rides_later) {
    gms_map *m = pf->map;
    // apply_rides_later: the caller's next launch is gms_launch_norm_raycast, which takes the pending pass along (single maps)
    if (!m->apply_pending || apply_rides_later) { gms_launch_pf_partials(pf, d_partials); return; }
    ProfScope ps(m, GMS_K_REDUCE);
    const int64_t nblk = nblk_global_of(pf);
    const int32_t all = ((m->gd.W + APPLY_TW - 1) / APPLY_TW) * ((m->gd.H + APPLY_TH - 1) / APPLY_TH);
    const uint32_t n_apply = (uint32_t)(all < GMS_APPLY_BLOCKS ? all : GMS_APPLY_BLOCKS);
    int32_t *cur = m->d_bbox + (size_t)m->bbox_cur * m->n_maps * 4, *idle = m->d_bbox + (size_t)(1 - m->bbox_cur) * m->n_maps * 4;
    PfStatsDev *lognorm_stats = gms_pf_lognorm_now(pf) ? pf->d_stats : (PfStatsDev *)nullptr;
    hipLaunchKernelGGL(k_partials_apply, dim3((uint32_t)nblk + n_apply, pf->n_maps), dim3(256), 0, m->stream, pf->d_w, pf->d_logw,
                       pf->d_pose, pf->n, pf->offset, nblk, d_partials,
                       pf->pending_nseg ? (const double *)pf->d_part : (const double *)nullptr, pf->pending_nseg, m->gd, m->d_log,
                       m->d_cnt_pend, cur, idle, lognorm_stats);
    pf->pending_nseg = 0;
    gms_apply_done(m);
}

// computeLikelihoodMap on the touched tiles (GridMap.java:233-250) beside resample() (SLAM.java:133-153)
void gms_launch_lik_resample(gms_pf *pf, double fraction) {
    gms_map *m = pf->map;
    gms_launch_pf_chunk_sums(pf);                     // no-op when level 0 is already there
    ProfScope ps(m, GMS_K_LIKELIHOOD);
    const int32_t k = m->lik_kh;
    const int32_t tiles_x = (m->gd.W + LK_TW - 1) / LK_TW, tiles_y = (m->gd.H + LK_TH - 1) / LK_TH;
    const size_t smem_l = gms_likelihood_lds_bytes(m->gd.khalf, k != 0);
    int32_t blocks = tiles_x * tiles_y;
    const int32_t cap = gms_likelihood_blocks_cap(m, smem_l);
    if (blocks > cap) blocks = cap;
    blocks = (blocks + 7) & ~7;
    const int64_t nch = nchunks_of(pf);
    const size_t smem_r = gms_resample_lds_bytes(nch);
    const size_t smem = smem_l > smem_r ? smem_l : smem_r;
    const uint32_t n_res = (uint32_t)((pf->n + 255) / 256);
    const int32_t *bb = m->d_bbox + (size_t)m->bbox_cur * m->n_maps * 4;
    const double *r01_maps = pf->n_maps == 1 ? (const double *)nullptr : pf->d_r01_src;
    int32_t lik_mode = m->lik_lazy ? 2 : 3;                   // the factor table only: likelihoodData follows on demand (gms_ensure_lik)
    if (m->lik_lazy) m->lik_stale = 1;
    if (m->fac_current && m->lik_skip) lik_mode |= 4;         // tiles whose codes this scan does not change are left alone (likelihood_body)
    m->fac_current = 1;
    const bool split = k != 0 && gms_likelihood_split(m, blocks);              // (likelihood_body, "split")
#define LR_LAUNCH(KH, SP)                                                                                                 \
    do {                                                                                                                  \
        if (smem > 48 * 1024)                                                                                             \
            hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lik_resample<KH, SP>),                                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                                   \
        hipLaunchKernelGGL((k_lik_resample<KH, SP>), dim3((uint32_t)blocks + n_res, pf->n_maps), dim3(256), smem, m->stream, m->gd, m->d_log, \
                           m->d_lik, m->d_fac, m->fac_stride, m->d_taps, bb, tiles_x, tiles_y, m->d_cnt, m->d_tile_state, n_res, pf->d_global, \
                           pf->n_global, nch, pf->d_cum, pf->d_chunk_tot, r01_maps, pf->r01_scalar, fraction, pf->n, pf->offset, \
                           pf->d_pose2, pf->d_cs2, pf->d_w2, pf->d_idx, pf->d_p2, nblk_global_of(pf), pf->d_stats,       \
                           pf->global_raw, m->d_bbox + (size_t)(1 - m->bbox_cur) * m->n_maps * 4, lik_mode);             \
    } while (0)
    if (k == 3) { if (split) LR_LAUNCH(3, 2); else LR_LAUNCH(3, 1); }
    else if (k == 5) { if (split) LR_LAUNCH(5, 2); else LR_LAUNCH(5, 1); }
    else LR_LAUNCH(0, 1);
#undef LR_LAUNCH
    pf->neff_folded = 1;
}

// ---- sharded filters, one all-gather per scan ----------------------------------------------------------------------
// this shard's partials at their global slots of d_partials and its raw pack at its slot of d_global_own (the payloads
// of the two all-gathers, both in place), beside the apply pass a previous paired step left pending
void gms_launch_partials_pack_apply(gms_pf *pf, bool apply_rides_later) {
    gms_map *m = pf->map;
    ProfScope ps(m, GMS_K_REDUCE);
    const uint32_t n_local = (uint32_t)((pf->n + GMS_BLOCK - 1) / GMS_BLOCK);
    uint32_t n_apply = 0;
    if (m->apply_pending && !apply_rides_later) {      // (rides later: beside this scan's ray cast, gms_launch_raycast_norm_chunks)
        const int32_t all = ((m->gd.W + APPLY_TW - 1) / APPLY_TW) * ((m->gd.H + APPLY_TH - 1) / APPLY_TH);
        n_apply = (uint32_t)(all < GMS_APPLY_BLOCKS ? all : GMS_APPLY_BLOCKS);
    }
    int32_t *cur = m->d_bbox + (size_t)m->bbox_cur * 4, *idle = m->d_bbox + (size_t)(1 - m->bbox_cur) * 4;
    hipLaunchKernelGGL(k_partials_pack_apply, dim3(n_local + n_apply), dim3(256), 0, m->stream, pf->d_w, pf->d_logw, pf->d_pose,
                       pf->n, pf->offset, nblk_global_of(pf), pf->d_partials,
                       pf->pending_nseg ? (const double *)pf->d_part : (const double *)nullptr, pf->pending_nseg,
                       pf->d_global_own + pf->offset, n_local, m->gd, m->d_log, m->d_cnt_pend, cur, idle);
    pf->pending_nseg = 0;
    pf->score_fresh = 0;               // the scoring pass has been consumed (raw weights packed for the exchange)
    if (n_apply) gms_apply_done(m);
}

// after the gathers: ray cast (optional) | normalise own + statistics | cumulative sums of the gathered raw population
void gms_launch_raycast_norm_chunks(gms_pf *pf, const gms_beam *d_beams, int32_t B, bool raycast) {
    gms_map *m = pf->map;
    ProfScope ps(m, GMS_K_RAYCAST);
    pf->d_global = pf->d_global_own;
    pf->global_raw = 1;
    const uint32_t n_near = raycast ? rc_near_blocks(m, B) : 0u;
    const uint32_t n_ray = raycast ? (uint32_t)((B + RCF_RAYS - 1) / RCF_RAYS) + n_near : 0u, n_norm = (uint32_t)((pf->n + 255) / 256);
    const uint32_t n_chunk = (uint32_t)nblk_global_of(pf);
    const size_t smem = rc_smem(m, RCF_RAYS, n_near);
    // a deferred apply pass rides beside the ray cast, as in gms_launch_norm_raycast
    uint32_t n_apply = 0;
    if (raycast && m->apply_pending) {
        const int32_t all = ((m->gd.W + APPLY_TW - 1) / APPLY_TW) * ((m->gd.H + APPLY_TH - 1) / APPLY_TH);
        n_apply = (uint32_t)(all < GMS_APPLY_BLOCKS_RIDING ? all : GMS_APPLY_BLOCKS_RIDING);
    }
    int32_t *pend = m->d_bbox + (size_t)m->bbox_cur * 4;
    int32_t *bb = n_apply ? m->d_bbox + (size_t)(1 - m->bbox_cur) * 4 : pend;
    if (smem > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_raycast_norm_chunks), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem);
    hipLaunchKernelGGL(k_raycast_norm_chunks, dim3(n_ray + n_norm + n_chunk + n_apply), dim3(256), smem, m->stream, m->gd, d_beams, B,
                       m->d_cnt, bb, rc_nw_max(m), n_ray, n_norm, pf->d_partials, nblk_global_of(pf), pf->d_w, pf->d_pose, pf->n,
                       pf->offset, pf->d_global_own, pf->n_global, nchunks_of(pf), pf->d_cum, pf->d_chunk_tot, pf->d_p2, pf->d_stats,
                       n_chunk, m->d_log, m->d_cnt_pend, pend, n_near);
    if (n_apply) gms_apply_done(m);
    pf->chunks_ready = 1;
    pf->neff_folded = 0;
}

// de-skew of one raw scan beside the motion-model sample of a single-map filter's particles
void gms_launch_deskew_motion(gms_pf *pf, const double *d_angle, const double *d_distance, const uint8_t *d_hit, int32_t length,
                              double d_center, double d_theta, uint64_t seed, uint64_t sequence) {
    gms_map *m = pf->map;
    const double d_center_sd = (0.01 + fabs(d_center) * 0.05) / 2;               // Odometry.java:63
    const double d_theta_sd = 5 * (3.141592653589793 / 180.0) + 0.1 * fabs(d_theta);   // :64
    const uint32_t n_dk = (uint32_t)((length + 255) / 256), n_mo = (uint32_t)((pf->n + 255) / 256);
    hipLaunchKernelGGL(k_deskew_motion, dim3(n_dk + n_mo), dim3(256), 0, m->stream, d_angle, d_distance, d_hit, length, d_center,
                       d_theta, m->d_beams, n_dk, pf->d_pose, pf->d_cs, pf->n, pf->offset, d_center_sd, d_theta_sd, seed, sequence);
}
