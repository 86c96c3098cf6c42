// gms_internal.h -- shared between the C-ABI host code and the gfx950 kernels of libgridmapslam.so.
// Not part of the public interface (that is include/gridmapslam.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "gridmapslam.h"

// ---- device-side view of a GridMap (J/slam/GridMap.java fields, widened once on the host) -------
struct GridDev {
    int32_t W, H;
    int64_t cells;        // W*H
    double posx, posy;    // (double) position.x / .y
    double res;           // (double) resolution
    double rinv;          // RN(1.0 / res), for j_cell_of
    float resf;           // resolution
    double l_free, l_occ; // log-odds increments
    int32_t extra;        // RayIterator additionalSteps
    float half_tol;       // hitTolerance / 2 (float arithmetic)
    double z_hit;         // zHit
    double c_rand;        // zRandom * 1.0 / SENSOR_MAX_RANGE
    double inv_max;       // 1.0 / SENSOR_MAX_RANGE
    int32_t ktaps, khalf;
    int32_t fpitch;       // factor table: row pitch W + 16; column W of every row and all of row H hold the neutral 1.0 (fac_index)
    uint32_t fneutral;    // factor table: index of one neutral entry (row H, column W)
    uint32_t *tile_stats; // [64][4] likelihood-tile census (gms_map_tile_stats) or nullptr: {left alone, constants kept, constants written, blurred}
};

// one ray of a scan in grid coordinates (GridMap.integrateObservation's locals)
struct RayIn {
    float sx, sy, ex, ey, measured;
    int32_t hit;
};

// device-resident statistics of one particle set (one per map)
struct PfStatsDev {
    double weight_sum;    // sum of raw weights
    double norm_sum;      // sum of normalised weights (calculateNeff's `sum`)
    double sq_sum;        // sum((w/norm_sum)^2)
    double xs, ys, ts;    // getWeightedPose numerators
    double max_w;         // largest raw weight
    double max_logw;
    int32_t strongest;
    int32_t n_zero;
    float wpose[3];       // weighted pose
    float spose[3];       // strongest particle's pose
    int32_t did_resample;
    int32_t n_ambiguous;
};

#define GMS_SCORE_MAXSEG 32
#ifndef GMS_SCORE_SEGLEN
#define GMS_SCORE_SEGLEN 45        // beams per segment product of the default scoring kernel, scans of more than GMS_SCORE_SHORT_SCAN beams
#endif
#define GMS_SCORE_SEGLEN_SHORT 12  // ... and of shorter scans (see gms_launch_pf_score)
#define GMS_SCORE_SHORT_SCAN 384
#define GMS_SCORE_SEGLEN_LONG 90   // ... and of scans of more than GMS_SCORE_LONG_SCAN beams
#define GMS_SCORE_LONG_SCAN 768
#define GMS_SCORE_SEGLEN_LONG_BATCHED 120   // ... on batched handles (which are never sharded)
static_assert(128 * GMS_SCORE_MAXSEG >= GMS_MAX_BEAMS, "a scan must fit GMS_SCORE_MAXSEG segments of 128 beams");
#define GMS_PARTIAL_STRIDE 9   // per block: sum w, max w, first argmax, n_zero, max logw, sum w^2, sum x*w, sum y*w, sum th*w

// packed particle exchanged by the all-gather (24 B)
struct PackedParticle {
    double w;
    float x, y, theta;
    uint32_t pad;
};

struct ProfSlot {
    hipEvent_t a, b;
    int32_t k;
};

// Pinned host staging for inputs that arrive as host buffers: a small ring, so that copying the next scan's inputs
// never waits for the stream to drain -- only (rarely) for the copy that last used the same slot.
#define GMS_STAGE_SLOTS 4
struct StageRing {
    void *slot[GMS_STAGE_SLOTS] = {};
    hipEvent_t ev[GMS_STAGE_SLOTS] = {};
    bool busy[GMS_STAGE_SLOTS] = {};
    int32_t next = 0;
};

struct gms_map {
    gms_params prm;
    GridDev gd;
    int32_t n_maps;
    int32_t device;
    int32_t n_cus;            // compute units of the device (hipDeviceAttributeMultiprocessorCount; 256 on MI355X)
    int32_t lds_per_cu;       // bytes of LDS per compute unit (160 KiB on MI355X)
    int32_t max_beams;
    int32_t n_filters;        // live gms_pf handles bound to this map (gms_map_destroy refuses while > 0)
    hipStream_t own_stream;
    hipStream_t stream;
    double *d_log;        // [n_maps][H][W]
    double *d_lik;        // [n_maps][H][W]
    double *d_fac;        // [n_maps][fac_stride]: per-cell scoring factor f(likelihood) (GridMap.java:285-288), rows of g.fpitch entries with a neutral border (fac_index),
                          // kept in step with d_lik; entry [cells] of each map is the neutral factor 1.0
    int64_t fac_stride;   // (H + 1) * fpitch
    uint32_t *d_cnt;      // [n_maps][H][W] per-scan packed counts (n_free | n_occ << 16): the grid the NEXT ray cast accumulates into, all zero between scans
    uint32_t *d_cnt_pend; // the second grid: while apply_pending, the counts of the scan whose apply pass is deferred (the two swap roles
                          // when a scan's apply is deferred, so that the next ray cast can share a launch with that apply pass)
    int32_t *d_bbox;      // [2][n_maps][4] encoded box of the cells changed since the last likelihood build;
                          // double-buffered: k_apply clears the idle half, so no memset is ever queued
    int32_t bbox_cur;     // half in use
    int32_t bbox_dirty;   // an integrate ran since the last likelihood build
    double *d_taps;       // [ktaps]
    uint8_t *d_tile_state;  // [n_maps][likelihood tiles][2 halves of a tile's rows]: 0 unknown, 1..3 the tile of d_lik/d_fac holds the constants of a uniform tile of code 0 / 0.5 / 1
    uint32_t *d_tile_stats; // [64][4] counters behind gd.tile_stats (always allocated; gd.tile_stats points at them while the census is on)
    gms_beam *d_beams;    // [n_maps][max_beams] staging
    float *d_poses;       // [n_maps][3] staging
    double *d_scratch;    // small device scratch
    int32_t need_full_build;  // likelihood field must be rebuilt everywhere (upload/reset/copy)
    int32_t apply_pending;    // the last scan's counts are not in logData yet (deferred apply pass, gms_flush_apply)
    int32_t raycast_tile;     // batched ray casts accumulate in LDS tiles (k_raycast_tile; GMS_RAYCAST_TILE=0 turns it off)
    int32_t raycast_tile_min; // ... when the launch has more rays than this in all (default 4096; GMS_RAYCAST_TILE_MIN)
    int32_t taps_plain;       // every tap is +0.0 or in [2^-900, 2^900]: sums of tap * {0, 1, 2} scale exactly by 0.5 and a tap * 0.0 may stand for a skipped one
    int32_t lik_kh;           // the likelihood kernels' compile-time half width (3 or 5), or 0 = the generic path: another kernel size, or a tap
                              // that is negative or outside 2^-900 .. 2^900 (the fast path computes twice the horizontal sums and halves them --
                              // exact only while nothing is subnormal -- and starts a sum with its first product instead of 0.0 + it -- the
                              // same bits only while that product is not -0.0; likelihood_body)
    int32_t lik_split;        // dirty-tile rebuilds of a single map give every tile two workgroups when the chip has them to spare (GMS_LIK_SPLIT=0: one)
    int32_t lik_lazy;         // scan steps' dirty-tile rebuilds write the factor table only, likelihoodData on demand (GMS_LIK_LAZY=0 turns it off)
    int32_t lik_stale;        // likelihoodData is behind the factor table somewhere (gms_ensure_lik brings it up to date)
    int32_t fac_current;      // the factor table is the field of logData + the pending counts as of the last rebuild, and logData has not moved since except by those counts
    int32_t lik_skip;         // dirty-tile rebuilds leave tiles alone whose codes the scan does not change (GMS_LIK_SKIP=0 turns it off; same bits)
    int32_t raycast_near;     // single-map ray casts: the first 64 steps of every ray go through near-field workgroups with an LDS tile (0 never, 1 for scans of 512 beams or more, 2 for every scan of 32 or more: GMS_RAYCAST_NEAR=0 / unset / 1)
    int32_t pair_launches;    // scan steps pair independent kernels in one launch (GMS_PAIR_LAUNCHES=0 turns it off)
    int32_t slam_threads;     // per-particle maps: lanes per workgroup of k_slam_particle (0 = the launcher decides; GMS_SLAM_THREADS = 512 / 1024)
    int32_t slam_tile_cells;  // per-particle maps: cap on the LDS count tile of k_slam_particle in cells (0 = what the LDS allows; GMS_SLAM_TILE_CELLS, for tests of the band walk)
    gms_beam *h_beams;    // pinned staging (de-skew inputs, single-ray entry)
    StageRing beam_ring;  // pinned staging of scans handed over as host buffers
    float *h_poses;       // pinned staging
    hipEvent_t pose_copy_ev;  // the last copy out of h_poses
    int32_t pose_copy_ev_set;
    int32_t *d_trace_cells; uint8_t *d_trace_cls; int32_t *d_trace_cnt;
    size_t trace_cap_cells, trace_cap_counts;   // capacities of d_trace_cells/d_trace_cls (cells) and d_trace_cnt (counts)
    // profiling
    int32_t prof_on;
    int32_t prof_stride;      // every prof_stride-th launch of an enabled class is bracketed (>= 1)
    int64_t prof_seen[GMS_K_COUNT];
    std::vector<ProfSlot> prof_pending;
    std::vector<ProfSlot> prof_free;
    double prof_ms[GMS_K_COUNT];
    int64_t prof_n[GMS_K_COUNT];
};

struct gms_pf {
    gms_map *map;
    int32_t n;            // particles held here (per map)
    int64_t offset;       // global index of particle 0
    int64_t n_global;
    int32_t n_maps;
    float *d_pose;                  // [n_maps][n][3] current poses x,y,theta (as at the boundary)
    float *d_pose2;                 // resample double buffer
    float *d_cs2;                   // ... and its trig
    double *d_w, *d_w2;             // [n_maps][n] weights
    double *d_logw, *d_logw2;       // [n_maps][n] sum(log factor)
    float *d_cs;                    // [n_maps][n][2] float-rounded cos/sin of theta
    double *d_hitbeams;             // [n_maps][max_beams][2] compacted hit beams
    double *d_part;                 // [n_maps][GMS_SCORE_MAXSEG][n] per-segment products (k_score_c)
    int32_t *d_nhit;                // [n_maps]
    double *d_partials;             // [n_maps][nblk_global][GMS_PARTIAL_STRIDE]
    double *d_p2;                   // [n_maps][nblk_global][2] {sum wn, sum wn^2} of the normalised global population
    int32_t neff_folded;            // stats.sq_sum has been folded from d_p2 (d_p2 is produced with the chunk sums)
    int32_t global_raw;             // d_global holds RAW weights (gathered before the normalisation): resample divides
    PackedParticle *d_global_own;   // the library's own buffer; d_global may alias a caller's all-gather result
    PackedParticle *d_global;       // [n_maps][n_global] source population (own copy when unsharded)
    double *d_chunk_tot;            // [n_maps][nchunks] scan chunk totals / offsets
    double *d_cum;                  // [n_maps][n_global] in-chunk inclusive sums
    PfStatsDev *d_stats;            // [2][n_maps]: [0] of the last normalise, [1] of the current particles (recomputed on demand)
    int32_t stats_current;          // d_stats[0] still describes the current particles
    PfStatsDev *h_stats;            // pinned
    double *d_r01;                  // [n_maps] (unused since the kernels read the pinned ring slot; kept for the allocation's alignment slack)
    const double *d_r01_src;        // [n_maps] where the resample kernel reads the draws: the current pinned ring slot (batched maps)
    double r01_scalar;
    int32_t *d_idx;                 // [n_maps][n]
    float *h_stage;                 // pinned staging for poses (read-back)
    StageRing pose_ring;            // pinned staging of pose proposals handed over as host buffers
    StageRing r01_ring;             // pinned staging of the per-map resampling draws (n_maps > 1)
    int32_t have_global;            // d_global holds the current normalised population
    int32_t chunks_ready;           // d_cum / d_chunk_tot hold level 0 of the scan of d_global
    int32_t refine;                 // scan steps run findBestPose on every particle before weighting (gms_pf_set_refine)
    int32_t pending_nseg;           // > 0: d_w is stale, the weights are still d_part's segment products
    float4 *d_ord;                  // [n_maps][n] {x, y, cos, sin} of the particles in locality order (k_order)
    int32_t *d_perm;                // [n_maps][n] the particle at each position of that order
    int32_t log_norm;               // gms_pf_set_log_normalize: weights = exp(logw - max logw) instead of the plain product (stand-alone filters)
    int32_t score_fresh;            // d_w / d_logw (or the segment products) come from a scoring pass nothing has consumed yet
    int32_t score_threads;          // 0 the launcher decides (the largest of 1024 / 512 / 256 lanes per scoring workgroup that still gives every CU one); GMS_SCORE_THREADS forces 64..1024
    int32_t reference_order;        // gms_pf_set_reference_order: the audit path -- every re-associated chain (the scan's product, weightSum, the
                                    // cumulative weights) as ONE chain in the reference's order (tests; slow)
    int32_t score_spread;           // -1 the launcher decides (launches of two or more workgroups per CU), 0 / 1 forced (GMS_SCORE_SPREAD, read at creation)
    int32_t order_mode;             // -1 the launcher decides (large launches only), 0 never, 1 always (GMS_SCORE_ORDER; results do not depend on it)
    int32_t slam_owned;             // the filter of a gms_slam: its particles own maps, so resampling, sharding and the shared-map scan steps are refused on it
    int32_t *d_epoch2;              // the filter of a gms_slam, during gms_slam_resample_maps[_if]: {draws that ran so far, the last resample() drew}, kept by the resampling kernels (NULL otherwise)
};

// one rank's side of the RCCL exchanges of a sharded filter (gms_host.hip)
struct gms_comm {
    void *nccl = nullptr;           // ncclComm_t
    int32_t rank = 0, world = 1, device = 0;
    int32_t overlap = 0;            // all-gather on the side stream (default: world > 2; GMS_COMM_OVERLAP=0/1 overrides)
    int32_t pending = 0;            // an all-gather is in flight
    int32_t broken = 0;             // an exchange failed: every later call on this communicator fails fast
    int32_t p2p = 0;                // GMS_EXCHANGE=p2p: the scan's exchange as grouped ncclSend/ncclRecv instead of all-gathers
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};

// Every particle's GridMapData of a gms_slam, both generations, as the kernels see it.  WHICH generation is current is a device-side
// fact: resample() writes its deep copies into the other one, and `if (neff < n / 2) resample()` (GridMapApp.java:185-186) is decided on
// the device, so the host cannot know without a round trip.  epoch[0] counts the draws that ran (kept by the resampling kernels,
// gms_pf::d_epoch2): its parity is the current generation; epoch[1] says whether the last resample() drew.
struct SlamBufs {
    double *log[2], *lik[2];        // [n][H][W]
    uint32_t *code[2];              // [n][2][code_words] class planes, or NULL
    int32_t *epoch;
};

// SLAM as the reference has it (J/slam/SLAM.java): N particles, each with its own GridMapData (gms_slam_host.hip, gms_slam_kernels.hip)
struct gms_slam {
    gms_map *map;                   // ONE map's worth of handle: the GridMap (geometry, constants, taps), the stream, staging, profiling; its own
                                    // logData / likelihoodData receive the combined map (gms_slam_combined, GridMapApp.calculateCombined)
    gms_pf *pf;                     // the N particles' poses, weights, statistics and resampling indices (one "map" of N particles)
    int32_t n;
    double *d_log[2], *d_lik[2];    // [n][H][W] every particle's GridMapData, double-buffered for resample()'s deep copies
    int32_t *d_epoch;               // {draws that ran so far, the last resample() drew}: the current generation is d_epoch[0] & 1 (SlamBufs)
    int64_t copies_base;            // maps copied by resampling steps before the last reset (the rest: d_epoch[0] * n)
    int32_t *d_plan;                // a shard's resample(): [3][n] device staging of {export list | local sources | positions in the received buffer}
    int32_t lazy_lik;               // resample() copies logData at once and likelihoodData when somebody asks for it: the next update's
                                    // computeLikelihoodMap overwrites every cell of it before anything on the path reads one (GMS_SLAM_LAZY_LIK_COPY=0: both at once)
    int32_t lik_behind;             // the current generation's likelihoodData does not hold the last resample()'s copies yet (if it drew): slot m's field is the other generation's [d_idx_lik[m]]
    int32_t *d_idx_lik;             // [n] the source indices of that resample()
    uint32_t *d_code[2];            // [n][2][code_words] every particle's class planes (gms_slam_kernels.hip), double-buffered with logData; NULL: not kept
                                    // (the blur kernel is wider than the on-demand evaluation takes, the plane does not fit the LDS, or GMS_SLAM_EAGER_LIK=1)
    int64_t code_words;             // 32-bit words per plane
    int32_t lik_from_codes;         // likelihoodData is behind: every particle's is the field of plane 1 of its class planes (made on demand)
    int32_t refine;                 // gms_slam_set_refine: update() runs findBestPose on every particle against its own field before weighting it (SLAM.java:96)
    int32_t refine_field;           // the field in front of the refinement: -1 from the class plane where logData exceeds the infinity cache, 0 from logData
                                    // always, 1 from the plane always (GMS_SLAM_REFINE_FIELD=log|codes: tests of both forms)
    int32_t refine_lds;             // -1 the field in LDS whenever it fits (computed there from the class plane where it can be), 0 never, 2 staged from
                                    // memory wherever it fits (GMS_SLAM_REFINE_LDS: tests of the other forms)
};

// the thread's last-error text + code (gms_host.hip); every C-ABI file reports through it
int gms_fail(int code, const char *fmt, ...);
// host beams [n_maps][B] -> the map's device staging buffer [n_maps][max_beams] through the pinned ring (gms_host.hip)
int gms_stage_beams(gms_map *m, const gms_beam *beams, int32_t B);

// ---- kernel launchers (gms_map_kernels.hip / gms_pf_kernels.hip) -----------------------------

void gms_launch_raycast(gms_map *m, const gms_beam *d_beams, int32_t B, int32_t beam_stride, const float *d_poses,
                        int32_t pose_stride, bool take_pending_apply = false);
bool gms_raycast_tiled(const gms_map *m, int32_t B);
void gms_launch_trace_scan(gms_map *m, const gms_beam *d_beams, int32_t B, const float *d_pose,
                           int32_t *d_cells, uint8_t *d_cls, int32_t cap, int32_t *d_counts);
void gms_launch_trace_ray(gms_map *m, float x0, float y0, float x1, float y1, int32_t extra,
                          int32_t *d_cells, int32_t cap, int32_t *d_count);
void gms_launch_apply_ray(gms_map *m, RayIn ray);
void gms_launch_apply_counts(gms_map *m);
void gms_launch_likelihood(gms_map *m, int32_t dirty_only, bool counts_pending = false, bool materialize = false);
void gms_ensure_lik(gms_map *m);        // likelihoodData up to date everywhere (the scan steps' rebuilds write the factor table only)
bool gms_likelihood_split(const gms_map *m, int32_t blocks);      // likelihood_body's SPLIT = 2 for a dirty-tile rebuild launched with `blocks` workgroups?
size_t gms_likelihood_lds_bytes(int32_t khalf, bool coded = true);   // coded: the byte-coded staging of the compile-time half widths (gms_map::lik_kh != 0)
int32_t gms_likelihood_blocks_cap(const gms_map *m, size_t smem);
void gms_launch_raycast_apply(gms_map *m, const gms_beam *d_beams, int32_t B, int32_t beam_stride, const float *d_poses, int32_t pose_stride);
void gms_defer_apply(gms_map *m);       // host bookkeeping: the scan just cast keeps its counts until a later launch applies them
void gms_launch_fill(gms_map *m, double *d, double v, int64_t n);
void gms_launch_combine(gms_map *src, gms_map *dst);
void gms_launch_deskew(gms_map *m, const double *d_angle, const double *d_distance, const uint8_t *d_hit, int32_t length,
                       double d_center, double d_theta, gms_beam *d_out);
void gms_launch_factors(gms_map *m);   // d_fac from d_lik (after an upload / copy)
void gms_launch_noop(gms_map *m);
bool gms_set_stamp_buffer(gms_map *m, void *dev_buffer);
void gms_launch_spin(gms_map *m, double us);
void gms_launch_copy(gms_map *m, void *dst, const void *src, size_t nbytes);   // src may be pinned host memory
void gms_invalidate_tile_state(gms_map *m);
void gms_launch_get_raw(gms_map *m, int32_t mi, int32_t x, int32_t y, double *d_out2);
void gms_launch_debug_f32(gms_map *m, int32_t op, const float *d_a, float *d_out, int64_t n);

void gms_launch_pf_init(gms_pf *pf);
void gms_launch_pf_pose_trig(gms_pf *pf, const float *d_src);
void gms_launch_pf_combine(gms_pf *pf);
void gms_launch_pf_motion(gms_pf *pf, double d_center, double d_theta, uint64_t seed, uint64_t sequence);
void gms_launch_pf_after_gather(gms_pf *pf);
void gms_launch_pf_chunk_sums(gms_pf *pf);
// paired launches (gms_fused_kernels.hip)
void gms_flush_apply(gms_map *m);
void gms_apply_done(gms_map *m);
void gms_launch_partials_apply(gms_pf *pf, double *d_partials, bool apply_rides_later = false);
bool gms_can_pair_launches(const gms_pf *pf, int32_t B);
void gms_launch_norm_raycast(gms_pf *pf, const double *d_partials, PackedParticle *d_packed_local, bool own,
                             const gms_beam *d_beams, int32_t B);
void gms_launch_lik_resample(gms_pf *pf, double fraction);
void gms_launch_deskew_motion(gms_pf *pf, const double *d_angle, const double *d_distance, const uint8_t *d_hit, int32_t length,
                              double d_center, double d_theta, uint64_t seed, uint64_t sequence);
void gms_launch_partials_pack_apply(gms_pf *pf, bool apply_rides_later = false);
void gms_launch_raycast_norm_chunks(gms_pf *pf, const gms_beam *d_beams, int32_t B, bool raycast);
void gms_launch_pf_fold_neff(gms_pf *pf);
struct MotionModel {            // one odometry step for gms_launch_pf_score's motion-model sample (Odometry.java:60-96)
    double d_center, d_theta;
    uint64_t seed, sequence;
};
void gms_launch_pf_score(gms_pf *pf, const gms_beam *d_beams, int32_t B, int32_t beam_stride, const float *d_pose_src = nullptr,
                         const MotionModel *motion = nullptr);   // d_pose_src: set the poses in the same launch
void gms_launch_pf_partials(gms_pf *pf, double *d_partials);
bool gms_pf_lognorm_now(const gms_pf *pf);
void gms_launch_pf_pack(gms_pf *pf, PackedParticle *d_packed);
void gms_launch_pf_apply_partials(gms_pf *pf, const double *d_partials, PackedParticle *d_packed_local, bool own);
void gms_launch_pf_stats_only(gms_pf *pf, const double *d_partials, PfStatsDev *d_stats_out);
void gms_launch_pf_resample(gms_pf *pf, double fraction /* <0: unconditional */);
void gms_launch_pf_refine(gms_pf *pf, const gms_beam *d_beams, int32_t B, int32_t beam_stride);
void gms_launch_pf_normalize_seq(gms_pf *pf, PfStatsDev *d_stats_out, bool normalise);
void gms_launch_pf_resample_seq(gms_pf *pf, double fraction);
// one GridMapData per particle (gms_slam_kernels.hip); the buffers' current generation is read on the device (SlamBufs)
SlamBufs gms_slam_bufs(const gms_slam *s);
void gms_launch_slam_likelihood(gms_map *m, const SlamBufs &sb, int32_t n);
void gms_launch_slam_particle(gms_pf *pf, const gms_beam *d_beams, int32_t B, const SlamBufs &sb, bool field_in_memory, const MotionModel *motion,
                              int32_t integrate, int64_t code_words);
void gms_launch_slam_likelihood_codes(gms_map *m, const SlamBufs &sb, int64_t code_words, int32_t n, int32_t plane);
void gms_launch_slam_codes_from_log(gms_map *m, const SlamBufs &sb, int32_t first, int32_t count, int64_t code_words);
int64_t gms_slam_code_words(int64_t cells);
void gms_launch_slam_trace(gms_pf *pf, const gms_beam *d_beams, int32_t B, int32_t particle, int32_t *d_cells, uint8_t *d_cls, int32_t cap, int32_t *d_counts);
bool gms_launch_slam_refine(gms_pf *pf, const gms_beam *d_beams, int32_t B, const SlamBufs &sb, const MotionModel *motion, int32_t field_in_lds,
                            int64_t code_words);
bool gms_slam_refine_from_planes(const gms_map *m, int32_t B, int32_t field_in_lds, int64_t code_words);
// resample()'s copies into the generation the draw has just made current, where it drew (epoch[1]); what: bit 0 logData (+ the class
// planes), bit 1 likelihoodData; d_idx_keep (may be NULL) receives the indices for a likelihoodData copy that is still owed
void gms_launch_slam_gather(gms_pf *pf, const SlamBufs &sb, int32_t what, const int32_t *d_idx, int32_t *d_idx_keep, int64_t code_words);
void gms_launch_slam_combine(gms_map *dst, const SlamBufs &sb, int32_t n);
void gms_launch_slam_export_records(gms_pf *pf, const SlamBufs &sb, const int32_t *d_list, int32_t count, int64_t code_words, double *d_dst);
void gms_launch_slam_shard_gather(gms_pf *pf, const SlamBufs &sb, const int32_t *d_src_local, const int32_t *d_recv_pos, const double *d_recv, int64_t code_words);

// profiling brackets
void gms_prof_begin(gms_map *m, int32_t k);
void gms_prof_end(gms_map *m);

struct ProfScope {
    gms_map *m;
    bool on;
    ProfScope(gms_map *mm, int32_t k) : m(mm), on(((mm->prof_on >> k) & 1) && (mm->prof_seen[k]++ % mm->prof_stride) == 0) {
        if (on) gms_prof_begin(m, k);
    }
    ~ProfScope() { if (on) gms_prof_end(m); }
};
