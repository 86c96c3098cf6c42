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
    float resf;           // resolution
    double l_free, l_occ; // log-odds increments
    int32_t extra;        // RayIterator additionalSteps
    float half_tol;       // hitTolerance / 2 (float arithmetic)
    double z_hit;         // zHit
    double c_rand;        // zRandom * 1.0 / SENSOR_MAX_RANGE
    double inv_max;       // 1.0 / SENSOR_MAX_RANGE
    int32_t ktaps, khalf;
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

#define GMS_PARTIAL_STRIDE 5   // {sum, max, first-argmax index, n_zero, max_logw} per block

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

struct gms_map {
    gms_params prm;
    GridDev gd;
    int32_t n_maps;
    int32_t device;
    int32_t max_beams;
    hipStream_t own_stream;
    hipStream_t stream;
    double *d_log;        // [n_maps][H][W]
    double *d_lik;        // [n_maps][H][W]
    uint32_t *d_cnt;      // [n_maps][H][W] per-scan packed counts, zero between calls
    int32_t *d_bbox;      // [n_maps][4] xmin,ymin,xmax,ymax of the cells the last scan changed
    double *d_taps;       // [ktaps]
    gms_beam *d_beams;    // [n_maps][max_beams] staging
    float *d_poses;       // [n_maps][3] staging
    double *d_scratch;    // small device scratch
    int32_t need_full_build;  // likelihood field must be rebuilt everywhere (upload/reset/copy)
    gms_beam *h_beams;    // pinned staging
    float *h_poses;       // pinned staging
    int32_t *d_trace_cells; uint8_t *d_trace_cls; int32_t *d_trace_cnt; size_t trace_cap_bytes;
    // profiling
    int32_t prof_on;
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
    float *d_x, *d_y, *d_th;        // [n_maps][n] current poses
    float *d_x2, *d_y2, *d_th2;     // resample double buffer
    double *d_w, *d_w2;             // [n_maps][n] weights
    double *d_logw, *d_logw2;       // [n_maps][n] sum(log factor)
    float *d_cs;                    // [n_maps][n][2] float-rounded cos/sin of theta
    double *d_hitbeams;             // [n_maps][max_beams][2] compacted hit beams
    int32_t *d_nhit;                // [n_maps]
    double *d_partials;             // [n_maps][nblk_global][GMS_PARTIAL_STRIDE]
    double *d_partials2;            // second-phase partials [n_maps][nblk_global][4]
    PackedParticle *d_global;       // [n_maps][n_global] source population (own copy when unsharded)
    double *d_chunk_tot;            // [n_maps][nchunks] scan chunk totals / offsets
    double *d_cum;                  // [n_maps][n_global] in-chunk inclusive sums
    PfStatsDev *d_stats;            // [n_maps]
    PfStatsDev *h_stats;            // pinned
    double *d_r01;                  // [n_maps]
    int32_t *d_idx;                 // [n_maps][n]
    float *h_stage;                 // pinned staging for poses
    int32_t have_global;            // d_global holds the current normalised population
};

// ---- kernel launchers (gms_map_kernels.hip / gms_pf_kernels.hip) -----------------------------
void gms_launch_raycast(gms_map *m, const gms_beam *d_beams, int32_t B, int32_t beam_stride, const float *d_poses);
void gms_launch_trace_scan(gms_map *m, const gms_beam *d_beams, int32_t B, const float *d_pose,
                           int32_t *d_cells, uint8_t *d_cls, int32_t cap, int32_t *d_counts);
void gms_launch_trace_ray(gms_map *m, float x0, float y0, float x1, float y1, int32_t extra,
                          int32_t *d_cells, int32_t cap, int32_t *d_count);
void gms_launch_apply_ray(gms_map *m, RayIn ray);
void gms_launch_apply_counts(gms_map *m);
void gms_launch_likelihood(gms_map *m, int32_t dirty_only);
void gms_launch_fill(gms_map *m, double *d, double v, int64_t n);
void gms_launch_get_raw(gms_map *m, int32_t mi, int32_t x, int32_t y, double *d_out2);
void gms_launch_debug_f32(gms_map *m, int32_t op, const float *d_a, float *d_out, int64_t n);

void gms_launch_pf_init(gms_pf *pf);
void gms_launch_pf_prep(gms_pf *pf, const gms_beam *d_beams, int32_t B, int32_t beam_stride);
void gms_launch_pf_set_poses_aos(gms_pf *pf, const float *d_xytheta);
void gms_launch_pf_score(gms_pf *pf, int32_t B);
void gms_launch_pf_partials(gms_pf *pf, double *d_partials);
void gms_launch_pf_pack(gms_pf *pf, PackedParticle *d_packed, int64_t stride);
void gms_launch_pf_apply_partials(gms_pf *pf, const double *d_partials, PackedParticle *d_packed_local);
void gms_launch_pf_global_stats(gms_pf *pf);
void gms_launch_pf_resample(gms_pf *pf, double fraction /* <0: unconditional */);
void gms_launch_pf_refine(gms_pf *pf, int32_t B);
void gms_launch_pose_from_pf(gms_map *m, gms_pf *pf, int32_t which, float *d_poses);

// profiling brackets
void gms_prof_begin(gms_map *m, int32_t k);
void gms_prof_end(gms_map *m);

struct ProfScope {
    gms_map *m;
    bool on;
    ProfScope(gms_map *mm, int32_t k) : m(mm), on((mm->prof_on >> k) & 1) { if (on) gms_prof_begin(m, k); }
    ~ProfScope() { if (on) gms_prof_end(m); }
};
