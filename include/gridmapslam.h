/*
 * gridmapslam.h -- C-ABI of the MI355X-native occupancy-grid SLAM core (libgridmapslam.so).
 *
 * Drop-in boundary for ONE hot path of antbern/gridmap-slam-robot's GridMapGL app: the log-odds
 * ray-cast map update, the likelihood-field build and the particle scan matcher (score,
 * normalise / Neff, weighted pose, systematic resample).  The reference has no FFI of its own; each
 * entry point below replaces one public Java method and cites it.  J/ =
 * java/GridMapGL/src/main/java/com/fmsz/gridmapgl/ in the reference tree.  The JNI / cgo-style
 * binding a maintainer would add is shown in INTEGRATION.md; a C++ mirror of the Java classes is in
 * include/gridmapslam.hpp.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, opaque handles, int status (GMS_OK = 0, negative =
 *    error; text via gms_last_error()).  No exception crosses the boundary.
 *  - The caller owns every host buffer; the library owns every device buffer.
 *  - One host thread per handle at a time (the reference calls this path from one thread only:
 *    J/app/DataEventHandler.java:24-26).
 *  - All work runs on the handle's HIP stream (gms_map_set_stream).  Entry points that only take
 *    inputs enqueue and return; entry points that fill a host buffer synchronise that stream
 *    before returning.
 *  - Grids are row-major `x + y*W` doubles exactly like GridMapData.logData / likelihoodData
 *    (J/slam/GridMap.java:72-74,135), so a JNI shim can Get/SetDoubleArrayRegion them as they are.
 *  - A handle with n_maps > 1 is a batch of independent maps (and particle sets): every per-map
 *    argument then carries a leading [n_maps] dimension.
 *  - There is no CPU fallback: without a HIP device gms_map_create fails with GMS_ERR_NO_DEVICE.
 */
#ifndef GRIDMAPSLAM_H
#define GRIDMAPSLAM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define GMS_VERSION_MAJOR 0
#define GMS_VERSION_MINOR 1

#define GMS_MAX_TAPS 129        /* size of gms_params.kernel; gms_map_create refuses a kernel whose halo does not fit the likelihood
                                   pass's LDS tile (MI355X: up to 65 taps) */
#define GMS_BLOCK 256           /* particles per reduction block; shard offsets are multiples of it */
#define GMS_MAX_BEAMS 4096      /* beams per scan (per map): gms_params.max_beams may not exceed it */
#define GMS_MAX_PARTICLES (1 << 20)   /* particles per map of a filter, and the global population of a sharded one: the resampling
                                         kernels keep two levels of the cumulative-weight scan in LDS (133 KiB at this size) */

enum {
    GMS_OK = 0,
    GMS_ERR_INVALID = -1,       /* bad argument */
    GMS_ERR_NO_DEVICE = -2,     /* no usable HIP device (there is no CPU path) */
    GMS_ERR_HIP = -3,           /* a HIP runtime call failed */
    GMS_ERR_NOMEM = -4,
    GMS_ERR_STATE = -5          /* call order violated (e.g. resample before normalise) */
};

/* One LIDAR beam = the fields of Observation.Measurement the path reads
 * (J/slam/Observation.java:37-41).  32 bytes. */
typedef struct gms_beam {
    double local_x;             /* Measurement.localX  (robot frame, metres) */
    double local_y;             /* Measurement.localY */
    double distance;            /* Measurement.distance (metres; SENSOR_MAX_RANGE on a miss) */
    uint8_t hit;                /* Measurement.wasHit */
    uint8_t pad_[7];
} gms_beam;

/* Everything the GridMap constructor and the constants it closes over decide
 * (J/slam/GridMap.java:80-100,210,223,259; J/slam/SensorModel.java:20-25).
 * Fill with gms_params_default() and override fields as needed. */
typedef struct gms_params {
    float width_m, height_m;    /* GridMap(width, height, ...) */
    float resolution;           /* metres per cell */
    float pos_x, pos_y;         /* lower-left corner in the world */
    int32_t n_maps;             /* batch of independent maps (1 = the reference's single map) */
    int32_t device;             /* HIP device ordinal */
    double l_free, l_occ;       /* Util.logOdds(P_FREE), Util.logOdds(P_OCCUPPIED); host-supplied so
                                   that the JVM's Math.log decides them */
    int32_t ktaps;              /* likelihood kernel (Util.generateGaussianKernel); host-supplied so */
    double kernel[GMS_MAX_TAPS];/* that the JVM's Math.exp decides the taps */
    int32_t extra_steps;        /* RayIterator additionalSteps, 2 */
    float hit_tolerance;        /* inverseSensorModel hitTolerance, 2 */
    double z_hit, z_random;     /* 0.9, 1 - 0.9 */
    float max_range;            /* SensorModel.SENSOR_MAX_RANGE, 10 */
    int32_t max_beams;          /* capacity per scan (per map); 0 = 2048; at most GMS_MAX_BEAMS */
} gms_params;

typedef struct gms_map gms_map;      /* GridMap + GridMapData (n_maps of them) */
typedef struct gms_pf gms_pf;        /* ParticleFilter / SLAM particle set bound to a gms_map */
typedef struct gms_comm gms_comm;    /* one rank's RCCL communicator (multi-GPU filters) */

/* Results of gms_pf_normalize, per map (SLAM.update's bookkeeping, J/slam/SLAM.java:87-129). */
typedef struct gms_pf_stats {
    double weight_sum;          /* sum of raw weights (SLAM.java:100) */
    double neff;                /* calculateNeff() (SLAM.java:180-190) */
    int32_t strongest;          /* global index of the first maximum raw weight (SLAM.java:110-115) */
    int32_t n_zero;             /* raw weights that were exactly 0 (underflow census; not in the reference) */
    double max_log_weight;      /* max over particles of sum(log factor); see gms_pf_score */
} gms_pf_stats;

/* ---- library ---------------------------------------------------------------------------------- */
int gms_version(void);                         /* major*1000 + minor */
/* 16 hex digits: sha256 prefix of the sources (the csrc directory, this header) the loaded binary was built from; build() prints the
 * same string ("built <hash>" / "reused <hash>"), so a log shows which binary ran. */
const char *gms_build_info(void);
const char *gms_last_error(void);              /* thread-local message of the last failing call */
int gms_device_count(void);                    /* HIP devices visible; 0 when none (never throws) */

/* ---- host-side helpers (pure; usable without a device) ----------------------------------------- */
/* GridMap ctor arithmetic: fills every field with the reference's values, computing l_free/l_occ
 * with libm log and the kernel with libm exp (J/slam/GridMap.java:80-100). */
int gms_params_default(gms_params *p, float width_m, float height_m, float resolution, float pos_x, float pos_y);
/* gridSize = ceil(width / resolution) in float (J/slam/GridMap.java:85). */
int gms_grid_size(const gms_params *p, int32_t *W, int32_t *H);
/* Util.generateGaussianKernel(sigma, size): out has 2*size+1 taps (J/app/Util.java:428-455). */
int gms_generate_gaussian_kernel(double sigma, int32_t size, double *out);
/* Util.logOdds / Util.invLogOdds (J/app/Util.java:35-37,46-48). */
double gms_log_odds(double p);
double gms_inv_log_odds(double l);

/* ---- GridMap / GridMapData --------------------------------------------------------------------- */
/* new GridMap(width,height,resolution,position) + createMapData(null)  (GridMap.java:80,106). */
int gms_map_create(const gms_params *p, gms_map **out);
/* GMS_ERR_STATE while particle filters created on this map are alive: destroy those first. */
int gms_map_destroy(gms_map *m);
int gms_map_get_size(const gms_map *m, int32_t *W, int32_t *H, int32_t *n_maps);
/* Run this handle's work on an existing hipStream_t (e.g. a torch stream); NULL restores the handle's own stream.
 * NB the legacy default stream's handle IS NULL: a host whose other work runs on the default stream gets the handle's
 * own stream here, unrelated to it -- give both sides one real stream when they exchange device buffers. */
int gms_map_set_stream(gms_map *m, void *hip_stream);
int gms_map_synchronize(gms_map *m);
/* GridMap.reset (GridMap.java:129-132): logData := logOdds(0.5) = 0 (likelihoodData is left alone,
 * as in the reference, until the next gms_map_build_likelihood). */
int gms_map_reset(gms_map *m);
/* GridMapData array access (GridMap.java:72-74; read by the renderer :371-388 and serialiser).
 * n_maps*W*H doubles. */
int gms_map_upload_log(gms_map *m, const double *log_data);
int gms_map_download_log(gms_map *m, double *log_data);
int gms_map_upload_likelihood(gms_map *m, const double *lik);
int gms_map_download_likelihood(gms_map *m, double *lik);
/* createMapData(other): device-to-device copy of both arrays (GridMap.java:106-124). */
int gms_map_copy(gms_map *dst, const gms_map *src);
/* GridMapApp.calculateCombined (J/app/GridMapApp.java:439-458): dst (one map) := logOdds(1 - prod_m (1 - prob_m)) over
 * the n_maps of src, maps in index order; rebuild dst's likelihood field afterwards as the reference does (:457). */
int gms_map_combine(gms_map *dst, gms_map *src);
/* The scan de-skew of GridMapApp.onHandleData (J/app/GridMapApp.java:143-175): `length` raw measurements
 * {angle, distance, wasHit} and the frame's odometry -> beams (Observation.java:69-76), computed on the device
 * into the handle's staging buffer.  beams_out (host, may be NULL) receives a copy; *dev_beams_out (may be
 * NULL) its device address, valid until the next call that stages beams, for the *_dev entry points. */
int gms_map_deskew(gms_map *m, const double *angle, const double *distance, const uint8_t *hit, int32_t length,
                   double d_center, double d_theta, gms_beam *beams_out, const gms_beam **dev_beams_out);
/* getRawAt(map,x,y) / getProbAt (GridMap.java:134-140) for map index mi. */
int gms_map_get_raw_at(gms_map *m, int32_t mi, int32_t x, int32_t y, double *raw, double *prob);
/* getRawAt(map, Vec2 point) / getLikelihood(map, Vec2 point) (GridMap.java:142-156): the world point goes through
 * the reference's float arithmetic, (point - position) / resolution, intValue(), index x + y*W -- only the flat
 * index is range-checked there (ArrayIndexOutOfBounds -> GMS_ERR_INVALID here), so an x beyond the row reads the
 * neighbouring row exactly as the Java does.  raw / likelihood may be NULL. */
int gms_map_get_at_point(gms_map *m, int32_t mi, float point_x, float point_y, double *raw, double *likelihood);

/* GridMap.integrateObservation(map, obs, pose) (GridMap.java:173-191): beams[n_maps][B],
 * poses[n_maps][3] = x,y,theta. */
int gms_map_integrate(gms_map *m, const gms_beam *beams, int32_t B, const float *poses);
/* The same with the pose taken from a particle filter's device-resident weighted pose
 * (SLAM.getWeightedPose) or strongest particle -- no host round trip.  which: 0 weighted, 1 strongest. */
int gms_map_integrate_at(gms_map *m, const gms_beam *beams, int32_t B, gms_pf *pf, int32_t which);
/* GridMap.applyMeasurement(map,startX,startY,endX,endY,measuredDistance,wasHit) on map 0
 * (GridMap.java:194-228). */
int gms_map_apply_ray(gms_map *m, float sx, float sy, float ex, float ey, float measured, int32_t hit);
/* RayIterator(W,H).init(x0,y0,x1,y1,extra) + the hasNext/next loop, run on the device
 * (J/slam/RayIterator.java:65-130): ordered cells x0,y0,x1,y1,...; *n = number visited (may
 * exceed cap; only cap cells are stored). */
int gms_map_trace_ray(gms_map *m, float x0, float y0, float x1, float y1, int32_t extra,
                      int32_t *cells_xy, int32_t cap, int32_t *n);
/* The cell walk of integrateObservation for map 0 without touching the map: for beam b,
 * counts[b] cells; cells_xy/classes hold [B][cap] entries (class 0 free, 1 prior, 2 occupied:
 * J/slam/SensorModel.java:31-41). */
int gms_map_trace_scan(gms_map *m, const gms_beam *beams, int32_t B, const float pose[3],
                       int32_t *cells_xy, uint8_t *classes, int32_t cap, int32_t *counts);
/* GridMap.computeLikelihoodMap(map) (GridMap.java:233-250 + Util.java:378-426). */
int gms_map_build_likelihood(gms_map *m);
/* integrate + rebuild only the part of the likelihood field the scan can have changed (bit-identical
 * to gms_map_integrate followed by gms_map_build_likelihood when the field was current before). */
int gms_map_update(gms_map *m, const gms_beam *beams, int32_t B, const float *poses);
/* gms_map_update with the pose taken from the filter on the device (see gms_map_integrate_at). */
int gms_map_update_at(gms_map *m, const gms_beam *beams, int32_t B, gms_pf *pf, int32_t which);

/* ---- ParticleFilter / SLAM particle set -------------------------------------------------------- */
/* new ParticleFilter(n) (J/slam/ParticleFilter.java:43): n particles per map, weights 1/n_global,
 * poses 0 (SLAM.reset, J/slam/SLAM.java:65-77). */
int gms_pf_create(gms_map *m, int32_t n_particles, gms_pf **out);
int gms_pf_destroy(gms_pf *pf);
/* This handle holds particles [offset, offset+n) of a filter of n_global particles sharded over
 * several GPUs (offset % GMS_BLOCK == 0).  Default: offset 0, n_global = n. */
int gms_pf_set_shard(gms_pf *pf, int64_t offset, int64_t n_global);
/* getParticles() pose access (ParticleFilter.java:50): [n_maps][n][3] floats x,y,theta. */
int gms_pf_set_poses(gms_pf *pf, const float *xytheta);
int gms_pf_get_poses(gms_pf *pf, float *xytheta);
int gms_pf_set_weights(gms_pf *pf, const double *w);
int gms_pf_get_weights(gms_pf *pf, double *w);
int gms_pf_get_log_weights(gms_pf *pf, double *lw);
/* weight[i] = GridMap.probabilityOf(map, obs, pose[i]) for every particle (GridMap.java:261-294,
 * SLAM.java:99).  Also stores sum(log factor) per particle (underflow-free companion, not in the
 * reference).  beams[n_maps][B]. */
int gms_pf_score(gms_pf *pf, const gms_beam *beams, int32_t B);
/* SLAM.update's bookkeeping: weightSum, strongest, weight /= weightSum, calculateNeff
 * (SLAM.java:87-129,180-190) and getWeightedPose's sums (SLAM.java:165-178).  stats may be NULL
 * (no host synchronisation then); stats[n_maps]. */
int gms_pf_normalize(gms_pf *pf, gms_pf_stats *stats);
int gms_pf_get_stats(gms_pf *pf, gms_pf_stats *stats);
/* SLAM.getWeightedPose() (SLAM.java:165-178): out[n_maps][3]. */
int gms_pf_weighted_pose(gms_pf *pf, float *out);
/* SLAM.resample() (SLAM.java:133-153; class surface ParticleFilter.resample
 * J/slam/ParticleFilter.java:59-82) with Math.random() passed in as r01[n_maps].
 * Particles are replaced by copies (pose and weight).  indices (may be NULL): [n_maps][n] source
 * index per slot; n_ambiguous (may be NULL): slots whose boundary lies within rounding distance of
 * a cumulative weight, i.e. where a sequential and a blocked scan may legitimately disagree. */
int gms_pf_resample(gms_pf *pf, const double *r01, int32_t *indices, int32_t *n_ambiguous);
/* if (neff < fraction * n) resample()   (J/app/GridMapApp.java:185-186), decided on the device. */
int gms_pf_resample_if(gms_pf *pf, const double *r01, double fraction);
/* indices [n_maps][n]: the source slot of every particle after the last resampling step on this handle -- gms_pf_resample,
 * gms_pf_resample_if or the resample inside a scan step (SLAM.java:140-149; slot m itself where the conditional resample
 * did not run).  Synchronises the stream. */
int gms_pf_last_resample_indices(gms_pf *pf, int32_t *indices);
/* Particles held by this handle per map (ParticleFilter.java:43: `new Particle[n]`), maps, and the global population
 * of a sharded filter; any pointer may be NULL. */
int gms_pf_count(const gms_pf *pf, int32_t *n, int32_t *n_maps, int64_t *n_global);
/* flags[n_maps]: whether the last gms_pf_resample / gms_pf_resample_if replaced the particles. */
int gms_pf_did_resample(gms_pf *pf, int32_t *flags);
/* What the last normalise / scan step left on the device, per map (any pointer may be NULL): the weighted pose
 * (SLAM.getWeightedPose of the SCORED population, i.e. before a resample replaced it: the pose a fused scan step
 * integrated the scan at), the strongest particle's pose (SLAM.getStrongestParticle), whether the conditional
 * resample ran, and how many of its slots were ambiguous (see gms_pf_resample).  Synchronises the stream. */
int gms_pf_last_step(gms_pf *pf, float *weighted_pose, float *strongest_pose, int32_t *did_resample, int32_t *n_ambiguous);
/* SLAM.sampleMotionModel -> Odometry.apply(pose) for every particle (J/slam/SLAM.java:155-163,
 * J/slam/Odometry.java:60-96): Gaussian step and heading change with the reference's standard deviations
 * ((0.01 + 0.05|dCenter|)/2 and 5 deg + 0.1|dTheta|).  The reference's random stream is unseeded and
 * cannot be reproduced; variates come from Philox4x32-10(seed; global particle index, sequence), so a
 * sharded filter draws the same numbers as a stand-alone one. */
int gms_pf_sample_motion(gms_pf *pf, double d_center, double d_theta, uint64_t seed, uint64_t sequence);
/* GridMap.findBestPose(map, obs, startPose) lattice search around every particle
 * (GridMap.java:319-346): poses are replaced by the argmax pose. */
int gms_pf_refine_poses(gms_pf *pf, const gms_beam *beams, int32_t B);

/* SLAM.update refines every particle's pose before weighting it (J/slam/SLAM.java:96-97; the reference calls
 * findBestPoseOptim there, whose objective is broken -- SURVEY.md section 3.1 -- and keeps the lattice search
 * findBestPose commented out beside it).  on != 0: gms_slam_update / gms_slam_update_dev / the sharded scan steps run
 * gms_pf_refine_poses' lattice search (GridMap.java:319-346) on the motion-model samples before scoring them.
 * Default off (the search is 1210 probabilityOf evaluations per particle). */
int gms_pf_set_refine(gms_pf *pf, int32_t on);

/* Opt-in robust normalisation (not in the reference).  The reference's weight is the plain product of up
 * to 720 factors in [0.01, 0.91] (GridMap.java:262-288), which underflows for all but a handful of particles of a wide cloud: the
 * filter then runs on one survivor.  on != 0: the normalise step of this filter (gms_pf_normalize, the scan steps) takes
 * weight[i] = exp(logw[i] - max_j logw[j]) from the log-weights gms_pf_score keeps beside the products, and everything after it
 * -- weight sum, strongest, weighted pose, Neff, resampling -- runs on those weights as before (same blocked reductions; one more
 * launch in front of the block partials).  gms_pf_stats.weight_sum is then the sum of the rescaled weights (the largest is 1),
 * max_log_weight the scale.  Default off: the parity outputs are the reference's arithmetic.  Stand-alone filters only
 * (GMS_ERR_STATE on a shard); gms_pf_set_shard turns it off. */
int gms_pf_set_log_normalize(gms_pf *pf, int32_t on);

/* Reference-order audit (tests; slow on purpose).  The default kernels re-associate three chains of the reference's arithmetic: the
 * product of a scan's factors (GridMap.java:262-288: segment products, combined), weightSum (SLAM.java:100: blocked sums) and the
 * cumulative weights of resample() (SLAM.java:137-144: a three-level scan).  on != 0: gms_pf_score, gms_pf_normalize, gms_pf_resample[_if],
 * gms_pf_weighted_pose and the scan steps of this filter (which then take the separate launches) run each of them as ONE chain in
 * the reference's own order -- one register per particle for the product, one lane for the sums -- so that raw weights (zeros and
 * denormals included), the weight sum, Neff, the weighted pose and the resampling indices can be compared with the oracle for
 * EQUALITY, and the default path with this one: what differs between the two is association and nothing else.  Stand-alone filters. */
int gms_pf_set_reference_order(gms_pf *pf, int32_t on);

/* ---- SLAM as the reference has it: one GridMapData per particle --------------------------------------------------------------
 * J/slam/SLAM.java keeps a map in every Particle (:30-47): update() scores a particle against ITS OWN likelihood field and
 * integrates the scan into ITS OWN map at ITS OWN pose (:88-107), resample() deep-copies both arrays of the surviving particle's
 * map (:41-45 -> GridMap.createMapData(other), J/slam/GridMap.java:106-124).  The gms_pf entry points above score N poses against
 * ONE shared map (what BASELINE's configurations need); this handle is the reference's filter literally, at its own operating
 * point (500 particles x 120 x 120 cells, SLAM.java:50,57) and beyond.  Not the reference's: findBestPoseOptim (:97; BOBYQA on an
 * objective that is 0 / NaN) is left out, and the motion-model draw (:90) comes from Philox keyed by the particle's
 * slot index (see gms_pf_sample_motion). */
typedef struct gms_slam gms_slam;
/* new SLAM() (SLAM.java:56-62) + reset() (:65-77): n_particles particles at Pose(0, 0, 0) with weight 1 / n and a blank map each
 * (createMapData(null)).  p as for gms_map_create, with p->n_maps == 1 (the GridMap whose GridMapData every particle instantiates). */
int gms_slam_create(const gms_params *p, int32_t n_particles, gms_slam **out);
int gms_slam_destroy(gms_slam *s);
int gms_slam_reset(gms_slam *s);                                    /* SLAM.reset() (SLAM.java:65-77) */
int gms_slam_count(const gms_slam *s, int32_t *n_particles, int32_t *W, int32_t *H);
/* The handles behind it, owned by the gms_slam (do not destroy them; the filter refuses gms_pf_resample[_if], gms_pf_set_shard and the
 * shared-map scan steps with GMS_ERR_STATE: they would move its particles without their maps): *map = SLAM.getGridMap() (:200) -- geometry, constants, the
 * stream every call of this handle runs on, and a GridMapData of its own that receives gms_slam_combined; *pf = getParticles()
 * (:192) without the maps: poses, weights and statistics through gms_pf_get_poses / gms_pf_set_poses / gms_pf_get_weights /
 * gms_pf_set_weights / gms_pf_get_stats / gms_pf_weighted_pose (getWeightedPose, :165-178) / gms_pf_last_step (strongest particle). */
int gms_slam_handles(gms_slam *s, gms_map **map, gms_pf **pf);
/* SLAM.update(z, u) (SLAM.java:80-131) for all particles: sampleMotionModel (:90; sample_motion == 0 is its `u == null` branch,
 * :159: the particles keep their poses, e.g. because the caller has set the samples with gms_pf_set_poses),
 * computeLikelihoodMap(p.m) (:93), p.weight = probabilityOf(p.m, z, p.pose) (:99; the product taken in beam order by one lane: the
 * reference's bits, underflow included), integrateObservation(p.m, z, p.pose) unless |dTheta| > 30 degrees (:82,102-107), weightSum,
 * strongest, weight /= weightSum (:100,110-121), calculateNeff (:124).  beams[B]; stats (may be NULL; when given the call
 * synchronises) receives update()'s return value as stats->neff, the weight sum and the strongest particle's index. */
int gms_slam_update_per_particle(gms_slam *s, const gms_beam *beams, int32_t B, int32_t sample_motion, double d_center, double d_theta,
                                 uint64_t seed, uint64_t sequence, gms_pf_stats *stats);
int gms_slam_update_per_particle_dev(gms_slam *s, const gms_beam *dev_beams, int32_t B, int32_t sample_motion, double d_center, double d_theta,
                                     uint64_t seed, uint64_t sequence, gms_pf_stats *stats);
/* SLAM.update's pose refinement (SLAM.java:96-97): on != 0, every update runs GridMap.findBestPose (J/slam/GridMap.java:319-346: the
 * lattice of 11 x 11 x 10 poses around the motion-model sample, float loop counters, strict `>` against maxProb = 0 so that the first
 * maximum wins) for every particle against ITS OWN likelihood field, between computeLikelihoodMap(p.m) (:93) and the weighting (:99);
 * the particle is then weighted, and its map updated, at the refined pose.  The reference calls findBestPoseOptim (:97: BOBYQA from
 * commons-math on an objective that is 0 / NaN) and keeps this search commented out beside it (:96).  One workgroup per
 * particle; where the particle's field fits the CU's LDS as probabilityOf's factors (120 x 120 cells: 115 KB of 160) it is COMPUTED there
 * from the particle's class plane (no launch writes it first; blur kernels of 7 or 11 plain taps) or staged there from memory
 * (GMS_SLAM_REFINE_LDS=2 forces that form), and read from memory otherwise (GMS_SLAM_REFINE_LDS=0 forces that one).  Default off. */
int gms_slam_set_refine(gms_slam *s, int32_t on);
/* SLAM.resample() (SLAM.java:133-153) with Math.random() = r01: the systematic draw over the particles' weights, then every slot's
 * deep copy -- pose, weight (:42-43) and both arrays of the map (:44, GridMap.java:118-121): map[m] <- map[idx[m]], double-buffered,
 * a pure HBM stream.  logData moves at once (16 bytes per cell); likelihoodData's copy is made when something reads it -- a download,
 * an upload into a slot, another resample -- because the next update's computeLikelihoodMap overwrites every cell of it before
 * anything on the path does (what a caller can see is the deep copy either way; GMS_SLAM_LAZY_LIK_COPY=0 moves both at once).
 * indices [n] / n_ambiguous as gms_pf_resample (either may be NULL). */
int gms_slam_resample_maps(gms_slam *s, double r01, int32_t *indices, int32_t *n_ambiguous);
/* `if (neff < fraction * n) resample()` -- the rule of SLAM.update's caller (J/app/GridMapApp.java:185-186) -- decided ON THE DEVICE from the
 * Neff of the last update: nothing is read back, so update + this is one revolution without a host round trip.  Where the rule says no,
 * nothing is drawn and NOTHING IS COPIED, as in the reference: which of the two generations of the maps is current is itself a device-side
 * fact (a counter of the draws that ran, kept by the resampling kernel; every kernel of the handle picks the generation from its
 * parity), and the copy kernels return at once.  gms_pf_last_resample_indices (on the filter of gms_slam_handles) tells afterwards what
 * happened.  The threshold is fraction * n in doubles; the reference's `numParticles / 2` is an integer division, so for an odd
 * particle count its threshold is half a particle lower than fraction = 0.5's. */
int gms_slam_resample_maps_if(gms_slam *s, double r01, double fraction);
/* ---- the reference-shape filter over several GPUs: particles WITH their maps, no replica ------------------------------------------
 * Rank r holds the contiguous block [r * n_local, (r + 1) * n_local) of the n_global particles (n_local a multiple of GMS_BLOCK) and
 * nothing else.  update(): gms_slam_update_local[_dev] (the per-particle body of SLAM.java:88-107 for this block; the motion model's
 * variates are keyed by the GLOBAL particle index), then the weight exchange of a sharded gms_pf on the filter of gms_slam_handles --
 * gms_pf_local_partials -> all-reduce(SUM) -> gms_pf_apply_partials -> all-gather -> gms_pf_import_global -- which gives every rank
 * weightSum, strongest, Neff and the weighted pose (SLAM.java:100-124,165-190) bit-identical to the one-GPU filter.  resample():
 * gms_slam_shard_draw on every rank with the same r01 (this rank's slots drawn from the gathered population; systematic resampling is
 * order-preserving, so a slot's source lives on this rank or a neighbouring one unless the weights have collapsed); the ranks
 * all-gather their `sources`, each sends the records (gms_slam_shard_export: logData + class planes, gms_slam_record_doubles doubles
 * per particle) of its particles that other ranks drew, and gms_slam_shard_gather makes the copies from the rank's own previous
 * generation and the received records.  The collectives stay with the caller (gridmap_slam_robot_amd/distributed.py:
 * ShardedSlamParticleMaps over torch.distributed = RCCL).  Poses, weights and every map equal the one-GPU gms_slam's for any number of
 * ranks.  A sharded handle needs the class planes (see gms_slam_create_shard's error text); its likelihoodData is produced on demand. */
int gms_slam_create_shard(const gms_params *p, int32_t n_local, int64_t offset, int64_t n_global, gms_slam **out);
int gms_slam_update_local(gms_slam *s, const gms_beam *beams, int32_t B, int32_t sample_motion, double d_center, double d_theta, uint64_t seed,
                          uint64_t sequence);
int gms_slam_update_local_dev(gms_slam *s, const gms_beam *dev_beams, int32_t B, int32_t sample_motion, double d_center, double d_theta,
                              uint64_t seed, uint64_t sequence);
/* fraction < 0: unconditional; else `if (neff < fraction * n_global) resample()`.  *did: it drew; sources [n_local]: global source indices.
 * Synchronises. */
int gms_slam_shard_draw(gms_slam *s, double r01, double fraction, int32_t *did, int32_t *sources);
int gms_slam_record_doubles(const gms_slam *s, int64_t *doubles);
int gms_slam_shard_export(gms_slam *s, const int32_t *local_indices, int32_t count, double *dev_dst);
int gms_slam_shard_gather(gms_slam *s, const int32_t *src_local, const int32_t *recv_pos, const double *dev_recv);
/* The same with the exchanges inside the library (RCCL; gms_comm_* below): SLAM.update(z, u) and SLAM.resample() of one rank's block as
 * ONE call each -- what a Java host with one JVM per GPU calls through the shim.  update: gms_slam_update_local, then the all-reduce of
 * the block partials and the all-gather of the packed particles (gms_pf_normalize_sharded_begin / _end); stats (may be NULL): identical
 * on every rank.  resample: the draw, an all-gather of the sources, the plan (gms_slam_plan_exchange: a pure host function), the
 * records that cross a rank boundary as one grouped launch of ncclSend / ncclRecv, the copies; *did (may be NULL).  With more than one
 * rank these have never executed (one GPU per box where they were written); the torch.distributed route above is the tested one. */
int gms_slam_update_sharded_maps(gms_slam *s, gms_comm *c, const gms_beam *beams, int32_t B, int32_t sample_motion, double d_center, double d_theta,
                                 uint64_t seed, uint64_t sequence, gms_pf_stats *stats);
int gms_slam_resample_sharded_maps(gms_slam *s, gms_comm *c, double r01, double fraction, int32_t *did);
int gms_slam_plan_exchange(const int32_t *all_sources, int32_t world, int32_t rank, int32_t n_local, int32_t *send_counts, int32_t *send_lists,
                           int32_t *recv_counts, int32_t *src_local, int32_t *recv_pos);
/* Particle i's GridMapData (SLAM.java:33; GridMap.java:72-74): W * H doubles each, either pointer may be NULL. */
int gms_slam_download_map(gms_slam *s, int32_t i, double *log_data, double *lik);
int gms_slam_upload_map(gms_slam *s, int32_t i, const double *log_data, const double *lik);
int gms_slam_download_maps(gms_slam *s, double *log_all, double *lik_all);        /* all particles: [n][H][W] */
/* GridMapApp.calculateCombined (J/app/GridMapApp.java:439-458) over the particles' maps, likelihood field included (:457), into the
 * GridMapData of the handle's own map (gms_slam_handles -> gms_map_download_log / gms_map_download_likelihood). */
int gms_slam_combined(gms_slam *s);
/* Diagnostics: the cell walk of integrateObservation(p.m, z, p.pose) (SLAM.java:105 -> GridMap.java:173-228) for particle i at its
 * current pose, exactly as the per-particle update kernel walks and classifies it, written out instead of counted: for beam b,
 * counts[b] cells in walk order (RayIterator.java:107-130), cells_xy / classes [B][cap] entries as gms_map_trace_scan (class 0 free,
 * 1 prior, 2 occupied; either may be NULL).  Touches no map.  Synchronises. */
int gms_slam_trace_scan(gms_slam *s, int32_t i, const gms_beam *beams, int32_t B, int32_t *cells_xy, uint8_t *classes, int32_t cap, int32_t *counts);
/* maps copied by resampling steps since creation (measurement: bytes moved = copies * W * H * 32) */
int gms_slam_copies(const gms_slam *s, int64_t *maps_copied);

/* ---- device-resident inputs ---------------------------------------------------------------------
 * The same entry points for callers whose scans / poses already live in HBM (a trace staged once, a
 * torch tensor, the output of a device-side motion model).  dev_beams is [n_maps][B] gms_beam,
 * dev_poses [n_maps][3], dev_xytheta [n_maps][n][3]; all are read on the handle's stream, asynchronously: the call
 * returns before the kernels run, so the buffers must stay valid and unmodified until the stream has passed them
 * (gms_map_synchronize, or an event of the caller's on that stream) -- in particular, memory handed back to a caching
 * allocator that serves another stream may be reused too early. */
int gms_map_integrate_dev(gms_map *m, const gms_beam *dev_beams, int32_t B, const float *dev_poses);
int gms_map_integrate_at_dev(gms_map *m, const gms_beam *dev_beams, int32_t B, gms_pf *pf, int32_t which);
int gms_map_update_dev(gms_map *m, const gms_beam *dev_beams, int32_t B, const float *dev_poses);
int gms_map_update_at_dev(gms_map *m, const gms_beam *dev_beams, int32_t B, gms_pf *pf, int32_t which);
int gms_pf_set_poses_dev(gms_pf *pf, const float *dev_xytheta);
int gms_pf_score_dev(gms_pf *pf, const gms_beam *dev_beams, int32_t B);
/* One scan through the whole path: SLAM.update(z, u) (J/slam/SLAM.java:80-131) followed by its caller's
 * `if (neff < fraction * N) resample()` (J/app/GridMapApp.java:185-186).  dev_xytheta (may be NULL) are the
 * motion-model samples; resample_fraction < 0 skips the resampling; integrate = 0 is the skipUpdate case
 * (SLAM.java:82).  r01[n_maps] is read on the host.  Nothing is read back. */
int gms_slam_update_dev(gms_pf *pf, const float *dev_xytheta, const gms_beam *dev_beams, int32_t B, const double *r01,
                        double resample_fraction, int32_t integrate);
/* SLAM.update(z, u) with the motion-model sample inside (J/slam/SLAM.java:80-131, line 90 included) and the caller's resampling
 * rule: every particle takes Odometry.apply (J/slam/Odometry.java:77-96; the variates of gms_pf_sample_motion for the same seed
 * and sequence -- bit-identical to that call followed by gms_slam_update_dev with dev_xytheta = NULL) on its way into the scoring
 * launch: four launches per scan, no launch for the motion model.  Stand-alone filters. */
int gms_slam_update_u_dev(gms_pf *pf, double d_center, double d_theta, uint64_t seed, uint64_t sequence, const gms_beam *dev_beams,
                          int32_t B, const double *r01, double resample_fraction, int32_t integrate);
/* One recorded revolution as GridMapApp.onHandleData treats it (J/app/GridMapApp.java:133-192), in one call: the raw polar
 * measurements are de-skewed with the frame's odometry (:143-175, as gms_map_deskew), every particle takes a motion-model sample
 * (J/slam/SLAM.java:90, as gms_pf_sample_motion with the same seed and sequence), then the scan step of gms_slam_update_dev on
 * the de-skewed scan.  Bit-identical to those three calls; the de-skew and the motion model share a launch (five launches per
 * frame).  Single-map, stand-alone filters.  angle / distance / hit [length] are host arrays and may be reused on return. */
int gms_slam_frame(gms_pf *pf, const double *angle, const double *distance, const uint8_t *hit, int32_t length, double d_center,
                   double d_theta, uint64_t seed, uint64_t sequence, const double *r01, double resample_fraction, int32_t integrate);
/* The same scan step with HOST inputs (what a JNI caller holds): poses (may be NULL) and the scan are copied into
 * pinned staging rings before the call returns (the caller may reuse its buffers at once) and pulled in by the
 * device without a stream synchronise; stats (may be NULL; when given the call synchronises) receives SLAM.update's
 * return values. */
int gms_slam_update(gms_pf *pf, const float *xytheta, const gms_beam *beams, int32_t B, const double *r01,
                    double resample_fraction, int32_t integrate, gms_pf_stats *stats);

/* ---- multi-GPU plumbing (particles sharded over ranks; collectives stay with the caller) ------- */
/* Number of doubles of the block-partial vector exchanged by an all-reduce(SUM):
 * 9 per global block of GMS_BLOCK particles {sum w, max w, first index of max, zero count, max
 * log-weight, sum w^2, sum x*w, sum y*w, sum theta*w}: everything SLAM.update reports follows from them. */
int gms_pf_partials_len(const gms_pf *pf, int64_t *n_doubles);
/* Phase 1 of normalise on a shard: writes this shard's block partials into dev_partials (device
 * pointer, zero elsewhere), ready for all-reduce(SUM) -- adding zeros is exact, so the reduced
 * vector is the same for any number of ranks. */
int gms_pf_local_partials(gms_pf *pf, double *dev_partials);
/* Phase 2: consumes the all-reduced partials: weightSum, strongest, Neff, weighted pose,
 * weight /= weightSum; then packs
 * this shard's {weight, x, y, theta} (24 B per particle) into dev_packed for the all-gather. */
int gms_pf_apply_partials(gms_pf *pf, const double *dev_partials, void *dev_packed);
/* Statistics only (weighted pose, Neff) of the CURRENT particles from an all-reduced partial vector,
 * nothing rewritten: e.g. getWeightedPose after a resample (J/app/GridMapApp.java:192). */
int gms_pf_stats_from_partials(gms_pf *pf, const double *dev_partials);
/* Packs this shard's current {weight, x, y, theta} without normalising (e.g. to all-gather the
 * population again after a resample, for getWeightedPose). */
int gms_pf_pack(gms_pf *pf, void *dev_packed);
/* Phase 3: consumes the all-gathered [n_global] packed particles: they become the source population
 * of gms_pf_resample (every rank then fills its own slots from the same global array).  Zero copy: the
 * buffer is read in place by later calls and must stay valid and unmodified until the next
 * gms_pf_import_global / gms_pf_normalize on this handle, or its destruction. */
int gms_pf_import_global(gms_pf *pf, const void *dev_packed_global);

/* ---- multi-GPU with the exchanges inside the library (RCCL, one process per GPU) ---------------- */
/* RCCL is bound at run time.  gms_comm_load names the shared object (NULL/"" = the copy the process already
 * holds, else the loader's librccl.so); the other entry points load the default on first use. */
int gms_comm_load(const char *librccl_path);
/* 128 opaque bytes (ncclUniqueId): one rank makes them, the host distributes them to every rank. */
int gms_comm_unique_id(void *id128);
/* Joins the communicator; blocks until all `world` ranks have called it.  One communicator per device. */
int gms_comm_create(gms_comm **out, const void *id128, int32_t rank, int32_t world, int32_t device);
int gms_comm_destroy(gms_comm *c);
int gms_comm_rank(const gms_comm *c, int32_t *rank, int32_t *world);
/* SLAM.update's weight bookkeeping (J/slam/SLAM.java:100-124) for a filter sharded with gms_pf_set_shard into
 * equal shards in rank order: block partials -> all-reduce(SUM) -> normalise -> START of the all-gather of the
 * packed normalised particles on the communicator's side stream.  On return (stream order) the statistics and
 * the weighted / strongest pose are complete, so the map update can be enqueued beside the gather. */
int gms_pf_normalize_sharded_begin(gms_pf *pf, gms_comm *c);
/* Joins the gather; the global population becomes the source of gms_pf_resample[_if]. */
int gms_pf_normalize_sharded_end(gms_pf *pf, gms_comm *c);
/* gms_slam_update_dev for a sharded filter: every rank calls it with its shard's motion-model samples and the
 * same scan and r01.  ONE exchange per scan: the ranks all-gather their RAW weights + poses and their block partials
 * (one grouped RCCL launch); normalisation, statistics, map update and resample are local after that.  Results equal
 * the stand-alone filter's bit for bit. */
int gms_slam_update_sharded_dev(gms_pf *pf, gms_comm *c, const float *dev_xytheta, const gms_beam *dev_beams, int32_t B,
                                const double *r01, double resample_fraction, int32_t integrate);
/* The same with HOST inputs (what a JNI caller holds; see gms_slam_update): this rank's shard of the motion-model samples
 * (may be NULL) and the scan are staged through the pinned rings; stats (may be NULL; synchronises) receives
 * SLAM.update's return values, identical on every rank. */
int gms_slam_update_sharded(gms_pf *pf, gms_comm *c, const float *xytheta, const gms_beam *beams, int32_t B, const double *r01,
                            double resample_fraction, int32_t integrate, gms_pf_stats *stats);
/* The same step for a host that brings its own collectives: _begin (poses, weights, this shard's payloads), then the
 * caller all-gathers BOTH buffers of gms_pf_gather_buffers in place (rank r's payload sits at r * per-rank size;
 * equal shards in rank order), then _end.  The buffers belong to the handle and keep their addresses until
 * gms_pf_set_shard / destroy. */
int gms_slam_update_sharded_begin_dev(gms_pf *pf, const float *dev_xytheta, const gms_beam *dev_beams, int32_t B);
int gms_pf_gather_buffers(gms_pf *pf, void **dev_packed_global, int64_t *packed_bytes_per_rank, double **dev_partials_global,
                          int64_t *partials_doubles_per_rank);
int gms_slam_update_sharded_end_dev(gms_pf *pf, const gms_beam *dev_beams, int32_t B, const double *r01,
                                    double resample_fraction, int32_t integrate);

/* ---- measurement ------------------------------------------------------------------------------- */
enum {
    GMS_K_RAYCAST = 0, GMS_K_APPLY = 1, GMS_K_LIKELIHOOD = 2, GMS_K_SCORE = 3, GMS_K_REDUCE = 4,
    GMS_K_RESAMPLE = 5, GMS_K_REFINE = 6, GMS_K_EXCHANGE = 7 /* the grouped RCCL launch of a sharded scan step */,
    GMS_K_ORDER = 8 /* the locality order of the particles ahead of a large scoring launch */,
    GMS_K_MAPCOPY = 9 /* resample()'s deep copies of the particles' maps (gms_slam_resample_maps) */, GMS_K_COUNT = 10
};
/* Bracket kernel launches of this map handle (and its filters) with HIP events on its stream:
 * bit k of `mask` enables kernel class k (GMS_K_*); 0 turns profiling off. */
int gms_profile_enable(gms_map *m, int32_t mask);
int gms_profile_reset(gms_map *m);
/* Bracket only every stride-th launch of an enabled class (default 1): the two event markers of a bracket cost
 * microseconds on the stream, which a 70 us scan step notices. */
int gms_profile_sample(gms_map *m, int32_t stride);
/* Total device milliseconds and launch count of kernel class k since the last reset. */
int gms_profile_get(gms_map *m, int32_t k, double *total_ms, int64_t *launches);
/* What such a bracket costs by itself: the mean event-to-event time around an EMPTY kernel (dispatch latency + the
 * ~1 us an empty kernel runs), over `reps` launches on the handle's stream.  A kernel-trace profiler reports the
 * kernel's own duration, i.e. roughly the bracketed time minus this. */
int gms_profile_calibrate(gms_map *m, int32_t reps, double *bracket_ms);
/* The same in its two parts: bracket_ms as above; kernel_ms = what one empty kernel takes when `reps` of them run back to back
 * with nothing between them (the floor of any launch in a kernel trace).  bracket_ms - kernel_ms is what the two event markers
 * add to a bracketed launch: subtract it from a bracketed duration to get what a kernel-trace profiler reports. */
int gms_profile_calibrate2(gms_map *m, int32_t reps, double *bracket_ms, double *kernel_ms);

/* Census of the likelihood-field tiles (64 x 32 cells) the rebuilds have walked since the last call, summed over the handle's
 * maps: out4 (may be NULL) = {left alone: no cell changes its thresholded code under the scan's counts; constants kept: a
 * uniform tile that already holds its constants; constants written: a uniform tile; blurred: both passes of
 * Util.doGaussianBlurdSeparable (J/app/Util.java:378-426)}.  Reads and clears the counters (synchronises when out4 is given);
 * enable != 0 keeps counting, 0 stops (the default: off).  The reference rebuilds every cell on every scan
 * (J/slam/GridMap.java:233-250); the sum of the four is what a dirty-tile rebuild looked at. */
int gms_map_tile_stats(gms_map *m, int32_t enable, int64_t *out4);

/* ---- diagnostics ------------------------------------------------------------------------------- */
/* The float-rounded device primitives the parity contract leans on, for tests: op 0 = (float)sqrt(a)
 * (GridMap.java:217), 1 = (float)cos((double)a), 2 = (float)sin((double)a) (J/math/MathUtil.java:30-40); 3 = self-check of the
 * wavefront butterflies every reduction uses (n a multiple of 64): out[i] = a bit code, bits 0-5 set where the exchange with lane
 * (i ^ 32, 16, 8, 4, 2, 1) does not deliver that lane's value (must be 0), bits 6-11 the same for the mirrored reading (must be 63 << 6 for the half-wave, row and quad-of-four steps that have one);
 * 4 / 5 = the squared thresholds the per-particle-map ray cast classifies cells with instead of a square root per cell: the smallest float s with
 * (float)sqrt(s) >= a, resp. the largest with (float)sqrt(s) <= a (SensorModel.java:31-41 compares (float)Math.sqrt(s) with a). */
int gms_debug_f32(gms_map *m, int32_t op, const float *in, float *out, int64_t n);
/* Development: instrumented builds (-DGMS_STAMPS) write wall-clock stamps of their kernels' stages to dev_buffer
 * ([workgroup][16] uint64, 10 ns units; NULL turns it off); a product build returns GMS_ERR_STATE.  tools/stamps.py. */
int gms_debug_set_stamps(gms_map *m, void *dev_buffer);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
