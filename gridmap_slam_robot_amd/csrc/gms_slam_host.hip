// gms_slam_host.hip -- C-ABI of the reference's own filter shape: SLAM (J/slam/SLAM.java), every particle with its own GridMapData.
// Handle lifetime and the launch sequences of SLAM.update / SLAM.resample; the kernels are in gms_slam_kernels.hip.  No CPU path.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "gms_internal.h"

#define HIPCHK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return gms_fail(GMS_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)
#define REQUIRE(cond, msg)                                         \
    do {                                                           \
        if (!(cond)) return gms_fail(GMS_ERR_INVALID, "%s", msg);  \
    } while (0)

SlamBufs gms_slam_bufs(const gms_slam *s) {
    SlamBufs b;
    for (int k = 0; k < 2; k++) { b.log[k] = s->d_log[k]; b.lik[k] = s->d_lik[k]; b.code[k] = s->d_code[k]; }
    b.epoch = s->d_epoch;
    return b;
}

// The current generation on the HOST: a stream synchronise and an 8-byte read.  Only where the host itself must address a particle's
// arrays -- downloads, uploads, reset, the copy counter -- never on the update / resample path.
static int slam_host_gen(gms_slam *s, int32_t *gen, int64_t *draws = nullptr) {
    int32_t e[2] = {0, 0};
    HIPCHK(hipStreamSynchronize(s->map->stream));
    HIPCHK(hipMemcpy(e, s->d_epoch, sizeof(e), hipMemcpyDeviceToHost));
    if (gen) *gen = e[0] & 1;
    if (draws) *draws = e[0];
    return GMS_OK;
}

extern "C" {

int gms_slam_destroy(gms_slam *s) {
    if (!s) return GMS_OK;
    if (s->map) { hipSetDevice(s->map->device); hipStreamSynchronize(s->map->stream); }
    for (int k = 0; k < 2; k++) { hipFree(s->d_log[k]); hipFree(s->d_lik[k]); }
    hipFree(s->d_idx_lik);
    hipFree(s->d_code[0]); hipFree(s->d_code[1]);
    hipFree(s->d_epoch);
    hipFree(s->d_plan);
    if (s->pf) gms_pf_destroy(s->pf);
    if (s->map) gms_map_destroy(s->map);
    delete s;
    return GMS_OK;
}

static int slam_create(const gms_params *p, int32_t n_particles, int64_t offset, int64_t n_global, gms_slam **out, bool shard_api = false);

int gms_slam_create(const gms_params *p, int32_t n_particles, gms_slam **out) {            // SLAM.java:56-62
    return slam_create(p, n_particles, 0, n_particles, out);
}

// One rank's block [offset, offset + n_local) of a filter of n_global particles with their maps (equal blocks in rank order, multiples
// of GMS_BLOCK): see gridmapslam.h "the reference-shape filter over several GPUs"
int gms_slam_create_shard(const gms_params *p, int32_t n_local, int64_t offset, int64_t n_global, gms_slam **out) {
    REQUIRE(n_global >= 1 && offset >= 0 && offset + n_local <= n_global, "gms_slam_create_shard: the block does not fit the population");
    REQUIRE(n_global == n_local || (n_local % GMS_BLOCK == 0 && offset % n_local == 0 && n_global % n_local == 0),
            "gms_slam_create_shard: equal blocks in rank order, each a multiple of GMS_BLOCK particles (the reductions' blocks must not straddle ranks)");
    return slam_create(p, n_local, offset, n_global, out, true);       // (a "shard" that is the whole population is both: one rank's view of a one-rank group)
}

static int slam_create(const gms_params *p, int32_t n_particles, int64_t offset, int64_t n_global, gms_slam **out, bool shard_api) {
    REQUIRE(p && out, "gms_slam_create: null argument");
    *out = nullptr;
    REQUIRE(p->n_maps == 1, "gms_slam_create: gms_params.n_maps must be 1 (every particle gets a map of its own)");
    REQUIRE(n_particles >= 1 && n_particles <= 65535, "gms_slam_create: particle count out of range (1 .. 65535)");
    gms_slam *s = new (std::nothrow) gms_slam();
    if (!s) return gms_fail(GMS_ERR_NOMEM, "out of host memory");
    s->n = n_particles;
    int rc = gms_map_create(p, &s->map);                                                   // new GridMap(...) :57
    if (!rc) rc = gms_pf_create(s->map, n_particles, &s->pf);                              // the particle list :59
    if (rc) { gms_slam_destroy(s); return rc; }
    gms_map *m = s->map;
    // the per-particle kernel keeps one row of the count tile at the very least (gms_launch_slam_particle)
    // (per beam: factor 8 B, thresholds 8 B, end-point cell 4 B; slots and ray records 13.3 KB; a class plane of at most 24 KiB)
    if (((size_t)m->max_beams + 8) * 20 + 16384 + 24576 + (size_t)m->gd.W * 4 + 4096 > (size_t)m->lds_per_cu) {
        gms_slam_destroy(s);
        return gms_fail(GMS_ERR_INVALID, "gms_slam_create: %d beams and rows of %d cells do not fit a workgroup's LDS", m->max_beams, m->gd.W);
    }
    const size_t bytes = (size_t)n_particles * (size_t)m->gd.cells * sizeof(double);
    const char *lazy_env = getenv("GMS_SLAM_LAZY_LIK_COPY");
    s->lazy_lik = !(lazy_env && lazy_env[0] == '0');
    const char *rl_env = getenv("GMS_SLAM_REFINE_LDS");
    s->refine_lds = rl_env && rl_env[0] == '0' ? 0 : (rl_env && rl_env[0] == '2' ? 2 : -1);
    const char *rf_env = getenv("GMS_SLAM_REFINE_FIELD");
    s->refine_field = rf_env && rf_env[0] == 'c' ? 1 : (rf_env && rf_env[0] == 'l' ? 0 : -1);
    bool ok = true;
    for (int k = 0; k < 2; k++)
        ok = ok && hipMalloc(&s->d_log[k], bytes) == hipSuccess && hipMalloc(&s->d_lik[k], bytes) == hipSuccess;
    ok = ok && hipMalloc(&s->d_idx_lik, (size_t)n_particles * sizeof(int32_t)) == hipSuccess;
    ok = ok && hipMalloc(&s->d_epoch, 2 * sizeof(int32_t)) == hipSuccess && hipMemset(s->d_epoch, 0, 2 * sizeof(int32_t)) == hipSuccess;
    s->pf->slam_owned = 1;          // its particles own maps: resampling goes through gms_slam_resample_maps[_if], which moves them
    // The class planes (gms_slam_kernels.hip): kept unless the blur kernel is wider than the on-demand evaluation takes, a plane would
    // crowd the count tile out of a workgroup's LDS, or GMS_SLAM_EAGER_LIK=1 asks for the reference's own schedule -- every cell of every
    // particle's likelihoodData rebuilt by every update (the like-for-like figure of bench.py)
    const char *eager_env = getenv("GMS_SLAM_EAGER_LIK");
    s->code_words = gms_slam_code_words(m->gd.cells);
    if (!(eager_env && eager_env[0] == '1') && m->gd.khalf <= 7 && s->code_words * 4 <= 24 * 1024 && m->gd.W <= 65535 && m->gd.H <= 65535) {
        const size_t cb = (size_t)n_particles * 2 * (size_t)s->code_words * sizeof(uint32_t);
        for (int k = 0; k < 2; k++) ok = ok && hipMalloc(&s->d_code[k], cb) == hipSuccess;
    }
    const bool sharded = shard_api || offset != 0 || n_global != n_particles;
    if (sharded) ok = ok && hipMalloc(&s->d_plan, (size_t)3 * n_particles * sizeof(int32_t)) == hipSuccess;
    if (!ok) {
        gms_slam_destroy(s);
        return gms_fail(GMS_ERR_NOMEM, "gms_slam_create: device allocation failed (%d particles x %lld cells x 32 bytes)", n_particles, (long long)m->gd.cells);
    }
    if (sharded) {
        if (!s->d_code[0]) {
            gms_slam_destroy(s);
            return gms_fail(GMS_ERR_INVALID, "gms_slam_create_shard: a sharded filter moves a particle as logData + its class planes; this map's planes are "
                                             "not kept (blur kernel wider than 15 taps, a plane over 24 KiB, or GMS_SLAM_EAGER_LIK=1)");
        }
        s->pf->d_epoch2 = s->d_epoch;                  // (lets gms_pf_set_shard through: the filter is otherwise closed to it)
        rc = gms_pf_set_shard(s->pf, offset, n_global);
        s->pf->d_epoch2 = nullptr;
        if (rc) { gms_slam_destroy(s); return rc; }
    }
    *out = s;
    return gms_slam_reset(s);
}

int gms_slam_reset(gms_slam *s) {                                                           // SLAM.java:65-77
    REQUIRE(s, "null handle");
    gms_map *m = s->map;
    HIPCHK(hipSetDevice(m->device));
    const size_t bytes = (size_t)s->n * (size_t)m->gd.cells * sizeof(double);
    int64_t draws = 0;
    int rc0 = slam_host_gen(s, nullptr, &draws);
    if (rc0) return rc0;
    s->copies_base += draws * s->n;                                                          // (the copy counter outlives a reset)
    HIPCHK(hipMemsetAsync(s->d_epoch, 0, 2 * sizeof(int32_t), m->stream));                   // generation 0 is current again
    // createMapData(null) per particle (GridMap.java:106-117): logData = logOdds(0.5) = 0.0, likelihoodData a fresh double[] = 0.0
    HIPCHK(hipMemsetAsync(s->d_log[0], 0, bytes, m->stream));
    HIPCHK(hipMemsetAsync(s->d_lik[0], 0, bytes, m->stream));
    if (s->d_code[0]) HIPCHK(hipMemsetAsync(s->d_code[0], 0, (size_t)s->n * 2 * (size_t)s->code_words * sizeof(uint32_t), m->stream));   // every class "logData == 0"
    s->lik_behind = 0;
    s->lik_from_codes = 0;
    gms_launch_pf_init(s->pf);                                                               // Pose(0, 0, 0), weight 1 / numParticles (:68-71)
    s->pf->pending_nseg = 0; s->pf->have_global = 0; s->pf->stats_current = 0; s->pf->score_fresh = 0;
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

int gms_slam_set_refine(gms_slam *s, int32_t on) {                                          // SLAM.java:96
    REQUIRE(s, "null handle");
    s->refine = on != 0;
    return GMS_OK;
}

int gms_slam_handles(gms_slam *s, gms_map **map, gms_pf **pf) {
    REQUIRE(s, "null handle");
    if (map) *map = s->map;
    if (pf) *pf = s->pf;
    return GMS_OK;
}

int gms_slam_count(const gms_slam *s, int32_t *n, int32_t *W, int32_t *H) {
    REQUIRE(s, "null handle");
    if (n) *n = s->n;
    if (W) *W = s->map->gd.W;
    if (H) *H = s->map->gd.H;
    return GMS_OK;
}

// the per-particle body of SLAM.update(z, u) (SLAM.java:88-107) for the particles this handle holds; the weights stay raw
static int slam_update_local(gms_slam *s, const gms_beam *dev_beams, int32_t B, int32_t sample_motion, double d_center, double d_theta,
                             uint64_t seed, uint64_t sequence) {
    REQUIRE(s && dev_beams, "null argument");
    gms_map *m = s->map;
    gms_pf *pf = s->pf;
    REQUIRE(B >= 0 && B <= m->max_beams, "beam count exceeds gms_params.max_beams");
    HIPCHK(hipSetDevice(m->device));
    const bool skip_update = fabs(d_theta) > (3.141592653589793 / 180.0) * 30;                              // :82
    MotionModel mo;
    mo.d_center = d_center; mo.d_theta = d_theta; mo.seed = seed; mo.sequence = sequence;
    // :93 for every particle.  With the class planes and no refinement the field is not written here: probabilityOf reads it under the
    // scan's end points only (GridMap.java:273-277), and k_slam_particle evaluates exactly those cells from the particle's plane; what a
    // caller may read afterwards -- the field of logData as it stands NOW -- stays defined by plane 1 and is written when asked for
    // (slam_lik_current).  The pose refinement looks up most of a field: it gets all of it.
    // ... unless it computes it itself: a field that fits a workgroup's LDS is computed there from the same plane.
    const bool planes = s->d_code[0] != nullptr;
    const bool on_demand = planes && (!s->refine || gms_slam_refine_from_planes(m, B, s->refine_lds, s->code_words));
    const SlamBufs sb = gms_slam_bufs(s);
    if (!on_demand) {
        // (plane 0 == the classes of logData, 1/32 of the bytes: 623 us against 793 at 4096 x 256^2, where logData is 2 GB; at
        //  500 x 120^2 -- 58 MB, inside the 256 MB infinity cache -- the blur's arithmetic binds and reading logData is 1.6 us FASTER)
        const bool big = (size_t)s->n * (size_t)m->gd.cells * sizeof(double) > ((size_t)256 << 20);
        if (s->d_code[0] && (s->refine_field < 0 ? big : s->refine_field == 1)) gms_launch_slam_likelihood_codes(m, sb, s->code_words, s->n, 0);
        else gms_launch_slam_likelihood(m, sb, s->n);
    }
    s->lik_behind = 0;                                                                                     // (every cell of every field is rewritten: an owed copy is moot)
    s->lik_from_codes = on_demand ? 1 : 0;
    bool drawn = false;
    if (s->refine) {                                                                                        // :90, then :96 (the lattice form of :97)
        if (!gms_launch_slam_refine(pf, dev_beams, B, sb, sample_motion ? &mo : nullptr, s->refine_lds, planes ? s->code_words : 0))
            return gms_fail(GMS_ERR_INVALID, "gms_slam_update_per_particle: the pose refinement's tables do not fit the LDS for a scan of %d beams", B);
        drawn = true;
    }
    gms_launch_slam_particle(pf, dev_beams, B, sb, !on_demand, sample_motion && !drawn ? &mo : nullptr, skip_update ? 0 : 1, s->code_words);   // :90, :99, :102-107
    pf->have_global = 0;
    pf->stats_current = 0;
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

// SLAM.update(z, u) on a device-resident scan (SLAM.java:80-131)
int gms_slam_update_per_particle_dev(gms_slam *s, const gms_beam *dev_beams, int32_t B, int32_t sample_motion, double d_center, double d_theta,
                                     uint64_t seed, uint64_t sequence, gms_pf_stats *stats) {
    REQUIRE(s, "null handle");
    if (s->pf->offset != 0 || s->pf->n_global != s->pf->n)
        return gms_fail(GMS_ERR_STATE, "a shard of a filter: gms_slam_update_local_dev, then the weight exchange (gms_pf_local_partials / apply_partials / import_global)");
    int rc = slam_update_local(s, dev_beams, B, sample_motion, d_center, d_theta, seed, sequence);
    if (rc) return rc;
    return gms_pf_normalize(s->pf, stats);                                                                 // :100, :110-124 (stats: synchronises)
}

// ... and for one rank's block of a sharded filter: the local half of update() -- motion sample (keyed by the GLOBAL particle index),
// field, weight, map update of this block's particles; weightSum / strongest / normalise / Neff follow from the exchange of the block
// partials (gms_pf_local_partials -> all-reduce -> gms_pf_apply_partials -> all-gather -> gms_pf_import_global), as for a sharded gms_pf
int gms_slam_update_local_dev(gms_slam *s, const gms_beam *dev_beams, int32_t B, int32_t sample_motion, double d_center, double d_theta,
                              uint64_t seed, uint64_t sequence) {
    return slam_update_local(s, dev_beams, B, sample_motion, d_center, d_theta, seed, sequence);
}
int gms_slam_update_local(gms_slam *s, const gms_beam *beams, int32_t B, int32_t sample_motion, double d_center, double d_theta, uint64_t seed,
                          uint64_t sequence) {
    REQUIRE(s && beams, "null argument");
    int rc = gms_stage_beams(s->map, beams, B);
    if (rc) return rc;
    return slam_update_local(s, s->map->d_beams, B, sample_motion, d_center, d_theta, seed, sequence);
}

int gms_slam_update_per_particle(gms_slam *s, const gms_beam *beams, int32_t B, int32_t sample_motion, double d_center, double d_theta,
                                 uint64_t seed, uint64_t sequence, gms_pf_stats *stats) {
    REQUIRE(s && beams, "null argument");
    int rc = gms_stage_beams(s->map, beams, B);
    if (rc) return rc;
    return gms_slam_update_per_particle_dev(s, s->map->d_beams, B, sample_motion, d_center, d_theta, seed, sequence, stats);
}

// likelihoodData as the last resample() left it, for whoever reads it before the next update (downloads; a second resample())
static int slam_lik_current(gms_slam *s) {
    if (s->lik_from_codes) {                                                                               // computeLikelihoodMap(p.m) of the last update (:93), late
        gms_launch_slam_likelihood_codes(s->map, gms_slam_bufs(s), s->code_words, s->n, 1);
        s->lik_from_codes = 0;
        HIPCHK(hipGetLastError());
        return GMS_OK;
    }
    if (!s->lik_behind) return GMS_OK;
    gms_launch_slam_gather(s->pf, gms_slam_bufs(s), 2, s->d_idx_lik, nullptr, s->code_words);              // GridMap.java:121, late (if that resample() drew)
    s->lik_behind = 0;
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

// SLAM.resample() (SLAM.java:133-153): the systematic draw over the particles' weights, then every slot's deep copy into the OTHER
// generation of the maps, which the draw makes current (:152); fraction >= 0: only where Neff < fraction * n (GridMapApp.java:185-186),
// decided on the device -- where the rule says no, nothing is drawn, no generation changes and the copy kernels return at once (the
// reference does nothing either)
static int slam_resample(gms_slam *s, double r01, double fraction, int32_t *indices, int32_t *n_ambiguous) {
    gms_map *m = s->map;
    HIPCHK(hipSetDevice(m->device));
    int rc = s->lik_behind ? slam_lik_current(s) : GMS_OK;                                                 // (two resample() calls in a row)
    if (rc) return rc;
    s->pf->d_epoch2 = s->d_epoch;                      // the draw counts itself (k_resample): the maps' generation follows it
    rc = fraction >= 0.0 ? gms_pf_resample_if(s->pf, &r01, fraction)
                         : gms_pf_resample(s->pf, &r01, indices, n_ambiguous);                             // :136-145 + pose, weight (:42-43)
    s->pf->d_epoch2 = nullptr;
    if (rc) return rc;
    const SlamBufs sb = gms_slam_bufs(s);
    if (s->lik_from_codes) {
        // likelihoodData is the field of plane 1 of a particle's class planes: they travel with logData, and so does it
        gms_launch_slam_gather(s->pf, sb, 1, s->pf->d_idx, nullptr, s->code_words);
    } else if (s->lazy_lik) {
        // logData now (GridMap.java:120); likelihoodData (:121) when it is asked for: SLAM.update starts with computeLikelihoodMap of
        // every particle (:93), which overwrites every cell of it -- nothing on the path ever reads the copies
        gms_launch_slam_gather(s->pf, sb, 1, s->pf->d_idx, s->d_idx_lik, s->code_words);                   // (keeps the indices for that)
        s->lik_behind = 1;
    } else {
        gms_launch_slam_gather(s->pf, sb, 3, s->pf->d_idx, nullptr, s->code_words);                        // :44
    }
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

int gms_slam_resample_maps(gms_slam *s, double r01, int32_t *indices, int32_t *n_ambiguous) {
    REQUIRE(s, "null handle");
    if (s->pf->offset != 0 || s->pf->n_global != s->pf->n)
        return gms_fail(GMS_ERR_STATE, "a shard of a filter: gms_slam_shard_draw / export / gather move its maps (the sources may live on other ranks)");
    return slam_resample(s, r01, -1.0, indices, n_ambiguous);
}

// ---- resample() of a sharded filter -------------------------------------------------------------------------------------------
// 1. the draw for this rank's slots from the gathered population (every rank: the same r01): poses and weights are filled from it, the
//    maps' generation advances if it drew.  sources[n_local] = the GLOBAL index of every slot's source particle; *did as the rule decided.
int gms_slam_shard_draw(gms_slam *s, double r01, double fraction, int32_t *did, int32_t *sources) {
    REQUIRE(s && did && sources, "null argument");
    gms_map *m = s->map;
    HIPCHK(hipSetDevice(m->device));
    if (s->lik_behind || !s->lik_from_codes) {
        // a shard's likelihoodData is never copied: it is the field of plane 1 of the class planes, which travel.  Whatever was written
        // out for a reader is dropped here (the planes still define it).
        s->lik_behind = 0;
        s->lik_from_codes = 1;
    }
    s->pf->d_epoch2 = s->d_epoch;
    int rc = fraction >= 0.0 ? gms_pf_resample_if(s->pf, &r01, fraction) : gms_pf_resample(s->pf, &r01, nullptr, nullptr);
    s->pf->d_epoch2 = nullptr;
    if (rc) return rc;
    int32_t e[2] = {0, 0};
    HIPCHK(hipStreamSynchronize(m->stream));
    HIPCHK(hipMemcpy(e, s->d_epoch, sizeof(e), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(sources, s->pf->d_idx, (size_t)s->n * sizeof(int32_t), hipMemcpyDeviceToHost));
    *did = e[1];
    return GMS_OK;
}
// doubles per record of a particle: logData + its two class planes
int gms_slam_record_doubles(const gms_slam *s, int64_t *doubles) {
    REQUIRE(s && doubles, "null argument");
    REQUIRE(s->d_code[0], "gms_slam_record_doubles: the class planes are not kept on this handle");
    *doubles = s->map->gd.cells + s->code_words;
    return GMS_OK;
}
// 2. the records of `count` local particles (indices into this rank's block) as they stood BEFORE the draw, into dev_dst
//    [count][record_doubles]: what the ranks whose slots drew them receive
int gms_slam_shard_export(gms_slam *s, const int32_t *local_indices, int32_t count, double *dev_dst) {
    REQUIRE(s && (count == 0 || (local_indices && dev_dst)), "null argument");
    REQUIRE(s->d_plan && count >= 0 && count <= s->n, "gms_slam_shard_export: not a shard, or more records than particles");
    if (count == 0) return GMS_OK;
    gms_map *m = s->map;
    HIPCHK(hipSetDevice(m->device));
    for (int32_t k = 0; k < count; k++) REQUIRE(local_indices[k] >= 0 && local_indices[k] < s->n, "gms_slam_shard_export: particle index out of range");
    HIPCHK(hipMemcpyAsync(s->d_plan, local_indices, (size_t)count * sizeof(int32_t), hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));           // (the host array may be pageable: the copy must not outlive the call)
    gms_launch_slam_export_records(s->pf, gms_slam_bufs(s), s->d_plan, count, s->code_words, dev_dst);
    HIPCHK(hipGetLastError());
    return GMS_OK;
}
// 3. the copies: slot m of the new generation <- local particle src_local[m] of the previous one, or, where src_local[m] < 0, record
//    recv_pos[m] of dev_recv (the records this rank received).  Both arrays [n_local], host.
int gms_slam_shard_gather(gms_slam *s, const int32_t *src_local, const int32_t *recv_pos, const double *dev_recv) {
    REQUIRE(s && src_local && recv_pos, "null argument");
    REQUIRE(s->d_plan, "gms_slam_shard_gather: not a shard");
    gms_map *m = s->map;
    HIPCHK(hipSetDevice(m->device));
    for (int32_t k = 0; k < s->n; k++) {
        REQUIRE(src_local[k] < s->n, "gms_slam_shard_gather: local source out of range");
        REQUIRE(src_local[k] >= 0 || (dev_recv && recv_pos[k] >= 0), "gms_slam_shard_gather: a remote source without a received record");
    }
    HIPCHK(hipMemcpyAsync(s->d_plan + s->n, src_local, (size_t)s->n * sizeof(int32_t), hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(s->d_plan + 2 * (size_t)s->n, recv_pos, (size_t)s->n * sizeof(int32_t), hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    gms_launch_slam_shard_gather(s->pf, gms_slam_bufs(s), s->d_plan + s->n, s->d_plan + 2 * (size_t)s->n, dev_recv, s->code_words);
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

int gms_slam_resample_maps_if(gms_slam *s, double r01, double fraction) {
    REQUIRE(s, "null handle");
    if (s->pf->offset != 0 || s->pf->n_global != s->pf->n)
        return gms_fail(GMS_ERR_STATE, "a shard of a filter: gms_slam_shard_draw / export / gather move its maps (the sources may live on other ranks)");
    REQUIRE(fraction >= 0.0, "gms_slam_resample_maps_if: fraction must be non-negative");
    return slam_resample(s, r01, fraction, nullptr, nullptr);
}

static int slam_map_xfer(gms_slam *s, int32_t i, int32_t count, double *dev_base, double *host, bool to_device) {
    gms_map *m = s->map;
    const size_t cells = (size_t)m->gd.cells;
    double *dev = dev_base + (size_t)i * cells;
    const size_t bytes = (size_t)count * cells * sizeof(double);
    if (to_device) HIPCHK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, m->stream));
    else HIPCHK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, m->stream));
    return GMS_OK;
}

int gms_slam_download_map(gms_slam *s, int32_t i, double *log_data, double *lik) {          // Particle.m (SLAM.java:33)
    REQUIRE(s && i >= 0 && i < s->n, "gms_slam_download_map: particle index out of range");
    HIPCHK(hipSetDevice(s->map->device));
    int rc = lik ? slam_lik_current(s) : GMS_OK;
    int32_t cur = 0;
    if (!rc) rc = slam_host_gen(s, &cur);
    if (!rc && log_data) rc = slam_map_xfer(s, i, 1, s->d_log[cur], log_data, false);
    if (!rc && lik) rc = slam_map_xfer(s, i, 1, s->d_lik[cur], lik, false);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(s->map->stream));
    return GMS_OK;
}

int gms_slam_download_maps(gms_slam *s, double *log_all, double *lik_all) {
    REQUIRE(s, "null handle");
    HIPCHK(hipSetDevice(s->map->device));
    int rc = lik_all ? slam_lik_current(s) : GMS_OK;
    int32_t cur = 0;
    if (!rc) rc = slam_host_gen(s, &cur);
    if (!rc && log_all) rc = slam_map_xfer(s, 0, s->n, s->d_log[cur], log_all, false);
    if (!rc && lik_all) rc = slam_map_xfer(s, 0, s->n, s->d_lik[cur], lik_all, false);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(s->map->stream));
    return GMS_OK;
}

int gms_slam_upload_map(gms_slam *s, int32_t i, const double *log_data, const double *lik) {
    REQUIRE(s && i >= 0 && i < s->n, "gms_slam_upload_map: particle index out of range");
    HIPCHK(hipSetDevice(s->map->device));
    int rc = lik ? slam_lik_current(s) : GMS_OK;                  // (the other slots' fields first, then this one's over its copy)
    int32_t cur = 0;
    if (!rc) rc = slam_host_gen(s, &cur);
    if (!rc && log_data) {
        rc = slam_map_xfer(s, i, 1, s->d_log[cur], const_cast<double *>(log_data), true);
        if (!rc && s->d_code[0])                                  // the slot's class plane 0 follows its logData (plane 1, its field's, does not)
            gms_launch_slam_codes_from_log(s->map, gms_slam_bufs(s), i, 1, s->code_words);
    }
    if (!rc && lik) rc = slam_map_xfer(s, i, 1, s->d_lik[cur], const_cast<double *>(lik), true);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(s->map->stream));
    return GMS_OK;
}

// GridMapApp.calculateCombined (J/app/GridMapApp.java:439-458) over the particles' maps, into the handle's own GridMap:
// read it with gms_map_download_log / gms_map_download_likelihood on the map of gms_slam_handles.
int gms_slam_combined(gms_slam *s) {
    REQUIRE(s, "null handle");
    gms_map *m = s->map;
    HIPCHK(hipSetDevice(m->device));
    gms_ensure_lik(m);
    gms_flush_apply(m);
    gms_launch_slam_combine(m, gms_slam_bufs(s), s->n);                                       // :441-455
    m->need_full_build = 1; m->fac_current = 0;
    HIPCHK(hipGetLastError());
    return gms_map_build_likelihood(m);                                                       // :457
}

// The cell walk of SLAM.update's integrateObservation(p.m, z, p.pose) for particle i at its current pose, as k_slam_particle walks and
// classifies it, written out instead of counted (tests): gms_map_trace_scan's layout.
int gms_slam_trace_scan(gms_slam *s, int32_t i, const gms_beam *beams, int32_t B, int32_t *cells_xy, uint8_t *classes, int32_t cap, int32_t *counts) {
    REQUIRE(s && beams && counts, "null argument");
    REQUIRE(i >= 0 && i < s->n, "gms_slam_trace_scan: particle index out of range");
    REQUIRE(B >= 0 && B <= s->map->max_beams && cap >= 0, "gms_slam_trace_scan: beam count or capacity out of range");
    gms_map *m = s->map;
    HIPCHK(hipSetDevice(m->device));
    int rc = gms_stage_beams(m, beams, B);
    if (rc) return rc;
    int32_t *d_cells = nullptr, *d_counts = nullptr;
    uint8_t *d_cls = nullptr;
    const size_t n = (size_t)B * (size_t)cap;
    bool ok = hipMalloc(&d_cells, (n ? n : 1) * 2 * sizeof(int32_t)) == hipSuccess && hipMalloc(&d_cls, n ? n : 1) == hipSuccess &&
              hipMalloc(&d_counts, (size_t)(B ? B : 1) * sizeof(int32_t)) == hipSuccess;
    if (ok) {
        gms_launch_slam_trace(s->pf, m->d_beams, B, i, d_cells, d_cls, cap, d_counts);
        ok = hipGetLastError() == hipSuccess;
        if (ok && B) ok = hipMemcpyAsync(counts, d_counts, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, m->stream) == hipSuccess;
        if (ok && n && cells_xy) ok = hipMemcpyAsync(cells_xy, d_cells, n * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, m->stream) == hipSuccess;
        if (ok && n && classes) ok = hipMemcpyAsync(classes, d_cls, n, hipMemcpyDeviceToHost, m->stream) == hipSuccess;
        ok = hipStreamSynchronize(m->stream) == hipSuccess && ok;
    }
    hipFree(d_cells); hipFree(d_cls); hipFree(d_counts);
    if (!ok) return gms_fail(GMS_ERR_HIP, "gms_slam_trace_scan: device allocation, launch or copy failed");
    return GMS_OK;
}

int gms_slam_copies(const gms_slam *s, int64_t *maps_copied) {
    REQUIRE(s && maps_copied, "null argument");
    int64_t draws = 0;
    gms_slam *sm = const_cast<gms_slam *>(s);
    HIPCHK(hipSetDevice(sm->map->device));
    int rc = slam_host_gen(sm, nullptr, &draws);               // (the draws are counted on the device: a conditional resample() may not have run)
    if (rc) return rc;
    *maps_copied = s->copies_base + draws * s->n;
    return GMS_OK;
}

}  // extern "C"
