// gridmapslam.hpp -- C++ host-side mirror of the reference's Java class surface over the C-ABI of
// include/gridmapslam.h.  Header-only, C++17; link with -lgridmapslam (and the HIP runtime it needs).
//
// The reference is compiled Java and no JVM exists in this image, so the host side above the C-ABI is
// written in C++ with the reference's names, argument meaning and error behaviour
// (J/ = java/GridMapGL/src/main/java/com/fmsz/gridmapgl/ in the reference tree):
//
//   gms::Pose             J/slam/Pose.java:21-35
//   gms::Observation      J/slam/Observation.java:29-106   (Measurement = gms_beam)
//   gms::GridMap          J/slam/GridMap.java:47-432       (one object = GridMap + its GridMapData)
//   gms::ParticleFilter   J/slam/ParticleFilter.java:19-84 (resample semantics of SLAM.resample)
//   gms::SLAM             J/slam/SLAM.java:26-204          (one shared map, N poses: SURVEY.md fact 3)
//   gms::SlamParticleMaps J/slam/SLAM.java:26-204          (the reference's own shape: every Particle owns a GridMapData)
//
// Java signals nothing on this path except ArrayIndexOutOfBounds from getRawAt; here every failing
// C-ABI call throws gms::Error carrying gms_last_error().
#pragma once

#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <array>
#include <vector>

#include "gridmapslam.h"

namespace gms {

struct Error : std::runtime_error {
    int code;
    Error(int c, const char *msg) : std::runtime_error(std::string("libgridmapslam: ") + msg), code(c) {}
};

inline void check(int rc) {
    if (rc != GMS_OK) throw Error(rc, gms_last_error());
}

/** Pose{float x, y, theta} (J/slam/Pose.java:21-35) */
struct Pose {
    float x = 0, y = 0, theta = 0;
    Pose() = default;
    Pose(float x_, float y_, float theta_) : x(x_), y(y_), theta(theta_) {}
};

/** One LIDAR revolution (J/slam/Observation.java) */
class Observation {
public:
    using Measurement = gms_beam;

    /** addMeasurement(angle, distance, wasHit): localX = distance * cos(angle) (Observation.java:44-51,87-89) */
    void addMeasurement(float angle, float distance, bool wasHit) {
        Measurement m{};
        m.distance = (double)distance;
        m.local_x = (double)distance * std::cos((double)angle);
        m.local_y = (double)distance * std::sin((double)angle);
        m.hit = wasHit ? 1 : 0;
        measurements_.push_back(m);
    }
    /** Measurement(x, y, wasHit, dummy) in the robot frame (Observation.java:69-76) */
    void addLocal(double x, double y, bool wasHit) {
        Measurement m{};
        m.local_x = x; m.local_y = y; m.distance = std::sqrt(x * x + y * y); m.hit = wasHit ? 1 : 0;
        measurements_.push_back(m);
    }
    void addMeasurement(const Measurement &m) { measurements_.push_back(m); }
    const std::vector<Measurement> &getMeasurements() const { return measurements_; }
    int getNumberOfMeasurements() const { return (int)measurements_.size(); }
    void reset() { measurements_.clear(); }

private:
    std::vector<Measurement> measurements_;
};

class ParticleFilter;

/** GridMap(width, height, resolution, position) + createMapData(null) (GridMap.java:80-132) */
class GridMap {
public:
    GridMap(float width, float height, float resolution, float posX, float posY, int device = 0) {
        check(gms_params_default(&params_, width, height, resolution, posX, posY));
        params_.device = device;
        check(gms_map_create(&params_, &h_));
        check(gms_map_get_size(h_, &w_, &hgt_, nullptr));
    }
    explicit GridMap(const gms_params &p) : params_(p) {
        check(gms_map_create(&params_, &h_));
        check(gms_map_get_size(h_, &w_, &hgt_, nullptr));
    }
    ~GridMap() { gms_map_destroy(h_); }
    GridMap(const GridMap &) = delete;
    GridMap &operator=(const GridMap &) = delete;

    /** createMapData(other): a new map with the same geometry holding a copy of both arrays (GridMap.java:106-124) */
    GridMap *createMapData() const {
        GridMap *m = new GridMap(params_);
        check(gms_map_copy(m->h_, h_));
        return m;
    }
    void reset() { check(gms_map_reset(h_)); }                                               // :129-132
    double getRawAt(int x, int y) { double v; check(gms_map_get_raw_at(h_, 0, x, y, &v, nullptr)); return v; }   // :134
    double getProbAt(int x, int y) { double v; check(gms_map_get_raw_at(h_, 0, x, y, nullptr, &v)); return v; }  // :138
    double getRawAt(float px, float py) { double v; check(gms_map_get_at_point(h_, 0, px, py, &v, nullptr)); return v; }      // :142 (Vec2 point)
    double getLikelihood(float px, float py) { double v; check(gms_map_get_at_point(h_, 0, px, py, nullptr, &v)); return v; } // :150
    bool pointInMap(float px, float py) const {                                              // :164-170
        const float tx = (px - params_.pos_x) / params_.resolution, ty = (py - params_.pos_y) / params_.resolution;
        return !(tx < 0 || ty < 0 || tx >= (float)w_ || ty >= (float)hgt_);
    }
    /** integrateObservation(map, obs, pose) (:173-191) */
    void integrateObservation(const Observation &obs, const Pose &p) {
        const float pose[3] = {p.x, p.y, p.theta};
        check(gms_map_integrate(h_, obs.getMeasurements().data(), obs.getNumberOfMeasurements(), pose));
    }
    /** applyMeasurement(map, startX, startY, endX, endY, measuredDistance, wasHit) (:194-228) */
    void applyMeasurement(float startX, float startY, float endX, float endY, float measuredDistance, bool wasHit) {
        check(gms_map_apply_ray(h_, startX, startY, endX, endY, measuredDistance, wasHit ? 1 : 0));
    }
    /** computeLikelihoodMap(map) (:233-250) */
    void computeLikelihoodMap() { check(gms_map_build_likelihood(h_)); }
    /** integrateObservation + computeLikelihoodMap, rebuilding only what the scan changed */
    void update(const Observation &obs, const Pose &p) {
        const float pose[3] = {p.x, p.y, p.theta};
        check(gms_map_update(h_, obs.getMeasurements().data(), obs.getNumberOfMeasurements(), pose));
    }
    /** probabilityOf(map, obs, pose) (:261-294) */
    double probabilityOf(const Observation &obs, const Pose &p);
    /** findBestPose(map, obs, startPose) (:319-346) */
    Pose findBestPose(const Observation &obs, const Pose &start);

    /** what the dirty-tile rebuilds of computeLikelihoodMap did since the last call: 64 x 32-cell tiles {left alone, constants
     *  kept, constants written, blurred}; keepCounting = false stops the census (gms_map_tile_stats) */
    std::array<int64_t, 4> tileStats(bool keepCounting = true) {
        std::array<int64_t, 4> o{};
        check(gms_map_tile_stats(h_, keepCounting ? 1 : 0, o.data()));
        return o;
    }

    std::vector<double> logData() { std::vector<double> v((size_t)w_ * hgt_); check(gms_map_download_log(h_, v.data())); return v; }
    std::vector<double> likelihoodData() { std::vector<double> v((size_t)w_ * hgt_); check(gms_map_download_likelihood(h_, v.data())); return v; }
    void setLogData(const std::vector<double> &v) { check(gms_map_upload_log(h_, v.data())); }

    float getResolution() const { return params_.resolution; }                               // :426-428
    std::pair<float, float> getPosition() const { return {params_.pos_x, params_.pos_y}; }   // :430-432
    std::pair<float, float> getWorldSize() const { return {(float)w_ * params_.resolution, (float)hgt_ * params_.resolution}; }   // :88,422
    int getGridWidth() const { return w_; }
    int getGridHeight() const { return hgt_; }
    gms_map *handle() { return h_; }

private:
    gms_params params_{};
    gms_map *h_ = nullptr;
    int32_t w_ = 0, hgt_ = 0;
};

/** ParticleFilter(numberOfParticles) (J/slam/ParticleFilter.java:43), bound to the map it scores against */
class ParticleFilter {
public:
    struct Particle {               // ParticleFilter.Particle (ParticleFilter.java:21-38)
        double weight;
        Pose pose;
    };

    ParticleFilter(GridMap &map, int numberOfParticles) : n_(numberOfParticles) {
        check(gms_pf_create(map.handle(), numberOfParticles, &h_));
    }
    ~ParticleFilter() { gms_pf_destroy(h_); }
    ParticleFilter(const ParticleFilter &) = delete;
    ParticleFilter &operator=(const ParticleFilter &) = delete;

    /** getParticles() (ParticleFilter.java:50): a snapshot */
    std::vector<Particle> getParticles() {
        std::vector<float> p((size_t)n_ * 3);
        std::vector<double> w(n_);
        check(gms_pf_get_poses(h_, p.data()));
        check(gms_pf_get_weights(h_, w.data()));
        std::vector<Particle> out(n_);
        for (int i = 0; i < n_; i++) out[i] = Particle{w[i], Pose(p[3 * i], p[3 * i + 1], p[3 * i + 2])};
        return out;
    }
    void setPoses(const std::vector<Pose> &poses) {
        if (poses.size() < (size_t)n_) throw Error(GMS_ERR_INVALID, "SlamParticleMaps::setPoses: fewer poses than particles");
        std::vector<float> p((size_t)n_ * 3);
        for (int i = 0; i < n_; i++) { p[3 * i] = poses[i].x; p[3 * i + 1] = poses[i].y; p[3 * i + 2] = poses[i].theta; }
        check(gms_pf_set_poses(h_, p.data()));
    }
    /** weight[i] = probabilityOf(map, obs, pose[i]) (SLAM.java:99) */
    void score(const Observation &obs) { check(gms_pf_score(h_, obs.getMeasurements().data(), obs.getNumberOfMeasurements())); }
    /** weightSum, strongest, weight /= weightSum, Neff (SLAM.java:87-129) */
    gms_pf_stats normalize() { gms_pf_stats s{}; check(gms_pf_normalize(h_, &s)); return s; }
    /** resample() (ParticleFilter.java:59-82 surface, SLAM.java:133-153 semantics); r01 stands for Math.random() */
    void resample(double r01) { check(gms_pf_resample(h_, &r01, nullptr, nullptr)); }
    void refinePoses(const Observation &obs) { check(gms_pf_refine_poses(h_, obs.getMeasurements().data(), obs.getNumberOfMeasurements())); }
    /** opt-in, not in the reference: normalise from the log-weights (exp(logw - max logw)) instead of the plain product of up to
     *  720 factors, which underflows for nearly every particle of a wide cloud (gms_pf_set_log_normalize) */
    void setLogNormalize(bool on) { check(gms_pf_set_log_normalize(h_, on ? 1 : 0)); }
    Pose getWeightedPose() { float o[3]; check(gms_pf_weighted_pose(h_, o)); return Pose(o[0], o[1], o[2]); }
    int size() const { return n_; }
    gms_pf *handle() { return h_; }

    /** multi-GPU: this rank holds particles [offset, offset + size()) of nGlobal (equal shards in rank order) */
    void setShard(int64_t offset, int64_t nGlobal) { check(gms_pf_set_shard(h_, offset, nGlobal)); }

private:
    gms_pf *h_ = nullptr;
    int n_;
};

/** One rank's RCCL communicator (one process per GPU).  `id` is the 128-byte token of uniqueId(), made by one
 *  rank and handed to the others by whatever channel the host has (a file, a socket, MPI ...). */
class Comm {
public:
    static std::vector<char> uniqueId() { std::vector<char> id(128); check(gms_comm_unique_id(id.data())); return id; }
    Comm(const std::vector<char> &id, int rank, int world, int device) { check(gms_comm_create(&h_, id.data(), rank, world, device)); }
    ~Comm() { gms_comm_destroy(h_); }
    Comm(const Comm &) = delete;
    Comm &operator=(const Comm &) = delete;
    gms_comm *handle() { return h_; }
    /** SLAM.update (SLAM.java:80-131) + `if (neff < fraction * N) resample()` (GridMapApp.java:185-186) for a sharded
     *  filter with device-resident inputs; both exchanges happen inside. */
    void slamUpdate(ParticleFilter &pf, const float *devPoses, const gms_beam *devBeams, int B, double r01,
                    double fraction = 0.5, bool integrate = true) {
        check(gms_slam_update_sharded_dev(pf.handle(), h_, devPoses, devBeams, B, &r01, fraction, integrate ? 1 : 0));
    }

private:
    gms_comm *h_ = nullptr;
};

inline double GridMap::probabilityOf(const Observation &obs, const Pose &p) {
    ParticleFilter pf(*this, 1);
    pf.setPoses({p});
    pf.score(obs);
    return pf.getParticles()[0].weight;
}

inline Pose GridMap::findBestPose(const Observation &obs, const Pose &start) {
    ParticleFilter pf(*this, 1);
    pf.setPoses({start});
    pf.refinePoses(obs);
    return pf.getParticles()[0].pose;
}

/** The particle filter the reference runs (J/slam/SLAM.java), pose proposals being an input */
class SLAM {
public:
    explicit SLAM(int numParticles = 500, float width = 6.0f, float height = 6.0f, float resolution = 0.05f,
                  float posX = -3.0f, float posY = -3.0f)                                   // SLAM.java:50,57
        : gridMap_(width, height, resolution, posX, posY), pf_(gridMap_, numParticles) {
        gridMap_.computeLikelihoodMap();
    }
    /** update(z, u) (SLAM.java:80-131); `poses` are the motion-model samples, dTheta is u.dTheta */
    double update(const Observation &z, const std::vector<Pose> *poses = nullptr, double dTheta = 0.0) {
        const bool skipUpdate = std::fabs(dTheta) > (3.141592653589793 / 180.0) * 30;       // :82
        std::vector<float> p;
        if (poses) {                                                                        // :90
            p.resize(poses->size() * 3);
            for (size_t i = 0; i < poses->size(); i++) { p[3 * i] = (*poses)[i].x; p[3 * i + 1] = (*poses)[i].y; p[3 * i + 2] = (*poses)[i].theta; }
        }
        const double unused_r01 = 0.0;
        gms_pf_stats st{};
        // one call: score (:99), bookkeeping (:100-124), map update at the weighted pose (:102-105, :93); no resample here
        check(gms_slam_update(pf_.handle(), poses ? p.data() : nullptr, z.getMeasurements().data(), z.getNumberOfMeasurements(),
                              &unused_r01, -1.0, skipUpdate ? 0 : 1, &st));
        strongest_ = st.strongest;
        return st.neff;
    }
    void resample(double r01) { pf_.resample(r01); }                                        // :133-153
    Pose getWeightedPose() { return pf_.getWeightedPose(); }                                // :165-178
    double calculateNeff() { gms_pf_stats s{}; check(gms_pf_get_stats(pf_.handle(), &s)); return s.neff; }   // :180-190
    std::vector<ParticleFilter::Particle> getParticles() { return pf_.getParticles(); }     // :192
    int getStrongestParticle() const { return strongest_; }                                 // :196
    GridMap &getGridMap() { return gridMap_; }                                              // :200

private:
    GridMap gridMap_;
    ParticleFilter pf_;
    int strongest_ = 0;
};

/** SLAM as the reference has it (J/slam/SLAM.java:26-204): every Particle owns a pose, a weight AND a GridMapData.  update() scores a
 *  particle against its own likelihood field and integrates the scan into its own map at its own pose (:88-107); resample()
 *  deep-copies the surviving particles' maps (:41-45 -> GridMap.createMapData(other), GridMap.java:106-124).  findBestPoseOptim (:97)
 *  is left out; the motion-model draw of Odometry.apply is Philox(seed; particle slot, sequence). */
class SlamParticleMaps {
public:
    struct Odometry { double dCenter = 0, dTheta = 0; };                                   // J/slam/Odometry.java:31
    struct Particle { double weight; Pose pose; };                                          // SLAM.Particle without its map: mapOf(i)

    explicit SlamParticleMaps(int numParticles = 500, float width = 6.0f, float height = 6.0f, float resolution = 0.05f,
                              float posX = -3.0f, float posY = -3.0f, int maxBeams = 0) : n_(numParticles) {   // SLAM.java:50,57
        gms_params p{};
        check(gms_params_default(&p, width, height, resolution, posX, posY));
        p.max_beams = maxBeams;
        check(gms_slam_create(&p, numParticles, &h_));
        check(gms_slam_handles(h_, &map_, &pf_));
        check(gms_slam_count(h_, nullptr, &w_, &hgt_));
    }
    /** One rank's block [offset, offset + numParticlesLocal) of a filter of numParticlesGlobal particles sharded WITH their maps over
     *  the GPUs of a node (gms_slam_create_shard; blocks of a multiple of 256 particles in rank order): updateSharded / resampleSharded */
    SlamParticleMaps(int numParticlesLocal, long long offset, long long numParticlesGlobal, float width, float height, float resolution,
                     float posX, float posY, int maxBeams = 0, int device = 0) : n_(numParticlesLocal) {
        gms_params p{};
        check(gms_params_default(&p, width, height, resolution, posX, posY));
        p.max_beams = maxBeams;
        p.device = device;
        check(gms_slam_create_shard(&p, numParticlesLocal, offset, numParticlesGlobal, &h_));
        check(gms_slam_handles(h_, &map_, &pf_));
        check(gms_slam_count(h_, nullptr, &w_, &hgt_));
    }
    ~SlamParticleMaps() { gms_slam_destroy(h_); }
    SlamParticleMaps(const SlamParticleMaps &) = delete;
    SlamParticleMaps &operator=(const SlamParticleMaps &) = delete;

    void reset() { check(gms_slam_reset(h_)); }                                             // :65-77
    /** update() refines every particle's pose with GridMap.findBestPose against the particle's own field before weighting it (:96) */
    void setRefine(bool on) { check(gms_slam_set_refine(h_, on ? 1 : 0)); }
    /** update(z, u) (:80-131); returns Neff.  seed / sequence select the motion model's variates (one sequence per frame). */
    double update(const Observation &z, const Odometry &u, uint64_t seed, uint64_t sequence, bool sampleMotion = true) {
        gms_pf_stats st{};
        check(gms_slam_update_per_particle(h_, z.getMeasurements().data(), z.getNumberOfMeasurements(), sampleMotion ? 1 : 0, u.dCenter, u.dTheta,
                                           seed, sequence, &st));
        strongest_ = st.strongest;
        return st.neff;
    }
    /** update(z, u) over all ranks of a sharded filter (every rank: the same scan, odometry, seed and sequence); returns Neff */
    double updateSharded(Comm &comm, const Observation &z, const Odometry &u, uint64_t seed, uint64_t sequence, bool sampleMotion = true) {
        gms_pf_stats st{};
        check(gms_slam_update_sharded_maps(h_, comm.handle(), z.getMeasurements().data(), z.getNumberOfMeasurements(), sampleMotion ? 1 : 0, u.dCenter,
                                           u.dTheta, seed, sequence, &st));
        strongest_ = st.strongest;
        return st.neff;
    }
    /** resample() over all ranks (every rank: the same r01; fraction < 0: unconditional); returns whether it drew */
    bool resampleSharded(Comm &comm, double r01, double fraction = -1.0) {
        int32_t did = 0;
        check(gms_slam_resample_sharded_maps(h_, comm.handle(), r01, fraction, &did));
        return did != 0;
    }
    /** resample() (:133-153); r01 stands for Math.random() */
    void resample(double r01) { check(gms_slam_resample_maps(h_, r01, nullptr, nullptr)); }
    // `if (neff < fraction * n) resample()` (GridMapApp.java:185-186) decided on the device: update + this is a revolution without a host round trip
    void resampleIf(double r01, double fraction = 0.5) { check(gms_slam_resample_maps_if(h_, r01, fraction)); }
    Pose getWeightedPose() { float o[3]; check(gms_pf_weighted_pose(pf_, o)); return Pose(o[0], o[1], o[2]); }      // :165-178
    double calculateNeff() { gms_pf_stats s{}; check(gms_pf_get_stats(pf_, &s)); return s.neff; }                    // :180-190
    std::vector<Particle> getParticles() {                                                  // :192
        std::vector<float> p((size_t)n_ * 3);
        std::vector<double> w(n_);
        check(gms_pf_get_poses(pf_, p.data()));
        check(gms_pf_get_weights(pf_, w.data()));
        std::vector<Particle> out(n_);
        for (int i = 0; i < n_; i++) out[i] = Particle{w[i], Pose(p[3 * i], p[3 * i + 1], p[3 * i + 2])};
        return out;
    }
    void setPoses(const std::vector<Pose> &poses) {
        if (poses.size() < (size_t)n_) throw Error(GMS_ERR_INVALID, "SlamParticleMaps::setPoses: fewer poses than particles");
        std::vector<float> p((size_t)n_ * 3);
        for (int i = 0; i < n_; i++) { p[3 * i] = poses[i].x; p[3 * i + 1] = poses[i].y; p[3 * i + 2] = poses[i].theta; }
        check(gms_pf_set_poses(pf_, p.data()));
    }
    int getStrongestParticle() const { return strongest_; }                                 // :196
    /** Particle i's GridMapData.logData (or .likelihoodData), W * H doubles, row-major x + y * W (SLAM.java:33, GridMap.java:72-74) */
    std::vector<double> mapOf(int i, bool likelihood = false) {
        std::vector<double> out((size_t)w_ * hgt_);
        check(gms_slam_download_map(h_, i, likelihood ? nullptr : out.data(), likelihood ? out.data() : nullptr));
        return out;
    }
    /** GridMapApp.calculateCombined (J/app/GridMapApp.java:439-458): the combined logData */
    std::vector<double> calculateCombined() {
        check(gms_slam_combined(h_));
        std::vector<double> out((size_t)w_ * hgt_);
        check(gms_map_download_log(map_, out.data()));
        return out;
    }
    int size() const { return n_; }
    int width() const { return w_; }
    int height() const { return hgt_; }
    gms_slam *handle() { return h_; }

private:
    gms_slam *h_ = nullptr;
    gms_map *map_ = nullptr;     // owned by h_
    gms_pf *pf_ = nullptr;       // owned by h_
    int n_;
    int32_t w_ = 0, hgt_ = 0;
    int strongest_ = 0;
};

}  // namespace gms
