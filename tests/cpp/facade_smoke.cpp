// C++ consumer of the C-ABI through include/gridmapslam.hpp: one small SLAM step on the GPU.
// Built and run by tests/test_cpp_facade.py on the GPU box; prints "ok <neff> <weight of pose 0>".
#include <cstdio>

#include "gridmapslam.hpp"

int main() {
    try {
        gms::SLAM slam(256, 6.4f, 6.4f, 0.05f, -3.2f, -3.2f);
        gms::Observation z;
        for (int b = 0; b < 90; b++) z.addMeasurement((float)(b * 2.0 * 3.14159265358979 / 90.0), 2.0f, true);
        gms::Pose origin(0.f, 0.f, 0.f);
        for (int i = 0; i < 3; i++) slam.getGridMap().update(z, origin);
        std::vector<gms::Pose> poses(256);
        for (int i = 0; i < 256; i++) poses[i] = gms::Pose(0.002f * (i % 16), 0.002f * (i / 16), 0.001f * i);
        const double neff = slam.update(z, &poses, 0.0);
        const double p0 = slam.getGridMap().probabilityOf(z, origin);
        slam.resample(0.5);
        const gms::Pose wp = slam.getWeightedPose();
        if (!(neff >= 1.0 && neff <= 256.0) || !(p0 > 0.0) || !(wp.x == wp.x)) { std::printf("bad values\n"); return 1; }
        // out-of-bounds getRawAt throws in Java (ArrayIndexOutOfBounds): here an Error
        bool threw = false;
        try { slam.getGridMap().getRawAt(100000, 0); } catch (const gms::Error &) { threw = true; }
        if (!threw) { std::printf("no error on bad index\n"); return 1; }
        // the reference's own shape: every particle with a map of its own (SLAM.java:30-47)
        gms::SlamParticleMaps pm(64, 6.0f, 6.0f, 0.05f, -3.0f, -3.0f, 128);
        gms::SlamParticleMaps::Odometry u;
        u.dCenter = 0.05; u.dTheta = 0.02;
        double neff2 = 0.0;
        for (int k = 0; k < 3; k++) neff2 = pm.update(z, u, 7, (uint64_t)k);
        pm.resample(0.3);
        pm.update(z, u, 7, 3);
        pm.resampleIf(0.6, 0.5);                                                            // GridMapApp.java:185-186, decided on the device
        pm.setRefine(true);                                                                 // SLAM.java:96: findBestPose of every particle against its own field
        const auto before = pm.getParticles();
        pm.update(z, u, 7, 4, /*sampleMotion=*/false);
        const auto after = pm.getParticles();
        bool moved = false;
        for (size_t i = 0; i < after.size(); i++) moved = moved || after[i].pose.x != before[i].pose.x || after[i].pose.theta != before[i].pose.theta;
        if (!moved) { std::printf("the pose refinement moved no particle\n"); return 1; }
        pm.setRefine(false);
        const std::vector<double> m0 = pm.mapOf(0), comb = pm.calculateCombined();
        bool touched = false;
        for (double v : m0) touched = touched || v != 0.0;
        if (!(neff2 >= 1.0 && neff2 <= 64.0) || !touched || m0.size() != 120u * 120u || comb.size() != m0.size() || pm.getParticles().size() != 64u) {
            std::printf("bad per-particle-map values\n");
            return 1;
        }
        std::printf("ok %.6f %.6e\n", neff, p0);
        return 0;
    } catch (const gms::Error &e) {
        std::printf("error %d: %s\n", e.code, e.what());
        return 2;
    }
}
