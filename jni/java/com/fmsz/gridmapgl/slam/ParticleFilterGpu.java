package com.fmsz.gridmapgl.slam;

/**
 * {@link ParticleFilter} (ParticleFilter.java:19-84) with the particles resident on the GPU and the
 * scoring / bookkeeping of SLAM.update (SLAM.java:87-129) next to them.  resample() follows
 * SLAM.resample (i starts at 0, SLAM.java:138), not ParticleFilter.resample's off-by-one (i = 1,
 * ParticleFilter.java:66).
 */
public class ParticleFilterGpu {
    private final long handle;
    private final int numberOfParticles;
    private final float[] poses;
    private final double[] weights;
    private final double[] stats = new double[3];

    public ParticleFilterGpu(GridMapGpu map, int numberOfParticles) {
        this.numberOfParticles = numberOfParticles;
        this.handle = NativeSlam.pfCreate(map.nativeHandle(), numberOfParticles);
        this.poses = new float[3 * numberOfParticles];
        this.weights = new double[numberOfParticles];
    }

    /** getParticles(): a snapshot of {weight, pose} (ParticleFilter.java:50). */
    public ParticleFilter.Particle[] getParticles() {
        NativeSlam.pfGetParticles(handle, poses, weights);
        ParticleFilter.Particle[] out = new ParticleFilter.Particle[numberOfParticles];
        for (int i = 0; i < numberOfParticles; i++)
            out[i] = new ParticleFilter.Particle(weights[i], new Pose(poses[3 * i], poses[3 * i + 1], poses[3 * i + 2]));
        return out;
    }

    public void setPoses(Pose[] p) {
        for (int i = 0; i < numberOfParticles; i++) { poses[3 * i] = p[i].x; poses[3 * i + 1] = p[i].y; poses[3 * i + 2] = p[i].theta; }
        NativeSlam.pfSetPoses(handle, poses);
    }

    /** weight[i] = probabilityOf(map, z, pose[i]); weights normalised; returns Neff (SLAM.java:99-129). */
    public double scoreAndNormalize(Observation z) {
        NativeSlam.pfScore(handle, NativeSlam.flatten(z), z.getNumberOfMeasurements());
        NativeSlam.pfNormalize(handle, stats);
        return stats[1];
    }

    public int strongest() { return (int) stats[2]; }
    public void resample() { NativeSlam.pfResample(handle, Math.random()); }        // SLAM.java:136
    public Pose getWeightedPose() { float[] o = new float[3]; NativeSlam.pfWeightedPose(handle, o); return new Pose(o[0], o[1], o[2]); }
    long nativeHandle() { return handle; }
    public void dispose() { NativeSlam.pfDestroy(handle); }
}
