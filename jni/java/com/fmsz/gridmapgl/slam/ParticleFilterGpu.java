package com.fmsz.gridmapgl.slam;

/**
 * {@link ParticleFilter} (ParticleFilter.java:19-84) with the particles resident on the GPU and the whole of
 * SLAM.update (SLAM.java:80-131) next to them: ONE shared map, N poses scored against it (DESIGN.md section 1).
 * A subclass, so it can stand where a ParticleFilter is held; getParticles() / resample() keep their signatures.
 * resample() follows SLAM.resample (i starts at 0, SLAM.java:138), not ParticleFilter.resample's off-by-one
 * (i = 1, ParticleFilter.java:66).
 */
public class ParticleFilterGpu extends ParticleFilter {
    private final GridMapGpu map;
    private final GridMap.GridMapData mapData;
    private final long handle;
    private final int numberOfParticles;
    private final float[] poses;
    private final double[] weights;
    private final double[] stats = new double[3];       // weightSum, neff, strongest
    private long comm;                                  // RCCL communicator of a sharded filter (0: stand-alone)

    public ParticleFilterGpu(GridMapGpu map, GridMap.GridMapData mapData, int numberOfParticles) {
        super(numberOfParticles);
        this.map = map;
        this.mapData = mapData;
        this.numberOfParticles = numberOfParticles;
        this.handle = NativeSlam.pfCreate(map.nativeHandle(), numberOfParticles);
        this.poses = new float[3 * numberOfParticles];
        this.weights = new double[numberOfParticles];
    }

    /** getParticles(): a snapshot of {weight, pose} (ParticleFilter.java:50). */
    @Override
    public ParticleFilter.Particle[] getParticles() {
        NativeSlam.pfGetParticles(handle, poses, weights, numberOfParticles);
        ParticleFilter.Particle[] out = new ParticleFilter.Particle[numberOfParticles];
        for (int i = 0; i < numberOfParticles; i++)
            out[i] = new ParticleFilter.Particle(weights[i], new Pose(poses[3 * i], poses[3 * i + 1], poses[3 * i + 2]));
        return out;
    }

    public void setPoses(Pose[] p) {
        for (int i = 0; i < numberOfParticles; i++) { poses[3 * i] = p[i].x; poses[3 * i + 1] = p[i].y; poses[3 * i + 2] = p[i].theta; }
        NativeSlam.pfSetPoses(handle, poses, numberOfParticles);
    }

    /** findBestPose on every particle before it is weighted (SLAM.java:96-97) */
    public void setRefine(boolean on) { NativeSlam.pfSetRefine(handle, on); }

    /** opt-in, not in the reference: weights from the log-weights (exp(logw - max logw)) instead of the plain product, which
     *  underflows at several hundred beams (gms_pf_set_log_normalize) */
    public void setLogNormalize(boolean on) { NativeSlam.pfSetLogNormalize(handle, on); }

    /**
     * SLAM.update(z, u) (SLAM.java:80-131) followed by its caller's `if (neff < fraction * N) resample()`
     * (GridMapApp.java:185-186) in ONE native call.  `sampled` are the motion-model samples (sampleMotionModel stays on
     * the JVM, SLAM.java:90,155-163; null keeps the current poses); skipUpdate is SLAM.java:82; resampleFraction < 0
     * never resamples.  Returns Neff (SLAM.java:126).
     */
    public double update(Observation z, Pose[] sampled, boolean skipUpdate, double resampleFraction) {
        float[] xyt = null;
        if (sampled != null) {
            for (int i = 0; i < numberOfParticles; i++) { poses[3 * i] = sampled[i].x; poses[3 * i + 1] = sampled[i].y; poses[3 * i + 2] = sampled[i].theta; }
            xyt = poses;
        }
        // A sharded filter resamples the all-gathered population on EVERY rank: the draw must be the same number in every
        // JVM, and Math.random() is not.  updateSharded takes it as an argument.
        if (comm != 0)
            throw new IllegalStateException("sharded filter: call updateSharded(z, sampled, skipUpdate, resampleFraction, r01) with the same r01 on every rank");
        double[] beams = NativeSlam.flatten(z);
        NativeSlam.slamUpdate(handle, xyt, numberOfParticles, beams, z.getNumberOfMeasurements(), Math.random(), resampleFraction, !skipUpdate, stats);
        if (!skipUpdate) map.deviceChanged(mapData);
        return stats[1];
    }

    /** weight[i] = probabilityOf(map, z, pose[i]); weights normalised; returns Neff (SLAM.java:99-129). */
    public double scoreAndNormalize(Observation z) {
        NativeSlam.pfScore(handle, NativeSlam.flatten(z), z.getNumberOfMeasurements());
        NativeSlam.pfNormalize(handle, stats);
        return stats[1];
    }

    /**
     * Multi-GPU, one JVM per GPU: this filter holds particles [rank * n, (rank + 1) * n) of world * n.  id128 comes
     * from NativeSlam.commUniqueId on one rank and reaches the others by whatever channel the host has.  Every rank
     * then calls updateSharded() with its shard of the samples, the same scan and the same resampling draw r01; update()
     * throws on a sharded filter (its Math.random() would differ from JVM to JVM).
     */
    public void joinShardedFilter(byte[] id128, int rank, int world, int device) {
        NativeSlam.pfSetShard(handle, (long) rank * numberOfParticles, (long) world * numberOfParticles);
        comm = NativeSlam.commCreate(id128, rank, world, device);
    }

    /** update() for a sharded filter with the resampling draw r01 (the same on every rank) passed in */
    public double updateSharded(Observation z, Pose[] sampled, boolean skipUpdate, double resampleFraction, double r01) {
        float[] xyt = null;
        if (sampled != null) {
            for (int i = 0; i < numberOfParticles; i++) { poses[3 * i] = sampled[i].x; poses[3 * i + 1] = sampled[i].y; poses[3 * i + 2] = sampled[i].theta; }
            xyt = poses;
        }
        NativeSlam.slamUpdateSharded(handle, comm, xyt, numberOfParticles, NativeSlam.flatten(z), z.getNumberOfMeasurements(), r01, resampleFraction, !skipUpdate, stats);
        if (!skipUpdate) map.deviceChanged(mapData);
        return stats[1];
    }

    public int strongest() { return (int) stats[2]; }                                  // SLAM.getStrongestParticle (:196)
    @Override public void resample() { NativeSlam.pfResample(handle, Math.random()); }  // SLAM.java:133-153
    public Pose getWeightedPose() { float[] o = new float[3]; NativeSlam.pfWeightedPose(handle, o); return new Pose(o[0], o[1], o[2]); }   // SLAM.java:165
    long nativeHandle() { return handle; }
    public void dispose() { if (comm != 0) NativeSlam.commDestroy(comm); NativeSlam.pfDestroy(handle); }
}
