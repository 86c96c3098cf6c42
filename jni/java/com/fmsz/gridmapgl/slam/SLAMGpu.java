package com.fmsz.gridmapgl.slam;

import java.util.ArrayList;

import com.fmsz.gridmapgl.app.Util;
import com.fmsz.gridmapgl.slam.GridMap.GridMapData;

/**
 * {@link SLAM} (SLAM.java:26-204) with every Particle -- pose, weight AND its GridMapData -- resident on the GPU: update() scores a
 * particle against its own likelihood field and integrates the scan into its own map at its own pose (SLAM.java:88-107), resample()
 * deep-copies the surviving particles' maps on the device (SLAM.java:41-45 -> GridMap.createMapData(other)).  A subclass, so it can
 * stand where GridMapApp holds its SLAM (GridMapApp.java:122); the Java-side particle list of the superclass is a snapshot that
 * getParticles() refreshes (poses and weights always; a particle's maps when asked for with mapOf(i)).
 * Not the reference's: findBestPoseOptim (SLAM.java:97; BOBYQA on an objective that is 0 / NaN) is left out, and the motion-model
 * draw of Odometry.apply (unseeded Well1024a) is Philox(seed; particle slot, frame number) on the device.
 * UNBUILT in the image this was written in (no JDK): see INTEGRATION.md.
 */
public class SLAMGpu extends SLAM {
    private final long handle;
    private final int n;
    private final float[] poses;
    private final double[] weights;
    private final double[] stats = new double[3];       // weightSum, neff, strongest
    private final long seed;
    private long frame;
    private final ArrayList<Particle> mine;

    private long comm;              // != 0: this object is one rank's block of a filter sharded over the GPUs of a node (joinSharded)
    private long shardOffset;

    /**
     * One rank's block of a filter whose particles AND maps are split over `world` GPUs, one JVM per GPU: numParticles is this rank's share
     * (a multiple of 256); id128 comes from NativeSlam.commUniqueId on one rank and reaches the others by the host's own channel. update(z, u)
     * and resample() then run over all ranks (every rank passes the same scan and odometry; resample()'s Math.random() must be the same number
     * on every rank: resampleSharded(r01)).
     */
    public static SLAMGpu joinSharded(int numParticlesLocal, long seed, int device, byte[] id128, int rank, int world) {
        SLAMGpu s = new SLAMGpu(numParticlesLocal, seed, device, (long) rank * numParticlesLocal, (long) world * numParticlesLocal);
        s.comm = NativeSlam.commCreate(id128, rank, world, device);
        return s;
    }

    private SLAMGpu(int numParticlesLocal, long seed, int device, long offset, long nGlobal) {
        super();
        this.mine = new ArrayList<>(numParticlesLocal);
        for (int i = 0; i < numParticlesLocal; i++) {
            Particle p = new Particle(new Pose(0, 0, 0), getGridMap().createMapData(null));
            p.weight = 1.0 / nGlobal;
            mine.add(p);
        }
        GridMap g = getGridMap();
        float res = g.getResolution();
        double sigma = Math.sqrt(0.05 / res);
        double[] kernel = Util.generateGaussianKernel(sigma, (int) Math.ceil(sigma * 3));
        this.n = numParticlesLocal;
        this.seed = seed;
        this.shardOffset = offset;
        this.handle = NativeSlam.pmCreateShard(g.getWorldSize().getX(), g.getWorldSize().getY(), res, g.getPosition().getX(), g.getPosition().getY(),
                Util.logOdds(SensorModel.P_FREE), Util.logOdds(SensorModel.P_OCCUPPIED), kernel, 0, device, numParticlesLocal, offset, nGlobal);
        this.poses = new float[3 * numParticlesLocal];
        this.weights = new double[numParticlesLocal];
    }

    /** resample() of a sharded filter: r01 stands for Math.random() and must be the same on every rank; fraction &lt; 0: unconditional */
    public boolean resampleSharded(double r01, double fraction) { return NativeSlam.pmResampleSharded(handle, comm, r01, fraction); }

    public SLAMGpu(int numParticles, long seed, int device) {
        super();                                                                    // SLAM.java:56-62 (its Java-side maps stay blank)
        // the superclass's list has SLAM.numParticles = 500 entries whatever numParticles is (a private field, SLAM.java:50): the
        // Java-side snapshot of THIS filter is a list of its own, one Particle per device particle
        this.mine = new ArrayList<>(numParticles);
        for (int i = 0; i < numParticles; i++) {                                    // SLAM.java:35-38,68-71
            Particle p = new Particle(new Pose(0, 0, 0), getGridMap().createMapData(null));
            p.weight = 1.0 / numParticles;
            mine.add(p);
        }
        GridMap g = getGridMap();
        float res = g.getResolution();
        double sigma = Math.sqrt(0.05 / res);                                       // GridMap.java:94-95
        double[] kernel = Util.generateGaussianKernel(sigma, (int) Math.ceil(sigma * 3));
        this.n = numParticles;
        this.seed = seed;
        this.handle = NativeSlam.pmCreate(g.getWorldSize().getX(), g.getWorldSize().getY(), res, g.getPosition().getX(), g.getPosition().getY(),
                Util.logOdds(SensorModel.P_FREE), Util.logOdds(SensorModel.P_OCCUPPIED), kernel, 0, device, numParticles);
        this.poses = new float[3 * numParticles];
        this.weights = new double[numParticles];
    }

    @Override
    public void reset() {                                                           // SLAM.java:65-77
        super.reset();
        if (handle != 0) NativeSlam.pmReset(handle);
        frame = 0;
    }

    /**
     * update() refines every particle's pose with GridMap.findBestPose (GridMap.java:319-346) against the particle's own likelihood
     * field before weighting it: the search SLAM.java:96 keeps commented out beside findBestPoseOptim (:97)
     */
    public void setRefine(boolean on) { NativeSlam.pmSetRefine(handle, on); }

    /** update(z, u) (SLAM.java:80-131); returns Neff */
    @Override
    public double update(Observation z, Odometry u) {
        if (comm != 0) {
            NativeSlam.pmUpdateSharded(handle, comm, NativeSlam.flatten(z), z.getNumberOfMeasurements(), u != null, u != null ? u.dCenter : 0.0,
                    u != null ? u.dTheta : 0.0, seed, frame++, stats);
            return stats[1];
        }
        NativeSlam.pmUpdate(handle, NativeSlam.flatten(z), z.getNumberOfMeasurements(), u != null, u != null ? u.dCenter : 0.0, u != null ? u.dTheta : 0.0,
                seed, frame++, stats);
        return stats[1];
    }

    @Override
    public void resample() { NativeSlam.pmResample(handle, Math.random()); }        // SLAM.java:133-153

    /** {@code if (neff < fraction * n) resample()} (GridMapApp.java:185-186) without reading Neff back: update(z, u) then this is one revolution on the device */
    public void resampleIf(double fraction) { NativeSlam.pmResampleIf(handle, Math.random(), fraction); }

    @Override
    public Pose getWeightedPose() {                                                 // SLAM.java:165-178
        float[] o = new float[3];
        NativeSlam.pmWeightedPose(handle, o);
        return new Pose(o[0], o[1], o[2]);
    }

    @Override
    public double calculateNeff() { return stats[1]; }                              // SLAM.java:180-190 (of the last update)

    /** getParticles() (SLAM.java:192): poses and weights refreshed from the device; a particle's maps are filled by mapOf(i) */
    @Override
    public ArrayList<Particle> getParticles() {
        ArrayList<Particle> list = mine;
        NativeSlam.pmGetParticles(handle, poses, weights, n);
        for (int i = 0; i < n; i++) {
            Particle p = list.get(i);
            p.pose = new Pose(poses[3 * i], poses[3 * i + 1], poses[3 * i + 2]);
            p.weight = weights[i];
        }
        return list;
    }

    /** Particle i with its GridMapData brought over from the device (what the renderer reads: GridMapApp.java:384-393) */
    public Particle mapOf(int i) {
        Particle p = getParticles().get(i);
        NativeSlam.pmDownloadMap(handle, i, p.m.logData, p.m.likelihoodData);
        return p;
    }

    @Override
    public Particle getStrongestParticle() { return mapOf((int) stats[2]); }        // SLAM.java:196

    /** GridMapApp.calculateCombined (GridMapApp.java:439-458) on the device, into `combined` */
    public void calculateCombined(GridMapData combined) { NativeSlam.pmCombined(handle, combined.logData, combined.likelihoodData); }

    public void dispose() {
        NativeSlam.pmDestroy(handle);
        if (comm != 0) NativeSlam.commDestroy(comm);
    }
}
