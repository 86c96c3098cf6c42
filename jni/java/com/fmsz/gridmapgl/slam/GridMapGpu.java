package com.fmsz.gridmapgl.slam;

import com.fmsz.gridmapgl.app.Util;
import glm_.vec2.Vec2;

/**
 * {@link GridMap} with the hot path on the GPU: a SUBCLASS, so that it can stand wherever the reference holds a
 * GridMap (SLAM.java:27,57; GridMapApp's renderer).  Every public method of the path is overridden with the
 * reference's signature (GridMap.java:106-346) and forwards to libgridmapslam.so; the geometry getters
 * (pointInMap, getWorldSize, getResolution, getPosition: GridMap.java:164,422-432) are inherited unchanged.
 *
 * The reference keeps one GridMap ("manager") and many GridMapData (one per particle).  The device holds ONE map:
 * the GridMapData most recently passed in is "bound" to it -- its arrays are the host mirror, downloaded lazily when
 * Java code reads them (render, getRawAt, the serialiser) and uploaded when a different GridMapData is passed.  With
 * the shared-map filter of DESIGN.md section 1 (ParticleFilterGpu) there is exactly one GridMapData and no switching.
 */
public class GridMapGpu extends GridMap {
    private final long handle;
    private final long probe;               // one-particle filter for probabilityOf / findBestPose on a single pose
    private final int width, height;
    private GridMapData bound;              // whose arrays mirror the device map
    private boolean hostStale;              // device is ahead of bound's arrays
    private final double[] two = new double[2];

    public GridMapGpu(float width, float height, float resolution, Vec2 position) { this(width, height, resolution, position, 2048, 0); }

    public GridMapGpu(float width, float height, float resolution, Vec2 position, int maxBeams, int device) {
        super(width, height, resolution, position);
        // the JVM's own Math.log / Math.exp decide the constants and taps (SURVEY.md 9.5)
        double sigma = Math.sqrt(0.05 / resolution);                                   // GridMap.java:94
        double[] kernel = Util.generateGaussianKernel(sigma, (int) Math.ceil(sigma * 3));
        this.width = (int) Math.ceil(width / resolution);                              // GridMap.java:85
        this.height = (int) Math.ceil(height / resolution);
        this.handle = NativeSlam.mapCreate(width, height, resolution, position.getX(), position.getY(),
                Util.logOdds(SensorModel.P_FREE), Util.logOdds(SensorModel.P_OCCUPPIED), kernel, maxBeams, device);
        this.probe = NativeSlam.pfCreate(handle, 1);
    }

    /** makes `map` the GridMapData the device map mirrors */
    private void bind(GridMapData map) {
        if (map == bound) return;
        if (bound != null && hostStale) NativeSlam.mapDownload(handle, bound.logData, bound.likelihoodData);
        NativeSlam.mapUpload(handle, map.logData, map.likelihoodData);
        bound = map;
        hostStale = false;
    }

    /** Brings logData / likelihoodData of the bound map up to date before Java code reads them. */
    public void sync(GridMapData map) {
        bind(map);
        if (hostStale) { NativeSlam.mapDownload(handle, map.logData, map.likelihoodData); hostStale = false; }
    }

    @Override
    public GridMapData createMapData(GridMapData other) {                              // GridMap.java:106-124
        if (other != null) sync(other);
        return super.createMapData(other);     // plain arrays; bound on first use
    }

    @Override
    public void reset(GridMapData map) {                                               // :129-132
        bind(map);
        NativeSlam.mapReset(handle);
        hostStale = true;
    }

    @Override public double getRawAt(GridMapData map, int x, int y) { sync(map); return super.getRawAt(map, x, y); }     // :134
    @Override public double getProbAt(GridMapData map, int x, int y) { sync(map); return super.getProbAt(map, x, y); }   // :138

    @Override
    public double getRawAt(GridMapData map, Vec2 point) {                              // :142-148
        bind(map);
        NativeSlam.mapGetAtPoint(handle, point.getX(), point.getY(), two);
        return two[0];
    }

    @Override
    public double getLikelihood(GridMapData map, Vec2 point) {                         // :150-156
        bind(map);
        NativeSlam.mapGetAtPoint(handle, point.getX(), point.getY(), two);
        return two[1];
    }

    @Override
    public void integrateObservation(GridMapData map, Observation obs, Pose p) {       // :173-191
        bind(map);
        NativeSlam.mapIntegrate(handle, NativeSlam.flatten(obs), obs.getNumberOfMeasurements(), p.x, p.y, p.theta);
        hostStale = true;
    }

    @Override
    public void applyMeasurement(GridMapData map, float startX, float startY, float endX, float endY, float measuredDistance, boolean wasHit) {   // :194-228
        bind(map);
        NativeSlam.mapApplyRay(handle, startX, startY, endX, endY, measuredDistance, wasHit);
        hostStale = true;
    }

    @Override
    public void computeLikelihoodMap(GridMapData map) {                                // :233-250
        bind(map);
        NativeSlam.mapBuildLikelihood(handle);
        hostStale = true;
    }

    @Override
    public double probabilityOf(GridMapData map, Observation obs, Pose p) {            // :261-294
        bind(map);
        return NativeSlam.pfProbabilityOf(probe, NativeSlam.flatten(obs), obs.getNumberOfMeasurements(), p.x, p.y, p.theta);
    }

    @Override
    public Pose findBestPose(GridMapData map, Observation obs, Pose startPose) {       // :319-346
        bind(map);
        float[] o = new float[3];
        NativeSlam.pfFindBestPose(probe, NativeSlam.flatten(obs), obs.getNumberOfMeasurements(), startPose.x, startPose.y, startPose.theta, o);
        // the reference returns startPose itself when no lattice pose scores above 0 (:320,:334)
        return (o[0] == startPose.x && o[1] == startPose.y && o[2] == startPose.theta) ? startPose : new Pose(o[0], o[1], o[2]);
    }

    /** the filter's map update wrote the device map (ParticleFilterGpu.update) */
    void deviceChanged(GridMapData map) { bound = map; hostStale = true; }

    int gridWidth() { return width; }
    int gridHeight() { return height; }
    long nativeHandle() { return handle; }
    public void dispose() { NativeSlam.pfDestroy(probe); NativeSlam.mapDestroy(handle); }
}
