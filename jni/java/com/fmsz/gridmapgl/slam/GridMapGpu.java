package com.fmsz.gridmapgl.slam;

import com.fmsz.gridmapgl.app.Util;
import glm_.vec2.Vec2;

/**
 * Drop-in for the hot-path methods of {@link GridMap} (GridMap.java:80-294): same signatures, the work
 * done by libgridmapslam.so on the GPU.  A maintainer adds this class next to GridMap.java and lets
 * SLAM.java construct it instead; GridMapData keeps its two double[] arrays, filled lazily from the
 * device when the renderer or serialiser asks (GridMap.java:371-388).
 */
public class GridMapGpu {
    private final long handle;
    private final int width, height;
    private boolean hostStale = true;

    public GridMapGpu(float width, float height, float resolution, Vec2 position) {
        // the JVM's own Math.log / Math.exp decide the constants and taps (SURVEY.md 9.5)
        double sigma = Math.sqrt(0.05 / resolution);
        double[] kernel = Util.generateGaussianKernel(sigma, (int) Math.ceil(sigma * 3));
        this.width = (int) Math.ceil(width / resolution);
        this.height = (int) Math.ceil(height / resolution);
        this.handle = NativeSlam.mapCreate(width, height, resolution, position.getX(), position.getY(),
                Util.logOdds(SensorModel.P_FREE), Util.logOdds(SensorModel.P_OCCUPPIED), kernel, 2048);
    }

    public GridMap.GridMapData createMapData(GridMap.GridMapData other) {
        GridMap.GridMapData d = new GridMap.GridMapData();
        d.logData = new double[width * height];
        d.likelihoodData = new double[width * height];
        if (other != null) {
            System.arraycopy(other.logData, 0, d.logData, 0, d.logData.length);
            NativeSlam.mapUploadLog(handle, d.logData);
            NativeSlam.mapBuildLikelihood(handle);
        } else {
            NativeSlam.mapReset(handle);
        }
        hostStale = true;
        return d;
    }

    public void reset(GridMap.GridMapData map) { NativeSlam.mapReset(handle); hostStale = true; }

    public void integrateObservation(GridMap.GridMapData map, Observation obs, Pose p) {
        NativeSlam.mapIntegrate(handle, NativeSlam.flatten(obs), obs.getNumberOfMeasurements(), p.x, p.y, p.theta);
        hostStale = true;
    }

    public void applyMeasurement(GridMap.GridMapData map, float startX, float startY, float endX, float endY, float measuredDistance, boolean wasHit) {
        NativeSlam.mapApplyRay(handle, startX, startY, endX, endY, measuredDistance, wasHit);
        hostStale = true;
    }

    public void computeLikelihoodMap(GridMap.GridMapData map) { NativeSlam.mapBuildLikelihood(handle); hostStale = true; }

    /** Brings logData / likelihoodData up to date before render(), getRawAt(), the serialiser. */
    public void sync(GridMap.GridMapData map) {
        if (hostStale) { NativeSlam.mapDownload(handle, map.logData, map.likelihoodData); hostStale = false; }
    }

    public double getRawAt(GridMap.GridMapData map, int x, int y) { sync(map); return map.logData[x + y * width]; }
    public double getProbAt(GridMap.GridMapData map, int x, int y) { sync(map); return Util.invLogOdds(map.logData[x + y * width]); }

    long nativeHandle() { return handle; }
    public void dispose() { NativeSlam.mapDestroy(handle); }
}
