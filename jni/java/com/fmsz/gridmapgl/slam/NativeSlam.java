package com.fmsz.gridmapgl.slam;

/** Native entry points of libgms_jni.so (jni/gms_jni.c) -> libgridmapslam.so (include/gridmapslam.h). */
final class NativeSlam {
    static { System.loadLibrary("gms_jni"); }
    private NativeSlam() {}

    // GridMap
    static native long mapCreate(float w, float h, float res, float px, float py, double lFree, double lOcc, double[] kernel, int maxBeams, int device);
    static native void mapDestroy(long m);
    static native void mapReset(long m);
    static native void mapIntegrate(long m, double[] beams, int B, float x, float y, float theta);
    static native void mapApplyRay(long m, float sx, float sy, float ex, float ey, float measured, boolean hit);
    static native void mapBuildLikelihood(long m);
    static native void mapDownload(long m, double[] logDataOrNull, double[] likelihoodDataOrNull);
    static native void mapUpload(long m, double[] logDataOrNull, double[] likelihoodDataOrNull);
    static native void mapGetAtPoint(long m, float px, float py, double[] rawAndLikelihood);
    static native void mapUpdateAt(long m, double[] beams, int B, long pf);
    // ParticleFilter / SLAM
    static native long pfCreate(long m, int n);
    static native void pfDestroy(long pf);
    static native void pfSetShard(long pf, long offset, long nGlobal);
    static native void pfSetRefine(long pf, boolean on);
    static native void pfSetLogNormalize(long pf, boolean on);
    static native void pfSetPoses(long pf, float[] xytheta, int n);
    static native void pfGetParticles(long pf, float[] xytheta, double[] weights, int n);
    static native void pfScore(long pf, double[] beams, int B);
    static native double pfProbabilityOf(long pf1, double[] beams, int B, float x, float y, float theta);
    static native void pfFindBestPose(long pf1, double[] beams, int B, float x, float y, float theta, float[] out3);
    static native void pfNormalize(long pf, double[] weightSumNeffStrongest);
    static native void pfResample(long pf, double r01);
    static native void pfWeightedPose(long pf, float[] out3);
    /** SLAM.update + conditional resample in one native call (four kernel launches on the device). */
    static native void slamUpdate(long pf, float[] xythetaOrNull, int n, double[] beams, int B, double r01, double resampleFraction,
                                  boolean integrate, double[] weightSumNeffStrongest);
    // multi-GPU: one JVM per GPU; the 128-byte id is made by one rank and handed to the others by the host
    static native void commUniqueId(byte[] id128);
    static native long commCreate(byte[] id128, int rank, int world, int device);
    static native void commDestroy(long comm);
    static native void slamUpdateSharded(long pf, long comm, float[] xythetaOrNull, int n, double[] beams, int B, double r01,
                                         double resampleFraction, boolean integrate, double[] weightSumNeffStrongest);

    // SLAM with one GridMapData per particle (SLAM.java:26-204): SLAMGpu
    static native long pmCreate(float w, float h, float res, float px, float py, double lFree, double lOcc, double[] kernel, int maxBeams, int device, int numParticles);
    // ... sharded with its maps over the GPUs of a node, one JVM per GPU (gms_slam_create_shard; the exchanges inside the library)
    static native long pmCreateShard(float w, float h, float res, float px, float py, double lFree, double lOcc, double[] kernel, int maxBeams, int device,
                                     int nLocal, long offset, long nGlobal);
    static native void pmUpdateSharded(long s, long comm, double[] beams, int B, boolean sampleMotion, double dCenter, double dTheta, long seed, long sequence,
                                       double[] weightSumNeffStrongest);
    static native boolean pmResampleSharded(long s, long comm, double r01, double fraction);
    static native void pmDestroy(long s);
    static native void pmReset(long s);
    static native void pmSetRefine(long s, boolean on);                         // SLAM.java:96: findBestPose of every particle against its own field
    static native void pmUpdate(long s, double[] beams, int B, boolean sampleMotion, double dCenter, double dTheta, long seed, long sequence, double[] weightSumNeffStrongest);
    static native void pmResample(long s, double r01);
    static native void pmResampleIf(long s, double r01, double fraction);      // the rule of GridMapApp.java:185-186, decided on the device
    static native void pmGetParticles(long s, float[] xytheta, double[] weights, int n);
    static native void pmWeightedPose(long s, float[] out3);
    static native void pmDownloadMap(long s, int i, double[] logDataOrNull, double[] likelihoodDataOrNull);
    static native void pmCombined(long s, double[] logDataOrNull, double[] likelihoodDataOrNull);

    /** Observation -> double[4*B] {localX, localY, distance, wasHit} (Observation.java:37-41). */
    static double[] flatten(Observation obs) {
        java.util.List<Observation.Measurement> ms = obs.getMeasurements();
        double[] out = new double[4 * ms.size()];
        int i = 0;
        for (Observation.Measurement m : ms) {
            out[i++] = m.localX; out[i++] = m.localY; out[i++] = m.distance; out[i++] = m.wasHit ? 1.0 : 0.0;
        }
        return out;
    }
}
