/*
 * gms_jni.c -- JNI shim between the reference's Java classes and libgridmapslam.so.
 *
 * NOT built in this image (no JDK, no jni.h); jni/Makefile builds it where JAVA_HOME exists.
 * It is the binding INTEGRATION.md describes: the Java facade under jni/java keeps the method
 * signatures of com.fmsz.gridmapgl.slam.GridMap / ParticleFilter (J/slam/GridMap.java:80-432,
 * J/slam/ParticleFilter.java:43-82) and forwards to these natives; each native is a thin call into
 * the C-ABI of include/gridmapslam.h.  The native handle travels in a Java long.
 *
 * Called from one thread only, like the path it replaces (J/app/DataEventHandler.java:24-26), so
 * GetPrimitiveArrayCritical is safe.  Math.log / Math.exp results (log-odds constants, blur taps) are
 * computed on the Java side and passed in, so the JVM's libm decides them.
 */
#include <jni.h>
#include <stdint.h>
#include <string.h>

#include "gridmapslam.h"

#define CLS(name) Java_com_fmsz_gridmapgl_slam_NativeSlam_##name

static void throw_gms(JNIEnv *env, int rc) {
    if (rc == GMS_OK) return;
    jclass ex = (*env)->FindClass(env, rc == GMS_ERR_INVALID ? "java/lang/IllegalArgumentException" : "java/lang/IllegalStateException");
    if (ex) (*env)->ThrowNew(env, ex, gms_last_error());
}

/* Observation -> gms_beam[]: the Java side flattens Measurement{localX, localY, distance, wasHit} into
 * double[4*B] (J/slam/Observation.java:37-41). */
static int beams_from(JNIEnv *env, jdoubleArray flat, jint B, gms_beam *out) {
    jdouble *p = (*env)->GetPrimitiveArrayCritical(env, flat, NULL);
    if (!p) return GMS_ERR_NOMEM;
    for (jint b = 0; b < B; b++) {
        memset(&out[b], 0, sizeof(gms_beam));
        out[b].local_x = p[4 * b]; out[b].local_y = p[4 * b + 1]; out[b].distance = p[4 * b + 2];
        out[b].hit = p[4 * b + 3] != 0.0;
    }
    (*env)->ReleasePrimitiveArrayCritical(env, flat, p, JNI_ABORT);
    return GMS_OK;
}

/* new GridMap(width, height, resolution, position) (GridMap.java:80) */
JNIEXPORT jlong JNICALL CLS(mapCreate)(JNIEnv *env, jclass c, jfloat w, jfloat h, jfloat res, jfloat px, jfloat py,
                                       jdouble lFree, jdouble lOcc, jdoubleArray kernel, jint maxBeams) {
    gms_params p;
    int rc = gms_params_default(&p, w, h, res, px, py);
    if (rc) { throw_gms(env, rc); return 0; }
    p.l_free = lFree; p.l_occ = lOcc; p.max_beams = maxBeams;
    jsize k = (*env)->GetArrayLength(env, kernel);
    if (k > GMS_MAX_TAPS) { throw_gms(env, GMS_ERR_INVALID); return 0; }
    (*env)->GetDoubleArrayRegion(env, kernel, 0, k, p.kernel);      /* Util.generateGaussianKernel on the JVM */
    p.ktaps = k;
    gms_map *m = NULL;
    rc = gms_map_create(&p, &m);
    throw_gms(env, rc);
    return (jlong)(intptr_t)m;
}
JNIEXPORT void JNICALL CLS(mapDestroy)(JNIEnv *env, jclass c, jlong m) { gms_map_destroy((gms_map *)(intptr_t)m); }
JNIEXPORT void JNICALL CLS(mapReset)(JNIEnv *env, jclass c, jlong m) { throw_gms(env, gms_map_reset((gms_map *)(intptr_t)m)); }

/* integrateObservation(map, obs, pose) (GridMap.java:173) */
JNIEXPORT void JNICALL CLS(mapIntegrate)(JNIEnv *env, jclass c, jlong m, jdoubleArray beams, jint B, jfloat x, jfloat y, jfloat theta) {
    gms_beam buf[2048];
    if (B > 2048) { throw_gms(env, GMS_ERR_INVALID); return; }
    int rc = beams_from(env, beams, B, buf);
    const float pose[3] = { x, y, theta };
    if (!rc) rc = gms_map_integrate((gms_map *)(intptr_t)m, buf, B, pose);
    throw_gms(env, rc);
}
/* applyMeasurement (GridMap.java:194) */
JNIEXPORT void JNICALL CLS(mapApplyRay)(JNIEnv *env, jclass c, jlong m, jfloat sx, jfloat sy, jfloat ex, jfloat ey, jfloat d, jboolean hit) {
    throw_gms(env, gms_map_apply_ray((gms_map *)(intptr_t)m, sx, sy, ex, ey, d, hit));
}
/* computeLikelihoodMap (GridMap.java:233) */
JNIEXPORT void JNICALL CLS(mapBuildLikelihood)(JNIEnv *env, jclass c, jlong m) { throw_gms(env, gms_map_build_likelihood((gms_map *)(intptr_t)m)); }

/* GridMapData.logData / likelihoodData back into the Java arrays (read by the renderer, GridMap.java:371-388) */
JNIEXPORT void JNICALL CLS(mapDownload)(JNIEnv *env, jclass c, jlong m, jdoubleArray logData, jdoubleArray likData) {
    if (logData) {
        jdouble *p = (*env)->GetPrimitiveArrayCritical(env, logData, NULL);
        int rc = gms_map_download_log((gms_map *)(intptr_t)m, p);
        (*env)->ReleasePrimitiveArrayCritical(env, logData, p, 0);
        throw_gms(env, rc);
    }
    if (likData) {
        jdouble *p = (*env)->GetPrimitiveArrayCritical(env, likData, NULL);
        int rc = gms_map_download_likelihood((gms_map *)(intptr_t)m, p);
        (*env)->ReleasePrimitiveArrayCritical(env, likData, p, 0);
        throw_gms(env, rc);
    }
}
JNIEXPORT void JNICALL CLS(mapUploadLog)(JNIEnv *env, jclass c, jlong m, jdoubleArray logData) {
    jdouble *p = (*env)->GetPrimitiveArrayCritical(env, logData, NULL);
    int rc = gms_map_upload_log((gms_map *)(intptr_t)m, p);
    (*env)->ReleasePrimitiveArrayCritical(env, logData, p, JNI_ABORT);
    throw_gms(env, rc);
}

/* new ParticleFilter(n) (ParticleFilter.java:43) */
JNIEXPORT jlong JNICALL CLS(pfCreate)(JNIEnv *env, jclass c, jlong m, jint n) {
    gms_pf *pf = NULL;
    throw_gms(env, gms_pf_create((gms_map *)(intptr_t)m, n, &pf));
    return (jlong)(intptr_t)pf;
}
JNIEXPORT void JNICALL CLS(pfDestroy)(JNIEnv *env, jclass c, jlong pf) { gms_pf_destroy((gms_pf *)(intptr_t)pf); }
JNIEXPORT void JNICALL CLS(pfSetPoses)(JNIEnv *env, jclass c, jlong pf, jfloatArray xyt) {
    jfloat *p = (*env)->GetPrimitiveArrayCritical(env, xyt, NULL);
    int rc = gms_pf_set_poses((gms_pf *)(intptr_t)pf, p);
    (*env)->ReleasePrimitiveArrayCritical(env, xyt, p, JNI_ABORT);
    throw_gms(env, rc);
}
JNIEXPORT void JNICALL CLS(pfGetParticles)(JNIEnv *env, jclass c, jlong pf, jfloatArray xyt, jdoubleArray w) {
    jfloat *p = (*env)->GetPrimitiveArrayCritical(env, xyt, NULL);
    int rc = gms_pf_get_poses((gms_pf *)(intptr_t)pf, p);
    (*env)->ReleasePrimitiveArrayCritical(env, xyt, p, 0);
    if (!rc) {
        jdouble *q = (*env)->GetPrimitiveArrayCritical(env, w, NULL);
        rc = gms_pf_get_weights((gms_pf *)(intptr_t)pf, q);
        (*env)->ReleasePrimitiveArrayCritical(env, w, q, 0);
    }
    throw_gms(env, rc);
}
/* weight[i] = probabilityOf(map, obs, pose[i]) (GridMap.java:261, SLAM.java:99) */
JNIEXPORT void JNICALL CLS(pfScore)(JNIEnv *env, jclass c, jlong pf, jdoubleArray beams, jint B) {
    gms_beam buf[2048];
    if (B > 2048) { throw_gms(env, GMS_ERR_INVALID); return; }
    int rc = beams_from(env, beams, B, buf);
    if (!rc) rc = gms_pf_score((gms_pf *)(intptr_t)pf, buf, B);
    throw_gms(env, rc);
}
/* SLAM.update bookkeeping (SLAM.java:87-129): out = {weightSum, neff, strongest} */
JNIEXPORT void JNICALL CLS(pfNormalize)(JNIEnv *env, jclass c, jlong pf, jdoubleArray out3) {
    gms_pf_stats st;
    int rc = gms_pf_normalize((gms_pf *)(intptr_t)pf, &st);
    if (!rc) {
        const jdouble v[3] = { st.weight_sum, st.neff, (jdouble)st.strongest };
        (*env)->SetDoubleArrayRegion(env, out3, 0, 3, v);
    }
    throw_gms(env, rc);
}
/* resample() with r = Math.random() drawn on the Java side (SLAM.java:136) */
JNIEXPORT void JNICALL CLS(pfResample)(JNIEnv *env, jclass c, jlong pf, jdouble r01) {
    throw_gms(env, gms_pf_resample((gms_pf *)(intptr_t)pf, &r01, NULL, NULL));
}
/* getWeightedPose() (SLAM.java:165) */
JNIEXPORT void JNICALL CLS(pfWeightedPose)(JNIEnv *env, jclass c, jlong pf, jfloatArray out3) {
    float o[3];
    int rc = gms_pf_weighted_pose((gms_pf *)(intptr_t)pf, o);
    if (!rc) (*env)->SetFloatArrayRegion(env, out3, 0, 3, o);
    throw_gms(env, rc);
}
/* integrateObservation at the filter's weighted pose + likelihood rebuild, no host round trip */
JNIEXPORT void JNICALL CLS(mapUpdateAt)(JNIEnv *env, jclass c, jlong m, jdoubleArray beams, jint B, jlong pf) {
    gms_beam buf[2048];
    if (B > 2048) { throw_gms(env, GMS_ERR_INVALID); return; }
    int rc = beams_from(env, beams, B, buf);
    if (!rc) rc = gms_map_update_at((gms_map *)(intptr_t)m, buf, B, (gms_pf *)(intptr_t)pf, 0);
    throw_gms(env, rc);
}

/* SLAM.update(z, u) (SLAM.java:80-131) + `if (neff < fraction * N) resample()` (GridMapApp.java:185-186) in ONE call:
 * poses (may be null) are the motion-model samples drawn on the JVM; out3 = {weightSum, neff, strongest}.
 * resampleFraction < 0 skips the resampling; integrate = false is the skipUpdate case (SLAM.java:82). */
JNIEXPORT void JNICALL CLS(slamUpdate)(JNIEnv *env, jclass c, jlong pf, jfloatArray xyt, jdoubleArray beams, jint B, jdouble r01,
                                       jdouble resampleFraction, jboolean integrate, jdoubleArray out3) {
    gms_beam buf[2048];
    if (B > 2048) { throw_gms(env, GMS_ERR_INVALID); return; }
    int rc = beams_from(env, beams, B, buf);
    if (rc) { throw_gms(env, rc); return; }
    gms_pf_stats st;
    jfloat *p = xyt ? (*env)->GetPrimitiveArrayCritical(env, xyt, NULL) : NULL;
    rc = gms_slam_update((gms_pf *)(intptr_t)pf, p, buf, B, &r01, resampleFraction, integrate ? 1 : 0, &st);
    if (p) (*env)->ReleasePrimitiveArrayCritical(env, xyt, p, JNI_ABORT);
    if (!rc && out3) {
        const jdouble v[3] = { st.weight_sum, st.neff, (jdouble)st.strongest };
        (*env)->SetDoubleArrayRegion(env, out3, 0, 3, v);
    }
    throw_gms(env, rc);
}
