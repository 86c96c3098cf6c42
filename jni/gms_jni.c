/*
 * gms_jni.c -- JNI shim between the reference's Java classes and libgridmapslam.so.
 *
 * NOT built in this image (no JDK, no jni.h): jni/Makefile builds it where JAVA_HOME exists; tests/test_jni_syntax.py
 * runs `gcc -fsyntax-only` on it against a minimal tests-local jni.h (a SYNTAX check only, nothing is linked or run).
 * It is the binding INTEGRATION.md describes: the Java facade under jni/java (GridMapGpu extends GridMap,
 * ParticleFilterGpu) keeps the method signatures of com.fmsz.gridmapgl.slam.GridMap / ParticleFilter
 * (J/slam/GridMap.java:80-432, J/slam/ParticleFilter.java:43-82) and forwards to these natives; each native is a thin
 * call into the C-ABI of include/gridmapslam.h.  Native handles travel in Java longs.
 *
 * Rules kept here:
 *  - a JNI critical region (GetPrimitiveArrayCritical) only ever brackets a memcpy-like loop: no call that may block
 *    or synchronise a stream runs inside one (inputs are copied to heap buffers first, outputs come back through
 *    heap buffers);
 *  - beam buffers live on the heap (B * 32 bytes), not on the native stack;
 *  - Math.log / Math.exp results (log-odds constants, blur taps) are computed on the Java side and passed in, so the
 *    JVM's libm decides them.
 * Called from one thread at a time, like the path it replaces (J/app/DataEventHandler.java:24-26).
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "gridmapslam.h"

#define CLS(name) Java_com_fmsz_gridmapgl_slam_NativeSlam_##name
#define MAP(h) ((gms_map *)(intptr_t)(h))
#define PF(h) ((gms_pf *)(intptr_t)(h))
#define COMM(h) ((gms_comm *)(intptr_t)(h))
#define SLAM(h) ((gms_slam *)(intptr_t)(h))

static void throw_named(JNIEnv *env, const char *cls, const char *msg) {
    jclass ex = (*env)->FindClass(env, cls);
    if (ex) (*env)->ThrowNew(env, ex, msg);
}
static void throw_gms(JNIEnv *env, int rc) {
    if (rc == GMS_OK) return;
    throw_named(env, rc == GMS_ERR_INVALID ? "java/lang/IllegalArgumentException" : "java/lang/IllegalStateException", gms_last_error());
}

/* Observation -> gms_beam[B] on the heap: the Java side flattens Measurement{localX, localY, distance, wasHit} into
 * double[4*B] (J/slam/Observation.java:37-41).  NULL (with a pending exception) on failure. */
static gms_beam *beams_from(JNIEnv *env, jdoubleArray flat, jint B) {
    if (B < 0 || !flat || (*env)->GetArrayLength(env, flat) < 4 * B) { throw_named(env, "java/lang/IllegalArgumentException", "beam array shorter than 4*B"); return NULL; }
    gms_beam *out = (gms_beam *)calloc((size_t)(B > 0 ? B : 1), sizeof(gms_beam));
    if (!out) { throw_named(env, "java/lang/OutOfMemoryError", "beam buffer"); return NULL; }
    jdouble *p = (jdouble *)(*env)->GetPrimitiveArrayCritical(env, flat, NULL);
    if (!p) { free(out); return NULL; }                             /* OutOfMemoryError is pending */
    for (jint b = 0; b < B; b++) {
        out[b].local_x = p[4 * b]; out[b].local_y = p[4 * b + 1]; out[b].distance = p[4 * b + 2];
        out[b].hit = p[4 * b + 3] != 0.0;
    }
    (*env)->ReleasePrimitiveArrayCritical(env, flat, p, JNI_ABORT);
    return out;
}
/* The C-ABI reads and writes pf->n * n_maps elements whatever the caller's `n` says: a mismatch is refused here
 * (IllegalArgumentException) instead of overrunning the heap buffers sized from `n`.  want_single: the handle must
 * hold exactly one particle on one map (the one-pose entry points). */
static int particle_count_ok(JNIEnv *env, jlong pf, jint n, int want_single) {
    int32_t have = 0, maps = 0;
    if (gms_pf_count(PF(pf), &have, &maps, NULL) != GMS_OK) { throw_gms(env, GMS_ERR_INVALID); return 0; }
    if (want_single ? (have != 1 || maps != 1) : ((int64_t)n != (int64_t)have * maps)) {
        throw_named(env, "java/lang/IllegalArgumentException",
                    want_single ? "this entry point needs a one-particle filter on a single map" : "particle count differs from the native filter's");
        return 0;
    }
    return 1;
}
/* float[] -> heap copy (NULL array -> NULL, *ok stays 1) */
static float *floats_from(JNIEnv *env, jfloatArray a, size_t n, int *ok) {
    *ok = 1;
    if (!a) return NULL;
    if ((size_t)(*env)->GetArrayLength(env, a) < n) { *ok = 0; throw_named(env, "java/lang/IllegalArgumentException", "pose array too short"); return NULL; }
    float *out = (float *)malloc(n * sizeof(float) + 4);
    if (!out) { *ok = 0; throw_named(env, "java/lang/OutOfMemoryError", "pose buffer"); return NULL; }
    (*env)->GetFloatArrayRegion(env, a, 0, (jsize)n, out);
    return out;
}
static void stats_out(JNIEnv *env, jdoubleArray out3, const gms_pf_stats *st) {
    if (!out3) return;
    const jdouble v[3] = { st->weight_sum, st->neff, (jdouble)st->strongest };
    (*env)->SetDoubleArrayRegion(env, out3, 0, 3, v);
}

/* ---- GridMap ------------------------------------------------------------------------------------------------- */
/* new GridMap(width, height, resolution, position) (GridMap.java:80) */
JNIEXPORT jlong JNICALL CLS(mapCreate)(JNIEnv *env, jclass c, jfloat w, jfloat h, jfloat res, jfloat px, jfloat py,
                                       jdouble lFree, jdouble lOcc, jdoubleArray kernel, jint maxBeams, jint device) {
    gms_params p;
    int rc = gms_params_default(&p, w, h, res, px, py);
    if (rc) { throw_gms(env, rc); return 0; }
    p.l_free = lFree; p.l_occ = lOcc; p.max_beams = maxBeams; p.device = device;
    jsize k = (*env)->GetArrayLength(env, kernel);
    if (k > GMS_MAX_TAPS) { throw_named(env, "java/lang/IllegalArgumentException", "likelihood kernel longer than GMS_MAX_TAPS"); return 0; }
    (*env)->GetDoubleArrayRegion(env, kernel, 0, k, p.kernel);      /* Util.generateGaussianKernel on the JVM */
    p.ktaps = k;
    gms_map *m = NULL;
    rc = gms_map_create(&p, &m);
    throw_gms(env, rc);
    return (jlong)(intptr_t)m;
}
JNIEXPORT void JNICALL CLS(mapDestroy)(JNIEnv *env, jclass c, jlong m) { throw_gms(env, gms_map_destroy(MAP(m))); }
JNIEXPORT void JNICALL CLS(mapReset)(JNIEnv *env, jclass c, jlong m) { throw_gms(env, gms_map_reset(MAP(m))); }

/* integrateObservation(map, obs, pose) (GridMap.java:173) */
JNIEXPORT void JNICALL CLS(mapIntegrate)(JNIEnv *env, jclass c, jlong m, jdoubleArray beams, jint B, jfloat x, jfloat y, jfloat theta) {
    gms_beam *buf = beams_from(env, beams, B);
    if (!buf) return;
    const float pose[3] = { x, y, theta };
    const int rc = gms_map_integrate(MAP(m), buf, B, pose);
    free(buf);
    throw_gms(env, rc);
}
/* applyMeasurement (GridMap.java:194) */
JNIEXPORT void JNICALL CLS(mapApplyRay)(JNIEnv *env, jclass c, jlong m, jfloat sx, jfloat sy, jfloat ex, jfloat ey, jfloat d, jboolean hit) {
    throw_gms(env, gms_map_apply_ray(MAP(m), sx, sy, ex, ey, d, hit));
}
/* computeLikelihoodMap (GridMap.java:233) */
JNIEXPORT void JNICALL CLS(mapBuildLikelihood)(JNIEnv *env, jclass c, jlong m) { throw_gms(env, gms_map_build_likelihood(MAP(m))); }

/* GridMapData.logData / likelihoodData back into the Java arrays (read by the renderer, GridMap.java:371-388).  The
 * download synchronises the stream, so it goes through a heap buffer; the critical region covers the memcpy only. */
static int download_into(JNIEnv *env, jlong m, jdoubleArray dst, int lik) {
    int32_t W, H, M;
    int rc = gms_map_get_size(MAP(m), &W, &H, &M);
    if (rc) return rc;
    const size_t n = (size_t)W * H * M;
    if ((size_t)(*env)->GetArrayLength(env, dst) < n) { throw_named(env, "java/lang/IllegalArgumentException", "map array too short"); return GMS_OK; }
    double *tmp = (double *)malloc(n * sizeof(double));
    if (!tmp) { throw_named(env, "java/lang/OutOfMemoryError", "map buffer"); return GMS_OK; }
    rc = lik ? gms_map_download_likelihood(MAP(m), tmp) : gms_map_download_log(MAP(m), tmp);
    if (!rc) {
        jdouble *p = (jdouble *)(*env)->GetPrimitiveArrayCritical(env, dst, NULL);
        if (p) { memcpy(p, tmp, n * sizeof(double)); (*env)->ReleasePrimitiveArrayCritical(env, dst, p, 0); }
    }
    free(tmp);
    return rc;
}
JNIEXPORT void JNICALL CLS(mapDownload)(JNIEnv *env, jclass c, jlong m, jdoubleArray logData, jdoubleArray likData) {
    int rc = GMS_OK;
    if (logData) rc = download_into(env, m, logData, 0);
    if (!rc && likData && !(*env)->ExceptionCheck(env)) rc = download_into(env, m, likData, 1);
    throw_gms(env, rc);
}
JNIEXPORT void JNICALL CLS(mapUpload)(JNIEnv *env, jclass c, jlong m, jdoubleArray logData, jdoubleArray likData) {
    int32_t W, H, M;
    int rc = gms_map_get_size(MAP(m), &W, &H, &M);
    const size_t n = (size_t)W * H * M;
    double *tmp = rc ? NULL : (double *)malloc(n * sizeof(double));
    if (!rc && !tmp) { throw_named(env, "java/lang/OutOfMemoryError", "map buffer"); return; }
    if (!rc && logData) {
        (*env)->GetDoubleArrayRegion(env, logData, 0, (jsize)n, tmp);
        if (!(*env)->ExceptionCheck(env)) rc = gms_map_upload_log(MAP(m), tmp);
    }
    if (!rc && likData && !(*env)->ExceptionCheck(env)) {
        (*env)->GetDoubleArrayRegion(env, likData, 0, (jsize)n, tmp);
        if (!(*env)->ExceptionCheck(env)) rc = gms_map_upload_likelihood(MAP(m), tmp);
    }
    free(tmp);
    throw_gms(env, rc);
}
/* getRawAt(map, Vec2) / getLikelihood(map, Vec2) (GridMap.java:142-156): out2 = {raw, likelihood}.  The Java code
 * throws ArrayIndexOutOfBoundsException for a point whose flat index leaves the array; so does this. */
JNIEXPORT void JNICALL CLS(mapGetAtPoint)(JNIEnv *env, jclass c, jlong m, jfloat px, jfloat py, jdoubleArray out2) {
    double v[2];
    const int rc = gms_map_get_at_point(MAP(m), 0, px, py, &v[0], &v[1]);
    if (rc == GMS_ERR_INVALID) { throw_named(env, "java/lang/ArrayIndexOutOfBoundsException", gms_last_error()); return; }
    if (!rc) (*env)->SetDoubleArrayRegion(env, out2, 0, 2, v);
    throw_gms(env, rc);
}
/* integrateObservation at the filter's weighted pose + likelihood rebuild, no host round trip */
JNIEXPORT void JNICALL CLS(mapUpdateAt)(JNIEnv *env, jclass c, jlong m, jdoubleArray beams, jint B, jlong pf) {
    gms_beam *buf = beams_from(env, beams, B);
    if (!buf) return;
    const int rc = gms_map_update_at(MAP(m), buf, B, PF(pf), 0);
    free(buf);
    throw_gms(env, rc);
}

/* ---- ParticleFilter / SLAM ------------------------------------------------------------------------------------ */
/* new ParticleFilter(n) (ParticleFilter.java:43) */
JNIEXPORT jlong JNICALL CLS(pfCreate)(JNIEnv *env, jclass c, jlong m, jint n) {
    gms_pf *pf = NULL;
    throw_gms(env, gms_pf_create(MAP(m), n, &pf));
    return (jlong)(intptr_t)pf;
}
JNIEXPORT void JNICALL CLS(pfDestroy)(JNIEnv *env, jclass c, jlong pf) { throw_gms(env, gms_pf_destroy(PF(pf))); }
JNIEXPORT void JNICALL CLS(pfSetShard)(JNIEnv *env, jclass c, jlong pf, jlong offset, jlong nGlobal) { throw_gms(env, gms_pf_set_shard(PF(pf), offset, nGlobal)); }
JNIEXPORT void JNICALL CLS(pfSetRefine)(JNIEnv *env, jclass c, jlong pf, jboolean on) { throw_gms(env, gms_pf_set_refine(PF(pf), on ? 1 : 0)); }
JNIEXPORT void JNICALL CLS(pfSetLogNormalize)(JNIEnv *env, jclass c, jlong pf, jboolean on) { throw_gms(env, gms_pf_set_log_normalize(PF(pf), on ? 1 : 0)); }
JNIEXPORT void JNICALL CLS(pfSetPoses)(JNIEnv *env, jclass c, jlong pf, jfloatArray xyt, jint n) {
    int ok;
    if (!particle_count_ok(env, pf, n, 0)) return;
    float *p = floats_from(env, xyt, (size_t)n * 3, &ok);
    if (!p) return;
    const int rc = gms_pf_set_poses(PF(pf), p);                     /* copies into a pinned ring before returning */
    free(p);
    throw_gms(env, rc);
}
JNIEXPORT void JNICALL CLS(pfGetParticles)(JNIEnv *env, jclass c, jlong pf, jfloatArray xyt, jdoubleArray w, jint n) {
    if (!particle_count_ok(env, pf, n, 0)) return;
    float *p = (float *)malloc((size_t)n * 3 * sizeof(float));
    double *q = (double *)malloc((size_t)n * sizeof(double));
    int rc = (p && q) ? gms_pf_get_poses(PF(pf), p) : GMS_ERR_NOMEM;
    if (!rc) rc = gms_pf_get_weights(PF(pf), q);
    if (!rc) {
        (*env)->SetFloatArrayRegion(env, xyt, 0, 3 * n, p);
        (*env)->SetDoubleArrayRegion(env, w, 0, n, q);
    }
    free(p); free(q);
    throw_gms(env, rc);
}
/* weight[i] = probabilityOf(map, obs, pose[i]) (GridMap.java:261, SLAM.java:99) */
JNIEXPORT void JNICALL CLS(pfScore)(JNIEnv *env, jclass c, jlong pf, jdoubleArray beams, jint B) {
    gms_beam *buf = beams_from(env, beams, B);
    if (!buf) return;
    const int rc = gms_pf_score(PF(pf), buf, B);
    free(buf);
    throw_gms(env, rc);
}
/* probabilityOf(map, obs, pose) for ONE pose (GridMap.java:261): pf is a one-particle filter on the map */
JNIEXPORT jdouble JNICALL CLS(pfProbabilityOf)(JNIEnv *env, jclass c, jlong pf, jdoubleArray beams, jint B, jfloat x, jfloat y, jfloat theta) {
    if (!particle_count_ok(env, pf, 1, 1)) return 0.0;           /* gms_pf_get_weights writes one double per particle */
    gms_beam *buf = beams_from(env, beams, B);
    if (!buf) return 0.0;
    const float pose[3] = { x, y, theta };
    double w = 0.0;
    int rc = gms_pf_set_poses(PF(pf), pose);
    if (!rc) rc = gms_pf_score(PF(pf), buf, B);
    if (!rc) rc = gms_pf_get_weights(PF(pf), &w);
    free(buf);
    throw_gms(env, rc);
    return w;
}
/* findBestPose(map, obs, startPose) (GridMap.java:319): pf is a one-particle filter on the map; out3 = best pose */
JNIEXPORT void JNICALL CLS(pfFindBestPose)(JNIEnv *env, jclass c, jlong pf, jdoubleArray beams, jint B, jfloat x, jfloat y, jfloat theta,
                                           jfloatArray out3) {
    if (!particle_count_ok(env, pf, 1, 1)) return;
    gms_beam *buf = beams_from(env, beams, B);
    if (!buf) return;
    float pose[3] = { x, y, theta };
    int rc = gms_pf_set_poses(PF(pf), pose);
    if (!rc) rc = gms_pf_refine_poses(PF(pf), buf, B);
    if (!rc) rc = gms_pf_get_poses(PF(pf), pose);
    free(buf);
    if (!rc) (*env)->SetFloatArrayRegion(env, out3, 0, 3, pose);
    throw_gms(env, rc);
}
/* SLAM.update bookkeeping (SLAM.java:87-129): out = {weightSum, neff, strongest} */
JNIEXPORT void JNICALL CLS(pfNormalize)(JNIEnv *env, jclass c, jlong pf, jdoubleArray out3) {
    gms_pf_stats st;
    const int rc = gms_pf_normalize(PF(pf), &st);
    if (!rc) stats_out(env, out3, &st);
    throw_gms(env, rc);
}
/* resample() with r = Math.random() drawn on the Java side (SLAM.java:136) */
JNIEXPORT void JNICALL CLS(pfResample)(JNIEnv *env, jclass c, jlong pf, jdouble r01) {
    throw_gms(env, gms_pf_resample(PF(pf), &r01, NULL, NULL));
}
/* getWeightedPose() (SLAM.java:165) */
JNIEXPORT void JNICALL CLS(pfWeightedPose)(JNIEnv *env, jclass c, jlong pf, jfloatArray out3) {
    float o[3];
    const int rc = gms_pf_weighted_pose(PF(pf), o);
    if (!rc) (*env)->SetFloatArrayRegion(env, out3, 0, 3, o);
    throw_gms(env, rc);
}

/* SLAM.update(z, u) (SLAM.java:80-131) + `if (neff < fraction * N) resample()` (GridMapApp.java:185-186) in ONE call:
 * poses (may be null) are the motion-model samples drawn on the JVM; out3 = {weightSum, neff, strongest}.
 * resampleFraction < 0 skips the resampling; integrate = false is the skipUpdate case (SLAM.java:82).
 * The inputs are copied to the heap first: gms_slam_update synchronises the stream when it returns the statistics,
 * which must not happen inside a JNI critical region. */
JNIEXPORT void JNICALL CLS(slamUpdate)(JNIEnv *env, jclass c, jlong pf, jfloatArray xyt, jint n, jdoubleArray beams, jint B, jdouble r01,
                                       jdouble resampleFraction, jboolean integrate, jdoubleArray out3) {
    if (xyt && !particle_count_ok(env, pf, n, 0)) return;
    gms_beam *buf = beams_from(env, beams, B);
    if (!buf) return;
    int ok;
    float *p = floats_from(env, xyt, (size_t)n * 3, &ok);
    if (!ok) { free(buf); return; }
    gms_pf_stats st;
    const int rc = gms_slam_update(PF(pf), p, buf, B, &r01, resampleFraction, integrate ? 1 : 0, &st);
    free(p); free(buf);
    if (!rc) stats_out(env, out3, &st);
    throw_gms(env, rc);
}

/* ---- multi-GPU: one JVM per GPU, the exchange inside the library -------------------- */
JNIEXPORT void JNICALL CLS(commUniqueId)(JNIEnv *env, jclass c, jbyteArray id128) {
    jbyte id[128];
    const int rc = gms_comm_unique_id(id);
    if (!rc) (*env)->SetByteArrayRegion(env, id128, 0, 128, id);
    throw_gms(env, rc);
}
/* blocks until all `world` ranks have called it (no JNI critical region is held) */
JNIEXPORT jlong JNICALL CLS(commCreate)(JNIEnv *env, jclass c, jbyteArray id128, jint rank, jint world, jint device) {
    jbyte id[128];
    (*env)->GetByteArrayRegion(env, id128, 0, 128, id);
    if ((*env)->ExceptionCheck(env)) return 0;
    gms_comm *cm = NULL;
    throw_gms(env, gms_comm_create(&cm, id, rank, world, device));
    return (jlong)(intptr_t)cm;
}
JNIEXPORT void JNICALL CLS(commDestroy)(JNIEnv *env, jclass c, jlong cm) { throw_gms(env, gms_comm_destroy(COMM(cm))); }
/* slamUpdate for this rank's shard of a sharded filter: every rank passes the same scan and r01 */
JNIEXPORT void JNICALL CLS(slamUpdateSharded)(JNIEnv *env, jclass c, jlong pf, jlong cm, jfloatArray xyt, jint n, jdoubleArray beams, jint B,
                                              jdouble r01, jdouble resampleFraction, jboolean integrate, jdoubleArray out3) {
    if (xyt && !particle_count_ok(env, pf, n, 0)) return;
    gms_beam *buf = beams_from(env, beams, B);
    if (!buf) return;
    int ok;
    float *p = floats_from(env, xyt, (size_t)n * 3, &ok);
    if (!ok) { free(buf); return; }
    gms_pf_stats st;
    const int rc = gms_slam_update_sharded(PF(pf), COMM(cm), p, buf, B, &r01, resampleFraction, integrate ? 1 : 0, &st);
    free(p); free(buf);
    if (!rc) stats_out(env, out3, &st);
    throw_gms(env, rc);
}

/* ---- SLAM as the reference has it: one GridMapData per particle (J/slam/SLAM.java:26-204; gms_slam_* of gridmapslam.h) -------- */
/* new SLAM() (SLAM.java:56-62): the GridMap's constants come from the JVM as for mapCreate */
JNIEXPORT jlong JNICALL CLS(pmCreate)(JNIEnv *env, jclass c, jfloat w, jfloat h, jfloat res, jfloat px, jfloat py, jdouble lFree, jdouble lOcc,
                                      jdoubleArray kernel, jint maxBeams, jint device, jint numParticles) {
    gms_params p;
    int rc = gms_params_default(&p, w, h, res, px, py);
    if (rc) { throw_gms(env, rc); return 0; }
    p.l_free = lFree; p.l_occ = lOcc; p.max_beams = maxBeams; p.device = device;
    jsize k = (*env)->GetArrayLength(env, kernel);
    if (k > GMS_MAX_TAPS) { throw_named(env, "java/lang/IllegalArgumentException", "likelihood kernel longer than GMS_MAX_TAPS"); return 0; }
    (*env)->GetDoubleArrayRegion(env, kernel, 0, k, p.kernel);
    p.ktaps = k;
    gms_slam *s = NULL;
    rc = gms_slam_create(&p, numParticles, &s);
    throw_gms(env, rc);
    return (jlong)(intptr_t)s;
}
/* one rank's block of a filter whose particles AND maps are split over the ranks (one JVM per GPU): gms_slam_create_shard */
JNIEXPORT jlong JNICALL CLS(pmCreateShard)(JNIEnv *env, jclass c, jfloat w, jfloat h, jfloat res, jfloat px, jfloat py, jdouble lFree, jdouble lOcc,
                                           jdoubleArray kernel, jint maxBeams, jint device, jint nLocal, jlong offset, jlong nGlobal) {
    gms_params p;
    int rc = gms_params_default(&p, w, h, res, px, py);
    if (rc) { throw_gms(env, rc); return 0; }
    p.l_free = lFree; p.l_occ = lOcc; p.max_beams = maxBeams; p.device = device;
    jsize k = (*env)->GetArrayLength(env, kernel);
    if (k > GMS_MAX_TAPS) { throw_named(env, "java/lang/IllegalArgumentException", "likelihood kernel longer than GMS_MAX_TAPS"); return 0; }
    (*env)->GetDoubleArrayRegion(env, kernel, 0, k, p.kernel);
    p.ktaps = k;
    gms_slam *s = NULL;
    rc = gms_slam_create_shard(&p, nLocal, offset, nGlobal, &s);
    throw_gms(env, rc);
    return (jlong)(intptr_t)s;
}
/* SLAM.update(z, u) over all ranks (gms_slam_update_sharded_maps): every rank passes the same scan, odometry, seed and sequence */
JNIEXPORT void JNICALL CLS(pmUpdateSharded)(JNIEnv *env, jclass c, jlong s, jlong cm, jdoubleArray beams, jint B, jboolean sampleMotion, jdouble dCenter,
                                            jdouble dTheta, jlong seed, jlong sequence, jdoubleArray out3) {
    gms_beam *buf = beams_from(env, beams, B);
    if (!buf) return;
    gms_pf_stats st;
    const int rc = gms_slam_update_sharded_maps(SLAM(s), COMM(cm), buf, B, sampleMotion ? 1 : 0, dCenter, dTheta, (uint64_t)seed, (uint64_t)sequence, &st);
    free(buf);
    if (!rc) stats_out(env, out3, &st);
    throw_gms(env, rc);
}
/* SLAM.resample() over all ranks (gms_slam_resample_sharded_maps; fraction < 0: unconditional): every rank the same r01; returns whether it drew */
JNIEXPORT jboolean JNICALL CLS(pmResampleSharded)(JNIEnv *env, jclass c, jlong s, jlong cm, jdouble r01, jdouble fraction) {
    int32_t did = 0;
    throw_gms(env, gms_slam_resample_sharded_maps(SLAM(s), COMM(cm), r01, fraction, &did));
    return (jboolean)(did != 0);
}
JNIEXPORT void JNICALL CLS(pmDestroy)(JNIEnv *env, jclass c, jlong s) { throw_gms(env, gms_slam_destroy(SLAM(s))); }
JNIEXPORT void JNICALL CLS(pmReset)(JNIEnv *env, jclass c, jlong s) { throw_gms(env, gms_slam_reset(SLAM(s))); }     /* SLAM.reset() :65-77 */
JNIEXPORT void JNICALL CLS(pmSetRefine)(JNIEnv *env, jclass c, jlong s, jboolean on) { throw_gms(env, gms_slam_set_refine(SLAM(s), on ? 1 : 0)); }   /* :96 */
/* SLAM.update(z, u) (:80-131): out3 = {weightSum, neff, strongest}; the motion-model variates are Philox(seed; particle, sequence) */
JNIEXPORT void JNICALL CLS(pmUpdate)(JNIEnv *env, jclass c, jlong s, jdoubleArray beams, jint B, jboolean sampleMotion, jdouble dCenter,
                                     jdouble dTheta, jlong seed, jlong sequence, jdoubleArray out3) {
    gms_beam *buf = beams_from(env, beams, B);
    if (!buf) return;
    gms_pf_stats st;
    const int rc = gms_slam_update_per_particle(SLAM(s), buf, B, sampleMotion ? 1 : 0, dCenter, dTheta, (uint64_t)seed, (uint64_t)sequence, &st);
    free(buf);
    if (!rc) stats_out(env, out3, &st);
    throw_gms(env, rc);
}
/* SLAM.resample() (:133-153) with r = Math.random() drawn on the Java side */
JNIEXPORT void JNICALL CLS(pmResample)(JNIEnv *env, jclass c, jlong s, jdouble r01) { throw_gms(env, gms_slam_resample_maps(SLAM(s), r01, NULL, NULL)); }
JNIEXPORT void JNICALL CLS(pmResampleIf)(JNIEnv *env, jclass c, jlong s, jdouble r01, jdouble fraction) { throw_gms(env, gms_slam_resample_maps_if(SLAM(s), r01, fraction)); }
/* getParticles() without the maps: {x, y, theta} and weight of every particle */
JNIEXPORT void JNICALL CLS(pmGetParticles)(JNIEnv *env, jclass c, jlong s, jfloatArray xyt, jdoubleArray w, jint n) {
    gms_pf *pf = NULL;
    int32_t have = 0;
    int rc = gms_slam_handles(SLAM(s), NULL, &pf);
    if (!rc) rc = gms_slam_count(SLAM(s), &have, NULL, NULL);
    if (rc) { throw_gms(env, rc); return; }
    if (have != n) { throw_named(env, "java/lang/IllegalArgumentException", "particle count differs from the native filter's"); return; }
    float *p = (float *)malloc((size_t)n * 3 * sizeof(float));
    double *ww = (double *)malloc((size_t)n * sizeof(double));
    if (!p || !ww) { free(p); free(ww); throw_named(env, "java/lang/OutOfMemoryError", "particle buffers"); return; }
    rc = gms_pf_get_poses(pf, p);
    if (!rc) rc = gms_pf_get_weights(pf, ww);
    if (!rc) { (*env)->SetFloatArrayRegion(env, xyt, 0, 3 * n, p); (*env)->SetDoubleArrayRegion(env, w, 0, n, ww); }
    free(p); free(ww);
    throw_gms(env, rc);
}
/* getWeightedPose() (:165-178) */
JNIEXPORT void JNICALL CLS(pmWeightedPose)(JNIEnv *env, jclass c, jlong s, jfloatArray out3) {
    gms_pf *pf = NULL;
    float o[3];
    int rc = gms_slam_handles(SLAM(s), NULL, &pf);
    if (!rc) rc = gms_pf_weighted_pose(pf, o);
    if (!rc) (*env)->SetFloatArrayRegion(env, out3, 0, 3, o);
    throw_gms(env, rc);
}
/* Particle i's GridMapData into the Java arrays (either may be null); the download synchronises, so it goes through heap buffers */
JNIEXPORT void JNICALL CLS(pmDownloadMap)(JNIEnv *env, jclass c, jlong s, jint i, jdoubleArray logData, jdoubleArray likelihoodData) {
    int32_t W = 0, H = 0;
    int rc = gms_slam_count(SLAM(s), NULL, &W, &H);
    if (rc) { throw_gms(env, rc); return; }
    const size_t n = (size_t)W * H;
    double *a = logData ? (double *)malloc(n * sizeof(double)) : NULL, *b = likelihoodData ? (double *)malloc(n * sizeof(double)) : NULL;
    if ((logData && !a) || (likelihoodData && !b)) { free(a); free(b); throw_named(env, "java/lang/OutOfMemoryError", "map buffer"); return; }
    rc = gms_slam_download_map(SLAM(s), i, a, b);
    if (!rc && a) (*env)->SetDoubleArrayRegion(env, logData, 0, (jsize)n, a);
    if (!rc && b) (*env)->SetDoubleArrayRegion(env, likelihoodData, 0, (jsize)n, b);
    free(a); free(b);
    throw_gms(env, rc);
}
/* GridMapApp.calculateCombined (J/app/GridMapApp.java:439-458) into the Java arrays of the combined GridMapData */
JNIEXPORT void JNICALL CLS(pmCombined)(JNIEnv *env, jclass c, jlong s, jdoubleArray logData, jdoubleArray likelihoodData) {
    gms_map *m = NULL;
    int32_t W = 0, H = 0;
    int rc = gms_slam_handles(SLAM(s), &m, NULL);
    if (!rc) rc = gms_slam_count(SLAM(s), NULL, &W, &H);
    if (!rc) rc = gms_slam_combined(SLAM(s));
    if (rc) { throw_gms(env, rc); return; }
    const size_t n = (size_t)W * H;
    double *tmp = (double *)malloc(n * sizeof(double));
    if (!tmp) { throw_named(env, "java/lang/OutOfMemoryError", "map buffer"); return; }
    if (logData) { rc = gms_map_download_log(m, tmp); if (!rc) (*env)->SetDoubleArrayRegion(env, logData, 0, (jsize)n, tmp); }
    if (!rc && likelihoodData) { rc = gms_map_download_likelihood(m, tmp); if (!rc) (*env)->SetDoubleArrayRegion(env, likelihoodData, 0, (jsize)n, tmp); }
    free(tmp);
    throw_gms(env, rc);
}
