/* A minimal JNIEnv for EXECUTING jni/gms_jni.c without a JVM (tests/test_gpu_jni_stub.py; this image has no JDK).
 * Test infrastructure, not a JNI implementation: exactly the twelve JNIEnv entries the shim uses (tests/jni_stub/jni.h), over
 * "Java arrays" that are plain C buffers with a length.  It enforces the two JNI rules the shim must keep: array regions are
 * bounds-checked (ArrayIndexOutOfBoundsException pending, nothing copied), and between GetPrimitiveArrayCritical and its Release no
 * other JNI function may be called (counted in stub_violations).  A thrown exception stays pending until the driver takes it
 * (stub_take_exception), as it would until the native returns to the JVM. */
#include <jni.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct _jobject {
    int32_t kind;          /* 1 double[], 2 float[], 3 byte[], 9 class */
    jsize len;
    void *data;
};

static int g_pending, g_critical, g_violations;
static char g_exc_class[160], g_exc_msg[600];

static void guard(void) { if (g_critical) g_violations++; }
static size_t elem_size(int32_t kind) { return kind == 1 ? 8 : (kind == 2 ? 4 : 1); }

static jclass env_FindClass(JNIEnv *env, const char *name) {
    (void)env; guard();
    struct _jobject *c = (struct _jobject *)calloc(1, sizeof(*c));      /* (leaked: a handful per test run) */
    c->kind = 9; c->data = strdup(name);
    return c;
}
static jint env_ThrowNew(JNIEnv *env, jclass cls, const char *msg) {
    (void)env; guard();
    g_pending = 1;
    snprintf(g_exc_class, sizeof(g_exc_class), "%s", cls && cls->kind == 9 ? (const char *)cls->data : "?");
    snprintf(g_exc_msg, sizeof(g_exc_msg), "%s", msg ? msg : "");
    return 0;
}
static jboolean env_ExceptionCheck(JNIEnv *env) { (void)env; guard(); return (jboolean)g_pending; }
static jsize env_GetArrayLength(JNIEnv *env, jarray a) { (void)env; guard(); return a->len; }
static void *env_GetPrimitiveArrayCritical(JNIEnv *env, jarray a, jboolean *is_copy) {
    (void)env; guard();
    if (is_copy) *is_copy = 0;
    g_critical++;
    return a->data;
}
static void env_ReleasePrimitiveArrayCritical(JNIEnv *env, jarray a, void *p, jint mode) {
    (void)env; (void)mode;
    if (p != a->data || g_critical <= 0) g_violations++;
    else g_critical--;
}
static int region_ok(JNIEnv *env, jarray a, int32_t kind, jsize start, jsize len) {
    guard();
    if (!a || a->kind != kind || start < 0 || len < 0 || (int64_t)start + len > a->len) {
        struct _jobject c = { 9, 0, (void *)"java/lang/ArrayIndexOutOfBoundsException" };
        env_ThrowNew(env, &c, "array region out of bounds");
        return 0;
    }
    return 1;
}
#define REGION(NAME, KIND, T, ARR)                                                                                     \
    static void env_Get##NAME##ArrayRegion(JNIEnv *env, ARR a, jsize start, jsize len, T *buf) {                        \
        if (region_ok(env, a, KIND, start, len)) memcpy(buf, (const T *)a->data + start, (size_t)len * sizeof(T));     \
    }                                                                                                                   \
    static void env_Set##NAME##ArrayRegion(JNIEnv *env, ARR a, jsize start, jsize len, const T *buf) {                  \
        if (region_ok(env, a, KIND, start, len)) memcpy((T *)a->data + start, buf, (size_t)len * sizeof(T));           \
    }
REGION(Double, 1, jdouble, jdoubleArray)
REGION(Float, 2, jfloat, jfloatArray)
REGION(Byte, 3, jbyte, jbyteArray)

static const struct JNINativeInterface_ g_table = {
    env_FindClass, env_ThrowNew, env_ExceptionCheck, env_GetArrayLength, env_GetPrimitiveArrayCritical, env_ReleasePrimitiveArrayCritical,
    env_GetDoubleArrayRegion, env_SetDoubleArrayRegion, env_GetFloatArrayRegion, env_SetFloatArrayRegion, env_GetByteArrayRegion,
    env_SetByteArrayRegion,
};
static JNIEnv g_env = &g_table;

/* ---- what the driver (ctypes) calls ---- */
#define STUB __attribute__((visibility("default")))
STUB JNIEnv *stub_env(void) { return &g_env; }
STUB jarray stub_new_array(int32_t kind, jsize len) {
    struct _jobject *a = (struct _jobject *)calloc(1, sizeof(*a));
    a->kind = kind; a->len = len;
    a->data = calloc((size_t)(len > 0 ? len : 1), elem_size(kind));
    return a;
}
STUB void *stub_array_data(jarray a) { return a->data; }
STUB void stub_free_array(jarray a) { if (a) { free(a->data); free(a); } }
/* the pending exception's class and message (empty strings: none); clears it, as the return to Java would deliver it */
STUB int stub_take_exception(char *cls, char *msg, int32_t cap) {
    const int had = g_pending;
    snprintf(cls, (size_t)cap, "%s", had ? g_exc_class : "");
    snprintf(msg, (size_t)cap, "%s", had ? g_exc_msg : "");
    g_pending = 0; g_exc_class[0] = 0; g_exc_msg[0] = 0;
    return had;
}
STUB int stub_violations(void) { return g_violations + g_critical; }     /* JNI calls inside a critical region + regions never released */
