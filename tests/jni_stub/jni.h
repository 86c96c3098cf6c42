/* Minimal declarations for a `gcc -fsyntax-only` pass over jni/gms_jni.c in an image without a JDK.
 * NOT a JNI implementation and never linked against: it only names the types and the JNIEnv entries that file uses,
 * with the signatures of the JNI specification, so that a typo or a wrong argument list in the shim is caught here.
 * A real build uses $JAVA_HOME/include/jni.h (jni/Makefile). */
#ifndef GMS_TEST_JNI_STUB_H
#define GMS_TEST_JNI_STUB_H
#include <stdint.h>
typedef uint8_t jboolean; typedef int8_t jbyte; typedef int32_t jint; typedef int64_t jlong; typedef float jfloat; typedef double jdouble;
typedef jint jsize;
struct _jobject; typedef struct _jobject *jobject;
typedef jobject jclass, jarray, jdoubleArray, jfloatArray, jbyteArray, jthrowable;
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;
struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv *, const char *);
    jint (*ThrowNew)(JNIEnv *, jclass, const char *);
    jboolean (*ExceptionCheck)(JNIEnv *);
    jsize (*GetArrayLength)(JNIEnv *, jarray);
    void *(*GetPrimitiveArrayCritical)(JNIEnv *, jarray, jboolean *);
    void (*ReleasePrimitiveArrayCritical)(JNIEnv *, jarray, void *, jint);
    void (*GetDoubleArrayRegion)(JNIEnv *, jdoubleArray, jsize, jsize, jdouble *);
    void (*SetDoubleArrayRegion)(JNIEnv *, jdoubleArray, jsize, jsize, const jdouble *);
    void (*GetFloatArrayRegion)(JNIEnv *, jfloatArray, jsize, jsize, jfloat *);
    void (*SetFloatArrayRegion)(JNIEnv *, jfloatArray, jsize, jsize, const jfloat *);
    void (*GetByteArrayRegion)(JNIEnv *, jbyteArray, jsize, jsize, jbyte *);
    void (*SetByteArrayRegion)(JNIEnv *, jbyteArray, jsize, jsize, const jbyte *);
};
#endif
