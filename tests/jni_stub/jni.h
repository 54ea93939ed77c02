/* TEST STUB — not a JDK header.  Declares only the JNI types, constants and JNIEnv members that jni/lpx_jni.c
 * uses, with the signatures the JNI specification gives them, so that `cc -fsyntax-only -Werror` can type-check the
 * shim against include/lpx.h on a machine without a JDK (tests/test_jni_shim_syntax.py).  Nothing links against
 * it; jni/build.sh uses the real <jni.h> of a JDK. */
#ifndef LPX_TEST_JNI_STUB_H
#define LPX_TEST_JNI_STUB_H
#include <stdint.h>

typedef int32_t jint;
typedef int64_t jlong;
typedef double jdouble;
typedef uint8_t jboolean;
typedef jint jsize;

struct _jobject;
typedef struct _jobject* jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jdoubleArray;
typedef jarray jintArray;
typedef jarray jlongArray;

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2

struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;

struct JNINativeInterface_ {
  jdouble* (*GetDoubleArrayElements)(JNIEnv* env, jdoubleArray array, jboolean* isCopy);
  void (*ReleaseDoubleArrayElements)(JNIEnv* env, jdoubleArray array, jdouble* elems, jint mode);
  jint* (*GetIntArrayElements)(JNIEnv* env, jintArray array, jboolean* isCopy);
  void (*ReleaseIntArrayElements)(JNIEnv* env, jintArray array, jint* elems, jint mode);
  void (*SetDoubleArrayRegion)(JNIEnv* env, jdoubleArray array, jsize start, jsize len, const jdouble* buf);
  void (*SetLongArrayRegion)(JNIEnv* env, jlongArray array, jsize start, jsize len, const jlong* buf);
  void (*GetLongArrayRegion)(JNIEnv* env, jlongArray array, jsize start, jsize len, jlong* buf);
  jstring (*NewStringUTF)(JNIEnv* env, const char* utf);
  jsize (*GetArrayLength)(JNIEnv* env, jarray array);
};
#endif
