/* JNI glue between lpsolver.LpxNative (jni/java/lpsolver/LpxNative.java) and the C ABI of liblpx.so
 * (include/lpx.h).  Build with jni/build.sh on a machine that has a JDK; the build image has none, so this
 * file is shipped uncompiled. */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>

#include "../include/lpx.h"

#define NAME(fn) Java_lpsolver_LpxNative_##fn

/* Arithmetic of the handles this shim creates (LpxNative.setFusedArithmetic), as lpx_solve_options.fused takes it: 0 (never
 * called) = the library's choice by size (LPX_OPT_FUSED = 2: fused multiply-add updates from 0.5 GiB of tableau); 1 = fused
 * everywhere; -1 = product and difference of every update rounded separately, as the reference rounds them (the opt-out). */
static int g_fused = 0;
JNIEXPORT void JNICALL NAME(setFusedArithmetic)(JNIEnv* env, jclass cls, jboolean on) {
  (void)env; (void)cls;
  g_fused = on ? 1 : -1;
}

static double* get_d(JNIEnv* env, jdoubleArray a) { return a ? (*env)->GetDoubleArrayElements(env, a, NULL) : NULL; }
static void put_d(JNIEnv* env, jdoubleArray a, double* p, jint mode) { if (a && p) (*env)->ReleaseDoubleArrayElements(env, a, p, mode); }
static jint* get_i(JNIEnv* env, jintArray a) { return a ? (*env)->GetIntArrayElements(env, a, NULL) : NULL; }
static void put_i(JNIEnv* env, jintArray a, jint* p, jint mode) { if (a && p) (*env)->ReleaseIntArrayElements(env, a, p, mode); }

JNIEXPORT jint JNICALL NAME(solve)(JNIEnv* env, jclass cls, jint m, jint n, jdoubleArray ja, jdoubleArray jb,
                                   jdoubleArray jc, jboolean maximize, jintArray jorder, jdoubleArray jout,
                                   jlongArray jpivots, jintArray jperm) {
  (void)cls;
  double *a = get_d(env, ja), *b = get_d(env, jb), *c = get_d(env, jc);
  jint* order = get_i(env, jorder);
  jint* perm = get_i(env, jperm);
  lpx_solve_options opts = {0};
  opts.device = 0;
  opts.fused = g_fused;
  opts.max_pivots = -1;
  opts.restore_order = (const int32_t*)order;
  opts.restore_order_len = jorder ? (int32_t)(*env)->GetArrayLength(env, jorder) : 0;
  opts.perm_out = (int32_t*)perm;
  lpx_solve_result res;
  int status = lpx_solve(m, n, a, n, b, c, maximize ? 1 : 0, &opts, &res);
  if (jout) {
    jdouble o[2] = {res.objective, res.objective_rounded};
    (*env)->SetDoubleArrayRegion(env, jout, 0, 2, o);
  }
  if (jpivots) {
    jlong p[2] = {res.pivots_phase1, res.pivots_phase2};
    (*env)->SetLongArrayRegion(env, jpivots, 0, 2, p);
  }
  put_d(env, ja, a, JNI_ABORT); put_d(env, jb, b, JNI_ABORT); put_d(env, jc, c, JNI_ABORT);
  put_i(env, jorder, order, JNI_ABORT);
  put_i(env, jperm, perm, 0);
  return status;
}

/* lpx_solve_multi: LPSolver.solve with the row blocks of the tableau on several GPUs of the node (devices[] = HIP
 * ordinals).  Same outputs as solve. */
JNIEXPORT jint JNICALL NAME(solveMulti)(JNIEnv* env, jclass cls, jint m, jint n, jdoubleArray ja, jdoubleArray jb,
                                        jdoubleArray jc, jboolean maximize, jintArray jorder, jintArray jdevices,
                                        jint ndev, jdoubleArray jout, jlongArray jpivots, jintArray jperm) {
  (void)cls;
  double *a = get_d(env, ja), *b = get_d(env, jb), *c = get_d(env, jc);
  jint* order = get_i(env, jorder);
  jint* devices = get_i(env, jdevices);
  jint* perm = get_i(env, jperm);
  lpx_solve_options opts = {0};
  opts.max_pivots = -1;
  opts.fused = g_fused;
  opts.restore_order = (const int32_t*)order;
  opts.restore_order_len = jorder ? (int32_t)(*env)->GetArrayLength(env, jorder) : 0;
  opts.perm_out = (int32_t*)perm;
  lpx_solve_result res;
  int status = lpx_solve_multi(m, n, a, n, b, c, maximize ? 1 : 0, &opts, (const int32_t*)devices, ndev, &res);
  if (jout) {
    jdouble o[2] = {res.objective, res.objective_rounded};
    (*env)->SetDoubleArrayRegion(env, jout, 0, 2, o);
  }
  if (jpivots) {
    jlong p[2] = {res.pivots_phase1, res.pivots_phase2};
    (*env)->SetLongArrayRegion(env, jpivots, 0, 2, p);
  }
  put_d(env, ja, a, JNI_ABORT); put_d(env, jb, b, JNI_ABORT); put_d(env, jc, c, JNI_ABORT);
  put_i(env, jorder, order, JNI_ABORT);
  put_i(env, jdevices, devices, JNI_ABORT);
  put_i(env, jperm, perm, 0);
  return status;
}

JNIEXPORT jlong JNICALL NAME(stateCreate)(JNIEnv* env, jclass cls, jint m, jint n, jdoubleArray ja,
                                          jdoubleArray jb, jdoubleArray jc, jdouble v, jintArray jperm) {
  (void)cls;
  double *a = get_d(env, ja), *b = get_d(env, jb), *c = get_d(env, jc);
  jint* perm = get_i(env, jperm);
  lpx_state* s = NULL;
  int rc = lpx_state_create(m, n, a, n, b, c, v, (const int32_t*)perm, 0, m, 0, &s);
  if (rc == 0 && g_fused) rc = lpx_state_set_option(s, LPX_OPT_FUSED, g_fused > 0 ? 1 : 0);
  put_d(env, ja, a, JNI_ABORT); put_d(env, jb, b, JNI_ABORT); put_d(env, jc, c, JNI_ABORT);
  put_i(env, jperm, perm, JNI_ABORT);
  return rc == 0 ? (jlong)(intptr_t)s : 0;
}

JNIEXPORT void JNICALL NAME(stateDestroy)(JNIEnv* env, jclass cls, jlong h) {
  (void)env; (void)cls;
  lpx_state_destroy((lpx_state*)(intptr_t)h);
}

JNIEXPORT jint JNICALL NAME(getEntering)(JNIEnv* env, jclass cls, jlong h) {
  (void)env; (void)cls;
  int32_t e = -1;
  int rc = lpx_get_entering((lpx_state*)(intptr_t)h, &e);
  return rc == 0 ? e : -1000 - rc;
}

JNIEXPORT jint JNICALL NAME(getLeaving)(JNIEnv* env, jclass cls, jlong h, jint entering) {
  (void)env; (void)cls;
  int32_t l = -1;
  int rc = lpx_get_leaving((lpx_state*)(intptr_t)h, entering, &l, NULL);
  if (rc == LPX_BAD_ARGUMENT) return -2; /* Validate.isTrue -> IllegalArgumentException */
  return rc == 0 ? l : -1000 - rc;
}

JNIEXPORT jint JNICALL NAME(pivot)(JNIEnv* env, jclass cls, jlong h, jint e, jint l) {
  (void)env; (void)cls;
  return lpx_pivot((lpx_state*)(intptr_t)h, e, l);
}

JNIEXPORT jint JNICALL NAME(simplexLoop)(JNIEnv* env, jclass cls, jlong h, jlong max_pivots, jlongArray jio) {
  (void)cls;
  jlong io[2] = {-1, 0};
  if (jio) (*env)->GetLongArrayRegion(env, jio, 0, 2, io);
  int32_t track = (int32_t)io[0], status = 0;
  int64_t done = 0;
  int rc = lpx_simplex_loop((lpx_state*)(intptr_t)h, max_pivots, &done, &status, io[0] >= 0 ? &track : NULL);
  io[0] = track;
  io[1] = done;
  if (jio) (*env)->SetLongArrayRegion(env, jio, 0, 2, io);
  return rc ? rc : status;
}

JNIEXPORT jint JNICALL NAME(stateRead)(JNIEnv* env, jclass cls, jlong h, jdoubleArray ja, jdoubleArray jb,
                                       jdoubleArray jc, jdoubleArray jv, jintArray jperm) {
  (void)cls;
  lpx_state* s = (lpx_state*)(intptr_t)h;
  int32_t m = 0, n = 0;
  lpx_state_dims(s, &m, &n, NULL, NULL);
  double *a = get_d(env, ja), *b = get_d(env, jb), *c = get_d(env, jc), *v = get_d(env, jv);
  jint* perm = get_i(env, jperm);
  int rc = lpx_state_read(s, a, n, b, c, v, (int32_t*)perm);
  put_d(env, ja, a, 0); put_d(env, jb, b, 0); put_d(env, jc, c, 0); put_d(env, jv, v, 0);
  put_i(env, jperm, perm, 0);
  return rc;
}

JNIEXPORT jstring JNICALL NAME(statusMessage)(JNIEnv* env, jclass cls, jint status) {
  (void)cls;
  return (*env)->NewStringUTF(env, lpx_status_message(status));
}
