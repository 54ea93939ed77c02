package lpsolver;

/**
 * JNI binding of liblpx.so (include/lpx.h) for the reference's Java host.  Thin by design: arrays in,
 * status + scalars out.  Built by jni/build.sh into liblpxjni.so; NOT compiled in the build image (no JDK
 * there) — see INTEGRATION.md.
 *
 * Each method names the lpx.h entry point it forwards to and the reference member that entry point
 * replaces.
 */
final class LpxNative {
  static {
    System.loadLibrary("lpxjni"); // liblpxjni.so links liblpx.so
  }

  private LpxNative() {}

  /** Arithmetic of every handle created afterwards (lpx.h LPX_OPT_FUSED / lpx_solve_options.fused).  Never called: the
   *  library chooses by size (fused multiply-add updates from 0.5 GiB of tableau, where they are 5-75 % faster; both binary
   *  modes leave the BigDecimal pivot sequence equally often, tests/golden/divergence_census.json).  true = fused everywhere;
   *  false = product and difference of every update rounded separately, as the reference's BigDecimal code rounds them
   *  (LPState.java:162): the opt-out. */
  static native void setFusedArithmetic(boolean on);

  /** lpx_solve — replaces LPSolver.solve(LPStandardForm) (LPSolver.java:78).
   *  a: row-major m*n.  restoreOrder: iteration order of stForm.coefficients.keySet() as variable indices
   *  (null: the default-name order).  out[0] = unrounded objective, out[1] = objective rounded to 6 decimals
   *  HALF_UP; pivots[0], pivots[1] = phase-1 / phase-2 pivot counts; perm (nullable, n+m) = final slot ->
   *  variable id.  Returns an lpx_status (0 = optimal). */
  static native int solve(int m, int n, double[] a, double[] b, double[] c, boolean maximize,
                          int[] restoreOrder, double[] out, long[] pivots, int[] perm);

  /** lpx_solve_multi — LPSolver.solve with the row blocks of the tableau on the GPUs `devices[0..ndev)` of this node
   *  (one handle, peer-to-peer exchange inside the decision kernels; phase 1 included).  Outputs as solve. */
  static native int solveMulti(int m, int n, double[] a, double[] b, double[] c, boolean maximize,
                               int[] restoreOrder, int[] devices, int ndev, double[] out, long[] pivots, int[] perm);

  /** lpx_state_create — replaces new LPState(A, b, c, v, variables, coefficients, m, n) (LPState.java:101). */
  static native long stateCreate(int m, int n, double[] a, double[] b, double[] c, double v, int[] perm);

  /** lpx_state_destroy. */
  static native void stateDestroy(long handle);

  /** lpx_get_entering — replaces LPState.getEntering() (LPState.java:274). */
  static native int getEntering(long handle);

  /** lpx_get_leaving — replaces LPState.getLeaving(int) (LPState.java:287); -2 = IllegalArgumentException. */
  static native int getLeaving(long handle, int entering);

  /** lpx_pivot — replaces LPState.pivot(int, int) (LPState.java:114).  Returns an lpx_status. */
  static native int pivot(long handle, int entering, int leaving);

  /** lpx_simplex_loop — the loop of LPSolver.simplex (LPSolver.java:101-107), device-resident.
   *  io[0] in/out: tracked slot (x0) or -1; io[1] out: pivots done.  Returns an lpx_status. */
  static native int simplexLoop(long handle, long maxPivots, long[] io);

  /** lpx_state_read — copies A (m*n), b, c, v[0] and perm (n+m) back; any array may be null. */
  static native int stateRead(long handle, double[] a, double[] b, double[] c, double[] v, int[] perm);

  /** lpx_status_message — the reference's exception text for a status. */
  static native String statusMessage(int status);
}
