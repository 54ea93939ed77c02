package lpsolver;

import java.math.BigDecimal;
import java.math.RoundingMode;
import java.util.HashMap;

/**
 * Drop-in for the hot path of {@link LPSolver}: the same {@code solve(LPStandardForm)} contract
 * (LPSolver.java:78-114) and the same {@code getEntering / getLeaving / pivot} operator triple (LPState.java:274, :287,
 * :114), executed by liblpx.so on an MI355X through {@link LpxNative}.  Lives in package {@code lpsolver} because
 * the reference keeps {@code LPStandardForm}'s and {@code LPState}'s fields package-private.
 *
 * <p>Differences a caller can observe, all deliberate (INTEGRATION.md section 2): arithmetic is IEEE fp64 with one
 * rounding per reference operation (same pivot sequence, objective within 1e-9, same 6-decimal result); the caller's
 * {@code stForm} is never modified (the reference pivots inside {@code stForm.A/b/c} and negates {@code stForm.c} for
 * a minimisation).
 *
 * <p>NOT compiled in this repository's build image (no JDK there); {@code tests/test_jni_shim_syntax.py} checks that
 * every {@code LpxNative} method used here is declared and that each declaration matches its C definition.
 */
public class LPSolverGpu {

  /** LPSolver.solve(LPStandardForm) (LPSolver.java:78): objective rounded to 6 decimals, HALF_UP (:113). */
  public BigDecimal solve(LPStandardForm stForm) throws LPException {
    final int m = stForm.m, n = stForm.n;
    final double[] a = new double[m * n], b = new double[m], c = new double[n];
    for (int i = 0; i < m; i++) {
      b[i] = stForm.b[i].doubleValue();
      for (int j = 0; j < n; j++) a[i * n + j] = stForm.A[i][j].doubleValue();
    }
    for (int j = 0; j < n; j++) c[j] = stForm.c[j].doubleValue();
    final double[] out = new double[2];
    final long[] pivots = new long[2];
    final int[] perm = new int[n + m];
    final int status =
        LpxNative.solve(m, n, a, b, c, stForm.maximize, keySetOrder(stForm), out, pivots, perm);
    throwFor(status);
    // the min -> max flip (:86-90) happens inside lpx_solve: out[0] is already sign-corrected
    return new BigDecimal(out[0]).setScale(6, RoundingMode.HALF_UP);
  }

  /** The same solve with the row blocks of the tableau on several GPUs of this node (lpx_solve_multi). */
  public BigDecimal solve(LPStandardForm stForm, int[] devices) throws LPException {
    final int m = stForm.m, n = stForm.n;
    final double[] a = new double[m * n], b = new double[m], c = new double[n];
    for (int i = 0; i < m; i++) {
      b[i] = stForm.b[i].doubleValue();
      for (int j = 0; j < n; j++) a[i * n + j] = stForm.A[i][j].doubleValue();
    }
    for (int j = 0; j < n; j++) c[j] = stForm.c[j].doubleValue();
    final double[] out = new double[2];
    final long[] pivots = new long[2];
    final int[] perm = new int[n + m];
    final int status =
        LpxNative.solveMulti(
            m, n, a, b, c, stForm.maximize, keySetOrder(stForm), devices, devices.length, out, pivots, perm);
    throwFor(status);
    return new BigDecimal(out[0]).setScale(6, RoundingMode.HALF_UP);
  }

  /**
   * Iteration order of {@code initial.coefficients.keySet()} in restoreInitialLP (LPSolver.java:213-217) as variable
   * indices: the objective rebuild there is an ORDERED sum of rounded terms, so the order is part of the result.  A
   * form without names gets default names x1..xn inside the library (null).  An empty key set substitutes nothing
   * (a zero-length array: lpx_solve_options.restore_order_len = 0).
   */
  static int[] keySetOrder(LPStandardForm stForm) {
    if (stForm.coefficients == null) return null;
    final int[] order = new int[stForm.coefficients.size()];
    int k = 0;
    for (String name : stForm.coefficients.keySet()) order[k++] = stForm.coefficients.get(name);
    return order;
  }

  /** lpx_status -> the reference's exception classes and messages (tests assert on the messages). */
  static void throwFor(int status) throws LPException {
    switch (status) {
      case 0: // LPX_OPTIMAL
        return;
      case 1: // LPX_UNBOUNDED            LPSolver.java:105
      case 3: // LPX_AUX_UNBOUNDED        :149
      case 4: // LPX_NO_DEGENERATE_PIVOT  :193
        throw new SolutionException(LpxNative.statusMessage(status));
      case 2: // LPX_INFEASIBLE           :173
        throw new LPException(LpxNative.statusMessage(status));
      case 5: // LPX_BAD_ARGUMENT         Validate.isTrue, LPState.java:288
        throw new IllegalArgumentException();
      case 6: // LPX_RESTORE_INDEX_FAULT  :231 (the reference's own defect, kept bug for bug)
        throw new ArrayIndexOutOfBoundsException();
      case 8: // LPX_DIVIDE_BY_ZERO       LPState.java:139
        throw new ArithmeticException("Division by zero");
      default: // LPX_DEVICE_ERROR, LPX_PIVOT_LIMIT
        throw new SolutionException("liblpx: " + LpxNative.statusMessage(status) + " (" + status + ")");
    }
  }

  /**
   * The LPState operator triple on the device.  Created from a reference LPState (slack form or auxiliary LP, as
   * convertIntoSlackForm / convertIntoAuxLP build it); {@link #writeBack} puts the tableau and the two name maps back
   * into a reference LPState, e.g. for the Spock specs that inspect them after a pivot.
   */
  public static final class GpuState implements AutoCloseable {
    private long handle;
    private final int m, n;
    private final String[] nameOfId; // variable id -> name (ids 0..n-1 nonbasic slots at creation, n..n+m-1 basic)

    public GpuState(LPState st) throws LPException {
      m = st.m;
      n = st.n;
      final double[] a = new double[m * n], b = new double[m], c = new double[n];
      for (int i = 0; i < m; i++) {
        b[i] = st.b[i].doubleValue();
        for (int j = 0; j < n; j++) a[i * n + j] = st.A[i][j].doubleValue();
      }
      for (int j = 0; j < n; j++) c[j] = st.c[j].doubleValue();
      nameOfId = new String[n + m];
      if (st.variables != null) {
        for (int slot = 0; slot < n + m; slot++) nameOfId[slot] = st.variables.get(slot);
      }
      handle = LpxNative.stateCreate(m, n, a, b, c, st.v == null ? 0.0 : st.v.doubleValue(), null);
      if (handle == 0) throw new SolutionException("liblpx: lpx_state_create failed");
    }

    /** LPState.getEntering() (LPState.java:274-285). */
    public int getEntering() {
      return LpxNative.getEntering(handle);
    }

    /** LPState.getLeaving(int) (LPState.java:287-305); IllegalArgumentException for a slot outside [0, n). */
    public int getLeaving(int entering) {
      final int l = LpxNative.getLeaving(handle, entering);
      if (l == -2) throw new IllegalArgumentException();
      return l;
    }

    /** LPState.pivot(int, int) (LPState.java:114-181, :311-320). */
    public void pivot(int entering, int leaving) throws LPException {
      throwFor(LpxNative.pivot(handle, entering, leaving));
    }

    /** The loop of LPSolver.simplex (LPSolver.java:101-107), device-resident; returns the pivots done. */
    public long simplex() throws LPException {
      final long[] io = {-1, 0};
      final int status = LpxNative.simplexLoop(handle, -1, io);
      throwFor(status);
      return io[1];
    }

    /**
     * A, b, c, v and the two HashMaps of {@code st} as the reference would hold them now: slot s carries the variable
     * perm[s], so {@code variables.put(s, name(perm[s]))} and {@code coefficients.put(name(perm[s]), s)} — what
     * exchangeIndexes (LPState.java:311-320) has done pivot by pivot.
     */
    public void writeBack(LPState st) throws LPException {
      final double[] a = new double[m * n], b = new double[m], c = new double[n], v = new double[1];
      final int[] perm = new int[n + m];
      final int rc = LpxNative.stateRead(handle, a, b, c, v, perm);
      if (rc != 0) throw new SolutionException("liblpx: lpx_state_read failed (" + rc + ")");
      for (int i = 0; i < m; i++) {
        st.b[i] = new BigDecimal(b[i]);
        for (int j = 0; j < n; j++) st.A[i][j] = new BigDecimal(a[i * n + j]);
      }
      for (int j = 0; j < n; j++) st.c[j] = new BigDecimal(c[j]);
      st.v = new BigDecimal(v[0]);
      if (st.variables != null && st.coefficients != null) {
        final HashMap<Integer, String> variables = new HashMap<>();
        final HashMap<String, Integer> coefficients = new HashMap<>();
        for (int slot = 0; slot < n + m; slot++) {
          final String name = nameOfId[perm[slot]];
          if (name == null) continue;
          variables.put(slot, name);
          coefficients.put(name, slot);
        }
        st.variables.clear();
        st.variables.putAll(variables);
        st.coefficients.clear();
        st.coefficients.putAll(coefficients);
      }
    }

    @Override
    public void close() {
      if (handle != 0) LpxNative.stateDestroy(handle);
      handle = 0;
    }
  }
}
