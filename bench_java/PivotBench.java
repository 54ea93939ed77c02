import java.math.BigDecimal;
import java.math.MathContext;
import java.math.RoundingMode;
import java.util.Random;
import java.util.concurrent.CountDownLatch;
import java.util.concurrent.ExecutorService;
import java.util.concurrent.Executors;

/**
 * CPU baseline harness (SURVEY §8d item 1): a build-authored restatement of the reference's concurrent pivot
 * (LPState.java:184-272) with the reference's own number type — java.math.BigDecimal under
 * MathContext(15, HALF_UP) — its fixed pool of 4 threads and its static [k*N/4, (k+1)*N/4) partitions in three
 * latch-separated phases, driven by the first-positive entering rule and the minimum-ratio test
 * (LPState.java:274-305).  Only run by bench.py when a JDK is present on the GPU box (none in the build image,
 * so this file has never been compiled there).
 *
 * Usage: java PivotBench m n pivots seed   ->  prints one line:  JAVA_PIVOTS_PER_SEC <value> <pivots> <seconds>
 * Data: A ~ U(0,1), b = (n/4) U(1,2), c ~ U(0,1) rounded to 6 decimals (same family as bench.py).
 */
public final class PivotBench {
  static final MathContext MC = new MathContext(15, RoundingMode.HALF_UP);
  static final BigDecimal EPS = new BigDecimal("1e-9");
  static final BigDecimal INF = new BigDecimal("1e50");
  static final int THREADS = 4;

  public static void main(String[] args) throws Exception {
    final int m = Integer.parseInt(args[0]), n = Integer.parseInt(args[1]);
    final int pivots = Integer.parseInt(args[2]);
    final long seed = args.length > 3 ? Long.parseLong(args[3]) : 1L;
    Random rng = new Random(seed);
    final BigDecimal[][] A = new BigDecimal[m][n];
    final BigDecimal[] b = new BigDecimal[m], c = new BigDecimal[n];
    for (int i = 0; i < m; i++) {
      for (int j = 0; j < n; j++) A[i][j] = rnd(rng, 0.0, 1.0);
      b[i] = rnd(rng, n / 4.0, n / 2.0);
    }
    for (int j = 0; j < n; j++) c[j] = rnd(rng, 0.0, 1.0);
    BigDecimal v = BigDecimal.ZERO;
    ExecutorService pool = Executors.newFixedThreadPool(THREADS);
    int done = 0;
    // two untimed warm-up pivots (JIT), then the timed ones
    long t0 = 0;
    for (int it = 0; it < pivots + 2; it++) {
      if (it == 2) t0 = System.nanoTime();
      int e = -1;
      for (int j = 0; j < n; j++) if (c[j].compareTo(EPS) > 0) { e = j; break; }
      if (e < 0) break;
      int l = -1;
      BigDecimal min = INF;
      for (int i = 0; i < m; i++) {
        BigDecimal a = A[i][e];
        BigDecimal s = a.compareTo(EPS) < 0 ? INF : b[i].divide(a, MC);
        if (s.compareTo(min) < 0) { min = s; l = i; }
      }
      if (l < 0) break;
      v = pivot(pool, A, b, c, v, m, n, e, l);
      if (it >= 2) done++;
    }
    double secs = (System.nanoTime() - t0) / 1e9;
    pool.shutdown();
    System.out.println("JAVA_PIVOTS_PER_SEC " + (done / secs) + " " + done + " " + secs);
  }

  static BigDecimal rnd(Random rng, double lo, double hi) {
    return new BigDecimal(lo + (hi - lo) * rng.nextDouble()).setScale(6, RoundingMode.HALF_UP);
  }

  static BigDecimal pivot(ExecutorService pool, final BigDecimal[][] A, final BigDecimal[] b, final BigDecimal[] c,
                          BigDecimal v, final int m, final int n, final int e, final int l) throws Exception {
    final BigDecimal[] prow = A[l];
    final BigDecimal piv = prow[e];
    prow[e] = BigDecimal.ONE.divide(piv, MC);
    final CountDownLatch l1 = new CountDownLatch(THREADS);
    for (int k = 0; k < THREADS; k++) {
      final int from = (k * n) / THREADS, to = ((k + 1) * n) / THREADS;
      pool.execute(() -> {
        for (int i = from; i < to; i++) if (i != e) prow[i] = prow[i].divide(piv, MC);
        l1.countDown();
      });
    }
    l1.await();
    b[l] = b[l].divide(piv, MC);
    final BigDecimal bEntering = b[l];
    final CountDownLatch l2 = new CountDownLatch(THREADS);
    for (int k = 0; k < THREADS; k++) {
      final int from = (k * m) / THREADS, to = ((k + 1) * m) / THREADS;
      pool.execute(() -> {
        for (int i = from; i < to; i++) {
          if (i == l) continue;
          BigDecimal[] row = A[i];
          BigDecimal ce = row[e];
          row[e] = ce.divide(piv, MC).negate();
          for (int j = 0; j < n; j++) {
            if (j == e) continue;
            row[j] = row[j].subtract(ce.multiply(prow[j], MC), MC);
          }
          b[i] = b[i].subtract(ce.multiply(bEntering, MC), MC);
        }
        l2.countDown();
      });
    }
    l2.await();
    final BigDecimal pc = c[e];
    v = v.add(b[l].multiply(pc, MC), MC);
    c[e] = pc.divide(piv, MC).negate();
    final CountDownLatch l3 = new CountDownLatch(THREADS);
    for (int k = 0; k < THREADS; k++) {
      final int from = (k * n) / THREADS, to = ((k + 1) * n) / THREADS;
      pool.execute(() -> {
        for (int i = from; i < to; i++) if (i != e) c[i] = c[i].subtract(pc.multiply(prow[i], MC), MC);
        l3.countDown();
      });
    }
    l3.await();
    return v;
  }
}
