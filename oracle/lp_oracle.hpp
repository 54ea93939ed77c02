// ORACLE — TEST INFRASTRUCTURE ONLY (see dec15.hpp).  CPU restatement of the reference's simplex hot
// path, written once as a template over the number type so that the decimal-15 instantiation (the
// reference's BigDecimal semantics) and the IEEE fp64 instantiation (what the HIP kernels compute)
// execute the same operations in the same order with one rounding per reference operation.
//
// Follows, line by line:
//   LPState.java:114-181   pivot / pivotSequentially          -> State::pivot
//   LPState.java:184-272   pivotConcurrently (4 static parts) -> State::pivot(threads > 1)
//   LPState.java:274-285   getEntering                        -> State::get_entering
//   LPState.java:287-305   getLeaving                         -> State::get_leaving
//   LPState.java:311-320   exchangeIndexes                    -> State::exchange_indexes
//   LPSolver.java:78-114   solve / simplex                    -> solve()
//   LPSolver.java:116-198  initializeSimplex .. performDegeneratePivot
//   LPSolver.java:200-246  restoreInitialLP (bug-for-bug, incl. the aux-slot indexing defect)
//   LPSolver.java:283-321  convertIntoAuxLP,  :375-386 minInB
// The reference cannot be compiled or run in this image (no JVM); this restatement is pinned
// by the reference's Spock vectors (tests/golden/reference_vectors.json) and by goldens generated with
// Python's decimal module (tests/golden/gen_golden.py).
#pragma once
#include <cmath>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../include/lpx.h"
#include "dec15.hpp"

namespace lporacle {

// ---------------------------------------------------------------- number policies
template <class T> struct Num;

template <> struct Num<dec15::Dec> {
  typedef dec15::Dec T;
  static T from_double(double x) { return dec15::from_double(x); }
  static double to_double(const T& a) { return dec15::to_double(a); }
  static T zero() { return T(0, 0); }
  static T one() { return dec15::from_int(1); }
  static T eps() { return dec15::from_string("1e-9"); }   // DEF_EPSILON  LPState.java:20
  static T inf() { return dec15::from_string("1e50"); }   // DEF_INF      LPState.java:21
  static T add(const T& a, const T& b) { return dec15::add(a, b); }
  static T sub(const T& a, const T& b) { return dec15::sub(a, b); }
  static T mul(const T& a, const T& b) { return dec15::mul(a, b); }
  static T div(const T& a, const T& b) { return dec15::div(a, b); }
  // x - round15(c*r) / x + round15(a*b): the update forms of LPState.java:162,164,171,177 (two roundings each)
  static T submul(const T& x, const T& c, const T& r) { return dec15::sub(x, dec15::mul(c, r)); }
  static T addmul(const T& x, const T& a, const T& b) { return dec15::add(x, dec15::mul(a, b)); }
  static T neg(const T& a) { return dec15::neg(a); }
  static T abs(const T& a) { return dec15::abs(a); }
  static int cmp(const T& a, const T& b) { return dec15::cmp(a, b); }
  static bool is_zero(const T& a) { return a.c == 0; }
  // Rows whose multiplier A[i][e] is zero may be skipped: x - round15(0*y) == x for every stored value
  // (stored values carry <= 15 digits and decimal has no signed zero), so the skip is value-identical to
  // the reference's arithmetic — it only avoids the work, as BigDecimal's own zero fast paths do.  Pinned
  // by the Python-decimal goldens, whose generator does not skip.
  static constexpr bool kSkipZeroMultiplier = true;
  static std::string str(const T& a) { return dec15::to_string(a); }
};

template <> struct Num<double> {
  typedef double T;
  static T from_double(double x) { return x; }
  static double to_double(T a) { return a; }
  static T zero() { return 0.0; }
  static T one() { return 1.0; }
  static T eps() { return 1e-9; }
  static T inf() { return 1e50; }
  // One IEEE operation per reference operation; this file is compiled with -ffp-contract=off so the
  // product and the difference of LPState.java:162 stay two roundings, as they are in the reference.
  static T add(T a, T b) { return a + b; }
  static T sub(T a, T b) { return a - b; }
  static T mul(T a, T b) { return a * b; }
  static T div(T a, T b) { return a / b; }
  static T submul(T x, T c, T r) { return x - c * r; }   // two roundings (-ffp-contract=off)
  static T addmul(T x, T a, T b) { return x + a * b; }
  static T neg(T a) { return -a; }
  static T abs(T a) { return std::fabs(a); }
  static int cmp(T a, T b) { return a < b ? -1 : (a > b ? 1 : 0); }
  static bool is_zero(T a) { return a == 0.0; }
  // No skip in binary: x - (0 * y) turns x = -0.0 into +0.0 when 0*y = -0.0, and the GPU kernel (which
  // never skips) must match this instantiation bit for bit, signs of zeros included.
  static constexpr bool kSkipZeroMultiplier = false;
  static std::string str(T a) {
    char buf[40];
    snprintf(buf, sizeof buf, "%a", a);
    return buf;
  }
};

// IEEE fp64 with every update x - c*r (LPState.java:162, :164, :177) and x + a*b (:171; LPSolver.java:223, :227)
// computed as ONE fused multiply-add: the checker of the GPU's opt-in fused-arithmetic mode (LPX_OPT_FUSED).  The
// reference rounds the product and the difference to 15 DECIMAL digits each; one binary rounding is no further from
// that than two are (and is the more accurate of the two), but the bits differ from the unfused instantiation, so
// the mode gets an instantiation of its own.  Everything else (divisions, comparisons) is the fp64 instantiation's.
struct F64Fused {
  double x;
  F64Fused() : x(0.0) {}
  explicit F64Fused(double v) : x(v) {}
};
template <> struct Num<F64Fused> {
  typedef F64Fused T;
  static T from_double(double x) { return T(x); }
  static double to_double(T a) { return a.x; }
  static T zero() { return T(0.0); }
  static T one() { return T(1.0); }
  static T eps() { return T(1e-9); }
  static T inf() { return T(1e50); }
  static T add(T a, T b) { return T(a.x + b.x); }
  static T sub(T a, T b) { return T(a.x - b.x); }
  static T mul(T a, T b) { return T(a.x * b.x); }
  static T div(T a, T b) { return T(a.x / b.x); }
  static T submul(T x, T c, T r) { return T(std::fma(-c.x, r.x, x.x)); }   // one rounding: v_fma_f64 with a negated factor
  static T addmul(T x, T a, T b) { return T(std::fma(a.x, b.x, x.x)); }
  static T neg(T a) { return T(-a.x); }
  static T abs(T a) { return T(std::fabs(a.x)); }
  static int cmp(T a, T b) { return a.x < b.x ? -1 : (a.x > b.x ? 1 : 0); }
  static bool is_zero(T a) { return a.x == 0.0; }
  static constexpr bool kSkipZeroMultiplier = false;   // fma(-0, y, -0.0) keeps -0.0, but the GPU never skips either
  static std::string str(T a) {
    char buf[40];
    snprintf(buf, sizeof buf, "%a", a.x);
    return buf;
  }
};

struct DivideByZero {};

// ---------------------------------------------------------------- LPState
template <class T> struct State {
  typedef Num<T> N;
  int m = 0, n = 0;
  std::vector<T> A;  // m*n row-major
  std::vector<T> b, c;
  T v;
  std::vector<int32_t> perm;  // slot -> variable id (n+m entries); empty == "no variable names"
  // 0 = the reference's rule (first slot with c > eps, LPState.java:274-285).  1 = Dantzig (largest c, lowest
  // slot on ties): an OPT-IN extension of this build that deliberately leaves the
  // reference's pivot sequence; it exists in the oracle only so that the GPU's opt-in mode has a checker.
  int pricing = 0;

  T& a(int i, int j) { return A[(size_t)i * n + j]; }
  const T& a(int i, int j) const { return A[(size_t)i * n + j]; }

  // LPState.java:274-285
  int get_entering() const {
    const T eps = N::eps();
    if (pricing == 1) {
      int best = -1;
      for (int i = 0; i < n; i++)
        if (N::cmp(c[i], eps) > 0 && (best < 0 || N::cmp(c[i], c[best]) > 0)) best = i;
      return best;
    }
    for (int i = 0; i < n; i++)
      if (N::cmp(c[i], eps) > 0) return i;
    return -1;
  }

  // LPState.java:287-305 (Validate.isTrue -> returns -2 here)
  int get_leaving(int entering, T* min_ratio = nullptr) const {
    if (!(entering >= 0 && entering < n)) return -2;
    const T eps = N::eps(), INF = N::inf();
    int leaving = -1;
    T min_slack = INF, slack;
    for (int i = 0; i < m; i++) {
      const T& aie = a(i, entering);
      if (N::cmp(aie, eps) < 0) slack = INF;
      else slack = N::div(b[i], aie);
      if (N::cmp(slack, min_slack) < 0) { min_slack = slack; leaving = i; }
    }
    if (min_ratio) *min_ratio = min_slack;
    return leaving;
  }

  // LPState.java:311-320
  void exchange_indexes(int entering, int leaving) {
    if (perm.empty()) return;
    std::swap(perm[entering], perm[leaving + n]);
  }

  // rows [from,to) of "recalculate other rows"  LPState.java:151-166 / :225-240
  void update_rows(int from, int to, int entering, int leaving, const T& piv, const T& b_entering) {
    const T* prow = &A[(size_t)leaving * n];
    for (int i = from; i < to; i++) {
      if (i == leaving) continue;
      T* row = &A[(size_t)i * n];
      const T ce = row[entering];
      row[entering] = N::neg(N::div(ce, piv));                                    // :157
      if (N::kSkipZeroMultiplier && N::is_zero(ce)) continue;
      for (int j = 0; j < n; j++) {
        if (j == entering) continue;
        row[j] = N::submul(row[j], ce, prow[j]);                                  // :162
      }
      b[i] = N::submul(b[i], ce, b_entering);                                     // :164
    }
  }

  // LPState.java:133-181.  threads > 1 reproduces pivotConcurrently's three latch-separated phases with
  // the static [k*N/T, (k+1)*N/T) partitions of :195-269 (element results are identical by construction).
  void pivot(int entering, int leaving, int threads = 1) {
    T* prow = &A[(size_t)leaving * n];
    const T piv = prow[entering];
    if (N::is_zero(piv)) throw DivideByZero();
    prow[entering] = N::div(N::one(), piv);                                       // :139
#pragma omp parallel for num_threads(threads) schedule(static) if (threads > 1)
    for (int i = 0; i < n; i++) {
      if (i == entering) continue;
      prow[i] = N::div(prow[i], piv);                                             // :144
    }
    b[leaving] = N::div(b[leaving], piv);                                         // :146
    const T b_entering = b[leaving];
    if (threads > 1) {
#pragma omp parallel num_threads(threads)
      {
#pragma omp for schedule(static, 1)
        for (int k = 0; k < threads; k++) {
          int from = (int)(((int64_t)k * m) / threads), to = (int)(((int64_t)(k + 1) * m) / threads);
          update_rows(from, to, entering, leaving, piv, b_entering);
        }
      }
    } else {
      update_rows(0, m, entering, leaving, piv, b_entering);
    }
    const T pc = c[entering];                                                      // :170
    v = N::addmul(v, b[leaving], pc);                                             // :171
    c[entering] = N::neg(N::div(pc, piv));                                        // :172
#pragma omp parallel for num_threads(threads) schedule(static) if (threads > 1)
    for (int i = 0; i < n; i++) {
      if (i == entering) continue;
      c[i] = N::submul(c[i], pc, prow[i]);                                        // :177
    }
    exchange_indexes(entering, leaving);                                           // :180
  }
};

// ---------------------------------------------------------------- LPSolver
struct PivotRecord { int32_t phase, entering, leaving; };

template <class T> struct SolveOut {
  int status = LPX_OPTIMAL;
  bool phase1_used = false;
  T v;                       // unrounded LPState.v (maximisation sign convention)
  bool negate_result = false;
  int64_t pivots1 = 0, pivots2 = 0;
  int x0_slot = -1;
  State<T> final_state;
  std::vector<PivotRecord> trace;
};

// LPSolver.java:375-386
template <class T> int min_in_b(const std::vector<T>& b) {
  typedef Num<T> N;
  T mn = N::inf();
  int idx = -1;
  for (size_t i = 0; i < b.size(); i++)
    if (N::cmp(mn, b[i]) > 0) { mn = b[i]; idx = (int)i; }
  return idx;
}

// java.lang.String.hashCode of "x<k>" and java.util.HashMap's iteration order after put("x1").."x<n>"
// into `new HashMap<>()` (LPSolver.addDefaultVariables, LPSolver.java:388-400): table doubles from 16
// whenever size exceeds 0.75*capacity; iteration walks buckets in index order, and entries of one bucket
// in insertion order (resize preserves relative order; a bucket only becomes a tree at >= 8 entries,
// which "x<k>" keys never reach for n below 2^24 — checked, falls back to insertion order otherwise).
inline std::vector<int32_t> java_default_name_order(int n) {
  std::vector<int32_t> order;
  if (n <= 0) return order;
  size_t cap = 16;
  while ((double)n > 0.75 * (double)cap) cap <<= 1;
  std::vector<std::vector<int32_t>> buckets(cap);
  for (int k = 1; k <= n; k++) {
    char name[32];
    int len = snprintf(name, sizeof name, "x%d", k);
    int32_t h = 0;
    for (int i = 0; i < len; i++) h = (int32_t)((uint32_t)h * 31u + (uint32_t)(unsigned char)name[i]);
    uint32_t hh = (uint32_t)h ^ ((uint32_t)h >> 16);
    buckets[hh & (cap - 1)].push_back(k - 1);
  }
  for (auto& bk : buckets) {
    if (bk.size() >= 8) {  // treeified bin: order no longer insertion order -> documented fallback
      order.clear();
      for (int k = 0; k < n; k++) order.push_back(k);
      return order;
    }
  }
  for (auto& bk : buckets)
    for (int32_t id : bk) order.push_back(id);
  return order;
}

template <class T>
int simplex_loop(State<T>& st, int phase, int64_t max_pivots, int64_t& pivots, int* track_slot,
                 std::vector<PivotRecord>* trace, int threads, int unbounded_status) {
  for (;;) {
    int e = st.get_entering();                      // LPSolver.java:101 / :142
    if (e == -1) return LPX_OPTIMAL;
    int l = st.get_leaving(e);                      // :102 / :146
    if (l == -1) return unbounded_status;           // :103-106 / :147-150
    if (max_pivots >= 0 && pivots >= max_pivots) return LPX_PIVOT_LIMIT;
    if (track_slot) {                               // :151-155
      if (e == *track_slot) *track_slot = l + st.n;
      else if (l + st.n == *track_slot) *track_slot = e;
    }
    st.pivot(e, l, threads);                        // :107 / :156
    pivots++;
    if (trace) trace->push_back(PivotRecord{phase, e, l});
  }
}

// convertIntoAuxLP  LPSolver.java:283-321.  Variable ids: originals 0..n-1, slacks n..n+m-1, x0 = n+m.
template <class T>
void convert_into_aux_lp(int m, int n, const std::vector<T>& A_in, const std::vector<T>& b_in, State<T>& aux) {
  typedef Num<T> N;
  const int na = n + 1;
  aux.m = m; aux.n = na;
  aux.A.resize((size_t)m * na);
  const T minus_one = N::neg(N::one());
  for (int i = 0; i < m; i++) {
    for (int j = 0; j < n; j++) aux.a(i, j) = A_in[(size_t)i * n + j];                // :292
    aux.a(i, n) = minus_one;                                                       // :293
  }
  aux.b = b_in;                                                                    // :296-297
  aux.c.assign(na, N::zero());                                                     // :299-301
  aux.c[n] = minus_one;
  aux.v = N::zero();
  aux.perm.resize(na + m);
  for (int j = 0; j < n; j++) aux.perm[j] = j;
  aux.perm[n] = n + m;
  for (int i = 0; i < m; i++) aux.perm[na + i] = n + i;
}

// solveAuxLP  LPSolver.java:135-164.  Returns the loop status; x0 receives x0CurrentIndex.
template <class T>
int solve_aux_lp(State<T>& aux, int index_of_x0, int mib, int64_t max_pivots, int64_t& pivots, int& x0,
                 std::vector<PivotRecord>* trace, int threads) {
  const int na = aux.n;
  aux.pivot(index_of_x0, mib, threads);                                            // :138
  pivots++;
  if (trace) trace->push_back(PivotRecord{1, index_of_x0, mib});
  x0 = mib + na;                                                                   // :139
  return simplex_loop(aux, 1, max_pivots, pivots, &x0, trace, threads, LPX_AUX_UNBOUNDED);
}

// restoreInitialLP  LPSolver.java:200-246 (bug-for-bug).  c0 = initial.c (already negated for `min`),
// restore_order = iteration order of initial.coefficients.keySet() as original-variable indices.
template <class T>
int restore_initial_lp(const State<T>& aux, const std::vector<T>& c0, int n, int x0,
                       const std::vector<int32_t>& restore_order, State<T>& st) {
  typedef Num<T> N;
  const int m = aux.m, na = aux.n;
  st.m = m; st.n = n;
  st.A.resize((size_t)m * n);
  for (int i = 0; i < m; i++) {                                                    // :208-211
    for (int j = 0; j < x0; j++) st.a(i, j) = aux.a(i, j);
    for (int j = x0; j < n; j++) st.a(i, j) = aux.a(i, j + 1);
  }
  std::vector<int32_t> slot_of(n + m + 1, -1);                                     // auxLP.coefficients
  for (int s = 0; s < na + m; s++) slot_of[aux.perm[s]] = s;
  T v = N::zero();
  std::vector<T> c(n, N::zero());
  for (int32_t index : restore_order) {                                            // :217
    const T& k = c0[index];                                                        // :219
    int cur = slot_of[index];                                                      // :220
    if (cur >= na) {                                                               // :221
      const int r = cur - na;
      v = N::addmul(v, aux.b[r], k);                                               // :223
      for (int j = 0; j < n; j++) {
        T coef = N::neg(st.a(r, j));                                               // :226
        c[j] = N::addmul(c[j], coef, k);                                           // :227
      }
    } else {
      if (cur >= n) return LPX_RESTORE_INDEX_FAULT;  // ArrayIndexOutOfBoundsException at :231
      c[cur] = N::add(c[cur], k);  // :231 — bug-for-bug: `cur` is an aux-LP slot, c is post-drop numbering
    }
  }
  st.b = aux.b;
  st.c = c;
  st.v = v;
  st.perm.clear();                                                                 // :235-244
  for (int s = 0; s < na + m; s++)
    if (s != x0) st.perm.push_back(aux.perm[s]);
  return LPX_OPTIMAL;
}

// LPSolver.solve (LPSolver.java:78-94) on a copy of the standard form.
// restore_order: iteration order of initial.coefficients.keySet() (original variable indices); empty =
// java_default_name_order(n).
template <class T>
void solve(int m, int n, const std::vector<T>& A_in, const std::vector<T>& b_in,
           const std::vector<T>& c_in, bool maximize, std::vector<int32_t> restore_order,
           int64_t max_pivots, int threads, bool want_trace, SolveOut<T>& out, int pricing = 0) {
  typedef Num<T> N;
  std::vector<PivotRecord>* trace = want_trace ? &out.trace : nullptr;
  out.final_state.pricing = pricing;
  std::vector<T> c0 = c_in;
  out.negate_result = !maximize;
  if (!maximize)
    for (auto& x : c0) x = N::neg(x);                                              // :86-89

  State<T>& st = out.final_state;
  int mib = min_in_b(b_in);                                                        // :118
  if (mib == -1 || N::cmp(b_in[mib], N::zero()) >= 0) {                            // :119
    // convertIntoSlackForm  :248-272
    st.m = m; st.n = n; st.A = A_in; st.b = b_in; st.c = c0; st.v = N::zero();
    st.perm.resize(n + m);
    for (int i = 0; i < n + m; i++) st.perm[i] = i;
  } else {
    out.phase1_used = true;
    State<T> aux;
    aux.pricing = pricing;
    convert_into_aux_lp(m, n, A_in, b_in, aux);                                    // :128
    const int na = aux.n;
    int x0 = -1;
    int stt = solve_aux_lp(aux, na - 1, mib, max_pivots, out.pivots1, x0, trace, threads);   // :130
    if (stt != LPX_OPTIMAL) { out.status = stt; out.v = aux.v; out.x0_slot = x0; out.final_state = aux; return; }
    // handleInitialization  :166-180
    T x0_value = (x0 < na) ? N::zero() : aux.b[x0 - na];
    if (N::cmp(N::abs(x0_value), N::eps()) > 0) {                                  // :171
      out.status = LPX_INFEASIBLE; out.v = aux.v; out.x0_slot = x0; out.final_state = aux; return;
    }
    if (x0 >= na) {                                                                // :175  performDegeneratePivot :182-198
      int entering = -1;
      const int row = x0 - na;
      for (int i = 0; i < na; i++)
        if (N::cmp(N::abs(aux.a(row, i)), N::eps()) > 0) { entering = i; break; }
      if (entering == -1) {
        out.status = LPX_NO_DEGENERATE_PIVOT; out.v = aux.v; out.x0_slot = x0; out.final_state = aux; return;
      }
      aux.pivot(entering, row, threads);                                           // :195
      out.pivots1++;
      if (trace) trace->push_back(PivotRecord{1, entering, row});
      x0 = entering;
    }
    out.x0_slot = x0;
    if (restore_order.empty()) restore_order = java_default_name_order(n);
    int rs = restore_initial_lp(aux, c0, n, x0, restore_order, st);                // :179
    st.pricing = pricing;
    if (rs != LPX_OPTIMAL) { out.status = rs; out.v = aux.v; out.final_state = aux; return; }
  }
  int64_t lim2 = max_pivots < 0 ? -1 : (max_pivots - out.pivots1);
  if (max_pivots >= 0 && lim2 < 0) lim2 = 0;
  out.status = simplex_loop(st, 2, lim2, out.pivots2, (int*)nullptr, trace, threads, LPX_UNBOUNDED);
  out.v = st.v;
}

}  // namespace lporacle
