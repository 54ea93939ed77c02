// LPSolver.solve behind the C ABI (include/lpx.h): the driver logic of the reference's LPSolver.java —
// min->max flip (:86-90), initializeSimplex (:116-133), the auxiliary LP of phase 1 (:135-198, :283-321)
// and restoreInitialLP (:200-246, reproduced bug-for-bug) — around the device-resident pivot loop.  Only
// O(n + m) scalars ever cross the host/device boundary after the upload: the tableau stays in HBM.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lpx.h"
#include "lpx_kernels.h"

namespace lpx_internal {
int alloc(int32_t m, int32_t n, int32_t n_cap, int device, lpx_state** out);
void destroy(lpx_state* s);
lpxk::Buffers& buffers(lpx_state* s);
hipStream_t stream(lpx_state* s);
lpxk::LpxCtl* host_ctl(lpx_state* s);
int pull_ctl(lpx_state* s);
int push(lpx_state* s);
void reset_ctl(lpx_state* s, double v);
void set_n(lpx_state* s, int32_t n);
int32_t get_n(lpx_state* s);
int32_t get_m(lpx_state* s);
int set_error(int status, const char* msg);
}  // namespace lpx_internal

using namespace lpx_internal;

#define HIP_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) return set_error(LPX_DEVICE_ERROR, hipGetErrorString(_e));   \
  } while (0)

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// java.util.HashMap<String,Integer> iteration order after put("x1") .. put("xn") into `new HashMap<>()`
// (LPSolver.addDefaultVariables, LPSolver.java:388-400): String.hashCode is s[0]*31^(k-1)+...; HashMap
// spreads h ^ (h >>> 16), indexes with (cap-1) & hash, doubles the table from 16 whenever size exceeds
// 0.75*cap (splitting preserves relative order) and iterates buckets in index order, entries of a bucket
// in insertion order.  A bucket is only treeified at >= 8 entries; "x<k>" keys stay far below that, and if
// they ever did not the order falls back to insertion order (documented deviation).
extern "C" int lpx_java_default_name_order(int32_t n, int32_t* order_out) {
  if (n < 0 || (n > 0 && !order_out)) return set_error(LPX_BAD_ARGUMENT, "lpx_java_default_name_order: bad argument");
  if (n == 0) return 0;
  size_t cap = 16;
  while ((double)n > 0.75 * (double)cap) cap <<= 1;
  std::vector<int32_t> count(cap, 0), bucket_of(n);
  for (int k = 1; k <= n; k++) {
    char name[32];
    const int len = snprintf(name, sizeof name, "x%d", k);
    uint32_t h = 0;
    for (int i = 0; i < len; i++) h = h * 31u + (uint32_t)(unsigned char)name[i];
    h ^= h >> 16;
    bucket_of[k - 1] = (int32_t)(h & (cap - 1));
    count[bucket_of[k - 1]]++;
  }
  bool treeified = false;
  for (size_t i = 0; i < cap; i++) treeified |= count[i] >= 8;
  if (treeified) {
    for (int k = 0; k < n; k++) order_out[k] = k;
    return 0;
  }
  std::vector<int32_t> start(cap + 1, 0);
  for (size_t i = 0; i < cap; i++) start[i + 1] = start[i] + count[i];
  std::vector<int32_t> fill(start.begin(), start.end() - 1);
  for (int k = 0; k < n; k++) order_out[fill[bucket_of[k]]++] = k;  // stable: insertion order inside a bucket
  return 0;
}

// new BigDecimal(v).setScale(6, RoundingMode.HALF_UP) (LPSolver.java:113) as text: the exact binary value
// of the double is expanded and the first discarded digit decides.
static void round6_text(double v, char* out, size_t cap) {
  static thread_local char buf[1400];
  if (!std::isfinite(v)) { snprintf(out, cap, "%g", v); return; }
  snprintf(buf, sizeof buf, "%.1100f", v);
  std::string s(buf);
  const bool neg = s[0] == '-';
  const size_t dot = s.find('.');
  std::string digits = s.substr(neg ? 1 : 0, dot - (neg ? 1 : 0)) + s.substr(dot + 1, 6);
  if (s[dot + 7] >= '5') {
    int i = (int)digits.size() - 1;
    while (i >= 0) {
      if (digits[i] == '9') { digits[i] = '0'; i--; }
      else { digits[i]++; break; }
    }
    if (i < 0) digits.insert(digits.begin(), '1');
  }
  bool all_zero = true;
  for (char ch : digits) all_zero &= ch == '0';
  std::string r = digits.substr(0, digits.size() - 6) + "." + digits.substr(digits.size() - 6);
  if (neg && !all_zero) r = "-" + r;  // BigDecimal has no negative zero
  snprintf(out, cap, "%s", r.c_str());
}

// LPSolver.java:375-386
static int min_in_b(const double* b, int m) {
  double mn = 1e50;
  int idx = -1;
  for (int i = 0; i < m; i++)
    if (mn > b[i]) { mn = b[i]; idx = i; }
  return idx;
}

namespace {
struct Cleanup {
  lpx_state* s = nullptr;
  lpxk::RestoreEntry* d_ent = nullptr;
  ~Cleanup() {
    if (d_ent) (void)hipFree(d_ent);
    if (s) destroy(s);
  }
};
}  // namespace

// restoreInitialLP(auxLP, initial, indexOfX0)                          LPSolver.java:200-246
// In place on the auxiliary-LP state `s` (m x (n+1)): drop x0's column (:206-211), rebuild c and v by
// substituting the basic original variables (:213-233, in keySet() order, bug-for-bug: a NONBASIC original
// variable is credited at its aux-LP slot although c is already in post-drop numbering; slot n faults like
// the reference's ArrayIndexOutOfBoundsException), renumber the slots above x0 (:235-244).  c0 = initial.c
// (already negated for `min`), order = iteration order of initial.coefficients.keySet() or NULL (default names).
extern "C" int lpx_restore_initial_lp(lpx_state* s, const double* c0, int32_t n, int32_t x0_slot,
                                      const int32_t* order_in, int32_t order_len) {
  if (!s || (n > 0 && !c0)) return set_error(LPX_BAD_ARGUMENT, "lpx_restore_initial_lp: NULL argument");
  const int na = get_n(s), m = get_m(s);
  if (na != n + 1 || x0_slot < 0 || x0_slot >= na) return set_error(LPX_BAD_ARGUMENT, "lpx_restore_initial_lp: bad shape/slot");
  hipStream_t st = stream(s);
  lpxk::Buffers& B = buffers(s);
  HIP_TRY(hipStreamSynchronize(st));
  std::vector<int32_t> perm((size_t)na + m);
  HIP_TRY(hipMemcpy(perm.data(), B.perm, perm.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  std::vector<int32_t> slot_of((size_t)n + m + 1, -1);                                // auxLP.coefficients
  for (int sl = 0; sl < na + m; sl++)
    if (perm[sl] >= 0 && perm[sl] <= n + m) slot_of[perm[sl]] = sl;
  // keySet() may hold fewer names than the LP has variables (a named form from getDual() with m > n names only
  // min(n, m) of them, LPStandardForm.java:139-142): the reference then substitutes only those
  if (order_in && (order_len < 0 || order_len > n)) return set_error(LPX_BAD_ARGUMENT, "lpx_restore_initial_lp: bad order_len");
  std::vector<int32_t> order(order_in ? (size_t)order_len : (size_t)n);
  if (order_in) order.assign(order_in, order_in + order_len);
  else lpx_java_default_name_order(n, order.data());
  std::vector<lpxk::RestoreEntry> ent;
  ent.reserve(n);
  for (int32_t index : order) {                                                      // :217
    if (index < 0 || index >= n) return set_error(LPX_BAD_ARGUMENT, "lpx_restore_initial_lp: bad order entry");
    const int cur = slot_of[index];                                                  // :220
    if (cur < 0) return set_error(LPX_BAD_ARGUMENT, "lpx_restore_initial_lp: variable missing from perm");
    lpxk::RestoreEntry e{};
    e.k = c0[index];                                                                 // :219
    if (cur >= na) { e.is_basic = 1; e.index = cur - na; }                           // :221-228
    else {
      if (cur >= n) return set_error(LPX_RESTORE_INDEX_FAULT, lpx_status_message(LPX_RESTORE_INDEX_FAULT));  // :231
      e.is_basic = 0; e.index = cur;  // :231 bug-for-bug: an aux-LP slot used as a post-drop index
    }
    ent.push_back(e);
  }
  lpxk::RestoreEntry* d_ent = nullptr;
  lpxk::launch_drop_column(B.A, B.ld, m, na, x0_slot, st);                             // :208-211
  HIP_TRY(hipMemsetAsync(B.c, 0, (size_t)B.ld * sizeof(double), st));
  if (!ent.empty()) {
    HIP_TRY(hipMalloc((void**)&d_ent, ent.size() * sizeof(lpxk::RestoreEntry)));
    hipError_t e = hipMemcpyAsync(d_ent, ent.data(), ent.size() * sizeof(lpxk::RestoreEntry), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) { (void)hipFree(d_ent); return set_error(LPX_DEVICE_ERROR, hipGetErrorString(e)); }
  }
  reset_ctl(s, 0.0);
  if (int rc = push(s)) { (void)hipFree(d_ent); return rc; }
  lpxk::launch_restore_objective(B, n, d_ent, (int)ent.size(), st);                  // :213-233 (sets ctl.v)
  std::vector<int32_t> np;                                                           // :235-244
  np.reserve((size_t)n + m);
  for (int sl = 0; sl < na + m; sl++)
    if (sl != x0_slot) np.push_back(perm[sl]);
  hipError_t e1 = hipMemcpyAsync(B.perm, np.data(), np.size() * sizeof(int32_t), hipMemcpyHostToDevice, st);
  hipError_t e2 = hipStreamSynchronize(st);
  hipError_t e3 = hipGetLastError();
  (void)hipFree(d_ent);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return set_error(LPX_DEVICE_ERROR, "lpx_restore_initial_lp: HIP error");
  set_n(s, n);
  return LPX_OPTIMAL;
}

extern "C" int lpx_solve(int32_t m, int32_t n, const double* A, int64_t lda, const double* b, const double* c,
                         int32_t maximize, const lpx_solve_options* opts, lpx_solve_result* res) {
  if (!res) return set_error(LPX_BAD_ARGUMENT, "lpx_solve: result is NULL");
  memset(res, 0, sizeof *res);
  res->x0_slot = -1;
  res->status = LPX_BAD_ARGUMENT;
  if (m < 0 || n < 0 || (m > 0 && n > 0 && !A) || (m > 0 && !b) || (n > 0 && !c) || (m > 0 && n > 0 && lda < n))
    return set_error(LPX_BAD_ARGUMENT, "lpx_solve: bad argument");
  const double t_start = now_s();
  lpx_solve_options o{};
  if (opts) o = *opts;
  const int64_t max_pivots = opts ? o.max_pivots : -1;
  if (o.keep_state) *o.keep_state = nullptr;

  // :86-89 — the reference negates stForm.c in place for `min`; here on a private copy
  std::vector<double> c0(c, c + n);
  if (!maximize)
    for (auto& x : c0) x = -x;

  const int mib = min_in_b(b, m);                                                   // :118
  const bool phase1 = !(mib == -1 || b[mib] >= 0.0);                                // :119
  const int n_cap = phase1 ? n + 1 : n;
  Cleanup guard;
  lpx_state* s = nullptr;
  if (int rc = alloc(m, n_cap, n_cap, o.device, &s)) { res->status = rc; return rc; }
  guard.s = s;
  if (o.pricing != 0) {
    if (int rc = lpx_state_set_pricing(s, o.pricing)) { res->status = rc; return rc; }
  }
  if (o.fused != 0) {   // 1: fused multiply-add updates, -1: two roundings per update (0: the library's choice by size)
    if (int rc = lpx_state_set_option(s, LPX_OPT_FUSED, o.fused > 0 ? 1 : 0)) { res->status = rc; return rc; }
  }
  hipStream_t st = stream(s);
  lpxk::Buffers& B = buffers(s);
  double t_pivots = 0.0;
  int status = LPX_OPTIMAL;

  // upload A (m x n) into the m x ld device tableau; b; perm
  if (m > 0 && n > 0)
    HIP_TRY(hipMemcpy2DAsync(B.A, B.ld * sizeof(double), A, lda * sizeof(double), (size_t)n * sizeof(double), (size_t)m,
                             hipMemcpyHostToDevice, st));
  if (m > 0) HIP_TRY(hipMemcpyAsync(B.b, b, (size_t)m * sizeof(double), hipMemcpyHostToDevice, st));

  std::vector<int32_t> perm;
  if (!phase1) {
    // convertIntoSlackForm :248-272 — ids: originals 0..n-1, slacks n..n+m-1
    if (n > 0) HIP_TRY(hipMemcpyAsync(B.c, c0.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
    perm.resize((size_t)n + m);
    for (size_t i = 0; i < perm.size(); i++) perm[i] = (int32_t)i;
    if (!perm.empty())
      HIP_TRY(hipMemcpyAsync(B.perm, perm.data(), perm.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    reset_ctl(s, 0.0);
    if (int rc = push(s)) { res->status = rc; return rc; }
    HIP_TRY(hipStreamSynchronize(st));
  } else {
    res->phase1_used = 1;
    // convertIntoAuxLP :283-321 — extra column of -1, objective -x0; x0 has id n+m and starts in slot n
    const int na = n + 1;
    lpxk::launch_fill_column(B.A, B.ld, m, n, -1.0, st);                              // :293
    std::vector<double> auxc(na, 0.0);
    auxc[n] = -1.0;                                                                  // :299-301
    HIP_TRY(hipMemcpyAsync(B.c, auxc.data(), (size_t)na * sizeof(double), hipMemcpyHostToDevice, st));
    perm.resize((size_t)na + m);
    for (int j = 0; j < n; j++) perm[j] = j;
    perm[n] = n + m;
    for (int i = 0; i < m; i++) perm[na + i] = n + i;
    HIP_TRY(hipMemcpyAsync(B.perm, perm.data(), perm.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    reset_ctl(s, 0.0);
    if (int rc = push(s)) { res->status = rc; return rc; }
    HIP_TRY(hipStreamSynchronize(st));

    // solveAuxLP :135-164
    double t0 = now_s();
    int rc = lpx_pivot(s, na - 1, mib);                                              // :138
    if (rc) { res->status = rc; return rc; }
    res->pivots_phase1 = 1;
    int32_t x0 = mib + na;                                                           // :139
    int64_t done = 0;
    int32_t lst = 0;
    const int64_t lim1 = max_pivots < 0 ? -1 : std::max<int64_t>(0, max_pivots - 1);
    rc = lpx_simplex_loop(s, lim1, &done, &lst, &x0);
    t_pivots += now_s() - t0;
    if (rc) { res->status = rc; return rc; }
    res->pivots_phase1 += done;
    res->x0_slot = x0;
    if (lst == LPX_UNBOUNDED) status = LPX_AUX_UNBOUNDED;                            // :147-150
    else if (lst == LPX_PIVOT_LIMIT) status = LPX_PIVOT_LIMIT;
    if (status == LPX_OPTIMAL) {
      // handleInitialization :166-180
      double x0_value = 0.0;
      if (x0 >= na) HIP_TRY(hipMemcpy(&x0_value, B.b + (x0 - na), sizeof(double), hipMemcpyDeviceToHost));
      if (std::fabs(x0_value) > 1e-9) status = LPX_INFEASIBLE;                       // :171-174
    }
    if (status == LPX_OPTIMAL && x0 >= na) {
      // performDegeneratePivot :182-198 — first slot with |A[row][i]| > eps
      const int row = x0 - na;
      std::vector<double> hrow(na);
      HIP_TRY(hipMemcpy(hrow.data(), B.A + (int64_t)row * B.ld, (size_t)na * sizeof(double), hipMemcpyDeviceToHost));
      int entering = -1;
      for (int i = 0; i < na; i++)
        if (std::fabs(hrow[i]) > 1e-9) { entering = i; break; }
      if (entering == -1) status = LPX_NO_DEGENERATE_PIVOT;                          // :192-194
      else {
        t0 = now_s();
        rc = lpx_pivot(s, entering, row);                                            // :195
        t_pivots += now_s() - t0;
        if (rc) { res->status = rc; return rc; }
        res->pivots_phase1 += 1;
        x0 = entering;
        res->x0_slot = x0;
      }
    }
    if (status == LPX_OPTIMAL) {
      // restoreInitialLP :200-246
      status = lpx_restore_initial_lp(s, c0.data(), n, x0, o.restore_order,
                                      o.restore_order ? (o.restore_order_len < 0 ? n : o.restore_order_len) : 0);
      if (status == LPX_DEVICE_ERROR) { res->status = status; return status; }
    }
  }

  if (status == LPX_OPTIMAL) {
    // LPSolver.simplex :96-114
    const double t0 = now_s();
    int64_t done = 0;
    int32_t lst = 0;
    int64_t lim2 = max_pivots < 0 ? -1 : std::max<int64_t>(0, max_pivots - res->pivots_phase1);
    int rc = lpx_simplex_loop(s, lim2, &done, &lst, nullptr);
    t_pivots += now_s() - t0;
    if (rc) { res->status = rc; return rc; }
    res->pivots_phase2 = done;
    status = lst;
  }

  // result
  if (int rc = pull_ctl(s)) { res->status = rc; return rc; }
  double v = host_ctl(s)->v;
  if (!maximize) v = -v;                                                             // :90
  res->objective = v;
  round6_text(v, res->objective_text, sizeof res->objective_text);                   // :113
  res->objective_rounded = strtod(res->objective_text, nullptr);
  res->status = status;
  const int32_t fn = get_n(s);
  if (o.perm_out || o.x_out) {
    std::vector<int32_t> fp((size_t)fn + m);
    if (!fp.empty()) HIP_TRY(hipMemcpy(fp.data(), B.perm, fp.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    // perm_out is int32[n+m]: a solve that ended inside phase 1 still holds the (n+1)-column auxiliary LP, whose
    // permutation has n+m+1 entries and numbers x0 — not the caller's problem any more: leave perm_out untouched
    if (o.perm_out && fn == n) memcpy(o.perm_out, fp.data(), ((size_t)n + m) * sizeof(int32_t));
    if (o.x_out && fn == n) {
      // solution vector (the reference's commented-out printSolution, LPSolver.java:344-374):
      // a basic original variable takes b[row], every nonbasic one is 0
      std::vector<double> hb(m);
      if (m > 0) HIP_TRY(hipMemcpy(hb.data(), B.b, (size_t)m * sizeof(double), hipMemcpyDeviceToHost));
      for (int j = 0; j < n; j++) o.x_out[j] = 0.0;
      for (int i = 0; i < m; i++) {
        const int32_t id = fp[(size_t)n + i];
        if (id >= 0 && id < n) o.x_out[id] = hb[i];
      }
    }
  }
  res->seconds_pivots = t_pivots;
  res->seconds_total = now_s() - t_start;
  if (o.keep_state) { *o.keep_state = s; guard.s = nullptr; }
  if (status != LPX_OPTIMAL) set_error(status, lpx_status_message(status));
  return status;
}

// ------------------------------------------------------------------------------------------------ lpx_solve_multi
// LPSolver.solve with the row blocks of the tableau on several GPUs (include/lpx.h): the same driver logic as
// lpx_solve — min -> max flip (:86-90), initializeSimplex (:116-133), the auxiliary LP (:135-198, :283-321),
// restoreInitialLP (:200-246, bug-for-bug) — over an lpx_multi.  What phase 1 adds on shards: the column of -1 is
// filled per shard, the forced first pivot and the degenerate pivot fetch their pivot row from the shard that owns
// it (lpx_multi_pivot), x0's value / row are read from their owner, and the objective rebuild of restoreInitialLP —
// an ORDERED sum over the rows of the basic original variables, wherever they live — gathers those rows to shard 0,
// runs there and the result is replicated.
#undef HIP_TRY
#include "lpx_internal.h"

namespace {
struct MultiCleanup {
  lpx_multi* M = nullptr;
  ~MultiCleanup() { if (M) lpx_multi_destroy(M); }
};

// upload a replicated vector (c or perm) to every shard
template <typename T>
int replicate(lpx_multi* M, T* lpxk::Buffers::*field, const T* host, size_t count) {
  for (int r = 0; r < multi_shards(M); r++) {
    lpx_state* s = multi_shard(M, r);
    HIP_TRY(hipSetDevice(multi_device(M, r)));
    HIP_TRY(hipMemcpyAsync(s->B.*field, host, count * sizeof(T), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  return 0;
}

int multi_reset_ctl(lpx_multi* M, double v) {
  for (int r = 0; r < multi_shards(M); r++) {
    lpx_state* s = multi_shard(M, r);
    HIP_TRY(hipSetDevice(multi_device(M, r)));
    init_ctl(s, v);
    if (int rc = push_ctl(s)) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  return 0;
}

// restoreInitialLP on shards (see the header of this section)
int multi_restore_initial_lp(lpx_multi* M, const double* c0, int32_t n, int32_t x0_slot, const int32_t* order_in,
                             int32_t order_len) {
  const int G = multi_shards(M);
  lpx_state* s0 = multi_shard(M, 0);
  const int na = state_n(s0), m = s0->m_global;
  if (na != n + 1 || x0_slot < 0 || x0_slot >= na) return fail(LPX_BAD_ARGUMENT, "restoreInitialLP: bad shape/slot");
  std::vector<int32_t> perm((size_t)na + m);
  HIP_TRY(hipSetDevice(multi_device(M, 0)));
  HIP_TRY(hipStreamSynchronize(s0->stream));
  HIP_TRY(hipMemcpy(perm.data(), s0->B.perm, perm.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  std::vector<int32_t> slot_of((size_t)n + m + 1, -1);
  for (int sl = 0; sl < na + m; sl++)
    if (perm[sl] >= 0 && perm[sl] <= n + m) slot_of[perm[sl]] = sl;
  if (order_in && (order_len < 0 || order_len > n)) return fail(LPX_BAD_ARGUMENT, "restoreInitialLP: bad order_len");
  std::vector<int32_t> order(order_in ? (size_t)order_len : (size_t)n);
  if (order_in) order.assign(order_in, order_in + order_len);
  else lpx_java_default_name_order(n, order.data());
  std::vector<lpxk::RestoreEntry> ent;
  std::vector<int32_t> basic_rows;   // global rows gathered to shard 0, in entry order
  for (int32_t index : order) {                                                      // :217
    if (index < 0 || index >= n) return fail(LPX_BAD_ARGUMENT, "restoreInitialLP: bad order entry");
    const int cur = slot_of[index];                                                  // :220
    if (cur < 0) return fail(LPX_BAD_ARGUMENT, "restoreInitialLP: variable missing from perm");
    lpxk::RestoreEntry e{};
    e.k = c0[index];                                                                 // :219
    if (cur >= na) { e.is_basic = 1; e.index = (int32_t)basic_rows.size(); basic_rows.push_back(cur - na); }  // :221-228
    else {
      if (cur >= n) return fail(LPX_RESTORE_INDEX_FAULT, lpx_status_message(LPX_RESTORE_INDEX_FAULT));   // :231
      e.is_basic = 0; e.index = cur;   // :231 bug-for-bug: an aux-LP slot used as a post-drop index
    }
    ent.push_back(e);
  }
  // drop x0's column on every shard (:206-211)
  for (int r = 0; r < G; r++) {
    lpx_state* s = multi_shard(M, r);
    HIP_TRY(hipSetDevice(multi_device(M, r)));
    lpxk::launch_drop_column(s->B.A, s->B.ld, s->m, na, x0_slot, s->stream);
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  // gather the rows (post-drop) and b values of the basic original variables to shard 0, in entry order
  const int64_t ld = s0->B.ld;
  const size_t nb_rows = std::max<size_t>(1, basic_rows.size());
  double *gA = nullptr, *gb = nullptr;
  lpxk::RestoreEntry* d_ent = nullptr;
  struct Temps {   // freed on every way out (on shard 0's device: hipFree does not depend on the current device)
    double *&a, *&b; lpxk::RestoreEntry*& e;
    ~Temps() { (void)hipFree(a); (void)hipFree(b); (void)hipFree(e); a = b = nullptr; e = nullptr; }
  } temps{gA, gb, d_ent};
  HIP_TRY(hipSetDevice(multi_device(M, 0)));
  HIP_TRY(hipMalloc((void**)&gA, nb_rows * (size_t)ld * sizeof(double)));
  HIP_TRY(hipMalloc((void**)&gb, nb_rows * sizeof(double)));
  int rc = 0;
  for (size_t t = 0; t < basic_rows.size() && rc == 0; t++) {
    const int o = multi_owner(M, basic_rows[t]);
    lpx_state* so = multi_shard(M, o);
    const int64_t lr = basic_rows[t] - multi_row_start(M, o);
    if (hipMemcpyPeer(gA + t * ld, multi_device(M, 0), so->B.A + lr * ld, multi_device(M, o), (size_t)ld * sizeof(double)) != hipSuccess ||
        hipMemcpyPeer(gb + t, multi_device(M, 0), so->B.b + lr, multi_device(M, o), sizeof(double)) != hipSuccess)
      rc = fail(LPX_DEVICE_ERROR, "restoreInitialLP: peer copy of a basic row failed");
  }
  // the copies above ran on the devices' null streams, the rebuild runs on shard 0's own (non-blocking) stream
  for (int r = 0; r < G && rc == 0; r++)
    if (hipSetDevice(multi_device(M, r)) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
      rc = fail(LPX_DEVICE_ERROR, "restoreInitialLP: device synchronisation failed");
  if (hipSetDevice(multi_device(M, 0)) != hipSuccess) rc = fail(LPX_DEVICE_ERROR, "hipSetDevice failed");
  std::vector<double> cnew((size_t)ld, 0.0);
  double vnew = 0.0;
  if (rc == 0) {
    hipStream_t st = s0->stream;
    if (!ent.empty()) {
      if (hipMalloc((void**)&d_ent, ent.size() * sizeof(lpxk::RestoreEntry)) != hipSuccess ||
          hipMemcpyAsync(d_ent, ent.data(), ent.size() * sizeof(lpxk::RestoreEntry), hipMemcpyHostToDevice, st) != hipSuccess)
        rc = fail(LPX_DEVICE_ERROR, "restoreInitialLP: entry upload failed");
    }
    if (rc == 0) {
      init_ctl(s0, 0.0);
      rc = push_ctl(s0);
    }
    if (rc == 0) {
      lpxk::Buffers Bg = s0->B;   // the gathered rows stand in for the tableau: entry t names row t
      Bg.A = gA;
      Bg.b = gb;
      (void)hipMemsetAsync(s0->B.c, 0, (size_t)ld * sizeof(double), st);
      lpxk::launch_restore_objective(Bg, n, d_ent, (int)ent.size(), st);              // :213-233 (sets ctl.v)
      if (hipMemcpyAsync(cnew.data(), s0->B.c, (size_t)ld * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
          sync_ctl_to_host(s0) != 0 || hipGetLastError() != hipSuccess)
        rc = fail(LPX_DEVICE_ERROR, "restoreInitialLP: objective rebuild failed");
      else vnew = s0->h_ctl->v;
    }
  }
  if (rc) return rc;
  std::vector<int32_t> np;                                                           // :235-244
  np.reserve((size_t)n + m);
  for (int sl = 0; sl < na + m; sl++)
    if (sl != x0_slot) np.push_back(perm[sl]);
  if (int r2 = replicate<double>(M, &lpxk::Buffers::c, cnew.data(), (size_t)ld)) return r2;
  if (int r2 = replicate<int32_t>(M, &lpxk::Buffers::perm, np.data(), np.size())) return r2;
  if (int r2 = multi_reset_ctl(M, vnew)) return r2;
  for (int r = 0; r < G; r++) multi_shard(M, r)->n = n;
  multi_set_n(M, n);
  return LPX_OPTIMAL;
}
}  // namespace

extern "C" int lpx_solve_multi(int32_t m, int32_t n, const double* A, int64_t lda, const double* b, const double* c,
                               int32_t maximize, const lpx_solve_options* opts, const int32_t* devices, int32_t n_dev,
                               lpx_solve_result* res) {
  if (!res) return fail(LPX_BAD_ARGUMENT, "lpx_solve_multi: result is NULL");
  DeviceRestore keep_device;
  memset(res, 0, sizeof *res);
  res->x0_slot = -1;
  res->status = LPX_BAD_ARGUMENT;
  if (m < 0 || n < 0 || (m > 0 && n > 0 && !A) || (m > 0 && !b) || (n > 0 && !c) || (m > 0 && n > 0 && lda < n))
    return fail(LPX_BAD_ARGUMENT, "lpx_solve_multi: bad argument");
  const double t_start = now_s();
  lpx_solve_options o{};
  if (opts) o = *opts;
  const int64_t max_pivots = opts ? o.max_pivots : -1;
  if (o.keep_state) return fail(LPX_BAD_ARGUMENT, "lpx_solve_multi: keep_state is not supported");

  std::vector<double> c0(c, c + n);                                                  // :86-89 on a private copy
  if (!maximize)
    for (auto& x : c0) x = -x;
  const int mib = min_in_b(b, m);                                                    // :118
  const bool phase1 = !(mib == -1 || b[mib] >= 0.0);                                 // :119
  const int n_cap = phase1 ? n + 1 : n;
  const int na = n + 1;
  MultiCleanup guard;
  lpx_multi* M = nullptr;
  std::vector<int32_t> perm((size_t)n_cap + m);
  std::vector<double> cinit((size_t)n_cap, 0.0);
  if (!phase1) {                                                                     // convertIntoSlackForm :248-272
    for (size_t i = 0; i < perm.size(); i++) perm[i] = (int32_t)i;
    std::copy(c0.begin(), c0.end(), cinit.begin());
  } else {                                                                           // convertIntoAuxLP :283-321
    for (int j = 0; j < n; j++) perm[j] = j;
    perm[n] = n + m;
    for (int i = 0; i < m; i++) perm[na + i] = n + i;
    cinit[n] = -1.0;                                                                 // :299-301
  }
  // the shards are created m x n_cap; in phase 1 column n is then filled with -1 (:293)
  {
    std::vector<double> Ax;
    const double* Asrc = A;
    int64_t ldsrc = lda;
    if (phase1) {   // one extra column: widen on the host (an O(m n) copy next to the PCIe upload)
      Ax.assign((size_t)m * n_cap, -1.0);
      for (int i = 0; i < m; i++) std::copy(A + (int64_t)i * lda, A + (int64_t)i * lda + n, Ax.begin() + (size_t)i * n_cap);
      Asrc = Ax.data();
      ldsrc = n_cap;
    }
    if (int rc = multi_create(m, n_cap, n_cap, Asrc, ldsrc, b, cinit.data(), 0.0, perm.data(), devices, n_dev, &M)) {
      res->status = rc;
      return rc;
    }
  }
  guard.M = M;
  if (o.pricing != 0) {
    if (int rc = lpx_multi_set_pricing(M, o.pricing)) { res->status = rc; return rc; }
  }
  if (o.fused != 0) {
    if (int rc = lpx_multi_set_option(M, LPX_OPT_FUSED, o.fused > 0 ? 1 : 0)) { res->status = rc; return rc; }
  }
  double t_pivots = 0.0;
  int status = LPX_OPTIMAL;
  if (phase1) {
    res->phase1_used = 1;
    double t0 = now_s();
    int rc = lpx_multi_pivot(M, na - 1, mib);                                        // solveAuxLP :138
    if (rc) { res->status = rc; return rc; }
    res->pivots_phase1 = 1;
    int32_t x0 = mib + na;                                                           // :139
    int64_t done = 0;
    int32_t lst = 0;
    const int64_t lim1 = max_pivots < 0 ? -1 : std::max<int64_t>(0, max_pivots - 1);
    rc = lpx_multi_simplex_loop(M, lim1, &done, &lst, &x0);
    t_pivots += now_s() - t0;
    if (rc) { res->status = rc; return rc; }
    res->pivots_phase1 += done;
    res->x0_slot = x0;
    if (lst == LPX_UNBOUNDED) status = LPX_AUX_UNBOUNDED;                            // :147-150
    else if (lst == LPX_PIVOT_LIMIT) status = LPX_PIVOT_LIMIT;
    if (status == LPX_OPTIMAL && x0 >= na) {                                         // handleInitialization :166-180
      const int o_sh = multi_owner(M, x0 - na);
      lpx_state* so = multi_shard(M, o_sh);
      const int64_t lr = (x0 - na) - multi_row_start(M, o_sh);
      HIP_TRY(hipSetDevice(multi_device(M, o_sh)));
      double x0_value = 0.0;
      HIP_TRY(hipMemcpy(&x0_value, so->B.b + lr, sizeof(double), hipMemcpyDeviceToHost));
      if (std::fabs(x0_value) > 1e-9) status = LPX_INFEASIBLE;                       // :171-174
      if (status == LPX_OPTIMAL) {                                                   // performDegeneratePivot :182-198
        std::vector<double> hrow(na);
        HIP_TRY(hipMemcpy(hrow.data(), so->B.A + lr * so->B.ld, (size_t)na * sizeof(double), hipMemcpyDeviceToHost));
        int entering = -1;
        for (int i = 0; i < na; i++)
          if (std::fabs(hrow[i]) > 1e-9) { entering = i; break; }
        if (entering == -1) status = LPX_NO_DEGENERATE_PIVOT;                        // :192-194
        else {
          t0 = now_s();
          rc = lpx_multi_pivot(M, entering, x0 - na);                                // :195
          t_pivots += now_s() - t0;
          if (rc) { res->status = rc; return rc; }
          res->pivots_phase1 += 1;
          x0 = entering;
          res->x0_slot = x0;
        }
      }
    }
    if (status == LPX_OPTIMAL) {                                                     // restoreInitialLP :200-246
      status = multi_restore_initial_lp(M, c0.data(), n, x0, o.restore_order,
                                        o.restore_order ? (o.restore_order_len < 0 ? n : o.restore_order_len) : 0);
      if (status == LPX_DEVICE_ERROR || status == LPX_BAD_ARGUMENT) { res->status = status; return status; }
    }
  }
  if (status == LPX_OPTIMAL) {                                                       // LPSolver.simplex :96-114
    const double t0 = now_s();
    int64_t done = 0;
    int32_t lst = 0;
    const int64_t lim2 = max_pivots < 0 ? -1 : std::max<int64_t>(0, max_pivots - res->pivots_phase1);
    const int rc = lpx_multi_simplex_loop(M, lim2, &done, &lst, nullptr);
    t_pivots += now_s() - t0;
    if (rc) { res->status = rc; return rc; }
    res->pivots_phase2 = done;
    status = lst;
  }
  double v = 0.0;
  const int32_t fn = state_n(multi_shard(M, 0));
  std::vector<int32_t> fp((size_t)fn + m);
  std::vector<double> hb((size_t)std::max(m, 1));
  if (int rc = lpx_multi_read(M, nullptr, 0, m > 0 ? hb.data() : nullptr, nullptr, &v, fp.data())) { res->status = rc; return rc; }
  if (!maximize) v = -v;                                                             // :90
  res->objective = v;
  round6_text(v, res->objective_text, sizeof res->objective_text);                   // :113
  res->objective_rounded = strtod(res->objective_text, nullptr);
  res->status = status;
  if (fn == n) {
    if (o.perm_out) memcpy(o.perm_out, fp.data(), ((size_t)n + m) * sizeof(int32_t));
    if (o.x_out) {
      for (int j = 0; j < n; j++) o.x_out[j] = 0.0;
      for (int i = 0; i < m; i++) {
        const int32_t id = fp[(size_t)n + i];
        if (id >= 0 && id < n) o.x_out[id] = hb[i];
      }
    }
  }
  res->seconds_pivots = t_pivots;
  res->seconds_total = now_s() - t_start;
  if (status != LPX_OPTIMAL) fail(status, lpx_status_message(status));
  return status;
}
