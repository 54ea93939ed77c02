// ORACLE — TEST INFRASTRUCTURE ONLY (see dec15.hpp / lp_oracle.hpp).  extern "C" surface of the CPU
// restatement, loaded with ctypes by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
// kind 0 = decimal-15 (reference BigDecimal semantics), kind 1 = IEEE fp64 (what the GPU computes), kind 2 = IEEE
// fp64 with fused multiply-add updates (what the GPU computes in its opt-in fused-arithmetic mode, LPX_OPT_FUSED).
// Inputs arrive as doubles; the decimal instantiation recovers the <=15-digit decimal each double was
// written from (%.17g then HALF_UP to 15 digits), so fixtures written with short decimals are exact.
#include <chrono>
#include <cstring>
#include <memory>
#include <string>

#include "lp_oracle.hpp"

using namespace lporacle;
using dec15::Dec;

namespace {

struct Handle {
  int kind;
  State<Dec> sd;
  State<double> sf;
  State<F64Fused> sx;
};

// run `fn` on the state of the handle's kind
template <class F> auto with_state(Handle* h, F&& fn) {
  if (h->kind == 0) return fn(h->sd);
  if (h->kind == 2) return fn(h->sx);
  return fn(h->sf);
}

template <class T>
void fill_state(State<T>& st, int m, int n, const double* A, const double* b, const double* c, double v,
                const int32_t* perm, int with_perm) {
  typedef Num<T> N;
  st.m = m; st.n = n;
  st.A.resize((size_t)m * n);
  const int64_t total = (int64_t)m * n;
#pragma omp parallel for schedule(static) if (total > 100000)
  for (int64_t i = 0; i < total; i++) st.A[i] = N::from_double(A[i]);
  st.b.resize(m);
  for (int i = 0; i < m; i++) st.b[i] = N::from_double(b[i]);
  st.c.resize(n);
  for (int i = 0; i < n; i++) st.c[i] = N::from_double(c[i]);
  st.v = N::from_double(v);
  st.perm.clear();
  if (with_perm) {
    st.perm.resize(n + m);
    for (int i = 0; i < n + m; i++) st.perm[i] = perm ? perm[i] : i;
  }
}

template <class T>
void read_state(const State<T>& st, double* A, double* b, double* c, double* v, int32_t* perm) {
  typedef Num<T> N;
  if (A) for (size_t i = 0; i < st.A.size(); i++) A[i] = N::to_double(st.A[i]);
  if (b) for (size_t i = 0; i < st.b.size(); i++) b[i] = N::to_double(st.b[i]);
  if (c) for (size_t i = 0; i < st.c.size(); i++) c[i] = N::to_double(st.c[i]);
  if (v) *v = N::to_double(st.v);
  if (perm) for (size_t i = 0; i < st.perm.size(); i++) perm[i] = st.perm[i];
}

// canonical text dump:  "A a00 a01 ...\nb ...\nc ...\nv ...\n"
template <class T> std::string dump_state(const State<T>& st) {
  typedef Num<T> N;
  std::string s = "A";
  for (auto& x : st.A) { s += ' '; s += N::str(x); }
  s += "\nb";
  for (auto& x : st.b) { s += ' '; s += N::str(x); }
  s += "\nc";
  for (auto& x : st.c) { s += ' '; s += N::str(x); }
  s += "\nv ";
  s += N::str(st.v);
  s += "\n";
  return s;
}

std::string round6_double(double v) {
  // new BigDecimal(double).setScale(6, HALF_UP): exact binary value, first discarded digit decides.
  static char buf[1400];
  snprintf(buf, sizeof buf, "%.1100f", v);  // exact expansion (a double has at most 1074 fraction digits)
  std::string s(buf);
  size_t dot = s.find('.');
  bool neg = s[0] == '-';
  std::string ip = s.substr(neg ? 1 : 0, dot - (neg ? 1 : 0));
  std::string fp = s.substr(dot + 1);
  std::string digits = ip + fp.substr(0, 6);
  bool up = fp[6] >= '5';
  if (up) {
    int i = (int)digits.size() - 1;
    while (i >= 0) {
      if (digits[i] == '9') { digits[i] = '0'; i--; }
      else { digits[i]++; break; }
    }
    if (i < 0) digits.insert(digits.begin(), '1');
  }
  std::string out = digits.substr(0, digits.size() - 6) + "." + digits.substr(digits.size() - 6);
  bool all_zero = true;
  for (char ch : digits) if (ch != '0') all_zero = false;
  if (neg && !all_zero) out = "-" + out;
  return out;
}

}  // namespace

extern "C" {

struct orc_result {
  int32_t status;
  int32_t phase1_used;
  double objective;           // unrounded v, sign-corrected for min
  char objective_repr[64];    // canonical text of the unrounded, sign-corrected v (dec: "ce" form, fp64: %a)
  char objective_text[64];    // setScale(6, HALF_UP), sign-corrected
  int64_t pivots1, pivots2;
  int32_t x0_slot;
  int32_t final_m, final_n;
  int32_t trace_len;          // number of pivot records available
  double seconds;
};

// ---- scalar ops for pinning dec15 against Python's decimal -------------------------------------------
// op: 0 add, 1 sub, 2 mul, 3 div, 4 cmp (result "-1|0|1"), 5 parse/normalise, 6 setScale(6,HALF_UP)
int orc_dec_op(int op, const char* a, const char* b, char* out, size_t cap) {
  try {
    Dec x = dec15::from_string(a);
    Dec y = (b && *b) ? dec15::from_string(b) : Dec(0, 0);
    std::string r;
    switch (op) {
      case 0: r = dec15::to_string(dec15::add(x, y)); break;
      case 1: r = dec15::to_string(dec15::sub(x, y)); break;
      case 2: r = dec15::to_string(dec15::mul(x, y)); break;
      case 3: r = dec15::to_string(dec15::div(x, y)); break;
      case 4: r = std::to_string(dec15::cmp(x, y)); break;
      case 5: r = dec15::to_string(x); break;
      case 6: r = dec15::set_scale6(x); break;
      default: return -1;
    }
    if (r.size() + 1 > cap) return -2;
    memcpy(out, r.c_str(), r.size() + 1);
    return 0;
  } catch (const std::domain_error&) {
    return 1;  // division by zero
  } catch (...) {
    return -3;
  }
}

int orc_round6_double(double v, char* out, size_t cap) {
  std::string r = round6_double(v);
  if (r.size() + 1 > cap) return -2;
  memcpy(out, r.c_str(), r.size() + 1);
  return 0;
}

// ---- LPState ---------------------------------------------------------------------------------------------
void* orc_state_new(int kind, int m, int n, const double* A, const double* b, const double* c, double v,
                    const int32_t* perm, int with_perm) {
  Handle* h = new Handle();
  h->kind = kind;
  with_state(h, [&](auto& st) { fill_state(st, m, n, A, b, c, v, perm, with_perm); return 0; });
  return h;
}

void orc_state_free(void* p) { delete (Handle*)p; }

// 0 = reference rule (first positive), 1 = Dantzig (opt-in extension, see lp_oracle.hpp)
void orc_state_set_pricing(void* p, int pricing) {
  Handle* h = (Handle*)p;
  h->sd.pricing = pricing;
  h->sf.pricing = pricing;
  h->sx.pricing = pricing;
}

int orc_get_entering(void* p) {
  Handle* h = (Handle*)p;
  return with_state(h, [](auto& st) { return st.get_entering(); });
}

// returns leaving row, -1 none, -2 IllegalArgumentException
int orc_get_leaving(void* p, int entering) {
  Handle* h = (Handle*)p;
  try {
    return with_state(h, [&](auto& st) { return st.get_leaving(entering); });
  } catch (const std::domain_error&) {
    return -3;
  }
}

// threads == 1: pivotSequentially; threads > 1: pivotConcurrently's partitioning
int orc_pivot(void* p, int entering, int leaving, int threads) {
  Handle* h = (Handle*)p;
  const int m = with_state(h, [](auto& st) { return st.m; }), n = with_state(h, [](auto& st) { return st.n; });
  if (entering < 0 || entering >= n || leaving < 0 || leaving >= m) return LPX_BAD_ARGUMENT;
  try {
    with_state(h, [&](auto& st) { st.pivot(entering, leaving, threads); return 0; });
  } catch (const DivideByZero&) {
    return LPX_DIVIDE_BY_ZERO;
  } catch (const std::domain_error&) {
    return LPX_DIVIDE_BY_ZERO;
  }
  return 0;
}

void orc_state_dims(void* p, int32_t* m, int32_t* n, int32_t* has_perm) {
  Handle* h = (Handle*)p;
  with_state(h, [&](auto& st) { *m = st.m; *n = st.n; *has_perm = !st.perm.empty(); return 0; });
}

void orc_state_read(void* p, double* A, double* b, double* c, double* v, int32_t* perm) {
  Handle* h = (Handle*)p;
  with_state(h, [&](auto& st) { read_state(st, A, b, c, v, perm); return 0; });
}

// canonical text of every entry; returns needed size (incl. NUL) if cap too small
int64_t orc_state_dump(void* p, char* out, int64_t cap) {
  Handle* h = (Handle*)p;
  std::string s = with_state(h, [](auto& st) { return dump_state(st); });
  if ((int64_t)s.size() + 1 > cap) return (int64_t)s.size() + 1;
  memcpy(out, s.c_str(), s.size() + 1);
  return 0;
}

// Runs LPSolver.simplex's loop on an existing state for at most max_pivots pivots.
// status: LPX_OPTIMAL / LPX_UNBOUNDED / LPX_PIVOT_LIMIT.  Returns seconds spent.
double orc_simplex_loop(void* p, int64_t max_pivots, int threads, int64_t* pivots_done, int32_t* status,
                        int32_t* trace /* 2 ints per pivot, may be NULL */, int64_t trace_cap) {
  Handle* h = (Handle*)p;
  std::vector<PivotRecord> tr;
  int64_t piv = 0;
  auto t0 = std::chrono::steady_clock::now();
  const int st = with_state(h, [&](auto& state) {
    return simplex_loop(state, 2, max_pivots, piv, (int*)nullptr, trace ? &tr : nullptr, threads, LPX_UNBOUNDED);
  });
  auto t1 = std::chrono::steady_clock::now();
  *pivots_done = piv;
  *status = st;
  if (trace)
    for (int64_t i = 0; i < (int64_t)tr.size() && i < trace_cap; i++) {
      trace[2 * i] = tr[i].entering;
      trace[2 * i + 1] = tr[i].leaving;
    }
  return std::chrono::duration<double>(t1 - t0).count();
}

// ---- LPSolver.solve ----------------------------------------------------------------------------------------
}  // extern "C" (reopened below; the template cannot have C linkage)

namespace {

template <class T> State<T>& handle_state(Handle* h);
template <> State<Dec>& handle_state<Dec>(Handle* h) { return h->sd; }
template <> State<double>& handle_state<double>(Handle* h) { return h->sf; }
template <> State<F64Fused>& handle_state<F64Fused>(Handle* h) { return h->sx; }

template <class T> std::string objective_text(const T& v);
template <> std::string objective_text<Dec>(const Dec& v) { return dec15::set_scale6(v); }
template <> std::string objective_text<double>(const double& v) { return round6_double(v); }
template <> std::string objective_text<F64Fused>(const F64Fused& v) { return round6_double(v.x); }

template <class T>
Handle* solve_impl(int kind, int m, int n, const double* A, const double* b, const double* c, int maximize,
                   const int32_t* restore_order, int64_t max_pivots, int threads, orc_result* res,
                   int32_t* trace_out, int64_t trace_cap, int pricing) {
  typedef Num<T> N;
  std::vector<T> Av((size_t)m * n), bv(m), cv(n);
  for (size_t i = 0; i < Av.size(); i++) Av[i] = N::from_double(A[i]);
  for (int i = 0; i < m; i++) bv[i] = N::from_double(b[i]);
  for (int i = 0; i < n; i++) cv[i] = N::from_double(c[i]);
  std::vector<int32_t> order;
  if (restore_order) order.assign(restore_order, restore_order + n);
  SolveOut<T> out;
  out.v = N::zero();
  auto t0 = std::chrono::steady_clock::now();
  int status;
  try {
    solve<T>(m, n, Av, bv, cv, maximize != 0, order, max_pivots, threads, trace_out != nullptr, out, pricing);
    status = out.status;
  } catch (const DivideByZero&) {
    status = LPX_DIVIDE_BY_ZERO;
  } catch (const std::domain_error&) {
    status = LPX_DIVIDE_BY_ZERO;
  }
  auto t1 = std::chrono::steady_clock::now();
  memset(res, 0, sizeof *res);
  res->status = status;
  res->phase1_used = out.phase1_used;
  // LPSolver.java:113 rounds first, :90 negates the rounded value; HALF_UP is symmetric so the order is moot.
  T v = out.negate_result ? N::neg(out.v) : out.v;
  res->objective = N::to_double(v);
  snprintf(res->objective_repr, sizeof res->objective_repr, "%s", N::str(v).c_str());
  snprintf(res->objective_text, sizeof res->objective_text, "%s", objective_text<T>(v).c_str());
  res->pivots1 = out.pivots1;
  res->pivots2 = out.pivots2;
  res->x0_slot = out.x0_slot;
  res->final_m = out.final_state.m;
  res->final_n = out.final_state.n;
  res->trace_len = (int32_t)out.trace.size();
  res->seconds = std::chrono::duration<double>(t1 - t0).count();
  if (trace_out)
    for (int64_t i = 0; i < (int64_t)out.trace.size() && i < trace_cap; i++) {
      trace_out[3 * i] = out.trace[i].phase;
      trace_out[3 * i + 1] = out.trace[i].entering;
      trace_out[3 * i + 2] = out.trace[i].leaving;
    }
  Handle* h = new Handle();
  h->kind = kind;
  handle_state<T>(h) = std::move(out.final_state);
  return h;
}

}  // namespace

extern "C" {

// Returns a state handle holding the final LPState (or the aux LPState if phase 1 failed); the caller
// frees it with orc_state_free.  trace_out: 3 ints per pivot (phase, entering, leaving), may be NULL.
void* orc_solve2(int kind, int m, int n, const double* A, const double* b, const double* c, int maximize,
                 const int32_t* restore_order, int64_t max_pivots, int threads, orc_result* res,
                 int32_t* trace_out, int64_t trace_cap, int pricing) {
  if (kind == 0)
    return solve_impl<Dec>(kind, m, n, A, b, c, maximize, restore_order, max_pivots, threads, res, trace_out, trace_cap, pricing);
  if (kind == 2)
    return solve_impl<F64Fused>(kind, m, n, A, b, c, maximize, restore_order, max_pivots, threads, res, trace_out, trace_cap, pricing);
  return solve_impl<double>(kind, m, n, A, b, c, maximize, restore_order, max_pivots, threads, res, trace_out, trace_cap, pricing);
}

void* orc_solve(int kind, int m, int n, const double* A, const double* b, const double* c, int maximize,
                const int32_t* restore_order, int64_t max_pivots, int threads, orc_result* res,
                int32_t* trace_out, int64_t trace_cap) {
  return orc_solve2(kind, m, n, A, b, c, maximize, restore_order, max_pivots, threads, res, trace_out, trace_cap, 0);
}

// solveAuxLP(auxLP, indexOfx0, minInB) on an existing aux state (LPSolver.java:135): returns x0CurrentIndex,
// or -1000 - status when the aux LP is unbounded.
int orc_solve_aux_lp(void* p, int index_of_x0, int mib) {
  Handle* h = (Handle*)p;
  int64_t piv = 0;
  int x0 = -1;
  const int st = with_state(h, [&](auto& state) { return solve_aux_lp(state, index_of_x0, mib, -1, piv, x0, nullptr, 1); });
  return st == LPX_OPTIMAL ? x0 : -1000 - st;
}

// convertIntoAuxLP (LPSolver.java:283): new aux-state handle from standard-form data
void* orc_convert_into_aux_lp(int kind, int m, int n, const double* A, const double* b) {
  Handle* h = new Handle();
  h->kind = kind;
  with_state(h, [&](auto& st) {
    typedef typename std::decay<decltype(st)>::type::N N;
    std::vector<typename N::T> Av((size_t)m * n), bv(m);
    for (size_t i = 0; i < Av.size(); i++) Av[i] = N::from_double(A[i]);
    for (int i = 0; i < m; i++) bv[i] = N::from_double(b[i]);
    convert_into_aux_lp(m, n, Av, bv, st);
    return 0;
  });
  return h;
}

// restoreInitialLP(auxLP, initial, indexOfX0) (LPSolver.java:200): returns a new state handle or NULL with
// *status = LPX_RESTORE_INDEX_FAULT.  c0 = initial.c, order = keySet() iteration order (n entries).
void* orc_restore_initial_lp(void* p, const double* c0, int n, int x0, const int32_t* order, int32_t* status) {
  Handle* a = (Handle*)p;
  Handle* h = new Handle();
  h->kind = a->kind;
  std::vector<int32_t> ord(order, order + n);
  const int st = with_state(a, [&](auto& aux) {
    typedef typename std::decay<decltype(aux)>::type::N N;
    std::vector<typename N::T> cv(n);
    for (int i = 0; i < n; i++) cv[i] = N::from_double(c0[i]);
    return restore_initial_lp(aux, cv, n, x0, ord, handle_state<typename N::T>(h));
  });
  *status = st;
  if (st != LPX_OPTIMAL) { delete h; return nullptr; }
  return h;
}

int orc_min_in_b(int kind, int m, const double* b) {
  if (kind == 0) {
    std::vector<Dec> bv(m);
    for (int i = 0; i < m; i++) bv[i] = dec15::from_double(b[i]);
    return min_in_b(bv);
  }
  std::vector<double> bv(b, b + m);
  return min_in_b(bv);
}

int orc_java_default_name_order(int n, int32_t* out) {
  std::vector<int32_t> o = java_default_name_order(n);
  for (int i = 0; i < n; i++) out[i] = o[i];
  return 0;
}

}  // extern "C"
