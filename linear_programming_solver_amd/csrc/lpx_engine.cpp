// Host engine behind include/lpx.h: owns the device-resident LPState, issues the pivot kernels on one HIP
// stream and polls the device loop state.  No decision about a pivot is ever taken on the host inside the
// loop: the host enqueues (select_pivot, update) pairs ahead of the GPU and reads LpxCtl::status from
// pinned memory once per batch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lpx.h"
#include "lpx_kernels.h"
#include "lpx_internal.h"

using lpxk::Buffers;
using lpxk::Geometry;
using lpxk::LpxCtl;
using lpxk::RatioRow;

// ------------------------------------------------------------------------------------------------ errors
thread_local std::string g_last_error;

int fail(int status, const std::string& msg) {
  g_last_error = msg;
  return status;
}

extern "C" const char* lpx_status_message(int status) {
  switch (status) {
    case LPX_UNBOUNDED: return "This linear program is unbounded";                 // LPSolver.java:105
    case LPX_INFEASIBLE: return "This linear program is infeasible";               // LPSolver.java:173
    case LPX_AUX_UNBOUNDED: return "Auxiliary lp is unbounded";                    // LPSolver.java:149
    case LPX_NO_DEGENERATE_PIVOT: return "Can't perform degenerate pivot";         // LPSolver.java:193
    case LPX_BAD_ARGUMENT: return "IllegalArgumentException";                      // LPState.java:288
    case LPX_RESTORE_INDEX_FAULT: return "ArrayIndexOutOfBoundsException";         // LPSolver.java:231
    case LPX_DEVICE_ERROR: return "HIP device error";
    case LPX_DIVIDE_BY_ZERO: return "Division by zero";                            // LPState.java:139
    default: return "";
  }
}

extern "C" const char* lpx_last_error(void) { return g_last_error.c_str(); }
extern "C" int lpx_abi_version(void) { return LPX_ABI_VERSION; }

extern "C" int lpx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return -1;
  return n;
}

static int64_t round_up(int64_t x, int64_t q) { return (x + q - 1) / q * q; }

// Option defaults.  The LPX_<NAME> environment variables are read exactly once per process (debugging aid for the
// scripts/ helpers, which run one configuration per process); everything after that goes through the handle
// (lpx_state_set_option): no getenv in the pivot path.
struct OptionSpec { const char* env; int64_t dflt, lo, hi; };
static const OptionSpec kOptionSpec[LPX_OPT_COUNT] = {
    {"LPX_BLOCK", 0, 0, lpxk::kBlockMax},   // LPX_OPT_BLOCK
    {"LPX_CHAIN", 1, 0, 1},                 // LPX_OPT_CHAIN
    {"LPX_OVERLAP", 1, 0, 1},               // LPX_OPT_OVERLAP
    {"LPX_OVERLAP_SERIAL", 0, 0, 1},        // LPX_OPT_OVERLAP_SERIAL
    {"LPX_OVERLAP_MASK", 1, 0, 1},          // LPX_OPT_OVERLAP_MASK
    {"LPX_CHAIN_WGS", 0, 0, lpxk::kChainMaxWgs},  // LPX_OPT_CHAIN_WGS
    {"LPX_CHAIN_FENCES", 2, 0, 3},          // LPX_OPT_CHAIN_FENCES
    {"LPX_SWEEP_ROWS", 0, 0, 8192},         // LPX_OPT_SWEEP_ROWS
    {"LPX_NT", -1, -1, 1},                  // LPX_OPT_NT
    {"LPX_BATCH", 0, 0, 4096},              // LPX_OPT_BATCH
    {"LPX_CHAIN_TRACE", 0, 0, 1},           // LPX_OPT_CHAIN_TRACE
    {"LPX_U", 1, 1, 4},                     // LPX_OPT_UPDATE_U
    {"LPX_ROWS_PER_TILE", 2, 2, 256},       // LPX_OPT_UPDATE_ROWS
    {"LPX_A2_OFFSET", 512, 0, 1 << 20},     // LPX_OPT_A2_OFFSET
    {"LPX_SWEEP_FORM", 0, 0, 4},            // LPX_OPT_SWEEP_FORM
    {"LPX_MULTI_ONEHOP", 0, 0, 1},          // LPX_OPT_MULTI_ONEHOP
    {"LPX_SWEEP_CUS", 0, 0, 256},           // LPX_OPT_SWEEP_CUS
    {"LPX_CHAIN_CUS", 0, 0, 16},            // LPX_OPT_CHAIN_CUS
    {"LPX_FUSED", 2, 0, 2},                 // LPX_OPT_FUSED
    {"LPX_CHAIN_FORM", 1, 0, 1},            // LPX_OPT_CHAIN_FORM
    {"LPX_FIXUP_SIDE", 4, 0, 4},            // LPX_OPT_FIXUP_SIDE
};

static const int64_t* env_defaults() {
  static int64_t d[LPX_OPT_COUNT];
  static const bool once = [] {
    for (int k = 0; k < LPX_OPT_COUNT; k++) {
      const OptionSpec& sp = kOptionSpec[k];
      int64_t v = sp.dflt;
      const char* e = getenv(sp.env);
      if (e && *e) v = std::max(sp.lo, std::min(sp.hi, (int64_t)atoll(e)));
      d[k] = v;
    }
    return true;
  }();
  (void)once;
  return d;
}

// LPX_OPT_FUSED resolved.  By size (2): the fused multiply-add pays where the sweep is the bound — measured, same box,
// pivots/s fused / plain: 512 MiB 81.2k / 76.9k (+5.6 %), 768 MiB 69.4k / 63.4k, cfg3 (1 GiB) 60.6k / 51.5k, cfg4 29.9k / 16.8k
// (profiles/r05_by_size_arith_grid.txt; 256 MiB and below: +2 % or less, profiles/r04_block_by_size_small.txt) — and it is no less faithful to the decimal
// reference than the two-rounding form (tests/golden/divergence_census.json: both leave the decimal-15 pivot sequence on
// the same 116 of 408 LPs, the 6-decimal result never differs).  Shards keep the plain arithmetic unless told.
void resolve_arithmetic(lpx_state* s) {
  const int64_t v = s->opt[LPX_OPT_FUSED];
  const double bytes = 8.0 * (double)s->m * (double)s->B.ld;
  const bool by_size = s->m == s->m_global && !s->multi_shard && bytes >= 0.5 * 1073741824.0;
  s->B.fused = v == 1 || (v == 2 && by_size);
}

// Tiling of k_update.
static Geometry choose_geometry(int m, int64_t ld, const int64_t* opt) {
  // Measured on MI355X (profiles/r01_sweep_*.log): one row PAIR x 512 columns per workgroup is fastest at
  // every size (cfg3 6.3 TB/s vs 5.3 TB/s for 32-row x 2048-column tiles): consecutive workgroups then sweep
  // the tableau in address order and the tail of the grid is negligible.
  Geometry g{};
  int U = (int)opt[LPX_OPT_UPDATE_U];
  if (U != 1 && U != 2 && U != 4) U = 1;
  g.U = U;
  const int W = 512 * U;
  g.nstrips = (int)((ld + W - 1) / W);
  int R = (int)opt[LPX_OPT_UPDATE_ROWS];
  if (R < 2) R = 2;
  if (R > 256) R = 256;  // the seed/peek kernels emit one partial per 256 rows into the same buffer
  if (R & 1) R += 1;  // rows are processed in pairs
  g.rows_per_tile = R;
  g.ntiles = m > 0 ? (m + R - 1) / R : 0;
  return g;
}

// The by-size choices that depend on options: k_update's tiling and the cache policy of the tableau accesses.
static void apply_layout_options(lpx_state* s) {
  s->g = choose_geometry(s->m, s->B.ld, s->opt);
  // non-temporal streaming only pays once the tableau no longer fits the 256 MiB Infinity Cache
  const int64_t bytes = (int64_t)s->m * s->B.ld * 8;
  const int64_t nt = s->opt[LPX_OPT_NT];
  s->nontemporal = nt < 0 ? bytes > (192ll << 20) : nt != 0;
  s->info.nontemporal = s->nontemporal ? 1 : 0;
}

// The loop state travels to the host mirror behind everything enqueued on the handle's stream so far; no wait.
static int enqueue_ctl_fetch(lpx_state* s) {
  HIP_TRY(hipMemcpyAsync(s->h_ctl, s->B.ctl, sizeof(LpxCtl), hipMemcpyDeviceToHost, s->stream));
  // a bounded wait inside a sweep kernel that ran out sets a device word: it travels with the loop state (no extra sync)
  unsigned* const fw = lpxk::sweep_fail_word(s->R, s->B.ld);
  unsigned* const h_fw = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(s->h_ctl) + sizeof(LpxCtl));
  if (fw) HIP_TRY(hipMemcpyAsync(h_fw, fw, sizeof(unsigned), hipMemcpyDeviceToHost, s->stream));
  return 0;
}
// ... and once the stream has been waited for: did a sweep kernel report a wait that ran out?
static int check_fetched_fail_word(lpx_state* s) {
  unsigned* const fw = lpxk::sweep_fail_word(s->R, s->B.ld);
  unsigned* const h_fw = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(s->h_ctl) + sizeof(LpxCtl));
  if (fw && *h_fw) {
    const unsigned f = *h_fw;
    *h_fw = 0;
    (void)hipMemset(fw, 0, sizeof(unsigned));
    return fail(LPX_DEVICE_ERROR, "k_sweep64_pull: a hand-over wait between the two stages hit its spin bound (code " +
                                      std::to_string(f) + "); the tableau of this handle is not valid");
  }
  return 0;
}
int sync_ctl_to_host(lpx_state* s) {
  if (int rc = enqueue_ctl_fetch(s)) return rc;
  HIP_TRY(hipStreamSynchronize(s->stream));
  return check_fetched_fail_word(s);
}

int push_ctl(lpx_state* s) {
  HIP_TRY(hipMemcpyAsync(s->B.ctl, s->h_ctl, sizeof(LpxCtl), hipMemcpyHostToDevice, s->stream));
  return 0;
}

// Everything ensure_block_ring allocates; leaves the handle as if no ring had ever been built.
static void free_block_ring(lpx_state* s) {
  (void)hipFree(s->R.prow);
  (void)hipFree(s->R.col);
  (void)hipFree(s->R.col0);
  (void)hipFree(s->R.row0);
  (void)hipFree(s->R.up);
  (void)hipFree(s->R.chain_part_a);
  (void)hipFree(s->R.chain_part_b);
  (void)hipFree(s->R.chain_bar);
  (void)hipFree(s->R.chain_own_col);
  (void)hipFree(s->R.chain_own_prow);
  (void)hipFree(s->R.chain_own_dvc);
  (void)hipFree(s->R.chain_own_b);
  (void)hipFree(s->R.chain_own_rs);
  (void)hipFree(s->R.chain_dbg);
  (void)hipFree(s->R.census);
  (void)hipFree(const_cast<double*>(s->R.zeros));
  (void)hipFree(s->R.tickets);
  (void)hipFree(s->R.sweep_fail);
  (void)hipFree(s->R.clk);
  (void)hipFree(s->R.col_packed);
  (void)hipFree(s->R.fix_col);
  (void)hipFree(s->R.fix_row);
  (void)hipFree(s->R.mg_mail);
  (void)hipFree(s->R.mg_arrive);
  (void)hipFree(s->R.mg_candrow);
  (void)hipFree(s->R.mg_arrive2);
  (void)hipFree(s->d_cand);
  s->R = lpxk::BlockRing{};
  s->d_cand = nullptr;
}

void free_state(lpx_state* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  for (hipEvent_t e : s->ev) (void)hipEventDestroy(e);
  (void)hipFree(s->A_base[0]);
  (void)hipFree(s->A_base[1]);
  (void)hipFree(s->b_base[0]);
  (void)hipFree(s->b_base[1]);
  (void)hipFree(s->B.c);
  (void)hipFree(s->B.prow);
  (void)hipFree(s->B.col[0]);
  (void)hipFree(s->B.col[1]);
  (void)hipFree(s->B.partial);
  (void)hipFree(s->B.perm);
  (void)hipFree(s->B.ctl);
  (void)hipFree(s->d_sum);
  (void)hipFree(s->ring);
  (void)hipFree(s->prow2);
  free_block_ring(s);
  if (s->ev_upd) (void)hipEventDestroy(s->ev_upd);

  if (s->ev_peek) (void)hipEventDestroy(s->ev_peek);
  if (s->ev_decide) (void)hipEventDestroy(s->ev_decide);
  if (s->h_ctl) (void)hipHostFree(s->h_ctl);
  if (s->h_snap) (void)hipHostFree(s->h_snap);
  for (hipEvent_t e : s->ev_batch) if (e) (void)hipEventDestroy(e);
  if (s->ov_chain) (void)hipStreamDestroy(s->ov_chain);
  if (s->ov_sweep) (void)hipStreamDestroy(s->ov_sweep);
  for (hipEvent_t e : s->ev_ov_chain) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : s->ev_ov_sweep) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : s->ev_ov_join) if (e) (void)hipEventDestroy(e);
  for (hipStream_t t : s->ov_fix) if (t) (void)hipStreamDestroy(t);
  for (hipEvent_t e : s->ev_ov_fix) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : s->ev_ov_pack) if (e) (void)hipEventDestroy(e);
  if (s->own_stream) (void)hipStreamDestroy(s->own_stream);
  delete s;
}

// Allocates buffers for an m_local x n_cap tableau (n columns in use) and zero-fills the padding.
int alloc_state(int32_t m_local, int32_t n, int32_t n_cap, int32_t row0, int32_t m_global, int device,
                       lpx_state** out) {
  if (m_local < 0 || n < 0 || n_cap < n || row0 < 0 || m_global < m_local || row0 + m_local > m_global)
    return fail(LPX_BAD_ARGUMENT, "lpx_state_create: bad dimensions");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(LPX_BAD_ARGUMENT, "lpx_state_create: no such device");
  HIP_TRY(hipSetDevice(device));
  lpx_state* s = new lpx_state();
  s->device = device;
  s->m = m_local; s->n = n; s->n_cap = n_cap; s->row0 = row0; s->m_global = m_global;
  const int64_t ld = std::max<int64_t>(16, round_up(n_cap, 16));
  const int64_t mp = std::max<int64_t>(2, round_up(m_local, 2)) + 2;
  s->B.ld = ld;
  memcpy(s->opt, env_defaults(), sizeof s->opt);
  resolve_arithmetic(s);
  s->B.chain_form = (int)s->opt[LPX_OPT_CHAIN_FORM];
  apply_layout_options(s);
#define ALLOC(ptr, count, type)                                                            \
  do {                                                                                     \
    hipError_t _e = hipMalloc((void**)&(ptr), std::max<size_t>(1, (size_t)(count)) * sizeof(type)); \
    if (_e != hipSuccess) { free_state(s); return fail(LPX_DEVICE_ERROR, std::string("hipMalloc: ") + hipGetErrorString(_e)); } \
    _e = hipMemset((ptr), 0, std::max<size_t>(1, (size_t)(count)) * sizeof(type));          \
    if (_e != hipSuccess) { free_state(s); return fail(LPX_DEVICE_ERROR, std::string("hipMemset: ") + hipGetErrorString(_e)); } \
  } while (0)
  ALLOC(s->B.A, mp * ld, double);
  s->A_base[0] = s->B.A;
  ALLOC(s->B.b, mp, double);
  s->b_base[0] = s->B.b;
  ALLOC(s->B.c, ld, double);
  ALLOC(s->B.prow, ld, double);
  ALLOC(s->B.col[0], mp, double);
  ALLOC(s->B.col[1], mp, double);
  ALLOC(s->B.partial, std::max(1, (m_local + 1) / 2 + 1), RatioRow);  // room for the finest tiling (2 rows per tile)
  ALLOC(s->B.perm, (int64_t)n_cap + m_global, int32_t);
  ALLOC(s->B.ctl, 1, LpxCtl);
  ALLOC(s->d_sum, 4, unsigned long long);
  ALLOC(s->ring, 2, LpxCtl);
  ALLOC(s->prow2, ld, double);
#undef ALLOC
  hipError_t e = hipHostMalloc((void**)&s->h_ctl, sizeof(LpxCtl) + 16, hipHostMallocDefault);   // + the sweep's fail word
  if (e != hipSuccess) { free_state(s); return fail(LPX_DEVICE_ERROR, "hipHostMalloc failed"); }
  memset(s->h_ctl, 0, sizeof(LpxCtl) + 16);
  if (hipHostMalloc((void**)&s->h_snap, 2 * sizeof(LpxCtl), hipHostMallocDefault) != hipSuccess ||
      hipEventCreateWithFlags(&s->ev_batch[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&s->ev_batch[1], hipEventDisableTiming) != hipSuccess) {
    free_state(s);
    return fail(LPX_DEVICE_ERROR, "pinned snapshot / event allocation failed");
  }
  e = hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess) { free_state(s); return fail(LPX_DEVICE_ERROR, "hipStreamCreate failed"); }
  s->stream = s->own_stream;
  if (hipEventCreateWithFlags(&s->ev_peek, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&s->ev_upd, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&s->ev_decide, hipEventDisableTiming) != hipSuccess) {
    free_state(s);
    return fail(LPX_DEVICE_ERROR, "hipEventCreate failed");
  }
  HIP_TRY(hipDeviceSynchronize());  // the memsets above ran on the null stream
  *out = s;
  return 0;
}

void init_ctl(lpx_state* s, double v) {
  LpxCtl& c = *s->h_ctl;
  memset(&c, 0, sizeof c);
  c.v = v;
  c.e_next = -1; c.e_cur = -1; c.l = -1;
  c.status = lpxk::kRunning;
  c.track = -1;
  c.max_pivots = -1;
  c.e_min = INT32_MAX;
  c.ticket = 0;
}

int upload_common(lpx_state* s, const double* A, int64_t lda, const double* b, const double* c, double v,
                         const int32_t* perm, hipMemcpyKind kind) {
  const int32_t m = s->m, n = s->n;
  if (m > 0 && n > 0) {
    if (lda < n) return fail(LPX_BAD_ARGUMENT, "lpx_state_create: lda < n");
    HIP_TRY(hipMemcpy2DAsync(s->B.A, s->B.ld * sizeof(double), A, lda * sizeof(double), (size_t)n * sizeof(double),
                             (size_t)m, kind, s->stream));
  }
  if (m > 0) HIP_TRY(hipMemcpyAsync(s->B.b, b, (size_t)m * sizeof(double), kind, s->stream));
  if (n > 0) HIP_TRY(hipMemcpyAsync(s->B.c, c, (size_t)n * sizeof(double), kind, s->stream));
  std::vector<int32_t> p((size_t)n + s->m_global);
  for (size_t i = 0; i < p.size(); i++) p[i] = perm ? perm[i] : (int32_t)i;
  if (!p.empty())
    HIP_TRY(hipMemcpyAsync(s->B.perm, p.data(), p.size() * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
  init_ctl(s, v);
  if (int rc = push_ctl(s)) return rc;
  HIP_TRY(hipStreamSynchronize(s->stream));
  return 0;
}

extern "C" int lpx_state_create(int32_t m_local, int32_t n, const double* A, int64_t lda, const double* b,
                                const double* c, double v, const int32_t* perm, int32_t row0, int32_t m_global,
                                int device, lpx_state** out) {
  if (!out) return fail(LPX_BAD_ARGUMENT, "lpx_state_create: out is NULL");
  *out = nullptr;
  if ((m_local > 0 && n > 0 && !A) || (m_local > 0 && !b) || (n > 0 && !c))
    return fail(LPX_BAD_ARGUMENT, "lpx_state_create: NULL array");
  lpx_state* s = nullptr;
  if (int rc = alloc_state(m_local, n, n, row0, m_global, device, &s)) return rc;
  if (int rc = upload_common(s, A, lda, b, c, v, perm, hipMemcpyHostToDevice)) { free_state(s); return rc; }
  *out = s;
  return 0;
}

extern "C" int lpx_state_create_from_device(int32_t m_local, int32_t n, const double* dA, int64_t lda,
                                            const double* db, const double* dc, double v, const int32_t* perm,
                                            int32_t row0, int32_t m_global, int device, lpx_state** out) {
  if (!out) return fail(LPX_BAD_ARGUMENT, "lpx_state_create_from_device: out is NULL");
  *out = nullptr;
  if ((m_local > 0 && n > 0 && !dA) || (m_local > 0 && !db) || (n > 0 && !dc))
    return fail(LPX_BAD_ARGUMENT, "lpx_state_create_from_device: NULL array");
  lpx_state* s = nullptr;
  if (int rc = alloc_state(m_local, n, n, row0, m_global, device, &s)) return rc;
  if (int rc = upload_common(s, dA, lda, db, dc, v, perm, hipMemcpyDeviceToDevice)) { free_state(s); return rc; }
  *out = s;
  return 0;
}

extern "C" void lpx_state_destroy(lpx_state* s) { free_state(s); }

extern "C" int lpx_state_set_stream(lpx_state* s, void* hip_stream) {
  if (!s) return fail(LPX_BAD_ARGUMENT, "NULL state");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->stream = hip_stream ? (hipStream_t)hip_stream : s->own_stream;
  return 0;
}

extern "C" int lpx_state_dims(const lpx_state* s, int32_t* m_local, int32_t* n, int32_t* row0, int32_t* m_global) {
  if (!s) return fail(LPX_BAD_ARGUMENT, "NULL state");
  if (m_local) *m_local = s->m;
  if (n) *n = s->n;
  if (row0) *row0 = s->row0;
  if (m_global) *m_global = s->m_global;
  return 0;
}

// ------------------------------------------------------------------------------------------------ launches
int launch_update_profiled(lpx_state* s, const double* prow, const LpxCtl* up,
                                  const Buffers* Bin, double* A_out, double* b_out) {
  if (!prow) prow = s->B.prow;
  if (!up) up = s->B.ctl;
  const Buffers& BB = Bin ? *Bin : s->B;
  if (s->prof > 0 && (s->prof_seq++ % s->prof) == 0) {
    if (s->ev_used + 2 > s->ev.size()) {
      for (int k = 0; k < 512; k++) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        s->ev.push_back(e);
      }
    }
    HIP_TRY(hipEventRecord(s->ev[s->ev_used], s->stream));
    lpxk::launch_update(BB, s->m, s->n, s->row0, s->g, s->nontemporal, prow, up, A_out, b_out, s->stream);
    HIP_TRY(hipEventRecord(s->ev[s->ev_used + 1], s->stream));
    s->ev_used += 2;
  } else {
    lpxk::launch_update(BB, s->m, s->n, s->row0, s->g, s->nontemporal, prow, up, A_out, b_out, s->stream);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

static int drain_profile(lpx_state* s) {
  if (s->ev_used == 0) return 0;
  HIP_TRY(hipStreamSynchronize(s->stream));
  for (size_t k = 0; k + 1 < s->ev_used; k += 2) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev[k], s->ev[k + 1]));
    s->prof_ms += ms;
    s->prof_launches += 1;
  }
  s->ev_used = 0;
  return 0;
}

extern "C" int lpx_profile_enable(lpx_state* s, int enable) {
  if (!s) return fail(LPX_BAD_ARGUMENT, "NULL state");
  HIP_TRY(hipSetDevice(s->device));
  if (int rc = drain_profile(s)) return rc;
  s->prof = enable > 0 ? enable : 0;
  s->prof_seq = 0;
  s->prof_launches = 0;
  s->prof_ms = 0.0;
  return 0;
}

extern "C" int lpx_profile_read(lpx_state* s, int64_t* launches, double* total_ms) {
  if (!s) return fail(LPX_BAD_ARGUMENT, "NULL state");
  HIP_TRY(hipSetDevice(s->device));
  if (int rc = drain_profile(s)) return rc;
  if (launches) *launches = s->prof_launches;
  if (total_ms) *total_ms = s->prof_ms;
  s->prof_launches = 0;
  s->prof_ms = 0.0;
  return 0;
}

// entering scan at the start of a loop / for getEntering(), honouring the handle's pricing rule
void launch_seed_entering(lpx_state* s, const lpxk::LoopStart& start) {
  if (s->pricing == 1) lpxk::launch_entering_dantzig(s->B, s->n, true, s->stream, start);
  else lpxk::launch_entering(s->B, s->n, s->stream, start);
}

extern "C" int lpx_state_set_pricing(lpx_state* s, int32_t pricing) {
  if (!s || (pricing != 0 && pricing != 1)) return fail(LPX_BAD_ARGUMENT, "lpx_state_set_pricing: bad argument");
  s->pricing = pricing;
  return 0;
}

// ------------------------------------------------------------------------------------------------ step API
static int require_single(lpx_state* s, const char* who) {
  if (!s) return fail(LPX_BAD_ARGUMENT, std::string(who) + ": NULL state");
  if (s->row0 != 0 || s->m_global != s->m)
    return fail(LPX_BAD_ARGUMENT, std::string(who) + ": not available on a row-block shard");
  return 0;
}

int set_running(lpx_state* s, int64_t max_pivots, int32_t track) {
  if (int rc = sync_ctl_to_host(s)) return rc;
  s->h_ctl->status = lpxk::kRunning;
  s->h_ctl->do_update = 0;
  s->h_ctl->pivots = 0;
  s->h_ctl->max_pivots = max_pivots;
  s->h_ctl->track = track;
  s->h_ctl->e_min = INT32_MAX;
  s->h_ctl->ticket = 0;
  return push_ctl(s);
}

extern "C" int lpx_get_entering(lpx_state* s, int32_t* entering) {
  if (!s || !entering) return fail(LPX_BAD_ARGUMENT, "lpx_get_entering: NULL argument");
  HIP_TRY(hipSetDevice(s->device));
  if (int rc = set_running(s, -1, -1)) return rc;
  launch_seed_entering(s);
  HIP_TRY(hipGetLastError());
  if (int rc = sync_ctl_to_host(s)) return rc;
  *entering = s->h_ctl->e_next;
  return 0;
}

extern "C" int lpx_get_leaving(lpx_state* s, int32_t entering, int32_t* leaving, double* ratio) {
  if (!s || !leaving) return fail(LPX_BAD_ARGUMENT, "lpx_get_leaving: NULL argument");
  if (!(entering >= 0 && entering < s->n))  // Validate.isTrue, LPState.java:288
    return fail(LPX_BAD_ARGUMENT, "lpx_get_leaving: entering outside [0, n)");
  HIP_TRY(hipSetDevice(s->device));
  if (int rc = set_running(s, -1, -1)) return rc;
  lpxk::launch_ratio_gather(s->B, s->m, s->row0, s->g, entering, s->stream);
  lpxk::launch_reduce_partials(s->B, s->g, s->stream);
  HIP_TRY(hipGetLastError());
  if (int rc = sync_ctl_to_host(s)) return rc;
  *leaving = s->h_ctl->l;
  if (ratio) *ratio = s->h_ctl->ratio;
  return 0;
}

extern "C" int lpx_pivot(lpx_state* s, int32_t entering, int32_t leaving) {
  if (int rc = require_single(s, "lpx_pivot")) return rc;
  if (!(entering >= 0 && entering < s->n) || !(leaving >= 0 && leaving < s->m))
    return fail(LPX_BAD_ARGUMENT, "lpx_pivot: index out of range");  // ArrayIndexOutOfBounds in the reference
  HIP_TRY(hipSetDevice(s->device));
  if (int rc = set_running(s, -1, -1)) return rc;
  lpxk::launch_ratio_gather(s->B, s->m, s->row0, s->g, entering, s->stream);  // column `entering` -> col[parity]
  lpxk::launch_select_pivot(s->B, s->n, s->m_global, s->g, entering, leaving, s->stream);
  if (int rc = launch_update_profiled(s)) return rc;
  HIP_TRY(hipGetLastError());
  if (int rc = sync_ctl_to_host(s)) return rc;
  if (s->h_ctl->status == LPX_DIVIDE_BY_ZERO) return fail(LPX_DIVIDE_BY_ZERO, "lpx_pivot: pivot element is zero");
  return 0;
}

// ------------------------------------------------------------------------------------------------ blocked loop
// Buffers that peer GPUs store into while this device's persistent kernel polls them (the pivot-row ring, the mailbox,
// the arrival words of an lpx_multi shard) must be fine-grained: on coarse-grained memory system-scope accesses give no
// cross-agent coherence inside a kernel, and the outcome would be a spin-bound failure or — worse — a stale pivot row.
// No downgrade: if the runtime has no fine-grained device memory the multi-GPU handle cannot be built, and says so.
hipError_t peer_visible_malloc(const lpx_state* s, void** ptr, size_t bytes) {
  if (s->peer_written) return hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocFinegrained);
  return hipMalloc(ptr, bytes);
}

int32_t state_n(const lpx_state* s) { return s->n; }

static int build_block_ring(lpx_state* s);
int ensure_block_ring(lpx_state* s) {
  if (s->R.prow) return 0;
  const int rc = build_block_ring(s);
  if (rc) {   // all or nothing: a half-built ring would make the next call return 0 with NULL pointers inside
    const std::string why = g_last_error;
    free_block_ring(s);
    if (s->peer_written) return fail(rc, (why + " (ring of a multi-GPU shard: peer-written buffers need fine-grained device memory)").c_str());
    g_last_error = why;
  }
  return rc;
}
static int build_block_ring(lpx_state* s) {
  const int64_t mp = std::max<int64_t>(2, round_up(s->m, 2)) + 2;
  s->R.mp = mp;
  const size_t K = 2 * lpxk::kBlockMax;  // two halves: the block being decided and the one being swept
  HIP_TRY(peer_visible_malloc(s, (void**)&s->R.prow, K * (size_t)s->B.ld * sizeof(double)));
  HIP_TRY(peer_visible_malloc(s, &s->R.mg_mail, 2 * lpxk::kMaxDevices * 32));
  HIP_TRY(peer_visible_malloc(s, (void**)&s->R.mg_arrive, lpxk::kChainMaxWgs * sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(s->R.mg_mail, 0, 2 * lpxk::kMaxDevices * 32, s->stream));
  HIP_TRY(hipMemsetAsync(s->R.mg_arrive, 0, lpxk::kChainMaxWgs * sizeof(unsigned long long), s->stream));
  if (s->m != s->m_global || s->peer_written || s->multi_shard) {   // a shard of an lpx_multi: the one-hop exchange buffers
    const size_t rows_bytes = 2 * (size_t)lpxk::kMaxDevices * (size_t)s->B.ld * sizeof(double);
    const size_t words_bytes = 2 * (size_t)lpxk::kMaxDevices * lpxk::kChainMaxWgs * sizeof(unsigned long long);
    HIP_TRY(peer_visible_malloc(s, (void**)&s->R.mg_candrow, rows_bytes));
    HIP_TRY(peer_visible_malloc(s, (void**)&s->R.mg_arrive2, words_bytes));
    HIP_TRY(hipMemsetAsync(s->R.mg_candrow, 0, rows_bytes, s->stream));
    HIP_TRY(hipMemsetAsync(s->R.mg_arrive2, 0, words_bytes, s->stream));
  }
  HIP_TRY(hipMalloc((void**)&s->R.col, K * (size_t)mp * sizeof(double)));
  HIP_TRY(hipMalloc((void**)&s->R.col0, K * (size_t)mp * sizeof(double)));
  HIP_TRY(hipMalloc((void**)&s->R.row0, K * (size_t)s->B.ld * sizeof(double)));
  HIP_TRY(hipMemsetAsync(s->R.col0, 0, K * (size_t)mp * sizeof(double), s->stream));
  HIP_TRY(hipMemsetAsync(s->R.row0, 0, K * (size_t)s->B.ld * sizeof(double), s->stream));
  HIP_TRY(hipMalloc((void**)&s->R.up, K * sizeof(LpxCtl)));
  HIP_TRY(hipMalloc(&s->R.chain_part_a, 2 * lpxk::kChainMaxWgs * 64));  // two sets, alternating by decision
  HIP_TRY(hipMemsetAsync(s->R.chain_part_a, 0, 2 * lpxk::kChainMaxWgs * 64, s->stream));  // no tag of any launch
  HIP_TRY(hipMalloc(&s->R.chain_part_b, lpxk::kChainMaxWgs * 16));
  HIP_TRY(hipMalloc((void**)&s->R.chain_bar, 512));  // two barrier counters and the hand-off word, a line each
  HIP_TRY(hipMemsetAsync(s->R.chain_bar, 0, 512, s->stream));
  HIP_TRY(hipMalloc((void**)&s->R.chain_own_col, K * (size_t)mp * sizeof(double)));
  HIP_TRY(hipMalloc((void**)&s->R.chain_own_prow, K * (size_t)s->B.ld * sizeof(double)));
  HIP_TRY(hipMalloc((void**)&s->R.chain_own_dvc, K * (size_t)mp * sizeof(double)));
  HIP_TRY(hipMalloc((void**)&s->R.chain_own_b, (size_t)mp * sizeof(double)));
  HIP_TRY(hipMalloc((void**)&s->R.chain_own_rs, (size_t)(mp + s->B.ld) * sizeof(int32_t)));
  HIP_TRY(hipMemsetAsync(s->R.chain_own_rs, 0xff, (size_t)(mp + s->B.ld) * sizeof(int32_t), s->stream));
  HIP_TRY(hipMemsetAsync(s->R.chain_own_col, 0, K * (size_t)mp * sizeof(double), s->stream));
  HIP_TRY(hipMemsetAsync(s->R.chain_own_prow, 0, K * (size_t)s->B.ld * sizeof(double), s->stream));
  HIP_TRY(hipMalloc((void**)&s->R.chain_dbg, 16 * lpxk::kBlockMax * sizeof(long long)));   // 5 (k_block_chain_t) or 8 stamps per decision
  HIP_TRY(hipMemsetAsync(s->R.chain_dbg, 0, 16 * lpxk::kBlockMax * sizeof(long long), s->stream));
  HIP_TRY(hipMalloc((void**)&s->R.census, (lpxk::kChainMaxWgs + 2 + 1200) * sizeof(unsigned)));   // + room for diagnostic builds
  HIP_TRY(hipMemsetAsync(s->R.census, 0, (lpxk::kChainMaxWgs + 2 + 1200) * sizeof(unsigned), s->stream));
  HIP_TRY(hipMalloc((void**)&s->R.zeros, 256));
  HIP_TRY(hipMemsetAsync(const_cast<double*>(s->R.zeros), 0, 256, s->stream));
  // k_sweep32_pull: a ticket counter per 128-column sub-strip (128 bytes apart) and the block's multipliers packed by
  // batches of 4 rows (1 KiB each)
  // (both once per ring half: the pack kernel of block k runs while the sweep of block k-1 still pulls and reads its own)
  HIP_TRY(hipMalloc((void**)&s->R.tickets, 2 * (size_t)lpxk::sweep_ticket_slots(s->B.ld) * 128));
  HIP_TRY(hipMemsetAsync(s->R.tickets, 0, 2 * (size_t)lpxk::sweep_ticket_slots(s->B.ld) * 128, s->stream));
  HIP_TRY(hipMalloc((void**)&s->R.sweep_fail, 128));
  HIP_TRY(hipMemsetAsync(s->R.sweep_fail, 0, 128, s->stream));
  HIP_TRY(hipMalloc((void**)&s->R.clk, 256));
  HIP_TRY(hipMemsetAsync(s->R.clk, 0, 256, s->stream));
  HIP_TRY(hipMalloc((void**)&s->R.col_packed, 2 * (size_t)(mp / 4 + 1) * 2048));
  HIP_TRY(hipMemsetAsync(s->R.col_packed, 0, 2 * (size_t)(mp / 4 + 1) * 2048, s->stream));
  HIP_TRY(hipMalloc((void**)&s->d_cand, (size_t)(LPX_CAND_HEADER + s->B.ld) * sizeof(double)));
  HIP_TRY(hipMemsetAsync(s->R.prow, 0, K * (size_t)s->B.ld * sizeof(double), s->stream));
  HIP_TRY(hipMemsetAsync(s->R.col, 0, K * (size_t)mp * sizeof(double), s->stream));
  HIP_TRY(hipMemsetAsync(s->R.up, 0, K * sizeof(LpxCtl), s->stream));
  HIP_TRY(hipMemsetAsync(s->d_cand, 0, (size_t)(LPX_CAND_HEADER + s->B.ld) * sizeof(double), s->stream));
  lpxk::preload_block_kernels(s->B, s->R, s->stream);
  return 0;
}

// The second tableau / b of the out-of-place forms (pipeline 2 of the shards, the overlapped blocked loop).
int ensure_spare_tableau(lpx_state* s) {
  if (s->A2) return 0;
  const int64_t mp = std::max<int64_t>(2, round_up(s->m, 2)) + 2;
  // 4 KiB skew between the two buffers: measured 346 us vs 354 us per cfg3 update with none (the read and the
  // write stream then do not hit the same HBM channel at the same time); profiles/r01_cu_mask.log
  const int64_t off = s->opt[LPX_OPT_A2_OFFSET];  // in doubles
  HIP_TRY(hipMalloc((void**)&s->A_base[1], (size_t)(mp * s->B.ld + off) * sizeof(double)));
  s->A2 = s->A_base[1] + off;
  HIP_TRY(hipMalloc((void**)&s->b_base[1], (size_t)mp * sizeof(double)));
  s->b2 = s->b_base[1];
  HIP_TRY(hipMemset(s->A2, 0, (size_t)(mp * s->B.ld) * sizeof(double)));
  HIP_TRY(hipMemset(s->b2, 0, (size_t)mp * sizeof(double)));
  HIP_TRY(hipDeviceSynchronize());
  return 0;
}

// Pivots per sweep.  Measured on MI355X: one decision (peek + pack + commit, three latency-bound launches) costs
// ~18 us + ~0.4 us per pending pivot whatever the size; a sweep moves the tableau once at ~5.6 TB/s up to K = 16
// and at ~4.7 TB/s at K = 32 (there the 2K fp64 operations per entry co-limit it); the one-pass form costs one
// pass at ~6.3 TB/s + ~9 us per pivot.  Per pivot: blocked(K) ~ 18 + 0.4 K/2 + sweep(K)/K.
int choose_block(const lpx_state* s) {
  int K = (int)s->opt[LPX_OPT_BLOCK];
  if (K == 0) {
    const double sweep_us = 16.0 * (double)s->m * (double)s->B.ld / 6.0e6;
    // Re-measured with the round-5 decision kernel and launches (profiles/r05_block_policy_{small,mid}_sizes.txt,
    // r05_block_policy_64_and_tiny.txt; same box, 2048 pivots, pivots/s).  A block costs ~17 us besides its decisions (~12 us
    // from the end of one decision launch to the entry of the next, 4-6 us of prologue: profiles/r05_chain_launch_stamps.txt),
    // so longer blocks win as soon as the ladder of pending pivots is cheap enough:
    //   two launches per pivot against blocks of 16: 256 x 512 58-62k vs 99k, 1024 x 2048 (16 MiB) 50k vs 96k — blocks win
    //     at EVERY size once their ring exists; but building the ring (buffers, streams, one preparing launch of every
    //     kernel) takes ~24 ms, what ~2 400 pivots of a small tableau save, and a tableau of a few hundred rows is solved
    //     in fewer: below ~18 MiB a handle that has no ring yet keeps the two-launch loop, one that has takes blocks of 16;
    //   16 against 32: 1024 x 2048 95.8k / 95.0k, 1536 x 2048 (24 MiB) 95.0k / 94.2k, 2048 x 2048 (32 MiB) 92.9k / 94.7k,
    //     2048 x 4096 93.8k / 96.1k, 4096 x 4096 88.2k / 91.2k, 4096 x 8192 (256 MiB) 85.9k / 89.5k, 6144 x 6144 85.2k / 88.6k;
    //   32 against 64 (plain: k_sweep64_one): 4096 x 8192 89.5k / 80.0k.
    if (sweep_us < 6.3) K = s->R.prow ? 16 : 1;
    else if (sweep_us < 10.0) K = 16;   // up to ~28 MiB
    else K = 32;
    // 64: the two-stage sweep moves half the bytes per pivot and takes 0.74x the time per pivot alone on the chip, but
    // 64-slot decisions cost twice as much each (their ring reads grow with K^2) and take bandwidth from the sweep
    // beside them: +3..5 % over 32 from 4 GiB to 12 GiB tableaux (profiles/r02_block64_policy.txt), -22 % at 2 GiB.
    // By size only from ~7 GiB up (8 GiB +4.3 %, 12 GiB +5.4 %), where the kernel applies; opt-in below.
    if (sweep_us >= 2500.0 && s->m % 4 == 0 && s->B.ld >= 512) K = 64;
    // fused arithmetic: blocks of 33..64 run on the matrix cores (k_sweep64_mfma2, 16-row tiles).  Round 5, blocks of 32 / 64:
    // 0.5 GiB (8192 x 8192) 86.8k / 81.8k, 0.625 GiB 84.9k / 79.6k, 0.75 GiB 79.8k / 77.9k (four shapes, all for 32),
    // 0.875 GiB (8192 x 14336) 73.3k / 75.7k, 1 GiB (cfg3) 67.7k / 73.7k; round 4 above: 1.25 GiB 48.0k / 53.0k,
    // cfg4 18.2k / 27.5k (profiles/r04_block_by_size_fused.txt).  From ~0.85 GiB.
    if (s->B.fused && sweep_us >= 300.0 && s->m % 16 == 0 && s->B.ld >= 512 && s->m_global == s->m) K = 64;
  }
  // three launches per decision (option chain = 0, the form the shards use): its kernels hold at most 32 pending pivots
  if (!s->opt[LPX_OPT_CHAIN] || s->m_global != s->m) K = std::min(K, (int)lpxk::kShardBlockMax);
  return std::max(1, std::min(K, (int)lpxk::kBlockMax));
}

// Decisions of the next block: K while the budget lasts, then whatever is left INCLUDING the decision that only
// reports the end of the budget — one block, one sweep for the tail (the sweep kernels take any number of pending
// pivots up to their template size at full speed, see sweep_apply).
int block_len(int K, int64_t max_pivots, int64_t decided) {
  if (max_pivots < 0) return K;
  const int64_t room = max_pivots + 1 - decided;  // +1: the decision that reports PIVOT_LIMIT
  return (int)std::max<int64_t>(0, std::min<int64_t>(K, room));
}

// Residency of the persistent decision kernel: every workgroup spins at grid barriers, so the grid must not exceed
// what the CUs it may use can hold at once.  cus = the CUs of the stream's mask (all of them without a mask).
int clamp_chain_wgs(lpx_state* s, int want, int cus) {
  const int per_cu = lpxk::chain_blocks_per_cu();
  const int cap = std::max(1, per_cu * std::max(1, cus));
  s->info.chain_blocks_per_cu = per_cu;
  s->info.chain_resident_max = cap;
  s->info.chain_wgs_requested = want;
  const int G = std::max(1, std::min(std::min(want, cap), (int)lpxk::kChainMaxWgs));
  s->info.chain_wgs = G;
  return G;
}

int device_cus(const lpx_state* s) {
  int ncu = 0;
  if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, s->device) != hipSuccess) ncu = 0;
  return ncu;
}

// B / R: destination buffers and the ring half of the block; A_src / b_src != NULL: out of place
int launch_sweep_profiled(lpx_state* s, int K, hipStream_t stream, const Buffers& B, const lpxk::BlockRing& R,
                                 const double* A_src, const double* b_src, const lpxk::FixSide* side, hipEvent_t stop) {
  const bool sample = s->prof > 0 && (s->prof_seq++ % s->prof) == 0;
  if (sample) {
    if (s->ev_used + 2 > s->ev.size()) {
      for (int k = 0; k < 512; k++) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        s->ev.push_back(e);
      }
    }
  }
  // CUs the sweep's stream may use: all but the decisions' reserved ones on the masked overlap stream
  const int cus = (stream == s->ov_sweep && s->ov_masked) ? s->ov_sweep_cus : device_cus(s);
  int kernel_used = 0;
  s->info.sweep_cus = cus;
  s->info.sweep_rows = lpxk::launch_block_sweep(B, R, s->n, s->m, s->row0, K, (int)s->opt[LPX_OPT_SWEEP_ROWS],
                                                s->nontemporal, stream, A_src, b_src,
                                                sample ? s->ev[s->ev_used + 1] : nullptr, cus,
                                                (int)s->opt[LPX_OPT_SWEEP_FORM], &kernel_used, side, stop,
                                                sample ? s->ev[s->ev_used] : nullptr);
  s->info.sweep_kernel = kernel_used;
  if (sample) s->ev_used += 2;
  HIP_TRY(hipGetLastError());
  return 0;
}
static int launch_sweep_profiled(lpx_state* s, int K) {
  return launch_sweep_profiled(s, K, s->stream, s->B, s->R, nullptr, nullptr, nullptr, nullptr);
}

// Ring half h as a ring of its own (what the sweep / fix-up kernels take).
lpxk::BlockRing ring_half(const lpx_state* s, int h) {
  lpxk::BlockRing R = s->R;
  const int64_t o = (int64_t)h * lpxk::kBlockMax;
  R.prow += o * s->B.ld;
  R.row0 += o * s->B.ld;
  R.col += o * R.mp;
  R.col0 += o * R.mp;
  if (R.fix_col) R.fix_col += o * R.mp;
  if (R.fix_row) R.fix_row += o * s->B.ld;
  if (R.col_packed) R.col_packed += (int64_t)h * (R.mp / 4 + 1) * (2048 / sizeof(double));
  if (R.tickets) R.tickets += (int64_t)h * lpxk::sweep_ticket_slots(s->B.ld) * 32;
  R.up += o;
  return R;
}

// CU masks on MI355X (measured, scripts/micro/cu_mask.hip -> profiles/r02_cu_mask.txt): mask bit i names CU i / 8 of
// XCD i % 8, and an XCD whose bits are ALL clear is not excluded — it runs unmasked.  So a stream cannot be kept off
// an XCD; what a mask can do is reserve the same few CUs on every XCD.  The decisions get the last kChainCusPerXcd CUs
// of each XCD (one resident workgroup each), the sweep the others: the two kernels never wait for each other's CUs.
#ifndef LPX_CHAIN_CUS_PER_XCD
#define LPX_CHAIN_CUS_PER_XCD 4
#endif
// By size (LPX_OPT_CHAIN_CUS = 0).  One row / one column per thread is the fastest decision (every further pass of a
// thread is a serial round trip): up to 8192 rows / columns that is 32 workgroups on LPX_CHAIN_CUS_PER_XCD = 4 CUs per
// XCD; above, where the loop is bound by the decisions and the sweep has slack (tableaus up to 1.5 GiB: cfg3), 8 CUs
// per XCD hold 64 workgroups — cfg3 45.6k -> 49.8k pivots/s, same box; at 2 GiB and above (cfg4) the sweep is the
// bound and keeps its 224 CUs (8 CUs per XCD there: -4 %).  profiles/r03_decision_grid.txt.
static int chain_cus_per_xcd(const lpx_state* s, int per_xcd) {
  int k = LPX_CHAIN_CUS_PER_XCD;
  const int64_t work = std::max<int64_t>(s->m, s->B.ld);
  const double bytes = 8.0 * (double)s->m * (double)s->B.ld;
  if (!s->multi_shard && work > 8192 && bytes <= 1.5 * 1073741824.0) k = 8;
  // fused arithmetic: the sweep is bound by its memory pass and keeps its time on 192 CUs (cfg4: 1.548 vs 1.544 ms,
  // profiles/r04_arith_grid_cfg4_a.txt), and one row per thread halves the decisions' phase A at 32768 rows
  if (!s->multi_shard && work > 8192 && s->B.fused) k = 8;
  // Multiples of 4 only: workgroups are dealt round the four shader engines of an XCD, so the reserved CUs must be the
  // same number on each of them — with 6 per XCD (2 + 2 + 1 + 1) the 48-workgroup grid was not resident (measured:
  // bounded wait -> LPX_DEVICE_ERROR; likewise one extra CU on one XCD, profiles/r03_decision_passes.txt).
  if (s->opt[LPX_OPT_CHAIN_CUS] > 0) k = std::max(4, (int)s->opt[LPX_OPT_CHAIN_CUS] / 4 * 4);
  return std::max(1, std::min(k, (per_xcd - 1) / 4 * 4));
}
int ensure_overlap_streams(lpx_state* s) {
  if (s->ov_chain) return 0;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, s->device));
  const int ncu = prop.multiProcessorCount;
  const int per_xcd = ncu / 8;
  std::vector<uint32_t> m_sweep((ncu + 31) / 32, 0u), m_chain((ncu + 31) / 32, 0u);
  // LPX_OPT_SWEEP_CUS (a multiple of 8; 0 = all that is left): the sweep's share per XCD.  Fewer CUs = less fp64 power
  // beside the decisions, whose latency follows the shader clock the power cap leaves (EXPERIMENTS.md section 0).
  const int kChainCusPerXcd = chain_cus_per_xcd(s, per_xcd);
  int sweep_per_xcd = per_xcd - kChainCusPerXcd;
  if (s->opt[LPX_OPT_SWEEP_CUS] > 0) sweep_per_xcd = std::max(1, std::min(sweep_per_xcd, (int)s->opt[LPX_OPT_SWEEP_CUS] / 8));
  for (int cu = 0; cu < ncu; cu++) {
    if (cu / 8 >= per_xcd - kChainCusPerXcd) m_chain[cu / 32] |= 1u << (cu % 32);
    else if (cu / 8 < sweep_per_xcd) m_sweep[cu / 32] |= 1u << (cu % 32);
  }
  bool masked = false;
  if (s->opt[LPX_OPT_OVERLAP_MASK] != 0 && ncu >= 64) {
    masked = hipExtStreamCreateWithCUMask(&s->ov_chain, (uint32_t)m_chain.size(), m_chain.data()) == hipSuccess &&
             hipExtStreamCreateWithCUMask(&s->ov_sweep, (uint32_t)m_sweep.size(), m_sweep.data()) == hipSuccess;
    if (!masked) {  // no CU masking on this runtime: plain streams below (same results, less isolation)
      (void)hipGetLastError();
      if (s->ov_chain) { (void)hipStreamDestroy(s->ov_chain); s->ov_chain = nullptr; }
      if (s->ov_sweep) { (void)hipStreamDestroy(s->ov_sweep); s->ov_sweep = nullptr; }
    }
  }
  if (!masked) {
    int lo = 0, hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIP_TRY(hipStreamCreateWithPriority(&s->ov_chain, hipStreamNonBlocking, hi));
    HIP_TRY(hipStreamCreateWithFlags(&s->ov_sweep, hipStreamNonBlocking));
  }
  s->ov_masked = masked;
  s->ov_chain_cus = masked ? 8 * kChainCusPerXcd : ncu;
  s->ov_sweep_cus = masked ? 8 * sweep_per_xcd : ncu;
  for (int k = 0; k < 2; k++) {
    HIP_TRY(hipEventCreateWithFlags(&s->ev_ov_chain[k], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&s->ev_ov_sweep[k], hipEventDisableTiming));
  }
  for (int k = 0; k < 3; k++) HIP_TRY(hipEventCreateWithFlags(&s->ev_ov_join[k], hipEventDisableTiming));
  return 0;
}

// The fix-up of block k beside sweep k (LPX_OPT_FIXUP_SIDE): its chains read ring values only, so they need not wait for
// the sweep — only their copy into the tableau does.  mode 1: a stream with the sweep's CU mask, 2 (default): the
// decisions' CUs, 3: no mask.  Measured at cfg4 (profiles/r05_fixup_beside_the_sweep.txt): behind the sweep the fix-up
// takes 76-95 us of every 2.05 ms block, its copy kernel 19-23; the chains themselves take 120-150 us on the decisions'
// 64 CUs (which have 0.8 ms of every block to spare there) and 700-800 us squeezed in between the sweep's own waves.
int ensure_fix_side(lpx_state* s, int mode) {
  if (mode < 1 || mode > 3) return fail(LPX_BAD_ARGUMENT, "ensure_fix_side: mode");
  const size_t K = 2 * lpxk::kBlockMax;
  if (!s->R.fix_col) {
    HIP_TRY(hipMalloc((void**)&s->R.fix_col, K * (size_t)s->R.mp * sizeof(double)));
    HIP_TRY(hipMemsetAsync(s->R.fix_col, 0, K * (size_t)s->R.mp * sizeof(double), s->stream));
  }
  if (!s->R.fix_row) {
    HIP_TRY(hipMalloc((void**)&s->R.fix_row, K * (size_t)s->B.ld * sizeof(double)));
    HIP_TRY(hipMemsetAsync(s->R.fix_row, 0, K * (size_t)s->B.ld * sizeof(double), s->stream));
  }
  for (int k = 0; k < 2; k++) {
    if (!s->ev_ov_fix[k]) HIP_TRY(hipEventCreateWithFlags(&s->ev_ov_fix[k], hipEventDisableTiming));
    if (!s->ev_ov_pack[k]) HIP_TRY(hipEventCreateWithFlags(&s->ev_ov_pack[k], hipEventDisableTiming));
  }
  if (s->ov_fix[mode]) return 0;
  bool made = false;
  if (s->ov_masked && mode != 3) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, s->device));
    const int ncu = prop.multiProcessorCount, per_xcd = ncu / 8;
    const int chain_per_xcd = s->ov_chain_cus / 8, sweep_per_xcd = s->ov_sweep_cus / 8;
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    for (int cu = 0; cu < ncu; cu++) {
      const bool chain_cu = cu / 8 >= per_xcd - chain_per_xcd, sweep_cu = !chain_cu && cu / 8 < sweep_per_xcd;
      if (mode == 1 ? sweep_cu : chain_cu) mask[cu / 32] |= 1u << (cu % 32);
    }
    made = hipExtStreamCreateWithCUMask(&s->ov_fix[mode], (uint32_t)mask.size(), mask.data()) == hipSuccess;
    if (!made) { (void)hipGetLastError(); s->ov_fix[mode] = nullptr; }
  }
  if (!made) HIP_TRY(hipStreamCreateWithFlags(&s->ov_fix[mode], hipStreamNonBlocking));
  return 0;
}

// The blocked loop with the decisions one block ahead of the sweeps.  Block k's decisions (k_block_chain) read
// the tableau as it was BEFORE block k-1's sweep and see that block's pivots as pending ones, like their own; so
// sweep k-1 (out of place, buffer (k-1)&1 -> k&1) and decisions k run side by side, on disjoint XCDs:
//
//     chain stream :  chain 0 | chain 1 | chain 2 | ...          chain k   waits for sweep k-2 (its input, and the
//     sweep stream :          | sweep 0 | sweep 1 | ...                     ring half it overwrites)
//                                                                sweep k   waits for chain k (and follows sweep k-1)
// Same arithmetic, same order, per tableau entry: bit-identical to the serial forms.
static int blocked_loop_overlapped(lpx_state* s, int K, int64_t max_pivots, const lpxk::LoopStart& start) {
  if (int rc = ensure_block_ring(s)) return rc;
  if (int rc = ensure_spare_tableau(s)) return rc;
  if (int rc = ensure_overlap_streams(s)) return rc;
  launch_seed_entering(s, start);
  HIP_TRY(hipEventRecord(s->ev_ov_join[0], s->stream));
  HIP_TRY(hipStreamWaitEvent(s->ov_chain, s->ev_ov_join[0], 0));
  HIP_TRY(hipStreamWaitEvent(s->ov_sweep, s->ev_ov_join[0], 0));
  LpxCtl* h2 = s->h_snap;
  LpxCtl* d_snap = nullptr;
  HIP_TRY(hipHostGetDevicePointer((void**)&d_snap, s->h_snap, 0));
  // 32 workgroups = one per CU of the reserved XCD; more of them slow the concurrent sweep down more than they
  // speed the decisions up (cfg4: 16 / 32 / 64 workgroups -> 14.1k / 15.0k / 14.0k pivots/s)
  const int64_t work = std::max<int64_t>(s->m, s->B.ld);
  // (+1: with the first-positive rule workgroup 0 serves slots 0..255 only, see k_block_chain)
  // one row / one column per thread, as far as the reserved CUs go (two per thread, the round-2 choice, was measured
  // 5-18 % slower from 64 MiB to 512 MiB once the candidates travelled as tagged granules)
  const int chain_cap = s->ov_masked ? std::max(32, s->ov_chain_cus) : 32;
  int auto_wgs = (int)std::min<int64_t>(chain_cap, std::max<int64_t>(1, (work + 255) / 256));
  if (s->pricing == 0 && auto_wgs >= 8) auto_wgs += 1;
  // never more workgroups than the chain stream's CUs can hold at once: they spin at grid barriers
  const int chain_wgs = clamp_chain_wgs(s, s->opt[LPX_OPT_CHAIN_WGS] > 0 ? (int)s->opt[LPX_OPT_CHAIN_WGS] : auto_wgs,
                                        s->ov_chain_cus);
  const int fences = (int)s->opt[LPX_OPT_CHAIN_FENCES];
  const bool trace = s->opt[LPX_OPT_CHAIN_TRACE] != 0;
  s->info.chain_stream_masked = s->ov_masked ? 1 : 0;
  s->info.overlapped = 1;
  double* Abuf[2] = {s->B.A, s->A2};
  double* bbuf[2] = {s->B.b, s->b2};
  int64_t decided = 0;
  int nb_prev = 0, nblk = 0;
  const bool serial = s->opt[LPX_OPT_OVERLAP_SERIAL] != 0;
  int fix_mode = serial ? 0 : (int)s->opt[LPX_OPT_FIXUP_SIDE];
  // 4 = by size: beside the sweep on the decisions' CUs where the SWEEP sets the pace and the decisions have time to spare
  // (cfg4: +1.2 %); behind the sweep where the decisions set it — there anything that shares their CUs costs them more than the
  // sweep stream's idle time is worth (same box, behind / beside: 2048 x 4096 100.2k / 98.0k pivots/s, 4096 x 8192 93.6k / 91.1k,
  // 8192 x 8192 88.5k / 86.3k, cfg3 74.2k / 73.5k; profiles/r05_fixup_side_decision_bound.txt)
  if (fix_mode == 4) fix_mode = 8.0 * (double)s->m * (double)s->B.ld >= 2.0 * 1073741824.0 ? 2 : 0;
  if (fix_mode > 0 && ensure_fix_side(s, fix_mode) != 0) {   // no memory for the images / no third stream: the fix-up
    (void)hipGetLastError();                                   // stays behind the sweep (same results)
    fix_mode = 0;
  }
  // (The decision kernel's private ring copies — identity padding behind a block's last pivot, its start indices — are
  // only meaningful to the kernel FORM that wrote them.  A launch treats the other ring half as pending pivots only for
  // k > 0 of THIS call (n_old = 0 for the first block of every call), and the form is read once per launch from the
  // handle, which no option call can reach while this function runs: the two forms never meet inside one loop.)
  auto issue_block = [&](int k) -> int {  // 1: the budget is spent, nothing issued
    const int nb = block_len(K, max_pivots, decided);
    if (nb <= 0) return 1;
    const int h = k & 1;
    Buffers Brd = s->B;
    if (serial) {  // diagnostics (LPX_OVERLAP_SERIAL=1): same kernels and streams, no concurrency
      if (k >= 1) HIP_TRY(hipStreamWaitEvent(s->ov_chain, s->ev_ov_sweep[h ^ 1], 0));  // sweep k-1
      Brd.A = Abuf[h];
      Brd.b = bbuf[h];
      s->info.chain_wgs = lpxk::launch_block_chain(Brd, s->R, s->n, s->m, nb, h, h ^ 1, 0, 1, s->chain_seq++, s->pricing == 1,
                                                   chain_wgs, fences, trace, d_snap + h, s->ov_chain, nullptr, s->ev_ov_chain[h]);
    } else {
      if (k >= 2) HIP_TRY(hipStreamWaitEvent(s->ov_chain, s->ev_ov_sweep[h], 0));  // sweep k-2
      Brd.A = Abuf[k == 0 ? 0 : (k - 1) & 1];
      Brd.b = bbuf[k == 0 ? 0 : (k - 1) & 1];
      s->info.chain_wgs = lpxk::launch_block_chain(Brd, s->R, s->n, s->m, nb, h, h ^ 1, k > 0 ? nb_prev : 0, k == 0,
                                                   s->chain_seq++, s->pricing == 1, chain_wgs, fences, trace, d_snap + h,
                                                   s->ov_chain, nullptr, s->ev_ov_chain[h]);
    }
    s->chain_nb_last = nb;
    // (ev_ov_chain[h] is the launch's own stop event: an event recorded behind it would be a queue packet of its own between
    // this decision launch and the next, 4.7 us beside the sweep — scripts/micro/launch_gap.hip)
    if (max_pivots >= 0 && decided == max_pivots) {  // the budget is spent: this decision can only report the end
      decided += nb;                                 // (LIMIT / UNBOUNDED), there is nothing to sweep
      return 0;
    }
    HIP_TRY(hipStreamWaitEvent(s->ov_sweep, s->ev_ov_chain[h], 0));
    Buffers Bdst = s->B;
    Bdst.A = Abuf[h ^ 1];
    Bdst.b = bbuf[h ^ 1];
    // (the side stream: chains k start when decisions k are through — their ring half and their image half are free by
    // then: chain k itself waited for sweep k-2, which ends with the copy of those images —, and run in stream order, so
    // the b they read is the one chains k-1 wrote)
    // The sweep's pack kernel goes the same way, in front of the chains: its multiplier buffer and ticket counters are the
    // ring half's own, free since sweep k-2.
    const lpxk::FixSide side{fix_mode > 0 ? s->ov_fix[fix_mode] : nullptr, s->ev_ov_chain[h], s->ev_ov_fix[h], s->ev_ov_pack[h]};
    if (int rc = launch_sweep_profiled(s, nb, s->ov_sweep, Bdst, ring_half(s, h), Abuf[h], bbuf[h], fix_mode > 0 ? &side : nullptr,
                                       s->ev_ov_sweep[h])) return rc;   // (likewise the stop event of the block's last kernel)
    decided += nb;
    nb_prev = nb;
    nblk = k + 1;
    return 0;
  };
  int rc = issue_block(0);
  if (rc == 1) rc = 0;  // max_pivots < 0 never gets here; a zero budget issues one probing decision, so neither does 0
  for (int k = 1; rc == 0; k++) {
    const int r = issue_block(k);
    if (r != 0 && r != 1) { rc = r; break; }
    hipError_t e = hipEventSynchronize(s->ev_ov_chain[(k - 1) & 1]);
    if (e != hipSuccess) { rc = fail(LPX_DEVICE_ERROR, hipGetErrorString(e)); break; }
    if (h2[(k - 1) & 1].status != lpxk::kRunning || r == 1) break;
  }
  // join: the caller's stream continues after both; the result lives in buffer nblk & 1
  (void)hipEventRecord(s->ev_ov_join[1], s->ov_chain);
  (void)hipEventRecord(s->ev_ov_join[2], s->ov_sweep);
  (void)hipStreamWaitEvent(s->stream, s->ev_ov_join[1], 0);
  (void)hipStreamWaitEvent(s->stream, s->ev_ov_join[2], 0);
  if (nblk & 1) {  // the two allocations swap roles (both are the state's own; freed through A_base / b_base)
    std::swap(s->B.A, s->A2);
    std::swap(s->B.b, s->b2);
  }
  if (rc == 0) rc = enqueue_ctl_fetch(s);   // the loop state comes back with the same wait
  hipError_t e2 = hipStreamSynchronize(s->stream);
  if (rc == 0 && e2 != hipSuccess) rc = fail(LPX_DEVICE_ERROR, hipGetErrorString(e2));
  return rc;
}

// LPSolver.simplex's loop with K pivot decisions per pass over the tableau (bit-identical results).
// start: what the loop's first launch resets in the loop state (lpxk::LoopStart).  On return (0) the host mirror of the
// loop state is current: it was fetched behind the last launch, in front of the one wait at the end.
static int blocked_loop(lpx_state* s, int K, int64_t max_pivots, const lpxk::LoopStart& start) {
  // a budget that fits one block has nothing to run beside: the serial form below gives its decisions and its sweep
  // the whole chip (20 pivots, the bench driver's command: cfg3 24.7k vs 20.0k pivots/s, same box)
  const bool one_block = max_pivots >= 0 && max_pivots + 1 <= K;
  if (s->opt[LPX_OPT_CHAIN] != 0 && s->opt[LPX_OPT_OVERLAP] != 0 && s->row0 == 0 && s->m == s->m_global && !one_block) {
    // the overlapped form needs a second tableau: a tableau of more than half the HBM keeps the in-place form
    if (s->A2 || ensure_spare_tableau(s) == 0) return blocked_loop_overlapped(s, K, max_pivots, start);
    (void)hipGetLastError();
    if (s->A_base[1]) { (void)hipFree(s->A_base[1]); s->A_base[1] = nullptr; }
    if (s->b_base[1]) { (void)hipFree(s->b_base[1]); s->b_base[1] = nullptr; }
    s->A2 = nullptr;
    s->b2 = nullptr;
  }
  if (int rc = ensure_block_ring(s)) return rc;
  launch_seed_entering(s, start);
  hipEvent_t* evs = s->ev_batch;
  LpxCtl* h2 = s->h_snap;
  int64_t decided = 0;  // decisions issued (each either pivots or reports the end)
  // one persistent launch per block (k_block_chain) instead of three launches per decision; LPX_CHAIN=0: off
  const bool fused = s->opt[LPX_OPT_CHAIN] != 0 && s->row0 == 0 && s->m == s->m_global;
  // serial form on the handle's own stream: one row / column per thread up to 64 workgroups (the exchange grows with
  // the grid: cfg4 alone 20.3 us per decision at 64 workgroups, 20.6 at 128), never more than the device holds at once
  const int64_t work = std::max<int64_t>(s->m, s->B.ld);
  const int auto_wgs = (int)std::min<int64_t>(64, std::max<int64_t>(1, (work + 255) / 256));
  const int chain_wgs = fused ? clamp_chain_wgs(s, s->opt[LPX_OPT_CHAIN_WGS] > 0 ? (int)s->opt[LPX_OPT_CHAIN_WGS] : auto_wgs,
                                                device_cus(s))
                              : 0;
  const int fences = (int)s->opt[LPX_OPT_CHAIN_FENCES];
  const bool trace = s->opt[LPX_OPT_CHAIN_TRACE] != 0;
  s->info.chain_stream_masked = 0;
  s->info.overlapped = 0;
  LpxCtl* d_snap = nullptr;  // the pinned snapshots as the device sees them
  if (fused) HIP_TRY(hipHostGetDevicePointer((void**)&d_snap, s->h_snap, 0));
  // (the fix-up stays BEHIND the sweep here: beside it — in place that is just as valid, its chains read ring values only —
  // the sweep, which has the whole chip and is bound by HBM in this form, lost more than the fix-up takes: the driver's
  // 20-pivot command at cfg4 1.515 -> 1.59 ms per sweep, 10.5k -> 10.2k pivots/s, profiles/r05_fixup_beside_the_sweep.txt)
  auto issue_block = [&](int slot) -> int {
    const int nb = block_len(K, max_pivots, decided);
    if (fused && nb > 0) {
      s->info.chain_wgs = lpxk::launch_block_chain(s->B, s->R, s->n, s->m, nb, 0, 0, 0, 1, s->chain_seq++, s->pricing == 1,
                                                   chain_wgs, fences, trace, d_snap + slot, s->stream);
      s->chain_nb_last = nb;
    } else {
      for (int k = 0; k < nb; k++) {
        lpxk::launch_block_peek(s->B, s->R, s->n, s->m, s->row0, k, s->d_cand, s->stream);
        lpxk::launch_block_decide(s->B, s->R, s->n, s->m_global, s->d_cand, 1, k, s->stream);
        if (s->pricing == 1) lpxk::launch_entering_dantzig(s->B, s->n, false, s->stream);
      }
    }
    const bool probe_only = max_pivots >= 0 && decided == max_pivots;  // can only report the end: nothing to sweep
    decided += nb;
    if (nb > 0 && !probe_only) {
      if (int rc = launch_sweep_profiled(s, nb)) return rc;
    }
    // (the fused launch writes the snapshot itself; a block of no decisions — the budget is spent, the decision that says
    // so is in the block before — has nothing new to show and its snapshot is never looked at)
    if (!fused)
      HIP_TRY(hipMemcpyAsync(&h2[slot], s->B.ctl, sizeof(LpxCtl), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipEventRecord(evs[slot], s->stream));
    return 0;
  };
  int rc = issue_block(0);
  int cur = 0;
  while (rc == 0) {
    rc = issue_block(cur ^ 1);
    if (rc) break;
    hipError_t e = hipEventSynchronize(evs[cur]);
    if (e != hipSuccess) { rc = fail(LPX_DEVICE_ERROR, hipGetErrorString(e)); break; }
    if (h2[cur].status != lpxk::kRunning) break;
    cur ^= 1;
  }
  if (rc == 0) rc = enqueue_ctl_fetch(s);   // the loop state comes back with the same wait
  hipError_t e2 = hipStreamSynchronize(s->stream);
  if (rc == 0 && e2 != hipSuccess) rc = fail(LPX_DEVICE_ERROR, hipGetErrorString(e2));
  return rc;
}

extern "C" int lpx_state_get_block(const lpx_state* s) { return s ? choose_block(s) : -1; }

// ---- blocked pivoting on row-block shards: the host exchanges the candidate of every decision ---------------
extern "C" int lpx_shard_block_peek(lpx_state* s, double* d_candidate, int32_t slot) {
  if (!s || !d_candidate || slot < 0 || slot >= lpxk::kShardBlockMax)
    return fail(LPX_BAD_ARGUMENT, "lpx_shard_block_peek: bad argument");
  HIP_TRY(hipSetDevice(s->device));
  if (int rc = ensure_block_ring(s)) return rc;
  lpxk::launch_block_peek(s->B, s->R, s->n, s->m, s->row0, slot, d_candidate, s->stream);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int lpx_shard_block_decide(lpx_state* s, const double* d_gathered, int32_t nranks, int32_t slot) {
  if (!s || !d_gathered || nranks < 1 || slot < 0 || slot >= lpxk::kShardBlockMax)
    return fail(LPX_BAD_ARGUMENT, "lpx_shard_block_decide: bad argument");
  HIP_TRY(hipSetDevice(s->device));
  if (int rc = ensure_block_ring(s)) return rc;
  lpxk::launch_block_decide(s->B, s->R, s->n, s->m_global, d_gathered, nranks, slot, s->stream);
  if (s->pricing == 1) lpxk::launch_entering_dantzig(s->B, s->n, false, s->stream);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int lpx_shard_block_sweep(lpx_state* s, int32_t nslots) {
  if (!s || nslots < 0 || nslots > lpxk::kShardBlockMax) return fail(LPX_BAD_ARGUMENT, "lpx_shard_block_sweep: bad argument");
  HIP_TRY(hipSetDevice(s->device));
  if (nslots == 0) return 0;
  if (int rc = ensure_block_ring(s)) return rc;
  return launch_sweep_profiled(s, nslots);
}

extern "C" int lpx_state_set_block(lpx_state* s, int32_t pivots_per_sweep) {
  if (!s || pivots_per_sweep < 0 || pivots_per_sweep > lpxk::kBlockMax)
    return fail(LPX_BAD_ARGUMENT, "lpx_state_set_block: 0 (auto), 1 (off) .. 64");
  s->opt[LPX_OPT_BLOCK] = pivots_per_sweep;
  return 0;
}

extern "C" int lpx_state_set_option(lpx_state* s, int32_t key, int64_t value) {
  if (!s || key < 0 || key >= LPX_OPT_COUNT) return fail(LPX_BAD_ARGUMENT, "lpx_state_set_option: unknown option");
  const OptionSpec& sp = kOptionSpec[key];
  if (value < sp.lo || value > sp.hi) return fail(LPX_BAD_ARGUMENT, std::string("lpx_state_set_option: value out of range for ") + sp.env);
  if (key == LPX_OPT_UPDATE_U && value == 3) return fail(LPX_BAD_ARGUMENT, "lpx_state_set_option: UPDATE_U is 1, 2 or 4");
  if (key == LPX_OPT_A2_OFFSET && s->A2) return fail(LPX_BAD_ARGUMENT, "lpx_state_set_option: the second tableau exists already");
  if ((key == LPX_OPT_CHAIN_CUS || key == LPX_OPT_SWEEP_CUS) && s->ov_chain && value != s->opt[key])
    return fail(LPX_BAD_ARGUMENT, "lpx_state_set_option: the CU-masked stream pair of this handle exists already (set CHAIN_CUS / SWEEP_CUS before the first blocked loop)");
  s->opt[key] = value;
  if (key == LPX_OPT_CHAIN_FORM) s->B.chain_form = (int)value;
  if (key == LPX_OPT_FUSED) resolve_arithmetic(s);   // which of the two compilations of the kernels the launches take
  if (key == LPX_OPT_UPDATE_U || key == LPX_OPT_UPDATE_ROWS || key == LPX_OPT_NT) apply_layout_options(s);
  return 0;
}

extern "C" int lpx_state_get_option(const lpx_state* s, int32_t key, int64_t* value) {
  if (!s || !value || key < 0 || key >= LPX_OPT_COUNT) return fail(LPX_BAD_ARGUMENT, "lpx_state_get_option: bad argument");
  *value = s->opt[key];
  return 0;
}

// Placement census of the last decision launch / blocked sweep: every chain workgroup stores its XCC id + 1, sampled
// sweep workgroups OR (1 << xcc) into one word (lpx_kernels.hip); folded here, on demand.
extern "C" int lpx_state_get_info(lpx_state* s, lpx_state_info* out) {
  if (!s || !out) return fail(LPX_BAD_ARGUMENT, "lpx_state_get_info: NULL argument");
  HIP_TRY(hipSetDevice(s->device));
  s->info.block = choose_block(s);
  s->info.arith_fused = s->B.fused ? 1 : 0;
  s->info.chain_xcd_mask = 0;
  s->info.sweep_xcd_mask = 0;
  if (s->R.census) {
    HIP_TRY(hipStreamSynchronize(s->stream));
    std::vector<unsigned> h(lpxk::kChainMaxWgs + 2);
    HIP_TRY(hipMemcpy(h.data(), s->R.census, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    for (int w = 0; w < s->info.chain_wgs && w < (int)lpxk::kChainMaxWgs; w++)
      if (h[w]) s->info.chain_xcd_mask |= 1 << ((h[w] - 1) & 15);
    s->info.sweep_xcd_mask = (int32_t)h[lpxk::kChainMaxWgs];
  }
  s->info.sweep_clock_mhz = 0;
  if (s->R.clk) {   // shader clock over the last pulled sweep: s_memtime ticks per 100 MHz tick between the two probes,
                    // XCD by XCD (the counter is per XCD), median over the XCDs that took both stamps
    long long c32[32] = {};
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipMemcpy(c32, s->R.clk, sizeof c32, hipMemcpyDeviceToHost));
    std::vector<double> mhz;
    for (int x = 0; x < 8; x++) {
      const long long* c4 = c32 + 4 * x;
      if (c4[3] > c4[1] && c4[2] > c4[0] && c4[3] - c4[1] > 1000) {   // (> 10 us apart)
        const double f = 100.0 * (double)(c4[2] - c4[0]) / (double)(c4[3] - c4[1]);
        if (f > 300.0 && f < 3500.0) mhz.push_back(f);
      }
    }
    if (!mhz.empty()) {
      std::sort(mhz.begin(), mhz.end());
      s->info.sweep_clock_mhz = (int32_t)(mhz[mhz.size() / 2] + 0.5);
    }
  }
  *out = s->info;
  return 0;
}

extern "C" const char* lpx_sweep_kernel_name(int32_t code) { return lpxk::sweep_kernel_name(code); }

// diagnostic builds (LPX_SWEEP_STAMPS): raw copy of the census buffer behind the sweep's sample word
extern "C" int lpx_debug_read_census(lpx_state* s, uint32_t* out, int32_t count) {
  if (!s || !out || !s->R.census || count < 0 || count > (int)lpxk::kChainMaxWgs + 2 + 1200) return LPX_BAD_ARGUMENT;
  if (hipSetDevice(s->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return LPX_DEVICE_ERROR;
  return hipMemcpy(out, s->R.census, (size_t)count * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess ? 0 : LPX_DEVICE_ERROR;
}

// diagnostic builds (LPX_CHAIN2_LAUNCH_STAMPS): raw copy of the decision kernel's stamp buffer (16 x kBlockMax words)
extern "C" int lpx_debug_read_chain_dbg(lpx_state* s, int64_t* out, int32_t count) {
  if (!s || !out || !s->R.chain_dbg || count < 0 || count > 16 * (int)lpxk::kBlockMax) return LPX_BAD_ARGUMENT;
  if (hipSetDevice(s->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return LPX_DEVICE_ERROR;
  return hipMemcpy(out, s->R.chain_dbg, (size_t)count * sizeof(int64_t), hipMemcpyDeviceToHost) == hipSuccess ? 0 : LPX_DEVICE_ERROR;
}

// stamps per decision the decision kernel of this handle writes: k_block_chain_t 5, k_block_chain2_t 8
#ifdef LPX_CHAIN2_FINE
static int chain_trace_stride(const lpx_state* s) { return (s->B.chain_form == 1 && !(s->multi_shard && s->opt[LPX_OPT_MULTI_ONEHOP] != 0)) ? 16 : 5; }
#else
static int chain_trace_stride(const lpx_state* s) { return (s->B.chain_form == 1 && !(s->multi_shard && s->opt[LPX_OPT_MULTI_ONEHOP] != 0)) ? 8 : 5; }
#endif

extern "C" int lpx_state_read_chain_trace(lpx_state* s, int64_t* ticks, int32_t cap, int32_t* ndecisions) {
  if (!s || !ticks || cap < 0 || !ndecisions) return fail(LPX_BAD_ARGUMENT, "lpx_state_read_chain_trace: bad argument");
  *ndecisions = 0;
  if (!s->R.chain_dbg || s->opt[LPX_OPT_CHAIN_TRACE] == 0) return 0;
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  long long h[16 * lpxk::kBlockMax];
  HIP_TRY(hipMemcpy(h, s->R.chain_dbg, sizeof h, hipMemcpyDeviceToHost));
  const int nd = std::min<int>(std::min<int>(cap, s->chain_nb_last), lpxk::kBlockMax);
  // the five stamps of the documented interface; k_block_chain2 keeps eight (lpx_state_read_chain_trace_fine)
  static const int pick8[5] = {0, 2, 3, 6, 7};
  const int st = chain_trace_stride(s);
  for (int d = 0; d < nd; d++)
    for (int k = 0; k < 5; k++) ticks[5 * d + k] = h[st * d + (st >= 8 ? pick8[k] : k)];
  *ndecisions = nd;
  return 0;
}

// k_block_chain2_t's eight stamps per decision (100 MHz ticks): start, phase A's loads here, candidate published,
// every candidate read, phase B's loads here, hand-off record stored, phase B done, next entering slot known.
// *stamps = stamps per decision present (8, or 5 as lpx_state_read_chain_trace for k_block_chain_t).
extern "C" int lpx_state_read_chain_trace_fine(lpx_state* s, int64_t* ticks, int32_t cap, int32_t* ndecisions, int32_t* stamps) {
  if (!s || !ticks || cap < 0 || !ndecisions || !stamps) return fail(LPX_BAD_ARGUMENT, "lpx_state_read_chain_trace_fine: bad argument");
  *ndecisions = 0;
  *stamps = chain_trace_stride(s);
  if (!s->R.chain_dbg || s->opt[LPX_OPT_CHAIN_TRACE] == 0) return 0;
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  long long h[16 * lpxk::kBlockMax];
  HIP_TRY(hipMemcpy(h, s->R.chain_dbg, sizeof h, hipMemcpyDeviceToHost));
  const int nd = std::min<int>(std::min<int>(cap, s->chain_nb_last), lpxk::kBlockMax);
  for (int k = 0; k < *stamps * nd; k++) ticks[k] = h[k];
  *ndecisions = nd;
  return 0;
}

// ------------------------------------------------------------------------------------------------ the loop
extern "C" int lpx_simplex_loop(lpx_state* s, int64_t max_pivots, int64_t* pivots_done, int32_t* status,
                                int32_t* track_slot) {
  if (int rc = require_single(s, "lpx_simplex_loop")) return rc;
  HIP_TRY(hipSetDevice(s->device));
  const int K = choose_block(s);
  if (K >= 2) {
    // one host round trip per call: the first launch of the loop starts the loop state over on the device (no
    // read-modify-write of the host mirror in front), the last copy brings it back in front of the loop's own final wait
    lpxk::LoopStart start;
    start.reset = 1;
    start.track = track_slot ? *track_slot : -1;
    start.max_pivots = max_pivots;
    if (int rc = blocked_loop(s, K, max_pivots, start)) return rc;
    if (int r2 = check_fetched_fail_word(s)) return r2;
    if (pivots_done) *pivots_done = s->h_ctl->pivots;
    if (status) *status = s->h_ctl->status;
    if (track_slot) *track_slot = s->h_ctl->track;
    if (s->h_ctl->status == LPX_DIVIDE_BY_ZERO) return fail(LPX_DIVIDE_BY_ZERO, "pivot element is zero");
    if (s->h_ctl->status == LPX_DEVICE_ERROR)   // written by the decision kernel itself: never a result, always an error
      return fail(LPX_DEVICE_ERROR, "decision kernel: a wait between its workgroups hit its spin bound (code " +
                                        std::to_string(s->h_ctl->reserved) + ": 0 / 1 grid barrier or a candidate record, "
                                        "4 hand-off; + 1000 x decision) — were all of its " + std::to_string(s->info.chain_wgs) +
                                        " workgroups resident?  The tableau of this handle is not valid");
    return 0;
  }
  if (int rc = set_running(s, max_pivots, track_slot ? *track_slot : -1)) return rc;
  // seed: entering scan + strided column gather / partials for the first pivot
  launch_seed_entering(s);
  lpxk::launch_ratio_gather(s->B, s->m, s->row0, s->g, -1, s->stream);
  HIP_TRY(hipGetLastError());

  // batch size: ~1-2 ms of GPU work between host polls
  const double est_us = 16.0 * (double)s->m * (double)s->B.ld / 4.0e6 + 12.0;
  int batch = (int)std::max(1.0, std::min(256.0, 1500.0 / est_us));
  if (s->opt[LPX_OPT_BATCH] > 0) batch = (int)s->opt[LPX_OPT_BATCH];

  hipEvent_t* evs = s->ev_batch;
  LpxCtl* h2 = s->h_snap;  // two pinned snapshots so that batch k+1 can be in flight while k is inspected

  int64_t enqueued = 0;  // select/update pairs issued
  auto issue_batch = [&](int slot) -> int {
    int nb = batch;
    if (max_pivots >= 0) {
      // never issue more pairs than the budget allows (+1 so that the LIMIT status itself is reached)
      const int64_t room = max_pivots + 1 - enqueued;
      nb = (int)std::max<int64_t>(0, std::min<int64_t>(nb, room));
    }
    for (int k = 0; k < nb; k++) {
      lpxk::launch_select_pivot(s->B, s->n, s->m_global, s->g, -1, -1, s->stream);
      if (s->pricing == 1) lpxk::launch_entering_dantzig(s->B, s->n, false, s->stream);
      if (k == nb - 1 && max_pivots >= 0 && enqueued + k + 1 == max_pivots + 1) break;  // last one only reports LIMIT
      if (int rc = launch_update_profiled(s)) return rc;
    }
    enqueued += nb;
    HIP_TRY(hipMemcpyAsync(&h2[slot], s->B.ctl, sizeof(LpxCtl), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipEventRecord(evs[slot], s->stream));
    return 0;
  };

  int rc = issue_batch(0);
  int cur = 0;
  int result = 0;
  while (rc == 0) {
    rc = issue_batch(cur ^ 1);
    if (rc) break;
    hipError_t e = hipEventSynchronize(evs[cur]);
    if (e != hipSuccess) { rc = fail(LPX_DEVICE_ERROR, hipGetErrorString(e)); break; }
    if (h2[cur].status != lpxk::kRunning) break;
    cur ^= 1;
  }
  hipError_t e2 = hipStreamSynchronize(s->stream);
  if (rc == 0 && e2 != hipSuccess) rc = fail(LPX_DEVICE_ERROR, hipGetErrorString(e2));
  if (rc) return rc;
  if (int r2 = sync_ctl_to_host(s)) return r2;
  if (pivots_done) *pivots_done = s->h_ctl->pivots;
  if (status) *status = s->h_ctl->status;
  if (track_slot) *track_slot = s->h_ctl->track;
  if (s->h_ctl->status == LPX_DIVIDE_BY_ZERO) result = fail(LPX_DIVIDE_BY_ZERO, "pivot element is zero");
  return result;
}

// ------------------------------------------------------------------------------------------------ shards
extern "C" int lpx_shard_propose(lpx_state* s, double* d_candidate) {
  if (!s || !d_candidate) return fail(LPX_BAD_ARGUMENT, "lpx_shard_propose: NULL argument");
  HIP_TRY(hipSetDevice(s->device));
  lpxk::launch_propose(s->B, s->n, s->row0, s->m, s->g, d_candidate, s->stream);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int lpx_shard_commit(lpx_state* s, const double* d_gathered, int32_t nranks) {
  if (!s || !d_gathered || nranks < 1) return fail(LPX_BAD_ARGUMENT, "lpx_shard_commit: bad argument");
  HIP_TRY(hipSetDevice(s->device));
  lpxk::launch_commit(s->B, s->n, s->m_global, d_gathered, nranks, s->B.prow, s->B.ctl, -1, s->stream);
  if (s->pricing == 1) lpxk::launch_entering_dantzig(s->B, s->n, false, s->stream);
  if (int rc = launch_update_profiled(s)) return rc;
  return 0;
}

// Same decision step as lpx_shard_commit but without the row update: the driver issues it as the LAST step
// of a budgeted run, where the only possible outcomes are UNBOUNDED / PIVOT_LIMIT (budget exhausted).
extern "C" int lpx_shard_probe(lpx_state* s, const double* d_gathered, int32_t nranks) {
  if (!s || !d_gathered || nranks < 1) return fail(LPX_BAD_ARGUMENT, "lpx_shard_probe: bad argument");
  HIP_TRY(hipSetDevice(s->device));
  lpxk::launch_commit(s->B, s->n, s->m_global, d_gathered, nranks, s->B.prow, s->B.ctl, -1, s->stream);
  HIP_TRY(hipGetLastError());
  return 0;
}

// A CU mask cannot keep a stream off an XCD (an XCD whose bits are all clear runs unmasked: profiles/r02_cu_mask.txt),
// so "reserve_xcds" reserves one XCD's WORTH of CUs — 32 — as the same 4 CUs on each of the 8 XCDs, per unit.  The
// HBM-bound row update keeps its rate on the remaining CUs; kernels of other streams find the reserved ones free.
extern "C" int lpx_state_use_masked_stream(lpx_state* s, int32_t reserve_xcds, void** stream_out) {
  if (!s || reserve_xcds < 0 || reserve_xcds > 4) return fail(LPX_BAD_ARGUMENT, "lpx_state_use_masked_stream: bad argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  hipStream_t ns = nullptr;
  if (reserve_xcds == 0) {
    HIP_TRY(hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
  } else {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, s->device));
    const int ncu = prop.multiProcessorCount;
    std::vector<uint32_t> mask((ncu + 31) / 32, 0xFFFFFFFFu);
    if (ncu % 32) mask.back() = (1u << (ncu % 32)) - 1u;
    const int per_xcd = ncu / 8;
    for (int cu = 0; cu < ncu; cu++)  // bit i = CU i / 8 of XCD i % 8: clear the last 4 * reserve_xcds CUs of EVERY XCD
      if (cu / 8 >= per_xcd - 4 * reserve_xcds) mask[cu / 32] &= ~(1u << (cu % 32));
    HIP_TRY(hipExtStreamCreateWithCUMask(&ns, (uint32_t)mask.size(), mask.data()));
  }
  if (s->own_stream) (void)hipStreamDestroy(s->own_stream);
  s->own_stream = ns;
  s->stream = ns;
  if (stream_out) *stream_out = (void*)ns;
  return 0;
}

// ---- look-ahead pipeline (include/lpx.h "Row-block shards, look-ahead form") -------------------------------
extern "C" int lpx_shard_set_comm_stream(lpx_state* s, void* hip_stream) {
  if (!s) return fail(LPX_BAD_ARGUMENT, "NULL state");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (s->comm_stream) HIP_TRY(hipStreamSynchronize(s->comm_stream));
  s->comm_stream = (hipStream_t)hip_stream;
  return 0;
}

static hipStream_t comm_of(lpx_state* s) { return s->comm_stream ? s->comm_stream : s->stream; }

// physical tableau/b of logical slot k in the overlapped pipeline: slot 0 = where the loop started (B.A)
static Buffers slot_buffers(lpx_state* s, int k) {
  Buffers BB = s->B;
  if (s->pipeline == 2 && k == 1) { BB.A = s->A2; BB.b = s->b2; }
  return BB;
}

extern "C" int lpx_shard_set_pipeline(lpx_state* s, int32_t mode) {
  if (!s || (mode != 1 && mode != 2)) return fail(LPX_BAD_ARGUMENT, "lpx_shard_set_pipeline: mode must be 1 or 2");
  HIP_TRY(hipSetDevice(s->device));
  if (mode == 2) {
    if (int rc = ensure_spare_tableau(s)) return rc;
  }
  s->pipeline = mode;
  return 0;
}

extern "C" int lpx_shard_peek(lpx_state* s, double* d_candidate, int32_t slot, int32_t pending) {
  if (!s || !d_candidate || (slot != 0 && slot != 1)) return fail(LPX_BAD_ARGUMENT, "lpx_shard_peek: bad argument");
  HIP_TRY(hipSetDevice(s->device));
  const int prev = slot ^ 1;
  const double* prow_t = prev ? s->prow2 : s->B.prow;
  if (s->pipeline == 2) {
    // overlapped form: runs on the comm stream beside update(t) and reads the tableau update(t) READS (slot
    // prev; for the prologue, without a pending pivot, the start buffer itself); only update(t-1) must be done
    hipStream_t cs = comm_of(s);
    if (s->upd_recorded && cs != s->stream) HIP_TRY(hipStreamWaitEvent(cs, s->ev_upd, 0));
    const Buffers BB = slot_buffers(s, pending ? prev : slot);
    lpxk::launch_peek(BB, s->n, s->m, s->row0, prow_t, s->B.col[prev], s->B.col[slot],
                      pending ? &s->ring[prev] : nullptr, d_candidate, cs);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  lpxk::launch_peek(s->B, s->n, s->m, s->row0, prow_t, s->B.col[prev], s->B.col[slot],
                    pending ? &s->ring[prev] : nullptr, d_candidate, s->stream);
  HIP_TRY(hipGetLastError());
  if (comm_of(s) != s->stream) {
    HIP_TRY(hipEventRecord(s->ev_peek, s->stream));
    HIP_TRY(hipStreamWaitEvent(comm_of(s), s->ev_peek, 0));
  }
  return 0;
}

extern "C" int lpx_shard_decide(lpx_state* s, const double* d_gathered, int32_t nranks, int32_t slot) {
  if (!s || !d_gathered || nranks < 1 || (slot != 0 && slot != 1))
    return fail(LPX_BAD_ARGUMENT, "lpx_shard_decide: bad argument");
  HIP_TRY(hipSetDevice(s->device));
  double* prow = slot ? s->prow2 : s->B.prow;
  // parity = slot^1 makes k_update(slot) read col[slot], the column k_peek produced for this pivot
  lpxk::launch_commit(s->B, s->n, s->m_global, d_gathered, nranks, prow, &s->ring[slot], slot ^ 1, comm_of(s));
  if (s->pricing == 1) lpxk::launch_entering_dantzig(s->B, s->n, false, comm_of(s));
  HIP_TRY(hipGetLastError());
  if (comm_of(s) != s->stream) {
    HIP_TRY(hipEventRecord(s->ev_decide, comm_of(s)));
    HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_decide, 0));
  }
  return 0;
}

extern "C" int lpx_shard_update(lpx_state* s, int32_t slot) {
  if (!s || (slot != 0 && slot != 1)) return fail(LPX_BAD_ARGUMENT, "lpx_shard_update: bad argument");
  HIP_TRY(hipSetDevice(s->device));
  if (s->pipeline == 2) {
    const Buffers in = slot_buffers(s, slot), out = slot_buffers(s, slot ^ 1);
    if (int rc = launch_update_profiled(s, slot ? s->prow2 : s->B.prow, &s->ring[slot], &in, out.A, out.b)) return rc;
    HIP_TRY(hipEventRecord(s->ev_upd, s->stream));
    s->upd_recorded = true;
    s->settled = false;  // the tableau may now live in either buffer until lpx_shard_poll sees the final status
    return 0;
  }
  return launch_update_profiled(s, slot ? s->prow2 : s->B.prow, &s->ring[slot]);
}

// Starts (or restarts) a sharded loop: resets the replicated loop state and seeds column/partials.
extern "C" int lpx_shard_begin(lpx_state* s, int64_t max_pivots, int32_t track_slot) {
  if (!s) return fail(LPX_BAD_ARGUMENT, "lpx_shard_begin: NULL state");
  HIP_TRY(hipSetDevice(s->device));
  if (s->comm_stream) HIP_TRY(hipStreamSynchronize(s->comm_stream));
  if (!s->settled) return fail(LPX_BAD_ARGUMENT, "lpx_shard_begin: previous overlapped loop was not polled to its end");
  if (int rc = set_running(s, max_pivots, track_slot)) return rc;
  s->upd_recorded = false;
  HIP_TRY(hipMemsetAsync(s->ring, 0, 2 * sizeof(LpxCtl), s->stream));
  launch_seed_entering(s);
  lpxk::launch_ratio_gather(s->B, s->m, s->row0, s->g, -1, s->stream);
  HIP_TRY(hipGetLastError());
  if (s->pipeline == 2) {  // the prologue peek runs on the comm stream: order it behind the seed kernels
    HIP_TRY(hipEventRecord(s->ev_upd, s->stream));
    s->upd_recorded = true;
  }
  return 0;
}

extern "C" int lpx_shard_poll(lpx_state* s, int64_t* pivots_done, int32_t* status) {
  if (!s) return fail(LPX_BAD_ARGUMENT, "lpx_shard_poll: NULL state");
  HIP_TRY(hipSetDevice(s->device));
  if (s->comm_stream) HIP_TRY(hipStreamSynchronize(s->comm_stream));
  if (int rc = sync_ctl_to_host(s)) return rc;
  if (s->pipeline == 2 && !s->settled && s->h_ctl->status != lpxk::kRunning) {
    // every performed pivot moved the tableau to the other buffer; steps after the last one were no-ops
    if (s->h_ctl->pivots & 1) { std::swap(s->B.A, s->A2); std::swap(s->B.b, s->b2); }
    s->settled = true;
  }
  if (pivots_done) *pivots_done = s->h_ctl->pivots;
  if (status) *status = s->h_ctl->status;  // LPX_RUNNING (-1) while the loop is live
  return 0;
}

// ------------------------------------------------------------------------------------------------ read-back
extern "C" int lpx_state_read(lpx_state* s, double* A, int64_t lda, double* b, double* c, double* v, int32_t* perm) {
  if (!s) return fail(LPX_BAD_ARGUMENT, "NULL state");
  HIP_TRY(hipSetDevice(s->device));
  if (A && s->m > 0 && s->n > 0) {
    if (lda < s->n) return fail(LPX_BAD_ARGUMENT, "lpx_state_read: lda < n");
    HIP_TRY(hipMemcpy2DAsync(A, lda * sizeof(double), s->B.A, s->B.ld * sizeof(double), (size_t)s->n * sizeof(double),
                             (size_t)s->m, hipMemcpyDeviceToHost, s->stream));
  }
  if (b && s->m > 0) HIP_TRY(hipMemcpyAsync(b, s->B.b, (size_t)s->m * sizeof(double), hipMemcpyDeviceToHost, s->stream));
  if (c && s->n > 0) HIP_TRY(hipMemcpyAsync(c, s->B.c, (size_t)s->n * sizeof(double), hipMemcpyDeviceToHost, s->stream));
  if (perm)
    HIP_TRY(hipMemcpyAsync(perm, s->B.perm, ((size_t)s->n + s->m_global) * sizeof(int32_t), hipMemcpyDeviceToHost,
                           s->stream));
  if (int rc = sync_ctl_to_host(s)) return rc;
  if (v) *v = s->h_ctl->v;
  return 0;
}

extern "C" int lpx_state_checksum(lpx_state* s, uint64_t out[3]) {
  if (!s || !out) return fail(LPX_BAD_ARGUMENT, "NULL argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipMemsetAsync(s->d_sum, 0, 4 * sizeof(unsigned long long), s->stream));
  lpxk::launch_checksum(s->B, s->m, s->n, s->row0, s->d_sum, s->stream);
  HIP_TRY(hipGetLastError());
  unsigned long long h[4];
  HIP_TRY(hipMemcpyAsync(h, s->d_sum, sizeof h, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  out[0] = h[0]; out[1] = h[1]; out[2] = h[2];
  return 0;
}

extern "C" int lpx_transpose(int32_t m, int32_t n, const double* A, int64_t lda, double* At, int64_t ldat, int device) {
  if (m < 0 || n < 0 || (m > 0 && n > 0 && (!A || !At)) || lda < n || ldat < m)
    return fail(LPX_BAD_ARGUMENT, "lpx_transpose: bad argument");
  if (m == 0 || n == 0) return 0;
  HIP_TRY(hipSetDevice(device));
  double *dA = nullptr, *dAt = nullptr;
  HIP_TRY(hipMalloc((void**)&dA, (size_t)m * n * sizeof(double)));
  hipError_t e = hipMalloc((void**)&dAt, (size_t)m * n * sizeof(double));
  if (e != hipSuccess) { (void)hipFree(dA); return fail(LPX_DEVICE_ERROR, "hipMalloc failed"); }
  int rc = 0;
  do {
    if (hipMemcpy2D(dA, (size_t)n * 8, A, (size_t)lda * 8, (size_t)n * 8, m, hipMemcpyHostToDevice) != hipSuccess) { rc = 1; break; }
    lpxk::launch_transpose(dA, n, dAt, m, m, n, nullptr);
    if (hipGetLastError() != hipSuccess) { rc = 1; break; }
    if (hipMemcpy2D(At, (size_t)ldat * 8, dAt, (size_t)m * 8, (size_t)m * 8, n, hipMemcpyDeviceToHost) != hipSuccess) { rc = 1; break; }
  } while (0);
  (void)hipFree(dA);
  (void)hipFree(dAt);
  return rc ? fail(LPX_DEVICE_ERROR, "lpx_transpose: HIP error") : 0;
}

// ------------------------------------------------------------------------------------------------ internals
// used by lpx_solver.cpp (same shared object, not exported through lpx.h)
namespace lpx_internal {

int alloc(int32_t m, int32_t n, int32_t n_cap, int device, lpx_state** out) { return alloc_state(m, n, n_cap, 0, m, device, out); }
void destroy(lpx_state* s) { free_state(s); }
lpxk::Buffers& buffers(lpx_state* s) { return s->B; }
hipStream_t stream(lpx_state* s) { return s->stream; }
LpxCtl* host_ctl(lpx_state* s) { return s->h_ctl; }
int pull_ctl(lpx_state* s) { return sync_ctl_to_host(s); }
int push(lpx_state* s) { return push_ctl(s); }
void reset_ctl(lpx_state* s, double v) { init_ctl(s, v); }
void set_n(lpx_state* s, int32_t n) { s->n = n; }
int32_t get_n(lpx_state* s) { return s->n; }
int32_t get_m(lpx_state* s) { return s->m; }
int set_error(int status, const char* msg) { return fail(status, msg); }

}  // namespace lpx_internal
