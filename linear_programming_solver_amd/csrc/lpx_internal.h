// Internal interface of the host engine: the handle behind include/lpx.h and the helpers that lpx_multi.cpp (row-block
// shards on several devices, one process) shares with lpx_engine.cpp.  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/lpx.h"
#include "lpx_kernels.h"

using lpxk::Buffers;
using lpxk::Geometry;
using lpxk::LpxCtl;
using lpxk::RatioRow;

int fail(int status, const std::string& msg);

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return fail(LPX_DEVICE_ERROR, std::string(#expr) + ": " + hipGetErrorString(_e));       \
  } while (0)

// The multi-GPU entry points set the current HIP device shard by shard; the caller (torch, a JVM thread) gets its own
// device back when the call returns.
struct DeviceRestore {
  int d = -1;
  DeviceRestore() { if (hipGetDevice(&d) != hipSuccess) { d = -1; (void)hipGetLastError(); } }
  ~DeviceRestore() { if (d >= 0) (void)hipSetDevice(d); }
};

// ------------------------------------------------------------------------------------------------ state
struct lpx_state {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  int32_t m = 0;         // local rows
  int32_t n = 0;         // nonbasic slots (columns in use)
  int32_t row0 = 0, m_global = 0;
  int32_t n_cap = 0;     // n the buffers were sized for (phase 1 allocates n+1 and later shrinks n)
  Buffers B{};
  Geometry g{};
  bool nontemporal = false;
  int pricing = 0;                  // 0 = reference rule (first positive), 1 = Dantzig (opt-in extension)
  int64_t opt[LPX_OPT_COUNT] = {};  // lpx_option values (include/lpx.h); initial values: env_defaults()
  lpx_state_info info{};            // what the last loop actually did (lpx_state_get_info)
  int chain_nb_last = 0;            // decisions of the last k_block_chain launch (chain trace)
  bool peer_written = false;        // a shard of an lpx_multi on several devices: buffers that peers store into
  bool multi_shard = false;     // a shard of an lpx_multi (also with a single shard / all shards on one GPU)
                                    // (pivot-row ring, mailbox, arrival words) are allocated fine-grained
  lpxk::BlockRing R{};
  int chain_seq = 0;                // k_block_chain launches so far (its two barrier counters alternate)
  // overlapped blocked loop: decisions of block k+1 (one reserved XCD) beside the sweep of block k (the other 7)
  hipStream_t ov_chain = nullptr, ov_sweep = nullptr;
  bool ov_masked = false;           // the pair was created with CU masks (else: plain streams, chain at high priority)
  int ov_chain_cus = 0;             // CUs the chain stream may use
  int ov_sweep_cus = 0;           // CUs of the sweep stream's mask (LPX_OPT_SWEEP_CUS)
  hipEvent_t ev_ov_chain[2] = {nullptr, nullptr}, ev_ov_sweep[2] = {nullptr, nullptr}, ev_ov_join[3] = {nullptr, nullptr, nullptr};
  // the fix-up's chains beside the sweep (LPX_OPT_FIXUP_SIDE = 1..3: on the sweep's CUs, the decisions' CUs, all CUs)
  hipStream_t ov_fix[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_ov_fix[2] = {nullptr, nullptr};     // chains of block k done (the copy kernel waits for it)
  hipEvent_t ev_ov_pack[2] = {nullptr, nullptr};    // multipliers of block k packed on the side stream (sweep k waits for it)
  double* d_cand = nullptr;         // candidate record of the single-GPU blocked loop (8 + n doubles)
  LpxCtl* h_ctl = nullptr;          // pinned mirror
  LpxCtl* h_snap = nullptr;         // 2 pinned snapshots for the batched loop (batch k+1 in flight while k is read)
  hipEvent_t ev_batch[2] = {nullptr, nullptr};
  // look-ahead pipeline of the sharded loop: parameter ring, second pivot-row buffer, second stream
  LpxCtl* ring = nullptr;           // 2 device blocks
  double* prow2 = nullptr;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_peek = nullptr, ev_decide = nullptr, ev_upd = nullptr;
  // pipeline = 2: fully overlapped form — out-of-place row update between two tableau buffers so that the peek
  // of pivot t+1 (comm stream) reads the un-updated tableau while update(t) streams (main stream)
  int pipeline = 1;
  double* A2 = nullptr;             // spare tableau / b of the out-of-place update (B.A/A2 and B.b/b2 swap roles)
  double* A_base[2] = {nullptr, nullptr};   // the two hipMalloc'ed tableau allocations, for hipFree
  double* b_base[2] = {nullptr, nullptr};
  double* b2 = nullptr;
  bool upd_recorded = false;
  bool settled = true;              // B.A/B.b point at the buffer that holds the current tableau
  unsigned long long* d_sum = nullptr;
  // row-update profiling (HIP events on `stream`)
  int prof = 0;                     // 0 = off, N = bracket every N-th row-update launch with events
  int64_t prof_seq = 0;
  std::vector<hipEvent_t> ev;       // pairs
  size_t ev_used = 0;
  int64_t prof_launches = 0;
  double prof_ms = 0.0;
};


int sync_ctl_to_host(lpx_state* s);
int push_ctl(lpx_state* s);
void free_state(lpx_state* s);
int alloc_state(int32_t m_local, int32_t n, int32_t n_cap, int32_t row0, int32_t m_global, int device, lpx_state** out);
int upload_common(lpx_state* s, const double* A, int64_t lda, const double* b, const double* c, double v,
                  const int32_t* perm, hipMemcpyKind kind);
void init_ctl(lpx_state* s, double v);
int set_running(lpx_state* s, int64_t max_pivots, int32_t track);
void resolve_arithmetic(lpx_state* s);      // LPX_OPT_FUSED (0 / 1 / 2 = by size) -> Buffers::fused
int ensure_block_ring(lpx_state* s);
int ensure_spare_tableau(lpx_state* s);     // the second tableau / b of the out-of-place forms
int ensure_overlap_streams(lpx_state* s);   // ov_chain / ov_sweep (CU-masked when possible) and their events
// hipMalloc, or fine-grained device memory when peers store into the buffer while a kernel of this device reads it
hipError_t peer_visible_malloc(const lpx_state* s, void** ptr, size_t bytes);
int32_t state_n(const lpx_state* s);
int choose_block(const lpx_state* s);
int block_len(int K, int64_t max_pivots, int64_t decided);
int clamp_chain_wgs(lpx_state* s, int want, int cus);
int device_cus(const lpx_state* s);
int launch_sweep_profiled(lpx_state* s, int K, hipStream_t stream, const Buffers& B, const lpxk::BlockRing& R,
                          const double* A_src, const double* b_src, const lpxk::FixSide* side = nullptr,
                          hipEvent_t stop = nullptr);   // stop: signalled when everything the call enqueued on `stream` is through
int ensure_fix_side(lpx_state* s, int mode);   // ov_fix[mode], ev_ov_fix and the ring's images (R.fix_col / R.fix_row)
lpxk::BlockRing ring_half(const lpx_state* s, int h);
void launch_seed_entering(lpx_state* s, const lpxk::LoopStart& start = lpxk::LoopStart{});   // start.reset: also starts the loop state over
int launch_update_profiled(lpx_state* s, const double* prow = nullptr, const LpxCtl* up = nullptr,
                           const Buffers* Bin = nullptr, double* A_out = nullptr, double* b_out = nullptr);

// ---- lpx_multi (lpx_multi.cpp) hooks used by lpx_solve_multi (lpx_solver.cpp)
int multi_create(int32_t m, int32_t n, int32_t n_cap, const double* A, int64_t lda, const double* b, const double* c,
                 double v, const int32_t* perm, const int32_t* devices, int32_t n_dev, lpx_multi** out);
namespace lpx_internal {
int multi_shards(lpx_multi* M);
lpx_state* multi_shard(lpx_multi* M, int r);
int multi_device(lpx_multi* M, int r);
int32_t multi_row_start(lpx_multi* M, int r);
int multi_owner(lpx_multi* M, int32_t row);
void multi_set_n(lpx_multi* M, int32_t n);
}  // namespace lpx_internal
