// lpx_multi: ONE handle over several GPUs of a node (include/lpx.h "Several GPUs of one node behind ONE handle").
//
// The tableau is cut into contiguous row blocks — shard r holds rows [r*m/G, (r+1)*m/G), the partition the
// reference's pivotConcurrently uses for its row phase (LPState.java:222-223); c, v, perm and the loop state are
// replicated and updated identically on every device.  One process, one host thread: per block of K pivot decisions
// the host launches the persistent decision kernel (k_block_chain_t<true>, lpx_kernels.hip) on every device — the
// kernels exchange the minimum-ratio candidates and the normalised pivot row among themselves by direct stores into
// peer memory over xGMI — and then one K-fold sweep of each device's own rows.  The host takes no decision: it polls
// the replicated loop state that every device writes into pinned memory when its launch ends.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lpx.h"
#include "lpx_internal.h"
#include "lpx_kernels.h"

struct lpx_multi {
  int n_dev = 0;
  int32_t m = 0, n = 0;
  std::vector<int> device;            // HIP ordinal of shard r (ordinals may repeat: shards sharing a GPU)
  std::vector<lpx_state*> sh;
  std::vector<int32_t> row_start;     // n_dev + 1
  LpxCtl* h_snap = nullptr;           // pinned, [n_dev][2]: the loop state each device publishes per block
  std::vector<LpxCtl*> d_snap;        // the same memory as device r sees it
  std::vector<hipEvent_t> ev[2];
  int seq = 0;                        // decision-kernel launches so far (the SAME number on every shard: tags)
  bool distinct_devices = false;
  bool fences_chosen = false;         // lpx_multi_set_option(LPX_OPT_CHAIN_FENCES) was called: honour it
  unsigned spin_max = 0;              // bound of the waits between devices (polls), sized from the warm launches below
  double first_launch_us = 0.0;       // slowest launch + completion of a full-grid decision kernel that decides nothing
};

static int shard_of(const lpx_multi* M, int32_t row) {
  for (int r = 0; r < M->n_dev; r++)
    if (row < M->row_start[r + 1]) return r;
  return M->n_dev - 1;
}

static void multi_free(lpx_multi* M) {
  if (!M) return;
  for (int r = 0; r < (int)M->sh.size(); r++) {
    if (!M->sh[r]) continue;
    (void)hipSetDevice(M->device[r]);
    for (int k = 0; k < 2; k++)
      if (r < (int)M->ev[k].size() && M->ev[k][r]) (void)hipEventDestroy(M->ev[k][r]);
    free_state(M->sh[r]);
  }
  if (M->h_snap) (void)hipHostFree(M->h_snap);
  delete M;
}

// Creates the shards for an m x n tableau whose buffers hold n_cap >= n columns (phase 1 allocates n + 1 once).
// c may be NULL (the objective is set later, multi_set_objective).
int multi_create(int32_t m, int32_t n, int32_t n_cap, const double* A, int64_t lda, const double* b, const double* c,
                 double v, const int32_t* perm, const int32_t* devices, int32_t n_dev, lpx_multi** out) {
  if (!out) return fail(LPX_BAD_ARGUMENT, "lpx_multi_create: out is NULL");
  *out = nullptr;
  if (m < 0 || n < 0 || n_cap < n || n_dev < 1 || n_dev > LPX_MAX_DEVICES || !devices || n_dev > std::max(m, 1))
    return fail(LPX_BAD_ARGUMENT, "lpx_multi_create: bad dimensions or device list");
  if ((m > 0 && n > 0 && (!A || lda < n)) || (m > 0 && !b)) return fail(LPX_BAD_ARGUMENT, "lpx_multi_create: NULL array");
  int ndev_visible = 0;
  HIP_TRY(hipGetDeviceCount(&ndev_visible));
  for (int r = 0; r < n_dev; r++)
    if (devices[r] < 0 || devices[r] >= ndev_visible) return fail(LPX_BAD_ARGUMENT, "lpx_multi_create: no such device");
  lpx_multi* M = new lpx_multi();
  M->n_dev = n_dev; M->m = m; M->n = n;
  M->device.assign(devices, devices + n_dev);
  M->sh.assign(n_dev, nullptr);
  M->row_start.resize(n_dev + 1);
  for (int r = 0; r <= n_dev; r++) M->row_start[r] = (int32_t)(((int64_t)r * m) / n_dev);   // LPState.java:222-223
  for (int a = 0; a < n_dev; a++)
    for (int b2 = 0; b2 < n_dev; b2++)
      if (devices[a] != devices[b2]) M->distinct_devices = true;
  // peer access between every pair of distinct devices of the set
  for (int a = 0; a < n_dev; a++) {
    for (int b2 = 0; b2 < n_dev; b2++) {
      if (devices[a] == devices[b2]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, devices[a], devices[b2]) != hipSuccess || !can) {
        multi_free(M);
        return fail(LPX_DEVICE_ERROR, "lpx_multi_create: devices " + std::to_string(devices[a]) + " and " +
                                          std::to_string(devices[b2]) + " cannot access each other's memory");
      }
      if (hipSetDevice(devices[a]) != hipSuccess) { multi_free(M); return fail(LPX_DEVICE_ERROR, "hipSetDevice failed"); }
      const hipError_t e = hipDeviceEnablePeerAccess(devices[b2], 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
        multi_free(M);
        return fail(LPX_DEVICE_ERROR, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
      }
      (void)hipGetLastError();
    }
  }
  std::vector<double> czero((size_t)std::max(n, 1), 0.0);
  for (int r = 0; r < n_dev; r++) {
    const int32_t r0 = M->row_start[r], ml = M->row_start[r + 1] - r0;
    lpx_state* s = nullptr;
    if (int rc = alloc_state(ml, n, n_cap, r0, m, devices[r], &s)) { multi_free(M); return rc; }
    M->sh[r] = s;
    s->peer_written = M->distinct_devices;
    s->multi_shard = true;
    resolve_arithmetic(s);   // (the by-size choice of the fused arithmetic is for unsharded handles)
    if (int rc = upload_common(s, A ? A + (int64_t)r0 * lda : nullptr, lda, b ? b + r0 : nullptr, c ? c : czero.data(), v,
                               perm, hipMemcpyHostToDevice)) {
      multi_free(M);
      return rc;
    }
  }
  if (hipHostMalloc((void**)&M->h_snap, (size_t)n_dev * 2 * sizeof(LpxCtl), hipHostMallocPortable | hipHostMallocMapped) !=
      hipSuccess) {
    multi_free(M);
    return fail(LPX_DEVICE_ERROR, "lpx_multi_create: pinned snapshot allocation failed");
  }
  memset(M->h_snap, 0, (size_t)n_dev * 2 * sizeof(LpxCtl));
  M->d_snap.assign(n_dev, nullptr);
  for (int k = 0; k < 2; k++) M->ev[k].assign(n_dev, nullptr);
  for (int r = 0; r < n_dev; r++) {
    if (hipSetDevice(devices[r]) != hipSuccess ||
        hipHostGetDevicePointer((void**)&M->d_snap[r], M->h_snap + 2 * r, 0) != hipSuccess ||
        hipEventCreateWithFlags(&M->ev[0][r], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&M->ev[1][r], hipEventDisableTiming) != hipSuccess) {
      multi_free(M);
      return fail(LPX_DEVICE_ERROR, "lpx_multi_create: snapshot mapping / event creation failed");
    }
  }
  *out = M;
  return 0;
}

extern "C" int lpx_multi_create(int32_t m, int32_t n, const double* A, int64_t lda, const double* b, const double* c,
                                double v, const int32_t* perm, const int32_t* devices, int32_t n_dev, lpx_multi** out) {
  DeviceRestore keep_device;
  if (n > 0 && !c) return fail(LPX_BAD_ARGUMENT, "lpx_multi_create: NULL array");
  return multi_create(m, n, n, A, lda, b, c, v, perm, devices, n_dev, out);
}

extern "C" void lpx_multi_destroy(lpx_multi* M) {
  DeviceRestore keep_device;
  multi_free(M);
}

extern "C" int lpx_multi_set_option(lpx_multi* M, int32_t key, int64_t value) {
  if (!M) return fail(LPX_BAD_ARGUMENT, "lpx_multi_set_option: NULL handle");
  DeviceRestore keep_device;
  if (key == LPX_OPT_CHAIN_FENCES) M->fences_chosen = true;
  for (lpx_state* s : M->sh)
    if (int rc = lpx_state_set_option(s, key, value)) return rc;
  return 0;
}

extern "C" int lpx_multi_set_pricing(lpx_multi* M, int32_t pricing) {
  if (!M) return fail(LPX_BAD_ARGUMENT, "lpx_multi_set_pricing: NULL handle");
  for (lpx_state* s : M->sh)
    if (int rc = lpx_state_set_pricing(s, pricing)) return rc;
  return 0;
}

extern "C" int lpx_multi_get_entering(lpx_multi* M, int32_t* entering) {
  DeviceRestore keep_device;
  if (!M || !entering) return fail(LPX_BAD_ARGUMENT, "lpx_multi_get_entering: NULL argument");
  return lpx_get_entering(M->sh[0], entering);   // c is replicated
}

extern "C" int lpx_multi_get_leaving(lpx_multi* M, int32_t entering, int32_t* leaving, double* ratio) {
  DeviceRestore keep_device;
  if (!M || !leaving) return fail(LPX_BAD_ARGUMENT, "lpx_multi_get_leaving: NULL argument");
  // the strict '<' scan from row 0 upwards (LPState.java:292-303) = lexicographic minimum of (ratio, global row)
  double best = lpxk::kInf;
  int32_t row = -1;
  for (lpx_state* s : M->sh) {
    int32_t l = -1;
    double r = lpxk::kInf;
    if (int rc = lpx_get_leaving(s, entering, &l, &r)) return rc;
    if (l >= 0 && r < best) { best = r; row = l; }   // shards ascend: '<' keeps the lowest row among equal ratios
  }
  *leaving = row;
  if (ratio) *ratio = best;
  return 0;
}

// What shard r's decision kernel needs to know about its peers.
static lpxk::MgPeers peers_of(const lpx_multi* M, int r) {
  lpxk::MgPeers P{};
  P.n_dev = M->n_dev; P.dev = r; P.row0 = M->row_start[r]; P.m_global = M->m;
  for (int d = 0; d < M->n_dev; d++) {
    P.mail[d] = M->sh[d]->R.mg_mail;
    P.prow[d] = M->sh[d]->R.prow;
    P.arrive[d] = M->sh[d]->R.mg_arrive;
    P.candrow[d] = M->sh[d]->R.mg_candrow;
    P.arrive2[d] = M->sh[d]->R.mg_arrive2;
  }
  P.spin_max = M->spin_max;
  P.onehop = M->sh[0]->opt[LPX_OPT_MULTI_ONEHOP] != 0 && M->sh[0]->R.mg_candrow != nullptr;
  M->sh[r]->info.multi_onehop = P.onehop;
  return P;
}

// pivot(entering, leaving) on shards: the owner of the leaving row hands its raw row to every shard (through the
// host: this is the forced first pivot / the degenerate pivot of phase 1, LPSolver.java:138, :195, not the loop),
// then every shard finishes the pivot identically (k_commit) and updates its own rows.
extern "C" int lpx_multi_pivot(lpx_multi* M, int32_t entering, int32_t leaving) {
  DeviceRestore keep_device;
  if (!M) return fail(LPX_BAD_ARGUMENT, "lpx_multi_pivot: NULL handle");
  const int32_t n = state_n(M->sh[0]);
  if (!(entering >= 0 && entering < n) || !(leaving >= 0 && leaving < M->m))
    return fail(LPX_BAD_ARGUMENT, "lpx_multi_pivot: index out of range");
  const int o = shard_of(M, leaving);
  lpx_state* so = M->sh[o];
  std::vector<double> rec((size_t)LPX_CAND_HEADER + so->B.ld, 0.0);
  HIP_TRY(hipSetDevice(M->device[o]));
  HIP_TRY(hipStreamSynchronize(so->stream));
  const int32_t ll = leaving - so->row0;
  HIP_TRY(hipMemcpy(rec.data() + LPX_CAND_HEADER, so->B.A + (int64_t)ll * so->B.ld, (size_t)n * sizeof(double),
                    hipMemcpyDeviceToHost));
  double bl = 0.0;
  HIP_TRY(hipMemcpy(&bl, so->B.b + ll, sizeof(double), hipMemcpyDeviceToHost));
  rec[0] = 0.0; rec[1] = (double)entering; rec[2] = 0.0; rec[3] = (double)leaving; rec[4] = bl;
  for (int r = 0; r < M->n_dev; r++) {
    lpx_state* s = M->sh[r];
    HIP_TRY(hipSetDevice(M->device[r]));
    if (int rc = ensure_block_ring(s)) return rc;   // d_cand lives there
    if (int rc = sync_ctl_to_host(s)) return rc;
    LpxCtl& c = *s->h_ctl;
    c.status = lpxk::kRunning; c.do_update = 0; c.pivots = 0; c.max_pivots = -1; c.track = -1;
    c.e_min = INT32_MAX; c.ticket = 0; c.e_next = entering;
    if (int rc = push_ctl(s)) return rc;
    lpxk::launch_ratio_gather(s->B, s->m, s->row0, s->g, entering, s->stream);   // column `entering` -> col[parity]
    HIP_TRY(hipMemcpyAsync(s->d_cand, rec.data(), ((size_t)LPX_CAND_HEADER + n) * sizeof(double), hipMemcpyHostToDevice,
                           s->stream));
    lpxk::launch_commit(s->B, n, s->m_global, s->d_cand, 1, s->B.prow, s->B.ctl, -1, s->stream);
    if (int rc = launch_update_profiled(s)) return rc;
  }
  int result = 0;
  for (int r = 0; r < M->n_dev; r++) {
    HIP_TRY(hipSetDevice(M->device[r]));
    if (int rc = sync_ctl_to_host(M->sh[r])) return rc;   // also drains the stream (rec must outlive the copies)
    if (M->sh[r]->h_ctl->status == LPX_DIVIDE_BY_ZERO) result = fail(LPX_DIVIDE_BY_ZERO, "lpx_multi_pivot: pivot element is zero");
  }
  return result;
}

// After a loop: the replicated loop state must agree on every device; hand it to the caller.
static int multi_finish_loop(lpx_multi* M, int64_t* pivots_done, int32_t* status, int32_t* track_slot) {
  const int G = M->n_dev;
  for (int r = 0; r < G; r++) {
    HIP_TRY(hipSetDevice(M->device[r]));
    if (int r2 = sync_ctl_to_host(M->sh[r])) return r2;
  }
  const LpxCtl& c0 = *M->sh[0]->h_ctl;
  for (int r = 0; r < G; r++)
    if (M->sh[r]->h_ctl->status == LPX_DEVICE_ERROR)
      return fail(LPX_DEVICE_ERROR, "decision kernel of shard " + std::to_string(r) + ": a wait hit its spin bound (code " +
                  std::to_string(M->sh[r]->h_ctl->reserved) + ": 1 grid barrier, 2 a peer's candidate, 3 the owner's row, "
                  "4 hand-off; + 16 x lane + 1000 x decision).  Shards that share a GPU need one hardware queue per "
                  "stream: GPU_MAX_HW_QUEUES");
  for (int r = 1; r < G; r++) {
    const LpxCtl& cr = *M->sh[r]->h_ctl;
    if (cr.status != c0.status || cr.pivots != c0.pivots || cr.track != c0.track || memcmp(&cr.v, &c0.v, sizeof(double)) != 0)
      return fail(LPX_DEVICE_ERROR, "lpx_multi_simplex_loop: the replicated loop state diverged between devices");
  }
  if (pivots_done) *pivots_done = c0.pivots;
  if (status) *status = c0.status;
  if (track_slot) *track_slot = c0.track;
  if (c0.status == LPX_DIVIDE_BY_ZERO) return fail(LPX_DIVIDE_BY_ZERO, "pivot element is zero");
  return 0;
}

static int multi_loop_overlapped(lpx_multi* M, int K, int64_t max_pivots, int want_wgs, int fences, bool trace, int dantzig,
                                 int64_t* pivots_done, int32_t* status, int32_t* track_slot) {
  const int G = M->n_dev;
  // the decision kernels' grid: the same on every device, within what the CUs of each shard's decision stream hold
  // (shards that share a GPU share those CUs)
  int wgs = want_wgs;
  for (int r = 0; r < G; r++) {
    int same = 0;
    for (int q = 0; q < G; q++) same += M->device[q] == M->device[r];
    HIP_TRY(hipSetDevice(M->device[r]));
    wgs = std::min(wgs, clamp_chain_wgs(M->sh[r], want_wgs, std::max(1, M->sh[r]->ov_chain_cus / same)));
  }
  std::vector<double*> Abuf0(G), Abuf1(G), bbuf0(G), bbuf1(G);
  for (int r = 0; r < G; r++) {
    lpx_state* s = M->sh[r];
    HIP_TRY(hipSetDevice(M->device[r]));
    s->info.chain_wgs = wgs; s->info.overlapped = 1; s->info.chain_stream_masked = s->ov_masked ? 1 : 0;
    Abuf0[r] = s->B.A; Abuf1[r] = s->A2; bbuf0[r] = s->B.b; bbuf1[r] = s->b2;
    // the seed (entering scan) ran on the shard's own stream: both work streams continue after it
    HIP_TRY(hipEventRecord(s->ev_ov_join[0], s->stream));
    HIP_TRY(hipStreamWaitEvent(s->ov_chain, s->ev_ov_join[0], 0));
    HIP_TRY(hipStreamWaitEvent(s->ov_sweep, s->ev_ov_join[0], 0));
  }
  int64_t decided = 0;
  int nb_prev = 0, nblk = 0;
  auto issue_block = [&](int k) -> int {   // 1: the budget is spent, nothing issued
    const int nb = block_len(K, max_pivots, decided);
    if (nb <= 0) return 1;
    const int h = k & 1;
    const bool probe_only = max_pivots >= 0 && decided == max_pivots;   // can only report the end: nothing to sweep
    for (int r = 0; r < G; r++) {
      lpx_state* s = M->sh[r];
      HIP_TRY(hipSetDevice(M->device[r]));
      if (k >= 2) HIP_TRY(hipStreamWaitEvent(s->ov_chain, s->ev_ov_sweep[h], 0));   // sweep k-2: its input, its ring half
      Buffers Brd = s->B;
      const int src = k == 0 ? 0 : (k - 1) & 1;
      Brd.A = src ? Abuf1[r] : Abuf0[r];
      Brd.b = src ? bbuf1[r] : bbuf0[r];
      lpxk::MgPeers P = peers_of(M, r);
      P.mail_slot0 = (int)(decided & 1);
      lpxk::launch_block_chain(Brd, s->R, s->n, s->m, nb, h, h ^ 1, k > 0 ? nb_prev : 0, k == 0, M->seq, dantzig, wgs,
                               fences, trace, M->d_snap[r] + h, s->ov_chain, &P, s->ev_ov_chain[h]);   // (its own stop event)
      s->chain_nb_last = nb;
      HIP_TRY(hipGetLastError());
    }
    M->seq++;
    if (!probe_only) {
      for (int r = 0; r < G; r++) {
        lpx_state* s = M->sh[r];
        HIP_TRY(hipSetDevice(M->device[r]));
        HIP_TRY(hipStreamWaitEvent(s->ov_sweep, s->ev_ov_chain[h], 0));
        Buffers Bdst = s->B;
        Bdst.A = h ? Abuf0[r] : Abuf1[r];   // buffer h -> buffer h ^ 1
        Bdst.b = h ? bbuf0[r] : bbuf1[r];
        if (int rc = launch_sweep_profiled(s, nb, s->ov_sweep, Bdst, ring_half(s, h), h ? Abuf1[r] : Abuf0[r],
                                           h ? bbuf1[r] : bbuf0[r], nullptr, s->ev_ov_sweep[h]))
          return rc;
      }
      nb_prev = nb;
      nblk = k + 1;
    }
    decided += nb;
    return 0;
  };
  int rc = issue_block(0);
  if (rc == 1) rc = 0;
  for (int k = 1; rc == 0; k++) {
    const int r1 = issue_block(k);
    if (r1 != 0 && r1 != 1) { rc = r1; break; }
    bool running = true;
    for (int r = 0; r < G && rc == 0; r++) {
      const hipError_t e = hipEventSynchronize(M->sh[r]->ev_ov_chain[(k - 1) & 1]);
      if (e != hipSuccess) { rc = fail(LPX_DEVICE_ERROR, hipGetErrorString(e)); break; }
      if (M->h_snap[2 * r + ((k - 1) & 1)].status != lpxk::kRunning) running = false;
    }
    if (rc || !running || r1 == 1) break;
  }
  // join: every shard's own stream continues after both work streams; the result lives in buffer nblk & 1
  for (int r = 0; r < G; r++) {
    lpx_state* s = M->sh[r];
    (void)hipSetDevice(M->device[r]);
    (void)hipEventRecord(s->ev_ov_join[1], s->ov_chain);
    (void)hipEventRecord(s->ev_ov_join[2], s->ov_sweep);
    (void)hipStreamWaitEvent(s->stream, s->ev_ov_join[1], 0);
    (void)hipStreamWaitEvent(s->stream, s->ev_ov_join[2], 0);
    if (nblk & 1) {
      std::swap(s->B.A, s->A2);
      std::swap(s->B.b, s->b2);
    }
    const hipError_t e = hipStreamSynchronize(s->stream);
    if (rc == 0 && e != hipSuccess) rc = fail(LPX_DEVICE_ERROR, hipGetErrorString(e));
  }
  if (rc) return rc;
  return multi_finish_loop(M, pivots_done, status, track_slot);
}

// The loop of LPSolver.simplex (LPSolver.java:101-107) over the shards: blocks of K decisions (one persistent launch
// per device, exchanging among themselves), then one sweep per device.
extern "C" int lpx_multi_simplex_loop(lpx_multi* M, int64_t max_pivots, int64_t* pivots_done, int32_t* status,
                                      int32_t* track_slot) {
  DeviceRestore keep_device;
  if (!M) return fail(LPX_BAD_ARGUMENT, "lpx_multi_simplex_loop: NULL handle");
  const int G = M->n_dev;
  // pivots per sweep: by the largest shard (every device must take the same decisions in the same blocks)
  int K = 2;
  int64_t work = 1;
  for (int r = 0; r < G; r++) {
    K = std::max(K, choose_block(M->sh[r]));
    work = std::max<int64_t>(work, std::max<int64_t>(M->sh[r]->m, M->sh[r]->B.ld));
  }
  K = std::min(K, (int)lpxk::kShardBlockMax);
  bool ring_built = false;
  for (int r = 0; r < G; r++) {
    lpx_state* s = M->sh[r];
    HIP_TRY(hipSetDevice(M->device[r]));
    if (int rc = set_running(s, max_pivots, track_slot ? *track_slot : -1)) return rc;
    ring_built = ring_built || s->R.prow == nullptr;
    if (int rc = ensure_block_ring(s)) return rc;
    launch_seed_entering(s);
    HIP_TRY(hipGetLastError());
  }
  if (ring_built) {
    // A new ring's mailbox, arrival words and pivot-row ring are zeroed by memsets queued on the shard's OWN stream,
    // and a peer's decision kernel stores into them as soon as IT runs: every memset must have landed before any
    // decision kernel of the set is launched (a late memset would erase a candidate and its reader would spin out).
    for (int r = 0; r < G; r++) {
      HIP_TRY(hipSetDevice(M->device[r]));
      HIP_TRY(hipStreamSynchronize(M->sh[r]->stream));
    }
  }
  // grid of the decision kernels: the same on every device, never more than any device holds resident — shards
  // that share a GPU share its CUs (their kernels spin on each other)
  int want = M->sh[0]->opt[LPX_OPT_CHAIN_WGS] > 0 ? (int)M->sh[0]->opt[LPX_OPT_CHAIN_WGS]
                                                   : (int)std::min<int64_t>(32, std::max<int64_t>(1, (work + 511) / 512));
  if (M->sh[0]->pricing == 0 && want >= 8 && M->sh[0]->opt[LPX_OPT_CHAIN_WGS] == 0) want += 1;
  int wgs = want;
  for (int r = 0; r < G; r++) {
    int same = 0;
    for (int q = 0; q < G; q++) same += M->device[q] == M->device[r];
    HIP_TRY(hipSetDevice(M->device[r]));
    const int cus = std::max(1, device_cus(M->sh[r]) / same);
    wgs = std::min(wgs, clamp_chain_wgs(M->sh[r], want, cus));
  }
  for (int r = 0; r < G; r++) { M->sh[r]->info.chain_wgs = wgs; M->sh[r]->info.overlapped = 0; M->sh[r]->info.chain_stream_masked = 0; }
  if (M->spin_max == 0) {
    // First use of this device set.  The decision kernels of a block wait for each other, and ONE host thread launches
    // them device after device: a kernel's first launch on a device (code object load, queue creation) must not fall
    // inside such a wait.  So every device first runs the very kernel at the very grid with nothing to decide (nb = 0:
    // every workgroup returns after reading the loop state), timed from launch to completion and waited for; the
    // slowest one sizes the bound of the cross-device waits: 128 x that time in polls of >= ~0.5 us each, never below
    // the default 2^22 polls (seconds).
    // (twice: the FIRST launch on a device loads the code object and creates the queue — 0.1-1 s that say nothing about a
    // wait inside the loop; the bound follows the second, warm one, and never exceeds 2^26 polls: a bug still ends in
    // LPX_DEVICE_ERROR after tens of seconds, not minutes)
    double worst_us = 0.0, first_us = 0.0;
    for (int pass = 0; pass < 2; pass++) {
      worst_us = 0.0;
      for (int r = 0; r < G; r++) {
        lpx_state* s = M->sh[r];
        HIP_TRY(hipSetDevice(M->device[r]));
        HIP_TRY(hipStreamSynchronize(s->stream));
        lpxk::MgPeers P = peers_of(M, r);
        const auto t0 = std::chrono::steady_clock::now();
        lpxk::launch_block_chain(s->B, s->R, s->n, s->m, 0, 0, 0, 0, 1, M->seq, M->sh[0]->pricing == 1, wgs, 3, false, nullptr,
                                 s->stream, &P);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s->stream));
        worst_us = std::max(worst_us, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
      }
      if (pass == 0) first_us = worst_us;
    }
    M->first_launch_us = first_us;
    M->spin_max = (unsigned)std::min<double>((double)(1u << 26), std::max<double>((double)(1u << 22), 128.0 * worst_us));
  }
  // across real devices the conservative barrier form (release + acquire around every exchange) unless the caller
  // chose one with lpx_multi_set_option: the fence-free form is validated inside one device only
  const int fences = (M->distinct_devices && !M->fences_chosen) ? 3 : (int)M->sh[0]->opt[LPX_OPT_CHAIN_FENCES];
  const bool trace = M->sh[0]->opt[LPX_OPT_CHAIN_TRACE] != 0;
  const int dantzig = M->sh[0]->pricing == 1;

  // Decisions one block ahead of the sweeps, as on one GPU (lpx_engine.cpp blocked_loop_overlapped): every shard
  // keeps two tableau buffers; the decision kernels of block k read the buffers sweep k-1 reads (its pivots are
  // pending ones for them, like their own) and run beside it on their own streams.
  // (a budget that fits one block has nothing to run beside: the serial form, as in lpx_engine.cpp blocked_loop)
  const bool one_block = max_pivots >= 0 && max_pivots + 1 <= K;
  if (M->sh[0]->opt[LPX_OPT_OVERLAP] != 0 && !one_block) {
    bool ok = true;
    for (int r = 0; r < G && ok; r++) {
      HIP_TRY(hipSetDevice(M->device[r]));
      // Shards that share a GPU (rehearsal) get plain streams: streams created with IDENTICAL CU masks were seen to
      // be served by one hardware queue, and two decision kernels that wait for each other must not queue behind
      // one another (the wait then runs into its spin bound).
      int same = 0;
      for (int q = 0; q < G; q++) same += M->device[q] == M->device[r];
      if (same > 1 && !M->sh[r]->ov_chain) M->sh[r]->opt[LPX_OPT_OVERLAP_MASK] = 0;
      ok = ensure_spare_tableau(M->sh[r]) == 0 && ensure_overlap_streams(M->sh[r]) == 0;
    }
    if (ok) return multi_loop_overlapped(M, K, max_pivots, want, fences, trace, dantzig, pivots_done, status, track_slot);
    (void)hipGetLastError();   // no room for the second tableau: the serial form below
  }

  int64_t decided = 0;
  auto issue_block = [&](int slot) -> int {
    const int nb = block_len(K, max_pivots, decided);
    if (nb <= 0) {   // nothing left to decide: only mark the slot as issued
      for (int r = 0; r < G; r++) {
        HIP_TRY(hipSetDevice(M->device[r]));
        HIP_TRY(hipMemcpyAsync(&M->h_snap[2 * r + slot], M->sh[r]->B.ctl, sizeof(LpxCtl), hipMemcpyDeviceToHost, M->sh[r]->stream));
        HIP_TRY(hipEventRecord(M->ev[slot][r], M->sh[r]->stream));
      }
      return 0;
    }
    const bool probe_only = max_pivots >= 0 && decided == max_pivots;   // can only report the end: nothing to sweep
    for (int r = 0; r < G; r++) {
      lpx_state* s = M->sh[r];
      HIP_TRY(hipSetDevice(M->device[r]));
      lpxk::MgPeers P = peers_of(M, r);
      P.mail_slot0 = (int)(decided & 1);   // every earlier block of this loop took all its decisions
      lpxk::launch_block_chain(s->B, s->R, s->n, s->m, nb, 0, 0, 0, 1, M->seq, dantzig, wgs, fences, trace,
                               M->d_snap[r] + slot, s->stream, &P);
      s->chain_nb_last = nb;
      HIP_TRY(hipGetLastError());
    }
    M->seq++;
    for (int r = 0; r < G; r++) {
      lpx_state* s = M->sh[r];
      HIP_TRY(hipSetDevice(M->device[r]));
      if (!probe_only) {
        if (int rc = launch_sweep_profiled(s, nb, s->stream, s->B, s->R, nullptr, nullptr)) return rc;
      }
      HIP_TRY(hipEventRecord(M->ev[slot][r], s->stream));
    }
    decided += nb;
    return 0;
  };

  int rc = issue_block(0);
  int cur = 0;
  while (rc == 0) {
    rc = issue_block(cur ^ 1);
    if (rc) break;
    bool running = true;
    for (int r = 0; r < G && rc == 0; r++) {
      const hipError_t e = hipEventSynchronize(M->ev[cur][r]);
      if (e != hipSuccess) { rc = fail(LPX_DEVICE_ERROR, hipGetErrorString(e)); break; }
      if (M->h_snap[2 * r + cur].status != lpxk::kRunning) running = false;
    }
    if (rc || !running) break;
    cur ^= 1;
  }
  for (int r = 0; r < G; r++) {
    (void)hipSetDevice(M->device[r]);
    const hipError_t e = hipStreamSynchronize(M->sh[r]->stream);
    if (rc == 0 && e != hipSuccess) rc = fail(LPX_DEVICE_ERROR, hipGetErrorString(e));
  }
  if (rc) return rc;
  return multi_finish_loop(M, pivots_done, status, track_slot);
}

extern "C" int lpx_multi_read(lpx_multi* M, double* A, int64_t lda, double* b, double* c, double* v, int32_t* perm) {
  DeviceRestore keep_device;
  if (!M) return fail(LPX_BAD_ARGUMENT, "lpx_multi_read: NULL handle");
  for (int r = 0; r < M->n_dev; r++) {
    const int32_t r0 = M->row_start[r];
    if (int rc = lpx_state_read(M->sh[r], A ? A + (int64_t)r0 * lda : nullptr, lda, b ? b + r0 : nullptr,
                                r == 0 ? c : nullptr, r == 0 ? v : nullptr, r == 0 ? perm : nullptr))
      return rc;
  }
  return 0;
}

extern "C" int lpx_multi_checksum(lpx_multi* M, uint64_t out[3]) {
  DeviceRestore keep_device;
  if (!M || !out) return fail(LPX_BAD_ARGUMENT, "lpx_multi_checksum: NULL argument");
  out[0] = out[1] = out[2] = 0;
  for (int r = 0; r < M->n_dev; r++) {
    uint64_t h[3];
    if (int rc = lpx_state_checksum(M->sh[r], h)) return rc;
    out[0] += h[0];   // position-keyed sums over disjoint rows add up to the whole tableau's
    out[1] += h[1];
    if (r == 0) out[2] = h[2];   // c is replicated
  }
  return 0;
}

extern "C" int lpx_multi_profile_enable(lpx_multi* M, int enable) {
  if (!M) return fail(LPX_BAD_ARGUMENT, "lpx_multi_profile_enable: NULL handle");
  for (lpx_state* s : M->sh)
    if (int rc = lpx_profile_enable(s, enable)) return rc;
  return 0;
}

extern "C" int lpx_multi_profile_read(lpx_multi* M, int32_t shard, int64_t* launches, double* total_ms) {
  DeviceRestore keep_device;
  if (!M || shard < 0 || shard >= M->n_dev) return fail(LPX_BAD_ARGUMENT, "lpx_multi_profile_read: bad argument");
  return lpx_profile_read(M->sh[shard], launches, total_ms);
}

extern "C" int lpx_multi_get_info(lpx_multi* M, lpx_state_info* out) {
  DeviceRestore keep_device;
  if (!M || !out) return fail(LPX_BAD_ARGUMENT, "lpx_multi_get_info: NULL argument");
  return lpx_state_get_info(M->sh[0], out);
}

// ---- hooks for lpx_solve_multi (lpx_solver.cpp): the phase-1 constructions of LPSolver.java on shards ------------
namespace lpx_internal {

int multi_shards(lpx_multi* M) { return M->n_dev; }
lpx_state* multi_shard(lpx_multi* M, int r) { return M->sh[r]; }
int multi_device(lpx_multi* M, int r) { return M->device[r]; }
int32_t multi_row_start(lpx_multi* M, int r) { return M->row_start[r]; }
int multi_owner(lpx_multi* M, int32_t row) { return shard_of(M, row); }
void multi_set_n(lpx_multi* M, int32_t n) { M->n = n; }

}  // namespace lpx_internal
