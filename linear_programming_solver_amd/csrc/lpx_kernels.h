// Internal interface between the HIP kernels (lpx_kernels.hip) and the host engine (lpx_engine.cpp).
// Not part of the public ABI (that is include/lpx.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lpxk {

constexpr int kRunning = -1;     // LpxCtl::status while the loop is live; otherwise an lpx_status value
constexpr double kEps = 1e-9;    // DEF_EPSILON  reference LPState.java:20
constexpr double kInf = 1e50;    // DEF_INF      reference LPState.java:21

// (ratio, row) candidate of the minimum-ratio test; "none" is {kInf, INT32_MAX}.
struct RatioRow {
  double ratio;
  int32_t row;   // GLOBAL row index
  int32_t pad;
};

// Device-resident loop control: the host never decides anything per pivot, it only polls `status`.
struct LpxCtl {
  double v;            // objective constant (LPState.v)
  double p;            // pivot element A[l][e] of the pivot being applied
  double bl;           // b[l] / p
  double pc;           // c[e] before the pivot
  double ratio;        // winning ratio of the last ratio test
  int32_t e_next;      // entering slot chosen for the NEXT pivot (-1: none)
  int32_t e_cur;       // entering slot of the pivot being applied by k_update
  int32_t l;           // leaving row (GLOBAL index) of the pivot being applied / last ratio test winner
  int32_t status;      // kRunning or lpx_status
  int32_t do_update;   // 1 iff k_select_pivot performed a pivot that k_update must apply
  int32_t track;       // slot of the tracked variable (x0 in phase 1), -1 = none  (LPSolver.java:151-155)
  int32_t parity;      // col[parity] receives / holds column e_next
  int32_t e_min;       // scratch of the pivot-finish kernels: atomicMin target, INT32_MAX when idle
  int32_t ticket;      // scratch: arrival counter of the pivot-finish workgroups, 0 when idle
  int32_t reserved;
  int64_t pivots;      // pivots performed since lpx_simplex_loop started
  int64_t max_pivots;  // budget for `pivots` (<0: unlimited)
};

struct Geometry {
  int U;              // double2 per thread per row in k_update (strip width = 512*U columns)
  int rows_per_tile;  // rows handled by one k_update block
  int nstrips, ntiles;
};

struct Buffers {
  double* A;          // m_local x ld, row-major, columns [n, ld) are zero
  int64_t ld;
  double* b;          // m_local
  double* c;          // ld (padding zero)
  double* prow;       // ld: normalised pivot row of the pivot being applied
  double* col[2];     // m_local each: column e of the tableau (ping-pong, see LpxCtl::parity)
  RatioRow* partial;  // ntiles
  int32_t* perm;      // n + m_global
  LpxCtl* ctl;
};

// ---- launch wrappers (all asynchronous on `s`) -------------------------------------------------------------
void launch_entering(const Buffers& B, int n, hipStream_t s);
// opt-in Dantzig pricing: overrides ctl->e_next after a decision (seed: at the start of a loop)
void launch_entering_dantzig(const Buffers& B, int n, bool seed, hipStream_t s);
// forced_e >= 0: use it as e_next instead of ctl->e_next (step API)
void launch_ratio_gather(const Buffers& B, int m_local, int row0, const Geometry& g, int forced_e, hipStream_t s);
void launch_reduce_partials(const Buffers& B, const Geometry& g, hipStream_t s);
// forced_l >= 0 (global row): pivot(forced_e, forced_l) of the step API, no ratio test
void launch_select_pivot(const Buffers& B, int n, int m_global, const Geometry& g, int forced_e, int forced_l,
                         hipStream_t s);
// prow / up: the normalised pivot row and the parameter block of the pivot to apply (B.prow / B.ctl in the
// two-launch loop, one ring slot in the look-ahead pipeline)
// A_out/b_out == nullptr: update (B.A, B.b) in place; otherwise read (B.A, B.b), write (A_out, b_out)
void launch_update(const Buffers& B, int m_local, int n, int row0, const Geometry& g, bool nontemporal,
                   const double* prow, const LpxCtl* up, double* A_out, double* b_out, hipStream_t s);
// shards
void launch_propose(const Buffers& B, int n, int row0, int m_local, const Geometry& g, double* d_candidate,
                    hipStream_t s);
void launch_commit(const Buffers& B, int n, int m_global, const double* d_gathered, int nranks, double* prow,
                   LpxCtl* up, int up_parity, hipStream_t s);
// look-ahead: candidate of the NEXT pivot computed from the tableau BEFORE the pending update `pend`
void launch_peek(const Buffers& B, int n, int m_local, int row0, const double* prow_t, const double* col_t,
                 double* col_next, const LpxCtl* pend, double* d_candidate, hipStream_t s);
// blocked pivoting: ring of pending pivots (see lpx_kernels.hip "blocked pivoting")
constexpr int kBlockMax = 64;        // pivots per block / slots per ring half (single-device loop)
constexpr int kShardBlockMax = 32;   // the step-wise shard interface and lpx_multi decide at most this many per block
struct BlockRing {  // every ring has 2 * kBlockMax slots: two halves, one per block in flight
  double* prow;   // kBlockMax x ld : normalised pivot row of pending pivot s
  double* col;    // kBlockMax x mp : column e_s of the tableau just before pivot s
  double* col0;   // kBlockMax x mp : the same column as it stands in the stale tableau (for the fix-up)
  double* row0;   // kBlockMax x ld : the stale pivot row l_s (for the fix-up)
  LpxCtl* up;     // kBlockMax parameter blocks (written by finish_pivot)
  int64_t mp;
  // scratch of k_block_chain (single-shard blocks decided in one persistent launch)
  void* chain_part_a;      // 2 x kChainMaxWgs x 32 B
  void* chain_part_b;      // kChainMaxWgs x 16 B
  unsigned* chain_bar;     // 2 barrier counters + the hand-off word, 128 B apart; a launch zeroes the next one's counter
  double* chain_own_col;   // kBlockMax x mp: copy of `col` that only its writer re-reads (plain, cache-resident)
  double* chain_own_prow;  // kBlockMax x ld: likewise for `prow`
  double* chain_own_dvc;   // kBlockMax x mp: column e_s of the tableau just AFTER pivot s (restart point)
  double* chain_own_b;     // mp: b with all pending pivots applied
  long long* chain_dbg;    // diagnostics (LPX_OPT_CHAIN_TRACE): 5 timestamps per decision of the last block
  unsigned* census;        // [w] = XCC id + 1 of chain workgroup w; [kChainMaxWgs] = OR of (1 << XCC id) of sampled sweep workgroups
  double* col_packed;      // k_sweep32_pull: the block's multipliers as [batch of 4 rows][pivot][row]: (mp / 4) x 1 KiB (2 KiB for blocks of 64)
  unsigned* tickets;       // k_sweep32_pull: one batch counter per 128-column sub-strip, 128 bytes apart (ld / 128 of them)
  const double* zeros;     // 256 bytes of +0.0: what k_sweep32_dma's multiplier DMA reads for the identity steps of a partly filled block
  // shards of an lpx_multi only (else NULL): written by the peers' decision kernels
  void* mg_mail;                   // MgMail[2][kMaxDevices]
  unsigned long long* mg_arrive;   // [kChainMaxWgs]
  double* mg_candrow;              // one-hop form: candidate rows [2][kMaxDevices][ld]
  unsigned long long* mg_arrive2;  // one-hop form: arrival words of the candidate rows [2][kMaxDevices][kChainMaxWgs]
};
constexpr int kChainMaxWgs = 256;   // <= one workgroup per CU: the whole grid is resident
constexpr int kMaxDevices = 8;      // row-block shards of one lpx_multi (the GPUs of one node)
// decision number `np` of a block (np pivots pending): candidate record like k_propose's
void launch_block_peek(const Buffers& B, const BlockRing& R, int n, int m_local, int row0, int np, double* d_candidate,
                       hipStream_t s);
void launch_block_decide(const Buffers& B, const BlockRing& R, int n, int m_global, const double* d_gathered, int nranks,
                         int slot, hipStream_t s);
// all nb decisions of a block in one persistent launch (single shard: row0 == 0, m == m_global); wgs <= 0: auto.
// The rings hold 2*kBlockMax slots: `half` is the block's, `old_half` that of the previous block when its sweep has
// not yet reached the tableau (B.A, B.b) this launch reads (n_old pivots; 0: none).  b_from_tableau: first launch
// of a loop.  seq: launch counter (the two barrier counters alternate).  host_snap: device-visible pointer to a
// pinned host LpxCtl that receives the loop state when the launch ends.
// Row-block shards on several devices deciding together (lpx_multi): what a shard's launch needs to know about its
// peers.  Pointers are peer-mapped device pointers, index = shard rank; [dev] is the shard's own memory.
struct MgPeers {
  int n_dev, dev;
  int row0, m_global;
  int mail_slot0;                          // parity of the decisions taken by earlier launches of the loop
  void* mail[kMaxDevices];                 // MgMail[2][kMaxDevices] (32-byte records) of every shard
  double* prow[kMaxDevices];               // base of every shard's pivot-row ring (BlockRing::prow)
  unsigned long long* arrive[kMaxDevices]; // arrival words [kChainMaxWgs] of every shard
  int onehop;                              // 1: every shard ships its candidate's ROW with the candidate (one hop per decision)
  double* candrow[kMaxDevices];            // candidate rows [2][kMaxDevices][ld] of every shard
  unsigned long long* arrive2[kMaxDevices];// their arrival words [2][kMaxDevices][kChainMaxWgs]
};
// fences: grid-barrier form (bit 0 release fence, bit 1 acquire fence); trace: record phase timestamps in R.chain_dbg.
// mg != NULL: the launch of one shard of an lpx_multi (m = the shard's rows); every shard must use the same wgs.
void launch_block_chain(const Buffers& B, const BlockRing& R, int n, int m, int nb, int half, int old_half, int n_old,
                        int b_from_tableau, int seq, int dantzig, int wgs, int fences, bool trace, LpxCtl* host_snap,
                        hipStream_t s, const MgPeers* mg = nullptr);
// one trivial launch of every kernel of the blocked loop, once per device (the runtime prepares a kernel at its first
// launch); needs the handle's buffers and its ring (with chain_bar) allocated
void preload_block_kernels(const Buffers& B, const BlockRing& R, hipStream_t s);
// device word the pull sweep kernels set when one of their (bounded) LDS waits ran out: 0 in a healthy run
unsigned* sweep_fail_word(const BlockRing& R, int64_t ld);
int64_t sweep_ticket_slots(int64_t ld);   // 128-byte slots of BlockRing::tickets for a row pitch of ld doubles
// hipOccupancyMaxActiveBlocksPerMultiprocessor for k_block_chain (256 threads, its static LDS); >= 1
int chain_blocks_per_cu();
// apply the valid leading pending pivots (at most K) in one pass
// A_src / b_src != NULL: out of place — read the tableau and b there, write the updated ones to B.A / B.b
// rows_per_wg: rows one workgroup walks down (multiple of 64; <= 0: by size and by `cus`, the CUs the stream may use,
// 0 = the whole device); returns the value used.  after_sweep: recorded between the sweep and the fix-up.
// form (blocks of 17..32 over the full strips): 0 = k_sweep32_pull (round 3: LDS-DMA staging, every wave pulls its
// batches in address order), 1 = k_sweep32_steady (round 2: register staging, runs of rows), 2 = k_sweep32_dma (LDS-DMA
// staging, runs of rows).  *kernel_used (may be NULL): the SweepKernel that took the bulk of the tableau.
enum SweepKernel { kSweepNone = 0, kSweepTiles = 1, kSweepMulti = 2, kSweepSteady = 3, kSweepPipe64 = 4, kSweepDma = 5, kSweepPull = 6, kSweepPull64 = 7 };
const char* sweep_kernel_name(int code);
int launch_block_sweep(const Buffers& B, const BlockRing& R, int n, int m_local, int row0, int K, int rows_per_wg,
                       bool nt, hipStream_t s, const double* A_src = nullptr, const double* b_src = nullptr,
                       hipEvent_t after_sweep = nullptr, int cus = 0, int form = 0, int* kernel_used = nullptr);
// phase 1 / restore helpers
void launch_fill_column(double* A, int64_t ld, int m, int col, double value, hipStream_t s);
void launch_drop_column(double* A, int64_t ld, int m, int n_old, int col, hipStream_t s);
struct RestoreEntry { int32_t is_basic; int32_t index; double k; };  // index = row r (basic) or post-drop slot
void launch_restore_objective(const Buffers& B, int n, const RestoreEntry* d_entries, int n_entries, hipStream_t s);
void launch_checksum(const Buffers& B, int m_local, int n, int row0, unsigned long long* d_out3, hipStream_t s);
void launch_transpose(const double* dA, int64_t lda, double* dAt, int64_t ldat, int m, int n, hipStream_t s);

}  // namespace lpxk
