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

// What the first launch of a loop call resets in LpxCtl (k_entering / k_entering_dantzig): see loop_start().
struct LoopStart { int32_t reset = 0; int32_t track = -1; int64_t max_pivots = -1; };

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
  int chain_form;     // decision kernel of the one-device blocked loop: 0 = k_block_chain_t, 1 = k_block_chain2_t (LPX_OPT_CHAIN_FORM)
  int fused;          // 0: lpxk::plain kernels (two roundings per update, the default); 1: lpxk::fused (LPX_OPT_FUSED)
};

// blocked pivoting: ring of pending pivots (see lpx_kernels.hip "blocked pivoting")
constexpr int kBlockMax = 64;        // pivots per block / slots per ring half (single-device loop)
constexpr int kShardBlockMax = 32;   // the step-wise shard interface and lpx_multi decide at most this many per block
constexpr int kChainMaxWgs = 256;   // <= one workgroup per CU: the whole grid is resident
constexpr int kMaxDevices = 8;      // row-block shards of one lpx_multi (the GPUs of one node)
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
  int32_t* chain_own_rs;   // mp + ld: k_block_chain2's start indices per row / per slot (last pending pivot that replaced it)
  long long* chain_dbg;    // diagnostics (LPX_OPT_CHAIN_TRACE): 5 timestamps per decision of the last block
  unsigned* census;        // [w] = XCC id + 1 of chain workgroup w; [kChainMaxWgs] = OR of (1 << XCC id) of sampled sweep workgroups
  double* fix_col;         // overlapped loop: images of the fix-up's entering columns [chain][mp] (NULL: the fix-up follows the sweep and writes the tableau itself)
  double* fix_row;         // ... and of its pivot rows [chain][ld]; both 2 x kBlockMax chains like the rings
  double* col_packed;      // k_sweep32_pull: the block's multipliers as [batch of 4 rows][pivot][row]: (mp / 4) x 1 KiB (2 KiB for blocks of 64)
  unsigned* tickets;       // k_sweep32_pull: one batch counter per 128-column sub-strip, 128 bytes apart (ld / 128 of them); like col_packed once per ring half
  unsigned* sweep_fail;    // the word a pull kernel sets when a bounded wait ran out (k_sweep64_pull: variants library only)
  long long* clk;          // clock probe of the last pulled sweep, per XCD x: clk[4 x + 0..1] = {s_memtime, 100 MHz} in front of it, [4 x + 2..3] behind it
  const double* zeros;     // 256 bytes of +0.0: what k_sweep32_dma's multiplier DMA reads for the identity steps of a partly filled block
  // shards of an lpx_multi only (else NULL): written by the peers' decision kernels
  void* mg_mail;                   // MgMail[2][kMaxDevices]
  unsigned long long* mg_arrive;   // [kChainMaxWgs]
  double* mg_candrow;              // one-hop form: candidate rows [2][kMaxDevices][ld]
  unsigned long long* mg_arrive2;  // one-hop form: arrival words of the candidate rows [2][kMaxDevices][kChainMaxWgs]
};
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
  unsigned spin_max;                       // bound of the waits between devices in polls (0: the default, 2^22 = seconds)
};
// which kernel swept the bulk of the tableau (lpx_state_info.sweep_kernel)
enum SweepKernel { kSweepNone = 0, kSweepTiles = 1, kSweepMulti = 2, kSweepSteady = 3, kSweepPipe64 = 4, kSweepDma = 5, kSweepPull = 6, kSweepPull64 = 7, kSweepOne64 = 8, kSweepMfma64 = 9, kSweepMfma642 = 10 };
// The fix-up of a block beside its sweep (launch_block_sweep): where its chains run and the two events that tie them in.
// ready: the block's decisions are through (the side stream waits for it); packed: the sweep's pack kernel is through (may be
// NULL: it then runs on the sweep's stream); done: the fix-up's chains are through (the copy kernel waits for it).
struct FixSide { hipStream_t stream; hipEvent_t ready, done, packed; };
struct RestoreEntry { int32_t is_basic; int32_t index; double k; };  // index = row r (basic) or post-drop slot

// ---- launch wrappers --------------------------------------------------------------------------------------------
// lpx_kernels.hip is compiled twice: plain (one rounding per reference operation) and fused (updates as one FMA).
namespace plain {
#include "lpx_launchers.inc"
}  // namespace plain
namespace fused {
#include "lpx_launchers.inc"
}  // namespace fused
// what the host engine calls: plain:: or fused:: by Buffers::fused (lpx_dispatch.cpp)
#include "lpx_launchers.inc"

}  // namespace lpxk
