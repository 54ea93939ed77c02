/*
 * lpx.h — C ABI of the MI355X-native dense simplex pivot engine (liblpx.so).
 *
 * This is the drop-in boundary for ONE path of Toptachamann/Linear_Programming_Solver: the simplex
 * pivot loop.  The reference has no FFI seam of its own (pure Java); the seam this library sits
 * behind is the Java API
 *
 *     BigDecimal LPSolver.solve(LPStandardForm) throws LPException      (LPSolver.java:78)
 *     int  LPState.getEntering()                                         (LPState.java:274)
 *     int  LPState.getLeaving(int entering)                              (LPState.java:287)
 *     void LPState.pivot(int entering, int leaving)                      (LPState.java:114)
 *
 * and every entry point below names the reference member it replaces.  Plain pointers and sizes only;
 * no torch, no C++ types.  A JNI / ctypes binding over these symbols is shown in INTEGRATION.md.
 *
 * Data model (reference LPState.java:25-30, "condensed" CLRS slack form):
 *   A   m x n row-major fp64, A[i][j] = coefficient of the variable in nonbasic SLOT j in the equation
 *       of the basic variable of row i:   x_basic(i) = b[i] - sum_j A[i][j] * x_slot(j)
 *   b   m, c n, scalar v (objective constant).  Maximisation sign convention (c[j] > 0 improves).
 *   perm int32[n+m]: slot -> variable id.  Slots 0..n-1 are nonbasic, slot n+i is the basic variable
 *       of row i.  Ids 0..n-1 are the caller's original variables, n..n+m-1 the slacks of rows 0..m-1
 *       (this replaces the reference's two HashMaps `variables` / `coefficients`, LPState.java:27-28,
 *       swapped by exchangeIndexes, :311-320).
 * Arithmetic is IEEE fp64.  Default: one rounding per reference operation, never fused: t = ce*row[j]; a - t
 * (the reference rounds the product and the difference separately, LPState.java:162) — bit-identical to a plain
 * unfused fp64 restatement of the reference.  Opt-in (LPX_OPT_FUSED, lpx_solve_options.fused): every update
 * a - ce*row[j] is ONE fused multiply-add — one binary rounding where the reference has two decimal ones; same
 * tolerance class against the reference's BigDecimal results (objective, basis), different bits, half the fp64
 * instructions in the row-update kernels.
 *
 * Threading: a handle is single-caller (the reference classes are not thread-safe either); different
 * handles may be driven from different host threads.  All device work of a handle is issued on one HIP
 * stream (lpx_state_set_stream); calls that return values to the host synchronise that stream.
 */
#ifndef LPX_H
#define LPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LPX_ABI_VERSION 5

/* Status codes.  One per exception message of the reference; the host shim maps them back
 * to the exact exception class + message because the reference's tests assert on the text. */
typedef enum lpx_status {
  LPX_OPTIMAL = 0,             /* getEntering() == -1: loop finished                 LPSolver.java:101      */
  LPX_UNBOUNDED = 1,           /* SolutionException "This linear program is unbounded"          :103-106    */
  LPX_INFEASIBLE = 2,          /* LPException "This linear program is infeasible"               :171-174    */
  LPX_AUX_UNBOUNDED = 3,       /* SolutionException "Auxiliary lp is unbounded"                 :147-150    */
  LPX_NO_DEGENERATE_PIVOT = 4, /* SolutionException "Can't perform degenerate pivot"            :192-194    */
  LPX_BAD_ARGUMENT = 5,        /* IllegalArgumentException (Validate.isTrue)             LPState.java:288   */
  LPX_RESTORE_INDEX_FAULT = 6, /* ArrayIndexOutOfBoundsException in restoreInitialLP     LPSolver.java:231  */
  LPX_DEVICE_ERROR = 7,        /* HIP runtime error (no reference analogue)                                 */
  LPX_DIVIDE_BY_ZERO = 8,      /* ArithmeticException: pivot on a zero element           LPState.java:139   */
  LPX_PIVOT_LIMIT = 9          /* max_pivots reached, loop still running (no reference analogue)            */
} lpx_status;

/* Exact reference exception text for a status ("" for OPTIMAL / PIVOT_LIMIT). Never NULL. */
const char* lpx_status_message(int status);
/* Last error text of the calling thread (HIP error string, argument complaint). Never NULL. */
const char* lpx_last_error(void);
int lpx_abi_version(void);
/* Number of HIP devices visible; <0 on error. */
int lpx_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * LPState: device-resident slack-form tableau + the three simplex primitives.
 * ---------------------------------------------------------------------------------------------- */
typedef struct lpx_state lpx_state;

/* new LPState(A, b, c, v, variables, coefficients, m, n)              LPState.java:88-112
 * Copies the host arrays into HBM on `device` (the reference aliases the caller's arrays,
 * LPSolver.java:267; this library never writes caller memory except in the read-back calls).
 * A is row-major with leading dimension lda >= n (elements).  perm may be NULL: identity.
 * Row-block shard form (multi-GPU): this handle owns global rows
 * [row0, row0 + m_local) of an m_global-row tableau; pass row0 = 0, m_global = m_local for one GPU.
 * b has m_local entries (the shard's rows); c, v and perm (n + m_global entries) are replicated. */
int lpx_state_create(int32_t m_local, int32_t n, const double* A, int64_t lda, const double* b,
                     const double* c, double v, const int32_t* perm, int32_t row0, int32_t m_global,
                     int device, lpx_state** out);
/* Same, but A/b/c are DEVICE pointers already resident in HBM on `device` (copied device-to-device). */
int lpx_state_create_from_device(int32_t m_local, int32_t n, const double* dA, int64_t lda,
                                 const double* db, const double* dc, double v, const int32_t* perm,
                                 int32_t row0, int32_t m_global, int device, lpx_state** out);
void lpx_state_destroy(lpx_state* s);
/* Issue all further device work of this handle on `hip_stream` (a hipStream_t; NULL = the handle's own). */
int lpx_state_set_stream(lpx_state* s, void* hip_stream);

/* Replace the handle's own stream by one whose CU mask leaves 32 * reserve_units (0..4) CUs to OTHER streams — the
 * same 4 * reserve_units CUs on every one of the 8 XCDs (a mask cannot exclude a whole XCD on MI355X: an XCD whose
 * mask bits are all clear runs unmasked) — and make it the handle's stream.  The row update is HBM-bound and keeps
 * its rate on the remaining CUs; the reserved ones guarantee that a collective or a decision kernel issued on
 * another stream runs beside it (look-ahead pipeline).  *stream_out receives the hipStream_t (wrap it, e.g. torch
 * ExternalStream). */
int lpx_state_use_masked_stream(lpx_state* s, int32_t reserve_units, void** stream_out);

/* Entering rule of this handle.  0 (default) = the reference's rule: first slot with c[j] > 1e-9
 * (LPState.java:274-285).  1 = Dantzig: largest c[j], lowest slot on ties — an OPT-IN extension of this
 * library: it reaches the same optimum in ~10x fewer pivots on dense LPs but deliberately leaves
 * the reference's pivot sequence, so basis/trace parity with the reference no longer applies. */
int lpx_state_set_pricing(lpx_state* s, int32_t pricing);

/* int getEntering()                                                    LPState.java:274-285
 * *entering = min{ j in [0,n) : c[j] > 1e-9 } or -1. */
int lpx_get_entering(lpx_state* s, int32_t* entering);
/* int getLeaving(int entering)                                         LPState.java:287-305
 * ratio_i = (A[i][e] < 1e-9) ? 1e50 : b[i]/A[i][e]; strict '<' scan from 1e50: lowest row wins ties;
 * *leaving = -1 if no ratio is below 1e50.  entering outside [0,n) -> LPX_BAD_ARGUMENT.
 * On a shard the result is the shard-local candidate (global row index) and *ratio its ratio. */
int lpx_get_leaving(lpx_state* s, int32_t entering, int32_t* leaving, double* ratio);
/* void pivot(int entering, int leaving)                                LPState.java:114-181, :311-320
 * Pivot row normalise, rank-1 update of all other rows and of b, objective row and v update,
 * slot/basis swap in perm.  `leaving` is a row index in [0,m).  A zero pivot element returns
 * LPX_DIVIDE_BY_ZERO and leaves the state untouched.  Single-GPU handles only. */
int lpx_pivot(lpx_state* s, int32_t entering, int32_t leaving);

/* Tuning and diagnostic options of ONE handle (they replace the LPX_* environment knobs of the first version:
 * a JVM host sets them through the handle, not through its process environment).  Every option has a measured
 * default; the LPX_<NAME> environment variables are read ONCE per process, before the first handle is created, and
 * only supply the initial values (debugging aid for the scripts/ helpers).  Setting an option between two loops
 * is allowed; inside a loop nothing reads them.  Unknown key / value out of range -> LPX_BAD_ARGUMENT. */
typedef enum lpx_option {
  LPX_OPT_BLOCK = 0,          /* pivots per sweep: 0 = by size (1 below ~18 MiB of tableau on a handle without a ring, 16 up to ~28 MiB, 32 up to ~7 GiB, 64 above; fused arithmetic: 64 from ~0.85 GiB), 1 = off, 2..64 (same as lpx_state_set_block) */
  LPX_OPT_CHAIN = 1,          /* 1 = all decisions of a block in one persistent launch (default); 0 = three launches */
  LPX_OPT_OVERLAP = 2,        /* 1 = decisions of block k+1 beside the sweep of block k (default); 0 = serial        */
  LPX_OPT_OVERLAP_SERIAL = 3, /* 1 = the overlapped loop's kernels and buffers without concurrency (diagnostics)     */
  LPX_OPT_OVERLAP_MASK = 4,   /* 1 = CU-masked streams: 4 CUs of every XCD for the decisions, the rest for the sweep  */
  LPX_OPT_CHAIN_WGS = 5,      /* workgroups of the decision kernel; 0 = by size; always clamped to what is resident  */
  LPX_OPT_CHAIN_FENCES = 6,   /* grid barrier of the decision kernel: bit 0 release fence, bit 1 acquire fence       */
  LPX_OPT_SWEEP_ROWS = 7,     /* rows one workgroup of the blocked sweep walks down (multiple of 64); 0 = by size    */
  LPX_OPT_NT = 8,             /* non-temporal tableau loads/stores: -1 = by size, 0, 1                               */
  LPX_OPT_BATCH = 9,          /* one-pass loop: pivots issued between two host polls; 0 = by size                    */
  LPX_OPT_CHAIN_TRACE = 10,   /* 1 = keep phase timestamps of the last decision launch (lpx_state_read_chain_trace)  */
  LPX_OPT_UPDATE_U = 11,      /* one-pass update: 16-byte accesses per thread per row (1, 2, 4)                      */
  LPX_OPT_UPDATE_ROWS = 12,   /* one-pass update: rows per workgroup (even, 2..256)                                  */
  LPX_OPT_A2_OFFSET = 13,     /* skew between the two tableau buffers in doubles (before the second one exists)      */
  LPX_OPT_SWEEP_FORM = 14,    /* 0 (default) = the sweep kernels of the library: blocks of 17..32 k_sweep32_pull (LDS-DMA staging, batches pulled in address order); blocks of 33..64 k_sweep64_one (one wave per 64-column sub-strip), in the fused-arithmetic mode with 16-row tiles (m % 16 == 0) k_sweep64_mfma2 (the matrix cores); 3 = k_sweep64_one in the fused mode too.  1, 2, 4 name the superseded kernels of csrc/variants/ (k_sweep32_steady / k_sweep64_pipe, k_sweep32_dma / k_sweep64_pull, k_sweep64_mfma): they select them in the variants library (make variants) only and mean the default here */
  LPX_OPT_MULTI_ONEHOP = 15,  /* lpx_multi: 1 = every shard ships its candidate's row with its candidate (one cross-device hop per decision instead of two); 0 (default) = candidates, then the winner's normalised row */
  LPX_OPT_SWEEP_CUS = 16,     /* overlapped loop: CUs of the sweep stream's mask (multiple of 8; 0 = all but the decisions'); set before the first blocked loop: LPX_BAD_ARGUMENT once the handle's stream pair exists */
  LPX_OPT_CHAIN_CUS = 17,     /* overlapped loop: CUs per XCD reserved for the decision kernel (4, 8, 12 or 16; other values are rounded down to a multiple of 4 but never below 4; 0 = by size: 8 for decision-bound tableaus above 8192 rows or columns, else 4); set before the first blocked loop: LPX_BAD_ARGUMENT once the handle's stream pair exists */
  LPX_OPT_FUSED = 18,         /* arithmetic of the updates x - c*r (LPState.java:162, :164, :177) and v + b*c (:171): 0 = product and difference rounded separately, as the reference rounds them (bit-identical to the unfused fp64 oracle); 1 = one fused multiply-add each (bit-identical to the oracle's fused instantiation); 2 (default) = by size: fused on an unsharded tableau of 0.5 GiB and more, where it is 5-75 % faster, otherwise 0.  Both binary modes leave the decimal-15 pivot sequence of the reference equally often (tests/golden/divergence_census.json).  lpx_state_info.arith_fused reports the mode in effect.  Every kernel of the handle switches together; set it before the first pivot of a solve (the two modes give different bits, so a switch in mid-solve matches neither checker) */
  LPX_OPT_CHAIN_FORM = 19,    /* decision kernel of the blocked loop: 0 = k_block_chain_t (round 2/3), 1 (default) = k_block_chain2_t (round 4/5: a phase asks for everything at once, nothing is drained on the critical path, branch-free pending-pivot ladder; workgroups of 256 threads; also on the shards of an lpx_multi unless LPX_OPT_MULTI_ONEHOP is set) */
  LPX_OPT_FIXUP_SIDE = 20,    /* overlapped loop: where the fix-up of a block (its entering columns and pivot rows recomputed from the ring, LPState.java:139-164) runs.  0 = behind the block's sweep, writing the tableau; 2 = its chains (and the sweep's pack kernel) BESIDE the sweep on a stream with the decisions' CU mask, into compact images, and only the copy of those images (k_block_fixup_scatter) behind the sweep; 1 = likewise on the sweep's CUs; 3 = likewise without a mask; 4 (default) = by size: 2 from 2 GiB of tableau, where the sweep sets the pace, 0 below, where the decisions do.  Unsharded handles only (the shards of an lpx_multi keep 0).  Same values, same order per entry: bit-identical results in every setting */
  LPX_OPT_COUNT = 21
} lpx_option;
int lpx_state_set_option(lpx_state* s, int32_t key, int64_t value);
int lpx_state_get_option(const lpx_state* s, int32_t key, int64_t* value);

/* What the handle actually did (the by-size choices, and whether the placement assumptions of the blocked loop
 * held on this runtime); filled from the last lpx_simplex_loop. */
typedef struct lpx_state_info {
  int32_t block;                /* pivots per sweep in effect                                                      */
  int32_t chain_wgs;            /* workgroups of the last decision launch (after the residency clamp)              */
  int32_t chain_wgs_requested;  /* before the clamp                                                                */
  int32_t chain_resident_max;   /* hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs the launch could use         */
  int32_t chain_blocks_per_cu;  /* the occupancy API's answer for the decision kernel                              */
  int32_t chain_stream_masked;  /* 1: the CU-masked stream pair was created; 0: plain streams (priority only)      */
  int32_t chain_xcd_mask;       /* bit x set: a workgroup of the last decision launch ran on XCD x (HW_REG_XCC_ID) */
  int32_t sweep_xcd_mask;       /* likewise for the sampled workgroups of the last blocked sweep                   */
  int32_t overlapped;           /* 1: the last blocked loop ran decisions beside sweeps (two tableau buffers)      */
  int32_t nontemporal;          /* tableau accesses are non-temporal                                               */
  int32_t sweep_rows;           /* rows per workgroup (run length) of the last blocked sweep                       */
  int32_t sweep_kernel;         /* the kernel that swept the bulk of the tableau last (lpx_sweep_kernel_name)      */
  int32_t multi_onehop;         /* lpx_multi: 1 = the last decision launches used the one-hop exchange              */
  int32_t sweep_clock_mhz;      /* shader clock the chip held over the last pulled sweep launch (k_sweep32_pull, k_sweep64_one,
                                   k_sweep64_mfma2, ...: in-kernel s_memtime against the 100 MHz counter, one pair of stamps per XCD
                                   as the sweep starts and in the launch behind it); 0 = not measured (another sweep kernel ran last) */
  int32_t sweep_cus;            /* CUs the stream of the last blocked sweep could use (all of them outside the overlapped loop) */
  int32_t arith_fused;          /* arithmetic in effect (LPX_OPT_FUSED resolved): 0 = two roundings per update, 1 = fused multiply-add */
} lpx_state_info;
int lpx_state_get_info(lpx_state* s, lpx_state_info* out);
/* Name of a lpx_state_info.sweep_kernel code as rocprofv3 prints it ("k_sweep32_dma", "k_update_tiles", ...; "" = none). */
const char* lpx_sweep_kernel_name(int32_t code);
/* Phase timestamps of the last decision launch (LPX_OPT_CHAIN_TRACE = 1): 5 ticks (100 MHz) per decision —
 * start, phase A done, barrier passed, phase B done, next entering slot known.  ticks has room for 5 * cap
 * values; *ndecisions = decisions of the last launch that are present. */
int lpx_state_read_chain_trace(lpx_state* s, int64_t* ticks, int32_t cap, int32_t* ndecisions);
/* The same with every stamp the decision kernel keeps: *stamps per decision (k_block_chain2_t, option chain_form = 1: 8 —
 * start, phase A's loads arrived, candidate published, every candidate read, phase B's loads arrived, hand-off record
 * stored, phase B done, next entering slot known; k_block_chain_t: the 5 above).  ticks has room for 16 * cap values
 * (a diagnostic build keeps 16 stamps). */
int lpx_state_read_chain_trace_fine(lpx_state* s, int64_t* ticks, int32_t cap, int32_t* ndecisions, int32_t* stamps);

/* Pivots per pass over the tableau in lpx_simplex_loop / lpx_solve ("blocked pivoting"): K pivot decisions are
 * taken from the not-yet-updated tableau (each needs one column and one row, recovered by the pending pivots'
 * rank-1 formulas) and then applied in ONE sweep that runs every entry through the K updates in order —
 * bit-identical to K separate updates, 1/K of the HBM traffic per pivot.  0 = choose by size (default),
 * 1 = off (one update pass per pivot), 2..64 = fixed (powers of two sweep fastest; the by-size choice is 32 up to
 * ~7 GiB of tableau and 64 above: blocks of 33..64 use a two-stage sweep and 64-slot decisions, worth +3..5 % from 4 GiB;
 * shards and lpx_multi handles, and the loop with LPX_OPT_CHAIN = 0, use at most 32).  On an unsharded handle the
 * K decisions are one persistent launch and run beside the previous block's sweep, which then works out of place:
 * the handle allocates a second tableau (same size) at the first blocked loop. */
int lpx_state_set_block(lpx_state* s, int32_t pivots_per_sweep);
/* The value in effect (after the by-size choice): 1 = one update pass per pivot, K > 1 = K pivots per sweep. */
int lpx_state_get_block(const lpx_state* s);

/* The loop of LPSolver.simplex                                         LPSolver.java:101-107
 * while ((e = getEntering()) != -1) { l = getLeaving(e); if (l == -1) unbounded; pivot(e, l); }
 * run device-resident (no host decision per pivot) for at most max_pivots pivots (<0: unlimited).
 * *status = LPX_OPTIMAL | LPX_UNBOUNDED | LPX_PIVOT_LIMIT.  track_slot >= 0 follows one variable's slot
 * through the pivots exactly as solveAuxLP does for x0 (LPSolver.java:151-155); the updated slot is
 * written back to *track_slot.  Pass NULL to track nothing. */
int lpx_simplex_loop(lpx_state* s, int64_t max_pivots, int64_t* pivots_done, int32_t* status,
                     int32_t* track_slot);

/* Read-back (host pointers; any may be NULL).  A is written row-major with leading dimension lda. */
int lpx_state_read(lpx_state* s, double* A, int64_t lda, double* b, double* c, double* v,
                   int32_t* perm);
int lpx_state_dims(const lpx_state* s, int32_t* m_local, int32_t* n, int32_t* row0, int32_t* m_global);
/* Order-independent 64-bit checksums of the bit patterns of A (xor/sum of per-element hashes keyed by
 * position), b and c, computed on the device: lets full-size parity tests compare tableaux without a
 * 1 GiB read-back.  out[0..2] = A, b, c. */
int lpx_state_checksum(lpx_state* s, uint64_t out[3]);

/* Per-kernel timing of the row-update kernel with HIP events on the handle's stream (bench.py's
 * roofline figure).  enable = N > 0 brackets every N-th row-update launch with an event pair (N = 1: all;
 * an event pair costs a few microseconds of stream time, so bench.py samples sparsely); 0 stops.
 * lpx_profile_read returns the number of sampled launches and their total ms since enable, and resets. */
int lpx_profile_enable(lpx_state* s, int enable);
int lpx_profile_read(lpx_state* s, int64_t* launches, double* total_ms);

/* ------------------------------------------------------------------------------------------------
 * Row-block shards (one process per GPU; the exchange itself is done by the host with
 * torch.distributed / RCCL between these two calls).
 * Candidate record, LPX_CAND_HEADER + n doubles:
 *   [0] status flag (0 running, else an lpx_status + 1 decided from the replicated c)
 *   [1] entering slot e        [2] local best ratio (1e50 if none)   [3] its GLOBAL row (-1 if none)
 *   [4] b[row] (un-normalised) [5..7] reserved
 *   [8 .. 8+n) the un-normalised candidate pivot row A[row][0..n)
 * ---------------------------------------------------------------------------------------------- */
#define LPX_CAND_HEADER 8
/* Start (or restart) a sharded loop: reset the replicated loop state (pivot budget, tracked slot as in
 * lpx_simplex_loop), run the entering scan on the replicated c and seed this shard's pivot column and
 * ratio partials.  Every rank calls it with the same arguments.  No host sync. */
int lpx_shard_begin(lpx_state* s, int64_t max_pivots, int32_t track_slot);
/* Phase A: entering scan on the replicated c, local ratio test on the shard's slice of column e, pack the
 * shard's candidate into d_candidate (DEVICE pointer, LPX_CAND_HEADER + n doubles).  No host sync. */
int lpx_shard_propose(lpx_state* s, double* d_candidate);
/* Phase B: d_gathered holds nranks candidate records back to back (DEVICE pointer, all-gathered).
 * Every rank picks the same winner (min ratio, lowest global row), normalises the pivot row, updates its
 * replicas of c, v, perm and runs the row update on its own block.  No host sync. */
int lpx_shard_commit(lpx_state* s, const double* d_gathered, int32_t nranks);
/* The decision step of lpx_shard_commit without the row update: issued as the LAST step of a budgeted run
 * (after max_pivots commits), where it can only report LPX_UNBOUNDED or LPX_PIVOT_LIMIT.  No host sync. */
int lpx_shard_probe(lpx_state* s, const double* d_gathered, int32_t nranks);
/* Row-block shards, look-ahead form: the exchange and decision of pivot t+1 overlap the row update of
 * pivot t.  The candidate of pivot t+1 only depends on the updated tableau through one column and one row,
 * which follow from the NOT-yet-updated tableau by the rank-1 formula (bit-identical to what the update
 * writes), so per pivot t the host issues, with slot = t & 1:
 *     lpx_shard_peek(s, cand, (t+1)&1, 1)      main stream: candidate of pivot t+1 (pending update: pivot t)
 *     lpx_shard_update(s, t&1)                 main stream: row update of pivot t
 *     all-gather of cand                       comm stream (after the peek: ordered by the library)
 *     lpx_shard_decide(s, gathered, G, (t+1)&1) comm stream: pick the winner, finish pivot t+1's decision
 * after a prologue  lpx_shard_begin; lpx_shard_peek(s, cand, 0, 0); all-gather; lpx_shard_decide(s, .., 0).
 * Parameter blocks, pivot rows and pivot columns are double-buffered by slot.  The library records/waits
 * the cross-stream events; without lpx_shard_set_comm_stream everything runs on the main stream. */
int lpx_shard_set_comm_stream(lpx_state* s, void* hip_stream);
/* mode 1 (default): look-ahead as above, row update in place, the peek runs on the main stream before the
 * update.  mode 2: fully overlapped — a second tableau buffer is allocated and the row update becomes
 * out-of-place (same HBM traffic), so the peek of pivot t+1 also runs on the comm stream, beside update(t),
 * reading the buffer update(t) reads; the critical path per pivot is then the row update alone.  Same host
 * call sequence in both modes.  In mode 2 the loop must be polled to a final status before the state is
 * read or another loop is started (the library then points the handle at the buffer holding the result). */
int lpx_shard_set_pipeline(lpx_state* s, int32_t mode);
int lpx_shard_peek(lpx_state* s, double* d_candidate, int32_t slot, int32_t pending);
int lpx_shard_decide(lpx_state* s, const double* d_gathered, int32_t nranks, int32_t slot);
int lpx_shard_update(lpx_state* s, int32_t slot);
/* Row-block shards, blocked form (see lpx_state_set_block): per block of K <= 32 decisions the host issues, for
 * slot = 0..K-1,   lpx_shard_block_peek(s, cand, slot) -> all-gather of cand -> lpx_shard_block_decide(s, gathered,
 * G, slot),   then ONE lpx_shard_block_sweep(s, K) that applies the (valid) decided pivots to the shard's rows in
 * a single pass.  After lpx_shard_begin; everything on the handle's main stream; no host sync. */
int lpx_shard_block_peek(lpx_state* s, double* d_candidate, int32_t slot);
int lpx_shard_block_decide(lpx_state* s, const double* d_gathered, int32_t nranks, int32_t slot);
int lpx_shard_block_sweep(lpx_state* s, int32_t nslots);
/* Host poll of the replicated loop state: pivots done so far and LPX_RUNNING (-1, loop still live) or
 * LPX_OPTIMAL / LPX_UNBOUNDED / LPX_PIVOT_LIMIT.  Synchronises the stream. */
#define LPX_RUNNING (-1)
int lpx_shard_poll(lpx_state* s, int64_t* pivots_done, int32_t* status);

/* ------------------------------------------------------------------------------------------------
 * LPSolver.solve                                                       LPSolver.java:78-94
 * ---------------------------------------------------------------------------------------------- */
typedef struct lpx_solve_result {
  int32_t status;          /* lpx_status */
  int32_t phase1_used;     /* 1 if the auxiliary LP was needed (min b < 0)          LPSolver.java:119-131 */
  double objective;        /* unrounded LPState.v, sign-corrected for minimisation                        */
  double objective_rounded;/* v.setScale(6, HALF_UP) as the nearest double           LPSolver.java:113    */
  char objective_text[64]; /* the same as decimal text with 6 fractional digits ("8.000000")              */
  int64_t pivots_phase1;   /* incl. the forced first pivot and the degenerate pivot  LPSolver.java:138,195 */
  int64_t pivots_phase2;
  int32_t x0_slot;         /* final aux-LP slot of x0 (-1 if phase 1 unused)         LPSolver.java:162    */
  int32_t reserved;
  double seconds_total;    /* wall time inside lpx_solve, upload included                                 */
  double seconds_pivots;   /* wall time of the device pivot loops only                                    */
} lpx_solve_result;

typedef struct lpx_solve_options {
  int32_t device;               /* HIP device ordinal                                                     */
  int32_t has_variable_names;   /* LPStandardForm.hasVariableNames(); only affects nothing numerically,   */
                                /* kept for shim symmetry                                                 */
  int64_t max_pivots;           /* <0: unlimited (the reference has no limit)                             */
  const int32_t* restore_order; /* iteration order of initial.coefficients.keySet() in restoreInitialLP   */
                                /* (LPSolver.java:213-217) as original-variable indices; NULL = the order */
                                /* java.util.HashMap gives the default names "x1".."xn" (:388-400)        */
  int32_t* perm_out;            /* optional int32[n+m]: final slot -> variable id.  Written only when the result */
                                /* is an m x n state (phase 2 reached): a solve that ends inside phase 1        */
                                /* (INFEASIBLE, AUX_UNBOUNDED, ...) leaves it untouched                          */
  double* x_out;                /* optional double[n]: primal solution (basic slot -> b[i], else 0)       */
  lpx_state** keep_state;       /* optional: receive the final LPState handle instead of destroying it    */
  int32_t pricing;              /* 0 = the reference's entering rule (default); 1 = Dantzig, see            */
                                /* lpx_state_set_pricing                                                   */
  int32_t restore_order_len;    /* entries of restore_order; < 0: n (every original variable has a name).  With a
                                 * non-NULL restore_order, 0 means NO variable is substituted (an empty keySet());
                                 * a NULL restore_order selects the default-name order over all n variables          */
  int32_t fused;                /* arithmetic of the handle(s) of this solve: 0 = the library's choice by size (LPX_OPT_FUSED = 2),
                                 * 1 = fused multiply-add updates, -1 = two roundings per update (the opt-out)         */
  int32_t reserved;
} lpx_solve_options;

/* BigDecimal LPSolver.solve(LPStandardForm stForm)                      LPSolver.java:78
 * A (lda), b, c are HOST arrays of the standard form  max/min c.x  s.t.  A x <= b, x >= 0
 * (LPStandardForm.java:11-16).  Unlike the reference, the caller's arrays are never modified (the
 * reference negates stForm.c in place for `min`, :86-89, and pivots inside stForm.A/b/c, :267).
 * Returns LPX_OPTIMAL (0) or the status that the reference would have thrown as an exception;
 * result->status carries the same code. */
int lpx_solve(int32_t m, int32_t n, const double* A, int64_t lda, const double* b, const double* c,
              int32_t maximize, const lpx_solve_options* opts, lpx_solve_result* result);

/* ------------------------------------------------------------------------------------------------
 * Several GPUs of one node behind ONE handle (SURVEY §8e, §8b "a handle owns its device set"): the tableau is cut
 * into contiguous row blocks — device r holds rows [r*m/G, (r+1)*m/G), the partition pivotConcurrently uses for its
 * row phase (LPState.java:222-223) — c, v, perm and the loop state are replicated.  One process, one host thread;
 * peer access (hipDeviceEnablePeerAccess) between all devices of the set.  Per pivot decision every device runs
 * the same persistent decision kernel on its rows and exchanges, by direct stores into its peers' memory over xGMI,
 * (1) its 32-byte minimum-ratio candidate {ratio, row, pivot element, b[row]} with every device — the
 * allreduce(min+loc), lowest global row winning ties — and (2) the normalised pivot row, n doubles, from the device
 * that owns the leaving row to all others; K decisions, then one K-fold sweep of each device's own rows.  No host
 * decision, no collective library, no Python.  `devices` may name the same ordinal more than once (the shards then
 * share that GPU: how the path is rehearsed on a one-GPU machine; their decision kernels wait for each other, so the
 * runtime must give each its own hardware queue — GPU_MAX_HW_QUEUES >= n_dev in the environment, default 4 — or the
 * loop ends with LPX_DEVICE_ERROR after a bounded spin).  Results are bit-identical to the one-device handle: every
 * entry sees the same operations in the same order.
 * ---------------------------------------------------------------------------------------------- */
#define LPX_MAX_DEVICES 8
typedef struct lpx_multi lpx_multi;
/* new LPState(...) over n_dev devices; arguments as lpx_state_create (HOST arrays, A is m x n with leading
 * dimension lda).  1 <= n_dev <= LPX_MAX_DEVICES, n_dev <= max(m, 1). */
int lpx_multi_create(int32_t m, int32_t n, const double* A, int64_t lda, const double* b, const double* c, double v,
                     const int32_t* perm, const int32_t* devices, int32_t n_dev, lpx_multi** out);
void lpx_multi_destroy(lpx_multi* s);
int lpx_multi_set_option(lpx_multi* s, int32_t key, int64_t value);   /* lpx_state_set_option on every shard */
int lpx_multi_set_pricing(lpx_multi* s, int32_t pricing);
/* getEntering / getLeaving / pivot and the loop of LPSolver.simplex, as lpx_get_entering ... lpx_simplex_loop */
int lpx_multi_get_entering(lpx_multi* s, int32_t* entering);
int lpx_multi_get_leaving(lpx_multi* s, int32_t entering, int32_t* leaving, double* ratio);
int lpx_multi_pivot(lpx_multi* s, int32_t entering, int32_t leaving);
int lpx_multi_simplex_loop(lpx_multi* s, int64_t max_pivots, int64_t* pivots_done, int32_t* status,
                           int32_t* track_slot);
/* Read-back of the whole tableau (rows gathered from the shards) / position-keyed checksums as lpx_state_checksum */
int lpx_multi_read(lpx_multi* s, double* A, int64_t lda, double* b, double* c, double* v, int32_t* perm);
int lpx_multi_checksum(lpx_multi* s, uint64_t out[3]);
/* Sweep timing of shard `shard` as lpx_profile_enable / lpx_profile_read; what shard 0 did as lpx_state_get_info */
int lpx_multi_profile_enable(lpx_multi* s, int enable);
int lpx_multi_profile_read(lpx_multi* s, int32_t shard, int64_t* launches, double* total_ms);
int lpx_multi_get_info(lpx_multi* s, lpx_state_info* out);
/* BigDecimal LPSolver.solve(LPStandardForm stForm) over n_dev devices: lpx_solve with the row blocks of the tableau
 * on several GPUs, phase 1 (auxiliary LP, forced first pivot, x0 tracking, degenerate pivot, column drop and
 * objective restore — LPSolver.java:116-246) included.  opts->device is ignored; keep_state is not supported. */
int lpx_solve_multi(int32_t m, int32_t n, const double* A, int64_t lda, const double* b, const double* c,
                    int32_t maximize, const lpx_solve_options* opts, const int32_t* devices, int32_t n_dev,
                    lpx_solve_result* result);

/* LPState restoreInitialLP(auxLP, initial, indexOfX0)                   LPSolver.java:200-246
 * In place on the auxiliary-LP handle (m x (n+1), as left by phase 1): drops x0's column, rebuilds c and v by
 * substitution in keySet() order (`order`, order_len <= n original-variable indices; NULL = default-name order of all n), renumbers
 * the slots above x0; the handle then is the m x n LPState the reference constructs at :245.  Bug-for-bug with
 * the reference (a nonbasic original variable is credited at its aux-LP slot; slot n returns
 * LPX_RESTORE_INDEX_FAULT).  c0 = initial.c (already negated for `min`). */
int lpx_restore_initial_lp(lpx_state* aux, const double* c0, int32_t n, int32_t x0_slot, const int32_t* order,
                           int32_t order_len);

/* Iteration order of a java.util.HashMap<String,Integer> filled by put("x1"), put("x2"), ... put("xn")
 * into a default-constructed map (LPSolver.addDefaultVariables, LPSolver.java:388-400): writes the
 * 0-based variable indices in keySet() order.  Host-only helper (no device work). */
int lpx_java_default_name_order(int32_t n, int32_t* order_out);

/* LPStandardForm.getDual()                                              LPStandardForm.java:129-152
 * Device transpose: At (n x m, leading dimension ldat) = transpose of A (m x n, lda).  HOST pointers. */
int lpx_transpose(int32_t m, int32_t n, const double* A, int64_t lda, double* At, int64_t ldat, int device);

#ifdef __cplusplus
}
#endif
#endif /* LPX_H */
