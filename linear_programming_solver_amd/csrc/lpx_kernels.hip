// HIP kernels of the dense simplex pivot loop for gfx950 (MI355X / CDNA4).
//
// One pivot of the reference (LPState.java:114-181, :274-305, :311-320) is two launches:
//
//   k_select_pivot (1 workgroup)  finishes the minimum-ratio test from per-tile partials -> leaving row l,
//                                 normalises the pivot row (:139-146), updates the objective row, v and the
//                                 slot permutation (:170-180, :311-320) and picks the NEXT entering slot
//                                 (first c[j] > 1e-9, :274-285);
//   k_update      (whole chip)    the rank-1 update of every other row and of b (:151-166): each fp64
//                                 tableau entry is read once and written once (16 B/entry, HBM-bound,
//                                 0.125 flop/B -> no MFMA), and — because the next entering slot is already
//                                 known — the thread that owns that column also emits the next pivot column
//                                 and the per-tile partial of the next ratio test (:287-305) in the same pass.
//
// Arithmetic is IEEE fp64.  This file is compiled TWICE, into two namespaces, and a handle picks one (LPX_OPT_FUSED,
// Buffers::fused; lpx_dispatch.cpp):
//   lpxk::plain (LPX_FUSED = 0, the default)  exactly one rounding per reference operation — the reference rounds the
//       product and the difference of :162 separately, so the update is __dmul_rn then __dsub_rn, never an FMA;
//   lpxk::fused (LPX_FUSED = 1, opt-in)       every update x - c*r (:162, :164, :177) and x + a*b (:171; LPSolver.java
//       :223, :227) is ONE v_fma_f64: half the fp64 instructions of the sweep (which is at its VALU instruction floor
//       in the plain form, profiles/r03_pmc) and one binary rounding instead of two where the reference has two decimal
//       ones.  Checked bit for bit against the oracle's Num<F64Fused> instantiation.
// The two differ in submul() / addmul() below and nowhere else; -ffp-contract=off keeps the compiler from forming
// (or splitting) anything on its own.
#include "lpx_kernels.h"

#include <hip/hip_ext.h>

#include <atomic>

#include <limits.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#ifndef LPX_FUSED
#define LPX_FUSED 0
#endif

namespace lpxk {
#if LPX_FUSED
namespace fused {
#else
namespace plain {
#endif

typedef double d2 __attribute__((ext_vector_type(2)));  // one 16-byte global access per lane

// x - c*r and x + a*b: the ONLY place where the two arithmetic modes differ
__device__ __forceinline__ double submul(double x, double c, double r) {
#if LPX_FUSED
  return __fma_rn(-c, r, x);
#else
  return __dsub_rn(x, __dmul_rn(c, r));
#endif
}
__device__ __forceinline__ double addmul(double x, double a, double b) {
#if LPX_FUSED
  return __fma_rn(a, b, x);
#else
  return __dadd_rn(x, __dmul_rn(a, b));
#endif
}

// ------------------------------------------------------------------------------------------------ helpers
__device__ __forceinline__ RatioRow rr_none() { return RatioRow{kInf, INT_MAX, 0}; }

// Lexicographic min on (ratio, row): the sequential scan of LPState.java:292-303 keeps the first row that
// is STRICTLY smaller than everything before it, i.e. the lowest row among equal minimal ratios.
__device__ __forceinline__ RatioRow rr_min(RatioRow a, RatioRow b) {
  const bool take_b = (b.ratio < a.ratio) || (b.ratio == a.ratio && b.row < a.row);
  return take_b ? b : a;
}

__device__ __forceinline__ RatioRow rr_wave_min(RatioRow x) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    RatioRow y;
    y.ratio = __shfl_down(x.ratio, off, 64);
    y.row = __shfl_down(x.row, off, 64);
    y.pad = 0;
    x = rr_min(x, y);
  }
  return x;
}

// Block-wide lexmin; result valid in every thread.  `sh` needs blockDim.x/64 entries.
__device__ __forceinline__ RatioRow rr_block_min(RatioRow x, RatioRow* sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  x = rr_wave_min(x);
  __syncthreads();
  if (lane == 0) sh[wave] = x;
  __syncthreads();
  RatioRow r = sh[0];
  for (int w = 1; w < nw; ++w) r = rr_min(r, sh[w]);
  return r;
}

__device__ __forceinline__ int block_min_int(int x, int* sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) x = min(x, __shfl_down(x, off, 64));
  __syncthreads();
  if (lane == 0) sh[wave] = x;
  __syncthreads();
  int r = sh[0];
  for (int w = 1; w < nw; ++w) r = min(r, sh[w]);
  return r;
}

// ratio of one row: LPState.java:293-298
__device__ __forceinline__ double ratio_of(double a, double bi) {
  return (a < kEps) ? kInf : __ddiv_rn(bi, a);
}

// ------------------------------------------------------------------------------------------------ k_entering
// getEntering(): first slot with c[j] > 1e-9 (LPState.java:274-285).  One workgroup.
// start.reset = 1 (first launch of a loop call): the loop state starts over here — status, counters, budget, tracked slot —
// instead of by a read-modify-write of the host mirror (two small copies and a host round trip in front of every call).
__device__ __forceinline__ void loop_start(LpxCtl* ctl, const LoopStart& st) {
  if (threadIdx.x == 0) {
    ctl->status = kRunning;
    ctl->do_update = 0;
    ctl->pivots = 0;
    ctl->max_pivots = st.max_pivots;
    ctl->track = st.track;
    ctl->e_min = INT_MAX;
    ctl->ticket = 0;
  }
  __syncthreads();
}
__global__ __launch_bounds__(1024) void k_entering(const double* __restrict__ c, int n, LpxCtl* ctl, const LoopStart start) {
  __shared__ int sh[16];
  if (start.reset) loop_start(ctl, start);
  else if (ctl->status != kRunning) return;
  int best = INT_MAX;
  for (int j = threadIdx.x; j < n; j += blockDim.x)
    if (c[j] > kEps) { best = j; break; }  // per-thread indices ascend, the first hit is this thread's min
  best = block_min_int(best, sh);
  if (threadIdx.x == 0) {
    ctl->e_next = (best == INT_MAX) ? -1 : best;
    if (best == INT_MAX) ctl->status = 0 /* LPX_OPTIMAL */;
  }
}

// Opt-in Dantzig pricing (an extension of this build; NOT the reference's rule): entering
// slot = argmax c[j] over c[j] > 1e-9, lowest slot on ties.  Runs as one extra small launch after the pivot
// decision and overrides ctl->e_next (k_select_pivot / k_commit have already set status = OPTIMAL when no
// c[j] > 1e-9 exists, which is rule-independent).  `seed` = 1: also decide OPTIMAL (start of a loop).
__global__ __launch_bounds__(1024) void k_entering_dantzig(const double* __restrict__ c, int n, LpxCtl* ctl,
                                                           int seed, const LoopStart start) {
  __shared__ RatioRow sh[16];
  if (start.reset) loop_start(ctl, start);
  else if (ctl->status != kRunning) return;
  RatioRow best{-kInf, INT_MAX, 0};  // reuse the (value,index) lexmin machinery on (-c[j], j)
  best.ratio = kInf;
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const double cj = c[j];
    if (cj > kEps) {
      const RatioRow cand{-cj, j, 0};
      if (cand.ratio < best.ratio) best = cand;  // j ascends per thread: strict < keeps the lowest slot
    }
  }
  best = rr_block_min(best, sh);
  if (threadIdx.x == 0) {
    const bool none = best.row == INT_MAX;
    ctl->e_next = none ? -1 : best.row;
    if (none && seed) ctl->status = 0 /* LPX_OPTIMAL */;
  }
}

// ------------------------------------------------------------------------------------------------ k_ratio_gather
// Seeds the pipeline: strided gather of column e_next into col[parity] plus the per-tile partials of
// getLeaving (LPState.java:287-305).  Only used for the first pivot of a loop and by the step API; inside
// the loop k_update produces both as a by-product.
__global__ __launch_bounds__(256) void k_ratio_gather(const double* __restrict__ A, int64_t ld,
                                                      const double* __restrict__ b, int m_local, int row0,
                                                      double* col0, double* col1, RatioRow* partial,
                                                      const LpxCtl* __restrict__ ctl, int nparts,
                                                      int forced_e) {
  __shared__ RatioRow sh[4];
  if (ctl->status != kRunning) return;
  const int e = forced_e >= 0 ? forced_e : ctl->e_next;
  if (e < 0) return;
  double* col = ctl->parity ? col1 : col0;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // one row per thread
  RatioRow best = rr_none();
  if (i < m_local) {
    const double a = A[(int64_t)i * ld + e];
    col[i] = a;
    const double r = ratio_of(a, b[i]);
    if (r < kInf) best = RatioRow{r, row0 + i, 0};
  }
  best = rr_block_min(best, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = best;
  // the consumers fold all `nparts` (= k_update's tile count) slots: blank the ones this grid does not produce
  for (int k = gridDim.x + blockIdx.x * blockDim.x + threadIdx.x; k < nparts; k += gridDim.x * blockDim.x)
    partial[k] = rr_none();
}

// getLeaving() of the step API: fold the partials into ctl->l / ctl->ratio, no pivot.
__global__ __launch_bounds__(256) void k_reduce_partials(const RatioRow* __restrict__ partial, int nparts,
                                                         LpxCtl* ctl) {
  __shared__ RatioRow sh[4];
  RatioRow best = rr_none();
  for (int k = threadIdx.x; k < nparts; k += blockDim.x) best = rr_min(best, partial[k]);
  best = rr_block_min(best, sh);
  if (threadIdx.x == 0) {
    ctl->l = (best.ratio < kInf) ? best.row : -1;
    ctl->ratio = best.ratio;
  }
}

// ------------------------------------------------------------------------------------------------ pivot finish
// Shared tail of k_select_pivot (one GPU) and k_commit (shards): given the winning row (raw, un-normalised)
// normalise it into prow (LPState.java:139-146), update c, v, perm (:170-180, :311-320), follow the tracked
// slot (LPSolver.java:151-155) and choose the next entering slot (:274-285).
//
// Runs on gridDim.x <= 16 workgroups of 1024: every workgroup derives the same (e, l, p) from the same
// inputs, then owns a grid-strided slice of the columns (one fp64 division per column is the expensive
// part).  The next entering slot is the minimum over workgroups of "first improving column": each
// workgroup folds its candidate into ctl->e_min with a returning device-scope atomicMin and then takes a
// ticket (the ticket's operand depends on the atomic's return value, so the min is performed first); the
// workgroup that draws the last ticket finalises the replicated loop state.  Nobody spins, so no residency
// assumption is needed.  ctl fields are only ever written by that last workgroup (or by workgroup 0 on
// the early exits that all workgroups take alike), after every workgroup has read what it needs.
//
// `up` is the parameter block the row update of THIS pivot will read: ctl itself in the two-launch loop,
// one slot of a ring in the look-ahead pipeline (where the decision for pivot t+1 is taken while the update
// of pivot t is still streaming, see k_peek); up_parity >= 0 then fixes which col buffer that update reads.
__device__ __forceinline__ void finish_pivot(const double* __restrict__ raw_row, double raw_b, int e,
                                             int l_global, double ratio, double* __restrict__ prow,
                                             double* __restrict__ c, int n, int64_t ld, int32_t* perm,
                                             LpxCtl* ctl, LpxCtl* up, int up_parity, int* sh_int) {
  const double p = raw_row[e];
  if (p == 0.0) {  // ArithmeticException in the reference (BigDecimal.divide by zero), :139
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      ctl->status = 8 /* LPX_DIVIDE_BY_ZERO */;
      ctl->do_update = 0;
      up->do_update = 0;
    }
    return;
  }
  const double pc = c[e];  // c[e] itself is rewritten only by the finalising workgroup
  const double bl = __ddiv_rn(raw_b, p);                                           // :146
  const double inv_p = __ddiv_rn(1.0, p);                                          // :139
  int first_pos = INT_MAX;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < (int)ld; j += gridDim.x * blockDim.x) {
    if (j == e) {
      prow[j] = inv_p;
    } else {
      const double x = (j < n) ? raw_row[j] : 0.0;
      const double pr = __ddiv_rn(x, p);                                           // :144
      const double cn = submul(c[j], pc, pr);                        // :177
      prow[j] = pr;
      c[j] = cn;
      if (j < n && cn > kEps && first_pos == INT_MAX) first_pos = j;  // j ascends per thread
    }
  }
  // thread 0 fetches what the finalisation needs while the block reduction runs (saves dependent round trips)
  int32_t perm_e = 0, perm_l = 0, track = -1;
  double v_old = 0.0;
  if (threadIdx.x == 0) {
    perm_e = perm[e];
    perm_l = perm[n + l_global];
    track = ctl->track;
    v_old = ctl->v;
  }
  first_pos = block_min_int(first_pos, sh_int);
  if (threadIdx.x == 0) {
    bool last = true;
    int e_min = first_pos;
    if (gridDim.x > 1) {  // one workgroup (small n): nothing to fold, no atomics
      int old = 0;
      if (first_pos != INT_MAX) old = atomicMin(&ctl->e_min, first_pos);
      int inc = 1;
      asm volatile("" : "+v"(inc) : "v"(old));  // the ticket below is issued after the min has returned
      const int ticket = atomicAdd(&ctl->ticket, inc);
      last = ticket == (int)gridDim.x - 1;
      if (last) e_min = atomicMin(&ctl->e_min, INT_MAX);  // returning atomic: the folded minimum
    }
    if (last) {
      const double ce_new = -__ddiv_rn(pc, p);                                     // :172
      c[e] = ce_new;
      if (e < n && ce_new > kEps) e_min = min(e_min, e);  // possible on forced / degenerate pivots only
      ctl->v = addmul(v_old, bl, pc);                                // :171
      perm[e] = perm_l;                                                            // exchangeIndexes :311-320
      perm[n + l_global] = perm_e;
      if (track >= 0) {                                                            // LPSolver.java:151-155
        if (e == track) ctl->track = l_global + n;
        else if (l_global + n == track) ctl->track = e;
      }
      ctl->p = p;
      ctl->bl = bl;
      ctl->pc = pc;
      ctl->ratio = ratio;
      ctl->e_cur = e;
      ctl->l = l_global;
      ctl->e_next = (e_min == INT_MAX) ? -1 : e_min;
      if (e_min == INT_MAX) ctl->status = 0 /* LPX_OPTIMAL: reached after k_update applies this pivot */;
      ctl->parity ^= 1;  // k_update reads col[parity^1] (column e_cur) and fills col[parity] (column e_next)
      ctl->pivots += 1;
      ctl->do_update = 1;
      if (up != ctl) {  // look-ahead ring slot: only what k_update / k_peek read
        up->p = p;
        up->bl = bl;
        up->e_cur = e;
        up->l = l_global;
        up->e_next = -1;           // no by-products: the next column / ratios come from k_peek
        up->parity = up_parity;
        up->do_update = 1;
      }
      if (gridDim.x > 1) {
        atomicExch(&ctl->e_min, INT_MAX);
        atomicExch(&ctl->ticket, 0);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ k_select_pivot
__global__ __launch_bounds__(1024) void k_select_pivot(const double* __restrict__ A, int64_t ld, int n,
                                                       int m_global, const double* __restrict__ b, double* c,
                                                       double* prow, const RatioRow* __restrict__ partial,
                                                       int nparts, int32_t* perm, LpxCtl* ctl, int forced_e,
                                                       int forced_l) {
  __shared__ RatioRow sh_rr[16];
  __shared__ int sh_int[16];
  const bool writer = blockIdx.x == 0 && threadIdx.x == 0;
  if (ctl->status != kRunning) {
    if (writer) ctl->do_update = 0;
    return;
  }
  int e, l;
  double ratio = 0.0;
  if (forced_l >= 0) {  // pivot(entering, leaving) of the step API
    e = forced_e;
    l = forced_l;
  } else {
    e = ctl->e_next;
    const int64_t pivots = ctl->pivots, max_pivots = ctl->max_pivots;
    RatioRow best = rr_none();
    for (int k = threadIdx.x; k < nparts; k += blockDim.x) best = rr_min(best, partial[k]);
    best = rr_block_min(best, sh_rr);
    l = (best.ratio < kInf) ? best.row : -1;
    ratio = best.ratio;
    if (l < 0) {  // getLeaving() == -1: unbounded (LPSolver.java:103-106 / :147-150)
      if (writer) { ctl->status = 1 /* LPX_UNBOUNDED */; ctl->do_update = 0; ctl->l = -1; ctl->ratio = best.ratio; }
      return;
    }
    if (max_pivots >= 0 && pivots >= max_pivots) {
      if (writer) { ctl->status = 9 /* LPX_PIVOT_LIMIT */; ctl->do_update = 0; }
      return;
    }
  }
  (void)m_global;
  finish_pivot(A + (int64_t)l * ld, b[l], e, l, ratio, prow, c, n, ld, perm, ctl, ctl, -1, sh_int);
}

// ------------------------------------------------------------------------------------------------ k_update
// The row update (LPState.java:151-166), the kernel the HBM roofline is quoted on.
//
// Work split: workgroup = (tile of rows_per_tile rows) x (strip of 512*U columns); thread t owns the U
// double2 columns  strip*512*U + k*512 + 2t  (k = 0..U-1), so every wave-level access is 64 x 16 B = 1 KiB
// contiguous and the thread's slice of the pivot row stays in registers for the whole tile: the pivot row
// is read once per tile from L2, the multiplier col[i] is one scalar load per row, and every tableau entry
// moves HBM -> register -> HBM exactly once.  Columns [n, ld) are zero padding that the update maps to zero.
//
// The thread that owns column e_next (or column 0 when the pivot being applied ends the loop) also owns
// the b update, writes the next pivot column into col[parity] and reduces its rows' ratios into
// partial[tile] — the next getLeaving() costs no extra pass over the tableau.
//
// OOP = out-of-place: read the tableau from (Asrc, bsrc) and write the updated one to (A, b).  Same HBM traffic
// (every entry read once, written once); used by the fully overlapped pipeline, where the decision of pivot
// t+1 reads the un-updated tableau WHILE this kernel streams, which an in-place update cannot allow.
template <int U, bool NT, bool OOP>
__global__ __launch_bounds__(256) void k_update(double* __restrict__ A, const double* __restrict__ Asrc_, int64_t ld,
                                                int m_local, int row0, double* __restrict__ b,
                                                const double* __restrict__ bsrc_, const double* __restrict__ prow,
                                                double* col0, double* col1, RatioRow* __restrict__ partial,
                                                const LpxCtl* __restrict__ ctl, int rows_per_tile,
                                                int nstrips) {
  if (ctl->do_update == 0) return;
  const double* Asrc = OOP ? Asrc_ : A;
  const double* bsrc = OOP ? bsrc_ : b;
  const int strip = blockIdx.x % nstrips;
  const int tile = blockIdx.x / nstrips;
  const int e = ctl->e_cur;
  const int en = ctl->e_next;
  const int l = ctl->l - row0;  // local index of the pivot row; outside [0, m_local) on other shards
  const double p = ctl->p;
  const double bl = ctl->bl;
  const double* __restrict__ colcur = ctl->parity ? col0 : col1;  // column e_cur (old values)
  double* __restrict__ colnxt = ctl->parity ? col1 : col0;        // receives column e_next (new values)

  const int cbase = strip * (512 * U) + 2 * threadIdx.x;
  d2 pr[U];
  bool act[U];
  int eslot = -1, oslot = -1;
  const int oc = en >= 0 ? en : 0;  // owner column of the b update / next-column emission
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const int cj = cbase + k * 512;
    act[k] = cj < (int)ld;
    pr[k] = act[k] ? *reinterpret_cast<const d2*>(prow + cj) : d2{0.0, 0.0};
    if (cj == e) eslot = 2 * k;
    if (cj + 1 == e) eslot = 2 * k + 1;
    if (cj == oc) oslot = 2 * k;
    if (cj + 1 == oc) oslot = 2 * k + 1;
  }

  const int r_begin = tile * rows_per_tile;
  const int r_end = min(m_local, r_begin + rows_per_tile);
  RatioRow best = rr_none();

  // two rows per iteration: 2U independent 16-byte loads in flight per thread before the first store
  for (int i0 = r_begin; i0 < r_end; i0 += 2) {
    const int i1 = i0 + 1;
    const bool has1 = i1 < r_end;
    double* row0p = A + (int64_t)i0 * ld;
    double* row1p = A + (int64_t)i1 * ld;
    const double* src0p = Asrc + (int64_t)i0 * ld;
    const double* src1p = Asrc + (int64_t)i1 * ld;
    const double ce0 = colcur[i0];
    const double ce1 = has1 ? colcur[i1] : 0.0;
    d2 x0[U], x1[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      if (act[k]) {
        const d2* q0 = reinterpret_cast<const d2*>(src0p + cbase + k * 512);
        x0[k] = NT ? __builtin_nontemporal_load(q0) : *q0;
      }
    }
    if (has1) {
#pragma unroll
      for (int k = 0; k < U; ++k) {
        if (act[k]) {
          const d2* q1 = reinterpret_cast<const d2*>(src1p + cbase + k * 512);
          x1[k] = NT ? __builtin_nontemporal_load(q1) : *q1;
        }
      }
    }
    // ---- row i0
    if (i0 == l) {
#pragma unroll
      for (int k = 0; k < U; ++k) x0[k] = pr[k];                                  // pivot row := normalised row
    } else {
#pragma unroll
      for (int k = 0; k < U; ++k) {
        x0[k].x = submul(x0[k].x, ce0, pr[k].x);                     // :162
        x0[k].y = submul(x0[k].y, ce0, pr[k].y);
      }
      if (eslot >= 0) {                                                            // :157
        const double ne = -__ddiv_rn(ce0, p);
#pragma unroll
        for (int k = 0; k < U; ++k) {
          if (eslot == 2 * k) x0[k].x = ne;
          if (eslot == 2 * k + 1) x0[k].y = ne;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      if (act[k]) {
        d2* q0 = reinterpret_cast<d2*>(row0p + cbase + k * 512);
        if (NT) __builtin_nontemporal_store(x0[k], q0); else *q0 = x0[k];
      }
    }
    // ---- row i1
    if (has1) {
      if (i1 == l) {
#pragma unroll
        for (int k = 0; k < U; ++k) x1[k] = pr[k];
      } else {
#pragma unroll
        for (int k = 0; k < U; ++k) {
          x1[k].x = submul(x1[k].x, ce1, pr[k].x);
          x1[k].y = submul(x1[k].y, ce1, pr[k].y);
        }
        if (eslot >= 0) {
          const double ne = -__ddiv_rn(ce1, p);
#pragma unroll
          for (int k = 0; k < U; ++k) {
            if (eslot == 2 * k) x1[k].x = ne;
            if (eslot == 2 * k + 1) x1[k].y = ne;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < U; ++k) {
        if (act[k]) {
          d2* q1 = reinterpret_cast<d2*>(row1p + cbase + k * 512);
          if (NT) __builtin_nontemporal_store(x1[k], q1); else *q1 = x1[k];
        }
      }
    }
    // ---- owner thread: b update (:164 / :146), next pivot column, next ratio partial (:293-302)
    if (oslot >= 0) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int i = r ? i1 : i0;
        if (r && !has1) break;
        const double ce = r ? ce1 : ce0;
        const double bn = (i == l) ? bl : submul(bsrc[i], ce, bl);
        b[i] = bn;
        if (en >= 0) {
          double a = 0.0;
#pragma unroll
          for (int k = 0; k < U; ++k) {
            const d2 xv = r ? x1[k] : x0[k];
            if (oslot == 2 * k) a = xv.x;
            if (oslot == 2 * k + 1) a = xv.y;
          }
          colnxt[i] = a;
          const double ratio = ratio_of(a, bn);
          if (ratio < best.ratio) best = RatioRow{ratio, row0 + i, 0};  // rows ascend: lowest row wins ties
        }
      }
    }
  }
  if (oslot >= 0 && en >= 0) partial[tile] = best;
}

// ------------------------------------------------------------------------------------------------ shards
// k_propose: fold this shard's partials into its candidate and pack {header, raw row} for the all-gather.
// Every workgroup reduces the partials alike; workgroup 0 writes the header, all copy a slice of the row.
__global__ __launch_bounds__(1024) void k_propose(const double* __restrict__ A, int64_t ld, int n, int row0,
                                                  int m_local, const double* __restrict__ b,
                                                  const RatioRow* __restrict__ partial, int nparts,
                                                  const LpxCtl* __restrict__ ctl, double* __restrict__ cand) {
  __shared__ RatioRow sh_rr[16];
  const int st = ctl->status;
  RatioRow best = rr_none();
  if (st == kRunning) {
    for (int k = threadIdx.x; k < nparts; k += blockDim.x) best = rr_min(best, partial[k]);
    best = rr_block_min(best, sh_rr);
  }
  const bool have = best.ratio < kInf;
  const int lr = have ? best.row - row0 : -1;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    cand[0] = (st == kRunning) ? 0.0 : (double)(st + 1);
    cand[1] = (double)ctl->e_next;
    cand[2] = best.ratio;
    cand[3] = have ? (double)best.row : -1.0;
    cand[4] = have ? b[lr] : 0.0;
    cand[5] = cand[6] = cand[7] = 0.0;
  }
  if (have) {
    const double* row = A + (int64_t)lr * ld;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) cand[8 + j] = row[j];
  }
  (void)m_local;
}

// k_commit: identical on every rank — pick the winner (min ratio, lowest global row) and finish the pivot.
__global__ __launch_bounds__(1024) void k_commit(const double* __restrict__ gathered, int nranks, int n,
                                                 int64_t ld, int m_global, double* c, double* prow,
                                                 int32_t* perm, LpxCtl* ctl, LpxCtl* up, int up_parity) {
  __shared__ int sh_int[16];
  const bool writer = blockIdx.x == 0 && threadIdx.x == 0;
  if (ctl->status != kRunning) {
    if (writer) { ctl->do_update = 0; up->do_update = 0; }
    return;
  }
  const int64_t pivots = ctl->pivots, max_pivots = ctl->max_pivots;
  const int e = ctl->e_next;
  const int64_t rec = 8 + (int64_t)n;
  RatioRow best = rr_none();
  int win = -1;
  for (int r = 0; r < nranks; ++r) {  // nranks <= 8: every thread scans the headers
    const double* h = gathered + r * rec;
    if (h[3] >= 0.0) {
      const RatioRow cr{h[2], (int32_t)h[3], 0};
      const bool take = (cr.ratio < best.ratio) || (cr.ratio == best.ratio && cr.row < best.row);
      if (take) { best = cr; win = r; }
    }
  }
  if (win < 0 || !(best.ratio < kInf)) {
    if (writer) {
      ctl->status = 1 /* LPX_UNBOUNDED */; ctl->do_update = 0; ctl->l = -1; ctl->ratio = best.ratio;
      up->do_update = 0;
    }
    return;
  }
  if (max_pivots >= 0 && pivots >= max_pivots) {
    if (writer) { ctl->status = 9 /* LPX_PIVOT_LIMIT */; ctl->do_update = 0; up->do_update = 0; }
    return;
  }
  const double* h = gathered + win * rec;
  (void)m_global;
  finish_pivot(h + 8, h[4], e, best.row, best.ratio, prow, c, n, ld, perm, ctl, up, up_parity, sh_int);
}

// ------------------------------------------------------------------------------------------------ look-ahead
// The ratio test of pivot t+1 and the winning row itself depend on the tableau AFTER pivot t only through
// one column and one row, and both follow from the tableau BEFORE pivot t by the rank-1 formula:
//     A'[i][e'] = A[i][e'] - col[i]*prow[e']   (i != l; = prow[e'] for i == l; = -(col[i]/p) for e' == e)
//     b'[i]     = b[i] - col[i]*b_l            (i != l; = b_l for i == l)
// (the very operations k_update performs, so the values are bit-identical).  k_peek therefore produces the
// shard's candidate for pivot t+1 in O(m + n) work BEFORE k_update(t) starts streaming, and the exchange
// (all-gather) and decision (k_commit) of pivot t+1 overlap the row update of pivot t on a second stream.
// `pend` = parameter block of the pivot whose update has not been applied yet (NULL / do_update == 0: none).
__global__ __launch_bounds__(256) void k_peek(const double* __restrict__ A, int64_t ld,
                                              const double* __restrict__ b, int m_local, int row0,
                                              const double* __restrict__ prow_t, const double* __restrict__ col_t,
                                              double* __restrict__ col_next, RatioRow* __restrict__ partial,
                                              const LpxCtl* __restrict__ ctl, const LpxCtl* __restrict__ pend) {
  __shared__ RatioRow sh[4];
  if (ctl->status != kRunning) return;
  const int en = ctl->e_next;
  if (en < 0) return;
  const bool pending = pend != nullptr && pend->do_update != 0;
  int e_t = -1, l_t = -1;
  double p = 1.0, bl = 0.0, pe = 0.0;
  if (pending) {
    e_t = pend->e_cur;
    l_t = pend->l - row0;
    p = pend->p;
    bl = pend->bl;
    pe = prow_t[en];
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  RatioRow best = rr_none();
  if (i < m_local) {
    const double a_old = A[(int64_t)i * ld + en];
    double a, bn;
    if (!pending) {
      a = a_old;
      bn = b[i];
    } else {
      const double ce = col_t[i];
      if (i == l_t) {
        a = pe;
        bn = bl;
      } else {
        a = (en == e_t) ? -__ddiv_rn(ce, p) : submul(a_old, ce, pe);
        bn = submul(b[i], ce, bl);
      }
    }
    col_next[i] = a;
    const double r = ratio_of(a, bn);
    if (r < kInf) best = RatioRow{r, row0 + i, 0};
  }
  best = rr_block_min(best, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = best;
}

// Folds k_peek's partials into the shard's candidate record and computes the candidate ROW as it will be
// after the pending update (same record layout as k_propose).
__global__ __launch_bounds__(1024) void k_peek_pack(const double* __restrict__ A, int64_t ld, int n, int row0,
                                                    const double* __restrict__ b,
                                                    const double* __restrict__ prow_t,
                                                    const double* __restrict__ col_t,
                                                    const RatioRow* __restrict__ partial, int nparts,
                                                    const LpxCtl* __restrict__ ctl,
                                                    const LpxCtl* __restrict__ pend, double* __restrict__ cand) {
  __shared__ RatioRow sh_rr[16];
  const int st = ctl->status;
  RatioRow best = rr_none();
  if (st == kRunning && ctl->e_next >= 0) {
    for (int k = threadIdx.x; k < nparts; k += blockDim.x) best = rr_min(best, partial[k]);
    best = rr_block_min(best, sh_rr);
  }
  const bool have = best.ratio < kInf;
  const int lr = have ? best.row - row0 : -1;
  const bool pending = pend != nullptr && pend->do_update != 0;
  int e_t = -1, l_t = -1;
  double p = 1.0, bl = 0.0, ce = 0.0;
  if (pending && have) {
    e_t = pend->e_cur;
    l_t = pend->l - row0;
    p = pend->p;
    bl = pend->bl;
    ce = col_t[lr];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    cand[0] = (st == kRunning) ? 0.0 : (double)(st + 1);
    cand[1] = (double)ctl->e_next;
    cand[2] = best.ratio;
    cand[3] = have ? (double)best.row : -1.0;
    double bn = 0.0;
    if (have) bn = !pending ? b[lr] : (lr == l_t ? bl : submul(b[lr], ce, bl));
    cand[4] = bn;
    cand[5] = cand[6] = cand[7] = 0.0;
  }
  if (have) {
    const double* row = A + (int64_t)lr * ld;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
      double x = row[j];
      if (pending) {
        if (lr == l_t) x = prow_t[j];
        else x = (j == e_t) ? -__ddiv_rn(ce, p) : submul(x, ce, prow_t[j]);
      }
      cand[8 + j] = x;
    }
  }
}

// ---- blocked pivoting: the decisions (lpx_decisions.inc) and the sweeps + fix-up (lpx_sweeps.inc) --------------------------
#include "lpx_decisions.inc"
#include "lpx_sweeps.inc"

// ------------------------------------------------------------------------------------------------ phase 1 helpers
// convertIntoAuxLP: auxA[i][n] = -1 (LPSolver.java:293)
__global__ void k_fill_column(double* A, int64_t ld, int m, int col, double value) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) A[(int64_t)i * ld + col] = value;
}

// restoreInitialLP: drop x0's column in place (LPSolver.java:208-211).  One workgroup per row; chunks are
// shifted left by one in increasing order, with a barrier between the chunk's reads and its writes.
__global__ __launch_bounds__(256) void k_drop_column(double* A, int64_t ld, int m, int n_old, int col) {
  const int i = blockIdx.x;
  if (i >= m) return;
  double* row = A + (int64_t)i * ld;
  for (int base = col; base < n_old; base += blockDim.x) {
    const int j = base + threadIdx.x;
    double x = 0.0;
    if (j + 1 < n_old) x = row[j + 1];  // the vacated last column becomes zero padding again
    __syncthreads();
    if (j < n_old) row[j] = x;
    __syncthreads();
  }
}

// restoreInitialLP: rebuild c and v by substitution (LPSolver.java:213-233), entries in keySet() order.
// Thread j accumulates c[j] over the entries in order — the same sequence of rounded additions per element
// as the reference; thread 0 of block 0 accumulates v.
__global__ __launch_bounds__(256) void k_restore_objective(const double* __restrict__ A, int64_t ld,
                                                           const double* __restrict__ b, double* c, int n,
                                                           const RestoreEntry* __restrict__ ent, int n_ent,
                                                           LpxCtl* ctl) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) {
    double acc = 0.0;
    for (int t = 0; t < n_ent; ++t) {
      const RestoreEntry en = ent[t];
      if (en.is_basic) {
        const double coef = -A[(int64_t)en.index * ld + j];                        // :226
        acc = addmul(acc, coef, en.k);                               // :227
      } else if (en.index == j) {
        acc = __dadd_rn(acc, en.k);                                                // :231
      }
    }
    c[j] = acc;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double v = 0.0;
    for (int t = 0; t < n_ent; ++t)
      if (ent[t].is_basic) v = addmul(v, b[ent[t].index], ent[t].k); // :223
    ctl->v = v;
  }
}

// Position-keyed, order-independent checksum of bit patterns (parity of full-size tableaux without a
// read-back): sum over elements of mix(bits + (pos+1)*K1) mod 2^64.
__device__ __forceinline__ unsigned long long mix64(unsigned long long bits, unsigned long long pos) {
  unsigned long long h = bits + (pos + 1ull) * 0x9E3779B97F4A7C15ull;
  h ^= h >> 30; h *= 0xBF58476D1CE4E5B9ull;
  h ^= h >> 27; h *= 0x94D049BB133111EBull;
  h ^= h >> 31;
  return h;
}

__global__ __launch_bounds__(256) void k_checksum(const double* __restrict__ A, int64_t ld, int m_local, int n,
                                                  int row0, const double* __restrict__ b,
                                                  const double* __restrict__ c, unsigned long long* out) {
  unsigned long long sa = 0, sb = 0, sc = 0;
  const int64_t total = (int64_t)m_local * n;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx / n, j = idx - i * n;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(A[i * ld + j]);
    sa += mix64(bits, (unsigned long long)((row0 + i) * (int64_t)n + j));
  }
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < m_local; i += blockDim.x)
      sb += mix64((unsigned long long)__double_as_longlong(b[i]), (unsigned long long)(row0 + i));
    for (int j = threadIdx.x; j < n; j += blockDim.x)
      sc += mix64((unsigned long long)__double_as_longlong(c[j]), (unsigned long long)j);
  }
  // wave reduce then one atomic per wave
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    sa += __shfl_down(sa, off, 64);
    sb += __shfl_down(sb, off, 64);
    sc += __shfl_down(sc, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&out[0], sa);
    if (blockIdx.x == 0) { atomicAdd(&out[1], sb); atomicAdd(&out[2], sc); }
  }
}

// LPStandardForm.getDual(): tiled transpose through LDS (64x64 tile, +1 padding against bank conflicts).
__global__ __launch_bounds__(256) void k_transpose(const double* __restrict__ A, int64_t lda,
                                                   double* __restrict__ At, int64_t ldat, int m, int n) {
  __shared__ double tile[64][65];
  const int bx = blockIdx.x * 64, by = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  for (int r = ty; r < 64; r += 4) {
    const int i = by + r, j = bx + tx;
    if (i < m && j < n) tile[r][tx] = A[(int64_t)i * lda + j];
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int j = bx + r, i = by + tx;
    if (j < n && i < m) At[(int64_t)j * ldat + i] = tile[tx][r];
  }
}

// ------------------------------------------------------------------------------------------------ launchers
// A launch that also signals `stop_` (may be NULL) when the kernel is through.  hipExtLaunchKernelGGL attaches the event to
// the kernel's own completion signal: an event RECORDED behind a launch is a packet of its own on the queue and costs the
// next launch of the stream 2.9 us alone and 4.7 us beside a kernel that streams through HBM; attached it costs nothing
// (scripts/micro/launch_gap.hip, profiles/r05_launch_gap.txt).
#define LPX_LAUNCH_STOP(kernel_, grid_, block_, stream_, stop_, ...)                                        \
  do {                                                                                                      \
    hipEvent_t lpx_stop_ = (stop_);                                                                          \
    if (lpx_stop_) hipExtLaunchKernelGGL(kernel_, grid_, block_, 0, stream_, nullptr, lpx_stop_, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel_, grid_, block_, 0, stream_, __VA_ARGS__);                               \
  } while (0)
// ... and one that is timed by a pair of events of its own (profiling of the sweeps: the kernel's duration, nothing around it)
#define LPX_LAUNCH_TIMED(kernel_, grid_, block_, stream_, t0_, t1_, ...)                                    \
  do {                                                                                                      \
    hipEvent_t lpx_t0_ = (t0_), lpx_t1_ = (t1_);                                                             \
    if (lpx_t0_ && lpx_t1_) hipExtLaunchKernelGGL(kernel_, grid_, block_, 0, stream_, lpx_t0_, lpx_t1_, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel_, grid_, block_, 0, stream_, __VA_ARGS__);                               \
  } while (0)
void launch_entering(const Buffers& B, int n, hipStream_t s, const LoopStart& start) {
  hipLaunchKernelGGL(k_entering, dim3(1), dim3(1024), 0, s, B.c, n, B.ctl, start);
}

void launch_entering_dantzig(const Buffers& B, int n, bool seed, hipStream_t s, const LoopStart& start) {
  // 256 threads: this kernel may run beside the row update on the comm stream (look-ahead pipeline)
  hipLaunchKernelGGL(k_entering_dantzig, dim3(1), dim3(256), 0, s, B.c, n, B.ctl, seed ? 1 : 0, start);
}

void launch_ratio_gather(const Buffers& B, int m_local, int row0, const Geometry& g, int forced_e, hipStream_t s) {
  if (g.ntiles <= 0) return;
  const int nblk = (m_local + 255) / 256;  // <= ntiles: choose_geometry caps a tile at 256 rows
  hipLaunchKernelGGL(k_ratio_gather, dim3(nblk), dim3(256), 0, s, B.A, B.ld, B.b, m_local, row0, B.col[0],
                     B.col[1], B.partial, B.ctl, g.ntiles, forced_e);
}

void launch_reduce_partials(const Buffers& B, const Geometry& g, hipStream_t s) {
  hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, s, B.partial, g.ntiles, B.ctl);
}

// workgroups of the pivot-finish kernels: one column per thread up to 16 workgroups
// (a single workgroup up to 4096 columns: it skips the three atomic round trips of the multi-workgroup fold)
static int finish_blocks(int64_t ld) {
  if (ld <= 4096) return 1;
  return (int)std::min<int64_t>(16, (ld + 1023) / 1024);
}

void launch_select_pivot(const Buffers& B, int n, int m_global, const Geometry& g, int forced_e, int forced_l,
                         hipStream_t s) {
  hipLaunchKernelGGL(k_select_pivot, dim3(finish_blocks(B.ld)), dim3(1024), 0, s, B.A, B.ld, n, m_global, B.b, B.c, B.prow,
                     B.partial, g.ntiles, B.perm, B.ctl, forced_e, forced_l);
}

template <int U, bool NT, bool OOP>
static void launch_update_t(const Buffers& B, int m_local, int row0, const Geometry& g, const double* prow,
                            const LpxCtl* up, double* A_out, double* b_out, hipStream_t s) {
  hipLaunchKernelGGL((k_update<U, NT, OOP>), dim3(g.nstrips * g.ntiles), dim3(256), 0, s, OOP ? A_out : B.A, B.A, B.ld,
                     m_local, row0, OOP ? b_out : B.b, B.b, prow, B.col[0], B.col[1], B.partial, up, g.rows_per_tile,
                     g.nstrips);
}

template <int U>
static void launch_update_u(const Buffers& B, int m_local, int row0, const Geometry& g, bool nt, const double* prow,
                            const LpxCtl* up, double* A_out, double* b_out, hipStream_t s) {
  const bool oop = A_out != nullptr;
  if (oop) {
    if (nt) launch_update_t<U, true, true>(B, m_local, row0, g, prow, up, A_out, b_out, s);
    else launch_update_t<U, false, true>(B, m_local, row0, g, prow, up, A_out, b_out, s);
  } else {
    if (nt) launch_update_t<U, true, false>(B, m_local, row0, g, prow, up, A_out, b_out, s);
    else launch_update_t<U, false, false>(B, m_local, row0, g, prow, up, A_out, b_out, s);
  }
}

// A_out/b_out == nullptr: in place (B.A, B.b); otherwise read (B.A, B.b) and write (A_out, b_out)
void launch_update(const Buffers& B, int m_local, int n, int row0, const Geometry& g, bool nt, const double* prow,
                   const LpxCtl* up, double* A_out, double* b_out, hipStream_t s) {
  (void)n;
  if (g.ntiles <= 0 || g.nstrips <= 0) return;
  switch (g.U) {
    case 1: launch_update_u<1>(B, m_local, row0, g, nt, prow, up, A_out, b_out, s); break;
    case 2: launch_update_u<2>(B, m_local, row0, g, nt, prow, up, A_out, b_out, s); break;
    default: launch_update_u<4>(B, m_local, row0, g, nt, prow, up, A_out, b_out, s); break;
  }
}

void launch_propose(const Buffers& B, int n, int row0, int m_local, const Geometry& g, double* d_candidate,
                    hipStream_t s) {
  hipLaunchKernelGGL(k_propose, dim3(finish_blocks(B.ld)), dim3(1024), 0, s, B.A, B.ld, n, row0, m_local, B.b, B.partial,
                     g.ntiles, B.ctl, d_candidate);
}

// Kernels that must slip in BESIDE a running row update (look-ahead pipeline) use 256-thread workgroups: a
// 1024-thread workgroup needs 16 free wave slots on one CU at once, which never happens while k_update's
// 256-thread workgroups keep refilling every CU (measured: such a kernel only completes when k_update drains).
static int small_blocks(int64_t ld) { return (int)std::max<int64_t>(1, std::min<int64_t>(64, (ld + 255) / 256)); }

void launch_commit(const Buffers& B, int n, int m_global, const double* d_gathered, int nranks, double* prow,
                   LpxCtl* up, int up_parity, hipStream_t s) {
  hipLaunchKernelGGL(k_commit, dim3(small_blocks(B.ld)), dim3(256), 0, s, d_gathered, nranks, n, B.ld, m_global, B.c,
                     prow, B.perm, B.ctl, up, up_parity);
}

void launch_peek(const Buffers& B, int n, int m_local, int row0, const double* prow_t, const double* col_t,
                 double* col_next, const LpxCtl* pend, double* d_candidate, hipStream_t s) {
  const int nblk = std::max(1, (m_local + 255) / 256);
  hipLaunchKernelGGL(k_peek, dim3(nblk), dim3(256), 0, s, B.A, B.ld, B.b, m_local, row0, prow_t, col_t, col_next,
                     B.partial, B.ctl, pend);
  hipLaunchKernelGGL(k_peek_pack, dim3(small_blocks(B.ld)), dim3(256), 0, s, B.A, B.ld, n, row0, B.b, prow_t, col_t,
                     B.partial, nblk, B.ctl, pend, d_candidate);
}

void launch_block_peek(const Buffers& B, const BlockRing& R, int n, int m_local, int row0, int np, double* d_candidate,
                       hipStream_t s) {
  const int nblk = std::max(1, (m_local + 255) / 256);
  hipLaunchKernelGGL(k_peek_multi, dim3(nblk), dim3(256), 0, s, B.A, B.ld, B.b, m_local, row0, R.prow, R.col, R.mp, R.up,
                     np, R.col + (int64_t)np * R.mp, R.col0 + (int64_t)np * R.mp, B.partial, B.ctl);
  hipLaunchKernelGGL(k_pack_multi, dim3(small_blocks(B.ld)), dim3(256), 0, s, B.A, B.ld, n, row0, B.b, R.prow, R.col,
                     R.mp, R.up, np, B.partial, nblk, B.ctl, d_candidate, R.row0 + (int64_t)np * B.ld);
}

void launch_block_decide(const Buffers& B, const BlockRing& R, int n, int m_global, const double* d_gathered, int nranks,
                         int slot, hipStream_t s) {
  hipLaunchKernelGGL(k_commit, dim3(small_blocks(B.ld)), dim3(256), 0, s, d_gathered, nranks, n, B.ld, m_global, B.c,
                     R.prow + (int64_t)slot * B.ld, B.perm, B.ctl, R.up + slot, 0);
}

// half / old_half: which half of the 2*kBlockMax-slot rings this block / the not-yet-swept previous block uses
// (n_old = 0: no such block, the tableau read is current); seq: launch counter of the loop (barrier counters
// alternate); B.A / B.b: the tableau version to read.
int launch_block_chain(const Buffers& B, const BlockRing& R, int n, int m, int nb, int half, int old_half, int n_old,
                        int b_from_tableau, int seq, int dantzig, int wgs, int fences, bool trace, LpxCtl* host_snap,
                        hipStream_t s, const MgPeers* mg, hipEvent_t stop) {
  // fences = 2 (the engine's default), acquire only: everything that crosses workgroups inside the launch is stored
  // write-through (st_agent = sc1), every storing wave drains (s_waitcnt vmcnt(0)), the workgroup meets, ONE lane
  // arrives with an agent-scope atomic add, the poller's loads of the handed-off bytes are all sc1 loads (ld_agent)
  // behind a workgroup barrier — the first row of MI355X_MICROARCH.md's table of hand-offs measured valid without
  // a release — and the acquire is kept on top.  A release (bit 0: buffer_wbl2) writes back the private plain-store
  // copies (own_col ...) too: ~2 us per barrier AND a slower sweep beside the launch (cfg3: 39.4k vs 45.7k pivots/s).
  const int64_t work = std::max<int64_t>(m, B.ld);
  int G = wgs > 0 ? wgs : (int)std::min<int64_t>(64, std::max<int64_t>(1, (work + 511) / 512));
  G = std::max(1, std::min(G, kChainMaxWgs));
  // chain_form 1 (one device): k_block_chain2, workgroups of kChain2Threads — one row / one column per thread needs
  // fewer of them (+1: workgroup 0 serves the hand-off window only); never more than the caller found resident (wgs
  // counts workgroups of ONE per CU, which holds for both kernels)
  // (shards: the two-hop exchange is grafted onto k_block_chain2 too; the opt-in one-hop form keeps k_block_chain_t.  Their
  // grid is the caller's, identical on every device — a shard waits for one arrival word per workgroup of the owner)
#ifdef LPX_WITH_VARIANTS
  const bool form2 = B.chain_form == 1 && (mg == nullptr || !mg->onehop);
#else   // (chain_form = 0 on one device names round 3's kernel: the variants library only)
  const bool form2 = mg == nullptr || (B.chain_form == 1 && !mg->onehop);
#endif
  if (form2 && mg == nullptr) {
    const int64_t rows_wgs = (m + kChain2Threads - 1) / kChain2Threads;
    const int64_t cols_wgs = (std::max<int64_t>(B.ld - 256, 0) + kChain2Threads - 1) / kChain2Threads + 1;
    G = (int)std::max<int64_t>(1, std::min<int64_t>(G, std::max(rows_wgs, cols_wgs)));
  }
  const int64_t K = kBlockMax;
  const bool wide = nb > 32 || n_old > 32;   // a block of more than 32 pivots on either side: the 64-slot form
  ChainArgs P{};
  P.A = B.A; P.b = B.b; P.ld = B.ld; P.mp = R.mp; P.n = n; P.m = m;
  P.c = B.c; P.perm = B.perm; P.ctl = B.ctl;
  const int64_t ho = half * K, oo = old_half * K;
  P.prow = R.prow + ho * B.ld; P.col = R.col + ho * R.mp; P.col0 = R.col0 + ho * R.mp; P.row0 = R.row0 + ho * B.ld;
  P.own_col = R.chain_own_col + ho * R.mp; P.own_prow = R.chain_own_prow + ho * B.ld;
  P.own_dvc = R.chain_own_dvc + ho * R.mp; P.up = R.up + ho;
  P.prow_o = R.prow + oo * B.ld; P.col_o = R.col + oo * R.mp;
  P.own_col_o = R.chain_own_col + oo * R.mp; P.own_prow_o = R.chain_own_prow + oo * B.ld;
  P.own_dvc_o = R.chain_own_dvc + oo * R.mp; P.up_o = R.up + oo;
  P.n_old = n_old; P.own_b = R.chain_own_b; P.b_from_tableau = b_from_tableau; P.nb = nb;
  P.own_rs_a = R.chain_own_rs; P.own_rs_b = R.chain_own_rs + R.mp;
  P.partA = reinterpret_cast<ChainPart*>(R.chain_part_a); P.partB = reinterpret_cast<RatioRow*>(R.chain_part_b);
  P.bar = R.chain_bar + 32 * (seq & 1); P.bar_next = R.chain_bar + 32 * ((seq + 1) & 1);
  P.hand = reinterpret_cast<unsigned long long*>(R.chain_bar + 64);   // its own 128-byte line
  P.hand_base = (unsigned)(seq + 1) * 64u;  // > any sequence of earlier launches (<= kBlockMax decisions each)
  P.dantzig = dantzig; P.fences = fences; P.host_snap = host_snap; P.dbg = trace ? R.chain_dbg : nullptr;
  P.spin_max = (mg && mg->spin_max) ? mg->spin_max : (1u << 22);
#ifdef LPX_DIAG_BUILD   // timing experiments only (results are wrong): never in the release library
  static const int chain_diag = getenv("LPX_CHAIN_DIAG") ? atoi(getenv("LPX_CHAIN_DIAG")) : 0;
  P.diag = chain_diag;
#endif
  P.census = R.census;
  if (mg) {
    P.shard_row0 = mg->row0; P.m_global = mg->m_global; P.n_dev = mg->n_dev; P.dev = mg->dev;
    P.mail_slot0 = mg->mail_slot0;
    P.onehop = mg->onehop;
    for (int d = 0; d < mg->n_dev && d < kMaxDevices; ++d) {
      P.mail_peer[d] = reinterpret_cast<MgMail*>(mg->mail[d]);
      P.prow_peer[d] = mg->prow[d] + ho * B.ld;   // the same ring half on every shard
      P.arrive_peer[d] = mg->arrive[d];
      P.candrow_peer[d] = mg->candrow[d];
      P.arrive2_peer[d] = mg->arrive2[d];
    }
    if (form2) {
      if (wide) LPX_LAUNCH_STOP((k_block_chain2_t<64, kChain2Threads, true>), dim3(G), dim3(kChain2Threads), s, stop, P);
      else LPX_LAUNCH_STOP((k_block_chain2_t<32, kChain2Threads, true>), dim3(G), dim3(kChain2Threads), s, stop, P);
    } else LPX_LAUNCH_STOP((k_block_chain_t<true, 32>), dim3(G), dim3(256), s, stop, P);   // (shards decide at most kShardBlockMax = 32 per block)
  } else if (form2) {
    P.m_global = m; P.n_dev = 1;
#ifdef LPX_CHAIN2_ONE_XCD
    G = std::min(G, 32);
    if (wide) hipLaunchKernelGGL((k_block_chain2_t<64, kChain2Threads, false>), dim3(8 * G), dim3(kChain2Threads), 0, s, P);
    else hipLaunchKernelGGL((k_block_chain2_t<32, kChain2Threads, false>), dim3(8 * G), dim3(kChain2Threads), 0, s, P);
    if (stop) (void)hipEventRecord(stop, s);
#else
    if (wide) LPX_LAUNCH_STOP((k_block_chain2_t<64, kChain2Threads, false>), dim3(G), dim3(kChain2Threads), s, stop, P);
    else LPX_LAUNCH_STOP((k_block_chain2_t<32, kChain2Threads, false>), dim3(G), dim3(kChain2Threads), s, stop, P);
#endif
  } else {
#ifdef LPX_WITH_VARIANTS
    P.m_global = m; P.n_dev = 1;
    if (wide) LPX_LAUNCH_STOP((k_block_chain_t<false, 64>), dim3(G), dim3(256), s, stop, P);
    else LPX_LAUNCH_STOP((k_block_chain_t<false, 32>), dim3(G), dim3(256), s, stop, P);
#else
    if (stop) (void)hipEventRecord(stop, s);
#endif
  }
  return G;
}

// The runtime prepares a kernel for a device the first time it is LAUNCHED (looking the function up does not do it;
// measured on MI355X boxes: 0.2-0.3 ms on the first block that uses the steady-state sweep kernel, i.e. a tenth of a
// block's time at cfg4).  A loop whose first blocks have another length than its later ones (a short warm-up, the tail
// of a budget) would pay that inside its own run, so every kernel of the blocked loop is launched once per device
// when the first ring is built — one workgroup each, with arguments that make it return at once (no pending pivots,
// no rows, no decisions) and touch nothing but the ring's own words.
// The ticket buffer: one 128-byte slot per 128-column sub-strip (at least the four the preparing launches pull from),
// then spare slots.  (The word a pull kernel sets when a bounded wait ran out is BlockRing::sweep_fail.)
int64_t sweep_ticket_slots(int64_t ld) { return std::max<int64_t>(ld / 64, 4) + 4; }   // (k_sweep64_one: a counter per 64 columns)
unsigned* sweep_fail_word(const BlockRing& R, int64_t ld) {
#ifdef LPX_WITH_VARIANTS   // (the only kernel that sets it, k_sweep64_pull, is in the variants library only)
  (void)ld;
  return R.sweep_fail;
#else
  (void)R; (void)ld;
  return nullptr;
#endif
}

void preload_block_kernels(const Buffers& B, const BlockRing& R, hipStream_t s) {
  static std::atomic<unsigned> done{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return;
  if (done.fetch_or(1u << dev) & (1u << dev)) return;
  double* const A = B.A;
  const int64_t ld = B.ld;
  unsigned* const no_census = nullptr;
#define LPX_EACH_NT_OOP(X) X(true, true) X(true, false) X(false, true) X(false, false)
#define LPX_PRE_TILES(K_, NT_, OOP_) \
  hipLaunchKernelGGL((k_update_tiles<K_, NT_, OOP_>), dim3(1), dim3(256), 0, s, A, A, ld, 0, R.prow, R.col, R.mp, R.up, 0, 16, 1, no_census);
#define LPX_PRE_T2(NT_, OOP_) LPX_PRE_TILES(2, NT_, OOP_)
#define LPX_PRE_T4(NT_, OOP_) LPX_PRE_TILES(4, NT_, OOP_)
#define LPX_PRE_T8(NT_, OOP_) LPX_PRE_TILES(8, NT_, OOP_)
#define LPX_PRE_T16(NT_, OOP_) LPX_PRE_TILES(16, NT_, OOP_)
#define LPX_PRE_T32(NT_, OOP_) LPX_PRE_TILES(32, NT_, OOP_)
#define LPX_PRE_MULTI(NT_, OOP_) \
  hipLaunchKernelGGL((k_update_multi<32, NT_, OOP_>), dim3(1), dim3(256), 0, s, A, A, ld, 0, R.prow, R.col, R.mp, R.up, 0, 64, 1, no_census, 0, 0);
#define LPX_PRE_STEADY(NT_, OOP_) \
  hipLaunchKernelGGL((k_sweep32_steady<NT_, OOP_>), dim3(1), dim3(256), 0, s, A, A, ld, 0, R.prow, R.col, R.mp, R.up, 0, 48, 1);
#define LPX_PRE_PIPE(NT_, OOP_) \
  hipLaunchKernelGGL((k_sweep64_pipe<NT_, OOP_>), dim3(1), dim3(512), 0, s, A, A, ld, 0, R.prow, R.col, R.mp, R.up, 0, 48, 1);
#define LPX_PRE_DMA(NT_, OOP_) \
  hipLaunchKernelGGL((k_sweep32_dma<NT_, OOP_>), dim3(1), dim3(256), 0, s, A, A, ld, 0, R.prow, R.col, R.mp, R.up, 0, 2, 1, 1, R.zeros, 0);
  LPX_EACH_NT_OOP(LPX_PRE_T2) LPX_EACH_NT_OOP(LPX_PRE_T4) LPX_EACH_NT_OOP(LPX_PRE_T8) LPX_EACH_NT_OOP(LPX_PRE_T16)
  LPX_EACH_NT_OOP(LPX_PRE_T32) LPX_EACH_NT_OOP(LPX_PRE_MULTI)
#ifdef LPX_WITH_VARIANTS
  LPX_EACH_NT_OOP(LPX_PRE_STEADY) LPX_EACH_NT_OOP(LPX_PRE_PIPE) LPX_EACH_NT_OOP(LPX_PRE_DMA)
#endif
#define LPX_PRE_PULL(NT_, OOP_) \
  hipLaunchKernelGGL((k_sweep32_pull<NT_, OOP_>), dim3(1), dim3(256), 0, s, A, A, ld, 0, R.prow, R.col, R.mp, R.up, 0, 1, R.col_packed, R.tickets, (long long*)nullptr);
  if (R.tickets && R.col_packed) {   // m_local = 0: the first ticket already names nothing
    LPX_EACH_NT_OOP(LPX_PRE_PULL)
    hipLaunchKernelGGL(k_pack_multipliers<32>, dim3(1), dim3(256), 0, s, R.col, R.mp, R.up, 0, 0, R.col_packed, R.tickets, 0, (long long*)nullptr);
    hipLaunchKernelGGL(k_pack_multipliers<64>, dim3(1), dim3(256), 0, s, R.col, R.mp, R.up, 0, 0, R.col_packed, R.tickets, 0, (long long*)nullptr);
#ifdef LPX_WITH_VARIANTS
#define LPX_PRE_PULL64(NT_, OOP_) \
    hipLaunchKernelGGL((k_sweep64_pull<NT_, OOP_>), dim3(1), dim3(256), 0, s, A, A, ld, 0, R.prow, R.up, 0, 1, R.col_packed, R.tickets, (sweep_fail_word)(R, ld));
    LPX_EACH_NT_OOP(LPX_PRE_PULL64)
#undef LPX_PRE_PULL64
#endif
#define LPX_PRE_ONE64(NT_, OOP_) \
    hipLaunchKernelGGL((k_sweep64_one<NT_, OOP_>), dim3(1), dim3(256), 0, s, A, A, ld, 0, R.prow, R.up, 0, 1, R.col_packed, R.tickets, (long long*)nullptr);
    LPX_EACH_NT_OOP(LPX_PRE_ONE64)
#undef LPX_PRE_ONE64
#if LPX_FUSED
    hipLaunchKernelGGL(k_pack_multipliers_mfma, dim3(1), dim3(256), 0, s, R.col, R.mp, R.up, 0, 0, R.col_packed, R.tickets, 0, (long long*)nullptr, 0);
#ifdef LPX_WITH_VARIANTS
#define LPX_PRE_MFMA64(NT_, OOP_) \
    hipLaunchKernelGGL((k_sweep64_mfma<NT_, OOP_>), dim3(1), dim3(256), 0, s, A, A, ld, 0, R.prow, R.up, 0, 1, R.col_packed, R.tickets);
    LPX_EACH_NT_OOP(LPX_PRE_MFMA64)
#undef LPX_PRE_MFMA64
#endif
#define LPX_PRE_MFMA642(NT_, OOP_) \
    hipLaunchKernelGGL((k_sweep64_mfma2<NT_, OOP_>), dim3(1), dim3(256), 0, s, A, A, ld, 0, R.prow, R.up, 0, 1, R.col_packed, R.tickets, 33, (long long*)nullptr);
    LPX_EACH_NT_OOP(LPX_PRE_MFMA642)
#undef LPX_PRE_MFMA642
#endif
  }
#undef LPX_PRE_PULL
#undef LPX_PRE_DMA
#undef LPX_PRE_PIPE
#undef LPX_PRE_STEADY
#undef LPX_PRE_MULTI
#undef LPX_PRE_T32
#undef LPX_PRE_T16
#undef LPX_PRE_T8
#undef LPX_PRE_T4
#undef LPX_PRE_T2
#undef LPX_PRE_TILES
#undef LPX_EACH_NT_OOP
  hipLaunchKernelGGL(k_block_fixup, dim3(1, 1, 3), dim3(256), 0, s, A, ld, 0, 0, 0, B.b, R.prow, R.col, R.col0, R.row0,
                     R.mp, R.up, 0, B.b, (long long*)nullptr, (double*)nullptr, (double*)nullptr);
  hipLaunchKernelGGL(k_block_fixup_scatter, dim3(1, 1, 2), dim3(256), 0, s, A, ld, 0, 0, (const double*)nullptr, (const double*)nullptr,
                     R.mp, R.up, 0, (long long*)nullptr);
  ChainArgs P{};   // nb = 0: every workgroup returns after reading the loop state (no barrier, nothing published)
  P.ctl = B.ctl; P.up = R.up; P.nb = 0;
  P.bar = R.chain_bar; P.bar_next = R.chain_bar + 32;
  P.spin_max = 1u << 22;
#ifdef LPX_WITH_VARIANTS
  hipLaunchKernelGGL((k_block_chain_t<false, 32>), dim3(1), dim3(256), 0, s, P);
  hipLaunchKernelGGL((k_block_chain_t<false, 64>), dim3(1), dim3(256), 0, s, P);
#endif
  hipLaunchKernelGGL((k_block_chain_t<true, 32>), dim3(1), dim3(256), 0, s, P);
  hipLaunchKernelGGL((k_block_chain2_t<32, kChain2Threads, false>), dim3(1), dim3(kChain2Threads), 0, s, P);
  hipLaunchKernelGGL((k_block_chain2_t<64, kChain2Threads, false>), dim3(1), dim3(kChain2Threads), 0, s, P);
  hipLaunchKernelGGL((k_block_chain2_t<32, kChain2Threads, true>), dim3(1), dim3(kChain2Threads), 0, s, P);
  (void)hipGetLastError();
}

int chain_blocks_per_cu() {
  int nb = 0;
  int nw = 0;
  int n2 = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (k_block_chain_t<true, 32>), 256, 0) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nw, (k_block_chain2_t<32, kChain2Threads, false>), kChain2Threads, 0) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&n2, (k_block_chain2_t<64, kChain2Threads, true>), kChain2Threads, 0) != hipSuccess) {
    (void)hipGetLastError();
    nb = nw = n2 = 1;
  }
  return std::max(1, std::min(std::min(nb, nw), n2));
}

template <int K>
static void launch_sweep_tiles(const Buffers& B, const BlockRing& R, int m_local, int kmax, int rows_per_tile, bool nt,
                               const double* A_src, hipStream_t s) {
  const int nstrips = (int)((B.ld + 511) / 512);
  const int ntiles = (m_local + rows_per_tile - 1) / rows_per_tile;
  const dim3 grid(nstrips * ntiles), block(256);
#define LPX_LAUNCH_SWEEP(NT_, OOP_)                                                                              \
  hipLaunchKernelGGL((k_update_tiles<K, NT_, OOP_>), grid, block, 0, s, B.A, A_src, B.ld, m_local, R.prow, R.col, \
                     R.mp, R.up, kmax, rows_per_tile, nstrips, R.census ? R.census + kChainMaxWgs : nullptr)
  if (A_src) { if (nt) LPX_LAUNCH_SWEEP(true, true); else LPX_LAUNCH_SWEEP(false, true); }
  else { if (nt) LPX_LAUNCH_SWEEP(true, false); else LPX_LAUNCH_SWEEP(false, false); }
#undef LPX_LAUNCH_SWEEP
}

template <int K>
static void launch_sweep_k(const Buffers& B, const BlockRing& R, int m_local, int kmax, int rows_per_wg, bool nt,
                           const double* A_src, hipStream_t s, int complement = 0, int slot0 = 0) {
  const int nstrips = (int)((B.ld + 511) / 512);
  const int ngroups = (m_local + rows_per_wg - 1) / rows_per_wg;
  const dim3 grid(nstrips * ngroups), block(256);
#define LPX_LAUNCH_SWEEP(NT_, OOP_)                                                                              \
  hipLaunchKernelGGL((k_update_multi<K, NT_, OOP_>), grid, block, 0, s, B.A, A_src, B.ld, m_local, R.prow, R.col, \
                     R.mp, R.up, kmax, rows_per_wg, nstrips, R.census ? R.census + kChainMaxWgs : nullptr, complement, \
                     slot0)
  if (A_src) { if (nt) LPX_LAUNCH_SWEEP(true, true); else LPX_LAUNCH_SWEEP(false, true); }
  else { if (nt) LPX_LAUNCH_SWEEP(true, false); else LPX_LAUNCH_SWEEP(false, false); }
#undef LPX_LAUNCH_SWEEP
}

// every wave on its own, batches pulled from per-sub-strip ticket counters (R.tickets: zeroed here, on the stream);
// the grid is what is resident: G workgroups per strip, G x strips <= slots (two workgroups per CU the stream may use)
// Where a pulled sweep's pack kernel runs.  With a side stream (the overlapped loop: R is a ring HALF with a packed-multiplier
// buffer and ticket counters of its own) it runs there, as soon as the block's decisions are through — normally while the
// sweep of the block before still streams — and the sweep stream only waits for its event; otherwise in front of the sweep.
static hipStream_t pack_stream(const FixSide* side, hipStream_t s) { return (side && side->stream && side->packed) ? side->stream : s; }
static hipEvent_t pack_stop(const FixSide* side) { return (side && side->stream && side->packed) ? side->packed : nullptr; }
static void pack_done(const FixSide* side, hipStream_t s) {   // (the event is the pack launch's own stop event)
  if (pack_stop(side)) (void)hipStreamWaitEvent(s, side->packed, 0);
}

static void launch_sweep_pull(const Buffers& B, const BlockRing& R, int m_local, int kmax, bool nt, const double* A_src,
                              hipStream_t s, int slots = 512, const FixSide* side = nullptr, hipEvent_t t0 = nullptr,
                              hipEvent_t t1 = nullptr) {
  const int nstrips_full = (int)(B.ld / 512);
  const int nbt = m_local / 4;
  const int G = std::max(1, std::min(nbt, slots / std::max(1, nstrips_full)));
  LPX_LAUNCH_STOP(k_pack_multipliers<32>, dim3((nbt + 7) / 8), dim3(256), pack_stream(side, s), pack_stop(side), R.col, R.mp, R.up, kmax, nbt, R.col_packed,
                     R.tickets, nstrips_full * 4, (long long*)nullptr);
  pack_done(side, s);
  const dim3 grid(nstrips_full * G), block(256);
#define LPX_LAUNCH_PULL(NT_, OOP_)                                                                                \
  LPX_LAUNCH_TIMED((k_sweep32_pull<NT_, OOP_>), grid, block, s, t0, t1, B.A, A_src, B.ld, m_local, R.prow, R.col, R.mp, \
                   R.up, kmax, nstrips_full, R.col_packed, R.tickets, R.clk)
  if (A_src) { if (nt) LPX_LAUNCH_PULL(true, true); else LPX_LAUNCH_PULL(false, true); }
  else { if (nt) LPX_LAUNCH_PULL(true, false); else LPX_LAUNCH_PULL(false, false); }
#undef LPX_LAUNCH_PULL
}

// blocks of 33..64 by single waves on 64-column sub-strips (k_sweep64_one); G workgroups per group of four sub-strips
static void launch_sweep64_one(const Buffers& B, const BlockRing& R, int m_local, int kmax, bool nt, const double* A_src,
                               hipStream_t s, int slots = 512, const FixSide* side = nullptr, hipEvent_t t0 = nullptr,
                               hipEvent_t t1 = nullptr) {
  const int nstrips_full = (int)(B.ld / 512);
  const int ngroups = nstrips_full * 2;
  const int nbt = m_local / 4;
  const int G = std::max(1, std::min(nbt, slots / std::max(1, ngroups)));
  LPX_LAUNCH_STOP(k_pack_multipliers<64>, dim3((nbt + 3) / 4), dim3(256), pack_stream(side, s), pack_stop(side), R.col, R.mp, R.up, kmax, nbt, R.col_packed,
                     R.tickets, nstrips_full * 8, (long long*)nullptr);
  pack_done(side, s);
  const dim3 grid(ngroups * G), block(256);
#define LPX_LAUNCH_ONE64(NT_, OOP_)                                                                               \
  LPX_LAUNCH_TIMED((k_sweep64_one<NT_, OOP_>), grid, block, s, t0, t1, B.A, A_src, B.ld, m_local, R.prow, R.up, kmax, \
                   nstrips_full, R.col_packed, R.tickets, R.clk)
  if (A_src) { if (nt) LPX_LAUNCH_ONE64(true, true); else LPX_LAUNCH_ONE64(false, true); }
  else { if (nt) LPX_LAUNCH_ONE64(true, false); else LPX_LAUNCH_ONE64(false, false); }
#undef LPX_LAUNCH_ONE64
}

#if LPX_FUSED
// blocks of 33..64 on the matrix cores (fused arithmetic only): one wave per SIMD, G workgroups per group of four
// 64-column sub-strips, 16-row tiles pulled from the sub-strip's ticket counter
static void launch_sweep64_mfma(const Buffers& B, const BlockRing& R, int m_local, int kmax, bool nt, const double* A_src,
                                hipStream_t s, int slots = 256, bool two_waves = false, int kmin = 33, const FixSide* side = nullptr,
                                hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr) {
  const int nstrips_full = (int)(B.ld / 512);
  const int ngroups = nstrips_full * 2;
  const int ntiles = m_local / 16;
  const int G = std::max(1, std::min(ntiles, slots / std::max(1, ngroups)));
  LPX_LAUNCH_STOP(k_pack_multipliers_mfma, dim3(ntiles), dim3(256), pack_stream(side, s), pack_stop(side), R.col, R.mp, R.up, kmax, ntiles, R.col_packed,
                     R.tickets, nstrips_full * 8, two_waves ? (long long*)nullptr : R.clk, two_waves ? 1 : 0);
  pack_done(side, s);
  if (two_waves) {   // k_sweep64_mfma2: groups of 128 columns, two workgroups per CU
    const int ng2 = nstrips_full * 4;
    const int G2 = std::max(1, std::min(ntiles, 2 * slots / std::max(1, ng2)));
    const dim3 grid2(ng2 * G2), block2(256);
#define LPX_LAUNCH_MFMA642(NT_, OOP_)                                                                               \
    LPX_LAUNCH_TIMED((k_sweep64_mfma2<NT_, OOP_>), grid2, block2, s, t0, t1, B.A, A_src, B.ld, m_local, R.prow, R.up, kmax, \
                     nstrips_full, R.col_packed, R.tickets, kmin, R.clk)
    if (A_src) { if (nt) LPX_LAUNCH_MFMA642(true, true); else LPX_LAUNCH_MFMA642(false, true); }
    else { if (nt) LPX_LAUNCH_MFMA642(true, false); else LPX_LAUNCH_MFMA642(false, false); }
#undef LPX_LAUNCH_MFMA642
    return;
  }
#ifdef LPX_WITH_VARIANTS   // k_sweep64_mfma (first version, one wave per SIMD): sweep_form = 4 of the variants library
  const dim3 grid(ngroups * G), block(256);
#define LPX_LAUNCH_MFMA64(NT_, OOP_)                                                                               \
  hipLaunchKernelGGL((k_sweep64_mfma<NT_, OOP_>), grid, block, 0, s, B.A, A_src, B.ld, m_local, R.prow, R.up, kmax, \
                     nstrips_full, R.col_packed, R.tickets)
  if (A_src) { if (nt) LPX_LAUNCH_MFMA64(true, true); else LPX_LAUNCH_MFMA64(false, true); }
  else { if (nt) LPX_LAUNCH_MFMA64(true, false); else LPX_LAUNCH_MFMA64(false, false); }
#undef LPX_LAUNCH_MFMA64
#else
  (void)G;
#endif
}
#endif

static int choose_sweep_rows(int m_local, int64_t ld, int K, int cus) {
  const int64_t nstrips = (ld + 511) / 512;
  const int64_t slots = std::max(1, 2 * cus);
  const int prologue_rows = std::max(8, (K * 3) / 2);
  int best_rows = kSweepChunk;
  int64_t best_cost = INT64_MAX;
  const int max_rows = (int)std::min<int64_t>(4096, (((int64_t)1 << 32) - 1) / (ld * 8) / kSweepChunk * kSweepChunk);
  for (int rows = kSweepChunk; rows <= std::max(kSweepChunk, max_rows); rows += kSweepChunk) {
    const int64_t groups = (m_local + rows - 1) / rows;
    const int64_t rounds = (nstrips * groups + slots - 1) / slots;
    // one round of huge runs leaves no slack for uneven CUs: ask for at least three rounds when there is enough work
    const int64_t want_rounds = (int64_t)m_local * nstrips >= 3 * slots * kSweepChunk ? 3 : 1;
    const int64_t cost = std::max(rounds, want_rounds) * (rows + prologue_rows);
    if (cost < best_cost || (cost == best_cost && rows > best_rows)) { best_cost = cost; best_rows = rows; }
    if (groups == 1) break;
  }
  return best_rows;
}

const char* sweep_kernel_name(int code) {
  switch (code) {
    case kSweepTiles: return "k_update_tiles";
    case kSweepMulti: return "k_update_multi";
    case kSweepSteady: return "k_sweep32_steady";
    case kSweepPipe64: return "k_sweep64_pipe";
    case kSweepDma: return "k_sweep32_dma";
    case kSweepPull: return "k_sweep32_pull";
    case kSweepPull64: return "k_sweep64_pull";
    case kSweepOne64: return "k_sweep64_one";
    case kSweepMfma64: return "k_sweep64_mfma";
    case kSweepMfma642: return "k_sweep64_mfma2";
    default: return "";
  }
}

#ifdef LPX_WITH_VARIANTS
#define LPX_VARIANT_PART 6
#include "variants/lpx_variants.inc"
#undef LPX_VARIANT_PART
#endif

// rows_per_wg <= 0: chosen here (see choose_sweep_rows); cus: CUs the stream may use (0: the whole device)
int launch_block_sweep(const Buffers& B, const BlockRing& R, int n, int m_local, int row0, int K, int rows_per_wg,
                       bool nt, hipStream_t s, const double* A_src, const double* b_src, hipEvent_t after_sweep,
                       int cus, int form, int* kernel_used, const FixSide* side, hipEvent_t stop, hipEvent_t before_sweep) {
  int used = kSweepNone;
  if (kernel_used) *kernel_used = used;
  if (K < 1) {
    if (before_sweep) (void)hipEventRecord(before_sweep, s);
    if (after_sweep) (void)hipEventRecord(after_sweep, s);
    if (stop) (void)hipEventRecord(stop, s);
    return 0;
  }
  // profiling (before_sweep / after_sweep): ONE pulled sweep kernel takes the pair as its own start / stop events (its duration,
  // no packet on the queue); any other form is bracketed by two recorded events
  bool timed_own = false, bracket_open = false;
  auto bracket = [&]() { if (before_sweep && !bracket_open) { (void)hipEventRecord(before_sweep, s); bracket_open = true; } };
  if (cus <= 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  // The fix-up's chains read ring values only (and b, which the sweep leaves alone): with a side stream they are computed
  // BESIDE the sweep into the images of this ring half, and only their copy into the tableau follows the sweep.
  // Likewise the pack kernel of a pulled sweep (pack_stream): first on the side stream, then the chains.
  const bool side_fix = side && side->stream && R.fix_col && R.fix_row;
  if (side_fix) (void)hipStreamWaitEvent(side->stream, side->ready, 0);
  else side = nullptr;
  if (K <= 16) {
    // tiles of a few rows (k_update_tiles): up to K = 16 the sweep is HBM-bound and 16-row tiles stream best
    // (profiles/r01_sweep_rows.txt: larger tiles widen the set of DRAM pages in flight, -10 %)
    bracket();
    int rows_per_tile = rows_per_wg <= 0 ? 16 : rows_per_wg;
    rows_per_tile = std::max(8, std::min(rows_per_tile, kSweepMaxRows)) & ~7;  // rows go four or eight at a time
    while (rows_per_tile > 8 && (int64_t)rows_per_tile * B.ld * 8 >= (int64_t)1 << 32) rows_per_tile -= 8;  // 32-bit offsets
    if (K <= 2) launch_sweep_tiles<2>(B, R, m_local, K, rows_per_tile, nt, A_src, s);
    else if (K <= 4) launch_sweep_tiles<4>(B, R, m_local, K, rows_per_tile, nt, A_src, s);
    else if (K <= 8) launch_sweep_tiles<8>(B, R, m_local, K, rows_per_tile, nt, A_src, s);
    else launch_sweep_tiles<16>(B, R, m_local, K, rows_per_tile, nt, A_src, s);
    rows_per_wg = rows_per_tile;
    used = kSweepTiles;
  } else if (K > 32) {
    // blocks of up to 64 pivots.  A block of 33..64 valid pivots over the full strips goes through a 64-step kernel in one
    // pass (k_sweep64_one: one wave per 64-column sub-strip; in the fused arithmetic with 16-row tiles k_sweep64_mfma2, the
    // matrix cores); whatever that leaves (a block that ended early, the partial last strip, m not a multiple of 4) takes
    // two passes of the generic kernel: slots 0..31 (out of place when asked), then slots 32.. in place on the result.
    // An entry's update reads ring values only, so the split changes nothing.
    // (sweep_form 1 / 2 / 4 name the superseded kernels of csrc/variants/ — k_sweep64_pipe, k_sweep64_pull, k_sweep64_mfma —
    // and select them in the variants library only; here they mean the default.)
    const int nstrips_full = (int)(B.ld / 512);
    // (the pull kernels address a batch's rows by 32-bit byte offsets: 3 * ld * 8 + 1 KiB must stay below 2^32)
    const bool geom = m_local % 4 == 0 && nstrips_full >= 1 && 3 * B.ld * 8 + 1024 < ((int64_t)1 << 32);
#ifdef LPX_WITH_VARIANTS
    const bool pull = geom && form != 1 && R.tickets && R.col_packed;   // round 3 / 4: blocks of 33..64 valid pivots
    const bool pipe = geom && !pull && K == 64;                         // round 2: full blocks of 64 only
    const bool one = pull && form != 2;
    [[maybe_unused]] const bool mfma_form = form == 0 || form == 4;
    const bool mfma2 = form == 0;
#else
    const bool pull = geom && R.tickets && R.col_packed;
    const bool pipe = false;
    const bool one = pull;
    [[maybe_unused]] const bool mfma_form = form != 3;   // (3: k_sweep64_one in the fused arithmetic too)
    const bool mfma2 = true;
#endif
    bool mfma = false;
#if LPX_FUSED
    mfma = one && mfma_form && m_local % 16 == 0 && 16 * B.ld * 8 + 1024 < ((int64_t)1 << 32);
#endif
    int rows64 = 0;
    // the matrix-core sweep over whole strips takes ANY number of valid pivots: no generic launches behind it (they cost
    // two launches and their gaps per block, ~20 us of a 2.1 ms block at cfg4, only to find nothing to do)
    const bool whole = mfma && mfma2 && B.ld % 512 == 0;
    timed_own = whole && before_sweep && after_sweep;
    if (!timed_own) bracket();
    if (mfma) {
#if LPX_FUSED
      launch_sweep64_mfma(B, R, m_local, K, nt, A_src, s, cus, mfma2, whole ? 1 : 33, side, timed_own ? before_sweep : nullptr,
                          timed_own ? after_sweep : nullptr);
#endif
      rows64 = 16;
    } else if (one) {
      launch_sweep64_one(B, R, m_local, K, nt, A_src, s, 2 * cus, side);
      rows64 = 4;
    }
#ifdef LPX_WITH_VARIANTS
    else if (pull) {
      launch_sweep64_pull(B, R, m_local, K, nt, A_src, s, 2 * cus);
      rows64 = 4;
    } else if (pipe) {
      rows64 = rows_per_wg > 0 ? std::max(4, rows_per_wg / 4 * 4) : choose_pipe_rows(m_local, nstrips_full, cus);
      while (rows64 > 4 && (int64_t)rows64 * B.ld * 8 >= (int64_t)1 << 32) rows64 -= 4;  // 32-bit offsets
      launch_sweep64_pipe(B, R, m_local, K, rows64, nt, A_src, s);
    }
#endif
    // what the one-pass kernel does not take (the partial last strip; a block with fewer valid pivots than it wants:
    // < 33 for the pull forms, < 64 for the two-stage pipe) goes through two passes of the generic kernel
    const int complement = pull ? 34 : (pipe ? 65 : 0);
    int rows = choose_sweep_rows(m_local, B.ld, 32, cus);
    while (rows > kSweepChunk && (int64_t)rows * B.ld * 8 >= (int64_t)1 << 32) rows -= kSweepChunk;
    if (!whole) {
      launch_sweep_k<32>(B, R, m_local, K, rows, nt, A_src, s, complement, 0);
      launch_sweep_k<32>(B, R, m_local, K, rows, nt, nullptr, s, complement, 32);
    }
    rows_per_wg = (pull || pipe) ? rows64 : rows;
    used = mfma ? (mfma2 ? kSweepMfma642 : kSweepMfma64) : one ? kSweepOne64 : (pull ? kSweepPull64 : (pipe ? kSweepPipe64 : kSweepMulti));
  } else if (K < kMaxBlock && !(m_local % 4 == 0 && B.ld >= 512)) {
    // a partly filled block of 17..31 pivots (the tail of a pivot budget) where the pulled kernel does not apply: the tile
    // kernel's guarded path took such blocks faster than the long-run kernel's (cfg3, 20 pivots, same box: 490 vs 615 us);
    // 64-row tiles as long as the grid keeps a few thousand workgroups (profiles/r01_sweep_rows.txt)
    bracket();
    const int64_t nstrips = (B.ld + 511) / 512;
    int rows_per_tile = 16;
    for (int rows : {64, 32})
      if ((int64_t)((m_local + rows - 1) / rows) * nstrips >= 4096) { rows_per_tile = rows; break; }
    while (rows_per_tile > 8 && (int64_t)rows_per_tile * B.ld * 8 >= (int64_t)1 << 32) rows_per_tile -= 8;
    launch_sweep_tiles<32>(B, R, m_local, K, rows_per_tile, nt, A_src, s);
    rows_per_wg = rows_per_tile;
    used = kSweepTiles;
  } else {
    // blocks of 17..32.  Long runs of rows per workgroup (k_update_multi) where the pulled kernel does not apply; otherwise
    // k_sweep32_pull over the full strips (LDS-DMA staging, every wave pulls its batches in address order) and the generic
    // kernel for the partial last strip.  (sweep_form 1 / 2: k_sweep32_steady / k_sweep32_dma of the variants library.)
    if (rows_per_wg <= 0) rows_per_wg = choose_sweep_rows(m_local, B.ld, K, cus);
    rows_per_wg = std::max(kSweepChunk, (rows_per_wg + kSweepChunk - 1) / kSweepChunk * kSweepChunk);
    while (rows_per_wg > kSweepChunk && (int64_t)rows_per_wg * B.ld * 8 >= (int64_t)1 << 32) rows_per_wg -= kSweepChunk;  // 32-bit offsets
    const bool wide32 = 3 * B.ld * 8 + 1024 >= ((int64_t)1 << 32);   // 32-bit row offsets of the LDS-DMA kernels would wrap
    const bool pullable = m_local % 4 == 0 && B.ld >= 512 && !wide32 && R.zeros && R.tickets && R.col_packed;
    bool done = false;
#ifdef LPX_WITH_VARIANTS
    const bool variant32 = m_local % 4 == 0 && B.ld >= 512 && (form == 1 || form == 2 || !pullable);
#else
    const bool variant32 = false;
#endif
    timed_own = !variant32 && pullable && B.ld % 512 == 0 && before_sweep && after_sweep;   // k_sweep32_pull alone
    if (!timed_own) bracket();
#ifdef LPX_WITH_VARIANTS
    if (variant32) {
      int rows48 = choose_pipe_rows(m_local, (int)(B.ld / 512), 2 * cus, 48);
      while (rows48 > 4 && (int64_t)rows48 * B.ld * 8 >= (int64_t)1 << 32) rows48 -= 4;   // 32-bit offsets
      if (form == 2 && pullable) {   // LDS-DMA staging, runs of rows (the step between the two)
        launch_sweep_dma(B, R, m_local, K, rows48, nt, A_src, s, 2 * cus);
        used = kSweepDma;
      } else {                       // round 2: batches parked in registers, runs of rows
        launch_sweep_steady(B, R, m_local, K, rows48, nt, A_src, s);
        used = kSweepSteady;
      }
      if (B.ld % 512 != 0) launch_sweep_k<32>(B, R, m_local, K, rows_per_wg, nt, A_src, s, 1);   // the partial last strip
      rows_per_wg = rows48;
      done = true;
    }
#endif
    if (!done && pullable) {
      launch_sweep_pull(B, R, m_local, K, nt, A_src, s, 2 * cus, side, timed_own ? before_sweep : nullptr, timed_own ? after_sweep : nullptr);
      used = kSweepPull;
      if (B.ld % 512 != 0) launch_sweep_k<32>(B, R, m_local, K, rows_per_wg, nt, A_src, s, 1);   // the partial last strip
      rows_per_wg = 4;   // (what lpx_state_get_info reports as the run length: one batch)
    } else if (!done) {
      launch_sweep_k<32>(B, R, m_local, K, rows_per_wg, nt, A_src, s);
      used = kSweepMulti;
    }
  }
  if (kernel_used) *kernel_used = used;
  if (after_sweep && !timed_own) { bracket(); (void)hipEventRecord(after_sweep, s); }  // profiling: the sweep kernels alone
  // the clock probe: only the pulled sweeps stamp at their START (sweep_front_stamp; the variants' through their pack kernels); behind any other form the
  // fix-up must not pair its stamp with a front stamp of an older launch (lpx_state_info.sweep_clock_mhz then says 0)
  const bool probed = used == kSweepPull || used == kSweepPull64 || used == kSweepOne64 || used == kSweepMfma64 || used == kSweepMfma642;
  const int gx = (int)((std::max<int64_t>(m_local, B.ld) + 255) / 256);
  if (side_fix) {   // the chains beside the sweep (behind its pack kernel on the side stream); only their copy into the tableau follows it
    LPX_LAUNCH_STOP(k_block_fixup, dim3(gx, (K + kFixChunk - 1) / kFixChunk, 3), dim3(256), side->stream, side->done, B.A, B.ld, n, m_local, row0,
                    B.b, R.prow, R.col, R.col0, R.row0, R.mp, R.up, K, b_src ? b_src : B.b, (long long*)nullptr, R.fix_col, R.fix_row);
    (void)hipStreamWaitEvent(s, side->done, 0);
  }
  // `stop` (the caller's "this block's sweep is through"): the stop event of the last kernel — unless the probe's memset follows
  const bool memset_behind = !probed && R.clk;
  hipEvent_t last_stop = memset_behind ? nullptr : stop;
  if (side_fix) {
    LPX_LAUNCH_STOP(k_block_fixup_scatter, dim3(gx, (K + kFixChunk - 1) / kFixChunk, 2), dim3(256), s, last_stop, B.A, B.ld, m_local, row0,
                    (const double*)R.fix_col, (const double*)R.fix_row, R.mp, R.up, K, probed ? R.clk : (long long*)nullptr);
  } else {
    LPX_LAUNCH_STOP(k_block_fixup, dim3(gx, (K + kFixChunk - 1) / kFixChunk, 3), dim3(256), s, last_stop, B.A, B.ld, n, m_local, row0, B.b, R.prow, R.col,
                    R.col0, R.row0, R.mp, R.up, K, b_src ? b_src : B.b, probed ? R.clk : (long long*)nullptr, (double*)nullptr, (double*)nullptr);
  }
  if (memset_behind) {
    (void)hipMemsetAsync(R.clk, 0, 256, s);
    if (stop) (void)hipEventRecord(stop, s);
  }
  return rows_per_wg;
}

void launch_fill_column(double* A, int64_t ld, int m, int col, double value, hipStream_t s) {
  if (m <= 0) return;
  hipLaunchKernelGGL(k_fill_column, dim3((m + 255) / 256), dim3(256), 0, s, A, ld, m, col, value);
}

void launch_drop_column(double* A, int64_t ld, int m, int n_old, int col, hipStream_t s) {
  if (m <= 0) return;
  hipLaunchKernelGGL(k_drop_column, dim3(m), dim3(256), 0, s, A, ld, m, n_old, col);
}

void launch_restore_objective(const Buffers& B, int n, const RestoreEntry* d_entries, int n_entries, hipStream_t s) {
  const int blocks = n > 0 ? (n + 255) / 256 : 1;
  hipLaunchKernelGGL(k_restore_objective, dim3(blocks), dim3(256), 0, s, B.A, B.ld, B.b, B.c, n, d_entries,
                     n_entries, B.ctl);
}

void launch_checksum(const Buffers& B, int m_local, int n, int row0, unsigned long long* d_out3, hipStream_t s) {
  hipLaunchKernelGGL(k_checksum, dim3(1024), dim3(256), 0, s, B.A, B.ld, m_local, n, row0, B.b, B.c, d_out3);
}

void launch_transpose(const double* dA, int64_t lda, double* dAt, int64_t ldat, int m, int n, hipStream_t s) {
  if (m <= 0 || n <= 0) return;
  hipLaunchKernelGGL(k_transpose, dim3((n + 63) / 64, (m + 63) / 64), dim3(256), 0, s, dA, lda, dAt, ldat, m, n);
}

}  // namespace plain / fused
}  // namespace lpxk
