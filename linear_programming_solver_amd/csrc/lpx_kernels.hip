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

// ------------------------------------------------------------------------------------------------ blocked pivoting
// K pivots per pass over the tableau ("delayed updates").  k_peek shows that the decision of the next pivot
// needs the updated tableau only through ONE column and ONE row, both obtainable from the un-updated tableau
// by the pending pivot's rank-1 formula.  The same holds for any number of pending pivots applied in order:
//
//     decision s (s = 0..K-1):  k_peek_multi  column e_s of the tableau with pivots 0..s-1 applied, from the
//                                             stale column + s sequential corrections  -> ratio test
//                               k_pack_multi  the winning row with pivots 0..s-1 applied, likewise
//                               k_commit      normalise it, update c, v, perm, choose e_{s+1}  (unchanged)
//     then ONE k_update_multi:  every tableau entry is read once, run through the K rank-1 updates in pivot
//                               order in registers, and written once.
//
// Every element sees exactly the operation sequence of K separate k_update passes (one rounded product and
// one rounded difference per pivot, same special cases for the pivot row and the entering column), so the
// result is bit-identical — but the HBM traffic per pivot is 16*m*n/K bytes instead of 16*m*n.  Pending
// pivots live in a ring: prow_ring[s] (ld doubles), col_ring[s] (mp doubles: column e_s BEFORE pivot s),
// ring[s] (LpxCtl-shaped parameter block written by finish_pivot).
constexpr int kMaxBlock = 32;

__global__ __launch_bounds__(256) void k_peek_multi(const double* __restrict__ A, int64_t ld,
                                                    const double* __restrict__ b, int m_local, int row0,
                                                    const double* __restrict__ prow_ring,
                                                    const double* __restrict__ col_ring, int64_t mp,
                                                    const LpxCtl* __restrict__ ring, int np,
                                                    double* __restrict__ col_out, double* __restrict__ col0_out,
                                                    RatioRow* __restrict__ partial,
                                                    const LpxCtl* __restrict__ ctl) {
  __shared__ RatioRow sh[4];
  __shared__ double sh_pe[kMaxBlock], sh_p[kMaxBlock], sh_bl[kMaxBlock];
  __shared__ int sh_e[kMaxBlock], sh_l[kMaxBlock];
  if (ctl->status != kRunning) return;
  const int en = ctl->e_next;
  if (en < 0) return;
  if ((int)threadIdx.x < np) {
    const LpxCtl& q = ring[threadIdx.x];
    sh_e[threadIdx.x] = q.e_cur;
    sh_l[threadIdx.x] = q.l - row0;
    sh_p[threadIdx.x] = q.p;
    sh_bl[threadIdx.x] = q.bl;
    sh_pe[threadIdx.x] = prow_ring[(int64_t)threadIdx.x * ld + en];
  }
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  RatioRow best = rr_none();
  if (i < m_local) {
    double a = A[(int64_t)i * ld + en];
    double bi = b[i];
    col0_out[i] = a;  // the stale column, kept for k_block_fixup
    for (int s = 0; s < np; ++s) {  // pending pivots in order: exactly what the K row updates would do
      const double cs = col_ring[(int64_t)s * mp + i];
      if (i == sh_l[s]) {
        a = sh_pe[s];
        bi = sh_bl[s];
      } else {
        a = (en == sh_e[s]) ? -__ddiv_rn(cs, sh_p[s]) : submul(a, cs, sh_pe[s]);
        bi = submul(bi, cs, sh_bl[s]);
      }
    }
    col_out[i] = a;
    const double r = ratio_of(a, bi);
    if (r < kInf) best = RatioRow{r, row0 + i, 0};
  }
  best = rr_block_min(best, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = best;
}

__global__ __launch_bounds__(256) void k_pack_multi(const double* __restrict__ A, int64_t ld, int n, int row0,
                                                    const double* __restrict__ b,
                                                    const double* __restrict__ prow_ring,
                                                    const double* __restrict__ col_ring, int64_t mp,
                                                    const LpxCtl* __restrict__ ring, int np,
                                                    const RatioRow* __restrict__ partial, int nparts,
                                                    const LpxCtl* __restrict__ ctl, double* __restrict__ cand,
                                                    double* __restrict__ row0_out) {
  __shared__ RatioRow sh_rr[4];
  __shared__ double sh_cs[kMaxBlock], sh_p[kMaxBlock], sh_bl[kMaxBlock];
  __shared__ int sh_e[kMaxBlock], sh_l[kMaxBlock];
  const int st = ctl->status;
  RatioRow best = rr_none();
  if (st == kRunning && ctl->e_next >= 0) {
    for (int k = threadIdx.x; k < nparts; k += blockDim.x) best = rr_min(best, partial[k]);
    best = rr_block_min(best, sh_rr);
  }
  const bool have = best.ratio < kInf;
  const int lr = have ? best.row - row0 : -1;
  if (have && (int)threadIdx.x < np) {
    const LpxCtl& q = ring[threadIdx.x];
    sh_e[threadIdx.x] = q.e_cur;
    sh_l[threadIdx.x] = q.l - row0;
    sh_p[threadIdx.x] = q.p;
    sh_bl[threadIdx.x] = q.bl;
    sh_cs[threadIdx.x] = col_ring[(int64_t)threadIdx.x * mp + lr];
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    cand[0] = (st == kRunning) ? 0.0 : (double)(st + 1);
    cand[1] = (double)ctl->e_next;
    cand[2] = best.ratio;
    cand[3] = have ? (double)best.row : -1.0;
    double bn = 0.0;
    if (have) {
      bn = b[lr];
      for (int s = 0; s < np; ++s) bn = (lr == sh_l[s]) ? sh_bl[s] : submul(bn, sh_cs[s], sh_bl[s]);
    }
    cand[4] = bn;
    cand[5] = cand[6] = cand[7] = 0.0;
  }
  if (have) {
    const double* row = A + (int64_t)lr * ld;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
      double x = row[j];
      row0_out[j] = x;  // the stale row, kept for k_block_fixup
      for (int s = 0; s < np; ++s) {
        const double pr = prow_ring[(int64_t)s * ld + j];
        if (lr == sh_l[s]) x = pr;
        else x = (j == sh_e[s]) ? -__ddiv_rn(sh_cs[s], sh_p[s]) : submul(x, sh_cs[s], pr);
      }
      cand[8 + j] = x;
    }
  }
}

// ------------------------------------------------------------------------------------------------ k_block_chain
// All decisions of one block in ONE launch (one shard owning every row, i.e. the single-GPU loop).  The three
// launches per decision above are latency chains (status -> entering slot -> column -> ratio -> row -> ...), ~6 us
// each however little they compute; here the same steps run inside a persistent grid of <= 256 workgroups (one
// per CU at most, so every workgroup is resident) separated by two grid barriers per decision:
//
//   phase A  every thread owns rows i = gid, gid+T, ...: column e of the tableau with the s pending pivots applied
//            (stale column + s sequential corrections, as k_peek_multi), ratio test, workgroup minimum -> partA[wg]
//   barrier
//   phase B  every workgroup reduces partA to the leaving row l (same data, same result everywhere); every thread
//            owns columns j = gid, gid+T, ...: row l with the s pending pivots applied (as k_pack_multi), normalised
//            row, update of c, candidate for the next entering slot (as finish_pivot) -> partB[wg]; workgroup 0's
//            first thread updates v, perm, the tracked slot and the ring's parameter block
//   barrier  every workgroup reduces partB to the next entering slot.
//
// Ownership is fixed for the whole launch, so a thread re-reads only ring entries it wrote itself; what crosses
// workgroups (the partial records, c[e], prow_s[e], col_s[l]) is read with agent-scope loads behind an agent-scope
// release / acquire pair around the barrier's counter (MI355X: per-XCD L2s are not coherent with each other).
// The arithmetic is statement for statement that of k_peek_multi / k_pack_multi / finish_pivot.
struct ChainPart {   // 64 bytes per workgroup and decision parity (as a record, or as eight 8-byte granules)
  double ratio, a, bi;
  int32_t row, pad;
  double unused[4];
};
#ifndef LPX_CHAIN_TAGGED
#define LPX_CHAIN_TAGGED 1
#endif
__device__ __forceinline__ unsigned lo32(double x) { return (unsigned)__double2loint(x); }
__device__ __forceinline__ unsigned hi32(double x) { return (unsigned)__double2hiint(x); }
__device__ __forceinline__ double from32(unsigned lo, unsigned hi) { return __hiloint2double((int)hi, (int)lo); }

// ---- multi-device decisions -----------------------------------------------------------------------------------
// Row-block shards on several GPUs of one node, one process (lpx_multi_*): every device runs this same persistent
// kernel on its shard; the replicated state (c, v, perm, the pivot-row ring) is updated identically everywhere.
// What crosses devices per decision, by direct stores into peer memory over xGMI (no collective, no host):
//   (1) the shard's minimum-ratio candidate {ratio, row, a, b_row}: 32 bytes + a sequence tag into a mailbox slot on
//       every device — allreduce(min+loc) as an all-to-all of 40-byte records, the lowest global row winning ties
//       exactly as the sequential scan of LPState.java:292-303 does;
//   (2) the normalised pivot row: the shard that owns the leaving row computes it (thread = column) and stores every
//       value into every device's replica of the ring ((n+1) doubles = 128 KiB at n = 16384, 7 links in parallel),
//       then each of its workgroups raises its arrival word on every device.
// One-hop form (LPX_OPT_MULTI_ONEHOP, off by default — like the rest of this path it has only ever run with all shards
// on one GPU): a decision above costs two DEPENDENT cross-device hops (candidates, then the winner's row).  Instead
// every shard computes the row of its OWN candidate (thread = column, pending pivots applied) before it knows whether
// it wins, stores it un-normalised into slot `dev` of every device's candidate-row buffer together with the candidate
// record, and raises per-workgroup arrival words; every device then picks the winner from its mailbox, waits for THAT
// shard's arrival words only and normalises the row itself (x / p is the same correctly rounded division everywhere).
// One hop per decision, 128 KiB x 7 links per device of wire (~1 us), n_dev times the row arithmetic in parallel.
// Stores to peers and loads of peer-written data are system-scope (sc0 sc1: write-through / cache-bypassing) and a
// tag or arrival word is stored only after the storing waves have drained (s_waitcnt vmcnt(0)) and met; with
// fences bit 0 / 1 a system-scope release / acquire fence is added around every exchange (the default across real
// devices: the fence-free form is the one MI355X_MICROARCH.md measured valid INSIDE one device only).
struct MgMail {
  double ratio, a, bi;
  int32_t row;
  uint32_t tag;   // sequence number of the decision, stored last
};
static_assert(sizeof(MgMail) == 32, "mailbox record is one 32-byte granule");

__device__ __forceinline__ double ld_sys(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys(double* p, double x) {
  __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ double ld_agent(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_agent(const int32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// write-through store (sc1): the line does not stay dirty in this XCD's L2, so the barrier's release has nothing
// to write back
__device__ __forceinline__ void st_agent(double* p, double x) {
  __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(int32_t* p, int32_t x) {
  __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifndef LPX_BARRIER_SLEEP
#define LPX_BARRIER_SLEEP __builtin_amdgcn_s_sleep(1)
#endif
// Monotonic-counter grid barrier.  Every wave drains its stores, the workgroup meets, one lane (optionally after an
// agent-scope release) arrives, polls (relaxed, bounded) and acquires; the second workgroup barrier holds the other
// waves until the invalidate has completed.  Returns false when the spin bound was hit (never in a healthy run:
// it only keeps a bug from hanging the device).  fences: bit 0 = release fence, bit 1 = acquire fence.
__device__ __forceinline__ bool grid_barrier(unsigned* bar, unsigned target, int* sh_fail, int fences, unsigned spin_max) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (fences & 1) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      LPX_BARRIER_SLEEP;
      if (++spins > spin_max) { *sh_fail = 1; break; }   // 1: grid barrier
    }
    if (fences & 2) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  return *sh_fail == 0;
}

// One pending pivot applied to the entering-column value of row i (phase A) / to the row value of column j
// (phase B).  `on` is wave-uniform: steps outside the live range leave the value alone.
#define LPX_CHAIN_STEP_A(on, cs_r, pe_r, l_r)                                      \
  {                                                                                \
    const double t_ = submul(a, (cs_r), (pe_r));                     \
    const double nv_ = (ig == (l_r)) ? (pe_r) : t_;                                \
    a = (on) ? nv_ : a;                                                            \
  }
#define LPX_CHAIN_STEP_B(on, cs_r, prv_r, e_r, dv_r)                               \
  {                                                                                \
    const double t_ = submul(x, (cs_r), (prv_r));                    \
    const double nv_ = (j == (e_r)) ? (dv_r) : t_;                                 \
    x = (on) ? nv_ : x;                                                            \
  }
// eight consecutive steps [r0, r0+8) of one half (LDS offset `off`: 0 = previous block, kMaxBlock = this block),
// live range [first, last); the chunk's parameters are read first, then the arithmetic (the LDS latencies overlap)
#define LPX_CHAIN_LIVE(r0, first, last) ((r0) < (last) && (r0) + 8 > (first))
#define LPX_CHAIN_CHUNK_A(off, r0, first, last, csv)                                                   \
  if (LPX_CHAIN_LIVE(r0, first, last)) {                                                           \
    double pe8[8];                                                                                     \
    int l8[8];                                                                                         \
    _Pragma("unroll") for (int q = 0; q < 8; ++q) { pe8[q] = sh_pe[(off) + (r0) + q]; l8[q] = sh_l[(off) + (r0) + q]; } \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    _Pragma("unroll") for (int q = 0; q < 8; ++q)                                                      \
        LPX_CHAIN_STEP_A(((r0) + q >= (first) && (r0) + q < (last)), csv[(r0) + q], pe8[q], l8[q])     \
  }
#define LPX_CHAIN_CHUNK_B(off, r0, first, last, prvv)                                                  \
  if (LPX_CHAIN_LIVE(r0, first, last)) {                                                           \
    double cs8[8], dv8[8];                                                                             \
    int e8[8];                                                                                         \
    _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                    \
      cs8[q] = sh_cs[(off) + (r0) + q]; dv8[q] = sh_dv[(off) + (r0) + q]; e8[q] = sh_e[(off) + (r0) + q]; \
    }                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    _Pragma("unroll") for (int q = 0; q < 8; ++q)                                                      \
        LPX_CHAIN_STEP_B(((r0) + q >= (first) && (r0) + q < (last)), cs8[q], prvv[(r0) + q], e8[q], dv8[q]) \
  }

// The host's view of the loop (status, pivots, ...) goes straight into its pinned snapshot: no copy-engine
// transfer (and its ~40 us of stream idle time) per block.  Called by the one thread that wrote ctl.
__device__ __forceinline__ void chain_publish(const LpxCtl* ctl, LpxCtl* host_snap) {
  if (!host_snap) return;
  *host_snap = *ctl;
  __threadfence_system();
}

__device__ __forceinline__ unsigned xcc_id() {  // which XCD this wave runs on (placement census; never used for correctness)
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 15u;
}

// Launch parameters.  "own" = the ring half of the block being decided (written here), "old" = the half of the
// PREVIOUS block when its sweep has not been applied to the tableau this launch reads (n_old pivots; 0 = none):
// the decisions then see the tableau through n_old + s pending pivots, and that sweep can run beside this launch.
struct ChainArgs {
  const double* A;   // tableau version read (never written during the launch)
  const double* b;
  int64_t ld, mp;
  int n, m;
  double* c;
  int32_t* perm;
  LpxCtl* ctl;
  double *prow, *col, *col0, *row0, *own_col, *own_prow, *own_dvc;  // own half (slot s of the block)
  LpxCtl* up;
  const double *prow_o, *col_o, *own_col_o, *own_prow_o, *own_dvc_o;  // old half
  const LpxCtl* up_o;
  int n_old;
  double* own_b;        // b with every decided pivot applied (kept across the launches of one loop)
  int32_t *own_rs_a, *own_rs_b;   // k_block_chain2: per row / slot, the last pending pivot that replaced it (LDS index, -1: none)
  int b_from_tableau;   // 1: first launch of a loop, own_b is not valid before decision 0
  int nb;
  ChainPart* partA;
  RatioRow* partB;
  unsigned long long* hand;  // {sequence << 32 | entering slot + 2}: workgroup 0's one-way hand-off (see phase B)
  unsigned hand_base;   // sequence number of this launch's decision 0 (strictly increasing over launches)
  unsigned* bar;        // this launch's barrier counter (zero on entry)
  unsigned* bar_next;   // the next launch's: zeroed here
  int dantzig, fences;
  LpxCtl* host_snap;
  long long* dbg;
  unsigned* census;     // [workgroup] = XCC id + 1
  // ---- row-block shards on several devices (k_block_chain_t<true>; see "multi-device decisions" below)
  int shard_row0;       // first global row of this shard (0 on one device)
  int m_global;
  int n_dev, dev;       // shards taking part / this shard's rank
  int mail_slot0;       // parity of the decisions taken by earlier launches of this loop (see the exchange)
  MgMail* mail_peer[kMaxDevices];            // every shard's mailbox [2][kMaxDevices] (peer-mapped); [dev] = own
  double* prow_peer[kMaxDevices];            // every shard's replica of this block's pivot-row ring half
  unsigned long long* arrive_peer[kMaxDevices];  // every shard's arrival words [kChainMaxWgs]
  // one-hop form (onehop != 0): every shard ships the ROW of its own candidate with the candidate, see k_block_chain_t
  int onehop;
  double* candrow_peer[kMaxDevices];             // every shard's candidate rows [2][kMaxDevices][ld]
  unsigned long long* arrive2_peer[kMaxDevices]; // every shard's arrival words of the candidate rows [2][kMaxDevices][kChainMaxWgs]
  unsigned spin_max;    // bound of every wait between workgroups / devices (polls; ~0.5-1 us each): a bug never hangs the GPU
  int diag;             // k_block_chain2 diagnostics (-DLPX_DIAG_BUILD libraries only: LPX_CHAIN_DIAG, timing experiments — results are wrong): bit 0 = read the
                        // entering column from CONTIGUOUS addresses (what a column-major mirror of the window would cost)
};

// KB: capacity of one ring half seen by the launch (32, or 64 for blocks of more than 32 pivots — the same code with
// longer register arrays; its decisions cost more, so the launcher picks it only when a block needs it).
// LDS layout of the per-pivot parameters: [0, KB) old half, [KB, 2 KB) own half.
template <int KB>
__device__ __forceinline__ int chain_restart(const unsigned long long* sh_mask) {
  // highest LDS index whose pending pivot touched the same slot / row (ballots of the first 2 KB / 64 waves)
  if constexpr (KB > 32) {
    const unsigned long long hi = sh_mask[1], lo = sh_mask[0];
    return hi ? 127 - __clzll((long long)hi) : (lo ? 63 - __clzll((long long)lo) : -1);
  } else {
    const unsigned long long lo = sh_mask[0];
    return lo ? 63 - __clzll((long long)lo) : -1;
  }
}

template <bool MG, int KB>
__global__ __launch_bounds__(256) void k_block_chain_t(const ChainArgs P) {
  static_assert(KB == 32 || KB == 64, "ring half of 32 or 64 slots");
  __shared__ RatioRow sh_rr[4];
  __shared__ unsigned long long sh_mask[2];
#if LPX_CHAIN_TAGGED
  __shared__ unsigned sh_gran[8], sh_part[kChainMaxWgs * 8];
#endif
  __shared__ double sh_pe[2 * KB], sh_cs[2 * KB], sh_dv[2 * KB], sh_p[2 * KB], sh_bl[2 * KB], sh_win[2];
  __shared__ int sh_e[2 * KB], sh_l[2 * KB];
  __shared__ int sh_fail, sh_restart;
  const int row0 = MG ? P.shard_row0 : 0;   // global index of local row 0 (a shard of an lpx_multi; 0 on one device)
  const double* __restrict__ A = P.A;
  const double* __restrict__ b = P.b;
  const int64_t ld = P.ld, mp = P.mp;
  const int n = P.n, m = P.m, nb = P.nb, n_old = P.n_old;
  LpxCtl* const ctl = P.ctl;
  const int G = gridDim.x, T = G * 256, tid = threadIdx.x, gid = blockIdx.x * 256 + tid;
  const bool lead = gid == 0;
  if (tid == 0) {
    sh_fail = 0;
    if (P.census) P.census[blockIdx.x] = xcc_id() + 1u;
  }
  if (lead) st_agent(reinterpret_cast<int32_t*>(P.bar_next), 0);
  // loop state at entry: written by earlier launches, identical in every workgroup
  int e = ctl->e_next;
  if (ctl->status != kRunning || e < 0 || nb < 1) {
    if (lead && nb >= 1) P.up[0].do_update = 0;
    if (lead) chain_publish(ctl, P.host_snap);
    return;
  }
  if (tid < n_old) {
    const LpxCtl& q = P.up_o[tid];
    sh_e[tid] = q.e_cur; sh_l[tid] = q.l; sh_p[tid] = q.p; sh_bl[tid] = q.bl;
  }
  int64_t pivots = ctl->pivots;
  const int64_t max_pivots = ctl->max_pivots;
  double v = ctl->v;
  int track = ctl->track, parity = ctl->parity;
  unsigned target = 0;

  for (int s = 0; s < nb; ++s) {
    // ------------------------------------------------------------------ phase A: column e, ratio test
    // Column e of the current tableau = the stale column run through the pending pivots in order (old half, then
    // this block's 0..s-1).  If a pending pivot u* entered at the same slot, its update REPLACED the column by
    // -(col/p) (1/p in its own row) whatever it was before: the chain restarts there, from own_dvc[u*], and only
    // the pivots after u* remain — every step has the one generic form and the loop is branch-free.
    // b needs no chain at all: own_b holds it with every decided pivot applied (one step added per decision).
    if (P.dbg && lead) P.dbg[s * 5 + 0] = wall_clock64();
    const double pc = ld_agent(&P.c[e]);  // c[e] is rewritten only in phase B, after the next barrier
    if (tid < 2 * KB) {
      const int r = tid & (KB - 1);
      const bool old = tid < KB;
      const bool valid = old ? r < n_old : r < s;
      bool same = false;
      if (valid) {  // one lane per pending pivot u fetches prow_u[e]
        const double* pe_src = (old ? P.prow_o : P.prow) + (int64_t)r * ld + e;
        sh_pe[tid] = MG ? ld_sys(pe_src) : ld_agent(pe_src);  // on a shard: possibly stored by a peer device
        same = sh_e[tid] == e;
      }
      const unsigned long long mask = __ballot(same);
      if ((tid & 63) == 0) sh_mask[tid >> 6] = mask;
    }
    __syncthreads();
    const int ra = chain_restart<KB>(sh_mask);                   // LDS index of the restart pivot, -1: none
    const int fo_a = ra < 0 ? 0 : (ra < KB ? ra + 1 : n_old);    // first live step of the old half
    const int fn_a = ra >= KB ? ra - KB + 1 : 0;                 // first live step of this block's half
    const bool use_b = P.b_from_tableau && s == 0;
    RatioRow best = rr_none();
    double best_a = 0.0, best_b = 0.0;
    for (int i = gid; i < m; i += T) {
      const double* src_a = ra < 0 ? &A[(int64_t)i * ld + e]
                                   : (ra < KB ? &P.own_dvc_o[(int64_t)ra * mp + i] : &P.own_dvc[(int64_t)(ra - KB) * mp + i]);
      const double* src_b = use_b ? &b[i] : &P.own_b[i];
      double a = *src_a;
      const double bi = *src_b;
      const int ig = row0 + i;   // global row: what the ring's parameter blocks name
      // this thread's own stores, only the chunks with a live step (after a restart most are dead: the rings
      // exceed the L2, every dead chunk is HBM traffic taken from the sweep running beside this launch); all
      // loads are issued before the first use — one round trip
      double cs[KB], cso[KB];
#pragma unroll
      for (int r0 = 0; r0 < KB; r0 += 8)
        if (LPX_CHAIN_LIVE(r0, fo_a, n_old)) {
#pragma unroll
          for (int q = 0; q < 8; ++q) cso[r0 + q] = P.own_col_o[(int64_t)(r0 + q) * mp + i];
        }
#pragma unroll
      for (int r0 = 0; r0 < KB; r0 += 8)
        if (LPX_CHAIN_LIVE(r0, fn_a, s)) {
#pragma unroll
          for (int q = 0; q < 8; ++q) cs[r0 + q] = P.own_col[(int64_t)(r0 + q) * mp + i];
        }
#pragma unroll
      for (int r0 = 0; r0 < KB; r0 += 8) LPX_CHAIN_CHUNK_A(0, r0, fo_a, n_old, cso)
      // here `a` is the entry of the tableau the sweep of THIS block will read (all older pivots applied): what
      // k_block_fixup restarts from.  (After a restart inside this block the value is not that entry, but then
      // the fix-up's own chain replaces it at the same pivot, whatever it starts from.)
      st_agent(&P.col0[(int64_t)s * mp + i], a);
#pragma unroll
      for (int r0 = 0; r0 < KB; r0 += 8) LPX_CHAIN_CHUNK_A(KB, r0, fn_a, s, cs)
      st_agent(&P.col[(int64_t)s * mp + i], a);
      P.own_col[(int64_t)s * mp + i] = a;
      const double rt = ratio_of(a, bi);
      if (rt < best.ratio) {  // i ascends per thread: strict < keeps the lowest row among equal ratios
        best = RatioRow{rt, ig, 0};
        best_a = a;
        best_b = bi;
      }
    }
    {
      const RatioRow w = rr_block_min(best, sh_rr);
      if (w.row != INT_MAX && best.row == w.row) { sh_win[0] = best_a; sh_win[1] = best_b; }
      __syncthreads();
#if !LPX_CHAIN_TAGGED
      if (tid == 0) {
        ChainPart* rec = &P.partA[(s & 1) * kChainMaxWgs + blockIdx.x];  // two sets: see the hand-off below
        st_agent(&rec->ratio, w.ratio);
        st_agent(&rec->row, w.row);
        st_agent(&rec->a, (w.row != INT_MAX) ? sh_win[0] : 0.0);
        st_agent(&rec->bi, (w.row != INT_MAX) ? sh_win[1] : 0.0);
      }
    }
    const unsigned xtag = P.hand_base + (unsigned)s;  // sequence number of this decision (unique over launches)
    if (P.dbg && lead) P.dbg[s * 5 + 1] = wall_clock64();
    target += (unsigned)G;
    if (!grid_barrier(P.bar, target, &sh_fail, P.fences, P.spin_max)) {
      if (lead) { ctl->status = 7 /* LPX_DEVICE_ERROR */; chain_publish(ctl, P.host_snap); }
      return;
    }
    if (P.dbg && lead) P.dbg[s * 5 + 2] = wall_clock64();

    // ------------------------------------------------------------------ phase B: the leaving row
    RatioRow mine = rr_none();
    double mine_a = 0.0, mine_b = 0.0;
    if (tid < G) {
      const ChainPart* rec = &P.partA[(s & 1) * kChainMaxWgs + tid];
      mine.ratio = ld_agent(&rec->ratio);
      mine.row = ld_agent(&rec->row);
      mine_a = ld_agent(&rec->a);
      mine_b = ld_agent(&rec->bi);
    }
#else
      // The workgroup's candidate {ratio, a, b_row, row} goes out as seven self-validating 8-byte granules
      // {32 data bits, sequence tag}, one aligned store each (MI355X_MICROARCH.md, hand-off by data-tagged granules):
      // a reader that sees the tag has the data — no counter, no flag, no second read.  Stored only after every wave
      // of the workgroup has drained its stores of this phase (col, col0: what later decisions read across
      // workgroups), so seeing a record still implies what arriving at the counter barrier implied.
      sh_gran[0] = lo32(w.ratio); sh_gran[1] = hi32(w.ratio);
      const double wa = (w.row != INT_MAX) ? sh_win[0] : 0.0, wb = (w.row != INT_MAX) ? sh_win[1] : 0.0;
      sh_gran[2] = lo32(wa); sh_gran[3] = hi32(wa); sh_gran[4] = lo32(wb); sh_gran[5] = hi32(wb);
      sh_gran[6] = (unsigned)w.row;   // (every thread writes the same values)
    }
    const unsigned xtag = P.hand_base + (unsigned)s;  // sequence number of this decision (unique over launches)
#ifndef LPX_CHAIN_DBG2
    if (P.dbg && lead) P.dbg[s * 5 + 1] = wall_clock64();
#endif
    {
      unsigned long long* const gran = reinterpret_cast<unsigned long long*>(P.partA) + (size_t)(s & 1) * kChainMaxWgs * 8;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid < 7) {
        if (P.fences & 1) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __hip_atomic_store(&gran[blockIdx.x * 8 + tid], ((unsigned long long)xtag << 32) | sh_gran[tid], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
      for (int idx = tid; idx < G * 8; idx += 256) {   // one lane per granule of every workgroup's record
        if ((idx & 7) == 7) continue;
        unsigned long long g;
        unsigned spins = 0;
        while ((unsigned)((g = __hip_atomic_load(&gran[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != xtag) {
          LPX_BARRIER_SLEEP;
          if (++spins > P.spin_max) { sh_fail = 1; break; }   // 1: a workgroup's candidate record
        }
        sh_part[idx] = (unsigned)g;
      }
      if (P.fences & 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __syncthreads();
      if (sh_fail) {
        if (lead) { ctl->status = 7 /* LPX_DEVICE_ERROR */; ctl->reserved = sh_fail + 1000 * s; chain_publish(ctl, P.host_snap); }
        return;
      }
    }
#ifndef LPX_CHAIN_DBG2
    if (P.dbg && lead) P.dbg[s * 5 + 2] = wall_clock64();
#endif

    // ------------------------------------------------------------------ phase B: the leaving row
    RatioRow mine = rr_none();
    double mine_a = 0.0, mine_b = 0.0;
    if (tid < G) {
      const unsigned* q = &sh_part[tid * 8];
      mine.ratio = from32(q[0], q[1]);
      mine_a = from32(q[2], q[3]);
      mine_b = from32(q[4], q[5]);
      mine.row = (int)q[6];
    }
#endif
    RatioRow w = rr_block_min(mine, sh_rr);
    if (w.row != INT_MAX && tid < G && mine.row == w.row) { sh_win[0] = mine_a; sh_win[1] = mine_b; }
    __syncthreads();
    const bool window = !P.dantzig && ld >= 256;  // every thread of workgroup 0 then owns one of the slots 0..255
    // Column ownership: with the hand-off, workgroup 0 owns slots 0..255 and nothing else (it is on everybody's
    // critical path), the other workgroups share the rest; fixed for the launch, as the private copies require.
    const bool solo0 = window && G > 1;
    const int jstep = solo0 ? T - 256 : T;
    const bool onehop = MG && P.onehop != 0;
    // the row `ll` (local index) of the current tableau at column j: the stale row run through the pending pivots,
    // restarted at pivot rb if that one left through the same row; needs sh_cs / sh_dv / sh_e of row ll
    auto row_value = [&](int j, int ll, int rb, int fo_b, int fn_b) -> double {
      const double* rowl = A + (int64_t)ll * ld;
      const double* src_x = rb < 0 ? &rowl[j]
                                   : (rb < KB ? &P.own_prow_o[(int64_t)rb * ld + j] : &P.own_prow[(int64_t)(rb - KB) * ld + j]);
      double x = *src_x;
      double prv[KB], prvo[KB];  // this thread's own stores, live chunks only (see phase A)
#pragma unroll
      for (int r0 = 0; r0 < KB; r0 += 8)
        if (LPX_CHAIN_LIVE(r0, fo_b, n_old)) {
#pragma unroll
          for (int q = 0; q < 8; ++q) prvo[r0 + q] = P.own_prow_o[(int64_t)(r0 + q) * ld + j];
        }
#pragma unroll
      for (int r0 = 0; r0 < KB; r0 += 8)
        if (LPX_CHAIN_LIVE(r0, fn_b, s)) {
#pragma unroll
          for (int q = 0; q < 8; ++q) prv[r0 + q] = P.own_prow[(int64_t)(r0 + q) * ld + j];
        }
#pragma unroll
      for (int r0 = 0; r0 < KB; r0 += 8) LPX_CHAIN_CHUNK_B(0, r0, fo_b, n_old, prvo)
      st_agent(&P.row0[(int64_t)s * ld + j], x);  // the row as the sweep of this block will read it (see col0)
#pragma unroll
      for (int r0 = 0; r0 < KB; r0 += 8) LPX_CHAIN_CHUNK_B(KB, r0, fn_b, s, prv)
      return x;
    };
    // per pending pivot u: col_u[ll] and -(col_u[ll] / p_u) into LDS, the restart pivot of row ll (LDS index or -1)
    auto row_params = [&](bool valid_row, int ll, int l_global) -> int {
      if (tid < 2 * KB) {
        const int r = tid & (KB - 1);
        const bool old = tid < KB;
        const bool valid = valid_row && (old ? r < n_old : r < s);
        bool same = false;
        if (valid) {
          const double csv = ld_agent((old ? P.col_o : P.col) + (int64_t)r * mp + ll);
          sh_cs[tid] = csv;
          sh_dv[tid] = -__ddiv_rn(csv, sh_p[tid]);
          same = sh_l[tid] == l_global;
        }
        const unsigned long long mask = __ballot(same);
        if ((tid & 63) == 0) sh_mask[tid >> 6] = mask;
      }
      __syncthreads();
      return chain_restart<KB>(sh_mask);
    };
    if constexpr (MG) {
      if (onehop) {
        // ---- this shard's candidate row goes to every device BEFORE anybody knows the winner
        const int mslot = (P.mail_slot0 + s) & 1;
        const bool have = w.row != INT_MAX;          // (identical in every workgroup of this shard)
        const int lc = have ? w.row - row0 : 0;
        const int rbc = row_params(have, lc, w.row);
        const int fo_c = rbc < 0 ? 0 : (rbc < KB ? rbc + 1 : n_old);
        const int fn_c = rbc >= KB ? rbc - KB + 1 : 0;
        if (have) {
          const int64_t base = (int64_t)(mslot * kMaxDevices + P.dev) * ld;
          for (int j = gid; j < (int)ld; j += (solo0 && blockIdx.x == 0) ? (int)ld : jstep) {
            const double x = j < n ? row_value(j, lc, rbc, fo_c, fn_c) : 0.0;
            for (int d = 0; d < P.n_dev; ++d) st_sys(&P.candrow_peer[d][base + j], x);
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
          if (P.fences & 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          for (int d = 0; d < P.n_dev; ++d)
            __hip_atomic_store(&P.arrive2_peer[d][(int64_t)(mslot * kMaxDevices + P.dev) * kChainMaxWgs + blockIdx.x],
                               (unsigned long long)xtag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();   // sh_cs / sh_dv / sh_mask are rewritten below
      }
    }
    int dstar = 0;   // one-hop: the shard whose candidate won
    if constexpr (MG) {
      // The two mailbox slots alternate with the decisions of the whole LOOP, not of the launch: a device may be one
      // decision ahead of a peer — also across the boundary between two launches (a block of odd length would reuse
      // the slot its last decision used) — never two.
      const int mslot = (P.mail_slot0 + s) & 1;
      // allreduce(min+loc) over the shards: this shard's winner goes into slot `dev` of every device's mailbox,
      // then every workgroup reduces the n_dev records of its own device's mailbox (lowest global row wins ties)
      if (lead) {
        const double wa = (w.row != INT_MAX) ? sh_win[0] : 0.0, wb = (w.row != INT_MAX) ? sh_win[1] : 0.0;
        for (int d = 0; d < P.n_dev; ++d) {
          MgMail* rec = &P.mail_peer[d][mslot * kMaxDevices + P.dev];
          st_sys(&rec->ratio, w.ratio);
          st_sys(&rec->a, wa);
          st_sys(&rec->bi, wb);
          __hip_atomic_store(&rec->row, w.row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (P.fences & 1) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        for (int d = 0; d < P.n_dev; ++d)
          __hip_atomic_store(&P.mail_peer[d][mslot * kMaxDevices + P.dev].tag, xtag, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_SYSTEM);
      }
      __syncthreads();  // sh_win is rewritten below
      RatioRow theirs = rr_none();
      double theirs_a = 0.0, theirs_b = 0.0;
      if (tid < P.n_dev) {
        const MgMail* rec = &P.mail_peer[P.dev][mslot * kMaxDevices + tid];
        unsigned spins = 0;
        while (__hip_atomic_load(&rec->tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != xtag) {
          LPX_BARRIER_SLEEP;
          if (++spins > P.spin_max) { sh_fail = 2 + 16 * tid; break; }   // 2: a peer's candidate record
        }
        if (P.fences & 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        theirs.ratio = ld_sys(&rec->ratio);
        theirs.row = __hip_atomic_load(&rec->row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        theirs_a = ld_sys(&rec->a);
        theirs_b = ld_sys(&rec->bi);
      }
      w = rr_block_min(theirs, sh_rr);
      if (sh_fail) { if (lead) { ctl->status = 7; ctl->reserved = sh_fail + 1000 * s; chain_publish(ctl, P.host_snap); } return; }
      if (w.row != INT_MAX && tid < P.n_dev && theirs.row == w.row) { sh_win[0] = theirs_a; sh_win[1] = theirs_b; sh_restart = tid; }
      __syncthreads();
      dstar = sh_restart;   // (only read in the one-hop form, and only when there is a winner)
      __syncthreads();      // sh_restart is reused further down
    }
    if (w.row == INT_MAX || !(w.ratio < kInf)) {  // getLeaving() == -1: unbounded
      if (lead) {
        ctl->status = 1; ctl->do_update = 0; ctl->l = -1; ctl->ratio = w.ratio; P.up[s].do_update = 0;
        chain_publish(ctl, P.host_snap);
      }
      return;
    }
    if (max_pivots >= 0 && pivots >= max_pivots) {
      if (lead) {
        ctl->status = 9 /* LPX_PIVOT_LIMIT */; ctl->do_update = 0; P.up[s].do_update = 0;
        chain_publish(ctl, P.host_snap);
      }
      return;
    }
    const int l = w.row;
    const double p = sh_win[0], raw_b = sh_win[1];
    if (p == 0.0) {  // ArithmeticException in the reference, LPState.java:139
      if (lead) { ctl->status = 8; ctl->do_update = 0; P.up[s].do_update = 0; chain_publish(ctl, P.host_snap); }
      return;
    }
    // Row l of the current tableau likewise; a pending pivot u* with the same leaving row REPLACED the row by its
    // normalised row: restart there (own_prow[u*]).  Column e_u of the row becomes -(col_u[l]/p_u) at pivot u.
    // On shards only the device that holds row l computes it; the others receive the normalised row (see below).
    const bool owner = !MG || (l >= row0 && l < row0 + m);
    const int ll = l - row0;  // local index of the leaving row on its owner
    int rb = -1;
    if (!onehop) rb = row_params(owner, ll, l);
    const int fo_b = rb < 0 ? 0 : (rb < KB ? rb + 1 : n_old);
    const int fn_b = rb >= KB ? rb - KB + 1 : 0;
#ifdef LPX_CHAIN_DBG2   // diagnostic build: stamp 1 = leaving row known and its column values fetched
    if (P.dbg && lead) P.dbg[s * 5 + 1] = wall_clock64();
#endif
    const double bl = __ddiv_rn(raw_b, p);                                         // :146
    const double inv_p = __ddiv_rn(1.0, p);                                        // :139
    RatioRow cand = rr_none();  // (key, slot): key 0 = first slot (reference), -c = largest coefficient (Dantzig)
    const double* cand_row = nullptr;   // one-hop: the winner's un-normalised row in THIS device's buffer
    if constexpr (MG) {
      if (onehop) {
        // wait until every workgroup of the WINNING shard has stored its columns of the candidate row here
        const int mslot = (P.mail_slot0 + s) & 1;
        if (tid < G) {
          const unsigned long long* aw = &P.arrive2_peer[P.dev][(int64_t)(mslot * kMaxDevices + dstar) * kChainMaxWgs + tid];
          unsigned spins = 0;
          while (__hip_atomic_load(aw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (unsigned long long)xtag) {
            LPX_BARRIER_SLEEP;
            if (++spins > P.spin_max) { sh_fail = 5 + 16 * tid; break; }   // 5: the winner's candidate-row arrival word
          }
          if (P.fences & 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        }
        __syncthreads();
        if (sh_fail) { if (lead) { ctl->status = 7; ctl->reserved = sh_fail + 1000 * s; chain_publish(ctl, P.host_snap); } return; }
        cand_row = P.candrow_peer[P.dev] + (int64_t)(mslot * kMaxDevices + dstar) * ld;
      } else if (!owner) {
        // wait until every workgroup of the owner has stored its columns of the normalised row into THIS device's
        // replica of the ring (one arrival word per owner workgroup, raised after its stores have drained)
        if (tid < G) {
          const unsigned long long* aw = &P.arrive_peer[P.dev][tid];
          unsigned spins = 0;
          while (__hip_atomic_load(aw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (unsigned long long)xtag) {
            LPX_BARRIER_SLEEP;
            if (++spins > P.spin_max) { sh_fail = 3 + 16 * tid; break; }   // 3: the owner's arrival word
          }
          if (P.fences & 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        }
        __syncthreads();
        if (sh_fail) { if (lead) { ctl->status = 7; ctl->reserved = sh_fail + 1000 * s; chain_publish(ctl, P.host_snap); } return; }
      }
    }
    const bool local_row = owner || onehop;   // this device computes the normalised row itself
    for (int j = gid; j < (int)ld; j += (solo0 && blockIdx.x == 0) ? (int)ld : jstep) {
      double x = 0.0;
      const double cj = ld_agent(&P.c[j]);  // this thread's own store (or the initial value)
      if (j < n) {
        if (onehop) x = ld_sys(&cand_row[j]);
        else if (owner) x = row_value(j, ll, rb, fo_b, fn_b);
      }
      double cn, pr;
      if (j == e) {
        pr = inv_p;
        cn = -__ddiv_rn(pc, p);                                                    // :172
      } else {
        pr = local_row ? __ddiv_rn(x, p) : ld_sys(&P.prow[(int64_t)s * ld + j]);   // :144 (the owner's value)
        cn = submul(cj, pc, pr);                                     // :177
      }
      if constexpr (MG) {
        if (owner && !onehop) {  // broadcast: the value goes into every other device's replica of the ring
          for (int d = 0; d < P.n_dev; ++d)
            if (d != P.dev) st_sys(&P.prow_peer[d][(int64_t)s * ld + j], pr);
        }
      }
      st_agent(&P.prow[(int64_t)s * ld + j], pr);
      P.own_prow[(int64_t)s * ld + j] = pr;
      st_agent(&P.c[j], cn);
      if (j < n && cn > kEps) {
        const RatioRow k2{P.dantzig ? -cn : 0.0, j, 0};
        cand = rr_min(cand, k2);
      }
      if (window && blockIdx.x == 0 && j == gid) {
        // Workgroup 0 has just finished slots 0..255.  Under the first-positive rule the next entering slot is the
        // lowest one with c > eps, so if there is one among them it is the answer, and everything the next phase A
        // reads about that slot (c[e], the pending pivot rows at e) was written by THIS workgroup, write-through:
        // drain, meet, and one lane publishes {sequence, slot} in a single 8-byte store.  The others then need no
        // second grid barrier, only this word.  No candidate here: publish "none", everybody takes the barrier.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const RatioRow w0 = rr_block_min(cand, sh_rr);
        if (tid == 0) {
          const unsigned long long rec = ((unsigned long long)(P.hand_base + (unsigned)s) << 32) |
                                         (unsigned)((w0.row == INT_MAX ? -2 : w0.row) + 2);
          __hip_atomic_store(P.hand, rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if constexpr (MG) {
      if (owner && !onehop && P.n_dev > 1) {  // this workgroup's columns are on their way to every peer: drain, meet, signal
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
          if (P.fences & 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          for (int d = 0; d < P.n_dev; ++d)
            if (d != P.dev)
              __hip_atomic_store(&P.arrive_peer[d][blockIdx.x], (unsigned long long)xtag, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
#ifdef LPX_CHAIN_DBG2   // diagnostic build: stamp 2 = this thread's columns done (workgroup 0: word published)
    if (P.dbg && lead) P.dbg[s * 5 + 2] = wall_clock64();
#endif
    // the row owners add pivot s to their two running columns: the entering column after the pivot (what a later
    // decision restarts from) and b — after the columns, so that workgroup 0 publishes the entering slot first
    for (int i = gid; i < m; i += T) {
      const double colv = P.own_col[(int64_t)s * mp + i];
      const double* src_b = use_b ? &b[i] : &P.own_b[i];
      const double bcur = *src_b;
      P.own_dvc[(int64_t)s * mp + i] = (row0 + i == l) ? inv_p : -__ddiv_rn(colv, p);    // :157 / :139
      P.own_b[i] = (row0 + i == l) ? bl : submul(bcur, colv, bl);          // :146 / :164
    }
    {
      const RatioRow w2 = rr_block_min(cand, sh_rr);
      if (tid == 0) {
        st_agent(&P.partB[blockIdx.x].ratio, w2.ratio);
        st_agent(&P.partB[blockIdx.x].row, w2.row);
        sh_e[KB + s] = e; sh_l[KB + s] = l; sh_p[KB + s] = p; sh_bl[KB + s] = bl;
      }
    }
    if (lead) {
      v = addmul(v, bl, pc);                                         // :171
      const int32_t perm_e = P.perm[e], perm_l = P.perm[n + l];                    // exchangeIndexes :311-320
      P.perm[e] = perm_l;
      P.perm[n + l] = perm_e;
      if (track >= 0) {                                                            // LPSolver.java:151-155
        if (e == track) track = l + n;
        else if (l + n == track) track = e;
      }
      LpxCtl& up = P.up[s];
      up.p = p; up.bl = bl; up.e_cur = e; up.l = l; up.e_next = -1; up.parity = 0; up.do_update = 1;
    }
    if (P.dbg && lead) P.dbg[s * 5 + 3] = wall_clock64();
    int e_next = -2;
    if (window) {  // one lane per workgroup waits for workgroup 0's word (bounded, like the barrier)
      if (tid == 0) {
        const unsigned want = P.hand_base + (unsigned)s;
        unsigned long long rec = 0;
        unsigned spins = 0;
        for (;;) {
          rec = __hip_atomic_load(P.hand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((unsigned)(rec >> 32) == want) break;
          LPX_BARRIER_SLEEP;
          if (++spins > P.spin_max) { sh_fail = 4; break; }   // 4: workgroup 0's hand-off word
        }
        sh_restart = (int)(unsigned)(rec & 0xffffffffu) - 2;
      }
      __syncthreads();
      if (sh_fail) { if (lead) { ctl->status = 7; ctl->reserved = sh_fail + 1000 * s; chain_publish(ctl, P.host_snap); } return; }
      e_next = sh_restart;
      __syncthreads();  // sh_restart is reused by the next phase
    }
    if (e_next < 0) {  // no hand-off (Dantzig, narrow tableau) or no candidate in its window: the full exchange
      target += (unsigned)G;
      if (!grid_barrier(P.bar, target, &sh_fail, P.fences, P.spin_max)) {
        if (lead) { ctl->status = 7; chain_publish(ctl, P.host_snap); }
        return;
      }
      RatioRow m2 = rr_none();
      if (tid < G) { m2.ratio = ld_agent(&P.partB[tid].ratio); m2.row = ld_agent(&P.partB[tid].row); }
      const RatioRow w3 = rr_block_min(m2, sh_rr);
      e_next = (w3.row == INT_MAX) ? -1 : w3.row;
    }
    if (P.dbg && lead) P.dbg[s * 5 + 4] = wall_clock64();
    pivots += 1;
    parity ^= 1;
    if (lead) {
      ctl->v = v; ctl->p = p; ctl->bl = bl; ctl->pc = pc; ctl->ratio = w.ratio;
      ctl->e_cur = e; ctl->l = l; ctl->e_next = e_next; ctl->parity = parity; ctl->pivots = pivots;
      ctl->track = track; ctl->do_update = 1;
      if (e_next < 0) {
        ctl->status = 0 /* LPX_OPTIMAL once the sweep has applied this pivot */;
        if (s + 1 < nb) P.up[s + 1].do_update = 0;  // the block ends here: the sweep counts leading valid slots
      }
    }
    if (e_next < 0 || s + 1 == nb) {
      if (lead) chain_publish(ctl, P.host_snap);
      return;
    }
    e = e_next;
  }
}

// ---- k_block_chain2: the same decisions on a shorter critical path (one device and shards; option chain_form = 1) ----
// k_block_chain_t spends a decision in ~8 dependent memory round trips and ~3 000 instructions that a lone wave per SIMD
// executes one after the other (13-14 us alone, 18-20 us beside a sweep, whatever it computes).  What is not a data
// dependency of the algorithm is taken off that path here:
//   round trips  A phase asks for everything at once — the few lanes that serve the pending pivots issue their loads
//                FIRST (loads return in order), then every thread its row / column entry and its own ring values; one
//                workgroup meeting later the arithmetic runs.  Nothing is drained in front of a publication: what the
//                NEXT phase A needs of the decision just taken — c[e'] and prow_s[e'] — travels INSIDE workgroup 0's
//                hand-off record (five self-validating 8-byte granules {32 data bits, sequence tag}); everything older a
//                later decision reads across workgroups (col_u[l], prow_u[e] of EARLIER decisions) was stored at least
//                one decision before it is asked for, and every wave passes an s_waitcnt vmcnt(0) (in front of the
//                meeting of its next phase, when its loads are here anyway) before its workgroup publishes anything
//                newer: a candidate record of decision s+1 implies col_s is visible, a hand-off record of decision s+1
//                implies prow_s and c of decision s are.  The stale copies kept for the fix-up (col0, row0) are plain
//                stores: only the next kernel reads them.
//   instructions wave minima by DPP butterflies + four v_readlane instead of six rounds of LDS shuffles, ONE workgroup meeting
//                per minimum (rr_block_min_rec), the workgroups' candidates reduced by every wave for itself; the pending
//                pivots' parameters come from LDS one chunk of eight AHEAD of the arithmetic that uses them, as 16-byte
//                reads; the pending pivots themselves are a ladder without branches or selects (chain8 / chain8_from, round
//                5): a thread's chain STARTS behind the last pending pivot that replaced its value, one window of live
//                chunks in registers.
//   workgroup 0  (the one everybody waits for) only does what the hand-off needs: the loop state (v, perm, the ring's
//                parameter block, the host's snapshot) is kept by the LAST workgroup's first thread, the candidates for
//                the rare full exchange are reduced only when that exchange happens, and the column a later restart
//                starts from (-(col_s / p_s)) is computed when a restart needs it, not after every decision.
// When workgroup 0 finds no entering slot in its window (2 % of the decisions; always under Dantzig pricing or with
// fewer than 256 columns) the decision ends as in k_block_chain_t: drain, grid barrier, every workgroup reduces the
// candidates, and the next phase A loads c[e] and prow_s[e] from memory.
// Arithmetic, ownership, ring layout and the bounded spins are those of k_block_chain_t (own_dvc is not used; the start
// indices own_rs_a / own_rs_b are this kernel's own); the kernels must not take turns INSIDE one loop (the private ring
// copies differ in that — lpx_engine.cpp blocked_loop_overlapped says why they cannot), between loops they may.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
  return __hiloint2double(dpp_i32<CTRL>(__double2hiint(x)), dpp_i32<CTRL>(__double2loint(x)));
}
__device__ __forceinline__ double lane_f64(double x, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane));
}
// Lexicographic (ratio, row) minimum of a wave, valid in EVERY lane.  Butterflies inside the rows of 16 lanes (quad
// permutes, half mirror, mirror: each lane ends with its row's minimum), the four rows by v_readlane.  Rows are distinct
// (or INT_MAX = none); equal ratios -> the lowest row, as rr_min.
__device__ __forceinline__ RatioRow rr_wave_min_all(RatioRow x) {
  double r = x.ratio;
  { const double o = dpp_f64<0xB1>(r); r = o < r ? o : r; }     // quad_perm [1,0,3,2]
  { const double o = dpp_f64<0x4E>(r); r = o < r ? o : r; }     // quad_perm [2,3,0,1]
  { const double o = dpp_f64<0x141>(r); r = o < r ? o : r; }    // row_half_mirror
  { const double o = dpp_f64<0x140>(r); r = o < r ? o : r; }    // row_mirror
  const double r0 = lane_f64(r, 0), r1 = lane_f64(r, 16), r2 = lane_f64(r, 32), r3 = lane_f64(r, 48);
  const double ra = r1 < r0 ? r1 : r0, rb = r3 < r2 ? r3 : r2;
  const double rmin = rb < ra ? rb : ra;
  int row = (x.ratio == rmin) ? x.row : INT_MAX;
  row = min(row, dpp_i32<0xB1>(row));
  row = min(row, dpp_i32<0x4E>(row));
  row = min(row, dpp_i32<0x141>(row));
  row = min(row, dpp_i32<0x140>(row));
  const int q = min(min(__builtin_amdgcn_readlane(row, 0), __builtin_amdgcn_readlane(row, 16)),
                    min(__builtin_amdgcn_readlane(row, 32), __builtin_amdgcn_readlane(row, 48)));
  return RatioRow{rmin, q, 0};
}
// Workgroup meeting for LDS traffic only.  __syncthreads() is a workgroup-scope FENCE and a barrier: the fence waits for
// every global store of the wave (s_waitcnt vmcnt(0)) — 0.5-1 us behind a handful of write-through stores, i.e. exactly
// the drain k_block_chain2 keeps off its critical path.  Here only the wave's LDS operations are waited for.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// workgroup-wide; result valid in every thread.  `sh` needs blockDim.x / 64 entries.
__device__ __forceinline__ RatioRow rr_block_min2(RatioRow x, RatioRow* sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  x = rr_wave_min_all(x);
  lds_barrier();
  if (lane == 0) sh[wave] = x;
  lds_barrier();
  RatioRow r = sh[0];
  for (int w = 1; w < nw; ++w) r = rr_min(r, sh[w]);
  return r;
}

// ---- the pending-pivot ladder of k_block_chain2 ---------------------------------------------------------------------
// Eight pending pivots applied in order to one value: x <- x - c[q] * r[q] (submul: product and difference rounded
// separately, or ONE v_fma_f64 in the fused compilation).  What a step costs a lone wave per SIMD, measured
// (scripts/micro/chain_step_cost.hip, profiles/r05_chain_step_cost_and_traces_before.txt, shader cycles per step):
//   dependent v_fma_f64 6.6 | v_mul_f64 + v_add_f64 11.6 | + a scalar bit test and a branch NOT taken (round 4, old ring
//   half) 22.7 | the same with the rare path inline, i.e. a branch TAKEN over it every step (round 4, own half) 36.5 |
//   v_cmp + two v_cndmask 21.3 | s_mov exec + v_cmpx 24.9 | eight v_cmp into SGPR pairs per chunk, then s_mov exec per step 16.5.
// So the ladder has no branch and no select: a chunk of eight steps is either `chain8` — straight multiply-adds — or
// `chain8_from`, where a lane takes step q only if q >= stc (its start index relative to the chunk): the steps in front of
// a lane's restart point (its row / slot was REPLACED by a pending pivot, LPState.java:139-145 / :157) are skipped by
// EXEC.  Which of the two a chunk needs is uniform per wave (does any of its lanes start behind the chunk's first step).
#if LPX_FUSED
#define LPX_C2_T_OUT
#define LPX_C2_STEP(q) "v_fma_f64 %[x], -%[c" #q "], %[r" #q "], %[x]\n\t"
#else
#define LPX_C2_T_OUT , [t] "=&v"(t)
#define LPX_C2_STEP(q) "v_mul_f64 %[t], %[c" #q "], %[r" #q "]\n\tv_add_f64 %[x], %[x], -%[t]\n\t"
#endif
#define LPX_C2_INS                                                                                                           \
  [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]), [c4] "v"(c[4]), [c5] "v"(c[5]), [c6] "v"(c[6]), [c7] "v"(c[7]), \
  [r0] "v"(r[0]), [r1] "v"(r[1]), [r2] "v"(r[2]), [r3] "v"(r[3]), [r4] "v"(r[4]), [r5] "v"(r[5]), [r6] "v"(r[6]), [r7] "v"(r[7])
__device__ __forceinline__ void chain8(double& x, const double* c, const double* r) {
#if !LPX_FUSED
  double t;
#endif
  asm volatile(LPX_C2_STEP(0) LPX_C2_STEP(1) LPX_C2_STEP(2) LPX_C2_STEP(3) LPX_C2_STEP(4) LPX_C2_STEP(5) LPX_C2_STEP(6) LPX_C2_STEP(7)
               : [x] "+v"(x) LPX_C2_T_OUT : LPX_C2_INS);
}
__device__ __forceinline__ void chain8_from(double& x, const double* c, const double* r, int stc) {
#if !LPX_FUSED
  double t;
#endif
  unsigned long long sv, m0, m1, m2, m3, m4, m5, m6, m7;
  // the eight masks under the FULL mask of the enclosing code (inactive lanes compare as 0 and stay inactive), then one
  // s_mov exec per step; EXEC is restored before the block ends (the compiler never sees it change)
  asm volatile("s_mov_b64 %[sv], exec\n\t"
               "v_cmp_ge_i32 %[m0], 0, %[stc]\n\tv_cmp_ge_i32 %[m1], 1, %[stc]\n\tv_cmp_ge_i32 %[m2], 2, %[stc]\n\t"
               "v_cmp_ge_i32 %[m3], 3, %[stc]\n\tv_cmp_ge_i32 %[m4], 4, %[stc]\n\tv_cmp_ge_i32 %[m5], 5, %[stc]\n\t"
               "v_cmp_ge_i32 %[m6], 6, %[stc]\n\tv_cmp_ge_i32 %[m7], 7, %[stc]\n\t"
               "s_mov_b64 exec, %[m0]\n\t" LPX_C2_STEP(0) "s_mov_b64 exec, %[m1]\n\t" LPX_C2_STEP(1)
               "s_mov_b64 exec, %[m2]\n\t" LPX_C2_STEP(2) "s_mov_b64 exec, %[m3]\n\t" LPX_C2_STEP(3)
               "s_mov_b64 exec, %[m4]\n\t" LPX_C2_STEP(4) "s_mov_b64 exec, %[m5]\n\t" LPX_C2_STEP(5)
               "s_mov_b64 exec, %[m6]\n\t" LPX_C2_STEP(6) "s_mov_b64 exec, %[m7]\n\t" LPX_C2_STEP(7)
               "s_mov_b64 exec, %[sv]"
               : [x] "+v"(x) LPX_C2_T_OUT, [sv] "=&s"(sv), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3),
                 [m4] "=&s"(m4), [m5] "=&s"(m5), [m6] "=&s"(m6), [m7] "=&s"(m7)
               : LPX_C2_INS, [stc] "v"(stc));
}
#undef LPX_C2_INS
#undef LPX_C2_STEP
#undef LPX_C2_T_OUT
// index of the highest set bit of a pair of ballots (bit u of lo = slot u, of hi = slot 64 + u), -1 if none
__device__ __forceinline__ int top_bit(unsigned long long lo, unsigned long long hi) {
  return hi ? 127 - __clzll((long long)hi) : (lo ? 63 - __clzll((long long)lo) : -1);
}

// Workgroup minimum of (ratio, row) WITH the winner's two payload values in ONE meeting: every wave reduces itself (DPP),
// the lane that holds its wave's minimum writes the wave's record, the workgroup meets, every thread reads the NW records.
// `slot` alternates from call to call: a record is rewritten two calls later, behind the meeting of the call in between.
// (rr_block_min2 + a winner's exchange took three meetings and an LDS round trip each: 0.8-0.9 us of a decision.)
struct MinRec {
  double ratio, a, b;
  int row, pad;
};
template <int NW>
__device__ __forceinline__ RatioRow rr_block_min_rec(RatioRow x, double& a, double& b, MinRec (*sh)[NW], int slot) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const RatioRow w = rr_wave_min_all(x);
  if (w.row == INT_MAX ? lane == 0 : x.row == w.row) sh[slot][wave] = MinRec{w.ratio, a, b, w.row, 0};
  lds_barrier();
  MinRec r = sh[slot][0];
#pragma unroll
  for (int k = 1; k < NW; ++k) {
    const MinRec o = sh[slot][k];
    if (o.ratio < r.ratio || (o.ratio == r.ratio && o.row < r.row)) r = o;
  }
  a = r.a;
  b = r.b;
  return RatioRow{r.ratio, r.row, 0};
}

constexpr int kChain2Threads = 256;
// MG: the shards of an lpx_multi (row blocks on several devices, see "multi-device decisions" above).  The two-hop exchange
// of k_block_chain_t on this kernel's shorter path: after the workgroups of a device have agreed on the device's candidate,
// its first thread stores it into every device's mailbox and every WAVE reduces the mailbox of its own device; the shard
// that owns the leaving row computes the normalised row, stores it into every peer's replica of the ring as it goes and
// raises its workgroups' arrival words behind a drain; the other shards wait for those words and read the row from their
// replica.  Everything replicated (c, v, perm, the hand-off of workgroup 0) is computed on every device, identically.
template <int KB, int NT, bool MG>
__global__ __launch_bounds__(NT) void k_block_chain2_t(const ChainArgs P) {
  static_assert(KB == 32 || KB == 64, "ring half of 32 or 64 slots");
  static_assert(NT % 64 == 0 && NT >= 256 && NT >= 2 * KB + 64, "lanes for the pending pivots and the loader lane");
  constexpr int NC = 2 * KB / 8;   // chunks of eight pending pivots: [0, KB / 8) the previous block's, then this block's
  constexpr int kWin = 8;          // live chunks whose ring copies a thread holds in registers at a time
  __shared__ RatioRow sh_rr[NT / 64];
  __shared__ MinRec sh_rec[2][NT / 64];
  __shared__ unsigned sh_part[kChainMaxWgs * 8], sh_hand[8];
  __shared__ __attribute__((aligned(16))) double sh_pe[2 * KB], sh_cs[2 * KB], sh_dv[2 * KB], sh_p[2 * KB], sh_bl[2 * KB];
  __shared__ __attribute__((aligned(16))) int sh_e[2 * KB], sh_l[2 * KB];
  __shared__ int sh_fail;
#ifdef LPX_CHAIN2_FINE   // diagnostic build: 16 stamps per decision (slots 8..15: inside the phases)
#define LPX_C2_STAMP(k) if (P.dbg && lead) P.dbg[s * 16 + (k)] = wall_clock64();
#define LPX_C2_STRIDE 16
#else
#define LPX_C2_STAMP(k)
#define LPX_C2_STRIDE 8
#endif
  const int row0 = MG ? P.shard_row0 : 0;   // global index of local row 0 (the ring's parameter blocks name GLOBAL rows)
  const double* __restrict__ A = P.A;
  const double* __restrict__ b = P.b;
  const int64_t ld = P.ld, mp = P.mp;
  const int n = P.n, m = P.m, nb = P.nb, n_old = P.n_old;
  LpxCtl* const ctl = P.ctl;
#ifdef LPX_CHAIN2_ONE_XCD   // experiment build: eight times the grid, only the workgroups the dispatcher deals to XCD 0 take part
  if (blockIdx.x & 7u) return;
  const unsigned bid = blockIdx.x >> 3;
  const int G = gridDim.x >> 3;
#else
  const unsigned bid = blockIdx.x;   // (workgroup index, as every use below names it)
  const int G = gridDim.x;
#endif
  const int T = G * NT, tid = threadIdx.x, gid = bid * NT + tid;
  const bool lead = gid == 0;                                   // stamps, the next launch's barrier counter
  const bool book = bid == (unsigned)(G - 1) && tid == 0;   // keeps the loop state (off workgroup 0's path)
  if (tid == 0) {
    sh_fail = 0;
    if (P.census) P.census[bid] = xcc_id() + 1u;
  }
  if (lead) st_agent(reinterpret_cast<int32_t*>(P.bar_next), 0);
  int e = ctl->e_next;
  if (ctl->status != kRunning || e < 0 || nb < 1) {
    if (book && nb >= 1) P.up[0].do_update = 0;
    if (book) chain_publish(ctl, P.host_snap);
    return;
  }
  // Identity padding instead of per-step predicates.  Inside a chunk of eight pending pivots every step runs without a
  // test; a step that is not (yet) a pivot must leave the value alone for every input, -0.0 included: multiplier +0 and
  // row / column value +0 (x - (+0 * +0) = x, fma(-(+0), +0, x) = x), entering slot / leaving row -1 (matches nothing).
  // The LDS parameters of such steps hold these from here on, and the thread zeroes its own copies of THIS block's ring
  // half (slot s is filled at decision s); the previous block's half keeps the zeros behind its last pivot.
  // (Per-step uniform predicates cost ~130 SGPR masks, spilled through v_writelane / v_readlane at every decision.)
  if (tid < 2 * KB) {
    sh_pe[tid] = 0.0; sh_cs[tid] = 0.0; sh_dv[tid] = 0.0; sh_p[tid] = 1.0; sh_bl[tid] = 0.0;
    sh_e[tid] = -1; sh_l[tid] = -1;
  }
  lds_barrier();
  if (tid < n_old) {
    const LpxCtl& q = P.up_o[tid];
    sh_e[tid] = q.e_cur; sh_l[tid] = q.l; sh_p[tid] = q.p; sh_bl[tid] = q.bl;
  }
  int64_t pivots = ctl->pivots;
  const int64_t max_pivots = ctl->max_pivots;
  double v = ctl->v;
  int track = ctl->track, parity = ctl->parity;
  unsigned target = 0;
  // Column ownership, fixed for the launch.  With the hand-off, workgroup 0 owns slots 0..255 and nothing else (it is
  // on everybody's critical path); the other workgroups share the rest.  Without it every thread strides over all slots.
  const bool window = !P.dantzig && ld >= 256 && G > 1;
  const int jfirst = !window ? gid : (bid == 0 ? (tid < 256 ? tid : (int)ld) : 256 + (int)(bid - 1) * NT + tid);
  const int jstep = !window ? T : (bid == 0 ? (int)ld : T - NT);
  const char* const oc_o = reinterpret_cast<const char*>(P.own_col_o);
  const char* const oc_n = reinterpret_cast<const char*>(P.own_col);
  const char* const op_o = reinterpret_cast<const char*>(P.own_prow_o);
  const char* const op_n = reinterpret_cast<const char*>(P.own_prow);
  const uint32_t mp8_0 = (uint32_t)mp * 8u, ld8_0 = (uint32_t)ld * 8u;   // (launcher: 64 * max(mp, ld) * 8 < 2^32)
  bool have_rec = false;   // sh_hand[1..4] hold c[e] and prow_{s-1}[e] of the record that named e (uniform)
  unsigned long long* const hand = P.hand;
  for (int i = gid; i < m; i += T) {     // this block's half of the thread's own column copies: +0 until a decision fills a slot
#pragma unroll 8
    for (int u = 0; u < KB; ++u) P.own_col[(int64_t)u * mp + i] = 0.0;
  }
  for (int j = jfirst; j < (int)ld; j += jstep) {
#pragma unroll 8
    for (int u = 0; u < KB; ++u) P.own_prow[(int64_t)u * ld + j] = 0.0;
  }
  lds_barrier();
  // Per row (column) of this thread: the LAST pending pivot that left through it (entered at it), as an index into the
  // LDS parameters, -1 if none.  That pivot REPLACED the row's (column's) values, so the thread's chain of a later decision
  // starts behind it (chain8_from).  Here the previous block's pivots; a decision of this launch adds its own (the
  // owner of row l / slot e stores KB + s).  The thread's own plain stores, re-read only by itself.
  for (int i = gid; i < m; i += T) {
    int r = -1;
    for (int u = 0; u < n_old; ++u) r = sh_l[u] == row0 + i ? u : r;
    P.own_rs_a[i] = r;
  }
  for (int j = jfirst; j < (int)ld; j += jstep) {
    int r = -1;
    for (int u = 0; u < n_old; ++u) r = sh_e[u] == j ? u : r;
    P.own_rs_b[j] = r;
  }

  for (int s = 0; s < nb; ++s) {
    // ------------------------------------------------------------------ phase A: column e, ratio test
    if (P.dbg && lead) P.dbg[s * LPX_C2_STRIDE + 0] = wall_clock64();
    // the ring pitches, made opaque per decision: otherwise the 128 slot offsets of the ring loads are hoisted out of the
    // decision loop as loop invariants and live (spilled) across it
    uint32_t mp8 = mp8_0, ld8 = ld8_0;
    asm volatile("" : "+s"(mp8), "+s"(ld8));
    const int r_mine = tid & (KB - 1);
    const bool old_mine = tid < KB;
    const bool valid_mine = tid < 2 * KB && (old_mine ? r_mine < n_old : r_mine < s);
    // Restart pivot of column e: the LAST pending pivot that entered at slot e left -(col / p) there (1 / p in its own
    // row); the column restarts from it.  Every wave asks for itself (lane u looks at slot u; an empty slot holds -1 and
    // matches nothing): a question to LDS only, no meeting.
    int ra;
    {
      const unsigned long long lo = __ballot(sh_e[tid & 63] == e);
      unsigned long long hi = 0;
      if constexpr (KB == 64) hi = __ballot(sh_e[64 + (tid & 63)] == e);
      ra = top_bit(lo, hi);
    }
    const int fo_a = ra < 0 ? 0 : (ra < KB ? ra + 1 : n_old);
    const int fn_a = ra >= KB ? ra - KB + 1 : 0;
    const bool use_b = P.b_from_tableau && s == 0;
    // The LIVE chunks of the ladder in order (a chunk = eight LDS slots): the previous block's [co0, co1), then this block's
    // [cn0, cn1) — what lies behind the restart pivot.  Under the first-positive rule the same few slots and rows come back
    // all the time (dense random LPs: three decisions of four restart inside the last 48 pending pivots), so the live part is
    // short; a thread holds ONE window of kWin chunks of its ring copies in registers (128 VGPRs whatever KB is — the whole
    // ring of a 64-slot launch would be 256) and a ladder longer than that takes a second round trip for the next window.
    const int co0_a = fo_a >> 3, co1_a = fo_a < n_old ? (n_old + 7) >> 3 : co0_a;
    const int cn0_a = fn_a >> 3, cn1_a = fn_a < s ? (s + 7) >> 3 : cn0_a;
    const int Lo_a = co1_a - co0_a, L_a = Lo_a + (cn1_a - cn0_a);
    auto chunk_a = [&](int k) { return (k < Lo_a ? co0_a + k : KB / 8 + cn0_a + (k - Lo_a)) & (NC - 1); };   // LDS chunk of live chunk k
    // every load of the phase in ONE round trip; the lanes of the pending pivots first
    double pe_mine = 0.0, pc_mine = 0.0;
    const bool pe_from_rec = have_rec && tid == KB + s - 1;   // (s >= 1 whenever have_rec)
    if (valid_mine && !pe_from_rec) {   // (on a shard: possibly stored by a peer device)
      const double* const pe_src = (old_mine ? P.prow_o : P.prow) + (int64_t)r_mine * ld + e;
      pe_mine = MG ? ld_sys(pe_src) : ld_agent(pe_src);
    }
    if (tid == NT - 1 && !have_rec) pc_mine = ld_agent(&P.c[e]);
    RatioRow best = rr_none();
    double best_a = 0.0, best_b = 0.0;
    double a = 0.0, bi = 0.0, a_first = 0.0, b_first = 0.0;
    int rs = -1;         // the row's own start index (see the launch prologue)
    double cv[kWin * 8];   // this thread's own stores: one window of live chunks
    auto load_window_a = [&](int i, int w0) {
#pragma unroll
      for (int k = 0; k < kWin; ++k) {
        if (w0 + k < L_a) {   // (uniform)
          const int c = chunk_a(w0 + k);
          const char* const base = (c < KB / 8 ? oc_o : oc_n) + (uint32_t)((c * 8) & (KB - 1)) * mp8;
#pragma unroll
          for (int q = 0; q < 8; ++q) cv[k * 8 + q] = *reinterpret_cast<const double*>(base + ((uint32_t)q * mp8 + (uint32_t)i * 8u));
        }
      }
    };
    auto load_row = [&](int i) {
      const uint32_t i8 = (uint32_t)i * 8u;   // uniform base (SGPRs) + one 32-bit lane offset per load: the rings are < 4 GiB
      // a pending pivot entered at the same slot: the column restarts from what that pivot left there, -(col / p)
      // (1 / p in its own row), computed from the thread's copy of the column as it was BEFORE that pivot
#ifdef LPX_DIAG_BUILD
      a = ra < 0 ? ((P.diag & 1) ? A[(int64_t)(e & 7) * ld + i] : A[(int64_t)i * ld + e])
#else
      a = ra < 0 ? A[(int64_t)i * ld + e]
#endif
                 : *reinterpret_cast<const double*>((ra < KB ? oc_o : oc_n) + ((uint32_t)(ra & (KB - 1)) * mp8 + i8));
      bi = use_b ? b[i] : P.own_b[i];
      rs = P.own_rs_a[i];
      load_window_a(i, 0);
    };
    int i = gid;
    if (i < m) load_row(i);
    // While the loads fly: how far into the ladder does some lane of this wave start late?  reach[p] = 1 + the highest
    // pending pivot whose OWN row is among the wave's 64 rows of pass p (lane u asks for slot u; all lanes are active
    // here), at least ra + 1; chunks from there on are straight multiply-adds for the whole wave.
    int reach_a[2];
    {
      const int row_w = row0 + (int)(bid * NT) + (tid & ~63);
      unsigned long long h0[2] = {0, 0}, h1[2] = {0, 0};
#pragma unroll
      for (int h = 0; h < 2 * KB / 64; ++h) {
        const int lu = sh_l[h * 64 + (tid & 63)];
        h0[h] = __ballot((unsigned)(lu - row_w) < 64u);
        h1[h] = __ballot((unsigned)(lu - row_w - T) < 64u);
      }
      reach_a[0] = max(ra, top_bit(h0[0], h0[1])) + 1;
      reach_a[1] = max(ra, top_bit(h1[0], h1[1])) + 1;
    }
    // everything has been asked for; by the time it is here, whatever this wave stored during the previous decision has
    // long landed: the drain that makes those stores visible before this workgroup publishes anything newer is free
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (P.dbg && lead) P.dbg[s * LPX_C2_STRIDE + 1] = wall_clock64();   // this wave's loads of phase A are here
    if (valid_mine) sh_pe[tid] = pe_from_rec ? from32(sh_hand[3], sh_hand[4]) : pe_mine;
    if (tid == NT - 1 && !have_rec) { sh_hand[1] = lo32(pc_mine); sh_hand[2] = hi32(pc_mine); }
    lds_barrier();
    LPX_C2_STAMP(8)
    int pass_a = 0;
    while (i < m) {
      const int ig = row0 + i;   // global row
      const int reach = pass_a == 0 ? reach_a[0] : pass_a == 1 ? reach_a[1] : 2 * KB;   // (later passes: every chunk masked)
      ++pass_a;
      // where this lane's chain starts, and from what: behind the later of the column's restart pivot and the row's own
      // last pivot.  The row's own pivot rs REPLACED the row by the normalised pivot row: its entry in column e is
      // prow_rs[e] (:139-145); otherwise the restart value -(col / p) (:157; rs == ra: the row of the restart pivot
      // itself, 1 / p = prow_ra[e], the first case again), otherwise the tableau's entry.
      const int st = max(ra, rs) + 1;
      if (rs >= ra && rs >= 0) a = sh_pe[rs];
      else if (ra >= 0) a = -__ddiv_rn(a, sh_p[ra]);                                  // :157
      // the pending pivots in order; their parameters one chunk ahead of the arithmetic, 16-byte LDS reads.  No memory
      // operation inside the ladder: a store there makes the compiler wait for it (vmcnt(0)) in front of the next chunk.
      double pe8[2][8];
      double a_mid = a;   // the value between the two blocks' pivots: the entry the sweep of THIS block will read
      auto ladder_a = [&](const int w0) {
        auto params = [&](int k) {
          if (k < kWin) {
            const double* const src = &sh_pe[chunk_a(w0 + k) * 8];
#pragma unroll
            for (int q = 0; q < 8; ++q) pe8[k & 1][q] = src[q];
          }
        };
        params(0);
#pragma unroll
        for (int k = 0; k < kWin; ++k) {
          if (w0 + k >= L_a) break;   // (uniform) what lies behind the last pivot inside a live chunk is identity by its data
          params(k + 1);
          __builtin_amdgcn_sched_barrier(0);
          if (w0 + k == Lo_a) a_mid = a;
          const int c8 = chunk_a(w0 + k) * 8;
          if (__builtin_expect(c8 >= reach, 1)) chain8(a, &cv[k * 8], pe8[k & 1]);
          else chain8_from(a, &cv[k * 8], pe8[k & 1], st - c8);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      ladder_a(0);
      if constexpr (NC > kWin) {   // (a 64-slot launch, rare: more than kWin live chunks — a second round trip, no loop:
        if (L_a > kWin) {          //  a loop made the compiler copy the whole window between its prologue and its body)
          load_window_a(i, kWin);
          ladder_a(kWin);
        }
      }
      if (Lo_a >= L_a) a_mid = a;   // no pivot of this block behind the boundary
      // (fix-up: next kernel.  A lane that starts inside this block's half stores its start value: the fix-up replaces that
      // entry at the same pivot, whatever it held.)
      P.col0[(int64_t)s * mp + i] = a_mid;
      LPX_C2_STAMP(9)
      st_agent(&P.col[(int64_t)s * mp + i], a);
      P.own_col[(int64_t)s * mp + i] = a;
      if (i == gid) { a_first = a; b_first = bi; }   // (kept for the b update behind phase B: no reload)
      const double rt = ratio_of(a, bi);
      if (rt < best.ratio) {  // i ascends per thread: strict < keeps the lowest row among equal ratios
        best = RatioRow{rt, ig, 0};
        best_a = a;
        best_b = bi;
      }
      i += T;
      __builtin_amdgcn_sched_barrier(0);   // (the next pass's loads stay behind this pass's arithmetic: one set of registers)
      if (i < m) load_row(i);   // (tableaus taller than the grid: a round trip per further row)
    }
    LPX_C2_STAMP(10)
    const double pc = from32(sh_hand[1], sh_hand[2]);   // c[e]: from the record that named e, or the loader lane
    const unsigned xtag = P.hand_base + (unsigned)s;    // sequence number of this decision (unique over launches)
    {
      double wa = best_a, wb = best_b;
      const RatioRow w = rr_block_min_rec<NT / 64>(best, wa, wb, sh_rec, s & 1);
      LPX_C2_STAMP(11)
      if (w.row == INT_MAX) { wa = 0.0; wb = 0.0; }
      if (P.dbg && lead) P.dbg[s * LPX_C2_STRIDE + 2] = wall_clock64();
      // the workgroup's candidate as seven tagged granules — NOT behind a drain: nobody reads col_s across workgroups
      // before decision s + 1, and every wave has passed a vmcnt(0) (above) since its stores of decision s - 1
      unsigned long long* const gran = reinterpret_cast<unsigned long long*>(P.partA) + (size_t)(s & 1) * kChainMaxWgs * 8;
      if (tid < 7) {
        const unsigned d = tid == 0 ? lo32(w.ratio) : tid == 1 ? hi32(w.ratio) : tid == 2 ? lo32(wa) : tid == 3 ? hi32(wa)
                         : tid == 4 ? lo32(wb) : tid == 5 ? hi32(wb) : (unsigned)w.row;
        __hip_atomic_store(&gran[bid * 8 + tid], ((unsigned long long)xtag << 32) | d, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
      for (int idx = tid; idx < G * 8; idx += NT) {   // one lane per granule of every workgroup's record
        if ((idx & 7) == 7) continue;
        unsigned long long g;
        unsigned spins = 0;
        while ((unsigned)((g = __hip_atomic_load(&gran[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != xtag) {
          LPX_BARRIER_SLEEP;
          if (++spins > P.spin_max) { sh_fail = 1; break; }   // 1: a workgroup's candidate record
        }
        sh_part[idx] = (unsigned)g;
      }
      if (P.fences & 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      lds_barrier();
      if (sh_fail) {
        if (book) { ctl->status = 7 /* LPX_DEVICE_ERROR */; ctl->reserved = sh_fail + 1000 * s; chain_publish(ctl, P.host_snap); }
        return;
      }
    }
    if (P.dbg && lead) P.dbg[s * LPX_C2_STRIDE + 3] = wall_clock64();

    // ------------------------------------------------------------------ phase B: the leaving row
    // every WAVE reduces the workgroups' candidates for itself (a lane per candidate): no meeting, and the winner's pivot
    // element and right-hand side come out of the winning lane by v_readlane
    RatioRow mine = rr_none();
    double mine_a = 0.0, mine_b = 0.0;
    for (int t = tid & 63; t < G; t += 64) {
      const unsigned* q = &sh_part[t * 8];
      const RatioRow o{from32(q[0], q[1]), (int)q[6], 0};
      if (o.ratio < mine.ratio || (o.ratio == mine.ratio && o.row < mine.row)) {
        mine = o;
        mine_a = from32(q[2], q[3]);
        mine_b = from32(q[4], q[5]);
      }
    }
    RatioRow w = rr_wave_min_all(mine);
    double win_p = 0.0, win_b = 0.0;
    if (w.row != INT_MAX) {
      const int wl = __ffsll((long long)__ballot(mine.row == w.row)) - 1;
      win_p = lane_f64(mine_a, wl);
      win_b = lane_f64(mine_b, wl);
    }
    if constexpr (MG) {
      // allreduce(min+loc) over the shards: this shard's winner goes into slot `dev` of every device's mailbox (the two
      // slots alternate with the decisions of the whole LOOP, see k_block_chain_t), then every wave reduces the n_dev
      // records of its own device's mailbox (lowest global row wins ties, LPState.java:292-303)
      const int mslot = (P.mail_slot0 + s) & 1;
      if (lead) {
        for (int d = 0; d < P.n_dev; ++d) {
          MgMail* rec = &P.mail_peer[d][mslot * kMaxDevices + P.dev];
          st_sys(&rec->ratio, w.ratio);
          st_sys(&rec->a, win_p);
          st_sys(&rec->bi, win_b);
          __hip_atomic_store(&rec->row, w.row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (P.fences & 1) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        for (int d = 0; d < P.n_dev; ++d)
          __hip_atomic_store(&P.mail_peer[d][mslot * kMaxDevices + P.dev].tag, xtag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      RatioRow theirs = rr_none();
      double theirs_a = 0.0, theirs_b = 0.0;
      if ((tid & 63) < P.n_dev) {
        const MgMail* rec = &P.mail_peer[P.dev][mslot * kMaxDevices + (tid & 63)];
        unsigned spins = 0;
        while (__hip_atomic_load(&rec->tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != xtag) {
          LPX_BARRIER_SLEEP;
          if (++spins > P.spin_max) { sh_fail = 2 + 16 * (tid & 63); break; }   // 2: a peer's candidate record
        }
        if (P.fences & 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        theirs.ratio = ld_sys(&rec->ratio);
        theirs.row = __hip_atomic_load(&rec->row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        theirs_a = ld_sys(&rec->a);
        theirs_b = ld_sys(&rec->bi);
      }
      w = rr_wave_min_all(theirs);
      win_p = 0.0; win_b = 0.0;
      if (w.row != INT_MAX) {
        const int wl = __ffsll((long long)__ballot(theirs.row == w.row)) - 1;
        win_p = lane_f64(theirs_a, wl);
        win_b = lane_f64(theirs_b, wl);
      }
      lds_barrier();   // (sh_fail of another wave)
      if (sh_fail) { if (book) { ctl->status = 7; ctl->reserved = sh_fail + 1000 * s; chain_publish(ctl, P.host_snap); } return; }
    }
    if (w.row == INT_MAX || !(w.ratio < kInf)) {  // getLeaving() == -1: unbounded
      if (book) {
        ctl->status = 1; ctl->do_update = 0; ctl->l = -1; ctl->ratio = w.ratio; P.up[s].do_update = 0;
        chain_publish(ctl, P.host_snap);
      }
      return;
    }
    if (max_pivots >= 0 && pivots >= max_pivots) {
      if (book) {
        ctl->status = 9 /* LPX_PIVOT_LIMIT */; ctl->do_update = 0; P.up[s].do_update = 0;
        chain_publish(ctl, P.host_snap);
      }
      return;
    }
    LPX_C2_STAMP(12)
    const int l = w.row;
    const double p = win_p, raw_b = win_b;
    if (p == 0.0) {  // ArithmeticException in the reference, LPState.java:139
      if (book) { ctl->status = 8; ctl->do_update = 0; P.up[s].do_update = 0; chain_publish(ctl, P.host_snap); }
      return;
    }
    // restart pivot of row l: the LAST pending pivot that left through row l made it its normalised pivot row; the row
    // restarts from the thread's copy of that row (every wave asks for itself, as in phase A)
    int rb;
    {
      const unsigned long long lo = __ballot(sh_l[tid & 63] == l);
      unsigned long long hi = 0;
      if constexpr (KB == 64) hi = __ballot(sh_l[64 + (tid & 63)] == l);
      rb = top_bit(lo, hi);
    }
    // On shards only the device that holds row l computes it; the others receive the normalised row (below).
    const bool owner = !MG || (l >= row0 && l < row0 + m);
    const int ll = l - row0;   // local index of the leaving row on its owner
    const int fo_b = rb < 0 ? 0 : (rb < KB ? rb + 1 : n_old);
    const int fn_b = rb >= KB ? rb - KB + 1 : 0;
    const int co0_b = fo_b >> 3, co1_b = fo_b < n_old ? (n_old + 7) >> 3 : co0_b;   // the live chunks, as in phase A
    const int cn0_b = fn_b >> 3, cn1_b = fn_b < s ? (s + 7) >> 3 : cn0_b;
    const int Lo_b = owner ? co1_b - co0_b : 0, L_b = owner ? Lo_b + (cn1_b - cn0_b) : 0;
    auto chunk_b = [&](int k) { return (k < Lo_b ? co0_b + k : KB / 8 + cn0_b + (k - Lo_b)) & (NC - 1); };
    // again everything in one round trip: col_u[l] of the pending pivots first, then the thread's column of row l
    double cs_mine = 0.0;
    if (valid_mine && owner) cs_mine = ld_agent((old_mine ? P.col_o : P.col) + (int64_t)r_mine * mp + ll);
    int32_t perm_e = 0, perm_l = 0;   // exchangeIndexes :311-320: asked for with the phase's loads (the keeper's own stores)
    if (book) { perm_e = P.perm[e]; perm_l = P.perm[n + l]; }
    const double bl = __ddiv_rn(raw_b, p);                                         // :146
    const double inv_p = __ddiv_rn(1.0, p);                                        // :139
    RatioRow cand = rr_none();  // (key, slot): key 0 = first slot (reference), -c = largest coefficient (Dantzig)
    double rec_c = 0.0, rec_pr = 0.0;   // workgroup 0: what its thread of slot j would put into the hand-off record
    const double* rowl = A + (int64_t)(owner ? ll : 0) * ld;
    if constexpr (MG) {
      if (!owner) {
        // wait until every workgroup of the owner has stored its columns of the normalised row into THIS device's replica
        // of the ring (one arrival word per owner workgroup, raised after its stores have drained)
        if (tid < G) {
          const unsigned long long* aw = &P.arrive_peer[P.dev][tid];
          unsigned spins = 0;
          while (__hip_atomic_load(aw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (unsigned long long)xtag) {
            LPX_BARRIER_SLEEP;
            if (++spins > P.spin_max) { sh_fail = 3 + 16 * tid; break; }   // 3: the owner's arrival word
          }
          if (P.fences & 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        }
        lds_barrier();
        if (sh_fail) { if (book) { ctl->status = 7; ctl->reserved = sh_fail + 1000 * s; chain_publish(ctl, P.host_snap); } return; }
      }
    }
    double x = 0.0, cj = 0.0;
    int rsb = -1;        // the column's own start index
    auto load_window_b = [&](int j, int w0) {
#pragma unroll
      for (int k = 0; k < kWin; ++k) {
        if (w0 + k < L_b) {
          const int c = chunk_b(w0 + k);
          const char* const base = (c < KB / 8 ? op_o : op_n) + (uint32_t)((c * 8) & (KB - 1)) * ld8;
#pragma unroll
          for (int q = 0; q < 8; ++q) cv[k * 8 + q] = *reinterpret_cast<const double*>(base + ((uint32_t)q * ld8 + (uint32_t)j * 8u));
        }
      }
    };
    auto load_col = [&](int j) {
      cj = ld_agent(&P.c[j]);  // this thread's own store (or the initial value)
      x = 0.0;
      rsb = P.own_rs_b[j];
      if (!owner) {   // (shards) the owner's value, from this device's replica of the ring
        x = (j < n && j != e) ? ld_sys(&P.prow[(int64_t)s * ld + j]) : 0.0;
      } else if (j < n) {
        const uint32_t j8 = (uint32_t)j * 8u;
        x = rb < 0 ? rowl[j]
                   : *reinterpret_cast<const double*>((rb < KB ? op_o : op_n) + ((uint32_t)(rb & (KB - 1)) * ld8 + j8));
        load_window_b(j, 0);
      }
    };
    int j = jfirst;
    if (j < (int)ld) load_col(j);
    // while the loads fly, as in phase A: 1 + the highest pending pivot that ENTERED at one of this wave's 64 slots
    int reach_b[2];
    {
      const int col_w = __builtin_amdgcn_readfirstlane(jfirst);   // (a wave's lanes own consecutive slots)
      unsigned long long h0[2] = {0, 0}, h1[2] = {0, 0};
#pragma unroll
      for (int h = 0; h < 2 * KB / 64; ++h) {
        const int eu = sh_e[h * 64 + (tid & 63)];
        h0[h] = __ballot((unsigned)(eu - col_w) < 64u);
        h1[h] = __ballot((unsigned)(eu - col_w - jstep) < 64u);
      }
      reach_b[0] = max(rb, top_bit(h0[0], h0[1])) + 1;
      reach_b[1] = max(rb, top_bit(h1[0], h1[1])) + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (as in phase A: also the free drain of this wave's earlier stores)
    if (P.dbg && lead) P.dbg[s * LPX_C2_STRIDE + 4] = wall_clock64();   // this wave's loads of phase B are here
    if (valid_mine) {
      sh_cs[tid] = cs_mine;
      sh_dv[tid] = -__ddiv_rn(cs_mine, sh_p[tid]);
    }
    lds_barrier();
    LPX_C2_STAMP(13)
    int pass_b = 0;
    while (j < (int)ld) {
      const int reach = pass_b == 0 ? reach_b[0] : pass_b == 1 ? reach_b[1] : 2 * KB;
      ++pass_b;
      if (j < n && owner) {
        // this lane's chain starts behind the later of the row's restart pivot and the LAST pending pivot that entered at
        // slot j; that pivot left -(col[l] / p) in row l's entry of its column (:157)
        const int st = max(rb, rsb) + 1;
        if (rsb > rb) x = sh_dv[rsb];
        double cs8[2][8];
        double x_mid = x;   // the row as the sweep of this block will read it (fix-up)
        auto ladder_b = [&](const int w0) {
          auto params = [&](int k) {
            if (k < kWin) {
              const double* const src = &sh_cs[chunk_b(w0 + k) * 8];
#pragma unroll
              for (int q = 0; q < 8; ++q) cs8[k & 1][q] = src[q];
            }
          };
          params(0);
#pragma unroll
          for (int k = 0; k < kWin; ++k) {
            if (w0 + k >= L_b) break;
            params(k + 1);
            __builtin_amdgcn_sched_barrier(0);
            if (w0 + k == Lo_b) x_mid = x;
            const int c8 = chunk_b(w0 + k) * 8;
            if (__builtin_expect(c8 >= reach, 1)) chain8(x, cs8[k & 1], &cv[k * 8]);
            else chain8_from(x, cs8[k & 1], &cv[k * 8], st - c8);
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        ladder_b(0);
        if constexpr (NC > kWin) {
          if (L_b > kWin) {
            load_window_b(j, kWin);
            ladder_b(kWin);
          }
        }
        if (Lo_b >= L_b) x_mid = x;
        P.row0[(int64_t)s * ld + j] = x_mid;
      }
      LPX_C2_STAMP(14)
      double cn, pr;
      if (j == e) {
        pr = inv_p;
        cn = -__ddiv_rn(pc, p);                                                    // :172
      } else {
        pr = owner ? __ddiv_rn(x, p) : x;                                          // :144 (shards: the owner's value)
        cn = submul(cj, pc, pr);                                                   // :177
      }
      if constexpr (MG) {
        if (owner) {   // broadcast: the value goes into every other device's replica of the ring
          for (int d = 0; d < P.n_dev; ++d)
            if (d != P.dev) st_sys(&P.prow_peer[d][(int64_t)s * ld + j], pr);
        }
      }
      st_agent(&P.prow[(int64_t)s * ld + j], pr);
      P.own_prow[(int64_t)s * ld + j] = pr;
      if (j == e) P.own_rs_b[j] = KB + s;   // later chains of this slot start behind pivot s
      st_agent(&P.c[j], cn);
      if (j < n && cn > kEps) {
        const RatioRow k2{P.dantzig ? -cn : 0.0, j, 0};
        cand = rr_min(cand, k2);
      }
      rec_c = cn;
      rec_pr = pr;
      j += jstep;
      __builtin_amdgcn_sched_barrier(0);
      if (j < (int)ld) load_col(j);
    }
    LPX_C2_STAMP(15)
    if (window && bid == 0) {
      // Workgroup 0 has finished slots 0..255.  Under the first-positive rule the next entering slot is the lowest one
      // with c > eps: if there is one among them it is the answer, and the thread that owns it publishes {slot, c[slot],
      // prow_s[slot]} — all the next phase A needs of this decision — as five tagged granules, no drain.  No candidate
      // here: "none", and everybody takes the grid barrier below.
      // Slots ascend with the lane and with the wave: the first lane of a wave's ballot, the first wave that has one — one
      // meeting, no reduction.  (sh_rec[s & 1] was read by everybody several meetings ago.)
      MinRec* const ho = sh_rec[s & 1];
      const unsigned long long cm = __ballot(cand.row != INT_MAX);
      if ((tid & 63) == (cm ? __ffsll((long long)cm) - 1 : 0)) ho[tid >> 6] = MinRec{rec_c, rec_pr, 0.0, cm ? cand.row : -1, 0};
      lds_barrier();
      if (tid < 5) {
        int slot = -1;
        double hc = 0.0, hp = 0.0;
#pragma unroll
        for (int k = NT / 64 - 1; k >= 0; --k) {
          const MinRec o = ho[k];
          if (o.row >= 0) { slot = o.row; hc = o.ratio; hp = o.a; }
        }
        const unsigned d = tid == 0 ? (unsigned)((slot < 0 ? -2 : slot) + 2) : slot < 0 ? 0u : tid == 1 ? lo32(hc) : tid == 2 ? hi32(hc) : tid == 3 ? lo32(hp) : hi32(hp);
        __hip_atomic_store(&hand[tid], ((unsigned long long)xtag << 32) | d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if constexpr (MG) {
      if (owner && P.n_dev > 1) {   // this workgroup's columns are on their way to every peer: drain, meet, signal
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        if (tid == 0) {
          if (P.fences & 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          for (int d = 0; d < P.n_dev; ++d)
            if (d != P.dev)
              __hip_atomic_store(&P.arrive_peer[d][bid], (unsigned long long)xtag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    if (P.dbg && lead) P.dbg[s * LPX_C2_STRIDE + 5] = wall_clock64();   // workgroup 0: the hand-off record is on its way
    // the row owners add pivot s to b (kept with every decided pivot applied) — after the columns, so that workgroup 0
    // publishes the entering slot first
    // (the owner of row l also notes that later chains of this row start behind pivot s)
    if (gid < m) P.own_b[gid] = (row0 + gid == l) ? bl : submul(b_first, a_first, bl);      // :146 / :164
    if (gid < m && row0 + gid == l) P.own_rs_a[gid] = KB + s;
    for (int i2 = gid + T; i2 < m; i2 += T) {
      const double colv = P.own_col[(int64_t)s * mp + i2];
      const double bcur = use_b ? b[i2] : P.own_b[i2];
      P.own_b[i2] = (row0 + i2 == l) ? bl : submul(bcur, colv, bl);
      if (row0 + i2 == l) P.own_rs_a[i2] = KB + s;
    }
    if (tid == 0) { sh_e[KB + s] = e; sh_l[KB + s] = l; sh_p[KB + s] = p; sh_bl[KB + s] = bl; }
    if (book) {
      v = addmul(v, bl, pc);                                                       // :171
      P.perm[e] = perm_l;                                                          // exchangeIndexes :311-320
      P.perm[n + l] = perm_e;
      if (track >= 0) {                                                            // LPSolver.java:151-155
        if (e == track) track = l + n;
        else if (l + n == track) track = e;
      }
      LpxCtl& up = P.up[s];
      up.p = p; up.bl = bl; up.e_cur = e; up.l = l; up.e_next = -1; up.parity = 0; up.do_update = 1;
    }
    if (P.dbg && lead) P.dbg[s * LPX_C2_STRIDE + 6] = wall_clock64();
    int e_next = -2;
    have_rec = false;
    if (window) {  // five lanes per workgroup wait for workgroup 0's record (bounded, like the barrier)
      if (tid < 5) {
        unsigned long long rec = 0;
        unsigned spins = 0;
        for (;;) {
          rec = __hip_atomic_load(&hand[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((unsigned)(rec >> 32) == xtag) break;
          LPX_BARRIER_SLEEP;
          if (++spins > P.spin_max) { sh_fail = 4; break; }   // 4: workgroup 0's hand-off record
        }
        sh_hand[tid] = (unsigned)rec;
      }
      lds_barrier();
      if (sh_fail) { if (book) { ctl->status = 7; ctl->reserved = sh_fail + 1000 * s; chain_publish(ctl, P.host_snap); } return; }
      e_next = (int)sh_hand[0] - 2;
      have_rec = e_next >= 0;
    }
    if (e_next < 0) {  // no hand-off (Dantzig, narrow tableau) or no candidate in its window: the full exchange
      const RatioRow w2 = rr_block_min2(cand, sh_rr);
      if (tid == 0) {
        st_agent(&P.partB[bid].ratio, w2.ratio);
        st_agent(&P.partB[bid].row, w2.row);
      }
      target += (unsigned)G;
      if (!grid_barrier(P.bar, target, &sh_fail, P.fences, P.spin_max)) {
        if (book) { ctl->status = 7; chain_publish(ctl, P.host_snap); }
        return;
      }
      RatioRow m2 = rr_none();
      if (tid < G) { m2.ratio = ld_agent(&P.partB[tid].ratio); m2.row = ld_agent(&P.partB[tid].row); }
      const RatioRow w3 = rr_block_min2(m2, sh_rr);
      e_next = (w3.row == INT_MAX) ? -1 : w3.row;
    }
    if (P.dbg && lead) P.dbg[s * LPX_C2_STRIDE + 7] = wall_clock64();
    pivots += 1;
    parity ^= 1;
    if (book) {
      ctl->v = v; ctl->p = p; ctl->bl = bl; ctl->pc = pc; ctl->ratio = w.ratio;
      ctl->e_cur = e; ctl->l = l; ctl->e_next = e_next; ctl->parity = parity; ctl->pivots = pivots;
      ctl->track = track; ctl->do_update = 1;
      if (e_next < 0) {
        ctl->status = 0 /* LPX_OPTIMAL once the sweep has applied this pivot */;
        if (s + 1 < nb) P.up[s + 1].do_update = 0;  // the block ends here: the sweep counts leading valid slots
      }
    }
    if (e_next < 0 || s + 1 == nb) {
      if (book) chain_publish(ctl, P.host_snap);
      return;
    }
    e = e_next;
    lds_barrier();   // sh_e[KB + s] .. and sh_hand are read by the next decision; sh_mask / sh_win are rewritten
  }
}
#undef LPX_C2_STAMP
#undef LPX_C2_STRIDE
#undef LPX_CHAIN_STEP_A
#undef LPX_CHAIN_STEP_B
#undef LPX_CHAIN_CHUNK_A
#undef LPX_CHAIN_CHUNK_B
#undef LPX_CHAIN_LIVE

// The sweep is a pure streaming kernel: x -= col_s[i] * prow_s[j] for the valid pending pivots s, in order, for
// EVERY entry — also at the few positions where a pivot does something else (its own row becomes the normalised
// row, its entering column becomes -(col/p)).  Those positions (K rows and K columns) are recomputed afterwards
// by k_block_fixup from the stale values that k_peek_multi / k_pack_multi saved, with the full case analysis;
// no entry depends on another entry, so the garbage written there in between is never read.
// Workgroup = rows_per_tile (<= 64) rows x 512 columns, thread = one 16-byte double2 per row; the thread's
// slices of the K pivot rows stay in registers (2K doubles), the K x rows multipliers of the tile are staged
// once in LDS (coalesced load, broadcast reads); rows go RB at a time to keep RB loads in flight per thread.
// Cost: two fp64 VALU operations per entry per pivot (the product and the difference must stay two roundings,
// so no FMA): measured ~8 cycles per wave-instruction, which makes the sweep VALU-bound from K ~ 16 on
// (cfg4: 1.36 ms for one pass, 1.55 ms at K = 16, 2.4 ms at K = 32; a one-double-per-thread variant with
// twice the occupancy was not faster).
constexpr int kSweepMaxRows = 128;
#ifndef LPX_SWEEP_NBUF
#define LPX_SWEEP_NBUF 2   // register buffers of the sweep: batches in flight = NBUF - 1 (3: measured 4 % slower)
#endif

// Number of valid leading pending pivots (slots >= kmax were not decided this block): one parallel look at the
// ring by the first wave instead of a chain of dependent loads.  Result valid in every thread.
__device__ __forceinline__ int ring_count(const LpxCtl* __restrict__ ring, int kcap, int kmax, int* sh_np) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const bool ok = lane < kcap && lane < kmax && ring[lane].do_update != 0;
    const unsigned long long mask = __ballot(ok);
    if (lane == 0) *sh_np = (~mask == 0ull) ? 64 : (__ffsll((long long)~mask) - 1);
  }
  __syncthreads();
  return *sh_np;
}

// The K-fold update of one batch of RB rows x one double2 held in registers; the multipliers come from LDS
// (one 16-byte broadcast read = two rows).  PIPE: the steady-state form — straight-line code with the LDS reads of
// step s+D issued before the arithmetic of step s (the compiler, minimising registers, otherwise puts every read
// right in front of its use and the wave eats the full LDS latency 2K times per batch — measured: the fp64 VALU then
// idles half of the time).  In a partly filled block (the tail of a pivot budget, the end of the LP) the steps
// s >= np are skipped by a wave-uniform scalar branch per step; their read-ahead still runs (sh_col is filled for all
// K steps).
enum SweepMode { kSweepSimple = 0, kSweepAll = 1, kSweepGuarded = 2 };
template <int K, int RB, int MODE>
__device__ __forceinline__ void sweep_apply(d2 (&x)[RB], const d2 (&pr)[K], const double (*sh_col)[kSweepMaxRows],
                                            int np, int r0) {
  if constexpr (MODE != kSweepSimple) {
#ifndef LPX_SWEEP_D
#define LPX_SWEEP_D 2
#endif
    constexpr int D = LPX_SWEEP_D;  // read-ahead distance in steps (1, 3 and 4 measured the same or worse)
    d2 cc[D + 1][RB / 2];
#pragma unroll
    for (int s = 0; s < D && s < K; ++s)
#pragma unroll
      for (int r = 0; r < RB; r += 2)
        cc[s][r / 2] = *reinterpret_cast<const d2*>(&sh_col[s][(r0 + r) & (kSweepMaxRows - 1)]);
#pragma unroll
    for (int s = 0; s < K; ++s) {
      if (s + D < K) {
#pragma unroll
        for (int r = 0; r < RB; r += 2)
          cc[(s + D) % (D + 1)][r / 2] =
              *reinterpret_cast<const d2*>(&sh_col[s + D][(r0 + r) & (kSweepMaxRows - 1)]);
      }
      __builtin_amdgcn_sched_barrier(0);
      // kSweepAll (np == K, the steady state): no branch at all — 32 scalar branches per batch cost 17 % (cfg4 alone
      // in place: 1.64 ms vs 1.91 ms); kSweepGuarded: a wave-uniform branch per step for partly filled blocks
      if (MODE == kSweepAll || s < np) {
#pragma unroll
        for (int r = 0; r < RB; r += 2) {
          const d2 c2 = cc[s % (D + 1)][r / 2];
          x[r].x = submul(x[r].x, c2.x, pr[s].x);                      // LPState.java:162
          x[r].y = submul(x[r].y, c2.x, pr[s].y);
          x[r + 1].x = submul(x[r + 1].x, c2.y, pr[s].x);
          x[r + 1].y = submul(x[r + 1].y, c2.y, pr[s].y);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll
    for (int s = 0; s < K; ++s) {
      if (s < np) {  // wave-uniform
#pragma unroll
        for (int r = 0; r < RB; r += 2) {
          const d2 c2 = *reinterpret_cast<const d2*>(&sh_col[s][(r0 + r) & (kSweepMaxRows - 1)]);  // LDS broadcast
          x[r].x = submul(x[r].x, c2.x, pr[s].x);
          x[r].y = submul(x[r].y, c2.x, pr[s].y);
          x[r + 1].x = submul(x[r + 1].x, c2.y, pr[s].x);
          x[r + 1].y = submul(x[r + 1].y, c2.y, pr[s].y);
        }
      }
    }
  }
}

// The sweep for K <= 16: one workgroup per tile of rows_per_tile (16) rows x 512 columns.  With at most 16 pivot-row
// slices per thread the kernel keeps three waves per SIMD and is HBM-bound (5.6 TB/s); the long-run form below, built
// for K = 32, measured 14-20 % slower here (cfg4, K = 16, same box: 1.72-1.82 ms vs 1.51 ms) and is not used.
template <int K, bool NT, bool OOP>
__global__ __launch_bounds__(256) void k_update_tiles(double* __restrict__ A, const double* __restrict__ Asrc,
                                                      int64_t ld, int m_local,
                                                      const double* __restrict__ prow_ring,
                                                      const double* __restrict__ col_ring, int64_t mp,
                                                      const LpxCtl* __restrict__ ring, int kmax,
                                                      int rows_per_tile, int nstrips, unsigned* census) {
  __shared__ __attribute__((aligned(16))) double sh_col[K][kSweepMaxRows];
  __shared__ int sh_np;
  // rows per batch (register budget: 2K doubles of pivot rows); K = 32 with 8 rows measured 13 % slower (r02)
  constexpr int RB = (K <= 8) ? 8 : 4;
  const int strip = blockIdx.x % nstrips;
  const int tile = blockIdx.x / nstrips;
  const int cj = strip * 512 + 2 * threadIdx.x;
  const bool act = cj < (int)ld;
  const int r_begin = tile * rows_per_tile;
  const int nrows = min(m_local, r_begin + rows_per_tile) - r_begin;
  // uniform tile base (SGPRs) + 32-bit per-lane byte offset: one VGPR per address
  char* const tile_base = reinterpret_cast<char*>(A + (int64_t)r_begin * ld + strip * 512);
  const char* const src_base =
      OOP ? reinterpret_cast<const char*>(Asrc + (int64_t)r_begin * ld + strip * 512) : tile_base;
  const uint32_t row_bytes = (uint32_t)ld * 8u;  // the launcher checks rows_per_tile * ld * 8 < 2^32
  const uint32_t off0 = threadIdx.x * 16u;

  // Prologue: everything the workgroup needs is requested at once — the ring's flags, the multipliers, the
  // thread's slices of the K pivot rows and the first batch of rows — ONE memory round trip, not four.
  bool ok = false;
  if (threadIdx.x < 64) ok = (int)threadIdx.x < K && (int)threadIdx.x < kmax && ring[threadIdx.x].do_update != 0;
  for (int idx = threadIdx.x; idx < K * kSweepMaxRows; idx += blockDim.x) {
    const int sidx = idx / kSweepMaxRows, r = idx % kSweepMaxRows;
    sh_col[sidx][r] = (r < nrows) ? col_ring[(int64_t)sidx * mp + r_begin + r] : 0.0;  // slots >= np: never used
  }
  d2 pr[K];
#pragma unroll
  for (int s = 0; s < K; ++s)
    pr[s] = act ? *reinterpret_cast<const d2*>(prow_ring + (int64_t)s * ld + cj) : d2{0.0, 0.0};
  // Full strips (all but possibly the last one of a row) with at least one full batch take the fast path below.
  const bool fast_geom = (strip + 1) * 512 <= (int)ld && nrows >= RB;
  // NB register buffers of RB rows each: while one batch is computed, the next NB-1 are in flight (the kernel's time
  // was T_hbm + ~0.55 T_valu with one batch ahead: too few bytes in flight to keep HBM busy during the fp64 work)
  constexpr int NB = (K >= 32) ? LPX_SWEEP_NBUF : 2;  // K = 16: a third buffer would cost the third wave per SIMD
  const int nfull_geom = fast_geom ? nrows / RB : 0;
  d2 xb[NB][RB];
#pragma unroll
  for (int u = 0; u + 1 < NB; ++u) {
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      xb[u][r] = d2{0.0, 0.0};
      if (u < nfull_geom) {  // uniform
        const d2* q = reinterpret_cast<const d2*>(src_base + (off0 + (uint32_t)(u * RB + r) * row_bytes));
        xb[u][r] = NT ? __builtin_nontemporal_load(q) : *q;
      }
    }
  }
  if (threadIdx.x < 64) {
    const unsigned long long mask = __ballot(ok);
    if (threadIdx.x == 0) sh_np = (~mask == 0ull) ? 64 : (__ffsll((long long)~mask) - 1);
  }
  __syncthreads();  // sh_col and sh_np complete
  const int np = sh_np;
  if (np == 0 && !OOP) return;  // out of place: the tableau still has to be carried over

  if (census && blockIdx.x % 509u == 0 && threadIdx.x == 0) atomicOr(census, 1u << xcc_id());  // placement sample

  // Fast path, the steady state: straight-line batches without any per-lane guard, software-pipelined — the next
  // batch's loads are in flight while this one runs its 2 np fp64 operations per entry.
  const int full = np > 0 ? nfull_geom : 0;
  auto stream_batches = [&](auto mode) {
    constexpr int MODE = decltype(mode)::value;
#pragma unroll 1
    for (int bt = 0; bt < full; bt += NB) {
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        if (bt + u < full) {  // uniform
          const int r0 = (bt + u) * RB;
          if (bt + u + NB - 1 < full) {  // request batch bt+u+NB-1 into the buffer that was stored last
#pragma unroll
            for (int r = 0; r < RB; ++r) {
              const d2* q = reinterpret_cast<const d2*>(src_base + (off0 + (uint32_t)(r0 + (NB - 1) * RB + r) * row_bytes));
              xb[(u + NB - 1) % NB][r] = NT ? __builtin_nontemporal_load(q) : *q;
            }
          }
          sweep_apply<K, RB, MODE>(xb[u], pr, sh_col, np, r0);
#pragma unroll
          for (int r = 0; r < RB; ++r) {
            d2* q = reinterpret_cast<d2*>(tile_base + (off0 + (uint32_t)(r0 + r) * row_bytes));
            if (NT) __builtin_nontemporal_store(xb[u][r], q); else *q = xb[u][r];
          }
        }
      }
    }
  };
  if (full > 0) {
    if (np == K) stream_batches(std::integral_constant<int, kSweepAll>{});
    else stream_batches(std::integral_constant<int, kSweepGuarded>{});
  }
  // the rest (partial blocks, rows beyond the last full batch, the partial last strip): guarded
  for (int r0 = full * RB; r0 < nrows; r0 += RB) {
    d2 y[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      y[r] = d2{0.0, 0.0};
      if (r0 + r < nrows && act) {
        const d2* q = reinterpret_cast<const d2*>(src_base + (off0 + (uint32_t)(r0 + r) * row_bytes));
        y[r] = NT ? __builtin_nontemporal_load(q) : *q;
      }
    }
    sweep_apply<K, RB, kSweepSimple>(y, pr, sh_col, np, r0);
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      if (r0 + r < nrows && act) {
        d2* q = reinterpret_cast<d2*>(tile_base + (off0 + (uint32_t)(r0 + r) * row_bytes));
        if (NT) __builtin_nontemporal_store(y[r], q); else *q = y[r];
      }
    }
  }
}

// OOP: read the tableau from Asrc, write the updated one to A (same traffic; lets the NEXT block's decisions read
// the un-updated tableau while this sweep streams — see blocked_loop_overlapped in lpx_engine.cpp).
//
// Work split (round 2): a workgroup owns one 512-column strip and a LONG run of rows (rows_per_wg, several hundred to
// a few thousand) and walks down it in chunks of 64 rows.  Its 2K doubles of pivot-row slices per thread are loaded
// ONCE; per chunk only the K x 64 multipliers (16 KiB at K = 32) are staged — into the other half of the LDS array
// while the current chunk is computed, so a chunk boundary costs one workgroup barrier.  In-kernel timestamps of the
// first version (one workgroup per 64-row tile: profiles/r02_sweep_stamps_before.txt) showed why: inside the batch
// loop a wave spent 0.1 us of 2.2 us per batch waiting for its loads and the SIMD was ~94 % busy with the two waves'
// arithmetic — but a tile's 16 batches took 35 us of the 59 us a workgroup slot was held: the rest was the prologue,
// 128 KiB of pivot-row slices + 16 KiB of multipliers fetched from L2 / Infinity Cache for every 256 KiB of tableau.
constexpr int kSweepChunk = 64;   // rows per chunk; the LDS array holds two chunks' multipliers
#ifndef LPX_SWEEP_LB
#define LPX_SWEEP_LB(K) 2   // waves per SIMD the register allocation is held to (two workgroups per CU)
#endif
static_assert(kSweepMaxRows == 2 * kSweepChunk, "sh_col is double-buffered by chunk");

template <int K, bool NT, bool OOP>
__global__ __launch_bounds__(256, LPX_SWEEP_LB(K)) void k_update_multi(double* __restrict__ A, const double* __restrict__ Asrc,
                                                      int64_t ld, int m_local,
                                                      const double* __restrict__ prow_ring,
                                                      const double* __restrict__ col_ring, int64_t mp,
                                                      const LpxCtl* __restrict__ ring, int kmax,
                                                      int rows_per_wg, int nstrips, unsigned* census, int complement,
                                                      int slot0) {
  // slot0: this pass applies the pending pivots slot0 .. slot0 + K - 1 of the block (a block of more than 32 goes in
  // two passes where the one-pass kernel does not apply); the rings are passed at slot 0, kmax counts from slot 0.
  // complement (0, or 1 + the smallest number of valid pivots at which the steady-state kernel launched in front takes
  // the full strips: 1 for k_sweep32_steady — it takes them always —, 65 for k_sweep64_pipe): only what that kernel
  // leaves is done here.
  __shared__ __attribute__((aligned(16))) double sh_col[K][kSweepMaxRows];
  __shared__ int sh_np;
  constexpr int CH = kSweepChunk;
  if (complement) {
    const int np0 = ring_count(ring, kBlockMax, kmax, &sh_np);
    if (np0 >= complement - 1 && (blockIdx.x % nstrips + 1) * 512 <= (int)ld) return;
    __syncthreads();
  }
  prow_ring += (int64_t)slot0 * ld;
  col_ring += (int64_t)slot0 * mp;
  // rows per batch (register budget: 2K doubles of pivot rows); K = 32 with 8 rows measured 13 % slower (r02)
#ifndef LPX_STRIP_RB
#define LPX_STRIP_RB 4
#endif
#ifndef LPX_STRIP_NB
#define LPX_STRIP_NB 2
#endif
  constexpr int RB = (K <= 8) ? 8 : LPX_STRIP_RB;
  constexpr int NB = LPX_STRIP_NB;                   // register buffers: one batch computed, NB - 1 in flight
  static_assert((CH / RB) % NB == 0, "a chunk holds a whole number of buffer rotations");
  constexpr int PF = (K * CH + 255) / 256;           // multipliers of the next chunk held per thread meanwhile
  const int strip = blockIdx.x % nstrips;
  const int grp = blockIdx.x / nstrips;
  const int cj = strip * 512 + 2 * threadIdx.x;
  const bool act = cj < (int)ld;
  const int r_begin = grp * rows_per_wg;
  const int nrows = min(m_local, r_begin + rows_per_wg) - r_begin;
  if (nrows <= 0) return;
  // uniform run base (SGPRs) + 32-bit per-lane byte offset: one VGPR per address
  char* const tile_base = reinterpret_cast<char*>(A + (int64_t)r_begin * ld + strip * 512);
  const char* const src_base =
      OOP ? reinterpret_cast<const char*>(Asrc + (int64_t)r_begin * ld + strip * 512) : tile_base;
  const uint32_t row_bytes = (uint32_t)ld * 8u;  // the launcher checks rows_per_wg * ld * 8 < 2^32
  const uint32_t off0 = threadIdx.x * 16u;

  // Prologue, once per workgroup: the ring's flags, the first chunk's multipliers, the thread's slices of the K pivot
  // rows and the first batch of rows are requested together — ONE memory round trip, not four.
  bool ok = false;
  if (threadIdx.x < 64) ok = (int)threadIdx.x < kmax && ring[threadIdx.x].do_update != 0;   // kmax <= 64 slots
  for (int idx = threadIdx.x; idx < K * CH; idx += 256) {
    const int sidx = idx / CH, r = idx % CH;
    sh_col[sidx][r] = (r < nrows) ? col_ring[(int64_t)sidx * mp + r_begin + r] : 0.0;  // slots >= np: never used
  }
  d2 pr[K];
#pragma unroll
  for (int s = 0; s < K; ++s)
    pr[s] = act ? *reinterpret_cast<const d2*>(prow_ring + (int64_t)s * ld + cj) : d2{0.0, 0.0};
  // Full strips (all but possibly the last one of a row) take the pipelined path for their full batches.
  const bool fast_geom = (strip + 1) * 512 <= (int)ld;
  const int nfull_geom = fast_geom ? nrows / RB : 0;   // full batches of the whole run (chunks hold CH / RB each)
  d2 xb[NB][RB];
#pragma unroll
  for (int u = 0; u + 1 < NB; ++u) {
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      xb[u][r] = d2{0.0, 0.0};
      if (u < nfull_geom) {  // uniform
        const d2* q = reinterpret_cast<const d2*>(src_base + (off0 + (uint32_t)(u * RB + r) * row_bytes));
        xb[u][r] = NT ? __builtin_nontemporal_load(q) : *q;
      }
    }
  }
  if (threadIdx.x < 64) {
    const unsigned long long mask = __ballot(ok);
    if (threadIdx.x == 0) {
      const int np_block = (~mask == 0ull) ? 64 : (__ffsll((long long)~mask) - 1);   // leading valid slots of the block
      sh_np = max(0, min(K, np_block - slot0));
    }
  }
  __syncthreads();  // chunk 0's multipliers and sh_np complete
  const int np = sh_np;
  if (np == 0 && !OOP) return;  // out of place: the tableau still has to be carried over

  if (census && blockIdx.x % 61u == 0 && threadIdx.x == 0) atomicOr(census, 1u << xcc_id());  // placement sample
#ifdef LPX_SWEEP_STAMPS   // diagnostic build only: lifetime of sampled workgroups
  long long wg_t0 = wall_clock64();
#endif

  const int full = np > 0 ? nfull_geom : 0;   // batches of the pipelined path, numbered over the whole run
  const int nchunks = (nrows + CH - 1) / CH;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int half = (ch & 1) * CH;                      // this chunk's half of sh_col
    const int c_rows = min(CH, nrows - ch * CH);
    // the next chunk's multipliers: requested now, parked in registers, stored into the other half at the end
    double colpf[PF];
    const bool more = ch + 1 < nchunks;
    if (more) {
      const int n_next = min(CH, nrows - (ch + 1) * CH);
#pragma unroll
      for (int k = 0; k < PF; ++k) {
        const int idx = threadIdx.x + k * 256;
        const int sidx = idx / CH, r = idx % CH;
        colpf[k] = (idx < K * CH && r < n_next) ? col_ring[(int64_t)sidx * mp + r_begin + (ch + 1) * CH + r] : 0.0;
      }
    }
    // pipelined path: straight-line batches without any per-lane guard — the next batch's loads (possibly the next
    // chunk's first rows: the run is contiguous) are in flight while this one runs its 2 np fp64 operations per entry
    const int b_lo = ch * (CH / RB), b_hi = min(full, b_lo + CH / RB);
    auto stream_batches = [&](auto mode) {
      constexpr int MODE = decltype(mode)::value;
#pragma unroll 1
      for (int bt = b_lo; bt < b_hi; bt += NB) {   // b_lo is even: buffer u holds batch bt + u
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          if (bt + u < b_hi) {  // uniform
            const int r0 = (bt + u) * RB;
            if (bt + u + NB - 1 < full) {  // request batch bt+u+NB-1 into the buffer that was stored last
#pragma unroll
              for (int r = 0; r < RB; ++r) {
                const d2* q = reinterpret_cast<const d2*>(src_base + (off0 + (uint32_t)(r0 + (NB - 1) * RB + r) * row_bytes));
                xb[(u + NB - 1) % NB][r] = NT ? __builtin_nontemporal_load(q) : *q;
              }
            }
            sweep_apply<K, RB, MODE>(xb[u], pr, sh_col, np, half + (r0 & (CH - 1)));
#pragma unroll
            for (int r = 0; r < RB; ++r) {
              d2* q = reinterpret_cast<d2*>(tile_base + (off0 + (uint32_t)(r0 + r) * row_bytes));
              if (NT) __builtin_nontemporal_store(xb[u][r], q); else *q = xb[u][r];
            }
          }
        }
      }
    };
    if (b_hi > b_lo) {
      if (np == K) stream_batches(std::integral_constant<int, kSweepAll>{});
      else stream_batches(std::integral_constant<int, kSweepGuarded>{});
    }
    // the rest of the chunk (an empty block carried over out of place, rows beyond the last full batch, the partial
    // last strip): guarded
    const int done_rows = (b_hi > b_lo ? b_hi - b_lo : 0) * RB;
    for (int rr = done_rows; rr < c_rows; rr += RB) {
      const int r0 = ch * CH + rr;
      d2 y[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        y[r] = d2{0.0, 0.0};
        if (rr + r < c_rows && act) {
          const d2* q = reinterpret_cast<const d2*>(src_base + (off0 + (uint32_t)(r0 + r) * row_bytes));
          y[r] = NT ? __builtin_nontemporal_load(q) : *q;
        }
      }
      sweep_apply<K, RB, kSweepSimple>(y, pr, sh_col, np, half + rr);
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        if (rr + r < c_rows && act) {
          d2* q = reinterpret_cast<d2*>(tile_base + (off0 + (uint32_t)(r0 + r) * row_bytes));
          if (NT) __builtin_nontemporal_store(y[r], q); else *q = y[r];
        }
      }
    }
    if (more) {   // uniform: publish the next chunk's multipliers; everyone is done with the half they go into
#pragma unroll
      for (int k = 0; k < PF; ++k) {
        const int idx = threadIdx.x + k * 256;
        if (idx < K * CH) sh_col[idx / CH][(CH - half) + idx % CH] = colpf[k];
      }
      __syncthreads();
    }
  }
#ifdef LPX_SWEEP_STAMPS
  if (census && blockIdx.x % 16u == 0 && blockIdx.x / 16u < 140 && threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long* o = reinterpret_cast<long long*>(census + 8) + (blockIdx.x / 16u) * 2;
    o[0] = wg_t0; o[1] = wall_clock64();
  }
#endif
}

#ifdef LPX_WITH_VARIANTS   // superseded / experiment kernels: csrc/variants/, built by `make variants` only
#define LPX_VARIANT_PART 1
#include "variants/lpx_variants.inc"
#undef LPX_VARIANT_PART
#endif

// ---- the steady state of the sweep with the tableau staged through LDS by LDS-DMA (round 3) -------------------------
// Same work split and arithmetic as k_sweep32_steady; what changes is where a batch waits for the fp64 pipe.  There a
// wave parks its in-flight batches in registers (3 x 16 VGPRs on top of the 128 that hold the pivot-row slices: two
// batches in flight is all the register file allows, and two batches = ~4 us of a SIMD's arithmetic is less than the
// ~5 us the memory system takes under this load).  Here a wave issues global_load_lds_dwordx4 (1 KiB per instruction,
// lane t's 16 bytes land at slot + 16 t) into a ring of NS slots of its OWN in LDS and reads its batch back with one
// ds_read_b128 per row when the batch's turn comes: NS batches in flight per wave, no register parked, and the
// destination of a hand-issued load is LDS, not a VGPR — the compiler has nothing to copy, spill or reuse before the
// data has landed (the hazard class of k_sweep32_steady's register loads).  What stays hand-counted is
// vmcnt: the DMA is invisible to the compiler's own bookkeeping, so every read-back sits behind an asm wait with a
// memory clobber.  vmcnt counts this wave's loads, stores and LDS-DMAs in issue order, and "at most N outstanding"
// completes every operation that has at least N younger ones: operations the compiler adds only lengthen a wait; a
// wait is too short only if FEWER operations follow than assumed, which is what the tail counts below are for.
//
// LDS (80 KiB, two workgroups per CU): multipliers [2 chunks][32 pivots][32 rows] = 16 KiB, filled by LDS-DMA too
// (one piece = 4 pivots x 32 rows; each wave brings two pieces of the NEXT chunk right after the chunk barrier), then
// 4 waves x NS slots x 4 KiB.  No slot is shared between waves: the only workgroup barrier is the one per chunk that
// publishes the next chunk's multipliers.
constexpr int kDmaK = 32, kDmaRB = 4, kDmaCH = 32;
#ifndef LPX_DMA_NS
#define LPX_DMA_NS 4
#endif
#ifndef LPX_DMA_DIAG
#define LPX_DMA_DIAG 0   // 1, 2: diagnostic builds of scripts/micro/sweep_dma.hip only (memory pass alone / arithmetic alone)
#endif
constexpr int kDmaNS = LPX_DMA_NS;
constexpr int kDmaMultBytes = 2 * kDmaK * kDmaCH * 8;
constexpr int kDmaSlotBytes = kDmaRB * 64 * 16;
constexpr int kDmaLdsBytes = kDmaMultBytes + 4 * kDmaNS * kDmaSlotBytes;
static_assert(2 * kDmaLdsBytes <= 160 * 1024, "two workgroups per CU");

__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)(p);
}
template <int N>
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// sweep_apply's steady-state form over a linear multiplier image: element (s, row) at mrow[s * STRIDE + row]
template <int K, int RB, int STRIDE>
__device__ __forceinline__ void sweep_apply_lin(d2 (&x)[RB], const d2 (&pr)[K], const double* mrow) {
  constexpr int D = 2;
  d2 cc[D + 1][RB / 2];
#pragma unroll
  for (int s = 0; s < D && s < K; ++s)
#pragma unroll
    for (int r = 0; r < RB; r += 2) cc[s][r / 2] = *reinterpret_cast<const d2*>(mrow + s * STRIDE + r);
#pragma unroll
  for (int s = 0; s < K; ++s) {
    if (s + D < K) {
#pragma unroll
      for (int r = 0; r < RB; r += 2)
        cc[(s + D) % (D + 1)][r / 2] = *reinterpret_cast<const d2*>(mrow + (s + D) * STRIDE + r);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < RB; r += 2) {
      const d2 c2 = cc[s % (D + 1)][r / 2];
      x[r].x = submul(x[r].x, c2.x, pr[s].x);                      // LPState.java:162
      x[r].y = submul(x[r].y, c2.x, pr[s].y);
      x[r + 1].x = submul(x[r + 1].x, c2.y, pr[s].x);
      x[r + 1].y = submul(x[r + 1].y, c2.y, pr[s].y);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

#ifdef LPX_WITH_VARIANTS
#define LPX_VARIANT_PART 2
#include "variants/lpx_variants.inc"
#undef LPX_VARIANT_PART
#endif

// ---- the steady state of the sweep, every wave on its own, batches PULLED in address order (round 3) ----------------
// What bounds k_sweep32_dma is its memory pass: on the same GPU a one-shot copy of the tableau in 4-row x 512-column
// tiles dispatched in address order streams at 6.1 TB/s, the same tiles copied by persistent workgroups that walk
// down their strips at 5.0-5.4 (runs of rows, every G-th tile, any depth of prefetch), and persistent workgroups that
// PULL their next tile from a per-strip counter at 6.0-6.2 (scripts/micro/copy_patterns.hip,
// profiles/r03_copy_patterns.txt): what the hardware dispatcher gives one-shot workgroups for free is that tiles
// are handed out in address order to whoever is free, so the rows in flight chip-wide stay one dense, moving window
// (DRAM pages are used up while they are open); statically assigned rows drift apart.
// So here a WAVE is the worker: it is bound to a 128-column sub-strip (its 32 pivot-row slices stay in registers, as
// before) and takes the next 4-row batch of that sub-strip from the sub-strip's ticket counter.  A batch brings its
// own multipliers (32 pivots x 4 rows = 1 KiB = one LDS-DMA, per-lane source addresses), so nothing is shared between
// the waves of a workgroup: no barrier, no chunk, no run length, no rounds of workgroups, no tail — the grid is
// simply what is resident, and a slow CU pulls fewer tickets.
// Per wave and iteration i (one batch each), everything LDS-DMA / hand-counted as in k_sweep32_dma:
//   wait vmcnt(24)   -> what iteration i-3 issued has landed: the rows and multipliers of batch i, the ticket t(i+3)
//   read batch i from stage slot i % 3 (ds_read_b128 x 4)
//   issue: 4 LDS-DMAs of batch t(i+3) into that slot, 1 LDS-DMA of its multipliers into slot (i+3) % 4, 1 ticket atomic
//   32 steps on batch i, multipliers from slot i % 4;  4 stores
// i.e. 10 operations per iteration, and behind the youngest operation of iteration i-3 that must be complete (its
// atomic) come its 4 stores and 2 x 10: 24.  Whenever one of the last three iterations issued less (start, end of the
// tickets) the wait is vmcnt(0).  The ticket's destination is a VGPR of a hand-issued atomic: it is handed to the
// compiler behind the wait and a scheduling barrier, the same two-statement form as strip_wait4.
constexpr int kPullNS = 3, kPullNM = 4;
constexpr int kPullWaveBytes = kPullNS * kDmaSlotBytes + kPullNM * 1024;   // 16 KiB per wave
static_assert(8 * kPullWaveBytes <= 160 * 1024, "two workgroups per CU");

__device__ __forceinline__ void dma_piece1(const double* p, uint32_t lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(p), "s"(lds) : "memory");
}
// one returning atomic add by lane 0; EXEC is saved and restored inside the statement, so the call sites need not run
// with all lanes enabled (they do today: wave-uniform branches only)
__device__ __forceinline__ void ticket_pull(unsigned& tk, unsigned* ctr) {
  const unsigned zero = 0, one = 1;
  unsigned long long keep;
  asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\ts_nop 0\n\tglobal_atomic_add %0, %2, %3, %4 sc0\n\ts_mov_b64 exec, %1"
               : "=v"(tk), "=&s"(keep) : "v"(zero), "v"(one), "s"(ctr) : "memory");
}
__device__ __forceinline__ int ticket_take(unsigned& tk) {   // behind a wait that covers the atomic
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" : "+v"(tk) :: "memory");
  return __builtin_amdgcn_readfirstlane((int)tk);
}

// The multipliers as the pulled batches want them: colT[batch][pivot][row in batch], 1 KiB per batch — ONE contiguous
// LDS-DMA piece, where the ring's [pivot][row] layout makes a batch's multipliers 32 pieces of 32 bytes on 32 different
// lines (measured: the sweep of a block whose multipliers all come from one line ran 9 % faster than a real one).  The
// identity steps of a partly filled block (pivot >= np) get +0 here.  8 MiB at cfg4, a few microseconds in front of the
// sweep on its stream.
template <int KP>
__global__ __launch_bounds__(256) void k_pack_multipliers(const double* __restrict__ col_ring, int64_t mp,
                                                          const LpxCtl* __restrict__ ring, int kmax, int nbt,
                                                          double* __restrict__ colT, unsigned* __restrict__ tickets,
                                                          int nsub, long long* __restrict__ clk) {
  __shared__ int sh_np;
  const int np = ring_count(ring, KP, kmax, &sh_np);
  // clock probe (lpx_state_info.sweep_clock_mhz): shader-clock and 100 MHz stamps in front of the sweep; k_block_fixup
  // takes the matching pair behind it
  if (clk && blockIdx.x < 8 && threadIdx.x == 0) {   // (s_memtime is a per-XCD counter: a pair of stamps per XCD, by ONE workgroup of each — workgroup i runs on XCD i % 8)
    const unsigned x = xcc_id() & 7u;
    clk[x * 4 + 0] = __builtin_amdgcn_s_memtime(); clk[x * 4 + 1] = wall_clock64();
  }
  // the sweep's ticket counters start at zero (one per sub-strip, 128 bytes apart): cleared here, in the launch in
  // front of the sweep, instead of by a memset launch of their own
  if (blockIdx.x == 0)
    for (int u = threadIdx.x; u < nsub; u += 256) tickets[u * 32] = 0u;
  const int s = threadIdx.x % KP;
  const int64_t b = (int64_t)blockIdx.x * (256 / KP) + threadIdx.x / KP;
  if (b >= nbt) return;
  d2 lo = d2{0.0, 0.0}, hi = d2{0.0, 0.0};
  if (s < np) {
    const d2* p = reinterpret_cast<const d2*>(col_ring + (int64_t)s * mp + b * 4);   // mp is even: 16-byte aligned
    lo = p[0];
    hi = p[1];
  }
  d2* q = reinterpret_cast<d2*>(colT + b * (KP * 4) + s * 4);
  q[0] = lo;
  q[1] = hi;
}

// one batch: RB = 4 rows (1 KiB each for this wave) and the batch's 1 KiB of packed multipliers
template <bool NT>
__device__ __forceinline__ void dma_batch4m(const char* base, uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3,
                                            uint32_t lds, const char* mbase, uint32_t lds_m) {
  unsigned keep;
  if (NT)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %1 nt\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %3, %1 nt\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %4, %1 nt\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %5, %1 nt\n\ts_mov_b32 m0, %8\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %7\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(base), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(lds), "s"(mbase), "s"(lds_m)
                 : "memory", "scc");
  else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %1\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %3, %1\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %4, %1\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %5, %1\n\ts_mov_b32 m0, %8\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %7\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(base), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(lds), "s"(mbase), "s"(lds_m)
                 : "memory", "scc");
}

template <bool NT, bool OOP>
__global__ __launch_bounds__(256, 2) void k_sweep32_pull(double* __restrict__ A, const double* __restrict__ Asrc,
                                                         int64_t ld, int m_local,
                                                         const double* __restrict__ prow_ring,
                                                         const double* __restrict__ col_ring, int64_t mp,
                                                         const LpxCtl* __restrict__ ring, int kmax, int nstrips_full,
                                                         const double* __restrict__ col_packed,
                                                         unsigned* __restrict__ tickets) {
  constexpr int K = kDmaK, RB = kDmaRB, NS = kPullNS, NM = kPullNM;
  constexpr int kOps = 2 * RB + 2;                 // per iteration: RB row DMAs, 1 multiplier DMA, 1 atomic, RB stores
  constexpr int kAhead = RB + (NS - 1) * kOps;     // younger than the atomic of iteration i - NS at iteration i's wait
  static_assert(kAhead <= 60, "vmcnt is six bits wide");
  __shared__ __attribute__((aligned(16))) char sm[4 * kPullWaveBytes];
  const int np = ring_count(ring, K, kmax, reinterpret_cast<int*>(sm));
  __syncthreads();   // everyone has read the count before a DMA lands on it
  if (np == 0 && !OOP) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // The grid is G workgroups per strip.  blockIdx % 8 names the XCD's share of the grid; with strip = blockIdx % nstrips
  // an XCD would only see the strips s = x (mod 8), i.e. one eighth of the memory channels (the row pitch is a multiple
  // of the channel interleave).  Shifting each XCD's walk over the strips by x * nstrips / 8 gives every XCD all
  // strips and still every strip G workgroups.
  const int strip = (nstrips_full % 8 == 0)
                        ? (int)(((blockIdx.x >> 3) + (blockIdx.x & 7) * (unsigned)(nstrips_full / 8)) % (unsigned)nstrips_full)
                        : (int)(blockIdx.x % (unsigned)nstrips_full);
  const int sub = strip * 4 + wave;
  const int nbt = m_local / RB;                    // batches of the tableau (m_local % RB == 0: launcher)
  unsigned* const ctr = tickets + sub * 32;        // one counter per sub-strip, 128 bytes apart
  const int64_t row_bytes = ld * 8;
  const int64_t batch_bytes = RB * row_bytes;
  char* const dst_base = reinterpret_cast<char*>(A + sub * 128);
  const char* const src_base = OOP ? reinterpret_cast<const char*>(Asrc + sub * 128) : dst_base;
  const uint32_t off0 = lane * 16u;
  const uint32_t rb32 = (uint32_t)row_bytes;       // 3 rows x ld x 8 < 2^32 (launcher)
  char* const stage = sm + wave * kPullWaveBytes;
  char* const mult = stage + NS * kDmaSlotBytes;
  const uint32_t lds_stage = lds_addr_of(stage), lds_mult = lds_addr_of(mult);
  auto issue = [&](int t, int it) {   // batch t becomes iteration it's: rows -> stage slot it % NS, multipliers -> it % NM
    const char* const base = src_base + (int64_t)t * batch_bytes;                                     // uniform
    const char* const mbase = reinterpret_cast<const char*>(col_packed) + (int64_t)t * 1024;          // uniform
    dma_batch4m<NT>(base, off0, off0 + rb32, off0 + 2 * rb32, off0 + 3 * rb32,
                    (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_stage + (uint32_t)((it % NS) * kDmaSlotBytes))),
                    mbase, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_mult + (uint32_t)((it % NM) * 1024))));
  };

  // prologue: tickets of iterations 0 .. NS-1, their DMAs, the tickets of iterations NS .. 2 NS - 1 (pending), the
  // thread's 32 pivot-row slices; everything is waited for
  unsigned tk[NS];
  int bq[NS + 1];   // bq[k]: the batch of iteration i + k (>= nbt: none)
#pragma unroll
  for (int u = 0; u < NS; ++u) ticket_pull(tk[u], ctr);
  dma_wait<0>();
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    bq[u] = ticket_take(tk[u]);
    if (bq[u] < nbt) issue(bq[u], u);   // uniform
  }
#pragma unroll
  for (int u = 0; u < NS; ++u) ticket_pull(tk[u], ctr);
  d2 pr[K];
#pragma unroll
  for (int s = 0; s < K; ++s) {
    pr[s] = *reinterpret_cast<const d2*>(prow_ring + (int64_t)s * ld + sub * 128 + 2 * lane);
    if (s >= np) pr[s] = d2{0.0, 0.0};   // uniform
  }
  dma_wait<0>();

  int full = 0;   // consecutive most recent iterations that issued all kOps operations
#pragma unroll 1
  for (int i0 = 0;; i0 += NS) {
    if (bq[0] >= nbt) break;   // tickets only grow: nothing is left for this wave (its pending pulls are waited for below)
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int i = i0 + u;
      const int t = bq[0];
      if (t < nbt) {   // uniform
        // (the first NS iterations need what the prologue issued and waited for)
        if (full >= NS) dma_wait<kAhead>(); else if (i >= NS) dma_wait<0>();
        bq[NS] = ticket_take(tk[u]);   // pulled at iteration i - NS: the batch of iteration i + NS
        const char* const slot = stage + u * kDmaSlotBytes + lane * 16;
        d2 x[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) x[r] = *reinterpret_cast<const d2*>(slot + r * 1024);
        const bool more = bq[NS] < nbt;
        if (more) issue(bq[NS], i + NS);   // refills the slot just read (the statement waits for the reads first)
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ticket_pull(tk[u], ctr);
        full = more ? full + 1 : 0;
#if LPX_DMA_DIAG != 1   // (diagnostic build 1: the memory pass alone)
        sweep_apply_lin<K, RB, RB>(x, pr, reinterpret_cast<const double*>(mult + (i % NM) * 1024));
#endif
        char* const out = dst_base + (int64_t)t * batch_bytes;   // uniform
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          d2* q = reinterpret_cast<d2*>(out + (off0 + (uint32_t)r * rb32));
          if (NT) __builtin_nontemporal_store(x[r], q); else *q = x[r];
        }
      } else {
        bq[NS] = INT_MAX;   // (tickets only grow: what this wave still has pending names nothing either)
      }
#pragma unroll
      for (int k = 0; k < NS; ++k) bq[k] = bq[k + 1];
    }
  }
  dma_wait<0>();   // the pending ticket atomics write registers of this wave: let them land before it ends
}

#ifdef LPX_WITH_VARIANTS   // superseded / experiment kernels: csrc/variants/, built by `make variants` only
#define LPX_VARIANT_PART 3
#include "variants/lpx_variants.inc"
#undef LPX_VARIANT_PART
#endif

// ---- blocks of 33..64 pivots by ONE wave per sub-strip (round 4) -----------------------------------------------------
// k_sweep64_pull needs a PAIR of waves per 128-column sub-strip because 64 pivot-row slices of two doubles do not fit one
// wave.  They do when the sub-strip is 64 columns wide: a lane owns ONE column and keeps its 64 pivot-row values in 128
// VGPRs.  Then the worker is a single wave again and the loop is k_sweep32_pull's, nothing handed over, no second wave
// to wait for: tickets per sub-strip, LDS-DMA staging, hand-counted vmcnt.  What changes with the width:
//   * a batch is 4 rows x 512 bytes; one LDS-DMA instruction (64 lanes x 16 bytes) brings TWO rows — lanes 0..31 the
//     first, lanes 32..63 the second (per-lane offsets) — so a batch is two of them, landing as four contiguous rows;
//   * the batch's multipliers ([pivot][row], 64 x 4 doubles = 2 KiB, the layout k_pack_multipliers<64> writes) are two
//     LDS-DMA instructions; per step the wave reads its four multipliers as two 16-byte broadcasts;
//   * a lane reads and stores 8 bytes per row (ds_read_b64, global_store_dwordx2).
// Per iteration: 2 + 2 LDS-DMAs, 1 ticket atomic, 4 stores = 9 operations; behind the atomic of iteration i - 3 come its
// 4 stores and 2 x 9: vmcnt(22).  With fused arithmetic the 64 steps are 256 v_fma_f64 per batch and wave — the sweep's
// instruction floor at cfg4 is 1.2 ms at 2 GHz against a memory pass of ~1.45 ms for HALF the bytes per pivot of a
// block of 32.  Blocks with fewer than 33 valid pivots, the partial last strip and tableaus whose height is not a
// multiple of 4 are left to the generic kernels, as with k_sweep64_pull.
constexpr int kOneSlotBytes = kDmaRB * 512;                                  // a batch: 4 rows x 64 columns
constexpr int kOneWaveBytes = kPullNS * kOneSlotBytes + kPullNM * 2048;      // 14 KiB per wave
static_assert(8 * kOneWaveBytes <= 160 * 1024, "two workgroups per CU");

// (the instruction offset of an LDS-DMA load moves BOTH addresses: the second KiB of the multipliers needs no M0 step)
template <bool NT>
__device__ __forceinline__ void dma_batch2m2(const char* base, uint32_t o0, uint32_t o1, uint32_t lds, const char* mbase,
                                             uint32_t om, uint32_t lds_m) {
  unsigned keep;
  if (NT)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %1 nt\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %3, %1 nt\n\ts_mov_b32 m0, %7\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %6, %5\n\t"
                 "global_load_lds_dwordx4 %6, %5 offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(base), "v"(o0), "v"(o1), "s"(lds), "s"(mbase), "v"(om), "s"(lds_m)
                 : "memory", "scc");
  else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %1\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %3, %1\n\ts_mov_b32 m0, %7\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %6, %5\n\t"
                 "global_load_lds_dwordx4 %6, %5 offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(base), "v"(o0), "v"(o1), "s"(lds), "s"(mbase), "v"(om), "s"(lds_m)
                 : "memory", "scc");
}

// 64 steps on a batch of 4 rows x one column per lane; multipliers [pivot][4 rows] in LDS, read two steps ahead
template <int K>
__device__ __forceinline__ void sweep_apply_one(double (&x)[4], const double (&pr)[K], const double* mrow) {
  constexpr int D = 2;
  d2 cc[D + 1][2];
#pragma unroll
  for (int s = 0; s < D && s < K; ++s) {
    cc[s][0] = *reinterpret_cast<const d2*>(mrow + s * 4);
    cc[s][1] = *reinterpret_cast<const d2*>(mrow + s * 4 + 2);
  }
#pragma unroll
  for (int s = 0; s < K; ++s) {
    if (s + D < K) {
      cc[(s + D) % (D + 1)][0] = *reinterpret_cast<const d2*>(mrow + (s + D) * 4);
      cc[(s + D) % (D + 1)][1] = *reinterpret_cast<const d2*>(mrow + (s + D) * 4 + 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    const d2 c01 = cc[s % (D + 1)][0], c23 = cc[s % (D + 1)][1];
    x[0] = submul(x[0], c01.x, pr[s]);                                             // LPState.java:162
    x[1] = submul(x[1], c01.y, pr[s]);
    x[2] = submul(x[2], c23.x, pr[s]);
    x[3] = submul(x[3], c23.y, pr[s]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <bool NT, bool OOP>
__global__ __launch_bounds__(256, 2) void k_sweep64_one(double* __restrict__ A, const double* __restrict__ Asrc,
                                                        int64_t ld, int m_local,
                                                        const double* __restrict__ prow_ring,
                                                        const LpxCtl* __restrict__ ring, int kmax, int nstrips_full,
                                                        const double* __restrict__ col_packed,   // [batch][64][4]
                                                        unsigned* __restrict__ tickets) {
  constexpr int K = 64, RB = kDmaRB, NS = kPullNS, NM = kPullNM;
  constexpr int kOps = 2 + 2 + 1 + RB;             // per iteration: 2 row DMAs, 2 multiplier DMAs, 1 atomic, RB stores
  constexpr int kAhead = RB + (NS - 1) * kOps;     // younger than the atomic of iteration i - NS at iteration i's wait
  static_assert(kAhead <= 60 && RB == 4, "vmcnt is six bits wide; a batch is two two-row DMAs");
  __shared__ __attribute__((aligned(16))) char sm[4 * kOneWaveBytes];
  const int np = ring_count(ring, K, kmax, reinterpret_cast<int*>(sm));
  __syncthreads();   // everyone has read the count before a DMA lands on it
  if (np <= 32) return;   // 0..32 pivots: the generic kernels behind this launch take the block (two passes)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // the grid is G workgroups per group of four sub-strips (256 columns); XCD x walks the groups shifted by
  // x * ngroups / 8 (see k_sweep32_pull)
  const int ngroups = nstrips_full * 2;
  const int grp = (ngroups % 8 == 0)
                      ? (int)(((blockIdx.x >> 3) + (blockIdx.x & 7) * (unsigned)(ngroups / 8)) % (unsigned)ngroups)
                      : (int)(blockIdx.x % (unsigned)ngroups);
  const int sub = grp * 4 + wave;                  // 64-column sub-strip
  const int nbt = m_local / RB;                    // batches of the tableau (m_local % RB == 0: launcher)
  unsigned* const ctr = tickets + sub * 32;        // one counter per sub-strip, 128 bytes apart
  const int64_t row_bytes = ld * 8;
  const int64_t batch_bytes = RB * row_bytes;
  char* const dst_base = reinterpret_cast<char*>(A + sub * 64);
  const char* const src_base = OOP ? reinterpret_cast<const char*>(Asrc + sub * 64) : dst_base;
  const uint32_t rb32 = (uint32_t)row_bytes;       // 3 rows x ld x 8 < 2^32 (launcher)
  const uint32_t off_r = (uint32_t)(lane >> 5) * rb32 + (uint32_t)(lane & 31) * 16u;   // DMA: two rows per instruction
  const uint32_t off_m = (uint32_t)lane * 16u;
  const uint32_t off_x = (uint32_t)lane * 8u;      // the lane's column inside a row of the sub-strip
  char* const stage = sm + wave * kOneWaveBytes;
  char* const mult = stage + NS * kOneSlotBytes;
  const uint32_t lds_stage = lds_addr_of(stage), lds_mult = lds_addr_of(mult);
  auto issue = [&](int t, int it) {   // batch t becomes iteration it's: rows -> stage slot it % NS, multipliers -> it % NM
    const char* const base = src_base + (int64_t)t * batch_bytes;                                     // uniform
    const char* const mbase = reinterpret_cast<const char*>(col_packed) + (int64_t)t * 2048;          // uniform
    dma_batch2m2<NT>(base, off_r, off_r + 2 * rb32,
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_stage + (uint32_t)((it % NS) * kOneSlotBytes))),
                     mbase, off_m, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_mult + (uint32_t)((it % NM) * 2048))));
  };

  // prologue: tickets of iterations 0 .. NS-1, their DMAs, the tickets of iterations NS .. 2 NS - 1 (pending), the
  // lane's 64 pivot-row values; everything is waited for
  unsigned tk[NS];
  int bq[NS + 1];   // bq[k]: the batch of iteration i + k (>= nbt: none)
#pragma unroll
  for (int u = 0; u < NS; ++u) ticket_pull(tk[u], ctr);
  dma_wait<0>();
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    bq[u] = ticket_take(tk[u]);
    if (bq[u] < nbt) issue(bq[u], u);   // uniform
  }
#pragma unroll
  for (int u = 0; u < NS; ++u) ticket_pull(tk[u], ctr);
  double pr[K];
#pragma unroll
  for (int s = 0; s < K; ++s) {
    pr[s] = prow_ring[(int64_t)s * ld + sub * 64 + lane];
    if (s >= np) pr[s] = 0.0;   // uniform: identity steps of a partly filled block (multiplier +0 as well)
  }
  dma_wait<0>();

  int full = 0;   // consecutive most recent iterations that issued all kOps operations
#pragma unroll 1
  for (int i0 = 0;; i0 += NS) {
    if (bq[0] >= nbt) break;   // tickets only grow: nothing is left for this wave (its pending pulls are waited for below)
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int i = i0 + u;
      const int t = bq[0];
      if (t < nbt) {   // uniform
        // (the first NS iterations need what the prologue issued and waited for)
        if (full >= NS) dma_wait<kAhead>(); else if (i >= NS) dma_wait<0>();
        bq[NS] = ticket_take(tk[u]);   // pulled at iteration i - NS: the batch of iteration i + NS
        const char* const slot = stage + u * kOneSlotBytes + off_x;
        double x[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) x[r] = *reinterpret_cast<const double*>(slot + r * 512);
        const bool more = bq[NS] < nbt;
        if (more) issue(bq[NS], i + NS);   // refills the slot just read (the statement waits for the reads first)
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ticket_pull(tk[u], ctr);
        full = more ? full + 1 : 0;
        sweep_apply_one<K>(x, pr, reinterpret_cast<const double*>(mult + (i % NM) * 2048));
        char* const out = dst_base + (int64_t)t * batch_bytes;   // uniform
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          double* q = reinterpret_cast<double*>(out + (off_x + (uint32_t)r * rb32));
          if (NT) __builtin_nontemporal_store(x[r], q); else *q = x[r];
        }
      } else {
        bq[NS] = INT_MAX;   // (tickets only grow: what this wave still has pending names nothing either)
      }
#pragma unroll
      for (int k = 0; k < NS; ++k) bq[k] = bq[k + 1];
    }
  }
  dma_wait<0>();   // the pending ticket atomics write registers of this wave: let them land before it ends
}

// ---- blocks of 33..64 pivots on the matrix cores (round 4; fused-arithmetic mode only) --------------------------------
// v_mfma_f64_16x16x4_f64 computes D[i][j] = fma(A[i][3], B[3][j], fma(A[i][2], B[2][j], fma(A[i][1], B[1][j],
// fma(A[i][0], B[0][j], C[i][j])))) — a chain of fused multiply-adds in k order, bit for bit (measured: 512 000 random
// entries incl. cancellation cases, 0 mismatches; scripts/micro/mfma_f64_order.hip, profiles/r04_mfma_f64_order.txt).
// That chain IS what the fused mode applies to a tableau entry for four consecutive pending pivots: x <- fma(-col_s[i],
// prow_s[j], x).  So with A = the negated multipliers (16 rows x 4 pivots), B = the pivot rows (4 pivots x 16 columns)
// and C = a 16 x 16 tile of the tableau, sixteen MFMAs in pivot order apply a block of 64 to the tile with the bits of 64
// v_fma_f64 steps.  The rate is the vector unit's (MI355X: fp64 matrix = fp64 vector peak); what the matrix path saves
// is everything AROUND the arithmetic: one A operand per lane feeds 1024 multiply-adds, where the vector kernels read
// two 16-byte LDS broadcasts per four (k_sweep64_one is bound by exactly that: LDS issue).
// A wave is bound to a 64-column sub-strip: its B operands — 16 pivot groups x 4 column tiles, one double per lane —
// stay in 128 VGPRs.  It pulls 16-row tiles of the sub-strip from the sub-strip's ticket counter; a tile is 16 x 64
// entries (8 KiB) in the MFMA's C layout (C[4 r + lane / 16][lane % 16] in register r: one global_load_dwordx2 = four
// 128-byte row segments) plus its 8 KiB of negated multipliers in A layout (k_pack_multipliers_mfma: [tile][group][lane],
// lane = 16 k + i; -0.0 for the identity steps of a partly filled block, whose B values are +0.0: fma(-0, +0, x) = x).
// Three tile buffers rotate (ordinary loads, the compiler counts them): while tile t computes its 64 MFMAs, tiles t + 1
// and t + 2 are in flight.  One wave per SIMD (the register file is the budget), LDS unused.
#if LPX_FUSED
typedef double d4v __attribute__((ext_vector_type(4)));

// A operands of the MFMA sweep: colM[(tile * 16 + g) * 64 + 16 k + i] = -col_ring[4 g + k][16 tile + i]
// (pairs != 0, k_sweep64_mfma2: groups 2 p and 2 p + 1 interleaved per lane, colM[(tile * 8 + p) * 128 + 2 lane + (g & 1)], so that
// one 16-byte load per lane brings the A operands of two groups)
__global__ __launch_bounds__(256) void k_pack_multipliers_mfma(const double* __restrict__ col_ring, int64_t mp,
                                                               const LpxCtl* __restrict__ ring, int kmax, int ntiles,
                                                               double* __restrict__ colM, unsigned* __restrict__ tickets,
                                                               int nsub, long long* __restrict__ clk, int pairs) {
  __shared__ int sh_np;
  const int np = ring_count(ring, 64, kmax, &sh_np);
  if (clk && blockIdx.x < 8 && threadIdx.x == 0) {   // (s_memtime is a per-XCD counter: a pair of stamps per XCD, by ONE workgroup of each — workgroup i runs on XCD i % 8)
    const unsigned x = xcc_id() & 7u;
    clk[x * 4 + 0] = __builtin_amdgcn_s_memtime(); clk[x * 4 + 1] = wall_clock64();
  }
  if (blockIdx.x == 0)
    for (int u = threadIdx.x; u < nsub; u += 256) tickets[u * 32] = 0u;
  // one workgroup per tile: thread = (pivot group pair, lane); reads along rows (i fastest): 128-byte segments
  const int tile = blockIdx.x;
  if (tile >= ntiles) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = q * 256 + threadIdx.x;          // 0 .. 1023 = 16 groups x 64 lanes
    const int g = idx >> 6, l = idx & 63, k = l >> 4, i = l & 15;
    const int s = 4 * g + k;
    const double c = s < np ? col_ring[(int64_t)s * mp + (int64_t)tile * 16 + i] : 0.0;
    colM[(int64_t)tile * 1024 + (pairs ? (g >> 1) * 128 + 2 * l + (g & 1) : idx)] = -c;
  }
}

#ifdef LPX_WITH_VARIANTS
#define LPX_VARIANT_PART 4
#include "variants/lpx_variants.inc"
#undef LPX_VARIANT_PART
#endif


// Second form: TWO waves per SIMD.  k_sweep64_mfma keeps the B operands in 128 VGPRs and therefore runs one wave per
// SIMD: while that wave issues a tile's 48 loads and 16 stores and waits for the last MFMAs of a tile, the matrix pipe
// idles (measured: 2.3 ms per 64 pivots at cfg4 against 1.45 ms of arithmetic).  Here a workgroup is bound to a
// 128-column group, its four waves to the two 64-column sub-strips in pairs, and the B operands of both sub-strips live
// in LDS (2 x 32 KiB, [sub-strip][group][column tile][lane]: every MFMA's B is one conflict-free ds_read_b64 with an
// immediate offset); the registers hold three tiles of C and A.  Two workgroups per CU.
// Memory side (second version; the first one lost a third of its time here, profiles/r04_sweep64_mfma2_what_bounds.txt):
//  * buffer addressing — the tile's base in the resource descriptor (SALU), the row group 4 r in the scalar offset, the
//    lane part in ONE 32-bit VGPR, the column tile in the immediate: no per-load 64-bit VALU address, 30 fewer VGPRs;
//  * the ticket is a hand-issued atomic (ticket_pull) taken behind `s_waitcnt vmcnt(48)` = the 32 loads and 16 stores
//    issued after it.  (`if (lane == 0) atomicAdd` compiled to an aggregated atomic followed by `s_waitcnt vmcnt(0)`:
//    every tile drained the wave's whole queue, stores included.)  The compiler does not know of the atomic, so its own
//    counts are one too strict, never too lax;
//  * the loop body is straight-line: the tile of a ticket past the end is CLAMPED to the last tile for its loads (a
//    re-read, two per wave) and the loop is left before its arithmetic, so no conditional load makes the compiler's
//    vmcnt bookkeeping fall back to draining counts.
constexpr int kMfma2LdsBytes = 2 * 16 * 4 * 64 * 8;
static_assert(2 * kMfma2LdsBytes <= 160 * 1024, "two workgroups per CU");
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <bool NT, bool OOP>
__global__ __launch_bounds__(256, 2) void k_sweep64_mfma2(double* A, const double* Asrc,   // (no __restrict__: see below)
                                                          int64_t ld, int m_local,
                                                          const double* __restrict__ prow_ring,
                                                          const LpxCtl* __restrict__ ring, int kmax, int nstrips_full,
                                                          const double* colM,   // [tile][group][lane]
                                                          unsigned* tickets, int kmin) {
  constexpr int NG = 16, CT = 4;
  constexpr int kRsrcWord3 = 0x00020000;           // raw buffer, 32-bit data format (gfx9 family)
  constexpr int kAuxNt = NT ? 2 : 0;               // cache policy bit 1 = nt
  __shared__ __attribute__((aligned(16))) double sh_b[2 * NG * CT * 64];
  __shared__ int sh_np;
  const int np = ring_count(ring, 64, kmax, &sh_np);
  // kmin = 33: a block of at most 32 valid pivots (it ended early) is left to the generic kernels launched behind; kmin = 1
  // (whole strips only, no such launches): taken here too — the identity steps of a partly filled block are exact
  // (multiplier -0.0 against a pivot-row value of +0.0), whatever their number
  if (np < kmin) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ngroups = nstrips_full * 4;            // groups of 128 columns
  const int grp = (ngroups % 8 == 0)
                      ? (int)(((blockIdx.x >> 3) + (blockIdx.x & 7) * (unsigned)(ngroups / 8)) % (unsigned)ngroups)
                      : (int)(blockIdx.x % (unsigned)ngroups);
  // ONE ticket counter per 128-column group: ticket T = (16-row block T / 2, 64-column half T % 2), so the two halves of
  // a row's 1 KiB (one DRAM page) are pulled back to back by two waves (second version; with a counter per half the
  // halves of a page were fetched at unrelated times: 5.3 TB/s alone where k_sweep32_pull's 1 KiB segments stream at 6)
  const int ntiles = 2 * (m_local / 16);
  unsigned* const ctr = tickets + grp * 32;
  const int64_t row_bytes = ld * 8;
  char* const dst_base = reinterpret_cast<char*>(A + grp * 128);
  const char* const src_base = OOP ? reinterpret_cast<const char*>(Asrc + grp * 128) : dst_base;
  const uint32_t rb32 = (uint32_t)row_bytes;       // 16 rows x ld x 8 < 2^32 (launcher)
  const uint32_t off_c = (uint32_t)(lane >> 4) * rb32 + (uint32_t)(lane & 15) * 8u;   // row lane / 16, column lane % 16
  const uint32_t off_a = (uint32_t)lane * 8u;
  // the B operands of the group's two halves into LDS: the four waves share the 128 (half, group, column tile) images
  for (int q = wave; q < 2 * NG * CT; q += 4) {
    const int h = q / (NG * CT), g = (q / CT) % NG, ct = q % CT;
    const int s = 4 * g + (lane >> 4);
    sh_b[q * 64 + lane] = s < np ? prow_ring[(int64_t)s * ld + grp * 128 + h * 64 + ct * 16 + (lane & 15)] : 0.0;
  }
  __syncthreads();
  auto load_tile = [&](int t, d4v (&c)[CT], double (&a)[NG]) {
    const int tt = max(0, min(t, ntiles - 1));                          // uniform; past the end: the last tile again
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(src_base) + (int64_t)(tt >> 1) * 16 * row_bytes + (tt & 1) * 512, 0, -1, kRsrcWord3);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
        c[ct][r] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rc, off_c + (uint32_t)ct * 128u,
                                                                                    (int)((uint32_t)(4 * r) * rb32), kAuxNt));
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double*>(colM) + (int64_t)(tt >> 1) * 1024, 0, -1, kRsrcWord3);
#pragma unroll
    for (int p = 0; p < NG / 2; ++p) {   // (k_pack_multipliers_mfma, pairs: the A operands of groups 2 p and 2 p + 1 side by side)
      const v4u x = __builtin_amdgcn_raw_buffer_load_b128(ra, 2u * off_a + (uint32_t)(p & 3) * 1024u, (p >> 2) * 4096, 0);
      a[2 * p] = __builtin_bit_cast(double, v2u{x[0], x[1]});
      a[2 * p + 1] = __builtin_bit_cast(double, v2u{x[2], x[3]});
    }
  };
// scheduling barrier between a tile's loads and the previous tile's arithmetic (without it the compiler sinks the loads
// behind the MFMA chain: three times slower, profiles/r04_sweep64_mfma2_no_sched_barriers.txt)
#define LPX_MFMA_FENCE __builtin_amdgcn_sched_barrier(0)
  auto work_tile = [&](int t, d4v (&c)[CT], const double (&a)[NG]) {
    const double* const bl = sh_b + (max(0, min(t, ntiles - 1)) & 1) * (NG * CT * 64) + lane;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        c[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[g], bl[(g * CT + ct) * 64], c[ct], 0, 0, 0);
      }
    // the B operands of group g + 1 (two ds_read2st64_b64) are asked for in front of group g's four MFMAs
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
    // a ticket past the end: the arithmetic runs on the re-read last tile and the stores are DROPPED by the buffer's range
    // check (num_records 0), so the loop body has no exit but its back edge
    const int tt = max(0, min(t, ntiles - 1));
    char* const out = dst_base + (int64_t)(tt >> 1) * 16 * row_bytes + (tt & 1) * 512;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(out, 0, __builtin_amdgcn_readfirstlane((unsigned)t < (unsigned)ntiles ? -1 : 0), kRsrcWord3);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, (double)c[ct][r]), rd, off_c + (uint32_t)ct * 128u,
                                              (int)((uint32_t)(4 * r) * rb32), kAuxNt);
  };
  // A ticket is pulled a whole step before it is taken: pulled in front of step i - 1's loads, taken in front of step i's.
  // vmcnt counts in order, so the wait in front of the take covers exactly what is OLDER than step i - 1's loads, stores and
  // step i's pull (41 operations) — operations that have had a step's arithmetic to complete and that step i's MFMAs need
  // anyway.  (Taken behind the same step's stores, first version, every step ended by waiting for the loads and stores it
  // had just issued.)
  constexpr int kStepOps = 16 + 8 + 16;   // a step's tile loads, A-operand loads and stores
  static_assert(kStepOps + 1 <= 63, "vmcnt is six bits wide");
  auto take = [&](unsigned& tk) -> int {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kStepOps + 1) : "memory");
    return ticket_take(tk);
  };
  d4v c0[CT], c1[CT], c2[CT];
  double a0[NG], a1[NG], a2[NG];
  int t0, t1;
  unsigned k0, k1, k2;                  // three ticket registers, rotating with the tile buffers
  {
    ticket_pull(k0, ctr); ticket_pull(k1, ctr);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t0 = ticket_take(k0); t1 = ticket_take(k1);
  }
  if (t0 >= ntiles) return;
  load_tile(t0, c0, a0);
  load_tile(t1, c1, a1);
  ticket_pull(k2, ctr);                 // the ticket of the tile the loop's first step loads (covered by the wait below)
  // The loop is entered with nothing in flight (once per wave): the compiler's vmcnt bookkeeping at the loop head is
  // then the loop-carried state alone — merged with the prologue's tiles in flight it waited in the first step of EVERY
  // round for loads the previous step had just issued.  (A real s_waitcnt, which the compiler's pass reads.)
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt and lgkmcnt untouched
  // Straight-line body, ONE exit at the back edge.  (With `break`s between the steps the compiler folds the exits into
  // one latch block, and its vmcnt bookkeeping then sees paths that skip a step: it waited for the loads the previous
  // step had just issued.)  Tickets only grow, t0 < t1 < ...: once t0 is past the end everything later is.
#pragma unroll 1
  do {
    ticket_pull(k0, ctr);
    const int t2 = take(k2);
    load_tile(t2, c2, a2);
    LPX_MFMA_FENCE;
    work_tile(t0, c0, a0);
    LPX_MFMA_FENCE;
    ticket_pull(k1, ctr);
    const int t3 = take(k0);
    load_tile(t3, c0, a0);
    LPX_MFMA_FENCE;
    work_tile(t1, c1, a1);
    LPX_MFMA_FENCE;
    ticket_pull(k2, ctr);
    const int t4 = take(k1);
    load_tile(t4, c1, a1);
    LPX_MFMA_FENCE;
    work_tile(t2, c2, a2);
    LPX_MFMA_FENCE;
    t0 = t3; t1 = t4;
  } while (t0 < ntiles);
}

#undef LPX_MFMA_FENCE
#ifdef LPX_WITH_VARIANTS
#define LPX_VARIANT_PART 7
#include "variants/lpx_variants.inc"
#undef LPX_VARIANT_PART
#endif
#endif  // LPX_FUSED

#ifdef LPX_WITH_VARIANTS   // superseded / experiment kernels: csrc/variants/, built by `make variants` only
#define LPX_VARIANT_PART 5
#include "variants/lpx_variants.inc"
#undef LPX_VARIANT_PART
#endif

// One pivot applied to one value with the reference's full case analysis (LPState.java:139-164): the value at
// row i, column j before pivot r -> after pivot r.
__device__ __forceinline__ double apply_pivot(double v, int i, int j, int l_r, int e_r, double p_r, double ce,
                                              double pr_j) {
  if (i == l_r) return pr_j;                       // pivot row := normalised row (pr_j = 1/p at j == e_r)
  if (j == e_r) return -__ddiv_rn(ce, p_r);        // :157
  return submul(v, ce, pr_j);        // :162
}

// After the sweep: recompute the entering columns (job 0), the pivot rows (job 1) and b (job 2) of the valid
// pending pivots from the saved stale values.  grid = (ceil(max(m, ld)/256), ceil(K / 8), 3): a thread takes EIGHT
// pending pivots' columns (rows) of its row (column) at once, so that a ring value it loads serves eight chains (one
// pivot per thread re-read the K x m / K x ld ring values K times over: 1.6 GB of L2 traffic per block of 64 at cfg4).
// The case analysis of apply_pivot per (pending pivot r, chain q) made the kernel instruction-bound (~7 700 instructions
// per thread for 512 multiply-adds: 215 us per block of 64 at cfg4, and NOT its scattered 8-byte column writes — written to
// a compact image instead it took 195, EXPERIMENTS.md).  None of its tests depends on the thread:
//   * "chain q's pivot entered at (left through) the same slot (row) as pivot r" is a property of the block: one 8-bit
//     mask per r, computed once per workgroup (sh_eq);
//   * "this thread's row (column) is pivot r's own" can only hold in the one wave whose 64 rows (columns) contain it: one
//     __ballot per wave (lane r asks for pivot r), then a scalar bit test per r.
// A step without either is eight multiply-adds on eight LDS values; with one, the old selects behind a uniform branch.
constexpr int kFixChunk = 8;
// The chains of a job: the LAST pending pivot of every distinct key (entering slot / leaving row), in pivot order.  Called
// by the first wave of a workgroup (all 64 lanes) between two barriers; keys[q] = -1 behind the last pivot.
__device__ __forceinline__ void fix_pick_chains(const int* keys, int np, int* sh_pick, int* sh_npick) {
  const int q = threadIdx.x;
  const int key = keys[q];
  bool keep = q < np;
  for (int r = q + 1; r < np; ++r) keep = keep && keys[r] != key;
  const unsigned long long mask = __ballot(keep);
  if (keep) sh_pick[__popcll(mask & ((1ull << q) - 1ull))] = q;
  const int npick = __popcll(mask);
  if (q < kFixChunk) sh_pick[npick + q] = 0;   // (padding of the last chunk: reads stay inside the ring)
  if (q == 0) *sh_npick = npick;
}
__global__ __launch_bounds__(256) void k_block_fixup(double* __restrict__ A, int64_t ld, int n, int m_local, int row0,
                                                     double* b, const double* __restrict__ prow_ring,
                                                     const double* __restrict__ col_ring,
                                                     const double* __restrict__ col0_ring,
                                                     const double* __restrict__ row0_ring, int64_t mp,
                                                     const LpxCtl* __restrict__ ring, int kmax,
                                                     const double* b_src, long long* __restrict__ clk,
                                                     double* __restrict__ img_col, double* __restrict__ img_row) {
  // img_col / img_row != NULL (the overlapped loop): the chains are computed BESIDE the block's sweep, from ring values
  // only, into compact images [chain][row] / [chain][column]; k_block_fixup_scatter copies them into the tableau once the
  // sweep is through.  b (job 2) is written in place either way: the sweep does not touch it.
  static_assert(kBlockMax <= 64, "one lane per pending pivot in the hit ballots");
  if (clk && blockIdx.x < 8 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
    const unsigned x = xcc_id() & 7u;
    clk[x * 4 + 2] = __builtin_amdgcn_s_memtime(); clk[x * 4 + 3] = wall_clock64();
  }
  __shared__ double sh_p[kBlockMax], sh_bl[kBlockMax];
  __shared__ __attribute__((aligned(16))) double sh_x[kBlockMax][kFixChunk];   // job 0: prow_r[e_s]; job 1: col_r[l_s]   (r: all pivots, s: this chunk's)
  __shared__ int sh_e[kBlockMax], sh_l[kBlockMax], sh_eq[kBlockMax], sh_pick[kBlockMax + kFixChunk];
  __shared__ int sh_np, sh_npick;
  const int s0 = blockIdx.y * kFixChunk, job = blockIdx.z;
  // (the grid is max(m, ld) wide for all three jobs: a workgroup with nothing to do leaves before the ring is looked at)
  if ((int64_t)blockIdx.x * blockDim.x >= (job == 1 ? ld : (int64_t)m_local) || (job == 2 && s0 != 0)) return;
  const int np = ring_count(ring, kBlockMax, kmax, &sh_np);
  if (job == 2 ? s0 != 0 : s0 >= np) return;  // the b job also runs for an empty block (out of place: it copies b)
  if ((int)threadIdx.x < kBlockMax) {   // (slots behind the last pivot: -1, matches nothing)
    const bool live = (int)threadIdx.x < np;
    const LpxCtl& q = ring[live ? threadIdx.x : 0];
    sh_e[threadIdx.x] = live ? q.e_cur : -1;
    sh_l[threadIdx.x] = live ? q.l - row0 : -1;
    sh_p[threadIdx.x] = live ? q.p : 1.0;
    sh_bl[threadIdx.x] = live ? q.bl : 0.0;
  }
  __syncthreads();
  // Pivots that share their entering slot (their leaving row) end in the SAME column (row): a later pivot at the slot
  // restarts the column whatever it held (:157), a later pivot through the row replaces the row (:139-145).  So only the
  // LAST pivot of every slot (row) is a chain here — under the first-positive rule a block of 64 pivots touches 10-15
  // distinct slots and rows (three decisions of four come back to a slot of the last 48), i.e. a quarter of the chains
  // and of the scattered column writes.
  if (job != 2 && threadIdx.x < 64) fix_pick_chains(job == 1 ? sh_l : sh_e, np, sh_pick, &sh_npick);
  __syncthreads();
  const int npick = job == 2 ? 0 : sh_npick;
  if (job != 2 && s0 >= npick) return;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int ns = min(kFixChunk, npick - s0);   // chains of this chunk (job 0 / 1)
  int pk[kFixChunk];                         // their pivots
#pragma unroll
  for (int q = 0; q < kFixChunk; ++q) pk[q] = sh_pick[(job == 2 ? 0 : s0) + q];
  const int lane = threadIdx.x & 63;
  const int t_wave = t - lane;              // the wave's first row (job 0, 2) / column (job 1)
  if (job == 0) {  // entering columns of pending pivots s0 .. s0 + ns - 1, all local rows
    for (int idx = threadIdx.x; idx < np * kFixChunk; idx += blockDim.x) {
      const int r = idx / kFixChunk, q = idx % kFixChunk;
      sh_x[r][q] = q < ns ? prow_ring[(int64_t)r * ld + sh_e[sh_pick[s0 + q]]] : 0.0;
    }
    if ((int)threadIdx.x < np) {   // which chains of the chunk entered at pivot r's slot (the division of :157)
      int mask = 0;
      for (int q = 0; q < ns; ++q) mask |= (sh_e[sh_pick[s0 + q]] == sh_e[threadIdx.x]) << q;
      sh_eq[threadIdx.x] = mask;
    }
    __syncthreads();
    // pivots whose own row is one of this wave's 64 rows (all lanes active here: lane r asks for pivot r)
    const unsigned long long hit = __ballot((unsigned)(sh_l[lane] - t_wave) < 64u);
    if (t < m_local) {
      double v[kFixChunk];
#pragma unroll
      for (int q = 0; q < kFixChunk; ++q) v[q] = q < ns ? col0_ring[(int64_t)pk[q] * mp + t] : 0.0;
      // the ring values of eight steps are requested together (they do not depend on the running values), one batch
      // ahead of the arithmetic that uses them
      double cv[8], cvn[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) cv[u] = (u < np) ? col_ring[(int64_t)u * mp + t] : 0.0;
      for (int r0 = 0; r0 < np; r0 += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) cvn[u] = (r0 + 8 + u < np) ? col_ring[(int64_t)(r0 + 8 + u) * mp + t] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int r = r0 + u;
          if (r < np) {   // (uniform)
            double x[kFixChunk];
#pragma unroll
            for (int q = 0; q < kFixChunk; ++q) x[q] = sh_x[r][q];
            const int eq = __builtin_amdgcn_readfirstlane(sh_eq[r]);
            if (eq == 0 && !((hit >> r) & 1)) {   // (uniform, the common case) eight plain steps
#pragma unroll
              for (int q = 0; q < kFixChunk; ++q) v[q] = submul(v[q], cv[u], x[q]);                    // :162
            } else {
              const double pr = sh_p[r];
              const bool own = t == sh_l[r];
#pragma unroll
              for (int q = 0; q < kFixChunk; ++q) {
                double nv;
                if ((eq >> q) & 1) nv = -__ddiv_rn(cv[u], pr);                                          // :157
                else nv = submul(v[q], cv[u], x[q]);                                                   // :162
                v[q] = own ? x[q] : nv;   // pivot row := normalised row
              }
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) cv[u] = cvn[u];
      }
#pragma unroll
      for (int q = 0; q < kFixChunk; ++q)
        if (q < ns) {
          if (img_col) img_col[(int64_t)(s0 + q) * mp + t] = v[q];
          else A[(int64_t)t * ld + sh_e[pk[q]]] = v[q];
        }
    }
  } else if (job == 1) {  // pivot rows of pending pivots s0 .. (those that live on this shard), all columns
    for (int idx = threadIdx.x; idx < np * kFixChunk; idx += blockDim.x) {
      const int r = idx / kFixChunk, q = idx % kFixChunk;
      const int i = q < ns ? sh_l[sh_pick[s0 + q]] : -1;
      sh_x[r][q] = (i >= 0 && i < m_local) ? col_ring[(int64_t)r * mp + i] : 0.0;
    }
    if ((int)threadIdx.x < np) {   // which chains of the chunk left through pivot r's row (row := normalised row)
      int mask = 0;
      for (int q = 0; q < ns; ++q) mask |= (sh_l[sh_pick[s0 + q]] == sh_l[threadIdx.x]) << q;
      sh_eq[threadIdx.x] = mask;
    }
    __syncthreads();
    // pivots that entered at one of this wave's 64 slots (the division of :157 sits in THAT column)
    const unsigned long long hit = __ballot((unsigned)(sh_e[lane] - t_wave) < 64u);
    if (t < (int)ld) {
      double v[kFixChunk];
#pragma unroll
      for (int q = 0; q < kFixChunk; ++q) v[q] = (q < ns && t < n) ? row0_ring[(int64_t)pk[q] * ld + t] : 0.0;
      double pv[8], pvn[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) pv[u] = (u < np) ? prow_ring[(int64_t)u * ld + t] : 0.0;
      for (int r0 = 0; r0 < np; r0 += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) pvn[u] = (r0 + 8 + u < np) ? prow_ring[(int64_t)(r0 + 8 + u) * ld + t] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int r = r0 + u;
          if (r < np) {   // (uniform)
            double x[kFixChunk];
#pragma unroll
            for (int q = 0; q < kFixChunk; ++q) x[q] = sh_x[r][q];
            const int eq = __builtin_amdgcn_readfirstlane(sh_eq[r]);
            if (eq == 0 && !((hit >> r) & 1)) {
#pragma unroll
              for (int q = 0; q < kFixChunk; ++q) v[q] = submul(v[q], x[q], pv[u]);
            } else {
              const double pr = sh_p[r];
              const bool own = t == sh_e[r];   // (one thread of the grid per pivot r)
#pragma unroll
              for (int q = 0; q < kFixChunk; ++q) {
                double nv;
                if (own) nv = -__ddiv_rn(x[q], pr);
                else nv = submul(v[q], x[q], pv[u]);
                v[q] = ((eq >> q) & 1) ? pv[u] : nv;
              }
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) pv[u] = pvn[u];
      }
#pragma unroll
      for (int q = 0; q < kFixChunk; ++q) {
        const int i = q < ns ? sh_l[pk[q]] : -1;
        if (i >= 0 && i < m_local) {
          if (img_row) img_row[(int64_t)(s0 + q) * ld + t] = v[q];
          else A[(int64_t)i * ld + t] = v[q];
        }
      }
    }
  } else {  // b of every local row (LPState.java:164 / :146)
    const unsigned long long hit = __ballot((unsigned)(sh_l[lane] - t_wave) < 64u);
    if (t < m_local) {
      double bi = b_src[t];  // == b unless the sweep ran out of place
      for (int r0 = 0; r0 < np; r0 += 8) {
        double cv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) cv[q] = (r0 + q < np) ? col_ring[(int64_t)(r0 + q) * mp + t] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = r0 + q;
          if (r < np) {
            const double nb = submul(bi, cv[q], sh_bl[r]);
            bi = nb;
            if ((hit >> r) & 1) { if (t == sh_l[r]) bi = sh_bl[r]; }
          }
        }
      }
      b[t] = bi;
    }
  }
}

// Behind the sweep of the overlapped loop: the entering columns (job 0) and pivot rows (job 1) that k_block_fixup computed
// beside it go from the images into the tableau.  grid = (ceil(max(m, ld)/256), ceil(K / 8), 2), the chains found as there.
__global__ __launch_bounds__(256) void k_block_fixup_scatter(double* __restrict__ A, int64_t ld, int m_local, int row0,
                                                             const double* __restrict__ img_col,
                                                             const double* __restrict__ img_row, int64_t mp,
                                                             const LpxCtl* __restrict__ ring, int kmax,
                                                             long long* __restrict__ clk) {
  if (clk && blockIdx.x < 8 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
    const unsigned x = xcc_id() & 7u;
    clk[x * 4 + 2] = __builtin_amdgcn_s_memtime(); clk[x * 4 + 3] = wall_clock64();
  }
  __shared__ int sh_key[kBlockMax], sh_pick[kBlockMax + kFixChunk];
  __shared__ int sh_np, sh_npick;
  const int s0 = blockIdx.y * kFixChunk, job = blockIdx.z;
  if ((int64_t)blockIdx.x * blockDim.x >= (job == 1 ? ld : (int64_t)m_local)) return;
  const int np = ring_count(ring, kBlockMax, kmax, &sh_np);
  if (s0 >= np) return;
  if ((int)threadIdx.x < kBlockMax) {
    const bool live = (int)threadIdx.x < np;
    const LpxCtl& q = ring[live ? threadIdx.x : 0];
    sh_key[threadIdx.x] = live ? (job == 1 ? q.l - row0 : q.e_cur) : -1;
  }
  __syncthreads();
  if (threadIdx.x < 64) fix_pick_chains(sh_key, np, sh_pick, &sh_npick);
  __syncthreads();
  const int npick = sh_npick;
  if (s0 >= npick) return;
  const int ns = min(kFixChunk, npick - s0);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  double v[kFixChunk];
  if (job == 0) {
    if (t >= m_local) return;
#pragma unroll
    for (int q = 0; q < kFixChunk; ++q) v[q] = q < ns ? img_col[(int64_t)(s0 + q) * mp + t] : 0.0;
#pragma unroll
    for (int q = 0; q < kFixChunk; ++q)
      if (q < ns) A[(int64_t)t * ld + sh_key[sh_pick[s0 + q]]] = v[q];
  } else {
    if (t >= (int)ld) return;
#pragma unroll
    for (int q = 0; q < kFixChunk; ++q) {
      const int i = q < ns ? sh_key[sh_pick[s0 + q]] : -1;
      v[q] = (i >= 0 && i < m_local) ? img_row[(int64_t)(s0 + q) * ld + t] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < kFixChunk; ++q) {
      const int i = q < ns ? sh_key[sh_pick[s0 + q]] : -1;
      if (i >= 0 && i < m_local) A[(int64_t)i * ld + t] = v[q];
    }
  }
}

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
                        hipStream_t s, const MgPeers* mg) {
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
      if (wide) hipLaunchKernelGGL((k_block_chain2_t<64, kChain2Threads, true>), dim3(G), dim3(kChain2Threads), 0, s, P);
      else hipLaunchKernelGGL((k_block_chain2_t<32, kChain2Threads, true>), dim3(G), dim3(kChain2Threads), 0, s, P);
    } else hipLaunchKernelGGL((k_block_chain_t<true, 32>), dim3(G), dim3(256), 0, s, P);   // (shards decide at most kShardBlockMax = 32 per block)
  } else if (form2) {
    P.m_global = m; P.n_dev = 1;
#ifdef LPX_CHAIN2_ONE_XCD
    G = std::min(G, 32);
    if (wide) hipLaunchKernelGGL((k_block_chain2_t<64, kChain2Threads, false>), dim3(8 * G), dim3(kChain2Threads), 0, s, P);
    else hipLaunchKernelGGL((k_block_chain2_t<32, kChain2Threads, false>), dim3(8 * G), dim3(kChain2Threads), 0, s, P);
#else
    if (wide) hipLaunchKernelGGL((k_block_chain2_t<64, kChain2Threads, false>), dim3(G), dim3(kChain2Threads), 0, s, P);
    else hipLaunchKernelGGL((k_block_chain2_t<32, kChain2Threads, false>), dim3(G), dim3(kChain2Threads), 0, s, P);
#endif
  } else {
#ifdef LPX_WITH_VARIANTS
    P.m_global = m; P.n_dev = 1;
    if (wide) hipLaunchKernelGGL((k_block_chain_t<false, 64>), dim3(G), dim3(256), 0, s, P);
    else hipLaunchKernelGGL((k_block_chain_t<false, 32>), dim3(G), dim3(256), 0, s, P);
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
// then spare slots; the LAST slot is never a counter — it is the word a pull kernel sets when a bounded wait ran out.
int64_t sweep_ticket_slots(int64_t ld) { return std::max<int64_t>(ld / 64, 4) + 4; }   // (k_sweep64_one: a counter per 64 columns)
unsigned* sweep_fail_word(const BlockRing& R, int64_t ld) {
#ifdef LPX_WITH_VARIANTS   // (the only kernel that sets it, k_sweep64_pull, is in the variants library only)
  return R.tickets ? R.tickets + (sweep_ticket_slots(ld) - 1) * 32 : nullptr;
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
  hipLaunchKernelGGL((k_sweep32_pull<NT_, OOP_>), dim3(1), dim3(256), 0, s, A, A, ld, 0, R.prow, R.col, R.mp, R.up, 0, 1, R.col_packed, R.tickets);
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
    hipLaunchKernelGGL((k_sweep64_one<NT_, OOP_>), dim3(1), dim3(256), 0, s, A, A, ld, 0, R.prow, R.up, 0, 1, R.col_packed, R.tickets);
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
    hipLaunchKernelGGL((k_sweep64_mfma2<NT_, OOP_>), dim3(1), dim3(256), 0, s, A, A, ld, 0, R.prow, R.up, 0, 1, R.col_packed, R.tickets, 33);
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
static void launch_sweep_pull(const Buffers& B, const BlockRing& R, int m_local, int kmax, bool nt, const double* A_src,
                              hipStream_t s, int slots = 512) {
  const int nstrips_full = (int)(B.ld / 512);
  const int nbt = m_local / 4;
  const int G = std::max(1, std::min(nbt, slots / std::max(1, nstrips_full)));
  hipLaunchKernelGGL(k_pack_multipliers<32>, dim3((nbt + 7) / 8), dim3(256), 0, s, R.col, R.mp, R.up, kmax, nbt, R.col_packed,
                     R.tickets, nstrips_full * 4, R.clk);
  const dim3 grid(nstrips_full * G), block(256);
#define LPX_LAUNCH_PULL(NT_, OOP_)                                                                                \
  hipLaunchKernelGGL((k_sweep32_pull<NT_, OOP_>), grid, block, 0, s, B.A, A_src, B.ld, m_local, R.prow, R.col, R.mp, \
                     R.up, kmax, nstrips_full, R.col_packed, R.tickets)
  if (A_src) { if (nt) LPX_LAUNCH_PULL(true, true); else LPX_LAUNCH_PULL(false, true); }
  else { if (nt) LPX_LAUNCH_PULL(true, false); else LPX_LAUNCH_PULL(false, false); }
#undef LPX_LAUNCH_PULL
}

// blocks of 33..64 by single waves on 64-column sub-strips (k_sweep64_one); G workgroups per group of four sub-strips
static void launch_sweep64_one(const Buffers& B, const BlockRing& R, int m_local, int kmax, bool nt, const double* A_src,
                               hipStream_t s, int slots = 512) {
  const int nstrips_full = (int)(B.ld / 512);
  const int ngroups = nstrips_full * 2;
  const int nbt = m_local / 4;
  const int G = std::max(1, std::min(nbt, slots / std::max(1, ngroups)));
  hipLaunchKernelGGL(k_pack_multipliers<64>, dim3((nbt + 3) / 4), dim3(256), 0, s, R.col, R.mp, R.up, kmax, nbt, R.col_packed,
                     R.tickets, nstrips_full * 8, R.clk);
  const dim3 grid(ngroups * G), block(256);
#define LPX_LAUNCH_ONE64(NT_, OOP_)                                                                               \
  hipLaunchKernelGGL((k_sweep64_one<NT_, OOP_>), grid, block, 0, s, B.A, A_src, B.ld, m_local, R.prow, R.up, kmax, \
                     nstrips_full, R.col_packed, R.tickets)
  if (A_src) { if (nt) LPX_LAUNCH_ONE64(true, true); else LPX_LAUNCH_ONE64(false, true); }
  else { if (nt) LPX_LAUNCH_ONE64(true, false); else LPX_LAUNCH_ONE64(false, false); }
#undef LPX_LAUNCH_ONE64
}

#if LPX_FUSED
// blocks of 33..64 on the matrix cores (fused arithmetic only): one wave per SIMD, G workgroups per group of four
// 64-column sub-strips, 16-row tiles pulled from the sub-strip's ticket counter
static void launch_sweep64_mfma(const Buffers& B, const BlockRing& R, int m_local, int kmax, bool nt, const double* A_src,
                                hipStream_t s, int slots = 256, bool two_waves = false, int kmin = 33) {
  const int nstrips_full = (int)(B.ld / 512);
  const int ngroups = nstrips_full * 2;
  const int ntiles = m_local / 16;
  const int G = std::max(1, std::min(ntiles, slots / std::max(1, ngroups)));
  hipLaunchKernelGGL(k_pack_multipliers_mfma, dim3(ntiles), dim3(256), 0, s, R.col, R.mp, R.up, kmax, ntiles, R.col_packed,
                     R.tickets, nstrips_full * 8, R.clk, two_waves ? 1 : 0);
  if (two_waves) {   // k_sweep64_mfma2: groups of 128 columns, two workgroups per CU
    const int ng2 = nstrips_full * 4;
    const int G2 = std::max(1, std::min(ntiles, 2 * slots / std::max(1, ng2)));
    const dim3 grid2(ng2 * G2), block2(256);
#define LPX_LAUNCH_MFMA642(NT_, OOP_)                                                                               \
    hipLaunchKernelGGL((k_sweep64_mfma2<NT_, OOP_>), grid2, block2, 0, s, B.A, A_src, B.ld, m_local, R.prow, R.up, kmax, \
                       nstrips_full, R.col_packed, R.tickets, kmin)
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
                       int cus, int form, int* kernel_used, const FixSide* side) {
  int used = kSweepNone;
  if (kernel_used) *kernel_used = used;
  if (K < 1) return 0;
  if (cus <= 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  // The fix-up's chains read ring values only (and b, which the sweep leaves alone): with a side stream they are computed
  // BESIDE the sweep into the images of this ring half, and only their copy into the tableau follows the sweep.
  const bool side_fix = side && side->stream && R.fix_col && R.fix_row;
  if (side_fix) {
    const int gxf = (int)((std::max<int64_t>(m_local, B.ld) + 255) / 256);
    (void)hipStreamWaitEvent(side->stream, side->ready, 0);
    hipLaunchKernelGGL(k_block_fixup, dim3(gxf, (K + kFixChunk - 1) / kFixChunk, 3), dim3(256), 0, side->stream, B.A, B.ld, n, m_local, row0,
                       B.b, R.prow, R.col, R.col0, R.row0, R.mp, R.up, K, b_src ? b_src : B.b, (long long*)nullptr, R.fix_col, R.fix_row);
    (void)hipEventRecord(side->done, side->stream);
  }
  if (K <= 16) {
    // tiles of a few rows (k_update_tiles): up to K = 16 the sweep is HBM-bound and 16-row tiles stream best
    // (profiles/r01_sweep_rows.txt: larger tiles widen the set of DRAM pages in flight, -10 %)
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
    if (mfma) {
#if LPX_FUSED
      launch_sweep64_mfma(B, R, m_local, K, nt, A_src, s, cus, mfma2, whole ? 1 : 33);
#endif
      rows64 = 16;
    } else if (one) {
      launch_sweep64_one(B, R, m_local, K, nt, A_src, s, 2 * cus);
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
    if (m_local % 4 == 0 && B.ld >= 512 && (form == 1 || form == 2 || !pullable)) {
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
      launch_sweep_pull(B, R, m_local, K, nt, A_src, s, 2 * cus);
      used = kSweepPull;
      if (B.ld % 512 != 0) launch_sweep_k<32>(B, R, m_local, K, rows_per_wg, nt, A_src, s, 1);   // the partial last strip
      rows_per_wg = 4;   // (what lpx_state_get_info reports as the run length: one batch)
    } else if (!done) {
      launch_sweep_k<32>(B, R, m_local, K, rows_per_wg, nt, A_src, s);
      used = kSweepMulti;
    }
  }
  if (kernel_used) *kernel_used = used;
  if (after_sweep) (void)hipEventRecord(after_sweep, s);  // profiling: the sweep kernel alone
  // the clock probe: only the pack kernels of the pulled sweeps stamp in FRONT of a sweep; behind any other form the
  // fix-up must not pair its stamp with a front stamp of an older launch (lpx_state_info.sweep_clock_mhz then says 0)
  const bool probed = used == kSweepPull || used == kSweepPull64 || used == kSweepOne64 || used == kSweepMfma64 || used == kSweepMfma642;
  const int gx = (int)((std::max<int64_t>(m_local, B.ld) + 255) / 256);
  if (side_fix) {   // the chains were computed beside the sweep (launched above): only their copy into the tableau is left
    (void)hipStreamWaitEvent(s, side->done, 0);
    hipLaunchKernelGGL(k_block_fixup_scatter, dim3(gx, (K + kFixChunk - 1) / kFixChunk, 2), dim3(256), 0, s, B.A, B.ld, m_local, row0,
                       R.fix_col, R.fix_row, R.mp, R.up, K, probed ? R.clk : nullptr);
  } else {
    hipLaunchKernelGGL(k_block_fixup, dim3(gx, (K + kFixChunk - 1) / kFixChunk, 3), dim3(256), 0, s, B.A, B.ld, n, m_local, row0, B.b, R.prow, R.col,
                       R.col0, R.row0, R.mp, R.up, K, b_src ? b_src : B.b, probed ? R.clk : nullptr, nullptr, nullptr);
  }
  if (!probed && R.clk) (void)hipMemsetAsync(R.clk, 0, 256, s);
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
