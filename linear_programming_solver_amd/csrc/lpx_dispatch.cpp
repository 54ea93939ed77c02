// The host engine's launch wrappers: lpx_kernels.hip is compiled twice — lpxk::plain (one rounding per reference
// operation) and lpxk::fused (every update x - c*r as one v_fma_f64, LPX_OPT_FUSED) — and a handle's Buffers say which
// of the two its launches take.  Kernels without arithmetic (column fill / drop, transpose) and the geometry helpers
// exist in both and are taken from lpxk::plain.
#include "lpx_kernels.h"

#include <algorithm>

namespace lpxk {

#define LPX_PICK(B, f) ((B).fused ? fused::f : plain::f)

void launch_entering(const Buffers& B, int n, hipStream_t s, const LoopStart& start) { LPX_PICK(B, launch_entering)(B, n, s, start); }
void launch_entering_dantzig(const Buffers& B, int n, bool seed, hipStream_t s, const LoopStart& start) {
  LPX_PICK(B, launch_entering_dantzig)(B, n, seed, s, start);
}
void launch_ratio_gather(const Buffers& B, int m_local, int row0, const Geometry& g, int forced_e, hipStream_t s) {
  LPX_PICK(B, launch_ratio_gather)(B, m_local, row0, g, forced_e, s);
}
void launch_reduce_partials(const Buffers& B, const Geometry& g, hipStream_t s) {
  LPX_PICK(B, launch_reduce_partials)(B, g, s);
}
void launch_select_pivot(const Buffers& B, int n, int m_global, const Geometry& g, int forced_e, int forced_l,
                         hipStream_t s) {
  LPX_PICK(B, launch_select_pivot)(B, n, m_global, g, forced_e, forced_l, s);
}
void launch_update(const Buffers& B, int m_local, int n, int row0, const Geometry& g, bool nontemporal,
                   const double* prow, const LpxCtl* up, double* A_out, double* b_out, hipStream_t s) {
  LPX_PICK(B, launch_update)(B, m_local, n, row0, g, nontemporal, prow, up, A_out, b_out, s);
}
void launch_propose(const Buffers& B, int n, int row0, int m_local, const Geometry& g, double* d_candidate,
                    hipStream_t s) {
  LPX_PICK(B, launch_propose)(B, n, row0, m_local, g, d_candidate, s);
}
void launch_commit(const Buffers& B, int n, int m_global, const double* d_gathered, int nranks, double* prow,
                   LpxCtl* up, int up_parity, hipStream_t s) {
  LPX_PICK(B, launch_commit)(B, n, m_global, d_gathered, nranks, prow, up, up_parity, s);
}
void launch_peek(const Buffers& B, int n, int m_local, int row0, const double* prow_t, const double* col_t,
                 double* col_next, const LpxCtl* pend, double* d_candidate, hipStream_t s) {
  LPX_PICK(B, launch_peek)(B, n, m_local, row0, prow_t, col_t, col_next, pend, d_candidate, s);
}
void launch_block_peek(const Buffers& B, const BlockRing& R, int n, int m_local, int row0, int np, double* d_candidate,
                       hipStream_t s) {
  LPX_PICK(B, launch_block_peek)(B, R, n, m_local, row0, np, d_candidate, s);
}
void launch_block_decide(const Buffers& B, const BlockRing& R, int n, int m_global, const double* d_gathered, int nranks,
                         int slot, hipStream_t s) {
  LPX_PICK(B, launch_block_decide)(B, R, n, m_global, d_gathered, nranks, slot, s);
}
int launch_block_chain(const Buffers& B, const BlockRing& R, int n, int m, int nb, int half, int old_half, int n_old,
                       int b_from_tableau, int seq, int dantzig, int wgs, int fences, bool trace, LpxCtl* host_snap,
                       hipStream_t s, const MgPeers* mg, hipEvent_t stop) {
  return LPX_PICK(B, launch_block_chain)(B, R, n, m, nb, half, old_half, n_old, b_from_tableau, seq, dantzig, wgs, fences, trace,
                                  host_snap, s, mg, stop);
}
// both sets: the arithmetic mode is an option of the handle and may be set after its ring has been built
void preload_block_kernels(const Buffers& B, const BlockRing& R, hipStream_t s) {
  plain::preload_block_kernels(B, R, s);
  fused::preload_block_kernels(B, R, s);
}
unsigned* sweep_fail_word(const BlockRing& R, int64_t ld) { return plain::sweep_fail_word(R, ld); }
int64_t sweep_ticket_slots(int64_t ld) { return plain::sweep_ticket_slots(ld); }
// the grid of the persistent decision kernel must be resident in either mode
int chain_blocks_per_cu() { return std::min(plain::chain_blocks_per_cu(), fused::chain_blocks_per_cu()); }
const char* sweep_kernel_name(int code) { return plain::sweep_kernel_name(code); }
int launch_block_sweep(const Buffers& B, const BlockRing& R, int n, int m_local, int row0, int K, int rows_per_wg,
                       bool nt, hipStream_t s, const double* A_src, const double* b_src, hipEvent_t after_sweep, int cus,
                       int form, int* kernel_used, const FixSide* side, hipEvent_t stop, hipEvent_t before_sweep) {
  return LPX_PICK(B, launch_block_sweep)(B, R, n, m_local, row0, K, rows_per_wg, nt, s, A_src, b_src, after_sweep, cus,
                                         form, kernel_used, side, stop, before_sweep);
}
void launch_fill_column(double* A, int64_t ld, int m, int col, double value, hipStream_t s) {
  plain::launch_fill_column(A, ld, m, col, value, s);
}
void launch_drop_column(double* A, int64_t ld, int m, int n_old, int col, hipStream_t s) {
  plain::launch_drop_column(A, ld, m, n_old, col, s);
}
void launch_restore_objective(const Buffers& B, int n, const RestoreEntry* d_entries, int n_entries, hipStream_t s) {
  LPX_PICK(B, launch_restore_objective)(B, n, d_entries, n_entries, s);
}
void launch_checksum(const Buffers& B, int m_local, int n, int row0, unsigned long long* d_out3, hipStream_t s) {
  plain::launch_checksum(B, m_local, n, row0, d_out3, s);
}
void launch_transpose(const double* dA, int64_t lda, double* dAt, int64_t ldat, int m, int n, hipStream_t s) {
  plain::launch_transpose(dA, lda, dAt, ldat, m, n, s);
}

}  // namespace lpxk
