// What would a block of 128 pivots cost on the matrix cores?  k_sweep64_mfma2 (a block of 64: 16 MFMA groups per 16 x 64 tile)
// is not at the matrix pipe's floor for its CUs — its pulled copy structure alone takes 1.63 ms of its 1.75 on 192 CUs
// (EXPERIMENTS 00.2) — so twice the arithmetic per memory pass might hide behind that pass: a block of 128 pivots in
// 2.6-2.8 ms instead of 2 x 1.9 would be +35-45 % pivots/s where the sweep sets the pace (cfg4).  Timing experiment on the
// diagnostic copy of the kernel (csrc/variants/, -DLPX_MFMA_REPEAT=2: every tile runs its 16 MFMA groups and their LDS
// reads twice; results wrong; the A operands of the second half are NOT loaded: +8 KiB per tile from L2 in a real kernel,
// which the a_mask experiment of round 4 found free).  With -DLPX_MFMA_REPEAT=1 the result is compared bit for bit with
// the product kernel's.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DLPX_FUSED=1 -DLPX_WITH_VARIANTS -DLPX_MFMA_REPEAT=2
//          -I linear_programming_solver_amd/csrc scripts/micro/sweep_mfma_k128.hip -o scripts/micro/sweep_mfma_k128_r2
// Run:   sweep_mfma_k128_rN [m] [n] [reps] [CUs per XCD left to the sweep, 32 = no mask]
#include "lpx_kernels.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace lpxk::fused;
using lpxk::Buffers; using lpxk::BlockRing; using lpxk::LpxCtl;

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } \
  } while (0)

__global__ void k_fill(double* p, int64_t n, unsigned long long seed, double scale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    p[i] = ((double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5) * scale;
  }
}
__global__ void k_diff(const double* a, const double* b, int64_t n, unsigned long long* out) {
  unsigned long long bad = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    bad += __double_as_longlong(a[i]) != __double_as_longlong(b[i]);
  if (bad) atomicAdd(out, bad);
}

int main(int argc, char** argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 32768;
  const int n = argc > 2 ? atoi(argv[2]) : 16384;
  const int reps = argc > 3 ? atoi(argv[3]) : 10;
  const int per_xcd_keep = argc > 4 ? atoi(argv[4]) : 24;
  const int64_t ld = (n + 15) / 16 * 16, mp = (m + 1) / 2 * 2 + 2;
  if (m % 16 != 0 || ld % 512 != 0) { printf("m must be a multiple of 16, n of 512\n"); return 1; }
  const int KT = 64;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount, per_xcd = ncu / 8;
  hipStream_t st = 0;
  int cus = ncu;
  if (per_xcd_keep < per_xcd) {   // bit i = CU i / 8 of XCD i % 8 (profiles/r02_cu_mask.txt)
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    for (int cu = 0; cu < ncu; cu++) if (cu / 8 < per_xcd_keep) mask[cu / 32] |= 1u << (cu % 32);
    CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    cus = 8 * per_xcd_keep;
  }
  double *src, *dst, *ref, *prow, *col, *col_packed;
  LpxCtl* up;
  unsigned long long* bad;
  unsigned* tickets;
  long long* clk;
  CK(hipMalloc(&src, (size_t)m * ld * 8)); CK(hipMalloc(&dst, (size_t)m * ld * 8)); CK(hipMalloc(&ref, (size_t)m * ld * 8));
  CK(hipMalloc(&prow, (size_t)KT * ld * 8)); CK(hipMalloc(&col, (size_t)KT * mp * 8));
  CK(hipMalloc(&up, 128 * sizeof(LpxCtl))); CK(hipMalloc(&bad, 8));
  CK(hipMalloc(&col_packed, (size_t)(mp / 16 + 2) * 8192));
  CK(hipMalloc(&tickets, (size_t)(ld / 64 + 8) * 128));
  CK(hipMalloc(&clk, 256));
  hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, src, (int64_t)m * ld, 1ull, 2.0);
  hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, prow, (int64_t)KT * ld, 2ull, 0.25);
  hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, col, (int64_t)KT * mp, 3ull, -0.25);
  std::vector<LpxCtl> h(128);
  for (int s = 0; s < 128; ++s) { h[s] = LpxCtl{}; h[s].do_update = s < KT ? 1 : 0; h[s].e_cur = s; h[s].l = s; h[s].p = 1.0; }
  CK(hipMemcpy(up, h.data(), 128 * sizeof(LpxCtl), hipMemcpyHostToDevice));
  CK(hipDeviceSynchronize());
  Buffers B{}; B.ld = ld; B.fused = 1;
  BlockRing R{}; R.prow = prow; R.col = col; R.up = up; R.mp = mp; R.tickets = tickets; R.col_packed = col_packed; R.clk = clk;
  const int nstrips_full = (int)(ld / 512), ntiles = m / 16, ng2 = nstrips_full * 4;
  const int G2 = std::max(1, std::min(ntiles, 2 * cus / std::max(1, ng2)));
  printf("m %d n %d ld %lld  sweep on %d CUs%s  grid %d x 256  LPX_MFMA_REPEAT %d (one launch = the arithmetic of %d pivots)\n", m, n,
         (long long)ld, cus, st ? " (CU-masked stream)" : "", ng2 * G2, LPX_MFMA_REPEAT, 64 * LPX_MFMA_REPEAT);
  // the product kernel (with its pack kernel), for the time beside and as the reference of the bit comparison
  B.A = ref;
  auto product = [&] { launch_sweep64_mfma(B, R, m, KT, true, src, st, cus, true, 1); };
  auto diag = [&] {
    hipLaunchKernelGGL(k_pack_multipliers_mfma, dim3(ntiles), dim3(256), 0, st, R.col, R.mp, R.up, KT, ntiles, R.col_packed, R.tickets,
                       nstrips_full * 8, (long long*)nullptr, 1);
    hipLaunchKernelGGL((k_sweep64_mfma2_diag<true, true>), dim3(ng2 * G2), dim3(256), 0, st, dst, (const double*)src, ld, m,
                       (const double*)R.prow, (const LpxCtl*)R.up, KT, nstrips_full, (const double*)R.col_packed, R.tickets, -1);
  };
  auto time_ms = [&](auto&& f) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps;
  };
  const double el = (double)m * ld;
  for (int round = 0; round < 2; ++round) {
    const float tp = time_ms(product);
    const float td = time_ms(diag);
    printf("k_sweep64_mfma2 (product, 64 pivots)       %.3f ms  %.2f TB/s  %.1f k pivots/s\n", tp, 16 * el / tp * 1e-9, 64.0 / tp);
    printf("k_sweep64_mfma2_diag, MFMA groups x %d      %.3f ms  %.2f TB/s  %.1f k pivots/s (as a block of %d)\n", LPX_MFMA_REPEAT, td,
           16 * el / td * 1e-9, 64.0 * LPX_MFMA_REPEAT / td, 64 * LPX_MFMA_REPEAT);
  }
  int rc = 0;
  if (LPX_MFMA_REPEAT == 1 && LPX_MFMA_DIAG == 0) {
    CK(hipMemset(bad, 0, 8));
    hipLaunchKernelGGL(k_diff, dim3(2048), dim3(256), 0, 0, dst, ref, (int64_t)m * ld, bad);
    unsigned long long hb = 0;
    CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
    printf("entries of the diagnostic copy's result that differ from the product kernel's: %llu\n", hb);
    if (hb != 0) rc = 2;
  }
  return rc;
}
