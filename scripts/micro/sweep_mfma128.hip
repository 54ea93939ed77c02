// k_sweep128_mfma (csrc/variants/: a block of up to 128 pivots per pass on the matrix cores, EXPERIMENTS 000.55) alone on a
// synthetic ring, out of place, against TWO passes of the product's k_sweep64_mfma2 (pivots 0..63 out of place, 64..127 in
// place on the result): time and a bit-for-bit comparison of every entry, for full and partly filled blocks.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DLPX_FUSED=1 -DLPX_WITH_VARIANTS
//          -I linear_programming_solver_amd/csrc scripts/micro/sweep_mfma128.hip -o scripts/micro/sweep_mfma128
// Run:   sweep_mfma128 [m] [n] [reps] [CUs per XCD left to the sweep, 32 = no mask] [row ranges per sub-strip]
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
  const int nranges = argc > 5 ? atoi(argv[5]) : 6;
  const int64_t ld = (n + 15) / 16 * 16, mp = (m + 1) / 2 * 2 + 2;
  if (m % 16 != 0 || ld % 512 != 0 || nranges < 1 || 16 * ld * 8 + 1024 >= ((int64_t)1 << 32)) { printf("m must be a multiple of 16, n of 512\n"); return 1; }
  const int KT = 128;
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
  const int nsub = (int)(ld / 64), rowblocks = m / 16, nchunks = nsub * nranges;
  double *src, *dst, *ref, *prow, *col, *col_packed, *col_packed128;
  LpxCtl* up;
  unsigned long long* bad;
  unsigned *tickets, *tickets128;
  long long* clk;
  CK(hipMalloc(&src, (size_t)m * ld * 8)); CK(hipMalloc(&dst, (size_t)m * ld * 8)); CK(hipMalloc(&ref, (size_t)m * ld * 8));
  CK(hipMalloc(&prow, (size_t)KT * ld * 8)); CK(hipMalloc(&col, (size_t)KT * mp * 8));
  CK(hipMalloc(&up, 128 * sizeof(LpxCtl))); CK(hipMalloc(&bad, 8));
  CK(hipMalloc(&col_packed, (size_t)(mp / 16 + 2) * 8192));
  CK(hipMalloc(&col_packed128, (size_t)(rowblocks + 2) * 16384));
  CK(hipMalloc(&tickets, (size_t)(ld / 64 + 8) * 128));
  CK(hipMalloc(&tickets128, (size_t)(nchunks + 2) * 128));
  CK(hipMalloc(&clk, 256));
  CK(hipMemset(clk, 0, 256));
  hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, src, (int64_t)m * ld, 1ull, 2.0);
  hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, prow, (int64_t)KT * ld, 2ull, 0.25);
  hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, col, (int64_t)KT * mp, 3ull, -0.25);
  CK(hipDeviceSynchronize());
  Buffers B{}; B.ld = ld; B.fused = 1;
  BlockRing R{}; R.prow = prow; R.col = col; R.up = up; R.mp = mp; R.tickets = tickets; R.col_packed = col_packed; R.clk = clk;
  BlockRing R2 = R; R2.prow = prow + 64 * ld; R2.col = col + 64 * mp; R2.up = up + 64;   // pivots 64..127 as a ring of their own
  const double el = (double)m * ld;
  printf("m %d n %d ld %lld  sweep on %d CUs%s  %d sub-strips x %d row ranges = %d chunks, grid %d x 256\n", m, n, (long long)ld, cus,
         st ? " (CU-masked stream)" : "", nsub, nranges, nchunks, 2 * cus);
  auto time_ms = [&](auto&& f, int nrep) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < nrep; ++i) f();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / nrep;
  };
  int rc = 0;
  for (int np : {128, 100, 64, 40}) {
    std::vector<LpxCtl> h(128);
    for (int s = 0; s < 128; ++s) { h[s] = LpxCtl{}; h[s].do_update = s < np ? 1 : 0; h[s].e_cur = s; h[s].l = s; h[s].p = 1.0; }
    CK(hipMemcpy(up, h.data(), 128 * sizeof(LpxCtl), hipMemcpyHostToDevice));
    // reference: two passes of the product kernel (a pass of no valid pivots copies / leaves the tableau)
    auto two_passes = [&] {
      B.A = ref;
      launch_sweep64_mfma(B, R, m, 64, true, src, st, cus, true, 1);
      if (np > 64) launch_sweep64_mfma(B, R2, m, 64, true, nullptr, st, cus, true, 1);
    };
    auto one_pass = [&] {
      hipLaunchKernelGGL(k_pack_multipliers_mfma128, dim3(rowblocks), dim3(256), 0, st, (const double*)col, mp, (const LpxCtl*)up, KT,
                         rowblocks, col_packed128, tickets128, nchunks + 1);
      hipLaunchKernelGGL((k_sweep128_mfma<true, true>), dim3(2 * cus), dim3(256), 0, st, dst, (const double*)src, ld, m,
                         (const double*)prow, (const LpxCtl*)up, KT, nsub, (const double*)col_packed128, tickets128, nranges);
    };
    const int nrep = np == 128 ? reps : 2;
    CK(hipMemset(dst, 0xff, (size_t)m * ld * 8));
    const float t2 = time_ms(two_passes, nrep);
    const float t1 = time_ms(one_pass, nrep);
    CK(hipMemset(bad, 0, 8));
    hipLaunchKernelGGL(k_diff, dim3(2048), dim3(256), 0, 0, dst, ref, (int64_t)m * ld, bad);
    unsigned long long hb = 0;
    CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
    printf("np %3d  two passes of k_sweep64_mfma2 %.3f ms (%.1f k pivots/s)   k_sweep128_mfma %.3f ms (%.1f k pivots/s, %.2f TB/s)   "
           "entries that differ: %llu\n", np, t2, np / t2, t1, np / t1, 16 * el / t1 * 1e-9, hb);
    if (hb != 0) rc = 2;
  }
  return rc;
}
