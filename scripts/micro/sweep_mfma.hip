// The MFMA sweeps of blocks of 64 pivots (fused arithmetic) alone on a synthetic ring: k_sweep64_mfma2 (two waves per
// SIMD, B operands in LDS, buffer addressing) against k_sweep64_mfma (one wave per SIMD), out of place, on all CUs or on
// a CU-masked stream (what the sweep gets beside the decision kernel); results compared bit for bit between the two.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DLPX_FUSED=1 -I linear_programming_solver_amd/csrc
//   scripts/micro/sweep_mfma.hip -o scripts/micro/sweep_mfma      (-DLPX_MFMA_DIAG=bits: timing experiments, results wrong:
//   2 no stores, 4 no tile loads, 8 no MFMAs, 16 no A loads)
// Run: sweep_mfma [m] [n] [reps] [CUs per XCD left to the sweep, 32 = no mask] [nt 0/1] [row pitch in doubles, 0 = n rounded up to 16]
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
__global__ __launch_bounds__(256) void k_copy_flat(const d2* __restrict__ a, d2* __restrict__ b, int64_t n2) {
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (i < n2) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i);
}

template <typename F>
static float time_ms(F&& f, int reps, hipStream_t s) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); f();
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1, s));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char** argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 32768;
  const int n = argc > 2 ? atoi(argv[2]) : 16384;
  const int reps = argc > 3 ? atoi(argv[3]) : 10;
  const int per_xcd_keep = argc > 4 ? atoi(argv[4]) : 32;
  const bool nt = argc > 5 ? atoi(argv[5]) != 0 : true;
  const int64_t ld = argc > 6 && atoi(argv[6]) > 0 ? atoi(argv[6]) : (n + 15) / 16 * 16, mp = (m + 1) / 2 * 2 + 2;
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
  double *src, *dst, *ref, *prow, *col;
  LpxCtl* up;
  unsigned long long* bad;
  CK(hipMalloc(&src, (size_t)m * ld * 8)); CK(hipMalloc(&dst, (size_t)m * ld * 8)); CK(hipMalloc(&ref, (size_t)m * ld * 8));
  CK(hipMalloc(&prow, (size_t)KT * ld * 8)); CK(hipMalloc(&col, (size_t)KT * mp * 8));
  CK(hipMalloc(&up, 128 * sizeof(LpxCtl))); CK(hipMalloc(&bad, 8));
  double* col_packed;
  CK(hipMalloc(&col_packed, (size_t)(mp / 16 + 2) * 8192));
  unsigned* tickets;
  CK(hipMalloc(&tickets, (size_t)(ld / 64 + 8) * 128));
  long long* clk;
  CK(hipMalloc(&clk, 256));
  hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, src, (int64_t)m * ld, 1ull, 2.0);
  hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, prow, (int64_t)KT * ld, 2ull, 0.25);
  hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, col, (int64_t)KT * mp, 3ull, -0.25);
  CK(hipDeviceSynchronize());
  Buffers B{}; B.ld = ld; B.fused = 1;
  BlockRing R{}; R.prow = prow; R.col = col; R.up = up; R.mp = mp; R.tickets = tickets; R.col_packed = col_packed; R.clk = clk;
  const double el = (double)m * ld;
  printf("m %d n %d ld %lld  nt %d  sweep on %d CUs%s  LPX_MFMA_DIAG %d\n", m, n, (long long)ld, (int)nt, cus,
         st ? " (CU-masked stream)" : "", LPX_MFMA_DIAG);
  const float tf = time_ms([&] { hipLaunchKernelGGL(k_copy_flat, dim3((unsigned)(el / 2 / 256)), dim3(256), 0, st,
                                                    (const d2*)src, (d2*)dst, (int64_t)(el / 2)); }, reps, st);
  printf("this box, flat nt copy of the tableau on that stream: %.3f ms  %.2f TB/s\n", tf, 16 * el / tf * 1e-9);
  int rc = 0;
  for (int np : {64, 40}) {
    std::vector<LpxCtl> h(128);
    for (int s = 0; s < 128; ++s) { h[s] = LpxCtl{}; h[s].do_update = s < np ? 1 : 0; h[s].e_cur = s; h[s].l = s; h[s].p = 1.0; }
    CK(hipMemcpy(up, h.data(), 128 * sizeof(LpxCtl), hipMemcpyHostToDevice));
    B.A = ref;
    launch_sweep64_mfma(B, R, m, KT, nt, src, st, cus, false);
    CK(hipStreamSynchronize(st));
    B.A = dst;
    for (int two = 2; two >= 1; --two) {
      CK(hipMemset(dst, 0xff, (size_t)m * ld * 8));
      const float t = time_ms([&] { launch_sweep64_mfma(B, R, m, KT, nt, src, st, cus, two == 2); }, np == 64 ? reps : 2, st);
      CK(hipMemset(bad, 0, 8));
      hipLaunchKernelGGL(k_diff, dim3(2048), dim3(256), 0, 0, dst, ref, (int64_t)m * ld, bad);
      unsigned long long hb = 0;
      CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
      printf("np %2d  %-16s %.3f ms (with its pack kernel)  %.2f TB/s   entries that differ from k_sweep64_mfma's: %llu\n", np,
             two == 2 ? "k_sweep64_mfma2" : "k_sweep64_mfma", t, 16 * el / t * 1e-9, hb);
      if (hb != 0 && LPX_MFMA_DIAG == 0) rc = 2;
    }
  }
  return rc;
}
