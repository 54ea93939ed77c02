// The two steady-state sweep kernels alone on a synthetic ring: k_sweep32_steady (tableau batches parked in registers,
// round 2) against k_sweep32_dma (batches staged through LDS by LDS-DMA, round 3), out of place, full and partly
// filled blocks; results compared bit for bit with the generic kernel (k_update_multi<32>).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I linear_programming_solver_amd/csrc
//   scripts/micro/sweep_dma.hip -o scripts/micro/sweep_dma        (-DLPX_DMA_NS=3 ... to vary the ring depth)
// Run: sweep_dma [m] [n] [reps] [rows per workgroup, 0 = by size] [nt 0/1]
#include "lpx_kernels.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

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

// calibration of the box: plain copies of the same buffer (what "the streaming ceiling" is on THIS GPU today)
__global__ __launch_bounds__(256) void k_copy_flat(const lpxk::d2* __restrict__ a, lpxk::d2* __restrict__ b, int64_t n2) {
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (i < n2) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i);
}
// k_update's shape: workgroup = 2 rows x 512 columns, consecutive workgroups sweep the buffer in address order
__global__ __launch_bounds__(256) void k_copy_rows(const double* __restrict__ a, double* __restrict__ b, int64_t ld, int nstrips) {
  const int strip = blockIdx.x % nstrips, tile = blockIdx.x / nstrips;
  const int64_t o = (int64_t)tile * 2 * ld + strip * 512 + 2 * threadIdx.x;
  const lpxk::d2 x0 = __builtin_nontemporal_load(reinterpret_cast<const lpxk::d2*>(a + o));
  const lpxk::d2 x1 = __builtin_nontemporal_load(reinterpret_cast<const lpxk::d2*>(a + o + ld));
  __builtin_nontemporal_store(x0, reinterpret_cast<lpxk::d2*>(b + o));
  __builtin_nontemporal_store(x1, reinterpret_cast<lpxk::d2*>(b + o + ld));
}

using namespace lpxk;

template <typename F>
static float time_ms(F f, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) f();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char** argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 32768;
  const int n = argc > 2 ? atoi(argv[2]) : 16384;
  const int reps = argc > 3 ? atoi(argv[3]) : 10;
  const int rows_arg = argc > 4 ? atoi(argv[4]) : 0;
  const bool nt = argc > 5 ? atoi(argv[5]) != 0 : true;
  const int cus = argc > 6 ? atoi(argv[6]) : 256;
  const int64_t ld = (n + 15) / 16 * 16, mp = (m + 1) / 2 * 2 + 2;
  const int KT = 32;
  double *src, *dst, *ref, *prow, *col, *zeros;
  LpxCtl* up;
  unsigned long long* bad;
  CK(hipMalloc(&src, (size_t)m * ld * 8)); CK(hipMalloc(&dst, (size_t)m * ld * 8)); CK(hipMalloc(&ref, (size_t)m * ld * 8));
  CK(hipMalloc(&prow, (size_t)KT * ld * 8)); CK(hipMalloc(&col, (size_t)KT * mp * 8));
  CK(hipMalloc(&up, 64 * sizeof(LpxCtl))); CK(hipMalloc(&bad, 8)); CK(hipMalloc(&zeros, 256));
  CK(hipMemset(zeros, 0, 256));
  double* col_packed;
  CK(hipMalloc(&col_packed, (size_t)(mp / 4 + 1) * 1024));
  unsigned* tickets;
  CK(hipMalloc(&tickets, (size_t)(ld / 128 + 4) * 128));
  hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, src, (int64_t)m * ld, 1ull, 2.0);
  hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, prow, (int64_t)KT * ld, 2ull, 0.25);
  hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, col, (int64_t)KT * mp, 3ull, -0.25);
  CK(hipDeviceSynchronize());

  Buffers B{}; B.ld = ld;
  BlockRing R{}; R.prow = prow; R.col = col; R.up = up; R.mp = mp; R.zeros = zeros; R.tickets = tickets; R.col_packed = col_packed;
  const int nstrips_full = (int)(ld / 512);
  int rows = rows_arg > 0 ? rows_arg / 4 * 4 : choose_pipe_rows(m, nstrips_full, 2 * cus, 48);
  const int rows_generic = std::max(64, choose_sweep_rows(m, ld, 32, cus) / 64 * 64);
  printf("m %d n %d ld %lld  rows/wg %d  nt %d  LDS ring slots %d\n", m, n, (long long)ld, rows, (int)nt, kDmaNS);
  const double el = (double)m * ld;
  int rc = 0;
  if (ld % 512 == 0 && m % 2 == 0) {
    const float tm = time_ms([&] { CK(hipMemcpyAsync(dst, src, (size_t)m * ld * 8, hipMemcpyDeviceToDevice, 0)); }, reps);
    const float tf = time_ms([&] { hipLaunchKernelGGL(k_copy_flat, dim3((unsigned)(el / 2 / 256)), dim3(256), 0, 0,
                                                      (const d2*)src, (d2*)dst, (int64_t)(el / 2)); }, reps);
    const float tr = time_ms([&] { hipLaunchKernelGGL(k_copy_rows, dim3((unsigned)(m / 2 * (ld / 512))), dim3(256), 0, 0,
                                                      src, dst, ld, (int)(ld / 512)); }, reps);
    printf("this box, plain copies of the tableau: hipMemcpy D2D %.3f ms %.2f TB/s | flat nt copy %.3f ms %.2f TB/s | "
           "2-row x 512-column tiles %.3f ms %.2f TB/s\n", tm, 16 * el / tm * 1e-9, tf, 16 * el / tf * 1e-9, tr, 16 * el / tr * 1e-9);
  }
  for (int np : {32, 20, 0}) {
    std::vector<LpxCtl> h(64);
    for (int s = 0; s < 64; ++s) { h[s] = LpxCtl{}; h[s].do_update = s < np ? 1 : 0; h[s].e_cur = s; h[s].l = s; h[s].p = 1.0; }
    CK(hipMemcpy(up, h.data(), 64 * sizeof(LpxCtl), hipMemcpyHostToDevice));
    B.A = ref;
    launch_sweep_k<32>(B, R, m, KT, rows_generic, nt, src, 0);
    CK(hipDeviceSynchronize());
    B.A = dst;
    for (int form = 0; form < 6; ++form) {
      if (form == 2 || form == 1) continue;
      CK(hipMemset(dst, 0xff, (size_t)m * ld * 8));
      const float t = time_ms([&] {
        if (form == 0) launch_sweep_steady(B, R, m, KT, std::max(48, rows / 48 * 48), nt, src, 0);
        else if (form == 1) launch_sweep_dma(B, R, m, KT, rows, nt, src, 0, 2 * cus, 0);
        else if (form == 2) launch_sweep_dma(B, R, m, KT, 0, nt, src, 0, 2 * cus, 0);
        else if (form == 3) launch_sweep_dma(B, R, m, KT, rows, nt, src, 0, 2 * cus, 1);
        else if (form == 4) launch_sweep_dma(B, R, m, KT, 0, nt, src, 0, 2 * cus, 1);
        else launch_sweep_pull(B, R, m, KT, nt, src, 0, 2 * cus);
      }, np == 32 ? reps : 2);
      CK(hipMemset(bad, 0, 8));
      hipLaunchKernelGGL(k_diff, dim3(2048), dim3(256), 0, 0, dst, ref, (int64_t)m * ld, bad);
      unsigned long long hb = 0;
      CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
      // the kernels take the full strips only: columns [nstrips_full * 512, ld) stay 0xff and are counted
      const unsigned long long untouched = (unsigned long long)m * (ld - (int64_t)nstrips_full * 512);
      printf("np %2d  %-22s %.3f ms  %.2f TB/s  %.1f T fp64 op/s   mismatches %llu (of which outside the full strips: %llu)\n",
             np, form == 0 ? "k_sweep32_steady" : form == 1 ? "k_sweep32_dma runs" : form == 2 ? "k_sweep32_dma ordered" :
             form == 3 ? "dma runs, xcd remap" : form == 4 ? "dma ordered, xcd remap" : "k_sweep32_pull", t, 16 * el / t * 1e-9, 2 * el * np / t * 1e-9, hb,
             untouched);
      if (hb != untouched) rc = 2;
    }
  }
  return rc;
}
