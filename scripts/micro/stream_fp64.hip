// Does the streaming ceiling of the part move when unfused fp64 work rides on the stream?  
// A grid-stride copy of a 4 GiB buffer (16 B per lane per access, nt loads and stores, out of place) with STEPS
// dependent steps  x <- x - c_s * p  per entry (one rounded product, one rounded difference: what the blocked sweep
// does per pending pivot), c_s wave-uniform kernel arguments (SGPR operands), p one double2 per thread: ~30 VGPRs,
// eight waves per SIMD — the arithmetic and the stream at the best occupancy the chip offers, nothing else in the way.
// Reports, per STEPS: time, TB/s moved, T fp64 op/s, and the shader clock the chip held (s_memtime ticks per
// s_memrealtime tick x 100 MHz, median over sampled workgroups).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scripts/micro/stream_fp64.hip -o scripts/micro/stream_fp64
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } \
  } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));
struct Coef { double c[32]; };

template <int STEPS, int UNROLL>
__global__ __launch_bounds__(256) void k_stream(const d2* __restrict__ src, d2* __restrict__ dst, int64_t n2, Coef cf,
                                                double p0, long long* stamps) {
  const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  const d2 p = d2{p0 + threadIdx.x * 1e-6, p0 - threadIdx.x * 1e-6};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += stride * UNROLL) {
    d2 x[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) x[u] = (i + u * stride < n2) ? __builtin_nontemporal_load(src + i + u * stride) : d2{0, 0};
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        x[u].x = __dsub_rn(x[u].x, __dmul_rn(cf.c[s], p.x));
        x[u].y = __dsub_rn(x[u].y, __dmul_rn(cf.c[s], p.y));
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (i + u * stride < n2) __builtin_nontemporal_store(x[u], dst + i + u * stride);
  }
  if (threadIdx.x == 0 && blockIdx.x % 64 == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamps[2 * (blockIdx.x / 64)] = __builtin_amdgcn_s_memtime() - t0;
    stamps[2 * (blockIdx.x / 64) + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

template <int STEPS>
static void run(const d2* src, d2* dst, int64_t n2, long long* d_stamps, int reps, int wgs_per_cu) {
  Coef cf;
  for (int s = 0; s < 32; ++s) cf.c[s] = 1e-3 * (s + 1) * ((s & 1) ? -1 : 1);
  const int grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_stream<STEPS, 4>), dim3(grid), dim3(256), 0, 0, src, dst, n2, cf, 0.37, d_stamps);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_stream<STEPS, 4>), dim3(grid), dim3(256), 0, 0, src, dst, n2, cf, 0.37, d_stamps);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  std::vector<long long> h(2 * (grid / 64));
  CK(hipMemcpy(h.data(), d_stamps, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> ghz;
  for (size_t k = 0; k + 1 < h.size(); k += 2)
    if (h[k + 1] > 0) ghz.push_back((double)h[k] / (double)h[k + 1] * 0.1);
  std::sort(ghz.begin(), ghz.end());
  const double bytes = 32.0 * n2;   // 16 B read + 16 B written per double2
  printf("steps %2d  wg/CU %d  %.3f ms  %.2f TB/s  %.1f T fp64 op/s  shader clock %.2f GHz\n", STEPS, wgs_per_cu, ms,
         bytes / ms * 1e-9, 4.0 * n2 * STEPS / ms * 1e-9, ghz.empty() ? 0.0 : ghz[ghz.size() / 2]);
}

int main(int argc, char** argv) {
  const int64_t gib = argc > 1 ? atoll(argv[1]) : 4;
  const int reps = argc > 2 ? atoi(argv[2]) : 10;
  const int64_t n2 = gib * (1ll << 30) / 16;
  d2 *src, *dst;
  long long* stamps;
  CK(hipMalloc(&src, n2 * 16)); CK(hipMalloc(&dst, n2 * 16)); CK(hipMalloc(&stamps, 8 * 2 * 64));
  CK(hipMemset(src, 0x3c, n2 * 16));   // 0x3c3c...: a small normal double (~1.5e-18), no denormal / NaN paths
  CK(hipMemset(stamps, 0, 8 * 2 * 64));
  for (int wpc : {8, 4}) {
    run<0>(src, dst, n2, stamps, reps, wpc);
    run<8>(src, dst, n2, stamps, reps, wpc);
    run<16>(src, dst, n2, stamps, reps, wpc);
    run<24>(src, dst, n2, stamps, reps, wpc);
    run<32>(src, dst, n2, stamps, reps, wpc);
  }
  return 0;
}
