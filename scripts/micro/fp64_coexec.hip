// Do the fp64 matrix pipe (v_mfma_f64_16x16x4_f64: 2048 flop, 64 cycles of a SIMD's MFMA pipe) and the fp64 vector pipe
// (v_fma_f64: 128 flop, 4 cycles of a SIMD's VALU) of gfx950 run SIDE BY SIDE, or do they share their multipliers?  Both
// peak at 32 flop per cycle and SIMD (78.6 TFLOP/s on the chip).  If they co-execute, a sweep whose waves are part MFMA
// workers, part v_fma_f64 workers (the ticket counters of k_sweep64_mfma2 / k_sweep64_one hand tiles to whoever asks) has
// twice the fp64 rate of either kind alone — and the block of 64 at cfg4, bound by the MFMA pipe on its 192 CUs,
// would be bound by HBM again.  No memory traffic here; one "unit" = 4 MFMAs = 64 v_fma_f64 = 256 pipe cycles of either kind.
//   mode 0: MFMAs only            mode 1: v_fma_f64 only
//   mode 2: both in EVERY wave (one MFMA, then sixteen independent v_fma_f64, four times per unit)
//   mode 3: waves 0..3 of a workgroup of 8 are MFMA workers, waves 4..7 v_fma_f64 workers (one of each per SIMD)
// Prints wall time, shader cycles per unit (s_memtime of wave 0) and the fp64 rate; co-execution <=> modes 2 / 3 take about
// the cycles of mode 0, not of modes 0 + 1.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off fp64_coexec.hip -o fp64_coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));

template <bool MFMA, bool VALU>
__device__ __forceinline__ void body(double* out, double a, double b, int iters, long long* cyc) {
  d4 acc[4];
  double x[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = d4{a + i, a - i, a * 0.5 + i, a * 0.25 - i};
#pragma unroll
  for (int c = 0; c < 16; ++c) x[c] = a + threadIdx.x * 1e-9 + c;
  const double ma = a * 1e-3 + (threadIdx.x & 15) * 1e-6, mb = b * 1e-3;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MFMA) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc[u], 0, 0, 0);
      if (MFMA && VALU) __builtin_amdgcn_sched_barrier(0);   // keep the program order: MFMA, sixteen v_fma_f64, MFMA, ...
      if (VALU) {
#pragma unroll
        for (int c = 0; c < 16; ++c) x[c] = __fma_rn(b, x[c], a);
      }
      if (MFMA && VALU) __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
#pragma unroll
  for (int c = 0; c < 16; ++c) s += x[c];
  if (s == 123.456) out[0] = s;
  if (cyc && blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int MODE>
__global__ __launch_bounds__(512) void k(double* out, double a, double b, int iters, long long* cyc) {
  if (MODE == 0) body<true, false>(out, a, b, iters, cyc);
  else if (MODE == 1) body<false, true>(out, a, b, iters, cyc);
  else if (MODE == 2) body<true, true>(out, a, b, iters, cyc);
  else {
    if ((threadIdx.x >> 6) < 4) body<true, false>(out, a, b, iters, cyc);   // uniform per wave
    else body<false, true>(out, a, b, iters, cyc);
  }
}

template <int MODE>
void run(const char* name, int threads, int blocks_per_cu, int ncu) {
  double* d;
  long long* dc;
  hipMalloc(&d, 8);
  hipMalloc(&dc, 8 * sizeof(long long));
  hipMemset(dc, 0, 8 * sizeof(long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 8000, grid = ncu * blocks_per_cu;
  hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(threads), 0, 0, d, 1.0, 0.999999, 200, (long long*)nullptr);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(threads), 0, 0, d, 1.0, 0.999999, iters, dc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  long long hc[8] = {0};
  hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
  const int waves = threads / 64;
  // flop of the whole launch: an MFMA wave does 4 x 2048 per unit, a v_fma_f64 wave 64 x 128 (the same), a wave of mode 2 both
  double wave_units = 0;
  if (MODE == 0 || MODE == 1) wave_units = waves;
  else if (MODE == 2) wave_units = 2.0 * waves;
  else wave_units = waves;   // 4 + 4 waves, one kind each
  const double flop = wave_units * 8192.0 * iters * grid;
  printf("%-34s %d waves/SIMD  %.3f ms  %6.1f TFLOP/s fp64   cycles per unit: wave0 %.1f", name, waves * blocks_per_cu / 4, ms,
         flop / (ms * 1e-3) / 1e12, (double)hc[0] / iters);
  if (MODE == 3) printf("  (MFMA wave) wave4 %.1f (v_fma wave)", (double)hc[4] / iters);
  printf("   clock %.2f GHz\n", (double)hc[0] / (ms * 1e-3) / 1e9);
  hipFree(d);
  hipFree(dc);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  printf("# %s, %d CUs; one unit = 4 v_mfma_f64_16x16x4 or 64 v_fma_f64 per wave = 256 pipe cycles of either kind\n", p.name, ncu);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("mode 0: MFMA only", 256, 1, ncu);
    run<1>("mode 1: v_fma_f64 only", 256, 1, ncu);
    run<2>("mode 2: both in every wave", 256, 1, ncu);
    run<0>("mode 0: MFMA only", 512, 1, ncu);
    run<1>("mode 1: v_fma_f64 only", 512, 1, ncu);
    run<2>("mode 2: both in every wave", 512, 1, ncu);
    run<3>("mode 3: 4 MFMA waves + 4 v_fma waves", 512, 1, ncu);
    run<3>("mode 3: 4 MFMA waves + 4 v_fma waves", 512, 2, ncu);
  }
  return 0;
}
