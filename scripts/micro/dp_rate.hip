// Micro-benchmark: issue rate of v_mul_f64 / v_add_f64 / v_fma_f64 on gfx950 (no memory traffic).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off dp_rate.hip -o dp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE, int CH>
__global__ __launch_bounds__(256) void k(double* out, double a, double b, int iters) {
  double x[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) x[c] = a + threadIdx.x * 1e-9 + c;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (MODE == 0) x[c] = __dsub_rn(x[c], __dmul_rn(b, x[c]));        // dependent mul -> sub (2 instr)
        else if (MODE == 1) x[c] = __fma_rn(b, x[c], a);                  // fma (1 instr)
        else if (MODE == 2) x[c] = __dmul_rn(x[c], b);                    // mul only
        else x[c] = __dadd_rn(x[c], b);                                   // add only
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += x[c];
  if (s == 123.456) out[0] = s;
}

template <int MODE, int CH>
void run(const char* name, int blocks_per_cu) {
  double* d; hipMalloc(&d, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL((k<MODE, CH>), dim3(grid), dim3(256), 0, 0, d, 1.0, 0.999999, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, CH>), dim3(grid), dim3(256), 0, 0, d, 1.0, 0.999999, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_thread = (double)iters * 8 * CH * (MODE == 0 ? 2 : 1);
  const double lane_ops = instr_per_thread * 256.0 * grid;
  printf("%-10s CH=%d waves/SIMD=%d  %.3f ms  %.2f T lane-instr/s  (%.1f cycles per wave-instr per SIMD at 2.4 GHz)\n", name, CH,
         blocks_per_cu, ms, lane_ops / ms / 1e9, 2.4e9 * (ms * 1e-3) / (instr_per_thread * blocks_per_cu));
  hipFree(d);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0, 8>("mul+sub", w);
    run<1, 8>("fma", w);
    run<2, 8>("mul", w);
    run<3, 8>("add", w);
  }
  run<0, 2>("mul+sub", 1);
  run<0, 4>("mul+sub", 1);
  run<0, 16>("mul+sub", 1);
  return 0;
}
