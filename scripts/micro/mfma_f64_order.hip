// Does v_mfma_f64_16x16x4_f64 accumulate its four products as a chain of fused multiply-adds in k order, i.e. is
//     D[i][j] = fma(A[i][3], B[3][j], fma(A[i][2], B[2][j], fma(A[i][1], B[1][j], fma(A[i][0], B[0][j], C[i][j]))))
// bit for bit?  (Then a block's rank-K update in the fused-arithmetic mode could run on the matrix cores: the operation
// sequence per entry is exactly that chain.)  One wave, random operands incl. heavy cancellation, several candidate
// orders; also discovers the operand layout.   hipcc --offload-arch=gfx950 -O2 mfma_f64_order.hip -o mfma_f64_order
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_mfma(const double* A, const double* B, const double* C, double* D) {
  // operands as the hardware lays them out (found with this test): A[i][k] in lane k*16 + i, B[k][j] in lane k*16 + j,
  // C/D[4*r + lane/16][lane%16] in vgpr r
  const int lane = threadIdx.x;
  const double a = A[(lane % 16) * 4 + lane / 16];
  const double b = B[(lane / 16) * 16 + lane % 16];
  d4 c;
  for (int r = 0; r < 4; ++r) c[r] = C[(4 * r + lane / 16) * 16 + lane % 16];
  d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * r + lane / 16) * 16 + lane % 16] = d[r];
}

static double rnd() { return (double)rand() / RAND_MAX; }

int main() {
  double *dA, *dB, *dC, *dD;
  hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dC, 256 * 8); hipMalloc(&dD, 256 * 8);
  std::vector<double> A(64), B(64), C(256), D(256);
  const char* names[] = {"fma chain k=0..3", "fma chain k=3..0", "exact sum, one rounding", "unfused mul+add k=0..3", "pairs (0+1)+(2+3) fused"};
  long bad[5] = {0, 0, 0, 0, 0}, total = 0, bad_mode[4] = {0, 0, 0, 0}, bad_r[4] = {0, 0, 0, 0};
  srand(12345);
  for (int trial = 0; trial < 2000; ++trial) {
    const int mode = trial % 4;
    for (int i = 0; i < 64; ++i) {
      A[i] = (rnd() - 0.5) * (mode == 1 ? 1e8 : 2.0);
      B[i] = (rnd() - 0.5) * (mode == 2 ? 1e-8 : 2.0);
    }
    for (int i = 0; i < 256; ++i) C[i] = (rnd() - 0.5) * (mode == 3 ? 1e-6 : 2.0);
    if (mode == 3)   // cancellation: make the products nearly cancel C
      for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) C[i * 16 + j] = -(A[i * 4 + 0] * B[0 * 16 + j]) * (1.0 + 1e-13 * rnd());
    hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), 256 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        const double c = C[i * 16 + j];
        double r[5];
        r[0] = c; for (int k = 0; k < 4; ++k) r[0] = fma(A[i * 4 + k], B[k * 16 + j], r[0]);
        r[1] = c; for (int k = 3; k >= 0; --k) r[1] = fma(A[i * 4 + k], B[k * 16 + j], r[1]);
        { long double s = c; for (int k = 0; k < 4; ++k) s += (long double)A[i * 4 + k] * (long double)B[k * 16 + j]; r[2] = (double)s; }
        r[3] = c; for (int k = 0; k < 4; ++k) { volatile double p = A[i * 4 + k] * B[k * 16 + j]; r[3] = r[3] + p; }
        { double p01 = fma(A[i * 4 + 1], B[16 + j], A[i * 4] * B[j]); double p23 = fma(A[i * 4 + 3], B[48 + j], A[i * 4 + 2] * B[32 + j]); r[4] = c + (p01 + p23); }
        const double got = D[i * 16 + j];
        for (int q = 0; q < 5; ++q) bad[q] += memcmp(&got, &r[q], 8) != 0;
        if (memcmp(&got, &r[0], 8) != 0) { bad_mode[mode]++; bad_r[i % 4]++; if (bad[0] <= 6) printf("  trial %d i %d j %d got %a chain %a exact %a\n", trial, i, j, got, r[0], r[2]); }
        total++;
      }
  }
  printf("fma-chain mismatches by operand mode %ld %ld %ld %ld, by row%%4 %ld %ld %ld %ld\n", bad_mode[0], bad_mode[1], bad_mode[2], bad_mode[3], bad_r[0], bad_r[1], bad_r[2], bad_r[3]);
  for (int q = 0; q < 5; ++q) printf("%-28s mismatches %ld of %ld\n", names[q], bad[q], total);
  return 0;
}
