// Which CUs / XCDs does a CU-masked stream really use on this runtime?  Launches a census kernel (one workgroup per
// CU-slot, each records HW_REG_XCC_ID and HW_REG_HW_ID) on streams created with different hipExtStreamCreateWithCUMask
// layouts and prints the XCD histogram.
// hipcc --offload-arch=gfx950 -O3 cu_mask.hip -o cu_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <set>

__global__ __launch_bounds__(256) void census(unsigned* out, int spin) {
  unsigned x, h;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
  // keep the workgroup resident for a while so that the grid spreads over every CU the stream may use
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = x & 15u; out[2 * blockIdx.x + 1] = h; }
}

static void run(const char* name, const std::vector<uint32_t>& mask, int grid) {
  hipStream_t s;
  hipError_t e = mask.empty() ? hipStreamCreateWithFlags(&s, hipStreamNonBlocking)
                              : hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
  if (e != hipSuccess) { printf("%-28s stream creation failed: %s\n", name, hipGetErrorString(e)); (void)hipGetLastError(); return; }
  unsigned* d; (void)hipMalloc(&d, 2 * grid * sizeof(unsigned));
  hipLaunchKernelGGL(census, dim3(grid), dim3(256), 0, s, d, 20000 /* 200 us */);
  (void)hipStreamSynchronize(s);
  std::vector<unsigned> h(2 * grid);
  (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  int per_xcd[16] = {0};
  std::set<unsigned> cus;
  for (int b = 0; b < grid; b++) { per_xcd[h[2 * b] & 15]++; cus.insert((h[2 * b] << 16) | ((h[2 * b + 1] >> 8) & 0xff) | ((h[2 * b + 1] >> 13) & 7) << 8); }
  printf("%-28s grid %4d  workgroups per XCD:", name, grid);
  for (int x = 0; x < 8; x++) printf(" %3d", per_xcd[x]);
  printf("   distinct (xcd,se,cu) = %zu\n", cus.size());
  (void)hipFree(d); (void)hipStreamDestroy(s);
}

int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  printf("device %s, %d CUs\n", p.name, ncu);
  const int nw = (ncu + 31) / 32;
  std::vector<uint32_t> none;
  run("no mask", none, 256);
  { std::vector<uint32_t> m(nw, 0); for (int cu = 0; cu < ncu; cu++) if (cu % 8 == 7) m[cu / 32] |= 1u << (cu % 32); run("bits cu%8==7 (32 bits)", m, 64); }
  { std::vector<uint32_t> m(nw, 0); for (int cu = 0; cu < ncu; cu++) if (cu % 8 != 7) m[cu / 32] |= 1u << (cu % 32); run("bits cu%8!=7 (224 bits)", m, 512); }
  { std::vector<uint32_t> m(nw, 0); for (int cu = 224; cu < ncu; cu++) m[cu / 32] |= 1u << (cu % 32); run("bits 224..255 (32 bits)", m, 64); }
  { std::vector<uint32_t> m(nw, 0); for (int cu = 0; cu < 224; cu++) m[cu / 32] |= 1u << (cu % 32); run("bits 0..223 (224 bits)", m, 512); }
  { std::vector<uint32_t> m(nw, 0); for (int cu = 0; cu < 32; cu++) m[cu / 32] |= 1u << (cu % 32); run("bits 0..31 (32 bits)", m, 64); }
  { std::vector<uint32_t> m(nw, 0); for (int cu = 0; cu < ncu; cu++) if (cu % 8 == 0) m[cu / 32] |= 1u << (cu % 32); run("bits cu%8==0 (32 bits)", m, 64); }
  { std::vector<uint32_t> m(nw, 0); m[0] = 0xff; run("bits 0..7 (8 bits)", m, 64); }
  return 0;
}
