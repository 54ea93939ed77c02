#include <hip/hip_runtime.h>
#include <cstdio>
int main() {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
  printf("name %s CUs %d sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu regsPerBlock %d clock %d kHz l2 %d\n", p.name,
         p.multiProcessorCount, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock, p.clockRate, p.l2CacheSize);
  int v = 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, 0) == hipSuccess) printf("attr MaxSharedMemoryPerBlock %d\n", v);
  return 0;
}
