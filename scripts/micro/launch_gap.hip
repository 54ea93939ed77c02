// What does a kernel boundary cost on a stream, by what sits between the two launches?  The decision launches of the blocked
// loop follow each other with ~20 us between them (profiles/r05_timeline_final_cfg4_cfg3.txt): kernel k, an event record, a
// wait for another stream's event, kernel k + 1.  Here: a chain of small kernels (64 workgroups of 256 threads, ~30 us each)
// that stamp the 100 MHz counter when their first wave starts and when their last workgroup leaves; the gap = start(k + 1) -
// end(k), by what the host enqueues between two launches — alone and beside a kernel of another stream that streams
// through HBM (the sweep's role).
// hipcc --offload-arch=gfx950 -O3 launch_gap.hip -o launch_gap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void work(long long* stamps, unsigned* counter, int k, int spin) {
  __shared__ int last;
  if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * k] = wall_clock64();
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(2);
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(&counter[k], 1u) == gridDim.x - 1;
  __syncthreads();
  if (last && threadIdx.x == 0) stamps[2 * k + 1] = wall_clock64();
}

// the decision kernel's launch shape: a kilobyte of launch parameters, and a last act that publishes the loop state to
// pinned HOST memory with a system-scope fence behind it (chain_publish)
struct Big { long long a[120]; };
__global__ __launch_bounds__(256) void work_big(long long* stamps, unsigned* counter, int k, int spin, Big big, long long* host) {
  __shared__ int last;
  if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * k] = wall_clock64() + (big.a[k % 120] & 0);
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(2);
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(&counter[k], 1u) == gridDim.x - 1;
  __syncthreads();
  if (last && threadIdx.x == 0) {
    if (host) {
      for (int q = 0; q < 12; q++) host[q] = t0 + q;
      __threadfence_system();
    }
    stamps[2 * k + 1] = wall_clock64();
  }
}

__global__ __launch_bounds__(256) void stream_copy(double2* dst, const double2* src, size_t n, int passes) {
  for (int p = 0; p < passes; ++p)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

enum Between { kNothing, kRecord, kRecordAndWait, kExtStop, kExtStopAndWait, kRecordDevScope };
static const char* kNames[] = {"nothing", "event record", "event record + wait(other stream's event)", "stop event of hipExtLaunchKernelGGL",
                               "stop event of hipExtLaunchKernelGGL + wait(other)", "event record (hipEventReleaseToDevice)"};

int main(int argc, char** argv) {
  const int N = 40, spin = 3000;   // 30 us per kernel
  long long* stamps; unsigned* counter;
  (void)hipMalloc(&stamps, 2 * N * sizeof(long long));
  (void)hipMalloc(&counter, N * sizeof(unsigned));
  // the decisions' stream: the same 8 CUs of every XCD the engine reserves (mask bit i = CU i / 8 of XCD i % 8)
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount, per_xcd = ncu / 8;
  std::vector<uint32_t> m_chain((ncu + 31) / 32, 0u), m_sweep((ncu + 31) / 32, 0u);
  for (int cu = 0; cu < ncu; cu++) {
    if (cu / 8 >= per_xcd - 8) m_chain[cu / 32] |= 1u << (cu % 32); else m_sweep[cu / 32] |= 1u << (cu % 32);
  }
  hipStream_t sc, ss, so;
  (void)hipExtStreamCreateWithCUMask(&sc, (uint32_t)m_chain.size(), m_chain.data());
  (void)hipExtStreamCreateWithCUMask(&ss, (uint32_t)m_sweep.size(), m_sweep.data());
  (void)hipStreamCreateWithFlags(&so, hipStreamNonBlocking);
  hipEvent_t ev[N], ev_dev[N], other;
  for (int k = 0; k < N; k++) {
    (void)hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&ev_dev[k], hipEventDisableTiming | hipEventReleaseToDevice);
  }
  (void)hipEventCreateWithFlags(&other, hipEventDisableTiming);
  (void)hipEventRecord(other, so);
  (void)hipStreamSynchronize(so);
  const size_t n = (size_t)1 << 27;   // 2 GiB each way
  double2 *src, *dst;
  (void)hipMalloc(&src, n * sizeof(double2)); (void)hipMalloc(&dst, n * sizeof(double2));
  (void)hipMemset(src, 0, n * sizeof(double2));
  for (int busy = 0; busy < 2; busy++) {
    for (int mode = 0; mode < 6; mode++) {
      (void)hipMemset(counter, 0, N * sizeof(unsigned));
      (void)hipMemset(stamps, 0, 2 * N * sizeof(long long));
      (void)hipDeviceSynchronize();
      if (busy) hipLaunchKernelGGL(stream_copy, dim3(192 * 8), dim3(256), 0, ss, dst, src, n, 6);   // ~4 ms of HBM streaming
      for (int k = 0; k < N; k++) {
        if (mode == kExtStop || mode == kExtStopAndWait)
          hipExtLaunchKernelGGL(work, dim3(64), dim3(256), 0, sc, nullptr, ev[k], 0, stamps, counter, k, spin);
        else
          hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, sc, stamps, counter, k, spin);
        if (mode == kRecord || mode == kRecordAndWait) (void)hipEventRecord(ev[k], sc);
        if (mode == kRecordDevScope) (void)hipEventRecord(ev_dev[k], sc);
        if (mode == kRecordAndWait || mode == kExtStopAndWait) (void)hipStreamWaitEvent(sc, other, 0);
      }
      (void)hipDeviceSynchronize();
      std::vector<long long> h(2 * N);
      (void)hipMemcpy(h.data(), stamps, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
      std::vector<double> gaps;
      for (int k = 5; k + 1 < N; k++) gaps.push_back((h[2 * (k + 1)] - h[2 * k + 1]) / 100.0);
      std::sort(gaps.begin(), gaps.end());
      printf("%-18s between two launches: %-52s gap median %6.2f us  (min %6.2f, max %6.2f)\n", busy ? "beside HBM stream" : "alone",
             kNames[mode], gaps[gaps.size() / 2], gaps.front(), gaps.back());
    }
  }
  // the launch shape of the decision kernel (beside the copy, stop event attached): big parameter block, host publication
  long long* host; (void)hipHostMalloc(&host, 4096, hipHostMallocDefault);
  long long* d_host; (void)hipHostGetDevicePointer((void**)&d_host, host, 0);
  Big big{};
  for (int shape = 0; shape < 3; shape++) {
    (void)hipMemset(counter, 0, N * sizeof(unsigned));
    (void)hipMemset(stamps, 0, 2 * N * sizeof(long long));
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(stream_copy, dim3(192 * 8), dim3(256), 0, ss, dst, src, n, 6);
    for (int k = 0; k < N; k++) {
      if (shape == 0) hipExtLaunchKernelGGL(work, dim3(64), dim3(256), 0, sc, nullptr, ev[k], 0, stamps, counter, k, spin);
      else hipExtLaunchKernelGGL(work_big, dim3(64), dim3(256), 0, sc, nullptr, ev[k], 0, stamps, counter, k, spin, big, shape == 2 ? d_host : nullptr);
      (void)hipStreamWaitEvent(sc, other, 0);
    }
    (void)hipDeviceSynchronize();
    std::vector<long long> h(2 * N);
    (void)hipMemcpy(h.data(), stamps, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<double> gaps;
    for (int k = 5; k + 1 < N; k++) gaps.push_back((h[2 * (k + 1)] - h[2 * k + 1]) / 100.0);
    std::sort(gaps.begin(), gaps.end());
    printf("beside HBM stream  launch shape: %-58s gap median %6.2f us  (min %6.2f, max %6.2f)\n",
           shape == 0 ? "three scalars" : shape == 1 ? "+ 960 bytes of parameters" : "+ 960 bytes of parameters + publication to host memory",
           gaps[gaps.size() / 2], gaps.front(), gaps.back());
  }
  return 0;
}
