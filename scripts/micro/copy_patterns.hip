// What a PERSISTENT workgroup costs a streaming copy on MI355X, by access pattern (round 3: the blocked sweep's memory
// pass reaches 5.4-5.6 TB/s where a one-shot flat copy of the same buffer reaches 6.4-6.6 — why?).
// Unit of work: a tile of 4 rows x 512 columns of an m x ld fp64 tableau (4 x 4 KiB pieces, ld * 8 bytes apart), copied
// src -> dst through registers, nt loads and stores, D tiles in flight per workgroup.
//   one-shot : one workgroup per tile, dispatched in address order (what k_update does)
//   stride   : W persistent workgroups, workgroup w takes tiles w, w + W, ... in address order (dense moving window)
//   runs     : W persistent workgroups = strips x groups; each walks down a contiguous run of rows of ONE strip
//              (what k_sweep32_steady / k_sweep32_dma do: the pivot-row slices of a strip live in its registers)
//   ordered  : same workgroups, each takes every G-th tile of its strip (all workgroups inside one window of rows)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/copy_patterns.hip -o scripts/micro/copy_patterns
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } \
  } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void tile_load(d2 (&x)[4], const double* src, int64_t ld, int64_t batch, int strip) {
  const double* p = src + batch * 4 * ld + strip * 512 + 2 * threadIdx.x;
#pragma unroll
  for (int r = 0; r < 4; ++r) x[r] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(p + r * ld));
}
__device__ __forceinline__ void tile_store(const d2 (&x)[4], double* dst, int64_t ld, int64_t batch, int strip) {
  double* p = dst + batch * 4 * ld + strip * 512 + 2 * threadIdx.x;
#pragma unroll
  for (int r = 0; r < 4; ++r) __builtin_nontemporal_store(x[r], reinterpret_cast<d2*>(p + r * ld));
}

__global__ __launch_bounds__(256) void k_oneshot(const double* __restrict__ src, double* __restrict__ dst, int64_t ld, int nstrips) {
  d2 x[4];
  tile_load(x, src, ld, blockIdx.x / nstrips, blockIdx.x % nstrips);
  tile_store(x, dst, ld, blockIdx.x / nstrips, blockIdx.x % nstrips);
}

// one workgroup per tile of R rows x 512 columns; strip_major: consecutive workgroups go DOWN one strip (no two
// neighbours in the same row) instead of ACROSS the rows
template <int R>
__global__ __launch_bounds__(256) void k_oneshot_r(const double* __restrict__ src, double* __restrict__ dst, int64_t ld,
                                                   int nstrips, int ntiles_down, int strip_major) {
  const int strip = strip_major ? blockIdx.x / ntiles_down : blockIdx.x % nstrips;
  const int64_t tile = strip_major ? blockIdx.x % ntiles_down : blockIdx.x / nstrips;
  const double* p = src + tile * R * ld + strip * 512 + 2 * threadIdx.x;
  double* q = dst + tile * R * ld + strip * 512 + 2 * threadIdx.x;
  d2 x[R];
#pragma unroll
  for (int r = 0; r < R; ++r) x[r] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(p + r * ld));
#pragma unroll
  for (int r = 0; r < R; ++r) __builtin_nontemporal_store(x[r], reinterpret_cast<d2*>(q + r * ld));
}

// mode 0 stride, 1 runs, 2 ordered; nb = batches (of 4 rows) in the tableau; G = groups per strip (modes 1, 2)
template <int D>
__global__ __launch_bounds__(256) void k_persist(const double* __restrict__ src, double* __restrict__ dst, int64_t ld,
                                                 int nstrips, int nb, int mode, int G) {
  int strip, cnt;
  int64_t first, step;
  if (mode == 0) {
    // tiles in address order: tile t = (batch t / nstrips, strip t % nstrips); handled as a per-iteration decode
    strip = -1; first = blockIdx.x; step = gridDim.x;
    const int64_t ntiles = (int64_t)nb * nstrips;
    cnt = (int)((ntiles - first + step - 1) / step);
  } else {
    strip = blockIdx.x % nstrips;
    const int g = blockIdx.x / nstrips;
    if (mode == 1) { const int L = (nb + G - 1) / G; first = (int64_t)g * L; step = 1; cnt = max(0, min(L, nb - (int)first)); }
    else { first = g; step = G; cnt = (nb - g + G - 1) / G; }
  }
  auto where = [&](int k, int64_t& batch, int& s) {
    const int64_t t = first + (int64_t)k * step;
    if (mode == 0) { batch = t / nstrips; s = (int)(t % nstrips); } else { batch = t; s = strip; }
  };
  d2 x[D][4];
#pragma unroll
  for (int u = 0; u < D; ++u)
    if (u < cnt) { int64_t b; int s; where(u, b, s); tile_load(x[u], src, ld, b, s); }
  for (int k = 0; k < cnt; k += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      if (k + u < cnt) {
        int64_t b; int s;
        where(k + u, b, s);
        tile_store(x[u], dst, ld, b, s);
        if (k + u + D < cnt) { where(k + u + D, b, s); tile_load(x[u], src, ld, b, s); }
      }
    }
  }
}

// persistent workgroups that PULL their tiles from a counter: tile order = address order, handed to whichever
// workgroup is free (what the hardware's dispatcher does for one-shot workgroups).  per_strip: a workgroup is bound to
// a strip (as the sweep's are, by the pivot-row slices in its registers) and pulls the next 4-row batch OF ITS STRIP.
__global__ __launch_bounds__(256) void k_pull(const double* __restrict__ src, double* __restrict__ dst, int64_t ld,
                                              int nstrips, int nb, int per_strip, unsigned* ctr) {
  __shared__ unsigned sh_t;
  const int my_strip = blockIdx.x % nstrips;
  const unsigned limit = per_strip ? (unsigned)nb : (unsigned)nb * nstrips;
  for (;;) {
    if (threadIdx.x == 0) sh_t = atomicAdd(ctr + (per_strip ? my_strip * 32 : 0), 1u);
    __syncthreads();
    const unsigned t = sh_t;
    __syncthreads();
    if (t >= limit) break;
    const int64_t batch = per_strip ? t : t / nstrips;
    const int strip = per_strip ? my_strip : (int)(t % nstrips);
    d2 x[4];
    tile_load(x, src, ld, batch, strip);
    tile_store(x, dst, ld, batch, strip);
  }
}

// every WAVE on its own: bound to a 128-column sub-strip (1 KiB per row), pulls 4-row batches of that sub-strip from
// the sub-strip's counter L pulls ahead; no workgroup-level coupling at all (blockDim = 256: four independent waves)
template <int L>
__global__ __launch_bounds__(256) void k_pull_wave(const double* __restrict__ src, double* __restrict__ dst, int64_t ld,
                                                   int nsub, int nb, unsigned* ctr) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = (blockIdx.x * 4 + wave) % nsub;
  unsigned q[L + 1];
  auto pull = [&]() -> unsigned {
    unsigned t = 0;
    if (lane == 0) t = atomicAdd(ctr + sub * 32, 1u);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)t);
  };
#pragma unroll
  for (int k = 0; k <= L; ++k) q[k] = pull();
  for (;;) {
    if (q[0] >= (unsigned)nb) break;
    const double* p = src + (int64_t)q[0] * 4 * ld + sub * 128 + 2 * lane;
    double* o = dst + (int64_t)q[0] * 4 * ld + sub * 128 + 2 * lane;
    d2 x[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(p + r * ld));
#pragma unroll
    for (int r = 0; r < 4; ++r) __builtin_nontemporal_store(x[r], reinterpret_cast<d2*>(o + r * ld));
#pragma unroll
    for (int k = 0; k < L; ++k) q[k] = q[k + 1];
    q[L] = pull();
  }
}

// independent waves with tiles of R rows: a wave bound to a sub-strip of W8 bytes per row (1024 = 128 columns, one 16-byte
// access per lane; 512 = 64 columns, one 8-byte access per lane — k_sweep64_mfma2's tile is 16 rows x 512 bytes)
template <int R, int W8>
__global__ __launch_bounds__(256) void k_pull_wave_r(const double* __restrict__ src, double* __restrict__ dst, int64_t ld,
                                                     int nsub, int nt, unsigned* ctr) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = (blockIdx.x * 4 + wave) % nsub;
  auto pull = [&]() -> unsigned {
    unsigned t = 0;
    if (lane == 0) t = atomicAdd(ctr + sub * 32, 1u);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)t);
  };
  unsigned q0 = pull(), q1 = pull();
  for (;;) {
    if (q0 >= (unsigned)nt) break;
    if (W8 == 1024) {
      const double* p = src + (int64_t)q0 * R * ld + sub * 128 + 2 * lane;
      double* o = dst + (int64_t)q0 * R * ld + sub * 128 + 2 * lane;
      d2 x[R];
#pragma unroll
      for (int r = 0; r < R; ++r) x[r] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(p + r * ld));
#pragma unroll
      for (int r = 0; r < R; ++r) __builtin_nontemporal_store(x[r], reinterpret_cast<d2*>(o + r * ld));
    } else {
      const double* p = src + (int64_t)q0 * R * ld + sub * 64 + lane;
      double* o = dst + (int64_t)q0 * R * ld + sub * 64 + lane;
      double x[R];
#pragma unroll
      for (int r = 0; r < R; ++r) x[r] = __builtin_nontemporal_load(p + r * ld);
#pragma unroll
      for (int r = 0; r < R; ++r) __builtin_nontemporal_store(x[r], o + r * ld);
    }
    q0 = q1;
    q1 = pull();
  }
}

// the same with look-ahead, as a real kernel needs it: a ticket names a unit of U batches (4 U rows) of the
// workgroup's strip and is pulled L units before it is used; D batches in flight in registers
template <int U, int L, int D>
__global__ __launch_bounds__(256) void k_pull_ahead(const double* __restrict__ src, double* __restrict__ dst, int64_t ld,
                                                    int nstrips, int nb, unsigned* ctr) {
  __shared__ unsigned sh_t[2];
  const int strip = blockIdx.x % nstrips;
  const unsigned nunits = (unsigned)((nb + U - 1) / U);
  unsigned q[L + 1];   // tickets of the next L + 1 units
  auto pull = [&](int par) -> unsigned {
    if (threadIdx.x == 0) sh_t[par] = atomicAdd(ctr + strip * 32, 1u);
    __syncthreads();
    return sh_t[par];
  };
#pragma unroll
  for (int k = 0; k <= L; ++k) q[k] = pull(k & 1);
  __syncthreads();
  // batches in flight: a small software pipeline over the flattened batch sequence of the pulled units
  d2 x[D][4];
  // position of the loader: unit slot lu (index into q, 0 = current), batch lb within it
  int par = (L + 1) & 1;
  for (;;) {
    if (q[0] >= nunits) break;
    // this unit's batches, D at a time (no overlap across units in this model beyond the tickets: keeps it simple)
#pragma unroll
    for (int b0 = 0; b0 < U; b0 += D) {
#pragma unroll
      for (int u = 0; u < D; ++u)
        if (b0 + u < U && (int64_t)q[0] * U + b0 + u < nb) tile_load(x[u], src, ld, (int64_t)q[0] * U + b0 + u, strip);
#pragma unroll
      for (int u = 0; u < D; ++u)
        if (b0 + u < U && (int64_t)q[0] * U + b0 + u < nb) tile_store(x[u], dst, ld, (int64_t)q[0] * U + b0 + u, strip);
    }
#pragma unroll
    for (int k = 0; k < L; ++k) q[k] = q[k + 1];
    q[L] = pull(par);
    par ^= 1;
  }
}

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
  const int64_t ld = argc > 2 ? atoi(argv[2]) : 16384;
  const int reps = argc > 3 ? atoi(argv[3]) : 10;
  const int64_t skew = argc > 4 ? atoi(argv[4]) : 512;   // doubles between the two buffers' alignments
  const int nstrips = (int)(ld / 512), nb = m / 4;
  double *src, *dst0;
  CK(hipMalloc(&src, (size_t)m * ld * 8));
  CK(hipMalloc(&dst0, ((size_t)m * ld + skew) * 8));
  double* dst = dst0 + skew;
  CK(hipMemset(src, 0x3c, (size_t)m * ld * 8));
  const double bytes = 16.0 * m * ld;
  auto report = [&](const char* what, float t) { printf("%-44s %.3f ms  %.2f TB/s\n", what, t, bytes / t * 1e-9); };
  report("one-shot, 4-row tiles in address order", time_ms([&] {
    hipLaunchKernelGGL(k_oneshot, dim3(nb * nstrips), dim3(256), 0, 0, src, dst, ld, nstrips); }, reps));
#define ONE(R_)                                                                                                    \
  for (int sm = 0; sm < 2; ++sm) {                                                                                \
    char nm[96];                                                                                                  \
    snprintf(nm, sizeof nm, "one-shot %2d-row tiles, %s", R_, sm ? "down the strips" : "across the rows");        \
    report(nm, time_ms([&] { hipLaunchKernelGGL((k_oneshot_r<R_>), dim3(m / R_ * nstrips), dim3(256), 0, 0, src, dst, ld, \
                                                nstrips, m / R_, sm); }, reps));                                  \
  }
  ONE(1) ONE(2) ONE(4) ONE(8) ONE(16) ONE(32)
#undef ONE
  unsigned* ctr;
  CK(hipMalloc(&ctr, 4 * 32 * 64));
  for (int W : {512, 1024, 2048})
    for (int ps = 0; ps < 2; ++ps) {
      char nm[96];
      snprintf(nm, sizeof nm, "pull, %4d wgs, %s", W, ps ? "bound to a strip" : "any tile");
      report(nm, time_ms([&] {
        CK(hipMemsetAsync(ctr, 0, 4 * 32 * 64, 0));
        hipLaunchKernelGGL(k_pull, dim3(W), dim3(256), 0, 0, src, dst, ld, nstrips, nb, ps, ctr); }, reps));
    }
#define PA(U_, L_, D_)                                                                                              \
  {                                                                                                                \
    char nm[96];                                                                                                   \
    snprintf(nm, sizeof nm, "pull ahead: unit %d batches, %d units ahead, %d in flight", U_, L_, D_);              \
    report(nm, time_ms([&] {                                                                                       \
      CK(hipMemsetAsync(ctr, 0, 4 * 32 * 64, 0));                                                                  \
      hipLaunchKernelGGL((k_pull_ahead<U_, L_, D_>), dim3(512), dim3(256), 0, 0, src, dst, ld, nstrips, nb, ctr); }, reps)); \
  }
  PA(1, 0, 1) PA(1, 1, 1) PA(1, 2, 1) PA(1, 4, 1) PA(2, 0, 2) PA(2, 1, 2) PA(2, 2, 2) PA(2, 3, 2) PA(4, 1, 2) PA(4, 2, 4) PA(8, 1, 4) PA(8, 2, 4)
#undef PA
  unsigned* ctr2;
  CK(hipMalloc(&ctr2, 4 * 32 * 1024));
#define PW(L_, W_)                                                                                                  \
  {                                                                                                                \
    char nm[96];                                                                                                   \
    snprintf(nm, sizeof nm, "independent waves, %d pulls ahead, %d wgs", L_, W_);                                  \
    report(nm, time_ms([&] {                                                                                       \
      CK(hipMemsetAsync(ctr2, 0, 4 * 32 * 1024, 0));                                                               \
      hipLaunchKernelGGL((k_pull_wave<L_>), dim3(W_), dim3(256), 0, 0, src, dst, ld, nstrips * 4, nb, ctr2); }, reps)); \
  }
  PW(0, 512) PW(2, 512) PW(3, 512) PW(3, 448) PW(3, 1024)
#undef PW
#define PR(R_, W8_, WG_)                                                                                            \
  {                                                                                                                \
    char nm[96];                                                                                                   \
    snprintf(nm, sizeof nm, "independent waves, %2d-row x %4d-byte tiles, %d wgs", R_, W8_, WG_);                   \
    report(nm, time_ms([&] {                                                                                       \
      CK(hipMemsetAsync(ctr2, 0, 4 * 32 * 1024, 0));                                                               \
      hipLaunchKernelGGL((k_pull_wave_r<R_, W8_>), dim3(WG_), dim3(256), 0, 0, src, dst, ld,                       \
                         (int)(ld * 8 / W8_), m / R_, ctr2); }, reps));                                            \
  }
  PR(2, 1024, 512) PR(4, 1024, 512) PR(8, 1024, 512) PR(16, 1024, 512) PR(4, 512, 512) PR(16, 512, 512) PR(16, 512, 384) PR(4, 1024, 384) PR(16, 1024, 384)
#undef PR
  for (int W : {512}) {
    const int G = W / nstrips;
    char name[96];
#define RUN(D_)                                                                                                     \
    for (int mode = 0; mode < 3; ++mode) {                                                                          \
      snprintf(name, sizeof name, "persistent %4d wgs, %d in flight, %s", W, D_,                                   \
               mode == 0 ? "stride" : mode == 1 ? "runs" : "ordered");                                              \
      report(name, time_ms([&] { hipLaunchKernelGGL((k_persist<D_>), dim3(W), dim3(256), 0, 0, src, dst, ld, nstrips, \
                                                    nb, mode, G); }, reps));                                        \
    }
    RUN(1) RUN(2) RUN(4)
#undef RUN
  }
  return 0;
}
