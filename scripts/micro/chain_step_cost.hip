// Micro-benchmark: what ONE step of a pending-pivot chain costs a lone wave per SIMD on gfx950 (Next 1).
// The decision kernel (k_block_chain2_t) replays 48-96 pending pivots per phase on one wave per SIMD; round 4 measured
// ~100-170 shader cycles per step for what is one v_fma_f64.  This file times the candidate instruction patterns of a step,
// 64 steps each, between two s_memtime stamps, with one wave per SIMD (256 threads on one CU) and with one wave per CU:
//   hipcc --offload-arch=gfx950 -O3 chain_step_cost.hip -o chain_step_cost && ./chain_step_cost
// Output: shader cycles per step (s_memtime) and ns per step (s_memrealtime, 100 MHz), median of 32 repeats per pattern.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
#define RSTAMP(t) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")

// every pattern: 64 steps on the dependent value x; c (per lane), p (per lane), st (per-lane start index), hm (uniform mask)
template <int PAT>
__device__ __forceinline__ void body(double& x, double& y, double c, double p, int st, unsigned hm, const double* lds) {
  if constexpr (PAT == 0) {   // one dependent v_fma_f64 per step (the floor of the fused mode)
    asm volatile(".rept 64\n\tv_fma_f64 %0, -%1, %2, %0\n\t.endr" : "+v"(x) : "v"(c), "v"(p));
  } else if constexpr (PAT == 1) {   // v_mul_f64 + v_add_f64 (the floor of the default arithmetic)
    double t;
    asm volatile(".rept 64\n\tv_mul_f64 %1, %2, %3\n\tv_add_f64 %0, %0, -%1\n\t.endr" : "+v"(x), "=&v"(t) : "v"(c), "v"(p));
  } else if constexpr (PAT == 2) {   // round 4, first ring half: scalar bit test + branch NOT taken
    asm volatile(".rept 64\n\ts_and_b32 s2, %3, 1\n\ts_cmp_eq_u32 s2, 0\n\tv_fma_f64 %0, -%1, %2, %0\n\ts_cbranch_scc0 9f\n\t.endr\n9:"
                 : "+v"(x) : "v"(c), "v"(p), "s"(hm) : "s2", "scc");
  } else if constexpr (PAT == 3) {   // round 4, second ring half: the rare path inline, the branch over it TAKEN every step
    asm volatile(".rept 64\n\ts_and_b32 s2, %3, 1\n\ts_cmp_eq_u32 s2, 0\n\tv_fma_f64 %0, -%1, %2, %0\n\ts_cbranch_scc1 1f\n\t"
                 "v_mov_b32 %4, 0\n\tv_mov_b32 %4, 0\n\tv_mov_b32 %4, 0\n\tv_mov_b32 %4, 0\n\tv_mov_b32 %4, 0\n\tv_mov_b32 %4, 0\n1:\n\t.endr"
                 : "+v"(x) : "v"(c), "v"(p), "s"(hm), "v"(st) : "s2", "scc");
  } else if constexpr (PAT == 4) {   // proposal: per-lane start index as an EXEC mask (s_mov + v_cmpx + v_fma)
    asm volatile(".set k, 0\n\t.rept 64\n\ts_mov_b64 exec, -1\n\tv_cmpx_ge_i32 k, %3\n\tv_fma_f64 %0, -%1, %2, %0\n\t.set k, k+1\n\t.endr\n\t"
                 "s_mov_b64 exec, -1" : "+v"(x) : "v"(c), "v"(p), "v"(st) : "vcc", "exec");
  } else if constexpr (PAT == 5) {   // select form (compiler-generated): v_cmp + v_fma + two v_cndmask on the dependent path
#pragma unroll
    for (int k = 0; k < 64; ++k) {
      const double t = __fma_rn(-c, p, x);
      x = (k == st) ? p : t;
      asm volatile("" : "+v"(x));
    }
  } else if constexpr (PAT == 6) {   // v_fma + one ds_read_b128 per two steps (the parameters of the next chunk)
    double q0, q1;
    asm volatile(".rept 32\n\tds_read_b128 %1, %4\n\tv_fma_f64 %0, -%2, %3, %0\n\tv_fma_f64 %0, -%2, %3, %0\n\t.endr\n\ts_waitcnt lgkmcnt(0)"
                 : "+v"(x), "=&v"(*(double2*)&q0) : "v"(c), "v"(p), "v"((unsigned)(size_t)lds) : "memory");
    (void)q1;
  } else if constexpr (PAT == 7) {   // two independent chains interleaved (two rows per thread)
    asm volatile(".rept 64\n\tv_fma_f64 %0, -%2, %3, %0\n\tv_fma_f64 %1, -%2, %3, %1\n\t.endr" : "+v"(x), "+v"(y) : "v"(c), "v"(p));
  } else if constexpr (PAT == 8) {   // nothing (cost of the stamps)
  } else if constexpr (PAT == 9) {   // 64 dependent SALU instructions
    asm volatile(".rept 64\n\ts_add_u32 s2, s2, 1\n\t.endr" ::: "s2", "scc");
  } else if constexpr (PAT == 10) {  // 64 independent 32-bit VALU moves
    asm volatile(".rept 64\n\tv_mov_b32 %0, 0\n\t.endr" : "=v"(st));
    x += st;
  } else if constexpr (PAT == 11) {  // a branch NOT taken alone (+ the fma)
    asm volatile(".rept 64\n\tv_fma_f64 %0, -%1, %2, %0\n\ts_cbranch_scc1 9f\n\t.endr\n9:" : "+v"(x) : "v"(c), "v"(p) : "scc");
  } else if constexpr (PAT == 12) {  // a branch TAKEN to the next instruction (+ the fma)
    asm volatile("s_cmp_eq_u32 0, 0\n\t.rept 64\n\tv_fma_f64 %0, -%1, %2, %0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:\n\t.endr" : "+v"(x) : "v"(c), "v"(p) : "scc");
  } else if constexpr (PAT == 13) {  // per-chunk masks: 8 x v_cmp into SGPR pairs, then s_mov exec + v_fma per step
    asm volatile(".set k, 0\n\t.rept 8\n\t"
                 "v_cmp_le_i32 s[4:5], %3, k\n\tv_cmp_le_i32 s[6:7], %3, k+1\n\tv_cmp_le_i32 s[8:9], %3, k+2\n\tv_cmp_le_i32 s[10:11], %3, k+3\n\t"
                 "v_cmp_le_i32 s[12:13], %3, k+4\n\tv_cmp_le_i32 s[14:15], %3, k+5\n\tv_cmp_le_i32 s[16:17], %3, k+6\n\tv_cmp_le_i32 s[18:19], %3, k+7\n\t"
                 "s_mov_b64 exec, s[4:5]\n\tv_fma_f64 %0, -%1, %2, %0\n\ts_mov_b64 exec, s[6:7]\n\tv_fma_f64 %0, -%1, %2, %0\n\t"
                 "s_mov_b64 exec, s[8:9]\n\tv_fma_f64 %0, -%1, %2, %0\n\ts_mov_b64 exec, s[10:11]\n\tv_fma_f64 %0, -%1, %2, %0\n\t"
                 "s_mov_b64 exec, s[12:13]\n\tv_fma_f64 %0, -%1, %2, %0\n\ts_mov_b64 exec, s[14:15]\n\tv_fma_f64 %0, -%1, %2, %0\n\t"
                 "s_mov_b64 exec, s[16:17]\n\tv_fma_f64 %0, -%1, %2, %0\n\ts_mov_b64 exec, s[18:19]\n\tv_fma_f64 %0, -%1, %2, %0\n\t"
                 "s_mov_b64 exec, -1\n\t.set k, k+8\n\t.endr"
                 : "+v"(x) : "v"(c), "v"(p), "v"(st)
                 : "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19", "exec");
  }
}

template <int PAT>
__global__ __launch_bounds__(256) void k(long long* out, double* sink, double c, double p, int st_base, unsigned hm, int reps) {
  __shared__ __attribute__((aligned(16))) double lds[64];
  if (threadIdx.x < 64) lds[threadIdx.x] = 1.0;
  __syncthreads();
  double x = 1.0 + threadIdx.x * 1e-9, y = x + 1.0;
  const int st = st_base < 0 ? -1 : (int)(threadIdx.x & 63) % (st_base + 1);   // per-lane start index
  for (int r = 0; r < reps; ++r) {
    long long t0, t1, w0, w1;
    RSTAMP(w0);
    STAMP(t0);
    body<PAT>(x, y, c, p, st, hm, lds);
    STAMP(t1);
    RSTAMP(w1);
    if ((threadIdx.x & 63) == 0) {
      out[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * reps * 2 + 2 * r] = t1 - t0;
      out[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * reps * 2 + 2 * r + 1] = w1 - w0;
    }
  }
  if (x + y == 123.456) sink[0] = x;
}

// straight-line code the first time through (instruction fetch from a cold instruction cache) against the second time
__global__ __launch_bounds__(64) void k_cold(long long* out, double* sink, double c, double p) {
  double x = 1.0 + threadIdx.x * 1e-9;
  for (int r = 0; r < 4; ++r) {
    long long t0, t1;
    STAMP(t0);
    asm volatile(".rept 2048\n\tv_fma_f64 %0, -%1, %2, %0\n\t.endr" : "+v"(x) : "v"(c), "v"(p));   // 16 KiB of code
    STAMP(t1);
    if (threadIdx.x == 0) out[r] = t1 - t0;
  }
  if (x == 123.456) sink[0] = x;
}

static const char* kNames[] = {
    "v_fma_f64 (dependent)", "v_mul_f64 + v_add_f64", "r04 half 1: s_and,s_cmp,v_fma,s_cbranch not taken",
    "r04 half 2: ... s_cbranch TAKEN over 6 instr", "s_mov exec,-1 + v_cmpx + v_fma", "v_cmp + v_fma + 2 v_cndmask",
    "v_fma x2 + ds_read_b128 per 2 steps", "two independent chains (2 fma per step)", "empty (stamps only)",
    "64 dependent s_add_u32", "64 v_mov_b32", "v_fma + s_cbranch not taken", "v_fma + s_cbranch taken (+1)",
    "8 v_cmp->sgpr per chunk, then s_mov exec + v_fma"};

template <int PAT>
void run(int threads, int st_base) {
  const int reps = 32;
  long long* d;
  double* sink;
  hipMalloc(&d, sizeof(long long) * 4 * reps * 2);
  hipMalloc(&sink, 8);
  hipMemset(d, 0, sizeof(long long) * 4 * reps * 2);
  hipLaunchKernelGGL((k<PAT>), dim3(1), dim3(threads), 0, 0, d, sink, 1e-9, 0.5, st_base, 0u, reps);
  hipDeviceSynchronize();
  std::vector<long long> h(4 * reps * 2);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cyc, ns;
  for (int r = 4; r < reps; ++r) { cyc.push_back((double)h[2 * r]); ns.push_back((double)h[2 * r + 1] * 10.0); }
  std::sort(cyc.begin(), cyc.end());
  std::sort(ns.begin(), ns.end());
  printf("%-52s waves/CU %d  st %3d : %7.1f cycles, %7.1f ns per 64 steps = %5.1f cycles / step\n", kNames[PAT], threads / 64, st_base,
         cyc[cyc.size() / 2], ns[ns.size() / 2], cyc[cyc.size() / 2] / 64.0);
  hipFree(d);
  hipFree(sink);
}

int main() {
  for (int threads : {256, 64}) {
    run<8>(threads, -1);
    run<0>(threads, -1);
    run<1>(threads, -1);
    run<2>(threads, -1);
    run<3>(threads, -1);
    run<11>(threads, -1);
    run<12>(threads, -1);
    run<4>(threads, -1);
    run<4>(threads, 40);
    run<13>(threads, 40);
    run<5>(threads, 40);
    run<6>(threads, -1);
    run<7>(threads, -1);
    run<9>(threads, -1);
    run<10>(threads, -1);
  }
  long long* d;
  double* sink;
  hipMalloc(&d, 64);
  hipMalloc(&sink, 8);
  hipLaunchKernelGGL(k_cold, dim3(1), dim3(64), 0, 0, d, sink, 1e-9, 0.5);
  hipDeviceSynchronize();
  long long h[4];
  hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
  printf("2048 straight-line v_fma_f64 (16 KiB of code), passes 1..4: %lld %lld %lld %lld cycles (%.1f / %.1f per instruction)\n", h[0], h[1], h[2],
         h[3], h[0] / 2048.0, h[3] / 2048.0);
  return 0;
}
