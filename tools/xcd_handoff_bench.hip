// Micro-benchmark behind the design of the XCD-local triangular solve (docs/HISTORY.md section 3):
// cost of one "level" hand-off between W single-wave workgroups
//   mode 0: the W waves of a group all sit on ONE XCD (group chosen from HW_REG_XCC_ID at run time):
//           plain stores (stay in that XCD's L2) + per-wave flag words, sc1 polls / sc1 loads
//   mode 1: groups mix XCDs: write-through (sc1) stores + flags, sc1 loads
// Each level: wait for all flags of the previous level, gather 13 doubles written in the previous
// level, store one double, drain, set the own flag.   hipcc --offload-arch=gfx950 -O3 xcd_handoff_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF; } // HW_REG_XCC_ID[3:0]

__global__ __launch_bounds__(64) void k_chain(int mode, int nlev, unsigned *tickets, unsigned *arrived, unsigned *flags, double *x, unsigned *census,
                                               unsigned *err)
{
  const int lane = threadIdx.x;
  unsigned xcc = 0, t = 0;
  if (lane == 0) {
    xcc = xcc_id();
    t = atomicAdd(&tickets[mode == 0 ? xcc : 0], 1u);
    __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    census[blockIdx.x] = xcc;
  }
  xcc = __builtin_amdgcn_readfirstlane(xcc);
  t = __builtin_amdgcn_readfirstlane(t);
  // grid barrier: every wave has drawn its ticket
  for (unsigned spins = 0; __hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x; ++spins) {
    if (spins > (1u << 22)) { *err = 1; return; }
    __builtin_amdgcn_s_sleep(2);
  }
  unsigned grp, rank, W;
  if (mode == 0) { grp = xcc; rank = t; W = __hip_atomic_load(&tickets[xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  else { grp = t % 8; rank = t / 8; W = gridDim.x / 8; }
  unsigned *fl = flags + (size_t)grp * nlev * 64;
  double *xg = x + (size_t)grp * nlev * 64;
  double acc = 1.0;
  for (int lev = 0; lev < nlev; ++lev) {
    if (lev > 0) {
      const unsigned *fp = fl + (size_t)(lev - 1) * 64;
      for (unsigned spins = 0;; ++spins) {
        unsigned v = lane < (int)W ? __hip_atomic_load(fp + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
        if (__all(v == 1u)) break;
        if (spins > (1u << 22)) { *err = 2; return; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // 13 gathers of values of the previous level (sc1: bypass the non-coherent L1)
      double s = 0;
      if (lane < 13) {
        unsigned long long u = __hip_atomic_load((unsigned long long *)(xg + (size_t)(lev - 1) * 64 + (rank + lane) % W), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s = __longlong_as_double((long long)u);
      }
      for (int o = 8; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
      acc = 0.5 * acc + 1e-3 * __shfl(s, 0, 64);
    }
    if (lane == 0) {
      if (mode == 0) xg[(size_t)lev * 64 + rank] = acc;   // plain: stays in this XCD's L2
      else __hip_atomic_store((unsigned long long *)(xg + (size_t)lev * 64 + rank), (unsigned long long)__double_as_longlong(acc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      if (mode == 0) fl[(size_t)lev * 64 + rank] = 1u;
      else __hip_atomic_store(fl + (size_t)lev * 64 + rank, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// mode 2: "slab ownership": a wave reads back values IT stored one level earlier (no flag, no drain: same-wave
// store -> sc1 load through L2) plus ONE progress word of its predecessor wave; 13 x 64 scattered gathers,
// 64 scattered stores per level, progress published lazily once the gathers have returned (in-order vmcnt
// then implies that the previous level's stores are complete).
__global__ __launch_bounds__(64) void k_slab(int nlev, unsigned *tickets, unsigned *arrived, unsigned *progress, double *x, unsigned *err, double *sink)
{
  const int lane = threadIdx.x;
  unsigned xcc = 0, t = 0;
  if (lane == 0) {
    xcc = xcc_id();
    t = atomicAdd(&tickets[xcc], 1u);
    __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  xcc = __builtin_amdgcn_readfirstlane(xcc);
  t = __builtin_amdgcn_readfirstlane(t);
  for (unsigned spins = 0; __hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x; ++spins) {
    if (spins > (1u << 22)) { *err = 1; return; }
    __builtin_amdgcn_s_sleep(2);
  }
  const unsigned W = __hip_atomic_load(&tickets[xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  double *xs = x + ((size_t)xcc * 64 + t) * (size_t)(1 << 17);   // 1 MB slab per wave
  unsigned *prog = progress + xcc * 64 * 32;                      // one 128-B line per wave
  double acc = 1.0 + lane;
  for (int lev = 0; lev < nlev; ++lev) {
    if (t > 0 && lev > 0) { // predecessor must have completed level lev-1
      for (unsigned spins = 0;; ++spins) {
        const unsigned p = __hip_atomic_load(prog + (t - 1) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p >= (unsigned)lev) break;
        if (spins > (1u << 22)) { *err = 2; return; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    double s = 0.0;
    if (lev > 0) {
#pragma unroll
      for (int u = 0; u < 13; ++u) {
        const size_t idx = ((size_t)(lev - 1) * 64 + ((lane * 7 + u * 5) & 63)) * 17 % (1 << 17); // scattered, written one level ago
        s += __longlong_as_double((long long)__hip_atomic_load((unsigned long long *)(xs + idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      }
    }
    // gathers returned => stores of level lev-1 are complete (vmcnt retires in order): publish lazily
    if (lev > 0 && lane == 0) *(volatile unsigned *)(prog + t * 32) = (unsigned)lev;
    acc = 0.5 * acc + 1e-3 * s;
    xs[((size_t)lev * 64 + lane) * 17 % (1 << 17)] = acc; // plain store: stays in this XCD's L2
  }
  if (lane == 0) sink[blockIdx.x] = acc;
  (void)W;
}

int main()
{
  const int nlev = 2000, G = 256;
  unsigned *tickets, *arrived, *flags, *census, *err;
  double *x;
  CHECK(hipMalloc(&tickets, 64));
  CHECK(hipMalloc(&arrived, 64));
  CHECK(hipMalloc(&err, 64));
  CHECK(hipMalloc(&census, G * 4));
  CHECK(hipMalloc(&flags, (size_t)8 * nlev * 64 * 4));
  CHECK(hipMalloc(&x, (size_t)8 * nlev * 64 * 8));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipMemset(tickets, 0, 64));
      CHECK(hipMemset(arrived, 0, 64));
      CHECK(hipMemset(err, 0, 64));
      CHECK(hipMemset(flags, 0, (size_t)8 * nlev * 64 * 4));
      CHECK(hipMemset(x, 0, (size_t)8 * nlev * 64 * 8));
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_chain, dim3(G), dim3(64), 0, 0, mode, nlev, tickets, arrived, flags, x, census, err);
      hipEventRecord(e1);
      CHECK(hipDeviceSynchronize());
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      unsigned herr, htk[8];
      std::vector<unsigned> hc(G);
      hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
      hipMemcpy(htk, tickets, 32, hipMemcpyDeviceToHost);
      hipMemcpy(hc.data(), census, G * 4, hipMemcpyDeviceToHost);
      int hist[16] = {0};
      for (auto c : hc) hist[c & 15]++;
      printf("mode %d rep %d: %.3f ms total, %.3f us per level, err %u, waves per xcc:", mode, rep, ms, 1e3 * ms / nlev, herr);
      for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
      printf("\n");
    }
  {
    unsigned *progress;
    double *xs, *sink;
    CHECK(hipMalloc(&progress, 8 * 64 * 32 * 4));
    CHECK(hipMalloc(&xs, (size_t)8 * 64 * (1 << 17) * 8));
    CHECK(hipMalloc(&sink, G * 8));
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipMemset(tickets, 0, 64));
      CHECK(hipMemset(arrived, 0, 64));
      CHECK(hipMemset(err, 0, 64));
      CHECK(hipMemset(progress, 0, 8 * 64 * 32 * 4));
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_slab, dim3(G), dim3(64), 0, 0, nlev, tickets, arrived, progress, xs, err, sink);
      hipEventRecord(e1);
      CHECK(hipDeviceSynchronize());
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      unsigned herr;
      hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
      printf("mode 2 (slab ownership) rep %d: %.3f ms total, %.3f us per level, err %u\n", rep, ms, 1e3 * ms / nlev, herr);
    }
  }
  return 0;
}
