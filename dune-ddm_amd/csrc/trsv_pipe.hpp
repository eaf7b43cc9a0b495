// Device side of the "pipe" triangular-solve engine; schedule format and rationale: trsv_pipe_host.hpp.
//
// One workgroup = PIPE_NC compute waves + PIPE_NL loader waves; two workgroups per CU.  Workgroups pull tasks (64
// chains walked level by level) from a per-(subdomain, sweep) queue in topological order.  The loader waves stream the
// task's tiles HBM -> LDS with LDS-DMA (global_load_lds_dwordx4, 1 KiB per instruction) into a byte ring, far ahead of
// the compute wave.
// Per step of the compute wave: operands of the own task from the LDS result ring, operands of other tasks by sc1
// gathers from the position arrays (guarded by the producers' progress words, which are normally ahead), the row sum in
// ascending column order (= the sequential back-solve), one coalesced 512-byte store of the 64 results.  The wave is
// software-pipelined by hand: header and operand words of the NEXT tile, the progress check and the next step's 15 gathers
// are issued before the row sums of the current step.  A check that fails there is DEFERRED: the wave computes the current
// step first (it has the operands), then waits for the producers and issues the gathers (fetch_late) -- a consumer that
// runs right behind its producer does not idle with work at hand.
// PIPE_NC = 2 (built, bit-exact, not faster -- docs/HISTORY.md section 3) lets two compute waves alternate the steps of a task:
// wave w takes steps w, w + NC, ...; only the operands flagged in the tile's late mask are read after the previous
// step has signalled (one LDS word).
//
// Visibility protocol (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"):
//   * XCD-local mode (every XCD that owns a subdomain hosts workgroups): all tasks of a subdomain run on ONE XCD; results
//     are plain stores (they stay in that XCD's L2), every read of another wave's result is an sc1 load (bypasses
//     the non-coherent L1), the progress word is stored after the result stores have completed;
//   * placement-independent ("spread") mode: results and progress words are sc1 (write-through) stores;
//   * "completed": the wave drains its stores (s_waitcnt vmcnt(0)) at the top of the next step, behind that step's first
//     LDS reads, and then publishes.  (Publishing a step later without a drain - vmcnt retires in issue order, so a
//     returned younger load vouches for the store - was measured slower.)  The non-blocking poll of the producers' progress
//     words is issued together with the result store, so the same drain makes it valid: the producer check of the next
//     prefetch sees progress that is one L2 round trip old.
#pragma once
#include "trsv_pipe_host.hpp"

namespace ddm {

#ifndef DDM_PIPE_NC
#define DDM_PIPE_NC 1
#endif
#ifndef DDM_PIPE_RING_KIB
#define DDM_PIPE_RING_KIB 64
#endif
constexpr int PIPE_NC = DDM_PIPE_NC; // compute waves per workgroup (steps of a task alternate between them); experiments: -DDDM_PIPE_NC=2 -DDDM_PIPE_RING_KIB=112
constexpr int PIPE_NL = 2;          // loader waves per workgroup
constexpr int PIPE_DEPTH = 3;       // tiles a loader keeps in flight
constexpr int PIPE_RING_KIB = DDM_PIPE_RING_KIB; // LDS byte ring of tiles per workgroup (two workgroups per CU).  2 NC tiles (the current and the
                                    // prefetched step of every compute wave) + one in flight must fit, or loaders and compute waves
                                    // would wait for each other: 5 tiles of the widest row the builder accepts (pipe::MAX_W)
constexpr int PIPE_STAMP_WORDS = 16 + 256; // stamped build: words per task (16 sums, then the time of every step's result store)
constexpr int PIPE_WIDE = 12;       // further entries of a wide row handled with one gather latency
constexpr int PIPE_READY = 32;      // ready words (tiles in flight < 32: the smallest tile is 3 KiB)
constexpr int PIPE_CHUNK = pipe::MIN_W; // entries of a row held in registers; every tile has at least that many (padded), wider rows take the rest from the tile
constexpr int PIPE_RINGAREA = (pipe::RING_BYTES + 4 * (PIPE_READY + 16) + 1023) / 1024 * 1024; // result ring + control words
constexpr size_t PIPE_LDS_BYTES = (size_t)PIPE_RINGAREA + (size_t)PIPE_RING_KIB * 1024; // dynamic part

struct PipeStep { // registers of one step, filled NC steps ahead of their use (the factor entries are read
                  // from the tile when the step is computed: two sets of them would not fit the register budget)
  double xg[PIPE_CHUNK];
  uint32_t lofs[PIPE_CHUNK];   // LDS ring addresses of the operands (the zero row for operands that are gathered)
  double s0;
  int W;
  unsigned late;       // entries whose ring operand may come from one of the NC - 1 steps before this one
  unsigned tpos, vend; // tile position in the LDS ring (KiB), virtual ring offset behind the tile
  int deferred;        // (uniform) the producers were not far enough when the step was prefetched: its gathers are issued by fetch_late()
};
__device__ __forceinline__ double pipe_ld_sc1_off(const double *base, uint32_t byte_off)
{
  typedef const __attribute__((address_space(1))) unsigned char *gbytes; // global, not flat: base is a kernel-argument array
  const __attribute__((address_space(1))) unsigned long long *p = (const __attribute__((address_space(1))) unsigned long long *)((gbytes)(uintptr_t)base + byte_off);
  return __longlong_as_double((long long)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// waits until at most n of the wave's vector-memory operations are outstanding (rounded down to a multiple of 4: stricter)
__device__ __forceinline__ void pipe_wait_vmcnt_le(int n)
{
  switch (n >> 2) {
  case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
  case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
  case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
  case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
  case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
  case 6: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
  case 7: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
  case 8: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
  case 9: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
  case 10: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
  case 11: asm volatile("s_waitcnt vmcnt(44)" ::: "memory"); break;
  default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
  }
}
// the result ring sits at LDS address 0 (checked at kernel start): a ring operand is the ds_read address itself
__device__ __forceinline__ double pipe_lds_f64(uint32_t lds_addr) { return *(const __attribute__((address_space(3))) double *)(uintptr_t)lds_addr; }
// Gather issued behind the compiler's back: hipcc's waitcnt insertion cannot keep loads in flight across the back edge of
// the software-pipelined step loop (it drains them before the next tile is fetched), so the loads are inline asm and the
// wait is explicit: pipe_wait_gathers<N> names every register the loads write, which pins all their uses behind it.
__device__ __forceinline__ const double *pipe_uniform_ptr(const double *p)
{
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (const double *)(((uint64_t)hi << 32) | lo);
}
// The asm blocks start with s_nop 4: if the compiler has just restored a base pointer with v_readlane / v_readfirstlane
// (a VALU write of an SGPR), a VMEM instruction may only read it 5 wait states later, and the hazard recognizer does not
// look into inline asm.
// RHS_SC1: the right-hand side entry is read past the L1 (backward sweep: the forward results were written by other waves of this
// launch); the forward sweep reads the caller's vector, which nobody writes during the launch: through the L1, where a lane finds
// the cache line of its grid line again for 16 steps.
#define PIPE_GATHER8_REST                       \
  "global_load_dwordx2 %1, %9, %17 sc1\n\t"  \
  "global_load_dwordx2 %2, %10, %17 sc1\n\t" \
  "global_load_dwordx2 %3, %11, %17 sc1\n\t" \
  "global_load_dwordx2 %4, %12, %17 sc1\n\t" \
  "global_load_dwordx2 %5, %13, %17 sc1\n\t" \
  "global_load_dwordx2 %6, %14, %17 sc1\n\t" \
  "global_load_dwordx2 %7, %15, %17 sc1"
template <bool RHS_SC1>
__device__ __forceinline__ void pipe_gather_asm8(const double *rhs_uniform, uint32_t own, const double *src_uniform, const uint32_t (&off)[PIPE_CHUNK], double &s0,
                                                 double (&x)[PIPE_CHUNK])
{
  static_assert(PIPE_CHUNK == 14, "operand lists below");
  if (RHS_SC1)
    asm volatile("s_nop 4\n\t"
                 "global_load_dwordx2 %0, %8, %16 sc1\n\t" PIPE_GATHER8_REST
                 : "=&v"(s0), "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6])
                 : "v"(own), "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "v"(off[6]), "s"(rhs_uniform), "s"(src_uniform)
                 : "memory");
  else
    asm volatile("s_nop 4\n\t"
                 "global_load_dwordx2 %0, %8, %16\n\t" PIPE_GATHER8_REST
                 : "=&v"(s0), "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6])
                 : "v"(own), "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "v"(off[6]), "s"(rhs_uniform), "s"(src_uniform)
                 : "memory");
  asm volatile("s_nop 4\n\t"
               "global_load_dwordx2 %0, %7, %14 sc1\n\t"
               "global_load_dwordx2 %1, %8, %14 sc1\n\t"
               "global_load_dwordx2 %2, %9, %14 sc1\n\t"
               "global_load_dwordx2 %3, %10, %14 sc1\n\t"
               "global_load_dwordx2 %4, %11, %14 sc1\n\t"
               "global_load_dwordx2 %5, %12, %14 sc1\n\t"
               "global_load_dwordx2 %6, %13, %14 sc1"
               : "=&v"(x[7]), "=&v"(x[8]), "=&v"(x[9]), "=&v"(x[10]), "=&v"(x[11]), "=&v"(x[12]), "=&v"(x[13])
               : "v"(off[7]), "v"(off[8]), "v"(off[9]), "v"(off[10]), "v"(off[11]), "v"(off[12]), "v"(off[13]), "s"(src_uniform)
               : "memory");
}
// non-blocking poll of two progress words per lane (issued in front of a step's result store, valid behind the next drain)
__device__ __forceinline__ void pipe_poll_asm(const unsigned long long *p0, const unsigned long long *p1, unsigned long long &w0, unsigned long long &w1)
{
  asm volatile("global_load_dwordx2 %0, %2, off sc1\n\t"
               "global_load_dwordx2 %1, %3, off sc1"
               : "=&v"(w0), "=&v"(w1)
               : "v"(p0), "v"(p1)
               : "memory");
}
constexpr int PIPE_INFLIGHT = PIPE_CHUNK + 1; // loads per step issued ahead: right-hand side, PIPE_CHUNK operands
// PIPE_WIDE / 2 further operands of a wide row, issued in front of the next step's gathers (same in-flight rules)
constexpr int PIPE_WHALF = PIPE_WIDE / 2;
__device__ __forceinline__ void pipe_gather_asm_wide(const double *src_uniform, const uint32_t (&off)[PIPE_WHALF], double (&x)[PIPE_WHALF])
{
  static_assert(PIPE_WHALF == 6, "operand lists below");
  asm volatile("s_nop 4\n\t"
               "global_load_dwordx2 %0, %6, %12 sc1\n\t"
               "global_load_dwordx2 %1, %7, %12 sc1\n\t"
               "global_load_dwordx2 %2, %8, %12 sc1\n\t"
               "global_load_dwordx2 %3, %9, %12 sc1\n\t"
               "global_load_dwordx2 %4, %10, %12 sc1\n\t"
               "global_load_dwordx2 %5, %11, %12 sc1"
               : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5])
               : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "s"(src_uniform)
               : "memory");
}
__device__ __forceinline__ void pipe_pin_wide(double (&x)[PIPE_WHALF])
{
  asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5])::"memory");
}
template <int N>
__device__ __forceinline__ void pipe_wait_gathers(double &s0, double (&x)[PIPE_CHUNK])
{
  static_assert(PIPE_CHUNK == 14, "operand list below");
  asm volatile("s_waitcnt vmcnt(%15)"
               : "+v"(s0), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]),
                 "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13])
               : "n"(N)
               : "memory");
}
__device__ __forceinline__ void pipe_pin(double &s, double (&x)[PIPE_CHUNK])
{
  asm volatile("" : "+v"(s), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]),
               "+v"(x[12]), "+v"(x[13])::"memory");
}
__device__ __forceinline__ uint32_t pipe_lofs(int32_t op) { return min((uint32_t)op, (uint32_t)pipe::RING_Z); } // LDS byte offset (zero row if remote)
__device__ __forceinline__ uint32_t pipe_gofs(int32_t op) { return max((uint32_t)op, (uint32_t)pipe::RING_Z); } // global byte offset (the reserved zero if local)
__device__ __forceinline__ double pipe_or(double a, double b) { return __longlong_as_double(__double_as_longlong(a) | __double_as_longlong(b)); }

struct PipeParams {
  int ngroups;
  const pipe::Group *groups;
  const pipe::Task *tasks;
  const unsigned char *stream;
  const int32_t *koff;
  const double *d;              // right-hand side of the solve, natural order (the forward tasks read it row by row: tile word "own")
  double *ypos, *xpos;          // forward / backward results in position order
  unsigned long long *progress; // one word per task at stride 16 (128 B): (epoch << 32) | steps stored
  unsigned *queue;              // per group 4 words at stride 32: next task of L, of U; finished tasks of L, of U
  XcdState *st;
  unsigned *err;
  unsigned long long *stamps;   // diagnostics (nullptr in the product path)
  int spread;                   // != 0: placement-independent mode even if the XCD-local one is possible (few, large subdomains)
};

__global__ void k_pipe_prologue(XcdState *st, unsigned *queue, int nwords)
{
  for (int i = threadIdx.x; i < nwords; i += blockDim.x) queue[(size_t)i * 32] = 0;
  if (threadIdx.x < 8) st->tickets[threadIdx.x] = 0;
  if (threadIdx.x == 0) {
    st->global_ticket = 0;
    st->arrived = 0;
    st->epoch += 1;
  }
}
// The permutation from position order back to natural order (position = task base + 64 * step + lane: lane = chain, on a structured
// grid a grid line).  Read by position, the natural side touches 64 different cache lines per 64 positions and uses 8 bytes of each
// (the next step's positions hit the same lines again: L2-request bound).  The kernel therefore works on tiles of 64 lanes x 16
// steps and transposes through LDS: the position side is read in position order, the natural side written with the step index
// running fastest, i.e. 16 consecutive rows of a chain by 16 neighbouring threads (whole 128-byte segments when a chain is a grid
// line; any other chain shape is merely not faster than before).  (The opposite direction is gone since round 4: the forward tasks
// gather their right-hand side entries from the caller's vector themselves -- a lane re-uses one cache line for 16 steps.)
constexpr int PERM_STEPS = 16, PERM_TILE = 64 * PERM_STEPS, PERM_WG = 256, PERM_PAD = 65;
// x[row(pos)] = xpos[pos], optionally followed by the Schwarz level's "x *= pou" and "x += coarse correction" (same operations in
// the same order as the separate kernels, so the result is bit-identical; saves their passes over the overlapping vector)
__global__ __launch_bounds__(PERM_WG) void k_pipe_permute_out(int64_t npos, const int32_t *__restrict__ rowU, const double *__restrict__ xpos, double *__restrict__ x,
                                                             const double *__restrict__ scale, const double *__restrict__ add)
{
  __shared__ int32_t idx[PERM_STEPS * PERM_PAD];
  __shared__ double val[PERM_STEPS * PERM_PAD];
  const int64_t ntile = (npos + PERM_TILE - 1) / PERM_TILE;
  for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int64_t p0 = tile * PERM_TILE;
#pragma unroll
    for (int q = 0; q < PERM_TILE / PERM_WG; ++q) {
      const int e = q * PERM_WG + threadIdx.x;
      const int64_t p = p0 + e;
      const int o = (e >> 6) * PERM_PAD + (e & 63);
      idx[o] = p < npos ? rowU[p] : -1;
      val[o] = p < npos ? xpos[p] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PERM_TILE / PERM_WG; ++q) {
      const int e = q * PERM_WG + threadIdx.x;
      const int l = e / PERM_STEPS, st = e % PERM_STEPS;
      const int32_t r = idx[st * PERM_PAD + l];
      if (r >= 0) {
        double v = val[st * PERM_PAD + l];
        if (scale) v *= scale[r];
        if (add) v += add[r];
        x[r] = v;
      }
    }
    __syncthreads();
  }
}

__device__ __forceinline__ void pipe_glds16(const unsigned char *gsrc, unsigned char *lds_dst)
{
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc, (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

// STAMP (diagnostic build only): per task 8 words -- start / first step / end (s_memrealtime, 100 MHz), cycles of compute
// wave 0 waiting for tiles / for producers / between the previous step's signal and its own (s_memtime), steps, XCC id
template <bool STAMP>
__global__ __launch_bounds__(64 * (PIPE_NC + PIPE_NL)) void k_trsv_pipe(PipeParams P)
{
  static_assert(PIPE_NC >= 1 && PIPE_NC <= pipe::MAX_NC, "late masks exist for up to MAX_NC compute waves");
  // every compute wave holds its current tile and prefetches its next one while a loader fills a further one: 2 NC + 1 tiles give full
  // overlap.  With -DDDM_PIPE_TIGHT_RING the ring only has to hold NC + 1 of the WIDEST tiles (all current tiles and one more): a wave whose
  // next tile does not fit yet waits for the oldest step to release its tile -- less overlap on the wide steps, no deadlock (steps complete
  // in order, and a completed step frees the space the youngest prefetch is waiting for).
#ifdef DDM_PIPE_TIGHT_RING
  static_assert((PIPE_NC + 1) * (pipe::Geometry(pipe::MAX_W).tile_bytes / 1024) <= PIPE_RING_KIB, "tile ring too small for the widest tiles");
  static_assert((2 * PIPE_NC + 1) * (pipe::Geometry(pipe::MIN_W).tile_bytes / 1024) <= PIPE_RING_KIB, "tile ring too small for full overlap on the narrow tiles");
#else
  static_assert((2 * PIPE_NC + 1) * (pipe::Geometry(pipe::MAX_W).tile_bytes / 1024) <= PIPE_RING_KIB, "tile ring too small for the widest tiles");
#endif
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  // All LDS is dynamic and the result ring sits at LDS address 0, so that ring operands are ds_read addresses as they
  // stand.  Control words between the waves: relaxed workgroup-scope atomics (plain ds_read / ds_write; ordering is by
  // the s_waitcnt instructions below).
  double *ringd = reinterpret_cast<double *>(smem); // [RING rows of 64 results][zero row]
  unsigned *ctl = reinterpret_cast<unsigned *>(smem + pipe::RING_BYTES);
  unsigned *sh_ready = ctl;                    // [PIPE_READY] tile t is in LDS when sh_ready[t % PIPE_READY] == t + 1
  unsigned &sh_vconsumed = ctl[PIPE_READY];    // virtual ring offset (KiB) behind the last tile that is no longer needed
  unsigned &sh_xcc = ctl[PIPE_READY + 1], &sh_gt = ctl[PIPE_READY + 2], &sh_fail = ctl[PIPE_READY + 3], &sh_q = ctl[PIPE_READY + 4];
  unsigned &sh_xt = ctl[PIPE_READY + 6];       // this workgroup's arrival number on its XCD
  unsigned &sh_stepdone = ctl[PIPE_READY + 5]; // steps of the task whose results are in the LDS ring
  unsigned *sh_stored = ctl + PIPE_READY + 8;  // [PIPE_NC] own steps of compute wave w whose global stores have completed
  unsigned char *tiles = smem + PIPE_RINGAREA; // byte ring of tiles
  auto lds_load = [](const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  auto lds_store = [](unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // (uniform: scalar control flow)
  XcdState *st = P.st;
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) { // static LDS in front of the ring: not this build
    if (threadIdx.x == 0) __hip_atomic_store(P.err, 6u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  // the ring is zeroed once: its last row stays zero for good, every other row is written before it is read
  for (int k = wave; k <= pipe::RING; k += PIPE_NC + PIPE_NL) ringd[k * 64 + lane] = 0.0;
  if (threadIdx.x == 0) {
    const unsigned xcc = hw_xcc_id();
    sh_xcc = xcc;
    sh_xt = __hip_atomic_fetch_add(&st->tickets[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh_gt = __hip_atomic_fetch_add(&st->global_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&st->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned fail_ = 0;
    for (unsigned spins = 0; __hip_atomic_load(&st->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x; ++spins) {
      if (spins > (1u << 22)) {
        __hip_atomic_store(P.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        fail_ = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    sh_fail = fail_;
  }
  __syncthreads();
  if (sh_fail) return;
  const unsigned xcc = sh_xcc;
  const unsigned xt = sh_xt;
  const unsigned epoch = __hip_atomic_load(&st->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned tk = lane < 8 ? __hip_atomic_load(&st->tickets[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
  // XCD-local mode: subdomain g lives on XCD g % 8 -- possible when every XCD that owns a subdomain hosts workgroups (with
  // fewer than 8 subdomains the other XCDs idle: one subdomain has work for ~30 waves at a time, and same-XCD L2 hand-overs
  // are what keeps a step short)
  const bool local_ok = !P.spread && __all(lane >= min(P.ngroups, 8) || tk >= 1u);
  const bool wt = !local_ok;
  const int gfirst = local_ok ? (int)xcc : (int)(sh_gt % (unsigned)P.ngroups);
  const int gcount = local_ok ? ((int)xcc < P.ngroups ? (P.ngroups - (int)xcc + 7) / 8 : 0) : P.ngroups;

  for (int gi = 0; gi < gcount; ++gi) {
    // an XCD that owns several subdomains works on all of them at once: its workgroups start on different ones (each sweep is
    // latency-bound and keeps only some tens of workgroups busy) and move on to the others as their queues run dry
    const int g = local_ok ? gfirst + 8 * (int)((gi + xt) % (unsigned)gcount) : (gfirst + gi) % P.ngroups;
    const pipe::Group *Gp = P.groups + g;
    const int ntaskL = Gp->ntask[0];
    unsigned *qbase = P.queue + (size_t)g * 4 * 32;
    for (int sweep = 0; sweep < 2; ++sweep) {
      const bool upper = sweep == 1;
      const int ntask = Gp->ntask[sweep], task0 = Gp->task0[sweep];
      for (;;) {
        if (threadIdx.x == 0) {
          sh_q = __hip_atomic_fetch_add(qbase + sweep * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (int k = 0; k < PIPE_READY; ++k) lds_store(&sh_ready[k], 0u);
          lds_store(&sh_vconsumed, 0u);
          lds_store(&sh_stepdone, 0u);
          for (int k = 0; k < PIPE_NC; ++k) lds_store(&sh_stored[k], 0u);
        }
        __syncthreads();
        const unsigned q = sh_q;
        if (q >= (unsigned)ntask) {
          __syncthreads(); // everybody has read sh_q before it is written again
          break;
        }
        const int tid = __builtin_amdgcn_readfirstlane(task0 + (int)q);
        const pipe::Task *T = P.tasks + tid;
        const int nsteps = T->nsteps;
        unsigned long long *myword = P.progress + (size_t)tid * 16;
        auto publish = [&](int steps) __attribute__((always_inline)) {
          if (lane == 0) {
            const unsigned long long w = ((unsigned long long)epoch << 32) | (unsigned)steps;
            if (wt) __hip_atomic_store(myword, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(myword), "v"(w) : "memory"); // plain store, no drain
          }
        };

        if (wave >= PIPE_NC) {
          // ---------------- loader: tiles wl, wl + NL, ... of the task by LDS-DMA, up to PIPE_DEPTH tiles in flight ----------------
          const int wl = wave - PIPE_NC;
          const int32_t *ko = P.koff + T->koff_base;
          const unsigned char *src_tiles = P.stream + T->tile_off;
          unsigned v = 0;       // virtual ring offset in KiB
          int head = wl;        // oldest own tile not yet published
          int fl[PIPE_DEPTH];   // pieces of the own tiles in flight
          int nfl = 0;
#pragma unroll
          for (int k = 0; k < PIPE_DEPTH; ++k) fl[k] = 0;
          auto publish_oldest = [&](int younger_pieces) __attribute__((always_inline)) {
            pipe_wait_vmcnt_le(younger_pieces);
            if (lane == 0) lds_store(&sh_ready[head % PIPE_READY], (unsigned)head + 1u);
            head += PIPE_NL;
#pragma unroll
            for (int k = 0; k + 1 < PIPE_DEPTH; ++k) fl[k] = fl[k + 1];
            fl[PIPE_DEPTH - 1] = 0;
            --nfl;
          };
          for (int t = 0; t < nsteps; ++t) {
            const int k0 = ko[t], sz = ko[t + 1] - k0;
            unsigned pos = v % PIPE_RING_KIB;
            if (pos + sz > PIPE_RING_KIB) {
              v += PIPE_RING_KIB - pos;
              pos = 0;
            }
            if (t % PIPE_NL != wl) { // another loader's tile: only its ring space is accounted
              v += sz;
              continue;
            }
            if ((int)(v + sz - lds_load(&sh_vconsumed)) > PIPE_RING_KIB) {
              // about to wait for ring space: first hand over everything that is in flight (the compute waves may need it)
              while (nfl > 0) publish_oldest(0);
              for (unsigned spins = 0; (int)(v + sz - lds_load(&sh_vconsumed)) > PIPE_RING_KIB; ++spins) {
                if (spins > (1u << 24)) {
                  if (lane == 0) __hip_atomic_store(P.err, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                  return;
                }
                __builtin_amdgcn_s_sleep(1);
              }
            }
            const unsigned char *gs = src_tiles + (size_t)k0 * 1024 + lane * 16;
            unsigned char *ld = tiles + pos * 1024;
            for (int p = 0; p < sz; ++p) pipe_glds16(gs + (size_t)p * 1024, ld + p * 1024);
#pragma unroll
            for (int k = 0; k < PIPE_DEPTH; ++k)
              if (k == nfl) fl[k] = sz; // (no dynamic indexing: the array stays in registers)
            ++nfl;
            v += sz;
            if (nfl == PIPE_DEPTH) {
              int younger = 0;
#pragma unroll
              for (int k = 1; k < PIPE_DEPTH; ++k) younger += fl[k];
              publish_oldest(younger);
            }
          }
          while (nfl > 0) publish_oldest(0);
        } else {
          // ---------------- compute wave w: steps w, w + NC, ... ----------------
          const int w = wave;
          const int nprod = T->nprod;
          const int64_t pos_base = T->pos_base;
          // lane l watches producers 2l and 2l + 1 (16-bit requirements, two per header word)
          const unsigned long long *pword0 = P.progress + (size_t)(2 * lane < nprod ? T->prod[2 * lane] : tid) * 16;
          const unsigned long long *pword1 = P.progress + (size_t)(2 * lane + 1 < nprod ? T->prod[2 * lane + 1] : tid) * 16;
          const double *src = pipe_uniform_ptr(upper ? P.xpos : P.ypos);
          const double *rhs = pipe_uniform_ptr(upper ? P.ypos : P.d);
          double *dst = upper ? P.xpos : P.ypos;
          int have0 = 0, have1 = 0;
          unsigned long long st_start = 0, st_first = 0;
          unsigned st_tile = 0, st_prog = 0, st_sum = 0; // cycles (32 bits are plenty for one task)
          unsigned st_nblock = 0, st_cblock = 0; // steps behind the first one that had to poll their producers, cycles spent there
          unsigned st_a = 0, st_b = 0, st_c = 0, st_d = 0, st_t = 0; // step top -> gathers issued -> early part done -> previous step signalled; signal -> step end
          if (STAMP) st_start = __builtin_amdgcn_s_memrealtime();
          if (upper) { // the forward sweep of this subdomain must be complete (its results are this sweep's right-hand side)
            for (unsigned spins = 0; __hip_atomic_load(qbase + 2 * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)ntaskL; ++spins) {
              if (spins > (1u << 22)) {
                if (lane == 0) __hip_atomic_store(P.err, 5u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return;
              }
              __builtin_amdgcn_s_sleep(2);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          }
          // ring positions of the tiles: every compute wave follows ALL tiles (sizes come with the headers it reads)
          auto place = [](unsigned &v, int sz, unsigned &pos) __attribute__((always_inline)) {
            pos = v % PIPE_RING_KIB;
            if (pos + sz > PIPE_RING_KIB) {
              v += PIPE_RING_KIB - pos;
              pos = 0;
            }
            v += sz;
          };
          unsigned v = 0;              // virtual offset in front of this wave's next tile
          int next_kib = T->first_kib; // its size
          if (PIPE_NC == 2 && w == 1) {
            unsigned dummy;
            place(v, next_kib, dummy);
            next_kib = T->second_kib;
          }
          bool failed = false;
          int nstored = 0; // own steps whose result store has completed
          bool publish_pending = false; // the previous step's store (and the poll in front of it) has been issued but not yet drained and published
          unsigned long long pw0 = 0, pw1 = 0; // progress words of the two producers this lane watches (non-blocking poll)
          // waits (polling) until the producers have stored what the lanes need
          auto block_on_producers = [&](int t, int need0, int need1, auto &&on_first_drain) __attribute__((always_inline)) {
            unsigned cb = 0;
            if (STAMP) cb = (unsigned)__builtin_amdgcn_s_memtime();
            for (unsigned spins = 0;; ++spins) {
              { // both words of every lane in one round trip
                unsigned long long q0, q1;
                pipe_poll_asm(pword0, pword1, q0, q1);
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(q0), "+v"(q1)::"memory");
                if ((unsigned)(q0 >> 32) == epoch) have0 = max(have0, (int)(unsigned)q0);
                if ((unsigned)(q1 >> 32) == epoch) have1 = max(have1, (int)(unsigned)q1);
              }
              if (spins == 0) on_first_drain();
              if (__all(have0 >= need0 && have1 >= need1)) break;
              if (spins > (1u << 22)) {
                if (lane == 0) __hip_atomic_store(P.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                failed = true;
                return;
              }
              __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (STAMP && t > w) {
              st_nblock += 1;
              st_cblock += (unsigned)__builtin_amdgcn_s_memtime() - cb;
            }
          };
          auto issue_gathers = [&](PipeStep &S, const int32_t (&op)[PIPE_CHUNK], int own) __attribute__((always_inline)) {
            uint32_t goff[PIPE_CHUNK];
#pragma unroll
            for (int u = 0; u < PIPE_CHUNK; ++u) {
              goff[u] = pipe_gofs(op[u]);
              S.lofs[u] = pipe_lofs(op[u]);
            }
            if (upper) pipe_gather_asm8<true>(rhs, (uint32_t)own, src, goff, S.s0, S.xg);
            else pipe_gather_asm8<false>(rhs, (uint32_t)own, src, goff, S.s0, S.xg);
          };
          // may_defer: when the producers are not far enough, do not wait here (the caller still has the current step to compute):
          // S.deferred is set and fetch_late() finishes the job
          auto fetch = [&](int t, PipeStep &S, bool may_defer, auto &&after_issue) __attribute__((always_inline)) {
            unsigned pos;
            place(v, next_kib, pos);
            S.tpos = pos;
            S.vend = v;
            unsigned c0 = 0;
            if (STAMP) c0 = (unsigned)__builtin_amdgcn_s_memtime();
            // one batch of LDS reads: the ready word first, then everything of the tile that sits at fixed offsets; LDS
            // serves a wave in order, so a matching ready word vouches for the data read behind it (normally the tile has
            // been resident for several steps and this is a single round trip)
            const unsigned char *tile = tiles + pos * 1024;
            const int32_t *hdr = reinterpret_cast<const int32_t *>(tile);
            const int4 *idxp = reinterpret_cast<const int4 *>(tile + 1024 * (1 + PIPE_CHUNK / 2)) + lane;
            int need, own;
            int4 wk; // hdr[2..5]: W, size of tile t+1, late mask (1 step), size of tile t+2
            int late2;
            int32_t op[PIPE_CHUNK];
            bool drained = false; // the caller's hook has waited for every outstanding load: the poll issued with the last store is valid
            for (unsigned spins = 0;; ++spins) {
              asm volatile("" ::: "memory");
              const unsigned rdy_v = lds_load(&sh_ready[t % PIPE_READY]);
              wk = make_int4(hdr[2], hdr[3], hdr[4], hdr[5]);
              late2 = hdr[6];
              need = hdr[pipe::HDR_REQ0 + (lane < 56 ? lane : 0)];
              own = reinterpret_cast<const int32_t *>(tile + 256)[lane];
#pragma unroll
              for (int q4 = 0; q4 < (PIPE_CHUNK + 3) / 4; ++q4) {
                const int4 o = idxp[q4 * 64];
                if (4 * q4 < PIPE_CHUNK) op[4 * q4] = o.x;
                if (4 * q4 + 1 < PIPE_CHUNK) op[4 * q4 + 1] = o.y;
                if (4 * q4 + 2 < PIPE_CHUNK) op[4 * q4 + 2] = o.z;
                if (4 * q4 + 3 < PIPE_CHUNK) op[4 * q4 + 3] = o.w;
              }
              if (spins == 0) drained = after_issue(); // further LDS reads of the caller, queued behind the batch
              const unsigned rdy = (unsigned)__builtin_amdgcn_readfirstlane((int)rdy_v); // (uniform anyway: scalar branch)
              if (rdy == (unsigned)t + 1u) break;
              if (spins > (1u << 24)) {
                if (lane == 0) __hip_atomic_store(P.err, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                failed = true;
                return;
              }
              __builtin_amdgcn_s_sleep(0);
            }
            if (STAMP) {
              const unsigned c = (unsigned)__builtin_amdgcn_s_memtime();
              st_tile += c - c0;
              c0 = c;
            }
            S.W = __builtin_amdgcn_readfirstlane(wk.x);
            S.late = PIPE_NC == 1 ? 0u : (unsigned)__builtin_amdgcn_readfirstlane(PIPE_NC == 2 ? wk.z : late2);
            // this wave's next tile is NC tiles further
            if (PIPE_NC == 1) next_kib = __builtin_amdgcn_readfirstlane(wk.y);
            else {
              unsigned dummy;
              place(v, __builtin_amdgcn_readfirstlane(wk.y), dummy);
              next_kib = __builtin_amdgcn_readfirstlane(wk.w);
            }
            const int need0 = 2 * lane < nprod ? (need & 0xffff) : 0, need1 = 2 * lane + 1 < nprod ? (int)((unsigned)need >> 16) : 0;
            if (drained) { // (only a drain follows a poll)
              asm volatile("" : "+v"(pw0), "+v"(pw1));
              if ((unsigned)(pw0 >> 32) == epoch) have0 = max(have0, (int)(unsigned)pw0);
              if ((unsigned)(pw1 >> 32) == epoch) have1 = max(have1, (int)(unsigned)pw1);
            }
            S.deferred = 0;
            if (!__all(have0 >= need0 && have1 >= need1)) { // producers far enough? (normally yes: they run ahead)
              if (may_defer) {
                S.deferred = 1;
#pragma unroll
                for (int u = 0; u < PIPE_CHUNK; ++u) S.lofs[u] = pipe_lofs(op[u]); // (the ring operands are read at the top of the step)
                if (STAMP) st_prog += (unsigned)__builtin_amdgcn_s_memtime() - c0;
                return;
              }
              block_on_producers(t, need0, need1, []() {});
              if (failed) return;
            }
            if (STAMP) st_prog += (unsigned)__builtin_amdgcn_s_memtime() - c0;
            issue_gathers(S, op, own);
          };
          // steps 0 .. prog-1 are stored: the first step of each compute wave that is not known to be stored bounds it
          auto publish_progress = [&]() __attribute__((always_inline)) {
            int prog = w + PIPE_NC * nstored;
#pragma unroll
            for (int k = 0; k < PIPE_NC; ++k)
              if (k != w) prog = min(prog, k + PIPE_NC * (int)lds_load(&sh_stored[k]));
            publish(prog);
          };
          // second half of a deferred fetch(), behind the caller's own step: the operand words again (the tile is resident), the
          // wait for the producers, the gathers.  The first poll's drain also covers the caller's result store: published there.
          auto fetch_late = [&](int t, PipeStep &S, auto &&on_first_drain) __attribute__((always_inline)) {
            const unsigned char *tile = tiles + S.tpos * 1024;
            const int4 *idxp = reinterpret_cast<const int4 *>(tile + 1024 * (1 + PIPE_CHUNK / 2)) + lane;
            const int need = reinterpret_cast<const int32_t *>(tile)[pipe::HDR_REQ0 + (lane < 56 ? lane : 0)];
            const int own = reinterpret_cast<const int32_t *>(tile + 256)[lane];
            int32_t op[PIPE_CHUNK];
#pragma unroll
            for (int q4 = 0; q4 < (PIPE_CHUNK + 3) / 4; ++q4) {
              const int4 o = idxp[q4 * 64];
              if (4 * q4 < PIPE_CHUNK) op[4 * q4] = o.x;
              if (4 * q4 + 1 < PIPE_CHUNK) op[4 * q4 + 1] = o.y;
              if (4 * q4 + 2 < PIPE_CHUNK) op[4 * q4 + 2] = o.z;
              if (4 * q4 + 3 < PIPE_CHUNK) op[4 * q4 + 3] = o.w;
            }
            const int need0 = 2 * lane < nprod ? (need & 0xffff) : 0, need1 = 2 * lane + 1 < nprod ? (int)((unsigned)need >> 16) : 0;
            block_on_producers(t, need0, need1, on_first_drain);
            if (failed) return;
            issue_gathers(S, op, own);
            S.deferred = 0;
          };
          // one step: "k4" = number of leading groups of 4 entries that no lane takes from the previous NC - 1 steps
          auto step = [&](int t, PipeStep &cur, PipeStep &nxt) __attribute__((always_inline)) {
            asm volatile("" ::: "memory");
            if (STAMP) st_t = (unsigned)__builtin_amdgcn_s_memtime();
            double xl[PIPE_CHUNK], p[PIPE_CHUNK], dinv; // p: factor entries, then the products
            const unsigned char *ctile = tiles + cur.tpos * 1024; // this step's tile stays resident until it is released below
            // LDS serves the wave in order: the next tile's header and operand words first (fetch waits for them at once, so they
            // must not queue behind anything), then the ring operands of this step, then this step's factor entries
            // rows wider than the register chunk (steps next to the overlap shell): their further operand words are read first
            // of all, the gathers of those PIPE_WIDE operands are issued in front of the next step's gathers (so the explicit
            // wait below covers them) and consumed behind the row sum of the first PIPE_CHUNK entries
            const int W = __builtin_amdgcn_readfirstlane(cur.W);
            const unsigned char *wtile = ctile;
            const pipe::Geometry G(W);
            // (two halves of PIPE_WHALF entries; the operand words of a half are one batch of LDS reads: pieces that the tile
            // does not have are read from its last piece instead and masked)
            auto wide_ops = [&](int u0, int32_t (&o)[PIPE_WHALF]) __attribute__((always_inline)) {
              const int q0 = u0 >> 2, last = G.idx_pieces - 1;
              const int4 a = *reinterpret_cast<const int4 *>(wtile + 1024 * G.idx_piece(min(q0, last)) + lane * 16);
              const int4 b = *reinterpret_cast<const int4 *>(wtile + 1024 * G.idx_piece(min(q0 + 1, last)) + lane * 16);
              // u0 = 14: entries 14, 15 | 16 .. 19;  u0 = 20: entries 20 .. 23 | 24, 25
              if ((u0 & 3) == 2) {
                o[0] = a.z, o[1] = a.w, o[2] = b.x, o[3] = b.y, o[4] = b.z, o[5] = b.w;
              } else {
                o[0] = a.x, o[1] = a.y, o[2] = a.z, o[3] = a.w, o[4] = b.x, o[5] = b.y;
              }
#pragma unroll
              for (int k = 0; k < PIPE_WHALF; ++k)
                if (u0 + k >= W) o[k] = pipe::PAD_OP; // (wave-uniform)
            };
            static_assert(PIPE_CHUNK == 14 && PIPE_WHALF == 6, "piece arithmetic of wide_ops");
            uint32_t ego0[PIPE_WHALF], ego1[PIPE_WHALF];
            double eg0[PIPE_WHALF], eg1[PIPE_WHALF];
            if (W > PIPE_CHUNK) {
              int32_t o[PIPE_WHALF];
              wide_ops(PIPE_CHUNK, o);
#pragma unroll
              for (int k = 0; k < PIPE_WHALF; ++k) ego0[k] = pipe_gofs(o[k]);
              if (W > PIPE_CHUNK + PIPE_WHALF) {
                wide_ops(PIPE_CHUNK + PIPE_WHALF, o);
#pragma unroll
                for (int k = 0; k < PIPE_WHALF; ++k) ego1[k] = pipe_gofs(o[k]);
              }
            }
            auto read_ring = [&]() __attribute__((always_inline)) -> bool {
              bool drained = false;
#pragma unroll
              for (int u = 0; u < PIPE_CHUNK; ++u) xl[u] = pipe_lds_f64(cur.lofs[u]);
              if (publish_pending) {
                drained = true; // the store of the previous step has had these instructions' time to complete
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                nstored = (t - w) / PIPE_NC;
                if (lane == 0) lds_store(&sh_stored[w], (unsigned)nstored);
                publish_progress();
                publish_pending = false;
              }
              if (W > PIPE_CHUNK) pipe_gather_asm_wide(src, ego0, eg0); // (behind the drain above, in front of the next step's gathers)
              if (W > PIPE_CHUNK + PIPE_WHALF) pipe_gather_asm_wide(src, ego1, eg1);
              return drained;
            };
            auto read_entries = [&]() __attribute__((always_inline)) {
              const double2 *valp = reinterpret_cast<const double2 *>(ctile + 1024) + lane;
#pragma unroll
              for (int q2 = 0; q2 < PIPE_CHUNK / 2; ++q2) {
                const double2 vv = valp[q2 * 64];
                p[2 * q2] = vv.x;
                p[2 * q2 + 1] = vv.y;
              }
              dinv = reinterpret_cast<const double *>(ctile + 512)[lane];
            };
            const bool fetched_next = t + PIPE_NC < nsteps;
            if (fetched_next) fetch(t + PIPE_NC, nxt, true, read_ring);
            else read_ring();
            if (failed) return;
            if (STAMP) {
              const unsigned c = (unsigned)__builtin_amdgcn_s_memtime();
              st_a += c - st_t;
              st_t = c;
            }
            read_entries(); // (behind the fetch: its temporaries and the factor entries need not be live together)
            // behind this step's gathers: the result store of this wave's previous step and, if there is a next step, its
            // PIPE_INFLIGHT loads => at most that many operations may still be outstanding
            // (the drain for the last step has no register operands: no copies of in-flight registers in front of it)
            const bool deferred_next = fetched_next && __builtin_amdgcn_readfirstlane(nxt.deferred) != 0; // (no gathers of the next step in flight)
            if (!fetched_next || deferred_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pipe_wait_gathers<PIPE_INFLIGHT>(cur.s0, cur.xg);
            if (W > PIPE_CHUNK) pipe_pin_wide(eg0); // (older than the gathers the wait left in flight: valid too)
            if (W > PIPE_CHUNK + PIPE_WHALF) pipe_pin_wide(eg1);
            // Products of the groups of 4 entries that no lane takes from the previous NC - 1 steps; in the other groups p keeps
            // the factor entry until the ring operands can be read (behind the previous step's signal).  The head of the row
            // sum up to the first such group is final already.
            const unsigned late = __builtin_amdgcn_readfirstlane(cur.late);
            auto group = [&](int g0, bool final_operands) __attribute__((always_inline)) {
#pragma unroll
              for (int u = g0; u < g0 + 4 && u < PIPE_CHUNK; ++u) {
                if (final_operands) p[u] = p[u] * pipe_or(cur.xg[u], xl[u]);
              }
            };
            group(0, (late & 0x000fu) == 0u);
            group(4, (late & 0x00f0u) == 0u);
            group(8, (late & 0x0f00u) == 0u);
            group(12, (late & 0xf000u) == 0u);
            const int k4 = late == 0u ? 4 : (__builtin_ctz(late) >> 2); // first group with late operands (4: none)
            double s = cur.s0;
            if (k4 >= 1) s = (((s - p[0]) - p[1]) - p[2]) - p[3];
            if (k4 >= 2) s = (((s - p[4]) - p[5]) - p[6]) - p[7];
            if (k4 >= 3) s = (((s - p[8]) - p[9]) - p[10]) - p[11];
            if (k4 >= 4) s = (s - p[12]) - p[13];
            // everything above is computed BEFORE the wait below (the compiler would sink it behind the wait otherwise)
            pipe_pin(s, p);
            if (STAMP) {
              const unsigned c = (unsigned)__builtin_amdgcn_s_memtime();
              st_b += c - st_t;
              st_t = c;
            }
            unsigned c0 = 0;
            if (PIPE_NC > 1) { // the previous step's results must be in the ring now
              for (unsigned spins = 0; (unsigned)__builtin_amdgcn_readfirstlane((int)lds_load(&sh_stepdone)) < (unsigned)t; ++spins) {
                if (spins > (1u << 24)) {
                  if (lane == 0) __hip_atomic_store(P.err, 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                  failed = true;
                  return;
                }
              }
              asm volatile("" ::: "memory");
              if (STAMP) {
                c0 = (unsigned)__builtin_amdgcn_s_memtime();
                st_c += c0 - st_t;
              }
              auto late_group = [&](int g0) __attribute__((always_inline)) {
                double x2[4];
#pragma unroll
                for (int u = g0; u < g0 + 4 && u < PIPE_CHUNK; ++u) x2[u - g0] = pipe_lds_f64(cur.lofs[u]);
#pragma unroll
                for (int u = g0; u < g0 + 4 && u < PIPE_CHUNK; ++u) p[u] = p[u] * pipe_or(cur.xg[u], x2[u - g0]);
              };
              if (late & 0x000fu) late_group(0);
              if (late & 0x00f0u) late_group(4);
              if (late & 0x0f00u) late_group(8);
              if (late & 0xf000u) late_group(12);
            }
            if (k4 < 1) s = (((s - p[0]) - p[1]) - p[2]) - p[3];
            if (k4 < 2) s = (((s - p[4]) - p[5]) - p[6]) - p[7];
            if (k4 < 3) s = (((s - p[8]) - p[9]) - p[10]) - p[11];
            if (k4 < 4) s = (s - p[12]) - p[13];
            if (W > PIPE_CHUNK) {
              // a half: operand words and factor entries in one batch of LDS reads, then the ring operands, then the row sum
              auto wide_half = [&](int u0, const double (&eg)[PIPE_WHALF]) __attribute__((always_inline)) {
                int32_t o[PIPE_WHALF];
                wide_ops(u0, o);
                const int lastv = G.val_pieces - 1;
                double av[PIPE_WHALF], xr[PIPE_WHALF];
#pragma unroll
                for (int k = 0; k < PIPE_WHALF / 2; ++k) {
                  const double2 vv = *reinterpret_cast<const double2 *>(wtile + 1024 * G.val_piece(min((u0 >> 1) + k, lastv)) + lane * 16);
                  av[2 * k] = vv.x;
                  av[2 * k + 1] = vv.y;
                }
#pragma unroll
                for (int k = 0; k < PIPE_WHALF; ++k) xr[k] = pipe_lds_f64(pipe_lofs(o[k]));
#pragma unroll
                for (int k = 0; k < PIPE_WHALF; ++k)
                  if (u0 + k < W) s -= av[k] * pipe_or(eg[k], xr[k]); // (wave-uniform)
              };
              wide_half(PIPE_CHUNK, eg0);
              if (W > PIPE_CHUNK + PIPE_WHALF) wide_half(PIPE_CHUNK + PIPE_WHALF, eg1);
              for (int u = PIPE_CHUNK + PIPE_WIDE; u < W; ++u) { // still wider (not seen on the stencils of SURVEY section 8): entry by entry
                const int32_t o = *reinterpret_cast<const int32_t *>(wtile + G.idx_off(u, lane));
                const double av = *reinterpret_cast<const double *>(wtile + G.val_off(u, lane));
                s -= av * pipe_or(pipe_ld_sc1_off(src, pipe_gofs(o)), pipe_lds_f64(pipe_lofs(o)));
              }
            }
            const double out = s * dinv; // (forward sweep: scale 1)
            ringd[(t % pipe::RING) * 64 + lane] = out;
            asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(out) : "memory"); // the result is in the ring, every LDS read of this step has returned
            if (lane == 0) {
              lds_store(&sh_vconsumed, cur.vend);       // the tile's ring space goes back to the loaders ...
              lds_store(&sh_stepdone, (unsigned)t + 1u); // ... and the next step may read the ring
            }
            if (STAMP && PIPE_NC > 1) {
              st_t = (unsigned)__builtin_amdgcn_s_memtime();
              st_sum += st_t - c0;
            }
            const int64_t mypos = pos_base + (int64_t)t * 64 + lane;
            pipe_poll_asm(pword0, pword1, pw0, pw1); // (as late as possible: the drain of the next step is the first to need it)
            if (wt) st_sc1(dst + mypos, out);
            else dst[mypos] = out;
            publish_pending = true; // drained and published at the top of the next step, behind its first LDS reads
            if (deferred_next) { // the next step's producers were not far enough at the top of this step: now is the time to wait for them
              fetch_late(t + PIPE_NC, nxt, [&]() {
                nstored = (t - w) / PIPE_NC + 1; // (the drain behind the first poll has completed this step's store)
                if (lane == 0) lds_store(&sh_stored[w], (unsigned)nstored);
                publish_progress();
                publish_pending = false;
              });
              if (failed) return;
            }
            if (STAMP && w == 0 && lane == 0 && P.stamps && t < PIPE_STAMP_WORDS - 16) P.stamps[(size_t)tid * PIPE_STAMP_WORDS + 16 + t] = __builtin_amdgcn_s_memrealtime();
            if (STAMP) st_d += (unsigned)__builtin_amdgcn_s_memtime() - st_t;
          };
          if (w < nsteps) {
            PipeStep SA, SB;
            fetch(w, SA, false, []() { return false; });
            if (failed) return;
            if (STAMP) st_first = __builtin_amdgcn_s_memrealtime();
            for (int t = w; t < nsteps; t += 2 * PIPE_NC) {
              step(t, SA, SB);
              if (failed) return;
              if (t + PIPE_NC < nsteps) step(t + PIPE_NC, SB, SA);
              if (failed) return;
            }
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every result store of this wave has completed
          if (STAMP && w == 0 && lane == 0 && P.stamps) {
            unsigned long long *o = P.stamps + (size_t)tid * PIPE_STAMP_WORDS;
            o[0] = st_start;
            o[1] = st_first;
            o[2] = __builtin_amdgcn_s_memrealtime();
            o[3] = st_tile;
            o[4] = st_prog;
            o[5] = st_sum;
            o[6] = (unsigned long long)nsteps;
            o[7] = xcc;
            o[8] = st_a;
            o[9] = st_b;
            o[10] = st_c;
            o[11] = st_d;
            o[12] = st_nblock;
            o[13] = st_cblock;
          }
        }
        __syncthreads(); // all steps are stored (every compute wave has drained)
        if (threadIdx.x == 0) {
          publish(nsteps);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_fetch_add(qbase + (2 + sweep) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
}

} // namespace ddm
