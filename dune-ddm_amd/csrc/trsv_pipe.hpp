// Device side of the "pipe" triangular-solve engine; schedule format and rationale: trsv_pipe_host.hpp.
//
// One workgroup = one compute wave + PIPE_NL loader waves; two workgroups per CU.  Workgroups pull tasks (64 chains walked level by level)
// from a per-(subdomain, sweep) queue in topological order.  The loader waves stream the task's tiles HBM -> LDS
// with LDS-DMA (global_load_lds_dwordx4, 1 KiB per instruction) into a byte ring, far ahead of the compute wave.
// The compute wave per step: operands of the own task from the LDS result ring, operands of other tasks by sc1
// gathers from the position arrays (guarded by the producers' progress words, which are normally far ahead), the
// row sum in ascending column order (= the sequential back-solve), one coalesced 512-byte store of the 64 results.
//
// Visibility protocol (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"):
//   * XCD-local mode (>= 8 subdomains, every XCD hosts workgroups): all tasks of a subdomain run on ONE XCD; results
//     are plain stores (they stay in that XCD's L2), every read of another wave's result is an sc1 load (bypasses
//     the non-coherent L1), the progress word is stored after the result stores have completed;
//   * placement-independent mode: results and progress words are sc1 (write-through) stores.
//   * "completed": vmcnt retires in issue order, so once a load issued after a store has returned, the store has
//     completed; PIPE_LAZY publishes on that basis, otherwise the wave drains (s_waitcnt vmcnt(0)) before publishing.
#pragma once
#include "trsv_pipe_host.hpp"

namespace ddm {

constexpr int PIPE_NL = 2;          // loader waves per workgroup (two workgroups per CU = 6 waves: at most 256 VGPRs per wave)
constexpr int PIPE_DEPTH = 3;       // tiles a loader keeps in flight
constexpr int PIPE_RING_KIB = 64;   // LDS byte ring of tiles per workgroup
constexpr int PIPE_READY = 32;      // ready words (tiles in flight < 32: the smallest tile is 3 KiB)
constexpr int PIPE_CHUNK = pipe::MIN_W; // entries of a row held in registers; every tile has at least that many (padded), wider rows take the rest from the tile
constexpr int PIPE_RINGAREA = (pipe::RING_BYTES + 4 * (PIPE_READY + 8) + 1023) / 1024 * 1024; // result ring + control words
constexpr size_t PIPE_LDS_BYTES = (size_t)PIPE_RINGAREA + (size_t)PIPE_RING_KIB * 1024; // dynamic part

struct PipeStep { // registers of one step, filled one step ahead of their use (the factor entries themselves are read
                  // from the tile when the step is computed: two sets of them would not fit the register budget)
  int32_t op[PIPE_CHUNK];
  double xg[PIPE_CHUNK];
  double s0;
  int W;
  unsigned tpos, vend; // tile position in the LDS ring (KiB), virtual ring offset behind the tile
};
__device__ __forceinline__ double pipe_ld_sc1_off(const double *base, uint32_t byte_off)
{
  const unsigned long long *p = reinterpret_cast<const unsigned long long *>(reinterpret_cast<const unsigned char *>(base) + byte_off);
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
__device__ __forceinline__ void pipe_gather_asm8(const double *rhs_uniform, uint32_t own, const double *src_uniform, const uint32_t (&off)[PIPE_CHUNK], double &s0,
                                                 double (&x)[PIPE_CHUNK])
{
  static_assert(PIPE_CHUNK == 14, "operand lists below");
  asm volatile("s_nop 4\n\t"
               "global_load_dwordx2 %0, %8, %16 sc1\n\t"
               "global_load_dwordx2 %1, %9, %17 sc1\n\t"
               "global_load_dwordx2 %2, %10, %17 sc1\n\t"
               "global_load_dwordx2 %3, %11, %17 sc1\n\t"
               "global_load_dwordx2 %4, %12, %17 sc1\n\t"
               "global_load_dwordx2 %5, %13, %17 sc1\n\t"
               "global_load_dwordx2 %6, %14, %17 sc1\n\t"
               "global_load_dwordx2 %7, %15, %17 sc1"
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
__device__ __forceinline__ uint32_t pipe_lofs(int32_t op) { return min((uint32_t)op, (uint32_t)pipe::RING_Z); } // LDS byte offset (zero row if remote)
__device__ __forceinline__ uint32_t pipe_gofs(int32_t op) { return max((uint32_t)op, (uint32_t)pipe::RING_Z); } // global byte offset (the reserved zero if local)
__device__ __forceinline__ double pipe_or(double a, double b) { return __longlong_as_double(__double_as_longlong(a) | __double_as_longlong(b)); }

struct PipeParams {
  int ngroups;
  const pipe::Group *groups;
  const pipe::Task *tasks;
  const unsigned char *stream;
  const int32_t *koff;
  const double *dperm;          // right-hand side in L position order
  double *ypos, *xpos;          // forward / backward results in position order
  unsigned long long *progress; // one word per task at stride 16 (128 B): (epoch << 32) | steps stored
  unsigned *queue;              // per group 4 words at stride 32: next task of L, of U; finished tasks of L, of U
  XcdState *st;
  unsigned *err;
  unsigned long long *stamps;   // diagnostics (nullptr in the product path)
  unsigned long long *dbg;      // diagnostics: 8 words describing the first out-of-range operand (stamped build only)
  unsigned nposL_bytes, nposU_bytes;
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
// dperm[pos] = d[row(pos)] (0 on padding positions)
__global__ void k_pipe_permute_in(int64_t npos, const int32_t *__restrict__ rowL, const double *__restrict__ d, double *__restrict__ dperm)
{
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npos; p += (int64_t)gridDim.x * blockDim.x) {
    const int32_t r = rowL[p];
    dperm[p] = r >= 0 ? d[r] : 0.0;
  }
}
__global__ void k_pipe_permute_out(int64_t n, const int32_t *__restrict__ posU, const double *__restrict__ xpos, double *__restrict__ x)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] = xpos[posU[i]];
}

__device__ __forceinline__ void pipe_glds16(const unsigned char *gsrc, unsigned char *lds_dst)
{
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc, (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

// STAMP (diagnostic build only): per task 8 words -- start / first step / end (s_memrealtime, 100 MHz), cycles waiting for
// tiles / for producers / in the gather+sum part (s_memtime), steps, XCC id
template <bool LAZY, bool STAMP>
__global__ __launch_bounds__(64 * (1 + PIPE_NL)) void k_trsv_pipe(PipeParams P)
{
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  // All LDS is dynamic and the result ring sits at LDS address 0, so that ring operands are ds_read addresses as they
  // stand.  Control words between the loader and the compute wave: relaxed workgroup-scope atomics (plain ds_read /
  // ds_write; ordering is by the s_waitcnt instructions below).
  double *ringd = reinterpret_cast<double *>(smem); // [RING rows of 64 results][zero row]
  unsigned *ctl = reinterpret_cast<unsigned *>(smem + pipe::RING_BYTES);
  unsigned *sh_ready = ctl;                    // [PIPE_READY]
  unsigned &sh_vconsumed = ctl[PIPE_READY];
  unsigned &sh_xcc = ctl[PIPE_READY + 1], &sh_gt = ctl[PIPE_READY + 2], &sh_fail = ctl[PIPE_READY + 3], &sh_q = ctl[PIPE_READY + 4];
  unsigned char *tiles = smem + PIPE_RINGAREA; // byte ring of tiles
  auto lds_load = [](const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  auto lds_store = [](unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  XcdState *st = P.st;
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) { // static LDS in front of the ring: not this build
    if (threadIdx.x == 0) __hip_atomic_store(P.err, 6u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (threadIdx.x == 0) {
    const unsigned xcc = hw_xcc_id();
    sh_xcc = xcc;
    __hip_atomic_fetch_add(&st->tickets[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh_gt = __hip_atomic_fetch_add(&st->global_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&st->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned fail_ = 0;
    for (unsigned spins = 0; __hip_atomic_load(&st->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x; ++spins) {
      if (spins > (1u << 22)) {
        __hip_atomic_store(P.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
  const unsigned epoch = __hip_atomic_load(&st->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned tk = lane < 8 ? __hip_atomic_load(&st->tickets[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
  const bool local_ok = P.ngroups >= 8 && __all(tk >= 1u); // every XCD hosts workgroups: subdomain g lives on XCD g % 8
  const bool wt = !local_ok;
  const int gfirst = local_ok ? (int)xcc : (int)(sh_gt % (unsigned)P.ngroups);
  const int gcount = local_ok ? (P.ngroups - (int)xcc + 7) / 8 : P.ngroups;

  for (int gi = 0; gi < gcount; ++gi) {
    const int g = local_ok ? gfirst + 8 * gi : (gfirst + gi) % P.ngroups;
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
        const int32_t *ko = P.koff + T->koff_base;
        const unsigned char *src_tiles = P.stream + T->tile_off;

        if (wave > 0) {
          // ---------------- loader: all tiles of the task by LDS-DMA, up to PIPE_DEPTH tiles in flight ----------------
          unsigned v = 0;       // virtual ring offset in KiB
          int head = wave - 1;  // oldest own tile not yet published (this wave loads tiles wave-1, wave-1+NL, ...)
          int fl[PIPE_DEPTH];   // pieces of the tiles in flight (head, head+1, ...)
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
            if (t % PIPE_NL != wave - 1) { // another loader's tile: only its ring space is accounted
              v += sz;
              continue;
            }
            if ((int)(v + sz - lds_load(&sh_vconsumed)) > PIPE_RING_KIB) {
              // about to wait for ring space: first hand over everything that is in flight (the compute wave may need it)
              while (nfl > 0) publish_oldest(0);
              for (unsigned spins = 0; (int)(v + sz - lds_load(&sh_vconsumed)) > PIPE_RING_KIB; ++spins) {
                if (spins > (1u << 24)) {
                  if (lane == 0) __hip_atomic_store(P.err, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
          // ---------------- compute wave ----------------
          // Software pipeline: fetch(t + 1) (tile -> registers, progress check, remote gathers issued) runs between the
          // LDS ring reads of step t and its row sums, so the L2 latency of the gathers of step t + 1 overlaps step t.
          const int nprod = T->nprod;
          const int64_t pos_base = T->pos_base;
          const unsigned long long *pword = P.progress + (size_t)(lane < nprod ? T->prod[lane] : tid) * 16;
          unsigned long long *myword = P.progress + (size_t)tid * 16;
          const double *src = pipe_uniform_ptr(upper ? P.xpos : P.ypos);
          const double *rhs = pipe_uniform_ptr(upper ? P.ypos : P.dperm);
          double *dst = upper ? P.xpos : P.ypos;
          int have = 0;
          unsigned long long st_start = 0, st_first = 0;
          unsigned st_tile = 0, st_prog = 0, st_sum = 0; // cycles (32 bits are plenty for one task)
          if (STAMP) st_start = __builtin_amdgcn_s_memrealtime();
          for (int k = 0; k <= pipe::RING; ++k) ringd[k * 64 + lane] = 0.0; // the last row stays zero
          if (upper) { // the forward sweep of this subdomain must be complete (its results are this sweep's right-hand side)
            for (unsigned spins = 0; __hip_atomic_load(qbase + 2 * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)ntaskL; ++spins) {
              if (spins > (1u << 22)) {
                if (lane == 0) __hip_atomic_store(P.err, 5u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
              }
              __builtin_amdgcn_s_sleep(2);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          }
          auto publish = [&](int steps) __attribute__((always_inline)) {
            if (lane == 0) {
              const unsigned long long w = ((unsigned long long)epoch << 32) | (unsigned)steps;
              if (wt) __hip_atomic_store(myword, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              else asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(myword), "v"(w) : "memory"); // plain store, no drain
            }
          };
          unsigned v = 0;
          int next_kib = T->first_kib;
          bool failed = false;
          auto fetch = [&](int t, PipeStep &S) __attribute__((always_inline)) {
            const int sz = next_kib; // (from the previous tile's header: no descriptor loads on this wave's path)
            unsigned pos = v % PIPE_RING_KIB;
            if (pos + sz > PIPE_RING_KIB) {
              v += PIPE_RING_KIB - pos;
              pos = 0;
            }
            v += sz;
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
            int2 wk;
            for (unsigned spins = 0;; ++spins) {
              asm volatile("" ::: "memory");
              const unsigned rdy = lds_load(&sh_ready[t % PIPE_READY]);
              wk = *reinterpret_cast<const int2 *>(hdr + 2);
              need = hdr[pipe::HDR_REQ0 + (lane < nprod ? lane : 0)];
              own = reinterpret_cast<const int32_t *>(tile + 256)[lane];
#pragma unroll
              for (int q4 = 0; q4 < (PIPE_CHUNK + 3) / 4; ++q4) {
                const int4 o = idxp[q4 * 64];
                if (4 * q4 < PIPE_CHUNK) S.op[4 * q4] = o.x;
                if (4 * q4 + 1 < PIPE_CHUNK) S.op[4 * q4 + 1] = o.y;
                if (4 * q4 + 2 < PIPE_CHUNK) S.op[4 * q4 + 2] = o.z;
                if (4 * q4 + 3 < PIPE_CHUNK) S.op[4 * q4 + 3] = o.w;
              }
              if (rdy == (unsigned)t + 1u) break;
              if (spins > (1u << 24)) {
                if (lane == 0) __hip_atomic_store(P.err, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
            next_kib = __builtin_amdgcn_readfirstlane(wk.y);
            if (lane >= nprod) need = 0;
            if (!__all(have >= need)) { // producers far enough? (normally yes: they run ahead)
              for (unsigned spins = 0;; ++spins) {
                if (have < need) {
                  const unsigned long long w = __hip_atomic_load(pword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  if ((unsigned)(w >> 32) == epoch) have = (int)(unsigned)w;
                }
                if (__all(have >= need)) break;
                if (spins > (1u << 22)) {
                  if (lane == 0) __hip_atomic_store(P.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  failed = true;
                  return;
                }
                __builtin_amdgcn_s_sleep(1);
              }
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            if (STAMP) st_prog += (unsigned)__builtin_amdgcn_s_memtime() - c0;
            uint32_t goff[PIPE_CHUNK];
#pragma unroll
            for (int u = 0; u < PIPE_CHUNK; ++u) goff[u] = pipe_gofs(S.op[u]);
            pipe_gather_asm8(rhs, (uint32_t)own, src, goff, S.s0, S.xg);
          };
          auto finish = [&](int t, PipeStep &S, const double(&xl)[PIPE_CHUNK], const double(&a)[PIPE_CHUNK], double dinv, bool fetched_next) __attribute__((always_inline)) {
            unsigned c0 = 0;
            if (STAMP) c0 = (unsigned)__builtin_amdgcn_s_memtime();
            // behind this step's gathers: the result store of the previous step and, if there is a next step, its
            // PIPE_CHUNK + 1 gathers => at most that many operations may still be outstanding
            // (the drain for the last step has no register operands: no copies of in-flight registers in front of it)
            if (!fetched_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pipe_wait_gathers<PIPE_CHUNK + 1>(S.s0, S.xg);
            double s = S.s0;
#pragma unroll
            for (int u = 0; u < PIPE_CHUNK; ++u) s -= a[u] * pipe_or(S.xg[u], xl[u]);
            if (S.W > PIPE_CHUNK) { // wide rows (rare): the rest of the row from the tile, which is still resident, in chunks
              const unsigned char *tile = tiles + S.tpos * 1024;
              const pipe::Geometry G(S.W);
              constexpr int WCH = 6; // (small chunks: this path must not cost the fast path registers)
              for (int u0 = PIPE_CHUNK; u0 < S.W; u0 += WCH) {
                int32_t o[WCH];
                double av[WCH], g[WCH], l[WCH];
#pragma unroll
                for (int u = 0; u < WCH; ++u) {
                  o[u] = pipe::PAD_OP;
                  av[u] = 0.0;
                  if (u0 + u < S.W) { // wave-uniform
                    o[u] = *reinterpret_cast<const int32_t *>(tile + G.idx_off(u0 + u, lane));
                    av[u] = *reinterpret_cast<const double *>(tile + G.val_off(u0 + u, lane));
                  }
                }
#pragma unroll
                for (int u = 0; u < WCH; ++u) {
                  g[u] = 0.0;
                  if (u0 + u < S.W) g[u] = pipe_ld_sc1_off(src, pipe_gofs(o[u]));
                  l[u] = pipe_lds_f64(pipe_lofs(o[u]));
                }
#pragma unroll
                for (int u = 0; u < WCH; ++u) s -= av[u] * pipe_or(g[u], l[u]);
              }
            }
            const double out = s * dinv; // (forward sweep: scale 1)
            asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(out) : "memory"); // "out" is complete: every load of this step has returned
            if (lane == 0) lds_store(&sh_vconsumed, S.vend);             // the tile's ring space goes back to the loaders
            ringd[(t % pipe::RING) * 64 + lane] = out;
            // the wait above left at most the next step's gathers outstanding; the result store of step t-1 is older than those
            if (LAZY) publish(t);
            const int64_t mypos = pos_base + (int64_t)t * 64 + lane;
            if (wt) st_sc1(dst + mypos, out);
            else dst[mypos] = out;
            if (!LAZY) {
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              publish(t + 1);
            }
            if (STAMP) st_sum += (unsigned)__builtin_amdgcn_s_memtime() - c0;
          };
          auto step = [&](int t, PipeStep &cur, PipeStep &nxt) __attribute__((always_inline)) {
            asm volatile("" ::: "memory");
            double xl[PIPE_CHUNK], a[PIPE_CHUNK];
#pragma unroll
            for (int u = 0; u < PIPE_CHUNK; ++u) xl[u] = pipe_lds_f64(pipe_lofs(cur.op[u]));
            const unsigned char *ctile = tiles + cur.tpos * 1024; // this step's tile stays resident until finish() releases it
            const double2 *valp = reinterpret_cast<const double2 *>(ctile + 1024) + lane;
#pragma unroll
            for (int q2 = 0; q2 < PIPE_CHUNK / 2; ++q2) {
              const double2 vv = valp[q2 * 64];
              a[2 * q2] = vv.x;
              a[2 * q2 + 1] = vv.y;
            }
            const double dinv = reinterpret_cast<const double *>(ctile + 512)[lane];
            asm volatile("" ::: "memory"); // these reads are issued before the next tile's
            if (t + 1 < nsteps) fetch(t + 1, nxt);
            if (failed) return;
            finish(t, cur, xl, a, dinv, t + 1 < nsteps);
          };
          PipeStep SA, SB;
          fetch(0, SA);
          if (failed) return;
          if (STAMP) st_first = __builtin_amdgcn_s_memrealtime();
          for (int t = 0; t < nsteps; t += 2) {
            step(t, SA, SB);
            if (failed) return;
            if (t + 1 < nsteps) step(t + 1, SB, SA);
            if (failed) return;
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          publish(nsteps);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) __hip_atomic_fetch_add(qbase + (2 + sweep) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (STAMP && lane == 0 && P.stamps) {
            unsigned long long *o = P.stamps + (size_t)tid * 8;
            o[0] = st_start;
            o[1] = st_first;
            o[2] = __builtin_amdgcn_s_memrealtime();
            o[3] = st_tile;
            o[4] = st_prog;
            o[5] = st_sum;
            o[6] = (unsigned long long)nsteps;
            o[7] = xcc;
          }
        }
        __syncthreads();
      }
    }
  }
}

} // namespace ddm
