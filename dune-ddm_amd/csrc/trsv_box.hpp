// Device side of the "box" triangular-solve engine; data layout and rationale: trsv_box_host.hpp.
//
// k_box_sweep<UPPER>: one workgroup = one compute wavefront + one prefetch wavefront, working on one plane K of one block at a time
// (planes are handed out in order by a ticket per block and sweep, so the plane a wave waits for is always held by a running wave).
//   compute wave, step l (rows I = l - 2 J of the lines J = lane + 64 c):
//     * operands of the own plane: x(I-1, J) is the lane's previous result (register); the line J-1 is the neighbour lane's line -- its
//       newest value comes out of a two-slot LDS ring, the two older ones were read in the steps before (register window);
//     * operands of plane K-1: three register windows (lines J-1, J, J+1), one new column per step, read from the plane's
//       position-ordered results xs[K-1][step][lane] with sc1 loads THREE steps ahead (guarded by the plane's progress word);
//     * the row's 14 stream values (7 sixteen-byte loads), its right-hand side entry, (backward) its products with the shell and the
//       product descriptor of three steps later: all requested three steps ahead (three register sets, the step loop is unrolled
//       three times; the products travel by LDS-DMA into a four-slot ring), every step issues the same number of vector-memory
//       operations, so one s_waitcnt vmcnt(N) with a constant N is the hand-over between the steps (vmcnt retires in issue order);
//     * row sum in ascending column order, every product rounded before it is subtracted (-ffp-contract=off): the sequential solve.
//   prefetch wave: touches the stream tiles, right-hand side lines and product descriptors BOX_AHEAD steps ahead of the compute wave, so
//     that the compute wave's two-step request distance only has to cover an L2 hit.
// Visibility as in the pipe engine: a block's planes run on ONE XCD (workgroups read their XCC id and serve the blocks g = xcc mod 8);
// results are plain stores (they stay in that XCD's L2), reads of another wave's results are sc1 loads; without that placement
// (fewer XCDs with workgroups than blocks need, or P.spread) results and progress words are agent-scope stores.
#pragma once
#include "trsv_box_host.hpp"
#include <type_traits>

namespace ddm {

constexpr int BOX_WG = 128;
constexpr int BOX_NEL = box::MAX_EXT / 2;   // two-double loads of products per row
constexpr int BOX_AHEAD = 8;                // steps the prefetch wave runs ahead of the compute wave
constexpr int BOX_DIST = 3;                 // steps between a request and its use (three register sets, the step loop is unrolled three times)
constexpr int BOX_ESLOT = BOX_NEL * 1024;   // bytes of one step's products in LDS: [q][lane] 16 bytes
constexpr unsigned BOX_SPIN_LIMIT = 1u << 24;
constexpr int BOX_MAX_STEPS = box::MAX_STEPS;  // the block's step table sits in LDS (a table in global memory would be read with vector loads: a full drain per step)

struct BoxParams {
  int nblocks;
  const box::Block *blocks;
  const box::StepTab *steps;
  const double *stream;
  const unsigned long long *einfo;
  const double *E;
  double *xs;
  unsigned long long *prog;
  unsigned *queue;    // tickets: word (2 g + sweep) * 32
  XcdState *st;
  unsigned *err;
  const double *rhs;  // lower sweep: d; upper sweep: x (the forward result, read and overwritten in place)
  double *out;
  const double *scale, *add; // upper sweep: x = x * scale + add (either may be null)
  int spread;
  // diagnostic (DDM_BOX_CHECK=1): per plane of block 0 and sweep four words {start, end (100 MHz clock), polls of the previous
  // plane's progress word, XCC}: dbg[(sweep * 128 + plane) * 4 ..]
  unsigned long long *dbg;
  int64_t n, stream_len, xs_len, prog_len, einfo_len, e_len;
};

typedef double bx_d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void bx_ld16(bx_d2 &v, const double *p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ void bx_ld8(double &v, const double *p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ void bx_ld8_sc1(double &v, const double *p) { asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ void bx_ld8u_sc1(unsigned long long &v, const unsigned long long *p) { asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ void bx_ld8u(unsigned long long &v, const unsigned long long *p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ void bx_st8(double *p, double v) { asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void bx_st8_sc1(double *p, double v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void bx_st8u(unsigned long long *p, unsigned long long v) { asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void bx_st8u_sc1(unsigned long long *p, unsigned long long v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
// the same with a wave-uniform base (SGPR pair) and a 32-bit byte offset per lane: one address register instead of two and no 64-bit
// address arithmetic per load -- the sweep kernels hold three sets of requests in registers.  The destination is a READ-WRITE operand
// ("+v"): the request registers are loop-carried, and with a plain output the compiler is free to define a new register at every
// request and copy it into the loop's register at the back edge -- while the load is in flight (tools/check_box_isa.py found that)
template <class T>
__device__ __forceinline__ const T *bx_uni(const T *p)
{
  const unsigned long long u = (unsigned long long)(uintptr_t)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
  return (const T *)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void bx_ld16o(bx_d2 &v, const void *base, unsigned off) { asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(v) : "v"(off), "s"(base) : "memory"); }
__device__ __forceinline__ void bx_ld8o(double &v, const void *base, unsigned off) { asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(v) : "v"(off), "s"(base) : "memory"); }
__device__ __forceinline__ void bx_ld8o_sc1(double &v, const void *base, unsigned off) { asm volatile("global_load_dwordx2 %0, %1, %2 sc1" : "+v"(v) : "v"(off), "s"(base) : "memory"); }
__device__ __forceinline__ void bx_ld8uo(unsigned long long &v, const void *base, unsigned off) { asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(v) : "v"(off), "s"(base) : "memory"); }
__device__ __forceinline__ void bx_ld8uo_sc1(unsigned long long &v, const void *base, unsigned off) { asm volatile("global_load_dwordx2 %0, %1, %2 sc1" : "+v"(v) : "v"(off), "s"(base) : "memory"); }
__device__ __forceinline__ void bx_st8o(void *base, unsigned off, double v) { asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(off), "v"(v), "s"(base) : "memory"); }
__device__ __forceinline__ void bx_st8o_sc1(void *base, unsigned off, double v) { asm volatile("global_store_dwordx2 %0, %1, %2 sc1" ::"v"(off), "v"(v), "s"(base) : "memory"); }

// brings the cache line of p into the L2 without a destination register (LDS-DMA into a scrap area of the workgroup)
// (inline asm, not the builtin: the compiler knows that the builtin writes LDS and puts s_waitcnt vmcnt(0) in front of the wave's next
// LDS read -- the prefetch wave reads the compute wave's step counter from LDS every step and would drain its touches each time)
__device__ __forceinline__ void bx_touch16(const void *p, unsigned char *scrap)
{
  const unsigned a = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)scrap);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(p), "s"(a) : "memory");
}
__device__ __forceinline__ void bx_touch4(const void *p, unsigned char *scrap)
{
  const unsigned a = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)scrap);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(p), "s"(a) : "memory");
}

// LDS-DMA the compiler does not know about (16 bytes per lane to lds_byte_address + 16 lane): with the builtin the compiler drains the
// whole vector-memory queue (s_waitcnt vmcnt(0)) in front of the first LDS read that might alias the destination -- once per step
__device__ __forceinline__ void bx_dma16(const void *base, unsigned off, unsigned lds_byte_address)
{
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds_byte_address) : "memory");   // (m0 is written here; the compiler sets it itself in front of each of its own uses)
}

// results not written yet carry this NaN pattern (k_box_fill before every sweep)
constexpr unsigned long long BOX_UNWRITTEN = 0xFFF7BADC0FFEE000ull;
__device__ __forceinline__ bool bx_unwritten(double v) { return (unsigned long long)__double_as_longlong(v) == BOX_UNWRITTEN; }
__global__ void k_box_fill(int64_t n, unsigned long long *__restrict__ p)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = BOX_UNWRITTEN;
}

// what one step requests BOX_DIST steps ahead
template <bool UPPER>
struct BoxSet {
  bx_d2 t[7];
  double r0, r1, r2, rh;
  double sc, ad;       // backward sweep: the row's scale / add entries of the level's tail
  unsigned ecnt;       // backward sweep: shell entries of the row (not a load: set when the request is issued)
};   // (the row's shell products of the backward sweep travel by LDS-DMA into a ring of BOX_DIST + 1 slots: 40 registers per set less)
// the compiler must not move uses of the registers in front of the wait that makes them valid
template <bool UPPER>
__device__ __forceinline__ void bx_tie(BoxSet<UPPER> &S)
{
  asm volatile("" : "+v"(S.t[0]), "+v"(S.t[1]), "+v"(S.t[2]), "+v"(S.t[3]), "+v"(S.t[4]), "+v"(S.t[5]), "+v"(S.t[6]));
  asm volatile("" : "+v"(S.r0), "+v"(S.r1), "+v"(S.r2), "+v"(S.rh));
  if constexpr (UPPER) {
    asm volatile("" : "+v"(S.sc), "+v"(S.ad));
  }
}

__global__ void k_box_products(int64_t n, const double *__restrict__ val, const int32_t *__restrict__ col, const double *__restrict__ xs, double *__restrict__ E)
{
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
    const int32_t c = col[p];
    E[p] = c < 0 ? 0.0 : val[p] * xs[c];
  }
}
// right-hand side of the shell system: d_s - L_sb y_b, products in ascending column order
__global__ void k_box_shell_rhs(int64_t ns, const int64_t *__restrict__ rp, const int32_t *__restrict__ ci, const double *__restrict__ va,
                                const int32_t *__restrict__ srow, const double *__restrict__ d, const double *__restrict__ y, double *__restrict__ ds)
{
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < ns; t += (int64_t)gridDim.x * blockDim.x) {
    double s = d[srow[t]];
    for (int64_t p = rp[t]; p < rp[t + 1]; ++p) s -= va[p] * y[ci[p]];
    ds[t] = s;
  }
}
// the shell rows of the result (with the tail of the Schwarz level, as k_pipe_permute_out applies it)
__global__ void k_box_shell_out(int64_t ns, const int32_t *__restrict__ srow, const double *__restrict__ xs, double *__restrict__ x, const double *__restrict__ scale,
                                const double *__restrict__ add)
{
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < ns; t += (int64_t)gridDim.x * blockDim.x) {
    const int32_t r = srow[t];
    double v = xs[t];
    if (scale) v *= scale[r];
    if (add) v += add[r];
    x[r] = v;
  }
}

template <bool UPPER>
__global__ __launch_bounds__(BOX_WG) void k_box_sweep(BoxParams P)
{
  __shared__ double ring[2][64];
  __shared__ unsigned sh_xcc, sh_xt, sh_gt, sh_fail, sh_q;
  __shared__ int sh_step;
  __shared__ box::StepTab sh_tab[BOX_MAX_STEPS + 1];
  // the prefetch wave's loads are LDS-DMA into this scrap area: a load into registers nobody reads would land, when it returns, in
  // registers the compiler has given to something else by then (an address of a later load: found the hard way)
  __shared__ __attribute__((aligned(16))) unsigned char sh_scrap[1024];
  __shared__ __attribute__((aligned(16))) unsigned char sh_E[UPPER ? (BOX_DIST + 1) * BOX_ESLOT : 16];
  __shared__ __attribute__((aligned(16))) unsigned char sh_I[UPPER ? (BOX_DIST + 1) * 1024 : 16];   // product descriptors of the steps in flight ([lane] 16 bytes, the first 8 used)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  XcdState *st = P.st;
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
  const unsigned xcc = sh_xcc, xt = sh_xt;
  const unsigned tk = lane < 8 ? __hip_atomic_load(&st->tickets[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
  const bool local_ok = !P.spread && __all(lane >= min(P.nblocks, 8) || tk >= 1u);
  const bool wt = !local_ok;
  const int gfirst = local_ok ? (int)xcc : (int)(sh_gt % (unsigned)P.nblocks);
  const int gcount = local_ok ? ((int)xcc < P.nblocks ? (P.nblocks - (int)xcc + 7) / 8 : 0) : P.nblocks;
  constexpr int sweep = UPPER ? 1 : 0;

  for (int gi = 0; gi < gcount; ++gi) {
    const int g = local_ok ? gfirst + 8 * (int)((gi + xt) % (unsigned)gcount) : (gfirst + gi) % P.nblocks;
    const box::Block *B = P.blocks + g;
    const int nx = B->nx, ny = B->ny, nz = B->nz, nsteps = B->nsteps;
    const int64_t r0 = B->r0;
    __syncthreads();
    for (int k = threadIdx.x; k <= nsteps; k += BOX_WG) sh_tab[k] = P.steps[B->step_off + k];
    const box::StepTab *T = sh_tab;
    for (;;) {
      if (threadIdx.x == 0) {
        sh_q = __hip_atomic_fetch_add(P.queue + (size_t)(2 * g + sweep) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh_step = -2;
      }
      __syncthreads();
      const int K = __builtin_amdgcn_readfirstlane((int)sh_q);
      if (K >= nz) {
        __syncthreads();
        break;
      }
      const double *plane = bx_uni(P.stream + B->stream_off[sweep] + (int64_t)K * B->plane_len[sweep]);
      double *xsK = const_cast<double *>(bx_uni((const double *)(P.xs + B->xs_off + (int64_t)K * (nsteps + 1) * 64)));    // [nsteps + 1][64]: the last slot takes the stores of steps without rows
      const double *xsP = K > 0 ? xsK - (int64_t)(nsteps + 1) * 64 : xsK;
      const unsigned long long *einfoK = bx_uni(P.einfo + B->einfo_off + (int64_t)K * nsteps * 64);
      const int ktrue = UPPER ? nz - 1 - K : K;
      const double *rhsU = bx_uni(P.rhs), *scaleU = bx_uni(P.scale ? P.scale : P.rhs), *addU = bx_uni(P.add ? P.add : P.rhs), *EU = bx_uni(P.E);
      const unsigned sh_E_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)sh_E);
      const unsigned sh_I_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)sh_I);
      // (J, I) of this lane at step s; I in [-3, 124] while the lane is on a line, -1000 otherwise
      auto line_of = [&](int s, int &J, int &I) __attribute__((always_inline)) {
        const int t = s + 3 - 2 * lane;
        const int c = t >> 7;
        J = lane + 64 * c;
        I = (t >= 0 && J < ny) ? (t & 127) - 3 : -1000;
      };
      auto row_of = [&](int I, int J) __attribute__((always_inline)) -> int64_t {
        const int i = UPPER ? nx - 1 - I : I, j = UPPER ? ny - 1 - J : J;
        return r0 + i + (int64_t)nx * (j + (int64_t)ny * ktrue);
      };

      if (wave == 1) {
        // ---------------- prefetch wave: brings what step s needs into the L2 ----------------
        for (int s = 0; s < nsteps; ++s) {
          unsigned spins = 0;
          while (__hip_atomic_load(&sh_step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < s - BOX_AHEAD) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > BOX_SPIN_LIMIT) break;
          }
          const box::StepTab ts = T[s];
          const double *tile = plane + (int64_t)ts.off * box::NV;
          const int tile_d = ts.nact * box::NV;       // doubles
          for (int o = 0; o < tile_d; o += 128) {     // 1 KiB per wavefront load
            const int e = min(o + 2 * lane, tile_d - 2);
            bx_touch16(tile + e, sh_scrap);
          }
          int J, I;
          line_of(s, J, I);
          const bool act = I >= 0 && I < nx;
          bx_touch4(P.rhs + (act ? row_of(I, J) : r0), sh_scrap);
          if (UPPER) bx_touch4(einfoK + (int64_t)s * 64 + lane, sh_scrap);
          asm volatile("s_waitcnt vmcnt(40)" ::: "memory");   // (four to five steps of touches in flight: two would tie this wave to 2 steps per HBM round trip)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        // ---------------- compute wave ----------------
        BoxSet<UPPER> SA, SB, SC;
        const unsigned long long t_begin = P.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;   // (diagnostic stamps, DDM_BOX_CHECK=1)
        unsigned polls = 0;
        double wA[3] = {0, 0, 0}, wB[3] = {0, 0, 0}, wC[3] = {0, 0, 0}, uu[3] = {0, 0, 0}, xprev = 0;   // windows: rotated by the unroll, not by moves
        unsigned failed = 0;
        // requests of step s into set S (the same number of vector-memory operations whatever s is)
        auto request = [&](BoxSet<UPPER> &S, int s, unsigned long long ei_now) __attribute__((always_inline)) {
          int J, I;
          line_of(s, J, I);
          const bool act = I >= 0 && I < nx;
          const int sc = min(max(s, 0), nsteps - 1);
          const box::StepTab ts = T[sc];
          const int a = act ? J - ts.jlo : 0;
          const unsigned toff = (unsigned)(ts.off * box::NV + 2 * a) * 8u, qs = (unsigned)ts.nact * 16u;     // bytes inside the plane's stream
#pragma unroll
          for (int q = 0; q < 7; ++q) bx_ld16o(S.t[q], plane, toff + q * qs);
          // previous plane: column I + 1 of the lines J - 1, J, J + 1 = its steps s - 1, s + 1, s + 3, lanes lane - 1, lane, lane + 1
          const int l0 = min(max(s - 1, 0), nsteps - 1), l1 = min(max(s + 1, 0), nsteps - 1), l2 = min(s + 3, nsteps - 1);
          bx_ld8o_sc1(S.r0, xsP, (unsigned)(l0 * 64 + ((lane + 63) & 63)) * 8u);
          bx_ld8o_sc1(S.r1, xsP, (unsigned)(l1 * 64 + lane) * 8u);
          bx_ld8o_sc1(S.r2, xsP, (unsigned)(l2 * 64 + ((lane + 1) & 63)) * 8u);
          const unsigned roff = (unsigned)((act ? row_of(I, J) : r0) * 8);                                    // rank-local rows: < 2^31
          bx_ld8o(S.rh, rhsU, roff);
          if constexpr (UPPER) {
            const int s2 = min(s + BOX_DIST, nsteps - 1);
            const unsigned ptr = (unsigned)ei_now, cnt = (unsigned)(ei_now >> 32);
            S.ecnt = cnt;
            bx_ld8o(S.sc, scaleU, roff);
            bx_ld8o(S.ad, addU, roff);
            const unsigned slot = sh_E_addr + (unsigned)((s + 4) & 3) * BOX_ESLOT;
#pragma unroll
            for (int q = 0; q < BOX_NEL; ++q) bx_dma16(EU, ((unsigned)(2 * q) < cnt ? ptr + 2 * q : 0u) * 8u, slot + q * 1024);
            // the descriptor of step s + BOX_DIST: by LDS-DMA as well (in a register it is loop-carried through an inline-asm load, and the
            // compiler kept finding ways to copy it while the load was in flight); 16 bytes per lane land, the first 8 are the lane's
            bx_dma16(einfoK, (unsigned)(s2 * 64 + lane) * 8u, sh_I_addr + (unsigned)((s + 4) & 3) * 1024);
          }
        };
        auto step = [&](BoxSet<UPPER> &S, int l, auto RC) __attribute__((always_inline)) {
          constexpr int R = decltype(RC)::value, i0 = (R + 1) % 3, i1 = (R + 2) % 3, i2 = R;   // columns I - 1, I, I + 1 of the windows in this turn
          // (the caller has waited for the requests of step l)
          int J, I;
          line_of(l, J, I);
          const bool act = I >= 0 && I < nx;
          const bool colok = I + 1 >= 0 && I + 1 < nx;          // column I + 1 exists (I = -1000 off line: false)
          const bool prev = K > 0 && colok;
          // hand-over by DATA: the previous plane's results were pre-set to a NaN pattern (k_box_fill) -- a value that still shows the
          // pattern has not been written yet.  In the steady state this never happens (every plane runs the same code at the same
          // speed; a plane that got too close waits here once and is far enough behind from then on): no progress words, no polls
          const bool n0 = prev && J >= 1, n1 = prev, n2 = prev && J + 1 < ny;
          {
            bool bad = (n0 && bx_unwritten(S.r0)) || (n1 && bx_unwritten(S.r1)) || (n2 && bx_unwritten(S.r2));
            unsigned spins = 0;
            while (__any(bad)) {
              ++polls;
              const int l0 = min(max(l - 1, 0), nsteps - 1), l1 = min(max(l + 1, 0), nsteps - 1), l2 = min(l + 3, nsteps - 1);
              bx_ld8o_sc1(S.r0, xsP, (unsigned)(l0 * 64 + ((lane + 63) & 63)) * 8u);
              bx_ld8o_sc1(S.r1, xsP, (unsigned)(l1 * 64 + lane) * 8u);
              bx_ld8o_sc1(S.r2, xsP, (unsigned)(l2 * 64 + ((lane + 1) & 63)) * 8u);
              asm volatile("s_waitcnt vmcnt(0)" : "+v"(S.r0), "+v"(S.r1), "+v"(S.r2)::"memory");
              bad = (n0 && bx_unwritten(S.r0)) || (n1 && bx_unwritten(S.r1)) || (n2 && bx_unwritten(S.r2));
              if (++spins > BOX_SPIN_LIMIT) {
                failed = 1;
                break;
              }
              if (__any(bad)) __builtin_amdgcn_s_sleep(1);
            }
          }
          wA[i2] = n0 ? S.r0 : 0.0;      // (overwrites the column that dropped out of the window)
          wB[i2] = n1 ? S.r1 : 0.0;
          wC[i2] = n2 ? S.r2 : 0.0;
          const double nbv = ring[(l + 1) & 1][(lane + 63) & 63];
          uu[i2] = (colok && J >= 1) ? nbv : 0.0;
          const double wA0 = wA[i0], wA1 = wA[i1], wA2 = wA[i2], wB0 = wB[i0], wB1 = wB[i1], wB2 = wB[i2], wC0 = wC[i0], wC1 = wC[i1], wC2 = wC[i2];
          const double u0 = uu[i0], u1 = uu[i1], u2 = uu[i2];
          if (I <= 0) xprev = 0.0;
          double v = S.rh;
          if constexpr (!UPPER) {
            v -= S.t[0].x * wA0; v -= S.t[0].y * wA1; v -= S.t[1].x * wA2;
            v -= S.t[1].y * wB0; v -= S.t[2].x * wB1; v -= S.t[2].y * wB2;
            v -= S.t[3].x * wC0; v -= S.t[3].y * wC1; v -= S.t[4].x * wC2;
            v -= S.t[4].y * u0; v -= S.t[5].x * u1; v -= S.t[5].y * u2;
            v -= S.t[6].x * xprev;
          } else {
            v -= S.t[0].x * xprev;
            v -= S.t[0].y * u2; v -= S.t[1].x * u1; v -= S.t[1].y * u0;
            v -= S.t[2].x * wC2; v -= S.t[2].y * wC1; v -= S.t[3].x * wC0;
            v -= S.t[3].y * wB2; v -= S.t[4].x * wB1; v -= S.t[4].y * wB0;
            v -= S.t[5].x * wA2; v -= S.t[5].y * wA1; v -= S.t[6].x * wA0;
            const unsigned char *slot = sh_E + ((l + 4) & 3) * BOX_ESLOT + lane * 16;
            const unsigned mycnt = S.ecnt;
#pragma unroll
            for (int q = 0; q < BOX_NEL; ++q) {
              if (!__any(mycnt > (unsigned)(2 * q))) break;     // (no row of this step has that many shell entries; the skipped slots hold zeros)
              const bx_d2 ep = *reinterpret_cast<const bx_d2 *>(slot + q * 1024);
              v -= ep.x;
              v -= ep.y;
            }
            v *= S.t[6].y;
          }
          const double xnew = act ? v : 0.0;
          xprev = xnew;
          ring[l & 1][lane] = xnew;
          // results: position-ordered for the next plane, natural order for the caller (inactive lanes store to a scratch slot)
          const unsigned xoff = (unsigned)((l < 0 || l >= nsteps ? nsteps : l) * 64 + lane) * 8u;   // (steps without rows: the spare slot)
          double *xp = xsK + xoff / 8;
          if (wt) bx_st8o_sc1(xsK, xoff, xnew);
          else bx_st8o(xsK, xoff, xnew);
          const int64_t row = act ? row_of(I, J) : -1;
          double o = xnew;
          if constexpr (UPPER) {
            if (P.scale) o *= S.sc;
            if (P.add) o += S.ad;
          }
          bx_st8(act ? P.out + row : xp, o);
        };

        // head: the product descriptors of steps 0 and 1, then the requests of steps -1, 0 and 1.  The walk starts at step -1: the
        // run-in of line 0 (column 0 of the previous plane's lines 0 and 1)
        unsigned long long e0 = 0, e1 = 0;
        if constexpr (UPPER) {
          bx_ld8u(e0, einfoK + lane);
          bx_ld8u(e1, einfoK + (int64_t)min(1, nsteps - 1) * 64 + lane);
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(e0), "+v"(e1)::"memory");
        }
        request(SA, -1, 0ull);
        request(SB, 0, e0);
        request(SC, 1, e1);
        // per step: NB requests; stores: 2 results + 1 progress word (lane 0 only: still one operation of the wave)
        constexpr int NB = UPPER ? 14 + BOX_NEL : 11;   // requests per step; 2 result stores per step
        static_assert(BOX_NEL == 10 && BOX_DIST == 3, "the s_waitcnt immediates below are written for 10 product loads and three sets");
        // the loop is entered with all three sets complete: whatever register moves the compiler places in front of the loop then move
        // data that has arrived (all planes start together, so this wait is not on the chain of hand-overs)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bx_tie(SA);
        bx_tie(SB);
        bx_tie(SC);
        // one step with set S: its requests are complete when at most the operations behind them are outstanding: two steps' stores
        // (3 each) and requests (NB each)
        auto turn = [&](BoxSet<UPPER> &S, int l, auto RC) __attribute__((always_inline)) {
          if constexpr (NB == 11) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(52)" ::: "memory");
          bx_tie(S);
          const unsigned long long ei = UPPER ? *reinterpret_cast<const unsigned long long *>(sh_I + ((l + 4) & 3) * 1024 + lane * 16) : 0ull;
          step(S, l, RC);
          __hip_atomic_store(&sh_step, l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          request(S, l + BOX_DIST, ei);
        };
        // (whole trips only: the one or two turns behind the last step have no active row, their stores go to the plane's spare slot
        //  -- no exit in the middle of the body, so that every path from a request to its wait is the one the counts assume)
        for (int l = -1; l < nsteps && !failed; l += 3) {
          turn(SA, l, std::integral_constant<int, 0>{});
          turn(SB, l + 1, std::integral_constant<int, 1>{});
          turn(SC, l + 2, std::integral_constant<int, 2>{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bx_tie(SA);
        bx_tie(SB);
        bx_tie(SC);
        if (P.dbg && g == 0 && K < 128 && lane == 0) {     // per plane of block 0: start, end (100 MHz), polls of the previous plane's word, XCC
          unsigned long long *o = P.dbg + (size_t)(sweep * 128 + K) * 4;
          o[0] = t_begin;
          o[1] = __builtin_amdgcn_s_memrealtime();
          o[2] = polls;
          o[3] = xcc;
        }
        __hip_atomic_store(&sh_step, 1 << 30, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (failed && lane == 0) __hip_atomic_store(P.err, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      __syncthreads(); // both waves have left the plane before the ticket words are written again
    }
  }
}

} // namespace ddm
