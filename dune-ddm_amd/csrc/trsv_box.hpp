// Device side of the "box" triangular-solve engine; data layout and rationale: trsv_box_host.hpp.
//
// k_box_sweep<UPPER>: one workgroup = one compute wavefront + one prefetch wavefront, working on one plane K of one block at a time
// (planes are handed out in order by a ticket per block and sweep, so the plane a wave waits for is always held by a running wave).
//   compute wave, step l (rows I = l - 2 J of the lines J = lane + 64 c):
//     * operands of the own plane: x(I-1, J) is the lane's previous result (register); the line J-1 is the neighbour lane's line -- its
//       newest value comes out of a two-slot LDS ring, the two older ones were read in the steps before (register window);
//     * operands of plane K-1: three register windows (lines J-1, J, J+1), one new column per step, read from the plane's
//       position-ordered results xs[K-1][step][lane] with sc1 loads TWO steps ahead (guarded by the plane's progress word);
//     * the row's 14 stream values (7 sixteen-byte loads), its right-hand side entry, (backward) its products with the shell and the
//       product slice of the step after next: all requested two steps ahead, every step issues the same number of vector-memory
//       operations, so one s_waitcnt vmcnt(N) with a constant N is the hand-over between the steps (vmcnt retires in issue order);
//     * row sum in ascending column order, every product rounded before it is subtracted (-ffp-contract=off): the sequential solve.
//   prefetch wave: touches the stream tiles, right-hand side lines and product descriptors BOX_AHEAD steps ahead of the compute wave, so
//     that the compute wave's two-step request distance only has to cover an L2 hit.
// Visibility as in the pipe engine: a block's planes run on ONE XCD (workgroups read their XCC id and serve the blocks g = xcc mod 8);
// results are plain stores (they stay in that XCD's L2), reads of another wave's results are sc1 loads; without that placement
// (fewer XCDs with workgroups than blocks need, or P.spread) results and progress words are agent-scope stores.
#pragma once
#include "trsv_box_host.hpp"

namespace ddm {

constexpr int BOX_WG = 128;
constexpr int BOX_NEL = box::MAX_EXT / 2;   // two-double loads of products per row
constexpr int BOX_AHEAD = 6;                // steps the prefetch wave runs ahead of the compute wave
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
  // diagnostic (DDM_BOX_CHECK=1): every address of the sweep kernels is checked against its array; the first violation is recorded
  // in dbg[0..7] = {site, offset, length, lane, step, plane, block, 0} and the access is redirected to the array's first element
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

// (compiled in with -DDDM_BOX_CHECK_BUILD only: the extra registers of the check make the compiler copy registers of loads in flight --
// see the Makefile target check-box-isa -- so the checked build is for hunting a wild address, not for results)
template <class T>
__device__ __forceinline__ const T *bx_chk(const BoxParams &P, int site, const T *p, const T *base, int64_t len, int lane, int step, int plane, int block)
{
#ifndef DDM_BOX_CHECK_BUILD
  return p;
#endif
  if (!P.dbg) return p;
  const int64_t off = p - base;
  if (off >= 0 && off < len) return p;
  if (atomicCAS(P.dbg + 7, 0ull, 1ull) == 0ull) {
    P.dbg[0] = (unsigned long long)site;
    P.dbg[1] = (unsigned long long)off;
    P.dbg[2] = (unsigned long long)len;
    P.dbg[3] = (unsigned long long)lane;
    P.dbg[4] = (unsigned long long)(long long)step;
    P.dbg[5] = (unsigned long long)plane;
    P.dbg[6] = (unsigned long long)block;
  }
  return base;
}
template <class T>
__device__ __forceinline__ T *bx_chkw(const BoxParams &P, int site, T *p, T *base, int64_t len, int lane, int step, int plane, int block)
{
  return const_cast<T *>(bx_chk<T>(P, site, p, base, len, lane, step, plane, block));
}

// brings the cache line of p into the L2 without a destination register (LDS-DMA into a scrap area of the workgroup)
__device__ __forceinline__ void bx_touch16(const void *p, unsigned char *scrap)
{
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p, (__attribute__((address_space(3))) void *)scrap, 16, 0, 0);
}
__device__ __forceinline__ void bx_touch4(const void *p, unsigned char *scrap)
{
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p, (__attribute__((address_space(3))) void *)scrap, 4, 0, 0);
}

// what one step requests two steps ahead
template <bool UPPER>
struct BoxSet {
  bx_d2 t[7];
  double r0, r1, r2, rh;
  unsigned long long pl, ei;
  double sc, ad;       // backward sweep: the row's scale / add entries of the level's tail
  bx_d2 ep[UPPER ? BOX_NEL : 1];
};
// the compiler must not move uses of the registers in front of the wait that makes them valid
template <bool UPPER>
__device__ __forceinline__ void bx_tie(BoxSet<UPPER> &S)
{
  asm volatile("" : "+v"(S.t[0]), "+v"(S.t[1]), "+v"(S.t[2]), "+v"(S.t[3]), "+v"(S.t[4]), "+v"(S.t[5]), "+v"(S.t[6]));
  asm volatile("" : "+v"(S.r0), "+v"(S.r1), "+v"(S.r2), "+v"(S.rh), "+v"(S.pl), "+v"(S.ei));
  if constexpr (UPPER) {
    asm volatile("" : "+v"(S.sc), "+v"(S.ad));
    asm volatile("" : "+v"(S.ep[0]), "+v"(S.ep[1]), "+v"(S.ep[2]), "+v"(S.ep[3]), "+v"(S.ep[4]));
    asm volatile("" : "+v"(S.ep[5]), "+v"(S.ep[6]), "+v"(S.ep[7]), "+v"(S.ep[8]), "+v"(S.ep[9]));
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
  const int dmode = P.spread >> 4;     // diagnostic (DDM_BOX_SPREAD = 16 * mode): 1 leave here, 2 tickets and tables only, 3 no compute wave, 4 no prefetch wave
  if (dmode == 1) return;
  const unsigned xcc = sh_xcc, xt = sh_xt;
  const unsigned epoch = __hip_atomic_load(&st->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned tk = lane < 8 ? __hip_atomic_load(&st->tickets[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
  const bool local_ok = !(P.spread & 15) && __all(lane >= min(P.nblocks, 8) || tk >= 1u);
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
      const double *plane = P.stream + B->stream_off[sweep] + (int64_t)K * B->plane_len[sweep];
      double *xsK = P.xs + B->xs_off + (int64_t)K * nsteps * 64;
      const double *xsP = K > 0 ? xsK - (int64_t)nsteps * 64 : xsK;
      unsigned long long *progK = P.prog + B->prog_off + (int64_t)sweep * nz + K;
      const unsigned long long *progP = K > 0 ? progK - 1 : progK;
      const unsigned long long *einfoK = P.einfo + B->einfo_off + (int64_t)K * nsteps * 64;
      const int ktrue = UPPER ? nz - 1 - K : K;
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

      if (dmode == 2 || (dmode == 3 && wave == 0) || (dmode == 4 && wave == 1)) {
        if (wave == 0 && lane == 0) {     // (the plane counts as done, so that nobody waits for it)
          const unsigned long long w = ((unsigned long long)epoch << 32) | (unsigned)nsteps;
          __hip_atomic_store(progK, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&sh_step, 1 << 30, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      } else if (wave == 1) {
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
            bx_touch16(bx_chk(P, 1, tile + e, P.stream, P.stream_len - 1, lane, s, K, g), sh_scrap);
          }
          int J, I;
          line_of(s, J, I);
          const bool act = I >= 0 && I < nx;
          bx_touch4(bx_chk(P, 2, P.rhs + (act ? row_of(I, J) : r0), P.rhs, P.n, lane, s, K, g), sh_scrap);
          if (UPPER) bx_touch4(bx_chk(P, 3, einfoK + (int64_t)s * 64 + lane, P.einfo, P.einfo_len, lane, s, K, g), sh_scrap);
          asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        // ---------------- compute wave ----------------
        BoxSet<UPPER> SA, SB;
        double wA0 = 0, wA1 = 0, wA2 = 0, wB0 = 0, wB1 = 0, wB2 = 0, wC0 = 0, wC1 = 0, wC2 = 0, u0 = 0, u1 = 0, u2 = 0, xprev = 0;
        unsigned failed = 0;
        auto count_of = [&](unsigned long long w) __attribute__((always_inline)) -> int { return (unsigned)(w >> 32) == epoch ? (int)(unsigned)w : 0; };
        auto publish = [&](int steps) __attribute__((always_inline)) {
          if (lane == 0) {
            const unsigned long long w = ((unsigned long long)epoch << 32) | (unsigned)steps;
            unsigned long long *pk = bx_chkw(P, 4, progK, P.prog, P.prog_len, lane, steps, K, g);
            if (wt) bx_st8u_sc1(pk, w);
            else bx_st8u(pk, w);
          }
        };
        // requests of step s into set S (the same number of vector-memory operations whatever s is)
        auto request = [&](BoxSet<UPPER> &S, int s, unsigned long long ei_now) __attribute__((always_inline)) {
          int J, I;
          line_of(s, J, I);
          const bool act = I >= 0 && I < nx;
          const int sc = min(max(s, 0), nsteps - 1);
          const box::StepTab ts = T[sc];
          const int a = act ? J - ts.jlo : 0;
          const double *tile = plane + (int64_t)ts.off * box::NV + 2 * a;
          const int qs = 2 * ts.nact;
#pragma unroll
          for (int q = 0; q < 7; ++q) bx_ld16(S.t[q], bx_chk(P, 5, tile + q * qs, P.stream, P.stream_len - 1, lane, s, K, g));
          // previous plane: column I + 1 of the lines J - 1, J, J + 1 = its steps s - 1, s + 1, s + 3, lanes lane - 1, lane, lane + 1
          const int l0 = min(max(s - 1, 0), nsteps - 1), l1 = min(s + 1, nsteps - 1), l2 = min(s + 3, nsteps - 1);
          bx_ld8_sc1(S.r0, bx_chk(P, 6, xsP + (int64_t)l0 * 64 + ((lane + 63) & 63), (const double *)P.xs, P.xs_len, lane, s, K, g));
          bx_ld8_sc1(S.r1, bx_chk(P, 7, xsP + (int64_t)l1 * 64 + lane, (const double *)P.xs, P.xs_len, lane, s, K, g));
          bx_ld8_sc1(S.r2, bx_chk(P, 8, xsP + (int64_t)l2 * 64 + ((lane + 1) & 63), (const double *)P.xs, P.xs_len, lane, s, K, g));
          bx_ld8(S.rh, bx_chk(P, 9, P.rhs + (act ? row_of(I, J) : r0), P.rhs, P.n, lane, s, K, g));
          bx_ld8u_sc1(S.pl, bx_chk(P, 10, progP, (const unsigned long long *)P.prog, P.prog_len, lane, s, K, g));
          if constexpr (UPPER) {
            const int s2 = min(s + 2, nsteps - 1);
            const unsigned ptr = (unsigned)ei_now, cnt = (unsigned)(ei_now >> 32);
            bx_ld8u(S.ei, bx_chk(P, 11, einfoK + (int64_t)s2 * 64 + lane, P.einfo, P.einfo_len, lane, s, K, g));
            const int64_t rr = act ? row_of(I, J) : r0;
            bx_ld8(S.sc, bx_chk(P, 12, (P.scale ? P.scale : P.rhs) + rr, P.scale ? P.scale : P.rhs, P.n, lane, s, K, g));
            bx_ld8(S.ad, bx_chk(P, 13, (P.add ? P.add : P.rhs) + rr, P.add ? P.add : P.rhs, P.n, lane, s, K, g));
#pragma unroll
            for (int q = 0; q < BOX_NEL; ++q) bx_ld16(S.ep[q], bx_chk(P, 14, P.E + ((unsigned)(2 * q) < cnt ? ptr + 2 * q : 0u), P.E, P.e_len - 1, lane, s, K, g));
          }
        };
        // waits until the previous plane has published `need` steps (bounded)
        auto wait_prev = [&](int need, unsigned long long seen) __attribute__((always_inline)) {
          if (K == 0) return;
          need = min(need, nsteps);
          int have = __builtin_amdgcn_readfirstlane(count_of(seen));
          unsigned spins = 0;
          while (have < need) {
            unsigned long long w;
            bx_ld8u_sc1(w, bx_chk(P, 15, progP, (const unsigned long long *)P.prog, P.prog_len, lane, need, K, g));
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(w)::"memory");
            have = __builtin_amdgcn_readfirstlane(count_of(w));
            if (++spins > BOX_SPIN_LIMIT) {
              failed = 1;
              break;
            }
            if (have < need) __builtin_amdgcn_s_sleep(1);
          }
        };
        auto step = [&](BoxSet<UPPER> &S, int l) __attribute__((always_inline)) {
          // (the caller has waited for the requests of step l)
          int J, I;
          line_of(l, J, I);
          const bool act = I >= 0 && I < nx;
          const bool colok = I + 1 >= 0 && I + 1 < nx;          // column I + 1 exists (I = -1000 off line: false)
          const bool prev = K > 0 && colok;
          wA0 = wA1; wA1 = wA2; wA2 = (prev && J >= 1) ? S.r0 : 0.0;
          wB0 = wB1; wB1 = wB2; wB2 = prev ? S.r1 : 0.0;
          wC0 = wC1; wC1 = wC2; wC2 = (prev && J + 1 < ny) ? S.r2 : 0.0;
          const double nbv = ring[(l + 1) & 1][(lane + 63) & 63];
          u0 = u1; u1 = u2; u2 = (colok && J >= 1) ? nbv : 0.0;
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
#pragma unroll
            for (int q = 0; q < BOX_NEL; ++q) {
              v -= S.ep[q].x;
              v -= S.ep[q].y;
            }
            v *= S.t[6].y;
          }
          const double xnew = act ? v : 0.0;
          xprev = xnew;
          ring[l & 1][lane] = xnew;
          // results: position-ordered for the next plane, natural order for the caller (inactive lanes store to a scratch slot)
          double *xp = bx_chkw(P, 16, xsK + (int64_t)max(l, 0) * 64 + lane, P.xs, P.xs_len, lane, l, K, g);   // (step -1 has no active row: its zeros are overwritten by step 0)
          if (wt) bx_st8_sc1(xp, xnew);
          else bx_st8(xp, xnew);
          const int64_t row = act ? row_of(I, J) : -1;
          double o = xnew;
          if constexpr (UPPER) {
            if (P.scale) o *= S.sc;
            if (P.add) o += S.ad;
          }
          bx_st8(act ? bx_chkw(P, 17, P.out + row, P.out, P.n, lane, l, K, g) : xp, o);
        };

        // head: the product descriptor of step 0, then the requests of steps -1 and 0
        unsigned long long e0 = 0;
        if constexpr (UPPER) {
          bx_ld8u(e0, bx_chk(P, 18, einfoK + lane, P.einfo, P.einfo_len, lane, -1, K, g));
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(e0)::"memory");
        }
        // the walk starts at step -1: the run-in of line 0 (column 0 of the previous plane's lines 0 and 1)
        wait_prev(3, 0ull);
        request(SA, -1, 0ull);
        wait_prev(4, 0ull);
        request(SB, 0, e0);
        // per step: NB requests; stores: 2 results + 1 progress word (lane 0 only: still one operation of the wave)
        constexpr int NB = UPPER ? 15 + BOX_NEL : 12;
        static_assert(BOX_NEL == 10, "the s_waitcnt immediates below are written for 10 product loads");
        // the requests of step 0 are complete when only those of step 1 are outstanding (no stores in between yet)
        if constexpr (NB == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(25)" ::: "memory");
        for (int l = -1; l < nsteps && !failed; l += 2) {
          // ---- set A ----
          if constexpr (NB + 3 == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
          bx_tie(SA);
          publish(max(l - 1, 0));
          {
            const unsigned long long seen = SA.pl, ei = SA.ei;
            step(SA, l);
            __hip_atomic_store(&sh_step, l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            wait_prev(l + 6, seen);
            request(SA, l + 2, ei);
          }
          if (l + 1 >= nsteps) break;
          // ---- set B ----
          if constexpr (NB + 3 == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
          bx_tie(SB);
          publish(max(l, 0));
          {
            const unsigned long long seen = SB.pl, ei = SB.ei;
            step(SB, l + 1);
            __hip_atomic_store(&sh_step, l + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            wait_prev(l + 7, seen);
            request(SB, l + 3, ei);
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bx_tie(SA);
        bx_tie(SB);
        publish(nsteps);
        __hip_atomic_store(&sh_step, 1 << 30, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (failed && lane == 0) __hip_atomic_store(P.err, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      __syncthreads(); // both waves have left the plane before the ticket words are written again
    }
  }
}

} // namespace ddm
