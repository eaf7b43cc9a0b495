// Single-vector solves with the device supernodal factor (sn_chol.hpp): the local solve of SchwarzPreconditioner::apply with a
// sparse direct subdomain solver (dune/ddm/schwarz.hh:85-92,133: `type = cholmod | umfpack`).  Included by sn_chol.hpp.
//
// One right-hand side has no work for the matrix cores: every kernel here is a gather / dot-product kernel on the panels,
// bounded by the HBM stream of the panels (8 bytes per entry and sweep) and, in practice, by the number of dependent steps: the
// supernodal elimination tree of a 3-D subdomain has ~40-130 levels, most of them the chains of 128-column links the large
// separators are cut into (ONE supernode per subdomain and level).  Two regimes:
//   * bottom levels (many small supernodes): one launch per level and sweep, all subdomains together (k_sn_fwd1, k_sn_bwd1_*);
//   * top levels (from the first level on which every later level has only a few supernodes): ONE persistent launch walks them up
//     (forward) and down again (backward).  A subdomain's supernodes are worked on by the workgroups of ONE XCD (block b -> XCD
//     b % 8, XCC id read at run time, as the pipe engine does): results are plain stores that stay in that XCD's L2, every read of
//     another workgroup's result is an sc1 load, the levels are separated by an all-to-all flag barrier among the XCD's workgroups
//     (plain flag stores, sc1 polls: ~1 us instead of ~15 us per launch).  With fewer than 8 subdomains all workgroups form one
//     group and the hand-overs are write-through (sc1 stores).
// Determinism, no atomics anywhere: a supernode of the BOTTOM levels writes what it subtracts from its row rows[q] into slot q of a
// scratch array, and the owner of the row subtracts its slots in list order (Meta::tptr / tmid / tidx) -- bottom-level owners in
// k_sn_fwd1 (short lists), top-level owners in ONE parallel gather phase at the start of the persistent kernel.  A supernode of the
// TOP levels subtracts in place, colour by colour (supernodes of one level whose row lists intersect have different colours,
// Factor::h_colour): every entry receives its updates in one order.  The same bits every run.
#pragma once

namespace sn {

constexpr int S1_CHUNK_ROWS = BWD_ROWS; // rows per partial product of the backward sweep

__device__ __forceinline__ double s1_wave_sum(double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
template <bool SC1>
__device__ __forceinline__ double s1_ld(const double *p)
{
  if (SC1) return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  return *p;
}
__device__ __forceinline__ void s1_st(double *p, double v, bool wt)
{
  if (wt) __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// ---- pieces (called by all threads of a workgroup of 4 * NCMAX threads; NCMAX = 64 or 128 >= columns of the supernode) ------------
// bs[0 .. nc) = right-hand side of the columns of s (L U: in pivot order) minus the slots of the supernodes below; four threads
// share a column's list (strided), partial sums added in a fixed order.  Ends with a barrier.
template <bool LU, int NCMAX, bool SC1>
__device__ __forceinline__ void s1_gather(const Meta &M, int32_t s, const double *__restrict__ B, const double *__restrict__ contrib, double *bs, double *part)
{
  const int32_t f = M.first[s], nc = M.first[s + 1] - f;
  const int tid = threadIdx.x, i = tid & (NCMAX - 1), g = tid / NCMAX;
  double acc = 0.0;
  int64_t row = 0;
  if (i < nc) {
    row = f + (LU ? M.piv[f + i] : i);
    const int64_t q0 = M.tptr[row], q1 = M.tptr[row + 1];
    int64_t q = q0 + g;
    for (; q + 12 < q1; q += 16) { // four loads in flight per thread
      const double c0 = s1_ld<SC1>(contrib + M.tidx[q]), c1 = s1_ld<SC1>(contrib + M.tidx[q + 4]), c2 = s1_ld<SC1>(contrib + M.tidx[q + 8]),
                   c3 = s1_ld<SC1>(contrib + M.tidx[q + 12]);
      acc = (((acc + c0) + c1) + c2) + c3;
    }
    for (; q < q1; q += 4) acc += s1_ld<SC1>(contrib + M.tidx[q]);
  }
  part[g * NCMAX + i] = acc;
  __syncthreads();
  if (tid < nc) bs[tid] = B[row] - ((part[tid] + part[NCMAX + tid]) + (part[2 * NCMAX + tid] + part[3 * NCMAX + tid]));
  __syncthreads();
}
// ys[0 .. nc) = W_s bs (Cholesky: W = L_ss^-1, lower triangle; L U: unit lower L^-1, strict part stored).  Ends with a barrier.
template <bool LU, int NCMAX>
__device__ __forceinline__ void s1_lower_product(const Meta &M, int32_t s, const double *bs, double *ys, double *part)
{
  constexpr int YS = NCMAX / 4; // k slice of the four thread groups
  const int32_t f = M.first[s], nc = M.first[s + 1] - f;
  const int64_t ld = nc + M.nrow[s];
  const double *P = M.panels + M.pptr[s];
  const int tid = threadIdx.x, i = tid & (NCMAX - 1), sl = tid / NCMAX; // row i of W, k slice [YS sl, YS sl + YS)
  double acc = 0.0;
  if (i < nc) {
    const int k1 = min(YS * sl + YS, LU ? i : i + 1);
    for (int kb = YS * sl; kb < k1; kb += 8) {
      double w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = kb + u < k1 ? P[i + (int64_t)(kb + u) * ld] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += w[u] * bs[min(kb + u, nc - 1)];
    }
  }
  part[sl * NCMAX + i] = acc;
  __syncthreads();
  if (tid < nc) ys[tid] = (LU ? bs[tid] : 0.0) + ((part[tid] + part[NCMAX + tid]) + (part[2 * NCMAX + tid] + part[3 * NCMAX + tid]));
  __syncthreads();
}
// the rows [64 tile, 64 tile + 64) of s: IN_PLACE = false: their slots <- R_s ys (bottom levels); IN_PLACE = true: dst[row] -= R_s ys
// (top levels: dst = the work vector, read past the L1; no other workgroup of the phase touches these rows).  Ends with a barrier.
template <int NCMAX, bool IN_PLACE>
__device__ __forceinline__ void s1_forward_tile(const Meta &M, int32_t s, int tile, const double *ys, double *part, double *__restrict__ dst, bool wt)
{
  const int32_t nc = M.first[s + 1] - M.first[s], nr = M.nrow[s];
  const int64_t ld = nc + nr;
  const double *P = M.panels + M.pptr[s];
  const int tid = threadIdx.x;
  const int r0 = tile * TILE, rl = tid & 63, sl = tid >> 6; // row rl of the tile, k slice [16 sl, 16 sl + 16)
  double acc = 0.0;
  if (r0 + rl < nr) {
    double w[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) w[u] = 16 * sl + u < nc ? P[nc + r0 + rl + (int64_t)(16 * sl + u) * ld] : 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += w[u] * ys[min(16 * sl + u, nc - 1)];
  }
  part[sl * TILE + rl] = acc;
  __syncthreads();
  if (tid < TILE && r0 + tid < nr) {
    double sum = 0.0;
#pragma unroll
    for (int q = 0; q < NCMAX / 16; ++q) sum += part[q * TILE + tid];
    if (IN_PLACE) {
      double *p = dst + (M.rows + M.rptr[s])[r0 + tid];
      s1_st(p, s1_ld<true>(p) - sum, wt);
    } else {
      dst[M.rptr[s] + r0 + tid] = sum;
    }
  }
  __syncthreads();
}
// acc[u] (thread = (row rl, k slice sl)) += block[r][16 sl + u] x[rows[r]] for the rows of one 64-row tile
__device__ __forceinline__ void s1_tile_tdot(const double *__restrict__ blk, int64_t bld, int32_t nc, int rn, int rl, int sl, double xr, double (&acc)[16])
{
  double w[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) w[u] = (rl < rn && 16 * sl + u < nc) ? blk[rl + (int64_t)(16 * sl + u) * bld] : 0.0;
#pragma unroll
  for (int u = 0; u < 16; ++u) acc[u] += w[u] * xr;
}
// out[0 .. nc) = sum over the rows [r_begin, r_end) of s of R_s[r][k] x[rows[r]] (L U: U_{s,rows}^T); out may be LDS or global
template <bool LU, int NCMAX, bool SC1>
__device__ __forceinline__ void s1_backward_rows(const Meta &M, int32_t s, int r_begin, int r_end, const double *__restrict__ X, double *out, bool out_wt)
{
  const int32_t nc = M.first[s + 1] - M.first[s], nr = M.nrow[s];
  const double *blk = LU ? M.upanels + M.uptr[s] : M.panels + M.pptr[s] + nc;
  const int64_t bld = LU ? (int64_t)nr : (int64_t)nc + nr;
  const int32_t *R = M.rows + M.rptr[s];
  const int tid = threadIdx.x, rl = tid & 63, sl = tid >> 6;
  double acc[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) acc[u] = 0.0;
  for (int r0 = r_begin; r0 < r_end; r0 += TILE) {
    const int rn = min(TILE, r_end - r0);
    s1_tile_tdot(blk + r0, bld, nc, rn, rl, sl, rl < rn ? s1_ld<SC1>(X + R[r0 + rl]) : 0.0, acc);
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const double v = s1_wave_sum(acc[u]);
    if (rl == 0 && 16 * sl + u < nc) s1_st(out + 16 * sl + u, v, out_wt);
  }
}
// x_s = W_s^T t (Cholesky) resp. U_ss^-1 t (L U), written to B[first .. first + nc).  t in LDS, complete before the call.
template <bool LU, int NCMAX>
__device__ __forceinline__ void s1_upper_product(const Meta &M, int32_t s, const double *t, double *part, double *__restrict__ B, bool wt)
{
  constexpr int XS = NCMAX / 4;
  const int32_t f = M.first[s], nc = M.first[s + 1] - f;
  const int64_t ld = nc + M.nrow[s];
  const double *P = M.panels + M.pptr[s];
  const int tid = threadIdx.x, i = tid & (NCMAX - 1), q = tid / NCMAX;
  double acc = 0.0;
  if (i < nc) {
    const int k0 = max(XS * q, i), k1 = min(XS * q + XS, nc);
    for (int kb = k0; kb < k1; kb += 8) {
      double w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = kb + u < k1 ? (LU ? P[i + (int64_t)(kb + u) * ld] : P[kb + u + (int64_t)i * ld]) : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += w[u] * t[min(kb + u, nc - 1)];
    }
  }
  part[q * NCMAX + i] = acc;
  __syncthreads();
  if (tid < nc) s1_st(B + f + tid, (part[tid] + part[NCMAX + tid]) + (part[2 * NCMAX + tid] + part[3 * NCMAX + tid]), wt);
  __syncthreads();
}

// ---- one launch per level (bottom of the tree; all levels when the persistent kernel is not used) ---------------------------------
// forward: workgroup = (supernode, item); every workgroup of a supernode recomputes y = W b from the inverse block (cheaper than a
// second launch per level: the levels are latency-bound); item 0 stores y, item 1 + t fills the slots of row tile t
template <bool LU, int NCMAX>
__global__ __launch_bounds__(4 * NCMAX) void k_sn_fwd1(Meta M, const int32_t *__restrict__ lev_sn, const int32_t *__restrict__ pre, int cnt, const double *__restrict__ B,
                                                  double *__restrict__ Y, double *__restrict__ contrib)
{
  __shared__ double bs[NCMAX], ys[NCMAX], part[4 * NCMAX];
  int lo = 0, hi = cnt; // item -> (supernode, local item): largest i with pre[i] + i <= item (1 + T_i items per supernode)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (pre[mid] + mid <= (int)blockIdx.x) lo = mid;
    else hi = mid;
  }
  const int32_t s = lev_sn[lo];
  const int item = (int)blockIdx.x - pre[lo] - lo;
  s1_gather<LU, NCMAX, false>(M, s, B, contrib, bs, part);
  s1_lower_product<LU, NCMAX>(M, s, bs, ys, part);
  if (item == 0) {
    const int32_t f = M.first[s], nc = M.first[s + 1] - f;
    if ((int)threadIdx.x < nc) Y[f + threadIdx.x] = ys[threadIdx.x];
    return;
  }
  s1_forward_tile<NCMAX, false>(M, s, item - 1, ys, part, contrib, false);
}
template <bool LU, int NCMAX>
__global__ __launch_bounds__(4 * NCMAX) void k_sn_bwd1_partial(Meta M, const int32_t *__restrict__ big_sn, const int32_t *__restrict__ pre, int cnt, const double *__restrict__ X,
                                                          double *__restrict__ partial)
{
  const int it = find_item(pre, cnt, (int32_t)blockIdx.x);
  const int32_t s = big_sn[it];
  const int item = (int)blockIdx.x - pre[it];
  const int32_t nr = M.nrow[s];
  s1_backward_rows<LU, NCMAX, false>(M, s, item * S1_CHUNK_ROWS, min(nr, (item + 1) * S1_CHUNK_ROWS), X, partial + (int64_t)blockIdx.x * SN_MAX_COLS, false);
}
template <bool LU, int NCMAX>
__global__ __launch_bounds__(4 * NCMAX) void k_sn_bwd1_diag(Meta M, const int32_t *__restrict__ lev_sn, const int32_t *__restrict__ big_index, const int32_t *__restrict__ pre_big,
                                                       const double *__restrict__ partial, const double *__restrict__ Y, double *__restrict__ B)
{
  __shared__ double t[NCMAX], part[4 * NCMAX];
  const int32_t s = lev_sn[blockIdx.x];
  const int32_t f = M.first[s], nc = M.first[s + 1] - f, nr = M.nrow[s];
  const int tid = threadIdx.x;
  const int bi = big_index[blockIdx.x];
  if (bi >= 0) {
    if (tid < nc) {
      double acc = Y[f + tid];
      const double *pp = partial + (int64_t)pre_big[bi] * SN_MAX_COLS + tid;
      const int npart = (nr + S1_CHUNK_ROWS - 1) / S1_CHUNK_ROWS;
      for (int q = 0; q < npart; ++q) acc -= pp[(int64_t)q * SN_MAX_COLS];
      t[tid] = acc;
    }
  } else {
    s1_backward_rows<LU, NCMAX, false>(M, s, 0, nr, B, part, false);
    __syncthreads();
    if (tid < nc) t[tid] = Y[f + tid] - part[tid];
  }
  __syncthreads();
  s1_upper_product<LU, NCMAX>(M, s, t, part, B, false);
}

// ---- the persistent kernel for the top of the tree --------------------------------------------------------------------------------
// Plan (host: build_top_plan): top levels j = 0 .. ntop - 1 (tree level ltop + j), supernodes split by class = block % 8.
// Item i of p_* writes partial[i * SN_MAX_COLS ..); p_first[s] = first item of s.
struct TopPlan {
  int32_t ntop = 0, nph = 0;                          // top levels; forward phases = sum over the levels of their colours
  const int32_t *a_ptr = nullptr, *a_sn = nullptr;    // [8 ntop + 1]: supernodes of (class, level)
  const int32_t *fph = nullptr;                       // [ntop + 1]: first forward phase of a level
  const int32_t *f_ptr = nullptr, *f_items = nullptr; // [8 nph + 1]: (supernode, row tile) pairs of (class, phase)
  const int32_t *p_ptr = nullptr, *p_items = nullptr; // [8 ntop + 1]: (supernode, row chunk) pairs of (class, level)
  const int32_t *p_first = nullptr;                   // [nsn]
  const int32_t *g_ptr = nullptr, *g_items = nullptr; // [8 + 1]: (supernode, 16-column piece) pairs of a class: the gather phase
};
struct TopSync { // device words of one factor (zero-initialised; the epoch separates launches)
  unsigned tickets[8];
  unsigned global_ticket, arrived, epoch, pad[5];
};
constexpr int TOP_THREADS = 4 * SN_MAX_COLS;
constexpr int TOP_FLAG_STRIDE = 16;  // unsigned long long words between two flags (128 bytes)
constexpr int TOP_MAX_WG = 1024;     // flag slots per group
__global__ void k_sn_top_prologue(TopSync *st)
{
  if (threadIdx.x < 8) st->tickets[threadIdx.x] = 0;
  if (threadIdx.x == 0) {
    st->global_ticket = 0;
    st->arrived = 0;
    st->epoch += 1;
  }
}
__device__ __forceinline__ unsigned s1_xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u; } // HW_REG_XCC_ID[3:0]

template <bool LU>
__global__ __launch_bounds__(TOP_THREADS) void k_sn_top1(Meta M, TopPlan P, int nblocks, int spread, double *__restrict__ B, double *__restrict__ Y, double *__restrict__ contrib,
                                                     double *__restrict__ partial, TopSync *st, unsigned long long *flags, unsigned *err)
{
  constexpr int NCMAX = SN_MAX_COLS;
  __shared__ double bs[NCMAX], ys[NCMAX], part[4 * NCMAX];
  __shared__ unsigned sh_xcc, sh_xt, sh_gt, sh_fail;
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) {
    const unsigned xcc = s1_xcc_id();
    sh_xcc = xcc;
    sh_xt = __hip_atomic_fetch_add(&st->tickets[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh_gt = __hip_atomic_fetch_add(&st->global_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&st->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned fail_ = 0;
    for (unsigned spins = 0; __hip_atomic_load(&st->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x; ++spins) {
      if (spins > (1u << 22)) { // the workgroups are not co-resident (another process on the GPU?)
        __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
  // XCD-local mode: class c (blocks b with b % 8 == c) lives on XCD c -- possible when every XCD that owns a block hosts workgroups
  const bool local_ok = !spread && __all(lane >= min(nblocks, 8) || tk >= 1u);
  const bool wt = !local_ok;
  const unsigned W = local_ok ? (unsigned)__builtin_amdgcn_readfirstlane((int)__shfl((int)tk, (int)xcc)) : gridDim.x; // workgroups of my group
  const unsigned rank = local_ok ? sh_xt : sh_gt;
  unsigned long long *gflags = flags + (size_t)(local_ok ? xcc : 8u) * TOP_MAX_WG * TOP_FLAG_STRIDE;
  const int c_begin = local_ok ? (int)xcc : 0, c_end = local_ok ? (int)xcc + 1 : 8;
  if (W > (unsigned)TOP_MAX_WG) {
    if (tid == 0) __hip_atomic_store(err, 7u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  unsigned count = 0;
  bool failed = false;
  // all-to-all flag barrier among the W workgroups of the group: every storing wave drains, one lane publishes, wave 0 polls all flags
  auto group_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++count;
    const unsigned long long want = ((unsigned long long)epoch << 32) | count;
    if (tid == 0) {
      if (wt) __hip_atomic_store(gflags + (size_t)rank * TOP_FLAG_STRIDE, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else gflags[(size_t)rank * TOP_FLAG_STRIDE] = want;
    }
    if (tid < 64) {
      for (unsigned spins = 0;; ++spins) {
        bool ok = true;
        for (unsigned j = (unsigned)lane; j < W; j += 64u) {
          const unsigned long long v = __hip_atomic_load(gflags + (size_t)j * TOP_FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = ok && (unsigned)(v >> 32) == epoch && (unsigned)v >= count;
        }
        if (__all(ok)) break;
        if (spins > (1u << 22)) {
          if (lane == 0) {
            __hip_atomic_store(err, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            sh_fail = 1;
          }
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (sh_fail) failed = true;
  };
  const int ntop = P.ntop;
  // ---- gather: every column of the top supernodes collects the slots the bottom levels left for it (16 columns per item, 32 lanes
  // per column strided over the list, partial sums folded in a fixed order) ----
  for (int c = c_begin; c < c_end; ++c)
    for (int i = P.g_ptr[c] + (int)rank; i < P.g_ptr[c + 1]; i += (int)W) {
      const int32_t s = P.g_items[2 * i], piece = P.g_items[2 * i + 1];
      const int32_t f = M.first[s], nc = M.first[s + 1] - f;
      const int k = 16 * piece + (tid >> 5), l32 = tid & 31;
      double acc = 0.0;
      if (k < nc) {
        const int64_t q1 = M.tmid[f + k];
        for (int64_t q = M.tptr[f + k] + l32; q < q1; q += 32) acc += contrib[M.tidx[q]]; // (slots of earlier launches: plain loads)
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
      if (k < nc && l32 == 0) s1_st(B + f + k, B[f + k] - acc, wt);
    }
  group_barrier();
  // ---- forward, bottom-up ----
  for (int j = 0; j < ntop && !failed; ++j) {
    for (int c = c_begin; c < c_end; ++c) {
      const int seg = c * ntop + j;
      for (int i = P.a_ptr[seg] + (int)rank; i < P.a_ptr[seg + 1]; i += (int)W) {
        const int32_t s = P.a_sn[i];
        const int32_t f = M.first[s], nc = M.first[s + 1] - f;
        if (tid < nc) bs[tid] = s1_ld<true>(B + f + (LU ? M.piv[f + tid] : tid));
        __syncthreads();
        s1_lower_product<LU, NCMAX>(M, s, bs, ys, part);
        if (tid < nc) s1_st(Y + f + tid, ys[tid], wt);
        __syncthreads();
      }
    }
    group_barrier();
    if (failed) break;
    for (int ph = P.fph[j]; ph < P.fph[j + 1] && !failed; ++ph) { // colour by colour: in-place subtraction without conflicts
      bool any = false;
      for (int c = c_begin; c < c_end; ++c) {
        const int seg = c * P.nph + ph;
        any = any || P.f_ptr[seg + 1] > P.f_ptr[seg];
        for (int i = P.f_ptr[seg] + (int)rank; i < P.f_ptr[seg + 1]; i += (int)W) {
          const int32_t s = P.f_items[2 * i], tile = P.f_items[2 * i + 1];
          const int32_t f = M.first[s], nc = M.first[s + 1] - f;
          if (tid < nc) ys[tid] = s1_ld<true>(Y + f + tid);
          __syncthreads();
          s1_forward_tile<NCMAX, true>(M, s, tile, ys, part, B, wt);
        }
      }
      if (any) group_barrier(); // (uniform over the group: the plan is the same for all its workgroups)
    }
  }
  // ---- backward, top-down ----
  for (int j = ntop - 1; j >= 0 && !failed; --j) {
    bool any = false;
    for (int c = c_begin; c < c_end; ++c) {
      const int seg = c * ntop + j;
      any = any || P.p_ptr[seg + 1] > P.p_ptr[seg];
      for (int i = P.p_ptr[seg] + (int)rank; i < P.p_ptr[seg + 1]; i += (int)W) {
        const int32_t s = P.p_items[2 * i], chunk = P.p_items[2 * i + 1];
        const int32_t nr = M.nrow[s];
        s1_backward_rows<LU, NCMAX, true>(M, s, chunk * S1_CHUNK_ROWS, min(nr, (chunk + 1) * S1_CHUNK_ROWS), B, partial + (int64_t)i * SN_MAX_COLS, wt);
      }
    }
    if (any) group_barrier();
    if (failed) break;
    for (int c = c_begin; c < c_end; ++c) {
      const int seg = c * ntop + j;
      for (int i = P.a_ptr[seg] + (int)rank; i < P.a_ptr[seg + 1]; i += (int)W) {
        const int32_t s = P.a_sn[i];
        const int32_t f = M.first[s], nc = M.first[s + 1] - f, nr = M.nrow[s];
        if (tid < nc) {
          double acc = s1_ld<true>(Y + f + tid);
          const int npart = (nr + S1_CHUNK_ROWS - 1) / S1_CHUNK_ROWS;
          const double *pp = partial + (int64_t)P.p_first[s] * SN_MAX_COLS + tid;
          for (int q = 0; q < npart; ++q) acc -= s1_ld<true>(pp + (int64_t)q * SN_MAX_COLS);
          bs[tid] = acc;
        }
        __syncthreads();
        s1_upper_product<LU, NCMAX>(M, s, bs, part, B, wt);
      }
    }
    group_barrier();
  }
}

} // namespace sn
