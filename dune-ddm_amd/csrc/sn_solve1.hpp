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
    for (; q + 12 < q1; q += 16) { // four loads in flight per thread (the slots of a row are contiguous: target-major layout)
      const double c0 = s1_ld<SC1>(contrib + q), c1 = s1_ld<SC1>(contrib + q + 4), c2 = s1_ld<SC1>(contrib + q + 8), c3 = s1_ld<SC1>(contrib + q + 12);
      acc = (((acc + c0) + c1) + c2) + c3;
    }
    for (; q < q1; q += 4) acc += s1_ld<SC1>(contrib + q);
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
      dst[M.tpos[M.rptr[s] + r0 + tid]] = sum; // (slot of this (supernode, row) in the ROW's list: the owner reads its list contiguously)
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

// ---- levels of SMALL supernodes (at most 64 columns: the leaves and the first levels above them, thousands per level) ---------------
// One WAVEFRONT per supernode, four per workgroup, no LDS and no barrier: lane = column (products with the inverse diagonal block)
// resp. lane = row (tiles of the block below); the vector entries of the other lanes come by shuffles.  The 256-thread kernels above
// spend a whole workgroup (and, forward, one per row tile) on ~30 columns x ~40 rows: 70-84 us per level and sweep on the DG problem
// (22 700 leaves), measured; profiles/r04_kernel_stats_sn_solve_dg.csv.
template <bool LU>
__global__ __launch_bounds__(256) void k_sn_fwd1_small(Meta M, const int32_t *__restrict__ lev_sn, int cnt, const double *__restrict__ B, double *__restrict__ Y,
                                                       double *__restrict__ contrib)
{
  const int lane = threadIdx.x & 63;
  const int it = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (it >= cnt) return; // (the whole wavefront: nothing below synchronises across wavefronts)
  const int32_t s = lev_sn[it];
  const int32_t f = M.first[s], nc = M.first[s + 1] - f, nr = M.nrow[s];
  const int64_t ld = nc + nr;
  const double *P = M.panels + M.pptr[s];
  double b = 0.0;
  if (lane < nc) {
    const int64_t row = f + (LU ? M.piv[f + lane] : lane);
    b = B[row];
    for (int64_t q = M.tptr[row]; q < M.tptr[row + 1]; ++q) b -= contrib[q]; // slots of the supernodes below, in list order
  }
  double y = LU ? b : 0.0; // (L U: unit lower factor, strict part stored)
  for (int k0 = 0; k0 < nc; k0 += 8) {
    double w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = (lane < nc && k0 + u < (LU ? lane : lane + 1)) ? P[lane + (int64_t)(k0 + u) * ld] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) y += w[u] * __shfl(b, k0 + u);
  }
  if (lane < nc) Y[f + lane] = y;
  for (int r0 = 0; r0 < nr; r0 += 64) {
    const int r = r0 + lane;
    double acc = 0.0;
    for (int k0 = 0; k0 < nc; k0 += 8) {
      double w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = (r < nr && k0 + u < nc) ? P[nc + r + (int64_t)(k0 + u) * ld] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += w[u] * __shfl(y, k0 + u);
    }
    if (r < nr) contrib[M.tpos[M.rptr[s] + r]] = acc;
  }
}
template <bool LU>
__global__ __launch_bounds__(256) void k_sn_bwd1_small(Meta M, const int32_t *__restrict__ lev_sn, int cnt, const double *__restrict__ Y, double *__restrict__ B)
{
  const int lane = threadIdx.x & 63;
  const int it = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (it >= cnt) return;
  const int32_t s = lev_sn[it];
  const int32_t f = M.first[s], nc = M.first[s + 1] - f, nr = M.nrow[s];
  const int64_t ld = nc + nr;
  const double *P = M.panels + M.pptr[s];
  const double *blk = LU ? M.upanels + M.uptr[s] : P + nc; // rows x columns of the block the backward sweep multiplies with
  const int64_t bld = LU ? (int64_t)nr : ld;
  const int32_t *R = M.rows + M.rptr[s];
  double t = lane < nc ? Y[f + lane] : 0.0; // lane = column k: t_k = y_k - sum_r block[r][k] x[rows[r]], rows in order
  for (int r0 = 0; r0 < nr; r0 += 64) {
    const double xr = r0 + lane < nr ? B[R[r0 + lane]] : 0.0;
    const int rn = min(64, nr - r0);
    for (int rr = 0; rr < rn; rr += 8) {
      double w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = (lane < nc && rr + u < rn) ? blk[r0 + rr + u + (int64_t)lane * bld] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) t -= w[u] * __shfl(xr, rr + u);
    }
  }
  double x = 0.0; // lane = column i: x_i = sum_{k >= i} (W^T resp. U^-1)[i][k] t_k
  for (int k0 = 0; k0 < nc; k0 += 8) {
    double w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + u;
      w[u] = (lane < nc && k >= lane && k < nc) ? (LU ? P[lane + (int64_t)k * ld] : P[k + (int64_t)lane * ld]) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) x += w[u] * __shfl(t, k0 + u);
  }
  if (lane < nc) B[f + lane] = x;
}

// ---- the persistent kernel for the top of the tree --------------------------------------------------------------------------------
// Plan (host: build_top_plan): top levels j = 0 .. ntop - 1 (tree level ltop + j), supernodes split by class = block % 8.
// Item i of p_* (a 64-row tile of a supernode) writes partial[i * SN_MAX_COLS ..); p_first[s] = first item of s.
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
                                                     double *__restrict__ partial, TopSync *st, unsigned long long *flags, unsigned *err, unsigned long long *stamps)
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
  // all-to-all flag barrier among the W workgroups of the group, in two halves so that panel loads for the next phase can be issued
  // in between: barrier_arrive -- every storing wave drains, one lane publishes; barrier_wait -- wave 0 polls all flags
  unsigned long long t_arrive = 0;
  auto barrier_arrive = [&]() __attribute__((always_inline)) {
    if (stamps) t_arrive = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++count;
    if (tid == 0) {
      const unsigned long long want = ((unsigned long long)epoch << 32) | count;
      if (wt) __hip_atomic_store(gflags + (size_t)rank * TOP_FLAG_STRIDE, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else {
        gflags[(size_t)rank * TOP_FLAG_STRIDE] = want;
        asm volatile("" ::: "memory");
      }
    }
  };
  auto barrier_wait = [&]() __attribute__((always_inline)) {
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
    // diagnostics (DDM_SN_TOP_STAMPS): rank 0 of the group on XCD 0 records when it arrived at barrier `count` and when it left
    if (stamps && rank == 0 && xcc == 0 && tid == 0 && count < 2000) {
      stamps[2 * count] = t_arrive;
      stamps[2 * count + 1] = __builtin_amdgcn_s_memrealtime();
    }
  };
  const int ntop = P.ntop;
  // ---- gather: every column of the top supernodes collects the slots the bottom levels left for it (16 columns per item, 32 lanes
  // per column strided over the row's contiguous slots, four loads in flight, partial sums folded in a fixed order) ----
  for (int c = c_begin; c < c_end; ++c)
    for (int i = P.g_ptr[c] + (int)rank; i < P.g_ptr[c + 1]; i += (int)W) {
      const int32_t s = P.g_items[2 * i], piece = P.g_items[2 * i + 1];
      const int32_t f = M.first[s], nc = M.first[s + 1] - f;
      const int k = 16 * piece + (tid >> 5), l32 = tid & 31;
      double acc = 0.0;
      if (k < nc) {
        const int64_t q1 = M.tmid[f + k];
        int64_t q = M.tptr[f + k] + l32;
        for (; q + 96 < q1; q += 128) {
          double v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = contrib[q + 32 * u]; // (slots of earlier launches: plain loads)
          acc += (v[0] + v[1]) + (v[2] + v[3]);
        }
        for (; q < q1; q += 32) acc += contrib[q];
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
      if (k < nc && l32 == 0) s1_st(B + f + k, B[f + k] - acc, wt);
    }
  // Register prefetch: what an item needs from the PANELS does not depend on anybody's results, so it is loaded before the barrier
  // the item waits behind -- the dependent part of a level is then: barrier, one sc1 round trip for the vector entries, arithmetic
  // on registers / LDS, one store.  (Without it every level was a chain of 4-5 dependent HBM round trips by one workgroup: ~36 us
  // per level and sweep pair, measured; tools/sn_solve_probe.py.)
  double wreg[32], treg[16];
  int pref_item = -1;
  // W_s slice of thread (i, sl): forward W[i][32 sl + u] (lower product), backward the transposed access of s1_upper_product
  auto load_w_fwd = [&](int32_t s) __attribute__((always_inline)) {
    const int32_t nc = M.first[s + 1] - M.first[s];
    const int64_t ld = nc + M.nrow[s];
    const double *Pn = M.panels + M.pptr[s];
    const int i = tid & (NCMAX - 1), sl = tid / NCMAX;
    const int k1 = i < nc ? min(32 * sl + 32, LU ? i : i + 1) : 0;
#pragma unroll
    for (int u = 0; u < 32; ++u) wreg[u] = 32 * sl + u < k1 ? Pn[i + (int64_t)(32 * sl + u) * ld] : 0.0;
  };
  auto load_w_bwd = [&](int32_t s) __attribute__((always_inline)) {
    const int32_t nc = M.first[s + 1] - M.first[s];
    const int64_t ld = nc + M.nrow[s];
    const double *Pn = M.panels + M.pptr[s];
    const int i = tid & (NCMAX - 1), q = tid / NCMAX;
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int k = 32 * q + u;
      wreg[u] = (i < nc && k >= i && k < nc) ? (LU ? Pn[i + (int64_t)k * ld] : Pn[k + (int64_t)i * ld]) : 0.0;
    }
  };
  // rows [64 tile, 64 tile + 64) of the block below the diagonal (upper = false: L_{rows,s} in the panel; true: the block the
  // backward sweep multiplies with: the same for Cholesky, U_{s,rows}^T for L U): thread (rl, sl) holds columns 16 sl .. 16 sl + 15
  auto load_tile = [&](int32_t s, int tile, bool upper) __attribute__((always_inline)) {
    const int32_t nc = M.first[s + 1] - M.first[s], nr = M.nrow[s];
    const double *blk = (LU && upper) ? M.upanels + M.uptr[s] : M.panels + M.pptr[s] + nc;
    const int64_t bld = (LU && upper) ? (int64_t)nr : (int64_t)nc + nr;
    const int r = tile * TILE + (tid & 63), sl = tid >> 6;
#pragma unroll
    for (int u = 0; u < 16; ++u) treg[u] = (tile >= 0 && r < nr && 16 * sl + u < nc) ? blk[r + (int64_t)(16 * sl + u) * bld] : 0.0;
  };
  auto product_from_regs = [&](int32_t nc, const double *vec, double *out, bool add_unit) __attribute__((always_inline)) {
    // out[i] = sum_k wreg(i, k) vec[k] (+ vec[i] for the unit diagonal of the L U lower factor); ends with a barrier
    const int i = tid & (NCMAX - 1), sl = tid / NCMAX;
    double acc = 0.0;
#pragma unroll
    for (int u0 = 0; u0 < 32; u0 += 8) { // (eight LDS operands at a time: all 32 at once would not fit beside the prefetched slices)
      double v8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v8[u] = vec[min(32 * sl + u0 + u, NCMAX - 1)];
      asm volatile("" : "+v"(v8[0]), "+v"(v8[1]), "+v"(v8[2]), "+v"(v8[3]), "+v"(v8[4]), "+v"(v8[5]), "+v"(v8[6]), "+v"(v8[7]));
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += wreg[u0 + u] * v8[u];
    }
    part[sl * NCMAX + i] = acc;
    __syncthreads();
    if (tid < NCMAX) out[tid] = tid < nc ? (add_unit ? vec[tid] : 0.0) + ((part[tid] + part[NCMAX + tid]) + (part[2 * NCMAX + tid] + part[3 * NCMAX + tid])) : 0.0; // (zeros beyond nc: the callers multiply them with zero panel entries)
    __syncthreads();
  };
  // first item of a forward phase / backward partial phase / backward tail phase that this workgroup will work on (-1: none)
  auto first_fwd = [&](int ph) { for (int c = c_begin; c < c_end; ++c) { const int i = P.f_ptr[c * P.nph + ph] + (int)rank; if (i < P.f_ptr[c * P.nph + ph + 1]) return i; } return -1; };
  auto first_part = [&](int j) { for (int c = c_begin; c < c_end; ++c) { const int i = P.p_ptr[c * ntop + j] + (int)rank; if (i < P.p_ptr[c * ntop + j + 1]) return i; } return -1; };
  auto first_tail = [&](int j) { for (int c = c_begin; c < c_end; ++c) { const int i = P.a_ptr[c * ntop + j] + (int)rank; if (i < P.a_ptr[c * ntop + j + 1]) return i; } return -1; };
  barrier_arrive();
  if (ntop > 0 && P.fph[1] > P.fph[0]) {
    pref_item = first_fwd(0);
    if (pref_item >= 0) {
      load_w_fwd(P.f_items[2 * pref_item]);
      load_tile(P.f_items[2 * pref_item], P.f_items[2 * pref_item + 1], false);
    }
  }
  barrier_wait();
  // ---- forward, bottom-up.  An item = (supernode s, row tile t); its workgroup recomputes y_s = W_s b_s itself (no barrier between
  // "y" and "tiles"; the 128 x 128 inverse block comes from L2 for all but the first), tile 0 (or the pseudo tile -1 of a supernode
  // without rows) also stores y_s for the backward sweep.  Colour by colour: in-place subtraction without conflicts. ----
  const int nph = P.nph;
  for (int ph = 0; ph < nph && !failed; ++ph) {
    for (int c = c_begin; c < c_end; ++c) {
      const int seg = c * nph + ph;
      for (int i = P.f_ptr[seg] + (int)rank; i < P.f_ptr[seg + 1]; i += (int)W) {
        const int32_t s = P.f_items[2 * i], tile = P.f_items[2 * i + 1];
        const int32_t f = M.first[s], nc = M.first[s + 1] - f, nr = M.nrow[s];
        if (i != pref_item) {
          load_w_fwd(s);
          load_tile(s, tile, false);
        }
        if (tid < nc) bs[tid] = s1_ld<true>(B + f + (LU ? M.piv[f + tid] : tid));
        if (tid >= nc && tid < NCMAX) bs[tid] = 0.0;
        __syncthreads();
        product_from_regs(nc, bs, ys, LU);
        if (tile <= 0 && tid < nc) s1_st(Y + f + tid, ys[tid], wt);
        if (tile >= 0) {
          const int rl = tid & 63, sl = tid >> 6;
          double acc = 0.0;
#pragma unroll
          for (int u = 0; u < 16; ++u) acc += treg[u] * ys[min(16 * sl + u, NCMAX - 1)];
          part[sl * TILE + rl] = acc;
          __syncthreads();
          if (tid < TILE && tile * TILE + tid < nr) {
            double sum = 0.0;
#pragma unroll
            for (int q = 0; q < NCMAX / 16; ++q) sum += part[q * TILE + tid];
            double *pb = B + (M.rows + M.rptr[s])[tile * TILE + tid];
            s1_st(pb, s1_ld<true>(pb) - sum, wt);
          }
        }
        __syncthreads();
      }
    }
    // what the next phase (or the first backward phase) needs from the panels: issued between the two halves of the barrier
    barrier_arrive();
    pref_item = -1;
    if (ph + 1 < nph) {
      pref_item = first_fwd(ph + 1);
      if (pref_item >= 0) {
        load_w_fwd(P.f_items[2 * pref_item]);
        load_tile(P.f_items[2 * pref_item], P.f_items[2 * pref_item + 1], false);
      }
    } else {
      pref_item = first_part(ntop - 1);
      if (pref_item >= 0) load_tile(P.p_items[2 * pref_item], P.p_items[2 * pref_item + 1], true);
    }
    barrier_wait();
  }
  // ---- backward, top-down: partial[item] = (64 rows of the block)^T x(rows); barrier; x_s = W_s^T (y_s - sum of the partials) ----
  for (int j = ntop - 1; j >= 0 && !failed; --j) {
    for (int c = c_begin; c < c_end; ++c) {
      const int seg = c * ntop + j;
      for (int i = P.p_ptr[seg] + (int)rank; i < P.p_ptr[seg + 1]; i += (int)W) {
        const int32_t s = P.p_items[2 * i], tile = P.p_items[2 * i + 1];
        const int32_t nc = M.first[s + 1] - M.first[s], nr = M.nrow[s];
        if (i != pref_item) load_tile(s, tile, true);
        const int rl = tid & 63, sl = tid >> 6, r = tile * TILE + rl;
        const double xr = r < nr ? s1_ld<true>(B + (M.rows + M.rptr[s])[r]) : 0.0;
        double *out = partial + (int64_t)i * SN_MAX_COLS;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const double v = s1_wave_sum(treg[u] * xr);
          if (rl == 0 && 16 * sl + u < nc) s1_st(out + 16 * sl + u, v, wt);
        }
      }
    }
    barrier_arrive();
    pref_item = first_tail(j);
    if (pref_item >= 0) load_w_bwd(P.a_sn[pref_item]);
    barrier_wait();
    if (failed) break;
    for (int c = c_begin; c < c_end; ++c) {
      const int seg = c * ntop + j;
      for (int i = P.a_ptr[seg] + (int)rank; i < P.a_ptr[seg + 1]; i += (int)W) {
        const int32_t s = P.a_sn[i];
        const int32_t f = M.first[s], nc = M.first[s + 1] - f, nr = M.nrow[s];
        if (i != pref_item) load_w_bwd(s);
        { // t = y_s - sum of the tile partials: the four thread groups take every fourth partial (four loads in flight each), the
          // group sums are folded in a fixed order
          const int i = tid & (NCMAX - 1), g = tid / NCMAX;
          double acc = 0.0;
          if (i < nc) {
            const int npart = (nr + TILE - 1) / TILE;
            const double *pp = partial + (int64_t)P.p_first[s] * SN_MAX_COLS + i;
            int q = g;
            for (; q + 12 < npart; q += 16) {
              const double p0 = s1_ld<true>(pp + (int64_t)q * SN_MAX_COLS), p1 = s1_ld<true>(pp + (int64_t)(q + 4) * SN_MAX_COLS), p2 = s1_ld<true>(pp + (int64_t)(q + 8) * SN_MAX_COLS),
                           p3 = s1_ld<true>(pp + (int64_t)(q + 12) * SN_MAX_COLS);
              acc = (((acc + p0) + p1) + p2) + p3;
            }
            for (; q < npart; q += 4) acc += s1_ld<true>(pp + (int64_t)q * SN_MAX_COLS);
          }
          part[g * NCMAX + i] = acc;
          __syncthreads();
          if (tid < NCMAX) bs[tid] = tid < nc ? s1_ld<true>(Y + f + tid) - ((part[tid] + part[NCMAX + tid]) + (part[2 * NCMAX + tid] + part[3 * NCMAX + tid])) : 0.0;
        }
        __syncthreads();
        product_from_regs(nc, bs, ys, false);
        if (tid < nc) s1_st(B + f + tid, ys[tid], wt);
        __syncthreads();
      }
    }
    barrier_arrive();
    pref_item = j > 0 ? first_part(j - 1) : -1;
    if (pref_item >= 0) load_tile(P.p_items[2 * pref_item], P.p_items[2 * pref_item + 1], true);
    barrier_wait();
  }
}


// ---- CHAINS: the separators of the top levels as dense units ------------------------------------------------------------------------
// A separator wider than SN_MAX_COLS columns is a chain of links s -> s + 1 = parent(s), every link the only child of the next: m links
// are m tree levels, i.e. 4 m barriers (or 3 m launches) of a few microseconds each for panels of one or two megabytes -- on the
// elasticity problem the 20 links of the root separator and the chains below it are 1.1 of the 1.7 ms of a solve (measured, barrier
// log of DDM_SN_TOP_STAMPS).  For the single-vector solves a chain C (columns [col0, col0 + n), links i = 0 .. m - 1, external rows E =
// the rows of its last link) is therefore ONE unit with an explicitly inverted triangle:
//     W = L_CC^-1 (n x n, lower; L U: the row exchanges of the links are absorbed, y_C = W b_C),   E = L_EC (n_E x n),
//     L U only:  V = (U_CC^T)^-1 (lower, = the transpose of U_CC^-1),   U = U_CE^T (n_E x n);
//     forward   y_C = W b_C;   b_E -= E y_C                       backward   t = y_C - (E | U)^T x_E;   x_C = (W | V)^T t
// -- four dense matrix-vector products per chain, each spread over all workgroups of the chain's XCD, two barriers per sweep and
// chain LEVEL (chains whose children are all done).  The blocked inversion runs once per factorisation (k_chain_scatter,
// k_chain_invert): X_jj = M_j (the inverted diagonal blocks the factorisation already holds), X_ij = -M_i sum_{k = j}^{i-1} L_ik X_kj,
// all blocks of one distance d = i - j in one launch.  The dense triangle has as many entries as the in-chain parts of the panels it
// replaces in the solves: the bytes per solve do not grow.
struct ChainDev {
  int32_t nchain = 0;
  const int32_t *col0 = nullptr, *ncol = nullptr, *first_sn = nullptr, *nlinks = nullptr, *last_sn = nullptr, *nE = nullptr; // [nchain]
  const int64_t *woff = nullptr, *eoff = nullptr;                                                                        // [nchain]
  double *W = nullptr, *E = nullptr, *V = nullptr, *U = nullptr;
};
// dense copies of the chain's coupling blocks: thread block = (link, 64-row tile of [diagonal block; rows below])
template <bool LU>
__global__ __launch_bounds__(256) void k_chain_scatter(Meta M, ChainDev C, const int32_t *__restrict__ link_sn, const int32_t *__restrict__ link_chain,
                                                       const int32_t *__restrict__ pre, int nlinks_total, double *__restrict__ Ltmp, double *__restrict__ UTtmp)
{
  const int it = find_item(pre, nlinks_total, (int32_t)blockIdx.x);
  const int32_t s = link_sn[it], c = link_chain[it];
  const int tile = (int)blockIdx.x - pre[it];
  const int32_t f = M.first[s], nc = M.first[s + 1] - f, nr = M.nrow[s];
  const int64_t ld = nc + nr;
  const int64_t n = C.ncol[c], nE = C.nE[c];
  const int32_t col0 = C.col0[c], co = f - col0;
  const double *P = M.panels + M.pptr[s];
  double *W = C.W + C.woff[c], *E = C.E + C.eoff[c];
  double *V = LU ? C.V + C.woff[c] : nullptr, *U = LU ? C.U + C.eoff[c] : nullptr;
  double *Lt = Ltmp + C.woff[c], *Ut = LU ? UTtmp + C.woff[c] : nullptr;
  const int32_t *erows = M.rows + M.rptr[C.last_sn[c]];
  for (int idx = threadIdx.x; idx < TILE * nc; idx += 256) {
    const int rr = tile * TILE + idx % TILE, j = idx / TILE; // row rr of [diag; rows], column j of the link
    if (rr >= nc + nr) continue;
    if (rr < nc) { // diagonal block: M_k (and, L U, M2_k = (U_kk^-1)^T) straight into the diagonal block of the result
      const int i = rr;
      if (!LU) {
        if (j <= i) W[(co + i) + (int64_t)(co + j) * n] = P[i + (int64_t)j * ld];
      } else {
        // y = L^-1 (P b), (P b)[q] = b[piv[q]]:  M[i][piv[q]] = L^-1[i][q]  (unit diagonal implied)
        const double linv = j < i ? P[i + (int64_t)j * ld] : (j == i ? 1.0 : 0.0);
        if (j <= i) W[(co + i) + (int64_t)(co + M.piv[f + j]) * n] = linv;
        if (j >= i) V[(co + j) + (int64_t)(co + i) * n] = P[i + (int64_t)j * ld]; // U^-1[i][j] -> lower triangle of the transpose
      }
      continue;
    }
    const int r = rr - nc;
    const int32_t g = (M.rows + M.rptr[s])[r];
    const double l = P[nc + r + (int64_t)j * ld];
    const double u = LU ? (M.upanels + M.uptr[s])[r + (int64_t)j * nr] : 0.0;
    if (g < col0 + (int32_t)n) { // inside the chain
      Lt[(g - col0) + (int64_t)(co + j) * n] = l;
      if (LU) Ut[(g - col0) + (int64_t)(co + j) * n] = u;
    } else {
      const int32_t e = lower_bound_i32(erows, (int32_t)nE, g);
      E[e + (int64_t)(co + j) * nE] = l;
      if (LU) U[e + (int64_t)(co + j) * nE] = u;
    }
  }
}
// one distance d = i - j of the blocked inversion, all chains: thread block = (chain, i, 64-column half of block j);
// X_ij = -M_i (sum_{k = j}^{i-1} L_ik X_kj), L = Lsrc (dense copy of the in-chain coupling), X = the result array (diagonal blocks = M)
__global__ __launch_bounds__(512) void k_chain_invert(Meta M, ChainDev C, int d, const int32_t *__restrict__ item_chain, const int32_t *__restrict__ item_i, const double *__restrict__ Lsrc,
                                                      double *__restrict__ X)
{
  extern __shared__ __attribute__((aligned(16))) double sm[]; // T: 128 x 64, Xs: 32 x 64
  double *T = sm, *Xs = sm + SN_MAX_COLS * 64;
  const int c = item_chain[blockIdx.x >> 1], i = item_i[blockIdx.x >> 1], h = (int)blockIdx.x & 1, j = i - d;
  const int64_t n = C.ncol[c];
  const int32_t col0 = C.col0[c], s0 = C.first_sn[c];
  const int co_i = M.first[s0 + i] - col0, ni = M.first[s0 + i + 1] - M.first[s0 + i];
  const int co_j = M.first[s0 + j] - col0, nj = M.first[s0 + j + 1] - M.first[s0 + j];
  const int K = co_i - co_j; // columns of the links j .. i - 1
  const double *L = Lsrc + C.woff[c];
  double *Xc = X + C.woff[c];
  const int tid = threadIdx.x, r = tid & (SN_MAX_COLS - 1), cg = tid / SN_MAX_COLS; // row r of block i, columns 16 cg .. 16 cg + 15 of the half
  const int cbase = 64 * h;
  if (cbase >= nj) return;
  double acc[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) acc[u] = 0.0;
  for (int k0 = 0; k0 < K; k0 += 32) {
    __syncthreads();
    for (int idx = tid; idx < 32 * 64; idx += 512) { // X[co_j + k0 + kk][co_j + cbase + cc]
      const int kk = idx & 31, cc = idx >> 5;
      Xs[kk * 64 + cc] = (k0 + kk < K && cbase + cc < nj) ? Xc[(co_j + k0 + kk) + (int64_t)(co_j + cbase + cc) * n] : 0.0;
    }
    __syncthreads();
    const int kn = min(32, K - k0);
    for (int kk = 0; kk < kn; ++kk) {
      const double a = r < ni ? L[(co_i + r) + (int64_t)(co_j + k0 + kk) * n] : 0.0;
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u] += a * Xs[kk * 64 + 16 * cg + u];
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 16; ++u) T[r * 64 + 16 * cg + u] = acc[u];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 16; ++u) acc[u] = 0.0;
  for (int q = 0; q < ni; ++q) { // -M_i T
    const double m = r < ni ? Xc[(co_i + r) + (int64_t)(co_i + q) * n] : 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u] -= m * T[q * 64 + 16 * cg + u];
  }
  if (r < ni)
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (cbase + 16 * cg + u < nj) Xc[(co_i + r) + (int64_t)(co_j + cbase + 16 * cg + u) * n] = acc[u];
}

// Plan of the chain kernel (host: build_chain_plan): chain levels L = 0 .. nclev - 1, classes c = block % 8, segment (c, L) -> c nclev + L.
// Items are (chain, piece) pairs (y_*: triples with the number of columns the rows reach): y_*: 64-row blocks of the triangle (forward product, largest first); e_*: 64-row tiles of the
// external rows, by (class, phase) with phases = chain levels x colours (eph: first phase of a level); t_* / x_*: 64-column blocks
// (backward).  g_*: as in TopPlan (the gather of the bottom levels' slots).
struct ChainPlan {
  int32_t nclev = 0, nph = 0;
  const int32_t *y_ptr = nullptr, *y_items = nullptr;
  const int32_t *eph = nullptr, *e_ptr = nullptr, *e_items = nullptr;
  const int32_t *t_ptr = nullptr, *t_items = nullptr;
  const int32_t *x_ptr = nullptr, *x_items = nullptr;
  const int32_t *g_ptr = nullptr, *g_items = nullptr;
};

template <bool LU>
__global__ __launch_bounds__(TOP_THREADS) void k_sn_top_chain(Meta M, ChainDev C, ChainPlan P, int nblocks, int spread, double *__restrict__ B, double *__restrict__ Y,
                                                          const double *__restrict__ contrib, TopSync *st, unsigned long long *flags, unsigned *err, unsigned long long *stamps)
{
  extern __shared__ __attribute__((aligned(16))) double vec[]; // a chain's vector (max(n, n_E) doubles)
  __shared__ double part[8 * 64];
  __shared__ unsigned sh_xcc, sh_xt, sh_gt, sh_fail;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  const bool local_ok = !spread && __all(lane >= min(nblocks, 8) || tk >= 1u);
  const bool wt = !local_ok;
  const unsigned W = local_ok ? (unsigned)__builtin_amdgcn_readfirstlane((int)__shfl((int)tk, (int)xcc)) : gridDim.x;
  const unsigned rank = local_ok ? sh_xt : sh_gt;
  unsigned long long *gflags = flags + (size_t)(local_ok ? xcc : 8u) * TOP_MAX_WG * TOP_FLAG_STRIDE;
  const int c_begin = local_ok ? (int)xcc : 0, c_end = local_ok ? (int)xcc + 1 : 8;
  if (W > (unsigned)TOP_MAX_WG) {
    if (tid == 0) __hip_atomic_store(err, 7u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  unsigned count = 0;
  bool failed = false;
  auto group_barrier = [&]() __attribute__((always_inline)) { // all-to-all flags among the W workgroups of the group (see k_sn_top1)
    const unsigned long long t_arrive = stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++count;
    if (tid == 0) {
      const unsigned long long want = ((unsigned long long)epoch << 32) | count;
      if (wt) __hip_atomic_store(gflags + (size_t)rank * TOP_FLAG_STRIDE, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else {
        gflags[(size_t)rank * TOP_FLAG_STRIDE] = want;
        asm volatile("" ::: "memory");
      }
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
    if (stamps && rank == 0 && xcc == 0 && tid == 0 && count < 2000) {
      stamps[2 * count] = t_arrive;
      stamps[2 * count + 1] = __builtin_amdgcn_s_memrealtime();
    }
  };
  // sum over the 8 wavefronts of part[w * 64 + lane] in a fixed order (valid for tid < 64 after the call's barrier)
  auto fold8 = [&](double v) __attribute__((always_inline)) -> double {
    part[wave * 64 + lane] = v;
    __syncthreads();
    double sres = 0.0;
    if (tid < 64) sres = ((part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid])) + ((part[256 + tid] + part[320 + tid]) + (part[384 + tid] + part[448 + tid]));
    __syncthreads();
    return sres;
  };
  // rows [r0, r0 + 64) of a column-major matrix (leading dimension ldm) times vec[0 .. kmax): lane = row, the wavefronts share the
  // k range; `tri`: only k <= row (lower triangle)
  auto rows_times_vec = [&](const double *Mx, int64_t ldm, int nrows, int r0, int kmax, bool tri) __attribute__((always_inline)) -> double {
    const int row = r0 + lane;
    const int chunk = ((kmax + 127) / 128) * 16; // multiple of 16 per wavefront
    const int ks = wave * chunk, ke = min(kmax, ks + chunk);
    double acc = 0.0;
    for (int k = ks; k < ke; k += 16) { // sixteen loads in flight per thread (these products stream the triangle: bandwidth-bound)
      double w[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) w[u] = (row < nrows && k + u < ke && (!tri || k + u <= row)) ? Mx[row + (int64_t)(k + u) * ldm] : 0.0;
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += w[u] * vec[min(k + u, kmax - 1)];
    }
    return fold8(acc);
  };
  // column `col` of a column-major matrix, rows [k_begin, k_end), times vec[rows]: one wavefront, lanes stride over the rows
  auto col_times_vec = [&](const double *Mx, int64_t ldm, int col, int k_begin, int k_end) __attribute__((always_inline)) -> double {
    double acc = 0.0;
    const double *cp = Mx + (int64_t)col * ldm;
    int k = k_begin + lane;
    for (; k + 448 < k_end; k += 512) {
      double w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = cp[k + 64 * u];
      acc += ((w[0] * vec[k] + w[1] * vec[k + 64]) + (w[2] * vec[k + 128] + w[3] * vec[k + 192])) + ((w[4] * vec[k + 256] + w[5] * vec[k + 320]) + (w[6] * vec[k + 384] + w[7] * vec[k + 448]));
    }
    for (; k + 192 < k_end; k += 256) {
      const double w0 = cp[k], w1 = cp[k + 64], w2 = cp[k + 128], w3 = cp[k + 192];
      acc += ((w0 * vec[k] + w1 * vec[k + 64]) + (w2 * vec[k + 128] + w3 * vec[k + 192]));
    }
    for (; k < k_end; k += 64) acc += cp[k] * vec[k];
    return s1_wave_sum(acc);
  };
  // ---- gather of the bottom levels' slots (as in k_sn_top1) ----
  for (int c = c_begin; c < c_end; ++c)
    for (int i = P.g_ptr[c] + (int)rank; i < P.g_ptr[c + 1]; i += (int)W) {
      const int32_t s = P.g_items[2 * i], piece = P.g_items[2 * i + 1];
      const int32_t f = M.first[s], nc = M.first[s + 1] - f;
      const int k = 16 * piece + (tid >> 5), l32 = tid & 31;
      double acc = 0.0;
      if (k < nc) {
        const int64_t q1 = M.tmid[f + k];
        int64_t q = M.tptr[f + k] + l32;
        for (; q + 96 < q1; q += 128) {
          double v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = contrib[q + 32 * u];
          acc += (v[0] + v[1]) + (v[2] + v[3]);
        }
        for (; q < q1; q += 32) acc += contrib[q];
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
      if (k < nc && l32 == 0) s1_st(B + f + k, B[f + k] - acc, wt);
    }
  group_barrier();
  const int nclev = P.nclev;
  // ---- forward ----
  for (int L = 0; L < nclev && !failed; ++L) {
    for (int c = c_begin; c < c_end; ++c) { // y_C = W b_C
      const int seg = c * nclev + L;
      for (int i = P.y_ptr[seg] + (int)rank; i < P.y_ptr[seg + 1]; i += (int)W) {
        const int32_t ch = P.y_items[3 * i], rb = P.y_items[3 * i + 1], kmax = P.y_items[3 * i + 2]; // kmax: end of the last row's link (L U: full diagonal blocks)
        const int n = C.ncol[ch], col0 = C.col0[ch], r0 = 64 * rb;
        for (int k = tid; k < kmax; k += TOP_THREADS) vec[k] = s1_ld<true>(B + col0 + k);
        __syncthreads();
        const double y = rows_times_vec(C.W + C.woff[ch], n, n, r0, kmax, !LU);
        if (tid < 64 && r0 + tid < n) s1_st(Y + col0 + r0 + tid, y, wt);
        __syncthreads();
      }
    }
    group_barrier();
    if (failed) break;
    for (int ph = P.eph[L]; ph < P.eph[L + 1] && !failed; ++ph) { // b_E -= E y_C, colour by colour
      bool any = false;
      for (int c = c_begin; c < c_end; ++c) {
        const int seg = c * P.nph + ph;
        any = any || P.e_ptr[seg + 1] > P.e_ptr[seg];
        for (int i = P.e_ptr[seg] + (int)rank; i < P.e_ptr[seg + 1]; i += (int)W) {
          const int32_t ch = P.e_items[2 * i], t = P.e_items[2 * i + 1];
          const int n = C.ncol[ch], col0 = C.col0[ch], nE = C.nE[ch];
          for (int k = tid; k < n; k += TOP_THREADS) vec[k] = s1_ld<true>(Y + col0 + k);
          __syncthreads();
          const double v = rows_times_vec(C.E + C.eoff[ch], nE, nE, 64 * t, n, false);
          if (tid < 64 && 64 * t + tid < nE) {
            double *pb = B + (M.rows + M.rptr[C.last_sn[ch]])[64 * t + tid];
            s1_st(pb, s1_ld<true>(pb) - v, wt);
          }
          __syncthreads();
        }
      }
      if (any) group_barrier();
    }
  }
  // ---- backward ----
  for (int L = nclev - 1; L >= 0 && !failed; --L) {
    bool any = false;
    for (int c = c_begin; c < c_end; ++c) { // t = y_C - (E | U)^T x_E, in place in Y
      const int seg = c * nclev + L;
      any = any || P.t_ptr[seg + 1] > P.t_ptr[seg];
      for (int i = P.t_ptr[seg] + (int)rank; i < P.t_ptr[seg + 1]; i += (int)W) {
        const int32_t ch = P.t_items[2 * i], cb = P.t_items[2 * i + 1];
        const int n = C.ncol[ch], col0 = C.col0[ch], nE = C.nE[ch];
        const int32_t *er = M.rows + M.rptr[C.last_sn[ch]];
        for (int e = tid; e < nE; e += TOP_THREADS) vec[e] = s1_ld<true>(B + er[e]);
        __syncthreads();
        const double *Mx = (LU ? C.U : C.E) + C.eoff[ch];
        for (int q = 0; q < 8; ++q) {
          const int col = 64 * cb + 8 * wave + q;
          if (col < n) {
            const double v = col_times_vec(Mx, nE, col, 0, nE);
            if (lane == 0) s1_st(Y + col0 + col, s1_ld<true>(Y + col0 + col) - v, wt);
          }
        }
        __syncthreads();
      }
    }
    if (any) group_barrier();
    if (failed) break;
    for (int c = c_begin; c < c_end; ++c) { // x_C = (W | V)^T t
      const int seg = c * nclev + L;
      for (int i = P.x_ptr[seg] + (int)rank; i < P.x_ptr[seg + 1]; i += (int)W) {
        const int32_t ch = P.x_items[2 * i], cb = P.x_items[2 * i + 1];
        const int n = C.ncol[ch], col0 = C.col0[ch], k0 = 64 * cb;
        for (int k = k0 + tid; k < n; k += TOP_THREADS) vec[k] = s1_ld<true>(Y + col0 + k);
        __syncthreads();
        const double *Mx = (LU ? C.V : C.W) + C.woff[ch];
        for (int q = 0; q < 8; ++q) {
          const int col = k0 + 8 * wave + q;
          if (col < n) {
            const double v = col_times_vec(Mx, n, col, col, n);
            if (lane == 0) s1_st(B + col0 + col, v, wt);
          }
        }
        __syncthreads();
      }
    }
    group_barrier();
  }
}

} // namespace sn
