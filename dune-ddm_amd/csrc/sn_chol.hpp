// Supernodal sparse Cholesky ON THE DEVICE: numeric factorisation and the (multi-right-hand-side) triangular solves of the sparse
// direct local solver -- what the reference obtains from CHOLMOD / UMFPACK through the dune-istl solver factory
// (dune/ddm/schwarz.hh:85-92, examples/poisson.ini:23,26), from `SymShiftInvert` inside the GenEO eigensolver
// (dune/ddm/eigensolvers/spectra.hh:28-89) and from the multi-RHS solver dune/ddm/eigensolvers/umfpack.hh:16-333.
// Host: ordering + symbolic analysis only (sn_chol_host.hpp).  Everything with arithmetic is below.
//
// Data: all diagonal blocks (subdomains) of the rank share one supernode list in GLOBAL permuted numbering; supernode s owns the
// columns [first[s], first[s+1]) (at most SN_MAX_COLS) and a dense column-major panel [D_s; R_s] of (ncol + nrow) x ncol doubles,
// D_s = diagonal block (lower triangle used), R_s = the rows `rows[rptr[s] .. rptr[s+1])` below it.
// After the factorisation D_s holds W_s = L_ss^-1 (explicit inverse of the Cholesky factor of the diagonal block: every solve with
// it is a product) and R_s holds L_{rows, s}.
//
// Factorisation = right-looking, level by level of the supernodal elimination tree (supernodes of one level are independent):
//   k_sn_diag    one workgroup per supernode: Cholesky + in-place triangular inverse of the diagonal block in LDS;
//   k_sn_panel   R_s <- R_s W_s^T                                   (FP64 MFMA, one workgroup per 64 rows);
//   k_sn_update  U = R_s R_s^T (lower triangle, 64 x 64 tiles, FP64 MFMA) subtracted from the panels of the ancestors that own
//                the columns `rows[...]`: the row positions inside a target panel are found by binary search once per (tile,
//                target) in LDS, the subtraction is a hardware FP64 atomic add (two supernodes of one level may update the same
//                ancestor entry; the ORDER of these additions is not fixed, so the factor is reproducible to rounding only).
// Solves (row-major n x m work block in the permuted numbering, in place), level by level:
//   forward   k_sn_fwd_diag: Y_s = W_s B_s;   k_sn_fwd_update: B_rows -= R_s Y_s (atomic adds, as above);
//   backward  k_sn_bwd_partial (supernodes with many rows): per 64-row tile R_tile^T X_rows;   k_sn_bwd_diag: X_s = W_s^T (Y_s - R_s^T X_rows),
//             tile partials summed in tile order.
// Bounding roofline: the factorisation is FP64-MFMA work (flop count from the symbolic analysis), the solves stream the panels once
// per sweep: 8 * entries bytes (+ the work block).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "sn_chol_host.hpp"

namespace sn {
// runs f once per device (hipFuncSetAttribute is per device; callers may come from several host threads: late-comers wait
// until the first one has finished)
struct DeviceOnce {
  std::mutex m;
  uint64_t done = 0; // bit = device id
  template <class F>
  void run(F &&f)
  {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    std::lock_guard<std::mutex> lock(m);
    if (done & bit) return;
    f();
    done |= bit;
  }
};

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int TILE = 64;      // rows per panel / update tile
constexpr int BWD_SMALL = 8;  // supernodes with at most this many row tiles do their backward reduction inside k_sn_bwd_diag
constexpr int UPD_KC = 32;   // columns of the panels staged through LDS per trip of the update kernel
constexpr int BWD_ROWS = 256; // rows per partial product of the others (one workgroup, four 64-row sub-tiles)

struct Meta { // device pointers
  int32_t nsn;
  const int32_t *first;     // [nsn + 1]
  const int32_t *nrow;      // [nsn]
  const int64_t *rptr;      // [nsn + 1]
  const int32_t *rows;
  const int64_t *pptr;      // [nsn + 1] (doubles)
  const int32_t *sn_of_col; // [n]
  // transposed row lists: the entries q of `rows` with rows[q] == c are tidx[tptr[c] .. tptr[c + 1]), ascending in q (= ascending
  // source supernode) within each of the two parts tmid separates.  The SINGLE-VECTOR forward sweep writes what a supernode of the
  // bottom levels subtracts from its row rows[q] into slot q of a scratch array and the owner of column c subtracts its slots in list
  // order (sn_solve1.hpp): one summation order, no atomics.  (Top levels and the block solves push coloured updates instead.)
  const int64_t *tptr;      // [n + 1]
  const int64_t *tmid;      // [n]: tidx[tptr[c] .. tmid[c]) come from supernodes BELOW the top levels (Factor::ltop), the rest from top levels
  const int32_t *tidx;
  const int32_t *tpos;      // [entries of rows]: tidx[tpos[q]] == q: the slot of (supernode, row) q in its row's list (slots are stored in list order)
  double *panels;
  // L U variant (non-symmetric values on the symmetric pattern): the panel of s holds the FULL diagonal block and L_{rows, s};
  // upanels holds U_{s, rows}^T as an nrow x ncol column-major block at uptr[s]; piv[first[s] + k] = row of the diagonal block
  // (before pivoting) that ended up in position k
  double *upanels;
  const int64_t *uptr; // [nsn + 1]
  int32_t *piv;        // [n]
};

__device__ __forceinline__ int32_t lower_bound_i32(const int32_t *__restrict__ a, int32_t n, int32_t v)
{
  int32_t lo = 0, hi = n;
  while (lo < hi) {
    const int32_t mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}
// position of global row r in the index list [cols(t); rows(t)] of supernode t (r must be in it)
__device__ __forceinline__ int32_t row_pos(const Meta &M, int32_t t, int32_t r)
{
  const int32_t f = M.first[t], nc = M.first[t + 1] - f;
  if (r < f + nc) return r - f;
  return nc + lower_bound_i32(M.rows + M.rptr[t], M.nrow[t], r);
}
// work item -> (supernode of the level, local tile): pre[0 .. cnt] is the exclusive prefix of the per-supernode tile counts
__device__ __forceinline__ int find_item(const int32_t *__restrict__ pre, int cnt, int32_t item)
{
  int lo = 0, hi = cnt; // largest i with pre[i] <= item
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (pre[mid] <= item) lo = mid;
    else hi = mid;
  }
  return lo;
}

// ---- assembly: the lower triangle of the permuted matrix into the panels (one thread per row of A) ----------------------------
__global__ __launch_bounds__(256) void k_sn_assemble(Meta M, int64_t n, const int64_t *__restrict__ rp, const int32_t *__restrict__ ci, const double *__restrict__ va,
                                                    const int32_t *__restrict__ iperm)
{
  const int64_t io = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (io >= n) return;
  const int32_t i = iperm[io];
  for (int64_t k = rp[io]; k < rp[io + 1]; ++k) {
    const int32_t j = iperm[ci[k]];
    if (j > i) continue;
    const int32_t t = M.sn_of_col[j];
    const int32_t f = M.first[t], nc = M.first[t + 1] - f;
    M.panels[M.pptr[t] + row_pos(M, t, i) + (int64_t)(j - f) * (nc + M.nrow[t])] = va[k];
  }
}

// ---- diagonal block: Cholesky + triangular inverse in LDS --------------------------------------------------------------------
// err: first supernode (+1) whose diagonal block is not positive definite
__global__ __launch_bounds__(256) void k_sn_diag(Meta M, const int32_t *__restrict__ lev_sn, unsigned *__restrict__ err)
{
  extern __shared__ __attribute__((aligned(16))) double a[];
  __shared__ double rowbuf[SN_MAX_COLS];
  __shared__ double tmp[32 * 33];
  const int32_t s = lev_sn[blockIdx.x];
  const int32_t nc = M.first[s + 1] - M.first[s];
  const int64_t ld = nc + M.nrow[s];
  double *P = M.panels + M.pptr[s];
  const int ldl = nc | 1;
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  for (int j = ty; j < nc; j += 4)
    for (int i = tx; i < nc; i += 64) a[i + j * ldl] = i >= j ? P[i + j * ld] : 0.0;
  // Cholesky, right-looking, ONE barrier per column: the trailing update of step k works with the UNSCALED column k
  // (a_ij -= a_ik a_jk / d_k), the column is scaled one step later, when nobody reads it any more
  double rp_prev = 0.0, sq_prev = 0.0;
  for (int k = 0; k < nc; ++k) {
    __syncthreads();
    if (k > 0) {
      for (int i = k + tid; i < nc; i += 256) a[i + (k - 1) * ldl] *= rp_prev;
      if (tid == 0) a[(k - 1) + (k - 1) * ldl] = sq_prev;
    }
    double d = a[k + k * ldl];
    if (!(d > 0.0) || !(d < 1.7e308)) {
      if (tid == 0) atomicCAS(err, 0u, (unsigned)s + 1u);
      d = 1.0;
    }
    sq_prev = sqrt(d);
    rp_prev = 1.0 / sq_prev;
    const double invd = 1.0 / d;
    for (int j = k + 1 + ty; j < nc; j += 4) {
      const double ajk = a[j + k * ldl] * invd;
      for (int i = j + tx; i < nc; i += 64) a[i + j * ldl] -= a[i + k * ldl] * ajk;
    }
  }
  __syncthreads();
  if (tid == 0) a[(nc - 1) + (nc - 1) * ldl] = sq_prev;
  __syncthreads();
  // W = L^-1 in place, in 32 x 32 blocks.  (1) the diagonal blocks, all at once: row by row, row i of a block from row i of L (staged)
  // and the rows of W above it
  const int nb = (nc + 31) >> 5;
  {
    const int b = tid >> 5, j = tid & 31, b0 = b << 5; // thread = (block, column)
    const int bn = b < nb ? min(32, nc - b0) : 0;
    for (int i = 0; i < 32; ++i) {
      if (i < bn && j <= i) rowbuf[b0 + j] = a[(b0 + i) + (b0 + j) * ldl];
      __syncthreads();
      if (i < bn && j <= i) {
        double acc = (i == j) ? 1.0 : 0.0;
        for (int k = j; k < i; ++k) acc -= rowbuf[b0 + k] * a[(b0 + k) + (b0 + j) * ldl];
        a[(b0 + i) + (b0 + j) * ldl] = acc / rowbuf[b0 + i];
      }
      __syncthreads();
    }
  }
  // (2) the blocks below the diagonal, column block by column block (the blocks of L to the right are still intact), row blocks
  // downwards:  W_ib = -W_ii (sum_{k = b}^{i-1} L_ik W_kb)
  {
    const int r = tid & 31, cg = tid >> 5; // thread: row r of the 32 x 32 block, columns 4 cg .. 4 cg + 3
    for (int b = 0; b + 1 < nb; ++b) {
      const int b0 = b << 5;
      for (int ib = b + 1; ib < nb; ++ib) {
        const int i0 = ib << 5, in = min(32, nc - i0);
        double t4[4] = {0.0, 0.0, 0.0, 0.0};
        if (r < in)
          for (int k = b0; k < i0; ++k) {
            const double l = a[(i0 + r) + k * ldl];
#pragma unroll
            for (int c = 0; c < 4; ++c) t4[c] += l * ((k - b0) >= 0 && k >= b0 + 4 * cg + c ? a[k + (b0 + 4 * cg + c) * ldl] : 0.0); // W_kb is lower triangular inside block b
          }
#pragma unroll
        for (int c = 0; c < 4; ++c) tmp[r * 33 + 4 * cg + c] = t4[c];
        __syncthreads();
        double w4[4] = {0.0, 0.0, 0.0, 0.0};
        if (r < in)
          for (int k = 0; k <= r; ++k) {
            const double wik = a[(i0 + r) + (i0 + k) * ldl]; // W_ii (lower)
#pragma unroll
            for (int c = 0; c < 4; ++c) w4[c] -= wik * tmp[k * 33 + 4 * cg + c];
          }
        __syncthreads();
        if (r < in) {
#pragma unroll
          for (int c = 0; c < 4; ++c) a[(i0 + r) + (b0 + 4 * cg + c) * ldl] = w4[c];
        }
        __syncthreads();
      }
    }
  }
  for (int j = ty; j < nc; j += 4)
    for (int i = j + tx; i < nc; i += 64) P[i + j * ld] = a[i + j * ldl];
}

// ---- panel: R_s <- R_s W_s^T  (64 rows per workgroup, 16 per wavefront, all ncol <= 128 columns in registers) ---------------
__global__ __launch_bounds__(256) void k_sn_panel(Meta M, const int32_t *__restrict__ lev_sn, const int32_t *__restrict__ pre, int cnt)
{
  const int it = find_item(pre, cnt, (int32_t)blockIdx.x);
  const int32_t s = lev_sn[it];
  const int tile = (int)blockIdx.x - pre[it];
  const int32_t nc = M.first[s + 1] - M.first[s], nr = M.nrow[s];
  const int64_t ld = nc + nr;
  double *P = M.panels + M.pptr[s];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lc = lane & 15, lr = lane >> 4;
  const int row = tile * TILE + wave * 16 + lc; // A operand row of this lane
  const bool rok = row < nr;
  const int tb = (nc + 15) >> 4;
  v4d acc[SN_MAX_COLS / 16];
#pragma unroll
  for (int t = 0; t < SN_MAX_COLS / 16; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < nc; k0 += 4) {
    const int k = k0 + lr;
    const double av = (rok && k < nc) ? P[nc + row + (int64_t)k * ld] : 0.0;
#pragma unroll
    for (int t = 0; t < SN_MAX_COLS / 16; ++t) {
      if (t < tb) {
        const int c = (t << 4) + lc;                        // B[k][j = c] = W[c][k] (lower: k <= c)
        const double bv = (c < nc && k <= c) ? P[c + (int64_t)k * ld] : 0.0;
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[t], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < SN_MAX_COLS / 16; ++t) {
    if (t >= tb) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = tile * TILE + wave * 16 + lr + 4 * q, c = (t << 4) + lc;
      if (r < nr && c < nc) P[nc + r + (int64_t)c * ld] = acc[t][q];
    }
  }
}

// ---- update: lower-triangular 64 x 64 tiles of R_s R_s^T subtracted from the ancestors' panels ---------------------------------
// (base: first work item of this launch = of the colour; pre / cnt describe the whole level)
__global__ __launch_bounds__(256) void k_sn_update(Meta M, const int32_t *__restrict__ lev_sn, const int32_t *__restrict__ pre, int cnt, int base)
{
  __shared__ int32_t rowid[TILE], colid[TILE], slot_of_col[TILE], slot_t[TILE], slot_first[TILE];
  __shared__ int64_t slot_base[TILE], slot_ld[TILE];
  __shared__ int32_t rpos[TILE * TILE]; // [slot][row]
  __shared__ int nslots_s;
  const int item = (int)blockIdx.x + base;
  const int it = find_item(pre, cnt, (int32_t)item);
  const int32_t s = lev_sn[it];
  int u = item - pre[it]; // index into the lower triangle of the T x T tile grid, row-major: u = ti (ti + 1) / 2 + tj
  int ti = (int)((sqrt(8.0 * (double)u + 1.0) - 1.0) * 0.5);
  while ((ti + 1) * (ti + 2) / 2 <= u) ++ti;
  while (ti * (ti + 1) / 2 > u) --ti;
  const int tj = u - ti * (ti + 1) / 2;
  const int32_t nc = M.first[s + 1] - M.first[s], nr = M.nrow[s];
  const int64_t ld = nc + nr;
  const double *P = M.panels + M.pptr[s];
  const int32_t *R = M.rows + M.rptr[s];
  const int tid = threadIdx.x;
  if (tid < TILE) {
    const int r = ti * TILE + tid, c = tj * TILE + tid;
    rowid[tid] = r < nr ? R[r] : -1;
    colid[tid] = c < nr ? R[c] : -1;
  }
  __syncthreads();
  if (tid == 0) {
    int ns = 0;
    for (int c = 0; c < TILE; ++c) {
      if (colid[c] < 0) {
        slot_of_col[c] = -1;
        continue;
      }
      const int32_t t = M.sn_of_col[colid[c]];
      if (ns == 0 || slot_t[ns - 1] != t) {
        slot_t[ns] = t;
        const int32_t f = M.first[t];
        slot_first[ns] = f;
        slot_ld[ns] = (int64_t)(M.first[t + 1] - f) + M.nrow[t];
        slot_base[ns] = M.pptr[t];
        ++ns;
      }
      slot_of_col[c] = ns - 1;
    }
    nslots_s = ns;
  }
  __syncthreads();
  const int ns = nslots_s;
  for (int idx = tid; idx < ns * TILE; idx += 256) {
    const int sl = idx / TILE, r = idx % TILE;
    const int32_t g = rowid[r];
    rpos[idx] = (g >= 0 && g >= slot_first[sl]) ? row_pos(M, slot_t[sl], g) : -1;
  }
  // The product: K in chunks of UPD_KC columns staged through LDS (each panel element is read from global memory once per
  // workgroup; round 3's first version had every wavefront fetch its operands itself: 24 TFLOP/s), the NEXT chunk in registers while
  // the matrix cores work on the current one; wavefront w owns the 32 x 32 quadrant (w >> 1, w & 1): 2 x 2 MFMA tiles, two A and
  // two B operand reads per four MFMAs.
  __shared__ double As[UPD_KC * TILE], Bs[UPD_KC * TILE]; // [k][row]
  const int lane = tid & 63, wave = tid >> 6, lc = lane & 15, lr = lane >> 4;
  const int wi = wave >> 1, wj = wave & 1;
  const bool diag = ti == tj;
  v4d acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = v4d{0.0, 0.0, 0.0, 0.0};
  constexpr int NLD = UPD_KC * TILE / 256; // elements per thread and operand per chunk
  const int lrow = tid & 63, lk = tid >> 6; // loader mapping: 64 consecutive rows of one column per wavefront
  double pa[NLD], pb[NLD];
  auto load_chunk = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int k = k0 + lk + 4 * u;
      const int ra = ti * TILE + lrow, rb = tj * TILE + lrow;
      pa[u] = (k < nc && ra < nr) ? P[nc + ra + (int64_t)k * ld] : 0.0;
      pb[u] = (!diag && k < nc && rb < nr) ? P[nc + rb + (int64_t)k * ld] : 0.0;
    }
  };
  load_chunk(0);
  for (int k0 = 0; k0 < nc; k0 += UPD_KC) {
    __syncthreads(); // the previous chunk has been consumed (and, first trip, rpos is complete)
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      As[(lk + 4 * u) * TILE + lrow] = pa[u];
      if (!diag) Bs[(lk + 4 * u) * TILE + lrow] = pb[u];
    }
    __syncthreads();
    if (k0 + UPD_KC < nc) load_chunk(k0 + UPD_KC);
    const double *Bp = diag ? As : Bs;
#pragma unroll
    for (int kk = 0; kk < UPD_KC; kk += 4) {
      const double a0 = As[(kk + lr) * TILE + 32 * wi + lc], a1 = As[(kk + lr) * TILE + 32 * wi + 16 + lc];
      const double b0 = Bp[(kk + lr) * TILE + 32 * wj + lc], b1 = Bp[(kk + lr) * TILE + 32 * wj + 16 + lc];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int c = 32 * wj + 16 * b + lc;
    const int sl = slot_of_col[c];
    if (sl < 0) continue;
    const int32_t gc = colid[c];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 32 * wi + 16 * a + lr + 4 * q;
        const int32_t gr = rowid[r];
        if (gr < gc) continue; // (also gr == -1) lower triangle only
        const int32_t rp = rpos[sl * TILE + r];
        M.panels[slot_base[sl] + rp + (int64_t)(gc - slot_first[sl]) * slot_ld[sl]] -= acc[a][b][q]; // (no other workgroup of this launch touches the entry: colours)
      }
  }
}

// ---- L U variant -------------------------------------------------------------------------------------------------------------------
// Non-symmetric values on a symmetric pattern (the DG convection-diffusion operator; `type = umfpack`): same supernodes, same tree.
// Pivoting: threshold partial pivoting INSIDE the diagonal block of a supernode (UMFPACK's default: the diagonal entry is kept when
// |a_kk| >= 0.1 max_i |a_ik|, else the largest entry of the column inside the block becomes the pivot); rows are never exchanged
// between supernodes, so the structure stays static.  The row exchange is applied to the right-hand side when the forward sweep
// reaches the supernode (not retroactively to the columns on the left), which is the same factorisation P_s ... P_1 A = L U.
constexpr double LU_PIVOT_THRESHOLD = 0.1;

__global__ __launch_bounds__(256) void k_sn_assemble_lu(Meta M, int64_t n, const int64_t *__restrict__ rp, const int32_t *__restrict__ ci, const double *__restrict__ va,
                                                       const int32_t *__restrict__ iperm)
{
  const int64_t io = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (io >= n) return;
  const int32_t i = iperm[io];
  for (int64_t k = rp[io]; k < rp[io + 1]; ++k) {
    const int32_t j = iperm[ci[k]];
    if (j <= i) {
      const int32_t t = M.sn_of_col[j];
      const int32_t f = M.first[t], nc = M.first[t + 1] - f;
      M.panels[M.pptr[t] + row_pos(M, t, i) + (int64_t)(j - f) * (nc + M.nrow[t])] = va[k];
    } else {
      const int32_t t = M.sn_of_col[i];
      const int32_t f = M.first[t], nc = M.first[t + 1] - f, nr = M.nrow[t];
      if (j < f + nc) M.panels[M.pptr[t] + (i - f) + (int64_t)(j - f) * (nc + nr)] = va[k];
      else M.upanels[M.uptr[t] + lower_bound_i32(M.rows + M.rptr[t], nr, j) + (int64_t)(i - f) * nr] = va[k];
    }
  }
}

// diagonal block: P D = L U with threshold partial pivoting, then both triangular inverses in place (strict lower: L^-1 with its
// unit diagonal implied; upper incl. diagonal: U^-1)
__global__ __launch_bounds__(256) void k_sn_lu_diag(Meta M, const int32_t *__restrict__ lev_sn, unsigned *__restrict__ err, double tiny)
{
  extern __shared__ __attribute__((aligned(16))) double a[];
  __shared__ double rowbuf[SN_MAX_COLS];
  __shared__ int32_t prow[SN_MAX_COLS];
  __shared__ int pivot_row;
  const int32_t s = lev_sn[blockIdx.x];
  const int32_t f = M.first[s], nc = M.first[s + 1] - f;
  const int64_t ld = nc + M.nrow[s];
  double *P = M.panels + M.pptr[s];
  const int ldl = nc | 1;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < nc * nc; idx += 256) {
    const int i = idx % nc, j = idx / nc;
    a[i + j * ldl] = P[i + j * ld];
  }
  if (tid < nc) prow[tid] = tid;
  __syncthreads();
  for (int k = 0; k < nc; ++k) {
    if (tid == 0) {
      double amax = 0.0;
      int imax = k;
      for (int i = k; i < nc; ++i) {
        const double v = fabs(a[i + k * ldl]);
        if (v > amax) {
          amax = v;
          imax = i;
        }
      }
      const double dg = fabs(a[k + k * ldl]);
      int p = (dg > 0.0 && dg >= LU_PIVOT_THRESHOLD * amax) ? k : imax;
      if (!(amax < 1.7e308)) { // NaN / Inf
        atomicCAS(err, 0u, (unsigned)s + 1u);
        a[k + k * ldl] = 1.0;
        p = k;
      } else if (!(amax > 0.0)) { // the whole column of the block vanishes: static perturbation (sqrt(eps) max|a_ij|, as SuperLU_DIST
        a[k + k * ldl] = tiny;    // does for the pivots its static order cannot reach); counted, repaired by the iterative refinement
        atomicAdd(err + 2, 1u);
        p = k;
      }
      pivot_row = p;
    }
    __syncthreads();
    const int p = pivot_row;
    if (p != k) {
      for (int j = tid; j < nc; j += 256) {
        const double tmp = a[k + j * ldl];
        a[k + j * ldl] = a[p + j * ldl];
        a[p + j * ldl] = tmp;
      }
      if (tid == 0) {
        const int32_t t2 = prow[k];
        prow[k] = prow[p];
        prow[p] = t2;
      }
    }
    __syncthreads();
    const double piv = a[k + k * ldl];
    for (int i = k + 1 + tid; i < nc; i += 256) a[i + k * ldl] /= piv;
    __syncthreads();
    const int m = nc - k - 1;
    for (int idx = tid; idx < m * m; idx += 256) {
      const int i = k + 1 + idx % m, j = k + 1 + idx / m;
      a[i + j * ldl] -= a[i + k * ldl] * a[k + j * ldl];
    }
    __syncthreads();
  }
  // strict lower: W = L^-1 (unit diagonal), row by row
  for (int i = 1; i < nc; ++i) {
    if (tid < i) rowbuf[tid] = a[i + tid * ldl];
    __syncthreads();
    if (tid < i) {
      const int j = tid;
      double acc = -rowbuf[j];
      for (int k = j + 1; k < i; ++k) acc -= rowbuf[k] * a[k + j * ldl];
      a[i + j * ldl] = acc;
    }
    __syncthreads();
  }
  // upper incl. diagonal: V = U^-1, rows from the bottom
  for (int i = nc - 1; i >= 0; --i) {
    if (tid >= i && tid < nc) rowbuf[tid] = a[i + tid * ldl];
    __syncthreads();
    if (tid >= i && tid < nc) {
      const int j = tid;
      double acc = (i == j) ? 1.0 : 0.0;
      for (int k = i + 1; k <= j; ++k) acc -= rowbuf[k] * a[k + j * ldl];
      a[i + j * ldl] = acc / rowbuf[i];
    }
    __syncthreads();
  }
  for (int idx = tid; idx < nc * nc; idx += 256) {
    const int i = idx % nc, j = idx / nc;
    P[i + j * ld] = a[i + j * ldl];
  }
  if (tid < nc) M.piv[f + tid] = prow[tid];
}

// rows below: L_{rows,s} <- A_{rows,s} U_ss^-1  and  U_{s,rows}^T <- (A_{s,rows}^T with the pivot order applied to its columns) L_ss^-T
__global__ __launch_bounds__(256) void k_sn_lu_panel(Meta M, const int32_t *__restrict__ lev_sn, const int32_t *__restrict__ pre, int cnt)
{
  const int it = find_item(pre, cnt, (int32_t)blockIdx.x);
  const int32_t s = lev_sn[it];
  const int tile = (int)blockIdx.x - pre[it];
  const int32_t f = M.first[s], nc = M.first[s + 1] - f, nr = M.nrow[s];
  const int64_t ld = nc + nr;
  double *P = M.panels + M.pptr[s];
  double *UT = M.upanels + M.uptr[s];
  const int32_t *piv = M.piv + f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lc = lane & 15, lr = lane >> 4;
  const int row = tile * TILE + wave * 16 + lc;
  const bool rok = row < nr;
  const int tb = (nc + 15) >> 4;
  v4d accL[SN_MAX_COLS / 16], accU[SN_MAX_COLS / 16];
#pragma unroll
  for (int t = 0; t < SN_MAX_COLS / 16; ++t) accL[t] = accU[t] = v4d{0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < nc; k0 += 4) {
    const int k = k0 + lr;
    const bool kok = k < nc;
    const double al = (rok && kok) ? P[nc + row + (int64_t)k * ld] : 0.0;
    const double au = (rok && kok) ? UT[row + (int64_t)piv[k] * nr] : 0.0;
#pragma unroll
    for (int t = 0; t < SN_MAX_COLS / 16; ++t) {
      if (t < tb) {
        const int c = (t << 4) + lc;
        const double bl = (c < nc && kok && k <= c) ? P[k + (int64_t)c * ld] : 0.0;                       // U^-1[k][c]
        const double bu = (c < nc && kok) ? (k < c ? P[c + (int64_t)k * ld] : (k == c ? 1.0 : 0.0)) : 0.0; // L^-1[c][k]
        accL[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(al, bl, accL[t], 0, 0, 0);
        accU[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(au, bu, accU[t], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < SN_MAX_COLS / 16; ++t) {
    if (t >= tb) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = tile * TILE + wave * 16 + lr + 4 * q, c = (t << 4) + lc;
      if (r < nr && c < nc) {
        P[nc + r + (int64_t)c * ld] = accL[t][q];
        UT[r + (int64_t)c * nr] = accU[t][q];
      }
    }
  }
}

// update: ALL 64 x 64 tiles of L_{rows,s} U_{s,rows} subtracted from the ancestors (lower part and diagonal blocks: panel of the
// column's owner; strictly upper part outside a diagonal block: U^T block of the row's owner)
__global__ __launch_bounds__(256) void k_sn_lu_update(Meta M, const int32_t *__restrict__ lev_sn, const int32_t *__restrict__ pre, int cnt, int base)
{
  __shared__ int32_t rowid[TILE], colid[TILE], cslot[TILE], rslot[TILE], cs_t[TILE], rs_t[TILE];
  __shared__ int32_t posc[TILE * TILE]; // [column slot][row]: position of the row in the column owner's index list
  __shared__ int32_t posr[TILE * TILE]; // [row slot][column]: position of the column in the row owner's index list
  __shared__ int ncs_s, nrs_s;
  const int item = (int)blockIdx.x + base;
  const int it = find_item(pre, cnt, (int32_t)item);
  const int32_t s = lev_sn[it];
  const int u = item - pre[it];
  const int32_t nc = M.first[s + 1] - M.first[s], nr = M.nrow[s];
  const int T = (nr + TILE - 1) / TILE;
  const int ti = u / T, tj = u - ti * T;
  const int64_t ld = nc + nr;
  const double *P = M.panels + M.pptr[s];
  const double *UT = M.upanels + M.uptr[s];
  const int32_t *R = M.rows + M.rptr[s];
  const int tid = threadIdx.x;
  if (tid < TILE) {
    const int r = ti * TILE + tid, c = tj * TILE + tid;
    rowid[tid] = r < nr ? R[r] : -1;
    colid[tid] = c < nr ? R[c] : -1;
  }
  __syncthreads();
  if (tid == 0) {
    int ns = 0;
    for (int c = 0; c < TILE; ++c) {
      if (colid[c] < 0) {
        cslot[c] = -1;
        continue;
      }
      const int32_t t = M.sn_of_col[colid[c]];
      if (ns == 0 || cs_t[ns - 1] != t) cs_t[ns++] = t;
      cslot[c] = ns - 1;
    }
    ncs_s = ns;
  }
  if (tid == 64) {
    int ns = 0;
    for (int r = 0; r < TILE; ++r) {
      if (rowid[r] < 0) {
        rslot[r] = -1;
        continue;
      }
      const int32_t t = M.sn_of_col[rowid[r]];
      if (ns == 0 || rs_t[ns - 1] != t) rs_t[ns++] = t;
      rslot[r] = ns - 1;
    }
    nrs_s = ns;
  }
  __syncthreads();
  const int ncs = ncs_s, nrs = nrs_s;
  for (int idx = tid; idx < ncs * TILE; idx += 256) {
    const int sl = idx / TILE, r = idx % TILE;
    const int32_t g = rowid[r];
    posc[idx] = (g >= 0 && g >= M.first[cs_t[sl]]) ? row_pos(M, cs_t[sl], g) : -1;
  }
  for (int idx = tid; idx < nrs * TILE; idx += 256) {
    const int sl = idx / TILE, c = idx % TILE;
    const int32_t g = colid[c];
    posr[idx] = (g >= 0 && g >= M.first[rs_t[sl]]) ? row_pos(M, rs_t[sl], g) : -1;
  }
  const int lane = tid & 63, wave = tid >> 6, lc = lane & 15, lr = lane >> 4;
  v4d acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
  const int arow = ti * TILE + wave * 16 + lc;
  const bool aok = arow < nr;
  for (int k0 = 0; k0 < nc; k0 += 4) {
    const int k = k0 + lr;
    const bool kok = k < nc;
    const double av = (aok && kok) ? P[nc + arow + (int64_t)k * ld] : 0.0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int brow = tj * TILE + (t << 4) + lc;
      const double bv = (brow < nr && kok) ? UT[brow + (int64_t)k * nr] : 0.0;
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[t], 0, 0, 0);
    }
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = (t << 4) + lc;
    const int32_t gc = colid[c];
    if (gc < 0) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = wave * 16 + lr + 4 * q;
      const int32_t gr = rowid[r];
      if (gr < 0) continue;
      if (gr >= gc) { // lower part: panel of the owner of column gc
        const int32_t tt = cs_t[cslot[c]];
        const int32_t f = M.first[tt];
        M.panels[M.pptr[tt] + posc[cslot[c] * TILE + r] + (int64_t)(gc - f) * ((int64_t)(M.first[tt + 1] - f) + M.nrow[tt])] -= acc[t][q];
      } else { // upper part: owner of row gr
        const int32_t tt = rs_t[rslot[r]];
        const int32_t f = M.first[tt], ncc = M.first[tt + 1] - f, nrr = M.nrow[tt];
        const int32_t pc = posr[rslot[r] * TILE + c];
        if (pc < ncc) M.panels[M.pptr[tt] + (gr - f) + (int64_t)pc * (ncc + nrr)] -= acc[t][q]; // inside the diagonal block
        else M.upanels[M.uptr[tt] + (pc - ncc) + (int64_t)(gr - f) * nrr] -= acc[t][q];
      }
    }
  }
}

// ---- solves ----------------------------------------------------------------------------------------------------------------------
// All four are small dense products on the FP64 matrix cores: m <= 48 right-hand sides = up to three 16-column tiles, padded with
// zeros in LDS (mpad = 16 ceil(m / 16)); a wavefront owns 16-row strips of the result.
constexpr int SOLVE_MT = 3;
constexpr int SOLVE_UNROLL = 8; // k-steps (of 4) whose global operand loads are issued together
// Y_s = W_s B_s, in place in the work block (row-major, leading dimension ldb)
template <bool LU>
__global__ __launch_bounds__(256) void k_sn_fwd_diag(Meta M, const int32_t *__restrict__ lev_sn, int m, double *__restrict__ B, int64_t ldb)
{
  extern __shared__ __attribute__((aligned(16))) double bs[]; // nc x mpad
  const int32_t s = lev_sn[blockIdx.x];
  const int32_t f = M.first[s], nc = M.first[s + 1] - f;
  const int64_t ld = nc + M.nrow[s];
  const double *W = M.panels + M.pptr[s];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lc = lane & 15, lr = lane >> 4;
  const int mt = (m + 15) >> 4, mpad = mt << 4;
  for (int idx = tid; idx < nc * mpad; idx += 256) {
    const int k = idx / mpad, c = idx - k * mpad;
    const int ksrc = LU ? M.piv[f + k] : k; // the row exchanges of the diagonal block, applied to the right-hand side here
    bs[idx] = c < m ? B[(int64_t)(f + ksrc) * ldb + c] : 0.0;
  }
  __syncthreads();
  const int ns = (nc + 15) >> 4;
  for (int si = wave; si < ns; si += 4) {
    v4d acc[SOLVE_MT];
#pragma unroll
    for (int t = 0; t < SOLVE_MT; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
    const int row = (si << 4) + lc;
    const int kend = min(nc, (si + 1) << 4);
    for (int k0 = 0; k0 < kend; k0 += 4 * SOLVE_UNROLL) { // SOLVE_UNROLL operand loads in flight: these kernels are latency-, not work-bound
      double av[SOLVE_UNROLL];
#pragma unroll
      for (int u = 0; u < SOLVE_UNROLL; ++u) {
        const int k = k0 + 4 * u + lr;
        av[u] = LU ? ((row < nc && k < row) ? W[row + (int64_t)k * ld] : ((row < nc && k == row) ? 1.0 : 0.0))
                   : ((row < nc && k <= row) ? W[row + (int64_t)k * ld] : 0.0);
      }
#pragma unroll
      for (int u = 0; u < SOLVE_UNROLL; ++u) {
        const int k = k0 + 4 * u + lr;
#pragma unroll
        for (int t = 0; t < SOLVE_MT; ++t)
          if (t < mt) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], k < nc ? bs[k * mpad + (t << 4) + lc] : 0.0, acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < SOLVE_MT; ++t) {
      if (t >= mt) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = (si << 4) + lr + 4 * q, c = (t << 4) + lc;
        if (r < nc && c < m) B[(int64_t)(f + r) * ldb + c] = acc[t][q];
      }
    }
  }
}
// B[rows] -= R_s Y_s, one workgroup per 64 rows of R_s; launched colour by colour (Factor::lev_phase_ptr): no other workgroup of a
// launch touches the same rows, so the subtraction is a plain read-modify-write and every row receives its updates in one order
__global__ __launch_bounds__(256) void k_sn_fwd_update(Meta M, const int32_t *__restrict__ lev_sn, const int32_t *__restrict__ pre, int cnt, int m,
                                                      double *__restrict__ B, int64_t ldb, int base)
{
  extern __shared__ __attribute__((aligned(16))) double ys[]; // nc x mpad
  const int item = (int)blockIdx.x + base; // (base: first row tile of this launch = of the colour)
  const int it = find_item(pre, cnt, (int32_t)item);
  const int32_t s = lev_sn[it];
  const int tile = item - pre[it];
  const int32_t *R = M.rows + M.rptr[s];
  const int32_t f = M.first[s], nc = M.first[s + 1] - f, nr = M.nrow[s];
  const int64_t ld = nc + nr;
  const double *P = M.panels + M.pptr[s];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lc = lane & 15, lr = lane >> 4;
  const int mt = (m + 15) >> 4, mpad = mt << 4;
  for (int idx = tid; idx < nc * mpad; idx += 256) {
    const int k = idx / mpad, c = idx - k * mpad;
    ys[idx] = c < m ? B[(int64_t)(f + k) * ldb + c] : 0.0;
  }
  __syncthreads();
  v4d acc[SOLVE_MT];
#pragma unroll
  for (int t = 0; t < SOLVE_MT; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
  const int row = tile * TILE + (wave << 4) + lc;
  for (int k0 = 0; k0 < nc; k0 += 4 * SOLVE_UNROLL) {
    double av[SOLVE_UNROLL];
#pragma unroll
    for (int u = 0; u < SOLVE_UNROLL; ++u) {
      const int k = k0 + 4 * u + lr;
      av[u] = (row < nr && k < nc) ? P[nc + row + (int64_t)k * ld] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < SOLVE_UNROLL; ++u) {
      const int k = k0 + 4 * u + lr;
#pragma unroll
      for (int t = 0; t < SOLVE_MT; ++t)
        if (t < mt) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], k < nc ? ys[k * mpad + (t << 4) + lc] : 0.0, acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < SOLVE_MT; ++t) {
    if (t >= mt) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = tile * TILE + (wave << 4) + lr + 4 * q, c = (t << 4) + lc;
      if (r < nr && c < m) B[(int64_t)R[r] * ldb + c] -= acc[t][q];
    }
  }
}
// one 64-row tile of R_s^T X_rows into the accumulators of the calling wavefront: strips si = wave, wave + 4 of the nc result rows
// (P + nc = first row of the block of rows below: R_s in the panel, leading dimension ld; or U_{s,rows}^T with P = block - nc)
__device__ __forceinline__ void bwd_tile_mfma(const double *__restrict__ P, int32_t nc, int64_t ld, int r0, int rn, const double *__restrict__ xr, int mpad, int mt, int wave,
                                               int lc, int lr, v4d (&acc)[2][SOLVE_MT])
{
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int si = wave + 4 * h;
    const int kcol = (si << 4) + lc; // result row = column of the panel
    if ((si << 4) >= nc) continue;
    for (int rr = 0; rr < rn; rr += 4 * SOLVE_UNROLL) {
      double av[SOLVE_UNROLL];
#pragma unroll
      for (int u = 0; u < SOLVE_UNROLL; ++u) {
        const int r = rr + 4 * u + lr;
        av[u] = (kcol < nc && r < rn) ? P[nc + r0 + r + (int64_t)kcol * ld] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < SOLVE_UNROLL; ++u) {
        const int r = rr + 4 * u + lr;
#pragma unroll
        for (int t = 0; t < SOLVE_MT; ++t)
          if (t < mt) acc[h][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], r < rn ? xr[r * mpad + (t << 4) + lc] : 0.0, acc[h][t], 0, 0, 0);
      }
    }
  }
}
// partial[item][k * m + c] = sum over the rows r of the tile of R_s[r][k] X[rows[r]][c]   (supernodes with more than BWD_SMALL tiles)
template <bool LU>
__global__ __launch_bounds__(256) void k_sn_bwd_partial(Meta M, const int32_t *__restrict__ big_sn, const int32_t *__restrict__ pre, int cnt, int m,
                                                       const double *__restrict__ X, int64_t ldb, double *__restrict__ partial)
{
  extern __shared__ __attribute__((aligned(16))) double xr[]; // 64 x mpad
  const int it = find_item(pre, cnt, (int32_t)blockIdx.x);
  const int32_t s = big_sn[it];
  const int tile = (int)blockIdx.x - pre[it];
  const int32_t nc = M.first[s + 1] - M.first[s], nr = M.nrow[s];
  const int64_t ld = LU ? (int64_t)nr : (int64_t)nc + nr;                              // L U: the block U_{s,rows}^T (nrow x ncol)
  const double *P = LU ? M.upanels + M.uptr[s] - nc : M.panels + M.pptr[s];
  const int32_t *R = M.rows + M.rptr[s];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lc = lane & 15, lr = lane >> 4;
  const int mt = (m + 15) >> 4, mpad = mt << 4;
  v4d acc[2][SOLVE_MT];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int t = 0; t < SOLVE_MT; ++t) acc[h][t] = v4d{0.0, 0.0, 0.0, 0.0};
  for (int sub = 0; sub < BWD_ROWS / TILE; ++sub) {
    const int r0 = tile * BWD_ROWS + sub * TILE, rn = min(TILE, nr - r0);
    if (rn <= 0) break;
    __syncthreads();
    for (int idx = tid; idx < TILE * mpad; idx += 256) {
      const int r = idx / mpad, c = idx - r * mpad;
      xr[idx] = (r < rn && c < m) ? X[(int64_t)R[r0 + r] * ldb + c] : 0.0;
    }
    __syncthreads();
    bwd_tile_mfma(P, nc, ld, r0, rn, xr, mpad, mt, wave, lc, lr, acc);
  }
  double *out = partial + (int64_t)blockIdx.x * SN_MAX_COLS * m;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int t = 0; t < SOLVE_MT; ++t) {
      if (t >= mt) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = ((wave + 4 * h) << 4) + lr + 4 * q, c = (t << 4) + lc;
        if (k < nc && c < m) out[k * m + c] = acc[h][t][q];
      }
    }
}
// X_s = W_s^T (Y_s - R_s^T X_rows); big_index[s] >= 0: position of s in the level's list of big supernodes (partials), else -1
template <bool LU>
__global__ __launch_bounds__(256) void k_sn_bwd_diag(Meta M, const int32_t *__restrict__ lev_sn, const int32_t *__restrict__ big_index, const int32_t *__restrict__ pre_big,
                                                    const double *__restrict__ partial, int m, double *__restrict__ B, int64_t ldb)
{
  extern __shared__ __attribute__((aligned(16))) double sh[]; // t: nc x mpad, then xr: 64 x mpad
  const int32_t s = lev_sn[blockIdx.x];
  const int32_t f = M.first[s], nc = M.first[s + 1] - f, nr = M.nrow[s];
  const int64_t ld = nc + nr;
  const double *P = M.panels + M.pptr[s];
  const int32_t *R = M.rows + M.rptr[s];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lc = lane & 15, lr = lane >> 4;
  const int mt = (m + 15) >> 4, mpad = mt << 4;
  double *t = sh, *xr = sh + (int64_t)nc * mpad;
  for (int idx = tid; idx < nc * mpad; idx += 256) {
    const int k = idx / mpad, c = idx - k * mpad;
    t[idx] = c < m ? B[(int64_t)(f + k) * ldb + c] : 0.0;
  }
  const int ntile = (nr + TILE - 1) / TILE;
  const int bi = big_index[blockIdx.x];
  if (bi >= 0) {
    for (int idx = tid; idx < nc * mpad; idx += 256) { // (same idx -> thread mapping as the load above)
      const int k = idx / mpad, c = idx - k * mpad;
      if (c >= m) continue;
      double acc = t[idx];
      const double *pp = partial + (int64_t)pre_big[bi] * SN_MAX_COLS * m + k * m + c;
      const int npart = (nr + BWD_ROWS - 1) / BWD_ROWS;
      for (int tl = 0; tl < npart; ++tl) acc -= pp[(int64_t)tl * SN_MAX_COLS * m];
      t[idx] = acc;
    }
  } else if (ntile > 0) {
    v4d acc[2][SOLVE_MT];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int tt = 0; tt < SOLVE_MT; ++tt) acc[h][tt] = v4d{0.0, 0.0, 0.0, 0.0};
    for (int tl = 0; tl < ntile; ++tl) {
      const int r0 = tl * TILE, rn = min(TILE, nr - r0);
      __syncthreads();
      for (int idx = tid; idx < TILE * mpad; idx += 256) {
        const int r = idx / mpad, c = idx - r * mpad;
        xr[idx] = (r < rn && c < m) ? B[(int64_t)R[r0 + r] * ldb + c] : 0.0;
      }
      __syncthreads();
      if (LU) bwd_tile_mfma(M.upanels + M.uptr[s] - nc, nc, (int64_t)nr, r0, rn, xr, mpad, mt, wave, lc, lr, acc);
      else bwd_tile_mfma(P, nc, ld, r0, rn, xr, mpad, mt, wave, lc, lr, acc);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int tt = 0; tt < SOLVE_MT; ++tt) {
        if (tt >= mt) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k = ((wave + 4 * h) << 4) + lr + 4 * q, c = (tt << 4) + lc;
          if (k < nc) t[k * mpad + c] -= acc[h][tt][q]; // (every (k, c) belongs to exactly one lane)
        }
      }
  }
  __syncthreads();
  // X_s = W^T t (Cholesky: A[i][k] = W[k][i], k >= i) resp. U_ss^-1 t (L U: A[i][k] = U^-1[i][k], k >= i);  B[k][c] = t[k][c]
  const int ns = (nc + 15) >> 4;
  for (int si = wave; si < ns; si += 4) {
    v4d acc[SOLVE_MT];
#pragma unroll
    for (int tt = 0; tt < SOLVE_MT; ++tt) acc[tt] = v4d{0.0, 0.0, 0.0, 0.0};
    const int i = (si << 4) + lc;
    for (int k0 = si << 4; k0 < nc; k0 += 4 * SOLVE_UNROLL) {
      double av[SOLVE_UNROLL];
#pragma unroll
      for (int u = 0; u < SOLVE_UNROLL; ++u) {
        const int k = k0 + 4 * u + lr;
        av[u] = (i < nc && k < nc && k >= i) ? (LU ? P[i + (int64_t)k * ld] /* U^-1[i][k] */ : P[k + (int64_t)i * ld]) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < SOLVE_UNROLL; ++u) {
        const int k = k0 + 4 * u + lr;
#pragma unroll
        for (int tt = 0; tt < SOLVE_MT; ++tt)
          if (tt < mt) acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], k < nc ? t[k * mpad + (tt << 4) + lc] : 0.0, acc[tt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int tt = 0; tt < SOLVE_MT; ++tt) {
      if (tt >= mt) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = (si << 4) + lr + 4 * q, c = (tt << 4) + lc;
        if (r < nc && c < m) B[(int64_t)(f + r) * ldb + c] = acc[tt][q];
      }
    }
  }
}

// ---- ONE right-hand side (the Schwarz apply inside a Krylov loop): sn_solve1.hpp ---------------------------------------------------
} // namespace sn
#include "sn_solve1.hpp"
namespace sn {

// chains of the top levels, host side (sn_solve1.hpp: "CHAINS")
struct ChainHost {
  std::vector<int32_t> chain_of;                                                     // [nsn]: chain of a top supernode, -1 below the top levels
  std::vector<int32_t> first_sn, nlinks, col0, ncol, last_sn, nE, clevel, colour, block; // [nchain]
  std::vector<int64_t> woff, eoff;                                                   // [nchain] offsets (doubles) of the n x n triangle / the n_E x n block
  int32_t nclev = 0, max_vec = 0, max_links = 0;
  int64_t wtot = 0, etot = 0;
  // setup work lists: links of all chains (scatter) and the (chain, i) pairs of every distance d (inversion)
  std::vector<int32_t> link_sn, link_chain, link_pre, inv_chain, inv_i, inv_ptr;
  int nchain() const { return (int)first_sn.size(); }
};

// ---- host driver -------------------------------------------------------------------------------------------------------------------
struct Factor {
  int64_t n = 0, entries = 0;
  int nblocks = 1;
  int32_t nsn = 0, nlev = 0;
  double flops = 0.0;
  Meta M{};
  std::vector<int32_t> h_perm;           // perm[new] = old (global)
  std::vector<int32_t> lev_ptr;          // [nlev + 1] into lev_sn
  std::vector<int32_t> lev_big_ptr;      // [nlev + 1] into big_sn
  std::vector<int32_t> lev_maxnc;        // widest supernode of the level
  std::vector<int32_t> lev_maxnr;        // longest row list of the level
  // Colours (deterministic updates): supernodes of one level whose row lists intersect would subtract from the same ancestor entries.
  // They get different colours; the level's list in lev_sn is sorted by colour and the update kernels run colour by colour, so every
  // panel entry receives its contributions in ONE order (level, colour) whatever the hardware does -- no atomics.
  std::vector<int32_t> lev_phase_ptr;    // [nlev + 1] into phase_k
  std::vector<int32_t> phase_k;          // first position (relative to lev_ptr[l]) of every colour of every level, plus the level's end
  std::vector<int32_t> h_preU, h_preUF, h_preT; // host copies of the tile prefixes (launch bounds of a colour)
  std::vector<int32_t> h_colour;         // colour of every supernode
  std::vector<int32_t> h_first;          // host copy of `first`
  // device
  int32_t *d_first = nullptr, *d_nrow = nullptr, *d_rows = nullptr, *d_sn_of_col = nullptr, *d_iperm = nullptr, *d_perm = nullptr;
  int64_t *d_rptr = nullptr, *d_pptr = nullptr;
  double *d_panels = nullptr;
  int32_t *d_lev_sn = nullptr;   // supernodes sorted by level
  int32_t *d_preT = nullptr;     // per level: exclusive prefix of the row-tile counts (lev_ptr[l] + l .. : cnt + 1 entries)
  int32_t *d_preU = nullptr;     // the same for the update tiles T (T + 1) / 2
  int32_t *d_big_sn = nullptr, *d_big_index = nullptr, *d_preB = nullptr; // supernodes with more than BWD_SMALL row tiles, per level
  std::vector<int32_t> h_tilesT, h_tilesU, h_tilesB; // totals per level
  unsigned *d_err = nullptr;
  bool lu = false;               // L U variant
  double *d_upanels = nullptr;
  int64_t *d_uptr = nullptr;
  int32_t *d_piv = nullptr;
  int64_t uentries = 0;
  std::vector<int32_t> h_tilesUF; // L U: all T x T update tiles per level
  int32_t *d_preUF = nullptr;
  double *d_partial = nullptr;
  int64_t partial_cap = 0; // doubles
  // single-vector solves: the persistent kernel for the top levels (sn_solve1.hpp)
  std::vector<int32_t> sn_block;  // block of every supernode
  int32_t ltop = 0, ntop = 0;     // tree levels ltop .. nlev - 1 are walked by k_sn_top1 (ntop = 0: level kernels only)
  TopPlan top{};
  int32_t *d_top_ints = nullptr;  // all integer arrays of the plan in one allocation
  double *d_top_partial = nullptr;
  TopSync *d_top_sync = nullptr;
  unsigned long long *d_top_flags = nullptr, *d_top_stamps = nullptr; // stamps: diagnostics (DDM_SN_TOP_STAMPS)
  int top_grid = 0, top_spread = 0, chain_grid = 0;
  // CHAINS of the top levels (sn_solve1.hpp): a separator wider than SN_MAX_COLS is a chain of links s -> s + 1 = parent(s), each the
  // only child of the next.  For the single-vector solves a chain is ONE dense unit with an explicitly inverted triangle.
  ChainHost ch;
  ChainDev chd{};
  ChainPlan chp{};
  const int32_t *ch_link_sn = nullptr, *ch_link_chain = nullptr, *ch_link_pre = nullptr, *ch_inv_chain = nullptr, *ch_inv_i = nullptr;
  int32_t *d_chain_ints = nullptr;
  int64_t *d_chain_offs = nullptr;
  double *d_chain_w = nullptr, *d_chain_e = nullptr, *d_chain_v = nullptr, *d_chain_u = nullptr; // inverse triangles / blocks of the external rows (L; L U: also U^T)
  bool chains_ready = false;
  int64_t *d_tptr = nullptr, *d_tmid = nullptr;  // transposed row lists (Meta::tptr / tmid / tidx)
  int32_t *d_tidx = nullptr, *d_tpos = nullptr;
  double *d_contrib = nullptr; // slots of the forward sweep: one per entry of `rows` and right-hand side
  int64_t contrib_cap = 0, nrows_total = 0;
  int64_t max_big_tiles = 0;
  void release()
  {
    if (d_top_stamps) { // diagnostics: barrier log of the LAST launch of the persistent kernel
      std::vector<unsigned long long> h(4000);
      if (hipMemcpy(h.data(), d_top_stamps, 8 * h.size(), hipMemcpyDeviceToHost) == hipSuccess) {
        std::fprintf(stderr, "[ddm] k_sn_top1 barrier log (us since the first barrier; work = arrival - previous release, wait = release - arrival):\n");
        const unsigned long long t0 = h[3];
        for (int c = 1; c < 2000 && h[2 * c + 1]; ++c)
          std::fprintf(stderr, "  barrier %3d: arrive %8.2f release %8.2f  work %6.2f wait %6.2f\n", c, (double)(h[2 * c] - t0) / 100.0, (double)(h[2 * c + 1] - t0) / 100.0,
                       c > 1 ? (double)(h[2 * c] - h[2 * c - 1]) / 100.0 : 0.0, (double)(h[2 * c + 1] - h[2 * c]) / 100.0);
      }
      (void)hipFree(d_top_stamps);
      d_top_stamps = nullptr;
    }
    for (void *p : {(void *)d_first, (void *)d_nrow, (void *)d_rows, (void *)d_sn_of_col, (void *)d_iperm, (void *)d_perm, (void *)d_rptr, (void *)d_pptr, (void *)d_panels,
                    (void *)d_lev_sn, (void *)d_preT, (void *)d_preU, (void *)d_big_sn, (void *)d_big_index, (void *)d_preB, (void *)d_err, (void *)d_partial,
                    (void *)d_upanels, (void *)d_uptr, (void *)d_piv, (void *)d_preUF, (void *)d_tptr, (void *)d_tmid, (void *)d_tidx, (void *)d_tpos, (void *)d_chain_ints, (void *)d_chain_offs, (void *)d_chain_w, (void *)d_chain_e, (void *)d_chain_v, (void *)d_chain_u, (void *)d_contrib, (void *)d_top_ints, (void *)d_top_partial, (void *)d_top_sync, (void *)d_top_flags})
      if (p) (void)hipFree(p);
    d_first = d_nrow = d_rows = d_sn_of_col = d_iperm = d_perm = d_lev_sn = d_preT = d_preU = d_big_sn = d_big_index = d_preB = nullptr;
    d_rptr = d_pptr = nullptr;
    d_panels = d_partial = d_upanels = d_contrib = nullptr;
    d_tptr = d_tmid = nullptr;
    d_tidx = d_tpos = nullptr;
    d_chain_ints = nullptr;
    d_chain_offs = nullptr;
    d_chain_w = d_chain_e = d_chain_v = d_chain_u = nullptr;
    chains_ready = false;
    d_top_ints = nullptr;
    d_top_partial = nullptr;
    d_top_sync = nullptr;
    d_top_flags = nullptr;
    top = TopPlan{};
    ntop = 0;
    d_uptr = nullptr;
    d_piv = d_preUF = nullptr;
    d_err = nullptr;
  }
  ~Factor() { release(); }
};

template <class T>
static inline bool up(const std::vector<T> &h, T **d)
{
  if (hipMalloc((void **)d, sizeof(T) * std::max<size_t>(h.size(), 1)) != hipSuccess) return false;
  return h.empty() || hipMemcpy(*d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice) == hipSuccess;
}

// Top levels of the single-vector solve (sn_solve1.hpp): those from the first level on which every later level has at most
// `top_max` supernodes (the separator chains: one supernode per block and level); fewer than four such levels are not worth a launch
// of their own.  DDM_SN_TOP_MAX overrides the bound (0: level kernels only).  Needs lev_ptr; sets ltop / ntop.
static inline void decide_top_levels(Factor &F)
{
  int top_max = 128; // (measured: DG 512^2 1.51 / 1.46 / 1.54 / 1.75 ms per solve at 32 / 128 / 512 / 2048, elasticity 1.31 / 1.29 / 1.28 / 1.39; tools/gpu_r04_i.sh)
  if (const char *e = std::getenv("DDM_SN_TOP_MAX")) top_max = std::atoi(e);
  int32_t ltop = F.nlev;
  while (ltop > 0 && F.lev_ptr[(size_t)ltop] - F.lev_ptr[(size_t)ltop - 1] <= top_max) --ltop;
  F.ltop = F.nlev;
  F.ntop = 0;
  if (F.nlev - ltop < 4) return;
  int dev = 0, ncu = 0, per_cu = 0;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) return; // (no device: host-only use)
  const void *fn = F.lu ? (const void *)k_sn_top1<true> : (const void *)k_sn_top1<false>;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, TOP_THREADS, 0) != hipSuccess || per_cu < 1) return;
  F.top_grid = std::min(per_cu, 2) * (ncu / 8 * 8); // co-resident: at most two workgroups of 512 threads per CU
  if (F.top_grid < 8 || F.top_grid > TOP_MAX_WG) return;
  F.ltop = ltop;
  F.ntop = F.nlev - ltop;
}
// Plan of the persistent kernel: per class (block % 8) the top supernodes by level, their forward tiles by (level, colour), their
// backward chunks by level, and the 16-column pieces of all their columns for the gather of the bottom levels' slots.
static inline bool build_top_plan(Factor &F, const std::vector<int32_t> &lev_sn, const std::vector<int32_t> &nrow, const std::vector<int32_t> &first)
{
  if (F.ntop == 0) return true;
  const int32_t ntop = F.ntop, ltop = F.ltop;
  // colours per top level (global over the classes: the barrier count of a level must not depend on the class)
  std::vector<int32_t> fph((size_t)ntop + 1, 0);
  for (int32_t j = 0; j < ntop; ++j) {
    int nc = 1;
    for (int32_t k = F.lev_ptr[(size_t)(ltop + j)]; k < F.lev_ptr[(size_t)(ltop + j) + 1]; ++k) nc = std::max(nc, F.h_colour[(size_t)lev_sn[(size_t)k]] + 1);
    fph[(size_t)j + 1] = fph[(size_t)j] + nc;
  }
  const int32_t nph = fph[(size_t)ntop];
  std::vector<int32_t> a_ptr((size_t)8 * ntop + 1, 0), f_ptr((size_t)8 * nph + 1, 0), p_ptr((size_t)8 * ntop + 1, 0), g_ptr(9, 0), a_sn, f_items, p_items, g_items,
      p_first((size_t)F.nsn, 0);
  for (int c = 0; c < 8; ++c) {
    for (int32_t j = 0; j < ntop; ++j) {
      const int32_t l = ltop + j;
      for (int32_t k = F.lev_ptr[(size_t)l]; k < F.lev_ptr[(size_t)l + 1]; ++k) {
        const int32_t s = lev_sn[(size_t)k];
        if (F.sn_block[(size_t)s] % 8 != c) continue;
        a_sn.push_back(s);
        const int32_t nr = nrow[(size_t)s], ncs = first[(size_t)s + 1] - first[(size_t)s];
        p_first[(size_t)s] = (int32_t)(p_items.size() / 2);
        for (int32_t q = 0; q < (nr + TILE - 1) / TILE; ++q) { // backward: one partial product per 64-row tile
          p_items.push_back(s);
          p_items.push_back(q);
        }
        for (int32_t q = 0; q < (ncs + 15) / 16; ++q) {
          g_items.push_back(s);
          g_items.push_back(q);
        }
      }
      a_ptr[(size_t)c * ntop + j + 1] = (int32_t)a_sn.size();
      p_ptr[(size_t)c * ntop + j + 1] = (int32_t)(p_items.size() / 2);
      for (int32_t col = 0; col < fph[(size_t)j + 1] - fph[(size_t)j]; ++col) { // forward tiles, colour by colour
        for (int32_t k = F.lev_ptr[(size_t)l]; k < F.lev_ptr[(size_t)l + 1]; ++k) {
          const int32_t s = lev_sn[(size_t)k];
          if (F.sn_block[(size_t)s] % 8 != c || F.h_colour[(size_t)s] != col) continue;
          for (int32_t t = 0; t < (nrow[(size_t)s] + TILE - 1) / TILE; ++t) {
            f_items.push_back(s);
            f_items.push_back(t);
          }
          if (nrow[(size_t)s] == 0) { // no rows below (a root): the pseudo tile -1 computes and stores y_s
            f_items.push_back(s);
            f_items.push_back(-1);
          }
        }
        f_ptr[(size_t)c * nph + fph[(size_t)j] + col + 1] = (int32_t)(f_items.size() / 2);
      }
    }
    g_ptr[(size_t)c + 1] = (int32_t)(g_items.size() / 2);
  }
  std::vector<int32_t> all;
  auto put = [&](const std::vector<int32_t> &v) {
    const size_t o = all.size();
    all.insert(all.end(), v.begin(), v.end());
    return o;
  };
  const size_t o_ap = put(a_ptr), o_as = put(a_sn), o_fh = put(fph), o_fp = put(f_ptr), o_fi = put(f_items), o_pp = put(p_ptr), o_pi = put(p_items), o_pf = put(p_first),
               o_gp = put(g_ptr), o_gi = put(g_items);
  if (!up(all, &F.d_top_ints)) return false;
  if (hipMalloc((void **)&F.d_top_partial, sizeof(double) * std::max<size_t>(p_items.size() / 2, 1) * SN_MAX_COLS) != hipSuccess) return false;
  if (hipMalloc((void **)&F.d_top_sync, sizeof(TopSync)) != hipSuccess || hipMemset(F.d_top_sync, 0, sizeof(TopSync)) != hipSuccess) return false;
  const size_t fbytes = sizeof(unsigned long long) * 9 * TOP_MAX_WG * TOP_FLAG_STRIDE;
  if (hipMalloc((void **)&F.d_top_flags, fbytes) != hipSuccess || hipMemset(F.d_top_flags, 0, fbytes) != hipSuccess) return false;
  if (std::getenv("DDM_SN_TOP_STAMPS")) {
    if (hipMalloc((void **)&F.d_top_stamps, 8 * 4000) != hipSuccess || hipMemset(F.d_top_stamps, 0, 8 * 4000) != hipSuccess) return false;
  }
  F.top.ntop = ntop;
  F.top.nph = nph;
  F.top.a_ptr = F.d_top_ints + o_ap;
  F.top.a_sn = F.d_top_ints + o_as;
  F.top.fph = F.d_top_ints + o_fh;
  F.top.f_ptr = F.d_top_ints + o_fp;
  F.top.f_items = F.d_top_ints + o_fi;
  F.top.p_ptr = F.d_top_ints + o_pp;
  F.top.p_items = F.d_top_ints + o_pi;
  F.top.p_first = F.d_top_ints + o_pf;
  F.top.g_ptr = F.d_top_ints + o_gp;
  F.top.g_items = F.d_top_ints + o_gi;
  // one block over all XCDs (write-through hand-overs, one barrier group) only when a block is too large for the bandwidth of one
  // XCD; otherwise block b lives on XCD b % 8 also when fewer than 8 blocks leave XCDs idle
  F.top_spread = (F.nblocks < 8 && (double)F.entries * 8.0 / std::max(1, F.nblocks) > 256e6) ? 1 : 0;
  if (const char *e = std::getenv("DDM_SN_TOP_SPREAD")) F.top_spread = std::atoi(e) != 0;
  return true;
}

// Chains among the top supernodes: s -> s + 1 = parent(s) while the parent has no other child (the links a wide separator was cut
// into).  Chain levels: a chain is one level above the highest chain hanging below it.  Colours inside a chain level: chains whose
// external row sets intersect subtract from the same entries and run one after the other.
static inline bool build_chains(Factor &F, int64_t n, const std::vector<int32_t> &level, const std::vector<int32_t> &first, const std::vector<int32_t> &nrow,
                                const std::vector<int64_t> &rptr, const std::vector<int32_t> &rows, const std::vector<int32_t> &parent_g)
{
  ChainHost &H = F.ch;
  H = ChainHost();
  if (F.ntop == 0) return true;
  if (const char *e = std::getenv("DDM_SN_CHAINS"))
    if (e[0] == '0') return true;
  const int32_t nsn = F.nsn;
  std::vector<int32_t> nchild((size_t)nsn, 0);
  for (int32_t s = 0; s < nsn; ++s)
    if (parent_g[(size_t)s] >= 0) nchild[(size_t)parent_g[(size_t)s]]++;
  H.chain_of.assign((size_t)nsn, -1);
  for (int32_t s = 0; s < nsn; ++s) {
    if (level[(size_t)s] < F.ltop || H.chain_of[(size_t)s] >= 0) continue;
    const int32_t c = H.nchain();
    int32_t cur = s, links = 1;
    H.chain_of[(size_t)s] = c;
    for (;;) {
      const int32_t p = parent_g[(size_t)cur];
      if (p != cur + 1 || nchild[(size_t)p] != 1 || level[(size_t)p] < F.ltop || F.sn_block[(size_t)p] != F.sn_block[(size_t)s]) break;
      H.chain_of[(size_t)p] = c;
      cur = p;
      ++links;
    }
    H.first_sn.push_back(s);
    H.nlinks.push_back(links);
    H.col0.push_back(first[(size_t)s]);
    H.ncol.push_back(first[(size_t)cur + 1] - first[(size_t)s]);
    H.last_sn.push_back(cur);
    H.nE.push_back(nrow[(size_t)cur]);
    H.block.push_back(F.sn_block[(size_t)s]);
    H.max_links = std::max(H.max_links, links);
    H.max_vec = std::max(H.max_vec, std::max(H.ncol.back(), H.nE.back()));
  }
  const int nch = H.nchain();
  H.clevel.assign((size_t)nch, 0);
  for (int c = 0; c < nch; ++c) { // (ascending first supernode: every chain below has been seen)
    const int32_t p = parent_g[(size_t)H.last_sn[(size_t)c]];
    if (p >= 0 && H.chain_of[(size_t)p] >= 0) H.clevel[(size_t)H.chain_of[(size_t)p]] = std::max(H.clevel[(size_t)H.chain_of[(size_t)p]], H.clevel[(size_t)c] + 1);
  }
  // a chain hanging below an INNER link cannot exist (inner links have one child), but one below the first link raises the level
  // only through the loop above: levels are final because children have smaller numbers than the first link of their parent chain
  H.nclev = 0;
  for (int c = 0; c < nch; ++c) H.nclev = std::max(H.nclev, H.clevel[(size_t)c] + 1);
  H.colour.assign((size_t)nch, 0);
  {
    std::vector<uint64_t> rowmask((size_t)n, 0);
    std::vector<int32_t> rowstamp((size_t)n, -1);
    for (int32_t L = 0; L < H.nclev; ++L)
      for (int c = 0; c < nch; ++c) {
        if (H.clevel[(size_t)c] != L) continue;
        const int32_t sl = H.last_sn[(size_t)c];
        uint64_t used = 0;
        for (int64_t q = rptr[(size_t)sl]; q < rptr[(size_t)sl + 1]; ++q)
          if (rowstamp[(size_t)rows[(size_t)q]] == L) used |= rowmask[(size_t)rows[(size_t)q]];
        if (~used == 0) return false;
        const int col = __builtin_ctzll(~used);
        H.colour[(size_t)c] = col;
        for (int64_t q = rptr[(size_t)sl]; q < rptr[(size_t)sl + 1]; ++q) {
          const int32_t r = rows[(size_t)q];
          if (rowstamp[(size_t)r] != L) {
            rowstamp[(size_t)r] = L;
            rowmask[(size_t)r] = 0;
          }
          rowmask[(size_t)r] |= 1ull << col;
        }
      }
  }
  H.woff.assign((size_t)nch, 0);
  H.eoff.assign((size_t)nch, 0);
  for (int c = 0; c < nch; ++c) {
    H.woff[(size_t)c] = H.wtot;
    H.eoff[(size_t)c] = H.etot;
    H.wtot += (int64_t)H.ncol[(size_t)c] * H.ncol[(size_t)c];
    H.etot += (int64_t)H.nE[(size_t)c] * H.ncol[(size_t)c];
  }
  // setup work lists
  H.link_pre.push_back(0);
  for (int c = 0; c < nch; ++c)
    for (int32_t k = 0; k < H.nlinks[(size_t)c]; ++k) {
      const int32_t s = H.first_sn[(size_t)c] + k;
      H.link_sn.push_back(s);
      H.link_chain.push_back(c);
      H.link_pre.push_back(H.link_pre.back() + (first[(size_t)s + 1] - first[(size_t)s] + nrow[(size_t)s] + TILE - 1) / TILE);
    }
  H.inv_ptr.assign((size_t)std::max(H.max_links, 1) + 1, 0);
  for (int32_t d = 1; d < H.max_links; ++d) {
    for (int c = 0; c < nch; ++c)
      for (int32_t i = d; i < H.nlinks[(size_t)c]; ++i) {
        H.inv_chain.push_back(c);
        H.inv_i.push_back(i);
      }
    H.inv_ptr[(size_t)d + 1] = (int32_t)H.inv_chain.size();
  }
  if (H.max_links >= 1) H.inv_ptr[1] = 0;
  return true;
}
// device side of the chains: arrays, the plan of k_sn_top_chain (one allocation of integers)
static inline bool upload_chains(Factor &F)
{
  ChainHost &H = F.ch;
  const int nch = H.nchain();
  if (nch == 0) return true;
  const int nclev = H.nclev;
  // phases of the external-row updates: chain levels x colours
  std::vector<int32_t> eph((size_t)nclev + 1, 0);
  for (int L = 0; L < nclev; ++L) {
    int nc = 1;
    for (int c = 0; c < nch; ++c)
      if (H.clevel[(size_t)c] == L) nc = std::max(nc, H.colour[(size_t)c] + 1);
    eph[(size_t)L + 1] = eph[(size_t)L] + nc;
  }
  const int nph = eph[(size_t)nclev];
  std::vector<int32_t> y_ptr((size_t)8 * nclev + 1, 0), t_ptr((size_t)8 * nclev + 1, 0), x_ptr((size_t)8 * nclev + 1, 0), e_ptr((size_t)8 * nph + 1, 0), y_items, t_items, x_items, e_items;
  for (int cls = 0; cls < 8; ++cls)
    for (int L = 0; L < nclev; ++L) {
      for (int c = 0; c < nch; ++c) {
        if (H.block[(size_t)c] % 8 != cls || H.clevel[(size_t)c] != L) continue;
        const int nb = (H.ncol[(size_t)c] + 63) / 64;
        for (int rb = nb - 1; rb >= 0; --rb) { // the long rows first
          y_items.push_back(c);
          y_items.push_back(rb);
          // columns the rows of the block reach: up to the end of the link of the block's last row (the diagonal blocks of the L U
          // variant are full: the row exchanges are absorbed), for Cholesky the lower triangle is cut by the kernel
          const int32_t lastrow = H.col0[(size_t)c] + std::min(H.ncol[(size_t)c], 64 * rb + 64) - 1;
          int32_t sl = H.first_sn[(size_t)c];
          while (F.h_first[(size_t)sl + 1] <= lastrow) ++sl;
          y_items.push_back(F.h_first[(size_t)sl + 1] - H.col0[(size_t)c]);
        }
        for (int cb = 0; cb < nb; ++cb) { // (the long columns first)
          x_items.push_back(c);
          x_items.push_back(cb);
          if (H.nE[(size_t)c] > 0) {
            t_items.push_back(c);
            t_items.push_back(cb);
          }
        }
      }
      y_ptr[(size_t)cls * nclev + L + 1] = (int32_t)(y_items.size() / 3);
      x_ptr[(size_t)cls * nclev + L + 1] = (int32_t)(x_items.size() / 2);
      t_ptr[(size_t)cls * nclev + L + 1] = (int32_t)(t_items.size() / 2);
      for (int col = 0; col < eph[(size_t)L + 1] - eph[(size_t)L]; ++col) {
        for (int c = 0; c < nch; ++c) {
          if (H.block[(size_t)c] % 8 != cls || H.clevel[(size_t)c] != L || H.colour[(size_t)c] != col) continue;
          for (int t = 0; t < (H.nE[(size_t)c] + 63) / 64; ++t) {
            e_items.push_back(c);
            e_items.push_back(t);
          }
        }
        e_ptr[(size_t)cls * nph + eph[(size_t)L] + col + 1] = (int32_t)(e_items.size() / 2);
      }
    }
  std::vector<int32_t> all;
  auto put = [&](const std::vector<int32_t> &v) {
    const size_t o = all.size();
    all.insert(all.end(), v.begin(), v.end());
    return o;
  };
  const size_t o_c0 = put(H.col0), o_nc = put(H.ncol), o_fs = put(H.first_sn), o_nl = put(H.nlinks), o_ls = put(H.last_sn), o_ne = put(H.nE), o_yp = put(y_ptr), o_yi = put(y_items),
               o_eh = put(eph), o_ep = put(e_ptr), o_ei = put(e_items), o_tp = put(t_ptr), o_ti = put(t_items), o_xp = put(x_ptr), o_xi = put(x_items), o_lsn = put(H.link_sn),
               o_lch = put(H.link_chain), o_lpre = put(H.link_pre), o_ic = put(H.inv_chain), o_ii = put(H.inv_i);
  if (!up(all, &F.d_chain_ints)) return false;
  std::vector<int64_t> offs(H.woff);
  offs.insert(offs.end(), H.eoff.begin(), H.eoff.end());
  if (!up(offs, &F.d_chain_offs)) return false;
  auto dalloc = [](double **p, int64_t cnt) { return hipMalloc((void **)p, sizeof(double) * (size_t)std::max<int64_t>(cnt, 1)) == hipSuccess; };
  if (!dalloc(&F.d_chain_w, H.wtot) || !dalloc(&F.d_chain_e, H.etot)) return false;
  if (F.lu && (!dalloc(&F.d_chain_v, H.wtot) || !dalloc(&F.d_chain_u, H.etot))) return false;
  const int32_t *I = F.d_chain_ints;
  F.chd.nchain = nch;
  F.chd.col0 = I + o_c0;
  F.chd.ncol = I + o_nc;
  F.chd.first_sn = I + o_fs;
  F.chd.nlinks = I + o_nl;
  F.chd.last_sn = I + o_ls;
  F.chd.nE = I + o_ne;
  F.chd.woff = F.d_chain_offs;
  F.chd.eoff = F.d_chain_offs + nch;
  F.chd.W = F.d_chain_w;
  F.chd.E = F.d_chain_e;
  F.chd.V = F.d_chain_v;
  F.chd.U = F.d_chain_u;
  F.chp.nclev = nclev;
  F.chp.nph = nph;
  F.chp.y_ptr = I + o_yp;
  F.chp.y_items = I + o_yi;
  F.chp.eph = I + o_eh;
  F.chp.e_ptr = I + o_ep;
  F.chp.e_items = I + o_ei;
  F.chp.t_ptr = I + o_tp;
  F.chp.t_items = I + o_ti;
  F.chp.x_ptr = I + o_xp;
  F.chp.x_items = I + o_xi;
  F.chp.g_ptr = F.top.g_ptr;
  F.chp.g_items = F.top.g_items;
  F.ch_link_sn = I + o_lsn;
  F.ch_link_chain = I + o_lch;
  F.ch_link_pre = I + o_lpre;
  F.ch_inv_chain = I + o_ic;
  F.ch_inv_i = I + o_ii;
  { // the chain kernel holds one chain vector in dynamic LDS: it must fit and leave the co-resident grid of the top levels possible
    const void *fn = F.lu ? (const void *)k_sn_top_chain<true> : (const void *)k_sn_top_chain<false>;
    const size_t lds = (size_t)H.max_vec * 8;
    int per_cu = 0;
    if (lds > 150 * 1024 || hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, TOP_THREADS, lds) != hipSuccess || per_cu < 1) {
      H = ChainHost(); // (the link-by-link kernel serves the top levels)
      return true;
    }
    int dev = 0, ncu = 0;
    (void)hipGetDevice(&dev);
    F.chain_grid = F.top_grid;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0) F.chain_grid = std::min(per_cu, 2) * (ncu / 8 * 8);
    if (F.chain_grid > TOP_MAX_WG) F.chain_grid = F.top_grid;
  }
  return true;
}
// after the numeric factorisation: dense copies and the blocked inversion of every chain (enqueued; temporaries freed after a sync)
static inline hipError_t chain_setup(Factor &F, hipStream_t st)
{
  F.chains_ready = false;
  const ChainHost &H = F.ch;
  if (H.nchain() == 0) return hipSuccess;
  hipError_t e;
  double *Ltmp = nullptr, *Utmp = nullptr;
  if ((e = hipMalloc((void **)&Ltmp, sizeof(double) * (size_t)std::max<int64_t>(H.wtot, 1))) != hipSuccess) return e;
  if (F.lu && (e = hipMalloc((void **)&Utmp, sizeof(double) * (size_t)std::max<int64_t>(H.wtot, 1))) != hipSuccess) {
    (void)hipFree(Ltmp);
    return e;
  }
  (void)hipMemsetAsync(Ltmp, 0, sizeof(double) * (size_t)H.wtot, st);
  (void)hipMemsetAsync(F.d_chain_w, 0, sizeof(double) * (size_t)H.wtot, st);
  (void)hipMemsetAsync(F.d_chain_e, 0, sizeof(double) * (size_t)std::max<int64_t>(H.etot, 1), st);
  if (F.lu) {
    (void)hipMemsetAsync(Utmp, 0, sizeof(double) * (size_t)H.wtot, st);
    (void)hipMemsetAsync(F.d_chain_v, 0, sizeof(double) * (size_t)H.wtot, st);
    (void)hipMemsetAsync(F.d_chain_u, 0, sizeof(double) * (size_t)std::max<int64_t>(H.etot, 1), st);
  }
  const int nlinks_total = (int)H.link_sn.size();
  const unsigned stiles = (unsigned)H.link_pre.back();
  if (stiles > 0) {
    if (F.lu) hipLaunchKernelGGL(k_chain_scatter<true>, dim3(stiles), dim3(256), 0, st, F.M, F.chd, F.ch_link_sn, F.ch_link_chain, F.ch_link_pre, nlinks_total, Ltmp, Utmp);
    else hipLaunchKernelGGL(k_chain_scatter<false>, dim3(stiles), dim3(256), 0, st, F.M, F.chd, F.ch_link_sn, F.ch_link_chain, F.ch_link_pre, nlinks_total, Ltmp, Utmp);
  }
  static DeviceOnce attr_once;
  constexpr size_t inv_lds = (size_t)(SN_MAX_COLS * 64 + 32 * 64) * 8;
  attr_once.run([]() { (void)hipFuncSetAttribute((const void *)k_chain_invert, hipFuncAttributeMaxDynamicSharedMemorySize, (int)inv_lds); });
  for (int32_t d = 1; d < H.max_links; ++d) {
    const int32_t i0 = H.inv_ptr[(size_t)d], i1 = H.inv_ptr[(size_t)d + 1];
    if (i1 <= i0) continue;
    hipLaunchKernelGGL(k_chain_invert, dim3((unsigned)(2 * (i1 - i0))), dim3(512), inv_lds, st, F.M, F.chd, (int)d, F.ch_inv_chain + i0, F.ch_inv_i + i0, (const double *)Ltmp, F.d_chain_w);
    if (F.lu)
      hipLaunchKernelGGL(k_chain_invert, dim3((unsigned)(2 * (i1 - i0))), dim3(512), inv_lds, st, F.M, F.chd, (int)d, F.ch_inv_chain + i0, F.ch_inv_i + i0, (const double *)Utmp, F.d_chain_v);
  }
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(Ltmp);
  if (Utmp) (void)hipFree(Utmp);
  if (e == hipSuccess) F.chains_ready = true;
  return e;
}

// symbolic results of all blocks -> one global structure on the device.  Returns false on an allocation failure.
static inline bool build(Factor &F, int64_t n, int64_t nblocks, const int64_t *block_ptr, std::vector<BlockSym> &BS, bool lu = false)
{
  F.n = n;
  F.lu = lu;
  F.nblocks = (int)nblocks;
  std::vector<int32_t> first, nrow, rows, sn_of_col((size_t)n), iperm((size_t)n), level, parent_g;
  std::vector<int64_t> rptr(1, 0), pptr(1, 0);
  F.h_perm.resize((size_t)n);
  for (int64_t b = 0; b < nblocks; ++b) {
    BlockSym &S = BS[(size_t)b];
    const int32_t off = (int32_t)block_ptr[b];
    const int32_t nsn = (int32_t)S.first.size() - 1;
    const int32_t sn_base = (int32_t)first.size();
    for (int32_t s = 0; s < nsn; ++s) {
      const int32_t gs = (int32_t)first.size();
      first.push_back(off + S.first[(size_t)s]);
      const int64_t r0 = S.rptr[(size_t)s], r1 = S.rptr[(size_t)s + 1];
      nrow.push_back((int32_t)(r1 - r0));
      for (int64_t k = r0; k < r1; ++k) rows.push_back(off + S.rows[(size_t)k]);
      rptr.push_back((int64_t)rows.size());
      const int64_t nc = S.first[(size_t)s + 1] - S.first[(size_t)s];
      pptr.push_back(pptr.back() + nc * (nc + (r1 - r0)));
      level.push_back(S.level[(size_t)s]);
      parent_g.push_back(S.parent[(size_t)s] < 0 ? -1 : sn_base + S.parent[(size_t)s]);
      F.sn_block.push_back((int32_t)b);
      for (int32_t c = S.first[(size_t)s]; c < S.first[(size_t)s + 1]; ++c) sn_of_col[(size_t)(off + c)] = gs;
    }
    for (int32_t k = 0; k < S.n; ++k) {
      F.h_perm[(size_t)(off + k)] = off + S.perm[(size_t)k];
      iperm[(size_t)(off + S.perm[(size_t)k])] = off + k;
    }
    F.flops += S.flops;
    S = BlockSym();
  }
  first.push_back((int32_t)n);
  F.nsn = (int32_t)nrow.size();
  F.entries = pptr.back();
  int32_t nlev = 0;
  for (int32_t l : level) nlev = std::max(nlev, l + 1);
  F.nlev = nlev;
  // supernodes by level (stable) and the per-level tile prefixes
  F.lev_ptr.assign((size_t)nlev + 1, 0);
  for (int32_t l : level) F.lev_ptr[(size_t)l + 1]++;
  for (int32_t l = 0; l < nlev; ++l) F.lev_ptr[(size_t)l + 1] += F.lev_ptr[(size_t)l];
  std::vector<int32_t> lev_sn((size_t)F.nsn), pos(F.lev_ptr.begin(), F.lev_ptr.end() - 1);
  for (int32_t s = 0; s < F.nsn; ++s) lev_sn[(size_t)pos[(size_t)level[(size_t)s]]++] = s;
  // colours of the update phases (see Factor::lev_phase_ptr): greedy, in list order; a row remembers which colours of the CURRENT
  // level already subtract from it
  F.lev_phase_ptr.assign((size_t)nlev + 1, 0);
  F.phase_k.clear();
  {
    std::vector<uint64_t> rowmask((size_t)n, 0);
    std::vector<int32_t> rowstamp((size_t)n, -1), colour((size_t)F.nsn, 0), sorted;
    for (int32_t l = 0; l < nlev; ++l) {
      const int32_t k0 = F.lev_ptr[(size_t)l], k1 = F.lev_ptr[(size_t)l + 1];
      int ncol = 1;
      for (int32_t k = k0; k < k1; ++k) {
        const int32_t s = lev_sn[(size_t)k];
        uint64_t used = 0;
        for (int64_t q = rptr[(size_t)s]; q < rptr[(size_t)s + 1]; ++q) {
          const int32_t r = rows[(size_t)q];
          if (rowstamp[(size_t)r] == l) used |= rowmask[(size_t)r];
        }
        if (~used == 0) return false; // more than 64 mutually conflicting supernodes in one level (not seen: <= 11 on 3-D grids)
        const int c = __builtin_ctzll(~used);
        colour[(size_t)s] = c;
        ncol = std::max(ncol, c + 1);
        for (int64_t q = rptr[(size_t)s]; q < rptr[(size_t)s + 1]; ++q) {
          const int32_t r = rows[(size_t)q];
          if (rowstamp[(size_t)r] != l) {
            rowstamp[(size_t)r] = l;
            rowmask[(size_t)r] = 0;
          }
          rowmask[(size_t)r] |= 1ull << c;
        }
      }
      sorted.assign(lev_sn.begin() + k0, lev_sn.begin() + k1);
      std::stable_sort(sorted.begin(), sorted.end(), [&](int32_t a, int32_t b) { return colour[(size_t)a] < colour[(size_t)b]; });
      std::copy(sorted.begin(), sorted.end(), lev_sn.begin() + k0);
      F.lev_phase_ptr[(size_t)l] = (int32_t)F.phase_k.size();
      for (int32_t k = k0; k < k1; ++k)
        if (k == k0 || colour[(size_t)lev_sn[(size_t)k]] != colour[(size_t)lev_sn[(size_t)k - 1]]) F.phase_k.push_back(k - k0);
      F.phase_k.push_back(k1 - k0);
      (void)ncol;
    }
    F.lev_phase_ptr[(size_t)nlev] = (int32_t)F.phase_k.size();
    F.h_colour = colour;
  }
  std::vector<int32_t> preT((size_t)F.nsn + nlev), preU((size_t)F.nsn + nlev), preUF((size_t)F.nsn + nlev), big_sn, big_index((size_t)F.nsn, -1), preB;
  std::vector<int64_t> uptr(1, 0);
  for (int32_t s = 0; s < F.nsn; ++s) uptr.push_back(uptr.back() + (int64_t)nrow[(size_t)s] * (first[(size_t)s + 1] - first[(size_t)s]));
  F.uentries = lu ? uptr.back() : 0;
  F.h_tilesUF.assign((size_t)nlev, 0);
  F.h_tilesT.assign((size_t)nlev, 0);
  F.h_tilesU.assign((size_t)nlev, 0);
  F.h_tilesB.assign((size_t)nlev, 0);
  F.lev_maxnc.assign((size_t)nlev, 0);
  F.lev_maxnr.assign((size_t)nlev, 0);
  F.lev_big_ptr.assign((size_t)nlev + 1, 0);
  for (int32_t l = 0; l < nlev; ++l) {
    int64_t aT = 0, aU = 0, aB = 0, aUF = 0;
    const int32_t base = F.lev_ptr[(size_t)l] + l;
    const int32_t bbase = (int32_t)preB.size();
    preB.push_back(0);
    for (int32_t k = F.lev_ptr[(size_t)l]; k < F.lev_ptr[(size_t)l + 1]; ++k) {
      const int32_t s = lev_sn[(size_t)k];
      const int64_t T = (nrow[(size_t)s] + TILE - 1) / TILE;
      preT[(size_t)(base + k - F.lev_ptr[(size_t)l])] = (int32_t)aT;
      preU[(size_t)(base + k - F.lev_ptr[(size_t)l])] = (int32_t)aU;
      preUF[(size_t)(base + k - F.lev_ptr[(size_t)l])] = (int32_t)aUF;
      aT += T;
      aU += T * (T + 1) / 2;
      aUF += T * T;
      if (T > BWD_SMALL) {
        big_index[(size_t)k] = (int32_t)(big_sn.size() - (size_t)F.lev_big_ptr[(size_t)l]);
        big_sn.push_back(s);
        aB += (nrow[(size_t)s] + BWD_ROWS - 1) / BWD_ROWS;
        preB.push_back((int32_t)aB);
      }
      F.lev_maxnc[(size_t)l] = std::max(F.lev_maxnc[(size_t)l], first[(size_t)s + 1] - first[(size_t)s]);
      F.lev_maxnr[(size_t)l] = std::max(F.lev_maxnr[(size_t)l], nrow[(size_t)s]);
    }
    if (aU > 2000000000ll || (lu && aUF > 2000000000ll)) return false;
    preUF[(size_t)(base + F.lev_ptr[(size_t)l + 1] - F.lev_ptr[(size_t)l])] = (int32_t)aUF;
    F.h_tilesUF[(size_t)l] = (int32_t)aUF;
    preT[(size_t)(base + F.lev_ptr[(size_t)l + 1] - F.lev_ptr[(size_t)l])] = (int32_t)aT;
    preU[(size_t)(base + F.lev_ptr[(size_t)l + 1] - F.lev_ptr[(size_t)l])] = (int32_t)aU;
    F.h_tilesT[(size_t)l] = (int32_t)aT;
    F.h_tilesU[(size_t)l] = (int32_t)aU;
    F.h_tilesB[(size_t)l] = (int32_t)aB;
    F.lev_big_ptr[(size_t)l + 1] = (int32_t)big_sn.size();
    F.max_big_tiles = std::max(F.max_big_tiles, aB);
    (void)bbase;
  }
  F.h_preU = preU;
  F.h_preUF = preUF;
  F.h_preT = preT;
  // transposed row lists (counting sort of the entries of `rows` by value; per column first the entries of bottom-level supernodes,
  // then those of top-level ones, each part in ascending position = ascending source supernode)
  if (rows.size() >= (size_t)0x7fffffff) return false;
  F.nrows_total = (int64_t)rows.size();
  decide_top_levels(F);
  {
    std::vector<int64_t> tptr((size_t)n + 1, 0), tmid((size_t)n, 0);
    for (int32_t s = 0; s < F.nsn; ++s)
      for (int64_t q = rptr[(size_t)s]; q < rptr[(size_t)s + 1]; ++q) {
        tptr[(size_t)rows[(size_t)q] + 1]++;
        if (level[(size_t)s] < F.ltop) tmid[(size_t)rows[(size_t)q]]++;
      }
    for (int64_t c = 0; c < n; ++c) tptr[(size_t)c + 1] += tptr[(size_t)c];
    for (int64_t c = 0; c < n; ++c) tmid[(size_t)c] += tptr[(size_t)c];
    std::vector<int32_t> tidx(rows.size());
    std::vector<int64_t> fill_lo(tptr.begin(), tptr.end() - 1), fill_hi(tmid);
    for (int32_t s = 0; s < F.nsn; ++s)
      for (int64_t q = rptr[(size_t)s]; q < rptr[(size_t)s + 1]; ++q) {
        const int32_t r = rows[(size_t)q];
        if (level[(size_t)s] < F.ltop) tidx[(size_t)fill_lo[(size_t)r]++] = (int32_t)q;
        else tidx[(size_t)fill_hi[(size_t)r]++] = (int32_t)q;
      }
    std::vector<int32_t> tpos(rows.size());
    for (size_t k = 0; k < tidx.size(); ++k) tpos[(size_t)tidx[k]] = (int32_t)k;
    if (!up(tptr, &F.d_tptr) || !up(tmid, &F.d_tmid) || !up(tidx, &F.d_tidx) || !up(tpos, &F.d_tpos)) return false;
  }
  bool ok = up(first, &F.d_first) && up(nrow, &F.d_nrow) && up(rows, &F.d_rows) && up(sn_of_col, &F.d_sn_of_col) && up(iperm, &F.d_iperm) && up(F.h_perm, &F.d_perm) &&
            up(rptr, &F.d_rptr) && up(pptr, &F.d_pptr) && up(lev_sn, &F.d_lev_sn) && up(preT, &F.d_preT) && up(preU, &F.d_preU) && up(big_sn, &F.d_big_sn) &&
            up(big_index, &F.d_big_index) && up(preB, &F.d_preB);
  if (!ok) return false;
  if (hipMalloc((void **)&F.d_err, 128) != hipSuccess || hipMemset(F.d_err, 0, 128) != hipSuccess) return false;
  if (hipMalloc((void **)&F.d_panels, sizeof(double) * (size_t)std::max<int64_t>(F.entries, 1)) != hipSuccess) return false;
  if (lu) {
    if (!up(uptr, &F.d_uptr) || !up(preUF, &F.d_preUF)) return false;
    if (hipMalloc((void **)&F.d_upanels, sizeof(double) * (size_t)std::max<int64_t>(F.uentries, 1)) != hipSuccess) return false;
    if (hipMalloc((void **)&F.d_piv, sizeof(int32_t) * (size_t)std::max<int64_t>(n, 1)) != hipSuccess) return false;
  }
  if (!build_top_plan(F, lev_sn, nrow, first)) return false;
  F.h_first = first;
  if (!build_chains(F, n, level, first, nrow, rptr, rows, parent_g) || !upload_chains(F)) return false;
  F.M.nsn = F.nsn;
  F.M.first = F.d_first;
  F.M.nrow = F.d_nrow;
  F.M.rptr = F.d_rptr;
  F.M.rows = F.d_rows;
  F.M.pptr = F.d_pptr;
  F.M.sn_of_col = F.d_sn_of_col;
  F.M.tptr = F.d_tptr;
  F.M.tmid = F.d_tmid;
  F.M.tidx = F.d_tidx;
  F.M.tpos = F.d_tpos;
  F.M.panels = F.d_panels;
  F.M.upanels = F.d_upanels;
  F.M.uptr = F.d_uptr;
  F.M.piv = F.d_piv;
  return true;
}

// numeric factorisation of the matrix (device CSR, original numbering); *bad = supernode + 1 whose diagonal block was not positive definite
// tiny: replacement of a vanishing pivot column in the L U variant; *perturbed: how many were replaced
static inline hipError_t factorize(Factor &F, hipStream_t st, const int64_t *d_rp, const int32_t *d_ci, const double *d_va, unsigned *bad, double tiny = 0.0,
                                   unsigned *perturbed = nullptr)
{
  hipError_t e = hipMemsetAsync(F.d_panels, 0, sizeof(double) * (size_t)std::max<int64_t>(F.entries, 1), st);
  if (e != hipSuccess) return e;
  (void)hipMemsetAsync(F.d_err, 0, 16, st);
  if (F.lu) {
    e = hipMemsetAsync(F.d_upanels, 0, sizeof(double) * (size_t)std::max<int64_t>(F.uentries, 1), st);
    if (e != hipSuccess) return e;
  }
  if (F.n > 0) {
    if (F.lu) hipLaunchKernelGGL(k_sn_assemble_lu, dim3((unsigned)((F.n + 255) / 256)), dim3(256), 0, st, F.M, F.n, d_rp, d_ci, d_va, F.d_iperm);
    else hipLaunchKernelGGL(k_sn_assemble, dim3((unsigned)((F.n + 255) / 256)), dim3(256), 0, st, F.M, F.n, d_rp, d_ci, d_va, F.d_iperm);
  }
  static DeviceOnce attr_once;
  attr_once.run([]() {
    (void)hipFuncSetAttribute((const void *)k_sn_diag, hipFuncAttributeMaxDynamicSharedMemorySize, (SN_MAX_COLS + 1) * SN_MAX_COLS * 8);
    (void)hipFuncSetAttribute((const void *)k_sn_lu_diag, hipFuncAttributeMaxDynamicSharedMemorySize, (SN_MAX_COLS + 1) * SN_MAX_COLS * 8);
  });
  for (int32_t l = 0; l < F.nlev; ++l) {
    const int32_t cnt = F.lev_ptr[(size_t)l + 1] - F.lev_ptr[(size_t)l];
    if (cnt == 0) continue;
    const int32_t *lsn = F.d_lev_sn + F.lev_ptr[(size_t)l];
    const int nc = F.lev_maxnc[(size_t)l];
    if (F.lu) {
      hipLaunchKernelGGL(k_sn_lu_diag, dim3((unsigned)cnt), dim3(256), (size_t)(nc | 1) * nc * 8, st, F.M, lsn, F.d_err, tiny);
      if (F.h_tilesT[(size_t)l] > 0)
        hipLaunchKernelGGL(k_sn_lu_panel, dim3((unsigned)F.h_tilesT[(size_t)l]), dim3(256), 0, st, F.M, lsn, (const int32_t *)(F.d_preT + F.lev_ptr[(size_t)l] + l), cnt);
      for (int32_t ph = F.lev_phase_ptr[(size_t)l]; ph + 1 < F.lev_phase_ptr[(size_t)l + 1]; ++ph) { // colour by colour
        const int32_t b0 = F.h_preUF[(size_t)(F.lev_ptr[(size_t)l] + l + F.phase_k[(size_t)ph])], b1 = F.h_preUF[(size_t)(F.lev_ptr[(size_t)l] + l + F.phase_k[(size_t)ph + 1])];
        if (b1 > b0) hipLaunchKernelGGL(k_sn_lu_update, dim3((unsigned)(b1 - b0)), dim3(256), 0, st, F.M, lsn, (const int32_t *)(F.d_preUF + F.lev_ptr[(size_t)l] + l), cnt, (int)b0);
      }
      continue;
    }
    hipLaunchKernelGGL(k_sn_diag, dim3((unsigned)cnt), dim3(256), (size_t)(nc | 1) * nc * 8, st, F.M, lsn, F.d_err);
    if (F.h_tilesT[(size_t)l] > 0)
      hipLaunchKernelGGL(k_sn_panel, dim3((unsigned)F.h_tilesT[(size_t)l]), dim3(256), 0, st, F.M, lsn, (const int32_t *)(F.d_preT + F.lev_ptr[(size_t)l] + l), cnt);
    for (int32_t ph = F.lev_phase_ptr[(size_t)l]; ph + 1 < F.lev_phase_ptr[(size_t)l + 1]; ++ph) { // colour by colour
      const int32_t b0 = F.h_preU[(size_t)(F.lev_ptr[(size_t)l] + l + F.phase_k[(size_t)ph])], b1 = F.h_preU[(size_t)(F.lev_ptr[(size_t)l] + l + F.phase_k[(size_t)ph + 1])];
      if (b1 > b0) hipLaunchKernelGGL(k_sn_update, dim3((unsigned)(b1 - b0)), dim3(256), 0, st, F.M, lsn, (const int32_t *)(F.d_preU + F.lev_ptr[(size_t)l] + l), cnt, (int)b0);
    }
  }
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(st);
  if (e != hipSuccess) return e;
  unsigned words[4] = {0, 0, 0, 0};
  e = hipMemcpy(words, F.d_err, 16, hipMemcpyDeviceToHost);
  *bad = words[0];
  if (perturbed) *perturbed = words[2];
  if (e == hipSuccess && words[0] == 0) e = chain_setup(F, st); // the dense inverses of the top chains (single-vector solves)
  return e;
}

// scratch of the backward sweep for m right-hand sides (call OUTSIDE a stream capture)
static inline bool reserve(Factor &F, int m)
{
  const int64_t cneed = F.nrows_total; // slots of the single-vector forward sweep (the block solves push coloured updates)
  if (cneed > F.contrib_cap) {
    if (F.d_contrib) (void)hipFree(F.d_contrib);
    F.d_contrib = nullptr;
    F.contrib_cap = 0;
    if (hipMalloc((void **)&F.d_contrib, sizeof(double) * (size_t)std::max<int64_t>(cneed, 1)) != hipSuccess) return false;
    F.contrib_cap = cneed;
  }
  const int64_t need = F.max_big_tiles * SN_MAX_COLS * (int64_t)m;
  if (need <= F.partial_cap) return true;
  if (F.d_partial) (void)hipFree(F.d_partial);
  F.d_partial = nullptr;
  F.partial_cap = 0;
  if (hipMalloc((void **)&F.d_partial, sizeof(double) * (size_t)std::max<int64_t>(need, 1)) != hipSuccess) return false;
  F.partial_cap = need;
  return true;
}

template <bool LU>
static inline void solve_t(const Factor &F, hipStream_t st, int m, double *B, int64_t ldb, double *Yvec, unsigned *err)
{
  if (m == 1 && ldb == 1 && Yvec) { // the single-vector kernels (sn_solve1.hpp): level launches at the bottom, one persistent launch for the top
    const int32_t lbot = F.ntop > 0 ? F.ltop : F.nlev; // levels [0, lbot) by launches
    const char *sk = std::getenv("DDM_SN_SMALL_KERNELS");
    const bool small_kernels = !(sk && sk[0] == '0');
    for (int32_t l = 0; l < lbot; ++l) {
      const int32_t cnt = F.lev_ptr[(size_t)l + 1] - F.lev_ptr[(size_t)l];
      if (cnt == 0) continue;
      const int32_t *lsn = F.d_lev_sn + F.lev_ptr[(size_t)l], *preT = F.d_preT + F.lev_ptr[(size_t)l] + l;
      const unsigned grid = (unsigned)(F.h_tilesT[(size_t)l] + cnt);
      if (F.lev_maxnc[(size_t)l] <= 64 && F.lev_maxnr[(size_t)l] <= 192 && small_kernels) // a wavefront per supernode
        hipLaunchKernelGGL(k_sn_fwd1_small<LU>, dim3((unsigned)((cnt + 3) / 4)), dim3(256), 0, st, F.M, lsn, cnt, (const double *)B, Yvec, F.d_contrib);
      else if (F.lev_maxnc[(size_t)l] <= 64) hipLaunchKernelGGL((k_sn_fwd1<LU, 64>), dim3(grid), dim3(256), 0, st, F.M, lsn, preT, cnt, (const double *)B, Yvec, F.d_contrib);
      else hipLaunchKernelGGL((k_sn_fwd1<LU, 128>), dim3(grid), dim3(512), 0, st, F.M, lsn, preT, cnt, (const double *)B, Yvec, F.d_contrib);
    }
    if (F.ntop > 0 && F.chains_ready) {
      hipLaunchKernelGGL(k_sn_top_prologue, dim3(1), dim3(64), 0, st, F.d_top_sync);
      hipLaunchKernelGGL(k_sn_top_chain<LU>, dim3((unsigned)F.chain_grid), dim3(TOP_THREADS), (size_t)F.ch.max_vec * 8, st, F.M, F.chd, F.chp, F.nblocks, F.top_spread, B, Yvec,
                         (const double *)F.d_contrib, F.d_top_sync, F.d_top_flags, err ? err : F.d_err + 1, F.d_top_stamps);
    } else if (F.ntop > 0) {
      hipLaunchKernelGGL(k_sn_top_prologue, dim3(1), dim3(64), 0, st, F.d_top_sync);
      hipLaunchKernelGGL(k_sn_top1<LU>, dim3((unsigned)F.top_grid), dim3(TOP_THREADS), 0, st, F.M, F.top, F.nblocks, F.top_spread, B, Yvec, F.d_contrib, F.d_top_partial,
                         F.d_top_sync, F.d_top_flags, err ? err : F.d_err + 1, F.d_top_stamps);
    }
    for (int32_t l = lbot - 1; l >= 0; --l) {
      const int32_t cnt = F.lev_ptr[(size_t)l + 1] - F.lev_ptr[(size_t)l];
      if (cnt == 0) continue;
      const int32_t *lsn = F.d_lev_sn + F.lev_ptr[(size_t)l];
      const int32_t nbig = F.lev_big_ptr[(size_t)l + 1] - F.lev_big_ptr[(size_t)l];
      const int32_t *preB = F.d_preB + F.lev_big_ptr[(size_t)l] + l, *bsn = F.d_big_sn + F.lev_big_ptr[(size_t)l], *bidx = F.d_big_index + F.lev_ptr[(size_t)l];
      const bool small = F.lev_maxnc[(size_t)l] <= 64;
      if (small && F.lev_maxnr[(size_t)l] <= 192 && small_kernels) {
        hipLaunchKernelGGL(k_sn_bwd1_small<LU>, dim3((unsigned)((cnt + 3) / 4)), dim3(256), 0, st, F.M, lsn, cnt, (const double *)Yvec, B);
        continue;
      }
      if (nbig > 0) {
        if (small) hipLaunchKernelGGL((k_sn_bwd1_partial<LU, 64>), dim3((unsigned)F.h_tilesB[(size_t)l]), dim3(256), 0, st, F.M, bsn, preB, nbig, (const double *)B, F.d_partial);
        else hipLaunchKernelGGL((k_sn_bwd1_partial<LU, 128>), dim3((unsigned)F.h_tilesB[(size_t)l]), dim3(512), 0, st, F.M, bsn, preB, nbig, (const double *)B, F.d_partial);
      }
      if (small) hipLaunchKernelGGL((k_sn_bwd1_diag<LU, 64>), dim3((unsigned)cnt), dim3(256), 0, st, F.M, lsn, bidx, preB, (const double *)F.d_partial, (const double *)Yvec, B);
      else hipLaunchKernelGGL((k_sn_bwd1_diag<LU, 128>), dim3((unsigned)cnt), dim3(512), 0, st, F.M, lsn, bidx, preB, (const double *)F.d_partial, (const double *)Yvec, B);
    }
    return;
  }
  static DeviceOnce attr_once; // (one per instantiation: LU / Cholesky)
  attr_once.run([]() {
    (void)hipFuncSetAttribute((const void *)k_sn_fwd_diag<LU>, hipFuncAttributeMaxDynamicSharedMemorySize, SN_MAX_COLS * 48 * 8);
    (void)hipFuncSetAttribute((const void *)k_sn_fwd_update, hipFuncAttributeMaxDynamicSharedMemorySize, SN_MAX_COLS * 48 * 8);
    (void)hipFuncSetAttribute((const void *)k_sn_bwd_diag<LU>, hipFuncAttributeMaxDynamicSharedMemorySize, (SN_MAX_COLS + TILE) * 48 * 8);
  });
  const int mpad = ((m + 15) / 16) * 16;
  for (int32_t l = 0; l < F.nlev; ++l) {
    const int32_t cnt = F.lev_ptr[(size_t)l + 1] - F.lev_ptr[(size_t)l];
    if (cnt == 0) continue;
    const int32_t *lsn = F.d_lev_sn + F.lev_ptr[(size_t)l];
    const size_t lds = (size_t)F.lev_maxnc[(size_t)l] * mpad * 8;
    hipLaunchKernelGGL(k_sn_fwd_diag<LU>, dim3((unsigned)cnt), dim3(256), lds, st, F.M, lsn, m, B, ldb);
    for (int32_t ph = F.lev_phase_ptr[(size_t)l]; ph + 1 < F.lev_phase_ptr[(size_t)l + 1]; ++ph) { // colour by colour
      const int32_t b0 = F.h_preT[(size_t)(F.lev_ptr[(size_t)l] + l + F.phase_k[(size_t)ph])], b1 = F.h_preT[(size_t)(F.lev_ptr[(size_t)l] + l + F.phase_k[(size_t)ph + 1])];
      if (b1 > b0)
        hipLaunchKernelGGL(k_sn_fwd_update, dim3((unsigned)(b1 - b0)), dim3(256), lds, st, F.M, lsn, (const int32_t *)(F.d_preT + F.lev_ptr[(size_t)l] + l), cnt, m, B, ldb, (int)b0);
    }
  }
  for (int32_t l = F.nlev - 1; l >= 0; --l) {
    const int32_t cnt = F.lev_ptr[(size_t)l + 1] - F.lev_ptr[(size_t)l];
    if (cnt == 0) continue;
    const int32_t *lsn = F.d_lev_sn + F.lev_ptr[(size_t)l];
    const int32_t nbig = F.lev_big_ptr[(size_t)l + 1] - F.lev_big_ptr[(size_t)l];
    const int32_t *preB = F.d_preB + F.lev_big_ptr[(size_t)l] + l;
    if (nbig > 0)
      hipLaunchKernelGGL(k_sn_bwd_partial<LU>, dim3((unsigned)F.h_tilesB[(size_t)l]), dim3(256), (size_t)TILE * mpad * 8, st, F.M, (const int32_t *)(F.d_big_sn + F.lev_big_ptr[(size_t)l]),
                         preB, nbig, m, (const double *)B, ldb, F.d_partial);
    hipLaunchKernelGGL(k_sn_bwd_diag<LU>, dim3((unsigned)cnt), dim3(256), (size_t)(F.lev_maxnc[(size_t)l] + TILE) * mpad * 8, st, F.M, lsn,
                       (const int32_t *)(F.d_big_index + F.lev_ptr[(size_t)l]), preB, (const double *)F.d_partial, m, B, ldb);
  }
}

// in-place solve (L L^T resp. P^T L U) X = B on the permuted row-major work block (n x m, leading dimension ldb); enqueues only
// Yvec: a second n-vector for the single-vector kernels (m == 1), or nullptr = the block kernels
// err: status word of the persistent single-vector kernel (a device-visible word the caller watches; nullptr: the factor's own)
static inline void solve(const Factor &F, hipStream_t st, int m, double *B, int64_t ldb, double *Yvec = nullptr, unsigned *err = nullptr)
{
  if (F.lu) solve_t<true>(F, st, m, B, ldb, Yvec, err);
  else solve_t<false>(F, st, m, B, ldb, Yvec, err);
}

} // namespace sn
