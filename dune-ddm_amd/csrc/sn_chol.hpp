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
__global__ __launch_bounds__(256) void k_sn_update(Meta M, const int32_t *__restrict__ lev_sn, const int32_t *__restrict__ pre, int cnt)
{
  __shared__ int32_t rowid[TILE], colid[TILE], slot_of_col[TILE], slot_t[TILE], slot_first[TILE];
  __shared__ int64_t slot_base[TILE], slot_ld[TILE];
  __shared__ int32_t rpos[TILE * TILE]; // [slot][row]
  __shared__ int nslots_s;
  const int it = find_item(pre, cnt, (int32_t)blockIdx.x);
  const int32_t s = lev_sn[it];
  int u = (int)blockIdx.x - pre[it]; // index into the lower triangle of the T x T tile grid, row-major: u = ti (ti + 1) / 2 + tj
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
        unsafeAtomicAdd(M.panels + slot_base[sl] + rp + (int64_t)(gc - slot_first[sl]) * slot_ld[sl], -acc[a][b][q]);
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
__global__ __launch_bounds__(256) void k_sn_lu_diag(Meta M, const int32_t *__restrict__ lev_sn, unsigned *__restrict__ err)
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
      if (!(amax > 0.0) || !(amax < 1.7e308)) {
        atomicCAS(err, 0u, (unsigned)s + 1u);
        a[k + k * ldl] = 1.0;
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
__global__ __launch_bounds__(256) void k_sn_lu_update(Meta M, const int32_t *__restrict__ lev_sn, const int32_t *__restrict__ pre, int cnt)
{
  __shared__ int32_t rowid[TILE], colid[TILE], cslot[TILE], rslot[TILE], cs_t[TILE], rs_t[TILE];
  __shared__ int32_t posc[TILE * TILE]; // [column slot][row]: position of the row in the column owner's index list
  __shared__ int32_t posr[TILE * TILE]; // [row slot][column]: position of the column in the row owner's index list
  __shared__ int ncs_s, nrs_s;
  const int it = find_item(pre, cnt, (int32_t)blockIdx.x);
  const int32_t s = lev_sn[it];
  const int u = (int)blockIdx.x - pre[it];
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
        unsafeAtomicAdd(M.panels + M.pptr[tt] + posc[cslot[c] * TILE + r] + (int64_t)(gc - f) * ((int64_t)(M.first[tt + 1] - f) + M.nrow[tt]), -acc[t][q]);
      } else { // upper part: owner of row gr
        const int32_t tt = rs_t[rslot[r]];
        const int32_t f = M.first[tt], ncc = M.first[tt + 1] - f, nrr = M.nrow[tt];
        const int32_t pc = posr[rslot[r] * TILE + c];
        if (pc < ncc) unsafeAtomicAdd(M.panels + M.pptr[tt] + (gr - f) + (int64_t)pc * (ncc + nrr), -acc[t][q]); // inside the diagonal block
        else unsafeAtomicAdd(M.upanels + M.uptr[tt] + (pc - ncc) + (int64_t)(gr - f) * nrr, -acc[t][q]);
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
// B[rows] -= R_s Y_s, one workgroup per 64 rows of R_s
__global__ __launch_bounds__(256) void k_sn_fwd_update(Meta M, const int32_t *__restrict__ lev_sn, const int32_t *__restrict__ pre, int cnt, int m,
                                                      double *__restrict__ B, int64_t ldb)
{
  extern __shared__ __attribute__((aligned(16))) double ys[]; // nc x mpad
  const int it = find_item(pre, cnt, (int32_t)blockIdx.x);
  const int32_t s = lev_sn[it];
  const int tile = (int)blockIdx.x - pre[it];
  const int32_t f = M.first[s], nc = M.first[s + 1] - f, nr = M.nrow[s];
  const int64_t ld = nc + nr;
  const double *P = M.panels + M.pptr[s];
  const int32_t *R = M.rows + M.rptr[s];
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
      if (r < nr && c < m) unsafeAtomicAdd(B + (int64_t)R[r] * ldb + c, -acc[t][q]);
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

// ---- ONE right-hand side (the Schwarz apply inside a Krylov loop) -----------------------------------------------------------------
// GEMV-shaped variants of the four solve kernels: 512 threads, every thread's slice of loads in flight at once, the k range (or the
// rows) split over thread groups and summed in a fixed order; forward: ONE launch per level -- every workgroup of a supernode
// recomputes y_s = W_s b_s itself (at most 128^2 / 2 multiply-adds), item 0 stores it to Y (a separate vector: the others still read
// b_s), items 1.. apply their 64 rows of the update.
// NCMAX = 128 (4 x 128 threads) or 64 (levels whose widest supernode has at most 64 columns -- the many small supernodes at the bottom
// of the tree: half the threads per workgroup, twice the workgroups per CU)
__device__ __forceinline__ double s1_wave_sum(double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
template <bool LU, int NCMAX>
__global__ __launch_bounds__(4 * NCMAX) void k_sn_fwd1(Meta M, const int32_t *__restrict__ lev_sn, const int32_t *__restrict__ pre, int cnt, double *__restrict__ B,
                                                  double *__restrict__ Y)
{
  constexpr int YS = NCMAX / 4; // k slice of the four thread groups that compute y
  __shared__ double bs[NCMAX], ys[NCMAX], part[4 * NCMAX];
  int lo = 0, hi = cnt; // item -> (supernode, local item): largest i with pre[i] + i <= item (1 + T_i items per supernode)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (pre[mid] + mid <= (int)blockIdx.x) lo = mid;
    else hi = mid;
  }
  const int32_t s = lev_sn[lo];
  const int item = (int)blockIdx.x - pre[lo] - lo;
  const int32_t f = M.first[s], nc = M.first[s + 1] - f, nr = M.nrow[s];
  const int64_t ld = nc + nr;
  const double *P = M.panels + M.pptr[s];
  const int tid = threadIdx.x;
  if (tid < nc) bs[tid] = B[f + (LU ? M.piv[f + tid] : tid)];
  __syncthreads();
  {
    const int i = tid & (NCMAX - 1), sl = tid / NCMAX; // row i of W, k slice [YS sl, YS sl + YS)
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
  }
  __syncthreads();
  if (tid < nc) ys[tid] = (LU ? bs[tid] : 0.0) + ((part[tid] + part[NCMAX + tid]) + (part[2 * NCMAX + tid] + part[3 * NCMAX + tid]));
  __syncthreads();
  if (item == 0) {
    if (tid < nc) Y[f + tid] = ys[tid];
    return;
  }
  const int r0 = (item - 1) * TILE, rl = tid & 63, sl = tid >> 6; // row rl of the tile, k slice [16 sl, 16 sl + 16)
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
    unsafeAtomicAdd(B + (M.rows + M.rptr[s])[r0 + tid], -sum);
  }
}
// out[k] += sum over 64 rows of a tile of block[r][k] x[rows[r]]: thread = (row rl, k slice of 16); blk: first row of the tile in the
// nrow x ncol block (leading dimension bld); results of the 64 rows are summed over the wavefront
__device__ __forceinline__ void s1_tile_tdot(const double *__restrict__ blk, int64_t bld, int32_t nc, int rn, int rl, int sl, double xr, double (&acc)[16])
{
  double w[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) w[u] = (rl < rn && 16 * sl + u < nc) ? blk[rl + (int64_t)(16 * sl + u) * bld] : 0.0;
#pragma unroll
  for (int u = 0; u < 16; ++u) acc[u] += w[u] * xr;
}
template <bool LU, int NCMAX>
__global__ __launch_bounds__(4 * NCMAX) void k_sn_bwd1_partial(Meta M, const int32_t *__restrict__ big_sn, const int32_t *__restrict__ pre, int cnt, const double *__restrict__ X,
                                                          double *__restrict__ partial)
{
  const int it = find_item(pre, cnt, (int32_t)blockIdx.x);
  const int32_t s = big_sn[it];
  const int item = (int)blockIdx.x - pre[it];
  const int32_t nc = M.first[s + 1] - M.first[s], nr = M.nrow[s];
  const double *blk = LU ? M.upanels + M.uptr[s] : M.panels + M.pptr[s] + nc;
  const int64_t bld = LU ? (int64_t)nr : (int64_t)nc + nr;
  const int32_t *R = M.rows + M.rptr[s];
  const int tid = threadIdx.x, rl = tid & 63, sl = tid >> 6;
  double acc[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) acc[u] = 0.0;
  for (int sub = 0; sub < BWD_ROWS / TILE; ++sub) {
    const int r0 = item * BWD_ROWS + sub * TILE, rn = min(TILE, nr - r0);
    if (rn <= 0) break;
    s1_tile_tdot(blk + r0, bld, nc, rn, rl, sl, rl < rn ? X[R[r0 + rl]] : 0.0, acc);
  }
  double *out = partial + (int64_t)blockIdx.x * SN_MAX_COLS;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const double v = s1_wave_sum(acc[u]);
    if (rl == 0 && 16 * sl + u < nc) out[16 * sl + u] = v;
  }
}
template <bool LU, int NCMAX>
__global__ __launch_bounds__(4 * NCMAX) void k_sn_bwd1_diag(Meta M, const int32_t *__restrict__ lev_sn, const int32_t *__restrict__ big_index, const int32_t *__restrict__ pre_big,
                                                       const double *__restrict__ partial, const double *__restrict__ Y, double *__restrict__ B)
{
  constexpr int XS = NCMAX / 4;
  __shared__ double t[NCMAX], part[4 * NCMAX];
  const int32_t s = lev_sn[blockIdx.x];
  const int32_t f = M.first[s], nc = M.first[s + 1] - f, nr = M.nrow[s];
  const int64_t ld = nc + nr;
  const double *P = M.panels + M.pptr[s];
  const int32_t *R = M.rows + M.rptr[s];
  const int tid = threadIdx.x, rl = tid & 63, sl = tid >> 6;
  const int bi = big_index[blockIdx.x];
  if (bi >= 0) {
    if (tid < nc) {
      double acc = Y[f + tid];
      const double *pp = partial + (int64_t)pre_big[bi] * SN_MAX_COLS + tid;
      const int npart = (nr + BWD_ROWS - 1) / BWD_ROWS;
      for (int q = 0; q < npart; ++q) acc -= pp[(int64_t)q * SN_MAX_COLS];
      t[tid] = acc;
    }
  } else {
    const double *blk = LU ? M.upanels + M.uptr[s] : P + nc;
    const int64_t bld = LU ? (int64_t)nr : ld;
    double acc[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u] = 0.0;
    for (int r0 = 0; r0 < nr; r0 += TILE) {
      const int rn = min(TILE, nr - r0);
      s1_tile_tdot(blk + r0, bld, nc, rn, rl, sl, rl < rn ? B[R[r0 + rl]] : 0.0, acc);
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const double v = s1_wave_sum(acc[u]);
      if (rl == 0 && 16 * sl + u < nc) t[16 * sl + u] = Y[f + 16 * sl + u] - v;
    }
  }
  __syncthreads();
  { // x = W^T t (Cholesky: x_i = sum_{k >= i} W[k][i] t_k) resp. U^-1 t (x_i = sum_{k >= i} U^-1[i][k] t_k): thread = (i, k slice of 32)
    const int i = tid & (NCMAX - 1), q = tid / NCMAX;
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
  }
  __syncthreads();
  if (tid < nc) B[f + tid] = (part[tid] + part[NCMAX + tid]) + (part[2 * NCMAX + tid] + part[3 * NCMAX + tid]);
}

// ---- host driver -------------------------------------------------------------------------------------------------------------------
struct Factor {
  int64_t n = 0, entries = 0;
  int32_t nsn = 0, nlev = 0;
  double flops = 0.0;
  Meta M{};
  std::vector<int32_t> h_perm;           // perm[new] = old (global)
  std::vector<int32_t> lev_ptr;          // [nlev + 1] into lev_sn
  std::vector<int32_t> lev_big_ptr;      // [nlev + 1] into big_sn
  std::vector<int32_t> lev_maxnc;        // widest supernode of the level
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
  int64_t max_big_tiles = 0;
  void release()
  {
    for (void *p : {(void *)d_first, (void *)d_nrow, (void *)d_rows, (void *)d_sn_of_col, (void *)d_iperm, (void *)d_perm, (void *)d_rptr, (void *)d_pptr, (void *)d_panels,
                    (void *)d_lev_sn, (void *)d_preT, (void *)d_preU, (void *)d_big_sn, (void *)d_big_index, (void *)d_preB, (void *)d_err, (void *)d_partial,
                    (void *)d_upanels, (void *)d_uptr, (void *)d_piv, (void *)d_preUF})
      if (p) (void)hipFree(p);
    d_first = d_nrow = d_rows = d_sn_of_col = d_iperm = d_perm = d_lev_sn = d_preT = d_preU = d_big_sn = d_big_index = d_preB = nullptr;
    d_rptr = d_pptr = nullptr;
    d_panels = d_partial = d_upanels = nullptr;
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

// symbolic results of all blocks -> one global structure on the device.  Returns false on an allocation failure.
static inline bool build(Factor &F, int64_t n, int64_t nblocks, const int64_t *block_ptr, std::vector<BlockSym> &BS, bool lu = false)
{
  F.n = n;
  F.lu = lu;
  std::vector<int32_t> first, nrow, rows, sn_of_col((size_t)n), iperm((size_t)n), level;
  std::vector<int64_t> rptr(1, 0), pptr(1, 0);
  F.h_perm.resize((size_t)n);
  for (int64_t b = 0; b < nblocks; ++b) {
    BlockSym &S = BS[(size_t)b];
    const int32_t off = (int32_t)block_ptr[b];
    const int32_t nsn = (int32_t)S.first.size() - 1;
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
  std::vector<int32_t> preT((size_t)F.nsn + nlev), preU((size_t)F.nsn + nlev), preUF((size_t)F.nsn + nlev), big_sn, big_index((size_t)F.nsn, -1), preB;
  std::vector<int64_t> uptr(1, 0);
  for (int32_t s = 0; s < F.nsn; ++s) uptr.push_back(uptr.back() + (int64_t)nrow[(size_t)s] * (first[(size_t)s + 1] - first[(size_t)s]));
  F.uentries = lu ? uptr.back() : 0;
  F.h_tilesUF.assign((size_t)nlev, 0);
  F.h_tilesT.assign((size_t)nlev, 0);
  F.h_tilesU.assign((size_t)nlev, 0);
  F.h_tilesB.assign((size_t)nlev, 0);
  F.lev_maxnc.assign((size_t)nlev, 0);
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
  F.M.nsn = F.nsn;
  F.M.first = F.d_first;
  F.M.nrow = F.d_nrow;
  F.M.rptr = F.d_rptr;
  F.M.rows = F.d_rows;
  F.M.pptr = F.d_pptr;
  F.M.sn_of_col = F.d_sn_of_col;
  F.M.panels = F.d_panels;
  F.M.upanels = F.d_upanels;
  F.M.uptr = F.d_uptr;
  F.M.piv = F.d_piv;
  return true;
}

// numeric factorisation of the matrix (device CSR, original numbering); *bad = supernode + 1 whose diagonal block was not positive definite
static inline hipError_t factorize(Factor &F, hipStream_t st, const int64_t *d_rp, const int32_t *d_ci, const double *d_va, unsigned *bad)
{
  hipError_t e = hipMemsetAsync(F.d_panels, 0, sizeof(double) * (size_t)std::max<int64_t>(F.entries, 1), st);
  if (e != hipSuccess) return e;
  (void)hipMemsetAsync(F.d_err, 0, 4, st);
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
      hipLaunchKernelGGL(k_sn_lu_diag, dim3((unsigned)cnt), dim3(256), (size_t)(nc | 1) * nc * 8, st, F.M, lsn, F.d_err);
      if (F.h_tilesT[(size_t)l] > 0)
        hipLaunchKernelGGL(k_sn_lu_panel, dim3((unsigned)F.h_tilesT[(size_t)l]), dim3(256), 0, st, F.M, lsn, (const int32_t *)(F.d_preT + F.lev_ptr[(size_t)l] + l), cnt);
      if (F.h_tilesUF[(size_t)l] > 0)
        hipLaunchKernelGGL(k_sn_lu_update, dim3((unsigned)F.h_tilesUF[(size_t)l]), dim3(256), 0, st, F.M, lsn, (const int32_t *)(F.d_preUF + F.lev_ptr[(size_t)l] + l), cnt);
      continue;
    }
    hipLaunchKernelGGL(k_sn_diag, dim3((unsigned)cnt), dim3(256), (size_t)(nc | 1) * nc * 8, st, F.M, lsn, F.d_err);
    if (F.h_tilesT[(size_t)l] > 0)
      hipLaunchKernelGGL(k_sn_panel, dim3((unsigned)F.h_tilesT[(size_t)l]), dim3(256), 0, st, F.M, lsn, (const int32_t *)(F.d_preT + F.lev_ptr[(size_t)l] + l), cnt);
    if (F.h_tilesU[(size_t)l] > 0)
      hipLaunchKernelGGL(k_sn_update, dim3((unsigned)F.h_tilesU[(size_t)l]), dim3(256), 0, st, F.M, lsn, (const int32_t *)(F.d_preU + F.lev_ptr[(size_t)l] + l), cnt);
  }
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(st);
  if (e != hipSuccess) return e;
  return hipMemcpy(bad, F.d_err, 4, hipMemcpyDeviceToHost);
}

// scratch of the backward sweep for m right-hand sides (call OUTSIDE a stream capture)
static inline bool reserve(Factor &F, int m)
{
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
static inline void solve_t(const Factor &F, hipStream_t st, int m, double *B, int64_t ldb, double *Yvec)
{
  if (m == 1 && ldb == 1 && Yvec) { // the single-vector kernels: three launches per level
    for (int32_t l = 0; l < F.nlev; ++l) {
      const int32_t cnt = F.lev_ptr[(size_t)l + 1] - F.lev_ptr[(size_t)l];
      if (cnt == 0) continue;
      const int32_t *lsn = F.d_lev_sn + F.lev_ptr[(size_t)l], *preT = F.d_preT + F.lev_ptr[(size_t)l] + l;
      const unsigned grid = (unsigned)(F.h_tilesT[(size_t)l] + cnt);
      if (F.lev_maxnc[(size_t)l] <= 64) hipLaunchKernelGGL((k_sn_fwd1<LU, 64>), dim3(grid), dim3(256), 0, st, F.M, lsn, preT, cnt, B, Yvec);
      else hipLaunchKernelGGL((k_sn_fwd1<LU, 128>), dim3(grid), dim3(512), 0, st, F.M, lsn, preT, cnt, B, Yvec);
    }
    for (int32_t l = F.nlev - 1; l >= 0; --l) {
      const int32_t cnt = F.lev_ptr[(size_t)l + 1] - F.lev_ptr[(size_t)l];
      if (cnt == 0) continue;
      const int32_t *lsn = F.d_lev_sn + F.lev_ptr[(size_t)l];
      const int32_t nbig = F.lev_big_ptr[(size_t)l + 1] - F.lev_big_ptr[(size_t)l];
      const int32_t *preB = F.d_preB + F.lev_big_ptr[(size_t)l] + l, *bsn = F.d_big_sn + F.lev_big_ptr[(size_t)l], *bidx = F.d_big_index + F.lev_ptr[(size_t)l];
      const bool small = F.lev_maxnc[(size_t)l] <= 64;
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
    if (F.h_tilesT[(size_t)l] > 0)
      hipLaunchKernelGGL(k_sn_fwd_update, dim3((unsigned)F.h_tilesT[(size_t)l]), dim3(256), lds, st, F.M, lsn, (const int32_t *)(F.d_preT + F.lev_ptr[(size_t)l] + l), cnt, m, B, ldb);
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
static inline void solve(const Factor &F, hipStream_t st, int m, double *B, int64_t ldb, double *Yvec = nullptr)
{
  if (F.lu) solve_t<true>(F, st, m, B, ldb, Yvec);
  else solve_t<false>(F, st, m, B, ldb, Yvec);
}

} // namespace sn
