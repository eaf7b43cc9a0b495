// Device kernels of the two-level Schwarz hot path (gfx950 / CDNA4, wave64).
// All kernels are HBM-bandwidth or dependency-latency bound FP64 streaming kernels; none is
// GEMM-shaped, so no MFMA here (DESIGN.md section "Kernels").  Launch wrappers are at the
// bottom; every wrapper enqueues on the given stream and never synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ddm {

constexpr int WG = 256;          // 4 waves of 64
constexpr int SPMV_NNZ = 2048;   // non-zeros staged in LDS per workgroup (16 KiB of products)
constexpr int RED_MAX_BLOCKS = 1024;

// bijective XCD-aware remap (cdna_hip_programming.md T1): consecutive work items of one XCD
// become contiguous, so each XCD's L2 sees one contiguous slab of the matrix / x vector.
__device__ __forceinline__ int xcd_remap(int orig, int nwg)
{
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// block-wide sum in a fixed order; result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double *lds4)
{
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) lds4[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) r = (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
  __syncthreads();
  return r;
}

// ---------------------------------------------------------------------------------------------
// K1 / K11: CSR SpMV, "CSR-stream": a workgroup owns a block of consecutive rows holding at most
// SPMV_NNZ non-zeros; values / column indices are read fully coalesced, the products are staged
// in LDS and each row is then summed sequentially in column order (deterministic).
// reference: BCRSMatrix::mv / usmv at nonoverlapping_operator.hh:37,47; spectra.hh:100-105.
template <bool ACC>
__global__ __launch_bounds__(WG) void k_spmv_stream(const int64_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                     const double *__restrict__ va, const int32_t *__restrict__ blk_row,
                                                     int nblk, const double *__restrict__ x, double *__restrict__ y,
                                                     double alpha)
{
  __shared__ double prod[SPMV_NNZ];
  __shared__ double red[4];
  const int b = xcd_remap(blockIdx.x, nblk);
  const int r0 = blk_row[b], r1 = blk_row[b + 1];
  const int64_t z0 = rp[r0], z1 = rp[r1];
  const int64_t nz = z1 - z0;
  if (nz > SPMV_NNZ) { // a single long row: strided partial sums + block reduction
    double s = 0.0;
    for (int64_t k = z0 + threadIdx.x; k < z1; k += WG) s += va[k] * x[ci[k]];
    s = block_sum(s, red);
    if (threadIdx.x == 0) y[r0] = ACC ? y[r0] + alpha * s : s;
    return;
  }
  // all loads of a thread are issued before the first one is needed: indices past the block's end are clamped (loaded, not
  // stored) instead of branched around, so that nothing serialises the SPMV_NNZ / WG independent load -> gather chains
  constexpr int U = SPMV_NNZ / WG;
  const int last = (int)nz - 1;
  if (last >= 0) {
    int32_t c[U];
    double v[U], xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) c[u] = ci[z0 + min((int)threadIdx.x + u * WG, last)];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = va[z0 + min((int)threadIdx.x + u * WG, last)];
#pragma unroll
    for (int u = 0; u < U; ++u) xv[u] = x[c[u]];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = threadIdx.x + u * WG;
      if (k <= last) prod[k] = v[u] * xv[u];
    }
  }
  __syncthreads();
  const int r = r0 + threadIdx.x;
  if (r < r1) {
    const int k0 = (int)(rp[r] - z0), k1 = (int)(rp[r + 1] - z0);
    double s = 0.0;
    for (int kb = k0; kb < k1; kb += 8) { // eight LDS reads in flight, summed in column order
      double q[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = prod[min(kb + j, k1 - 1)];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (kb + j < k1) s += q[j];
    }
    y[r] = ACC ? y[r] + alpha * s : s;
  }
}

// ---------------------------------------------------------------------------------------------
// K3: level-scheduled sparse triangular solves with the ILU(0) factors (schwarz.hh:133).
// Rows of one level are independent; their entries are stored level-by-level in sliced-ELL
// (column-major inside the level) so that a thread-per-row kernel reads them coalesced and sums
// them in the same column order as the sequential back-solve.
// One row of a level: all column indices / values of the row are loaded first, then all x
// gathers are issued together (memory-level parallelism: two dependent round trips per level
// instead of two per entry), then the products are subtracted in column order.
constexpr int TRSV_UNROLL = 16;
template <class IDX>
__device__ __forceinline__ double trsv_row_sum(double s, int w, IDX m, IDX r, const int32_t *__restrict__ cols,
                                               const double *__restrict__ vals, const double *x)
{
  for (int k0 = 0; k0 < w; k0 += TRSV_UNROLL) {
    int32_t c[TRSV_UNROLL];
    double v[TRSV_UNROLL], xv[TRSV_UNROLL];
#pragma unroll
    for (int u = 0; u < TRSV_UNROLL; ++u) {
      const bool ok = k0 + u < w;
      c[u] = ok ? cols[(IDX)(k0 + u) * m + r] : -1;
      v[u] = ok ? vals[(IDX)(k0 + u) * m + r] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < TRSV_UNROLL; ++u) xv[u] = c[u] >= 0 ? x[c[u]] : 0.0;
#pragma unroll
    for (int u = 0; u < TRSV_UNROLL; ++u) s -= v[u] * xv[u];
  }
  return s;
}

__global__ __launch_bounds__(WG) void k_trsv_lower_level(int m, int w, const int32_t *__restrict__ rows,
                                                          const int32_t *__restrict__ cols, const double *__restrict__ vals,
                                                          const double *__restrict__ d, double *x)
{
  const int r = blockIdx.x * WG + threadIdx.x;
  if (r >= m) return;
  const int row = rows[r];
  x[row] = trsv_row_sum<int64_t>(d[row], w, m, r, cols, vals, x);
}

__global__ __launch_bounds__(WG) void k_trsv_upper_level(int m, int w, const int32_t *__restrict__ rows,
                                                          const int32_t *__restrict__ cols, const double *__restrict__ vals,
                                                          const double *__restrict__ dinv, double *x)
{
  const int r = blockIdx.x * WG + threadIdx.x;
  if (r >= m) return;
  const int row = rows[r];
  x[row] = trsv_row_sum<int64_t>(x[row], w, m, r, cols, vals, x) * dinv[r];
}

// Multi-right-hand-side variants (GenEO setup: block eigensolver; cf. the reference's own
// multi-RHS triangular solve, eigensolvers/umfpack.hh:131-197).  Block vectors are row-major
// n x nrhs, a thread owns one (row, rhs) pair; the nrhs lanes of a row read the same factor
// entries (broadcast) and gather nrhs consecutive doubles of x (coalesced).
template <class IDX>
__device__ __forceinline__ double trsv_row_sum_multi(double s, int w, IDX m, IDX r, int64_t nrhs, int j,
                                                     const int32_t *__restrict__ cols, const double *__restrict__ vals,
                                                     const double *x)
{
  for (int k0 = 0; k0 < w; k0 += TRSV_UNROLL) {
    int32_t c[TRSV_UNROLL];
    double v[TRSV_UNROLL], xv[TRSV_UNROLL];
#pragma unroll
    for (int u = 0; u < TRSV_UNROLL; ++u) {
      const bool ok = k0 + u < w;
      c[u] = ok ? cols[(IDX)(k0 + u) * m + r] : -1;
      v[u] = ok ? vals[(IDX)(k0 + u) * m + r] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < TRSV_UNROLL; ++u) xv[u] = c[u] >= 0 ? x[(int64_t)c[u] * nrhs + j] : 0.0;
#pragma unroll
    for (int u = 0; u < TRSV_UNROLL; ++u) s -= v[u] * xv[u];
  }
  return s;
}

// ldd / ldx: leading dimensions of the row-major blocks d and x (>= nrhs)
template <bool UPPER>
__global__ __launch_bounds__(WG) void k_trsv_level_multi(int m, int w, int nrhs, const int32_t *__restrict__ rows,
                                                          const int32_t *__restrict__ cols, const double *__restrict__ vals,
                                                          const double *__restrict__ dinv, const double *__restrict__ d, int64_t ldd, double *x, int64_t ldx)
{
  const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x;
  const int r = (int)(t / nrhs);
  if (r >= m) return;
  const int j = (int)(t - (int64_t)r * nrhs);
  const int64_t o = (int64_t)rows[r] * ldx + j;
  const double s = trsv_row_sum_multi<int64_t>(UPPER ? x[o] : d[(int64_t)rows[r] * ldd + j], w, m, r, ldx, j, cols, vals, x);
  x[o] = UPPER ? s * dinv[r] : s;
}
// nrhs % 4 == 0 and 32-byte aligned rows: one thread per (row, group of 4 right-hand sides) -- a quarter of the index / value /
// gather instructions and wavefronts per level (the 24-column solves of the block eigensolver move 27 GB of gathered x rows per
// triangle through the caches at 216^3: the large levels are throughput-bound).  Same sums in the same order as the scalar kernel.
template <bool UPPER>
__global__ __launch_bounds__(WG) void k_trsv_level_multi4(int m, int w, int nq /* nrhs / 4 */, const int32_t *__restrict__ rows,
                                                           const int32_t *__restrict__ cols, const double *__restrict__ vals,
                                                           const double *__restrict__ dinv, const double *__restrict__ d, int64_t ldd, double *x, int64_t ldx)
{
  typedef double d4 __attribute__((ext_vector_type(4)));
  const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x;
  const int r = (int)(t / nq);
  if (r >= m) return;
  const int j = 4 * (int)(t - (int64_t)r * nq);
  const int64_t o = (int64_t)rows[r] * ldx + j;
  d4 s = UPPER ? *reinterpret_cast<const d4 *>(x + o) : *reinterpret_cast<const d4 *>(d + (int64_t)rows[r] * ldd + j);
  for (int k0 = 0; k0 < w; k0 += TRSV_UNROLL) {
    int32_t c[TRSV_UNROLL];
    double v[TRSV_UNROLL];
    d4 xv[TRSV_UNROLL];
#pragma unroll
    for (int u = 0; u < TRSV_UNROLL; ++u) {
      const bool ok = k0 + u < w;
      c[u] = ok ? cols[(int64_t)(k0 + u) * m + r] : -1;
      v[u] = ok ? vals[(int64_t)(k0 + u) * m + r] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < TRSV_UNROLL; ++u) xv[u] = c[u] >= 0 ? *reinterpret_cast<const d4 *>(x + (int64_t)c[u] * ldx + j) : d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < TRSV_UNROLL; ++u) s -= v[u] * xv[u];
  }
  if (UPPER) s *= dinv[r];
  *reinterpret_cast<d4 *>(x + o) = s;
}
// SINGLE-PRECISION sweeps for uses where the triangular solve is a PRECONDITIONER inside an iteration that only needs a fixed, good
// search direction (the block eigensolver of the GenEO setup: W = T r): factor entries and the work block in float -- half the bytes of
// the gathered x rows, which is what the 24-column level solves are bound by -- right-hand side read and result written in double.
// One thread per (row, 4 right-hand sides); xf: n x nrhs floats (ld = nrhs).
__global__ void k_to_float(int64_t n, const double *__restrict__ src, float *__restrict__ dst)
{
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t < n) dst[t] = (float)src[t];
}
template <bool UPPER>
__global__ __launch_bounds__(WG) void k_trsv_level_multi4_f32(int m, int w, int nq /* nrhs / 4 */, const int32_t *__restrict__ rows, const int32_t *__restrict__ cols,
                                                               const float *__restrict__ vals, const float *__restrict__ dinv, const double *__restrict__ d, int64_t ldd,
                                                               float *xf, int64_t ldf, double *__restrict__ xout, int64_t ldx)
{
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef double d4 __attribute__((ext_vector_type(4)));
  const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x;
  const int r = (int)(t / nq);
  if (r >= m) return;
  const int j = 4 * (int)(t - (int64_t)r * nq);
  const int64_t row = rows[r], o = row * ldf + j;
  f4 s;
  if (UPPER) s = *reinterpret_cast<const f4 *>(xf + o);
  else {
    const d4 dv = *reinterpret_cast<const d4 *>(d + row * ldd + j);
    s = f4{(float)dv[0], (float)dv[1], (float)dv[2], (float)dv[3]};
  }
  for (int k0 = 0; k0 < w; k0 += TRSV_UNROLL) {
    int32_t c[TRSV_UNROLL];
    float v[TRSV_UNROLL];
    f4 xv[TRSV_UNROLL];
#pragma unroll
    for (int u = 0; u < TRSV_UNROLL; ++u) {
      const bool ok = k0 + u < w;
      c[u] = ok ? cols[(int64_t)(k0 + u) * m + r] : -1;
      v[u] = ok ? vals[(int64_t)(k0 + u) * m + r] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < TRSV_UNROLL; ++u) xv[u] = c[u] >= 0 ? *reinterpret_cast<const f4 *>(xf + (int64_t)c[u] * ldf + j) : f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int u = 0; u < TRSV_UNROLL; ++u) s -= v[u] * xv[u];
  }
  if (UPPER) {
    s *= dinv[r];
    *reinterpret_cast<d4 *>(xout + row * ldx + j) = d4{(double)s[0], (double)s[1], (double)s[2], (double)s[3]};
  }
  *reinterpret_cast<f4 *>(xf + o) = s;
}
// The same for levels of WIDE rows (factors of the sparse direct solver: separator rows have thousands of entries and a
// level often holds a single row): one workgroup per row, the 256 threads split the row's entries into 256 / nrhs slices
// per right-hand side, slice sums meet in LDS and are added in slice order (deterministic).
template <bool UPPER>
__global__ __launch_bounds__(WG) void k_trsv_level_multi_wide(int m, int w, int nrhs, const int32_t *__restrict__ rows,
                                                               const int32_t *__restrict__ cols, const double *__restrict__ vals,
                                                               const double *__restrict__ dinv, const double *__restrict__ d, int64_t ldd, double *x, int64_t ldx)
{
  __shared__ double part[WG];
  const int r = blockIdx.x;
  const int nsl = WG / nrhs;
  const int sl = threadIdx.x / nrhs, j = threadIdx.x - sl * nrhs;
  double s = 0.0;
  if (sl < nsl)
    for (int k = sl; k < w; k += nsl) {
      const double v = vals[(int64_t)k * m + r];
      s += v * x[(int64_t)cols[(int64_t)k * m + r] * ldx + j];
    }
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < nrhs) {
    const int64_t o = (int64_t)rows[r] * ldx + threadIdx.x;
    double acc = UPPER ? x[o] : d[(int64_t)rows[r] * ldd + threadIdx.x];
    for (int q = 0; q < nsl; ++q) acc -= part[q * nrhs + threadIdx.x];
    x[o] = UPPER ? acc * dinv[r] : acc;
  }
}

// Y = A X for row-major block vectors (n x nrhs): one thread per (row, rhs)
__global__ __launch_bounds__(WG) void k_spmm_rowmajor(int64_t n, int nrhs, const int64_t *__restrict__ rp,
                                                       const int32_t *__restrict__ ci, const double *__restrict__ va,
                                                       const double *__restrict__ x, int64_t ldx, double *__restrict__ y, int64_t ldy)
{
  const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x;
  const int64_t row = t / nrhs;
  if (row >= n) return;
  const int j = (int)(t - row * nrhs);
  const int64_t k0 = rp[row], k1 = rp[row + 1];
  double s = 0.0;
  for (int64_t kb = k0; kb < k1; kb += 8) {
    int32_t c[8];
    double v[8], xv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool ok = kb + u < k1;
      c[u] = ok ? ci[kb + u] : -1;
      v[u] = ok ? va[kb + u] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) xv[u] = c[u] >= 0 ? x[(int64_t)c[u] * ldx + j] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u] * xv[u];
  }
  y[row * ldy + j] = s;
}

// The same for nrhs % 4 == 0 and 32-byte aligned rows: one thread per (row, group of 4 columns) -- a quarter of the index / value /
// gather instructions per product; optionally TWO matrices on one pattern (va2, y2: the pencil matrices A~ and C~ of the block
// eigensolver share their pattern), which reads the block X once for both products.  Row sums in entry order, as above.
template <bool TWO>
__global__ __launch_bounds__(WG) void k_spmm_rowmajor4(int64_t n, int nq /* nrhs / 4 */, const int64_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                        const double *__restrict__ va, const double *__restrict__ va2, const double *__restrict__ x, int64_t ldx,
                                                        double *__restrict__ y, double *__restrict__ y2, int64_t ldy)
{
  typedef double d4 __attribute__((ext_vector_type(4)));
  // (An XCD-contiguous block order was measured in round 3 and changes nothing: 29.7 GB of L2 fills per call either way, 8.7 ms.  The
  // misses are not the eight L2s duplicating each other but the reuse distance itself -- two grid planes of X rows, 8 MB with the
  // 576-byte row stride of the [X | W | P] array, against 4 MB of L2: every row of X is fetched once per plane.  What would help is a
  // cache-blocked ROW ORDER (bricks instead of planes), i.e. a host-side permutation of the processing order.)
  const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x;
  const int64_t row = t / nq;
  if (row >= n) return;
  const int j = 4 * (int)(t - row * nq);
  const int64_t k0 = rp[row], k1 = rp[row + 1];
  d4 s = {0.0, 0.0, 0.0, 0.0}, s2 = {0.0, 0.0, 0.0, 0.0};
  for (int64_t kb = k0; kb < k1; kb += 4) {
    int32_t c[4];
    double v[4], w[4];
    d4 xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool ok = kb + u < k1;
      c[u] = ok ? ci[kb + u] : -1;
      v[u] = ok ? va[kb + u] : 0.0;
      if (TWO) w[u] = ok ? va2[kb + u] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) xv[u] = c[u] >= 0 ? *reinterpret_cast<const d4 *>(x + (int64_t)c[u] * ldx + j) : d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s += v[u] * xv[u];
      if (TWO) s2 += w[u] * xv[u];
    }
  }
  *reinterpret_cast<d4 *>(y + row * ldy + j) = s;
  if (TWO) *reinterpret_cast<d4 *>(y2 + row * ldy + j) = s2;
}

// The same product with the rows processed in a CACHE-BLOCKED order (order[]: a permutation of the rows, csr_row_order_tiled): rows of
// a 16 x 4 x 4 brick of a structured grid are neighbours in order[], a workgroup takes 64 of them (blockDim = 64 nq threads), and the
// workgroups of one XCD take a contiguous range of bricks (xcd_remap) -- the 27 gathered X rows of a row are then shared inside the
// brick and with the bricks running beside it on the same L2, instead of being fetched once per grid plane (29.7 GB of L2 fills per
// call for 12 GB of algorithmic traffic with the natural order at 216^3).  Every row is computed as before: same sums, same order.
template <bool TWO>
__global__ __launch_bounds__(512) void k_spmm_rowmajor4_tiled(int64_t n, int nq /* nrhs / 4 */, const int32_t *__restrict__ order, const int64_t *__restrict__ rp,
                                                               const int32_t *__restrict__ ci, const double *__restrict__ va, const double *__restrict__ va2,
                                                               const double *__restrict__ x, int64_t ldx, double *__restrict__ y, double *__restrict__ y2, int64_t ldy)
{
  typedef double d4 __attribute__((ext_vector_type(4)));
  const int rows_per_wg = blockDim.x / nq;
  const int64_t wg = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int lr = threadIdx.x / nq;
  const int64_t pos = wg * rows_per_wg + lr;
  if (lr >= rows_per_wg || pos >= n) return;
  const int64_t row = order[pos];
  const int j = 4 * (threadIdx.x - lr * nq);
  const int64_t k0 = rp[row], k1 = rp[row + 1];
  d4 s = {0.0, 0.0, 0.0, 0.0}, s2 = {0.0, 0.0, 0.0, 0.0};
  constexpr int CH = 8; // entries whose gathers are in flight together (the kernel is bound by these dependent gathers, not by bytes)
  for (int64_t kb = k0; kb < k1; kb += CH) {
    int32_t c[CH];
    double v[CH], w[CH];
    d4 xv[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const bool ok = kb + u < k1;
      c[u] = ok ? ci[kb + u] : -1;
      v[u] = ok ? va[kb + u] : 0.0;
      if (TWO) w[u] = ok ? va2[kb + u] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) xv[u] = c[u] >= 0 ? *reinterpret_cast<const d4 *>(x + (int64_t)c[u] * ldx + j) : d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      s += v[u] * xv[u];
      if (TWO) s2 += w[u] * xv[u];
    }
  }
  *reinterpret_cast<d4 *>(y + row * ldy + j) = s;
  if (TWO) *reinterpret_cast<d4 *>(y2 + row * ldy + j) = s2;
}

// Several consecutive small levels (each <= WG*TRSV_SMALL_ROWS rows) in ONE workgroup: the levels
// are separated by workgroup barriers instead of kernel boundaries.  desc[l] = {m, w, row_off,
// ent_off}.  All data of these levels is produced and consumed by this workgroup only.
struct LevelDesc {
  int32_t m, w;
  int64_t row_off, ent_off;
};
constexpr int TRSV_SMALL_WG = 1024;

template <bool UPPER>
__global__ __launch_bounds__(TRSV_SMALL_WG) void k_trsv_small_levels(int nlev, const LevelDesc *__restrict__ desc,
                                                                      const int32_t *__restrict__ rows,
                                                                      const int32_t *__restrict__ cols,
                                                                      const double *__restrict__ vals,
                                                                      const double *__restrict__ dinv,
                                                                      const double *__restrict__ d, double *x)
{
  for (int l = 0; l < nlev; ++l) {
    const LevelDesc L = desc[l];
    if (L.w >= 32 && 2 * L.m <= TRSV_SMALL_WG) {
      // few wide rows (separator rows of a sparse direct factor: thousands of entries, often one row per level): S lanes
      // of one wavefront share a row, each sums every S-th product, a butterfly adds the S partial sums (fixed order)
      int S = 64;
      while (S * L.m > TRSV_SMALL_WG) S >>= 1;
      const int r = threadIdx.x / S, sl = threadIdx.x - r * S;
      const bool act = r < L.m;
      double s = 0.0;
      if (act) {
        const int32_t *cl = cols + L.ent_off;
        const double *vl = vals + L.ent_off;
        for (int k = sl; k < L.w; k += S) s += vl[(int64_t)k * L.m + r] * x[cl[(int64_t)k * L.m + r]];
      }
      for (int o = S >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if (act && sl == 0) {
        const int row = rows[L.row_off + r];
        const double t = (UPPER ? x[row] : d[row]) - s;
        x[row] = UPPER ? t * dinv[L.row_off + r] : t;
      }
    } else {
      for (int r = threadIdx.x; r < L.m; r += TRSV_SMALL_WG) {
        const int row = rows[L.row_off + r];
        const double s = trsv_row_sum<int>(UPPER ? x[row] : d[row], L.w, L.m, r, cols + L.ent_off, vals + L.ent_off, x);
        x[row] = UPPER ? s * dinv[L.row_off + r] : s;
      }
    }
    __syncthreads(); // workgroup-scope release/acquire of the x entries just written
  }
}

// ---------------------------------------------------------------------------------------------
// Level-scheduled triangular solves for factors of the SPARSE DIRECT solver (csrc/sparse_chol_host.hpp).  Their rows differ from
// ILU(0) rows: a level near the root of the elimination tree holds a handful of rows with thousands of entries, a leaf level
// thousands of short rows, and there are thousands of levels -- sliced ELL would pad every level to its longest row.  Layout:
// rows sorted by level, entries of a row contiguous (CSR in level order: lrp); S lanes (a power of two <= 64, chosen per level
// from its row count) share a row, each sums every S-th product, a butterfly adds the S partial sums in a fixed order.
struct CsrLevel {
  int32_t m, S;
  int64_t row_off;
};
template <bool UPPER>
__device__ __forceinline__ void trsv_csr_rows(const CsrLevel L, int first_group, int ngroups_step, int tid, const int32_t *__restrict__ rows,
                                              const int32_t *__restrict__ rhs, const int64_t *__restrict__ lrp, const int32_t *__restrict__ cols, const double *__restrict__ vals,
                                              const double *__restrict__ dinv, const double *__restrict__ d, double *x)
{
  const int S = L.S;
  const int g = tid / S, sl = tid - g * S;
  for (int base = first_group; base < L.m; base += ngroups_step) { // every lane of a wavefront runs the same number of rounds
    const int r = base + g;
    const bool act = r < L.m;
    double s = 0.0;
    if (act) {
      const int64_t k1 = lrp[L.row_off + r + 1];
      // eight entries of the lane in flight: all column and value loads, then all gathers, then the products (in order)
      for (int64_t k = lrp[L.row_off + r] + sl; k < k1; k += 8 * (int64_t)S) {
        int32_t c[8];
        double v[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int64_t kk = k + (int64_t)u * S;
          const bool ok = kk < k1;
          c[u] = ok ? cols[kk] : -1;
          v[u] = ok ? vals[kk] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) xv[u] = c[u] >= 0 ? x[c[u]] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u] * xv[u];
      }
    }
    for (int o = S >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (act && sl == 0) { // rhs: index of the right-hand side (lower sweep: in d; upper sweep: in x, the forward result) or -1
      const int ri = rhs[L.row_off + r];
      const double t = (ri >= 0 ? (UPPER ? x[ri] : d[ri]) : 0.0) - s;
      x[rows[L.row_off + r]] = UPPER ? t * dinv[L.row_off + r] : t;
    }
  }
}
// a run of consecutive small levels in ONE workgroup (levels separated by workgroup barriers)
template <bool UPPER>
__global__ __launch_bounds__(TRSV_SMALL_WG) void k_trsv_csr_fused(int nlev, const CsrLevel *__restrict__ desc, const int32_t *__restrict__ rows,
                                                                   const int32_t *__restrict__ rhs, const int64_t *__restrict__ lrp, const int32_t *__restrict__ cols,
                                                                   const double *__restrict__ vals, const double *__restrict__ dinv,
                                                                   const double *__restrict__ d, double *x)
{
  for (int l = 0; l < nlev; ++l) {
    const CsrLevel L = desc[l];
    trsv_csr_rows<UPPER>(L, 0, TRSV_SMALL_WG / L.S, threadIdx.x, rows, rhs, lrp, cols, vals, dinv, d, x);
    __syncthreads();
  }
}
// one workgroup per independent diagonal block (subdomain): the whole triangular solve of the block, level after level
template <bool UPPER>
__global__ __launch_bounds__(TRSV_SMALL_WG) void k_trsv_csr_blocks(const int32_t *__restrict__ blk_lev_ptr, const CsrLevel *__restrict__ desc,
                                                                    const int32_t *__restrict__ rows, const int32_t *__restrict__ rhs, const int64_t *__restrict__ lrp,
                                                                    const int32_t *__restrict__ cols, const double *__restrict__ vals,
                                                                    const double *__restrict__ dinv, const double *__restrict__ d, double *x)
{
  const int l1 = blk_lev_ptr[blockIdx.x + 1];
  for (int l = blk_lev_ptr[blockIdx.x]; l < l1; ++l) {
    const CsrLevel L = desc[l];
    trsv_csr_rows<UPPER>(L, 0, TRSV_SMALL_WG / L.S, threadIdx.x, rows, rhs, lrp, cols, vals, dinv, d, x);
    __syncthreads();
  }
}
// one large level over the whole grid
template <bool UPPER>
__global__ __launch_bounds__(WG) void k_trsv_csr_level(CsrLevel L, const int32_t *__restrict__ rows, const int32_t *__restrict__ rhs, const int64_t *__restrict__ lrp,
                                                       const int32_t *__restrict__ cols, const double *__restrict__ vals,
                                                       const double *__restrict__ dinv, const double *__restrict__ d, double *x)
{
  const int gpb = WG / L.S;
  trsv_csr_rows<UPPER>(L, blockIdx.x * gpb, gridDim.x * gpb, threadIdx.x, rows, rhs, lrp, cols, vals, dinv, d, x);
}
// multi-right-hand-side level (row-major n x nrhs blocks): S slices x nrhs columns of threads per row, slice sums meet in LDS
template <bool UPPER>
__global__ __launch_bounds__(WG) void k_trsv_csr_level_multi(CsrLevel L, int S, int nrhs, const int32_t *__restrict__ rows, const int32_t *__restrict__ rhs, const int64_t *__restrict__ lrp,
                                                             const int32_t *__restrict__ cols, const double *__restrict__ vals,
                                                             const double *__restrict__ dinv, const double *__restrict__ d, int64_t ldd, double *x, int64_t ldx)
{
  __shared__ double part[WG];
  const int per_row = S * nrhs, rpb = WG / per_row;
  const int g = threadIdx.x / per_row, u = threadIdx.x - g * per_row;
  const int sl = u / nrhs, j = u - sl * nrhs;
  const int r = blockIdx.x * rpb + g;
  const bool act = g < rpb && r < L.m;
  double s = 0.0;
  if (act) {
    const int64_t k1 = lrp[L.row_off + r + 1];
    for (int64_t k = lrp[L.row_off + r] + sl; k < k1; k += 8 * (int64_t)S) {
      int32_t c[8];
      double v[8], xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t kk = k + (int64_t)u * S;
        const bool ok = kk < k1;
        c[u] = ok ? cols[kk] : -1;
        v[u] = ok ? vals[kk] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) xv[u] = c[u] >= 0 ? x[(int64_t)c[u] * ldx + j] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u] * xv[u];
    }
  }
  part[threadIdx.x] = s;
  __syncthreads();
  if (act && sl == 0) {
    const int ri = rhs[L.row_off + r];
    double t = ri >= 0 ? (UPPER ? x[(int64_t)ri * ldx + j] : d[(int64_t)ri * ldd + j]) : 0.0;
    for (int q = 0; q < S; ++q) t -= part[g * per_row + q * nrhs + j];
    x[(int64_t)rows[L.row_off + r] * ldx + j] = UPPER ? t * dinv[L.row_off + r] : t;
  }
}

// ---------------------------------------------------------------------------------------------
// Coherent accesses for hand-overs between waves inside one launch (cdna_hip_programming.md Guideline 16):
// sc1 loads bypass the non-coherent L1, sc1 stores are written through.
__device__ __forceinline__ double ld_sc1(const double *p)
{
  return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_sc1(double *p, double v)
{
  __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
// XCD-local placement shared by the single-launch engines (xcd2 fallback, pipe).  Measured on MI355X
// (tools/xcd_handoff_bench.hip): a hand-off between waves costs 0.9 us when producers and consumers share one XCD
// (plain stores stay in that XCD's L2, consumers read them with sc1 loads that bypass only L1) against 2.0 us across
// XCDs (write-through) and 7-9 us for a kernel boundary or a counter barrier.  Each independent diagonal block
// (= subdomain) is therefore solved by the waves of ONE XCD: every workgroup reads its XCD from HW_REG_XCC_ID and
// draws a ticket on that XCD; after a grid barrier it knows how many workgroups its XCD has and works only on the
// groups owned by its XCD (group % 8 == xcc).  If some XCD that owns a group received no workgroup, all take the
// placement-independent path (write-through stores).  epoch comes from a device word bumped by
// k_trsv_xcd_prologue, so flags never need zeroing.
constexpr int TRSV_X_MAXW = 64;
struct GroupDesc {
  int32_t nlevL, nlevU;
  int64_t lev_off; // first LevelDesc of the group (L levels, then U levels)
};
struct XcdState {
  unsigned tickets[8];
  unsigned global_ticket, arrived, epoch, pad;
};
__global__ void k_trsv_xcd_prologue(XcdState *st)
{
  if (threadIdx.x < 8) st->tickets[threadIdx.x] = 0;
  if (threadIdx.x == 0) {
    st->global_ticket = 0;
    st->arrived = 0;
    st->epoch += 1;
  }
}
__device__ __forceinline__ unsigned hw_xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u; } // HW_REG_XCC_ID[3:0]

// ---------------------------------------------------------------------------------------------
// XCD-local solve with a dedicated LOADER wave per workgroup ("xcd2" engine).
// vmcnt retires in order, so a wave that both streams factor entries from HBM and polls flags pays
// the HBM latency on every level.  Here each workgroup has two waves: wave 1 streams the tiles of
// the workgroup's chunks (row ids, permuted right-hand side / inverse pivots, 16 entries per row)
// from HBM into an LDS ring several levels ahead; wave 0 takes a tile from LDS (lgkmcnt), polls the
// previous level's flags, gathers x from L2, stores, drains only its own store and sets its flag.
// The two waves synchronise through two LDS words per ring (produced / consumed counts); they never
// meet at a barrier.  Cross-workgroup protocol and XCD grouping are those of k_trsv_xcd.
constexpr int TRSV_L_SLOTS = 8;
struct TrsvTileHdr { // everything the compute wave needs to know about a work item: no descriptor loads on its path
  int32_t valid, grp, lev, c, m, w, upper, nact_prev, level_done, pad;
  int64_t flag_base, ent_off;
};
struct TrsvLdsTile {
  int32_t row[64];
  double s0[64];  // L: right-hand side d[row];  U: inverse pivot
  int32_t cc[TRSV_UNROLL][64];
  double vv[TRSV_UNROLL][64];
  TrsvTileHdr hdr;
};
constexpr int TRSV_L_LOADERS = 2;
struct TrsvLds {
  TrsvLdsTile tile[TRSV_L_SLOTS];
  unsigned ready[TRSV_L_SLOTS]; // ready[slot] = seq + 1 once work item seq is in the slot (several loader waves)
  unsigned produced, consumed;  // work items written by the (single) loader / released by the compute wave
};
__global__ void k_permute_rhs(int64_t n, const int32_t *__restrict__ rows, const double *__restrict__ d, double *__restrict__ dperm)
{
  for (int64_t i = blockIdx.x * (int64_t)WG + threadIdx.x; i < n; i += (int64_t)gridDim.x * WG) dperm[i] = d[rows[i]];
}

// enumerates the work items (level, chunk) of one wave in processing order
struct TrsvWork {
  const GroupDesc *groups;
  const LevelDesc *desc;
  int ngroups, gstep, grp, lev, c, rank, W;
  __device__ bool valid() const { return grp < ngroups; }
  __device__ void init(const GroupDesc *g, const LevelDesc *d_, int ng, int first, int step, int rank_, int W_)
  {
    groups = g; desc = d_; ngroups = ng; gstep = step; grp = first; lev = 0; c = rank_; rank = rank_; W = W_;
    settle();
  }
  __device__ void settle()
  { // move to the next existing (level, chunk)
    while (grp < ngroups) {
      const GroupDesc G = groups[grp];
      const int nlev = G.nlevL + G.nlevU;
      while (lev < nlev) {
        const int nchunk = (desc[G.lev_off + lev].m + 63) >> 6;
        if (c < nchunk) return;
        ++lev;
        c = rank;
      }
      grp += gstep;
      lev = 0;
      c = rank;
    }
  }
  __device__ void advance()
  {
    c += W;
    settle();
  }
};

__global__ __launch_bounds__(64 * (1 + TRSV_L_LOADERS)) void k_trsv_xcd2(int ngroups, const GroupDesc *__restrict__ groups, const LevelDesc *__restrict__ desc,
                                                    const int64_t *__restrict__ flag_off, const int32_t *__restrict__ rowsA,
                                                    const int32_t *__restrict__ colsA, const double *__restrict__ valsA,
                                                    const double *__restrict__ dinvA, const double *__restrict__ dperm, double *x,
                                                    unsigned *flags, XcdState *st, unsigned *err, unsigned long long *stamps)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  TrsvLds &S = *reinterpret_cast<TrsvLds *>(smem_raw);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  if (threadIdx.x < TRSV_L_SLOTS) S.ready[threadIdx.x] = 0;
  if (threadIdx.x == 0) {
    S.produced = 0;
    S.consumed = 0;
  }
  // ---- placement: ticket on the own XCD, then a grid barrier (wave 0 lane 0 acts for the workgroup)
  __shared__ unsigned sh_xcc, sh_t, sh_gt, sh_fail;
  if (threadIdx.x == 0) {
    const unsigned xcc = hw_xcc_id();
    sh_xcc = xcc;
    sh_t = __hip_atomic_fetch_add(&st->tickets[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh_gt = __hip_atomic_fetch_add(&st->global_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&st->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned fail_ = 0;
    for (unsigned spins = 0; __hip_atomic_load(&st->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x; ++spins) {
      if (spins > (1u << 22)) {
        __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        fail_ = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    sh_fail = fail_;
  }
  __syncthreads(); // the only workgroup barrier: publishes the placement to both waves
  if (sh_fail) return;
  const unsigned xcc = sh_xcc, t = sh_t, gt = sh_gt;
  const unsigned epoch = __hip_atomic_load(&st->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned tk = lane < 8 ? __hip_atomic_load(&st->tickets[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
  const int owners = ngroups < 8 ? ngroups : 8;
  const bool local_ok = __all(lane >= owners || tk > 0u);
  int W, rank, first, step;
  bool wt;
  if (local_ok) {
    W = (int)__shfl((int)tk, (int)xcc, 64);
    if (W > TRSV_X_MAXW) W = TRSV_X_MAXW;
    rank = (int)t; first = (int)xcc; step = 8; wt = false;
  } else {
    W = (int)gridDim.x < TRSV_X_MAXW ? (int)gridDim.x : TRSV_X_MAXW;
    rank = (int)gt; first = 0; step = 1; wt = true;
  }
  if (rank >= W) return;
  TrsvWork wk;
  wk.init(groups, desc, ngroups, first, step, rank, W);
  volatile unsigned *ready = S.ready;
  volatile unsigned *consumed = &S.consumed;

  if (wave >= 1) {
    // =========================== loaders: HBM -> LDS ring ===========================
    // TRSV_L_LOADERS loader waves; loader j handles the work items seq = j, j + L, j + 2L, ... and announces each
    // slot separately (ready[slot] = seq + 1), so tiles of different loaders may land out of order.
    const int lj = wave - 1;
    // two tiles in flight: the loads of item seq+1 are issued before item seq is written to LDS.  The loader also
    // resolves all schedule descriptors and hands them over in the tile header.
    struct Regs {
      int32_t rowm;
      double s0;
      int32_t cc[TRSV_UNROLL];
      double vv[TRSV_UNROLL];
      TrsvTileHdr h;
    };
    auto issue = [&](Regs &R, const TrsvWork &k) {
      const GroupDesc G = groups[k.grp];
      const LevelDesc D = desc[G.lev_off + k.lev];
      const bool upper = k.lev >= G.nlevL;
      const int r = (k.c << 6) + lane;
      const int rr = r < D.m ? r : D.m - 1;
      const int32_t row = rowsA[D.row_off + rr];
      R.rowm = r < D.m ? row : -1 - row; // negative = shadow lane (no store)
      R.s0 = upper ? dinvA[D.row_off + rr] : dperm[D.row_off + rr];
      (void)*(volatile const unsigned long long *)(x + row); // warm this XCD's L2 with the line of x[row]
      const int32_t *cols = colsA + D.ent_off;
      const double *vals = valsA + D.ent_off;
#pragma unroll
      for (int u = 0; u < TRSV_UNROLL; ++u) {
        R.cc[u] = 0;
        R.vv[u] = 0.0;
        if (u < D.w) {
          R.cc[u] = cols[(int64_t)u * D.m + rr];
          R.vv[u] = vals[(int64_t)u * D.m + rr];
        }
      }
      const int nchunk = (D.m + 63) >> 6;
      int ncp = 0;
      if (k.lev > 0) ncp = (desc[G.lev_off + k.lev - 1].m + 63) >> 6;
      R.h.valid = 1;
      R.h.grp = k.grp;
      R.h.lev = k.lev;
      R.h.c = k.c;
      R.h.m = D.m;
      R.h.w = D.w;
      R.h.upper = upper ? 1 : 0;
      R.h.nact_prev = (k.lev > 0 && k.c == k.rank) ? (ncp < k.W ? ncp : k.W) : 0; // > 0: poll the previous level's flags first
      R.h.level_done = (k.c + k.W >= nchunk) ? 1 : 0;                               // last chunk of this wave in this level
      R.h.pad = 0;
      R.h.flag_base = flag_off[k.grp] + (int64_t)k.lev * TRSV_X_MAXW;
      R.h.ent_off = D.ent_off;
    };
    Regs A, B;
    for (int q = 0; q < lj && wk.valid(); ++q) wk.advance(); // first item of this loader
    unsigned seq = (unsigned)lj;
    auto end_marker = [&](unsigned sq) { // the first loader that runs out of items writes the end marker at its next sequence number
      for (unsigned spins = 0; sq >= *consumed + TRSV_L_SLOTS; ++spins) {
        if (spins > (1u << 24)) return;
        __builtin_amdgcn_s_sleep(1);
      }
      if (lane == 0) {
        S.tile[sq % TRSV_L_SLOTS].hdr.valid = 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ready[sq % TRSV_L_SLOTS] = sq + 1;
      }
    };
    if (!wk.valid()) {
      end_marker(seq);
      return;
    }
    issue(A, wk);
    for (;; seq += TRSV_L_LOADERS) {
      for (int q = 0; q < TRSV_L_LOADERS && wk.valid(); ++q) wk.advance();
      const bool more = wk.valid();
      if (more) issue(B, wk);
      for (unsigned spins = 0; seq >= *consumed + TRSV_L_SLOTS; ++spins) { // free ring slot?
        if (spins > (1u << 24)) {
          if (lane == 0) __hip_atomic_store(err, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          return;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      TrsvLdsTile &T = S.tile[seq % TRSV_L_SLOTS];
      T.row[lane] = A.rowm;
      T.s0[lane] = A.s0;
#pragma unroll
      for (int u = 0; u < TRSV_UNROLL; ++u)
        if (u < A.h.w) {
          T.cc[u][lane] = A.cc[u];
          T.vv[u][lane] = A.vv[u];
        }
      if (lane == 0) T.hdr = A.h;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // tile is in LDS before it is announced
      if (lane == 0) ready[seq % TRSV_L_SLOTS] = seq + 1;
      if (!more) {
        end_marker(seq + TRSV_L_LOADERS);
        return;
      }
      A = B;
    }
  }

  // =========================== compute wave ===========================
  // diagnostic stamps (only when a buffer is passed; production launches pass nullptr)
  const bool stamp = stamps != nullptr && rank == 0 && (local_ok ? xcc == 0 : true);
  unsigned long long t_tile = 0, t_poll = 0, t_gather = 0, t_drain = 0, n_items = 0, t_begin = 0, tq = 0;
#define DDM_STAMP(acc)                                                 \
  if (stamp) {                                                         \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();      \
    acc += now_ - tq;                                                  \
    tq = now_;                                                         \
  }
  if (stamp) t_begin = tq = __builtin_amdgcn_s_memtime();
  for (unsigned seq = 0;; ++seq) {
    for (unsigned spins = 0; ready[seq % TRSV_L_SLOTS] != seq + 1; ++spins) {
      if (spins > (1u << 24)) {
        if (lane == 0) __hip_atomic_store(err, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    const TrsvLdsTile &T = S.tile[seq % TRSV_L_SLOTS];
    const TrsvTileHdr H = T.hdr;
    if (!H.valid) break;
    const bool upper = H.upper != 0;
    const int32_t rowm = T.row[lane];
    const bool act = rowm >= 0;
    const int32_t row = act ? rowm : -1 - rowm;
    const double s0 = T.s0[lane];
    int32_t cc[TRSV_UNROLL];
    double vv[TRSV_UNROLL], xv[TRSV_UNROLL];
#pragma unroll
    for (int u = 0; u < TRSV_UNROLL; ++u) {
      cc[u] = 0;
      vv[u] = 0.0;
      if (u < H.w) {
        cc[u] = T.cc[u][lane];
        vv[u] = T.vv[u][lane];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) *consumed = seq + 1; // slot may be refilled
    DDM_STAMP(t_tile)
    // previous level of this group complete?
    if (H.nact_prev > 0) {
      const unsigned *fp = flags + H.flag_base - TRSV_X_MAXW;
      for (unsigned spins = 0;; ++spins) {
        const unsigned v = lane < H.nact_prev ? __hip_atomic_load(fp + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
        if (__all(v == epoch)) break;
        if (spins > (1u << 22)) {
          if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    DDM_STAMP(t_poll)
#pragma unroll
    for (int u = 0; u < TRSV_UNROLL; ++u) {
      xv[u] = 0.0;
      if (u < H.w) xv[u] = ld_sc1(x + cc[u]);
    }
    double s = upper ? ld_sc1(x + row) : s0;
#pragma unroll
    for (int u = 0; u < TRSV_UNROLL; ++u) s -= vv[u] * xv[u];
    if (H.w > TRSV_UNROLL) { // rows wider than a tile: the rest straight from global memory (rare)
      const int r = (H.c << 6) + lane;
      const int rr = r < H.m ? r : H.m - 1;
      for (int k = TRSV_UNROLL; k < H.w; ++k)
        s -= valsA[H.ent_off + (int64_t)k * H.m + rr] * ld_sc1(x + colsA[H.ent_off + (int64_t)k * H.m + rr]);
    }
    if (stamp) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      DDM_STAMP(t_gather)
    }
    const double out = upper ? s * s0 : s;
    if (act) {
      if (wt) st_sc1(x + row, out);
      else x[row] = out;
    }
    if (H.level_done) { // last chunk of this wave in this level -> drain and raise the flag
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) {
        unsigned *f = flags + H.flag_base + rank;
        if (wt) __hip_atomic_store(f, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *(volatile unsigned *)f = epoch;
      }
    }
    DDM_STAMP(t_drain)
    n_items += 1;
  }
  if (stamp && lane == 0) {
    stamps[0] = t_tile;
    stamps[1] = t_poll;
    stamps[2] = t_gather;
    stamps[3] = t_drain;
    stamps[4] = n_items;
    stamps[5] = __builtin_amdgcn_s_memtime() - t_begin;
  }
#undef DDM_STAMP
}

// ---------------------------------------------------------------------------------------------
// xp[pos] <- d[row(pos)] for the L positions of every group; x[row(pos)] <- xp[pos] for the U positions
__global__ void k_w_permute_in(int64_t n, const int64_t *__restrict__ lpos, const int32_t *__restrict__ rows, const double *__restrict__ d,
                               double *__restrict__ dperm)
{
  for (int64_t i = blockIdx.x * (int64_t)WG + threadIdx.x; i < n; i += (int64_t)gridDim.x * WG) dperm[lpos[i]] = d[rows[lpos[i]]];
}
__global__ void k_w_permute_out(int64_t n, const int64_t *__restrict__ upos, const int32_t *__restrict__ rows, const double *__restrict__ xp,
                                double *__restrict__ x)
{
  for (int64_t i = blockIdx.x * (int64_t)WG + threadIdx.x; i < n; i += (int64_t)gridDim.x * WG) x[rows[upos[i]]] = xp[upos[i]];
}

// ---------------------------------------------------------------------------------------------
// K2 extend (schwarz.hh:121-122), K5 restrict (schwarz.hh:146), K4 POU scaling (schwarz.hh:141)
// dst[i,:] = src[perm[i],:] / dst[perm[i],:] = src[i,:] for row-major n x nrhs blocks (fill-reducing order of the direct solver)
// (src / dst of the un-permuted side have leading dimension ld; the permuted side is packed, ld = nrhs)
__global__ void k_perm_gather(int64_t n, int nrhs, const int32_t *__restrict__ perm, const double *__restrict__ src, int64_t ld, double *__restrict__ dst)
{
  const int64_t tot = n * nrhs;
  for (int64_t t = blockIdx.x * (int64_t)WG + threadIdx.x; t < tot; t += (int64_t)gridDim.x * WG) {
    const int64_t i = t / nrhs;
    dst[t] = src[(int64_t)perm[i] * ld + (t - i * nrhs)];
  }
}
__global__ void k_perm_scatter(int64_t n, int nrhs, const int32_t *__restrict__ perm, const double *__restrict__ src, double *__restrict__ dst, int64_t ld)
{
  const int64_t tot = n * nrhs;
  for (int64_t t = blockIdx.x * (int64_t)WG + threadIdx.x; t < tot; t += (int64_t)gridDim.x * WG) {
    const int64_t i = t / nrhs;
    dst[(int64_t)perm[i] * ld + (t - i * nrhs)] = src[t];
  }
}
// dst[perm[i]] += src[i] (the correction of an iterative-refinement step, scattered back into the solution)
__global__ void k_perm_scatter_add(int64_t n, int nrhs, const int32_t *__restrict__ perm, const double *__restrict__ src, double *__restrict__ dst, int64_t ld)
{
  const int64_t tot = n * nrhs;
  for (int64_t t = blockIdx.x * (int64_t)WG + threadIdx.x; t < tot; t += (int64_t)gridDim.x * WG) {
    const int64_t i = t / nrhs;
    dst[(int64_t)perm[i] * ld + (t - i * nrhs)] += src[t];
  }
}
// R = D - A X for row-major blocks (n x nrhs): the residual of an iterative-refinement step (dune/ddm/eigensolvers/umfpack.hh:55-60);
// one thread per (row, rhs), row sums in column order
__global__ __launch_bounds__(WG) void k_residual_rowmajor(int64_t n, int nrhs, const int64_t *__restrict__ rp, const int32_t *__restrict__ ci, const double *__restrict__ va,
                                                          const double *__restrict__ x, int64_t ldx, const double *__restrict__ d, int64_t ldd, double *__restrict__ r, int64_t ldr)
{
  const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x;
  const int64_t row = t / nrhs;
  if (row >= n) return;
  const int j = (int)(t - row * nrhs);
  double s = d[row * ldd + j];
  for (int64_t k = rp[row]; k < rp[row + 1]; ++k) s -= va[k] * x[(int64_t)ci[k] * ldx + j];
  r[row * ldr + j] = s;
}
__global__ void k_extend(int64_t n, const int32_t *__restrict__ ext_map, const double *__restrict__ d, double *__restrict__ dov)
{
  for (int64_t i = blockIdx.x * (int64_t)WG + threadIdx.x; i < n; i += (int64_t)gridDim.x * WG) {
    const int32_t m = ext_map[i];
    dov[i] = m >= 0 ? d[m] : 0.0;
  }
}
// x[m] = (ACC ? x[m] : 0) + w[i]*xov[i] for the rows that have a non-overlapping image
template <bool ACC, bool SCALE>
__global__ void k_restrict(int64_t n, const int32_t *__restrict__ ext_map, const double *__restrict__ xov,
                           const double *__restrict__ w, double *__restrict__ x)
{
  for (int64_t i = blockIdx.x * (int64_t)WG + threadIdx.x; i < n; i += (int64_t)gridDim.x * WG) {
    const int32_t m = ext_map[i];
    if (m >= 0) {
      const double v = SCALE ? xov[i] * w[i] : xov[i];
      x[m] = ACC ? x[m] + v : v;
    }
  }
}
__global__ void k_scale(int64_t n, const double *__restrict__ w, double *__restrict__ x)
{
  for (int64_t i = blockIdx.x * (int64_t)WG + threadIdx.x; i < n; i += (int64_t)gridDim.x * WG) x[i] *= w[i];
}
__global__ void k_fill(int64_t n, double v, double *__restrict__ x)
{
  for (int64_t i = blockIdx.x * (int64_t)WG + threadIdx.x; i < n; i += (int64_t)gridDim.x * WG) x[i] = v;
}
// y = a*x + y  (a from host) ; z = x + y
__global__ void k_axpy(int64_t n, double a, const double *__restrict__ x, double *__restrict__ y)
{
  for (int64_t i = blockIdx.x * (int64_t)WG + threadIdx.x; i < n; i += (int64_t)gridDim.x * WG) y[i] += a * x[i];
}

// y -= (*h) x with the coefficient in device memory (modified Gram-Schmidt of restarted GMRES), x *= a
__global__ void k_axpy_negdev(int64_t n, const double *__restrict__ h, const double *__restrict__ x, double *__restrict__ y)
{
  const double a = h[0];
  for (int64_t i = blockIdx.x * (int64_t)WG + threadIdx.x; i < n; i += (int64_t)gridDim.x * WG) y[i] -= a * x[i];
}
__global__ void k_scal(int64_t n, double a, double *__restrict__ x)
{
  for (int64_t i = blockIdx.x * (int64_t)WG + threadIdx.x; i < n; i += (int64_t)gridDim.x * WG) x[i] *= a;
}

// ---------------------------------------------------------------------------------------------
// halo exchange: pack (gather) and deterministic unpack (copy / add), SURVEY.md 2.3 C1-C4
__global__ void k_pack(int64_t n, const int64_t *__restrict__ idx, const double *__restrict__ v, double *__restrict__ buf)
{
  for (int64_t i = blockIdx.x * (int64_t)WG + threadIdx.x; i < n; i += (int64_t)gridDim.x * WG) buf[i] = v[idx[i]];
}
template <bool ADD>
__global__ void k_unpack(int64_t ndst, const int64_t *__restrict__ dst_idx, const int64_t *__restrict__ dst_ptr,
                         const int64_t *__restrict__ src_pos, const double *__restrict__ buf, double *__restrict__ v)
{
  for (int64_t t = blockIdx.x * (int64_t)WG + threadIdx.x; t < ndst; t += (int64_t)gridDim.x * WG) {
    const int64_t i = dst_idx[t];
    double s = ADD ? v[i] : 0.0;
    for (int64_t k = dst_ptr[t]; k < dst_ptr[t + 1]; ++k) s = ADD ? s + buf[src_pos[k]] : buf[src_pos[k]];
    v[i] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// K20 / a3: owner-masked dot products (nonoverlapping_operator.hh:76-83), two-stage and
// deterministic: partial[b] per workgroup, then one workgroup sums the partials in index order.
template <bool MASKED>
__global__ __launch_bounds__(WG) void k_dot_partial(int64_t n, const uint8_t *__restrict__ mask, const double *__restrict__ x,
                                                     const double *__restrict__ y, double *__restrict__ partial)
{
  __shared__ double red[4];
  double s = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)WG + threadIdx.x; i < n; i += (int64_t)gridDim.x * WG)
    if (!MASKED || mask[i]) s += x[i] * y[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(WG) void k_reduce_final(int nb, const double *__restrict__ partial, double *__restrict__ out)
{
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += WG) s += partial[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = s;
}

// CG scalar recurrences on the device (dune-istl CGSolver), one thread.
// scal: [0]=rholast [1]=alpha=<p,q> [2]=lambda [3]=rho [4]=beta [5]=<b,b>
__global__ void k_cg_lambda(double *scal) { scal[2] = scal[0] / scal[1]; }
__global__ void k_cg_beta(double *scal)
{
  scal[4] = scal[3] / scal[0];
  scal[0] = scal[3];
}
// x += lambda p ; b -= lambda q
__global__ void k_cg_update(int64_t n, const double *__restrict__ scal, const double *__restrict__ p,
                            const double *__restrict__ q, double *__restrict__ x, double *__restrict__ b)
{
  const double lam = scal[2];
  for (int64_t i = blockIdx.x * (int64_t)WG + threadIdx.x; i < n; i += (int64_t)gridDim.x * WG) {
    x[i] += lam * p[i];
    b[i] -= lam * q[i];
  }
}
// the same update followed by the owner-masked partial sums of <b, b> (same grid and summation order as k_dot_partial,
// so the norm is bit-identical to the separate dot product; saves its pass over b)
template <bool MASKED>
__global__ __launch_bounds__(WG) void k_cg_update_norm(int64_t n, const double *__restrict__ scal, const uint8_t *__restrict__ mask,
                                                        const double *__restrict__ p, const double *__restrict__ q, double *__restrict__ x,
                                                        double *__restrict__ b, double *__restrict__ partial)
{
  __shared__ double red[4];
  const double lam = scal[2];
  double s = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)WG + threadIdx.x; i < n; i += (int64_t)gridDim.x * WG) {
    x[i] += lam * p[i];
    const double bi = b[i] - lam * q[i];
    b[i] = bi;
    if (!MASKED || mask[i]) s += bi * bi;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// p = beta p + q
__global__ void k_cg_direction(int64_t n, const double *__restrict__ scal, const double *__restrict__ q, double *__restrict__ p)
{
  const double beta = scal[4];
  for (int64_t i = blockIdx.x * (int64_t)WG + threadIdx.x; i < n; i += (int64_t)gridDim.x * WG) p[i] = beta * p[i] + q[i];
}

// ---------------------------------------------------------------------------------------------
// K6 coarse restriction d0[(s,j)] = <r_j^s, d_ovlp^s>  (galerkin_preconditioner.hh:165-167):
// a tall-skinny GEMV.  Work item = (chunk of rows inside one subdomain); wave w of the
// workgroup handles vectors w, w+4, ...; lanes stride the rows (coalesced).
static constexpr int COARSE_KMAX = 256; // basis vectors per subdomain (GenEO threshold mode doubles nev up to nev_max, spectra.hh:157-163)
struct RowChunk {
  int64_t r0, r1;
  int32_t sub, pad;
};
__global__ __launch_bounds__(WG) void k_coarse_restrict_partial(int kmax, int64_t ld, const double *__restrict__ basis,
                                                                 const double *__restrict__ d, const RowChunk *__restrict__ chunks,
                                                                 double *__restrict__ partial /* [nchunk][kmax] */, int nchunk)
{
  // (grid-stride over the chunks: a small grid keeps the kernel's footprint per CU low when it runs beside the local solve)
  for (int ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
  const RowChunk c = chunks[ch];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int j = w; j < kmax; j += 4) {
    const double *bj = basis + (int64_t)j * ld;
    double s = 0.0;
    // (eight loads of each stream in flight per lane; the products are added in the same order as one by one)
    int64_t r = c.r0 + lane;
    for (; r + 7 * 64 < c.r1; r += 8 * 64) {
      double bv[8], dv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) bv[u] = __builtin_nontemporal_load(bj + r + u * 64);
#pragma unroll
      for (int u = 0; u < 8; ++u) dv[u] = d[r + u * 64];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += bv[u] * dv[u];
    }
    for (; r < c.r1; r += 64) s += bj[r] * d[r];
    s = wave_sum(s);
    if (lane == 0) partial[(int64_t)ch * kmax + j] = s;
  }
  }
}
// one workgroup: sums the chunk partials of each (subdomain, vector) in chunk order and scatters
// them to the global coarse vector (zero elsewhere, so that an all-reduce assembles d0)
__global__ __launch_bounds__(WG) void k_coarse_restrict_final(int nsub, int kmax, const int32_t *__restrict__ sub_chunk_ptr,
                                                               const double *__restrict__ partial,
                                                               const int64_t *__restrict__ coarse_index, int64_t K,
                                                               double *__restrict__ d0)
{
  for (int64_t i = threadIdx.x; i < K; i += WG) d0[i] = 0.0;
  __syncthreads();
  for (int t = threadIdx.x; t < nsub * kmax; t += WG) {
    const int s = t / kmax, j = t % kmax;
    const int64_t gi = coarse_index[t];
    if (gi < 0) continue;
    double acc = 0.0;
    for (int c = sub_chunk_ptr[s]; c < sub_chunk_ptr[s + 1]; ++c) acc += partial[(int64_t)c * kmax + j];
    d0[gi] = acc;
  }
}
// K8: x0 = A0^-1 d0 with the replicated explicit inverse (K x K, row-major); one wave per row
__global__ __launch_bounds__(WG) void k_dense_mv(int64_t K, const double *__restrict__ M, const double *__restrict__ v,
                                                  double *__restrict__ out)
{
  const int lane = threadIdx.x & 63;
  const int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= K) return;
  double s = 0.0;
  for (int64_t c = lane; c < K; c += 64) s += M[row * K + c] * v[c];
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}
// K7 prolongation x_ovlp = sum_j c_j r_j  (galerkin_preconditioner.hh:186-188)
__global__ __launch_bounds__(WG) void k_coarse_prolong(int kmax, int64_t ld, const double *__restrict__ basis,
                                                        const double *__restrict__ x0, const int64_t *__restrict__ coarse_index,
                                                        const RowChunk *__restrict__ chunks, double *__restrict__ xov, int nchunk)
{
  __shared__ double cj[COARSE_KMAX];
  static_assert(COARSE_KMAX <= WG, "one thread per coefficient");
  for (int ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
  const RowChunk c = chunks[ch];
  __syncthreads(); // cj of the previous chunk is no longer read
  if (threadIdx.x < kmax) {
    const int64_t gi = coarse_index[(int64_t)c.sub * kmax + threadIdx.x];
    cj[threadIdx.x] = gi >= 0 ? x0[gi] : 0.0;
  }
  __syncthreads();
  // two rows per thread and trip, four basis vectors of each in flight (added in order; kmax is usually a multiple of 4)
  for (int64_t r = c.r0 + threadIdx.x; r < c.r1; r += 2 * WG) {
    const int64_t r2 = r + WG < c.r1 ? r + WG : r; // (the second row of the last trip may not exist: computed twice, stored once)
    double s = 0.0, s2 = 0.0;
    int j = 0;
    for (; j + 4 <= kmax; j += 4) {
      double bv[4], bw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) bv[u] = __builtin_nontemporal_load(basis + (int64_t)(j + u) * ld + r);
#pragma unroll
      for (int u = 0; u < 4; ++u) bw[u] = __builtin_nontemporal_load(basis + (int64_t)(j + u) * ld + r2);
#pragma unroll
      for (int u = 0; u < 4; ++u) s += cj[j + u] * bv[u];
#pragma unroll
      for (int u = 0; u < 4; ++u) s2 += cj[j + u] * bw[u];
    }
    for (; j < kmax; ++j) {
      s += cj[j] * basis[(int64_t)j * ld + r];
      s2 += cj[j] * basis[(int64_t)j * ld + r2];
    }
    xov[r] = s;
    if (r2 != r) xov[r2] = s2;
  }
  }
}

} // namespace ddm
