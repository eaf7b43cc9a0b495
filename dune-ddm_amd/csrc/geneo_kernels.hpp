// Device kernels of the GenEO coarse-basis builder (csrc/geneo.hpp): the dense contractions of the block eigensolver on
// tall-skinny row-major blocks, written for the FP64 matrix cores of gfx950 (v_mfma_f64_16x16x4_f64), plus the small
// column-wise helpers.  They replace, on the device, the dense pieces of the reference's eigensolver: Spectra's Gram /
// re-orthogonalisation products V^T B f and the basis compression V * Q (extern/spectra-1.2.0 .../Lanczos.h, Arnoldi.h:310-329;
// SURVEY K14, K15), and finalize_eigenvectors (dune/ddm/coarsespaces/coarse_spaces.hh:52-61).
//
// Operand maps of v_mfma_f64_16x16x4_f64 (cdna_hip_programming.md "Fragment layout"): lane l holds A[i = l & 15][k = l >> 4]
// and B[k = l >> 4][j = l & 15]; the four results of a lane are D[row = (l >> 4) + 4 * reg][col = l & 15].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ddm {

typedef double v4d __attribute__((ext_vector_type(4)));

struct GChunk { // a run of rows inside ONE subdomain
  int64_t r0, r1;
  int32_t sub, pad;
};

// ---------------------------------------------------------------------------------------------
// Gram product of two tall-skinny blocks, split over row chunks:  partial[chunk] = U[rows]^T V[rows]   (pu x pv, row-major).
// U^T V sums over the ROW index, so a 4-row slab of 16 columns of U is exactly one A operand (A[i = column][k = row]) and the
// same slab shape of V one B operand: both are read as four fully used 128-byte segments per instruction, no LDS staging.
// A workgroup (4 wavefronts) owns a chunk; wavefront w accumulates the tile rows ta = w, w + 4, ... against all tile columns,
// TAW x TB accumulator tiles in registers (template bounds; the actual tile counts are wave-uniform run-time values).
// HBM-bound: 8 (pu + pv) bytes per row for 2 pu pv flops.
template <int TAW, int TB>
// The pu x pv result is written as a sub-block at (out_i0, out_j0) of a per-chunk matrix with leading dimension out_ld (blocks wider
// than the register tiles are computed in column panels: GeneoWork::gram).
__global__ __launch_bounds__(256) void k_gram_mfma(const GChunk *__restrict__ chunks, const double *__restrict__ U, int64_t ldu, int pu,
                                                  const double *__restrict__ V, int64_t ldv, int pv, double *__restrict__ partial, int64_t out_stride, int out_ld,
                                                  int out_i0, int out_j0)
{
  const GChunk c = chunks[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ta_n = (pu + 15) >> 4, tb_n = (pv + 15) >> 4;
  const int lc = lane & 15, lr = lane >> 4;
  v4d acc[TAW][TB];
#pragma unroll
  for (int a = 0; a < TAW; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b) acc[a][b] = v4d{0.0, 0.0, 0.0, 0.0};
  constexpr int GU = (TAW + TB <= 7) ? 4 : 2; // 4-row slabs whose operand loads are issued together (the kernel is HBM-latency bound otherwise)
  for (int64_t r = c.r0; r < c.r1; r += 4 * GU) {
    double av[GU][TAW], bv[GU][TB];
#pragma unroll
    for (int g = 0; g < GU; ++g) {
      const int64_t row = r + 4 * g + lr;
      const bool rok = row < c.r1;
#pragma unroll
      for (int a = 0; a < TAW; ++a) {
        const int col = ((wave + 4 * a) << 4) + lc;
        av[g][a] = (rok && wave + 4 * a < ta_n && col < pu) ? U[row * ldu + col] : 0.0;
      }
#pragma unroll
      for (int b = 0; b < TB; ++b) {
        const int col = (b << 4) + lc;
        bv[g][b] = (rok && b < tb_n && col < pv) ? V[row * ldv + col] : 0.0;
      }
    }
#pragma unroll
    for (int g = 0; g < GU; ++g)
#pragma unroll
      for (int a = 0; a < TAW; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
          if (wave + 4 * a < ta_n && b < tb_n) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[g][a], bv[g][b], acc[a][b], 0, 0, 0);
  }
  double *out = partial + (int64_t)blockIdx.x * out_stride + (int64_t)out_i0 * out_ld + out_j0;
#pragma unroll
  for (int a = 0; a < TAW; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      if (wave + 4 * a >= ta_n || b >= tb_n) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = ((wave + 4 * a) << 4) + lr + 4 * q, j = (b << 4) + lc;
        if (i < pu && j < pv) out[(int64_t)i * out_ld + j] = acc[a][b][q];
      }
    }
}

// G[sub] = sum of the chunk partials of the subdomain, in chunk order (deterministic); one thread per entry
__global__ void k_gram_reduce(int nsub, const int32_t *__restrict__ sub_chunk_ptr, int64_t pp, const double *__restrict__ partial, double *__restrict__ G)
{
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= (int64_t)nsub * pp) return;
  const int s = (int)(t / pp);
  const int64_t e = t - (int64_t)s * pp;
  double acc = 0.0;
  for (int c = sub_chunk_ptr[s]; c < sub_chunk_ptr[s + 1]; ++c) acc += partial[(int64_t)c * pp + e];
  G[t] = acc;
}

// ---------------------------------------------------------------------------------------------
// Basis rotation  Out[rows, 0:q) = (Base[rows, 0:q) -) U[rows, 0:p) * Y[sub]   with Y (p x q, row-major) per subdomain:
// the Rayleigh-Ritz update X <- [X W P] Y of the block method (= Spectra's compress_V, Arnoldi.h:310-329) and the
// A-orthogonalisation W <- W - X (AX^T W).  The product sums over the COLUMN index of U, so a wavefront stages its 16 rows of
// U through LDS (coalesced row reads in, A operands A[i = row][k = column] out) and streams Y from LDS as B operands.
// blockIdx.y selects one of up to three (U, Out) pairs that share Y (the blocks S, A S, C S are rotated by the same Y).
struct RotArgs {
  const double *U[3];
  double *Out[3];
  const double *Base[3]; // nullptr: plain product; else Out = Base - U Y
};
constexpr int ROT_TQ = 3; // q <= 48
// Output columns j >= gap_from are written `gap` columns further right (the fused Rayleigh-Ritz rotation writes X_new to the X slot
// and P_new straight into the P slot, two slots further: no copy kernel behind it).
template <int ROT_PRE> // registers per lane for the prefetched slab: 20 (p <= 80), or 0 = row-by-row staging without prefetch (any p <= 144)
__global__ __launch_bounds__(256) void k_rotate_mfma(const GChunk *__restrict__ chunks, RotArgs args, int64_t ldu, int p, const double *__restrict__ Yall,
                                                    int q, int64_t ldo, int64_t ldb, int gap_from, int gap, int y_ld, int y_rows, int y_k0, int y_j0, int mode)
{
  // Panels (GeneoWork::rotate): this launch multiplies the p columns of U it is given by rows [y_k0, y_k0 + p), columns [y_j0, y_j0 + q)
  // of the per-subdomain coefficient matrix (y_rows x y_ld, row-major) and produces the output columns y_j0 .. y_j0 + q - 1.
  // mode 0: Out = U Y;  1: Out = Base - U Y (Base indexed by the plain column);  2 / 3: Out = Out -/+ U Y (further K panels: the
  // addend is the output itself, at its mapped column)
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const GChunk c = chunks[blockIdx.x];
  const double *__restrict__ U = args.U[blockIdx.y];
  double *__restrict__ Out = args.Out[blockIdx.y];
  const double *__restrict__ Base = args.Base[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lc = lane & 15, lr = lane >> 4;
  const int p4 = (p + 3) & ~3;              // k extent padded to the MFMA step
  const int q16 = ((q + 15) >> 4) << 4;     // columns of Y padded to whole tiles
  double *Ys = lds;                          // p4 x q16
  const int ustride = p4 + 1;                // odd stride: the 16 rows of an A operand fall into different banks
  double *Us = lds + (int64_t)p4 * q16 + (int64_t)wave * 16 * ustride;
  const double *Y = Yall + (int64_t)c.sub * y_rows * y_ld;
  for (int t = threadIdx.x; t < p4 * q16; t += 256) {
    const int k = t / q16, j = t - k * q16;
    Ys[t] = (k < p && j < q) ? Y[(int64_t)(y_k0 + k) * y_ld + y_j0 + j] : 0.0;
  }
  __syncthreads();
  const int tq_n = q16 >> 4;
  // The 16 x p4 slab of U is read as ONE flat run of 16 p4 elements (contiguous in memory when ldu == p: full 512-byte loads), and the
  // NEXT slab is fetched into registers while the matrix cores work on the current one (round 3: 1.5 TB/s before).
  constexpr int NPRE = ROT_PRE > 0 ? ROT_PRE : 1;
  const int nel = 16 * p4;
  double pre[NPRE];
  auto load_slab = [&](int64_t r0s) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
      const int idx = lane + (u << 6);
      const int i = idx / p4, k = idx - i * p4;
      const int64_t row = r0s + i;
      pre[u] = (idx < nel && row < c.r1 && k < p) ? U[row * ldu + k] : 0.0;
    }
  };
  if (ROT_PRE > 0 && c.r0 + 16 * wave < c.r1) load_slab(c.r0 + 16 * wave);
  for (int64_t r0 = c.r0 + 16 * wave; r0 < c.r1; r0 += 64) {
    if (ROT_PRE > 0) {
#pragma unroll
      for (int u = 0; u < NPRE; ++u) {
        const int idx = lane + (u << 6);
        if (idx < nel) {
          const int i = idx / p4;
          Us[i * ustride + (idx - i * p4)] = pre[u];
        }
      }
    } else { // stage 16 rows x p of U: row by row, lanes along the row (coalesced)
      for (int i = 0; i < 16; ++i) {
        const int64_t row = r0 + i;
        for (int k = lane; k < p4; k += 64) Us[i * ustride + k] = (row < c.r1 && k < p) ? U[row * ldu + k] : 0.0;
      }
    }
    __builtin_amdgcn_s_waitcnt(0); // the wave reads back what its own lanes wrote (same wave: no barrier needed, only completion)
    __builtin_amdgcn_wave_barrier();
    if (ROT_PRE > 0 && r0 + 64 < c.r1) load_slab(r0 + 64);
    v4d acc[ROT_TQ];
#pragma unroll
    for (int t = 0; t < ROT_TQ; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < p4; k0 += 4) {
      const double a = Us[lc * ustride + k0 + lr];
#pragma unroll
      for (int t = 0; t < ROT_TQ; ++t)
        if (t < tq_n) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Ys[(k0 + lr) * q16 + (t << 4) + lc], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < ROT_TQ; ++t) {
      if (t >= tq_n) continue;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int64_t row = r0 + lr + 4 * v;
        const int j = (t << 4) + lc;
        if (row < c.r1 && j < q) {
          const int ja = y_j0 + j;
          double *o = Out + row * ldo + (ja < gap_from ? ja : ja + gap);
          const double a = acc[t][v];
          *o = mode == 0 ? a : (mode == 1 ? Base[row * ldb + ja] - a : (mode == 2 ? *o - a : *o + a));
        }
      }
    }
    __builtin_amdgcn_wave_barrier(); // all lanes are done with Us before the next slab overwrites it
  }
}

// ---------------------------------------------------------------------------------------------
// column-wise helpers on row-major blocks (one thread per row, m <= 48 columns)

// R[:, j] = (CX[:, j] - mu[sub][j] AX[:, j])  (residual of the pencil C x = mu A~ x)
__global__ void k_geneo_residual(int64_t n, int m, const int32_t *__restrict__ sub_of_row, const double *__restrict__ mu, const double *__restrict__ AX,
                                 int64_t lda, const double *__restrict__ CX, int64_t ldc, double *__restrict__ R, int64_t ldr)
{
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n * m) return;
  const int64_t i = t / m;
  const int j = (int)(t - i * m);
  R[i * ldr + j] = CX[i * ldc + j] - mu[(int64_t)sub_of_row[i] * m + j] * AX[i * lda + j];
}
// X[:, j] *= s[sub][j]
__global__ void k_geneo_colscale(int64_t n, int m, const int32_t *__restrict__ sub_of_row, const double *__restrict__ s, double *__restrict__ X, int64_t ldx)
{
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n * m) return;
  const int64_t i = t / m;
  const int j = (int)(t - i * m);
  X[i * ldx + j] *= s[(int64_t)sub_of_row[i] * m + j];
}
// s[sub][j] = 1 / sqrt(max(G[sub][j][j], tiny))  from m x m Gram matrices; zero for non-positive diagonals (dropped directions)
__global__ void k_geneo_invsqrt_diag(int nsub, int m, const double *__restrict__ G, double *__restrict__ s)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nsub * m) return;
  const int sub = t / m, j = t - sub * m;
  const double g = G[((int64_t)sub * m + j) * m + j];
  s[t] = g > 1e-300 ? 1.0 / sqrt(g) : 0.0;
}
// X = w[row] * mask[row] * X  (partition of unity / free-DoF mask applied to all columns)
__global__ void k_geneo_rowscale(int64_t n, int m, const double *__restrict__ w, double *__restrict__ X, int64_t ldx)
{
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n * m) return;
  const int64_t i = t / m;
  X[i * ldx + (t - i * m)] *= w[i];
}
// finalize_eigenvectors + zero_at_dirichlet + transpose: basis[j][row] = scale[sub][j] * V[row][j]  (vector-major output, ld = n)
__global__ void k_geneo_finalize(int64_t n, int nev, const int32_t *__restrict__ sub_of_row, const double *__restrict__ scale, int mscale,
                                 const double *__restrict__ V, int64_t ldv, double *__restrict__ basis)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = sub_of_row[i];
  for (int j = 0; j < nev; ++j) basis[(int64_t)j * n + i] = scale[(int64_t)s * mscale + j] * V[i * ldv + j];
}
// dst[:, 0:m) = src[:, 0:m)  with independent leading dimensions
__global__ void k_geneo_copy_cols(int64_t n, int m, const double *__restrict__ src, int64_t lds_, double *__restrict__ dst, int64_t ldd)
{
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n * m) return;
  const int64_t i = t / m;
  const int j = (int)(t - i * m);
  dst[i * ldd + j] = src[i * lds_ + j];
}

} // namespace ddm
