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

// Guarded operand loads select their ADDRESS, not their result: an element outside the block is read from this page of zeros.  A load
// inside a conditional compiles to a branch around it (and a basic block per load); a select on the loaded VALUE puts a VALU
// instruction behind every load, which the compiler then waits for right there -- both break the software pipelines below.
// (NOT const: a constant-address-space object would make the selected pointer generic and the loads flat_load, which complete out of
// order and force full waits)
__device__ double ddm_zero_page[16];

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
  // tile rows are dealt to the wavefronts in turn, and with 5 tile rows (p = 72) the first one gets two: the FP64 MFMA (64 cycles per
  // 16x16x4 on a SIMD) makes that wavefront the critical one, so the deal starts at a different wavefront in every workgroup --
  // the workgroups that share a CU then load its four SIMDs evenly (4.5 -> 3.3 ms for the 72 x 72 products at 216^3)
  const int lane = threadIdx.x & 63, wave = ((threadIdx.x >> 6) + blockIdx.x) & 3;
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

// Gram products of NARROW blocks (pu, pv <= 32: the m x m inner products of the block eigensolver at nev = 20).  The general kernel
// above deals tile rows to the wavefronts, which leaves two of four idle here and 12 loads in flight per wavefront; this one
// splits the ROWS of the chunk over the four wavefronts (32 rows = 8 slabs per wavefront and trip, 16..32 loads in flight), every
// wavefront accumulates all (at most 2 x 2) tiles, and the four partial results meet in LDS in wavefront order (deterministic).
// SAME: U and V are the same block (column norms, R^T R): the A operand doubles as the B operand, nothing is loaded twice.
template <bool SAME, bool A1, bool B1> // A1 / B1: a second 16-column tile of U / V (pu > 16, pv > 16), compile-time: the accumulators keep their registers
__global__ __launch_bounds__(256) void k_gram_small(const GChunk *__restrict__ chunks, const double *__restrict__ U, int64_t ldu, int pu,
                                                   const double *__restrict__ V, int64_t ldv, int pv, double *__restrict__ partial, int64_t out_stride, int out_ld)
{
  __shared__ double red[4][4][256];
  const GChunk c = chunks[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lc = lane & 15, lr = lane >> 4;
  constexpr bool a1 = A1, b1 = B1;
  v4d acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = v4d{0.0, 0.0, 0.0, 0.0};
  constexpr int GU = 8;
  for (int64_t r = c.r0 + (int64_t)wave * 4 * GU; r < c.r1; r += 16 * GU) {
    double av[GU][2], bv[GU][2];
#pragma unroll
    for (int g = 0; g < GU; ++g) {
      const int64_t row = r + 4 * g + lr;
      const bool rok = row < c.r1;
      av[g][0] = *((rok && lc < pu) ? U + row * ldu + lc : ddm_zero_page);
      av[g][1] = a1 ? *((rok && 16 + lc < pu) ? U + row * ldu + 16 + lc : ddm_zero_page) : 0.0;
      if (!SAME) {
        bv[g][0] = *((rok && lc < pv) ? V + row * ldv + lc : ddm_zero_page);
        bv[g][1] = b1 ? *((rok && 16 + lc < pv) ? V + row * ldv + 16 + lc : ddm_zero_page) : 0.0;
      }
    }
#pragma unroll
    for (int g = 0; g < GU; ++g)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          if ((a == 0 || a1) && (b == 0 || b1)) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[g][a], SAME ? av[g][b] : bv[g][b], acc[a][b], 0, 0, 0);
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) red[wave][2 * a + b][4 * lane + q] = acc[a][b][q];
  __syncthreads();
  double *out = partial + (int64_t)blockIdx.x * out_stride;
#pragma unroll
  for (int t = 0; t < 4; ++t) { // thread = entry (4 lane + q) of tile t, summed over the wavefronts in order
    const int e = threadIdx.x, l = e >> 2, q = e & 3;
    const int i = ((t >> 1) << 4) + (l >> 4) + 4 * q, j = ((t & 1) << 4) + (l & 15);
    if (i < pu && j < pv) out[(int64_t)i * out_ld + j] = ((red[0][t][e] + red[1][t][e]) + red[2][t][e]) + red[3][t][e];
  }
}

// The two SYMMETRIC p x p products of the Rayleigh-Ritz step in one pass:  G1 = U^T V1,  G2 = U^T V2  with V1 = A~ U, V2 = C~ U
// (p <= 80: five 16-column tiles).  U is read once for both, and only the tiles on and above the diagonal are computed (15 of 25:
// the FP64 MFMA costs 64 cycles per 16x16x4 on a SIMD, which made the separate full products MFMA-bound on their critical wavefront,
// 2 x 4.3 ms at 216^3 for 2 x 12.5 GB); the host mirrors the upper triangle.  The 15 tiles are dealt to the four wavefronts in
// groups that share operands -- {00 01 02 03}, {11 12 13 14}, {22 23 24 04}, {33 34 44} -- and the group a wavefront takes rotates
// with the workgroup index, so that the workgroups of a CU load its four SIMDs evenly.  Operands of the next 8 rows are in flight
// while the matrix cores work on the current 8 (two register buffers).  Entries of tiles below the diagonal are NOT written.
template <int G>
struct Gram2Group;
template <>
struct Gram2Group<0> {
  static constexpr int NA = 1, NB = 4, NT = 4;
  static constexpr int AL[2] = {0, 0}, BL[4] = {0, 1, 2, 3}, TA[4] = {0, 0, 0, 0}, TB[4] = {0, 1, 2, 3};
};
template <>
struct Gram2Group<1> {
  static constexpr int NA = 1, NB = 4, NT = 4;
  static constexpr int AL[2] = {1, 1}, BL[4] = {1, 2, 3, 4}, TA[4] = {0, 0, 0, 0}, TB[4] = {0, 1, 2, 3};
};
template <>
struct Gram2Group<2> {
  static constexpr int NA = 2, NB = 3, NT = 4;
  static constexpr int AL[2] = {2, 0}, BL[4] = {2, 3, 4, 4}, TA[4] = {0, 0, 0, 1}, TB[4] = {0, 1, 2, 2};
};
template <>
struct Gram2Group<3> {
  static constexpr int NA = 2, NB = 2, NT = 3;
  static constexpr int AL[2] = {3, 4}, BL[4] = {3, 4, 4, 4}, TA[4] = {0, 0, 1, 1}, TB[4] = {0, 1, 1, 1};
};
template <int G, int TN> // TN = tiles per side = ceil(p / 16), compile-time
__device__ __forceinline__ void gram2_sym_group(const GChunk c, const double *__restrict__ U, int64_t ldu, const double *__restrict__ V1, const double *__restrict__ V2,
                                                int64_t ldv, int p, double *__restrict__ out1, double *__restrict__ out2)
{
  using GG = Gram2Group<G>;
  constexpr int GU = 2; // 4-row slabs per buffer
  const int lane = threadIdx.x & 63, lc = lane & 15, lr = lane >> 4;
  constexpr int tn = TN;
  v4d acc1[GG::NT], acc2[GG::NT];
#pragma unroll
  for (int t = 0; t < GG::NT; ++t) acc1[t] = acc2[t] = v4d{0.0, 0.0, 0.0, 0.0};
  double a0[GU][GG::NA], b0[GU][GG::NB], c0[GU][GG::NB], a1[GU][GG::NA], b1[GU][GG::NB], c1[GU][GG::NB];
  auto load = [&](double(&a)[GU][GG::NA], double(&b)[GU][GG::NB], double(&cc)[GU][GG::NB], int64_t r) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < GU; ++g) {
      const int64_t row = r + 4 * g + lr;
      const bool rok = row < c.r1;
#pragma unroll
      for (int i = 0; i < GG::NA; ++i) {
        const int col = (GG::AL[i] << 4) + lc;
        a[g][i] = *((rok && col < p) ? U + row * ldu + col : ddm_zero_page);
      }
#pragma unroll
      for (int i = 0; i < GG::NB; ++i) {
        const int col = (GG::BL[i] << 4) + lc;
        const bool ok = rok && col < p;
        b[g][i] = *(ok ? V1 + row * ldv + col : ddm_zero_page);
        cc[g][i] = *(ok ? V2 + row * ldv + col : ddm_zero_page);
      }
    }
  };
  auto compute = [&](const double(&a)[GU][GG::NA], const double(&b)[GU][GG::NB], const double(&cc)[GU][GG::NB]) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < GU; ++g)
#pragma unroll
      for (int t = 0; t < GG::NT; ++t)
        if (GG::AL[GG::TA[t]] < tn && GG::BL[GG::TB[t]] < tn) {
          acc1[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[g][GG::TA[t]], b[g][GG::TB[t]], acc1[t], 0, 0, 0);
          acc2[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[g][GG::TA[t]], cc[g][GG::TB[t]], acc2[t], 0, 0, 0);
        }
  };
  constexpr int RS = 4 * GU;
  load(a0, b0, c0, c.r0);
  for (int64_t r = c.r0; r < c.r1; r += 2 * RS) {
    load(a1, b1, c1, r + RS); // (rows beyond the chunk load zeros)
    compute(a0, b0, c0);
    load(a0, b0, c0, r + 2 * RS);
    compute(a1, b1, c1);
  }
#pragma unroll
  for (int t = 0; t < GG::NT; ++t) {
    const int ta = GG::AL[GG::TA[t]], tb = GG::BL[GG::TB[t]];
    if (ta >= tn || tb >= tn) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = (ta << 4) + lr + 4 * q, j = (tb << 4) + lc;
      if (i < p && j < p) {
        out1[(int64_t)i * p + j] = acc1[t][q];
        out2[(int64_t)i * p + j] = acc2[t][q];
      }
    }
  }
}
// partial: per chunk two p x p matrices (stride 2 p p)
template <int TN>
__global__ __launch_bounds__(256) void k_gram2_sym(const GChunk *__restrict__ chunks, const double *__restrict__ U, int64_t ldu, const double *__restrict__ V1,
                                                  const double *__restrict__ V2, int64_t ldv, int p, double *__restrict__ partial)
{
  const GChunk c = chunks[blockIdx.x];
  const int64_t pp = (int64_t)p * p;
  double *out1 = partial + (int64_t)blockIdx.x * 2 * pp, *out2 = out1 + pp;
  switch (__builtin_amdgcn_readfirstlane(((threadIdx.x >> 6) + blockIdx.x) & 3)) { // wave-uniform
  case 0: gram2_sym_group<0, TN>(c, U, ldu, V1, V2, ldv, p, out1, out2); break;
  case 1: gram2_sym_group<1, TN>(c, U, ldu, V1, V2, ldv, p, out1, out2); break;
  case 2: gram2_sym_group<2, TN>(c, U, ldu, V1, V2, ldv, p, out1, out2); break;
  default: gram2_sym_group<3, TN>(c, U, ldu, V1, V2, ldv, p, out1, out2); break;
  }
}

// G[sub] = sum of the chunk partials of the subdomain in a fixed order (deterministic): one thread per entry, eight independent running
// sums over the chunks (a single one is a chain of ~660 dependent loads at 216^3: 0.27 ms per call whatever the size of the product)
// (stride: distance of the chunks' partial matrices, pp unless two products share the buffer)
__global__ void k_gram_reduce(int nsub, const int32_t *__restrict__ sub_chunk_ptr, int64_t pp, const double *__restrict__ partial, int64_t stride, double *__restrict__ G)
{
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= (int64_t)nsub * pp) return;
  const int s = (int)(t / pp);
  const int64_t e = t - (int64_t)s * pp;
  double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  int c = sub_chunk_ptr[s];
  const int c1 = sub_chunk_ptr[s + 1];
  for (; c + 8 <= c1; c += 8)
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += partial[(int64_t)(c + u) * stride + e];
  for (; c < c1; ++c) a[0] += partial[(int64_t)c * stride + e];
  G[t] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
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
// TQ = 16-column output tiles of this launch (ceil(q / 16) <= ROT_TQ), compile-time: with a run-time tile count the compiler keeps ONE
// accumulator in the matrix-core registers and moves the others in and out around every MFMA
template <int ROT_PRE, int TQ> // ROT_PRE: registers per lane for a prefetched slab: 20 (p <= 80), or 0 = row-by-row staging without prefetch (any p <= 144)
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
  constexpr int q16 = 16 * TQ;              // columns of Y padded to whole tiles
  double *Ys = lds;                          // p4 x q16
  const int ustride = p4 + 1;                // odd stride: the 16 rows of an A operand fall into different banks
  double *Us = lds + (int64_t)p4 * q16 + (int64_t)wave * 16 * ustride;
  const double *Y = Yall + (int64_t)c.sub * y_rows * y_ld;
  for (int t = threadIdx.x; t < p4 * q16; t += 256) {
    const int k = t / q16, j = t - k * q16;
    Ys[t] = (k < p && j < q) ? Y[(int64_t)(y_k0 + k) * y_ld + y_j0 + j] : 0.0;
  }
  __syncthreads();
  // The 16 x p4 slab of U is read as ONE flat run of 16 p4 elements (contiguous in memory when ldu == p: full 512-byte loads), and the
  // slabs of the next TWO trips are on their way into registers while the matrix cores work on the current one: the workgroup's
  // LDS (Y + four staging slabs) allows two workgroups per CU, i.e. two wavefronts per SIMD, and the 54 MFMAs of a slab (1.4 us) are
  // shorter than an HBM round trip under load -- with one slab in flight (round 3, first version) the wavefronts waited for data
  // (3.2 TB/s for the fused rotation at 216^3, 1.5 TB/s before any prefetch).
  constexpr int NPRE = ROT_PRE > 0 ? ROT_PRE : 1;
  constexpr int DEPTH = 2;
  const int nel = 16 * p4;
  double pre[DEPTH][NPRE];
  auto load_slab = [&](double(&dst)[NPRE], int64_t r0s) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
      const int idx = lane + (u << 6);
      const int i = idx / p4, k = idx - i * p4;
      const int64_t row = r0s + i;
      dst[u] = *((idx < nel && row < c.r1 && k < p) ? U + row * ldu + k : ddm_zero_page);
    }
  };
  auto process = [&](const double(&src)[NPRE], int64_t r0) __attribute__((always_inline)) {
    if (ROT_PRE > 0) {
#pragma unroll
      for (int u = 0; u < NPRE; ++u) {
        const int idx = lane + (u << 6);
        if (idx < nel) {
          const int i = idx / p4;
          Us[i * ustride + (idx - i * p4)] = src[u];
        }
      }
    } else { // stage 16 rows x p of U: row by row, lanes along the row (coalesced)
      for (int i = 0; i < 16; ++i) {
        const int64_t row = r0 + i;
        for (int k = lane; k < p4; k += 64) Us[i * ustride + k] = (row < c.r1 && k < p) ? U[row * ldu + k] : 0.0;
      }
    }
    // the wave reads back what its own lanes wrote: LDS operations of a wave complete in order, the wait covers the LDS counter only
    // (the prefetched slabs stay in flight) and the barrier keeps the compiler from moving the reads up
    __builtin_amdgcn_s_waitcnt(ROT_PRE > 0 ? 0xC07F : 0);
    __builtin_amdgcn_wave_barrier();
    v4d acc[TQ];
#pragma unroll
    for (int t = 0; t < TQ; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
    const double *ua = Us + lc * ustride + lr, *yb = Ys + lr * q16 + lc;
    for (int k0 = 0; k0 < p4; k0 += 4) {
      const double a = ua[k0];
      double b[TQ];
#pragma unroll
      for (int t = 0; t < TQ; ++t) b[t] = yb[k0 * q16 + (t << 4)];
#pragma unroll
      for (int t = 0; t < TQ; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[t], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
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
  };
  const int64_t rw = c.r0 + 16 * wave;
  if (ROT_PRE > 0) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_slab(pre[d], rw + 64 * d); // (rows beyond the chunk load zeros)
    for (int64_t r0 = rw; r0 < c.r1; r0 += 64 * DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int64_t rd = r0 + 64 * d;
        if (rd < c.r1) { // wave-uniform
          process(pre[d], rd);
          load_slab(pre[d], rd + 64 * DEPTH);
        }
      }
    }
  } else {
    for (int64_t r0 = rw; r0 < c.r1; r0 += 64) process(pre[0], r0);
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
