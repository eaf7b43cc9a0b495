// GenEO coarse-basis builder on the device (included by ddm_hip.hip; C ABI: ddm_geneo_basis in include/ddm_hip.h).
//
// Reference: GenEOCoarseSpace::setup_geneo_impl (dune/ddm/coarsespaces/coarse_spaces.hh:319-331): C = D B_neu D
// (detail::scale_matrix_with_pou, :74-96), the lowest nev eigenpairs of A_neu x = lambda C x (solve_gevp ->
// spectra_gevp_op, dune/ddm/eigensolvers/spectra.hh:111-254: shift-invert implicitly restarted Lanczos, one vector at a
// time, on a sparse LU of A - sigma C), then v <- D v / ||D v||_2 (detail::finalize_eigenvectors, :52-61); the caller zeroes
// the Dirichlet entries (examples/poisson.cc:235-238).
//
// Here (SURVEY.md App. A.9 allows a block method whose span matches): LOBPCG on the reciprocal pencil
//     C~ x = mu A~ x,   A~ = A_neu + sigma C~   (largest mu;  lambda = 1 / mu - sigma,  identical eigenvectors),
// all subdomains of the rank in lock-step on their concatenated row-major blocks S = [X | W | P] (n x 3m, m = nev + extra):
//   * products with A~ and C~: row-major SpMM (k_spmm_rowmajor);
//   * preconditioner T ~ A~^-1 applied to all m columns at once by the multi-RHS triangular solves: the sparse Cholesky factor
//     of A~ when the symbolic analysis says it is affordable (ddm_chol_create: T is then exact and the iteration is block
//     inverse iteration with Rayleigh-Ritz acceleration -- the device analogue of the reference's shift-invert), else ILU(0);
//   * Gram matrices S^T (A~ S), S^T (C~ S) and all inner products by the FP64-MFMA split-K kernel k_gram_mfma; the basis update
//     [X P] <- S Y by k_rotate_mfma (S, A~S, C~S rotated in one launch);
//   * only the p x p (p = 3m) projected eigenproblems run on the host (dense_host.hpp, one thread per subdomain), in the
//     rank-revealing form of the robust LOBPCG (basis truncation instead of Cholesky factorisations that break down).
// C~ is C without the rows / columns of global Dirichlet DoFs: after the symmetric elimination (examples/pdelab_helper.hh:33-46)
// those are decoupled unit modes that zero_at_dirichlet turns into zero vectors (a singular R A R^T in the reference).
//
// Convergence test per wanted pair, for all subdomains: with the exact T the relative residual of the inverted operator in the
// A~-norm, sqrt(r^T A~^-1 r) / mu  (r = C~ x - mu A~ x, x^T A~ x = 1) -- the quantity Spectra bounds by tol for its B-norm
// Lanczos residual (HermEigsBase.h:158-175); with ILU(0) the Euclidean relative residual ||r|| / (mu ||A~ x||).
#pragma once
#include <functional>

#include "dense_host.hpp"
#include "geneo_kernels.hpp"

static constexpr int64_t GENEO_CHUNK_ROWS = 2048;

extern "C" int ddm_geneo_params_default(ddm_geneo_params *p)
{
  if (!p) return DDM_EINVAL;
  p->nev = 16;            // eigensolver_params.hh:11
  p->nev_max = 32;        // 2 nev (:24)
  p->tolerance = 1e-5;    // :14
  p->shift = 1e-3;        // :15
  p->threshold = -0.5;    // :16
  p->maxit = 400;
  p->extra = 4;
  p->seed = 0;
  p->preconditioner = 0;
  // The exact preconditioner (sparse Cholesky of the pencil) is decided PER RANK by time and memory, not by a fixed size: a time budget
  // (DDM_GENEO_DIRECT_SECONDS, default 6 s: what the ILU(0)-preconditioned iteration costs at the headline size) times the measured
  // rate of the device factorisation (1.1e13 multiply-adds / s, sn_chol.hpp) gives this bound on the multiply-adds -- 8 x 111^3 blocks
  // on one GPU (2.1e14, 154 GB) stay with ILU(0), ONE 111^3 block per GPU (2.6e13, 19 GB: the 8-GPU layout) gets the exact factor
  // in ~2.4 s and ~20 block iterations -- and the panels must fit into 85 % of the free device memory (sn_direct_create).
  {
    double seconds = 6.0;
    if (const char *e = std::getenv("DDM_GENEO_DIRECT_SECONDS")) seconds = std::atof(e);
    p->max_direct_flops = seconds * 1.1e13;
  }
  p->verbose = 0;
  p->raw = 0;
  return DDM_OK;
}

namespace {

__global__ void k_geneo_random(int64_t n, int m, int64_t ld, unsigned long long seed, const double *__restrict__ mask, double *__restrict__ X)
{
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n * m) return;
  const int64_t i = t / m;
  const int j = (int)(t - i * m);
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(t + 1); // splitmix64 of the entry index
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  X[i * ld + j] = ((double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5) * mask[i];
}

// widest block the eigensolver iterates (nev + extra): the dense kernels work in column panels beyond 48 (round 3; the threshold mode
// of the reference doubles nev up to nev_max, spectra.hh:157-163), the limit is the memory of the six n x 3m blocks
constexpr int GENEO_MAX_BLOCK = 132;
struct GeneoWork {
  ddm_ctx *ctx = nullptr;
  int64_t n = 0;
  int nsub = 0, m = 0, p = 0, nchunk = 0;
  std::vector<void *> allocs;
  GChunk *chunks = nullptr;
  int32_t *sub_chunk_ptr = nullptr, *sub_of_row = nullptr;
  double *partial = nullptr;
  ~GeneoWork()
  {
    for (void *q : allocs) (void)hipFree(q);
  }
  template <class T>
  int alloc(T **ptr, size_t count)
  {
    *ptr = nullptr;
    if (hipMalloc((void **)ptr, sizeof(T) * std::max<size_t>(count, 1)) != hipSuccess) return fail(ctx, DDM_EHIP, "GenEO: device allocation of %zu bytes failed", sizeof(T) * count);
    allocs.push_back(*ptr);
    return DDM_OK;
  }
  // G[sub] = U^T V per subdomain (pu x pv row-major, nsub matrices).  Blocks wider than the register tiles of the kernel (128 x 80)
  // are computed in column panels that land in their sub-block of the per-chunk partial matrices.
  int gram(const double *U, int64_t ldu, int pu, const double *V, int64_t ldv, int pv, double *G)
  {
    const int64_t pp = (int64_t)pu * pv;
    if (pu <= 32 && pv <= 32) {
      const bool same = U == V && ldu == ldv && pu == pv;
#define DDM_GRAM_SMALL(SAME, A1, B1) hipLaunchKernelGGL((k_gram_small<SAME, A1, B1>), dim3(nchunk), dim3(256), 0, ctx->stream, chunks, U, ldu, pu, V, ldv, pv, partial, pp, pv)
      if (same) {
        if (pu > 16) DDM_GRAM_SMALL(true, true, true);
        else DDM_GRAM_SMALL(true, false, false);
      } else if (pu > 16) {
        if (pv > 16) DDM_GRAM_SMALL(false, true, true);
        else DDM_GRAM_SMALL(false, true, false);
      } else {
        if (pv > 16) DDM_GRAM_SMALL(false, false, true);
        else DDM_GRAM_SMALL(false, false, false);
      }
#undef DDM_GRAM_SMALL
    } else if (pu <= 128 && pv <= 80)
      hipLaunchKernelGGL((k_gram_mfma<2, 5>), dim3(nchunk), dim3(256), 0, ctx->stream, chunks, U, ldu, pu, V, ldv, pv, partial, pp, pv, 0, 0);
    else if (pu <= 144 && pv <= 144)
      hipLaunchKernelGGL((k_gram_mfma<3, 9>), dim3(nchunk), dim3(256), 0, ctx->stream, chunks, U, ldu, pu, V, ldv, pv, partial, pp, pv, 0, 0);
    else
      for (int i0 = 0; i0 < pu; i0 += 128)
        for (int j0 = 0; j0 < pv; j0 += 80)
          hipLaunchKernelGGL((k_gram_mfma<2, 5>), dim3(nchunk), dim3(256), 0, ctx->stream, chunks, U + i0, ldu, std::min(128, pu - i0), V + j0, ldv, std::min(80, pv - j0),
                             partial, pp, pv, i0, j0);
    hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)((nsub * pp + 255) / 256)), dim3(256), 0, ctx->stream, nsub, sub_chunk_ptr, pp, partial, pp, G);
    HIPCHECK(ctx, hipGetLastError());
    return DDM_OK;
  }
  // G1[sub] = U^T V1, G2[sub] = U^T V2 (p x p each) for products that are symmetric by construction (V1 = A~ U, V2 = C~ U): one pass over
  // U, upper tiles only -- the entries BELOW the diagonal tiles of G1 / G2 are not defined, the caller mirrors the upper triangle
  // (gram2_mirror_host).  The partial buffer must hold 2 p p doubles per chunk.  p > 80: two general products.
  bool gram2_sym(const double *U, int64_t ldu, const double *V1, const double *V2, int64_t ldv, int p, double *G1, double *G2)
  {
    const int64_t pp = (int64_t)p * p;
    if (p > 80) {
      (void)gram(U, ldu, p, V1, ldv, p, G1);
      (void)gram(U, ldu, p, V2, ldv, p, G2);
      return false;
    }
    switch ((p + 15) >> 4) {
    case 1: hipLaunchKernelGGL(k_gram2_sym<1>, dim3(nchunk), dim3(256), 0, ctx->stream, chunks, U, ldu, V1, V2, ldv, p, partial); break;
    case 2: hipLaunchKernelGGL(k_gram2_sym<2>, dim3(nchunk), dim3(256), 0, ctx->stream, chunks, U, ldu, V1, V2, ldv, p, partial); break;
    case 3: hipLaunchKernelGGL(k_gram2_sym<3>, dim3(nchunk), dim3(256), 0, ctx->stream, chunks, U, ldu, V1, V2, ldv, p, partial); break;
    case 4: hipLaunchKernelGGL(k_gram2_sym<4>, dim3(nchunk), dim3(256), 0, ctx->stream, chunks, U, ldu, V1, V2, ldv, p, partial); break;
    default: hipLaunchKernelGGL(k_gram2_sym<5>, dim3(nchunk), dim3(256), 0, ctx->stream, chunks, U, ldu, V1, V2, ldv, p, partial); break;
    }
    hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)((nsub * pp + 255) / 256)), dim3(256), 0, ctx->stream, nsub, sub_chunk_ptr, pp, (const double *)partial, 2 * pp, G1);
    hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)((nsub * pp + 255) / 256)), dim3(256), 0, ctx->stream, nsub, sub_chunk_ptr, pp, (const double *)(partial + pp), 2 * pp, G2);
    return true;
  }
  static void gram2_mirror_host(int p, double *G)
  {
    for (int i = 1; i < p; ++i)
      for (int j = 0; j < i; ++j) G[(size_t)i * p + j] = G[(size_t)j * p + i];
  }
  // Out_k[:, 0:q) = (Base_k -) U_k[:, 0:pk) Y[sub]  for k < narr.  Y: nsub matrices pk x q, row-major.  More than 48 output columns
  // or more than 80 inner columns run as panels: 48 output columns per launch, the inner dimension in pieces of 72 whose products are
  // added onto the output.
  int rotate(int narr, const double *const *U, double *const *Out, const double *const *Base, int64_t ldu, int pk, const double *Y, int q, int64_t ldo, int64_t ldb,
             int gap_from = 1 << 30, int gap = 0)
  {
    const int KP = pk <= 80 ? pk : 72;
    for (int j0 = 0; j0 < q; j0 += 16 * ROT_TQ) {
      const int qq = std::min(16 * ROT_TQ, q - j0);
      for (int k0 = 0; k0 < pk; k0 += KP) {
        const int kk = std::min(KP, pk - k0);
        RotArgs a;
        for (int k = 0; k < 3; ++k) {
          a.U[k] = k < narr ? U[k] + k0 : nullptr;
          a.Out[k] = k < narr ? Out[k] : nullptr;
          a.Base[k] = (k < narr && Base) ? Base[k] : nullptr;
        }
        const int mode = k0 == 0 ? (Base ? 1 : 0) : (Base ? 2 : 3);
        const int p4 = (kk + 3) & ~3, q16 = ((qq + 15) >> 4) << 4;
        const size_t lds = sizeof(double) * ((size_t)p4 * q16 + 4 * 16 * (size_t)(p4 + 1));
#define DDM_ROTATE(PRE, TQ) hipLaunchKernelGGL((k_rotate_mfma<PRE, TQ>), dim3(nchunk, narr), dim3(256), lds, ctx->stream, chunks, a, ldu, kk, Y, qq, ldo, ldb, gap_from, gap, q, pk, k0, j0, mode)
        const int tq = q16 >> 4;
        if (16 * p4 <= 20 * 64) {
          if (tq == 1) DDM_ROTATE(20, 1);
          else if (tq == 2) DDM_ROTATE(20, 2);
          else DDM_ROTATE(20, 3);
        } else {
          if (tq == 1) DDM_ROTATE(0, 1);
          else if (tq == 2) DDM_ROTATE(0, 2);
          else DDM_ROTATE(0, 3);
        }
#undef DDM_ROTATE
      }
    }
    HIPCHECK(ctx, hipGetLastError());
    return DDM_OK;
  }
};

// A~ = A + sigma C~ and C~ = D B D without Dirichlet rows / columns, on the union pattern, as host CSR
static void build_pencil_host(const ddm_csr *A, const ddm_csr *B, const double *pou, const uint8_t *dir, double sigma, hvec<int64_t> &rpT,
                              hvec<int32_t> &ciT, hvec<double> &vaT, hvec<double> &vaC)
{
  const int64_t n = A->nrows;
  rpT.resize((size_t)n + 1);
  rpT[0] = 0;
  const unsigned hw = host_threads();
  const int nth = (int)std::min<int64_t>(hw, std::max<int64_t>(1, n / 65536));
  std::vector<std::thread> th;
  // pass 1: sizes of the merged rows (threads), then the prefix sum
  for (int t = 0; t < nth; ++t)
    th.emplace_back([&, t]() {
      for (int64_t i = n * t / nth; i < n * (t + 1) / nth; ++i) {
        int64_t a = A->h_rp[i], b = B->h_rp[i], cnt = 0;
        const int64_t a1 = A->h_rp[i + 1], b1 = B->h_rp[i + 1];
        while (a < a1 || b < b1) {
          const int32_t ca = a < a1 ? A->h_ci[a] : INT32_MAX, cb = b < b1 ? B->h_ci[b] : INT32_MAX;
          a += ca <= cb;
          b += cb <= ca;
          ++cnt;
        }
        rpT[i + 1] = cnt;
      }
    });
  for (auto &t : th) t.join();
  th.clear();
  for (int64_t i = 0; i < n; ++i) rpT[i + 1] += rpT[i];
  ciT.resize((size_t)rpT[n]);
  vaT.resize((size_t)rpT[n]);
  vaC.resize((size_t)rpT[n]);
  for (int t = 0; t < nth; ++t)
    th.emplace_back([&, t]() {
      for (int64_t i = n * t / nth; i < n * (t + 1) / nth; ++i) {
        int64_t a = A->h_rp[i], b = B->h_rp[i], q = rpT[i];
        const int64_t a1 = A->h_rp[i + 1], b1 = B->h_rp[i + 1];
        const bool di = dir && dir[i];
        while (a < a1 || b < b1) {
          const int32_t ca = a < a1 ? A->h_ci[a] : INT32_MAX, cb = b < b1 ? B->h_ci[b] : INT32_MAX;
          const int32_t c = std::min(ca, cb);
          double va = 0.0, vc = 0.0;
          if (ca == c) va = A->h_va[a++];
          if (cb == c) {
            vc = (di || (dir && dir[c])) ? 0.0 : B->h_va[b] * pou[i] * pou[c]; // scale_matrix_with_pou (coarse_spaces.hh:74-96)
            ++b;
          }
          ciT[(size_t)q] = c;
          vaC[(size_t)q] = vc;
          vaT[(size_t)q] = va + sigma * vc;
          ++q;
        }
      }
    });
  for (auto &t : th) t.join();
}

// X = keep .* X - Y  (rows: keep = 0 / 1): tail of the harmonic projection / extension
__global__ void k_geneo_project(int64_t n, int m, const double *__restrict__ keep, const double *__restrict__ Y, int64_t ldy, double *__restrict__ X, int64_t ldx)
{
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n * m) return;
  const int64_t i = t / m;
  const int j = (int)(t - i * m);
  X[i * ldx + j] = keep[i] * X[i * ldx + j] - Y[i * ldy + j];
}
// Y = w .* X
__global__ void k_geneo_rowscale_to(int64_t n, int m, const double *__restrict__ w, const double *__restrict__ X, int64_t ldx, double *__restrict__ Y, int64_t ldy)
{
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n * m) return;
  const int64_t i = t / m;
  const int j = (int)(t - i * m);
  Y[i * ldy + j] = w[i] * X[i * ldx + j];
}

static bool csr_values_symmetric(const ddm_csr *A)
{
  const int64_t n = A->nrows;
  for (int64_t i = 0; i < n; ++i)
    for (int64_t k = A->h_rp[i]; k < A->h_rp[i + 1]; ++k) {
      const int64_t j = A->h_ci[k];
      if (j <= i) continue;
      const auto b = A->h_ci.begin() + A->h_rp[j], e = A->h_ci.begin() + A->h_rp[j + 1];
      const auto it = std::lower_bound(b, e, (int32_t)i);
      const double vt = (it != e && *it == i) ? A->h_va[(size_t)(it - A->h_ci.begin())] : 0.0;
      if (std::fabs(vt - A->h_va[k]) > 1e-12 * (std::fabs(vt) + std::fabs(A->h_va[k]))) return false;
    }
  return true;
}

} // namespace

// Energy-minimal (a-harmonic) extension u_i = -A_ii^-1 A_ib u_b (EnergyMinimalExtension, energy_minimal_extension.hh:36-229) for
// row-major device blocks: the interior block is factorised once by the sparse direct solver, as part of the n x n matrix
//     A^ = [A_ii 0; 0 I]   (rows / columns outside the interior replaced by the identity),
// so that every operand is a full-length block and no index gathers are needed:  X <- keep .* X - A^^-1 (G_ib X)  with G_ib the
// interior rows x boundary columns of A.  The same object projects onto the a-harmonic subspace in the constrained eigensolver
// (MsGFEMCoarseSpace): P X = keep_b .* X - A^^-1 G_ib X,  P^T R = keep_b .* R - G_bi A^^-T (interior .* R).
struct ddm_harmonic {
  int64_t n = 0;
  ddm_csr *Gib = nullptr, *Gbi = nullptr;
  ddm_ilu0 *F = nullptr;
  double *keep = nullptr;   // 1 outside the interior (rows the extension leaves alone)
  double *keep_b = nullptr; // 1 on boundary rows only (projection: rows that are neither interior nor boundary are zeroed)
  double *isint = nullptr;  // 1 on interior rows
  double *t1 = nullptr, *t2 = nullptr;
  int tcols = 0;
  bool symmetric = true;
};
extern "C" void ddm_harmonic_destroy(ddm_harmonic *H)
{
  if (!H) return;
  ddm_csr_destroy(H->Gib);
  ddm_csr_destroy(H->Gbi);
  ddm_ilu0_destroy(H->F);
  (void)hipFree(H->keep);
  (void)hipFree(H->keep_b);
  (void)hipFree(H->isint);
  (void)hipFree(H->t1);
  (void)hipFree(H->t2);
  delete H;
}
// cls[i]: 0 = interior, 1 = boundary, anything else = neither (its values count as zero in the right-hand side, :109-118)
static int harmonic_create_impl(ddm_ctx *ctx, const ddm_csr *A, int64_t nblocks, const int64_t *block_ptr, const uint8_t *cls, bool want_transpose, ddm_harmonic **out)
{
  const int64_t n = A->nrows;
  std::vector<int64_t> rpI((size_t)n + 1, 0), rpG((size_t)n + 1, 0), rpT((size_t)n + 1, 0);
  for (int64_t i = 0; i < n; ++i) {
    int64_t ci = 0, cg = 0;
    if (cls[i] == 0) {
      for (int64_t k = A->h_rp[i]; k < A->h_rp[i + 1]; ++k) {
        const uint8_t c = cls[A->h_ci[k]];
        ci += c == 0;
        cg += c == 1;
        if (c == 1) rpT[(size_t)A->h_ci[k] + 1] += 1;
      }
      if (ci == 0) return fail(ctx, DDM_ENUMERIC, "harmonic extension: interior row %lld has no interior entries", (long long)i);
    } else ci = 1;
    rpI[(size_t)i + 1] = rpI[(size_t)i] + ci;
    rpG[(size_t)i + 1] = rpG[(size_t)i] + cg;
  }
  for (int64_t i = 0; i < n; ++i) rpT[(size_t)i + 1] += rpT[(size_t)i];
  std::vector<int32_t> ciI((size_t)rpI[n]), ciG((size_t)rpG[n]), ciT((size_t)rpT[n]);
  std::vector<double> vaI((size_t)rpI[n]), vaG((size_t)rpG[n]), vaT((size_t)rpT[n]);
  std::vector<int64_t> fill(rpT.begin(), rpT.end() - 1);
  for (int64_t i = 0; i < n; ++i) {
    int64_t qi = rpI[(size_t)i], qg = rpG[(size_t)i];
    if (cls[i] != 0) {
      ciI[(size_t)qi] = (int32_t)i;
      vaI[(size_t)qi] = 1.0;
      continue;
    }
    for (int64_t k = A->h_rp[i]; k < A->h_rp[i + 1]; ++k) {
      const int32_t j = A->h_ci[k];
      if (cls[j] == 0) {
        ciI[(size_t)qi] = j;
        vaI[(size_t)qi++] = A->h_va[k];
      } else if (cls[j] == 1) {
        ciG[(size_t)qg] = j;
        vaG[(size_t)qg++] = A->h_va[k];
        const int64_t q = fill[(size_t)j]++; // rows i ascending => sorted columns in the transpose
        ciT[(size_t)q] = (int32_t)i;
        vaT[(size_t)q] = A->h_va[k];
      }
    }
  }
  ddm_harmonic *H = new ddm_harmonic;
  H->n = n;
  ddm_csr *Ahat = nullptr;
  int rc = ddm_csr_create(ctx, n, n, rpI.data(), ciI.data(), vaI.data(), &Ahat);
  if (!rc) {
    H->symmetric = csr_values_symmetric(Ahat);
    rc = direct_create_impl(ctx, Ahat, nblocks, block_ptr, H->symmetric ? 0 : 1, 0.0, /*setup_use=*/true, &H->F);
  }
  ddm_csr_destroy(Ahat);
  if (!rc) rc = ddm_csr_create(ctx, n, n, rpG.data(), ciG.data(), vaG.data(), &H->Gib);
  if (!rc && want_transpose) rc = ddm_csr_create(ctx, n, n, rpT.data(), ciT.data(), vaT.data(), &H->Gbi);
  if (!rc) {
    std::vector<double> k0((size_t)n), k1((size_t)n), k2((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
      k0[(size_t)i] = cls[i] == 0 ? 0.0 : 1.0;
      k1[(size_t)i] = cls[i] == 1 ? 1.0 : 0.0;
      k2[(size_t)i] = cls[i] == 0 ? 1.0 : 0.0;
    }
    rc = upload(ctx, k0.data(), n, &H->keep);
    if (!rc) rc = upload(ctx, k1.data(), n, &H->keep_b);
    if (!rc) rc = upload(ctx, k2.data(), n, &H->isint);
  }
  if (rc) {
    ddm_harmonic_destroy(H);
    return rc;
  }
  *out = H;
  return DDM_OK;
}
static int harmonic_reserve(ddm_ctx *ctx, ddm_harmonic *H, int m)
{
  if (m <= H->tcols) return DDM_OK;
  (void)hipFree(H->t1);
  (void)hipFree(H->t2);
  H->t1 = H->t2 = nullptr;
  H->tcols = 0;
  HIPCHECK(ctx, hipMalloc((void **)&H->t1, sizeof(double) * (size_t)std::max<int64_t>(H->n, 1) * m));
  HIPCHECK(ctx, hipMalloc((void **)&H->t2, sizeof(double) * (size_t)std::max<int64_t>(H->n, 1) * m));
  H->tcols = m;
  return DDM_OK;
}
// X <- keep .* X - A^^-1 G_ib X  (keep = rows outside the interior, or boundary rows only)
static int harmonic_apply(ddm_ctx *ctx, ddm_harmonic *H, int m, double *X, int64_t ldx, bool boundary_only)
{
  DDMCHECK(harmonic_reserve(ctx, H, m));
  DDMCHECK(csr_mm_ld(ctx, H->Gib, m, X, ldx, H->t1, m));
  DDMCHECK(ilu0_solve_multi_ld(ctx, H->F, m, H->t1, m, H->t2, m));
  hipLaunchKernelGGL(k_geneo_project, dim3((unsigned)((H->n * (int64_t)m + 255) / 256)), dim3(256), 0, ctx->stream, H->n, m, boundary_only ? H->keep_b : H->keep,
                     (const double *)H->t2, (int64_t)m, X, ldx);
  HIPCHECK(ctx, hipGetLastError());
  return DDM_OK;
}
// R <- keep_b .* R - G_bi A^^-1 (interior .* R)   (transpose of the projection; symmetric A only)
static int harmonic_apply_transposed(ddm_ctx *ctx, ddm_harmonic *H, int m, double *R, int64_t ldr)
{
  DDMCHECK(harmonic_reserve(ctx, H, m));
  const unsigned g = (unsigned)((H->n * (int64_t)m + 255) / 256);
  hipLaunchKernelGGL(k_geneo_rowscale_to, dim3(g), dim3(256), 0, ctx->stream, H->n, m, H->isint, (const double *)R, ldr, H->t1, (int64_t)m);
  DDMCHECK(ilu0_solve_multi_ld(ctx, H->F, m, H->t1, m, H->t2, m));
  DDMCHECK(csr_mm_ld(ctx, H->Gbi, m, H->t2, m, H->t1, m));
  hipLaunchKernelGGL(k_geneo_project, dim3(g), dim3(256), 0, ctx->stream, H->n, m, H->keep_b, (const double *)H->t1, (int64_t)m, R, ldr);
  HIPCHECK(ctx, hipGetLastError());
  return DDM_OK;
}
extern "C" int ddm_harmonic_create(ddm_ctx *ctx, const ddm_csr *A, int64_t nblocks, const int64_t *block_ptr, int64_t n_interior, const int64_t *interior_host,
                                   int64_t n_boundary, const int64_t *boundary_host, ddm_harmonic **out)
{
  if (!ctx || !A || !out || nblocks < 1 || !block_ptr || n_interior < 0 || n_boundary < 0 || (n_interior && !interior_host) || (n_boundary && !boundary_host))
    return fail(ctx, DDM_EINVAL, "ddm_harmonic_create: bad arguments");
  if (A->nrows != A->ncols) return fail(ctx, DDM_EINVAL, "ddm_harmonic_create: square matrix expected");
  const int64_t n = A->nrows;
  std::vector<uint8_t> cls((size_t)n, 2);
  for (int64_t k = 0; k < n_boundary; ++k) {
    if (boundary_host[k] < 0 || boundary_host[k] >= n) return fail(ctx, DDM_EINVAL, "ddm_harmonic_create: boundary index out of range");
    cls[(size_t)boundary_host[k]] = 1; // listed twice is fine (coarse_spaces.hh:582-589 produces duplicates)
  }
  for (int64_t k = 0; k < n_interior; ++k) {
    if (interior_host[k] < 0 || interior_host[k] >= n) return fail(ctx, DDM_EINVAL, "ddm_harmonic_create: interior index out of range");
    if (cls[(size_t)interior_host[k]] == 1) return fail(ctx, DDM_EINVAL, "ddm_harmonic_create: index %lld is both interior and boundary", (long long)interior_host[k]);
    cls[(size_t)interior_host[k]] = 0;
  }
  return harmonic_create_impl(ctx, A, nblocks, block_ptr, cls.data(), false, out);
}
extern "C" int ddm_harmonic_extend(ddm_ctx *ctx, ddm_harmonic *H, int nrhs, double *X, int64_t ldx)
{
  if (!ctx || !H || !X || nrhs < 1 || ldx < nrhs) return fail(ctx, DDM_EINVAL, "ddm_harmonic_extend: bad arguments");
  for (int c0 = 0; c0 < nrhs; c0 += 48) DDMCHECK(harmonic_apply(ctx, H, std::min(48, nrhs - c0), X + c0, ldx, false));
  return DDM_OK;
}

static int geneo_run(ddm_ctx *ctx, const ddm_csr *A_neu, const ddm_csr *B_neu, int64_t nsub, const int64_t *sub_ptr, const double *pou_host,
                     const uint8_t *dirichlet_host, const ddm_geneo_params &P, int nev, double *basis_dev /* nev x n */, double *eig_host /* nsub x nev */,
                     ddm_geneo_info *info, ddm_harmonic *con = nullptr /* iterate in the a-harmonic subspace */, const double *pou_pencil_host = nullptr,
                     const std::function<int(int, const double *, int64_t, double *, int64_t)> *op_C = nullptr /* replaces the product with C~ */)
{
  const int64_t n = A_neu->nrows;
  const int m = nev + std::max(P.extra, 1);
  const int p = 3 * m;
  if (m > GENEO_MAX_BLOCK) return fail(ctx, DDM_ENOTIMPL, "GenEO: nev + extra = %d exceeds %d vectors per subdomain", m, GENEO_MAX_BLOCK);
  for (int64_t s = 0; s < nsub; ++s)
    if (sub_ptr[s + 1] - sub_ptr[s] < 3 * (int64_t)m) return fail(ctx, DDM_EINVAL, "GenEO: subdomain %lld has fewer than 3 (nev + extra) = %d rows", (long long)s, 3 * m);
  const auto t_begin = std::chrono::steady_clock::now();
  auto since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  // "auto": whether the sparse direct factor of A~ is affordable at all is estimated on a helper thread from the first separator of the
  // largest block of A_neu's pattern (the pencil's pattern, or a superset of B_neu's), while the pencil is being assembled
  struct Joiner { // (an early error return must not leave a helper thread running on this frame's variables)
    std::thread &t;
    ~Joiner()
    {
      if (t.joinable()) t.join();
    }
  };
  double probe_flops = 0.0;
  std::thread probe_thread;
  Joiner probe_joiner{probe_thread};
  if (P.preconditioner == 0 && P.max_direct_flops > 0.0 && nsub > 1 && !std::getenv("DDM_DIRECT_ENGINE"))
    probe_thread = std::thread([&]() { probe_flops = sn_probe_largest_block(A_neu->h_rp.data(), A_neu->h_ci.data(), nsub, sub_ptr, false); });
  // ---- pencil ----
  hvec<int64_t> rpT;
  hvec<int32_t> ciT;
  hvec<double> vaT, vaC;
  build_pencil_host(A_neu, B_neu, pou_pencil_host ? pou_pencil_host : pou_host, dirichlet_host, P.shift, rpT, ciT, vaT, vaC);
  const double t_pencil = since(t_begin);
  struct Owned {
    ddm_csr *At = nullptr, *C = nullptr;
    ddm_ilu0 *T = nullptr;
    ~Owned()
    {
      ddm_ilu0_destroy(T);
      ddm_csr_destroy(At); // (joins the upload thread that also fills C)
      ddm_csr_destroy(C);
    }
  } own;
  // the host arrays move into At; its device copy and the values of C~ (same pattern, device only) are uploaded by a helper thread
  // while this one factorises / analyses on the host
  own.At = csr_adopt(ctx, n, std::move(rpT), std::move(ciT), std::move(vaT), std::move(vaC), &own.C, nsub, sub_ptr);
  const double t_upload = since(t_begin);
  // A~ X and C~ X of the same block in one pass (the two matrices share their pattern: build_pencil_host)
  auto apply_AC = [&](int mm, const double *X, int64_t ldx, double *YA, double *YC, int64_t ldy) -> int {
    if (!op_C) return csr_mm2_ld(ctx, own.At, own.C, mm, X, ldx, YA, YC, ldy);
    DDMCHECK(csr_mm_ld(ctx, own.At, mm, X, ldx, YA, ldy));
    return (*op_C)(mm, X, ldx, YC, ldy);
  };
  // ---- preconditioner ----
  // "auto": the ILU(0) factorisation starts on a helper thread while this one orders and analyses for the sparse direct factor
  // (at the headline size the analysis ends in "too expensive" after 0.9 s and the ILU(0) setup takes 2.3 s); whichever is not
  // needed is dropped
  int direct = 0;
  ddm_ilu0 *T_ilu = nullptr;
  int rc_ilu = DDM_OK;
  std::string err_ilu;
  std::thread ilu_thread;
  Joiner ilu_joiner{ilu_thread};
  auto start_ilu = [&]() {
    ilu_thread = std::thread([&]() {
      (void)hipSetDevice(ctx->device);
      rc_ilu = ilu0_create_impl(ctx, own.At, nsub, sub_ptr, /*multi_rhs_only=*/true, &T_ilu);
      if (rc_ilu) err_ilu = last_error_of_this_thread();
    });
  };
  if (P.preconditioner != 2) start_ilu();
  int rc_direct = DDM_OK;
  std::string why_not;
  if (probe_thread.joinable()) probe_thread.join();
  if (P.preconditioner == 0 && probe_flops > 4.0 * P.max_direct_flops) { // declined by the early probe: no second analysis
    char buf[256];
    std::snprintf(buf, sizeof buf, "sparse direct solver: the factorisation needs about %.1g flops (estimate from the first separator of the largest block; limit %.3g)",
                  probe_flops, P.max_direct_flops);
    why_not = buf;
    rc_direct = DDM_ENOTIMPL;
  } else if (P.preconditioner != 1) {
    rc_direct = direct_create_impl(ctx, own.At, nsub, sub_ptr, 0, P.preconditioner == 2 ? 0.0 : P.max_direct_flops, /*setup_use=*/true, &own.T);
    if (rc_direct == DDM_OK) direct = 1;
    else why_not = last_error_of_this_thread();
  }
  const double t_direct = since(t_begin);
  if (ilu_thread.joinable()) ilu_thread.join();
  if (direct) {
    ddm_ilu0_destroy(T_ilu); // (speculative work, not needed)
  } else {
    if (P.preconditioner != 1 && (P.preconditioner == 2 || (rc_direct != DDM_ENOTIMPL && rc_direct != DDM_ENUMERIC))) {
      ddm_ilu0_destroy(T_ilu);
      return fail(ctx, rc_direct, "%s", why_not.c_str());
    }
    if (P.preconditioner != 1 && P.verbose) std::fprintf(stderr, "[ddm geneo] sparse Cholesky not used (%s): ILU(0) preconditioner\n", why_not.c_str());
    if (rc_ilu) return fail(ctx, rc_ilu, "%s", err_ilu.c_str());
    own.T = T_ilu;
  }
  // ILU(0) as the preconditioner of the block iteration: single-precision sweeps (the iteration only needs a fixed search direction
  // W = T r; eigenpairs and residuals are computed in double).  DDM_GENEO_ILU_F64=1 keeps the sweeps in double.
  const bool prec_f32 = !direct && !std::getenv("DDM_GENEO_ILU_F64");
  const int refresh_period = std::getenv("DDM_GENEO_REFRESH") ? std::max(1, std::atoi(std::getenv("DDM_GENEO_REFRESH"))) : (direct ? 2 : 8);
  // W <- W - X (A~X)^T W before the Rayleigh-Ritz step: twice with the exact T (W = A~^-1 r lies almost in span X near convergence: on
  // the elasticity pencil one pass gave 68-81 block iterations in two of eight runs, none 133 in one, against 12-18 -- measured before the products of P were refreshed, see below), once with
  // ILU(0) (216^3: the same 109 iterations and residuals with two, one or no pass; 5.6 / 5.2 / 4.9 s).  DDM_GENEO_ORTH_PASSES overrides.
  const int orth_passes = std::getenv("DDM_GENEO_ORTH_PASSES") ? std::max(0, std::atoi(std::getenv("DDM_GENEO_ORTH_PASSES"))) : (direct ? 2 : 1);
  const double t_prec = since(t_begin);
  DDMCHECK(csr_wait_upload(ctx, own.At));
  const double t_setup = since(t_begin);
  if (P.verbose)
    std::fprintf(stderr, "[ddm geneo] setup: pencil %.2f s, sparse direct attempt %.2f s (%s), rest of the ILU(0) setup (helper thread) %.2f s, rest of the matrix upload (helper thread) %.2f s\n",
                 t_pencil, t_direct - t_upload, direct ? "used" : "declined", t_prec - t_direct, t_setup - t_prec);
  // ---- work space ----
  GeneoWork W;
  W.ctx = ctx;
  W.n = n;
  W.nsub = (int)nsub;
  W.m = m;
  W.p = p;
  std::vector<GChunk> chunks;
  std::vector<int32_t> scp((size_t)nsub + 1, 0), sor((size_t)n);
  for (int64_t s = 0; s < nsub; ++s) {
    for (int64_t r = sub_ptr[s]; r < sub_ptr[s + 1]; r += GENEO_CHUNK_ROWS) chunks.push_back(GChunk{r, std::min(r + GENEO_CHUNK_ROWS, sub_ptr[s + 1]), (int32_t)s, 0});
    scp[(size_t)s + 1] = (int32_t)chunks.size();
    for (int64_t r = sub_ptr[s]; r < sub_ptr[s + 1]; ++r) sor[(size_t)r] = (int32_t)s;
  }
  W.nchunk = (int)chunks.size();
  DDMCHECK(W.alloc(&W.chunks, chunks.size()));
  DDMCHECK(W.alloc(&W.sub_chunk_ptr, scp.size()));
  DDMCHECK(W.alloc(&W.sub_of_row, (size_t)n));
  DDMCHECK(W.alloc(&W.partial, (size_t)W.nchunk * p * p * 2)); // (two products per chunk: gram2_sym)
  HIPCHECK(ctx, hipMemcpy(W.chunks, chunks.data(), sizeof(GChunk) * chunks.size(), hipMemcpyHostToDevice));
  HIPCHECK(ctx, hipMemcpy(W.sub_chunk_ptr, scp.data(), sizeof(int32_t) * scp.size(), hipMemcpyHostToDevice));
  HIPCHECK(ctx, hipMemcpy(W.sub_of_row, sor.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice));
  double *S[2], *AS[2], *CS[2], *R = nullptr, *maskd = nullptr, *poud = nullptr;
  for (int b = 0; b < 2; ++b) {
    DDMCHECK(W.alloc(&S[b], (size_t)n * p));
    DDMCHECK(W.alloc(&AS[b], (size_t)n * p));
    DDMCHECK(W.alloc(&CS[b], (size_t)n * p));
    HIPCHECK(ctx, hipMemsetAsync(S[b], 0, sizeof(double) * (size_t)n * p, ctx->stream));
    HIPCHECK(ctx, hipMemsetAsync(AS[b], 0, sizeof(double) * (size_t)n * p, ctx->stream));
    HIPCHECK(ctx, hipMemsetAsync(CS[b], 0, sizeof(double) * (size_t)n * p, ctx->stream));
  }
  DDMCHECK(W.alloc(&R, (size_t)n * m));
  DDMCHECK(W.alloc(&maskd, (size_t)n));
  DDMCHECK(W.alloc(&poud, (size_t)n));
  {
    std::vector<double> mk((size_t)n);
    for (int64_t i = 0; i < n; ++i) mk[(size_t)i] = (dirichlet_host && dirichlet_host[i]) ? 0.0 : 1.0;
    HIPCHECK(ctx, hipMemcpy(maskd, mk.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    HIPCHECK(ctx, hipMemcpy(poud, pou_host, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  }
  double *gA, *gC, *gmm[5], *svec[2], *Yd, *mud;
  DDMCHECK(W.alloc(&gA, (size_t)nsub * p * p));
  DDMCHECK(W.alloc(&gC, (size_t)nsub * p * p));
  for (int k = 0; k < 5; ++k) DDMCHECK(W.alloc(&gmm[k], (size_t)nsub * m * m));
  for (int k = 0; k < 2; ++k) DDMCHECK(W.alloc(&svec[k], (size_t)nsub * m));
  const int q2 = 2 * m; // fused rotation: [X_new | P_new]
  DDMCHECK(W.alloc(&Yd, (size_t)nsub * p * q2));
  DDMCHECK(W.alloc(&mud, (size_t)nsub * m));
  std::vector<double> hA((size_t)nsub * p * p), hC((size_t)nsub * p * p), hY((size_t)nsub * p * q2), hmu((size_t)nsub * m);
  std::vector<double> h_rr((size_t)nsub * m * m), h_rw((size_t)nsub * m * m), h_aa((size_t)nsub * m * m);
  std::vector<double> dscale((size_t)nsub * p, 1.0); // column scaling of S = [X | W | P] folded into the projected problem
  const unsigned gnm = (unsigned)((n * (int64_t)m + 255) / 256);
  const int64_t ld = p;
  // ---- initial block: random on the free DoFs, Rayleigh-Ritz on span X ----
  hipLaunchKernelGGL(k_geneo_random, dim3(gnm), dim3(256), 0, ctx->stream, n, m, ld, 0x5DEECE66Dull + (unsigned long long)P.seed, maskd, S[0]);
  if (con) DDMCHECK(harmonic_apply(ctx, con, m, S[0], ld, true));
  DDMCHECK(apply_AC(m, S[0], ld, AS[0], CS[0], ld));
  int cur = 0, it = 0, converged = 0, rank_min = p;
  double worst = 0.0;
  const double tau = 1e-11;
  bool have_residual = false;
  std::vector<int> rcs((size_t)nsub, 0);
  const auto t_iter = std::chrono::steady_clock::now();
  for (it = 0; it <= P.maxit; ++it) {
    if (it > 0) {
      // R = C X - mu A~ X ; column norms ; W = T R
      hipLaunchKernelGGL(k_geneo_residual, dim3(gnm), dim3(256), 0, ctx->stream, n, m, W.sub_of_row, mud, AS[cur], ld, CS[cur], ld, R, (int64_t)m);
      if (con) DDMCHECK(harmonic_apply_transposed(ctx, con, m, R, m)); // residual of the constrained problem: P^T r
      DDMCHECK(W.gram(R, m, m, R, m, m, gmm[0]));
      HIPCHECK(ctx, hipMemcpyAsync(h_rr.data(), gmm[0], sizeof(double) * h_rr.size(), hipMemcpyDeviceToHost, ctx->stream));
      // (W = T r with the residual columns as they are: their scaling is folded into the projected problem below, like W's and P's)
      double *Wb = S[cur] + m, *AWb = AS[cur] + m, *CWb = CS[cur] + m;
      DDMCHECK(ilu0_solve_multi_ld(ctx, own.T, m, R, m, Wb, ld, prec_f32));
      DDMCHECK(W.gram(R, m, m, Wb, ld, m, gmm[1]));   // r^T T r per column (diagonal)
      HIPCHECK(ctx, hipMemcpyAsync(h_rw.data(), gmm[1], sizeof(double) * h_rw.size(), hipMemcpyDeviceToHost, ctx->stream));
      if (con) DDMCHECK(harmonic_apply(ctx, con, m, Wb, ld, true));    // W = P T P^T r stays in the subspace
      DDMCHECK(W.gram(AS[cur], ld, m, AS[cur], ld, m, gmm[2]));
      HIPCHECK(ctx, hipMemcpyAsync(h_aa.data(), gmm[2], sizeof(double) * h_aa.size(), hipMemcpyDeviceToHost, ctx->stream));
      // W <- W - X (A~X)^T W   (orth_passes times), then A~-normalise the columns of W
      for (int pass = 0; pass < orth_passes; ++pass) {
        DDMCHECK(W.gram(AS[cur], ld, m, Wb, ld, m, gmm[3]));
        const double *Ux[1] = {S[cur]};
        double *Ox[1] = {Wb};
        const double *Bx[1] = {Wb};
        DDMCHECK(W.rotate(1, Ux, Ox, Bx, ld, m, gmm[3], m, ld, ld));
      }
      DDMCHECK(apply_AC(m, Wb, ld, AWb, CWb, ld)); // both products of W in one pass over it
      // A~-normalisation of the columns of W and P (zero columns stay zero): the blocks themselves are NOT rescaled (six passes over
      // n x m blocks per iteration in round 2) -- the diagonal scaling D is applied where it is cheap: to the p x p Gram matrices
      // (D G D) and to the rows of the Ritz coefficients (S D) Y = S (D Y), on the host; the norms W^T A~ W, P^T A~ P are the diagonal
      // of the unscaled S^T A~ S that is computed below anyway (two separate m x m products until round 3)
      have_residual = true;
    }
    const bool upper_only = W.gram2_sym(S[cur], ld, AS[cur], CS[cur], ld, p, gA, gC);
    HIPCHECK(ctx, hipGetLastError());
    HIPCHECK(ctx, hipMemcpyAsync(hA.data(), gA, sizeof(double) * hA.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(ctx, hipMemcpyAsync(hC.data(), gC, sizeof(double) * hC.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (upper_only)
      for (int64_t s = 0; s < nsub; ++s) {
        GeneoWork::gram2_mirror_host(p, hA.data() + (size_t)s * p * p);
        GeneoWork::gram2_mirror_host(p, hC.data() + (size_t)s * p * p);
      }
    if (have_residual) { // residuals of the block that entered this iteration (its Ritz values are still in hmu)
      worst = 0.0;
      int64_t nconv_cols = 0;
      int lead_min = nev; // converged columns in front of the first unconverged one, minimum over the subdomains
      for (int64_t s = 0; s < nsub; ++s) {
        int lead = 0;
        bool leading = true;
        for (int j = 0; j < nev; ++j) {
          const size_t dj = ((size_t)s * m + j) * m + j;
          const double rn = std::sqrt(std::max(h_rr[dj], 0.0)), mu = std::fabs(hmu[(size_t)s * m + j]);
          const double res = direct ? std::sqrt(std::fabs(h_rw[dj])) / std::max(mu, 1e-300) // sqrt(r^T T r) / mu
                                    : rn / std::max(mu * std::sqrt(std::max(h_aa[dj], 0.0)), 1e-300);
          worst = std::max(worst, res);
          if (res < P.tolerance) {
            ++nconv_cols;
            if (leading) ++lead;
          } else
            leading = false;
        }
        lead_min = std::min(lead_min, lead);
      }
      if (P.verbose)
        std::fprintf(stderr, "[ddm geneo] it %3d  worst residual %.3e  rank >= %d  lambda_min(sub 0) %.6g  converged columns %lld of %lld, leading (min over subdomains) %d\n", it, worst,
                     rank_min, 1.0 / hmu[0] - P.shift, (long long)nconv_cols, (long long)(nsub * nev), lead_min);
      if (worst < P.tolerance) {
        converged = 1;
        break;
      }
      if (it == P.maxit) break;
    }
    // Rayleigh-Ritz per subdomain on the host
    {
      const unsigned hw = host_threads();
      const int nth = (int)std::min<int64_t>(nsub, hw);
      std::vector<std::thread> th;
      std::vector<int> ranks((size_t)nsub, 0);
      for (int t = 0; t < nth; ++t)
        th.emplace_back([&, t]() {
          std::vector<double> Y1((size_t)p * m);
          for (int64_t s = t; s < nsub; s += nth) {
            double *d = dscale.data() + (size_t)s * p;
            for (int i = 0; i < p; ++i) d[i] = 1.0;
            if (have_residual)
              for (int j = 0; j < m; ++j) {
                const double gw = hA[(size_t)s * p * p + (size_t)(m + j) * p + (m + j)], gp = hA[(size_t)s * p * p + (size_t)(2 * m + j) * p + (2 * m + j)];
                d[m + j] = gw > 1e-300 ? 1.0 / std::sqrt(gw) : 0.0;
                d[2 * m + j] = gp > 1e-300 ? 1.0 / std::sqrt(gp) : 0.0;
              }
            double *gAs = hA.data() + (size_t)s * p * p, *gCs = hC.data() + (size_t)s * p * p;
            for (int i = 0; i < p; ++i)
              for (int j = 0; j < p; ++j) {
                gAs[(size_t)i * p + j] *= d[i] * d[j];
                gCs[(size_t)i * p + j] *= d[i] * d[j];
              }
            const int r = dense::rayleigh_ritz(p, gAs, gCs, m, tau, hmu.data() + (size_t)s * m, Y1.data());
            ranks[(size_t)s] = r;
            rcs[(size_t)s] = r < m ? 1 : 0;
            if (r < m) continue;
            double *Y = hY.data() + (size_t)s * p * q2; // [Y | Y with the X rows zeroed]: X_new = S Y, P_new = [W P] Y_{W,P}
            for (int i = 0; i < p; ++i)
              for (int j = 0; j < m; ++j) {
                const double y = d[i] * Y1[(size_t)i * m + j];
                Y[(size_t)i * q2 + j] = y;
                Y[(size_t)i * q2 + m + j] = i < m ? 0.0 : y;
              }
          }
        });
      for (auto &t : th) t.join();
      rank_min = p;
      for (int64_t s = 0; s < nsub; ++s) {
        if (rcs[(size_t)s]) return fail(ctx, DDM_ENUMERIC, "GenEO: Rayleigh-Ritz failed in subdomain %lld (rank %d of the block basis)", (long long)s, ranks[(size_t)s]);
        rank_min = std::min(rank_min, ranks[(size_t)s]);
      }
    }
    HIPCHECK(ctx, hipMemcpyAsync(Yd, hY.data(), sizeof(double) * hY.size(), hipMemcpyHostToDevice, ctx->stream));
    HIPCHECK(ctx, hipMemcpyAsync(mud, hmu.data(), sizeof(double) * hmu.size(), hipMemcpyHostToDevice, ctx->stream));
    // [X | . | P] of the other buffer <- S [Y | Y_wp]: two column blocks of one rotation (q = 2m, written with a gap of m columns)
    {
      const int nxt = cur ^ 1;
      const double *U3[3] = {S[cur], AS[cur], CS[cur]};
      double *O3[3] = {S[nxt], AS[nxt], CS[nxt]};
      // first m output columns -> X slot, the other m -> P slot (the W slot in between is overwritten by the next preconditioner solve)
      DDMCHECK(W.rotate(3, U3, O3, nullptr, ld, p, Yd, q2, ld, ld, /*gap_from=*/m, /*gap=*/m));
      cur = nxt;
    }
    // Refresh A~X, C X from X and A~P, C P from P: the products are carried along by the rotations and drift.  With the exact T every
    // second iteration: W and P shrink geometrically there (their scaling lives in the projected problem, the blocks are not
    // renormalised), the Ritz coefficients of such columns are large, and products of P that are never recomputed went wrong often
    // enough to matter -- on the elasticity pencil 12-18 block iterations in most runs but 36-120 or no convergence within 400 in
    // about one run of seven (the device factor's atomics make every run round differently); with P refreshed as well: 12-16 in 28
    // of 28 runs, refreshing X alone does not help (tools/geneo_variability.sh).  With ILU(0) every eighth iteration, as before.
    if (it > 0 && it % refresh_period == 0) {
      if (con) DDMCHECK(harmonic_apply(ctx, con, m, S[cur], ld, true));
      DDMCHECK(apply_AC(m, S[cur], ld, AS[cur], CS[cur], ld));
      DDMCHECK(apply_AC(m, S[cur] + 2 * m, ld, AS[cur] + 2 * m, CS[cur] + 2 * m, ld));
    }
  }
  const double t_loop = since(t_iter);
  // ---- output: eigenvalues, finalised basis ----
  for (int64_t s = 0; s < nsub; ++s)
    for (int j = 0; j < nev; ++j) eig_host[(size_t)s * nev + j] = 1.0 / hmu[(size_t)s * m + j] - P.shift;
  hipLaunchKernelGGL(k_geneo_copy_cols, dim3(gnm), dim3(256), 0, ctx->stream, n, m, (const double *)S[cur], ld, R, (int64_t)m);
  if (!P.raw) hipLaunchKernelGGL(k_geneo_rowscale, dim3(gnm), dim3(256), 0, ctx->stream, n, m, poud, R, (int64_t)m); // v <- D v
  DDMCHECK(W.gram(R, m, m, R, m, m, gmm[0]));
  hipLaunchKernelGGL(k_geneo_invsqrt_diag, dim3((unsigned)((nsub * m + 255) / 256)), dim3(256), 0, ctx->stream, (int)nsub, m, gmm[0], svec[0]); // 1 / ||D v||_2  (raw: 1 / ||v||_2)
  hipLaunchKernelGGL(k_geneo_rowscale, dim3(gnm), dim3(256), 0, ctx->stream, n, m, maskd, R, (int64_t)m);      // zero_at_dirichlet
  hipLaunchKernelGGL(k_geneo_finalize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, nev, W.sub_of_row, svec[0], m, R, (int64_t)m, basis_dev);
  HIPCHECK(ctx, hipGetLastError());
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (info) {
    info->iterations = it;
    info->converged = converged;
    info->used_direct = direct;
    info->worst_residual = worst;
    info->setup_s = t_setup;
    info->iterate_s = t_loop;
    info->nev = nev;
    info->direct_flops = direct ? own.T->direct_flops : 0.0;
  }
  return DDM_OK;
}

static int geneo_basis_impl(ddm_ctx *ctx, const char *who, const ddm_csr *A_neu, const ddm_csr *B_neu, int64_t nsub, const int64_t *sub_ptr, const double *pou_host,
                            const uint8_t *dirichlet_host, const ddm_geneo_params *params, int64_t kmax, double *basis_host, int32_t *nconv,
                            double *eigenvalues_host, ddm_geneo_info *info, ddm_harmonic *con, const double *pou_pencil_host,
                            const std::function<int(int, const double *, int64_t, double *, int64_t)> *op_C = nullptr)
{
  if (!ctx || !A_neu || !B_neu || !sub_ptr || !pou_host || !params || !basis_host || !nconv || !eigenvalues_host || nsub < 1)
    return fail(ctx, DDM_EINVAL, "%s: bad arguments", who);
  if (A_neu->nrows != A_neu->ncols || B_neu->nrows != A_neu->nrows || B_neu->ncols != A_neu->ncols)
    return fail(ctx, DDM_EINVAL, "The matrix and the partition of unity must have the same size"); // coarse_spaces.hh:323
  if (sub_ptr[0] != 0 || sub_ptr[nsub] != A_neu->nrows) return fail(ctx, DDM_EINVAL, "sub_ptr does not cover the matrix");
  const ddm_geneo_params &P = *params;
  if (P.nev < 1 || P.extra < 1 || !(P.tolerance > 0.0)) return fail(ctx, DDM_EINVAL, "%s: bad eigensolver parameters", who);
  const int64_t n = A_neu->nrows;
  int nev = P.nev;
  // threshold mode of spectra_gevp_op (eigensolvers/spectra.hh:157-163, 186-189): keep the eigenvalues below the threshold (at least
  // one), double nev until the largest computed one exceeds it or nev >= nev_max
  for (;;) {
    if (nev > kmax) return fail(ctx, DDM_EINVAL, "%s: kmax = %lld is smaller than nev = %d", who, (long long)kmax, nev);
    double *basis_dev = nullptr;
    HIPCHECK(ctx, hipMalloc((void **)&basis_dev, sizeof(double) * (size_t)nev * (size_t)std::max<int64_t>(n, 1)));
    std::vector<double> eig((size_t)nsub * nev);
    int rc = geneo_run(ctx, A_neu, B_neu, nsub, sub_ptr, pou_host, dirichlet_host, P, nev, basis_dev, eig.data(), info, con, pou_pencil_host, op_C);
    if (!rc && hipMemcpy(basis_host, basis_dev, sizeof(double) * (size_t)nev * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(ctx, DDM_EHIP, "%s: basis download failed", who);
    (void)hipFree(basis_dev);
    if (rc) return rc;
    bool done = true;
    if (P.threshold > 0.0) {
      for (int64_t s = 0; s < nsub; ++s) done = done && eig[(size_t)s * nev + nev - 1] >= P.threshold;
      // the block eigensolver takes nev + extra <= GENEO_MAX_BLOCK vectors per subdomain: the doubling stops there and the last
      // converged block is returned (info->nev tells the caller how far it got) instead of failing the whole basis build
      done = done || nev >= P.nev_max || 2 * nev > kmax || 2 * nev + P.extra > GENEO_MAX_BLOCK;
    }
    if (done) {
      for (int64_t s = 0; s < nsub; ++s) {
        int cnt = nev;
        if (P.threshold > 0.0) {
          cnt = 0;
          while (cnt < nev - 1 && eig[(size_t)s * nev + cnt] < P.threshold) ++cnt;
          cnt = std::max(cnt, 1);
        }
        nconv[s] = cnt;
        for (int j = 0; j < nev; ++j) eigenvalues_host[(size_t)s * kmax + j] = eig[(size_t)s * nev + j];
      }
      if (info) info->nev = nev;
      return DDM_OK;
    }
    nev *= 2;
  }
}

extern "C" int ddm_geneo_basis(ddm_ctx *ctx, const ddm_csr *A_neu, const ddm_csr *B_neu, int64_t nsub, const int64_t *sub_ptr, const double *pou_host,
                               const uint8_t *dirichlet_host, const ddm_geneo_params *params, int64_t kmax, double *basis_host, int32_t *nconv,
                               double *eigenvalues_host, ddm_geneo_info *info)
{
  return geneo_basis_impl(ctx, "ddm_geneo_basis", A_neu, B_neu, nsub, sub_ptr, pou_host, dirichlet_host, params, kmax, basis_host, nconv, eigenvalues_host, info, nullptr,
                          nullptr);
}

// MsGFEMCoarseSpace::setup_msgfem_impl (coarse_spaces.hh:712-826).  The reference assembles the saddle-point pencil
//     [A_nn  G^T; G  0] [u; p] = lambda [D A_ii D  0; 0  0] [u; p],      G = interior rows of A_dir (a-harmonicity),
// on the non-Dirichlet DoFs and hands it to the shift-invert Lanczos with an LU of the indefinite matrix.  Here the multipliers are
// eliminated instead: the same eigenpairs are the stationary points of the Rayleigh quotient of (A_neu, D A_ii D) on the
// a-harmonic subspace {u : G u = 0, u = 0 on Dirichlet DoFs}, and the block iteration of geneo_run stays inside that subspace
// with the projection P = [0 -A_ii^-1 A_ib; 0 I] (ddm_harmonic: one sparse Cholesky of the interior block, multi-RHS solves):
// start block P X0, search directions P T P^T r.  With the exact T = (A_neu + sigma C)^-1 the preconditioned operator is the
// inverse of the Schur complement onto the boundary unknowns, i.e. the iteration is the device analogue of the shift-invert.
extern "C" int ddm_msgfem_basis(ddm_ctx *ctx, const ddm_csr *A_neu, const ddm_csr *A_dir, int64_t nsub, const int64_t *sub_ptr, const double *pou_host,
                                const uint8_t *dirichlet_host, const uint8_t *boundary_host, const ddm_geneo_params *params, int64_t kmax, double *basis_host,
                                int32_t *nconv, double *eigenvalues_host, ddm_geneo_info *info)
{
  if (!ctx || !A_neu || !A_dir || !sub_ptr || !pou_host || !boundary_host || !params || nsub < 1) return fail(ctx, DDM_EINVAL, "ddm_msgfem_basis: bad arguments");
  if (A_dir->nrows != A_neu->nrows || A_dir->ncols != A_dir->nrows) return fail(ctx, DDM_EINVAL, "The two matrices must have the same size"); // :714
  const int64_t n = A_dir->nrows;
  std::vector<uint8_t> cls((size_t)n);
  std::vector<double> pou_int((size_t)n);
  for (int64_t i = 0; i < n; ++i) { // :722-740
    cls[(size_t)i] = (dirichlet_host && dirichlet_host[i]) ? 2 : boundary_host[i] ? 1 : 0;
    pou_int[(size_t)i] = cls[(size_t)i] == 0 ? pou_host[i] : 0.0; // the right-hand side has interior x interior entries only (:801-811)
  }
  ddm_harmonic *H = nullptr;
  DDMCHECK(harmonic_create_impl(ctx, A_dir, nsub, sub_ptr, cls.data(), true, &H));
  int rc = DDM_OK;
  if (!H->symmetric) rc = fail(ctx, DDM_ENOTIMPL, "ddm_msgfem_basis: the interior block of A_dir is not symmetric");
  if (!rc)
    rc = geneo_basis_impl(ctx, "ddm_msgfem_basis", A_neu, A_neu, nsub, sub_ptr, pou_host, dirichlet_host, params, kmax, basis_host, nconv, eigenvalues_host, info, H,
                          pou_int.data());
  ddm_harmonic_destroy(H);
  return rc;
}

// SVDCoarseSpace (coarse_spaces.hh:1268-1407): the leading left singular vectors of T = D A_ii^-1 A_{i,Gamma} (interior x subdomain
// boundary; the reference forms T densely column by column and calls Eigen's bdcSvd).  Here: the leading eigenvectors of
//     T T^T = D A_ii^-1 (A_{i,Gamma} A_{i,Gamma}^T) A_ii^-T D
// by the block eigensolver of geneo_run with the identity as left-hand matrix and T T^T applied as an operator -- two multi-RHS
// interior solves (the sparse direct factor of ddm_harmonic) around two sparse products per application; T is never formed.
extern "C" int ddm_svd_basis(ddm_ctx *ctx, const ddm_csr *A_dir, int64_t nsub, const int64_t *sub_ptr, const double *pou_host, const uint8_t *dirichlet_host,
                             const uint8_t *boundary_host, int n_vectors, int mult_pou, double tolerance, int maxit, double *basis_host, double *singular_values_host,
                             ddm_geneo_info *info)
{
  if (!ctx || !A_dir || !sub_ptr || !pou_host || !boundary_host || !basis_host || !singular_values_host || nsub < 1 || n_vectors < 1)
    return fail(ctx, DDM_EINVAL, "ddm_svd_basis: bad arguments");
  if (A_dir->nrows != A_dir->ncols || sub_ptr[0] != 0 || sub_ptr[nsub] != A_dir->nrows) return fail(ctx, DDM_EINVAL, "ddm_svd_basis: sub_ptr does not cover the matrix");
  const int64_t n = A_dir->nrows;
  std::vector<uint8_t> cls((size_t)n), notint((size_t)n);
  std::vector<double> pou_int((size_t)n), ones((size_t)n, 1.0);
  for (int64_t i = 0; i < n; ++i) { // :1293-1309
    cls[(size_t)i] = (dirichlet_host && dirichlet_host[i]) ? 2 : boundary_host[i] ? 1 : 0;
    notint[(size_t)i] = cls[(size_t)i] != 0;
    pou_int[(size_t)i] = cls[(size_t)i] == 0 ? pou_host[i] : 0.0;
  }
  ddm_harmonic *H = nullptr;
  DDMCHECK(harmonic_create_impl(ctx, A_dir, nsub, sub_ptr, cls.data(), true, &H));
  struct Guard {
    ddm_harmonic *H;
    ddm_csr *I = nullptr;
    double *d = nullptr;
    ~Guard()
    {
      ddm_harmonic_destroy(H);
      ddm_csr_destroy(I);
      (void)hipFree(d);
    }
  } g{H};
  if (!H->symmetric) return fail(ctx, DDM_ENOTIMPL, "ddm_svd_basis: the interior block of A_dir is not symmetric");
  std::vector<int64_t> rp((size_t)n + 1);
  std::vector<int32_t> ci((size_t)n);
  for (int64_t i = 0; i <= n; ++i) rp[(size_t)i] = i;
  for (int64_t i = 0; i < n; ++i) ci[(size_t)i] = (int32_t)i;
  DDMCHECK(ddm_csr_create(ctx, n, n, rp.data(), ci.data(), ones.data(), &g.I));
  DDMCHECK(upload(ctx, pou_int.data(), n, &g.d));
  const std::function<int(int, const double *, int64_t, double *, int64_t)> op = [&](int m, const double *X, int64_t ldx, double *Y, int64_t ldy) -> int {
    DDMCHECK(harmonic_reserve(ctx, H, m));
    const unsigned gr = (unsigned)((n * (int64_t)m + 255) / 256);
    hipLaunchKernelGGL(k_geneo_rowscale_to, dim3(gr), dim3(256), 0, ctx->stream, n, m, (const double *)g.d, X, ldx, H->t1, (int64_t)m); // D x
    DDMCHECK(ilu0_solve_multi_ld(ctx, H->F, m, H->t1, m, H->t2, m));                                                                 // A_ii^-T
    DDMCHECK(csr_mm_ld(ctx, H->Gbi, m, H->t2, m, H->t1, m));                                                                         // A_{i,Gamma}^T
    DDMCHECK(csr_mm_ld(ctx, H->Gib, m, H->t1, m, H->t2, m));                                                                         // A_{i,Gamma}
    DDMCHECK(ilu0_solve_multi_ld(ctx, H->F, m, H->t2, m, H->t1, m));                                                                 // A_ii^-1
    hipLaunchKernelGGL(k_geneo_rowscale_to, dim3(gr), dim3(256), 0, ctx->stream, n, m, (const double *)g.d, (const double *)H->t1, (int64_t)m, Y, ldy); // D
    HIPCHECK(ctx, hipGetLastError());
    return DDM_OK;
  };
  ddm_geneo_params P;
  ddm_geneo_params_default(&P);
  P.nev = n_vectors;
  P.extra = std::max(4, n_vectors / 2);
  P.shift = 0.0;            // A~ = I
  P.tolerance = tolerance > 0.0 ? tolerance : 1e-8;
  P.maxit = maxit > 0 ? maxit : 400;
  P.preconditioner = 1;     // ILU(0) of the identity: no preconditioner
  P.raw = mult_pou ? 0 : 1; // :1403: finalize_eigenvectors only with mult_pou
  std::vector<int32_t> nconv((size_t)nsub);
  std::vector<double> eig((size_t)nsub * n_vectors);
  DDMCHECK(geneo_basis_impl(ctx, "ddm_svd_basis", g.I, g.I, nsub, sub_ptr, pou_host, notint.data(), &P, n_vectors, basis_host, nconv.data(), eig.data(), info, nullptr,
                            ones.data(), &op));
  for (size_t k = 0; k < eig.size(); ++k) singular_values_host[k] = 1.0 / std::sqrt(std::max(eig[k], 1e-300)); // mu = sigma^2 = 1 / lambda
  return DDM_OK;
}

// ---- the two dense block kernels on their own (parity tests against an FP64 host reference; also usable by callers that keep
//      their block vectors on the device) ------------------------------------------------------------------------------------
static int blockvec_setup(ddm_ctx *ctx, GeneoWork &W, int64_t nsub, const int64_t *sub_ptr, int pmax_sq)
{
  W.ctx = ctx;
  W.nsub = (int)nsub;
  W.n = sub_ptr[nsub];
  std::vector<GChunk> chunks;
  std::vector<int32_t> scp((size_t)nsub + 1, 0);
  for (int64_t s = 0; s < nsub; ++s) {
    for (int64_t r = sub_ptr[s]; r < sub_ptr[s + 1]; r += GENEO_CHUNK_ROWS) chunks.push_back(GChunk{r, std::min(r + GENEO_CHUNK_ROWS, sub_ptr[s + 1]), (int32_t)s, 0});
    scp[(size_t)s + 1] = (int32_t)chunks.size();
  }
  W.nchunk = (int)chunks.size();
  DDMCHECK(W.alloc(&W.chunks, chunks.size()));
  DDMCHECK(W.alloc(&W.sub_chunk_ptr, scp.size()));
  DDMCHECK(W.alloc(&W.partial, (size_t)std::max(W.nchunk, 1) * (size_t)pmax_sq));
  HIPCHECK(ctx, hipMemcpy(W.chunks, chunks.data(), sizeof(GChunk) * chunks.size(), hipMemcpyHostToDevice));
  HIPCHECK(ctx, hipMemcpy(W.sub_chunk_ptr, scp.data(), sizeof(int32_t) * scp.size(), hipMemcpyHostToDevice));
  return DDM_OK;
}
extern "C" int ddm_blockvec_gram(ddm_ctx *ctx, int64_t nsub, const int64_t *sub_ptr, const double *U, int64_t ldu, int pu, const double *V, int64_t ldv,
                                 int pv, double *G_host)
{
  if (!ctx || !sub_ptr || !U || !V || !G_host || nsub < 1 || pu < 1 || pv < 1 || ldu < pu || ldv < pv) return fail(ctx, DDM_EINVAL, "ddm_blockvec_gram: bad arguments");
  GeneoWork W;
  DDMCHECK(blockvec_setup(ctx, W, nsub, sub_ptr, pu * pv));
  double *G = nullptr;
  DDMCHECK(W.alloc(&G, (size_t)nsub * pu * pv));
  DDMCHECK(W.gram(U, ldu, pu, V, ldv, pv, G));
  return ddm_memcpy_d2h(ctx, G_host, G, (int64_t)sizeof(double) * nsub * pu * pv);
}
extern "C" int ddm_blockvec_gram2_sym(ddm_ctx *ctx, int64_t nsub, const int64_t *sub_ptr, const double *U, int64_t ldu, const double *V1, const double *V2, int64_t ldv, int p,
                                      double *G1_host, double *G2_host)
{
  if (!ctx || !sub_ptr || !U || !V1 || !V2 || !G1_host || !G2_host || nsub < 1 || p < 1 || ldu < p || ldv < p) return fail(ctx, DDM_EINVAL, "ddm_blockvec_gram2_sym: bad arguments");
  GeneoWork W;
  DDMCHECK(blockvec_setup(ctx, W, nsub, sub_ptr, 2 * p * p));
  double *G = nullptr;
  DDMCHECK(W.alloc(&G, (size_t)nsub * p * p * 2));
  const bool upper_only = W.gram2_sym(U, ldu, V1, V2, ldv, p, G, G + (size_t)nsub * p * p);
  HIPCHECK(ctx, hipGetLastError());
  DDMCHECK(ddm_memcpy_d2h(ctx, G1_host, G, (int64_t)sizeof(double) * nsub * p * p));
  DDMCHECK(ddm_memcpy_d2h(ctx, G2_host, G + (size_t)nsub * p * p, (int64_t)sizeof(double) * nsub * p * p));
  if (upper_only)
    for (int64_t s = 0; s < nsub; ++s) {
      GeneoWork::gram2_mirror_host(p, G1_host + (size_t)s * p * p);
      GeneoWork::gram2_mirror_host(p, G2_host + (size_t)s * p * p);
    }
  return DDM_OK;
}
extern "C" int ddm_blockvec_rotate(ddm_ctx *ctx, int64_t nsub, const int64_t *sub_ptr, const double *U, int64_t ldu, int p, const double *Y_host, int q,
                                   const double *Base, int64_t ldb, double *Out, int64_t ldo)
{
  if (!ctx || !sub_ptr || !U || !Y_host || !Out || nsub < 1 || p < 1 || q < 1 || ldu < p || ldo < q || (Base && ldb < q) || U == Out)
    return fail(ctx, DDM_EINVAL, "ddm_blockvec_rotate: bad arguments");
  GeneoWork W;
  DDMCHECK(blockvec_setup(ctx, W, nsub, sub_ptr, 1));
  double *Y = nullptr;
  DDMCHECK(W.alloc(&Y, (size_t)nsub * p * q));
  DDMCHECK(ddm_memcpy_h2d(ctx, Y, Y_host, (int64_t)sizeof(double) * nsub * p * q));
  const double *Ux[1] = {U};
  double *Ox[1] = {Out};
  const double *Bx[1] = {Base};
  DDMCHECK(W.rotate(1, Ux, Ox, Base ? Bx : nullptr, ldu, p, Y, q, ldo, ldb));
  return ddm_ctx_sync(ctx);
}
