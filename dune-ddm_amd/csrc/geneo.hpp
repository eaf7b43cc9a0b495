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
  p->max_direct_flops = 3e11;
  p->verbose = 0;
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
  // G[sub] = U^T V per subdomain (pu x pv row-major, nsub matrices)
  int gram(const double *U, int64_t ldu, int pu, const double *V, int64_t ldv, int pv, double *G)
  {
    if (pu > 144 || pv > 144) return fail(ctx, DDM_ENOTIMPL, "GenEO: block wider than 144 columns");
    if (pu <= 128 && pv <= 80)
      hipLaunchKernelGGL((k_gram_mfma<2, 5>), dim3(nchunk), dim3(256), 0, ctx->stream, chunks, U, ldu, pu, V, ldv, pv, partial);
    else
      hipLaunchKernelGGL((k_gram_mfma<3, 9>), dim3(nchunk), dim3(256), 0, ctx->stream, chunks, U, ldu, pu, V, ldv, pv, partial);
    const int64_t pp = (int64_t)pu * pv;
    hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)((nsub * pp + 255) / 256)), dim3(256), 0, ctx->stream, nsub, sub_chunk_ptr, pp, partial, G);
    HIPCHECK(ctx, hipGetLastError());
    return DDM_OK;
  }
  // Out_k[:, 0:q) = (Base_k -) U_k[:, 0:pk) Y[sub]  for k < narr
  int rotate(int narr, const double *const *U, double *const *Out, const double *const *Base, int64_t ldu, int pk, const double *Y, int q, int64_t ldo, int64_t ldb)
  {
    if (q > 16 * ROT_TQ) return fail(ctx, DDM_ENOTIMPL, "GenEO: more than %d columns in a rotation", 16 * ROT_TQ);
    RotArgs a;
    for (int k = 0; k < 3; ++k) {
      a.U[k] = k < narr ? U[k] : nullptr;
      a.Out[k] = k < narr ? Out[k] : nullptr;
      a.Base[k] = (k < narr && Base) ? Base[k] : nullptr;
    }
    const int p4 = (pk + 3) & ~3, q16 = ((q + 15) >> 4) << 4;
    const size_t lds = sizeof(double) * ((size_t)p4 * q16 + 4 * 16 * (size_t)(p4 + 1));
    hipLaunchKernelGGL((k_rotate_mfma<0>), dim3(nchunk, narr), dim3(256), lds, ctx->stream, chunks, a, ldu, pk, Y, q, ldo, ldb);
    HIPCHECK(ctx, hipGetLastError());
    return DDM_OK;
  }
};

// A~ = A + sigma C~ and C~ = D B D without Dirichlet rows / columns, on the union pattern, as host CSR
static void build_pencil_host(const ddm_csr *A, const ddm_csr *B, const double *pou, const uint8_t *dir, double sigma, std::vector<int64_t> &rpT,
                              std::vector<int32_t> &ciT, std::vector<double> &vaT, std::vector<double> &vaC)
{
  const int64_t n = A->nrows;
  rpT.assign((size_t)n + 1, 0);
  // pass 1: sizes of the merged rows
  for (int64_t i = 0; i < n; ++i) {
    int64_t a = A->h_rp[i], b = B->h_rp[i], cnt = 0;
    const int64_t a1 = A->h_rp[i + 1], b1 = B->h_rp[i + 1];
    while (a < a1 || b < b1) {
      const int32_t ca = a < a1 ? A->h_ci[a] : INT32_MAX, cb = b < b1 ? B->h_ci[b] : INT32_MAX;
      a += ca <= cb;
      b += cb <= ca;
      ++cnt;
    }
    rpT[i + 1] = rpT[i] + cnt;
  }
  ciT.resize((size_t)rpT[n]);
  vaT.resize((size_t)rpT[n]);
  vaC.resize((size_t)rpT[n]);
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const int nth = (int)std::min<int64_t>(hw, std::max<int64_t>(1, n / 65536));
  std::vector<std::thread> th;
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

} // namespace

static int geneo_run(ddm_ctx *ctx, const ddm_csr *A_neu, const ddm_csr *B_neu, int64_t nsub, const int64_t *sub_ptr, const double *pou_host,
                     const uint8_t *dirichlet_host, const ddm_geneo_params &P, int nev, double *basis_dev /* nev x n */, double *eig_host /* nsub x nev */,
                     ddm_geneo_info *info)
{
  const int64_t n = A_neu->nrows;
  const int m = nev + std::max(P.extra, 1);
  const int p = 3 * m;
  if (p > 144) return fail(ctx, DDM_ENOTIMPL, "GenEO: nev + extra = %d exceeds 48 vectors per subdomain", m);
  for (int64_t s = 0; s < nsub; ++s)
    if (sub_ptr[s + 1] - sub_ptr[s] < 3 * (int64_t)m) return fail(ctx, DDM_EINVAL, "GenEO: subdomain %lld has fewer than 3 (nev + extra) = %d rows", (long long)s, 3 * m);
  const auto t_begin = std::chrono::steady_clock::now();
  auto since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  // ---- pencil ----
  std::vector<int64_t> rpT;
  std::vector<int32_t> ciT;
  std::vector<double> vaT, vaC;
  build_pencil_host(A_neu, B_neu, pou_host, dirichlet_host, P.shift, rpT, ciT, vaT, vaC);
  struct Owned {
    ddm_csr *At = nullptr, *C = nullptr;
    ddm_ilu0 *T = nullptr;
    ~Owned()
    {
      ddm_ilu0_destroy(T);
      ddm_csr_destroy(At);
      ddm_csr_destroy(C);
    }
  } own;
  DDMCHECK(ddm_csr_create(ctx, n, n, rpT.data(), ciT.data(), vaT.data(), &own.At));
  DDMCHECK(ddm_csr_create(ctx, n, n, rpT.data(), ciT.data(), vaC.data(), &own.C));
  { std::vector<double>().swap(vaT); std::vector<double>().swap(vaC); }
  // ---- preconditioner ----
  int direct = 0;
  if (P.preconditioner != 1) {
    const int rc = ddm_chol_create(ctx, own.At, nsub, sub_ptr, P.preconditioner == 2 ? 0.0 : P.max_direct_flops, &own.T);
    if (rc == DDM_OK) direct = 1;
    else if (P.preconditioner == 2 || (rc != DDM_ENOTIMPL && rc != DDM_ENUMERIC)) return rc;
    else if (P.verbose) std::fprintf(stderr, "[ddm geneo] sparse Cholesky not used (%s): ILU(0) preconditioner\n", ddm_last_error(ctx));
  }
  if (!direct) DDMCHECK(ilu0_create_impl(ctx, own.At, nsub, sub_ptr, /*multi_rhs_only=*/true, &own.T));
  const double t_setup = since(t_begin);
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
  DDMCHECK(W.alloc(&W.partial, (size_t)W.nchunk * p * p));
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
  double *gA, *gC, *gmm[4], *svec[2], *Yd, *mud;
  DDMCHECK(W.alloc(&gA, (size_t)nsub * p * p));
  DDMCHECK(W.alloc(&gC, (size_t)nsub * p * p));
  for (int k = 0; k < 4; ++k) DDMCHECK(W.alloc(&gmm[k], (size_t)nsub * m * m));
  for (int k = 0; k < 2; ++k) DDMCHECK(W.alloc(&svec[k], (size_t)nsub * m));
  const int q2 = 2 * m; // fused rotation: [X_new | P_new]
  DDMCHECK(W.alloc(&Yd, (size_t)nsub * p * q2));
  DDMCHECK(W.alloc(&mud, (size_t)nsub * m));
  std::vector<double> hA((size_t)nsub * p * p), hC((size_t)nsub * p * p), hY((size_t)nsub * p * q2), hmu((size_t)nsub * m);
  std::vector<double> h_rr((size_t)nsub * m * m), h_rw((size_t)nsub * m * m), h_aa((size_t)nsub * m * m);
  const unsigned gnm = (unsigned)((n * (int64_t)m + 255) / 256);
  const int64_t ld = p;
  // ---- initial block: random on the free DoFs, Rayleigh-Ritz on span X ----
  hipLaunchKernelGGL(k_geneo_random, dim3(gnm), dim3(256), 0, ctx->stream, n, m, ld, 0x5DEECE66Dull + (unsigned long long)P.seed, maskd, S[0]);
  DDMCHECK(csr_mm_ld(ctx, own.At, m, S[0], ld, AS[0], ld));
  DDMCHECK(csr_mm_ld(ctx, own.C, m, S[0], ld, CS[0], ld));
  int cur = 0, it = 0, converged = 0, rank_min = p;
  double worst = 0.0;
  const double tau = 1e-11;
  bool have_residual = false;
  std::vector<int> rcs((size_t)nsub, 0);
  const auto t_iter = std::chrono::steady_clock::now();
  for (it = 0; it <= P.maxit; ++it) {
    if (it > 0) {
      // R = C X - mu A~ X ; column norms ; W = T (R / ||R||)
      hipLaunchKernelGGL(k_geneo_residual, dim3(gnm), dim3(256), 0, ctx->stream, n, m, W.sub_of_row, mud, AS[cur], ld, CS[cur], ld, R, (int64_t)m);
      DDMCHECK(W.gram(R, m, m, R, m, m, gmm[0]));
      HIPCHECK(ctx, hipMemcpyAsync(h_rr.data(), gmm[0], sizeof(double) * h_rr.size(), hipMemcpyDeviceToHost, ctx->stream));
      hipLaunchKernelGGL(k_geneo_invsqrt_diag, dim3((unsigned)((nsub * m + 255) / 256)), dim3(256), 0, ctx->stream, (int)nsub, m, gmm[0], svec[0]);
      hipLaunchKernelGGL(k_geneo_colscale, dim3(gnm), dim3(256), 0, ctx->stream, n, m, W.sub_of_row, svec[0], R, (int64_t)m);
      double *Wb = S[cur] + m, *AWb = AS[cur] + m, *CWb = CS[cur] + m;
      DDMCHECK(ilu0_solve_multi_ld(ctx, own.T, m, R, m, Wb, ld));
      DDMCHECK(W.gram(R, m, m, Wb, ld, m, gmm[1]));   // r^T T r per column (diagonal)
      HIPCHECK(ctx, hipMemcpyAsync(h_rw.data(), gmm[1], sizeof(double) * h_rw.size(), hipMemcpyDeviceToHost, ctx->stream));
      DDMCHECK(W.gram(AS[cur], ld, m, AS[cur], ld, m, gmm[2]));
      HIPCHECK(ctx, hipMemcpyAsync(h_aa.data(), gmm[2], sizeof(double) * h_aa.size(), hipMemcpyDeviceToHost, ctx->stream));
      // W <- W - X (A~X)^T W   (twice), then A~-normalise the columns of W
      for (int pass = 0; pass < 2; ++pass) {
        DDMCHECK(W.gram(AS[cur], ld, m, Wb, ld, m, gmm[3]));
        const double *Ux[1] = {S[cur]};
        double *Ox[1] = {Wb};
        const double *Bx[1] = {Wb};
        DDMCHECK(W.rotate(1, Ux, Ox, Bx, ld, m, gmm[3], m, ld, ld));
      }
      DDMCHECK(csr_mm_ld(ctx, own.At, m, Wb, ld, AWb, ld));
      DDMCHECK(W.gram(Wb, ld, m, AWb, ld, m, gmm[3]));
      hipLaunchKernelGGL(k_geneo_invsqrt_diag, dim3((unsigned)((nsub * m + 255) / 256)), dim3(256), 0, ctx->stream, (int)nsub, m, gmm[3], svec[1]);
      hipLaunchKernelGGL(k_geneo_colscale, dim3(gnm), dim3(256), 0, ctx->stream, n, m, W.sub_of_row, svec[1], Wb, ld);
      hipLaunchKernelGGL(k_geneo_colscale, dim3(gnm), dim3(256), 0, ctx->stream, n, m, W.sub_of_row, svec[1], AWb, ld);
      DDMCHECK(csr_mm_ld(ctx, own.C, m, Wb, ld, CWb, ld));
      // A~-normalise the columns of P (zero columns stay zero)
      double *Pb = S[cur] + 2 * m, *APb = AS[cur] + 2 * m, *CPb = CS[cur] + 2 * m;
      DDMCHECK(W.gram(Pb, ld, m, APb, ld, m, gmm[3]));
      hipLaunchKernelGGL(k_geneo_invsqrt_diag, dim3((unsigned)((nsub * m + 255) / 256)), dim3(256), 0, ctx->stream, (int)nsub, m, gmm[3], svec[1]);
      hipLaunchKernelGGL(k_geneo_colscale, dim3(gnm), dim3(256), 0, ctx->stream, n, m, W.sub_of_row, svec[1], Pb, ld);
      hipLaunchKernelGGL(k_geneo_colscale, dim3(gnm), dim3(256), 0, ctx->stream, n, m, W.sub_of_row, svec[1], APb, ld);
      hipLaunchKernelGGL(k_geneo_colscale, dim3(gnm), dim3(256), 0, ctx->stream, n, m, W.sub_of_row, svec[1], CPb, ld);
      have_residual = true;
    }
    DDMCHECK(W.gram(S[cur], ld, p, AS[cur], ld, p, gA));
    DDMCHECK(W.gram(S[cur], ld, p, CS[cur], ld, p, gC));
    HIPCHECK(ctx, hipMemcpyAsync(hA.data(), gA, sizeof(double) * hA.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(ctx, hipMemcpyAsync(hC.data(), gC, sizeof(double) * hC.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (have_residual) { // residuals of the block that entered this iteration (its Ritz values are still in hmu)
      worst = 0.0;
      for (int64_t s = 0; s < nsub; ++s)
        for (int j = 0; j < nev; ++j) {
          const size_t dj = ((size_t)s * m + j) * m + j;
          const double rn = std::sqrt(std::max(h_rr[dj], 0.0)), mu = std::fabs(hmu[(size_t)s * m + j]);
          // T was applied to rhat = r / ||r||:  r^T T r = ||r||^2 rhat^T T rhat
          const double res = direct ? rn * std::sqrt(std::fabs(h_rw[dj])) / std::max(mu, 1e-300)
                                    : rn / std::max(mu * std::sqrt(std::max(h_aa[dj], 0.0)), 1e-300);
          worst = std::max(worst, res);
        }
      if (P.verbose) std::fprintf(stderr, "[ddm geneo] it %3d  worst residual %.3e  rank >= %d  lambda_min(sub 0) %.6g\n", it, worst, rank_min, 1.0 / hmu[0] - P.shift);
      if (worst < P.tolerance) {
        converged = 1;
        break;
      }
      if (it == P.maxit) break;
    }
    // Rayleigh-Ritz per subdomain on the host
    {
      const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
      const int nth = (int)std::min<int64_t>(nsub, hw);
      std::vector<std::thread> th;
      std::vector<int> ranks((size_t)nsub, 0);
      for (int t = 0; t < nth; ++t)
        th.emplace_back([&, t]() {
          std::vector<double> Y1((size_t)p * m);
          for (int64_t s = t; s < nsub; s += nth) {
            const int r = dense::rayleigh_ritz(p, hA.data() + (size_t)s * p * p, hC.data() + (size_t)s * p * p, m, tau, hmu.data() + (size_t)s * m, Y1.data());
            ranks[(size_t)s] = r;
            rcs[(size_t)s] = r < m ? 1 : 0;
            if (r < m) continue;
            double *Y = hY.data() + (size_t)s * p * q2; // [Y | Y with the X rows zeroed]: X_new = S Y, P_new = [W P] Y_{W,P}
            for (int i = 0; i < p; ++i)
              for (int j = 0; j < m; ++j) {
                Y[(size_t)i * q2 + j] = Y1[(size_t)i * m + j];
                Y[(size_t)i * q2 + m + j] = i < m ? 0.0 : Y1[(size_t)i * m + j];
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
      // first m output columns -> X slot
      DDMCHECK(W.rotate(3, U3, O3, nullptr, ld, p, Yd, q2, ld, ld));
      // the rotation wrote columns [0, 2m): move the second half into the P slot and clear the W slot
      // (done by writing directly: the kernel writes q2 contiguous columns, so P_new lands in the W slot; shift it)
      for (int k = 0; k < 3; ++k) {
        hipLaunchKernelGGL(k_geneo_copy_cols, dim3(gnm), dim3(256), 0, ctx->stream, n, m, (const double *)(O3[k] + m), ld, O3[k] + 2 * m, ld);
      }
      cur = nxt;
    }
    if (it > 0 && it % 8 == 0) { // refresh A~X, C X from X: the recursions drift
      DDMCHECK(csr_mm_ld(ctx, own.At, m, S[cur], ld, AS[cur], ld));
      DDMCHECK(csr_mm_ld(ctx, own.C, m, S[cur], ld, CS[cur], ld));
    }
  }
  const double t_loop = since(t_iter);
  // ---- output: eigenvalues, finalised basis ----
  for (int64_t s = 0; s < nsub; ++s)
    for (int j = 0; j < nev; ++j) eig_host[(size_t)s * nev + j] = 1.0 / hmu[(size_t)s * m + j] - P.shift;
  hipLaunchKernelGGL(k_geneo_copy_cols, dim3(gnm), dim3(256), 0, ctx->stream, n, m, (const double *)S[cur], ld, R, (int64_t)m);
  hipLaunchKernelGGL(k_geneo_rowscale, dim3(gnm), dim3(256), 0, ctx->stream, n, m, poud, R, (int64_t)m);       // v <- D v
  DDMCHECK(W.gram(R, m, m, R, m, m, gmm[0]));
  hipLaunchKernelGGL(k_geneo_invsqrt_diag, dim3((unsigned)((nsub * m + 255) / 256)), dim3(256), 0, ctx->stream, (int)nsub, m, gmm[0], svec[0]); // 1 / ||D v||_2
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

extern "C" int ddm_geneo_basis(ddm_ctx *ctx, const ddm_csr *A_neu, const ddm_csr *B_neu, int64_t nsub, const int64_t *sub_ptr, const double *pou_host,
                               const uint8_t *dirichlet_host, const ddm_geneo_params *params, int64_t kmax, double *basis_host, int32_t *nconv,
                               double *eigenvalues_host, ddm_geneo_info *info)
{
  if (!ctx || !A_neu || !B_neu || !sub_ptr || !pou_host || !params || !basis_host || !nconv || !eigenvalues_host || nsub < 1)
    return fail(ctx, DDM_EINVAL, "ddm_geneo_basis: bad arguments");
  if (A_neu->nrows != A_neu->ncols || B_neu->nrows != A_neu->nrows || B_neu->ncols != A_neu->ncols)
    return fail(ctx, DDM_EINVAL, "The matrix and the partition of unity must have the same size"); // coarse_spaces.hh:323
  if (sub_ptr[0] != 0 || sub_ptr[nsub] != A_neu->nrows) return fail(ctx, DDM_EINVAL, "sub_ptr does not cover the matrix");
  const ddm_geneo_params &P = *params;
  if (P.nev < 1 || P.extra < 1 || !(P.tolerance > 0.0)) return fail(ctx, DDM_EINVAL, "ddm_geneo_basis: bad eigensolver parameters");
  const int64_t n = A_neu->nrows;
  int nev = P.nev;
  // threshold mode of spectra_gevp_op (eigensolvers/spectra.hh:157-163, 186-189): keep the eigenvalues below the threshold (at least
  // one), double nev until the largest computed one exceeds it or nev >= nev_max
  for (;;) {
    if (nev > kmax) return fail(ctx, DDM_EINVAL, "ddm_geneo_basis: kmax = %lld is smaller than nev = %d", (long long)kmax, nev);
    double *basis_dev = nullptr;
    HIPCHECK(ctx, hipMalloc((void **)&basis_dev, sizeof(double) * (size_t)nev * (size_t)std::max<int64_t>(n, 1)));
    std::vector<double> eig((size_t)nsub * nev);
    int rc = geneo_run(ctx, A_neu, B_neu, nsub, sub_ptr, pou_host, dirichlet_host, P, nev, basis_dev, eig.data(), info);
    if (!rc && hipMemcpy(basis_host, basis_dev, sizeof(double) * (size_t)nev * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(ctx, DDM_EHIP, "GenEO: basis download failed");
    (void)hipFree(basis_dev);
    if (rc) return rc;
    bool done = true;
    if (P.threshold > 0.0) {
      for (int64_t s = 0; s < nsub; ++s) done = done && eig[(size_t)s * nev + nev - 1] >= P.threshold;
      done = done || nev >= P.nev_max || 2 * nev > kmax;
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
