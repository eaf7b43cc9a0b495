/* ORACLE (test infrastructure only -- never linked into or called by the product path).
 *
 * Plain-C restatement of the inner loops of dune-ddm's two-level Schwarz apply path.
 * The loops that live in dune-istl (not in the /root/reference snapshot: empty submodule
 * extern/dune-istl, DUNE 2.10 series) are restated from the published algorithms and anchored
 * on the reference's call sites, cited per function (paths relative to /root/reference).
 * Built by oracle/Makefile into oracle/_build/liboracle.so; loaded with ctypes by
 * oracle/apply_oracle.py.  Parity status: see the header of apply_oracle.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;
typedef int32_t i32;

/* BCRSMatrix::mv, y = A x -- call site dune/ddm/nonoverlapping_operator.hh:37,
 * galerkin_preconditioner.hh:293; same loop as MatOp::perform_op, eigensolvers/spectra.hh:100-105 */
void orc_csr_mv(i64 n, const i64 *rp, const i32 *ci, const double *v, const double *x, double *y)
{
  for (i64 i = 0; i < n; ++i) {
    double s = 0.0;
    for (i64 k = rp[i]; k < rp[i + 1]; ++k) s += v[k] * x[ci[k]];
    y[i] = s;
  }
}

/* BCRSMatrix::usmv, y += alpha A x -- call site dune/ddm/nonoverlapping_operator.hh:47 */
void orc_csr_usmv(i64 n, const i64 *rp, const i32 *ci, const double *v, double alpha, const double *x, double *y)
{
  for (i64 i = 0; i < n; ++i) {
    double s = 0.0;
    for (i64 k = rp[i]; k < rp[i + 1]; ++k) s += v[k] * x[ci[k]];
    y[i] += alpha * s;
  }
}

/* ILU(0) in the matrix pattern, natural row order (dune-istl ilu.hh `blockILU0Decomposition`,
 * reached through the solver factory at dune/ddm/schwarz.hh:85-92): IKJ elimination, the
 * multipliers are stored in L, the diagonal of U is stored INVERTED.  Rows must have sorted
 * column indices and a stored diagonal.  lu is a copy of v on entry.  diag[i] = position of
 * the diagonal of row i.  Returns 0, or i+1 if row i has no diagonal / a zero pivot. */
int orc_ilu0_factor(i64 n, const i64 *rp, const i32 *ci, double *lu, i64 *diag)
{
  for (i64 i = 0; i < n; ++i) {
    diag[i] = -1;
    for (i64 k = rp[i]; k < rp[i + 1]; ++k)
      if (ci[k] == i) { diag[i] = k; break; }
    if (diag[i] < 0) return (int)(i + 1);
  }
  for (i64 i = 0; i < n; ++i) {
    for (i64 kk = rp[i]; kk < diag[i]; ++kk) {     /* columns k < i of row i */
      const i64 k = ci[kk];
      lu[kk] *= lu[diag[k]];                          /* a_ik <- a_ik * (a_kk)^-1 (inverse stored) */
      const double lik = lu[kk];
      /* row_i -= lik * row_k for the columns j > k present in both rows (merge of two sorted rows) */
      i64 pi = kk + 1;
      for (i64 pk = diag[k] + 1; pk < rp[k + 1]; ++pk) {
        const i32 j = ci[pk];
        while (pi < rp[i + 1] && ci[pi] < j) ++pi;
        if (pi == rp[i + 1]) break;
        if (ci[pi] == j) lu[pi] -= lik * lu[pk];
      }
    }
    if (lu[diag[i]] == 0.0) return (int)(i + 1);
    lu[diag[i]] = 1.0 / lu[diag[i]];
  }
  return 0;
}

/* ILU back-solve (dune-istl ilu.hh `blockILUBacksolve`): v = L^-1 d (unit lower), then
 * v_i = a_ii^-1 (v_i - sum_{j>i} a_ij v_j) with the stored inverse diagonal.
 * This is the local `solver->apply(x_ovlp, d_ovlp, res)` of dune/ddm/schwarz.hh:133 for
 * [subdomain_solver] type=loopsolver maxit=1 + preconditioner type=ilu n=0 (SURVEY.md caveats). */
void orc_ilu0_solve(i64 n, const i64 *rp, const i32 *ci, const double *lu, const i64 *diag, const double *d, double *x)
{
  for (i64 i = 0; i < n; ++i) {
    double s = d[i];
    for (i64 k = rp[i]; k < diag[i]; ++k) s -= lu[k] * x[ci[k]];
    x[i] = s;
  }
  for (i64 i = n - 1; i >= 0; --i) {
    double s = x[i];
    for (i64 k = diag[i] + 1; k < rp[i + 1]; ++k) s -= lu[k] * x[ci[k]];
    x[i] = s * lu[diag[i]];
  }
}

/* OwnerOverlapCopyCommunication::dot -- owner-masked local part; the caller adds the ranks'
 * partial sums (MPI_Allreduce).  Call site dune/ddm/nonoverlapping_operator.hh:76-81. */
double orc_masked_dot(i64 n, const uint8_t *owner, const double *x, const double *y)
{
  double s = 0.0;
  for (i64 i = 0; i < n; ++i)
    if (owner[i]) s += x[i] * y[i];
  return s;
}

/* The same owner-masked sum in two OTHER summation orders -- not reference semantics: they exist so that the tests can show how far
 * two correct FP64 implementations of the same CG drift apart when only the order of the additions inside the global dot products
 * differs (tests/test_oracle_order_sensitivity.py; the HIP path sums wavefront-wise, the reference index by index).
 *   order 1: descending index;  order 2: pairwise (recursive halving, blocks of 64 summed ascending) */
static double masked_pairwise(const uint8_t *owner, const double *x, const double *y, i64 lo, i64 hi)
{
  if (hi - lo <= 64) {
    double s = 0.0;
    for (i64 i = lo; i < hi; ++i)
      if (owner[i]) s += x[i] * y[i];
    return s;
  }
  const i64 mid = lo + (hi - lo) / 2;
  return masked_pairwise(owner, x, y, lo, mid) + masked_pairwise(owner, x, y, mid, hi);
}
double orc_masked_dot_order(i64 n, const uint8_t *owner, const double *x, const double *y, int order)
{
  if (order == 1) {
    double s = 0.0;
    for (i64 i = n - 1; i >= 0; --i)
      if (owner[i]) s += x[i] * y[i];
    return s;
  }
  if (order == 2) return masked_pairwise(owner, x, y, 0, n);
  return orc_masked_dot(n, owner, x, y);
}

/* plain dot: restr_vecs[k] * y (dune/ddm/galerkin_preconditioner.hh:165-167, 294) */
double orc_dot(i64 n, const double *x, const double *y)
{
  double s = 0.0;
  for (i64 i = 0; i < n; ++i) s += x[i] * y[i];
  return s;
}

/* y += a x  (BlockVector::axpy; CG body, combined_preconditioner.hh:141) */
void orc_axpy(i64 n, double a, const double *x, double *y)
{
  for (i64 i = 0; i < n; ++i) y[i] += a * x[i];
}

/* p = beta p + q (CGSolver) */
void orc_xpby(i64 n, const double *q, double beta, double *p)
{
  for (i64 i = 0; i < n; ++i) p[i] = beta * p[i] + q[i];
}

/* x_ovlp[i] *= pou[i] (dune/ddm/schwarz.hh:141) */
void orc_scale(i64 n, const double *w, double *x)
{
  for (i64 i = 0; i < n; ++i) x[i] *= w[i];
}

/* dense LU with partial pivoting + solve; stands in for the factory coarse solver
 * (UMFPack/Cholmod on the K x K coarse matrix, galerkin_preconditioner.hh:338-346, 174-179). */
int orc_dense_lu(i64 n, double *a, i64 *piv)
{
  for (i64 k = 0; k < n; ++k) {
    i64 p = k;
    double m = fabs(a[k * n + k]);
    for (i64 i = k + 1; i < n; ++i)
      if (fabs(a[i * n + k]) > m) { m = fabs(a[i * n + k]); p = i; }
    piv[k] = p;
    if (m == 0.0) return (int)(k + 1);
    if (p != k)
      for (i64 j = 0; j < n; ++j) { double t = a[k * n + j]; a[k * n + j] = a[p * n + j]; a[p * n + j] = t; }
    for (i64 i = k + 1; i < n; ++i) {
      a[i * n + k] /= a[k * n + k];
      const double l = a[i * n + k];
      for (i64 j = k + 1; j < n; ++j) a[i * n + j] -= l * a[k * n + j];
    }
  }
  return 0;
}

void orc_dense_lu_solve(i64 n, const double *a, const i64 *piv, double *b)
{
  /* orc_dense_lu swaps FULL rows (the multipliers already stored move with them, as LAPACK's getrf does), so L refers to
   * the finally permuted rows: all interchanges are applied to b before the forward substitution (getrs: laswp, then trsm) */
  for (i64 k = 0; k < n; ++k)
    if (piv[k] != k) { double t = b[k]; b[k] = b[piv[k]]; b[piv[k]] = t; }
  for (i64 k = 0; k < n; ++k)
    for (i64 i = k + 1; i < n; ++i) b[i] -= a[i * n + k] * b[k];
  for (i64 i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (i64 j = i + 1; j < n; ++j) s -= a[i * n + j] * b[j];
    b[i] = s / a[i * n + i];
  }
}
