// Small dense symmetric eigenproblems on the host: the p x p (p = 3 (nev + 4) <= ~150) projected problems of the block
// eigensolver's Rayleigh-Ritz step.  Householder tridiagonalisation followed by the implicit QL iteration (the classical
// EISPACK tred2 / tql2 pair, restated), O(p^3), a few hundred microseconds per subdomain and iteration.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

namespace dense {

// V (n x n, row-major) holds the symmetric matrix on entry and the eigenvectors (as columns) on return; w ascending.
// Returns false if the QL iteration does not converge.
inline bool sym_eig(int n, double *V, double *w)
{
  if (n == 0) return true;
  std::vector<double> e(n, 0.0);
  double *d = w;
  auto at = [&](int i, int j) -> double & { return V[(size_t)i * n + j]; };
  // ---- Householder reduction to tridiagonal form (accumulating the transformations in V) ----
  for (int j = 0; j < n; ++j) d[j] = at(n - 1, j);
  for (int i = n - 1; i > 0; --i) {
    double scale = 0.0, h = 0.0;
    for (int k = 0; k < i; ++k) scale += std::fabs(d[k]);
    if (scale == 0.0) {
      e[i] = d[i - 1];
      for (int j = 0; j < i; ++j) {
        d[j] = at(i - 1, j);
        at(i, j) = 0.0;
        at(j, i) = 0.0;
      }
    } else {
      for (int k = 0; k < i; ++k) {
        d[k] /= scale;
        h += d[k] * d[k];
      }
      double f = d[i - 1];
      double g = std::sqrt(h);
      if (f > 0) g = -g;
      e[i] = scale * g;
      h -= f * g;
      d[i - 1] = f - g;
      for (int j = 0; j < i; ++j) e[j] = 0.0;
      for (int j = 0; j < i; ++j) {
        f = d[j];
        at(j, i) = f;
        g = e[j] + at(j, j) * f;
        for (int k = j + 1; k <= i - 1; ++k) {
          g += at(k, j) * d[k];
          e[k] += at(k, j) * f;
        }
        e[j] = g;
      }
      f = 0.0;
      for (int j = 0; j < i; ++j) {
        e[j] /= h;
        f += e[j] * d[j];
      }
      const double hh = f / (h + h);
      for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
      for (int j = 0; j < i; ++j) {
        f = d[j];
        g = e[j];
        for (int k = j; k <= i - 1; ++k) at(k, j) -= (f * e[k] + g * d[k]);
        d[j] = at(i - 1, j);
        at(i, j) = 0.0;
      }
    }
    d[i] = h;
  }
  for (int i = 0; i < n - 1; ++i) {
    at(n - 1, i) = at(i, i);
    at(i, i) = 1.0;
    const double h = d[i + 1];
    if (h != 0.0) {
      for (int k = 0; k <= i; ++k) d[k] = at(k, i + 1) / h;
      for (int j = 0; j <= i; ++j) {
        double g = 0.0;
        for (int k = 0; k <= i; ++k) g += at(k, i + 1) * at(k, j);
        for (int k = 0; k <= i; ++k) at(k, j) -= g * d[k];
      }
    }
    for (int k = 0; k <= i; ++k) at(k, i + 1) = 0.0;
  }
  for (int j = 0; j < n; ++j) {
    d[j] = at(n - 1, j);
    at(n - 1, j) = 0.0;
  }
  at(n - 1, n - 1) = 1.0;
  e[0] = 0.0;
  // ---- implicit QL on the tridiagonal matrix ----
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  double f = 0.0, tst1 = 0.0;
  const double eps = std::pow(2.0, -52.0);
  for (int l = 0; l < n; ++l) {
    tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
    int m = l;
    while (m < n) {
      if (std::fabs(e[m]) <= eps * tst1) break;
      ++m;
    }
    if (m >= n) m = n - 1;
    if (m > l) {
      int iter = 0;
      do {
        if (++iter > 300) return false;
        double g = d[l];
        double p = (d[l + 1] - g) / (2.0 * e[l]);
        double r = std::hypot(p, 1.0);
        if (p < 0) r = -r;
        d[l] = e[l] / (p + r);
        d[l + 1] = e[l] * (p + r);
        const double dl1 = d[l + 1];
        double h = g - d[l];
        for (int i = l + 2; i < n; ++i) d[i] -= h;
        f += h;
        p = d[m];
        double c = 1.0, c2 = c, c3 = c;
        const double el1 = e[l + 1];
        double s = 0.0, s2 = 0.0;
        for (int i = m - 1; i >= l; --i) {
          c3 = c2;
          c2 = c;
          s2 = s;
          g = c * e[i];
          h = c * p;
          r = std::hypot(p, e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          c = p / r;
          p = c * d[i] - s * g;
          d[i + 1] = h + s * (c * g + s * d[i]);
          for (int k = 0; k < n; ++k) {
            h = at(k, i + 1);
            at(k, i + 1) = s * at(k, i) + c * h;
            at(k, i) = c * at(k, i) - s * h;
          }
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = c * p;
      } while (std::fabs(e[l]) > eps * tst1);
    }
    d[l] = d[l] + f;
    e[l] = 0.0;
  }
  // ---- sort ascending ----
  for (int i = 0; i < n - 1; ++i) {
    int k = i;
    double p = d[i];
    for (int j = i + 1; j < n; ++j)
      if (d[j] < p) {
        k = j;
        p = d[j];
      }
    if (k != i) {
      d[k] = d[i];
      d[i] = p;
      for (int j = 0; j < n; ++j) std::swap(at(j, i), at(j, k));
    }
  }
  return true;
}

// Rayleigh-Ritz of the pencil  gC y = mu gA y  on a possibly rank-deficient basis (gA symmetric positive SEMI-definite, p x p,
// row-major): the basis is truncated to the directions whose eigenvalue of the diagonally scaled gA exceeds tau * largest
// (the "SVQB" orthonormalisation of Stathopoulos & Wu inside the Rayleigh-Ritz step, as in the robust LOBPCG of Duersch et al.),
// then the standard symmetric problem of the projected gC is solved.  Output: the `keep` largest mu (descending) and the
// coefficient matrix Y (p x keep, row-major) with Y^T gA Y = I.  Returns the rank used, or -1 on failure.
inline int rayleigh_ritz(int p, const double *gA, const double *gC, int keep, double tau, double *mu, double *Y)
{
  std::vector<double> G((size_t)p * p), d(p), lam(p);
  for (int i = 0; i < p; ++i) {
    const double g = gA[(size_t)i * p + i];
    d[i] = g > 1e-300 ? 1.0 / std::sqrt(g) : 0.0;
  }
  for (int i = 0; i < p; ++i)
    for (int j = 0; j < p; ++j) G[(size_t)i * p + j] = 0.5 * (gA[(size_t)i * p + j] + gA[(size_t)j * p + i]) * d[i] * d[j];
  if (!sym_eig(p, G.data(), lam.data())) return -1;
  const double lmax = lam[p - 1];
  if (!(lmax > 0.0)) return -1;
  int first = 0;
  while (first < p && !(lam[first] > tau * lmax)) ++first;
  const int r = p - first;
  if (r < 1) return -1;
  // B = diag(d) Q[:, first:] diag(lam^-1/2)   (p x r)
  std::vector<double> B((size_t)p * r);
  for (int i = 0; i < p; ++i)
    for (int k = 0; k < r; ++k) B[(size_t)i * r + k] = d[i] * G[(size_t)i * p + first + k] / std::sqrt(lam[first + k]);
  // M = B^T gC B  (r x r)
  std::vector<double> T((size_t)p * r, 0.0), M((size_t)r * r, 0.0), w(r);
  for (int i = 0; i < p; ++i)
    for (int j = 0; j < p; ++j) {
      const double c = 0.5 * (gC[(size_t)i * p + j] + gC[(size_t)j * p + i]);
      if (c == 0.0) continue;
      for (int k = 0; k < r; ++k) T[(size_t)i * r + k] += c * B[(size_t)j * r + k];
    }
  for (int i = 0; i < p; ++i)
    for (int a = 0; a < r; ++a) {
      const double b = B[(size_t)i * r + a];
      if (b == 0.0) continue;
      for (int k = 0; k < r; ++k) M[(size_t)a * r + k] += b * T[(size_t)i * r + k];
    }
  for (int a = 0; a < r; ++a)
    for (int k = a + 1; k < r; ++k) M[(size_t)a * r + k] = M[(size_t)k * r + a] = 0.5 * (M[(size_t)a * r + k] + M[(size_t)k * r + a]);
  if (!sym_eig(r, M.data(), w.data())) return -1;
  const int kk = std::min(keep, r);
  for (int c = 0; c < keep; ++c) {
    mu[c] = c < kk ? w[r - 1 - c] : 0.0;
    for (int i = 0; i < p; ++i) {
      double s = 0.0;
      if (c < kk)
        for (int k = 0; k < r; ++k) s += B[(size_t)i * r + k] * M[(size_t)k * r + (r - 1 - c)];
      Y[(size_t)i * keep + c] = s;
    }
  }
  return r;
}

} // namespace dense
