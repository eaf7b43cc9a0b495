// Host side of the sparse direct local solver: fill-reducing ordering, symbolic analysis and numeric Cholesky of a
// symmetric positive definite CSR block, producing the factor in the SAME storage convention the ILU(0) engines use
// (unit lower factor below the diagonal, INVERSE pivots on it, upper factor above), so that the device triangular
// solves (levels / xcd2 / multi-RHS kernels) run unchanged on it.
//
// What it stands in for: the reference's subdomain / coarse / eigenproblem solvers `type = cholmod` and `type = umfpack`
// (SuiteSparse, not in the snapshot) behind Dune::InverseOperator -- dune/ddm/schwarz.hh:85-92, every shipped .ini
// (examples/poisson.ini:23,26), and SymShiftInvert of the GenEO eigensolver (dune/ddm/eigensolvers/spectra.hh:28-89).
// Factorisation is setup work and runs on host threads (one per subdomain), like the ILU(0) factorisation; every solve
// with the factors runs on the device.
//
// Algorithms (textbook; own code):
//   ordering  : nested dissection by breadth-first level structures (George & Liu's automatic nested dissection):
//               pseudo-peripheral start vertex, the narrowest level in the middle third as vertex separator, separator
//               vertices without a neighbour on the far side are handed back, recursion on the parts, separator last;
//   symbolic  : elimination tree with path compression, row patterns by tree reach (also gives nnz(L) and the flop count,
//               which decides whether a block is affordable before any arithmetic is done);
//   numeric   : up-looking L L^T (row k of L from the rows it reaches in the elimination tree).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

namespace chol {

struct Graph { // symmetric pattern of one block, local indices, no diagonal
  int32_t n = 0;
  std::vector<int64_t> ptr;
  std::vector<int32_t> adj;
};

// pattern of A + A^T restricted to rows/cols [r0, r1)
inline Graph block_graph(const int64_t *rp, const int32_t *ci, int64_t r0, int64_t r1)
{
  Graph G;
  const int32_t n = (int32_t)(r1 - r0);
  G.n = n;
  std::vector<int64_t> cnt(n + 1, 0);
  for (int64_t i = r0; i < r1; ++i)
    for (int64_t k = rp[i]; k < rp[i + 1]; ++k) {
      const int64_t j = ci[k];
      if (j == i || j < r0 || j >= r1) continue;
      cnt[i - r0 + 1]++;
      cnt[j - r0 + 1]++;
    }
  for (int32_t i = 0; i < n; ++i) cnt[i + 1] += cnt[i];
  std::vector<int32_t> adj((size_t)cnt[n]);
  std::vector<int64_t> pos(cnt.begin(), cnt.end() - 1);
  for (int64_t i = r0; i < r1; ++i)
    for (int64_t k = rp[i]; k < rp[i + 1]; ++k) {
      const int64_t j = ci[k];
      if (j == i || j < r0 || j >= r1) continue;
      adj[(size_t)pos[i - r0]++] = (int32_t)(j - r0);
      adj[(size_t)pos[j - r0]++] = (int32_t)(i - r0);
    }
  // sort + unique per vertex
  G.ptr.assign(n + 1, 0);
  G.adj.reserve(adj.size() / 2 + 1);
  for (int32_t i = 0; i < n; ++i) {
    auto b = adj.begin() + cnt[i], e = adj.begin() + cnt[i + 1];
    std::sort(b, e);
    e = std::unique(b, e);
    G.adj.insert(G.adj.end(), b, e);
    G.ptr[i + 1] = (int64_t)G.adj.size();
  }
  return G;
}

// Nested-dissection elimination order: perm[new] = old (local indices).
// group_ends (optional): perm.size() after every emitted group (a leaf region or a separator) -- the candidate supernodes of the
// supernodal device factorisation (sn_chol_host.hpp): the vertices of one group are eliminated consecutively.
// probe_first_separator (optional): return right after the FIRST split with the size of its separator in it (perm stays empty) -- a
// cheap look at how expensive the factorisation is going to be (sn::estimate_flops) before the whole ordering is computed.
inline std::vector<int32_t> nested_dissection(const Graph &G, int leaf = 48, std::vector<int32_t> *group_ends = nullptr, int64_t *probe_first_separator = nullptr)
{
  const int32_t n = G.n;
  std::vector<int32_t> perm;
  perm.reserve(n);
  std::vector<int32_t> region(n, 0); // current region id of every vertex (-1 = already ordered)
  std::vector<int32_t> level(n, -1), queue;
  queue.reserve(n);
  struct Job {
    std::vector<int32_t> verts;
    bool emit; // true: append as is (separator / leaf)
  };
  std::vector<Job> stack;
  {
    Job j;
    j.verts.resize(n);
    for (int32_t i = 0; i < n; ++i) j.verts[i] = i;
    j.emit = false;
    stack.push_back(std::move(j));
  }
  int32_t next_region = 1;
  // BFS inside the region `rid` from `s`; fills level[] for reached vertices, returns them in BFS order
  auto bfs = [&](int32_t s, int32_t rid, std::vector<int32_t> &out) {
    out.clear();
    out.push_back(s);
    level[s] = 0;
    for (size_t h = 0; h < out.size(); ++h) {
      const int32_t v = out[h];
      for (int64_t k = G.ptr[v]; k < G.ptr[v + 1]; ++k) {
        const int32_t u = G.adj[(size_t)k];
        if (region[u] == rid && level[u] < 0) {
          level[u] = level[v] + 1;
          out.push_back(u);
        }
      }
    }
  };
  std::vector<int32_t> reach, reach2;
  while (!stack.empty()) {
    Job job = std::move(stack.back());
    stack.pop_back();
    std::vector<int32_t> &V = job.verts;
    if (V.empty()) continue;
    if (job.emit || (int)V.size() <= leaf) {
      for (int32_t v : V) {
        perm.push_back(v);
        region[v] = -1;
      }
      if (group_ends) group_ends->push_back((int32_t)perm.size());
      continue;
    }
    const int32_t rid = next_region++;
    for (int32_t v : V) region[v] = rid;
    // pseudo-peripheral vertex of the component of V[0]
    int32_t s = V[0];
    for (int pass = 0; pass < 3; ++pass) {
      bfs(s, rid, reach);
      const int32_t last = reach.back();
      const int32_t depth = level[last];
      for (int32_t v : reach) level[v] = -1;
      if (pass > 0 && last == s) break;
      (void)depth;
      s = last;
    }
    bfs(s, rid, reach);
    const int32_t nlev = level[reach.back()] + 1;
    if (reach.size() < V.size()) {
      // disconnected region: the reached component and the rest are independent jobs
      Job a, b;
      a.emit = b.emit = false;
      for (int32_t v : V) (level[v] >= 0 ? a.verts : b.verts).push_back(v);
      for (int32_t v : reach) level[v] = -1;
      stack.push_back(std::move(b));
      stack.push_back(std::move(a));
      continue;
    }
    if (nlev < 3) { // small diameter (clique-like): no useful separator
      for (int32_t v : reach) level[v] = -1;
      for (int32_t v : V) {
        perm.push_back(v);
        region[v] = -1;
      }
      if (group_ends) group_ends->push_back((int32_t)perm.size());
      continue;
    }
    // level sizes; separator = the narrowest level whose lower side holds 30-70 % of the vertices (else the median level)
    std::vector<int64_t> lsize(nlev, 0);
    for (int32_t v : reach) lsize[level[v]]++;
    int64_t cum = 0;
    int32_t best = -1, median = 1;
    const int64_t tot = (int64_t)reach.size();
    for (int32_t l = 0; l < nlev; ++l) {
      if (l >= 1 && l <= nlev - 2) {
        if (10 * cum >= 3 * tot && 10 * cum <= 7 * tot && (best < 0 || lsize[l] < lsize[best])) best = l;
        if (2 * cum <= tot) median = l;
      }
      cum += lsize[l];
    }
    const int32_t sep = best >= 0 ? best : std::max(1, std::min(median, nlev - 2));
    Job A, B, S;
    A.emit = B.emit = false;
    S.emit = true;
    for (int32_t v : reach) {
      if (level[v] < sep) A.verts.push_back(v);
      else if (level[v] > sep) B.verts.push_back(v);
      else {
        bool far = false; // a separator vertex without a neighbour beyond the separator belongs to the near side
        for (int64_t k = G.ptr[v]; k < G.ptr[v + 1] && !far; ++k) {
          const int32_t u = G.adj[(size_t)k];
          far = region[u] == rid && level[u] > sep;
        }
        (far ? S.verts : A.verts).push_back(v);
      }
    }
    for (int32_t v : reach) level[v] = -1;
    if (probe_first_separator) {
      *probe_first_separator = (int64_t)S.verts.size();
      return std::vector<int32_t>();
    }
    stack.push_back(std::move(S)); // popped last => ordered last
    stack.push_back(std::move(B));
    stack.push_back(std::move(A));
  }
  // the stack is LIFO: the order in which jobs finish is A-subtree, B-subtree, S -- exactly elimination order
  return perm;
}

struct BlockFactor {
  int32_t n = 0;
  std::vector<int32_t> perm;   // perm[new] = old (local)
  std::vector<int32_t> parent; // elimination tree
  std::vector<int64_t> Lp;     // CSC column pointers of L (diagonal first in every column)
  std::vector<int32_t> Li;
  std::vector<double> Lx;
  int64_t nnzL = 0;
  double flops = 0.0;
  std::string error;
};

// lower part (new numbering) of the permuted block: rows k, entries (j < k, value) and the diagonal
struct PermutedLower {
  std::vector<int64_t> ptr;
  std::vector<int32_t> col;
  std::vector<double> val;
  std::vector<double> diag;
};
inline PermutedLower permute_lower(const int64_t *rp, const int32_t *ci, const double *va, int64_t r0, int64_t r1, const std::vector<int32_t> &perm)
{
  const int32_t n = (int32_t)(r1 - r0);
  std::vector<int32_t> iperm(n);
  for (int32_t k = 0; k < n; ++k) iperm[perm[k]] = k;
  PermutedLower P;
  P.ptr.assign(n + 1, 0);
  P.diag.assign(n, 0.0);
  for (int32_t k = 0; k < n; ++k) {
    const int64_t i = r0 + perm[k];
    int64_t c = 0;
    for (int64_t p = rp[i]; p < rp[i + 1]; ++p) {
      const int64_t j = ci[p];
      if (j < r0 || j >= r1) continue;
      if (iperm[j - r0] < k) ++c;
    }
    P.ptr[k + 1] = P.ptr[k] + c;
  }
  P.col.resize((size_t)P.ptr[n]);
  P.val.resize((size_t)P.ptr[n]);
  for (int32_t k = 0; k < n; ++k) {
    const int64_t i = r0 + perm[k];
    int64_t q = P.ptr[k];
    for (int64_t p = rp[i]; p < rp[i + 1]; ++p) {
      const int64_t j = ci[p];
      if (j < r0 || j >= r1) continue;
      const int32_t jn = iperm[j - r0];
      if (jn < k) {
        P.col[(size_t)q] = jn;
        P.val[(size_t)q] = va ? va[p] : 0.0;
        ++q;
      } else if (jn == k && va) P.diag[k] = va[p];
    }
  }
  return P;
}

// elimination tree + column counts + flops (no arithmetic)
inline void analyze(const PermutedLower &P, int32_t n, BlockFactor &F)
{
  F.n = n;
  F.parent.assign(n, -1);
  std::vector<int32_t> anc(n, -1);
  for (int32_t k = 0; k < n; ++k)
    for (int64_t p = P.ptr[k]; p < P.ptr[k + 1]; ++p) {
      int32_t i = P.col[(size_t)p];
      while (i != -1 && i < k) { // walk to the root of i's current subtree, compressing the path to k
        const int32_t nxt = anc[i];
        anc[i] = k;
        if (nxt == -1) F.parent[i] = k;
        i = nxt;
      }
    }
  std::vector<int64_t> cnt(n, 1); // diagonal
  std::vector<int32_t> w(n, -1);
  for (int32_t k = 0; k < n; ++k) {
    w[k] = k;
    for (int64_t p = P.ptr[k]; p < P.ptr[k + 1]; ++p)
      for (int32_t i = P.col[(size_t)p]; i != -1 && w[i] != k; i = F.parent[i]) {
        w[i] = k;
        cnt[i]++; // L(k, i) != 0
      }
  }
  F.Lp.assign(n + 1, 0);
  double fl = 0.0;
  for (int32_t i = 0; i < n; ++i) {
    F.Lp[i + 1] = F.Lp[i] + cnt[i];
    fl += (double)cnt[i] * (double)cnt[i];
  }
  F.nnzL = F.Lp[n];
  F.flops = fl;
}

// up-looking numeric factorisation; false (F.error set) if the block is not positive definite
inline bool factorize(const PermutedLower &P, BlockFactor &F)
{
  const int32_t n = F.n;
  F.Li.resize((size_t)F.nnzL);
  F.Lx.resize((size_t)F.nnzL);
  std::vector<int64_t> c(F.Lp.begin(), F.Lp.end() - 1);
  std::vector<int32_t> w(n, -1), s(n);
  std::vector<double> x(n, 0.0);
  for (int32_t k = 0; k < n; ++k) {
    int32_t top = n;
    w[k] = k;
    for (int64_t p = P.ptr[k]; p < P.ptr[k + 1]; ++p) {
      int32_t i = P.col[(size_t)p];
      x[i] = P.val[(size_t)p];
      int32_t len = 0;
      for (; i != -1 && w[i] != k; i = F.parent[i]) {
        s[len++] = i;
        w[i] = k;
      }
      while (len > 0) s[--top] = s[--len];
    }
    double d = P.diag[k];
    for (; top < n; ++top) {
      const int32_t i = s[top];
      const double lki = x[i] / F.Lx[(size_t)F.Lp[i]];
      x[i] = 0.0;
      for (int64_t p = F.Lp[i] + 1; p < c[i]; ++p) x[F.Li[(size_t)p]] -= F.Lx[(size_t)p] * lki;
      d -= lki * lki;
      const int64_t q = c[i]++;
      F.Li[(size_t)q] = k;
      F.Lx[(size_t)q] = lki;
    }
    if (!(d > 0.0) || !std::isfinite(d)) {
      F.error = "matrix is not positive definite (pivot " + std::to_string(d) + " in eliminated row " + std::to_string(k) + ")";
      return false;
    }
    const int64_t q = c[k]++;
    F.Li[(size_t)q] = k;
    F.Lx[(size_t)q] = std::sqrt(d);
  }
  return true;
}

// ---- general (non-symmetric) blocks: L U without pivoting on the symmetrised pattern -----------------------------------
// For matrices whose symmetric part is positive definite (the SIPG / upwind DG operator of BASELINE configs[3]; what the
// reference hands to UMFPACK) Gaussian elimination without pivoting is stable; the pattern of L and of U^T is the Cholesky
// pattern of A + A^T, so ordering, elimination tree and storage are shared with the symmetric case: position q of column j
// holds L(k, j) in Lx and U(j, k) in Ux.  The diagonal of U sits at the head of each column in Ux (Lx there is 1).
struct PermutedLowerLU {
  PermutedLower lo;            // lo.val = a(k, j), j < k (new numbering), lo.diag = a(k, k)
  std::vector<double> valT;    // a(j, k) at the same positions
};
inline PermutedLowerLU permute_lower_lu(const int64_t *rp, const int32_t *ci, const double *va, int64_t r0, int64_t r1, const std::vector<int32_t> &perm)
{
  const int32_t n = (int32_t)(r1 - r0);
  std::vector<int32_t> iperm(n);
  for (int32_t k = 0; k < n; ++k) iperm[perm[k]] = k;
  // transpose of the block (local indices), rows sorted because the input rows are visited in order
  std::vector<int64_t> tp(n + 1, 0);
  for (int64_t i = r0; i < r1; ++i)
    for (int64_t p = rp[i]; p < rp[i + 1]; ++p)
      if (ci[p] >= r0 && ci[p] < r1) tp[ci[p] - r0 + 1]++;
  for (int32_t i = 0; i < n; ++i) tp[i + 1] += tp[i];
  std::vector<int32_t> tj((size_t)tp[n]);
  std::vector<double> tv((size_t)tp[n]);
  {
    std::vector<int64_t> pos(tp.begin(), tp.end() - 1);
    for (int64_t i = r0; i < r1; ++i)
      for (int64_t p = rp[i]; p < rp[i + 1]; ++p)
        if (ci[p] >= r0 && ci[p] < r1) {
          const int64_t q = pos[ci[p] - r0]++;
          tj[(size_t)q] = (int32_t)(i - r0);
          tv[(size_t)q] = va ? va[p] : 0.0;
        }
  }
  PermutedLowerLU P;
  P.lo.ptr.assign(n + 1, 0);
  P.lo.diag.assign(n, 0.0);
  // merged (pattern of A + A^T) lower rows in the new numbering; two passes: count, fill
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) {
      for (int32_t k = 0; k < n; ++k) P.lo.ptr[k + 1] += P.lo.ptr[k];
      P.lo.col.resize((size_t)P.lo.ptr[n]);
      P.lo.val.resize((size_t)P.lo.ptr[n]);
      P.valT.resize((size_t)P.lo.ptr[n]);
    }
    for (int32_t k = 0; k < n; ++k) {
      const int32_t io = perm[k];
      int64_t a = rp[r0 + io], b = tp[io];
      const int64_t a1 = rp[r0 + io + 1], b1 = tp[io + 1];
      int64_t q = pass ? P.lo.ptr[k] : 0;
      while (a < a1 || b < b1) {
        while (a < a1 && (ci[a] < r0 || ci[a] >= r1)) ++a;
        const int32_t ca = a < a1 ? (int32_t)(ci[a] - r0) : INT32_MAX, cb = b < b1 ? tj[(size_t)b] : INT32_MAX;
        if (ca == INT32_MAX && cb == INT32_MAX) break;
        const int32_t c = std::min(ca, cb);
        double v = 0.0, vt = 0.0;
        if (ca == c) v = va ? va[a++] : (a++, 0.0);
        if (cb == c) vt = tv[(size_t)b++];
        const int32_t jn = iperm[c];
        if (jn < k) {
          if (pass) {
            P.lo.col[(size_t)q] = jn;
            P.lo.val[(size_t)q] = v;
            P.valT[(size_t)q] = vt;
          }
          ++q;
        } else if (jn == k && pass) P.lo.diag[k] = v;
      }
      if (!pass) P.lo.ptr[k + 1] = q;
    }
  }
  return P;
}
// graph of A + A^T is what block_graph builds already (it inserts both directions)

inline bool factorize_lu(const PermutedLowerLU &P, BlockFactor &F, std::vector<double> &Ux)
{
  const int32_t n = F.n;
  F.Li.resize((size_t)F.nnzL);
  F.Lx.resize((size_t)F.nnzL);
  Ux.resize((size_t)F.nnzL);
  std::vector<int64_t> c(F.Lp.begin(), F.Lp.end() - 1);
  std::vector<int32_t> w(n, -1), s(n);
  std::vector<double> x(n, 0.0), y(n, 0.0);
  double amax = 0.0;
  for (double d : P.lo.diag) amax = std::max(amax, std::fabs(d));
  for (int32_t k = 0; k < n; ++k) {
    int32_t top = n;
    w[k] = k;
    for (int64_t p = P.lo.ptr[k]; p < P.lo.ptr[k + 1]; ++p) {
      int32_t i = P.lo.col[(size_t)p];
      x[i] = P.lo.val[(size_t)p];
      y[i] = P.valT[(size_t)p];
      int32_t len = 0;
      for (; i != -1 && w[i] != k; i = F.parent[i]) {
        s[len++] = i;
        w[i] = k;
      }
      while (len > 0) s[--top] = s[--len];
    }
    double d = P.lo.diag[k];
    for (; top < n; ++top) {
      const int32_t i = s[top];
      const double lki = x[i] / Ux[(size_t)F.Lp[i]]; // L(k, i)
      const double uik = y[i];                        // U(i, k)
      x[i] = 0.0;
      y[i] = 0.0;
      for (int64_t p = F.Lp[i] + 1; p < c[i]; ++p) {
        const int32_t r = F.Li[(size_t)p];
        x[r] -= Ux[(size_t)p] * lki; // a(k, r) -= L(k, i) U(i, r)
        y[r] -= F.Lx[(size_t)p] * uik; // a(r, k) -= L(r, i) U(i, k)
      }
      d -= lki * uik;
      const int64_t q = c[i]++;
      F.Li[(size_t)q] = k;
      F.Lx[(size_t)q] = lki;
      Ux[(size_t)q] = uik;
    }
    if (!(std::fabs(d) > 1e-14 * amax) || !std::isfinite(d)) {
      F.error = "zero pivot without pivoting (pivot " + std::to_string(d) + " in eliminated row " + std::to_string(k) + ")";
      return false;
    }
    const int64_t q = c[k]++;
    F.Li[(size_t)q] = k;
    F.Lx[(size_t)q] = 1.0;
    Ux[(size_t)q] = d;
  }
  return true;
}
// rows of an L U block factor in the ILU(0) storage convention: L(i, j) below the diagonal, 1 / U(i, i) on it, U(i, j) above
template <class VRP, class VCI, class VLU>
inline void append_rows_lu(const BlockFactor &F, const std::vector<double> &Ux, int64_t r0, VRP &rp, VCI &ci, VLU &lu,
                           std::vector<int64_t> &diag)
{
  const int32_t n = F.n;
  std::vector<int64_t> lcnt(n + 1, 0);
  for (int32_t j = 0; j < n; ++j)
    for (int64_t p = F.Lp[j] + 1; p < F.Lp[j + 1]; ++p) lcnt[F.Li[(size_t)p] + 1]++;
  const int64_t base = (int64_t)ci.size();
  std::vector<int64_t> start(n + 1, 0);
  for (int32_t i = 0; i < n; ++i) start[i + 1] = start[i] + lcnt[i + 1] + (F.Lp[i + 1] - F.Lp[i]);
  ci.resize((size_t)(base + start[n]));
  lu.resize((size_t)(base + start[n]));
  std::vector<int64_t> pos(n);
  for (int32_t i = 0; i < n; ++i) pos[i] = base + start[i];
  for (int32_t j = 0; j < n; ++j)
    for (int64_t p = F.Lp[j] + 1; p < F.Lp[j + 1]; ++p) {
      const int32_t i = F.Li[(size_t)p];
      ci[(size_t)pos[i]] = (int32_t)(r0 + j);
      lu[(size_t)pos[i]] = F.Lx[(size_t)p];
      pos[i]++;
    }
  for (int32_t i = 0; i < n; ++i) {
    diag.push_back(pos[i]);
    ci[(size_t)pos[i]] = (int32_t)(r0 + i);
    lu[(size_t)pos[i]] = 1.0 / Ux[(size_t)F.Lp[i]];
    pos[i]++;
    for (int64_t p = F.Lp[i] + 1; p < F.Lp[i + 1]; ++p) {
      ci[(size_t)pos[i]] = (int32_t)(r0 + F.Li[(size_t)p]);
      lu[(size_t)pos[i]] = Ux[(size_t)p];
      pos[i]++;
    }
    rp.push_back(pos[i]);
  }
}

// Appends the rows of this block's factor to a CSR in the storage convention of the ILU(0) engines (global permuted
// row numbers = r0 + new local index): strictly lower part l_ij / l_jj (unit lower factor), diagonal 1 / l_ii^2
// (inverse pivot), strictly upper part l_ii * l_ji (= D L^T).  Columns ascending in every row.
template <class VRP, class VCI, class VLU>
inline void append_rows(const BlockFactor &F, int64_t r0, VRP &rp, VCI &ci, VLU &lu, std::vector<int64_t> &diag)
{
  const int32_t n = F.n;
  std::vector<int64_t> lcnt(n + 1, 0); // strictly lower entries per row (transpose of the CSC factor)
  for (int32_t j = 0; j < n; ++j)
    for (int64_t p = F.Lp[j] + 1; p < F.Lp[j + 1]; ++p) lcnt[F.Li[(size_t)p] + 1]++;
  const int64_t base = (int64_t)ci.size();
  std::vector<int64_t> start(n + 1, 0);
  for (int32_t i = 0; i < n; ++i) start[i + 1] = start[i] + lcnt[i + 1] + (F.Lp[i + 1] - F.Lp[i]); // lower + (diag + upper)
  ci.resize((size_t)(base + start[n]));
  lu.resize((size_t)(base + start[n]));
  std::vector<int64_t> pos(n);
  for (int32_t i = 0; i < n; ++i) pos[i] = base + start[i];
  for (int32_t j = 0; j < n; ++j) { // columns ascending => lower parts of the rows fill in ascending column order
    const double ljj = F.Lx[(size_t)F.Lp[j]];
    for (int64_t p = F.Lp[j] + 1; p < F.Lp[j + 1]; ++p) {
      const int32_t i = F.Li[(size_t)p];
      ci[(size_t)pos[i]] = (int32_t)(r0 + j);
      lu[(size_t)pos[i]] = F.Lx[(size_t)p] / ljj;
      pos[i]++;
    }
  }
  for (int32_t i = 0; i < n; ++i) {
    const double lii = F.Lx[(size_t)F.Lp[i]];
    diag.push_back(pos[i]);
    ci[(size_t)pos[i]] = (int32_t)(r0 + i);
    lu[(size_t)pos[i]] = 1.0 / (lii * lii);
    pos[i]++;
    for (int64_t p = F.Lp[i] + 1; p < F.Lp[i + 1]; ++p) { // column i of L below the diagonal = row i of L^T
      ci[(size_t)pos[i]] = (int32_t)(r0 + F.Li[(size_t)p]);
      lu[(size_t)pos[i]] = lii * F.Lx[(size_t)p];
      pos[i]++;
    }
    rp.push_back(pos[i]);
  }
}

} // namespace chol
