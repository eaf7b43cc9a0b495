// Host side of the SUPERNODAL sparse Cholesky whose numeric factorisation and solves run on the device (sn_chol.hpp): ordering and
// symbolic analysis only -- no arithmetic.  (the reference gets this from CHOLMOD / UMFPACK behind the solver factory,
// dune/ddm/schwarz.hh:85-92, and inside the GenEO eigensolver, dune/ddm/eigensolvers/spectra.hh:28-89, umfpack.hh:16-333.)
//
//   ordering   : the nested dissection of sparse_chol_host.hpp; every emitted group (leaf region or separator) is a candidate
//                supernode, groups wider than SN_MAX_COLS are cut into a CHAIN of supernodes of at most SN_MAX_COLS columns;
//   supernodes : dense lower-triangular diagonal block (explicit zeros accepted inside a group) + ONE sorted list of rows below
//                shared by all columns (the union of the columns' structures), found by the usual bottom-up merge
//                struct(s) = adj_A(cols(s)) u U_{children c} struct(c), restricted to rows behind the supernode;
//   tree       : parent(s) = supernode of the first row below; level(s) = 1 + max level of the children.  Supernodes of one level
//                are independent: the device processes the tree level by level, all blocks (subdomains) at once.
// The panel of supernode s is the dense (ncol + nrow) x ncol column-major matrix [diagonal block; rows below].
#pragma once

#include <algorithm>
#include <cstdint>
#include <vector>

#include "sparse_chol_host.hpp"

namespace sn {

constexpr int SN_MAX_COLS = 128; // widest supernode: its diagonal block is factorised and inverted by ONE workgroup in LDS

struct BlockSym { // one diagonal block (subdomain), local permuted numbering
  int32_t n = 0;
  std::vector<int32_t> perm;  // perm[new] = old
  std::vector<int32_t> first; // [nsn + 1] first column of every supernode
  std::vector<int64_t> rptr;  // [nsn + 1] into rows
  std::vector<int32_t> rows;  // rows below, ascending
  std::vector<int32_t> parent, level;
  double flops = 0.0;
  int64_t entries = 0; // doubles in the panels
};

// Rough multiply-add count of the factorisation from the first separator alone: nested dissection costs c * s^3 with s the top
// separator (s = n^(2/3) in 3-D: 14 n^2 = 14 s^3 measured on 27-point grids; s = n^(1/2) in 2-D: ~10-30 s^3).  Used only to DECLINE
// hopeless sizes early (the caller compares 10 s^3 against a multiple of its limit); 0 if the block is too small to split.
inline double estimate_flops(const chol::Graph &G, int leaf = 48)
{
  if (G.n <= leaf) return 0.0;
  int64_t sep = -1;
  (void)chol::nested_dissection(G, leaf, nullptr, &sep);
  return sep > 0 ? 10.0 * (double)sep * (double)sep * (double)sep : 0.0;
}

// G: symmetric pattern of the block without the diagonal (chol::block_graph)
inline BlockSym analyse(const chol::Graph &G, int leaf = 48)
{
  BlockSym S;
  const int32_t n = G.n;
  S.n = n;
  std::vector<int32_t> ends;
  S.perm = chol::nested_dissection(G, leaf, &ends);
  std::vector<int32_t> iperm((size_t)n);
  for (int32_t k = 0; k < n; ++k) iperm[(size_t)S.perm[(size_t)k]] = k;
  // supernode boundaries: groups, wide ones cut into chains
  S.first.push_back(0);
  int32_t g0 = 0;
  for (int32_t e : ends) {
    for (int32_t c = g0; c < e;) {
      const int32_t w = std::min<int32_t>(SN_MAX_COLS, e - c);
      c += w;
      S.first.push_back(c);
    }
    g0 = e;
  }
  const int32_t nsn = (int32_t)S.first.size() - 1;
  std::vector<int32_t> sn_of((size_t)n);
  for (int32_t s = 0; s < nsn; ++s)
    for (int32_t c = S.first[(size_t)s]; c < S.first[(size_t)s + 1]; ++c) sn_of[(size_t)c] = s;
  S.parent.assign((size_t)nsn, -1);
  S.level.assign((size_t)nsn, 0);
  S.rptr.assign((size_t)nsn + 1, 0);
  std::vector<std::vector<int32_t>> children((size_t)nsn);
  std::vector<int32_t> mark((size_t)n, -1), cur;
  for (int32_t s = 0; s < nsn; ++s) {
    const int32_t c0 = S.first[(size_t)s], c1 = S.first[(size_t)s + 1];
    cur.clear();
    for (int32_t c = c0; c < c1; ++c) {
      const int32_t v = S.perm[(size_t)c];
      for (int64_t k = G.ptr[(size_t)v]; k < G.ptr[(size_t)v + 1]; ++k) {
        const int32_t i = iperm[(size_t)G.adj[(size_t)k]];
        if (i >= c1 && mark[(size_t)i] != s) {
          mark[(size_t)i] = s;
          cur.push_back(i);
        }
      }
    }
    int32_t lev = 0;
    for (int32_t c : children[(size_t)s]) {
      lev = std::max(lev, S.level[(size_t)c] + 1);
      for (int64_t k = S.rptr[(size_t)c]; k < S.rptr[(size_t)c + 1]; ++k) {
        const int32_t i = S.rows[(size_t)k];
        if (i >= c1 && mark[(size_t)i] != s) {
          mark[(size_t)i] = s;
          cur.push_back(i);
        }
      }
      std::vector<int32_t>().swap(children[(size_t)c]);
    }
    std::sort(cur.begin(), cur.end());
    S.level[(size_t)s] = lev;
    S.rows.insert(S.rows.end(), cur.begin(), cur.end());
    S.rptr[(size_t)s + 1] = (int64_t)S.rows.size();
    if (!cur.empty()) {
      S.parent[(size_t)s] = sn_of[(size_t)cur[0]];
      children[(size_t)S.parent[(size_t)s]].push_back(s);
    }
    const double nc = (double)(c1 - c0), nr = (double)cur.size();
    S.flops += nc * nc * nc / 3.0 + nr * nc * nc + nr * nr * nc; // potrf + panel solve + update (multiply-adds x 2 for the last two)
    S.entries += (int64_t)(c1 - c0) * (int64_t)(c1 - c0 + (int64_t)cur.size());
  }
  return S;
}

} // namespace sn
