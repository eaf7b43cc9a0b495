// Host-side generator of the benchmark's INPUT matrices (no device code, no ddm_ctx): the Q1 diffusion matrix of a structured node box
// restricted to a node subset, in the caller's numbering, Dirichlet rows/columns eliminated symmetrically -- what
// dune_ddm_amd/synth.py:StructuredPoisson.region_matrix computes with whole-array numpy passes (assemble_stencil -> stencil_to_csr ->
// COO renumbering -> sort -> eliminate_dirichlet; 3.7 s per matrix family at 216^3) done row by row on host threads.  It stands where
// PDELab's assembler stands in the reference (examples/pdelab_helper.hh:113-436 hands A_dir / A_neu / B_neu to the coarse spaces);
// it is input synthesis for bench.py and the tests, not a piece of the preconditioner.
//
// Bit-for-bit the numpy result: a stencil entry is the sum over the corners a = 0 .. 2^dim - 1 (ascending, as assemble_stencil's outer
// loop) of kappa_e K[a][b], e = the element whose corner a the row's node is, b = a + offset.
#pragma once
#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

namespace synth {

struct Q1Args {
    int dim;
    const int64_t *bshape;      // node box, x first
    const double *ke;           // element coefficients, C order (x fastest), shape eshape; elements that do not count hold 0
    const int64_t *eshape;      // x first
    const int64_t *eoff;        // element index of the element whose corner 0 is box node 0 (0, or 1 with a surrounding element layer)
    const double *K;            // 2^dim x 2^dim element matrix
    const uint8_t *inset;       // per box node: member of the node subset (NULL: all)
    const int64_t *loc_of_box;  // per box node: local index (NULL: identity)
    int64_t n;                  // local rows
    const int64_t *box_index;   // per local row: its box node (NULL: identity)
    const uint8_t *dmask;       // per local row: Dirichlet
    const double *diag;         // per local row: value of a Dirichlet diagonal (NULL: 1.0)
};

// one row: pattern (local columns, ascending) and values; returns the entry count.  cols/vals may be NULL (count only).
inline int q1_row(const Q1Args &A, int64_t i, int32_t *cols, double *vals)
{
    const int dim = A.dim, nc = 1 << dim;
    int64_t nbx[3] = {1, 1, 1}, c[3] = {0, 0, 0}, stride[3] = {1, 1, 1}, estride[3] = {1, 1, 1};
    for (int d = 0; d < dim; ++d) nbx[d] = A.bshape[d];
    for (int d = 1; d < dim; ++d) { stride[d] = stride[d - 1] * nbx[d - 1]; estride[d] = estride[d - 1] * A.eshape[d - 1]; }
    const int64_t p = A.box_index ? A.box_index[i] : i;
    if (A.inset && !A.inset[p]) return 0;
    { int64_t q = p; for (int d = 0; d < dim; ++d) { c[d] = q % nbx[d]; q /= nbx[d]; } }
    int cnt = 0;
    int32_t lc[27]; double lv[27];
    const int no = dim == 3 ? 27 : 9;
    for (int o = 0; o < no; ++o) {
        int off[3] = {0, 0, 0};
        { int q = o; for (int d = 0; d < dim; ++d) { off[d] = q % 3 - 1; q /= 3; } }
        bool ok = true; int64_t nb = p;
        for (int d = 0; d < dim; ++d) { int64_t x = c[d] + off[d]; ok = ok && x >= 0 && x < nbx[d]; nb += off[d] * stride[d]; }
        if (!ok || (A.inset && !A.inset[nb])) continue;
        const int64_t col = A.loc_of_box ? A.loc_of_box[nb] : nb;
        if (cols) {
            double s = 0.0;
            for (int a = 0; a < nc; ++a) {
                int b = 0; bool in = true; int64_t e = 0;
                for (int d = 0; d < dim; ++d) {
                    const int ab = (a >> d) & 1, bb = ab + off[d];
                    const int64_t ed = c[d] - ab + A.eoff[d];
                    in = in && (bb == 0 || bb == 1) && ed >= 0 && ed < A.eshape[d];
                    b |= (bb & 1) << d; e += ed * estride[d];
                }
                if (!in) continue;
                const double k = A.K[a * nc + b];
                if (k != 0.0) { const double t = A.ke[e] * k; s += t; }   // product rounded, then added: as numpy (no contraction)
            }
            lv[cnt] = s;
        }
        lc[cnt++] = (int32_t)col;
    }
    if (!cols) return cnt;
    // ascending local columns (box order is ascending already when the numbering is the box's own)
    int order[27];
    for (int k = 0; k < cnt; ++k) order[k] = k;
    if (A.loc_of_box) std::sort(order, order + cnt, [&](int x, int y) { return lc[x] < lc[y]; });
    const bool drow = A.dmask && A.dmask[i];
    for (int k = 0; k < cnt; ++k) {
        const int32_t col = lc[order[k]];
        double v = lv[order[k]];
        if (drow) v = (col == (int32_t)i) ? (A.diag ? A.diag[i] : 1.0) : 0.0;
        else if (A.dmask && A.dmask[col]) v = 0.0;
        cols[k] = col; vals[k] = v;
    }
    return cnt;
}

inline void q1_rows(const Q1Args &A, int64_t *indptr, int32_t *indices, double *data, int nthreads)
{
    nthreads = std::max(1, nthreads);
    auto run = [&](auto &&f) {
        std::vector<std::thread> th;
        const int64_t chunk = (A.n + nthreads - 1) / nthreads;
        for (int t = 0; t < nthreads; ++t) {
            const int64_t r0 = std::min<int64_t>(A.n, t * chunk), r1 = std::min<int64_t>(A.n, r0 + chunk);
            if (r0 < r1) th.emplace_back([=, &f] { for (int64_t i = r0; i < r1; ++i) f(i); });
        }
        for (auto &t : th) t.join();
    };
    if (!indices) {     // counting call: indptr[i + 1] = entries of row i, then the prefix sum
        indptr[0] = 0;
        run([&](int64_t i) { indptr[i + 1] = q1_row(A, i, nullptr, nullptr); });
        for (int64_t i = 0; i < A.n; ++i) indptr[i + 1] += indptr[i];
        return;
    }
    run([&](int64_t i) { q1_row(A, i, indices + indptr[i], data + indptr[i]); });
}

} // namespace synth
