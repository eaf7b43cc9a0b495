"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of dune-ddm's hot path over a list of
*simulated* ranks in one process: every class below mirrors the reference class of the same
name and cites the lines it follows (paths relative to /root/reference).  Inner loops are in
oracle/kernels.c (plain C, -ffp-contract=off).

Parity status
  * GalerkinPreconditioner.build_solver / gather layout: PINNED by the reference's golden 4x4
    coarse matrix (tests/test_galerkin_coarse_matrix.cc:50-67) -- tests/test_oracle_kat.py.
  * overlap extension / overlapping matrix: PINNED by the 9x9 chain KAT (same file :198-212).
  * SchwarzPreconditioner.apply, GalerkinPreconditioner.apply, CombinedPreconditioner.apply,
    NonOverlappingOperator, CG: the reference stores no expected outputs for these (SURVEY.md
    8c) => "parity unpinned" by the reference; they are literal restatements, cross-checked
    against scipy direct solves in tests/test_oracle_apply.py.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
        L = ctypes.CDLL(so)
        P = ctypes.c_void_p
        I = ctypes.c_int64
        D = ctypes.c_double
        L.orc_csr_mv.argtypes = [I, P, P, P, P, P]
        L.orc_csr_usmv.argtypes = [I, P, P, P, D, P, P]
        L.orc_ilu0_factor.argtypes = [I, P, P, P, P]
        L.orc_ilu0_factor.restype = ctypes.c_int
        L.orc_ilu0_solve.argtypes = [I, P, P, P, P, P, P]
        L.orc_masked_dot.argtypes = [I, P, P, P]
        L.orc_masked_dot.restype = D
        L.orc_masked_dot_order.argtypes = [I, P, P, P, ctypes.c_int]
        L.orc_masked_dot_order.restype = D
        L.orc_dot.argtypes = [I, P, P]
        L.orc_dot.restype = D
        L.orc_axpy.argtypes = [I, D, P, P]
        L.orc_xpby.argtypes = [I, P, D, P]
        L.orc_scale.argtypes = [I, P, P]
        L.orc_dense_lu.argtypes = [I, P, P]
        L.orc_dense_lu.restype = ctypes.c_int
        L.orc_dense_lu_solve.argtypes = [I, P, P, P]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# One host thread per simulated rank (the reference runs one MPI rank per subdomain); the C
# kernels release the GIL.  THREADS = 1 keeps everything sequential (default, used by the tests).
THREADS = 1
_POOL = None


def set_threads(n):
    global THREADS, _POOL
    THREADS = max(1, int(n))
    if _POOL is not None:
        _POOL.shutdown()
        _POOL = None
    if THREADS > 1:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(THREADS)


def pfor(fn, n):
    """for r in range(n): fn(r) -- ranks are independent between communication calls"""
    if THREADS > 1 and n > 1:
        list(_POOL.map(fn, range(n)))
    else:
        for r in range(n):
            fn(r)


# Summation order of the global dot products: 0 = the reference's (ascending index, ascending rank).  1 / 2 are used by the
# order-sensitivity tests only (see orc_masked_dot_order in kernels.c).
DOT_ORDER = 0


def set_dot_order(order):
    global DOT_ORDER
    assert order in (0, 1, 2)
    DOT_ORDER = int(order)


class Csr:
    """Flattened scalar BCRSMatrix (int64 row pointers, int32 columns, f64 values)."""

    def __init__(self, M):
        M = sp.csr_matrix(M)
        self.n = M.shape[0]
        self.rp = np.ascontiguousarray(M.indptr, dtype=np.int64)
        self.ci = np.ascontiguousarray(M.indices, dtype=np.int32)
        self.v = np.ascontiguousarray(M.data, dtype=np.float64)
        self.shape = M.shape

    def mv(self, x, y):
        lib().orc_csr_mv(self.n, _p(self.rp), _p(self.ci), _p(self.v), _p(x), _p(y))

    def usmv(self, alpha, x, y):
        lib().orc_csr_usmv(self.n, _p(self.rp), _p(self.ci), _p(self.v), float(alpha), _p(x), _p(y))

    def scipy(self):
        return sp.csr_matrix((self.v, self.ci, self.rp), shape=self.shape)


class Ilu0:
    """dune-istl SeqILU(n=0) applied once (loopsolver maxit=1) -- kernels.c:orc_ilu0_*."""

    def __init__(self, A: Csr):
        self.A = A
        self.lu = A.v.copy()
        self.diag = np.empty(A.n, dtype=np.int64)
        rc = lib().orc_ilu0_factor(A.n, _p(A.rp), _p(A.ci), _p(self.lu), _p(self.diag))
        if rc:
            raise ZeroDivisionError(f"ILU(0): zero/missing pivot in row {rc - 1}")

    def apply(self, x, d):
        lib().orc_ilu0_solve(self.A.n, _p(self.A.rp), _p(self.A.ci), _p(self.lu), _p(self.diag), _p(d), _p(x))


class DirectSolver:
    """Exact local solve (the shipped .ini files use cholmod/umfpack, examples/poisson.ini:23,26)."""

    def __init__(self, A: Csr):
        import scipy.sparse.linalg as spl
        self.lu = spl.splu(A.scipy().tocsc())

    def apply(self, x, d):
        x[:] = self.lu.solve(d)


# ---------------------------------------------------------------------------------------------
@dataclass
class Comm:
    """Simulated OwnerOverlapCopyCommunication over all ranks: pair lists per ordered (src,dst)."""
    nranks: int
    owner_to_all: dict      # copyOwnerToAll interface
    all_to_all: dict        # addOwnerCopyToOwnerCopy / addOwnerCopyToAll interface
    owner: list             # per rank uint8 owner mask (over that communicator's index set)

    def neighbours(self, r):
        return sorted({d for (s, d) in self.all_to_all if s == r} | {s for (s, d) in self.all_to_all if d == r})

    def _forward(self, pairs, vecs, add):
        bufs = {k: vecs[k[0]][s].copy() for k, (s, d) in pairs.items()}   # gather before any scatter
        for (src, dst) in sorted(pairs, key=lambda k: (k[1], k[0])):       # per receiver: ascending sender rank
            d = pairs[(src, dst)][1]
            if add:
                vecs[dst][d] += bufs[(src, dst)]
            else:
                vecs[dst][d] = bufs[(src, dst)]

    def copyOwnerToAll(self, vecs):
        self._forward(self.owner_to_all, vecs, add=False)

    def addOwnerCopyToOwnerCopy(self, vecs):
        self._forward(self.all_to_all, vecs, add=True)

    addOwnerCopyToAll = addOwnerCopyToOwnerCopy   # only owner/copy attributes exist (overlap_extension.hh:261)

    def dot(self, xs, ys):
        """OwnerOverlapCopyCommunication::dot: owner-masked local sums + MPI_Allreduce(SUM)
        (ranks added in rank order here).  DOT_ORDER != 0 (set_dot_order; NOT reference semantics, order-sensitivity tests only):
        the same sums with the additions in another order -- 1: descending index and rank, 2: pairwise."""
        if DOT_ORDER:
            parts = [lib().orc_masked_dot_order(len(xs[r]), _p(self.owner[r]), _p(xs[r]), _p(ys[r]), DOT_ORDER) for r in range(self.nranks)]
            tot = 0.0
            for v in (reversed(parts) if DOT_ORDER == 1 else parts):
                tot += v
            return tot
        tot = 0.0
        for r in range(self.nranks):
            tot += lib().orc_masked_dot(len(xs[r]), _p(self.owner[r]), _p(xs[r]), _p(ys[r]))
        return tot

    def norm(self, xs):
        return float(np.sqrt(self.dot(xs, xs)))


class NonOverlappingOperator:
    """dune/ddm/nonoverlapping_operator.hh:11-58."""

    def __init__(self, A: list, comm: Comm):
        self.A, self.comm = A, comm

    def apply(self, x, y):                                   # :34-39
        pfor(lambda r: self.A[r].mv(x[r], y[r]), len(self.A))
        self.comm.addOwnerCopyToOwnerCopy(y)

    def applyscaleadd(self, alpha, x, y):                    # :41-50
        y1 = [v.copy() for v in y]
        for r, A in enumerate(self.A):
            y[r][:] = 0.0
            A.usmv(alpha, x[r], y[r])
        self.comm.addOwnerCopyToOwnerCopy(y)
        for r in range(len(y)):
            y[r] += y1[r]


class NonOverlappingScalarProduct:
    """dune/ddm/nonoverlapping_operator.hh:63-89."""

    def __init__(self, comm: Comm):
        self.comm = comm

    def dot(self, x, y):
        return self.comm.dot(x, y)

    def norm(self, x):
        return self.comm.norm(x)


class SchwarzPreconditioner:
    """dune/ddm/schwarz.hh:54-220.  ``solver_factory(Csr) -> object with apply(x, d)``."""

    def __init__(self, Aovlp: list, comm: Comm, pou, type="restricted", solver_factory=Ilu0):
        if type not in ("restricted", "standard"):
            raise NotImplementedError("Unknown Schwarz type '" + type + "'")   # :83
        self.A, self.comm, self.pou, self.type = Aovlp, comm, pou, type
        self.solver = [None] * len(Aovlp)

        def factor(r):
            self.solver[r] = solver_factory(Aovlp[r])                          # :92
        pfor(factor, len(Aovlp))
        for r, A in enumerate(Aovlp):                                          # :186-193
            if len(comm.owner[r]) != A.n:
                raise RuntimeError("Remote indices size does not match overlapping matrix size")
            if pou is not None and len(pou[r]) != A.n:
                raise RuntimeError("Partition of unity size does not match overlapping matrix size")
        self.d_ovlp = [np.zeros(A.n) for A in Aovlp]
        self.x_ovlp = [np.zeros(A.n) for A in Aovlp]

    def apply(self, x, d):                                    # :115-149
        for r in range(len(d)):
            self.d_ovlp[r][:] = 0.0                            # :121
            self.d_ovlp[r][:len(d[r])] = d[r]                  # :122
        self.comm.copyOwnerToAll(self.d_ovlp)                 # :125
        def local_solve(r):
            self.x_ovlp[r][:] = 0.0                            # :132
            self.solver[r].apply(self.x_ovlp[r], self.d_ovlp[r])   # :133
        pfor(local_solve, len(d))
        if self.type == "restricted" and self.pou is not None:     # :139-141
            for r in range(len(d)):
                lib().orc_scale(len(self.pou[r]), _p(self.pou[r]), _p(self.x_ovlp[r]))
        self.comm.addOwnerCopyToOwnerCopy(self.x_ovlp)        # :138 / :142
        for r in range(len(x)):
            x[r][:] = self.x_ovlp[r][:len(x[r])]               # :146


def gather_matrix_from_rows_flat(rows_per_rank, n_cols, clip_tolerance=0.0):
    """gatherMatrixFromRowsFlat (dune/ddm/helpers.hh:204-339): each rank's slab is column-major
    ``rows[row + col * n_rows]`` (:252); entries with |v| <= clip are dropped (:253); the global
    matrix stacks the ranks' rows in rank order.  Returns the K x n_cols CSR that rank 0 holds."""
    rp, ci, vv = [0], [], []
    for rows in rows_per_rank:
        if len(rows) == 0:
            raise RuntimeError("No rows to build matrix from")                 # :214
        if len(rows) % n_cols != 0:
            raise RuntimeError("Rows size is not a multiple of the number of columns")  # :216
        n_rows = len(rows) // n_cols
        for row in range(n_rows):
            for col in range(n_cols):
                value = rows[row + col * n_rows]
                if abs(value) > clip_tolerance:
                    ci.append(col)
                    vv.append(value)
            rp.append(len(ci))
    return sp.csr_matrix((np.array(vv), np.array(ci, dtype=np.int64), np.array(rp, dtype=np.int64)),
                         shape=(len(rp) - 1, n_cols))


class DenseLU:
    def __init__(self, A0):
        self.n = A0.shape[0]
        self.a = np.ascontiguousarray(A0.toarray() if sp.issparse(A0) else A0, dtype=np.float64).copy()
        self.piv = np.empty(self.n, dtype=np.int64)
        if lib().orc_dense_lu(self.n, _p(self.a), _p(self.piv)):
            raise ZeroDivisionError("singular coarse matrix")

    def solve(self, b):
        x = np.ascontiguousarray(b, dtype=np.float64).copy()
        lib().orc_dense_lu_solve(self.n, _p(self.a), _p(self.piv), _p(x))
        return x


class GalerkinPreconditioner:
    """dune/ddm/galerkin_preconditioner.hh:40-363.  ``ts[r]`` = list of template vectors of rank r
    (each of the overlapping size)."""

    def __init__(self, A: list, ts: list, comm: Comm):
        self.comm = comm
        self.num_t = [len(t) for t in ts]
        for r, t in enumerate(ts):
            if len(t) == 0:
                raise RuntimeError("Must at least pass one template vector")   # :129
            if len(t[0]) != A[r].n:
                raise RuntimeError("Template vectors must match size of matrix")   # :131
        self.restr_vecs = [[np.array(v, dtype=np.float64) for v in t] for t in ts]  # copies (:138-139)
        self.d_ovlp = [np.zeros(a.n) for a in A]
        self.x_ovlp = [np.zeros(a.n) for a in A]
        self.build_solver(A)

    def build_solver(self, A):                                # :219-349
        P = self.comm.nranks
        self.num_t_per_rank = list(self.num_t)                # MPI_Allgather (:248)
        self.total_num_t = int(sum(self.num_t))
        self.offset_per_rank = [int(x) for x in np.concatenate([[0], np.cumsum(self.num_t)[:-1]])]  # :256
        pairs = self.comm.all_to_all
        slabs = []
        for me in range(P):
            k = self.num_t[me]
            rows = np.zeros(k * self.total_num_t)
            y = np.zeros(A[me].n)
            for idx in range(k):                              # local x local (:292-295)
                A[me].mv(self.restr_vecs[me][idx], y)
                for j in range(k):
                    rows[(self.offset_per_rank[me] + idx) * k + j] = lib().orc_dot(len(y), _p(self.restr_vecs[me][j]), _p(y))
            for nb in self.comm.neighbours(me):               # local x remote (:298-309, 321-327)
                if (nb, me) not in pairs:
                    continue
                s_idx, d_idx = pairs[(nb, me)]
                for idx in range(self.num_t[nb]):
                    other = np.zeros(A[me].n)                 # vd.others[rank]: zero outside the shared indices (:96-101)
                    other[d_idx] = self.restr_vecs[nb][idx][s_idx]
                    A[me].mv(other, y)
                    for j in range(k):
                        rows[(self.offset_per_rank[nb] + idx) * k + j] = lib().orc_dot(len(y), _p(self.restr_vecs[me][j]), _p(y))
            slabs.append(rows)
        self.slabs = slabs
        self.a0 = gather_matrix_from_rows_flat(slabs, self.total_num_t)   # :331
        self.solver = DenseLU(self.a0)                         # rank 0 (:335-347)

    def apply(self, x, d):                                    # :151-194
        P = self.comm.nranks
        for r in range(P):
            self.d_ovlp[r][:len(d[r])] = d[r]                  # :159 (no zero fill: the tail is overwritten by the copy below)
        self.comm.copyOwnerToAll(self.d_ovlp)                 # :162
        d0 = np.zeros(self.total_num_t)

        def restrict(r):                                       # :165-171
            for k in range(self.num_t[r]):
                d0[self.offset_per_rank[r] + k] = lib().orc_dot(len(self.d_ovlp[r]), _p(self.restr_vecs[r][k]), _p(self.d_ovlp[r]))
        pfor(restrict, P)
        x0 = self.solver.solve(d0)                             # :174-179

        def prolong(r):                                        # :186-188
            self.x_ovlp[r][:] = 0.0
            for k in range(self.num_t[r]):
                lib().orc_axpy(len(self.x_ovlp[r]), float(x0[self.offset_per_rank[r] + k]), _p(self.restr_vecs[r][k]), _p(self.x_ovlp[r]))
        pfor(prolong, P)
        self.comm.addOwnerCopyToAll(self.x_ovlp)              # :190
        for r in range(P):
            x[r][:] = self.x_ovlp[r][:len(x[r])]               # :193


class CombinedPreconditioner:
    """dune/ddm/combined_preconditioner.hh:39-180."""

    def __init__(self, mode="additive"):
        if mode not in ("additive", "multiplicative"):
            raise NotImplementedError("Unknown apply mode in CombinedPreconditioner, use either additive or multiplicative")  # :68
        self.mode, self.precs, self.A = mode, [], None

    def add(self, prec):
        self.precs.append(prec)

    def set_op(self, A):
        self.A = A

    def apply(self, x, d):                                    # :127-163
        assert self.precs
        for v in x:
            v[:] = 0.0
        self.precs[0].apply(x, d)
        if self.mode == "additive":
            xnext = [np.zeros_like(v) for v in x]
            for prec in self.precs[1:]:
                for v in xnext:
                    v[:] = 0.0
                prec.apply(xnext, d)
                for r in range(len(x)):
                    x[r] += xnext[r]
        else:
            if self.A is None:
                raise RuntimeError("ERROR: ApplyMode is multiplicative but operator A is not provided. Set with `set_op`")  # :146
            dnext = [v.copy() for v in d]
            for prec in self.precs[1:]:
                self.A.applyscaleadd(-1.0, x, dnext)
                xnext = [np.zeros_like(v) for v in x]
                prec.apply(xnext, dnext)
                for r in range(len(x)):
                    x[r] += xnext[r]


def cg_solve(op, sp_, prec, x, b, reduction=1e-10, maxit=1000):
    """dune-istl CGSolver::apply (DUNE 2.10 solvers.hh, not in the snapshot; recurrences restated
    in SURVEY.md 3.2; called at examples/poisson.cc:311-319).  b is overwritten by the defect.
    Returns (iterations, converged, [def_0, def_1, ...])."""
    op.applyscaleadd(-1.0, x, b)
    def0 = sp_.norm(b)
    hist = [def0]
    if def0 < 1e-30 or def0 == 0.0:
        return 0, True, hist
    p = [np.zeros_like(v) for v in x]
    q = [np.zeros_like(v) for v in x]
    prec.apply(p, b)
    rholast = sp_.dot(p, b)
    it, conv = 0, False
    for i in range(1, maxit + 1):
        op.apply(p, q)
        alpha = sp_.dot(p, q)
        lam = rholast / alpha
        for r in range(len(x)):
            lib().orc_axpy(len(x[r]), lam, _p(p[r]), _p(x[r]))
            lib().orc_axpy(len(x[r]), -lam, _p(q[r]), _p(b[r]))
        deff = sp_.norm(b)
        hist.append(deff)
        it = i
        if deff < def0 * reduction or deff < 1e-30:
            conv = True
            break
        for v in q:
            v[:] = 0.0
        prec.apply(q, b)
        rho = sp_.dot(q, b)
        beta = rho / rholast
        for r in range(len(x)):
            lib().orc_xpby(len(p[r]), _p(q[r]), beta, _p(p[r]))
        rholast = rho
    return it, conv, hist


def pou_coarse_space(pou, template_vecs=None):
    """POUCoarseSpace (dune/ddm/coarsespaces/coarse_spaces.hh:1175-1231) with
    detail::finalize_eigenvectors (:52-61): v *= pou; v /= ||v||_2."""
    out = []
    for r, w in enumerate(pou):
        vecs = [np.ones(len(w))] if template_vecs is None else [np.array(t, dtype=float) for t in template_vecs[r]]
        fin = []
        for v in vecs:
            v = v * w
            fin.append(v * (1.0 / np.sqrt(float(np.dot(v, v)))))
        out.append(fin)
    return out


def _generate_plane_rotation(dx, dy):
    ndx, ndy = abs(dx), abs(dy)
    if ndy < 1e-15:
        return 1.0, 0.0
    if ndx < 1e-15:
        return 0.0, 1.0
    if ndy > ndx:
        t = ndx / ndy
        cs = 1.0 / np.sqrt(1.0 + t * t)
        sn = cs
        cs *= t
        sn *= dx / ndx
        sn *= dy / ndy
        return cs, sn
    t = ndy / ndx
    cs = 1.0 / np.sqrt(1.0 + t * t)
    return cs, cs * (dy / dx)


def gmres_solve(op, sp_, prec, x, b, reduction=1e-10, maxit=1000, restart=100):
    """dune-istl RestartedGMResSolver::apply (DUNE 2.10 solvers.hh, not in the snapshot; selected by
    examples/poisson.ini:12-17 and dune/ddm/twolevel_schwarz.hh:121-130): left preconditioning,
    modified Gram-Schmidt, Givens rotations; the monitored norm is the preconditioned defect.
    Returns (iterations, converged, [norm_0, norm_1, ...])."""
    P = len(x)
    m = restart

    def zeros():
        return [np.zeros_like(v) for v in x]

    op.applyscaleadd(-1.0, x, b)
    V = [zeros()]
    prec.apply(V[0], b)
    norm = sp_.norm(V[0])
    def0 = norm
    hist = [def0]
    if def0 < 1e-30:
        return 0, True, hist
    j, conv = 0, False
    w = zeros()
    while j < maxit and not conv:
        for r in range(P):
            V[0][r] *= 1.0 / norm
        s = np.zeros(m + 1)
        s[0] = norm
        H = np.zeros((m + 1, m))
        cs, sn = np.zeros(m), np.zeros(m)
        i = 0
        while i < m and j < maxit and not conv:
            if len(V) <= i + 1:
                V.append(zeros())
            op.apply(V[i], V[i + 1])
            for v in w:
                v[:] = 0.0
            prec.apply(w, V[i + 1])
            for k in range(i + 1):
                H[k, i] = sp_.dot(V[k], w)
                for r in range(P):
                    lib().orc_axpy(len(w[r]), -H[k, i], _p(V[k][r]), _p(w[r]))
            H[i + 1, i] = sp_.norm(w)
            if abs(H[i + 1, i]) < 1e-80:
                raise ZeroDivisionError("breakdown in GMRes - |w| == 0.0")
            for r in range(P):
                V[i + 1][r][:] = w[r] * (1.0 / H[i + 1, i])
            for k in range(i):
                t = cs[k] * H[k, i] + sn[k] * H[k + 1, i]
                H[k + 1, i] = -sn[k] * H[k, i] + cs[k] * H[k + 1, i]
                H[k, i] = t
            cs[i], sn[i] = _generate_plane_rotation(H[i, i], H[i + 1, i])
            t = cs[i] * H[i, i] + sn[i] * H[i + 1, i]
            H[i + 1, i] = -sn[i] * H[i, i] + cs[i] * H[i + 1, i]
            H[i, i] = t
            t = cs[i] * s[i] + sn[i] * s[i + 1]
            s[i + 1] = -sn[i] * s[i] + cs[i] * s[i + 1]
            s[i] = t
            norm = abs(s[i + 1])
            hist.append(norm)
            i += 1
            j += 1
            if norm < def0 * reduction or norm < 1e-30:
                conv = True
        y = np.zeros(i)
        for a in range(i - 1, -1, -1):
            t = s[a]
            for c in range(a + 1, i):
                t -= H[a, c] * y[c]
            y[a] = t / H[a, a]
        upd = zeros()
        for a in range(i):
            for r in range(P):
                lib().orc_axpy(len(upd[r]), float(y[a]), _p(V[a][r]), _p(upd[r]))
        for r in range(P):
            x[r] += upd[r]
        if not conv and j < maxit:
            op.applyscaleadd(-1.0, upd, b)
            for v in V[0]:
                v[:] = 0.0
            prec.apply(V[0], b)
            norm = sp_.norm(V[0])
    return j, conv, hist


def bicgstab_solve(op, sp_, prec, x, b, reduction=1e-10, maxit=1000):
    """dune-istl BiCGSTABSolver::apply (DUNE 2.10 solvers.hh, not in the snapshot; selectable through the same solver factory as
    examples/poisson.ini:12-17): right preconditioning, two half steps per iteration, the defect norm is tested after each half
    step.  Returns (iterations = ceil of the half-step counter, converged, [norm after every half step, starting with norm_0])."""
    P = len(x)
    EPS = 1e-80

    def zeros():
        return [np.zeros(len(v)) for v in x]

    def axpy(a, u, w):
        for r in range(P):
            lib().orc_axpy(len(w[r]), float(a), _p(u[r]), _p(w[r]))

    r = b
    op.applyscaleadd(-1.0, x, r)
    rt = [v.copy() for v in r]
    norm = sp_.norm(r)
    def0 = norm
    hist = [def0]
    if def0 < 1e-30:
        return 0, True, hist
    p, v, y, t = zeros(), zeros(), zeros(), zeros()
    rho = alpha = omega = 1.0
    it = 0.5
    conv = False
    while it < maxit:
        rho_new = sp_.dot(rt, r)
        if abs(rho) <= EPS or abs(omega) <= EPS:
            raise ArithmeticError("breakdown in BiCGSTAB")
        if it < 1:
            for q in range(P):
                p[q][:] = r[q]
        else:
            beta = (rho_new / rho) * (alpha / omega)
            axpy(-omega, v, p)
            for q in range(P):
                p[q] *= beta
            axpy(1.0, r, p)
        for q in range(P):
            y[q][:] = 0.0
        prec.apply(y, p)
        op.apply(y, v)
        h = sp_.dot(rt, v)
        if abs(h) < EPS:
            raise ArithmeticError("abs(h) < EPSILON in BiCGSTAB - abort")
        alpha = rho_new / h
        axpy(alpha, y, x)
        axpy(-alpha, v, r)
        norm = sp_.norm(r)
        hist.append(norm)
        if norm <= def0 * reduction:
            conv = True
            break
        it += 0.5
        for q in range(P):
            y[q][:] = 0.0
        prec.apply(y, r)
        op.apply(y, t)
        omega = sp_.dot(t, r) / sp_.dot(t, t)
        axpy(omega, y, x)
        axpy(-omega, t, r)
        rho = rho_new
        norm = sp_.norm(r)
        hist.append(norm)
        if norm <= def0 * reduction:
            conv = True
            break
        it += 0.5
    return int(np.ceil(min(it, maxit))), conv, hist
