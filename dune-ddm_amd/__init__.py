"""dune_ddm_amd -- MI355X-native two-level additive Schwarz + GenEO hot path (host-side mirror).

The compute lives in ``libddm_hip.so`` (hand-written HIP for gfx950, C ABI in include/ddm_hip.h).
This package is the thin host layer used by the tests and bench.py: ctypes bindings with the
reference's class names (SchwarzPreconditioner, GalerkinPreconditioner, CombinedPreconditioner,
NonOverlappingOperator), the flattening of DUNE-style index sets into exchange plans, and the
synthetic problem generator.  There is no CPU fallback: importing works anywhere (so that the
symbol table can be checked), every compute call needs a HIP device.

The directory name contains a hyphen, so import it through ``__graft_entry__.import_package()``.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libddm_hip.so")

GENEO_AVAILABLE = True   # dune-ddm_amd/geneo.py: device GenEO coarse-basis builder

DDM_OK, DDM_EINVAL, DDM_EHIP, DDM_ENOTIMPL, DDM_ENUMERIC, DDM_ECOMM = 0, -1, -2, -3, -4, -5


class DdmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ddm_hip error {code}: {msg}")
        self.code = code


class SolveResult(ctypes.Structure):
    _fields_ = [("iterations", ctypes.c_int32), ("converged", ctypes.c_int32), ("def0", ctypes.c_double),
                ("reduction", ctypes.c_double), ("elapsed_s", ctypes.c_double)]


class GeneoParams(ctypes.Structure):
    """ddm_geneo_params: the keys of `<prefix>.eigensolver` (dune/ddm/eigensolvers/eigensolver_params.hh:8-62)"""
    _fields_ = [("nev", ctypes.c_int32), ("nev_max", ctypes.c_int32), ("tolerance", ctypes.c_double), ("shift", ctypes.c_double),
                ("threshold", ctypes.c_double), ("maxit", ctypes.c_int32), ("extra", ctypes.c_int32), ("seed", ctypes.c_int32),
                ("preconditioner", ctypes.c_int32), ("max_direct_flops", ctypes.c_double), ("verbose", ctypes.c_int32), ("raw", ctypes.c_int32)]


class GeneoInfo(ctypes.Structure):
    _fields_ = [("iterations", ctypes.c_int32), ("converged", ctypes.c_int32), ("used_direct", ctypes.c_int32), ("nev", ctypes.c_int32),
                ("worst_residual", ctypes.c_double), ("setup_s", ctypes.c_double), ("iterate_s", ctypes.c_double), ("direct_flops", ctypes.c_double)]


A2A_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64)

_P, _I64, _I32, _D = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double
_PP = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes); must list every symbol declared in include/ddm_hip.h
SYMBOLS = {
    "ddm_ctx_create": (_I32, [_I32, _P, _PP]),
    "ddm_ctx_destroy": (None, [_P]),
    "ddm_last_error": (ctypes.c_char_p, [_P]),
    "ddm_ctx_sync": (_I32, [_P]),
    "ddm_ctx_fence": (_I32, [_P]),
    "ddm_ctx_stream": (_P, [_P]),
    "ddm_ctx_set_comm": (_I32, [_P, _I32, _I32, A2A_FN, ALLREDUCE_FN, _P]),
    "ddm_rccl_unique_id": (_I32, [_P]),
    "ddm_ctx_set_rccl": (_I32, [_P, _I32, _I32, _P, _I32]),
    "ddm_ctx_rccl_size": (_I32, [_P, ctypes.POINTER(ctypes.c_int)]),
    "ddm_ctx_comm_counts": (_I32, [_P, _P]),
    "ddm_malloc": (_I32, [_P, _I64, _PP]),
    "ddm_free": (_I32, [_P, _P]),
    "ddm_memset_zero": (_I32, [_P, _P, _I64]),
    "ddm_memcpy_h2d": (_I32, [_P, _P, _P, _I64]),
    "ddm_memcpy_d2h": (_I32, [_P, _P, _P, _I64]),
    "ddm_csr_create": (_I32, [_P, _I64, _I64, _P, _P, _P, _PP]),
    "ddm_csr_create_host": (_I32, [_P, _I64, _I64, _P, _P, _P, _PP]),
    "ddm_csr_destroy": (None, [_P]),
    "ddm_csr_rows": (_I64, [_P]),
    "ddm_csr_nnz": (_I64, [_P]),
    "ddm_csr_mv": (_I32, [_P, _P, _P, _P]),
    "ddm_csr_usmv": (_I32, [_P, _P, _D, _P, _P]),
    "ddm_csr_mm": (_I32, [_P, _P, _I32, _P, _P]),
    "ddm_csr_row_order_tiled_host": (_I32, [_I64, _P, _P, _P, _P]),
    "ddm_ilu0_solve_multi": (_I32, [_P, _P, _I32, _P, _P]),
    "ddm_ilu0_solve_multi_f32": (_I32, [_P, _P, _I32, _P, _P]),
    "ddm_ilu0_create": (_I32, [_P, _P, _I64, _P, _PP]),
    "ddm_ilu0_destroy": (None, [_P]),
    "ddm_ilu0_solve": (_I32, [_P, _P, _P, _P]),
    "ddm_ilu0_debug_stamps": (_I32, [_P, _P, _P, _P, _P]),
    "ddm_ilu0_status": (_I32, [_P, _P, ctypes.POINTER(ctypes.c_int)]),
    "ddm_ilu0_peek_status": (_I32, [_P]),
    "ddm_halo_exchange_to": (_I32, [_P, _P, _P, _P]),
    "ddm_schwarz_local_solver": (_P, [_P]),
    "ddm_ilu0_pipe_trace": (_I32, [_P, _P, _P, _P, _P, _P, _I64, ctypes.POINTER(ctypes.c_int64)]),
    "ddm_ilu0_num_levels": (_I64, [_P, _I32]),
    "ddm_ilu0_wait": (_I32, [_P, _P]),
    "ddm_ilu0_box_check": (_I32, [_P, _P]),
    "ddm_ilu0_engine": (_I32, [_P]),
    "ddm_chol_create": (_I32, [_P, _P, _I64, _P, _D, _PP]),
    "ddm_ilu0_is_direct": (_I32, [_P]),
    "ddm_ilu0_refinement": (_I32, [_P, _P]),
    "ddm_sn_host_create": (_I32, [_I64, _P, _P, _I64, _P, _PP]),
    "ddm_sn_host_destroy": (None, [_P]),
    "ddm_sn_host_sizes": (_I32, [_P, _I64, _P, ctypes.POINTER(ctypes.c_double)]),
    "ddm_sn_host_get": (_I32, [_P, _I64, _P, _P, _P, _P, _P, _P]),
    "ddm_ilu0_nnz": (_I64, [_P]),
    "ddm_chol_host_create": (_I32, [_I64, _P, _P, _P, _I64, _P, _PP]),
    "ddm_direct_host_create": (_I32, [_I64, _P, _P, _P, _I64, _P, _I32, _PP]),
    "ddm_direct_create": (_I32, [_P, _P, _I64, _P, _I32, _D, _PP]),
    "ddm_chol_host_destroy": (None, [_P]),
    "ddm_chol_host_nnz": (_I64, [_P]),
    "ddm_chol_host_nnz_factor": (_I64, [_P]),
    "ddm_chol_host_flops": (_D, [_P]),
    "ddm_chol_host_get": (_I32, [_P, _P, _P, _P, _P]),
    "ddm_schwarz_create_ex": (_I32, [_P, _P, _I64, _P, _I64, _P, _P, _I32, ctypes.c_char_p, _P, _P, _PP]),
    "ddm_schwarz_engine": (_I32, [_P]),
    "ddm_schwarz_factor_nnz": (_I64, [_P]),
    "ddm_schwarz_status": (_I32, [_P, _P]),
    "ddm_combined_status": (_I32, [_P, _P]),
    "ddm_ilu0_get_factors_host": (_I32, [_P, _P, _P]),
    "ddm_halo_create": (_I32, [_P, _I32, _I32, _I64, _P, _P, _P, _I64, _P, _P, _P, _PP]),
    "ddm_halo_destroy": (None, [_P]),
    "ddm_halo_exchange": (_I32, [_P, _P, _P]),
    "ddm_halo_sendbuf": (_P, [_P]),
    "ddm_halo_recvbuf": (_P, [_P]),
    "ddm_op_create": (_I32, [_P, _P, _P, _P, _PP]),
    "ddm_op_destroy": (None, [_P]),
    "ddm_op_apply": (_I32, [_P, _P, _P, _P]),
    "ddm_op_applyscaleadd": (_I32, [_P, _P, _D, _P, _P]),
    "ddm_dot": (_I32, [_P, _P, _P, _P, _P]),
    "ddm_norm": (_I32, [_P, _P, _P, _P]),
    "ddm_schwarz_create": (_I32, [_P, _P, _I64, _P, _I64, _P, _P, _I32, _P, _P, _PP]),
    "ddm_schwarz_destroy": (None, [_P]),
    "ddm_schwarz_apply": (_I32, [_P, _P, _P, _P]),
    "ddm_schwarz_num_levels": (_I64, [_P, _I32]),
    "ddm_galerkin_create": (_I32, [_P, _I64, _I64, _P, _I64, _P, _I64, _P, _P, _I64, _P, _P, _P, _PP]),
    "ddm_galerkin_destroy": (None, [_P]),
    "ddm_galerkin_apply": (_I32, [_P, _P, _P, _P]),
    "ddm_galerkin_products": (_I32, [_P, _P, _I64, _P, _I64, _P, _I64, _I64, _P]),
    "ddm_geneo_params_default": (_I32, [ctypes.POINTER(GeneoParams)]),
    "ddm_geneo_basis": (_I32, [_P, _P, _P, _I64, _P, _P, _P, ctypes.POINTER(GeneoParams), _I64, _P, _P, _P, ctypes.POINTER(GeneoInfo)]),
    "ddm_msgfem_basis": (_I32, [_P, _P, _P, _I64, _P, _P, _P, _P, ctypes.POINTER(GeneoParams), _I64, _P, _P, _P, ctypes.POINTER(GeneoInfo)]),
    "ddm_svd_basis": (_I32, [_P, _P, _I64, _P, _P, _P, _P, _I32, _I32, _D, _I32, _P, _P, ctypes.POINTER(GeneoInfo)]),
    "ddm_harmonic_create": (_I32, [_P, _P, _I64, _P, _I64, _P, _I64, _P, _PP]),
    "ddm_harmonic_destroy": (None, [_P]),
    "ddm_harmonic_extend": (_I32, [_P, _P, _I32, _P, _I64]),
    "ddm_blockvec_gram": (_I32, [_P, _I64, _P, _P, _I64, _I32, _P, _I64, _I32, _P]),
    "ddm_blockvec_gram2_sym": (_I32, [_P, _I64, _P, _P, _I64, _P, _P, _I64, _I32, _P, _P]),
    "ddm_blockvec_rotate": (_I32, [_P, _I64, _P, _P, _I64, _I32, _P, _I32, _P, _I64, _P, _I64]),
    "ddm_dense_sym_eig_host": (_I32, [_I32, _P, _P]),
    "ddm_dense_rayleigh_ritz_host": (_I32, [_I32, _P, _P, _I32, _D, _P, _P]),
    "ddm_combined_create": (_I32, [_P, _I32, _P, _P, _P, _PP]),
    "ddm_combined_destroy": (None, [_P]),
    "ddm_combined_apply": (_I32, [_P, _P, _P, _P]),
    "ddm_cg_solve": (_I32, [_P, _P, _P, _P, _P, _D, _I32, _I32, _P, ctypes.POINTER(SolveResult)]),
    "ddm_gmres_solve": (_I32, [_P, _P, _P, _P, _P, _D, _I32, _I32, _P, ctypes.POINTER(SolveResult)]),
    "ddm_bicgstab_solve": (_I32, [_P, _P, _P, _P, _P, _D, _I32, _P, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(SolveResult)]),
    "ddm_cg_begin": (_I32, [_P, _P, _P, _P, _P, _PP]),
    "ddm_cg_steps": (_I32, [_P, _P, _I32]),
    "ddm_cg_defect": (_I32, [_P, _P, ctypes.POINTER(ctypes.c_double)]),
    "ddm_cg_def0": (_D, [_P]),
    "ddm_cg_end": (None, [_P, _P]),
    "ddm_timing_enable": (_I32, [_P, _I32]),
    "ddm_timing_get": (_I32, [_P, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]),
    "ddm_timing_reset": (_I32, [_P]),
    "ddm_synth_q1_matrix": (_I32, [_I32, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _P, _I32]),
}

_lib = None


def load_library():
    """Loads libddm_hip.so (built by __graft_entry__.build()).  Fails loudly if it is missing."""
    global _lib
    if _lib is None:
        try:
            # torch ships its own HIP runtime: it must be in the process BEFORE libddm_hip.so is opened, so that the library binds to
            # the same libamdhip64 (the streams torch hands over belong to that runtime; with the opposite order ddm_ctx_create fails
            # with "no HIP device" on a GPU box -- seen with build() followed by smoke() in one process)
            import torch  # noqa: F401
        except ImportError:
            pass
        path = os.environ.get("DDM_HIP_LIBRARY", LIB_PATH)   # (diagnostic: an experimental build of the same sources)
        if not os.path.exists(path):
            raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback for the HIP hot path)")
        L = ctypes.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)            # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _np(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _hp(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class Context:
    """ddm_ctx: one per process / GPU.  ``stream``: raw hipStream_t (e.g. torch's current stream)."""

    def __init__(self, device=0, stream=None):
        self.lib = load_library()
        h = ctypes.c_void_p()
        rc = self.lib.ddm_ctx_create(int(device), ctypes.c_void_p(stream) if stream else None, ctypes.byref(h))
        if rc != DDM_OK:
            raise DdmError(rc, "ddm_ctx_create failed (no HIP device? the hot path has no CPU fallback)")
        self.h = h
        self._keep = []
        self.rank, self.nranks = 0, 1

    def check(self, rc):
        if rc != DDM_OK:
            raise DdmError(rc, self.lib.ddm_last_error(self.h).decode())

    def sync(self):
        self.check(self.lib.ddm_ctx_sync(self.h))

    def set_comm(self, rank, nranks, alltoall, allreduce):
        """alltoall(tag, send_ptr, recv_ptr) -> int ; allreduce(ptr, n) -> int (device pointers)."""
        a = A2A_FN(lambda user, tag, s, r: int(alltoall(tag, s, r)))
        b = ALLREDUCE_FN(lambda user, p, n: int(allreduce(p, n)))
        self._keep += [a, b]
        self.check(self.lib.ddm_ctx_set_comm(self.h, rank, nranks, a, b, None))
        self.rank, self.nranks = rank, nranks

    def rccl_unique_id(self):
        """128 bytes identifying a new RCCL communicator (call on ONE rank, distribute to all)"""
        buf = ctypes.create_string_buffer(128)
        rc = self.lib.ddm_rccl_unique_id(buf)
        if rc != DDM_OK:
            raise DdmError(rc, "ddm_rccl_unique_id failed (librccl not loadable?)")
        return buf.raw

    def set_rccl(self, rank, nranks, unique_id, self_test=False):
        """in-library exchange over RCCL / xGMI (collective over all ranks: ncclCommInitRank)"""
        assert len(unique_id) == 128
        self.check(self.lib.ddm_ctx_set_rccl(self.h, int(rank), int(nranks), ctypes.c_char_p(unique_id), int(bool(self_test))))
        self.rank, self.nranks = rank, nranks

    def comm_counts(self):
        """(all-reduces, doubles carried, grouped halo exchanges) issued so far -- as a multi-rank run launches them."""
        c = np.zeros(3, dtype=np.int64)
        self.check(self.lib.ddm_ctx_comm_counts(self.h, _hp(c)))
        return [int(v) for v in c]

    def rccl_size(self):
        """ranks of the in-library communicator as RCCL reports them (ncclCommCount); 0 = no in-library exchange"""
        c = ctypes.c_int(0)
        self.check(self.lib.ddm_ctx_rccl_size(self.h, ctypes.byref(c)))
        return c.value

    def timing(self, on=True):
        self.check(self.lib.ddm_timing_enable(self.h, int(on)))

    def timer(self, name):
        ms, cnt = ctypes.c_double(), ctypes.c_int64()
        self.check(self.lib.ddm_timing_get(self.h, name.encode(), ctypes.byref(ms), ctypes.byref(cnt)))
        return ms.value, cnt.value

    def timing_reset(self):
        self.check(self.lib.ddm_timing_reset(self.h))

    def close(self):
        if self.h:
            self.lib.ddm_ctx_destroy(self.h)
            self.h = None


def torch_context(device=0):
    """Context bound to torch's current stream on ``device``.  If that is the legacy default stream
    (which can neither be captured into a HIP graph nor ordered against a non-blocking stream) a
    side stream is created and made current, so torch allocations / copies and the library's
    kernels are ordered on one stream."""
    import torch
    torch.cuda.set_device(device)
    s = torch.cuda.current_stream()
    if s.cuda_stream == 0:
        s = torch.cuda.Stream(device=device)
        torch.cuda.set_stream(s)
    ctx = Context(device, s.cuda_stream)
    ctx._torch_stream = s
    return ctx


def _ptr(t):
    """device pointer of a torch tensor / raw int"""
    return ctypes.c_void_p(t if isinstance(t, int) else t.data_ptr())


class CsrMatrix:
    """Flattened BCRSMatrix on the device (ddm_csr)."""

    def __init__(self, ctx: Context, M, host_only=False):
        """host_only: no device copy (inputs the library only reads on the host: A_neu, B_neu of geneo_basis)"""
        import scipy.sparse as sp
        M = sp.csr_matrix(M)
        if not M.has_sorted_indices:
            M = M.sorted_indices()
        self.ctx = ctx
        self.shape = M.shape
        rp, ci, va = _np(M.indptr, np.int64), _np(M.indices, np.int32), _np(M.data, np.float64)
        h = ctypes.c_void_p()
        create = ctx.lib.ddm_csr_create_host if host_only else ctx.lib.ddm_csr_create
        ctx.check(create(ctx.h, M.shape[0], M.shape[1], _hp(rp), _hp(ci), _hp(va), ctypes.byref(h)))
        self.h = h
        self.nnz = int(M.nnz)

    def mv(self, x, y):
        self.ctx.check(self.ctx.lib.ddm_csr_mv(self.ctx.h, self.h, _ptr(x), _ptr(y)))

    def usmv(self, alpha, x, y):
        self.ctx.check(self.ctx.lib.ddm_csr_usmv(self.ctx.h, self.h, float(alpha), _ptr(x), _ptr(y)))

    def mm(self, X, Y):
        """Y = A X for row-major (n, nrhs) device tensors"""
        assert X.shape == Y.shape and X.is_contiguous() and Y.is_contiguous() and X.shape[0] == self.shape[1]
        self.ctx.check(self.ctx.lib.ddm_csr_mm(self.ctx.h, self.h, int(X.shape[1]), _ptr(X), _ptr(Y)))

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.ctx.lib.ddm_csr_destroy(self.h)
        except Exception:
            pass


def chol_host(M, block_ptr=None, numeric=True, general=False):
    """The host part of the sparse direct solver alone (ordering, symbolic analysis, numeric Cholesky; no device needed).
    Returns dict(perm, rowptr, col, lu, nnzL, flops); lu follows the ILU(0) storage convention over the permuted indices."""
    import scipy.sparse as sp
    lib = load_library()
    M = sp.csr_matrix(M)
    if not M.has_sorted_indices:
        M = M.sorted_indices()
    n = M.shape[0]
    bp = _np([0, n] if block_ptr is None else block_ptr, np.int64)
    rp, ci, va = _np(M.indptr, np.int64), _np(M.indices, np.int32), _np(M.data, np.float64)
    h = ctypes.c_void_p()
    rc = lib.ddm_direct_host_create(n, _hp(rp), _hp(ci), _hp(va) if numeric else None, len(bp) - 1, _hp(bp), int(general), ctypes.byref(h))
    if rc != DDM_OK:
        raise DdmError(rc, "sparse Cholesky failed (DDM_ENUMERIC: matrix not positive definite)")
    try:
        nz = lib.ddm_chol_host_nnz(h)
        out = {"perm": np.empty(n, dtype=np.int32), "rowptr": np.empty(n + 1, dtype=np.int64) if nz else None,
               "col": np.empty(nz, dtype=np.int32) if nz else None, "lu": np.empty(nz, dtype=np.float64) if nz else None,
               "nnzL": int(lib.ddm_chol_host_nnz_factor(h)), "flops": float(lib.ddm_chol_host_flops(h))}
        lib.ddm_chol_host_get(h, _hp(out["perm"]), _hp(out["rowptr"]), _hp(out["col"]), _hp(out["lu"]))
    finally:
        lib.ddm_chol_host_destroy(h)
    return out


def sn_symbolic_host(M, block_ptr=None):
    """Host half of the device supernodal Cholesky (ordering + symbolic analysis, no device needed): list of per-block dicts
    perm, first, rptr, rows, parent, level, flops, entries."""
    import scipy.sparse as sp
    lib = load_library()
    M = sp.csr_matrix(M)
    if not M.has_sorted_indices:
        M = M.sorted_indices()
    n = M.shape[0]
    bp = _np([0, n] if block_ptr is None else block_ptr, np.int64)
    rp, ci = _np(M.indptr, np.int64), _np(M.indices, np.int32)
    h = ctypes.c_void_p()
    rc = lib.ddm_sn_host_create(n, _hp(rp), _hp(ci), len(bp) - 1, _hp(bp), ctypes.byref(h))
    if rc != DDM_OK:
        raise DdmError(rc, "ddm_sn_host_create failed")
    out = []
    try:
        for b in range(len(bp) - 1):
            sz = np.zeros(4, dtype=np.int64)
            fl = ctypes.c_double()
            lib.ddm_sn_host_sizes(h, b, _hp(sz), ctypes.byref(fl))
            nsn, nrows = int(sz[0]), int(sz[1])
            d = {"perm": np.empty(int(bp[b + 1] - bp[b]), dtype=np.int32), "first": np.empty(nsn + 1, dtype=np.int32), "rptr": np.empty(nsn + 1, dtype=np.int64),
                 "rows": np.empty(nrows, dtype=np.int32), "parent": np.empty(nsn, dtype=np.int32), "level": np.empty(nsn, dtype=np.int32),
                 "flops": fl.value, "entries": int(sz[2]), "levels": int(sz[3])}
            lib.ddm_sn_host_get(h, b, _hp(d["perm"]), _hp(d["first"]), _hp(d["rptr"]), _hp(d["rows"]), _hp(d["parent"]), _hp(d["level"]))
            out.append(d)
    finally:
        lib.ddm_sn_host_destroy(h)
    return out


class Ilu0:
    """ddm_ilu0: a local factor solver -- ILU(0) in natural order, or (direct=True) the sparse Cholesky of ddm_chol_create."""

    def __init__(self, ctx: Context, A: CsrMatrix, block_ptr=None, direct=False, max_flops=0.0, general=False):
        self.ctx, self.A = ctx, A
        bp = _np([0, A.shape[0]] if block_ptr is None else block_ptr, np.int64)
        h = ctypes.c_void_p()
        if direct:
            ctx.check(ctx.lib.ddm_direct_create(ctx.h, A.h, len(bp) - 1, _hp(bp), int(general), float(max_flops), ctypes.byref(h)))
        else:
            ctx.check(ctx.lib.ddm_ilu0_create(ctx.h, A.h, len(bp) - 1, _hp(bp), ctypes.byref(h)))
        self.h = h

    @property
    def nnz(self):
        return int(self.ctx.lib.ddm_ilu0_nnz(self.h))

    def refinement(self):
        """(steps per solve, backward errors of the probe after 0, 1, .. steps) of a device direct factor."""
        om = np.zeros(5)
        steps = int(self.ctx.lib.ddm_ilu0_refinement(self.h, _hp(om)))
        return steps, om

    def solve(self, d, x):
        self.ctx.check(self.ctx.lib.ddm_ilu0_solve(self.ctx.h, self.h, _ptr(d), _ptr(x)))

    def solve_multi(self, D, X, single_precision=False):
        """X = (LU)^-1 D for row-major (n, nrhs) device tensors; single_precision: float sweeps (preconditioner grade)"""
        assert D.shape == X.shape and D.is_contiguous() and X.is_contiguous()
        fn = self.ctx.lib.ddm_ilu0_solve_multi_f32 if single_precision else self.ctx.lib.ddm_ilu0_solve_multi
        self.ctx.check(fn(self.ctx.h, self.h, int(D.shape[1]), _ptr(D), _ptr(X)))

    def num_levels(self, upper=False):
        return int(self.ctx.lib.ddm_ilu0_num_levels(self.h, int(upper)))

    def box_check(self):
        """stamps of the box engine's last solve (DDM_BOX_CHECK=1 at creation): [sweep][plane of block 0] = (start, end in 10 ns ticks,
        polls of the previous plane's progress word, XCC id)"""
        out = np.zeros(1024, dtype=np.uint64)
        self.ctx.lib.ddm_ilu0_box_check(self.h, _hp(out))
        return out.reshape(2, 128, 4)

    def wait(self):
        """joins the part of the setup that runs in the background (ddm_ilu0_wait)"""
        self.ctx.check(self.ctx.lib.ddm_ilu0_wait(self.ctx.h, self.h))

    def debug_stamps(self, d, x):
        out = np.zeros(8, dtype=np.uint64)
        self.ctx.check(self.ctx.lib.ddm_ilu0_debug_stamps(self.ctx.h, self.h, _ptr(d), _ptr(x), _hp(out)))
        return out

    def pipe_trace(self, d, x):
        """diagnostic: (stamps[ntasks, 272] uint64, meta[ntasks, 2] int32 = group, sweep) of one pipe-engine solve"""
        nt = ctypes.c_int64()
        self.ctx.check(self.ctx.lib.ddm_ilu0_pipe_trace(self.ctx.h, self.h, None, None, None, None, 0, ctypes.byref(nt)))
        out = np.zeros((nt.value, 272), dtype=np.uint64)
        meta = np.zeros((nt.value, 2), dtype=np.int32)
        self.ctx.check(self.ctx.lib.ddm_ilu0_pipe_trace(self.ctx.h, self.h, _ptr(d), _ptr(x), _hp(out), _hp(meta), nt.value, ctypes.byref(nt)))
        return out, meta

    def engine(self):
        return SchwarzPreconditioner.ENGINES[int(self.ctx.lib.ddm_ilu0_engine(self.h))]

    def status(self):
        st = ctypes.c_int()
        self.ctx.check(self.ctx.lib.ddm_ilu0_status(self.ctx.h, self.h, ctypes.byref(st)))
        return st.value

    def factors(self):
        out = np.empty(self.A.nnz, dtype=np.float64)
        self.ctx.check(self.ctx.lib.ddm_ilu0_get_factors_host(self.ctx.h, self.h, _hp(out)))
        return out

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.ctx.lib.ddm_ilu0_destroy(self.h)
        except Exception:
            pass


class Halo:
    """One DUNE interface flattened into a pack / exchange / unpack plan (ddm_halo)."""
    COPY, ADD = 0, 1

    def __init__(self, ctx: Context, tag, mode, plan):
        self.ctx, self.tag, self.mode, self.plan = ctx, tag, mode, plan
        a = {k: _np(plan[k], np.int64) for k in ("send_idx", "send_counts", "recv_counts", "dst_idx", "dst_ptr", "src_pos")}
        assert len(a["send_counts"]) == ctx.nranks and len(a["recv_counts"]) == ctx.nranks
        h = ctypes.c_void_p()
        ctx.check(ctx.lib.ddm_halo_create(ctx.h, tag, mode, len(a["send_idx"]), _hp(a["send_idx"]), _hp(a["send_counts"]),
                                          _hp(a["recv_counts"]), len(a["dst_idx"]), _hp(a["dst_idx"]), _hp(a["dst_ptr"]),
                                          _hp(a["src_pos"]), ctypes.byref(h)))
        self.h = h
        self.send_counts = [int(c) for c in a["send_counts"]]
        self.recv_counts = [int(c) for c in a["recv_counts"]]

    def exchange(self, v):
        self.ctx.check(self.ctx.lib.ddm_halo_exchange(self.ctx.h, self.h, _ptr(v)))

    def exchange_to(self, src, dst):
        """pack from ``src``, unpack into ``dst`` (entries outside the destination list are left alone)"""
        self.ctx.check(self.ctx.lib.ddm_halo_exchange_to(self.ctx.h, self.h, _ptr(src), _ptr(dst)))

    @property
    def sendbuf(self):
        return self.ctx.lib.ddm_halo_sendbuf(self.h)

    @property
    def recvbuf(self):
        return self.ctx.lib.ddm_halo_recvbuf(self.h)


class NonOverlappingOperator:
    """dune/ddm/nonoverlapping_operator.hh:11-58 (+ the scalar product :63-89)."""

    def __init__(self, ctx: Context, A: CsrMatrix, novlp_add: Halo | None, owner_mask):
        self.ctx, self.A, self.halo = ctx, A, novlp_add
        m = _np(owner_mask, np.uint8)
        assert len(m) == A.shape[0]
        h = ctypes.c_void_p()
        ctx.check(ctx.lib.ddm_op_create(ctx.h, A.h, novlp_add.h if novlp_add else None, _hp(m), ctypes.byref(h)))
        self.h = h

    def apply(self, x, y):
        self.ctx.check(self.ctx.lib.ddm_op_apply(self.ctx.h, self.h, _ptr(x), _ptr(y)))

    def applyscaleadd(self, alpha, x, y):
        self.ctx.check(self.ctx.lib.ddm_op_applyscaleadd(self.ctx.h, self.h, float(alpha), _ptr(x), _ptr(y)))

    def dot(self, x, y):
        r = ctypes.c_double()
        self.ctx.check(self.ctx.lib.ddm_dot(self.ctx.h, self.h, _ptr(x), _ptr(y), ctypes.byref(r)))
        return r.value

    def norm(self, x):
        r = ctypes.c_double()
        self.ctx.check(self.ctx.lib.ddm_norm(self.ctx.h, self.h, _ptr(x), ctypes.byref(r)))
        return r.value


class SchwarzPreconditioner:
    """dune/ddm/schwarz.hh:54-220 with the ILU(0) local solver on the device."""
    TYPES = {"standard": 0, "restricted": 1}

    def __init__(self, ctx: Context, A_dir: CsrMatrix, block_ptr, n_novlp, ext_map, pou, type, ovlp_copy: Halo | None,
                 ovlp_add: Halo | None, subdomain_solver="ilu0"):
        if type not in self.TYPES:
            raise NotImplementedError("Unknown Schwarz type '" + str(type) + "'")   # schwarz.hh:83
        self.ctx = ctx
        bp = _np(block_ptr, np.int64)
        em = _np(ext_map, np.int32)
        pw = None if pou is None else _np(pou, np.float64)
        h = ctypes.c_void_p()
        ctx.check(ctx.lib.ddm_schwarz_create_ex(ctx.h, A_dir.h, len(bp) - 1, _hp(bp), int(n_novlp), _hp(em), _hp(pw), self.TYPES[type],
                                                str(subdomain_solver).encode(), ovlp_copy.h if ovlp_copy else None,
                                                ovlp_add.h if ovlp_add else None, ctypes.byref(h)))
        self.h = h
        self._keep = (A_dir, ovlp_copy, ovlp_add)

    def apply(self, x, d):
        self.ctx.check(self.ctx.lib.ddm_schwarz_apply(self.ctx.h, self.h, _ptr(x), _ptr(d)))

    def num_levels(self):
        return (int(self.ctx.lib.ddm_schwarz_num_levels(self.h, 0)), int(self.ctx.lib.ddm_schwarz_num_levels(self.h, 1)))

    ENGINES = {8: "pipe", 4: "xcd2", 0: "levels", 16: "supernodal", 32: "box"}

    def engine(self):
        """triangular-solve engine of the local solver: 'box' (structured blocks: csrc/trsv_box.hpp), 'pipe', 'xcd2' (also when pipe declined
        the matrix), 'levels', or 'supernodal'
        (sparse direct factor of the device engine: dense panels, csrc/sn_chol.hpp)"""
        return self.ENGINES[int(self.ctx.lib.ddm_schwarz_engine(self.h))]

    def wait_setup(self):
        """joins the local solver's background setup (the pipe engine's schedule is built while the caller goes on, e.g. into the
        GenEO eigensolver): ddm_ilu0_wait; the first apply would wait for it otherwise"""
        F = self.ctx.lib.ddm_schwarz_local_solver(self.h)
        if F:
            self.ctx.check(self.ctx.lib.ddm_ilu0_wait(self.ctx.h, ctypes.c_void_p(F)))

    def factor_nnz(self):
        """stored entries of the local solver's factor (roofline accounting)"""
        return int(self.ctx.lib.ddm_schwarz_factor_nnz(self.h))

    def check_status(self):
        """raises DdmError if a single-launch local solve timed out since creation (synchronous)"""
        self.ctx.check(self.ctx.lib.ddm_schwarz_status(self.ctx.h, self.h))


class GalerkinPreconditioner:
    """dune/ddm/galerkin_preconditioner.hh:40-363 (apply path; the coarse matrix is assembled by
    ``coarse.build_coarse_matrix`` and handed over as its replicated inverse)."""

    def __init__(self, ctx: Context, n, n_novlp, ext_map, sub_ptr, basis, coarse_index, a0inv, ovlp_copy, ovlp_add):
        self.ctx = ctx
        basis = _np(basis, np.float64)
        kmax = basis.shape[0]
        assert basis.shape[1] == n
        sp_ = _np(sub_ptr, np.int64)
        ci = _np(coarse_index, np.int64)
        inv = _np(a0inv, np.float64)
        K = inv.shape[0]
        em = _np(ext_map, np.int32)
        h = ctypes.c_void_p()
        ctx.check(ctx.lib.ddm_galerkin_create(ctx.h, int(n), int(n_novlp), _hp(em), len(sp_) - 1, _hp(sp_), kmax, _hp(basis), _hp(ci),
                                              K, _hp(inv), ovlp_copy.h if ovlp_copy else None, ovlp_add.h if ovlp_add else None,
                                              ctypes.byref(h)))
        self.h = h
        self._keep = (ovlp_copy, ovlp_add)

    def apply(self, x, d):
        self.ctx.check(self.ctx.lib.ddm_galerkin_apply(self.ctx.h, self.h, _ptr(x), _ptr(d)))


def galerkin_products(ctx: Context, A_dir: CsrMatrix, left, right, row0, row1):
    """out[i, j] = <left_i, A_dir right_j> over rows [row0, row1); left/right: torch (k x n) device tensors."""
    nl, nr = left.shape[0], right.shape[0]
    out = np.empty((nr, nl), dtype=np.float64)      # column-major nl x nr
    ctx.check(ctx.lib.ddm_galerkin_products(ctx.h, A_dir.h, nl, _ptr(left), nr, _ptr(right), int(row0), int(row1), _hp(out)))
    return out.T


class CombinedPreconditioner:
    """dune/ddm/combined_preconditioner.hh:39-180."""
    MODES = {"additive": 0, "multiplicative": 1}

    def __init__(self, ctx: Context, mode="additive", op=None, schwarz=None, galerkin=None):
        if mode not in self.MODES:
            raise NotImplementedError("Unknown apply mode in CombinedPreconditioner, use either additive or multiplicative")
        self.ctx = ctx
        h = ctypes.c_void_p()
        ctx.check(ctx.lib.ddm_combined_create(ctx.h, self.MODES[mode], op.h if op else None, schwarz.h if schwarz else None,
                                              galerkin.h if galerkin else None, ctypes.byref(h)))
        self.h = h
        self._keep = (op, schwarz, galerkin)

    def apply(self, x, d):
        self.ctx.check(self.ctx.lib.ddm_combined_apply(self.ctx.h, self.h, _ptr(x), _ptr(d)))

    def check_status(self):
        self.ctx.check(self.ctx.lib.ddm_combined_status(self.ctx.h, self.h))


def cg_solve(ctx: Context, op: NonOverlappingOperator, prec: CombinedPreconditioner, x, b, reduction=1e-10, maxit=1000,
             fixed_iterations=0, history=True):
    """dune-istl CGSolver::apply as driven by examples/poisson.cc:311-319.  x, b: device tensors;
    b is overwritten by the defect.  Returns (SolveResult, history ndarray or None)."""
    res = SolveResult()
    hist = np.zeros(max(maxit, fixed_iterations) + 1, dtype=np.float64) if history else None
    ctx.check(ctx.lib.ddm_cg_solve(ctx.h, op.h, prec.h, _ptr(x), _ptr(b), float(reduction), int(maxit), int(fixed_iterations),
                                   _hp(hist), ctypes.byref(res)))
    return res, (hist[:res.iterations + 1] if history else None)


def gmres_solve(ctx: Context, op: NonOverlappingOperator, prec: CombinedPreconditioner, x, b, reduction=1e-10, maxit=1000, restart=100,
                history=True):
    """dune-istl RestartedGMResSolver::apply ([solver] type = restartedgmressolver, examples/poisson.ini:12-17)."""
    res = SolveResult()
    hist = np.zeros(maxit + 1, dtype=np.float64) if history else None
    ctx.check(ctx.lib.ddm_gmres_solve(ctx.h, op.h, prec.h, _ptr(x), _ptr(b), float(reduction), int(maxit), int(restart), _hp(hist),
                                      ctypes.byref(res)))
    return res, (hist[:res.iterations + 1] if history else None)


def bicgstab_solve(ctx: Context, op: NonOverlappingOperator, prec: CombinedPreconditioner, x, b, reduction=1e-10, maxit=1000, history=True):
    """dune-istl BiCGSTABSolver::apply ([solver] type = bicgstabsolver); history: one entry per HALF step"""
    res = SolveResult()
    hist = np.zeros(2 * maxit + 2, dtype=np.float64)
    nh = ctypes.c_int32(0)
    ctx.check(ctx.lib.ddm_bicgstab_solve(ctx.h, op.h, prec.h, _ptr(x), _ptr(b), float(reduction), int(maxit), _hp(hist), ctypes.byref(nh), ctypes.byref(res)))
    return res, (hist[:nh.value] if history else None)


class CgIteration:
    """ddm_cg_begin / steps / defect / end: the CG loop in pieces (exact-K timing in bench.py)."""

    def __init__(self, ctx: Context, op: NonOverlappingOperator, prec: CombinedPreconditioner, x, b):
        self.ctx = ctx
        h = ctypes.c_void_p()
        ctx.check(ctx.lib.ddm_cg_begin(ctx.h, op.h, prec.h, _ptr(x), _ptr(b), ctypes.byref(h)))
        self.h = h
        self._keep = (op, prec, x, b)
        self.def0 = ctx.lib.ddm_cg_def0(h)

    def steps(self, k):
        self.ctx.check(self.ctx.lib.ddm_cg_steps(self.ctx.h, self.h, int(k)))

    def defect(self):
        d = ctypes.c_double()
        self.ctx.check(self.ctx.lib.ddm_cg_defect(self.ctx.h, self.h, ctypes.byref(d)))
        return d.value

    def end(self):
        if self.h:
            self.ctx.lib.ddm_cg_end(self.ctx.h, self.h)
            self.h = None


class HarmonicExtension:
    """ddm_harmonic: EnergyMinimalExtension (dune/ddm/coarsespaces/energy_minimal_extension.hh:36-229) on the device."""

    def __init__(self, ctx: Context, A: "CsrMatrix", interior, boundary, block_ptr=None):
        self.ctx = ctx
        ii, bb = _np(interior, np.int64), _np(boundary, np.int64)
        bp = _np(block_ptr if block_ptr is not None else [0, A.shape[0]], np.int64)
        h = ctypes.c_void_p()
        ctx.check(ctx.lib.ddm_harmonic_create(ctx.h, A.h, len(bp) - 1, _hp(bp), len(ii), _hp(ii), len(bb), _hp(bb), ctypes.byref(h)))
        self.h = h

    def extend(self, X):
        """in place on a row-major device tensor (n, nrhs) holding the boundary values: interior rows are overwritten"""
        assert X.dim() == 2 and X.stride(1) == 1
        self.ctx.check(self.ctx.lib.ddm_harmonic_extend(self.ctx.h, self.h, X.shape[1], _ptr(X), X.stride(0)))
        return X

    def close(self):
        if self.h:
            self.ctx.lib.ddm_harmonic_destroy(self.h)
            self.h = None

    __del__ = close


def row_order_tiled_host(block_ptr, A):
    """(found, order): the cache-blocked row order of the block products for a scipy CSR matrix (host only, no device needed)"""
    lib = load_library()
    bp = _np(block_ptr, np.int64)
    rp = _np(A.indptr, np.int64)
    ci = _np(A.indices, np.int32)
    order = np.empty(A.shape[0], dtype=np.int32)
    rc = lib.ddm_csr_row_order_tiled_host(len(bp) - 1, _hp(bp), _hp(rp), _hp(ci), _hp(order))
    if rc < 0:
        raise ValueError("ddm_csr_row_order_tiled_host: bad arguments")
    return bool(rc), order


def blockvec_gram(ctx: Context, sub_ptr, U, V):
    """per-subdomain U^T V of row-major device tensors (n, pu), (n, pv) -> ndarray (nsub, pu, pv)"""
    bp = _np(sub_ptr, np.int64)
    assert U.is_contiguous() and V.is_contiguous() and U.shape[0] == V.shape[0] == bp[-1]
    out = np.empty((len(bp) - 1, U.shape[1], V.shape[1]), dtype=np.float64)
    ctx.check(ctx.lib.ddm_blockvec_gram(ctx.h, len(bp) - 1, _hp(bp), _ptr(U), U.stride(0), U.shape[1], _ptr(V), V.stride(0), V.shape[1], _hp(out)))
    return out


def blockvec_gram2_sym(ctx: Context, sub_ptr, U, V1, V2):
    """per-subdomain U^T V1, U^T V2 for symmetric products (upper tiles computed, mirrored) -> two ndarrays (nsub, p, p)"""
    bp = _np(sub_ptr, np.int64)
    p = U.shape[1]
    assert V1.shape == U.shape == V2.shape and V1.stride(0) == V2.stride(0) and U.shape[0] == bp[-1]
    g1 = np.empty((len(bp) - 1, p, p), dtype=np.float64)
    g2 = np.empty_like(g1)
    ctx.check(ctx.lib.ddm_blockvec_gram2_sym(ctx.h, len(bp) - 1, _hp(bp), _ptr(U), U.stride(0), _ptr(V1), _ptr(V2), V1.stride(0), p, _hp(g1), _hp(g2)))
    return g1, g2


def blockvec_rotate(ctx: Context, sub_ptr, U, Y, out, base=None):
    """out[rows of s] = (base[rows of s] -) U[rows of s] @ Y[s]; U (n, p), Y ndarray (nsub, p, q), out (n, >= q) device tensors"""
    bp = _np(sub_ptr, np.int64)
    Yh = _np(Y, np.float64)
    ctx.check(ctx.lib.ddm_blockvec_rotate(ctx.h, len(bp) - 1, _hp(bp), _ptr(U), U.stride(0), U.shape[1], _hp(Yh), Yh.shape[2],
                                          _ptr(base) if base is not None else None, base.stride(0) if base is not None else 0, _ptr(out), out.stride(0)))
