"""Host side of the "box" triangular-solve engine (dune-ddm_amd/csrc/trsv_box_host.hpp): structure detection, stream packing and the
shell system, walked on the CPU in the order of the device kernels and compared BIT FOR BIT with the oracle's sequential ILU(0) solve
(the same check tests/test_pipe_schedule.py makes for the pipe engine).  CPU only."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


@pytest.fixture(scope="module")
def harness():
    subprocess.check_call(["make", "-C", CPP, "libbox_host_test.so"], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(os.path.join(CPP, "libbox_host_test.so"))
    lib.box_test_build_and_emulate.restype = ctypes.c_int
    return lib


def run_box(lib, M, block_ptr, d):
    from oracle import apply_oracle as ao
    M = sp.csr_matrix(M)
    M.sort_indices()
    n = M.shape[0]
    lu = np.empty(M.nnz)
    diag = np.empty(n, dtype=np.int64)
    xo = np.zeros(n)
    for b in range(len(block_ptr) - 1):
        r0, r1 = block_ptr[b], block_ptr[b + 1]
        Mb = sp.csr_matrix(M[r0:r1, r0:r1])
        Mb.sort_indices()
        f = ao.Ilu0(ao.Csr(Mb))
        lu[M.indptr[r0]:M.indptr[r1]] = f.lu
        diag[r0:r1] = f.diag + M.indptr[r0]
        xb = np.zeros(r1 - r0)
        f.apply(xb, np.ascontiguousarray(d[r0:r1]))
        xo[r0:r1] = xb
    rp = np.asarray(M.indptr, dtype=np.int64)
    ci = np.asarray(M.indices, dtype=np.int32)
    bp = np.asarray(block_ptr, dtype=np.int64)
    x = np.full(n, np.nan)
    stats = np.zeros(10, dtype=np.int64)
    err = ctypes.create_string_buffer(256)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib.box_test_build_and_emulate(ctypes.c_int64(n), p(rp), p(ci), p(lu), p(diag), ctypes.c_int(len(bp) - 1), p(bp),
                                        p(np.ascontiguousarray(d)), p(x), p(stats), err, 256)
    names = ["box_rows", "shell_rows", "stream_bytes", "ext_products", "shell_lower_entries", "nx", "ny", "nz", "nsteps", "shell_factor_nnz"]
    return rc, err.value.decode(), x, xo, dict(zip(names, stats.tolist()))


def _subdomain_system(N, P, overlap, kappa=None):
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import build_structured
    grid = synth.StructuredPoisson(N, P, kappa)
    dec = build_structured(grid, overlap=overlap, pou_type="distance", shrink=0)
    mats = [sd.A_dir.tocsr() for sd in dec.subs]
    M = sp.block_diag(mats, format="csr")
    bp = np.concatenate([[0], np.cumsum([m.shape[0] for m in mats])])
    return M, bp, dec


@pytest.mark.parametrize("N,P,overlap", [((21, 19, 18), (2, 1, 2), 2), ((17, 17, 17), (2, 2, 2), 1), ((30, 14, 12), (3, 1, 1), 2),
                                         ((13, 12, 25), (1, 1, 3), 3)])
def test_structured_subdomains_bit_exact(ddm, harness, N, P, overlap):
    from dune_ddm_amd import synth
    kappa = synth.islands_kappa(tuple(n - 1 for n in N), 1e4, 4, 2)
    M, bp, dec = _subdomain_system(N, P, overlap, kappa)
    rng = np.random.default_rng(3)
    d = rng.standard_normal(M.shape[0])
    rc, err, x, xo, st = run_box(harness, M, bp, d)
    assert rc == 0, err
    assert np.array_equal(x, xo)                               # same order of operations as the sequential solve
    assert st["box_rows"] + st["shell_rows"] == M.shape[0] and st["shell_rows"] > 0
    sd = dec.subs[0]
    assert st["nx"] * st["ny"] * st["nz"] == sd.n_o            # the box is what the rank owns; the shell is the overlap


def test_plain_box_without_shell(ddm, harness):
    M, bp, _ = _subdomain_system((12, 11, 10), (1, 1, 1), 1)
    d = np.random.default_rng(0).standard_normal(M.shape[0])
    rc, err, x, xo, st = run_box(harness, M, bp, d)
    assert rc == 0, err
    assert np.array_equal(x, xo) and st["shell_rows"] == 0 and (st["nx"], st["ny"], st["nz"]) == (12, 11, 10)


def test_declines_what_is_not_a_box(ddm, harness):
    rng = np.random.default_rng(1)
    R = sp.random(500, 500, density=0.02, random_state=np.random.RandomState(5), format="csr")
    M = sp.csr_matrix(R + R.T + sp.eye(500) * 30.0)
    rc, err, *_ = run_box(harness, M, np.array([0, 500]), rng.standard_normal(500))
    assert rc == 1 and "block 0" in err
    # a 2-D problem (9-point pattern) is not a 3-D box either
    M2, bp2, _ = _subdomain_system((40, 30), (2, 1), 2)
    rc, err, *_ = run_box(harness, M2, bp2, rng.standard_normal(M2.shape[0]))
    assert rc == 1
    # a permuted box: the pattern check must notice
    M3, bp3, _ = _subdomain_system((9, 9, 9), (1, 1, 1), 1)
    perm = np.arange(M3.shape[0])
    perm[[100, 101]] = perm[[101, 100]]
    M3p = sp.csr_matrix(M3[perm][:, perm])
    rc, err, *_ = run_box(harness, M3p, bp3, rng.standard_normal(M3.shape[0]))
    assert rc == 1
