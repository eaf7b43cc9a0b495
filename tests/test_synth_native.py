"""The host-side generator of the benchmark's input matrices (ddm_synth_q1_matrix, csrc/synth_host.hpp) against the numpy passes of
dune_ddm_amd/synth.py it replaces by default: bit for bit (pattern, column order, values, dtypes) on 2-D and 3-D boxes, non-integer
coefficients, overlap regions, thin boxes.  CPU only: the function is host code of libddm_hip.so."""
import numpy as np
import pytest

import __graft_entry__ as ge

ge.import_package()
from dune_ddm_amd import synth  # noqa: E402
from dune_ddm_amd.problem import build_structured  # noqa: E402


def _same(A, B):
    if A is None or B is None:
        return A is None and B is None
    A, B = A.tocsr(), B.tocsr()
    return (A.shape == B.shape and A.indices.dtype == B.indices.dtype and np.array_equal(A.indptr, B.indptr)
            and np.array_equal(A.indices, B.indices) and np.array_equal(A.data, B.data))


@pytest.mark.parametrize("N,P,overlap", [((9, 8, 7), (2, 2, 1), 1), ((12, 12, 12), (2, 2, 2), 2), ((17, 13), (3, 2), 1),
                                         ((3, 3, 3), (1, 1, 1), 1), ((2, 5, 4), (1, 2, 2), 1), ((24, 10), (4, 1), 3)])
def test_native_generator_is_bitwise_the_numpy_one(N, P, overlap, monkeypatch):
    rng = np.random.default_rng(len(N) * 100 + N[0])
    kappa = rng.random(tuple(n - 1 for n in N)[::-1]) * 3 + 0.1
    dec = {}
    for native in ("0", "1"):
        monkeypatch.setenv("DDM_SYNTH_NATIVE", native)
        dec[native] = build_structured(synth.StructuredPoisson(N, P, kappa), overlap=overlap, neumann=True)
    for a, b in zip(dec["0"].subs, dec["1"].subs):
        for name in ("A", "A_dir", "A_neu", "B_neu"):
            assert _same(getattr(a, name, None), getattr(b, name, None)), (a.id, name)
        assert np.array_equal(a.pou, b.pou)


def test_native_generator_rejects_bad_arguments():
    lib = ge.import_package().load_library()
    assert lib.ddm_synth_q1_matrix(4, None, None, None, None, None, None, None, 0, None, None, None, None, None, None, 1) != 0
    assert b"ddm_synth_q1_matrix" in lib.ddm_last_error(None)
