"""Evidence for the late-iteration tolerance of the CG parity tests ("what's weak" 1).

Two FP64 implementations of the SAME preconditioned CG that differ only in the order of the additions inside the global dot
products are run against each other -- oracle vs oracle, no GPU involved: the reference's order (ascending index, ascending rank),
descending order, and pairwise summation (oracle/kernels.c: orc_masked_dot_order).  Every other operation is bit-identical.
The residual histories agree to rounding level while the residual is large and then drift apart by many orders of magnitude
(CG amplifies the perturbation of its scalars roughly like ||r_0|| / ||r_k||), while iteration counts and SOLUTIONS keep agreeing.
The HIP path differs from the oracle in exactly this way (wavefront-wise / blocked sums instead of index-by-index sums), so its
late-iteration deviation from the oracle must be judged against this envelope, not against 1e-8: tests/test_gpu_fullsize.py does
that at 96^3 with the GenEO coarse space, bench.py reports both deviations side by side at 216^3.
"""
import numpy as np
import pytest

from oracle import apply_oracle as ao
from tests.oracle_bridge import oracle_solve


def order_histories(dec, orders=(0, 1, 2), **kw):
    out = {}
    try:
        for o in orders:
            ao.set_dot_order(o)
            it, conv, hist, x = oracle_solve(dec, **kw)
            out[o] = (it, conv, np.asarray(hist, dtype=float), np.concatenate(x))
    finally:
        ao.set_dot_order(0)
    return out


def envelope(h_ref, h_other):
    """running maximum over the iterations of | ||r_k||' - ||r_k|| | / ||r_k||"""
    m = min(len(h_ref), len(h_other))
    return np.maximum.accumulate(np.abs(h_other[:m] - h_ref[:m]) / h_ref[:m])


def test_dot_order_variants_are_the_same_sum():
    rng = np.random.default_rng(7)
    n = 100_003
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    owner = (rng.random(n) < 0.9).astype(np.uint8)
    L = ao.lib()
    ref = L.orc_masked_dot(n, ao._p(owner), ao._p(x), ao._p(y))
    exact = float(np.sum((x * y)[owner > 0].astype(np.longdouble)))
    for o in (0, 1, 2):
        v = L.orc_masked_dot_order(n, ao._p(owner), ao._p(x), ao._p(y), o)
        assert abs(v - exact) <= 1e-12 * np.sum(np.abs(x * y))
        if o == 0:
            assert v == ref
    assert L.orc_masked_dot_order(n, ao._p(owner), ao._p(x), ao._p(y), 1) != ref or L.orc_masked_dot_order(n, ao._p(owner), ao._p(x), ao._p(y), 2) != ref


def test_cg_histories_of_two_summation_orders_drift_apart(ddm):
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import build_structured
    dec = build_structured(synth.StructuredPoisson((40, 40, 40), (2, 2, 2)), overlap=2, pou_type="distance")
    H = order_histories(dec, reduction=1e-10, maxit=500, coarse="pou", schwarz_type="standard", mode="additive")
    it0, conv0, h0, x0 = H[0]
    assert conv0 and it0 >= 60
    for o in (1, 2):
        it, conv, h, x = H[o]
        assert conv and abs(it - it0) <= 1
        m = min(len(h), len(h0))
        dev = np.abs(h[:m] - h0[:m]) / h0[:m]
        early = h0[:m] >= 2e-3 * h0[0]
        # while the residual is large the two runs agree far below the 1e-8 of the parity tests ...
        assert early.sum() >= 20 and dev[early].max() <= 1e-11, dev[early].max()
        # ... and then drift apart by MANY orders of magnitude although nothing but the order of the additions in the dots differs
        assert dev.max() >= 1e-5, dev.max()
        assert dev.max() <= 0.5                                  # still the same curve on a log plot
        assert dev[:m // 2].max() < dev[m // 2:].max()
        # the solutions agree to 1e-10 all the same
        assert np.abs(x - x0).max() <= 1e-10 * np.abs(x0).max()
    e1, e2 = envelope(h0, H[1][2]), envelope(h0, H[2][2])
    k = min(len(e1), len(e2)) - 1
    print("k, r_k/r_0, envelope(reversed), envelope(pairwise):", [(j, f"{h0[j] / h0[0]:.1e}", f"{e1[j]:.1e}", f"{e2[j]:.1e}") for j in range(0, k + 1, 10)])
    # the amplification follows the decay of the residual: by the time ||r_k|| has dropped 1e-8 the drift is >= 1e4 times what it
    # was at a drop of 1e-2
    j2, j8 = int(np.argmax(h0 <= 1e-2 * h0[0])), int(np.argmax(h0 <= 1e-8 * h0[0]))
    assert max(e1[j8], e2[j8]) >= 1e4 * max(e1[j2], e2[j2], 1e-16)
