"""Diagnostic: per-iteration deviation of the HIP CG residual history from the oracle's with a GenEO coarse space (same basis),
and the conditioning of the coarse matrix.  usage: python tools/geneo_parity_probe.py N [nev]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.import_package()
from dune_ddm_amd import synth  # noqa: E402
from dune_ddm_amd.geneo import geneo_basis  # noqa: E402
from dune_ddm_amd.problem import build_structured  # noqa: E402
from dune_ddm_amd.solver import TwoLevelSchwarz  # noqa: E402
from tests.oracle_bridge import oracle_solve  # noqa: E402

N = int(sys.argv[1])
nev = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dec = build_structured(synth.StructuredPoisson((N, N, N), (2, 2, 2)), overlap=2, pou_type="distance", neumann=True)
tl = TwoLevelSchwarz(dec, coarse="none")
basis = geneo_basis(tl, nev=nev)
tl.set_coarse_basis(basis)
tl.rebuild_combined("additive")
print("K =", tl.K, "cond(A0) = %.3e" % np.linalg.cond(tl.a0), "||A0 - A0^T||/||A0|| = %.2e" % (np.abs(tl.a0 - tl.a0.T).max() / np.abs(tl.a0).max()))
res, hist, x = tl.solve(reduction=1e-10, maxit=1000)
it, conv, ho, _ = oracle_solve(dec, reduction=1e-10, maxit=1000, coarse={s: list(basis[s]) for s in basis}, schwarz_type="standard", mode="additive")
ho = np.array(ho)
m = min(len(hist), len(ho))
dev = np.abs(np.asarray(hist)[:m] - ho[:m]) / ho[:m]
print("iterations gpu/oracle", res.iterations, it)
for k in (1, 2, 5, 10, 20, 40, 60, 100, 150, 200, m - 1):
    if k < m:
        print(f"  k={k:4d}  ||r_k||/||r_0|| = {ho[k] / ho[0]:.3e}   rel dev = {dev[k]:.2e}")
tl.ctx.close()
