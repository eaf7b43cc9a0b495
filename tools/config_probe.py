"""Probe of BASELINE.json configs[3] / configs[4] on the device: engine the local solve uses (DDM_PIPE_VERBOSE=1 prints why pipe
declines), ms per local solve, device GenEO, outer GMRES.  usage: python tools/config_probe.py dg|elasticity SIZE [nev]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ddm = ge.import_package()
import torch  # noqa: E402
from dune_ddm_amd import synth  # noqa: E402
from dune_ddm_amd.geneo import geneo_basis  # noqa: E402
from dune_ddm_amd.problem import build_structured  # noqa: E402
from dune_ddm_amd.solver import TwoLevelSchwarz  # noqa: E402

which, size = sys.argv[1], int(sys.argv[2])
local = os.environ.get("DDM_LOCAL_SOLVER", "ilu0")     # ilu0 | cholmod | umfpack | direct
t0 = time.time()
if which == "dg":
    nev = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    dec = build_structured(synth.StructuredDG2D((size, size), (4, 2)), overlap=2, neumann=True)
    cfg = dict(schwarz_type="standard", mode="additive", restart=50, reduction=1e-8, tol=1e-5)
else:
    nev = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    dec = build_structured(synth.StructuredElasticity(refine=size, parts=8), overlap=1, neumann=True, second_region="all")
    cfg = dict(schwarz_type="restricted", mode="multiplicative", restart=100, reduction=1e-6, tol=1e-6)
print(f"{which}: {dec.nglobal} DoF, {dec.nsub} subdomains, n = {[sd.n for sd in dec.subs]}, host setup {time.time() - t0:.1f} s", flush=True)
t0 = time.time()
tl = TwoLevelSchwarz(dec, coarse="none", schwarz_type=cfg["schwarz_type"], mode=cfg["mode"], subdomain_solver=local)
print(f"subdomain solver '{local}': device setup {time.time() - t0:.1f} s", flush=True)
print("engine", tl.schwarz.engine(), "levels", tl.schwarz_levels(), "max row nnz", int(np.diff(tl.rl.A_dir.indptr).max()), flush=True)
d = tl.to_device(np.random.default_rng(0).standard_normal(tl.rl.n_o))
x = tl.zeros(tl.rl.n_o)
for _ in range(3):
    tl.schwarz.apply(x, d)
tl.ctx.sync()
tl.ctx.timing(True)
tl.ctx.timing_reset()
for _ in range(10):
    tl.schwarz.apply(x, d)
tl.ctx.timing(False)
ms, cnt = tl.ctx.timer("Schwarz/local solve")
z, n = tl.A_dir.nnz, tl.rl.n
print(f"local solve {ms / cnt:.3f} ms  ({(12.0 * z + 40.0 * n) / (ms / cnt) / 1e6:.1f} GB/s algorithmic)", flush=True)
tl.schwarz.check_status()
res, hist, _ = tl.solve(reduction=cfg["reduction"], maxit=1000, solver="restartedgmressolver", restart=cfg["restart"])
print(f"one-level GMRES: {res.iterations} iterations, converged {bool(res.converged)}, {res.elapsed_s:.3f} s", flush=True)
t1 = time.time()
basis, info = geneo_basis(tl, nev=nev, tol=cfg["tol"], return_info=True, verbose=os.environ.get("DDM_VERBOSE") == "1")
print(f"GenEO: {info['iterations']} block iterations, converged {info['converged']}, {time.time() - t1:.1f} s; lambda(sub 0) = {info['eigenvalues'][0][:4]} .. {info['eigenvalues'][0][-1]:.4g}", flush=True)
tl.set_coarse_basis(basis)
tl.rebuild_combined(cfg["mode"])
res, hist, _ = tl.solve(reduction=cfg["reduction"], maxit=1000, solver="restartedgmressolver", restart=cfg["restart"])
print(f"two-level GMRES: {res.iterations} iterations, converged {bool(res.converged)}, {res.elapsed_s:.3f} s, {res.iterations / max(res.elapsed_s, 1e-9):.1f} it/s", flush=True)
tl.prec.check_status()
