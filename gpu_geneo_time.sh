python - <<'PY'
import sys, time
sys.path.insert(0,'.')
import __graft_entry__ as ge
ge.import_package()
from dune_ddm_amd import synth
from dune_ddm_amd.problem import build_structured
from dune_ddm_amd.solver import TwoLevelSchwarz
from dune_ddm_amd.geneo import geneo_basis
import os; N=int(os.environ.get("GRID","120"))
t=time.time()
dec = build_structured(synth.StructuredPoisson((N,N,N), (2, 2, 2)), overlap=2, pou_type="distance", neumann=True)
print("host setup", time.time()-t)
tl = TwoLevelSchwarz(dec, coarse="none")
t=time.time()
basis, info = geneo_basis(tl, nev=20, tol=1e-5, return_info=True, verbose=True, maxit=12)
print("geneo 12 its", time.time()-t)
PY
