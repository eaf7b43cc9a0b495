python - <<'PY'
import sys, time
sys.path.insert(0,'.')
import __graft_entry__ as ge
ddm = ge.import_package()
import numpy as np, torch
from dune_ddm_amd import synth
from dune_ddm_amd.problem import build_structured, RankLocal
import os
N=int(os.environ.get("GRID","216"))
dec = build_structured(synth.StructuredPoisson((N,N,N), (2, 2, 2)), overlap=2, pou_type="distance")
rl = RankLocal(dec)
ctx = ddm.torch_context(0)
A = ddm.CsrMatrix(ctx, rl.A_dir)
F = ddm.Ilu0(ctx, A, rl.block_ptr)
d = torch.rand(rl.n, dtype=torch.float64, device="cuda")
x = torch.zeros_like(d)
for _ in range(3): F.solve(d, x)
ctx.sync()
t=time.time()
for _ in range(10): F.solve(d, x)
ctx.sync()
print("solve ms", (time.time()-t)*100, "levels", F.num_levels())
st = F.debug_stamps(d, x)
items = int(st[4])
print("direct chunks", int(st[6]))
tick_us = 1.0
tot = float(st[5])
print("items", items, "per item share of the kernel:")
for name, v in zip(("tile","poll","gather","drain"), st[:4]):
    print(f"  {name:7s} {100.0*float(v)/tot:6.1f} %")
PY
