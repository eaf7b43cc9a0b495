"""Wall-clock and per-launch timing of the single-vector solve with the device supernodal factor (diagnostic).
usage: python tools/sn_solve_probe.py dg|elasticity|poisson64 [K=V ...]   (K=V sets the environment variable K)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    os.environ[k] = v
os.environ["DDM_DIRECT_ENGINE"] = "device"
os.environ["DDM_PIPE_VERBOSE"] = "1"
import __graft_entry__ as ge
ddm = ge.import_package()
import torch
from dune_ddm_amd import synth
from dune_ddm_amd.problem import RankLocal, build_structured
which = sys.argv[1]
if which == "dg":
    dec = build_structured(synth.StructuredDG2D((512, 512), (4, 2)), overlap=2)
    general = True
elif which == "elasticity":
    dec = build_structured(synth.StructuredElasticity(refine=1, parts=8), overlap=1)
    general = False
else:
    dec = build_structured(synth.StructuredPoisson((64, 64, 64), (1, 1, 1)), overlap=1, pou_type="distance")
    general = False
rl = RankLocal(dec, 0, 1)
ctx = ddm.torch_context(0)
A = ddm.CsrMatrix(ctx, rl.A_dir)
t0 = time.perf_counter()
F = ddm.Ilu0(ctx, A, rl.block_ptr, direct=True, general=general)
ctx.sync()
print(f"factor: {time.perf_counter() - t0:.2f} s, refinement {F.refinement()[0]}")
n = rl.n
d = torch.as_tensor(np.random.default_rng(0).standard_normal(n)).cuda()
x = torch.zeros_like(d)
for _ in range(3):
    F.solve(d, x)
ctx.sync()
for reps in (1, 50):
    t0 = time.perf_counter()
    for _ in range(reps):
        F.solve(d, x)
    t_enq = time.perf_counter() - t0
    ctx.sync()
    t_all = time.perf_counter() - t0
    print(f"{reps} solve(s): enqueue {1e3 * t_enq / reps:.3f} ms each, wall {1e3 * t_all / reps:.3f} ms each; status {F.status()}")
r = rl.A_dir @ x.cpu().numpy() - d.cpu().numpy()
print("relative residual", np.linalg.norm(r) / np.linalg.norm(d.cpu().numpy()))
F.close() if hasattr(F, "close") else None
del F
import gc
gc.collect()
ctx.close()
