"""Times one ILU(0) solve (all subdomains of the rank) per engine on a structured Poisson decomposition.
usage: python tools/trsv_engines_bench.py N PX PY PZ engine[,engine...] [reps]   (engine = pipe, xcd2, levels, ...; pipe:K=V sets DDM_PIPE_K)"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ddm = ge.import_package()
import torch  # noqa: E402
from dune_ddm_amd import synth  # noqa: E402
from dune_ddm_amd.problem import build_structured  # noqa: E402

N = int(sys.argv[1])
P = tuple(int(a) for a in sys.argv[2:5])
engines = sys.argv[5].split(",")
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
t0 = time.time()
dec = build_structured(synth.StructuredPoisson((N, N, N), P), overlap=2, pou_type="distance", shrink=0)
mats = [sd.A_dir.tocsr() for sd in dec.subs]
M = sp.block_diag(mats, format="csr")
M.sort_indices()
bp = np.concatenate([[0], np.cumsum([m.shape[0] for m in mats])]).astype(np.int64)
n, z = M.shape[0], M.nnz
print(f"problem {N}^3 {P}: n={n} nnz={z} built in {time.time() - t0:.1f}s", flush=True)
ctx = ddm.torch_context(0)
A = ddm.CsrMatrix(ctx, M)
d = torch.as_tensor(np.random.default_rng(0).standard_normal(n)).cuda()
ref = None
alg_bytes = 12.0 * (z - n) + 40.0 * n
for spec in engines:
    parts = spec.split(":")
    os.environ["DDM_TRSV_MODE"] = parts[0]
    for k in [k for k in os.environ if k.startswith("DDM_PIPE_") and k != "DDM_PIPE_VERBOSE"]:
        del os.environ[k]
    for kv in parts[1:]:
        k, v = kv.split("=")
        os.environ["DDM_PIPE_" + k] = v
    t0 = time.time()
    F = ddm.Ilu0(ctx, A, bp)
    x = torch.zeros_like(d)
    F.solve(d, x)
    ctx.sync()
    tb = time.time() - t0
    st = F.status()
    for _ in range(3):
        F.solve(d, x)
    ctx.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    stream = torch.cuda.current_stream()
    e0.record(stream)
    for _ in range(reps):
        F.solve(d, x)
    e1.record(stream)
    ctx.sync()
    ms = e0.elapsed_time(e1) / reps
    xr = x.cpu().numpy()
    if ref is None:
        ref = xr
    same = bool(np.array_equal(xr, ref))
    print(f"engine {spec:28s} status {st} setup+first {tb:6.1f}s  {ms:8.3f} ms/solve  {alg_bytes / ms / 1e6:8.1f} GB/s (algorithmic)  identical-to-first {same}  status-after {F.status()}", flush=True)
    del F
ctx.close()
