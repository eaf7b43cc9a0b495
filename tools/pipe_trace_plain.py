"""Pipe-engine timing on 8 copies of a plain N^3 27-point matrix (no overlap shell: every triangular row has <= 13 entries, i.e. no
wide tiles) -- the configuration used to compare builds of the kernel (diagnostic).  usage: python tools/pipe_trace_plain.py N"""
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

N = int(sys.argv[1])
os.environ["DDM_TRSV_MODE"] = "pipe"
A = synth.StructuredPoisson((N, N, N), (1, 1, 1)).subdomains()[0].A.tocsr()
M = sp.block_diag([A] * 8, format="csr")
M.sort_indices()
bp = (np.arange(9) * A.shape[0]).astype(np.int64)
ctx = ddm.torch_context(0)
F = ddm.Ilu0(ctx, ddm.CsrMatrix(ctx, M), bp)
d = torch.as_tensor(np.random.default_rng(0).standard_normal(M.shape[0])).cuda()
x = torch.zeros_like(d)
for _ in range(3):
    F.solve(d, x)
ctx.sync()
x0 = x.clone()
t0 = time.perf_counter()
for _ in range(20):
    F.solve(d, x)
ctx.sync()
dt = (time.perf_counter() - t0) / 20
print(f"engine {F.engine()} status {F.status()} ms/solve {1e3 * dt:.3f} repeat-identical {bool((x == x0).all())} checksum {float(x.double().sum()):.17g} absmax {float(x.abs().max()):.17g}")
ctx.close()
