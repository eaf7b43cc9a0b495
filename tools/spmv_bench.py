"""Times CsrMatrix.mv (k_spmv_stream) on a 27-point stencil matrix of N^3 rows (diagnostic).  usage: python tools/spmv_bench.py [N]"""
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

N = int(sys.argv[1]) if len(sys.argv) > 1 else 216
T = sp.diags([np.ones(N - 1), 2.0 * np.ones(N), np.ones(N - 1)], [-1, 0, 1], format="csr")
t0 = time.time()
M = sp.kron(T, sp.kron(T, T, format="csr"), format="csr")
M.sort_indices()
print(f"matrix {M.shape[0]} rows {M.nnz} nnz built in {time.time() - t0:.1f}s", flush=True)
ctx = ddm.torch_context(0)
A = ddm.CsrMatrix(ctx, M)
x = torch.as_tensor(np.random.default_rng(0).standard_normal(M.shape[0])).cuda()
y = torch.zeros_like(x)
for _ in range(5):
    A.mv(x, y)
ctx.sync()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 30
e0.record()
for _ in range(reps):
    A.mv(x, y)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
ref = M @ x.cpu().numpy()
err = np.abs(y.cpu().numpy() - ref).max() / np.abs(ref).max()
print(f"spmv {ms:.4f} ms  {12.0 * M.nnz / ms / 1e6:.1f} GB/s (12 B per non-zero)  max rel dev vs scipy {err:.2e}")
ctx.close()
