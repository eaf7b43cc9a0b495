"""Measurement aid for k_rotate_mfma / k_gram2_sym at the headline block size (run under rocprofv3 --kernel-trace --stats):
fused-rotation-shaped calls (p = 72 inner, q = 48 output columns) and the fused pair of symmetric Gram products on n rows split into
8 subdomains."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ddm = ge.import_package()
import torch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_900_000
ctx = ddm.torch_context(0)
bp = np.linspace(0, n, 9).astype(np.int64)
U = torch.randn((n, 72), dtype=torch.float64, device="cuda")
out = torch.zeros((n, 72), dtype=torch.float64, device="cuda")
Y = np.random.default_rng(0).standard_normal((8, 72, 48))
for _ in range(3):
    ddm.blockvec_rotate(ctx, bp, U, Y, out)
V1 = torch.randn((n, 72), dtype=torch.float64, device="cuda")
for _ in range(3):
    ddm.blockvec_gram2_sym(ctx, bp, U, V1, out)
print("done")
ctx.close()
