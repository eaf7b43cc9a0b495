"""Timeline of one pipe-engine ILU(0) solve from the stamped kernel build (diagnostic).
usage: python tools/pipe_trace.py N PX PY PZ [K=V ...]   (K=V sets DDM_PIPE_K)"""
import os
import sys

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
os.environ["DDM_TRSV_MODE"] = "pipe"
for kv in sys.argv[5:]:
    k, v = kv.split("=")
    os.environ["DDM_PIPE_" + k] = v
dec = build_structured(synth.StructuredPoisson((N, N, N), P), overlap=2, pou_type="distance", shrink=0)
mats = [sd.A_dir.tocsr() for sd in dec.subs]
M = sp.block_diag(mats, format="csr")
M.sort_indices()
bp = np.concatenate([[0], np.cumsum([m.shape[0] for m in mats])]).astype(np.int64)
ctx = ddm.torch_context(0)
F = ddm.Ilu0(ctx, ddm.CsrMatrix(ctx, M), bp)
d = torch.as_tensor(np.random.default_rng(0).standard_normal(M.shape[0])).cuda()
x = torch.zeros_like(d)
for _ in range(3):
    F.solve(d, x)
ctx.sync()
st, meta = F.pipe_trace(d, x)
st, meta = F.pipe_trace(d, x)
ctx.sync()
st = st.astype(np.float64)
t0 = st[:, 0].min()
start, first, end = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0, (st[:, 2] - t0) / 100.0   # us
steps = st[:, 6]
print(f"tasks {len(st)}  kernel span {end.max():.1f} us  status {F.status()}")
for g in sorted(set(meta[:, 0]))[:2]:
    for sw in (0, 1):
        m = (meta[:, 0] == g) & (meta[:, 1] == sw)
        dur = end[m] - first[m]
        print(f"group {g} sweep {sw}: tasks {m.sum()} steps {int(steps[m].sum())} span [{start[m].min():.1f}, {end[m].max():.1f}] us; "
              f"per-step us: median {np.median(dur / steps[m]):.3f} mean {dur.sum() / steps[m].sum():.3f}; "
              f"cycles/step: tile-wait {st[m, 3].sum() / steps[m].sum():.0f} producer-wait {st[m, 4].sum() / steps[m].sum():.0f} critical {st[m, 5].sum() / steps[m].sum():.0f}; "
              f"wave-0 segments per step: top->gathers {st[m, 8].sum() / steps[m].sum():.0f} ->early done {st[m, 9].sum() / steps[m].sum():.0f} ->signalled {st[m, 10].sum() / steps[m].sum():.0f} post {st[m, 11].sum() / steps[m].sum():.0f}; steps (behind the first) that polled producers: {st[m, 12].sum() / steps[m].sum():.2%}, {st[m, 13].sum() / np.maximum(st[m, 12].sum(), 1):.0f} cycles each")
        idx = np.where(m)[0]
        sel = idx[:: max(1, len(idx) // 12)]
        for i in sel:
            print(f"    task {i - idx[0]:4d}: dequeued {start[i]:8.1f} first {first[i]:8.1f} end {end[i]:8.1f} steps {int(steps[i]):4d} "
                  f"us/step {(end[i] - first[i]) / steps[i]:.3f} tile {st[i, 3] / steps[i]:.0f} prod {st[i, 4] / steps[i]:.0f} crit {st[i, 5] / steps[i]:.0f} | a {st[i, 8] / steps[i]:.0f} b {st[i, 9] / steps[i]:.0f} c {st[i, 10] / steps[i]:.0f} d {st[i, 11] / steps[i]:.0f} cyc/step xcc {int(st[i, 7])}")
np.savez(os.path.join(ROOT, 'gpurun_out', f'pipe_trace{N}.npz'), st=st, meta=meta)
ctx.close()
