"""world_size-2 (and 4) rehearsal of the N>1 path over gloo."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(mode, nproc, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "mp_worker.py"), mode]
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("nproc", [2, 4])
def test_halo_plans_across_ranks_cpu(nproc):
    p = _launch("plans", nproc, 29531 + nproc)
    assert p.returncode == 0 and f"PLANS_OK {nproc}" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


@pytest.mark.gpu
def test_two_rank_solve_shares_one_gpu():
    p = _launch("solve", 2, 29541)
    assert p.returncode == 0 and "SOLVE_OK 2" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
