"""world_size-2 (and 4) rehearsal of the N>1 path over gloo."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(mode, nproc, port, **extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1", **extra)
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


@pytest.mark.gpu
def test_two_rank_solve_with_distributed_setup():
    """SURVEY 8f-1 end to end: overlap extension, overlapping matrix, partition of unity and halo plans built rank by rank over gloo
    (no global knowledge), then the device solve; history == oracle on the global-knowledge setup."""
    p = _launch("solve_dist", 2, 29561)
    assert p.returncode == 0 and "SOLVE_DIST_OK 2" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


@pytest.mark.gpu
def test_rccl_exchange_self_test_on_one_gpu(ddm):
    """The in-library RCCL exchange (ddm_ctx_set_rccl: dlopen of librccl, ncclCommInitRank, grouped ncclSend / ncclRecv per halo,
    ncclAllReduce for dots and the coarse defect, all on the context's stream) exercised on ONE GPU: a communicator of size 1 in
    self-test mode routes the rank's own halo segment and every reduction through RCCL.  Same solve as without it, bit for bit
    (a size-1 all-reduce and a self send/recv move data without arithmetic)."""
    import numpy as np
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import build_structured
    from dune_ddm_amd.solver import TwoLevelSchwarz
    dec = build_structured(synth.StructuredPoisson((15, 14, 13), (2, 2, 2)), overlap=2, pou_type="distance")
    tl0 = TwoLevelSchwarz(dec, coarse="pou")
    res0, hist0, x0 = tl0.solve(reduction=1e-10, maxit=300)
    tl0.ctx.close()
    os.environ["DDM_RCCL_SELFTEST"] = "1"
    try:
        tl = TwoLevelSchwarz(dec, coarse="pou")
    finally:
        del os.environ["DDM_RCCL_SELFTEST"]
    assert tl.exchange == "rccl"
    res, hist, x = tl.solve(reduction=1e-10, maxit=300)
    tl.prec.check_status()
    assert res.converged and res.iterations == res0.iterations
    assert np.array_equal(np.asarray(hist), np.asarray(hist0)) and np.array_equal(x.cpu().numpy(), x0.cpu().numpy())
    tl.ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["rccl", "callback"])
def test_two_rank_nccl_solve(exchange):
    """world size 2, one process per GPU over RCCL: in-library exchange and the callback variant.  Needs two visible devices
    (skipped on a one-GPU box: RCCL refuses two ranks on one device)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    p = _launch("solve_nccl", 2, 29551 if exchange == "rccl" else 29552, DDM_EXCHANGE=exchange)
    assert p.returncode == 0 and f"NCCL_SOLVE_OK 2" in p.stdout and f" {exchange} " in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
