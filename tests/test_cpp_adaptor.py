"""The DUNE-facing C++ adaptors (dune-ddm_amd/dune/ddm/hip/*.hh: the reference's class names and constructor
signatures on top of the C ABI) compiled against minimal DUNE stand-ins (tests/cpp/mock) and driven like
examples/poisson.cc drives the reference classes."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _build():
    subprocess.check_call(["make", "-C", CPP], stdout=subprocess.DEVNULL)
    return os.path.join(CPP, "poisson_adaptor")


def test_adaptors_compile_and_link(ddm):
    ddm.load_library()
    exe = _build()
    assert os.path.exists(exe)
    # every C-ABI symbol the adaptors use must be exported by the library
    assert os.path.exists(os.path.join(CPP, "mpi_exchange_check.o"))   # mpi_exchange.hh compiles against the image's MPI headers (-DHAVE_MPI=1)
    for e in (exe, os.path.join(CPP, "geneo_adaptor"), os.path.join(CPP, "coarse_adaptor"), os.path.join(CPP, "twolevel_adaptor"),
              os.path.join(CPP, "twolevel_pdelab")):   # (the last one: the PDELab-facing class compiled with HAVE_DUNE_PDELAB=1)
        out = subprocess.run(["nm", "-D", "--undefined-only", e], capture_output=True, text=True).stdout
        used = sorted({ln.split()[-1] for ln in out.splitlines() if " ddm_" in ln})
        assert used and all(u in ddm.SYMBOLS for u in used), [u for u in used if u not in ddm.SYMBOLS]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["additive", "multiplicative"])
def test_adaptor_cg_matches_oracle(ddm, tmp_path, mode):
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import build_structured
    from tests.oracle_bridge import oracle_solve
    exe = _build()
    dec = build_structured(synth.StructuredPoisson((14, 13, 12), (1, 1, 1)), overlap=1, pou_type="distance")
    sd = dec.subs[0]
    A = sd.A.tocsr()
    np.asarray(A.indptr, dtype=np.int64).tofile(tmp_path / "rowptr.bin")
    np.asarray(A.indices, dtype=np.int32).tofile(tmp_path / "col.bin")
    np.asarray(A.data, dtype=np.float64).tofile(tmp_path / "val.bin")
    sd.b.astype(np.float64).tofile(tmp_path / "b.bin")
    sd.dirichlet_ovlp.astype(np.uint8).tofile(tmp_path / "dirichlet.bin")
    sd.pou.astype(np.float64).tofile(tmp_path / "pou.bin")
    p = subprocess.run([exe, str(tmp_path), mode], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    hist = np.array([float(ln.split()[2]) for ln in p.stdout.splitlines() if ln.startswith("it ")])
    assert "errors_caught 5" in p.stdout          # + the two coarse-solver key errors (galerkin_preconditioner.hh:338-346)
    gs = [ln for ln in p.stdout.splitlines() if ln.startswith("getSolver")][0].split()
    # SchwarzPreconditioner::getSolver() (schwarz.hh:155): deterministic, and on one rank identical to Schwarz::apply (standard type)
    assert float(gs[2]) == 0.0 and float(gs[4]) == 0.0 and float(gs[6]) > 0 and gs[8] == "1", gs
    maxit = 500 if mode == "additive" else len(hist) - 1
    it, conv, hist_o, _ = oracle_solve(dec, reduction=1e-10, maxit=maxit, coarse="pou", schwarz_type="standard", mode=mode)
    ho = np.array(hist_o)
    assert len(hist) == len(ho)
    assert (np.abs(hist - ho) <= 1e-8 * ho + 1e-12 * ho[0]).all()


def _dump_csr(path, pre, M):
    M = M.tocsr()
    np.asarray(M.indptr, dtype=np.int64).tofile(path / f"{pre}_rowptr.bin")
    np.asarray(M.indices, dtype=np.int32).tofile(path / f"{pre}_col.bin")
    np.asarray(M.data, dtype=np.float64).tofile(path / f"{pre}_val.bin")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["device", "device_cholmod"])
def test_adaptor_factory_style_device_solver(ddm, tmp_path, mode):
    """examples/poisson.cc:229-321 through the device-resident pieces: coarse space from a CoarseSpaceBuilder task (POUCoarseSpace),
    zero_at_dirichlet, GalerkinPreconditioner, CombinedPreconditioner, solver from Dune::getHipSolver -- the whole CG runs on the
    device with one upload and one download; subdomain solver `ilu0` resp. `cholmod`; plus the Dune::InverseOperator plugin
    (HipSubdomainSolver).  Iteration count and solution against the oracle."""
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import build_structured
    from tests.oracle_bridge import oracle_solve
    exe = _build()
    dec = build_structured(synth.StructuredPoisson((14, 13, 12), (1, 1, 1)), overlap=1, pou_type="distance")
    sd = dec.subs[0]
    A = sd.A.tocsr()
    np.asarray(A.indptr, dtype=np.int64).tofile(tmp_path / "rowptr.bin")
    np.asarray(A.indices, dtype=np.int32).tofile(tmp_path / "col.bin")
    np.asarray(A.data, dtype=np.float64).tofile(tmp_path / "val.bin")
    sd.b.astype(np.float64).tofile(tmp_path / "b.bin")
    sd.dirichlet_ovlp.astype(np.uint8).tofile(tmp_path / "dirichlet.bin")
    sd.pou.astype(np.float64).tofile(tmp_path / "pou.bin")
    p = subprocess.run([exe, str(tmp_path), mode], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("device_solve")][0].split()
    its, conv = int(line[2]), int(line[4])
    it, convo, hist_o, xo = oracle_solve(dec, reduction=1e-10, maxit=500, coarse="pou", schwarz_type="standard", mode="additive",
                                         local_solver="ilu0" if mode == "device" else "direct")
    assert conv == 1 and convo and its == it, (its, it)
    x = np.fromfile(tmp_path / "x_device.bin", dtype=np.float64)
    assert np.abs(x - xo[0]).max() <= 1e-8 * np.abs(xo[0]).max()
    res = float([ln for ln in p.stdout.splitlines() if ln.startswith("plugin")][0].split()[2])
    assert res < 1e-10 and "errors_caught 2" in p.stdout
    if mode == "device":
        # the same run with the in-library RCCL exchange installed from C++ (dune/ddm/hip/rccl_exchange.hh: id, ncclCommInitRank on a
        # size-1 communicator, reductions routed through ncclAllReduce -- the self-test mode): same iteration count, same bits
        x0 = x.copy()
        q = subprocess.run([exe, str(tmp_path), mode], capture_output=True, text=True, timeout=300, env=dict(os.environ, DDM_TEST_RCCL="1"))
        assert q.returncode == 0, q.stdout[-2000:] + q.stderr[-2000:]
        assert "rccl_exchange installed rank 0 of 1" in q.stdout and "rccl_bad_rank_caught" in q.stdout
        line2 = [ln for ln in q.stdout.splitlines() if ln.startswith("device_solve")][0].split()
        assert line2[2] == line[2] and line2[4] == "1"
        assert np.array_equal(np.fromfile(tmp_path / "x_device.bin", dtype=np.float64), x0)


@pytest.mark.gpu
def test_geneo_coarse_space_adaptor_matches_oracle(ddm, tmp_path):
    """GenEOCoarseSpace(A, B, pou, ptree, taskflow) -> get_basis() on one subdomain of a 2 x 2 x 2 decomposition (one rank = one
    subdomain, as in the reference) against the oracle's Spectra restatement: eigenvalues 1e-6, spans 2e-3; the adaptor returns the
    vectors as the reference does (POU-scaled, 2-normalised; zero_at_dirichlet is the caller's job)."""
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import build_structured
    from oracle import geneo_oracle as go
    _build()
    exe = os.path.join(CPP, "geneo_adaptor")
    dec = build_structured(synth.StructuredPoisson((25, 25, 25), (2, 2, 2), synth.islands_kappa((24, 24, 24), 1e4, 6, 2)), overlap=2, pou_type="distance", neumann=True)
    sd = dec.subs[5]
    _dump_csr(tmp_path, "A", sd.A_neu)
    _dump_csr(tmp_path, "B", sd.B_neu)
    sd.pou.astype(np.float64).tofile(tmp_path / "pou.bin")
    nev = 4
    p = subprocess.run([exe, str(tmp_path), str(nev)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert f"size {nev} consumed {nev}" in p.stdout and "errors_caught 2" in p.stdout
    lam = np.array([float(ln.split()[1]) for ln in p.stdout.splitlines() if ln.startswith("lambda")])
    vecs, lam_o = go.geneo_basis(sd.A_neu, sd.B_neu, sd.pou, {"nev": nev})
    assert np.allclose(lam, lam_o, rtol=1e-6)
    Bd = np.fromfile(tmp_path / "basis.bin", dtype=np.float64).reshape(nev, sd.n)
    assert np.abs(np.linalg.norm(Bd, axis=1) - 1.0).max() < 1e-12
    Qd, _ = np.linalg.qr(Bd.T)
    Qo, _ = np.linalg.qr(np.array(vecs).T)
    assert np.linalg.norm(Qd - Qo @ (Qo.T @ Qd), 2) < 2e-3


@pytest.mark.gpu
def test_remaining_coarse_space_adaptors_match_oracle(ddm, tmp_path):
    """MsGFEMCoarseSpace, ConstraintGenEOCoarseSpace, GenEORingCoarseSpace, MsGFEMRingCoarseSpace, HarmonicExtensionCoarseSpace
    (+ EnergyMinimalExtension) constructed with the reference's signatures and run through a taskflow, one subdomain of a 2 x 2 x 2
    decomposition, against oracle/coarse_oracle.py: eigenvalues 1e-6, spans 2e-3, harmonic extension 1e-10."""
    import scipy.sparse as sp
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import build_structured
    from oracle import coarse_oracle as co
    from oracle import geneo_oracle as go
    _build()
    exe = os.path.join(CPP, "coarse_adaptor")
    overlap, nev = 2, 4
    grid = synth.StructuredPoisson((29, 27, 25), (2, 2, 2))
    dec = build_structured(grid, overlap=overlap, pou_type="distance", neumann=True, second_region="all")
    sd = dec.subs[3]

    def ring_of(width):
        ring = np.nonzero(sd.boundary_dist <= width)[0]
        M = grid.neumann_matrix(sd.glob, sd.boundary_dist <= width, sd.dirichlet_ovlp)
        return sp.csr_matrix(M)[ring][:, ring].tocsr(), ring

    (R1, ring1), (R2, ring2) = ring_of(2 * overlap + 1), ring_of(2 * overlap)
    _dump_csr(tmp_path, "N", sd.A_neu)
    _dump_csr(tmp_path, "D", sd.A_dir)
    _dump_csr(tmp_path, "R1", R1)
    _dump_csr(tmp_path, "R2", R2)
    sd.pou.astype(np.float64).tofile(tmp_path / "pou.bin")
    np.asarray(sd.dirichlet_ovlp, dtype=np.float64).tofile(tmp_path / "dirichlet.bin")
    sd.boundary.astype(np.float64).tofile(tmp_path / "boundary.bin")
    ring1.astype(np.int64).tofile(tmp_path / "ring1.bin")
    ring2.astype(np.int64).tofile(tmp_path / "ring2.bin")
    nb = int(sd.boundary.sum())
    bdata = np.array([np.ones(nb), np.sin(np.arange(nb))])
    bdata.tofile(tmp_path / "bdata.bin")
    p = subprocess.run([exe, str(tmp_path), str(nev), str(overlap)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert f"sizes {nev} {nev} {nev} {nev} 2 5" in p.stdout and "errors_caught 2" in p.stdout

    def lam_of(name):
        return np.array([float(ln.split()[2]) for ln in p.stdout.splitlines() if ln.startswith("lambda " + name + " ")])

    def span_ok(name, vecs, lam, k=nev):
        Bd = np.fromfile(tmp_path / (name + ".bin"), dtype=np.float64).reshape(k, sd.n)
        assert np.abs(np.linalg.norm(Bd, axis=1) - 1.0).max() < 1e-12
        Q, _ = np.linalg.qr(Bd.T)
        below = [v / np.linalg.norm(v) for v, l in zip(vecs, lam) if l < lam[-1] * (1 - 1e-3)]
        assert len(below) >= k - 2
        for u in below:
            assert np.linalg.norm(u - Q @ (Q.T @ u)) < 2e-3, name

    vecs, lam = co.msgfem_basis(sd.A_neu, sd.A_dir, sd.pou, sd.dirichlet_ovlp, sd.boundary, {"nev": nev})
    assert np.allclose(lam_of("msgfem"), lam, rtol=1e-6)
    span_ok("msgfem", vecs, lam)
    vecs, lam = go.geneo_basis(sd.A_neu, sd.A_neu, sd.pou, {"nev": nev})
    span_ok("constraint_geneo", vecs, lam)
    vecs, lam = co.geneo_ring_basis(sd.A_dir, R1, sd.pou, ring1, {"nev": nev})
    assert np.allclose(lam_of("geneo_ring"), lam, rtol=1e-6)
    span_ok("geneo_ring", vecs, lam)
    vecs, lam = co.msgfem_ring_basis(sd.A_dir, R2, overlap, sd.pou, 0, sd.dirichlet_ovlp, sd.boundary, ring2, {"nev": nev})
    assert np.allclose(lam_of("msgfem_ring"), lam, rtol=1e-6)
    span_ok("msgfem_ring", vecs, lam)
    vecs, sv = co.svd_basis(sd.A_dir, sd.pou, sd.boundary, sd.dirichlet_ovlp, n_vectors=5)
    assert np.allclose(lam_of("svd"), sv[:5], rtol=1e-6)
    Bd = np.fromfile(tmp_path / "svd.bin", dtype=np.float64).reshape(5, sd.n)
    Q, _ = np.linalg.qr(Bd.T)
    for u, s_ in zip(vecs, sv[:5]):
        if s_ > sv[4] * (1 + 1e-3):
            assert np.linalg.norm(u - Q @ (Q.T @ u)) < 2e-3
    ref = np.array(co.harmonic_extension_basis(sd.A_dir, sd.pou, list(bdata), sd.boundary))
    got = np.fromfile(tmp_path / "harmonic.bin", dtype=np.float64).reshape(2, sd.n)
    assert np.abs(got - ref).max() < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [("multiplicative", "ilu0", "restartedgmressolver"), ("additive", "ilu0", "bicgstabsolver"),
                                 ("multiplicative", "umfpack", "restartedgmressolver")])
def test_twolevel_schwarz_solver_adaptor(ddm, tmp_path, cfg):
    """The C++ TwoLevelSchwarzSolver adaptor (dune/ddm/hip/twolevel_schwarz.hh: statement sequence of twolevel_schwarz.hh:106-146 --
    POUCoarseSpace of the template vectors 1, x, y, xy, SchwarzPreconditioner "fine" with novlp_comm set, GalerkinPreconditioner
    "coarse" with its factory key, CombinedPreconditioner with the mode key in the sub-tree itself, NonOverlappingOperator, solver from
    the "solver" sub-tree, consistent right-hand side, solve) on the DG problem of examples/convectiondiffusiondg.cc, one rank,
    against the oracle assembled from the same pieces; two consecutive apply() calls agree bit for bit."""
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import build_structured
    from oracle import apply_oracle as ao
    from tests.oracle_bridge import oracle_solve
    mode, local, krylov = cfg
    _build()
    exe = os.path.join(CPP, "twolevel_adaptor")
    grid = synth.StructuredDG2D((16, 16), (1, 1))
    dec = build_structured(grid, overlap=1)
    sd = dec.subs[0]
    A = sd.A.tocsr()
    np.asarray(A.indptr, dtype=np.int64).tofile(tmp_path / "rowptr.bin")
    np.asarray(A.indices, dtype=np.int32).tofile(tmp_path / "col.bin")
    np.asarray(A.data, dtype=np.float64).tofile(tmp_path / "val.bin")
    sd.b.astype(np.float64).tofile(tmp_path / "b.bin")
    sd.pou.astype(np.float64).tofile(tmp_path / "pou.bin")
    X = grid.dof_coords(sd.glob)
    np.ascontiguousarray(X, dtype=np.float64).tofile(tmp_path / "coords.bin")
    p = subprocess.run([exe, str(tmp_path), mode, local, krylov], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "errors_caught 2" in p.stdout
    lines = [ln.split() for ln in p.stdout.splitlines() if ln.startswith("solve ")]
    assert len(lines) == 2 and lines[0][2:] == lines[1][2:]                      # second apply(): same result
    its, conv, red = int(lines[0][3]), int(lines[0][5]), float(lines[0][7])
    assert lines[0][11] == "1" and lines[0][13] == "4"                           # fine->novlp_comm set (:109); 4 template vectors
    templ = [[np.ones(sd.n), X[:, 0], X[:, 1], X[:, 0] * X[:, 1]]]
    basis = ao.pou_coarse_space([sd.pou], templ)
    it, convo, hist_o, xo = oracle_solve(dec, coarse={0: basis[0]}, schwarz_type="restricted", mode=mode, reduction=1e-8, maxit=300, solver=krylov, restart=50,
                                         local_solver="ilu0" if local == "ilu0" else "direct")
    # BiCGSTAB amplifies rounding more than the other two (tests/test_gpu_parity.py::test_bicgstab_history_matches_oracle): on this
    # problem the half step that crosses 1e-8 lands on either side of the threshold (measured 28 device / 29 oracle iterations)
    assert conv == 1 and convo and abs(its - it) <= (1 if krylov == "bicgstabsolver" else 0), (its, it)
    z0 = np.fromfile(tmp_path / "z0.bin", dtype=np.float64)
    z1 = np.fromfile(tmp_path / "z1.bin", dtype=np.float64)
    assert np.array_equal(z0, z1)
    assert np.abs(z0 - xo[0]).max() <= 1e-6 * np.abs(xo[0]).max()
    assert red <= 1e-8
    # The PDELab-facing class of the same header (HAVE_DUNE_PDELAB=1; stand-in PDELab containers and single-process stand-ins of the
    # reference's setup layer, tests/cpp/mock/dune/{pdelab,ddm}): constructor from function space + constraints, first apply() creating
    # the overlapping objects, second one refreshing the matrix values, norm(), the result storage -- must reproduce the core's
    # vectors bit for bit (one rank: the partition of unity is 1 everywhere, as in pou.bin).
    assert np.all(sd.pou == 1.0)
    q = subprocess.run([os.path.join(CPP, "twolevel_pdelab"), str(tmp_path), mode, local, krylov], capture_output=True, text=True, timeout=300)
    assert q.returncode == 0, q.stdout[-2000:] + q.stderr[-2000:]
    pl = [ln.split() for ln in q.stdout.splitlines() if ln.startswith("solve ")]
    assert len(pl) == 2 and pl[0][2:] == pl[1][2:] and pl[0][2:10] == lines[0][2:10], (pl, lines)
    zp0 = np.fromfile(tmp_path / "zp0.bin", dtype=np.float64)
    zp1 = np.fromfile(tmp_path / "zp1.bin", dtype=np.float64)
    assert np.array_equal(zp0, z0) and np.array_equal(zp1, z0)
    extra = {ln.split()[0]: ln.split() for ln in q.stdout.splitlines() if ln.startswith(("nonadditive_input", "constrained"))}
    assert extra["nonadditive_input"][2] == str(its) and extra["nonadditive_input"][4] == "1"
    assert extra["constrained"][4] == "1"
