"""Generates the committed fixtures of tests/golden/ IN THIS CONTAINER (the reference cannot be built here or on the GPU box:
DESIGN.md section 6).  Two kinds of data:
  * reference_kat.npz -- the golden data of the reference's OWN tests (tests/test_galerkin_coarse_matrix.cc: 9x9 matrix,
    per-rank additive blocks and index sets, hand-written partition of unity, expected 4x4 R A R^T), transcribed in
    tests/kat_data.py with the source lines; data only.
  * poisson12_2x2x2.npz -- outputs of the CPU oracle (oracle/, the restatement pinned by those KATs) on a down-scaled instance
    of BASELINE.json configs[2]: 12^3 Q1 Poisson, 2x2x2 subdomains, overlap 2, distance POU; index maps, POU, the Galerkin
    matrix of the POU coarse space, GenEO eigenvalues (nev = 4) of two subdomains, the first residual norms of the two-level CG
    and the iteration count to 1e-10.  "Parity unpinned by reference fixtures" for these: they freeze the oracle.
  * dg32_2x2.npz / elasticity32_4.npz -- the same kind of oracle outputs on down-scaled instances of BASELINE.json configs[3]
    (Q1-DG convection-diffusion, checkerboard coefficient, GenEO on the symmetric part, additive, restarted GMRES) and
    configs[4] (P1 elasticity, GenEO with B = A_neu, restricted Schwarz, multiplicative coarse level, restarted GMRES).
usage: python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.import_package()
from dune_ddm_amd import synth  # noqa: E402
from dune_ddm_amd.problem import build_structured  # noqa: E402
from oracle import geneo_oracle as go  # noqa: E402
from tests import kat_data  # noqa: E402
from tests.oracle_bridge import oracle_objects, oracle_solve  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def instance():
    return build_structured(synth.StructuredPoisson((12, 12, 12), (2, 2, 2)), overlap=2, pou_type="distance", shrink=0, neumann=True)


CFG_DG = dict(nev=8, eig={"nev": 8}, solver="restartedgmressolver", restart=50, reduction=1e-8, maxit=500, schwarz_type="standard", mode="additive")
CFG_EL = dict(nev=12, eig={"nev": 12, "tolerance": 1e-6}, solver="restartedgmressolver", restart=100, reduction=1e-6, maxit=500,
              schwarz_type="restricted", mode="multiplicative")


def dg_instance():
    return build_structured(synth.StructuredDG2D((32, 32), (2, 2)), overlap=2, pou_type="distance", shrink=0, neumann=True)


def elasticity_instance():
    return build_structured(synth.StructuredElasticity((32, 4, 6), 4), overlap=1, pou_type="distance", shrink=0, neumann=True, second_region="all")


def oracle_geneo_run(dec, cfg):
    """GenEO basis by the oracle's IRLM per subdomain, then the outer Krylov solve of the oracle with that basis"""
    out = {"nglobal": np.int64(dec.nglobal)}
    basis = {}
    for s, sd in enumerate(dec.subs):
        out[f"sub{s}_glob"] = np.asarray(sd.glob, dtype=np.int64)
        out[f"sub{s}_n_o"] = np.int64(sd.n_o)
        out[f"sub{s}_pou"] = np.asarray(sd.pou, dtype=np.float64)
        out[f"sub{s}_A_dir_nnz"] = np.int64(sd.A_dir.nnz)
        vecs, lam = go.geneo_basis(sd.A_neu, sd.B_neu, sd.pou, cfg["eig"])
        out[f"sub{s}_geneo_lambda"] = np.asarray(lam, dtype=np.float64)
        basis[s] = [np.array(v) for v in vecs]
    it, conv, hist, x = oracle_solve(dec, coarse=basis, schwarz_type=cfg["schwarz_type"], mode=cfg["mode"], reduction=cfg["reduction"],
                                     maxit=cfg["maxit"], solver=cfg["solver"], restart=cfg["restart"])
    out["iterations"] = np.int64(it)
    out["converged"] = np.bool_(conv)
    out["residuals"] = np.asarray(hist, dtype=np.float64)
    return out, basis


def main():
    for name, inst, cfg in (("dg32_2x2.npz", dg_instance, CFG_DG), ("elasticity32_4.npz", elasticity_instance, CFG_EL)):
        out, _ = oracle_geneo_run(inst(), cfg)
        np.savez(os.path.join(HERE, name), **out)
        print("wrote", name, int(out["iterations"]), "iterations, converged", bool(out["converged"]), "lambda_0", out["sub0_geneo_lambda"][:3])
    ranks = kat_data.chain()
    np.savez(os.path.join(HERE, "reference_kat.npz"), A_global=kat_data.A_GLOBAL, A0_expected=kat_data.A0_EXPECTED,
             **{f"rank{r.rank}_A": r.A.toarray() for r in ranks}, **{f"rank{r.rank}_glob": r.glob for r in ranks},
             **{f"rank{r.rank}_owner": r.owner for r in ranks}, **{f"rank{r.rank}_public": r.public for r in ranks},
             **{f"rank{k}_pou": np.array(v) for k, v in kat_data.POU.items()})
    dec = instance()
    out = {"nglobal": np.int64(dec.nglobal)}
    for s, sd in enumerate(dec.subs):
        out[f"sub{s}_glob"] = np.asarray(sd.glob, dtype=np.int64)            # overlapping local -> global id (arrival order)
        out[f"sub{s}_n_o"] = np.int64(sd.n_o)
        out[f"sub{s}_owner_novlp"] = np.asarray(sd.owner_novlp, dtype=np.uint8)
        out[f"sub{s}_pou"] = np.asarray(sd.pou, dtype=np.float64)
        out[f"sub{s}_A_dir_nnz"] = np.int64(sd.A_dir.nnz)
    op, sp_, prec, sch, gal = oracle_objects(dec, coarse="pou")
    out["A0_pou"] = np.asarray(gal.a0.toarray() if hasattr(gal.a0, "toarray") else gal.a0, dtype=np.float64)
    for s in (0, 7):
        sd = dec.subs[s]
        lam, X, _ = go.spectra_gevp(sd.A_neu, go.scale_matrix_with_pou(sd.B_neu, sd.pou), go.EigensolverParams({"nev": 4}))
        out[f"sub{s}_geneo_lambda"] = np.asarray(lam, dtype=np.float64)
    it, conv, hist, x = oracle_solve(dec, coarse="pou", schwarz_type="standard", mode="additive", reduction=1e-10, maxit=200)
    out["cg_iterations"] = np.int64(it)
    out["cg_converged"] = np.bool_(conv)
    out["cg_residuals"] = np.asarray(hist, dtype=np.float64)
    np.savez(os.path.join(HERE, "poisson12_2x2x2.npz"), **out)
    print("wrote fixtures:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in list(out.items())[:6]}, "...", it, "iterations")


if __name__ == "__main__":
    main()
