"""The solver half of the oracle (PETSc-semantics FGMRES, Jacobi, Schur
fieldsplit).  PARITY UNPINNED: nothing in the reference pins these results
(PETSc absent, no reference tests); they are checked against independent dense /
sparse-direct algebra and against the committed regression fixtures.  CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as sla

from conftest import relerr


def _dense_pc(A, B, fact):
    Ad, Bd = A.to_scipy().toarray(), B.to_scipy().toarray()
    n, m = Ad.shape[0], Bd.shape[0]
    D = np.diag(1.0 / np.diag(Ad))
    Sh = np.diag(np.diag(Bd @ D @ Bd.T))
    St = -Sh
    I, Im, Z = np.eye(n), np.eye(m), np.zeros
    L = np.block([[I, Z((n, m))], [Bd @ D, Im]])
    U = np.block([[I, D @ Bd.T], [Z((m, n)), Im]])
    Dg = np.block([[np.linalg.inv(D), Z((n, m))], [Z((m, n)), St]])
    return [np.block([[D, Z((n, m))], [Z((m, n)), np.linalg.inv(Sh)]]),
            np.linalg.inv(L @ Dg), np.linalg.inv(Dg @ U), np.linalg.inv(L @ Dg @ U)][fact]


@pytest.mark.parametrize("fact", [0, 1, 2, 3])
def test_schur_factorisations_match_dense_inverse(oracle, fact):
    A, _ = oracle.assemble(5)
    B, _ = oracle.assemble_constraints(5)
    v = np.random.default_rng(fact).standard_normal(A.nrows + 4)
    y = oracle.pc_apply(A, B, oracle.PC_SCHUR, fact, v)
    assert relerr(y, _dense_pc(A, B, fact) @ v) < 1e-13


def test_jacobi_zero_diagonal_rule(oracle):
    A, _ = oracle.assemble(5)
    B, _ = oracle.assemble_constraints(5)
    v = np.random.default_rng(1).standard_normal(A.nrows + 4)
    y = oracle.pc_apply(A, B, oracle.PC_JACOBI, 0, v)
    assert np.allclose(y[:-4], v[:-4] / A.to_scipy().diagonal())
    assert np.array_equal(y[-4:], v[-4:])          # PCJACOBI: zero diagonal -> 1


def test_fgmres_jacobi_m32(oracle, golden_m32):
    A, f = oracle.assemble(32)
    x, info = oracle.fgmres(A, f, pc_type=oracle.PC_JACOBI, rtol=1e-5)
    assert info["reason"] == 2 and info["its"] == len(golden_m32["jacobi_hist"]) - 1 == 75
    assert np.allclose(info["history"], golden_m32["jacobi_hist"], rtol=1e-9)
    assert relerr(x, golden_m32["u"]) < 1e-4
    # residual history is the true unpreconditioned residual at convergence
    r = f - A.to_scipy() @ x
    assert np.linalg.norm(r) == pytest.approx(info["rnorm"], rel=1e-6)
    assert info["rnorm"] <= 1e-5 * np.linalg.norm(f)


def test_fgmres_saddle_all_factorisations(oracle, golden_m32):
    A, f = oracle.assemble(32)
    B, g = oracle.assemble_constraints(32)
    rhs = np.concatenate([f, g])
    for fact in range(4):
        x, info = oracle.fgmres(A, rhs, B=B, pc_type=oracle.PC_SCHUR, schur_fact=fact, rtol=1e-8)
        assert info["reason"] == 2
        assert info["its"] == golden_m32["schur_its"][fact]
        assert relerr(x, golden_m32["saddle"]) < 1e-7
    assert np.allclose(info["history"], golden_m32["schur_full_hist"], rtol=1e-8)


def test_fgmres_tight_tolerance_matches_direct_solve(oracle):
    A, f = oracle.assemble(17)
    B, g = oracle.assemble_constraints(17)
    rhs = np.concatenate([f, g])
    K = sp.bmat([[A.to_scipy(), B.to_scipy().T], [B.to_scipy(), None]]).tocsc()
    x, info = oracle.fgmres(A, rhs, B=B, pc_type=oracle.PC_SCHUR, schur_fact=3, rtol=1e-12)
    assert info["reason"] == 2 and relerr(x, sla.spsolve(K, rhs)) < 1e-9


def test_fgmres_edge_cases(oracle):
    A, f = oracle.assemble(9)
    # zero right-hand side: converged at iteration 0 by the absolute tolerance
    x, info = oracle.fgmres(A, np.zeros_like(f))
    assert info["its"] == 0 and info["reason"] == 3 and not x.any()
    # iteration cap
    x, info = oracle.fgmres(A, f, pc_type=oracle.PC_NONE, max_it=3, rtol=1e-14)
    assert info["its"] == 3 and info["reason"] == -3
    # exact initial guess: nothing to do
    u = sla.spsolve(A.to_scipy().tocsc(), f)
    x, info = oracle.fgmres(A, f, x0=u, rtol=1e-8)
    assert info["its"] == 0 and info["reason"] in (2, 3)
    # restart shorter than the iteration count exercises the true-residual restart
    x, info = oracle.fgmres(A, f, pc_type=oracle.PC_JACOBI, restart=5, rtol=1e-10)
    assert info["reason"] == 2 and info["its"] > 5 and relerr(x, u) < 1e-8
    # NaN in the right-hand side
    b = f.copy(); b[3] = np.nan
    x, info = oracle.fgmres(A, b)
    assert info["reason"] == -9
