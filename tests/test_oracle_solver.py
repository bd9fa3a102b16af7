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


def _fgmres_textbook(K, Minv, b, restart, rtol, max_it):
    """Independent restatement (numpy, dense): flexible GMRES with right preconditioning (Saad,
    Iterative Methods, Alg. 9.6), classical Gram-Schmidt without refinement, Givens QR of the
    Hessenberg, residual estimate |g_{j+1}|, PETSc's default convergence test on the true-norm scale
    (||r|| <= max(rtol ||b||, 1e-50)), true residual recomputed at every restart, x0 = 0."""
    n = len(b)
    x = np.zeros(n)
    hist, its = [], 0
    ttol = max(rtol * np.linalg.norm(b), 1e-50)
    while True:
        r = b - K @ x
        beta = np.linalg.norm(r)
        if its == 0:
            hist.append(beta)
        if beta <= ttol or its >= max_it:
            return x, hist
        V = np.zeros((restart + 1, n)); Z = np.zeros((restart, n))
        H = np.zeros((restart + 1, restart)); cs = np.zeros(restart); sn = np.zeros(restart)
        gvec = np.zeros(restart + 1); gvec[0] = beta
        V[0] = r / beta
        j_done = 0
        for j in range(restart):
            Z[j] = Minv @ V[j]
            w = K @ Z[j]
            h = V[:j + 1] @ w                       # classical: all projections from the same w
            w = w - h @ V[:j + 1]
            tt = np.linalg.norm(w)
            H[:j + 1, j] = h; H[j + 1, j] = tt
            for i in range(j):
                a, c_ = H[i, j], H[i + 1, j]
                H[i, j] = cs[i] * a + sn[i] * c_
                H[i + 1, j] = cs[i] * c_ - sn[i] * a
            d = np.hypot(H[j, j], H[j + 1, j])
            cs[j], sn[j] = H[j, j] / d, H[j + 1, j] / d
            H[j, j] = d; H[j + 1, j] = 0.0
            gvec[j + 1] = -sn[j] * gvec[j]; gvec[j] = cs[j] * gvec[j]
            its += 1; j_done = j + 1
            hist.append(abs(gvec[j + 1]))
            if abs(gvec[j + 1]) <= ttol or its >= max_it:
                break
            V[j + 1] = w / tt
        y = np.linalg.solve(np.triu(H[:j_done, :j_done]), gvec[:j_done])
        x = x + y @ Z[:j_done]
        if hist[-1] <= ttol or its >= max_it:
            return x, hist


@pytest.mark.parametrize("pc,fact,restart", [("jacobi", 0, 30), ("schur", 3, 30), ("schur", 1, 7), ("schur", 0, 12)])
def test_fgmres_history_matches_textbook_restatement(oracle, pc, fact, restart):
    """The C oracle against a SECOND, structurally independent restatement of the same algorithm
    (dense numpy, written from the textbook): iteration counts equal, residual histories equal to
    rounding over the first cycles, solutions equal.  This pins the oracle's FGMRES mechanics
    (Arnoldi/CGS, Givens recurrence, restart, solution update) independently of its own C code."""
    A, f = oracle.assemble(12)
    B, g = oracle.assemble_constraints(12)
    Ad = A.to_scipy().toarray()
    if pc == "jacobi":
        K, b = Ad, f
        Minv = np.diag(1.0 / np.diag(Ad))
        x, info = oracle.fgmres(A, b, pc_type=oracle.PC_JACOBI, restart=restart, rtol=1e-10)
    else:
        Bd = B.to_scipy().toarray()
        K = np.block([[Ad, Bd.T], [Bd, np.zeros((4, 4))]])
        b = np.concatenate([f, g])
        Minv = _dense_pc(A, B, fact)
        x, info = oracle.fgmres(A, b, B=B, pc_type=oracle.PC_SCHUR, schur_fact=fact, restart=restart, rtol=1e-10)
    xt, hist = _fgmres_textbook(K, Minv, b, restart, 1e-10, 10000)
    assert info["reason"] == 2
    assert abs(info["its"] - (len(hist) - 1)) <= 1
    k = min(len(hist), len(info["history"]), 2 * restart)
    assert np.allclose(info["history"][:k], hist[:k], rtol=1e-7)
    assert relerr(x, xt) < 1e-8


def test_converged_default_reference_norm(oracle):
    """KSPConvergedDefault at iteration 0 (PETSc iterativ.c, as published; parity unpinned: nothing in
    the reference pins it): zero guess -> tolerances relative to the initial residual; non-zero guess ->
    relative to ||b||, or to the initial residual when b = 0; -ksp_divtol compares against the same norm."""
    A, f = oracle.assemble(9)
    n = A.nrows
    xs = sla.spsolve(A.to_scipy().tocsc(), f)
    # b = 0 with a non-zero guess: the solve must iterate towards x = 0 (ttol = rtol * ||r0||, not abstol)
    x, info = oracle.fgmres(A, np.zeros(n), x0=xs, pc_type=oracle.PC_JACOBI, rtol=1e-6)
    assert info["reason"] == 2 and info["its"] > 0
    assert info["rnorm"] <= 1e-6 * info["rnorm0"] and np.linalg.norm(x) < 1e-5 * np.linalg.norm(xs)
    # non-zero guess, b != 0: relative to ||b||, not to the (much larger) initial residual
    x, info = oracle.fgmres(A, f, x0=1e6 * xs, pc_type=oracle.PC_JACOBI, rtol=1e-3, dtol=1e30)
    assert info["reason"] == 2 and info["rnorm"] <= 1e-3 * np.linalg.norm(f) < 1e-3 * info["rnorm0"]
    # the divergence test refers to ||b|| too: an initial residual 1e6 times ||b|| is "diverged" at once
    x, info = oracle.fgmres(A, f, x0=1e6 * xs, pc_type=oracle.PC_JACOBI, rtol=1e-3)       # default divtol 1e4
    assert info["reason"] == -4 and info["its"] == 0
    # zero guess: DTOL can only come from growth during the iteration; a tiny divtol triggers it at iteration 0
    x, info = oracle.fgmres(A, f, pc_type=oracle.PC_JACOBI, rtol=1e-12, dtol=0.5)
    assert info["reason"] == -4 and info["its"] == 0


@pytest.mark.parametrize("fact", [0, 1, 2, 3])
def test_general_constraint_block_against_dense_algebra(oracle, fact):
    """A general sparse A10 block (40 rows, far beyond the build-defined 4): every factorisation against the
    dense block inverse, S^ against diag(B D B^T), and a converged solve against scipy's sparse LU."""
    A, f = oracle.assemble(7, 6)
    n, m = A.nrows, 40
    rng = np.random.default_rng(7)
    Bd = np.where(rng.random((m, n)) < 0.2, rng.standard_normal((m, n)), 0.0)
    rp = np.concatenate([[0], np.cumsum((Bd != 0).sum(1))]).astype(np.int32)
    B = oracle.CSR(rp, np.concatenate([np.nonzero(r)[0] for r in Bd]).astype(np.int32), Bd[Bd != 0], n)
    v = rng.standard_normal(n + m)
    assert relerr(oracle.pc_apply(A, B, oracle.PC_SCHUR, fact, v), _dense_pc(A, B, fact) @ v) < 1e-12
    D = 1.0 / A.to_scipy().diagonal()
    assert np.allclose(oracle.schur_setup(A, B)[0], np.einsum("ri,i,ri->r", Bd, D, Bd), rtol=1e-13)
    K = sp.bmat([[A.to_scipy(), B.to_scipy().T], [B.to_scipy(), None]], format="csc")
    rhs = np.concatenate([f, 1e-3 * rng.standard_normal(m)])
    x, info = oracle.fgmres(A, rhs, B=B, pc_type=oracle.PC_SCHUR, schur_fact=fact, rtol=1e-11, max_it=5000)
    assert info["reason"] == 2 and relerr(x, sla.spsolve(K, rhs)) < 1e-8
