"""The oracle against the known-answer values the survey extracted from the
reference's own element code and against the committed
oracle fixtures.  CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as sla


def test_element_stiffness_known_answer(oracle, appendix_b):
    Ke_ref = np.array(appendix_b["Ke"])
    for h in (1.0 / 3.0, 1.0 / 256.0):          # h-independent in 2-D
        for (x0, y0) in ((0.0, 0.0), (0.25, 0.5)):
            xe = [x0, y0, x0, y0 + h, x0 + h, y0 + h, x0 + h, y0]
            Ke = oracle.element_stiffness(xe)
            assert np.abs(Ke - Ke_ref).max() < appendix_b["Ke_tol"]
            assert np.allclose(Ke, Ke.T, atol=1e-15)
            assert np.abs(Ke.sum(axis=1)).max() < 1e-12          # rigid translations
    ev = np.sort(np.linalg.eigvalsh(oracle.element_stiffness([0, 0, 0, 1, 1, 1, 1, 0])))
    assert np.allclose(ev, [0, 0, 0, 1, 1, 2, 2, 2], atol=1e-10)


def test_element_stiffness_truncated_gauss_point(oracle):
    # defect A2: the 11-digit abscissa leaves Ke diagonal = 1 + O(1e-12), not exactly 1
    Ke = oracle.element_stiffness([0, 0, 0, 1, 1, 1, 1, 0])
    assert 1e-14 < abs(Ke[0, 0] - 1.0) < 1e-11


def test_element_load_known_answer(oracle, appendix_b):
    h = 1.0 / 3.0
    Fe = oracle.element_load([0, 0, 0, h, h, h, h, 0])
    assert np.allclose(Fe, h * h / 4 * np.array(appendix_b["Fe_over_h2_quarter"]), rtol=1e-11)


@pytest.mark.parametrize("M", [4, 32, 33])
def test_global_known_answers(oracle, appendix_b, M):
    ref = appendix_b["grids"][str(M)]
    A, f = oracle.assemble(M)
    assert A.nrows == ref["rows"] and A.nnz == ref["nnz"] == 4 * (3 * M - 2) ** 2
    assert np.linalg.norm(f) == pytest.approx(ref["f_norm"], rel=1e-11)
    S = A.to_scipy()
    assert abs(S - S.T).max() < 1e-14                      # symmetric after MatZeroRowsColumns
    u = sla.spsolve(S.tocsc(), f)
    assert np.linalg.norm(u) == pytest.approx(ref["u_norm"], rel=1e-10)
    assert np.abs(u).max() == pytest.approx(ref["u_max"], rel=1e-10)
    d = S.diagonal().reshape(M, M, 2)
    assert np.allclose(d[1:-1, 1:-1], appendix_b["interior_diag"], atol=1e-10)
    assert np.all(d[0] == 1.0) and np.all(d[-1] == 1.0) and np.all(d[:, 0] == 1.0) and np.all(d[:, -1] == 1.0)
    fb = f.reshape(M, M, 2)
    assert np.all(fb[0] == 0) and np.all(fb[-1] == 0) and np.all(fb[:, 0] == 0) and np.all(fb[:, -1] == 0)
    h = 1.0 / (M - 1)
    assert np.allclose(fb[1:-1, 1:-1, 0], h * h, rtol=1e-11) and np.allclose(fb[1:-1, 1:-1, 1], 2 * h * h, rtol=1e-11)


def test_m4_solution_vector(oracle, appendix_b, golden_m4):
    A, f = oracle.assemble(4)
    assert np.array_equal(A.rowptr, golden_m4["rowptr"]) and np.array_equal(A.colidx, golden_m4["colidx"])
    assert np.array_equal(A.val, golden_m4["val"]) and np.array_equal(f, golden_m4["f"])   # bit-for-bit
    u = sla.spsolve(A.to_scipy().tocsc(), f)
    for idx, v in appendix_b["m4_solution"].items():
        assert u[int(idx)] == pytest.approx(v, abs=2e-12)
    assert np.allclose(u, golden_m4["u"], rtol=0, atol=1e-15)


def test_constraint_block_known_answers(oracle, appendix_b, golden_m32):
    ref = appendix_b["constraints_m32"]
    A, f = oracle.assemble(32)
    B, g = oracle.assemble_constraints(32)
    assert B.nrows == 4 and B.ncols == A.nrows and B.nnz == ref["nnz_B"]
    assert list(g) == ref["g"]
    shat, G = oracle.schur_setup(A, B)
    assert np.allclose(shat, ref["shat"], rtol=1e-8)
    assert np.array_equal(shat, golden_m32["shat"])
    # G against dense algebra
    Bd = B.to_scipy().toarray()
    Gd = Bd @ np.diag(1.0 / A.to_scipy().diagonal()) @ Bd.T
    assert np.allclose(G, Gd, rtol=1e-12, atol=1e-20)
    # Dirichlet columns dropped
    node = B.colidx // 2
    i, j = node % 32, node // 32
    assert i.min() == 1 and i.max() == 30 and j.min() == 1 and j.max() == 30
    K = sp.bmat([[A.to_scipy(), B.to_scipy().T], [B.to_scipy(), None]]).tocsc()
    sol = sla.spsolve(K, np.concatenate([f, g]))
    assert np.linalg.norm(sol[:-4]) == pytest.approx(ref["u_norm"], rel=1e-10)
    assert np.allclose(sol[-4:], ref["lambda"], rtol=1e-7)
    assert np.allclose(sol, golden_m32["saddle"], rtol=1e-9, atol=1e-13)


def test_rectangular_grid_and_no_bc(oracle):
    A, f = oracle.assemble(7, 5, bc=False)
    S = A.to_scipy()
    assert A.nnz == 4 * (3 * 7 - 2) * (3 * 5 - 2)
    assert abs(S - S.T).max() < 1e-14
    # without Dirichlet rows the operator annihilates rigid translations
    for c in range(2):
        t = np.zeros(A.nrows)
        t[c::2] = 1.0
        assert np.abs(S @ t).max() < 1e-12
    assert f.sum() == pytest.approx(3.0, rel=1e-11)      # integral of (1,2) over the unit square


def test_input_fixture_is_derived_not_transcribed():
    """tests/golden/appendix_b.json is the output of an independent exact derivation (sympy on the
    reference's element formulas + a numpy/scipy assembly), re-run here and compared."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "golden", "derive_appendix_b.py"), "--check"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "matches a fresh derivation" in out.stdout
