"""General sparse constraint block A10: the reference declares B as a general
Mat (MatSetSizes(B, 4, nCols), SaddlePointProblem.c:45-53); beyond the build-defined 4 (2-D) / 6 (3-D) long
rows the library takes any CSR block -- thousands of short rows (CSR stream kernel) with a few long ones
(column-window kernel), S^ = diag(B D B^T) row by row.  Against the CPU oracle (parity unpinned)."""
import threading

import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu
KERNEL_TOL = 1e-13


def _random_block(spk, n, m, rng, long_rows=()):
    rows = []
    for r in range(m):
        k = int(n * 0.55) if r in long_rows else int(rng.integers(3, 41))
        cols = np.sort(rng.choice(n, k, replace=False)).astype(np.int32)
        rows.append((cols, rng.standard_normal(k) * (0.02 if r in long_rows else 1.0)))
    rp = np.concatenate([[0], np.cumsum([len(c) for c, _ in rows])]).astype(np.int32)
    return spk.CSR(rp, np.concatenate([c for c, _ in rows]), np.concatenate([v for _, v in rows]), n)


def test_random_sparse_block_m500(spk, oracle):
    """500 short rows (3..40 entries) + 2 long rows (13 500 entries: beyond the 8192 that send a row to the
    window kernel) on the 128 x 96 grid: operator, S^, all four factorisations, FGMRES."""
    mx, my = 128, 96
    A, f = spk.AssembleOperator_Laplace(mx, my)
    n = A.nrows
    rng = np.random.default_rng(2024)
    m = 502
    B = _random_block(spk, n, m, rng, long_rows=(17, 333))
    g = rng.standard_normal(m) * 1e-3
    rhs = np.concatenate([f, g])
    x = rng.uniform(-1, 1, n + m)
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        c.set_block(spk.BLOCK_A10, B)
        assert c.sizes()["m"] == m
        assert relerr(c.mult(x), oracle.apply_K(A, B, x)) < KERNEL_TOL
        for fact in range(4):
            c.pc_setup(spk.PC_SCHUR, fact)
            assert relerr(c.schur_diag(), oracle.schur_setup(A, B)[0]) < KERNEL_TOL
            assert relerr(c.pc_apply(x), oracle.pc_apply(A, B, oracle.PC_SCHUR, fact, x)) < KERNEL_TOL
        c.pc_setup(spk.PC_JACOBI)
        assert relerr(c.pc_apply(x), oracle.pc_apply(A, B, oracle.PC_JACOBI, 0, x)) < KERNEL_TOL
        c.pc_setup(spk.PC_SCHUR, spk.SCHUR_FULL)
        assert c.bd_planes() == 0                                     # no dense planes for a general block
        xt, it = c.fgmres(rhs, rtol=0.0, abstol=0.0, max_it=60)
        xs, info = c.fgmres(rhs, rtol=1e-8, max_it=4000)
        kx = c.mult(xs)
    _, io = oracle.fgmres(A, rhs, B=B, pc_type=oracle.PC_SCHUR, schur_fact=3, rtol=0.0, abstol=0.0, max_it=60)
    assert it["its"] == io["its"] == 60 and np.allclose(it["history"], io["history"], rtol=1e-6)
    xo, ioc = oracle.fgmres(A, rhs, B=B, pc_type=oracle.PC_SCHUR, schur_fact=3, rtol=1e-8, max_it=4000, threads=8)
    assert info["reason"] == ioc["reason"]
    if info["reason"] == 2:
        assert abs(info["its"] - ioc["its"]) <= max(2, ioc["its"] // 50) and relerr(xs, xo) < 1e-6
        assert np.linalg.norm(rhs - kx) <= 1.0001e-8 * np.linalg.norm(rhs)
    assert np.linalg.norm(rhs - oracle.apply_K(A, B, xs)) == pytest.approx(info["rnorm"], rel=1e-6)


def _div_block(spk, grid, b=0, e=None):
    Bm, g = spk.AssembleOperator_Constraints3D(*grid, b, e)
    Bd = spk.AssembleOperator_Divergence3D(*grid, b, e)
    return spk.CSR.vstack([Bm, Bd]), np.concatenate([g, np.zeros(Bd.nrows)])


@pytest.mark.parametrize("fact", [1, 3])
def test_divergence_block_3d(spk, oracle, fact):
    """ex42-style block on the 3-D grid: 6 long mean / moment rows + one divergence row per hexahedron
    (792 rows of <= 24 entries).  With Dirichlet data on every face the divergence rows carry the constant-
    pressure mode, so K is singular but consistent (g = 0 on those rows): a fixed number of iterations is
    compared with the oracle, and the true residual with the recurrence."""
    grid = (12, 10, 9)
    A, f = spk.AssembleOperator_Laplace3D(*grid)
    B, g = _div_block(spk, grid)
    Ao = oracle.CSR(A.rowptr, A.colidx, A.val, A.ncols)
    Bo = oracle.CSR(B.rowptr, B.colidx, B.val, B.ncols)
    assert np.array_equal(oracle.assemble_divergence3d(*grid).val, B.val[B.rowptr[6]:])     # generator parity, bitwise
    n, m = A.nrows, B.nrows
    assert m == 6 + 11 * 9 * 8
    rhs = np.concatenate([f, g])
    x = np.random.default_rng(3).uniform(-1, 1, n + m)
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        c.set_block(spk.BLOCK_A10, B)
        assert relerr(c.mult(x), oracle.apply_K(Ao, Bo, x)) < KERNEL_TOL
        c.pc_setup(spk.PC_SCHUR, fact)
        assert relerr(c.schur_diag(), oracle.schur_setup(Ao, Bo)[0]) < KERNEL_TOL
        assert relerr(c.pc_apply(x), oracle.pc_apply(Ao, Bo, oracle.PC_SCHUR, fact, x)) < KERNEL_TOL
        xt, it = c.fgmres(rhs, rtol=0.0, abstol=0.0, max_it=50)
    _, io = oracle.fgmres(Ao, rhs, B=Bo, pc_type=oracle.PC_SCHUR, schur_fact=fact, rtol=0.0, abstol=0.0, max_it=50)
    assert it["its"] == 50 and np.allclose(it["history"], io["history"], rtol=1e-6)
    assert np.linalg.norm(rhs - oracle.apply_K(Ao, Bo, xt)) == pytest.approx(it["rnorm"], rel=1e-6)


def test_divergence_block_two_z_slabs(spk, oracle):
    """The same block column-partitioned over two z-slabs (logical ranks): every rank holds all 798 rows
    restricted to its node planes; the 798 multipliers are replicated; sums over ranks of 798 values."""
    grid = (12, 10, 9)
    A, f = spk.AssembleOperator_Laplace3D(*grid)
    B, g = _div_block(spk, grid)
    Ao = oracle.CSR(A.rowptr, A.colidx, A.val, A.ncols)
    Bo = oracle.CSR(B.rowptr, B.colidx, B.val, B.ncols)
    n, m = A.nrows, B.nrows
    rhs = np.concatenate([f, g])
    xin = np.concatenate([np.sin(0.37 * np.arange(n)), 0.5 + 0.01 * np.arange(m)])
    grp = spk.LocalGroup(2)
    out, errs = [None, None], []

    def work(r):
        try:
            b, e = spk.partition_slab3d(*grid, r, 2)
            As, fs = spk.AssembleOperator_Laplace3D(*grid, b, e)
            Bs, _ = _div_block(spk, grid, b, e)
            c = spk.Context(0)
            c.comm_init_local(grp, r)
            c.set_block(spk.BLOCK_A00, As)
            c.set_block(spk.BLOCK_A10, Bs)
            c.pc_setup(spk.PC_SCHUR, spk.SCHUR_FULL)
            xl = np.concatenate([xin[b:e], xin[n:]])
            y = c.mult(xl)
            z = c.pc_apply(xl)
            xs, info = c.fgmres(np.concatenate([fs, g]), rtol=0.0, abstol=0.0, max_it=50)
            out[r] = (b, e, y, z, xs, info)
            c.close()
        except Exception as ex:  # noqa: BLE001
            errs.append(ex)
            raise

    th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    grp.close()
    assert not errs, errs
    y = np.zeros(n + m); z = np.zeros(n + m); x = np.zeros(n + m)
    for (b, e, yr, zr, xr, info) in out:
        y[b:e], z[b:e], x[b:e] = yr[:e - b], zr[:e - b], xr[:e - b]
        y[n:], z[n:], x[n:] = yr[-m:], zr[-m:], xr[-m:]
        assert np.array_equal(info["history"], out[0][5]["history"]) and np.array_equal(xr[-m:], out[0][4][-m:])
    assert relerr(y, oracle.apply_K(Ao, Bo, xin)) < KERNEL_TOL
    assert relerr(z, oracle.pc_apply(Ao, Bo, oracle.PC_SCHUR, 3, xin)) < KERNEL_TOL
    _, io = oracle.fgmres(Ao, rhs, B=Bo, pc_type=oracle.PC_SCHUR, schur_fact=3, rtol=0.0, abstol=0.0, max_it=50)
    assert np.allclose(out[0][5]["history"], io["history"], rtol=1e-6)
    assert np.linalg.norm(rhs - oracle.apply_K(Ao, Bo, x)) == pytest.approx(out[0][5]["rnorm"], rel=1e-6)


def test_block_sizes_between_the_paths(spk, oracle):
    """m = 8 (the last size of the dense-plane path) and m = 9 (the first of the general one) on the same
    grid give the same kind of answer; m = 0 rows is a valid (empty) block."""
    mx, my = 30, 22
    A, f = spk.AssembleOperator_Laplace(mx, my)
    n = A.nrows
    rng = np.random.default_rng(5)
    for m in (8, 9):
        B = _random_block(spk, n, m, rng)
        rhs = np.concatenate([f, rng.standard_normal(m) * 1e-3])
        with spk.Context(0) as c:
            c.set_block(spk.BLOCK_A00, A)
            c.set_block(spk.BLOCK_A10, B)
            c.pc_setup(spk.PC_SCHUR, spk.SCHUR_FULL)
            xs, info = c.fgmres(rhs, rtol=1e-9, max_it=3000)
        xo, io = oracle.fgmres(A, rhs, B=B, pc_type=oracle.PC_SCHUR, schur_fact=3, rtol=1e-9, max_it=3000)
        assert info["reason"] == io["reason"] == 2 and abs(info["its"] - io["its"]) <= 2 and relerr(xs, xo) < 1e-7
