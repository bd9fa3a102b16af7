"""Parity of the HIP path (through the C ABI of libspk.so) against the CPU
oracle on the same inputs.  Needs a real MI355X: run with -m gpu.

Tolerances (FP64, stated per SURVEY.md section 8(c)):
  kernel parity    <= 1e-13 relative (summation order only); the A-block SpMV is
                   checked for BITWISE equality, its sums run in CSR order
  solution parity  <= 1e-8 relative when both solves run to rtol 1e-10
  iteration parity same count +-1, residual history within 1e-6 relative
"""
import threading

import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu

KERNEL_TOL = 1e-13


@pytest.fixture(scope="module")
def ctx(spk):
    c = spk.Context(0)
    yield c
    c.close()


def _x(n, seed=12345):
    return np.random.default_rng(seed).uniform(-1.0, 1.0, n)


# --------------------------------------------------------------------------- SpMV
@pytest.mark.parametrize("mx,my", [(4, 4), (32, 32), (33, 33), (64, 64), (50, 7), (256, 256)])
def test_spmv_A_block_bitwise(spk, oracle, mx, my):
    A, _ = spk.AssembleOperator_Laplace(mx, my)
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        for seed in (1, 2):
            x = _x(A.nrows, seed)
            y = c.mult(x)
            y_ref = oracle.spmv(A, x)
            assert np.array_equal(y, y_ref), f"max diff {np.abs(y - y_ref).max()}"


def test_spmv_irregular_rows(spk, oracle):
    """Ragged CSR: empty rows, rows longer than one LDS tile (long-row path),
    unsorted columns, tile boundaries at odd offsets."""
    rng = np.random.default_rng(7)
    n = 3000
    lens = rng.integers(0, 40, n)
    lens[5] = 0; lens[6] = 0; lens[100] = 5000; lens[101] = 4097; lens[2999] = 1
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    colidx = rng.integers(0, n, rowptr[-1]).astype(np.int32)
    val = rng.standard_normal(rowptr[-1])
    A = spk.CSR(rowptr, colidx, val, n)
    x = _x(n)
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        y = c.mult(x)
    y_ref = oracle.spmv(A, x)
    short = lens <= 4096
    assert np.array_equal(y[short], y_ref[short])                      # CSR-order sums: bitwise
    assert np.allclose(y[~short], y_ref[~short], rtol=1e-12, atol=1e-12)  # long rows: tree order


@pytest.mark.parametrize("bs,fmt,long_len", [(2, "bcsr2x2", 700), (3, "bcsr3x3", 300)])
def test_spmv_blocked_irregular(spk, oracle, bs, fmt, long_len):
    """The blocked copies on ragged block structure: block rows of 1..40 blocks in arbitrary column order, one block
    row longer than a tile (strided path: tree order), tiles cut at odd places -- SpMV bitwise equal to the oracle's
    CSR loop on every row a tile holds; the FP32 sweeps (3 x 3: from the single-precision planes, the long block row by
    its own CSR-order path) bit for bit."""
    rng = np.random.default_rng(11 + bs)
    nb = 900
    lens = rng.integers(1, 41, nb)
    lens[37] = long_len
    lens[899] = 1
    rp, ci, va = [0], [], []
    for br in range(nb):
        cols = rng.choice(nb, size=lens[br], replace=False)
        if br not in cols:
            cols[0] = br                       # a diagonal block in every block row (Jacobi needs the diagonal)
        blocks = rng.standard_normal((lens[br], bs, bs))
        for j, c in enumerate(cols):
            if c == br:
                blocks[j] += 50.0 * np.eye(bs)
        for r in range(bs):
            for j, c in enumerate(cols):
                ci.extend(range(bs * c, bs * c + bs))
                va.extend(blocks[j, r])
            rp.append(len(ci))
    A = spk.CSR(np.array(rp, np.int32), np.array(ci, np.int32), np.array(va), bs * nb)
    x = _x(bs * nb, 5)
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        assert c.spmv_info()["format"] == fmt
        y = c.mult(x)
        c.pc_setup(spk.PC_JACOBI, 0, inner_sweeps=3, inner_omega=0.7)
        z = c.pc_apply(x)
    y_ref = oracle.spmv(A, x)
    short = np.repeat(lens <= (512 if bs == 2 else 256), bs)
    assert np.array_equal(y[short], y_ref[short])
    assert np.allclose(y[~short], y_ref[~short], rtol=1e-12, atol=1e-12) and (~short).sum() == bs
    assert np.array_equal(z, oracle.pc_apply_inner(A, None, oracle.PC_JACOBI, 0, 3, 0.7, x))


def test_spmv_full_size_1024(spk, oracle):
    """BASELINE config grid 1024 x 1024 (2.1 M rows, 37.7 M stored non-zeros)."""
    A, _ = spk.AssembleOperator_Laplace(1024)
    x = np.sin(0.37 * np.arange(A.nrows))
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        y = c.mult(x)
        # linearity: A(2x + e) = 2 A x + A e
        e = _x(A.nrows, 3)
        assert relerr(c.mult(2 * x + e), 2 * y + c.mult(e)) < 1e-14
    assert np.array_equal(y, oracle.spmv(A, x))


# --------------------------------------------------------------------------- nest operator, preconditioners
@pytest.mark.parametrize("mx,my", [(5, 5), (32, 32), (40, 23)])
def test_nest_mult_and_all_preconditioners(spk, oracle, mx, my):
    A, _ = spk.AssembleOperator_Laplace(mx, my)
    B, _ = spk.AssembleOperator_Constraints(mx, my)
    x = _x(A.nrows + 4)
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        c.set_block(spk.BLOCK_A10, B)
        assert relerr(c.mult(x), oracle.apply_K(A, B, x)) < KERNEL_TOL
        c.pc_setup(spk.PC_JACOBI)
        assert np.array_equal(c.jacobi_diag(), oracle.jacobi_dinv(A))
        assert relerr(c.pc_apply(x), oracle.pc_apply(A, B, oracle.PC_JACOBI, 0, x)) < KERNEL_TOL
        c.pc_setup(spk.PC_NONE)
        assert np.array_equal(c.pc_apply(x), x)
        for fact in range(4):
            c.pc_setup(spk.PC_SCHUR, fact)
            assert relerr(c.schur_diag(), oracle.schur_setup(A, B)[0]) < KERNEL_TOL
            assert relerr(c.pc_apply(x), oracle.pc_apply(A, B, oracle.PC_SCHUR, fact, x)) < KERNEL_TOL


def test_schur_diag_known_answer(spk, appendix_b):
    A, _ = spk.AssembleOperator_Laplace(32)
    B, _ = spk.AssembleOperator_Constraints(32)
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        c.set_block(spk.BLOCK_A10, B)
        c.pc_setup(spk.PC_SCHUR, spk.SCHUR_FULL)
        assert np.allclose(c.schur_diag(), appendix_b["constraints_m32"]["shat"], rtol=1e-8)


# --------------------------------------------------------------------------- Gram-Schmidt kernels
@pytest.mark.parametrize("n,nv", [(1, 1), (7, 3), (2047, 8), (2048, 9), (4099, 17), (100000, 31), (300001, 40), (65536, 0),
                                  (140001, 5), (262144, 13), (600000, 22), (1200001, 33)])
def test_mdot_maxpy(ctx, n, nv):
    """Every form of the two kernels: wave-split MDOT with 2 / 4 / 8 double2 per lane and 4 / 8 / 12
    vectors per wave, thin-workgroup MAXPY, the streaming forms of both (>= 1 M entries), ragged tails."""
    rng = np.random.default_rng(n + nv)
    V = rng.standard_normal((max(nv, 1), n))[:nv]
    w = rng.standard_normal(n)
    h, ww = ctx.mdot(V.reshape(nv, n), w)
    assert np.allclose(h, V @ w, rtol=1e-12, atol=1e-12 * np.sqrt(n))
    assert ww == pytest.approx(w @ w, rel=1e-13)
    a = rng.standard_normal(nv)
    w2, nrm2 = ctx.maxpy(a, V.reshape(nv, n), w)
    ref = w + a @ V if nv else w
    assert relerr(w2, ref) < KERNEL_TOL
    assert nrm2 == pytest.approx(ref @ ref, rel=1e-13)


# --------------------------------------------------------------------------- FGMRES
def _solve_both(spk, oracle, A, B, rhs, pc, fact=3, **kw):
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        if B is not None:
            c.set_block(spk.BLOCK_A10, B)
        c.pc_setup(pc, fact)
        x, info = c.fgmres(rhs, **kw)
    xo, io = oracle.fgmres(A, rhs, B=B, pc_type=pc, schur_fact=fact, **kw)
    return x, info, xo, io


def _check_iteration_parity(info, io, tight=20):
    """Same reason, iteration count +-1, residual history within 1e-6 relative
    over the first `tight` iterations.  Beyond that, classical Gram-Schmidt
    WITHOUT refinement (PETSc's default, kept here) amplifies summation-order
    differences: the oracle run with 1 vs 3 OpenMP threads already differs by
    14 % in the late history of the DIAG factorisation (parity
    notes), so later entries are only required to stay within a factor 2."""
    assert info["reason"] == io["reason"]
    assert abs(info["its"] - io["its"]) <= 1
    k = min(len(info["history"]), len(io["history"]))
    t = min(k, tight + 1)
    assert np.allclose(info["history"][:t], io["history"][:t], rtol=1e-6)
    ratio = info["history"][:k] / io["history"][:k]
    assert ratio.min() > 0.5 and ratio.max() < 2.0


def test_fgmres_jacobi_matches_oracle_and_fixture(spk, oracle, golden_m32):
    A, f = spk.AssembleOperator_Laplace(32)
    x, info, xo, io = _solve_both(spk, oracle, A, None, f, spk.PC_JACOBI, rtol=1e-5)
    _check_iteration_parity(info, io, tight=75)
    assert info["its"] == 75 and info["reason"] == 2
    assert np.allclose(info["history"], golden_m32["jacobi_hist"], rtol=1e-6)
    assert relerr(x, xo) < 1e-8
    # head-kernel path (default) against the step-by-step PCApply / MatMult / VecScale path
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        c.pc_setup(spk.PC_JACOBI)
        xu, iu = c.fgmres(f, rtol=1e-5, fused=0)
    assert iu["its"] == info["its"] and np.allclose(iu["history"], info["history"], rtol=1e-9) and relerr(xu, x) < 1e-10


@pytest.mark.parametrize("fact", [0, 1, 2, 3])
def test_fgmres_saddle_schur(spk, oracle, golden_m32, fact):
    A, f = spk.AssembleOperator_Laplace(32)
    B, g = spk.AssembleOperator_Constraints(32)
    rhs = np.concatenate([f, g])
    x, info, xo, io = _solve_both(spk, oracle, A, B, rhs, spk.PC_SCHUR, fact, rtol=1e-10)
    _check_iteration_parity(info, io)
    assert relerr(x, xo) < 1e-8
    assert relerr(x, golden_m32["saddle"]) < 1e-8                       # scipy sparse LU
    assert np.allclose(x[-4:], golden_m32["saddle"][-4:], rtol=1e-7)


@pytest.mark.parametrize("fact", [1, 3])
@pytest.mark.parametrize("mx,my", [(32, 32), (45, 18)])
def test_fused_schur_path_equals_unfused(spk, oracle, fact, mx, my):
    """opts.fused = 1 (default: VecScale + Schur PC + B^T product in one pass, B D w' inside
    the MAXPY pass, w1 = t - G y1) against the step-by-step PCApply / MatMult path."""
    A, f = spk.AssembleOperator_Laplace(mx, my)
    B, g = spk.AssembleOperator_Constraints(mx, my)
    rhs = np.concatenate([f, g])
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        c.set_block(spk.BLOCK_A10, B)
        c.pc_setup(spk.PC_SCHUR, fact)
        xf, inf_ = c.fgmres(rhs, rtol=1e-10, fused=1)
        xu, inu = c.fgmres(rhs, rtol=1e-10, fused=0)
    xo, io = oracle.fgmres(A, rhs, B=B, pc_type=oracle.PC_SCHUR, schur_fact=fact, rtol=1e-10)
    assert inf_["reason"] == inu["reason"] == 2
    assert abs(inf_["its"] - inu["its"]) <= 1 and abs(inf_["its"] - io["its"]) <= 1
    assert np.allclose(inf_["history"][:21], inu["history"][:21], rtol=1e-9)
    assert relerr(xf, xu) < 1e-8 and relerr(xf, xo) < 1e-8


@pytest.mark.parametrize("kw", [dict(orthog=1), dict(cgs_refine=2), dict(cgs_refine=1)])
@pytest.mark.parametrize("fused", [0, 1])
def test_orthogonalisation_options(spk, oracle, kw, fused):
    """-ksp_gmres_modifiedgramschmidt and -ksp_gmres_cgs_refinement_type {ifneeded,always}."""
    A, f = spk.AssembleOperator_Laplace(32)
    B, g = spk.AssembleOperator_Constraints(32)
    rhs = np.concatenate([f, g])
    okw = dict(orthog=kw.get("orthog", 0), refine=kw.get("cgs_refine", 0))
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        c.set_block(spk.BLOCK_A10, B)
        c.pc_setup(spk.PC_SCHUR, spk.SCHUR_FULL)
        x, info = c.fgmres(rhs, rtol=1e-10, fused=fused, **kw)
    xo, io = oracle.fgmres(A, rhs, B=B, pc_type=oracle.PC_SCHUR, schur_fact=3, rtol=1e-10, **okw)
    _check_iteration_parity(info, io)
    assert relerr(x, xo) < 1e-8


@pytest.mark.parametrize("fact", [1, 3])
def test_single_reduction_gram_schmidt(spk, oracle, fact):
    """opts.single_reduce = 1: one reduction per iteration (||w'||^2 = w.w - |h|^2, B D w' by
    recurrence) against the two-reduction fused path and the oracle."""
    A, f = spk.AssembleOperator_Laplace(40, 28)
    B, g = spk.AssembleOperator_Constraints(40, 28)
    rhs = np.concatenate([f, g])
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        c.set_block(spk.BLOCK_A10, B)
        c.pc_setup(spk.PC_SCHUR, fact)
        x1, i1 = c.fgmres(rhs, rtol=1e-10, single_reduce=1)
        x2, i2 = c.fgmres(rhs, rtol=1e-10, single_reduce=0)
    xo, io = oracle.fgmres(A, rhs, B=B, pc_type=oracle.PC_SCHUR, schur_fact=fact, rtol=1e-10)
    assert i1["reason"] == i2["reason"] == 2
    assert abs(i1["its"] - i2["its"]) <= 1 and abs(i1["its"] - io["its"]) <= 1
    assert np.allclose(i1["history"][:21], i2["history"][:21], rtol=1e-5)
    assert relerr(x1, x2) < 1e-8 and relerr(x1, xo) < 1e-8


@pytest.mark.parametrize("restart", [1, 2, 7, 45, 62])
@pytest.mark.parametrize("single", [0, 1])
def test_restart_lengths(spk, oracle, restart, single):
    """-ksp_gmres_restart from 1 to the 62 the kernels allow: beyond 40 basis vectors MDot runs as two
    launches; the single-reduction route handles first / last iterations of a cycle differently."""
    A, f = spk.AssembleOperator_Laplace(20, 18)
    B, g = spk.AssembleOperator_Constraints(20, 18)
    rhs = np.concatenate([f, g])
    rtol = 1e-9 if restart >= 7 else 1e-3          # short cycles stagnate: a loose target keeps the run short
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        c.set_block(spk.BLOCK_A10, B)
        c.pc_setup(spk.PC_SCHUR, spk.SCHUR_FULL)
        x, info = c.fgmres(rhs, restart=restart, rtol=rtol, max_it=4000, single_reduce=single)
    xo, io = oracle.fgmres(A, rhs, B=B, pc_type=oracle.PC_SCHUR, schur_fact=3, restart=restart, rtol=rtol, max_it=4000)
    assert info["reason"] == io["reason"]
    if info["reason"] == 2:
        # single reduction: once ||w'||^2 = w.w - |h|^2 drops into rounding noise (a 724-row system with a
        # 45-vector basis gets there) the kernel keeps a conservative floor, so convergence may only be
        # confirmed by the true residual of the next restart -- later, never earlier, than the oracle
        slack = restart if single else max(2, io["its"] // 50)
        assert -2 <= info["its"] - io["its"] <= slack
        assert relerr(x, xo) < (1e-6 if restart >= 7 else 1e-2)
    r = np.linalg.norm(rhs - oracle.apply_K(A, B, x))
    assert r == pytest.approx(info["rnorm"], rel=1e-6)      # the recurrence estimate is the true residual


@pytest.mark.parametrize("restart,orthog", [(63, 0), (100, 0), (200, 1), (500, 0)])
def test_long_restart_cycles(spk, oracle, restart, orthog):
    """-ksp_gmres_restart beyond the 62 of the fused kernels (PETSc takes any length through KSPSetFromOptions,
    SaddlePointProblem.c:67): Gram-Schmidt in chunks of 40 vectors, the Givens step and the back substitution from their
    large forms.  One cycle can hold the whole solve here (restart 500 > iterations)."""
    A, f = spk.AssembleOperator_Laplace(24, 20)
    B, g = spk.AssembleOperator_Constraints(24, 20)
    rhs = np.concatenate([f, g])
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        c.set_block(spk.BLOCK_A10, B)
        c.pc_setup(spk.PC_SCHUR, spk.SCHUR_FULL)
        x, info = c.fgmres(rhs, restart=restart, rtol=1e-10, max_it=4000, orthog=orthog)
        # round 3: a long restart keeps the head kernel (VecScale + PCApply + B^T part in one pass, B D w' out of the last
        # MAXPY chunk); only the Givens step is a launch of its own.  fused=0: the step-by-step launches, the same iterates
        assert c.iteration_form()[0] == 1
        xs, infos = c.fgmres(rhs, restart=restart, rtol=1e-10, max_it=4000, orthog=orthog, fused=0)
        assert c.iteration_form()[0] == -1
        ks = min(len(info["history"]), len(infos["history"]), 150) - 1
        assert abs(info["its"] - infos["its"]) <= 2 and np.allclose(info["history"][:ks], infos["history"][:ks], rtol=1e-5)
        assert relerr(x, xs) < 1e-7
        refined = {}
        if orthog == 0 and restart <= 200:
            # -ksp_gmres_cgs_refinement_type on a long restart (round 3: the second pass runs in the same chunks)
            for mode in (1, 2):
                refined[mode] = c.fgmres(rhs, restart=restart, rtol=1e-10, max_it=4000, cgs_refine=mode)
    for mode, (xr, ir) in refined.items():
        xro, iro = oracle.fgmres(A, rhs, B=B, pc_type=oracle.PC_SCHUR, schur_fact=3, restart=restart, rtol=1e-10, max_it=4000,
                                 refine=mode)
        assert ir["reason"] == iro["reason"] == 2 and abs(ir["its"] - iro["its"]) <= 2, mode
        kk = min(len(ir["history"]), len(iro["history"]), 150) - 1
        assert np.allclose(ir["history"][:kk], iro["history"][:kk], rtol=1e-5), mode
        assert relerr(xr, xro) < 1e-7
    xo, io = oracle.fgmres(A, rhs, B=B, pc_type=oracle.PC_SCHUR, schur_fact=3, restart=restart, rtol=1e-10, max_it=4000,
                           orthog=orthog)
    assert info["reason"] == io["reason"] == 2 and abs(info["its"] - io["its"]) <= 2
    k = min(len(info["history"]), len(io["history"]), 150) - 1
    assert np.allclose(info["history"][:k], io["history"][:k], rtol=1e-5)
    assert relerr(x, xo) < 1e-7
    r = np.linalg.norm(rhs - oracle.apply_K(A, B, x))
    assert r <= 1.01e-10 * np.linalg.norm(rhs) and r == pytest.approx(info["rnorm"], rel=1e-4)


def test_long_restart_jacobi_head_path(spk, oracle):
    """The same on the Jacobi head path (K = A, the reference as written, SaddlePointProblem.c:66): restart 90."""
    A, f = spk.AssembleOperator_Laplace(40, 28)
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        c.pc_setup(spk.PC_JACOBI)
        x, info = c.fgmres(f, restart=90, rtol=1e-10, max_it=4000)
        assert c.iteration_form()[0] == 1
        xr, ir = c.fgmres(f, restart=90, rtol=1e-10, max_it=4000, cgs_refine=1)
    xo, io = oracle.fgmres(A, f, pc_type=oracle.PC_JACOBI, restart=90, rtol=1e-10, max_it=4000)
    xro, iro = oracle.fgmres(A, f, pc_type=oracle.PC_JACOBI, restart=90, rtol=1e-10, max_it=4000, refine=1)
    for (xa, ia), (xb, ib) in (((x, info), (xo, io)), ((xr, ir), (xro, iro))):
        assert ia["reason"] == ib["reason"] == 2 and abs(ia["its"] - ib["its"]) <= 2
        k = min(len(ia["history"]), len(ib["history"]), 150) - 1
        assert np.allclose(ia["history"][:k], ib["history"][:k], rtol=1e-5)
        assert relerr(xa, xb) < 1e-7


def test_single_reduction_jacobi_head_path(spk, oracle):
    """The same single-reduction route on the Jacobi head path (K = A, the reference as written):
    ||w'||^2 = w.w - |h|^2, MAXPY + VecScale + PCApply_Jacobi + Givens in one launch."""
    A, f = spk.AssembleOperator_Laplace(40, 28)
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        c.pc_setup(spk.PC_JACOBI)
        x1, i1 = c.fgmres(f, rtol=1e-10, single_reduce=1)
        x2, i2 = c.fgmres(f, rtol=1e-10, single_reduce=0)
        x3, i3 = c.fgmres(f, rtol=1e-10, single_reduce=1, restart=1)     # degenerate cycle: no fused launch at all
    xo, io = oracle.fgmres(A, f, pc_type=oracle.PC_JACOBI, rtol=1e-10)
    assert i1["reason"] == i2["reason"] == 2
    assert abs(i1["its"] - i2["its"]) <= 1 and abs(i1["its"] - io["its"]) <= 1
    assert np.allclose(i1["history"][:21], i2["history"][:21], rtol=1e-5)
    assert relerr(x1, x2) < 1e-8 and relerr(x1, xo) < 1e-8
    assert i3["reason"] in (2, -3)


@pytest.mark.parametrize("pc,fact", [("jacobi", 0), ("schur", 0), ("schur", 1), ("schur", 2), ("schur", 3)])
def test_fp32_inner_solve(spk, oracle, pc, fact):
    """BASELINE config 5's "mixed FP32 inner solve": k damped-Jacobi Richardson sweeps on A in single
    precision stand for diag(A)^-1 in the preconditioner.  PC application against the oracle's float
    restatement (same operation order: expected bitwise on the u part), then the outer FP64 FGMRES."""
    A, f = spk.AssembleOperator_Laplace(32, 27)
    B, g = spk.AssembleOperator_Constraints(32, 27)
    saddle = pc == "schur"
    rhs = np.concatenate([f, g]) if saddle else f
    pct = spk.PC_SCHUR if saddle else spk.PC_JACOBI
    opc = oracle.PC_SCHUR if saddle else oracle.PC_JACOBI
    x = _x(len(rhs), 4)
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        if saddle:
            c.set_block(spk.BLOCK_A10, B)
        c.pc_setup(pct, fact, inner_sweeps=3, inner_omega=0.8)
        z = c.pc_apply(x)
        sol, info = c.fgmres(rhs, rtol=1e-9)
    zo = oracle.pc_apply_inner(A, B if saddle else None, opc, fact, 3, 0.8, x)
    assert relerr(z, zo) < 1e-6                       # FP32 arithmetic; wide B products differ in order
    if not saddle or fact == 0:
        assert np.array_equal(z[:A.nrows], zo[:A.nrows])   # pure inner solve: same float operations
    so, io = oracle.fgmres(A, rhs, B=B if saddle else None, pc_type=opc, schur_fact=fact, rtol=1e-9,
                           inner_its=3, inner_omega=0.8)
    assert info["reason"] == io["reason"] == 2 and abs(info["its"] - io["its"]) <= 2
    assert relerr(sol, so) < 1e-7
    # and it does what it is for: far fewer outer iterations than plain diag(A)^-1
    _, plain = oracle.fgmres(A, rhs, B=B if saddle else None, pc_type=opc, schur_fact=fact, rtol=1e-9)
    assert info["its"] < 0.75 * plain["its"]


def _six_row_constraints(spk, mx, my):
    """4 build-defined rows + 2 more (x-moment of Uy, y-moment of Ux): exercises the m in 5..8 kernels."""
    B, g = spk.AssembleOperator_Constraints(mx, my)
    rows = [(B.colidx[B.rowptr[r]:B.rowptr[r + 1]], B.val[B.rowptr[r]:B.rowptr[r + 1]]) for r in range(4)]
    hx, hy = 1.0 / (mx - 1), 1.0 / (my - 1)
    c1, _ = rows[1]; node = c1 // 2
    rows.append((c1, hx * hy * ((node % mx) * hx - 0.5)))            # x-moment of Uy
    c0, _ = rows[0]; node = c0 // 2
    rows.append((c0, hx * hy * ((node // mx) * hy - 0.5)))           # y-moment of Ux
    rp = np.concatenate([[0], np.cumsum([len(c) for c, _ in rows])]).astype(np.int32)
    B6 = spk.CSR(rp, np.concatenate([c for c, _ in rows]), np.concatenate([v for _, v in rows]), B.ncols)
    return B6, np.concatenate([g, [2e-3, -1e-3]])


@pytest.mark.parametrize("fact,fused", [(3, 1), (3, 0), (1, 1), (2, 0)])
def test_six_constraint_rows(spk, oracle, fact, fused):
    A, f = spk.AssembleOperator_Laplace(30, 22)
    B6, g6 = _six_row_constraints(spk, 30, 22)
    rhs = np.concatenate([f, g6])
    x = _x(len(rhs), 6)
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        c.set_block(spk.BLOCK_A10, B6)
        c.pc_setup(spk.PC_SCHUR, fact)
        assert relerr(c.mult(x), oracle.apply_K(A, B6, x)) < KERNEL_TOL
        assert relerr(c.pc_apply(x), oracle.pc_apply(A, B6, oracle.PC_SCHUR, fact, x)) < KERNEL_TOL
        sol, info = c.fgmres(rhs, rtol=1e-10, fused=fused)
    so, io = oracle.fgmres(A, rhs, B=B6, pc_type=oracle.PC_SCHUR, schur_fact=fact, rtol=1e-10)
    _check_iteration_parity(info, io)
    assert relerr(sol, so) < 1e-8


def test_fgmres_rtol_1e8_iteration_counts(spk, golden_m32):
    A, f = spk.AssembleOperator_Laplace(32)
    B, g = spk.AssembleOperator_Constraints(32)
    rhs = np.concatenate([f, g])
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        c.set_block(spk.BLOCK_A10, B)
        for fact in range(4):
            c.pc_setup(spk.PC_SCHUR, fact)
            _, info = c.fgmres(rhs, rtol=1e-8)
            assert abs(info["its"] - golden_m32["schur_its"][fact]) <= 1 and info["reason"] == 2


def test_fgmres_host_check_cadence_does_not_change_the_iterate(spk):
    A, f = spk.AssembleOperator_Laplace(24)
    B, g = spk.AssembleOperator_Constraints(24)
    rhs = np.concatenate([f, g])
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        c.set_block(spk.BLOCK_A10, B)
        c.pc_setup(spk.PC_SCHUR, spk.SCHUR_FULL)
        x0, i0 = c.fgmres(rhs, rtol=1e-9, check_every=0)
        x1, i1 = c.fgmres(rhs, rtol=1e-9, check_every=1)
        x7, i7 = c.fgmres(rhs, rtol=1e-9, check_every=7)
    assert i0["its"] == i1["its"] == i7["its"] and i0["reason"] == 2
    assert np.array_equal(x0, x1) and np.array_equal(x0, x7)           # deterministic, bit for bit
    assert np.array_equal(i0["history"], i1["history"])


def test_fgmres_edge_cases(spk, oracle):
    A, f = spk.AssembleOperator_Laplace(9)
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        c.pc_setup(spk.PC_JACOBI)
        x, info = c.fgmres(np.zeros_like(f))
        assert info["its"] == 0 and info["reason"] == 3 and not x.any()
        c.pc_setup(spk.PC_NONE)
        x, info = c.fgmres(f, max_it=3, rtol=1e-14)
        xo, io = oracle.fgmres(A, f, pc_type=oracle.PC_NONE, max_it=3, rtol=1e-14)
        assert info["its"] == 3 and info["reason"] == -3 and relerr(x, xo) < 1e-10
        x, info = c.fgmres(f, max_it=0)
        assert info["its"] == 0 and info["reason"] == -3
        c.pc_setup(spk.PC_JACOBI)
        x, info = c.fgmres(f, restart=5, rtol=1e-10)
        xo, io = oracle.fgmres(A, f, pc_type=oracle.PC_JACOBI, restart=5, rtol=1e-10)
        assert info["reason"] == 2 and abs(info["its"] - io["its"]) <= 1 and relerr(x, xo) < 1e-8
        assert info["cycles"] >= 2
        # non-zero initial guess
        x, info = c.fgmres(f, x0=xo, rtol=1e-8)
        assert info["its"] == 0 and info["reason"] in (2, 3)
        x, info = c.fgmres(f, x0=0.5 * xo, rtol=1e-9)
        xo2, io2 = oracle.fgmres(A, f, x0=0.5 * xo, pc_type=oracle.PC_JACOBI, rtol=1e-9)
        assert abs(info["its"] - io2["its"]) <= 1 and relerr(x, xo2) < 1e-8
        b = f.copy(); b[3] = np.nan
        x, info = c.fgmres(b)
        assert info["reason"] == -9


def test_fgmres_invariant_subspace_and_bad_data(spk, oracle):
    """K = 2 I: the Krylov space is invariant after one step (w' vanishes, PETSc's happy-breakdown
    corner); NaN inside the operator; a singular (zero) operator."""
    n = 8
    A = spk.CSR(np.arange(n + 1), np.arange(n), 2 * np.ones(n), n)
    b = np.arange(1, n + 1, dtype=float)
    for pc in (spk.PC_NONE, spk.PC_JACOBI):
        with spk.Context(0) as c:
            c.set_block(spk.BLOCK_A00, A)
            c.pc_setup(pc)
            x, info = c.fgmres(b, rtol=1e-12)
        xo, io = oracle.fgmres(oracle.CSR(A.rowptr, A.colidx, A.val, n), b, pc_type=pc, rtol=1e-12)
        assert info["its"] == io["its"] == 1 and info["reason"] == io["reason"] == 2
        assert np.allclose(x, b / 2, rtol=1e-14)
    An = spk.CSR(A.rowptr, A.colidx, np.where(np.arange(n) == 3, np.nan, 2.0), n)
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, An)
        c.pc_setup(spk.PC_NONE)
        _, info = c.fgmres(b)
    assert info["reason"] == -9                                          # KSP_DIVERGED_NANORINF
    Az = spk.CSR(A.rowptr, A.colidx, np.zeros(n), n)
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, Az)
        c.pc_setup(spk.PC_JACOBI)                                        # zero diagonal -> 1 (PCJACOBI)
        assert np.array_equal(c.jacobi_diag(), np.ones(n))
        _, info = c.fgmres(b, max_it=5)
    _, io = oracle.fgmres(oracle.CSR(Az.rowptr, Az.colidx, Az.val, n), b, pc_type=oracle.PC_JACOBI, max_it=5)
    assert info["reason"] == io["reason"] and info["reason"] < 0 and info["its"] == io["its"]


def test_fgmres_config3_512_truncated(spk, oracle):
    """BASELINE config 3 (512 x 512, full Schur path): 45 iterations of both
    implementations must agree (the oracle finishes this in seconds)."""
    A, f = spk.AssembleOperator_Laplace(512)
    B, g = spk.AssembleOperator_Constraints(512)
    rhs = np.concatenate([f, g])
    x, info, xo, io = _solve_both(spk, oracle, A, B, rhs, spk.PC_SCHUR, 3, rtol=1e-30, max_it=45)
    assert info["its"] == io["its"] == 45 and info["reason"] == io["reason"] == -3
    assert np.allclose(info["history"], io["history"], rtol=1e-7)
    # mid-solve iterates agree in the residual, not digit for digit: x = x0 + Z H^-1 g is
    # ill-conditioned in directions the residual barely sees (measured 6e-7 here)
    assert relerr(x, xo) < 1e-5
    K = oracle.apply_K(A, B, x)
    assert np.linalg.norm(rhs - K) == pytest.approx(np.linalg.norm(rhs - oracle.apply_K(A, B, xo)), rel=1e-6)


def test_full_size_1024_residual_property(spk, oracle):
    """BASELINE bench workload (1024 x 1024 saddle system): after K iterations the residual norm
    the device reports (Givens recurrence) must equal the TRUE residual ||b - K x|| evaluated
    independently by the oracle -- a size-independent property; also linearity of K and M^-1."""
    A, f = spk.AssembleOperator_Laplace(1024)
    B, g = spk.AssembleOperator_Constraints(1024)
    rhs = np.concatenate([f, g])
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        c.set_block(spk.BLOCK_A10, B)
        c.pc_setup(spk.PC_SCHUR, spk.SCHUR_FULL)
        x, info = c.fgmres(rhs, rtol=0.0, abstol=0.0, max_it=45)          # 1.5 restart cycles
        xs, infos = c.fgmres(rhs, rtol=0.0, abstol=0.0, max_it=45, single_reduce=1)
        e = _x(len(rhs), 9)
        assert relerr(c.pc_apply(3 * rhs - e), 3 * c.pc_apply(rhs) - c.pc_apply(e)) < 1e-13
        kx = c.mult(x)
    assert info["its"] == 45 and info["reason"] == -3
    r_true = np.linalg.norm(rhs - oracle.apply_K(A, B, x))
    assert r_true == pytest.approx(info["rnorm"], rel=1e-8)
    # K x ~ b is the small remainder of large terms here (|lambda| ~ 10, B^T lambda cancels against A u):
    # the summation-order bar is relative to |K| |x|, not to the cancelled result
    absA = type(A)(A.rowptr, A.colidx, np.abs(A.val), A.ncols)
    absB = type(B)(B.rowptr, B.colidx, np.abs(B.val), B.ncols)
    assert np.linalg.norm(kx - oracle.apply_K(A, B, x)) < KERNEL_TOL * np.linalg.norm(oracle.apply_K(absA, absB, np.abs(x)))
    assert np.all(np.diff(info["history"][:31]) <= 1e-14)                 # monotone inside a cycle
    # opt-in single-reduction mode: ||w'||^2 = w.w - |h|^2 cancels (measured 5e-6 drift here)
    assert np.allclose(infos["history"], info["history"], rtol=1e-4) and relerr(xs, x) < 1e-4


def test_converged_solve_config2_size(spk, oracle):
    """256 x 256 grid (BASELINE config 2 size) solved to rtol 1e-8 on the saddle system: the
    oracle needs ~1100 iterations; solution parity and iteration count within 1 %."""
    A, f = spk.AssembleOperator_Laplace(256)
    B, g = spk.AssembleOperator_Constraints(256)
    rhs = np.concatenate([f, g])
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        c.set_block(spk.BLOCK_A10, B)
        c.pc_setup(spk.PC_SCHUR, spk.SCHUR_FULL)
        x, info = c.fgmres(rhs, rtol=1e-8)
    xo, io = oracle.fgmres(A, rhs, B=B, pc_type=oracle.PC_SCHUR, schur_fact=3, rtol=1e-8, threads=8)
    assert info["reason"] == io["reason"] == 2
    assert abs(info["its"] - io["its"]) <= max(2, 0.01 * io["its"])
    assert relerr(x, xo) < 1e-6                       # both stop at rtol 1e-8: agreement to the tolerance
    assert np.linalg.norm(rhs - oracle.apply_K(A, B, x)) <= 1.0001e-8 * np.linalg.norm(rhs)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_spmv_random_csr(spk, oracle, seed):
    """Random rectangular-pattern CSR (no 2x2 structure -> CSR stream kernel): ragged rows,
    empty rows, duplicate columns, tile boundaries at arbitrary offsets."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 5000))
    lens = rng.integers(0, int(rng.integers(1, 60)), n)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    colidx = rng.integers(0, n, rowptr[-1]).astype(np.int32)
    val = rng.standard_normal(rowptr[-1])
    A = spk.CSR(rowptr, colidx, val, n)
    x = _x(n, seed)
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        assert np.array_equal(c.mult(x), oracle.spmv(A, x))


@pytest.mark.parametrize("n,m", [(1001, 0), (1001, 3), (777, 5), (4098, 4)])
def test_fgmres_general_matrix_odd_sizes(spk, oracle, n, m):
    """A general (non-grid) operator: random diagonally dominant CSR with ragged rows, an ODD number of
    rows (no 2x2 blocks, no fused head path: PCApply and MatMult as separate steps) and a random
    constraint block with m rows -- the solver is matrix-agnostic behind KSPSetOperators."""
    rng = np.random.default_rng(n + m)
    lens = rng.integers(1, 9, n)
    rowptr = np.concatenate([[0], np.cumsum(lens + 1)]).astype(np.int32)
    colidx = np.empty(rowptr[-1], np.int32)
    val = np.empty(rowptr[-1])
    for i in range(n):
        k0, k1 = rowptr[i], rowptr[i + 1]
        cols = np.sort(rng.choice(n, k1 - k0 - 1, replace=False))
        cols = np.sort(np.unique(np.concatenate([cols[cols != i], [i]])))
        cols = np.pad(cols, (0, k1 - k0 - len(cols)), mode="edge")            # duplicates allowed
        colidx[k0:k1] = cols
        v = rng.standard_normal(k1 - k0) * 0.3
        v[cols == i] = 0.0
        v[np.argmax(cols == i)] = 4.0 + np.abs(v).sum()
        val[k0:k1] = v
    A = spk.CSR(rowptr, colidx, val, n)
    B = None
    if m:
        Bd = np.where(rng.random((m, n)) < 0.4, rng.standard_normal((m, n)), 0.0)
        brp = np.concatenate([[0], np.cumsum((Bd != 0).sum(1))]).astype(np.int32)
        B = spk.CSR(brp, np.concatenate([np.nonzero(r)[0] for r in Bd]).astype(np.int32), Bd[Bd != 0], n)
    rhs = rng.standard_normal(n + m)
    Ao = oracle.CSR(A.rowptr, A.colidx, A.val, n)
    Bo = oracle.CSR(B.rowptr, B.colidx, B.val, n) if m else None
    pc, fact = (spk.PC_SCHUR, 3) if m else (spk.PC_JACOBI, 0)
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        assert c.spmv_info()["format"] == "csr"
        if m:
            c.set_block(spk.BLOCK_A10, B)
        c.pc_setup(pc, fact)
        x, info = c.fgmres(rhs, rtol=1e-10, max_it=3000)
    xo, io = oracle.fgmres(Ao, rhs, B=Bo, pc_type=pc, schur_fact=fact, rtol=1e-10, max_it=3000)
    assert info["reason"] == io["reason"] == 2 and abs(info["its"] - io["its"]) <= 2
    assert relerr(x, xo) < 1e-8
    Kx = oracle.apply_K(Ao, Bo, x) if m else oracle.spmv(Ao, x)
    assert np.linalg.norm(rhs - Kx) <= 2e-10 * np.linalg.norm(rhs)


def test_iteration_path_is_agreed_over_the_ranks(spk, oracle):
    """Two logical ranks, one with an ODD number of rows (501 | 500 of a general 1001-row operator): the head-kernel
    paths need an even local size, and the ranks' collective sequences differ between paths -- so the path is agreed at
    KSPSetUp (an all-gather of the local facts), not chosen per rank.  Before, rank 1 took the Jacobi head path and
    rank 0 the step-by-step one: mismatched all-reduces.  Result against the single-rank solve and the oracle."""
    n = 1001
    rng = np.random.default_rng(77)
    lens = rng.integers(2, 9, n)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    colidx = np.empty(rowptr[-1], np.int32)
    val = np.empty(rowptr[-1])
    for i in range(n):
        k0, k1 = rowptr[i], rowptr[i + 1]
        cols = np.sort(np.unique(np.concatenate([rng.choice(n, k1 - k0 - 1, replace=False), [i]])))[:k1 - k0]
        if i not in cols:
            cols[-1] = i
            cols = np.sort(cols)
        cols = np.pad(cols, (0, k1 - k0 - len(cols)), mode="edge")
        colidx[k0:k1] = cols
        v = rng.standard_normal(k1 - k0) * 0.3
        v[cols == i] = 0.0
        v[np.argmax(cols == i)] = 4.0 + np.abs(v).sum()
        val[k0:k1] = v
    rhs = rng.standard_normal(n)
    Ao = oracle.CSR(rowptr, colidx, val, n)
    xo, io = oracle.fgmres(Ao, rhs, pc_type=oracle.PC_JACOBI, rtol=1e-10, max_it=2000)
    cuts = [0, 501, n]
    grp = spk.LocalGroup(2)
    out, errs = [None, None], []

    def work(r):
        try:
            b, e = cuts[r], cuts[r + 1]
            k0, k1 = rowptr[b], rowptr[e]
            Ar = spk.CSR((rowptr[b:e + 1] - k0).astype(np.int32), colidx[k0:k1], val[k0:k1], n, row_begin=b)
            with spk.Context(0) as c:
                c.comm_init_local(grp, r)
                c.set_block(spk.BLOCK_A00, Ar)
                c.pc_setup(spk.PC_JACOBI, 0)
                x, info = c.fgmres(rhs[b:e], rtol=1e-10, max_it=2000)
                out[r] = (x, info, c.iteration_form()[0])
        except Exception as ex:  # noqa: BLE001
            errs.append(ex)
            raise
    th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    grp.close()
    assert not errs and all(not t.is_alive() for t in th), errs
    assert out[0][2] == out[1][2] == -1                                   # both on the step-by-step path
    assert np.array_equal(out[0][1]["history"], out[1][1]["history"])
    assert out[0][1]["reason"] == io["reason"] == 2 and abs(out[0][1]["its"] - io["its"]) <= 2
    assert relerr(np.concatenate([out[0][0], out[1][0]]), xo) < 1e-8


def test_device_resident_vectors(spk, oracle):
    """b and x already in HBM (SPK_MEM_DEVICE): same iterate as the host-pointer path."""
    A, f = spk.AssembleOperator_Laplace(32)
    B, g = spk.AssembleOperator_Constraints(32)
    rhs = np.concatenate([f, g])
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        c.set_block(spk.BLOCK_A10, B)
        c.pc_setup(spk.PC_SCHUR, spk.SCHUR_FULL)
        xh, ih = c.fgmres(rhs, rtol=1e-8)
        bd, xd = c.vec_create(rhs), c.vec_create(n=len(rhs))
        idv = c.fgmres_device(bd, xd, rtol=1e-8)
        x = c.vec_get(xd, len(rhs))
        c.vec_destroy(bd); c.vec_destroy(xd)
    assert idv["its"] == ih["its"] and np.array_equal(x, xh)


def test_error_reporting(spk):
    A, f = spk.AssembleOperator_Laplace(8)
    with spk.Context(0) as c:
        with pytest.raises(spk.SpkError, match="operator"):
            c.mult(np.zeros(0))
        c.set_block(spk.BLOCK_A00, A)
        with pytest.raises(spk.SpkError, match="pc_setup"):
            c.fgmres(f)
        with pytest.raises(spk.SpkError, match="A10"):
            c.pc_setup(spk.PC_SCHUR)
        bad = spk.CSR(A.rowptr, A.colidx + 1, A.val, A.ncols)
        with pytest.raises(spk.SpkError, match="out of range"):
            c.set_block(spk.BLOCK_A00, bad)
        c.pc_setup(spk.PC_JACOBI)
        with pytest.raises(spk.SpkError, match="restart"):
            c.fgmres(f, restart=0)


# --------------------------------------------------------------------------- the reference's call sequence
def test_ksp_call_sequence_like_the_reference(spk, oracle, golden_m32):
    """KSPCreate / SetOperators / SetFromOptions / SetUp / Solve / Destroy
    (SaddlePointProblem.c:65-72) with the finished nest and the options of
    SURVEY.md Appendix C."""
    A, f = spk.AssembleOperator_Laplace(32)
    B, g = spk.AssembleOperator_Constraints(32)
    ksp = spk.KSP()
    ksp.setOperators(A, B)
    ksp.setFromOptions("-ksp_type fgmres -ksp_rtol 1e-8 -pc_type fieldsplit -pc_fieldsplit_type schur "
                       "-pc_fieldsplit_schur_fact_type full -pc_fieldsplit_schur_precondition selfp "
                       "-fieldsplit_0_ksp_type preonly -fieldsplit_0_pc_type jacobi "
                       "-fieldsplit_1_ksp_type preonly -fieldsplit_1_pc_type jacobi")
    ksp.setUp()
    x = ksp.solve(np.concatenate([f, g]))
    assert ksp.getConvergedReason() == 2
    assert abs(ksp.getIterationNumber() - golden_m32["schur_its"][3]) <= 1
    assert relerr(x, golden_m32["saddle"]) < 1e-6
    assert len(ksp.getConvergenceHistory()) == ksp.getIterationNumber() + 1
    # as written in the reference: A alone, options decide the rest
    ksp2 = spk.KSP()
    ksp2.setOperators(A)
    ksp2.setFromOptions("-ksp_type fgmres -pc_type jacobi")
    u = ksp2.solve(f)                                   # KSPSolve sets up on demand
    assert ksp2.getIterationNumber() == 75 and relerr(u, golden_m32["u"]) < 1e-4
    ksp.destroy(); ksp2.destroy()


def test_driver_executable_reference_default_problem(spk, appendix_b, golden_m32):
    """saddle_point_run: the reference program's shape (main.c:7-19).  Default grid = the
    reference's Nx = Ny = 3 elements; as written there the solve is A u = f."""
    import os, subprocess
    exe = os.path.join(os.path.dirname(spk.LIB_PATH), "saddle_point_run")
    import tempfile
    wd = tempfile.mkdtemp()
    out = subprocess.run([exe, "-saddle", "0", "-ksp_type", "fgmres", "-pc_type", "jacobi", "-ksp_rtol", "1e-12",
                          "-solution_view"], capture_output=True, text=True, timeout=120, cwd=wd)
    assert "VECTORS U double" in open(os.path.join(wd, "test.vtk")).read()
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    assert "CONVERGED_RTOL" in lines[0] and "4 x 4 nodes, 32 rows" in lines[0]
    u = np.array([float(v) for v in lines[1:]])
    for idx, v in appendix_b["m4_solution"].items():
        assert u[int(idx)] == pytest.approx(v, abs=2e-12)
    out = subprocess.run([exe, "-da_grid_x", "32", "-da_grid_y", "32", "-ksp_type", "fgmres", "-ksp_rtol", "1e-8",
                          "-pc_type", "fieldsplit", "-pc_fieldsplit_type", "schur",
                          "-pc_fieldsplit_schur_fact_type", "full", "-ksp_converged_reason", "-no_vtk"],
                         capture_output=True, text=True, timeout=120, cwd=wd)
    assert out.returncode == 0, out.stdout + out.stderr
    its = int(out.stdout.split("after ")[1].split(" iterations")[0])
    assert abs(its - golden_m32["schur_its"][3]) <= 1
    bad = subprocess.run([exe, "-ksp_type", "cg"], capture_output=True, text=True, timeout=120, cwd=wd)
    assert bad.returncode == 1 and "not supported" in bad.stderr


# --------------------------------------------------------------------------- partitioned algorithm on one GPU
def _run_ranks(spk, P, mx, my, pc, fact, rhs_full, with_B, inner=0, try_peer=False, **kw):
    import threading
    grp = spk.LocalGroup(P)
    out, errs = [None] * P, []
    n = 2 * mx * my

    def work(r):
        try:
            b, e = spk.partition_slab(mx, my, r, P)
            A, _ = spk.AssembleOperator_Laplace(mx, my, b, e)
            c = spk.Context(0)
            c.comm_init_local(grp, r)
            if try_peer:   # refused for ranks of one process: the communicator set before must keep working
                assert not c.comm_enable_peer() and c.comm_backend() == "local"
            c.set_block(spk.BLOCK_A00, A)
            if with_B:
                Bs, _ = spk.AssembleOperator_Constraints(mx, my, b, e)
                c.set_block(spk.BLOCK_A10, Bs)
            c.pc_setup(pc, fact, inner_sweeps=inner, inner_omega=0.8)
            rhs = np.concatenate([rhs_full[b:e], rhs_full[n:]])
            y = c.mult(rhs)
            z = c.pc_apply(rhs)
            x, info = c.fgmres(rhs, **kw)
            out[r] = (b, e, y, z, x, info, c.sizes())
            c.close()
        except Exception as ex:  # noqa: BLE001
            errs.append(ex)
            raise

    th = [threading.Thread(target=work, args=(r,)) for r in range(P)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    grp.close()
    assert not errs, errs
    return out


@pytest.mark.parametrize("P,single", [(2, 0), (3, 0), (2, 1)])
def test_row_partitioned_solver_matches_single_rank(spk, oracle, P, single):
    mx, my = 24, 26
    A, f = spk.AssembleOperator_Laplace(mx, my)
    B, g = spk.AssembleOperator_Constraints(mx, my)
    rhs = np.concatenate([f, g])
    n = A.nrows
    out = _run_ranks(spk, P, mx, my, spk.PC_SCHUR, spk.SCHUR_FULL, rhs, True, rtol=1e-10, single_reduce=single,
                     try_peer=(P == 3))
    y_ref = oracle.apply_K(A, B, rhs)
    z_ref = oracle.pc_apply(A, B, oracle.PC_SCHUR, 3, rhs)
    xo, io = oracle.fgmres(A, rhs, B=B, pc_type=oracle.PC_SCHUR, schur_fact=3, rtol=1e-10)
    y = np.zeros(n + 4); z = np.zeros(n + 4); x = np.zeros(n + 4)
    for (b, e, yr, zr, xr, info, sz) in out:
        y[b:e], z[b:e], x[b:e] = yr[:-4], zr[:-4], xr[:-4]
        y[n:], z[n:], x[n:] = yr[-4:], zr[-4:], xr[-4:]
        assert sz["n_ghost"] == 2 * mx * ((b > 0) + (e < n))           # one node line per neighbour
        assert info["reason"] == 2 and abs(info["its"] - io["its"]) <= 1
        assert np.array_equal(xr[-4:], out[0][4][-4:])                   # multipliers replicated bit for bit
        assert np.array_equal(info["history"], out[0][5]["history"])     # every rank takes the same branch
    assert relerr(y, y_ref) < KERNEL_TOL and relerr(z, z_ref) < KERNEL_TOL
    assert relerr(x, xo) < 1e-8


def test_row_partitioned_long_restart(spk, oracle):
    """Restart 100 (the step-by-step path with chunked Gram-Schmidt) across two ranks: the all-reduce then carries up
    to 101 inner products at once, and every rank must still take the same branches."""
    mx, my = 24, 26
    A, f = spk.AssembleOperator_Laplace(mx, my)
    B, g = spk.AssembleOperator_Constraints(mx, my)
    rhs = np.concatenate([f, g])
    n = A.nrows
    out = _run_ranks(spk, 2, mx, my, spk.PC_SCHUR, spk.SCHUR_FULL, rhs, True, rtol=1e-10, restart=100)
    xo, io = oracle.fgmres(A, rhs, B=B, pc_type=oracle.PC_SCHUR, schur_fact=3, rtol=1e-10, restart=100)
    x = np.zeros(n + 4)
    for (b, e, yr, zr, xr, info, sz) in out:
        x[b:e], x[n:] = xr[:-4], xr[-4:]
        assert info["reason"] == io["reason"] == 2 and abs(info["its"] - io["its"]) <= 1
        assert np.array_equal(info["history"], out[0][5]["history"])
    assert relerr(x, xo) < 1e-8


def _launch_peer_worker(tmp_path, P, mode, port, env_extra=None, timeout=280):
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(P),
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(root, "tests", "_peer_worker.py"), str(tmp_path), mode],
                         capture_output=True, text=True, timeout=timeout, cwd=root, env=env)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]


@pytest.mark.parametrize("P,halo_max,fuse", [(2, None, "1"), (3, None, "1"), (2, "100", "1"), (3, None, "0")])
def test_peer_store_collectives_across_processes(spk, oracle, tmp_path, P, halo_max, fuse):
    """P PROCESSES on this GPU, each owning a row slab; the peer-store backend maps the other
    processes' windows through HIP IPC and the solver's own kernels write the Krylov all-reduces and
    the halo rows into them (tests/_peer_worker.py).  Checked per case: every rank holds the same
    scalars bit for bit (same residual history), the result equals the single-rank oracle to the
    usual bars, and with two ranks -- where the host-staged all-reduce adds in the same order -- it
    is bit-identical to the host-staged run.  halo_max = 100: with a halo segment beyond 100 doubles the
    exchange takes the BULK form (plain doubles in chunks + one flag per chunk: what the node plane of
    a 3-D slab does in production) instead of granules."""
    env = {"SPK_PEER_FUSE": fuse}      # "0": every collective as a launch of its own (granule kernels)
    if halo_max:
        env["SPK_PEER_HALO_MAX"] = halo_max
    _launch_peer_worker(tmp_path, P, "cases", 29650 + P + (10 if halo_max else 0) + (20 if fuse == "0" else 0), env)
    R = [np.load(tmp_path / f"rank{r}.npz") for r in range(P)]
    okws = {"mgs": dict(orthog=1), "refine": dict(refine=1), "r62": dict(restart=62)}
    cases = [("schur_full", 2, (24, 26), oracle.PC_SCHUR, 3, 0), ("schur_full_single", 2, (24, 26), oracle.PC_SCHUR, 3, 0),
             ("schur_full_mgs", 2, (24, 26), oracle.PC_SCHUR, 3, 0), ("schur_full_refine", 2, (24, 26), oracle.PC_SCHUR, 3, 0),
             ("schur_full_r62", 2, (24, 26), oracle.PC_SCHUR, 3, 0), ("schur_full_guess", 2, (24, 26), oracle.PC_SCHUR, 3, 0),
             ("schur_lower_unfused", 2, (24, 26), oracle.PC_SCHUR, 1, 0),
             ("jacobi", 2, (24, 26), oracle.PC_JACOBI, 0, 0), ("jacobi_single", 2, (24, 26), oracle.PC_JACOBI, 0, 0),
             ("schur_diag_fp32", 2, (24, 26), oracle.PC_SCHUR, 0, 3),
             ("jacobi_3d_fp32", 3, (10, 9, 12), oracle.PC_JACOBI, 0, 3)]
    for name, dim, grid, pc, fact, inner in cases:
        saddle = pc == oracle.PC_SCHUR
        if dim == 2:
            A, f = spk.AssembleOperator_Laplace(*grid)
            B, g = spk.AssembleOperator_Constraints(*grid) if saddle else (None, np.zeros(0))
        else:
            A, f = spk.AssembleOperator_Laplace3D(*grid)
            B, g = None, np.zeros(0)
        n, m = A.nrows, len(g)
        Ao = oracle.CSR(A.rowptr, A.colidx, A.val, A.ncols)
        Bo = oracle.CSR(B.rowptr, B.colidx, B.val, B.ncols) if saddle else None
        rhs = np.concatenate([f, g])
        xin = np.concatenate([np.sin(0.37 * np.arange(n)), 0.5 + np.arange(m)])
        kw = dict(inner_its=inner, inner_omega=0.8) if inner else {}
        kw.update(okws.get(name.rsplit("_", 1)[-1], {}))
        if name.endswith("_guess"):
            kw["x0"] = 0.01 * xin
        xo, io = oracle.fgmres(Ao, rhs, B=Bo, pc_type=pc, schur_fact=fact, rtol=1e-9, **kw)
        y_ref = oracle.apply_K(Ao, Bo, xin) if saddle else oracle.spmv(Ao, xin)
        got = {}
        for peer in (0, 1):
            y = np.zeros(n + m); z = np.zeros(n + m); x = np.zeros(n + m)
            for r in range(P):
                k = f"{name}/{peer}/"
                b, e, its, reason, is_peer, ngh = R[r][k + "meta"]
                assert is_peer == peer and reason == 2
                assert np.array_equal(R[r][k + "hist"], R[0][k + "hist"])            # every rank: same branch
                y[b:e], z[b:e], x[b:e] = R[r][k + "y"][:e - b], R[r][k + "z"][:e - b], R[r][k + "x"][:e - b]
                if m:
                    y[n:], z[n:], x[n:] = R[r][k + "y"][-m:], R[r][k + "z"][-m:], R[r][k + "x"][-m:]
                    assert np.array_equal(R[r][k + "x"][-m:], R[0][k + "x"][-m:])    # multipliers replicated
            got[peer] = (y, z, x, len(R[0][f"{name}/{peer}/hist"]))
            assert relerr(y, y_ref) < KERNEL_TOL, (name, peer)
            assert abs(got[peer][3] - 1 - io["its"]) <= 2, (name, peer)
            assert relerr(x, xo) < (1e-6 if inner else 1e-7), (name, peer)
        if P == 2:
            for a, b_ in zip(got[0][:3], got[1][:3]):
                assert np.array_equal(a, b_), name


def test_peer_store_operators_set_again(spk, oracle, tmp_path):
    """KSPSetOperators three times on one context with the peer-store backend on (different grid,
    then the first one again): the halo staging follows the new plan each time."""
    P = 2
    _launch_peer_worker(tmp_path, P, "reset", 29671)
    R = [np.load(tmp_path / f"rank{r}.npz") for r in range(P)]
    for it, (mx, my) in enumerate([(20, 18), (28, 33), (20, 18)]):
        A, f = spk.AssembleOperator_Laplace(mx, my)
        B, g = spk.AssembleOperator_Constraints(mx, my)
        rhs = np.concatenate([f, g])
        xo, io = oracle.fgmres(A, rhs, B=B, pc_type=oracle.PC_SCHUR, schur_fact=3, rtol=1e-9)
        x = np.zeros(A.nrows + 4)
        for r in range(P):
            b, e, its, reason = R[r][f"{it}/meta"]
            assert reason == 2 and abs(its - io["its"]) <= 1
            x[b:e], x[A.nrows:] = R[r][f"{it}/x"][:e - b], R[r][f"{it}/x"][-4:]
        assert relerr(x, xo) < 1e-7
    assert np.array_equal(R[0]["0/x"], R[0]["2/x"])        # same operators again: same bits


def test_peer_store_wait_is_bounded(spk, tmp_path):
    """A rank that never arrives must turn into SPK_ERR_COMM, not into a kernel that spins for ever:
    rank 0 multiplies (halo exchange + all-reduce) while rank 1 stays away."""
    import json
    _launch_peer_worker(tmp_path, 2, "timeout", 29659, {"SPK_PEER_TIMEOUT_MS": "300"}, timeout=120)
    d = json.load(open(tmp_path / "rank0.json"))
    assert d["code"] == -4 and "timed out" in d["msg"]


def test_peer_store_needs_one_process_per_rank(spk):
    """Logical ranks of ONE process are refused (the null stream and hipFree couple their streams):
    the staged local backend stays in place and says why."""
    import threading
    grp = spk.LocalGroup(2)
    got = [None, None]

    def work(r):
        c = spk.Context(0)
        c.comm_init_local(grp, r)
        got[r] = (c.comm_enable_peer(), c.comm_backend(), c.last_error())
        c.close()

    th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join(timeout=60) for t in th]
    grp.close()
    for on, backend, why in got:
        assert not on and backend == "local" and "one process per rank" in why


def test_peer_store_can_be_switched_off(spk, monkeypatch):
    import threading
    monkeypatch.setenv("SPK_COMM_PEER", "0")
    grp = spk.LocalGroup(2)
    got = [None, None]

    def work(r):
        c = spk.Context(0)
        c.comm_init_local(grp, r)
        got[r] = (c.comm_enable_peer(), c.comm_backend())
        c.close()

    th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join(timeout=60) for t in th]
    grp.close()
    assert got == [(False, "local"), (False, "local")]


def test_one_sided_halo(spk, oracle):
    """Structurally non-symmetric split: rank 1 needs a column of rank 0, rank 0 needs nothing -- it
    still has to SEND (its n_ghost is 0; PETSc's VecScatter has the same one-sided shape)."""
    import threading
    n, half = 64, 32
    rp = np.arange(0, 2 * n + 1, 2, dtype=np.int32) - 1
    rp[0] = 0
    ci = np.empty(2 * n - 1, np.int32)
    va = np.empty(2 * n - 1)
    ci[0], va[0] = 0, 2.0
    for i in range(1, n):
        ci[2 * i - 1], va[2 * i - 1] = i - 1, -1.0 - 0.01 * i      # sub-diagonal
        ci[2 * i], va[2 * i] = i, 2.0 + 0.1 * i
    A = spk.CSR(rp, ci, va, n)
    x = _x(n, 5)
    y_ref = oracle.spmv(oracle.CSR(A.rowptr, A.colidx, A.val, A.ncols), x)
    grp = spk.LocalGroup(2)
    out, errs = [None, None], []

    def work(r):
        try:
            sl = A.slab(r * half, (r + 1) * half)
            c = spk.Context(0)
            c.comm_init_local(grp, r)
            c.set_block(spk.BLOCK_A00, sl)
            out[r] = (c.mult(x[r * half:(r + 1) * half]), c.sizes()["n_ghost"])
            c.close()
        except Exception as ex:  # noqa: BLE001
            errs.append(ex)
            raise

    th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join(timeout=200) for t in th]
    grp.close()
    assert not errs, errs
    assert out[0][1] == 0 and out[1][1] == 1
    y = np.concatenate([out[0][0], out[1][0]])
    assert np.array_equal(y[:half], y_ref[:half]) and np.array_equal(y[half + 1:], y_ref[half + 1:])   # local rows: bitwise
    assert relerr(y, y_ref) < KERNEL_TOL      # the boundary row adds its off-rank column last (and fused)


def test_row_partitioned_fp32_inner_solve(spk, oracle):
    """The FP32 inner sweeps across 2 logical ranks (single-precision halo staged as doubles): the
    preconditioner must be the same operator as on one rank up to float rounding (the off-rank
    columns of a boundary row are added after its local ones, not in CSR order)."""
    mx, my = 24, 26
    A, f = spk.AssembleOperator_Laplace(mx, my)
    B, g = spk.AssembleOperator_Constraints(mx, my)
    rhs = np.concatenate([f, g])
    n = A.nrows
    out = _run_ranks(spk, 2, mx, my, spk.PC_SCHUR, spk.SCHUR_DIAG, rhs, True, inner=3, rtol=1e-9)
    z_ref = oracle.pc_apply_inner(A, B, oracle.PC_SCHUR, 0, 3, 0.8, rhs)
    xo, io = oracle.fgmres(A, rhs, B=B, pc_type=oracle.PC_SCHUR, schur_fact=0, rtol=1e-9, inner_its=3, inner_omega=0.8)
    z = np.zeros(n + 4); x = np.zeros(n + 4)
    for (b, e, yr, zr, xr, info, sz) in out:
        z[b:e], x[b:e] = zr[:-4], xr[:-4]
        z[n:], x[n:] = zr[-4:], xr[-4:]
        assert info["reason"] == 2 and abs(info["its"] - io["its"]) <= 2
    assert relerr(z, z_ref) < 1e-6
    interior = slice(2 * mx * 2, n // 2 - 2 * mx * 2)          # rows whose stencil stays on rank 0
    assert np.array_equal(z[interior], z_ref[interior])
    assert relerr(x, xo) < 1e-7


def test_3d_grid_config5_shape(spk, oracle):
    """BASELINE config 5 in miniature: 3-D grid (build-defined generator), dof 3 (81 entries per row in
    27 blocks of 3 x 3 -> the 3x3-blocked kernels), halo-exchange SpMV over 2 z-slabs, FGMRES with the mixed
    FP32 inner solve, and the six-row saddle system on one rank."""
    mx, my, mz = 10, 9, 12
    A, f = spk.AssembleOperator_Laplace3D(mx, my, mz)
    B, g = spk.AssembleOperator_Constraints3D(mx, my, mz)
    Ao = oracle.CSR(A.rowptr, A.colidx, A.val, A.ncols)
    Bo = oracle.CSR(B.rowptr, B.colidx, B.val, B.ncols)
    x = _x(A.nrows, 11)
    rhs = np.concatenate([f, g])
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        assert c.spmv_info()["format"] == "dict3x3"    # (row types + codes over the 3 x 3 blocks; tests/test_gpu_dict.py)
        assert c.spmv_info()["layout_bytes"] < 0.75 * (12 * A.nnz + 4 * (A.nrows + 1) + 16 * A.nrows)
        assert np.array_equal(c.mult(x), oracle.spmv(Ao, x))                      # bitwise
        c.pc_setup(spk.PC_JACOBI, 0, inner_sweeps=3, inner_omega=0.8)
        # the FP32 sweeps from the single-precision planes: the oracle's float loop, bit for bit
        assert np.array_equal(c.pc_apply(x), oracle.pc_apply_inner(Ao, None, oracle.PC_JACOBI, 0, 3, 0.8, x))
        u, info = c.fgmres(f, rtol=1e-9)
        uo, io = oracle.fgmres(Ao, f, pc_type=oracle.PC_JACOBI, rtol=1e-9, inner_its=3, inner_omega=0.8)
        assert info["reason"] == 2 and abs(info["its"] - io["its"]) <= 1 and relerr(u, uo) < 1e-7
        c.set_block(spk.BLOCK_A10, B)
        c.pc_setup(spk.PC_SCHUR, spk.SCHUR_FULL)
        assert relerr(c.mult(np.concatenate([x, g])), oracle.apply_K(Ao, Bo, np.concatenate([x, g]))) < KERNEL_TOL
        sol, info = c.fgmres(rhs, rtol=1e-10)
        so, io = oracle.fgmres(Ao, rhs, B=Bo, pc_type=oracle.PC_SCHUR, schur_fact=3, rtol=1e-10)
        _check_iteration_parity(info, io)
        assert relerr(sol, so) < 1e-8
    # two z-slabs (halo = one node plane per side), FP32 inner solve inside the partitioned loop
    import threading
    grp = spk.LocalGroup(2)
    out, errs = [None, None], []

    def work(r):
        try:
            b, e = spk.partition_slab3d(mx, my, mz, r, 2)
            As, fs = spk.AssembleOperator_Laplace3D(mx, my, mz, b, e)
            cc = spk.Context(0)
            cc.comm_init_local(grp, r)
            cc.set_block(spk.BLOCK_A00, As)
            cc.pc_setup(spk.PC_JACOBI, 0, inner_sweeps=3, inner_omega=0.8)
            y = cc.mult(x[b:e])
            ur, inf = cc.fgmres(fs, rtol=1e-9)
            out[r] = (b, e, y, ur, inf, cc.sizes())
            cc.close()
        except Exception as ex:  # noqa: BLE001
            errs.append(ex)
            raise

    th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    grp.close()
    assert not errs, errs
    y = np.zeros(A.nrows); u2 = np.zeros(A.nrows)
    for (b, e, yr, ur, inf, sz) in out:
        y[b:e], u2[b:e] = yr, ur
        assert sz["n_ghost"] == 3 * mx * my and inf["reason"] == 2
    assert relerr(y, oracle.spmv(Ao, x)) < KERNEL_TOL and relerr(u2, uo) < 1e-6


def test_rccl_single_rank_communicator(spk, oracle):
    """RCCL path with nranks = 1 (all a 1-GPU box can run): unique id, init,
    in-stream all-reduce of the Krylov scalars."""
    A, f = spk.AssembleOperator_Laplace(16)
    B, g = spk.AssembleOperator_Constraints(16)
    rhs = np.concatenate([f, g])
    with spk.Context(0) as c:
        c.comm_init_rccl(0, 1, spk.unique_id())
        c.set_block(spk.BLOCK_A00, A)
        c.set_block(spk.BLOCK_A10, B)
        c.pc_setup(spk.PC_SCHUR, spk.SCHUR_FULL)
        x, info = c.fgmres(rhs, rtol=1e-10)
    xo, io = oracle.fgmres(A, rhs, B=B, pc_type=oracle.PC_SCHUR, schur_fact=3, rtol=1e-10)
    assert info["reason"] == 2 and abs(info["its"] - io["its"]) <= 1 and relerr(x, xo) < 1e-8


def test_bench_multi_process_flow_over_gloo(spk):
    """bench.py's N > 1 flow (one process per rank, slabs, halo plan exchanged between processes,
    rank-0 JSON line) with 2 processes sharing this GPU.  RCCL refuses two ranks on one device, so
    the collectives go through the host-callback transport over gloo (SPK_BENCH_COMM=gloo); the
    partitioned result must match the single-process one."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--grid", "192", "--steps", "45", "--warmup", "5", "--no-cpu-baseline", "--spmv-reps", "5"]

    def last_json(out):
        return json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])

    one = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True,
                         timeout=240, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    env = dict(os.environ, SPK_BENCH_COMM="gloo")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29631", os.path.join(root, "bench.py"),
                          "--gpus", "2"] + common, capture_output=True, text=True, timeout=280, cwd=root, env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    d1, d2 = last_json(one.stdout), last_json(two.stdout)
    # AUTO: both take the resident restart-cycle kernel -- the two-process run with the ranks' sums and the halo rows crossing
    # the IPC windows inside the one launch per cycle
    assert d1["config"]["iteration_form_run"] == 6 and d2["config"]["iteration_form_run"] == 6
    assert d2["config"]["resident_fallback"] is None and d2["residual_check"]["consistent"]
    assert all(r["allreduce"]["fused"] > 0 and r["allreduce"]["inner"] == 0 for r in d2["ranks"])
    assert d1["n_gpus"] == 1 and d2["n_gpus"] == 2 and d2["steps"] == 45 and d2["scaling"] == "strong"
    assert d2["residual_after_steps"] == pytest.approx(d1["residual_after_steps"], rel=1e-6)
    assert d2["roofline"]["bytes_per_launch"] < d1["roofline"]["bytes_per_launch"]        # half the rows per rank
    # the two processes map each other's windows through HIP IPC: collectives by the solver's own kernels
    assert d2["config"]["collectives"] == "peer-store"
    # the launch-by-launch form 5 with the collectives by the solver's kernels, and the same with the backend off: same sums,
    # same order, same bits
    runs = []
    for port, e in ((29635, env), (29633, dict(env, SPK_BENCH_PEER="0"))):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                            "--gpus", "2", "--iter-form", "5"] + common, capture_output=True, text=True, timeout=280, cwd=root, env=e)
        assert r.returncode == 0, r.stderr[-2000:]
        runs.append(last_json(r.stdout))
    d5, d3 = runs
    assert d5["config"]["collectives"] == "peer-store" and d5["config"]["iteration_form_run"] == 5
    assert d3["config"]["collectives"] == "host-callback"
    assert d3["residual_after_steps"] == d5["residual_after_steps"]
    assert d5["residual_after_steps"] == pytest.approx(d2["residual_after_steps"], rel=1e-6)


def test_bench_inner_backend_rehearsal_four_processes_at_1024(spk):
    """The fallback route of the 8-GPU job, rehearsed at the bench's own size: bench.py --grid 1024 over FOUR processes with
    the peer-store backend switched off (SPK_BENCH_PEER=0), so that every Krylov all-reduce and every halo exchange goes
    through the INNER communicator's call sites in spk_comm.cpp (allreduce_sum / exchange: the ones ncclAllReduce and
    the grouped ncclSend / ncclRecv sit behind on a multi-GPU node; here the host-callback transport over gloo, because RCCL
    refuses two ranks on one device).  The line must carry its own integrity proof (residual_check) and say which route
    the collectives took (ranks[])."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SPK_BENCH_COMM="gloo", SPK_BENCH_PEER="0")
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
                          "--master-addr", "127.0.0.1", "--master-port", "29637", os.path.join(root, "bench.py"),
                          "--gpus", "4", "--grid", "1024", "--steps", "35", "--warmup", "3", "--no-cpu-baseline", "--spmv-reps", "5"],
                         capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    d = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 4 and d["steps"] == 35 and d["value"] is not None
    assert d["residual_check"]["consistent"] and d["config"]["collectives"] == "host-callback"
    assert len(d["ranks"]) == 4
    for r, info in enumerate(d["ranks"]):
        assert info["rank"] == r and not info["peer_enabled"] and info["backend"] == "host-callback"
        assert info["allreduce"]["inner"] >= 2 * 35 and info["allreduce"]["fused"] == 0
        assert info["halo_exchanges"]["inner"] >= 35
