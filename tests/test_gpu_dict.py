"""The "row types + deviation codes" layout of the A block (DictDev, spk_k_dict.hip): found in the caller's CSR at
KSPSetOperators, bit-identical products, automatic fall-back.  Needs a real MI355X: run with -m gpu.

The reference assembles the same element matrix for every element of a uniform grid
(/root/reference/src/Discretization.c:25, :293-332) up to the rounding of a Jacobian formed from node coordinates
(:96-128): A's entries scatter by a few hundred ulps around a handful of ideal values.  The layout stores a 16-bit row
type per block row and an integer deviation per value as a bit field (value = base + k 2^g exactly) -- under 2 bytes per stored
non-zero instead of 9 (blocked) or 12 (CSR) -- and forms the same products in the same order."""
import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu


def _x(n, seed=12345):
    return np.random.default_rng(seed).uniform(-1.0, 1.0, n)


def _with_format(monkeypatch, spk, fmt, A, fn):
    """runs fn(ctx) on a context whose operator was set under SPK_SPMV_FORMAT=fmt (None: default)"""
    if fmt is None:
        monkeypatch.delenv("SPK_SPMV_FORMAT", raising=False)
    else:
        monkeypatch.setenv("SPK_SPMV_FORMAT", fmt)
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        return fn(c)


@pytest.mark.parametrize("mx,my", [(4, 4), (33, 33), (64, 64), (50, 7), (300, 200)])
def test_dictionary_found_on_uniform_grids(spk, oracle, monkeypatch, mx, my):
    """dof-2 grids: the layout is the default, holds a few dozen row types and block classes, and its product is bit for
    bit the blocked kernel's, the CSR kernel's and the oracle's."""
    A, _ = spk.AssembleOperator_Laplace(mx, my)
    x = _x(A.nrows, 3)
    out = {}
    for fmt in (None, "bcsr", "csr"):
        out[fmt] = _with_format(monkeypatch, spk, fmt, A, lambda c: (c.spmv_info(), c.spmv_models(), c.mult(x)))
    assert out[None][0]["format"] == "dict2x2" and out["bcsr"][0]["format"] == "bcsr2x2" and out["csr"][0]["format"] == "csr"
    mod = out[None][1]
    assert 0 < mod["patterns"] <= 64 and 0 < mod["blocks"] <= 64    # row types, block classes
    assert mod["blocked_bytes"] < mod["csr_bytes"]
    if mx >= 33:
        assert mod["dict_bytes"] < 0.45 * mod["blocked_bytes"]
    assert mod["csr_bytes"] == 12 * A.nnz + 4 * (A.nrows + 1) + 16 * A.nrows
    assert out["bcsr"][1]["dict_bytes"] == 0
    y_ref = oracle.spmv(A, x)
    for fmt in (None, "bcsr", "csr"):
        assert np.array_equal(out[fmt][2], y_ref), fmt


def test_dictionary_3d_grid(spk, oracle, monkeypatch):
    """dof-3 grid (BASELINE config 5 shape): 27 blocks of 3 x 3 per row; product and FP32 sweeps bit for bit."""
    A, _ = spk.AssembleOperator_Laplace3D(14, 11, 9)
    x = _x(A.nrows, 4)

    def run(c):
        y = c.mult(x)
        c.pc_setup(spk.PC_JACOBI, 0, inner_sweeps=3, inner_omega=0.7)
        return c.spmv_info()["format"], c.spmv_models(), y, c.pc_apply(x)
    monkeypatch.delenv("SPK_DICT_NOUNIFORM", raising=False)
    d = _with_format(monkeypatch, spk, None, A, run)
    b = _with_format(monkeypatch, spk, "bcsr", A, run)
    monkeypatch.setenv("SPK_DICT_NOUNIFORM", "1")      # per-class bit fields (what a rougher matrix would get)
    p = _with_format(monkeypatch, spk, None, A, run)
    monkeypatch.delenv("SPK_DICT_NOUNIFORM", raising=False)
    monkeypatch.setenv("SPK_DICT3_PIPELINE", "1")      # the pipelined kernels of systems beyond a million block rows, here
    q = _with_format(monkeypatch, spk, None, A, run)
    monkeypatch.delenv("SPK_DICT3_PIPELINE", raising=False)
    assert q[0] == "dict3x3" and np.array_equal(q[2], d[2]) and np.array_equal(q[3], d[3])
    assert d[0] == "dict3x3" and b[0] == "bcsr3x3" and p[0] == "dict3x3"
    assert d[1]["patterns"] <= 343 and d[1]["blocks"] <= 128
    y_ref = oracle.spmv(A, x)
    assert np.array_equal(d[2], y_ref) and np.array_equal(b[2], y_ref) and np.array_equal(p[2], y_ref)
    z_ref = oracle.pc_apply_inner(A, None, oracle.PC_JACOBI, 0, 3, 0.7, x)
    assert np.array_equal(d[3], z_ref) and np.array_equal(b[3], z_ref) and np.array_equal(p[3], z_ref)


def test_dictionary_fp32_sweeps_2d(spk, oracle, monkeypatch):
    A, _ = spk.AssembleOperator_Laplace(45, 38)
    x = _x(A.nrows, 8)

    def run(c):
        c.pc_setup(spk.PC_JACOBI, 0, inner_sweeps=4, inner_omega=0.8)
        return c.spmv_info()["format"], c.pc_apply(x)
    d = _with_format(monkeypatch, spk, None, A, run)
    assert d[0] == "dict2x2"
    assert np.array_equal(d[1], oracle.pc_apply_inner(A, None, oracle.PC_JACOBI, 0, 4, 0.8, x))


def test_dictionary_takes_perturbed_entries(spk, oracle, monkeypatch):
    """A few entries changed by hand (another coefficient in some elements, as a user of the reference would get from a
    non-constant `coeff`, Discretization.c:156-157): more patterns, still a dictionary, still the oracle's bits."""
    A, _ = spk.AssembleOperator_Laplace(40, 40)
    val = A.val.copy()
    rng = np.random.default_rng(5)
    big = np.flatnonzero(np.abs(val) > 1e-3)          # (the stored zeros carry rounding residues of ~1e-13: left alone)
    for k in rng.choice(big, 200, replace=False):
        val[k] *= 1.0 + 0.25 * rng.standard_normal()
    A2 = spk.CSR(A.rowptr, A.colidx, val, A.ncols)
    x = _x(A.nrows, 6)
    fmt, mod, y = _with_format(monkeypatch, spk, None, A2, lambda c: (c.spmv_info()["format"], c.spmv_models(), c.mult(x)))
    A0fmt, mod0, _ = _with_format(monkeypatch, spk, None, A, lambda c: (c.spmv_info()["format"], c.spmv_models(), c.mult(x)))
    assert fmt == "dict2x2" and A0fmt == "dict2x2"
    assert mod["patterns"] > mod0["patterns"] and mod["blocks"] > mod0["blocks"]
    assert np.array_equal(y, oracle.spmv(A2, x))


def test_dictionary_fields_across_the_halves(spk, oracle, monkeypatch):
    """A class whose four widths allow no packing inside the 32-bit halves of its word but fit 64 bits (what the 2048^2
    grid brings: 20 + 14 + 14 + 13) keeps the layout with a field across the halves: the plain kernels extract it with
    64-bit shifts; the pipelined product and the resident cycle stand back.  Here: the first diagonal entry of every
    diagonal block scattered over 2^23 granules of 2^-46 by hand."""
    A, f = spk.AssembleOperator_Laplace(48, 40)
    B, g = spk.AssembleOperator_Constraints(48, 40)
    val = A.val.copy()
    rng = np.random.default_rng(11)
    rows = np.repeat(np.arange(A.nrows), np.diff(A.rowptr))
    big = np.abs(val) > 1e-3
    diag = big & (rows == A.colidx) & (rows % 2 == 0)      # entry (0, 0) of the diagonal blocks: 24 bits of codes
    val[diag] += rng.integers(-2**22, 2**22, int(diag.sum())) * 2.0**-46
    A2 = spk.CSR(A.rowptr, A.colidx, val, A.ncols)
    x = _x(A.nrows, 8)
    rhs = np.concatenate([f, g])

    def run(c):
        y = c.mult(x)
        c.set_block(spk.BLOCK_A10, B)
        c.pc_setup(spk.PC_SCHUR, spk.SCHUR_FULL)
        return c.spmv_info()["format"], y, c.fgmres(rhs, rtol=1e-30, max_it=40), c.iteration_form()[0]
    fmt, y, (xs, info), form = _with_format(monkeypatch, spk, None, A2, run)
    fb, yb, (xb, ib), formb = _with_format(monkeypatch, spk, "bcsr", A2, run)
    assert fmt == "dict2x2" and fb == "bcsr2x2"
    assert np.array_equal(y, oracle.spmv(A2, x)) and np.array_equal(y, yb)
    assert form == 5 and formb == 5            # (no resident cycle on such a layout)
    assert np.array_equal(info["history"], ib["history"]) and np.array_equal(xs, xb)


def test_dictionary_refuses_matrices_without_repetition(spk, oracle, monkeypatch):
    """every block different (random values on the grid's pattern): more than 1024 block classes -> the blocked layout
    stays, results unchanged; and too many distinct ROW TYPES with few block classes likewise."""
    A, _ = spk.AssembleOperator_Laplace(48, 48)
    rng = np.random.default_rng(9)
    A2 = spk.CSR(A.rowptr, A.colidx, A.val + rng.standard_normal(A.nnz), A.ncols)
    x = _x(A.nrows, 2)
    fmt, mod, y = _with_format(monkeypatch, spk, None, A2, lambda c: (c.spmv_info()["format"], c.spmv_models(), c.mult(x)))
    assert fmt == "bcsr2x2" and mod["dict_bytes"] == 0 and mod["patterns"] == 0
    assert np.array_equal(y, oracle.spmv(A2, x))
    # few blocks, every row its own sequence: block row br = a diagonal block + blocks at br's "digits" in base 5
    nb = 3000
    rp, ci, va = [0], [], []
    blocks = [np.array([[4.0, 1.0], [1.0, 3.0]]), np.array([[0.5, 0.0], [0.25, -0.5]]), np.array([[-1.0, 2.0], [0.0, 1.0]])]
    for br in range(nb):
        cols = sorted({br, (br * 7 + 1) % nb, (br * br + 3) % nb, (br // 5 * 11 + br % 5) % nb})
        for r in range(2):
            for c in cols:
                b = blocks[0] if c == br else blocks[1 + (c + br) % 2]
                ci.extend([2 * c, 2 * c + 1])
                va.extend(b[r])
            rp.append(len(ci))
    A3 = spk.CSR(np.array(rp, np.int32), np.array(ci, np.int32), np.array(va), 2 * nb)
    x3 = _x(2 * nb, 7)
    fmt3, mod3, y3 = _with_format(monkeypatch, spk, None, A3, lambda c: (c.spmv_info()["format"], c.spmv_models(), c.mult(x3)))
    assert fmt3 == "bcsr2x2" and mod3["patterns"] == 0
    assert np.array_equal(y3, oracle.spmv(A3, x3))


def test_dictionary_in_the_solver(spk, oracle, monkeypatch):
    """FGMRES on the saddle system with the dictionary product (default) and with the blocked one: the same products give
    the same iterates -- histories and solutions bit for bit -- and both follow the oracle."""
    A, f = spk.AssembleOperator_Laplace(64)
    B, g = spk.AssembleOperator_Constraints(64)
    rhs = np.concatenate([f, g])

    def run(c):
        c.set_block(spk.BLOCK_A10, B)
        c.pc_setup(spk.PC_SCHUR, spk.SCHUR_FULL)
        # (form 5 in both runs: the resident form 6 that AUTO would take needs the row-type layout, and groups the partial
        # sums of the inner products differently)
        return c.spmv_info()["format"], c.fgmres(rhs, rtol=1e-9, max_it=400, iteration_form=5)
    fd, (xd, idd) = _with_format(monkeypatch, spk, None, A, run)
    fb, (xb, ib) = _with_format(monkeypatch, spk, "bcsr", A, run)
    assert fd == "dict2x2" and fb == "bcsr2x2"
    assert idd["its"] == ib["its"] and np.array_equal(idd["history"], ib["history"]) and np.array_equal(xd, xb)
    xo, io = oracle.fgmres(A, rhs, B=B, pc_type=oracle.PC_SCHUR, schur_fact=oracle.SCHUR_FULL, rtol=1e-9, max_it=400)
    assert abs(idd["its"] - io["its"]) <= 1 and relerr(xd, xo) < 1e-7


@pytest.mark.parametrize("mx,my", [(33, 33), (300, 200), (1024, 40)])
def test_dictionary_per_class_fields(spk, oracle, monkeypatch, mx, my):
    """Where the widest need of any class per block entry fits one common bit-field layout (what these grids do) the
    pipelined product extracts without reading the field table; SPK_DICT_NOUNIFORM=1 keeps the per-class layout of larger
    systems: the same bits from both, in the product, in the resident cycle and in a solve."""
    A, f = spk.AssembleOperator_Laplace(mx, my)
    B, g = spk.AssembleOperator_Constraints(mx, my)
    rhs = np.concatenate([f, g])
    x = _x(A.nrows, 9)
    y_ref = oracle.spmv(A, x)

    def run(c):
        y = c.mult(x)
        c.set_block(spk.BLOCK_A10, B)
        c.pc_setup(spk.PC_SCHUR, spk.SCHUR_FULL)
        z = c.mult(np.concatenate([x, [0.5, -0.25, 0.125, 1.0]]))     # (the B^T rows in the product's epilogue)
        r5 = c.fgmres(rhs, rtol=1e-30, max_it=45, iteration_form=5)
        r0 = c.fgmres(rhs, rtol=1e-30, max_it=45)                      # (AUTO: the resident cycle where it fits)
        return y, z, r5, r0, c.iteration_form()
    monkeypatch.delenv("SPK_DICT_NOUNIFORM", raising=False)
    yu, zu, (xu5, iu5), (xu0, iu0), fu = _with_format(monkeypatch, spk, None, A, run)
    monkeypatch.setenv("SPK_DICT_NOUNIFORM", "1")
    yp, zp, (xp5, ip5), (xp0, ip0), fp = _with_format(monkeypatch, spk, None, A, run)
    assert np.array_equal(yu, y_ref) and np.array_equal(yp, y_ref)
    assert np.array_equal(zu, zp)
    assert np.array_equal(iu5["history"], ip5["history"]) and np.array_equal(xu5, xp5)
    assert fu == fp and np.array_equal(iu0["history"], ip0["history"]) and np.array_equal(xu0, xp0)


def test_product_launches_timed_inside_a_solve(spk, monkeypatch):
    """spk_debug_time_products / spk_get_product_timing (bench.py's roofline): one pair of HIP events per product launch of
    the iterations -- restart - 1 per cycle on the three-launch form, none on the resident form -- and nothing when off."""
    A, f = spk.AssembleOperator_Laplace(64)
    B, g = spk.AssembleOperator_Constraints(64)
    rhs = np.concatenate([f, g])
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        c.set_block(spk.BLOCK_A10, B)
        c.pc_setup(spk.PC_SCHUR, spk.SCHUR_FULL)
        x0, i0 = c.fgmres(rhs, rtol=0.0, abstol=0.0, max_it=60, iteration_form=5)
        c.time_products(200)
        x1, i1 = c.fgmres(rhs, rtol=0.0, abstol=0.0, max_it=60, iteration_form=5)
        pt = c.product_timing()
        # (29 per cycle; the two the host had enqueued ahead of the solve's end return at once and may or may not fall
        # below the cut on a system this small)
        assert 2 * 29 <= pt["launches"] <= 2 * 29 + 2 and 0.0 < pt["min_ms"] <= pt["median_ms"] <= pt["max_ms"] < 5.0
        assert np.array_equal(x0, x1) and np.array_equal(i0["history"], i1["history"])     # (timing changes nothing)
        c.time_products(200)
        c.fgmres(rhs, rtol=0.0, abstol=0.0, max_it=60, iteration_form=6)
        assert c.product_timing()["launches"] == 0
        c.time_products(0)
        c.fgmres(rhs, rtol=0.0, abstol=0.0, max_it=60, iteration_form=5)
        assert c.product_timing()["launches"] == 0


def test_dictionary_row_slabs(spk, oracle, monkeypatch):
    """three logical ranks: every slab finds its own dictionary (rows at a cut lose their off-rank blocks to the halo
    part), the partitioned product equals the single-rank one bit for bit away from the cuts."""
    import threading
    monkeypatch.delenv("SPK_SPMV_FORMAT", raising=False)
    mx = my = 36
    A, _ = spk.AssembleOperator_Laplace(mx, my)
    x = _x(A.nrows, 11)
    y_ref = oracle.spmv(A, x)
    P = 3
    grp = spk.LocalGroup(P)
    outs, fmts, errs = [None] * P, [None] * P, []

    def work(r):
        try:
            rb, re_ = spk.partition_slab(mx, my, r, P)
            Ar, _ = spk.AssembleOperator_Laplace(mx, my, rb, re_)
            with spk.Context(0) as c:
                c.comm_init_local(grp, r)
                c.set_block(spk.BLOCK_A00, Ar)
                fmts[r] = c.spmv_info()["format"]
                outs[r] = c.mult(x[rb:re_])
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(r,)) for r in range(P)]
    [t.start() for t in th]
    [t.join() for t in th]
    grp.close()
    assert not errs, errs
    assert fmts == ["dict2x2"] * P
    y = np.concatenate(outs)
    # rows of the node lines at a cut add their off-rank columns after the local ones: another order, not other values
    cut = np.zeros(A.nrows, bool)
    for r in range(1, P):
        rb, _ = spk.partition_slab(mx, my, r, P)
        cut[rb - 2 * mx:rb + 2 * mx] = True
    assert np.array_equal(y[~cut], y_ref[~cut])
    assert np.allclose(y[cut], y_ref[cut], rtol=1e-13, atol=1e-15)


def test_constraint_block_column_validation(spk):
    """A10 with a column number of exactly -1 (the value a sentinel once used) or n is refused with SPK_ERR_ARG and the
    operator set before stays usable."""
    A, f = spk.AssembleOperator_Laplace(16)
    B, g = spk.AssembleOperator_Constraints(16)
    rhs = np.concatenate([f, g])
    with spk.Context(0) as c:
        c.set_block(spk.BLOCK_A00, A)
        c.set_block(spk.BLOCK_A10, B)
        c.pc_setup(spk.PC_SCHUR, spk.SCHUR_FULL)
        x0, i0 = c.fgmres(rhs, rtol=1e-8)
        for badcol in (-1, A.nrows, -7):
            ci = B.colidx.copy()
            ci[len(ci) // 2] = badcol
            with pytest.raises(spk.SpkError, match="A10: column") as ei:
                c.set_block(spk.BLOCK_A10, spk.CSR(B.rowptr, ci, B.val, B.ncols))
            assert ei.value.code == -1   # SPK_ERR_ARG
            x1, i1 = c.fgmres(rhs, rtol=1e-8)
            assert i1["its"] == i0["its"] and np.array_equal(x1, x0)


@pytest.mark.parametrize("order", ["spk_first", "torch_first"])
def test_context_and_torch_share_one_hip_runtime(spk, order):
    """a context AND a torch CUDA tensor in one process, the package imported before or after torch (bench.py's N > 1 flow
    needs both; the import order used to decide whether the second runtime found the GPU)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    imp = "import saddle_point_petsc_amd as S, torch" if order == "spk_first" else "import torch, saddle_point_petsc_amd as S"
    code = imp + "; c = S.Context(0); t = torch.ones(4, device='cuda'); print(float(t.sum())); c.close()"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("4.0"), out.stdout + out.stderr
