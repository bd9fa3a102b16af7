"""ctypes front end of oracle/libsp_oracle.so (C restatement, see sp_oracle.c).

TEST INFRASTRUCTURE ONLY -- never imported by saddle_point_petsc_amd.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsp_oracle.so")

PC_NONE, PC_JACOBI, PC_SCHUR = 0, 1, 2
SCHUR_DIAG, SCHUR_LOWER, SCHUR_UPPER, SCHUR_FULL = 0, 1, 2, 3

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile the oracle with its Makefile (gcc only)."""
    if force or not os.path.exists(_SO) or (
        os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "sp_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])


class Operator(C.Structure):
    _fields_ = [
        ("n", C.c_int32),
        ("a_rowptr", C.c_void_p), ("a_colidx", C.c_void_p), ("a_val", C.c_void_p),
        ("m", C.c_int32),
        ("b_rowptr", C.c_void_p), ("b_colidx", C.c_void_p), ("b_val", C.c_void_p),
    ]


class Options(C.Structure):
    _fields_ = [
        ("pc_type", C.c_int32), ("schur_fact", C.c_int32), ("restart", C.c_int32),
        ("max_it", C.c_int32), ("rtol", C.c_double), ("abstol", C.c_double),
        ("dtol", C.c_double), ("guess_nonzero", C.c_int32), ("threads", C.c_int32),
        ("orthog", C.c_int32), ("refine", C.c_int32),
        ("inner_its", C.c_int32), ("pad1", C.c_int32), ("inner_omega", C.c_double),
    ]


class Result(C.Structure):
    _fields_ = [
        ("its", C.c_int32), ("reason", C.c_int32), ("rnorm", C.c_double),
        ("rnorm0", C.c_double), ("hist_len", C.c_int32), ("pad", C.c_int32),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.spo_grid_sizes.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.spo_assemble_A.argtypes = [C.c_int, C.c_int, _i32p, _i32p, _f64p]
        L.spo_assemble_f.argtypes = [C.c_int, C.c_int, _f64p]
        L.spo_apply_bc.argtypes = [C.c_int, C.c_int, _i32p, _i32p, _f64p, _f64p]
        L.spo_constraint_nnz.restype = C.c_int64
        L.spo_constraint_nnz.argtypes = [C.c_int, C.c_int]
        L.spo_assemble_B.argtypes = [C.c_int, C.c_int, _i32p, _i32p, _f64p]
        L.spo_constraint_rhs.argtypes = [_f64p]
        L.spo_element_stiffness.argtypes = [_f64p, _f64p, _f64p]
        L.spo_element_load.argtypes = [_f64p, _f64p]
        L.spo_spmv.argtypes = [C.c_int32, _i32p, _i32p, _f64p, _f64p, _f64p]
        L.spo_apply_K.argtypes = [C.POINTER(Operator), _f64p, _f64p]
        L.spo_pc_apply_once.argtypes = [C.POINTER(Operator), C.c_int, C.c_int, _f64p, _f64p]
        L.spo_pc_apply_inner.argtypes = [C.POINTER(Operator), C.c_int, C.c_int, C.c_int, C.c_double, _f64p, _f64p]
        L.spo_fgmres.argtypes = [C.POINTER(Operator), C.POINTER(Options), _f64p, _f64p,
                                 C.POINTER(Result), _f64p, C.c_int32]
        L.spo_vec_dot.restype = C.c_double
        L.spo_vec_dot.argtypes = [C.c_int64, _f64p, _f64p]
        L.spo_vec_norm.restype = C.c_double
        L.spo_vec_norm.argtypes = [C.c_int64, _f64p]
        L.spo_time_spmv.restype = C.c_double
        L.spo_time_spmv.argtypes = [C.c_int32, _i32p, _i32p, _f64p, _f64p, _f64p, C.c_int, C.c_int]
        L.spo_set_threads.argtypes = [C.c_int]
        L.spo_jacobi_setup.argtypes = [C.POINTER(Operator), _f64p]
        L.spo_schur_setup.argtypes = [C.POINTER(Operator), _f64p, _f64p, _f64p]
        _lib = L
    return _lib


class CSR:
    """Plain CSR triple (int32 rowptr/colidx, float64 val)."""

    def __init__(self, rowptr, colidx, val, ncols):
        self.rowptr = np.ascontiguousarray(rowptr, np.int32)
        self.colidx = np.ascontiguousarray(colidx, np.int32)
        self.val = np.ascontiguousarray(val, np.float64)
        self.nrows = len(self.rowptr) - 1
        self.ncols = int(ncols)

    @property
    def nnz(self):
        return int(self.rowptr[-1])

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.val, self.colidx, self.rowptr), shape=(self.nrows, self.ncols))


def element_stiffness(xe, coeff=None):
    xe = np.ascontiguousarray(xe, np.float64)
    coeff = np.ones(4) if coeff is None else np.ascontiguousarray(coeff, np.float64)
    Ke = np.zeros(64)
    lib().spo_element_stiffness(xe, coeff, Ke)
    return Ke.reshape(8, 8).T.copy()  # Ke[i + 8 j] -> [i, j]


def element_load(xe):
    xe = np.ascontiguousarray(xe, np.float64)
    Fe = np.zeros(8)
    lib().spo_element_load(xe, Fe)
    return Fe


def assemble(mx, my=None, bc=True):
    """A (CSR, structural zeros kept), f for an mx x my NODE grid; Dirichlet
    applied unless bc=False."""
    my = mx if my is None else my
    n, nnz = C.c_int64(), C.c_int64()
    lib().spo_grid_sizes(mx, my, C.byref(n), C.byref(nnz))
    rowptr = np.zeros(n.value + 1, np.int32)
    colidx = np.zeros(nnz.value, np.int32)
    val = np.zeros(nnz.value)
    f = np.zeros(n.value)
    assert lib().spo_assemble_A(mx, my, rowptr, colidx, val) == 0
    lib().spo_assemble_f(mx, my, f)
    if bc:
        lib().spo_apply_bc(mx, my, rowptr, colidx, val, f)
    return CSR(rowptr, colidx, val, n.value), f


def assemble_constraints(mx, my=None):
    """Build-defined B (4 x n) and g."""
    my = mx if my is None else my
    nnz = lib().spo_constraint_nnz(mx, my)
    rowptr = np.zeros(5, np.int32)
    colidx = np.zeros(nnz, np.int32)
    val = np.zeros(nnz)
    lib().spo_assemble_B(mx, my, rowptr, colidx, val)
    g = np.zeros(4)
    lib().spo_constraint_rhs(g)
    return CSR(rowptr, colidx, val, 2 * mx * my), g


def assemble3d(mx, my=None, mz=None, bc=True):
    """3-D input generator (BUILD-DEFINED, not in the reference): A, f on an mx x my x mz node grid."""
    my = mx if my is None else my
    mz = mx if mz is None else mz
    L = lib()
    L.spo3_grid_sizes.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.spo3_assemble.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f64p, _f64p]
    n, nnz = C.c_int64(), C.c_int64()
    L.spo3_grid_sizes(mx, my, mz, C.byref(n), C.byref(nnz))
    rowptr = np.zeros(n.value + 1, np.int32)
    colidx = np.zeros(nnz.value, np.int32)
    val = np.zeros(nnz.value)
    f = np.zeros(n.value)
    assert L.spo3_assemble(mx, my, mz, int(bc), rowptr, colidx, val, f) == 0
    return CSR(rowptr, colidx, val, n.value), f


def assemble_constraints3d(mx, my=None, mz=None):
    my = mx if my is None else my
    mz = mx if mz is None else mz
    L = lib()
    L.spo3_constraint_nnz.restype = C.c_int64
    L.spo3_constraint_nnz.argtypes = [C.c_int, C.c_int, C.c_int]
    L.spo3_assemble_B.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f64p, _f64p]
    nnz = L.spo3_constraint_nnz(mx, my, mz)
    rowptr = np.zeros(7, np.int32)
    colidx = np.zeros(nnz, np.int32)
    val = np.zeros(nnz)
    g = np.zeros(6)
    L.spo3_assemble_B(mx, my, mz, rowptr, colidx, val, g)
    return CSR(rowptr, colidx, val, 3 * mx * my * mz), g


def assemble_divergence3d(mx, my=None, mz=None):
    """Build-defined discrete divergence block (one row per hexahedron, Dirichlet columns dropped)."""
    my = mx if my is None else my
    mz = mx if mz is None else mz
    L = lib()
    L.spo3_divergence_nnz.restype = C.c_int64
    L.spo3_divergence_nnz.argtypes = [C.c_int, C.c_int, C.c_int]
    L.spo3_assemble_div.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f64p]
    nnz = L.spo3_divergence_nnz(mx, my, mz)
    m = (mx - 1) * (my - 1) * (mz - 1)
    rowptr = np.zeros(m + 1, np.int32)
    colidx = np.zeros(nnz, np.int32)
    val = np.zeros(nnz)
    L.spo3_assemble_div(mx, my, mz, rowptr, colidx, val)
    return CSR(rowptr, colidx, val, 3 * mx * my * mz)


def _operator(A, B=None):
    op = Operator()
    op.n = A.nrows
    op.a_rowptr, op.a_colidx, op.a_val = A.rowptr.ctypes.data, A.colidx.ctypes.data, A.val.ctypes.data
    if B is not None:
        op.m = B.nrows
        op.b_rowptr, op.b_colidx, op.b_val = B.rowptr.ctypes.data, B.colidx.ctypes.data, B.val.ctypes.data
    else:
        op.m = 0
    return op


def spmv(A, x):
    y = np.zeros(A.nrows)
    lib().spo_spmv(A.nrows, A.rowptr, A.colidx, A.val, np.ascontiguousarray(x, np.float64), y)
    return y


def apply_K(A, B, x):
    op = _operator(A, B)
    y = np.zeros(A.nrows + (B.nrows if B is not None else 0))
    lib().spo_apply_K(C.byref(op), np.ascontiguousarray(x, np.float64), y)
    return y


def pc_apply(A, B, pc_type, schur_fact, x):
    op = _operator(A, B)
    y = np.zeros(A.nrows + (B.nrows if B is not None else 0))
    lib().spo_pc_apply_once(C.byref(op), pc_type, schur_fact, np.ascontiguousarray(x, np.float64), y)
    return y


def pc_apply_inner(A, B, pc_type, schur_fact, inner_its, inner_omega, x):
    """PC with the FP32 Richardson/Jacobi inner solve standing for A^-1."""
    op = _operator(A, B)
    y = np.zeros(A.nrows + (B.nrows if B is not None else 0))
    lib().spo_pc_apply_inner(C.byref(op), pc_type, schur_fact, inner_its, inner_omega,
                             np.ascontiguousarray(x, np.float64), y)
    return y


def jacobi_dinv(A):
    op = _operator(A, None)
    d = np.zeros(A.nrows)
    lib().spo_jacobi_setup(C.byref(op), d)
    return d


def schur_setup(A, B):
    op = _operator(A, B)
    d = jacobi_dinv(A)
    shat = np.zeros(B.nrows)
    G = np.zeros(B.nrows * B.nrows)
    lib().spo_schur_setup(C.byref(op), d, shat, G)
    return shat, G.reshape(B.nrows, B.nrows)


def fgmres(A, b, B=None, x0=None, pc_type=PC_JACOBI, schur_fact=SCHUR_FULL, restart=30,
           max_it=10000, rtol=1e-5, abstol=1e-50, dtol=1e4, threads=1, orthog=0, refine=0,
           inner_its=0, inner_omega=1.0):
    """PETSc-semantics FGMRES on K = A or [A B^T; B 0].  Returns (x, info)."""
    op = _operator(A, B)
    N = A.nrows + (B.nrows if B is not None else 0)
    b = np.ascontiguousarray(b, np.float64)
    assert b.shape == (N,)
    x = np.zeros(N) if x0 is None else np.array(x0, np.float64)
    opt = Options(pc_type, schur_fact, restart, max_it, rtol, abstol, dtol,
                  0 if x0 is None else 1, threads, orthog, refine, inner_its, 0, inner_omega)
    res = Result()
    hist = np.zeros(max_it + 2)
    lib().spo_fgmres(C.byref(op), C.byref(opt), b, x, C.byref(res), hist, len(hist))
    return x, dict(its=res.its, reason=res.reason, rnorm=res.rnorm, rnorm0=res.rnorm0,
                   history=hist[:res.hist_len].copy())


def time_spmv(A, reps, threads):
    x = np.sin(0.37 * np.arange(A.ncols))
    y = np.zeros(A.nrows)
    return lib().spo_time_spmv(A.nrows, A.rowptr, A.colidx, A.val, x, y, reps, threads)
