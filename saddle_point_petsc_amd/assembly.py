"""Host assembler front end (include/spk_assembly.h): the PETSc-free
counterpart of /root/reference/src/Discretization.c used to generate A, f, B, g.
Function names follow the reference's (Discretization.h:36-41)."""
import ctypes as C

import numpy as np

from ._lib import lib, SpkError
from .csr import CSR


def _chk(rc, what):
    if rc != 0:
        raise SpkError(rc, what)


def grid_sizes(mx, my=None):
    my = mx if my is None else my
    n, nnz = C.c_int64(), C.c_int64()
    _chk(lib.SpkAssemblySizes(mx, my, C.byref(n), C.byref(nnz)), "SpkAssemblySizes")
    return n.value, nnz.value


def AssembleOperator_Laplace(mx, my=None, row_begin=0, row_end=None, apply_bc=True, with_rhs=True, nthreads=0):
    """A (CSR slab, global columns) and f for rows [row_begin,row_end) of an
    mx x my NODE grid.  Restates AssembleOperator_Laplace + AssembleRHS_Laplace
    + ApplyBC_Laplace (Discretization.c:130-274)."""
    my = mx if my is None else my
    n, _ = grid_sizes(mx, my)
    row_end = n if row_end is None else row_end
    nnz = lib.SpkAssemblySlabNnz(mx, my, row_begin, row_end)
    if nnz < 0:
        raise SpkError(-1, "row range must consist of whole node lines")
    nl = row_end - row_begin
    rowptr = np.zeros(nl + 1, np.int32)
    colidx = np.zeros(nnz, np.int32)
    val = np.zeros(nnz)
    f = np.zeros(nl) if with_rhs else None
    _chk(lib.SpkAssembleOperator_Laplace(mx, my, row_begin, row_end, rowptr, colidx, val,
                                         f.ctypes.data if with_rhs else None, int(apply_bc), nthreads),
         "SpkAssembleOperator_Laplace")
    return CSR(rowptr, colidx, val, n, row_begin), f


def AssembleOperator_Constraints(mx, my=None, row_begin=0, row_end=None):
    """Build-defined B (4 x n, columns restricted to [row_begin,row_end)) and g;
    the reference's assemblers are empty stubs (Discretization.c:277-290)."""
    my = mx if my is None else my
    n, _ = grid_sizes(mx, my)
    row_end = n if row_end is None else row_end
    nnz = lib.SpkConstraintsSlabNnz(mx, my, row_begin, row_end)
    if nnz < 0:
        raise SpkError(-1, "column range must consist of whole node lines (grid >= 3x3)")
    rowptr = np.zeros(5, np.int32)
    colidx = np.zeros(nnz, np.int32)
    val = np.zeros(nnz)
    _chk(lib.SpkAssembleOperator_Constraints(mx, my, row_begin, row_end, rowptr, colidx, val),
         "SpkAssembleOperator_Constraints")
    g = np.zeros(4)
    _chk(lib.SpkAssembleRHS_Constraints(g), "SpkAssembleRHS_Constraints")
    return CSR(rowptr, colidx, val, n, 0), g


def FormStressOperatorQ12D(xe, coeff=None):
    Ke = np.zeros(64)
    coeff = np.ones(4) if coeff is None else np.ascontiguousarray(coeff, np.float64)
    _chk(lib.SpkFormStressOperatorQ12D(np.ascontiguousarray(xe, np.float64), coeff, Ke), "SpkFormStressOperatorQ12D")
    return Ke.reshape(8, 8)


def FormLaplaceRHSQ12D(xe):
    Fe = np.zeros(8)
    _chk(lib.SpkFormLaplaceRHSQ12D(np.ascontiguousarray(xe, np.float64), Fe), "SpkFormLaplaceRHSQ12D")
    return Fe


def AssembleOperator_Laplace3D(mx, my=None, mz=None, row_begin=0, row_end=None, apply_bc=True, nthreads=0):
    """BUILD-DEFINED 3-D input generator (the reference is 2-D only; see include/spk_assembly.h):
    A (CSR slab of whole node planes, global columns) and f on an mx x my x mz node grid, dof 3."""
    my = mx if my is None else my
    mz = mx if mz is None else mz
    n, nnz = C.c_int64(), C.c_int64()
    _chk(lib.SpkAssemblySizes3D(mx, my, mz, C.byref(n), C.byref(nnz)), "SpkAssemblySizes3D")
    row_end = n.value if row_end is None else row_end
    nz = lib.SpkAssemblySlabNnz3D(mx, my, mz, row_begin, row_end)
    if nz < 0:
        raise SpkError(-1, "row range must consist of whole node planes")
    nl = row_end - row_begin
    rowptr = np.zeros(nl + 1, np.int32)
    colidx = np.zeros(nz, np.int32)
    val = np.zeros(nz)
    f = np.zeros(nl)
    _chk(lib.SpkAssembleOperator_Laplace3D(mx, my, mz, row_begin, row_end, rowptr, colidx, val, f.ctypes.data,
                                           int(apply_bc), nthreads), "SpkAssembleOperator_Laplace3D")
    return CSR(rowptr, colidx, val, n.value, row_begin), f


def AssembleOperator_Constraints3D(mx, my=None, mz=None, row_begin=0, row_end=None):
    my = mx if my is None else my
    mz = mx if mz is None else mz
    n = 3 * mx * my * mz
    row_end = n if row_end is None else row_end
    nz = lib.SpkConstraintsSlabNnz3D(mx, my, mz, row_begin, row_end)
    if nz < 0:
        raise SpkError(-1, "column range must consist of whole node planes (grid >= 3^3)")
    rowptr = np.zeros(7, np.int32)
    colidx = np.zeros(nz, np.int32)
    val = np.zeros(nz)
    _chk(lib.SpkAssembleOperator_Constraints3D(mx, my, mz, row_begin, row_end, rowptr, colidx, val),
         "SpkAssembleOperator_Constraints3D")
    g = np.zeros(6)
    _chk(lib.SpkAssembleRHS_Constraints3D(g), "SpkAssembleRHS_Constraints3D")
    return CSR(rowptr, colidx, val, n, 0), g


def AssembleOperator_Divergence3D(mx, my=None, mz=None, row_begin=0, row_end=None):
    """BUILD-DEFINED discrete divergence block of the 3-D grid (include/spk_assembly.h): one row per
    hexahedron, all rows restricted to the columns [row_begin, row_end) -- a general sparse A10 block."""
    my = mx if my is None else my
    mz = mx if mz is None else mz
    n = 3 * mx * my * mz
    row_end = n if row_end is None else row_end
    nz = lib.SpkDivergenceSlabNnz3D(mx, my, mz, row_begin, row_end)
    if nz < 0:
        raise SpkError(-1, "column range must consist of whole node planes (grid >= 3^3)")
    m = (mx - 1) * (my - 1) * (mz - 1)
    rowptr = np.zeros(m + 1, np.int32)
    colidx = np.zeros(nz, np.int32)
    val = np.zeros(nz)
    _chk(lib.SpkAssembleOperator_Divergence3D(mx, my, mz, row_begin, row_end, rowptr, colidx, val),
         "SpkAssembleOperator_Divergence3D")
    return CSR(rowptr, colidx, val, n, 0)


def WriteVTK(mx, my, u, filename):
    """Legacy-VTK file with the node grid and the solution field (the reference's
    WriteVTK, SaddlePointProblem.c:22, never wrote the field)."""
    u = np.ascontiguousarray(u, np.float64)
    if u.shape != (2 * mx * my,):
        raise ValueError("u must have 2*mx*my entries")
    _chk(lib.SpkWriteVTK(mx, my, u, str(filename).encode()), "SpkWriteVTK")


def partition_slab(mx, my, rank, nranks):
    """Rows of `rank` when the my node lines are dealt in contiguous slabs."""
    b, e = C.c_int64(), C.c_int64()
    _chk(lib.spk_partition_slab(my, 2 * mx, rank, nranks, C.byref(b), C.byref(e)), "spk_partition_slab")
    return b.value, e.value


def partition_slab3d(mx, my, mz, rank, nranks):
    """Rows of `rank` when the mz node planes are dealt in contiguous slabs (z-slabs)."""
    b, e = C.c_int64(), C.c_int64()
    _chk(lib.spk_partition_slab(mz, 3 * mx * my, rank, nranks, C.byref(b), C.byref(e)), "spk_partition_slab")
    return b.value, e.value
