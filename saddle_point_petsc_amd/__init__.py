"""saddle_point_petsc_amd -- MI355X-native replacement for the KSPSolve hot path
of p-m-mueller/saddle_point_petsc (FGMRES + Schur fieldsplit on
[A B^T; B 0]).  The product is the C-ABI library libspk.so (include/spk.h);
this package is its ctypes front end plus the host-side mirror of the
reference's call site.  No CPU fallback: importing needs the built library and
solving needs a GPU."""
from ._lib import lib, SpkError, LIB_PATH  # noqa: F401
from .csr import CSR  # noqa: F401
from .assembly import (  # noqa: F401
    AssembleOperator_Laplace, AssembleOperator_Constraints, FormStressOperatorQ12D,
    FormLaplaceRHSQ12D, grid_sizes, partition_slab, WriteVTK,
    AssembleOperator_Laplace3D, AssembleOperator_Constraints3D, AssembleOperator_Divergence3D, partition_slab3d,
)
from .solver import (  # noqa: F401
    Context, KSP, LocalGroup, default_opts, unique_id,
    PC_NONE, PC_JACOBI, PC_SCHUR, SCHUR_DIAG, SCHUR_LOWER, SCHUR_UPPER, SCHUR_FULL,
    BLOCK_A00, BLOCK_A10,
)
