"""ctypes binding of libspk.so (include/spk.h, spk_ksp.h, spk_assembly.h).

The library is built in-tree by `make -C saddle_point_petsc_amd/csrc` (or
`__graft_entry__.build()`).  There is NO fallback: if libspk.so is missing the
import fails, and without a GPU `spk_create` fails -- nothing here computes on
the CPU.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libspk.so")

i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


class SpkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libspk error {code}: {msg}")
        self.code = code


class Opts(C.Structure):
    _fields_ = [
        ("restart", C.c_int32), ("max_it", C.c_int32), ("rtol", C.c_double),
        ("abstol", C.c_double), ("dtol", C.c_double), ("guess_nonzero", C.c_int32),
        ("orthog", C.c_int32), ("check_every", C.c_int32), ("fused", C.c_int32),
        ("cgs_refine", C.c_int32), ("single_reduce", C.c_int32), ("iteration_form", C.c_int32), ("reserved", C.c_int32),
    ]


class Result(C.Structure):
    _fields_ = [
        ("its", C.c_int32), ("reason", C.c_int32), ("rnorm", C.c_double),
        ("rnorm0", C.c_double), ("hist_len", C.c_int32), ("cycles", C.c_int32),
        ("solve_seconds", C.c_double),
    ]


ALLREDUCE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)
EXCHANGE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double), C.c_int64)
ALLGATHER_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)


class HostComm(C.Structure):
    _fields_ = [("user", C.c_void_p), ("allreduce", ALLREDUCE_CB), ("exchange", EXCHANGE_CB),
                ("allgather", ALLGATHER_CB)]


class CommInfo(C.Structure):
    """spk_comm_info (include/spk.h)."""
    _fields_ = [
        ("rank", C.c_int32), ("nranks", C.c_int32), ("peer_enabled", C.c_int32), ("window_tier", C.c_int32),
        ("self_test_ok", C.c_int32), ("halo_mode", C.c_int32), ("halo_fused", C.c_int32), ("device", C.c_int32),
        ("n_allreduce_fused", C.c_int64), ("n_allreduce_kernel", C.c_int64), ("n_allreduce_inner", C.c_int64),
        ("n_halo_fused", C.c_int64), ("n_halo_kernel", C.c_int64), ("n_halo_inner", C.c_int64),
        ("wait_ticks", C.c_uint64 * 4), ("wait_count", C.c_uint64 * 4),
        ("backend", C.c_char * 32), ("inner_backend", C.c_char * 32), ("why", C.c_char * 256),
    ]


class MatCSR(C.Structure):
    _fields_ = [
        ("row_begin", C.c_int64), ("nrows_local", C.c_int32), ("pad", C.c_int32),
        ("ncols_global", C.c_int64), ("rowptr", C.c_void_p), ("colidx", C.c_void_p),
        ("val", C.c_void_p),
    ]


def hip_runtimes_mapped():
    """Paths of every libamdhip64 mapped into this process (two = two HIP runtimes: the second finds no device)."""
    try:
        with open("/proc/self/maps") as fh:
            return sorted({ln.split()[-1] for ln in fh if "libamdhip64" in ln})
    except OSError:
        return []


def _preload_hip_runtime():
    """ONE HIP runtime per process whatever the import order.  The torch wheel carries a private libamdhip64.so (SONAME
    libamdhip64.so.7, the name libspk.so needs).  Loaded first, it satisfies libspk's dependency by soname; loaded SECOND --
    `import saddle_point_petsc_amd` before `import torch` -- the process used to map two runtimes, and the second one found
    no device ("no ROCm-capable device").  So where torch is installed its copy is mapped here, before libspk.so, without
    importing torch; a later `import torch` finds it already loaded.  SPK_HIP_RUNTIME=system keeps /opt/rocm's."""
    if os.environ.get("SPK_HIP_RUNTIME") == "system":
        return None
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(path):
        return None
    try:
        return C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError:
        return None


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(hipcc, gfx950).  saddle_point_petsc_amd has no CPU fallback.")
    global _hip_preloaded
    _hip_preloaded = _preload_hip_runtime()
    L = C.CDLL(LIB_PATH)
    rts = hip_runtimes_mapped()
    if len(rts) > 1:
        raise ImportError("two HIP runtimes are mapped into this process (" + ", ".join(rts) + "): libspk.so would not find the "
                          "GPU.  Import saddle_point_petsc_amd before anything else that loads a HIP runtime of its own, or set "
                          "LD_LIBRARY_PATH so that both resolve libamdhip64.so.7 to the same file.")
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    L.spk_version.restype = C.c_int
    L.spk_last_error.restype = C.c_char_p
    L.spk_last_error.argtypes = [vp]
    L.spk_default_opts.argtypes = [C.POINTER(Opts)]
    L.spk_default_opts.restype = None
    L.spk_create.argtypes = [C.POINTER(vp), C.c_int]
    L.spk_destroy.argtypes = [vp]
    L.spk_comm_unique_id.argtypes = [C.c_char_p]
    L.spk_comm_init_rccl.argtypes = [vp, C.c_int, C.c_int, C.c_char_p]
    L.spk_comm_init_host.argtypes = [vp, C.c_int, C.c_int, C.POINTER(HostComm)]
    L.spk_local_group_create.argtypes = [C.POINTER(vp), C.c_int]
    L.spk_local_group_destroy.argtypes = [vp]
    L.spk_comm_init_local.argtypes = [vp, vp, C.c_int]
    L.spk_comm_enable_peer.argtypes = [vp, C.POINTER(i32)]
    L.spk_comm_backend.restype = C.c_char_p
    L.spk_comm_backend.argtypes = [vp]
    L.spk_comm_get_info.argtypes = [vp, C.POINTER(CommInfo)]
    L.spk_debug_peer_allreduce_loopback.argtypes = [vp, C.c_int, C.c_int, C.c_int, f64p, f64p]
    L.spk_debug_finish_timeout.argtypes = [vp, C.c_int]
    L.spk_debug_set_wait_bound.argtypes = [vp, C.c_uint32]
    L.spk_debug_time_products.argtypes = [vp, C.c_int32]
    L.spk_get_product_timing.argtypes = [vp, C.POINTER(C.c_int32)] + [C.POINTER(C.c_double)] * 4 + [C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.spk_set_block.argtypes = [vp, C.c_int, i64, i32, i64, i32p, i32p, f64p]
    L.spk_pc_setup.argtypes = [vp, C.c_int, C.c_int]
    L.spk_pc_set_inner.argtypes = [vp, C.c_int, C.c_double]
    L.spk_get_schur_diag.argtypes = [vp, f64p]
    L.spk_get_jacobi_diag.argtypes = [vp, f64p]
    L.spk_get_bd_planes.argtypes = [vp, C.POINTER(i32)]
    L.spk_mult.argtypes = [vp, f64p, f64p, C.c_int]
    L.spk_pc_apply.argtypes = [vp, f64p, f64p, C.c_int]
    L.spk_fgmres.argtypes = [vp, f64p, f64p, C.c_int, C.POINTER(Opts), C.POINTER(Result), vp, i32]
    L.spk_vec_create.argtypes = [vp, i64, C.POINTER(vp)]
    L.spk_vec_destroy.argtypes = [vp, vp]
    L.spk_vec_set.argtypes = [vp, vp, f64p, i64]
    L.spk_vec_get.argtypes = [vp, vp, f64p, i64]
    L.spk_get_sizes.argtypes = [vp, C.POINTER(i64), C.POINTER(i32), C.POINTER(i32), C.POINTER(i64), C.POINTER(i32)]
    L.spk_kernel_mdot.argtypes = [vp, i64, i32, f64p, i64, f64p, f64p]
    L.spk_kernel_maxpy.argtypes = [vp, i64, i32, f64p, f64p, i64, f64p, C.POINTER(dbl)]
    L.spk_time_spmv.argtypes = [vp, C.c_int, C.c_int, C.POINTER(dbl)]
    L.spk_get_spmv_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i64)]
    L.spk_get_iteration_form.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.spk_get_spmv_models.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i32), C.POINTER(i32)]
    L.spk_time_kernel.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(dbl)]
    L.spk_partition_slab.argtypes = [i64, i64, C.c_int, C.c_int, C.POINTER(i64), C.POINTER(i64)]
    L.spk_partition_split.argtypes = [i64, i32, i32p, i32p, f64p] + [vp] * 7 + [C.POINTER(i64), C.POINTER(i64), C.POINTER(i32)]
    # assembly
    L.SpkAssemblySizes.argtypes = [C.c_int, C.c_int, C.POINTER(i64), C.POINTER(i64)]
    L.SpkAssemblySlabNnz.restype = i64
    L.SpkAssemblySlabNnz.argtypes = [C.c_int, C.c_int, i64, i64]
    L.SpkAssembleOperator_Laplace.argtypes = [C.c_int, C.c_int, i64, i64, i32p, i32p, f64p, vp, C.c_int, C.c_int]
    L.SpkConstraintsSlabNnz.restype = i64
    L.SpkConstraintsSlabNnz.argtypes = [C.c_int, C.c_int, i64, i64]
    L.SpkAssembleOperator_Constraints.argtypes = [C.c_int, C.c_int, i64, i64, i32p, i32p, f64p]
    L.SpkAssembleRHS_Constraints.argtypes = [f64p]
    L.SpkAssemblySizes3D.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(i64), C.POINTER(i64)]
    L.SpkAssemblySlabNnz3D.restype = i64
    L.SpkAssemblySlabNnz3D.argtypes = [C.c_int, C.c_int, C.c_int, i64, i64]
    L.SpkAssembleOperator_Laplace3D.argtypes = [C.c_int, C.c_int, C.c_int, i64, i64, i32p, i32p, f64p, vp, C.c_int, C.c_int]
    L.SpkConstraintsSlabNnz3D.restype = i64
    L.SpkConstraintsSlabNnz3D.argtypes = [C.c_int, C.c_int, C.c_int, i64, i64]
    L.SpkAssembleOperator_Constraints3D.argtypes = [C.c_int, C.c_int, C.c_int, i64, i64, i32p, i32p, f64p]
    L.SpkAssembleRHS_Constraints3D.argtypes = [f64p]
    L.SpkDivergenceSlabNnz3D.restype = i64
    L.SpkDivergenceSlabNnz3D.argtypes = [C.c_int, C.c_int, C.c_int, i64, i64]
    L.SpkAssembleOperator_Divergence3D.argtypes = [C.c_int, C.c_int, C.c_int, i64, i64, i32p, i32p, f64p]
    L.SpkWriteVTK.argtypes = [C.c_int, C.c_int, f64p, C.c_char_p]
    L.SpkFormStressOperatorQ12D.argtypes = [f64p, f64p, f64p]
    L.SpkFormLaplaceRHSQ12D.argtypes = [f64p, f64p]
    # KSP facade
    L.SpkKSPCreate.argtypes = [C.c_int, C.POINTER(vp)]
    L.SpkKSPSetCommRCCL.argtypes = [vp, C.c_int, C.c_int, C.c_char_p]
    L.SpkKSPSetOperators.argtypes = [vp, C.POINTER(MatCSR), C.POINTER(MatCSR)]
    L.SpkKSPSetFromOptions.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p)]
    L.SpkKSPSetUp.argtypes = [vp]
    L.SpkKSPSolve.argtypes = [vp, f64p, f64p]
    L.SpkKSPDestroy.argtypes = [C.POINTER(vp)]
    L.SpkKSPGetIterationNumber.argtypes = [vp, C.POINTER(i32)]
    L.SpkKSPGetConvergedReason.argtypes = [vp, C.POINTER(i32)]
    L.SpkKSPGetResidualNorm.argtypes = [vp, C.POINTER(dbl)]
    L.SpkKSPGetResidualHistory.argtypes = [vp, C.POINTER(C.POINTER(dbl)), C.POINTER(i32)]
    L.SpkKSPGetSolveTime.argtypes = [vp, C.POINTER(dbl)]
    L.SpkKSPGetOptions.argtypes = [vp, C.POINTER(Opts), C.POINTER(i32), C.POINTER(i32)]
    L.SpkKSPGetContext.argtypes = [vp, C.POINTER(vp)]
    L.SpkKSPGetError.restype = C.c_char_p
    L.SpkKSPGetError.argtypes = [vp]
    L.SpkKSPConvergedReasonName.restype = C.c_char_p
    L.SpkKSPConvergedReasonName.argtypes = [i32]
    return L


_hip_preloaded = None
lib = _load()
