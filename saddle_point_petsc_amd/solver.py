"""Thin object wrappers over the C ABI (include/spk.h) and the KSP facade
(include/spk_ksp.h).  `KSP` mirrors the reference's call site
/root/reference/src/SaddlePointProblem.c:65-72 (petsc4py-style method names)."""
import ctypes as C

import numpy as np

from ._lib import lib, Opts, Result, MatCSR, SpkError

PC_NONE, PC_JACOBI, PC_SCHUR = 0, 1, 2
SCHUR_DIAG, SCHUR_LOWER, SCHUR_UPPER, SCHUR_FULL = 0, 1, 2, 3
BLOCK_A00, BLOCK_A10 = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1


def default_opts(**kw):
    o = Opts()
    lib.spk_default_opts(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError(f"unknown solver option {k}")
        setattr(o, k, v)
    return o


def unique_id():
    buf = C.create_string_buffer(128)
    rc = lib.spk_comm_unique_id(buf)
    if rc != 0:
        raise SpkError(rc, lib.spk_last_error(None).decode())
    return buf.raw


class LocalGroup:
    """In-process logical ranks on one device (parity tests only)."""

    def __init__(self, nranks):
        self.h = C.c_void_p()
        rc = lib.spk_local_group_create(C.byref(self.h), nranks)
        if rc != 0:
            raise SpkError(rc, "spk_local_group_create")
        self.nranks = nranks

    def close(self):
        if self.h:
            lib.spk_local_group_destroy(self.h)
            self.h = C.c_void_p()


class Context:
    """One solver context (= one KSP) on one GPU."""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        rc = lib.spk_create(C.byref(self.h), device)
        if rc != 0:
            raise SpkError(rc, lib.spk_last_error(None).decode())

    def _chk(self, rc):
        if rc != 0:
            raise SpkError(rc, lib.spk_last_error(self.h).decode())

    def close(self):
        if self.h:
            lib.spk_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def comm_init_rccl(self, rank, nranks, id128):
        self._chk(lib.spk_comm_init_rccl(self.h, rank, nranks, id128))

    def comm_init_torch(self, dist, rank, nranks):
        """Rehearsal transport over an initialised torch.distributed group (gloo): lets several
        processes share one GPU, which RCCL refuses.  Slow: every collective goes through the host."""
        import torch
        from ._lib import HostComm, ALLREDUCE_CB, EXCHANGE_CB, ALLGATHER_CB

        def allreduce(_u, buf, count):
            a = np.ctypeslib.as_array(buf, shape=(count,))
            t = torch.from_numpy(a.copy())
            dist.all_reduce(t)
            a[:] = t.numpy()
            return 0

        def exchange(_u, peer, send, nsend, recv, nrecv):
            reqs = []
            if nsend:
                reqs.append(dist.isend(torch.from_numpy(np.ctypeslib.as_array(send, shape=(nsend,)).copy()), peer))
            rt = torch.zeros(max(nrecv, 1), dtype=torch.float64)
            if nrecv:
                reqs.append(dist.irecv(rt[:nrecv], peer))
            for r in reqs:
                r.wait()
            if nrecv:
                np.ctypeslib.as_array(recv, shape=(nrecv,))[:] = rt[:nrecv].numpy()
            return 0

        def allgather(_u, inp, out, nbytes):
            mine = torch.frombuffer(bytearray(C.string_at(inp, nbytes)), dtype=torch.uint8)
            parts = [torch.zeros(nbytes, dtype=torch.uint8) for _ in range(nranks)]
            dist.all_gather(parts, mine)
            C.memmove(out, b"".join(bytes(p.numpy().tobytes()) for p in parts), nbytes * nranks)
            return 0

        self._cbs = (ALLREDUCE_CB(allreduce), EXCHANGE_CB(exchange), ALLGATHER_CB(allgather))   # keep alive
        self._hc = HostComm(None, *self._cbs)
        self._chk(lib.spk_comm_init_host(self.h, rank, nranks, C.byref(self._hc)))

    def comm_init_local(self, group, rank):
        self._chk(lib.spk_comm_init_local(self.h, group.h, rank))

    def comm_enable_peer(self):
        """Peer-store collectives over xGMI on top of the communicator set before (collective).
        Returns True when they are active, False when the previous backend stays (see
        spk_comm_enable_peer in include/spk.h)."""
        on = C.c_int32()
        self._chk(lib.spk_comm_enable_peer(self.h, C.byref(on)))
        return bool(on.value)

    def last_error(self):
        return lib.spk_last_error(self.h).decode()

    def comm_backend(self):
        return lib.spk_comm_backend(self.h).decode()

    def comm_info(self):
        """Per-rank diagnostics of the communicator (spk_comm_get_info): backend, why the peer-store
        backend is off if it is, window memory kind, collective counts, device-side wait times."""
        from ._lib import CommInfo
        ci = CommInfo()
        self._chk(lib.spk_comm_get_info(self.h, C.byref(ci)))
        kinds = ("allreduce_after_mdot", "allreduce_after_maxpy", "allreduce_standalone", "halo")
        return dict(rank=ci.rank, nranks=ci.nranks, device=ci.device, backend=ci.backend.decode(),
                    inner_backend=ci.inner_backend.decode(), peer_enabled=bool(ci.peer_enabled),
                    window_memory={0: "uncached", 1: "fine-grained", 2: "plain"}.get(ci.window_tier, "none"),
                    self_test_ok=bool(ci.self_test_ok),
                    halo={0: "none", 1: "granules", 2: "bulk", 3: "inner-backend"}[ci.halo_mode],
                    halo_fused=bool(ci.halo_fused),
                    allreduce=dict(fused=ci.n_allreduce_fused, kernel=ci.n_allreduce_kernel, inner=ci.n_allreduce_inner),
                    halo_exchanges=dict(fused=ci.n_halo_fused, kernel=ci.n_halo_kernel, inner=ci.n_halo_inner),
                    # 100 MHz ticks -> microseconds, mean per wait
                    wait_us={k: (ci.wait_ticks[i] / 100.0 / ci.wait_count[i] if ci.wait_count[i] else None)
                             for i, k in enumerate(kinds)},
                    wait_count={k: int(ci.wait_count[i]) for i, k in enumerate(kinds)},
                    why=ci.why.decode())

    def debug_peer_allreduce_loopback(self, vals, rounds=6):
        """Test hook: nranks x count inputs -> nranks x count rank-ordered sums (one launch, no IPC)."""
        vals = np.ascontiguousarray(vals, np.float64)
        out = np.zeros_like(vals)
        self._chk(lib.spk_debug_peer_allreduce_loopback(self.h, vals.shape[0], vals.shape[1], rounds,
                                                        vals.reshape(-1), out.reshape(-1)))
        return out

    def debug_finish_timeout(self, timeout_ms=50):
        """Test hook: a reduction with a partial that never arrives; raises SpkError (SPK_ERR_HIP)."""
        self._chk(lib.spk_debug_finish_timeout(self.h, timeout_ms))

    def time_products(self, max_launches):
        """HIP events around the product launches of the next solves' iterations (0: off); see include/spk.h"""
        self._chk(lib.spk_debug_time_products(self.h, max_launches))

    def product_timing(self):
        n, ng, gm = C.c_int32(), C.c_int32(), C.c_double()
        v = [C.c_double() for _ in range(4)]
        self._chk(lib.spk_get_product_timing(self.h, C.byref(n), *[C.byref(x) for x in v], C.byref(ng), C.byref(gm)))
        return dict(launches=n.value, mean_ms=v[0].value, median_ms=v[1].value, min_ms=v[2].value, max_ms=v[3].value,
                    gated=ng.value, gated_mean_ms=gm.value)

    def debug_set_wait_bound(self, ticks=0):
        """Test hook: bound of the device-side waits in 100 MHz ticks (0: the default, 4 s)."""
        self._chk(lib.spk_debug_set_wait_bound(self.h, ticks))

    def set_block(self, which, A):
        nrows = A.nrows
        self._chk(lib.spk_set_block(self.h, which, A.row_begin if which == BLOCK_A00 else 0, nrows,
                                    A.ncols, A.rowptr, A.colidx, A.val))

    def pc_setup(self, pc_type, schur_fact=SCHUR_FULL, inner_sweeps=0, inner_omega=1.0):
        """inner_sweeps > 0: FP32 damped-Jacobi Richardson sweeps stand for diag(A)^-1."""
        self._chk(lib.spk_pc_set_inner(self.h, inner_sweeps, inner_omega))
        self._chk(lib.spk_pc_setup(self.h, pc_type, schur_fact))

    def sizes(self):
        ng, nl, m, nnz, gh = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int64(), C.c_int32()
        lib.spk_get_sizes(self.h, C.byref(ng), C.byref(nl), C.byref(m), C.byref(nnz), C.byref(gh))
        return dict(n_global=ng.value, n_local=nl.value, m=m.value, nnz_local=nnz.value, n_ghost=gh.value)

    def spmv_info(self):
        fmt, b = C.c_int32(), C.c_int64()
        lib.spk_get_spmv_info(self.h, C.byref(fmt), C.byref(b))
        return dict(format={0: "csr", 1: "bcsr2x2", 2: "bcsr3x3", 3: "dict2x2", 4: "dict3x3"}[fmt.value], layout_bytes=b.value)

    def iteration_form(self):
        """(form, single_reduce) of the last fgmres on this context: the SPK_ITER_* actually run, -1 = step-by-step path."""
        f, sr = C.c_int32(), C.c_int32()
        self._chk(lib.spk_get_iteration_form(self.h, C.byref(f), C.byref(sr)))
        return f.value, sr.value

    def spmv_models(self):
        """Bytes of one product y = A x in the CSR, blocked and row-pattern-dictionary layouts (0: layout absent)."""
        a, b, d, p, q = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32()
        self._chk(lib.spk_get_spmv_models(self.h, C.byref(a), C.byref(b), C.byref(d), C.byref(p), C.byref(q)))
        return dict(csr_bytes=a.value, blocked_bytes=b.value, dict_bytes=d.value, patterns=p.value, blocks=q.value)

    def _n(self):
        s = self.sizes()
        return s["n_local"] + s["m"]

    def schur_diag(self):
        out = np.zeros(self.sizes()["m"])
        self._chk(lib.spk_get_schur_diag(self.h, out))
        return out

    def bd_planes(self):
        v = C.c_int32()
        self._chk(lib.spk_get_bd_planes(self.h, C.byref(v)))
        return v.value

    def jacobi_diag(self):
        out = np.zeros(self.sizes()["n_local"])
        self._chk(lib.spk_get_jacobi_diag(self.h, out))
        return out

    def mult(self, x):
        x = np.ascontiguousarray(x, np.float64)
        assert x.shape == (self._n(),)
        y = np.zeros_like(x)
        self._chk(lib.spk_mult(self.h, x, y, MEM_HOST))
        return y

    def pc_apply(self, x):
        x = np.ascontiguousarray(x, np.float64)
        assert x.shape == (self._n(),)
        y = np.zeros_like(x)
        self._chk(lib.spk_pc_apply(self.h, x, y, MEM_HOST))
        return y

    def fgmres(self, b, x0=None, **kw):
        b = np.ascontiguousarray(b, np.float64)
        assert b.shape == (self._n(),)
        o = default_opts(**kw)
        x = np.zeros_like(b)
        if x0 is not None:
            x[:] = x0
            o.guess_nonzero = 1
        res = Result()
        cap = int(min(o.max_it + 2, 1 << 22))
        hist = np.zeros(cap)
        self._chk(lib.spk_fgmres(self.h, b, x, MEM_HOST, C.byref(o), C.byref(res), hist.ctypes.data, cap))
        return x, dict(its=res.its, reason=res.reason, rnorm=res.rnorm, rnorm0=res.rnorm0,
                       cycles=res.cycles, solve_seconds=res.solve_seconds,
                       history=hist[:res.hist_len].copy())

    # ---- device-resident vectors (inputs already in HBM when a solve starts)
    def vec_create(self, host=None, n=None):
        n = len(host) if host is not None else n
        p = C.c_void_p()
        self._chk(lib.spk_vec_create(self.h, n, C.byref(p)))
        if host is not None:
            self._chk(lib.spk_vec_set(self.h, p, np.ascontiguousarray(host, np.float64), n))
        return p

    def vec_get(self, p, n):
        out = np.zeros(n)
        self._chk(lib.spk_vec_get(self.h, p, out, n))
        return out

    def vec_destroy(self, p):
        self._chk(lib.spk_vec_destroy(self.h, p))

    def fgmres_device(self, b_dev, x_dev, **kw):
        """KSPSolve on vectors that already live in device memory."""
        o = default_opts(**kw)
        res = Result()
        cap = int(min(o.max_it + 2, 1 << 22))
        hist = np.zeros(cap)
        f = lib.spk_fgmres
        old = f.argtypes
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Opts), C.POINTER(Result), C.c_void_p, C.c_int32]
        try:
            rc = f(self.h, b_dev, x_dev, MEM_DEVICE, C.byref(o), C.byref(res), hist.ctypes.data, cap)
        finally:
            f.argtypes = old
        self._chk(rc)
        return dict(its=res.its, reason=res.reason, rnorm=res.rnorm, rnorm0=res.rnorm0,
                    cycles=res.cycles, solve_seconds=res.solve_seconds,
                    history=hist[:res.hist_len].copy())

    def mdot(self, V, w):
        V = np.ascontiguousarray(V, np.float64)
        w = np.ascontiguousarray(w, np.float64)
        nv, n = V.shape
        h = np.zeros(nv + 1)
        self._chk(lib.spk_kernel_mdot(self.h, n, nv, V.reshape(-1), n, w, h))
        return h[:nv], h[nv]

    def maxpy(self, a, V, w):
        V = np.ascontiguousarray(V, np.float64)
        a = np.ascontiguousarray(a, np.float64)
        w = np.array(w, np.float64)
        nv, n = V.shape
        nrm2 = C.c_double()
        self._chk(lib.spk_kernel_maxpy(self.h, n, nv, a, V.reshape(-1), n, w, C.byref(nrm2)))
        return w, nrm2.value

    def time_spmv(self, warmup=5, reps=50):
        ms = C.c_double()
        self._chk(lib.spk_time_spmv(self.h, warmup, reps, C.byref(ms)))
        return ms.value


def _time_kernel(self, which, nv=0, warmup=5, reps=50):
    ms = C.c_double()
    self._chk(lib.spk_time_kernel(self.h, which.encode(), nv, warmup, reps, C.byref(ms)))
    return ms.value


Context.time_kernel = _time_kernel


def _mat(A):
    m = MatCSR()
    m.row_begin, m.nrows_local, m.ncols_global = A.row_begin, A.nrows, A.ncols
    m.rowptr, m.colidx, m.val = A.rowptr.ctypes.data, A.colidx.ctypes.data, A.val.ctypes.data
    return m


class KSP:
    """Mirror of the reference's solver object: create / setOperators /
    setFromOptions / setUp / solve / destroy (SaddlePointProblem.c:65-72)."""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        rc = lib.SpkKSPCreate(device, C.byref(self.h))
        if rc != 0:
            msg = lib.SpkKSPGetError(self.h).decode() if self.h else "SpkKSPCreate failed"
            lib.SpkKSPDestroy(C.byref(self.h))
            raise SpkError(rc, msg)
        self._keep = None

    def _chk(self, rc):
        if rc != 0:
            raise SpkError(rc, lib.SpkKSPGetError(self.h).decode())

    def setCommRCCL(self, rank, nranks, id128):
        self._chk(lib.SpkKSPSetCommRCCL(self.h, rank, nranks, id128))

    def setOperators(self, A, B=None):
        a = _mat(A)
        b = _mat(B) if B is not None else None
        self._chk(lib.SpkKSPSetOperators(self.h, C.byref(a), C.byref(b) if b is not None else None))
        self._n = A.nrows + (B.nrows if B is not None else 0)

    def setFromOptions(self, options):
        """options: PETSc-style string or list, e.g. '-ksp_type fgmres -ksp_rtol 1e-8'."""
        args = options.split() if isinstance(options, str) else list(options)
        arr = (C.c_char_p * max(len(args), 1))(*[a.encode() for a in args])
        self._chk(lib.SpkKSPSetFromOptions(self.h, len(args), arr))

    def setUp(self):
        self._chk(lib.SpkKSPSetUp(self.h))

    def solve(self, b, x=None):
        b = np.ascontiguousarray(b, np.float64)
        assert b.shape == (self._n,)
        x = np.zeros_like(b) if x is None else x
        self._chk(lib.SpkKSPSolve(self.h, b, x))
        return x

    def getIterationNumber(self):
        v = C.c_int32()
        lib.SpkKSPGetIterationNumber(self.h, C.byref(v))
        return v.value

    def getConvergedReason(self):
        v = C.c_int32()
        lib.SpkKSPGetConvergedReason(self.h, C.byref(v))
        return v.value

    def getResidualNorm(self):
        v = C.c_double()
        lib.SpkKSPGetResidualNorm(self.h, C.byref(v))
        return v.value

    def getSolveTime(self):
        v = C.c_double()
        lib.SpkKSPGetSolveTime(self.h, C.byref(v))
        return v.value

    def getConvergenceHistory(self):
        p, n = C.POINTER(C.c_double)(), C.c_int32()
        lib.SpkKSPGetResidualHistory(self.h, C.byref(p), C.byref(n))
        return np.array([p[i] for i in range(n.value)])

    def getOptions(self):
        o, pc, sf = Opts(), C.c_int32(), C.c_int32()
        lib.SpkKSPGetOptions(self.h, C.byref(o), C.byref(pc), C.byref(sf))
        return o, pc.value, sf.value

    def destroy(self):
        if self.h:
            lib.SpkKSPDestroy(C.byref(self.h))
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.destroy()
