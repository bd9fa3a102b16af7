#!/usr/bin/env python3
"""bench.py -- FGMRES iterations/s on the saddle-point system + achieved HBM
GB/s of the A-block SpMV (BASELINE.json's metric), on N GPUs of one node.

    python bench.py --gpus 1 --steps 300 --warmup 30
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (config.workload): the 1024 x 1024 DMDA node grid of BASELINE.json's
configs (2 097 152 rows, 37 699 600 stored non-zeros in A, 4 constraint rows),
K = [A B^T; B 0], FGMRES(30) with classical Gram-Schmidt and the full Schur
fieldsplit preconditioner S^ = diag(B diag(A)^-1 B^T): the call the reference
makes at /root/reference/src/SaddlePointProblem.c:70 with the finished nest.
The whole system fits one MI355X; N > 1 row-partitions the SAME system (strong
scaling) with RCCL all-reduces and halo send/recv on the solver's stream.

A "step" is one FGMRES iteration.  Exactly K iterations are timed: rtol = atol
= 0 so the solve cannot stop early, and spk_fgmres stops at max_it = K.  b and x
live in device memory before the clock starts.  The timed region contains the
whole KSPSolve for those K iterations (||b||, the restarts' true residuals, the
solution update).

Only the cpu_baseline leg touches oracle/ (the CPU restatement), as a reported
baseline on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0          # measured float4 copy ceiling, same guide
METRIC = "FGMRES iterations/sec + achieved HBM GB/s on A-block SpMV, 1/2/4/8 MI355X"


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:  # noqa: BLE001
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:  # noqa: BLE001
            pass
    return n


def cpu_threads():
    """Threads the CPU legs use: every core the process may use (SPK_CPU_THREADS overrides; N ranks share the host)."""
    cap = os.environ.get("SPK_CPU_THREADS")
    world = max(1, int(os.environ.get("WORLD_SIZE", "1")))
    return int(cap) if cap else max(1, host_cores() // world)


def spmv_bytes(nrows, nnz):
    """Algorithmic bytes of one CSR SpMV:
    12 B per stored non-zero + 4 B row pointer + x once + y once per row."""
    return 12 * nnz + 4 * (nrows + 1) + 16 * nrows


def solve_bytes(n, nnz, nnzB, restart, steps, planes, m, spmv_matrix_bytes=None, unnormalised=True):
    """Algorithmic bytes of ONE spk_fgmres call that executes exactly `steps` iterations of the
    head-kernel paths, summed over the iterations and restart cycles ACTUALLY
    executed -- iteration i of the solve has j = i mod restart basis vectors behind it, and a cycle end
    is paid once per started cycle -- so the figure is right for any --steps, not only for whole cycles.
    vec = 8 n bytes; planes = planes of B D streamed per pass (spk_get_bd_planes; 0: Jacobi on K = A);
    m = constraint rows.  Per iteration with j vectors behind it:
      head (VecScale + PC [+ B^T part])  : w', dinv, planes in; v, z [, c] out   (4 + [1] + planes) vec
      SpMV  y (+)= A z                   : matrix + 4 n (row pointers) + x + y out [+ y in]
      MDot                               : V_0..V_j and w                        (j + 2) vec
      MAXPY + norm [+ B D w']            : V_0..V_j, w in/out, planes            (j + 3 + planes) vec
    unnormalised (the default form, opts.iteration_form 0/5: no head launch after the first iteration of a cycle --
    the MAXPY pass also applies the preconditioner to the w' it holds, and nothing is normalised in memory):
      MDot [+ B D w~]                    : V_0..V_j, w, planes                   (j + 2 + planes) vec
      MAXPY + norm + next PCApply        : V_0..V_j, w, dinv, planes in; w', z [, c] out   (j + 5 + [1] + planes) vec
      (iteration 0 of a cycle still pays the head: (4 + [1] + planes) vec, and its MAXPY the same as above)
    Per started cycle with L iterations: ||r|| [+ B D r] (1 + m) vec, x += Z y (L + 2) vec, true residual
    (plain K x: matrix + B and B^T entries, b - K x: 5 vec).  Once per solve: ||b|| (1 vec).
    The matrix term is 12 B per stored non-zero (CSR) unless spmv_matrix_bytes gives the
    bytes of the layout the kernel really streams."""
    vec = 8 * n
    mat = 12 * nnz if spmv_matrix_bytes is None else spmv_matrix_bytes
    saddle = planes > 0
    total = vec                                              # ||b||
    done = 0
    while done < steps:
        L = min(restart, steps - done)
        total += (1 + (m if saddle else 0)) * vec            # cycle start
        for j in range(L):
            total += mat + 4 * (n + 1) + 2 * vec + (vec if saddle else 0)
            if unnormalised:
                if j == 0:
                    total += (4 + (1 if saddle else 0) + planes) * vec
                total += (j + 2 + planes) * vec + (j + 5 + (1 if saddle else 0) + planes) * vec
            else:
                total += (4 + (1 if saddle else 0) + planes) * vec
                total += (j + 2) * vec + (j + 3 + planes) * vec
        total += (L + 2) * vec                               # x += Z y
        total += mat + 4 * (n + 1) + 2 * vec + 2 * 12 * nnzB + 4 * n + 5 * vec   # true residual of the restart
        done += L
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--grid", type=int, default=1024, help="DMDA nodes per side")
    ap.add_argument("--dim", type=int, default=2, choices=[2, 3], help="3: build-defined 3-D grid (BASELINE config 5 "
                                                                       "shape, --grid nodes per side, dof 3)")
    ap.add_argument("--grid-y", type=int, default=0, help="node lines in y (default: square); e.g. 128 emulates one "
                                                          "rank's slab of the 8-GPU split on one GPU")
    ap.add_argument("--grid-z", type=int, default=0, help="--dim 3: node planes in z (default: --grid); e.g. --grid 256 --grid-z 32 "
                                                          "is one rank's z-slab of the 256^3 grid split 8 ways")
    ap.add_argument("--pc", default="schur-full", choices=["schur-full", "schur-lower", "schur-upper", "schur-diag", "jacobi"])
    ap.add_argument("--constraints", default="moments", choices=["moments", "div3d"],
                    help="the (1,0) block of the saddle system: the reference's few long rows (component means and moments: "
                         "4 rows in 2-D, 6 in 3-D; the fused Schur path), or -- with --dim 3 -- the discrete divergence block in "
                         "the manner of PETSc's ksp/ex42 (one row per hexahedron: a GENERAL sparse block, which runs PCApply and "
                         "MatMult step by step)")
    ap.add_argument("--restart", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="bound on the CPU baseline sample")
    ap.add_argument("--spmv-reps", type=int, default=200)
    ap.add_argument("--inner-sweeps", type=int, default=0, help="FP32 damped-Jacobi Richardson sweeps standing for "
                                                                "diag(A)^-1 in the PC (BASELINE config 5's mixed FP32 inner solve)")
    ap.add_argument("--inner-omega", type=float, default=0.8)
    ap.add_argument("--iter-form", type=int, default=0, choices=[0, 1, 2, 3, 4, 5, 6],
                    help="opts.iteration_form: 0 auto, 1 four launches per iteration, 2 two launches, 3 three launches, 4 BA, "
                         "5 un-normalised three-launch, 6 resident restart-cycle kernel")
    ap.add_argument("--single-reduce", type=int, default=0, help="1: single-reduction Gram-Schmidt (one all-reduce per iteration; see include/spk.h)")
    ap.add_argument("--watchdog", type=float, default=900.0, help="seconds after which a run that is still going says WHERE it "
                                                                  "is stuck (rank, phase) on stderr and exits 4; 0: off")
    args = ap.parse_args()

    # A multi-GPU run that hangs (a rendezvous, a collective of the fallback backend) would otherwise end in the
    # launcher's kill with nothing to read: the watchdog names the rank and the phase it never left.
    phase = ["start"]

    def watchdog():
        time.sleep(args.watchdog)
        print(f"[bench] watchdog: rank {os.environ.get('RANK', '0')} of {os.environ.get('WORLD_SIZE', '1')} still in phase "
              f"'{phase[0]}' after {args.watchdog:.0f} s -- giving up", file=sys.stderr, flush=True)
        os._exit(4)

    if args.watchdog > 0:
        import threading
        threading.Thread(target=watchdog, daemon=True).start()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    dist = None
    use_dist = world > 1 or os.environ.get("SPK_BENCH_FORCE_DIST") == "1"   # the latter: 1-GPU rehearsal
    if use_dist:
        # torch FIRST: its wheel carries a private libamdhip64 (DT_NEEDED "libamdhip64.so"); loaded
        # before libspk.so, the loader resolves libspk's "libamdhip64.so.7" to that same copy by
        # soname.  The other order maps two HIP runtimes into one process and the second one finds
        # no device (measured: spk_create -> "no ROCm-capable device").
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        phase[0] = "torch.distributed rendezvous"
        import torch
        import torch.distributed as dist
        if os.environ.get("SPK_BENCH_COMM") == "gloo":
            # rehearsal on a 1-GPU box: N processes share GPU 0 (RCCL/NCCL refuse that), collectives
            # go through the host over gloo.  Exercises the multi-process flow, not its speed.
            dist.init_process_group("gloo")
            local_rank = 0
            # (the resident cycle kernel wants every workgroup of its launch on the chip at once: N launches share one here)
            os.environ.setdefault("SPK_RES_WGS", str(max(1, 240 // world)))
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    phase[0] = "host assembly"
    import saddle_point_petsc_amd as S   # fails loudly if libspk.so is not built

    M = args.grid
    My = args.grid_y or M
    Mz = args.grid_z or M
    t_setup = time.time()
    t_asm = time.time()
    asm_threads = max(1, min(16, cpu_threads()))                   # N ranks share the host: stay far below
    saddle = args.pc != "jacobi"                                   # the box's thread limits
    B = g = None
    if args.dim == 3:
        n = 3 * M * My * Mz
        nnz_global = 9 * (3 * M - 2) * (3 * My - 2) * (3 * Mz - 2)
        rb, re_ = S.partition_slab3d(M, My, Mz, rank, world)
        A, f = S.AssembleOperator_Laplace3D(M, My, Mz, rb, re_, nthreads=asm_threads)
        if saddle and args.constraints == "div3d":
            # six long mean / moment rows + one divergence row per hexahedron (tests/test_gpu_general_b.py): with Dirichlet
            # data on every face the divergence rows carry the constant-pressure mode, so g = 0 on them (consistent system)
            Bm, g6 = S.AssembleOperator_Constraints3D(M, My, Mz, rb, re_)
            Bd = S.AssembleOperator_Divergence3D(M, My, Mz, rb, re_)
            B, g = S.CSR.vstack([Bm, Bd]), np.concatenate([g6, np.zeros(Bd.nrows)])
        elif saddle:
            B, g = S.AssembleOperator_Constraints3D(M, My, Mz, rb, re_)
    else:
        n, nnz_global = S.grid_sizes(M, My)
        rb, re_ = S.partition_slab(M, My, rank, world)
        A, f = S.AssembleOperator_Laplace(M, My, rb, re_, nthreads=asm_threads)
        if saddle:
            B, g = S.AssembleOperator_Constraints(M, My, rb, re_)
    t_asm = time.time() - t_asm
    t_ctx = time.time()
    ctx = S.Context(local_rank)     # HIP runtime start-up + stream + scratch: per process, not per KSPSetOperators
    t_ctx = time.time() - t_ctx
    t_up = time.time()
    phase[0] = "communicator set-up (RCCL unique id, init)"
    if use_dist and os.environ.get("SPK_BENCH_COMM") == "gloo":
        ctx.comm_init_torch(dist, rank, world)
    elif use_dist:
        ids = [S.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        ctx.comm_init_rccl(rank, world, ids[0])
    if use_dist and world > 1 and os.environ.get("SPK_BENCH_PEER", "1") != "0":
        # Krylov all-reduces and halo rows written straight into the peers' HBM over xGMI by the
        # solver's kernels; falls back (collectively) to the communicator above when a rank cannot
        # map a peer's window
        phase[0] = "peer-store windows (HIP IPC mapping, self-test)"
        if not ctx.comm_enable_peer() and rank == 0:
            print(f"[bench] peer-store collectives off ({ctx.last_error()}); using {ctx.comm_backend()}", file=sys.stderr)
    phase[0] = "KSPSetOperators (upload, split, halo plan)"
    ctx.set_block(S.BLOCK_A00, A)
    if saddle:
        ctx.set_block(S.BLOCK_A10, B)
    phase[0] = "KSPSetUp (preconditioner)"
    pc = S.PC_JACOBI if not saddle else S.PC_SCHUR
    fact = {"schur-full": S.SCHUR_FULL, "schur-lower": S.SCHUR_LOWER, "schur-upper": S.SCHUR_UPPER,
            "schur-diag": S.SCHUR_DIAG, "jacobi": S.SCHUR_FULL}[args.pc]
    t_up = time.time() - t_up
    t_pc = time.time()
    ctx.pc_setup(pc, fact, inner_sweeps=args.inner_sweeps, inner_omega=args.inner_omega)
    t_pc = time.time() - t_pc
    rhs = np.concatenate([f, g]) if saddle else f
    b_dev = ctx.vec_create(rhs)
    x_dev = ctx.vec_create(n=len(rhs))
    t_setup = time.time() - t_setup

    def barrier():
        if dist is not None:
            import torch
            if dist.get_backend() == "nccl":
                torch.cuda.synchronize()
            dist.barrier()

    kw = dict(restart=args.restart, rtol=0.0, abstol=0.0, dtol=1e300, single_reduce=args.single_reduce,
              iteration_form=args.iter_form)

    def timed_solve(steps, **kws):
        """exactly `steps` iterations between two barriers; MAX over ranks of the wall time"""
        barrier()
        t0 = time.perf_counter()
        inf = ctx.fgmres_device(b_dev, x_dev, max_it=steps, **kws)   # synchronous at return
        barrier()
        el = time.perf_counter() - t0
        assert inf["its"] == steps, inf
        if dist is not None:
            import torch
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, inf

    # ---- warm-up: W untimed iterations
    phase[0] = "warm-up solve"
    fallback = None
    if args.warmup > 0:
        # N > 1 with AUTO: the 1/8 slabs take the resident restart-cycle kernel, whose collectives run INSIDE one launch
        # per cycle -- first contact with real xGMI windows happens here.  If it fails on ANY rank (a bounded wait gave
        # up: SPK_ERR_COMM / SPK_ERR_HIP), every rank falls back to the launch-by-launch form 5 and the line says so.
        failed, msg = 0, ""
        try:
            ctx.fgmres_device(b_dev, x_dev, max_it=args.warmup, **kw)
        except S.SpkError as ex:
            if world == 1 or args.iter_form != 0:
                raise
            failed, msg = 1, str(ex)
        if dist is not None and world > 1 and args.iter_form == 0:
            import torch
            t = torch.tensor([failed], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if int(t.item()):
                fallback = msg or "another rank's warm-up solve failed"
                print(f"[bench] rank {rank}: warm-up with the resident cycle kernel failed ({fallback}); falling back to iteration form 5",
                      file=sys.stderr, flush=True)
                kw["iteration_form"] = 5
                ctx.fgmres_device(b_dev, x_dev, max_it=args.warmup, **kw)
    # ---- timed: exactly K iterations
    phase[0] = "timed solve"
    elapsed, info = timed_solve(args.steps, **kw)
    phase[0] = "post-solve checks (true residual, full cycles, kernel timings)"

    # ---- integrity of the timed solve (every N): the residual norm the device carries through its
    # Givens recurrence must equal the TRUE residual ||b - K x|| recomputed from the iterate with one
    # more (collective) product -- a wrong halo or all-reduce cannot satisfy this
    xh = ctx.vec_get(x_dev, len(rhs))
    yh = ctx.mult(xh)
    nl_ = len(f)
    r2 = float(np.sum((rhs[:nl_] - yh[:nl_]) ** 2)) + (float(np.sum((rhs[nl_:] - yh[nl_:]) ** 2)) if rank == 0 else 0.0)
    if dist is not None:
        import torch
        t = torch.tensor([r2], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t)
        r2 = float(t.item())
    true_rnorm = float(np.sqrt(r2))
    # (below ~1e-11 of the initial residual the true residual sits on its round-off floor while the
    # recurrence keeps falling: both "converged", not comparable digit by digit)
    # (a restart beyond the default lengths: classical Gram-Schmidt without refinement lets the recurrence drift from the
    # true residual inside a long cycle -- 5e-5 at restart 100, in the oracle as here; a wrong collective is off by O(1))
    res_tol = 1e-6 if args.restart <= 62 else 1e-3
    residual_ok = bool(abs(true_rnorm - info["rnorm"]) <= res_tol * max(true_rnorm, 1e-300) or
                       max(true_rnorm, info["rnorm"]) <= 1e-11 * info["rnorm0"])

    # ---- the same solver over WHOLE restart cycles (>= 3), whatever --steps says: a rate that does not
    # depend on where in a cycle the timed iterations happen to fall (early iterations of a cycle
    # orthogonalise against few vectors and are cheaper)
    full_steps = 3 * args.restart
    elapsed_full, _ = timed_solve(full_steps, **kw)
    form_run, _ = ctx.iteration_form()   # the form the solver took for these options (spk_get_iteration_form)

    # ---- the opt-in single-reduction mode on the same K iterations (one all-reduce and three launches
    # per iteration instead of two and four; ||w'||^2 by Pythagoras, see include/spk.h).  Reported
    # beside `value`, never as `value`: PETSc's default Gram-Schmidt makes two reductions.
    single_mode = None
    if args.pc in ("schur-full", "schur-lower", "jacobi") and args.inner_sweeps == 0 and args.single_reduce == 0:
        kws = dict(kw, single_reduce=1)
        ctx.fgmres_device(b_dev, x_dev, max_it=min(args.warmup, 30) or 1, **kws)
        es, infos = timed_solve(args.steps, **kws)
        single_mode = {"value": args.steps / es, "ms_per_step": es / args.steps * 1e3,
                       "residual_after_steps": infos["rnorm"] / infos["rnorm0"] if infos["rnorm0"] else None,
                       "reductions_per_iteration": 1, "launches_per_iteration": 3}

    # ---- the same K iterations with b and x handed over as HOST arrays (what a PCSHELL/KSP glue over
    # host Vecs does): adds one H2D of b and one D2H of x per solve over PCIe.  Reported, never `value`.
    host_rate = None
    if world == 1:
        t0h = time.perf_counter()
        _, ih = ctx.fgmres(rhs, max_it=args.steps, **kw)
        host_rate = ih["its"] / (time.perf_counter() - t0h)

    # ---- dominant kernel named by the metric: A-block SpMV, HIP events on the solver's stream
    spmv_ms = ctx.time_spmv(warmup=20, reps=args.spmv_reps)
    sz = ctx.sizes()
    alg_bytes = spmv_bytes(sz["n_local"], sz["nnz_local"])
    achieved = alg_bytes / (spmv_ms * 1e-3) / 1e9
    spi = ctx.spmv_info()
    achieved_layout = spi["layout_bytes"] / (spmv_ms * 1e-3) / 1e9
    models = ctx.spmv_models()      # bytes of one product in the CSR / blocked / row-type layouts (0: layout absent)
    # the variant the fused Schur iteration launches: y += A x (y pre-loaded with B^T lambda): 8n more bytes
    acc_ms = ctx.time_kernel("spmv_acc", 0, 20, args.spmv_reps)
    ride_ms = ctx.time_kernel("spmv_ride", 0, 20, args.spmv_reps)   # y = A x with the rider: the Jacobi iteration's launch
    # The same launch timed WHERE IT RUNS: the kernel's own start / stop time stamps (hipExtLaunchKernelGGL events on the
    # solver's stream) of every product launch of the iterations of one more solve over whole restart cycles -- behind the
    # MAXPY pass, caches as that pass leaves them.  This is the figure the roofline claims; it is what
    # `rocprofv3 --kernel-trace --stats` of the same command averages for that kernel's launches inside the solves
    # (tools/rocpd_product.py, profiles/*_product_in_solve.txt).  The resident form (and solves off the head path) launch no
    # such product per iteration: there the batch figure stands in, and says so.
    # (Inside a solve a pair of stamps reads longer than the launch itself -- the start stamp is taken when the packet is
    # picked up, before the previous kernel has drained.  The launches the device gates off at a solve's end show by how
    # much: they are the same launch returning at once, whose cost alone is measured on a batch.)
    phase[0] = "product timing inside a solve"
    ctx.time_products(4 * args.restart)
    timed_solve(full_steps, **kw)
    pt = ctx.product_timing()
    ctx.time_products(0)
    stamp_offset_ms = 0.0
    if pt["gated"] >= 1:
        gated_batch_ms = ctx.time_kernel("spmv_gated", 0, 20, args.spmv_reps)
        stamp_offset_ms = max(0.0, pt["gated_mean_ms"] - gated_batch_ms)
    # per-rank diagnostics of the communicator, gathered to rank 0: an N-GPU line explains itself
    rank_info = [ctx.comm_info()]
    if dist is not None and world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, rank_info[0])
        rank_info = gathered
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "spmv_traffic.json")
    if os.path.exists(tpath) and world == 1 and M == 1024 and My == 1024 and args.dim == 2:
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch_" + spi["format"])
        except Exception:  # noqa: BLE001
            traffic = None

    if rank != 0:
        ctx.vec_destroy(b_dev); ctx.vec_destroy(x_dev); ctx.close()   # RCCL communicator down before torch's
        if dist is not None:
            dist.destroy_process_group()
        return

    its_per_s = args.steps / elapsed
    nnzB_local = B.nnz if saddle else 0
    planes = ctx.bd_planes() if saddle else 0
    # the byte model covers the head-kernel paths: fused Schur (FULL / LOWER) and Jacobi on K = A
    head_path = args.inner_sweeps == 0 and ((saddle and planes > 0) or (not saddle and sz["n_local"] % 2 == 0))
    mrows = B.nrows if saddle else 0
    in_solver_acc = saddle and planes > 0            # the loop launches y += A x (else the plain product)
    batch_ms = acc_ms if in_solver_acc else ride_ms      # a batch of back-to-back launches, HIP events around the batch
    in_solve = pt["launches"] >= min(args.restart, 8)
    loop_ms = max(pt["mean_ms"] - stamp_offset_ms, batch_ms) if in_solve else batch_ms
    loop_alg = alg_bytes + (8 * sz["n_local"] if in_solver_acc else 0)
    loop_layout = spi["layout_bytes"] + (8 * sz["n_local"] if in_solver_acc else 0)
    loop_gbps, loop_layout_gbps = loop_alg / (loop_ms * 1e-3) / 1e9, loop_layout / (loop_ms * 1e-3) / 1e9
    extra = 8 * sz["n_local"] if in_solver_acc else 0
    byte_models = {k_: (models[k_ + "_bytes"] + extra if models[k_ + "_bytes"] else None) for k_ in ("csr", "blocked", "dict")}
    rates = {k_: (v / (loop_ms * 1e-3) / 1e9 if v else None) for k_, v in byte_models.items()}
    kernel_names = {"csr": "spmv_stream_kernel", "bcsr2x2": "spmv_bcsr_kernel", "bcsr3x3": "spmv_bcsr3_kernel",
                    # (row types of the 9-point / 27-point stencils take the pipelined kernels -- 3x3 from 2^20 block rows --
                    # other shapes spmv_dict_kernel<2 | 3, ..>: profiles/*_kernel_stats_*.csv name the one that ran)
                    "dict2x2": "spmv_dict2_kernel", "dict3x3": "spmv_dict3_kernel | spmv_dict_kernel<3>"}
    out = {
        "metric": METRIC,
        "value": its_per_s,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": (f"{M}x{My} DMDA node grid, dof 2" if args.dim == 2 else
                                f"{M}x{My}x{Mz} node grid (build-defined 3-D generator), dof 3")
                               + f" (n={n}, nnz(A)={nnz_global}), "
                               + (f"saddle K=[A B^T;B 0] with {B.nrows} constraint rows, " if saddle else "K=A, ")
                               + f"FGMRES({args.restart}) CGS, pc={args.pc}",
                   "grid": M, "rows": n + (B.nrows if saddle else 0), "dim": args.dim, "pc": args.pc, "restart": args.restart,
                   "inner_fp32_sweeps": args.inner_sweeps, "iteration_form": args.iter_form, "iteration_form_run": form_run,
                   "constraints": args.constraints if saddle else None, "resident_fallback": fallback,
                   "reductions_per_iteration": 1 if (saddle and args.pc in ("schur-full", "schur-lower") and
                                                     args.single_reduce == 1) else 2,
                   "parallelism": f"row-slab x{world}" if world > 1 else "single GPU",
                   "collectives": ctx.comm_backend()},
        "spmv_gbps": achieved,
        "spmv_ms": spmv_ms,
        "residual_after_steps": info["rnorm"] / info["rnorm0"] if info["rnorm0"] else None,
        "residual_check": {"recurrence": info["rnorm"], "true": true_rnorm, "consistent": residual_ok},
        # the representative rate: whole restart cycles (the first iterations of a cycle orthogonalise against few vectors and
        # are cheaper, so a --steps that is not a multiple of the restart length flatters `value`)
        "value_representative": full_steps / elapsed_full,
        "value_full_cycles": full_steps / elapsed_full,
        "full_cycles": {"steps": full_steps, "ms_per_step": elapsed_full / full_steps * 1e3,
                        "note": f"{full_steps // args.restart} whole restart cycles, timed like `value`"},
        "value_with_host_vectors": host_rate,
        "single_reduction_mode": single_mode,
        "setup_seconds": t_setup,
        "setup_breakdown": {"host_assembly": t_asm, "context_create": t_ctx, "set_operators_upload": t_up, "pc_setup": t_pc},
        # achieved = ALGORITHMIC (CSR) bytes / time.  When the kernel streams the
        # 2x2-blocked layout its true bytes are fewer: both rates are reported and `frac` is the
        # LOWER of the two fractions, as SURVEY 8(d) prescribes for compressed layouts.
        # The kernel described is the one the timed loop LAUNCHES for the A block: y += A x on the fused
        # Schur path (y pre-loaded with B^T lambda: one more vector read), the plain product otherwise.
        # `achieved` = the bytes the launched kernel really streams / time -- the LOWEST of the rates below, the one claimed
        # (`frac` = achieved / peak); the CSR-algorithmic figure of SURVEY 8(d) and the blocked layout's are beside it.
        "roofline": {"bound": "hbm",
                     "kernel": kernel_names[spi["format"]]
                               + (" <ACC=true, RIDE=true, BT=false, ..>: y += A x, as launched by the fused Schur iteration "
                                  "(Givens rider in workgroup 0)" if in_solver_acc
                                  else " <ACC=false, RIDE=true, BT=false, ..>: y = A x, as launched by the iteration"),
                     "format": spi["format"], "ms": loop_ms,
                     "ms_source": (f"mean of {pt['launches']} launches inside a solve of {full_steps} iterations: the kernel's own "
                                   f"start/stop stamps as HIP events on the solver's stream ({pt['mean_ms'] * 1e3:.2f} us; median "
                                   f"{pt['median_ms'] * 1e3:.2f}, {pt['min_ms'] * 1e3:.2f} .. {pt['max_ms'] * 1e3:.2f}) less the offset of such "
                                   f"a pair, read off the {pt['gated']} launches the device had gated off ({stamp_offset_ms * 1e3:.2f} us)" if in_solve else
                                   f"batch of {args.spmv_reps} back-to-back launches, HIP events around the batch (this solve launches "
                                   "no product per iteration)"),
                     "ms_back_to_back": batch_ms,
                     "achieved": min(loop_gbps, loop_layout_gbps), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": min(loop_gbps, loop_layout_gbps) / HBM_PEAK_GBS,
                     "byte_models": {"csr_algorithmic": byte_models["csr"], "blocked_layout": byte_models["blocked"],
                                     "row_types_codes_layout": byte_models["dict"], "streamed_by_the_kernel": loop_layout},
                     "rates_gbps": {"csr_algorithmic": rates["csr"], "blocked_layout": rates["blocked"],
                                    "row_types_codes_layout": rates["dict"]},
                     "layout": ({"row_types": models["patterns"], "block_classes": models["blocks"]} if models["dict_bytes"] else None),
                     "frac_algorithmic_csr_bytes": loop_gbps / HBM_PEAK_GBS,
                     "achieved_csr_model": loop_gbps,
                     "achieved_layout_bytes": loop_layout_gbps, "layout_bytes_per_launch": loop_layout,
                     "frac_of_measured_copy": min(loop_gbps, loop_layout_gbps) / HBM_COPY_GBS,
                     "bytes_per_launch": loop_layout, "bytes_per_launch_csr_model": loop_alg, "traffic": traffic,
                     "standalone_variant": {"kernel": "y = A x (spk_mult, restarts)", "ms": spmv_ms,
                                            "csr_gbps": achieved, "layout_gbps": achieved_layout,
                                            "frac": min(achieved, achieved_layout) / HBM_PEAK_GBS}},
        "ranks": rank_info,
    }
    if head_path:
        # bytes of the solves as executed (j runs over the iterations actually timed; cycle ends counted
        # as they happened), CSR model and the layout the kernel really streams; the LOWER fraction is claimed
        mat_layout = spi["layout_bytes"] - 4 * (sz["n_local"] + 1) - 16 * sz["n_local"]
        itmodels = {}
        for name, steps_, secs in (("timed_steps", args.steps, elapsed), ("full_cycles", full_steps, elapsed_full)):
            un = form_run in (5, 6)             # the form the solver took (spk_get_iteration_form), not a guess from the options
            b_csr = solve_bytes(sz["n_local"], sz["nnz_local"], nnzB_local, args.restart, steps_, planes, mrows, None, un)
            b_lay = solve_bytes(sz["n_local"], sz["nnz_local"], nnzB_local, args.restart, steps_, planes, mrows, mat_layout, un)
            itmodels[name] = {"steps": steps_, "bytes_per_iteration_csr_model": b_csr / steps_,
                            "bytes_per_iteration_layout": b_lay / steps_,
                            "achieved_gbps_csr_model": b_csr / secs / 1e9, "achieved_gbps_layout": b_lay / secs / 1e9,
                            "frac_of_peak": min(b_csr, b_lay) / secs / 1e9 / HBM_PEAK_GBS,
                            "form": ("resident restart-cycle kernel (basis in registers): the figures are what the three-launch "
                                     "form would stream -- this form moves ~3 vectors + the matrix codes per iteration and is bound by "
                                     "its two grid-wide exchanges per iteration, not by bandwidth") if form_run == 6
                                    else "unnormalised three-launch" if un else "head + SpMV + MDot + MAXPY"}
            if form_run == 6:
                vec = 8 * sz["n_local"]
                moved = spi["layout_bytes"] - 16 * sz["n_local"] + 3 * vec    # codes + z~ stored, gathered; Z read at the cycle end
                itmodels[name]["bytes_per_iteration_resident_form"] = moved
                itmodels[name]["achieved_gbps_resident_form"] = moved * steps_ / secs / 1e9
                itmodels[name]["frac_of_peak"] = moved * steps_ / secs / 1e9 / HBM_PEAK_GBS
                itmodels[name]["bound"] = "latency: two grid-wide exchanges per iteration (inner products all-to-all, z~ hand-off)"
        out["iteration_model"] = itmodels
        out["value_representative_model"] = itmodels["full_cycles"]
    if not residual_ok:
        # a wrong halo or all-reduce shows here: the number would describe a broken solver
        out["value"] = None
        out["error"] = "residual_check failed: the iterate's true residual does not match the recurrence"

    # ---- CPU baseline: the oracle (a port of PETSc's algorithm; PETSc itself is not
    # installable here) on a bounded sample of the SAME workload, all host cores.
    phase[0] = "cpu baseline (oracle)"
    if world == 1 and not args.no_cpu_baseline:
        import oracle as O
        cores = cpu_threads()
        Ao = O.CSR(A.rowptr, A.colidx, A.val, A.ncols)
        Bo = O.CSR(B.rowptr, B.colidx, B.val, B.ncols) if saddle else None
        okw = dict(B=Bo, pc_type=O.PC_SCHUR if saddle else O.PC_JACOBI, schur_fact=fact, restart=args.restart, rtol=0.0, abstol=0.0,
                   dtol=1e300, threads=cores)
        # warm-up (2 iterations: pages touched, threads up), then ONE WHOLE restart cycle -- the same mix of short and long
        # orthogonalisations as `value_representative` -- unless the time bound cuts it short (said in `sample`)
        t0 = time.perf_counter()
        O.fgmres(Ao, rhs, max_it=2, **okw)
        per_it = (time.perf_counter() - t0) / 2
        k = int(max(2, min(args.restart, args.cpu_seconds / max(per_it, 1e-9))))
        t0 = time.perf_counter()
        _, io = O.fgmres(Ao, rhs, max_it=k, **okw)
        tc = time.perf_counter() - t0
        O.time_spmv(Ao, 5, cores)                          # warm-up
        t_spmv = O.time_spmv(Ao, 60, cores) / 60
        whole = io["its"] == args.restart
        # the GPU over the SAME iterations (the first k of a cycle), for a like-for-like ratio
        el_same, _ = timed_solve(io["its"], **kw)
        out["cpu_baseline"] = {"value": io["its"] / tc, "unit": "iterations/s", "cores": cores, "cores_available": host_cores(),
                               "kind": "port",
                               "sample": (f"one whole restart cycle ({io['its']} FGMRES iterations)" if whole else
                                          f"the first {io['its']} iterations of a restart cycle (time bound {args.cpu_seconds:.0f} s)")
                                         + f" of the same {M}x{My} system with the oracle after a 2-iteration warm-up, OpenMP over "
                                           f"{cores} threads",
                               "spmv_gbps": spmv_bytes(A.nrows, A.nnz) / t_spmv / 1e9,
                               "label": "PETSc-equivalent CPU restatement (PETSc not installable offline)"}
        out["speedup_vs_cpu"] = (io["its"] / el_same) / out["cpu_baseline"]["value"]
        out["speedup_vs_cpu_note"] = f"GPU and CPU both over iterations 0..{io['its'] - 1} of a restart cycle"
    print(json.dumps(out))
    ctx.vec_destroy(b_dev); ctx.vec_destroy(x_dev); ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    if not residual_ok:
        sys.exit(3)


if __name__ == "__main__":
    main()
