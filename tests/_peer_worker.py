"""One rank of a multi-PROCESS run on ONE GPU (launched by torch.distributed.run, gloo):
each process owns a row slab, all share device 0.  RCCL refuses two ranks on one device, so
the communicator is the host-callback transport over gloo; on top of it the peer-store
backend maps the other processes' windows through HIP IPC -- the production mechanism, with
device 0's HBM standing in for the xGMI peers.  Every case is run with the collectives
staged through the host (reference) and written by the solver's kernels (peer-store);
results go to <out>/rank<r>.npz for the test to compare.  Test infrastructure."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, mode = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch  # noqa: F401  (before libspk: see bench.py)
    import torch.distributed as dist
    dist.init_process_group("gloo")
    import saddle_point_petsc_amd as S

    res = {}
    if mode == "timeout":
        # rank 0 multiplies (halo exchange + all-reduce) while the others stay away: bounded wait
        mx = my = 12
        b, e = S.partition_slab(mx, my, rank, world)
        A, _ = S.AssembleOperator_Laplace(mx, my, b, e)
        Bs, _ = S.AssembleOperator_Constraints(mx, my, b, e)
        c = S.Context(0)
        c.comm_init_torch(dist, rank, world)
        assert c.comm_enable_peer(), c.last_error()
        c.set_block(S.BLOCK_A00, A)
        c.set_block(S.BLOCK_A10, Bs)
        msg, code = "", 0
        if rank == 0:
            try:
                c.mult(np.ones(A.nrows + 4))
            except S.SpkError as ex:
                msg, code = str(ex), ex.code
        dist.barrier()
        c.close()
        json.dump({"msg": msg, "code": code}, open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
        dist.destroy_process_group()
        return

    if mode == "reset":
        # KSPSetOperators twice on ONE context: the halo staging is re-allocated and re-shared
        # (collectively), the all-reduce windows stay
        c = S.Context(0)
        c.comm_init_torch(dist, rank, world)
        assert c.comm_enable_peer(), c.last_error()
        for it, (mx, my) in enumerate([(20, 18), (28, 33), (20, 18)]):
            b, e = S.partition_slab(mx, my, rank, world)
            A, f = S.AssembleOperator_Laplace(mx, my, b, e)
            Bs, g = S.AssembleOperator_Constraints(mx, my, b, e)
            c.set_block(S.BLOCK_A00, A)
            c.set_block(S.BLOCK_A10, Bs)
            c.pc_setup(S.PC_SCHUR, S.SCHUR_FULL)
            x, info = c.fgmres(np.concatenate([f, g]), rtol=1e-9)
            res[f"{it}/x"], res[f"{it}/meta"] = x, np.array([b, e, info["its"], info["reason"]], np.int64)
        c.close()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
        dist.barrier()
        dist.destroy_process_group()
        return

    if mode == "resident":
        # the resident restart-cycle kernel (iteration form 6) ACROSS RANKS: the ranks' inner products through the all-reduce
        # windows inside the launch, the halo rows of z~ as granules between the edge workgroups.  SPK_RES_WGS (set by the
        # test) caps every process's grid so that all processes' launches are resident on the one device together.
        for name, grid, pc, fact, kw in (("schur_full", (96, 64), S.PC_SCHUR, S.SCHUR_FULL, dict(rtol=1e-9, max_it=900)),
                                         ("schur_lower", (96, 64), S.PC_SCHUR, S.SCHUR_LOWER, dict(rtol=0.0, abstol=0.0, max_it=75)),
                                         ("jacobi", (128, 100), S.PC_JACOBI, 0, dict(rtol=0.0, abstol=0.0, max_it=95)),
                                         ("jacobi_r7", (40, 36), S.PC_JACOBI, 0, dict(rtol=1e-7, restart=7, max_it=4000))):
            saddle = pc == S.PC_SCHUR
            mx, my = grid
            b, e = S.partition_slab(mx, my, rank, world)
            A, f = S.AssembleOperator_Laplace(mx, my, b, e)
            Bs, g = S.AssembleOperator_Constraints(mx, my, b, e) if saddle else (None, np.zeros(0))
            rhs = np.concatenate([f, g])
            c = S.Context(0)
            c.comm_init_torch(dist, rank, world)
            assert c.comm_enable_peer(), c.last_error()
            c.set_block(S.BLOCK_A00, A)
            if saddle:
                c.set_block(S.BLOCK_A10, Bs)
            c.pc_setup(pc, fact)
            for form in (6, 5):
                x, info = c.fgmres(rhs, iteration_form=form, **kw)
                k = f"{name}/{form}/"
                res[k + "x"], res[k + "hist"] = x, info["history"]
                res[k + "kx"] = c.mult(x)
                res[k + "meta"] = np.array([b, e, info["its"], info["reason"], c.iteration_form()[0]], np.int64)
                res[k + "rnorm"] = np.array([info["rnorm"]])
            ci = c.comm_info()
            res[name + "/fused"] = np.array([ci["allreduce"]["fused"], ci["halo_exchanges"]["fused"], ci["allreduce"]["inner"]], np.int64)
            c.close()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
        dist.barrier()
        dist.destroy_process_group()
        return

    # (name, dim, grid, pc, fact, inner sweeps, fused); solver options by name suffix, see OPTS
    OPTS = {"single": dict(single_reduce=1), "mgs": dict(orthog=1), "refine": dict(cgs_refine=1),
            "r62": dict(restart=62), "guess": dict()}
    cases = [("schur_full", 2, (24, 26), S.PC_SCHUR, S.SCHUR_FULL, 0, 1),
             ("schur_full_single", 2, (24, 26), S.PC_SCHUR, S.SCHUR_FULL, 0, 1),
             ("schur_full_mgs", 2, (24, 26), S.PC_SCHUR, S.SCHUR_FULL, 0, 1),
             ("schur_full_refine", 2, (24, 26), S.PC_SCHUR, S.SCHUR_FULL, 0, 1),
             ("schur_full_r62", 2, (24, 26), S.PC_SCHUR, S.SCHUR_FULL, 0, 1),
             ("schur_full_guess", 2, (24, 26), S.PC_SCHUR, S.SCHUR_FULL, 0, 1),
             ("schur_lower_unfused", 2, (24, 26), S.PC_SCHUR, S.SCHUR_LOWER, 0, 0),
             ("jacobi", 2, (24, 26), S.PC_JACOBI, 0, 0, 1),
             ("jacobi_single", 2, (24, 26), S.PC_JACOBI, 0, 0, 1),
             ("schur_diag_fp32", 2, (24, 26), S.PC_SCHUR, S.SCHUR_DIAG, 3, 1),
             ("jacobi_3d_fp32", 3, (10, 9, 12), S.PC_JACOBI, 0, 3, 1)]
    for name, dim, grid, pc, fact, inner, fused in cases:
        saddle = pc == S.PC_SCHUR
        if dim == 2:
            mx, my = grid
            b, e = S.partition_slab(mx, my, rank, world)
            A, f = S.AssembleOperator_Laplace(mx, my, b, e)
            Bs, g = S.AssembleOperator_Constraints(mx, my, b, e) if saddle else (None, np.zeros(0))
        else:
            mx, my, mz = grid
            b, e = S.partition_slab3d(mx, my, mz, rank, world)
            A, f = S.AssembleOperator_Laplace3D(mx, my, mz, b, e)
            Bs, g = None, np.zeros(0)
        rhs = np.concatenate([f, g])
        xin = np.sin(0.37 * np.arange(b, e))
        xin = np.concatenate([xin, 0.5 + np.arange(len(g))])
        for peer in (0, 1):
            c = S.Context(0)
            c.comm_init_torch(dist, rank, world)
            if peer:
                assert c.comm_enable_peer(), c.last_error()
            backend = c.comm_backend()
            c.set_block(S.BLOCK_A00, A)
            if saddle:
                c.set_block(S.BLOCK_A10, Bs)
            c.pc_setup(pc, fact, inner_sweeps=inner, inner_omega=0.8)
            y = c.mult(xin)
            z = c.pc_apply(xin)
            okw = OPTS.get(name.rsplit("_", 1)[-1], {})
            x0 = 0.01 * xin if name.endswith("_guess") else None          # -ksp_initial_guess_nonzero
            # (form 5 in both runs: with the peer-store backend AUTO would take the resident form 6, which the host-staged
            # run this one is compared with bit for bit cannot)
            x, info = c.fgmres(rhs, x0=x0, rtol=1e-9, fused=fused, iteration_form=5, **okw)
            k = f"{name}/{peer}/"
            res[k + "y"], res[k + "z"], res[k + "x"] = y, z, x
            res[k + "hist"] = info["history"]
            res[k + "meta"] = np.array([b, e, info["its"], info["reason"], int(backend == "peer-store"),
                                        c.sizes()["n_ghost"]], np.int64)
            c.close()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
