"""One rank of a multi-PROCESS run on ONE GPU at the BASELINE config shapes (launched by
torch.distributed.run over gloo): config 4 = the 1024 x 1024 saddle system in row slabs, config 5 =
a 3-D grid in z-slabs whose node plane is large enough for the bulk halo form.  Same mechanism as
tests/_peer_worker.py (host-callback transport over gloo for the set-up, peer-store windows through
HIP IPC for the collectives); parameters come as JSON in argv[2].  Test infrastructure."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, prm = sys.argv[1], json.loads(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch  # noqa: F401  (before libspk: see bench.py)
    import torch.distributed as dist
    dist.init_process_group("gloo")
    import saddle_point_petsc_amd as S

    dim, grid = prm["dim"], prm["grid"]
    saddle = prm.get("saddle", True)
    if dim == 2:
        mx, my = grid
        b, e = S.partition_slab(mx, my, rank, world)
        A, f = S.AssembleOperator_Laplace(mx, my, b, e)
        Bs, g = S.AssembleOperator_Constraints(mx, my, b, e) if saddle else (None, np.zeros(0))
    else:
        mx, my, mz = grid
        b, e = S.partition_slab3d(mx, my, mz, rank, world)
        A, f = S.AssembleOperator_Laplace3D(mx, my, mz, b, e)
        Bs, g = S.AssembleOperator_Constraints3D(mx, my, mz, b, e) if saddle else (None, np.zeros(0))
    rhs = np.concatenate([f, g])
    xin = np.concatenate([np.sin(0.37 * np.arange(b, e)), 0.5 + np.arange(len(g))])

    c = S.Context(0)
    c.comm_init_torch(dist, rank, world)
    if prm.get("peer", True):
        assert c.comm_enable_peer(), c.last_error()
    c.set_block(S.BLOCK_A00, A)
    if saddle:
        c.set_block(S.BLOCK_A10, Bs)
    c.pc_setup(S.PC_SCHUR if saddle else S.PC_JACOBI, prm.get("fact", 3), inner_sweeps=prm.get("inner", 0),
               inner_omega=0.8)
    res = {}
    res["y"] = c.mult(xin)
    for name, kw in prm["solves"].items():
        x, info = c.fgmres(rhs, **kw)
        res[name + "/x"] = x
        res[name + "/hist"] = info["history"]
        res[name + "/meta"] = np.array([info["its"], info["reason"]], np.int64)
        res[name + "/rnorm"] = np.array([info["rnorm"], info["rnorm0"]])
        res[name + "/kx"] = c.mult(x)           # for the true residual of the iterate (collective)
    res["range"] = np.array([b, e], np.int64)
    info = c.comm_info()
    c.close()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    json.dump(info, open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
