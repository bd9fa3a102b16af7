#!/usr/bin/env python3
"""Per-kernel timing table on one GPU (HIP events on the solver's stream):
    python tools/kbench.py [--grid 1024]
Prints time, algorithmic bytes and GB/s for every kernel of an FGMRES iteration."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import saddle_point_petsc_amd as S

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=1024)
ap.add_argument("--reps", type=int, default=100)
ap.add_argument("--grid-y", type=int, default=0)
ap.add_argument("--kernels", default="spmv,spmv_bcsr,spmv_dict,mult,pc,wide_dot,bt_update,scale,mdot,maxpy")
ap.add_argument("--nvs", default="1,8,9,16,17,24,25,30")
a = ap.parse_args()
M = a.grid
My = a.grid_y or M
A, f = S.AssembleOperator_Laplace(M, My)
B, g = S.AssembleOperator_Constraints(M, My)
n, nnz, nnzB = A.nrows, A.nnz, B.nnz
c = S.Context(0)
c.set_block(S.BLOCK_A00, A); c.set_block(S.BLOCK_A10, B); c.pc_setup(S.PC_SCHUR, S.SCHUR_FULL)
vec = 8 * n
model = {"spmv": 12 * nnz + 4 * n + 2 * vec, "spmv_bcsr": 12 * nnz + 4 * n + 2 * vec, "spmv_dict": 12 * nnz + 4 * n + 2 * vec, "mult": 12 * nnz + 4 * n + 2 * vec + 2 * (12 * nnzB) + 4 * n + vec,
         "pc": (12 * nnzB + 2 * vec) + (12 * nnzB + 4 * n + 3 * vec), "wide_dot": 12 * nnzB + vec,
         "bt_update": 12 * nnzB + 4 * n + 3 * vec, "scale": 2 * vec}
out = {}
for k in a.kernels.split(","):
    if k in ("mdot", "maxpy", "maxpy_nonorm"):
        for nv in [int(v) for v in a.nvs.split(",")]:
            ms = c.time_kernel(k, nv, 10, a.reps)
            b = (nv + 1) * vec if k == "mdot" else (nv + 2) * vec
            out[f"{k}{nv}"] = (ms * 1e3, b / ms / 1e6)
    else:
        ms = c.time_kernel(k, 0, 10, a.reps)
        out[k] = (ms * 1e3, model[k] / ms / 1e6)
for k, (us, gbs) in out.items():
    print(f"{k:12s} {us:9.2f} us  {gbs:8.1f} GB/s")
print(json.dumps({k: v[0] for k, v in out.items()}))
