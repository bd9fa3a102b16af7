import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import saddle_point_petsc_amd as S
M, My = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 128
A, f = S.AssembleOperator_Laplace(M, My); B, g = S.AssembleOperator_Constraints(M, My)
c = S.Context(0); c.set_block(S.BLOCK_A00, A); c.set_block(S.BLOCK_A10, B); c.pc_setup(S.PC_SCHUR, 3)
for name in ("spmv_bcsr", "spmv_ride", "spmv_acc"):
    print("before solve (no rider):", name, round(c.time_kernel(name, 0, 20, 300) * 1e3, 2), "us")
rhs = np.concatenate([f, g]); c.fgmres(rhs, max_it=40, rtol=0.0, abstol=0.0, dtol=1e300)
for name in ("spmv_bcsr", "spmv_ride", "spmv_acc"):
    print("after solve (rider on ride/acc):", name, round(c.time_kernel(name, 0, 20, 300) * 1e3, 2), "us")
