import sys, numpy as np
sys.path.insert(0, '/root/repo')
import saddle_point_petsc_amd as S, oracle as O
A, f = S.AssembleOperator_Laplace(32, 27)
x = np.random.default_rng(4).uniform(-1, 1, A.nrows)
for k in (1, 2, 3):
    with S.Context(0) as c:
        c.set_block(S.BLOCK_A00, A)
        c.pc_setup(S.PC_JACOBI, 0, inner_sweeps=k, inner_omega=0.8)
        z = c.pc_apply(x)
    zo = O.pc_apply_inner(A, None, O.PC_JACOBI, 0, k, 0.8, x)
    d = np.abs(z - zo)
    print(k, "max abs diff", d.max(), "n differing", (d > 0).sum(), "first idx", np.argmax(d > 0), "rel", d.max() / np.abs(zo).max())
    i = int(np.argmax(d))
    print("   at", i, z[i], zo[i], np.float32(z[i]).view(np.uint32) - np.float32(zo[i]).view(np.uint32))
