import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
import saddle_point_petsc_amd as S
t=time.time(); A,f=S.AssembleOperator_Laplace(1024, nthreads=16); B,g=S.AssembleOperator_Constraints(1024); print("assemble", time.time()-t)
t=time.time(); c=S.Context(0); print("ctx", time.time()-t)
for rep in range(3):
    t=time.time(); c.set_block(S.BLOCK_A00, A); ta=time.time()-t
    t=time.time(); c.set_block(S.BLOCK_A10, B); tb=time.time()-t
    t=time.time(); c.pc_setup(S.PC_SCHUR, S.SCHUR_FULL); tp=time.time()-t
    print(f"rep {rep}: set A {ta*1e3:.1f} ms, set B {tb*1e3:.1f} ms, pc_setup {tp*1e3:.1f} ms")
x=np.sin(0.37*np.arange(A.nrows+4)); y=c.mult(x)
import oracle as O
yo = O.apply_K(A, B, x)
print("K x vs oracle: relative", np.linalg.norm(y - yo) / np.linalg.norm(yo))
x0 = x.copy(); x0[A.nrows:] = 0.0   # A-block alone: bitwise
print("A-block rows bitwise equal to the oracle:", np.array_equal(c.mult(x0)[:A.nrows], O.apply_K(A, B, x0)[:A.nrows]))
