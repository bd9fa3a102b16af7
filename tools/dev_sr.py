import sys, numpy as np
sys.path.insert(0, '/root/repo')
import saddle_point_petsc_amd as S
import oracle as O
M=int(sys.argv[1]) if len(sys.argv)>1 else 1024
A,f=S.AssembleOperator_Laplace(M); B,g=S.AssembleOperator_Constraints(M); rhs=np.concatenate([f,g])
with S.Context(0) as c:
    c.set_block(S.BLOCK_A00,A); c.set_block(S.BLOCK_A10,B); c.pc_setup(S.PC_SCHUR,S.SCHUR_FULL)
    x2,i2=c.fgmres(rhs,rtol=0.0,abstol=0.0,max_it=90,single_reduce=2)
    x1,i1=c.fgmres(rhs,rtol=0.0,abstol=0.0,max_it=90,single_reduce=1)
    xu,iu=c.fgmres(rhs,rtol=0.0,abstol=0.0,max_it=90,fused=0)
xo,io=O.fgmres(O.CSR(A.rowptr,A.colidx,A.val,A.ncols),rhs,B=O.CSR(B.rowptr,B.colidx,B.val,B.ncols),pc_type=O.PC_SCHUR,schur_fact=3,rtol=0.0,abstol=0.0,max_it=90,threads=8)
def dev(a,b): return np.abs(a/b-1)
for name,h in (("single vs two",i1['history']),("unfused vs two",iu['history']),("oracle(8thr) vs two",io['history'])):
    d=dev(h,i2['history'])
    print(name, "max dev its<=30: %.2e  its<=60: %.2e its<=90: %.2e"%(d[:31].max(), d[:61].max(), d.max()))
print("x: single vs two %.2e, unfused vs two %.2e, oracle vs two %.2e"%(np.linalg.norm(x1-x2)/np.linalg.norm(x2), np.linalg.norm(xu-x2)/np.linalg.norm(x2), np.linalg.norm(xo-x2)/np.linalg.norm(x2)))
