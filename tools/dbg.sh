SPK_DICT_VERBOSE=1 timeout -k 10 120 python - <<'PY'
import numpy as np, saddle_point_petsc_amd as S
for m in [(4,4),(33,33),(32,32),(64,64),(50,7)]:
    A,_=S.AssembleOperator_Laplace(*m)
    with S.Context(0) as c:
        c.set_block(S.BLOCK_A00,A)
        print(m, c.spmv_info(), c.spmv_models(), flush=True)
PY
