"""Developer aid: class entry widths of the row-type layout for a dof-3 grid (3x3 blocks), computed on the host.
Mirrors build_dict (see tools/dict_widths.py)."""
import sys
import numpy as np
import saddle_point_petsc_amd as S

M, My, Mz = (int(a) for a in sys.argv[1:4])
rb, re_ = S.partition_slab3d(M, My, Mz, 0, 1)
A, f = S.AssembleOperator_Laplace3D(M, My, Mz, rb, re_, nthreads=8)
rp, ci, v = np.asarray(A.rowptr), np.asarray(A.colidx), np.asarray(A.val)
n = A.nrows
rows = np.repeat(np.arange(n), np.diff(rp))
br, bc = rows // 3, ci // 3
e = (rows % 3) * 3 + (ci % 3)
key = br.astype(np.int64) * (n // 3) + bc
uk, inv = np.unique(key, return_inverse=True)
B = np.zeros((len(uk), 9))
B[inv, e] = v
kq = np.rint(B * 2.0**20).astype(np.int64)
ukq, first, cinv = np.unique(kq, axis=0, return_index=True, return_inverse=True)
cinv = cinv.ravel()
print("blocks", len(B), "classes", len(ukq))
gw = np.zeros(9, int)
tot_max = 0
for c in range(len(ukq)):
    mem = B[cinv == c]
    base = B[first[c]]
    dev = mem - base
    ws = []
    for k in range(9):
        nz = dev[:, k][dev[:, k] != 0]
        if len(nz) == 0:
            ws.append(1)
            continue
        m, ex = np.frexp(nz)
        mi = (np.abs(m) * 2.0**53).astype(np.int64)
        low = mi & -mi
        g = np.min(ex - 53 + np.log2(low).astype(int))
        kabs = np.max(np.abs(nz)) / 2.0**g
        w = 2
        while (1 << (w - 1)) - 1 < kabs:
            w += 1
        ws.append(w)
    tot_max = max(tot_max, sum(ws))
    gw = np.maximum(gw, ws)
    print("class", c, "members", len(mem), "widths", ws, "sum", sum(ws))
print("max bits per block", tot_max, "global per-entry widths", list(gw), "sum", gw.sum(), "split 5/4:", gw[:5].sum(), gw[5:].sum(), "4/5:", gw[:4].sum(), gw[4:].sum())
