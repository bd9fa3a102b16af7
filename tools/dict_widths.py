"""Developer aid: class entry widths of the row-type layout, computed on the host from the assembled A (no GPU needed).
Mirrors build_dict: classes = 2x2 blocks equal to 2^-20 absolute, base = first member, granule = lowest set bit of the
deviations, width = two's-complement bits of the largest |k|."""
import sys
import numpy as np
import saddle_point_petsc_amd as S

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
My = int(sys.argv[2]) if len(sys.argv) > 2 else M
rb, re_ = S.partition_slab(M, My, 0, 1)
A, f = S.AssembleOperator_Laplace(M, My, rb, re_, nthreads=8)
rp, ci, v = np.asarray(A.rowptr), np.asarray(A.colidx), np.asarray(A.val)
n = A.nrows
nb = n // 2
# 2x2 blocks: rows 2i, 2i+1 have the same block columns (dof-2 grid)
r0 = np.arange(0, n, 2)
cnt = (rp[r0 + 1] - rp[r0]) // 2
assert np.all(rp[r0 + 2] - rp[r0 + 1] == 2 * cnt)
blocks = []
idx0 = np.concatenate([np.arange(rp[r], rp[r + 1]) for r in r0[:0]]) if False else None
# vectorised: entries of even rows in order, pairs (2 per block)
ev = np.concatenate([v[rp[r]:rp[r + 1]] for r in range(0, min(n, 2 * 4096 * 8), 2)]) if False else None
top = v[np.concatenate([np.arange(a, b) for a, b in zip(rp[r0], rp[r0 + 1])])] if nb < 300000 else None
if top is None:
    # large: use the regular structure -- rows are contiguous, so even/odd rows alternate in val
    starts = rp[r0]
    lens = rp[r0 + 1] - rp[r0]
    mask = np.zeros(len(v), bool)
    # even rows occupy [rp[2i], rp[2i+1])
    d = np.zeros(len(v) + 1, np.int32)
    np.add.at(d, rp[r0], 1)
    np.add.at(d, rp[r0 + 1], -1)
    mask = np.cumsum(d[:-1]) > 0
    top = v[mask]
    bot = v[~mask]
else:
    bot = v[np.concatenate([np.arange(a, b) for a, b in zip(rp[r0 + 1], rp[r0 + 2])])]
B = np.stack([top[0::2], top[1::2], bot[0::2], bot[1::2]], axis=1)   # nblocks x 4
key = np.rint(B * 2.0**20).astype(np.int64)
uk, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
print("blocks", len(B), "classes", len(uk))
tot_max = 0
gw = np.zeros(4, int)
for c in range(len(uk)):
    mem = B[inv.ravel() == c]
    base = B[first[c]]
    dev = mem - base
    ws = []
    for e in range(4):
        dd = dev[:, e]
        nz = dd[dd != 0]
        if len(nz) == 0:
            ws.append(1)
            continue
        m, ex = np.frexp(nz)
        # granule: lowest set bit over all deviations
        mi = (np.abs(m) * 2.0**53).astype(np.int64)
        low = (mi & -mi)
        g = np.min(ex - 53 + np.log2(low).astype(int))
        kabs = np.max(np.abs(nz)) / 2.0**g
        w = 2
        while (1 << (w - 1)) - 1 < kabs:
            w += 1
        ws.append(w)
    tot_max = max(tot_max, sum(ws))
    gw = np.maximum(gw, ws)
    print("class", c, "members", len(mem), "base", base, "widths", ws, "sum", sum(ws), "halves", ws[0] + ws[1], ws[2] + ws[3])
print("max bits per block", tot_max, "global per-entry widths", gw, "sum", gw.sum())
