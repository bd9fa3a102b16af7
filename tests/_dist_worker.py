"""World-size-N CPU rehearsal of the row-partitioned solve over torch.distributed
(gloo).  Each rank: product assembler slab -> libspk's host split (Ad, Ao, garray)
-> halo plan from the gathered ghost lists -> the SAME partitioned FGMRES the
device loop runs (unfused path: PCApply, MatMult with halo, CGS with one
all-reduce for the dots and one for the norm, lambda replicated and counted on
rank 0 only), local arithmetic done with the oracle's CSR kernels.  Test
infrastructure: no GPU, no product compute."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _split(S, A):
    i64, i32 = C.c_int64, C.c_int32
    nd, no, ng = i64(), i64(), i32()
    args = (A.row_begin, A.nrows, A.rowptr, A.colidx, A.val)
    assert S.lib.spk_partition_split(*args, *([None] * 7), C.byref(nd), C.byref(no), C.byref(ng)) == 0
    drp, orp = np.zeros(A.nrows + 1, np.int32), np.zeros(A.nrows + 1, np.int32)
    dci, oci = np.zeros(max(nd.value, 1), np.int32), np.zeros(max(no.value, 1), np.int32)
    dv, ov = np.zeros(max(nd.value, 1)), np.zeros(max(no.value, 1))
    ga = np.zeros(max(ng.value, 1), np.int32)
    p = lambda a: a.ctypes.data  # noqa: E731
    assert S.lib.spk_partition_split(*args, p(drp), p(dci), p(dv), p(orp), p(oci), p(ov), p(ga),
                                     C.byref(nd), C.byref(no), C.byref(ng)) == 0
    return (drp, dci[:nd.value], dv[:nd.value]), (orp, oci[:no.value], ov[:no.value]), ga[:ng.value]


def run(rank, world, port, mx, my, fact, rtol, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import saddle_point_petsc_amd as S
        import oracle as O
        n = 2 * mx * my
        b0, e0 = S.partition_slab(mx, my, rank, world)
        A, f = S.AssembleOperator_Laplace(mx, my, b0, e0)
        B, g = S.AssembleOperator_Constraints(mx, my, b0, e0)
        (drp, dci, dv), (orp, oci, ov), ga = _split(S, A)
        nl, m, ng = e0 - b0, 4, len(ga)
        Ad = O.CSR(drp, dci, dv, nl)
        Ao = O.CSR(orp, oci, ov, max(ng, 1))
        Bl = O.CSR(B.rowptr, B.colidx - b0, B.val, nl)

        # ---- halo plan: who owns my ghosts, what do I send (as libspk's set_block does)
        ranges = [None] * world
        dist.all_gather_object(ranges, (b0, e0))
        ghosts = [None] * world
        dist.all_gather_object(ghosts, ga.tolist())
        send = {p: np.array([c - b0 for c in ghosts[p] if b0 <= c < e0], np.int64) for p in range(world) if p != rank}
        recv = {p: np.array([i for i, c in enumerate(ga) if ranges[p][0] <= c < ranges[p][1]], np.int64)
                for p in range(world) if p != rank}

        def halo(x):
            xg = np.zeros(max(ng, 1))
            reqs, bufs = [], {}
            for p in sorted(send):
                if len(send[p]):
                    reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(x[send[p]])), p))
            for p in sorted(recv):
                if len(recv[p]):
                    bufs[p] = torch.zeros(len(recv[p]), dtype=torch.float64)
                    reqs.append(dist.irecv(bufs[p], p))
            for r in reqs:
                r.wait()
            for p, t in bufs.items():
                xg[recv[p]] = t.numpy()
            return xg

        def allreduce(v):
            t = torch.from_numpy(np.array(v, np.float64).reshape(-1).copy())
            dist.all_reduce(t)
            return t.numpy()

        dinv = O.jacobi_dinv(Ad)
        # S^ = diag(B D B^T): local column slab contribution, summed over ranks
        Bd = Bl.to_scipy()
        shat = allreduce((Bd.multiply(Bd).multiply(dinv[None, :])).sum(axis=1).A1)

        def mult(x):
            u, lam = x[:nl], x[nl:]
            y = O.spmv(Ad, u) + (O.spmv(Ao, halo(u)) if ng else 0.0) + Bd.T @ lam
            return np.concatenate([y, allreduce(Bd @ u)])

        def pc(x):
            x0, x1 = x[:nl], x[nl:]
            if fact == 0:
                return np.concatenate([x0 * dinv, x1 / shat])
            if fact == 2:
                y1 = -x1 / shat
                return np.concatenate([(x0 - Bd.T @ y1) * dinv, y1])
            y0 = x0 * dinv
            y1 = -(x1 - allreduce(Bd @ y0)) / shat
            if fact == 3:
                y0 = y0 - dinv * (Bd.T @ y1)
            return np.concatenate([y0, y1])

        def dots(V, w):          # lambda is replicated: count it on rank 0 only
            nd_ = nl + (m if rank == 0 else 0)
            return allreduce([v[:nd_] @ w[:nd_] for v in V])

        rhs = np.concatenate([f, g])
        bnorm = np.sqrt(dots([rhs], rhs)[0])
        x = np.zeros(nl + m)
        r = rhs.copy()
        its, reason, mk, hist = 0, 0, 30, []
        ttol = max(rtol * bnorm, 1e-50)
        while not reason:
            rn = np.sqrt(dots([r], r)[0])
            if its == 0:
                hist.append(rn)
            if rn <= ttol:
                reason = 2
                break
            V, Z = [r / rn], []
            H = np.zeros((mk + 2, mk + 1)); cc = np.zeros(mk + 1); ss = np.zeros(mk + 1); rs = np.zeros(mk + 2)
            rs[0] = rn
            loc = 0
            while not reason and loc < mk and its < 10000:
                Z.append(pc(V[loc]))
                w = mult(Z[loc])
                h = dots(V, w)
                for hv, v in zip(h, V):
                    w = w - hv * v
                tt = np.sqrt(dots([w], w)[0])
                V.append(w / tt)
                H[:loc + 1, loc] = h
                H[loc + 1, loc] = tt
                for j in range(1, loc + 1):
                    h0, h1 = H[j - 1, loc], H[j, loc]
                    H[j - 1, loc] = cc[j - 1] * h0 + ss[j - 1] * h1
                    H[j, loc] = cc[j - 1] * h1 - ss[j - 1] * h0
                h0, h1 = H[loc, loc], H[loc + 1, loc]
                d = np.hypot(h0, h1)
                cc[loc], ss[loc] = h0 / d, h1 / d
                rs[loc + 1] = -ss[loc] * rs[loc]
                rs[loc] = cc[loc] * rs[loc]
                H[loc, loc] = cc[loc] * h0 + ss[loc] * h1
                rn = abs(rs[loc + 1])
                loc += 1
                its += 1
                hist.append(rn)
                if rn <= ttol:
                    reason = 2
            y = np.zeros(loc)
            for k in range(loc - 1, -1, -1):
                y[k] = (rs[k] - H[k, k + 1:loc] @ y[k + 1:]) / H[k, k]
            for k in range(loc):
                x = x + y[k] * Z[k]
            if not reason:
                r = rhs - mult(x)
        q.put((rank, b0, e0, x, its, reason, np.array(hist), ng))
    finally:
        dist.destroy_process_group()
