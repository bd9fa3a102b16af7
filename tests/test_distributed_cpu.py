"""N > 1 on the CPU: world-size-2 (and 3) gloo rehearsal of the row-partitioned
solve (tests/_dist_worker.py) against the serial oracle.  Exercises the product's
slab assembler, spk_partition_slab / spk_partition_split and the halo / all-reduce
/ replicated-lambda scheme the device loop uses.  CPU only."""
import multiprocessing as mp
import socket

import numpy as np
import pytest

from conftest import relerr


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,fact", [(2, 3), (2, 1), (3, 3)])
def test_row_partitioned_fgmres_over_gloo(spk, oracle, world, fact):
    from _dist_worker import run
    mx, my, rtol = 20, 23, 1e-9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=run, args=(r, world, port, mx, my, fact, rtol, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=180) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)

    A, f = oracle.assemble(mx, my)
    B, g = oracle.assemble_constraints(mx, my)
    rhs = np.concatenate([f, g])
    xo, io = oracle.fgmres(A, rhs, B=B, pc_type=oracle.PC_SCHUR, schur_fact=fact, rtol=rtol)
    n = A.nrows
    x = np.zeros(n + 4)
    res.sort(key=lambda t: t[0])
    for (rank, b0, e0, xr, its, reason, hist, ng) in res:
        x[b0:e0] = xr[:-4]
        x[n:] = xr[-4:]
        assert reason == io["reason"] == 2 and abs(its - io["its"]) <= 1
        assert np.array_equal(xr[-4:], res[0][3][-4:])                 # multipliers identical on every rank
        assert np.array_equal(hist, res[0][6])                         # every rank takes the same branch
        assert ng == 2 * mx * ((b0 > 0) + (e0 < n))                    # one ghost node line per neighbour
        k = min(len(hist), len(io["history"]), 21)
        assert np.allclose(hist[:k], io["history"][:k], rtol=1e-8)
    assert relerr(x, xo) < 1e-7
