"""Regenerates tests/golden/appendix_b.json from an INDEPENDENT derivation -- no number in the file is
copied out of a document any more, and none comes from the oracle (oracle/sp_oracle.c) or the product.

What is derived, and from what
  Ke, Fe          exact rational arithmetic (sympy) of the reference's formulas: Q1 shape functions and their
                  gradients (/root/reference/src/Discretization.c:65-94), the 2 x 2 Gauss rule with the
                  abscissa LITERAL 0.57735026919 and unit weights (:52-55), the stress-form integrand
                  B^T D B detJ with D = diag(2, 2, 1) (:313-327), node order n0=(i,j), n1=(i,j+1),
                  n2=(i+1,j+1), n3=(i+1,j) (:377-395, the corner order of the commented block :40-43), load
                  f = (1, 2) (:397-402, :362-367).  The truncated abscissa is kept as the exact rational
                  57735026919 / 10^11, so Ke carries the same O(1e-12) defect (SURVEY defect A2) exactly.
  grids, m4       a plain numpy/scipy assembly written here (element loop, natural numbering (j M + i) 2 + c,
                  rows and columns of boundary degrees of freedom replaced by the identity, boundary load 0:
                  MatZeroRowsColumns(..., 1.0, NULL, NULL), :229-274) and scipy's sparse LU.
  constraints_m32 the BUILD-DEFINED constraint rows (component means and first moments with lumped weights,
                  Dirichlet columns dropped -- the reference's assemblers are empty stubs, :277-290), their
                  S^ = diag(B diag(A)^-1 B^T) by dense algebra, and the saddle system by scipy's sparse LU.

The checked-in file is the output of:   python tests/golden/derive_appendix_b.py
`--check` compares a fresh derivation with the checked-in file instead of writing (used by the CPU tests)."""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as sla
import sympy as sy

HERE = os.path.dirname(os.path.abspath(__file__))
GAUSS = sy.Rational(57735026919, 10 ** 11)     # the literal of Discretization.c:52-55, exactly


def element_matrices():
    """Ke (8 x 8) and Fe / (h^2/4) on an axis-aligned square of side h, exact rationals; h cancels in 2-D."""
    xi, eta, h = sy.symbols("xi eta h", positive=True)
    # reference nodes in the order n0..n3 above: (-1,-1), (-1,+1), (+1,+1), (+1,-1)
    sg = [(-1, -1), (-1, 1), (1, 1), (1, -1)]
    N = [sy.Rational(1, 4) * (1 + a * xi) * (1 + b * eta) for a, b in sg]
    # x = x0 + h (xi + 1) / 2: d/dx = (2/h) d/dxi; detJ = h^2 / 4
    dNx = [sy.diff(n, xi) * 2 / h for n in N]
    dNy = [sy.diff(n, eta) * 2 / h for n in N]
    detJ = h ** 2 / 4
    D = sy.diag(2, 2, 1)
    Ke = sy.zeros(8, 8)
    Fe = sy.zeros(8, 1)
    for gx in (-GAUSS, GAUSS):
        for gy in (-GAUSS, GAUSS):
            sub = {xi: gx, eta: gy}
            B = sy.zeros(3, 8)
            for a in range(4):
                B[0, 2 * a] = dNx[a].subs(sub)
                B[1, 2 * a + 1] = dNy[a].subs(sub)
                B[2, 2 * a] = dNy[a].subs(sub)
                B[2, 2 * a + 1] = dNx[a].subs(sub)
            Ke += (B.T * D * B) * detJ
            for a in range(4):
                Fe[2 * a] += N[a].subs(sub) * 1 * detJ
                Fe[2 * a + 1] += N[a].subs(sub) * 2 * detJ
    Ke = sy.simplify(Ke)
    assert not Ke.free_symbols, "Ke must not depend on h in 2-D"
    return (np.array(Ke.tolist(), dtype=object), [sy.nsimplify(v / (h ** 2 / 4)) for v in Fe])


def assemble(M, Ke):
    """A (CSR, structural zeros kept like DMCreateMatrix's preallocation) and f on an M x M node grid."""
    n = 2 * M * M
    h = 1.0 / (M - 1)
    rows, cols, vals = [], [], []
    f = np.zeros(n)
    off = [(0, 0), (0, 1), (1, 1), (1, 0)]          # (di, dj) of n0..n3
    for ej in range(M - 1):
        for ei in range(M - 1):
            eq = [((ej + dj) * M + (ei + di)) * 2 + c for (di, dj) in off for c in range(2)]
            for a in range(8):
                f[eq[a]] += (h * h / 4) * (1.0 if a % 2 == 0 else 2.0)
                for b in range(8):
                    rows.append(eq[a]); cols.append(eq[b]); vals.append(Ke[a, b])
    A = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    node = np.arange(n) // 2
    i, j = node % M, node // M
    bnd = (i == 0) | (i == M - 1) | (j == 0) | (j == M - 1)
    keep = sp.diags((~bnd).astype(float))
    A = keep @ A @ keep + sp.diags(bnd.astype(float))
    f[bnd] = 0.0
    return A.tocsr(), f, bnd


def constraints(M, bnd):
    """4 x n block: means of Ux, Uy and first moments (x - 1/2) Ux, (y - 1/2) Uy with lumped nodal weights
    h^2 {1 interior, 1/2 edge, 1/4 corner}; Dirichlet columns dropped (only interior nodes remain)."""
    h = 1.0 / (M - 1)
    n = 2 * M * M
    B = np.zeros((4, n))
    for j in range(M):
        for i in range(M):
            p = j * M + i
            w = h * h
            B[0, 2 * p] = w
            B[1, 2 * p + 1] = w
            B[2, 2 * p] = w * (i * h - 0.5)
            B[3, 2 * p + 1] = w * (j * h - 0.5)
    B[:, bnd] = 0.0
    return sp.csr_matrix(B)


def derive():
    KeR, FeR = element_matrices()
    Ke = np.array([[float(v) for v in row] for row in KeR])
    out = {"_source": "tests/golden/derive_appendix_b.py: exact sympy derivation of the reference's element formulas "
                      "(/root/reference/src/Discretization.c:49-128,293-402, corner order of :40-43) + an independent "
                      "numpy/scipy assembly and sparse LU.  Generated, not transcribed; no value comes from the oracle.",
           "Ke": [[float(v) for v in row] for row in Ke],
           "Ke_exact": [[str(v) for v in row] for row in KeR],
           "Ke_tol": 5e-14,   # floating-point evaluation of the element routine at arbitrary element positions; the
           # truncated-abscissa defect it must resolve is 3.2e-13
           "Fe_over_h2_quarter": [float(v) for v in FeR],
           "grids": {}, "interior_diag": float(4 * Ke[0, 0])}
    for M in (4, 32, 33, 256):
        A, f, bnd = assemble(M, Ke)
        u = sla.spsolve(A.tocsc(), f)
        nnz_pattern = 4 * (3 * M - 2) ** 2           # stored entries of the 9-point x 2 x 2 pattern
        out["grids"][str(M)] = {"rows": 2 * M * M, "nnz": nnz_pattern, "f_norm": float(np.linalg.norm(f)),
                                "u_norm": float(np.linalg.norm(u)), "u_max": float(np.abs(u).max())}
        if M == 4:
            out["m4_solution"] = {str(k): float(u[k]) for k in (10, 11, 12, 13, 18, 19, 20, 21)}
        if M == 32:
            B = constraints(M, bnd)
            D = sp.diags(1.0 / A.diagonal())
            shat = (B @ D @ B.T).diagonal()
            g = np.array([1e-2, -2e-2, 3e-3, 1e-3])
            K = sp.bmat([[A, B.T], [B, None]]).tocsc()
            sol = sla.spsolve(K, np.concatenate([f, g]))
            out["constraints_m32"] = {"nnz_B": int(B.nnz), "shat": [float(v) for v in shat],
                                      "u_norm": float(np.linalg.norm(sol[:-4])), "lambda": [float(v) for v in sol[-4:]],
                                      "g": [float(v) for v in g]}
    return out


def main():
    new = derive()
    path = os.path.join(HERE, "appendix_b.json")
    if "--check" in sys.argv:
        old = json.load(open(path))
        assert np.allclose(old["Ke"], new["Ke"], rtol=0, atol=1e-15)
        for M, ref in new["grids"].items():
            for k, v in ref.items():
                assert np.isclose(old["grids"][M][k], v, rtol=1e-9), (M, k)
        assert np.allclose(old["constraints_m32"]["shat"], new["constraints_m32"]["shat"], rtol=1e-9)
        assert np.allclose(old["constraints_m32"]["lambda"], new["constraints_m32"]["lambda"], rtol=1e-8)
        print("appendix_b.json matches a fresh derivation")
        return
    with open(path, "w") as fh:
        json.dump(new, fh, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
