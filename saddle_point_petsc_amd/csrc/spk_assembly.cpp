// spk_assembly.cpp -- threaded, PETSc-free assembler (see include/spk_assembly.h).
//
// Not the reference's scatter loop: rows are GATHERED.  A thread owns a run of
// node lines, computes the element matrices of the two adjacent element lines
// once, and writes each CSR row directly from the <= 4 elements around its
// node, visiting them in ascending (ej, ei) -- the order in which the
// reference's element loop (Discretization.c:146-147) would have added them --
// so every stored value is bit-identical to a sequential ADD_VALUES assembly
// while no two threads ever touch the same entry.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/spk_assembly.h"
#include "../../include/spk.h"

#pragma clang fp contract(off)

namespace {

// 2x2 Gauss points; the 11-digit abscissa is the reference's (Discretization.c:52-55)
const double kGp[4][2] = {{-0.57735026919, -0.57735026919},
                          {-0.57735026919, 0.57735026919},
                          {0.57735026919, 0.57735026919},
                          {0.57735026919, -0.57735026919}};

struct GaussPoint {
    double N[4];      // shape functions
    double dN[2][4];  // reference gradients
};

GaussPoint make_gp(int p)
{
    GaussPoint g;
    const double xi = kGp[p][0], eta = kGp[p][1];
    g.N[0] = 0.25 * (1.0 - xi) * (1.0 - eta);
    g.N[1] = 0.25 * (1.0 - xi) * (1.0 + eta);
    g.N[2] = 0.25 * (1.0 + xi) * (1.0 + eta);
    g.N[3] = 0.25 * (1.0 + xi) * (1.0 - eta);
    g.dN[0][0] = -0.25 * (1.0 - eta);
    g.dN[0][1] = -0.25 * (1.0 + eta);
    g.dN[0][2] = 0.25 * (1.0 + eta);
    g.dN[0][3] = 0.25 * (1.0 - eta);
    g.dN[1][0] = -0.25 * (1.0 - xi);
    g.dN[1][1] = 0.25 * (1.0 - xi);
    g.dN[1][2] = 0.25 * (1.0 + xi);
    g.dN[1][3] = -0.25 * (1.0 + xi);
    return g;
}

// physical gradients at one Gauss point; returns det J
double phys_grad(const GaussPoint &g, const double *xe, double gx[2][4])
{
    double J[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    for (int c = 0; c < 2; ++c)
        for (int d = 0; d < 2; ++d)
            for (int i = 0; i < 4; ++i) J[c][d] += g.dN[c][i] * xe[2 * i + d];
    const double det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
    const double i00 = J[1][1] / det, i01 = -J[0][1] / det, i10 = -J[1][0] / det, i11 = J[0][0] / det;
    for (int i = 0; i < 4; ++i) {
        gx[0][i] = i00 * g.dN[0][i] + i01 * g.dN[1][i];
        gx[1][i] = i10 * g.dN[0][i] + i11 * g.dN[1][i];
    }
    return det;
}

// Ke[a*8+b]; strain-displacement rows (exx, eyy, 2exy), D = diag(2,2,1)
void stiffness(const double *xe, const double *coeff, double *Ke)
{
    double acc[64];  // acc[i + 8 j], the reference's index (symmetric anyway)
    std::memset(acc, 0, sizeof acc);
    for (int p = 0; p < 4; ++p) {
        const GaussPoint g = make_gp(p);
        double gx[2][4], Bm[3][8], tD[3];
        const double det = phys_grad(g, xe, gx);
        for (int i = 0; i < 4; ++i) {
            Bm[0][2 * i] = gx[0][i]; Bm[0][2 * i + 1] = 0.0;
            Bm[1][2 * i] = 0.0;      Bm[1][2 * i + 1] = gx[1][i];
            Bm[2][2 * i] = gx[1][i]; Bm[2][2 * i + 1] = gx[0][i];
        }
        tD[0] = 2.0 * 1.0 * det * coeff[p];
        tD[1] = 2.0 * 1.0 * det * coeff[p];
        tD[2] = 1.0 * det * coeff[p];
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 8; ++j)
                for (int k = 0; k < 3; ++k) acc[i + 8 * j] += Bm[k][i] * tD[k] * Bm[k][j];
    }
    // the reference hands Ae to MatSetValuesStencil row-major: (row a, col b) = Ae[a*8+b]
    std::memcpy(Ke, acc, sizeof acc);
}

void load(const double *xe, double *Fe)
{
    std::memset(Fe, 0, 8 * sizeof(double));
    for (int p = 0; p < 4; ++p) {
        const GaussPoint g = make_gp(p);
        double gx[2][4];
        const double fac = 1.0 * phys_grad(g, xe, gx);
        const double body[2] = {1.0, 2.0};  // FormRHS, Discretization.c:397-402
        for (int i = 0; i < 4; ++i)
            for (int c = 0; c < 2; ++c) Fe[2 * i + c] += fac * g.N[i] * body[c];
    }
}

inline double coord(int i, int m) { return 0.0 + (1.0 / (double)(m - 1)) * (double)i; }

void element_coords(int mx, int my, int ei, int ej, double *xe)
{
    xe[0] = coord(ei, mx);     xe[1] = coord(ej, my);
    xe[2] = coord(ei, mx);     xe[3] = coord(ej + 1, my);
    xe[4] = coord(ei + 1, mx); xe[5] = coord(ej + 1, my);
    xe[6] = coord(ei + 1, mx); xe[7] = coord(ej, my);
}

// local node number of the corner at offset (oi, oj) from the element origin
inline int corner(int oi, int oj) { return oi == 0 ? (oj == 0 ? 0 : 1) : (oj == 0 ? 3 : 2); }

inline bool on_boundary(int mx, int my, int i, int j) { return i == 0 || i == mx - 1 || j == 0 || j == my - 1; }

// stored non-zeros of the two rows of node (i,j)
inline int node_row_nnz(int mx, int my, int i, int j)
{
    const int wi = (i > 0) + 1 + (i < mx - 1), wj = (j > 0) + 1 + (j < my - 1);
    return wi * wj * 2;
}

// ---- 3-D (build-defined; see include/spk_assembly.h) ----
const int kSgn3[8][3] = {{-1, -1, -1}, {-1, 1, -1}, {1, 1, -1}, {1, -1, -1}, {-1, -1, 1}, {-1, 1, 1}, {1, 1, 1}, {1, -1, 1}};

// Ke[a*24+b] and Fe[24] of one hexahedron (same operation order as the oracle's restatement)
void element3d(const double *xe, double *Ke, double *Fe)
{
    double acc[576];
    std::memset(acc, 0, sizeof acc);
    std::memset(Fe, 0, 24 * sizeof(double));
    const double gp1 = 0.57735026919;
    for (int p = 0; p < 8; ++p) {
        const double xi[3] = {kSgn3[p][0] * gp1, kSgn3[p][1] * gp1, kSgn3[p][2] * gp1};
        double N[8], G[3][8], Gx[3][8], Bm[6][24], tD[6], J[3][3], iJ[3][3];
        for (int a = 0; a < 8; ++a) {
            const double sx = kSgn3[a][0], sy = kSgn3[a][1], sz = kSgn3[a][2];
            N[a] = 0.125 * (1.0 + sx * xi[0]) * (1.0 + sy * xi[1]) * (1.0 + sz * xi[2]);
            G[0][a] = 0.125 * sx * (1.0 + sy * xi[1]) * (1.0 + sz * xi[2]);
            G[1][a] = 0.125 * sy * (1.0 + sx * xi[0]) * (1.0 + sz * xi[2]);
            G[2][a] = 0.125 * sz * (1.0 + sx * xi[0]) * (1.0 + sy * xi[1]);
        }
        for (int c = 0; c < 3; ++c)
            for (int d = 0; d < 3; ++d) {
                J[c][d] = 0.0;
                for (int a = 0; a < 8; ++a) J[c][d] += G[c][a] * xe[a * 3 + d];
            }
        const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                           J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
        iJ[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) / det;
        iJ[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
        iJ[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
        iJ[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) / det;
        iJ[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
        iJ[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
        iJ[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) / det;
        iJ[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
        iJ[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
        for (int a = 0; a < 8; ++a)
            for (int c = 0; c < 3; ++c) Gx[c][a] = iJ[c][0] * G[0][a] + iJ[c][1] * G[1][a] + iJ[c][2] * G[2][a];
        std::memset(Bm, 0, sizeof Bm);
        for (int a = 0; a < 8; ++a) {
            Bm[0][3 * a] = Gx[0][a];
            Bm[1][3 * a + 1] = Gx[1][a];
            Bm[2][3 * a + 2] = Gx[2][a];
            Bm[3][3 * a] = Gx[1][a];     Bm[3][3 * a + 1] = Gx[0][a];
            Bm[4][3 * a + 1] = Gx[2][a]; Bm[4][3 * a + 2] = Gx[1][a];
            Bm[5][3 * a] = Gx[2][a];     Bm[5][3 * a + 2] = Gx[0][a];
        }
        for (int k = 0; k < 6; ++k) tD[k] = (k < 3 ? 2.0 : 1.0) * 1.0 * det * 1.0;
        for (int i = 0; i < 24; ++i)
            for (int j = 0; j < 24; ++j)
                for (int k = 0; k < 6; ++k) acc[i + 24 * j] += Bm[k][i] * tD[k] * Bm[k][j];
        const double fac = 1.0 * det, body[3] = {1.0, 2.0, 3.0};
        for (int a = 0; a < 8; ++a)
            for (int c = 0; c < 3; ++c) Fe[3 * a + c] += fac * N[a] * body[c];
    }
    std::memcpy(Ke, acc, sizeof acc);
}

inline int corner3(int oi, int oj, int ok) { return corner(oi, oj) + 4 * ok; }
inline bool on_boundary3(int mx, int my, int mz, int i, int j, int k)
{
    return i == 0 || i == mx - 1 || j == 0 || j == my - 1 || k == 0 || k == mz - 1;
}
inline int width(int i, int m) { return (i > 0) + 1 + (i < m - 1); }

int threads_or_default(int n)
{
    if (n > 0) return n;
    const unsigned h = std::thread::hardware_concurrency();
    return h ? (int)h : 1;
}

}  // namespace

extern "C" {

int SpkAssemblySizes(int mx, int my, int64_t *nrows, int64_t *nnz)
{
    if (mx < 2 || my < 2) return SPK_ERR_ARG;
    if (nrows) *nrows = (int64_t)2 * mx * my;
    if (nnz) *nnz = (int64_t)4 * (3 * (int64_t)mx - 2) * (3 * (int64_t)my - 2);
    return SPK_OK;
}

int64_t SpkAssemblySlabNnz(int mx, int my, int64_t row_begin, int64_t row_end)
{
    const int64_t line = (int64_t)2 * mx;
    if (mx < 2 || my < 2 || row_begin % line || row_end % line || row_begin > row_end || row_end > line * my) return -1;
    int64_t nnz = 0;
    for (int64_t j = row_begin / line; j < row_end / line; ++j) {
        const int64_t wj = (j > 0) + 1 + (j < my - 1);
        nnz += 4 * wj * (3 * (int64_t)mx - 2);
    }
    return nnz;
}

int SpkAssembleOperator_Laplace(int mx, int my, int64_t row_begin, int64_t row_end, int32_t *rowptr,
                                int32_t *colidx, double *val, double *f, int apply_bc, int nthreads)
{
    const int64_t line = (int64_t)2 * mx;
    if (!rowptr || !colidx || !val) return SPK_ERR_ARG;
    if (SpkAssemblySlabNnz(mx, my, row_begin, row_end) < 0) return SPK_ERR_ARG;
    if ((int64_t)2 * mx * my > INT32_MAX) return SPK_ERR_UNSUPPORTED;
    const int j0 = (int)(row_begin / line), j1 = (int)(row_end / line);

    // row pointers (closed form per node)
    {
        int64_t k = 0;
        int64_t r = 0;
        for (int j = j0; j < j1; ++j)
            for (int i = 0; i < mx; ++i) {
                const int w = node_row_nnz(mx, my, i, j);
                rowptr[r++] = (int32_t)k; k += w;
                rowptr[r++] = (int32_t)k; k += w;
            }
        rowptr[r] = (int32_t)k;
        if (k > INT32_MAX) return SPK_ERR_UNSUPPORTED;
    }

    const int nt = std::max(1, std::min(threads_or_default(nthreads), j1 - j0));
    auto work = [&](int t) {
        const int ja = j0 + (int)((int64_t)(j1 - j0) * t / nt), jb = j0 + (int)((int64_t)(j1 - j0) * (t + 1) / nt);
        const int ne = mx - 1;
        // element matrices / loads of element lines below (ej = j-1) and above (ej = j)
        std::vector<double> KeLo((size_t)ne * 64), KeHi((size_t)ne * 64), FeLo((size_t)ne * 8), FeHi((size_t)ne * 8);
        const double coeff[4] = {1.0, 1.0, 1.0, 1.0};
        auto fill_line = [&](int ej, std::vector<double> &Ke, std::vector<double> &Fe) {
            if (ej < 0 || ej > my - 2) return;
            for (int ei = 0; ei < ne; ++ei) {
                double xe[8];
                element_coords(mx, my, ei, ej, xe);
                stiffness(xe, coeff, &Ke[(size_t)ei * 64]);
                load(xe, &Fe[(size_t)ei * 8]);
            }
        };
        fill_line(ja - 1, KeLo, FeLo);
        for (int j = ja; j < jb; ++j) {
            fill_line(j, KeHi, FeHi);
            for (int i = 0; i < mx; ++i) {
                const bool rb = on_boundary(mx, my, i, j);
                for (int c = 0; c < 2; ++c) {
                    const int64_t grow = ((int64_t)j * mx + i) * 2 + c;
                    const int64_t lrow = grow - row_begin;
                    int64_t k = rowptr[lrow];
                    for (int dj = -1; dj <= 1; ++dj) {
                        const int cj = j + dj;
                        if (cj < 0 || cj >= my) continue;
                        for (int di = -1; di <= 1; ++di) {
                            const int ci = i + di;
                            if (ci < 0 || ci >= mx) continue;
                            const bool cb = on_boundary(mx, my, ci, cj);
                            for (int d = 0; d < 2; ++d) {
                                const int64_t gcol = ((int64_t)cj * mx + ci) * 2 + d;
                                double v = 0.0;
                                // elements that hold both nodes, ascending (ej, ei)
                                for (int ej = std::max(j, cj) - 1; ej <= std::min(j, cj); ++ej) {
                                    if (ej < 0 || ej > my - 2) continue;
                                    const std::vector<double> &Ke = (ej == j - 1) ? KeLo : KeHi;
                                    for (int ei = std::max(i, ci) - 1; ei <= std::min(i, ci); ++ei) {
                                        if (ei < 0 || ei > mx - 2) continue;
                                        const int a = corner(i - ei, j - ej) * 2 + c;
                                        const int b = corner(ci - ei, cj - ej) * 2 + d;
                                        v += Ke[(size_t)ei * 64 + a * 8 + b];
                                    }
                                }
                                if (apply_bc && (rb || cb)) v = (gcol == grow) ? 1.0 : 0.0;
                                colidx[k] = (int32_t)gcol;
                                val[k] = v;
                                ++k;
                            }
                        }
                    }
                    if (f) {
                        double fv = 0.0;
                        for (int ej = j - 1; ej <= j; ++ej) {
                            if (ej < 0 || ej > my - 2) continue;
                            const std::vector<double> &Fe = (ej == j - 1) ? FeLo : FeHi;
                            for (int ei = i - 1; ei <= i; ++ei) {
                                if (ei < 0 || ei > mx - 2) continue;
                                fv += Fe[(size_t)ei * 8 + corner(i - ei, j - ej) * 2 + c];
                            }
                        }
                        f[lrow] = (apply_bc && rb) ? 0.0 : fv;
                    }
                }
            }
            std::swap(KeLo, KeHi);
            std::swap(FeLo, FeHi);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    return SPK_OK;
}

int64_t SpkConstraintsSlabNnz(int mx, int my, int64_t row_begin, int64_t row_end)
{
    const int64_t line = (int64_t)2 * mx;
    if (mx < 3 || my < 3 || row_begin % line || row_end % line || row_begin > row_end || row_end > line * my) return -1;
    int64_t lines = 0;
    for (int64_t j = row_begin / line; j < row_end / line; ++j) lines += (j > 0 && j < my - 1);
    return 4 * lines * (mx - 2);
}

int SpkAssembleOperator_Constraints(int mx, int my, int64_t row_begin, int64_t row_end, int32_t *rowptr,
                                    int32_t *colidx, double *val)
{
    if (!rowptr || !colidx || !val || SpkConstraintsSlabNnz(mx, my, row_begin, row_end) < 0) return SPK_ERR_ARG;
    const int64_t line = (int64_t)2 * mx;
    const int j0 = (int)(row_begin / line), j1 = (int)(row_end / line);
    const double hx = 1.0 / (double)(mx - 1), hy = 1.0 / (double)(my - 1), w = hx * hy;
    int64_t k = 0;
    for (int r = 0; r < 4; ++r) {
        rowptr[r] = (int32_t)k;
        for (int j = std::max(j0, 1); j < std::min(j1, my - 1); ++j)
            for (int i = 1; i < mx - 1; ++i) {
                double v = w;
                if (r == 2) v = w * (coord(i, mx) - 0.5);
                if (r == 3) v = w * (coord(j, my) - 0.5);
                colidx[k] = (int32_t)(((int64_t)j * mx + i) * 2 + (r & 1));
                val[k] = v;
                ++k;
            }
    }
    rowptr[4] = (int32_t)k;
    return SPK_OK;
}

int SpkAssembleRHS_Constraints(double *g)
{
    if (!g) return SPK_ERR_ARG;
    g[0] = 1e-2; g[1] = -2e-2; g[2] = 3e-3; g[3] = 1e-3;
    return SPK_OK;
}

int SpkAssemblySizes3D(int mx, int my, int mz, int64_t *nrows, int64_t *nnz)
{
    if (mx < 2 || my < 2 || mz < 2) return SPK_ERR_ARG;
    if (nrows) *nrows = (int64_t)3 * mx * my * mz;
    if (nnz) *nnz = (int64_t)9 * (3 * (int64_t)mx - 2) * (3 * (int64_t)my - 2) * (3 * (int64_t)mz - 2);
    return SPK_OK;
}

int64_t SpkAssemblySlabNnz3D(int mx, int my, int mz, int64_t row_begin, int64_t row_end)
{
    const int64_t plane = (int64_t)3 * mx * my;
    if (mx < 2 || my < 2 || mz < 2 || row_begin % plane || row_end % plane || row_begin > row_end || row_end > plane * mz) return -1;
    int64_t nnz = 0;
    for (int64_t k = row_begin / plane; k < row_end / plane; ++k)
        nnz += (int64_t)9 * width((int)k, mz) * (3 * (int64_t)mx - 2) * (3 * (int64_t)my - 2);
    return nnz;
}

int SpkAssembleOperator_Laplace3D(int mx, int my, int mz, int64_t row_begin, int64_t row_end, int32_t *rowptr,
                                  int32_t *colidx, double *val, double *f, int apply_bc, int nthreads)
{
    const int64_t plane = (int64_t)3 * mx * my;
    if (!rowptr || !colidx || !val) return SPK_ERR_ARG;
    const int64_t total = SpkAssemblySlabNnz3D(mx, my, mz, row_begin, row_end);
    if (total < 0) return SPK_ERR_ARG;
    if ((int64_t)3 * mx * my * mz > INT32_MAX || total > INT32_MAX) return SPK_ERR_UNSUPPORTED;
    const int k0 = (int)(row_begin / plane), k1 = (int)(row_end / plane);
    {
        int64_t q = 0, r = 0;
        for (int k = k0; k < k1; ++k)
            for (int j = 0; j < my; ++j)
                for (int i = 0; i < mx; ++i) {
                    const int w = width(i, mx) * width(j, my) * width(k, mz) * 3;
                    for (int c = 0; c < 3; ++c) { rowptr[r++] = (int32_t)q; q += w; }
                }
        rowptr[r] = (int32_t)q;
    }
    const int nlines = (k1 - k0) * my;  // node lines (j, k) of the slab
    const int nt = std::max(1, std::min(threads_or_default(nthreads), nlines));
    const int ne = mx - 1;
    auto work = [&](int t) {
        const int la = (int)((int64_t)nlines * t / nt), lb = (int)((int64_t)nlines * (t + 1) / nt);
        // element lines around the current node line: slot [dj][dk] holds (ej = j-1+dj, ek = k-1+dk)
        std::vector<double> Ke[2][2], Fe[2][2];
        int tag_j[2][2], tag_k[2][2];
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b) {
                Ke[a][b].resize((size_t)ne * 576);
                Fe[a][b].resize((size_t)ne * 24);
                tag_j[a][b] = tag_k[a][b] = -2;
            }
        auto ensure = [&](int dj, int dk, int ej, int ek) {
            if (ej < 0 || ej > my - 2 || ek < 0 || ek > mz - 2) return;
            if (tag_j[dj][dk] == ej && tag_k[dj][dk] == ek) return;
            // the line computed as "upper in j" a moment ago is the "lower in j" one now
            if (dj == 0 && tag_j[1][dk] == ej && tag_k[1][dk] == ek) {
                std::swap(Ke[0][dk], Ke[1][dk]);
                std::swap(Fe[0][dk], Fe[1][dk]);
                std::swap(tag_j[0][dk], tag_j[1][dk]);
                std::swap(tag_k[0][dk], tag_k[1][dk]);
                return;
            }
            for (int ei = 0; ei < ne; ++ei) {
                double xe[24];
                for (int a = 0; a < 8; ++a) {
                    xe[3 * a] = coord(ei + (kSgn3[a][0] > 0), mx);
                    xe[3 * a + 1] = coord(ej + (kSgn3[a][1] > 0), my);
                    xe[3 * a + 2] = coord(ek + (kSgn3[a][2] > 0), mz);
                }
                element3d(xe, &Ke[dj][dk][(size_t)ei * 576], &Fe[dj][dk][(size_t)ei * 24]);
            }
            tag_j[dj][dk] = ej;
            tag_k[dj][dk] = ek;
        };
        for (int l = la; l < lb; ++l) {
            const int k = k0 + l / my, j = l % my;
            for (int dk = 0; dk < 2; ++dk) {
                ensure(0, dk, j - 1, k - 1 + dk);
                ensure(1, dk, j, k - 1 + dk);
            }
            for (int i = 0; i < mx; ++i) {
                const bool rb = on_boundary3(mx, my, mz, i, j, k);
                for (int c = 0; c < 3; ++c) {
                    const int64_t grow = (((int64_t)k * my + j) * mx + i) * 3 + c;
                    const int64_t lrow = grow - row_begin;
                    int64_t q = rowptr[lrow];
                    for (int dk = -1; dk <= 1; ++dk) {
                        const int ck = k + dk;
                        if (ck < 0 || ck >= mz) continue;
                        for (int dj = -1; dj <= 1; ++dj) {
                            const int cj = j + dj;
                            if (cj < 0 || cj >= my) continue;
                            for (int di = -1; di <= 1; ++di) {
                                const int ci = i + di;
                                if (ci < 0 || ci >= mx) continue;
                                const bool cb = on_boundary3(mx, my, mz, ci, cj, ck);
                                for (int d = 0; d < 3; ++d) {
                                    const int64_t gcol = (((int64_t)ck * my + cj) * mx + ci) * 3 + d;
                                    double v = 0.0;
                                    for (int ek = std::max(k, ck) - 1; ek <= std::min(k, ck); ++ek) {
                                        if (ek < 0 || ek > mz - 2) continue;
                                        for (int ej = std::max(j, cj) - 1; ej <= std::min(j, cj); ++ej) {
                                            if (ej < 0 || ej > my - 2) continue;
                                            const std::vector<double> &K = Ke[ej - (j - 1)][ek - (k - 1)];
                                            for (int ei = std::max(i, ci) - 1; ei <= std::min(i, ci); ++ei) {
                                                if (ei < 0 || ei > mx - 2) continue;
                                                const int a = corner3(i - ei, j - ej, k - ek) * 3 + c;
                                                const int b = corner3(ci - ei, cj - ej, ck - ek) * 3 + d;
                                                v += K[(size_t)ei * 576 + a * 24 + b];
                                            }
                                        }
                                    }
                                    if (apply_bc && (rb || cb)) v = (gcol == grow) ? 1.0 : 0.0;
                                    colidx[q] = (int32_t)gcol;
                                    val[q++] = v;
                                }
                            }
                        }
                    }
                    if (f) {
                        double fv = 0.0;
                        for (int ek = k - 1; ek <= k; ++ek) {
                            if (ek < 0 || ek > mz - 2) continue;
                            for (int ej = j - 1; ej <= j; ++ej) {
                                if (ej < 0 || ej > my - 2) continue;
                                const std::vector<double> &F = Fe[ej - (j - 1)][ek - (k - 1)];
                                for (int ei = i - 1; ei <= i; ++ei) {
                                    if (ei < 0 || ei > mx - 2) continue;
                                    fv += F[(size_t)ei * 24 + corner3(i - ei, j - ej, k - ek) * 3 + c];
                                }
                            }
                        }
                        f[lrow] = (apply_bc && rb) ? 0.0 : fv;
                    }
                }
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    return SPK_OK;
}

int64_t SpkConstraintsSlabNnz3D(int mx, int my, int mz, int64_t row_begin, int64_t row_end)
{
    const int64_t plane = (int64_t)3 * mx * my;
    if (mx < 3 || my < 3 || mz < 3 || row_begin % plane || row_end % plane || row_begin > row_end || row_end > plane * mz) return -1;
    int64_t planes = 0;
    for (int64_t k = row_begin / plane; k < row_end / plane; ++k) planes += (k > 0 && k < mz - 1);
    return 6 * planes * (mx - 2) * (my - 2);
}

int SpkAssembleOperator_Constraints3D(int mx, int my, int mz, int64_t row_begin, int64_t row_end, int32_t *rowptr,
                                      int32_t *colidx, double *val)
{
    if (!rowptr || !colidx || !val || SpkConstraintsSlabNnz3D(mx, my, mz, row_begin, row_end) < 0) return SPK_ERR_ARG;
    const int64_t plane = (int64_t)3 * mx * my;
    const int k0 = (int)(row_begin / plane), k1 = (int)(row_end / plane);
    const double w = (1.0 / (mx - 1)) * (1.0 / (my - 1)) * (1.0 / (mz - 1));
    int64_t q = 0;
    for (int r = 0; r < 6; ++r) {
        rowptr[r] = (int32_t)q;
        for (int k = std::max(k0, 1); k < std::min(k1, mz - 1); ++k)
            for (int j = 1; j < my - 1; ++j)
                for (int i = 1; i < mx - 1; ++i) {
                    double v = w;
                    if (r == 3) v = w * (coord(i, mx) - 0.5);
                    if (r == 4) v = w * (coord(j, my) - 0.5);
                    if (r == 5) v = w * (coord(k, mz) - 0.5);
                    colidx[q] = (int32_t)((((int64_t)k * my + j) * mx + i) * 3 + r % 3);
                    val[q++] = v;
                }
    }
    rowptr[6] = (int32_t)q;
    return SPK_OK;
}

// Discrete divergence block, one row per hexahedron (see include/spk_assembly.h): the entries of a row over
// the columns [row_begin, row_end) this rank owns.
static int64_t divergence_slab(int mx, int my, int mz, int64_t row_begin, int64_t row_end, int32_t *rowptr, int32_t *colidx,
                               double *val)
{
    const double hx = 1.0 / (mx - 1), hy = 1.0 / (my - 1), hz = 1.0 / (mz - 1);
    const double fx = (hy * hz) / 4.0, fy = (hx * hz) / 4.0, fz = (hx * hy) / 4.0;
    int64_t q = 0;
    int64_t e = 0;
    for (int ek = 0; ek < mz - 1; ++ek)
        for (int ej = 0; ej < my - 1; ++ej)
            for (int ei = 0; ei < mx - 1; ++ei) {
                if (rowptr) rowptr[e] = (int32_t)q;
                ++e;
                for (int dk = 0; dk < 2; ++dk)
                    for (int dj = 0; dj < 2; ++dj)
                        for (int di = 0; di < 2; ++di) {
                            const int i = ei + di, j = ej + dj, k = ek + dk;
                            if (i == 0 || i == mx - 1 || j == 0 || j == my - 1 || k == 0 || k == mz - 1) continue;
                            const int64_t c0 = (((int64_t)k * my + j) * mx + i) * 3;
                            if (c0 < row_begin || c0 >= row_end) continue;   // whole nodes belong to one rank
                            if (colidx) {
                                colidx[q] = (int32_t)c0;       val[q] = di ? fx : -fx;
                                colidx[q + 1] = (int32_t)c0 + 1; val[q + 1] = dj ? fy : -fy;
                                colidx[q + 2] = (int32_t)c0 + 2; val[q + 2] = dk ? fz : -fz;
                            }
                            q += 3;
                        }
            }
    if (rowptr) rowptr[e] = (int32_t)q;
    return q;
}

int64_t SpkDivergenceSlabNnz3D(int mx, int my, int mz, int64_t row_begin, int64_t row_end)
{
    const int64_t plane = (int64_t)3 * mx * my;
    if (mx < 3 || my < 3 || mz < 3 || row_begin % plane || row_end % plane || row_begin > row_end || row_end > plane * mz) return -1;
    return divergence_slab(mx, my, mz, row_begin, row_end, nullptr, nullptr, nullptr);
}

int SpkAssembleOperator_Divergence3D(int mx, int my, int mz, int64_t row_begin, int64_t row_end, int32_t *rowptr,
                                     int32_t *colidx, double *val)
{
    if (!rowptr || !colidx || !val || SpkDivergenceSlabNnz3D(mx, my, mz, row_begin, row_end) < 0) return SPK_ERR_ARG;
    divergence_slab(mx, my, mz, row_begin, row_end, rowptr, colidx, val);
    return SPK_OK;
}

int SpkAssembleRHS_Constraints3D(double *g)
{
    if (!g) return SPK_ERR_ARG;
    g[0] = 1e-2; g[1] = -2e-2; g[2] = 3e-3; g[3] = 1e-3; g[4] = 2e-3; g[5] = -1e-3;
    return SPK_OK;
}

int SpkWriteVTK(int mx, int my, const double *u, const char *filename)
{
    if (mx < 2 || my < 2 || !u || !filename) return SPK_ERR_ARG;
    FILE *fp = std::fopen(filename, "w");
    if (!fp) return SPK_ERR_ARG;
    std::fprintf(fp, "# vtk DataFile Version 3.0\nsaddle point solution U on a %d x %d node grid\nASCII\n", mx, my);
    std::fprintf(fp, "DATASET STRUCTURED_GRID\nDIMENSIONS %d %d 1\nPOINTS %lld double\n", mx, my, (long long)mx * my);
    for (int j = 0; j < my; ++j)
        for (int i = 0; i < mx; ++i) std::fprintf(fp, "%.17g %.17g 0\n", coord(i, mx), coord(j, my));
    std::fprintf(fp, "POINT_DATA %lld\nVECTORS U double\n", (long long)mx * my);
    for (int64_t p = 0; p < (int64_t)mx * my; ++p) std::fprintf(fp, "%.17g %.17g 0\n", u[2 * p], u[2 * p + 1]);
    const bool ok = std::ferror(fp) == 0;
    return (std::fclose(fp) == 0 && ok) ? SPK_OK : SPK_ERR_ARG;
}

int SpkFormStressOperatorQ12D(const double *xe, const double *coeff4, double *Ke64)
{
    if (!xe || !coeff4 || !Ke64) return SPK_ERR_ARG;
    stiffness(xe, coeff4, Ke64);
    return SPK_OK;
}

int SpkFormLaplaceRHSQ12D(const double *xe, double *Fe8)
{
    if (!xe || !Fe8) return SPK_ERR_ARG;
    load(xe, Fe8);
    return SPK_OK;
}

}  // extern "C"
