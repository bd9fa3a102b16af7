// spk_partition.cpp -- host-only row-slab partition and diagonal/off-rank
// split of a CSR slab.  This is the job MatMPIAIJ does inside PETSc when the
// reference creates its DMDA on PETSC_COMM_WORLD
// (/root/reference/src/Discretization.c:17, SaddlePointProblem.c:42): local
// rows, a "diagonal" block with local column numbers, an "off-diagonal" block
// whose columns are renumbered into a sorted ghost list (garray).
// No HIP call in this file: it is exercised by the CPU-only tests.
#include <algorithm>
#include <cstdarg>
#include <cstring>

#include "spk_internal.hpp"

namespace spk {

void fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error{code, std::string(buf)};
}

void Comm::memcpy_self(const void *in, void *out, size_t b) { std::memcpy(out, in, b); }

void split_csr(int64_t row_begin, int32_t nrows_local, const int32_t *rowptr, const int32_t *colidx,
               const double *val, SplitCsr &out)
{
    const int64_t lo = row_begin, hi = row_begin + nrows_local;
    const int64_t nnz = rowptr[nrows_local];
    // ghost list = sorted unique off-range columns
    std::vector<int32_t> ghosts;
    for (int64_t k = 0; k < nnz; ++k)
        if (colidx[k] < lo || colidx[k] >= hi) ghosts.push_back(colidx[k]);
    std::sort(ghosts.begin(), ghosts.end());
    ghosts.erase(std::unique(ghosts.begin(), ghosts.end()), ghosts.end());
    out.garray = ghosts;

    out.d_rowptr.assign(nrows_local + 1, 0);
    out.o_rowptr.assign(nrows_local + 1, 0);
    out.d_colidx.clear();
    out.d_val.clear();
    out.o_colidx.clear();
    out.o_val.clear();
    out.d_colidx.reserve(nnz);
    out.d_val.reserve(nnz);
    for (int32_t r = 0; r < nrows_local; ++r) {
        for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
            const int32_t c = colidx[k];
            if (c >= lo && c < hi) {
                out.d_colidx.push_back((int32_t)(c - lo));
                out.d_val.push_back(val[k]);
            } else {
                const int32_t g = (int32_t)(std::lower_bound(ghosts.begin(), ghosts.end(), c) - ghosts.begin());
                out.o_colidx.push_back(g);
                out.o_val.push_back(val[k]);
            }
        }
        out.d_rowptr[r + 1] = (int32_t)out.d_colidx.size();
        out.o_rowptr[r + 1] = (int32_t)out.o_colidx.size();
    }
}

}  // namespace spk

extern "C" int spk_partition_slab(int64_t nlines, int64_t line_rows, int rank, int nranks,
                                  int64_t *row_begin, int64_t *row_end)
{
    if (nranks <= 0 || rank < 0 || rank >= nranks || nlines < 0 || line_rows <= 0) return SPK_ERR_ARG;
    // PETSc's DMDA split of `nlines` lines over `nranks`: the first (nlines % nranks)
    // ranks get one extra line.
    const int64_t base = nlines / nranks, extra = nlines % nranks;
    const int64_t l0 = rank * base + std::min<int64_t>(rank, extra);
    const int64_t l1 = l0 + base + (rank < extra ? 1 : 0);
    if (row_begin) *row_begin = l0 * line_rows;
    if (row_end) *row_end = l1 * line_rows;
    return SPK_OK;
}

extern "C" int spk_partition_split(int64_t row_begin, int32_t nrows_local, const int32_t *rowptr,
                                   const int32_t *colidx, const double *val, int32_t *d_rowptr,
                                   int32_t *d_colidx, double *d_val, int32_t *o_rowptr,
                                   int32_t *o_colidx, double *o_val, int32_t *garray, int64_t *nnz_d,
                                   int64_t *nnz_o, int32_t *n_ghost)
{
    if (!rowptr || !colidx || !val || nrows_local < 0) return SPK_ERR_ARG;
    try {
        spk::SplitCsr s;
        spk::split_csr(row_begin, nrows_local, rowptr, colidx, val, s);
        if (nnz_d) *nnz_d = (int64_t)s.d_colidx.size();
        if (nnz_o) *nnz_o = (int64_t)s.o_colidx.size();
        if (n_ghost) *n_ghost = (int32_t)s.garray.size();
        auto cp = [](auto *dst, const auto &v) {
            if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(v[0]));
        };
        cp(d_rowptr, s.d_rowptr);
        cp(d_colidx, s.d_colidx);
        cp(d_val, s.d_val);
        cp(o_rowptr, s.o_rowptr);
        cp(o_colidx, s.o_colidx);
        cp(o_val, s.o_val);
        cp(garray, s.garray);
    } catch (...) {
        return SPK_ERR_NOMEM;
    }
    return SPK_OK;
}
