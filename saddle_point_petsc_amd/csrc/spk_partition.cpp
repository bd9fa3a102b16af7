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
#include <thread>

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

void parallel_for(int64_t n, const std::function<void(int64_t, int64_t, int)> &fn, int max_threads)
{
    int nt = max_threads > 0 ? max_threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 32) nt = 32;  // setup work is memory bound; stay well inside the host's limits
    if ((int64_t)nt > n / 4096 + 1) nt = (int)(n / 4096 + 1);
    if (nt <= 1) {
        fn(0, n, 0);
        return;
    }
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back([&, t] { fn(n * t / nt, n * (t + 1) / nt, t); });
    fn(0, n / nt, 0);
    for (auto &th : pool) th.join();
}

// Two threaded passes over the rows: count (and collect the off-range columns), then fill.
void split_csr(int64_t row_begin, int32_t nrows_local, const int32_t *rowptr, const int32_t *colidx,
               const double *val, SplitCsr &out, int64_t ncols_global)
{
    const int64_t lo = row_begin, hi = row_begin + nrows_local;
    out.d_rowptr.alloc((size_t)nrows_local + 1);
    out.o_rowptr.alloc((size_t)nrows_local + 1);
    out.bad_column = false;
    std::vector<std::vector<int32_t>> ghost_parts(64);
    std::vector<int> bad(64, 0);
    std::vector<int32_t> badv(64, 0);
    parallel_for(nrows_local, [&](int64_t r0, int64_t r1, int t) {
        auto &gp = ghost_parts[(size_t)t];
        for (int64_t r = r0; r < r1; ++r) {
            int32_t nd = 0, no = 0;
            for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
                const int32_t c = colidx[k];
                if (c >= lo && c < hi) ++nd;
                else {
                    ++no;
                    if (ncols_global >= 0 && (c < 0 || c >= ncols_global)) { bad[(size_t)t] = 1; badv[(size_t)t] = c; }
                    if (gp.empty() || gp.back() != c) gp.push_back(c);
                }
            }
            out.d_rowptr[(size_t)r + 1] = nd;
            out.o_rowptr[(size_t)r + 1] = no;
        }
    });
    for (size_t t = 0; t < bad.size(); ++t)
        if (bad[t]) { out.bad_column = true; out.bad_value = badv[t]; }
    std::vector<int32_t> ghosts;
    for (auto &gp : ghost_parts) ghosts.insert(ghosts.end(), gp.begin(), gp.end());
    std::sort(ghosts.begin(), ghosts.end());
    ghosts.erase(std::unique(ghosts.begin(), ghosts.end()), ghosts.end());
    out.garray = ghosts;
    out.d_rowptr[0] = 0;
    out.o_rowptr[0] = 0;
    for (int32_t r = 0; r < nrows_local; ++r) {
        out.d_rowptr[(size_t)r + 1] += out.d_rowptr[(size_t)r];
        out.o_rowptr[(size_t)r + 1] += out.o_rowptr[(size_t)r];
    }
    const size_t nd = (size_t)out.d_rowptr[(size_t)nrows_local], no = (size_t)out.o_rowptr[(size_t)nrows_local];
    out.d_colidx.alloc(nd);
    out.d_val.alloc(nd);
    out.o_colidx.alloc(no);
    out.o_val.alloc(no);
    if (out.bad_column) return;
    parallel_for(nrows_local, [&](int64_t r0, int64_t r1, int) {
        for (int64_t r = r0; r < r1; ++r) {
            int32_t kd = out.d_rowptr[(size_t)r], ko = out.o_rowptr[(size_t)r];
            for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
                const int32_t c = colidx[k];
                if (c >= lo && c < hi) {
                    out.d_colidx[(size_t)kd] = (int32_t)(c - lo);
                    out.d_val[(size_t)kd++] = val[k];
                } else {
                    out.o_colidx[(size_t)ko] = (int32_t)(std::lower_bound(ghosts.begin(), ghosts.end(), c) - ghosts.begin());
                    out.o_val[(size_t)ko++] = val[k];
                }
            }
        }
    });
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
            if (dst && v.size()) std::memcpy(dst, v.data(), v.size() * sizeof(v[0]));
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
