// spk_k_spmv.hip -- MatMult on the A block (CSR stream kernel, 2x2- and 3x3-blocked kernels, the FP32 Richardson
// sweeps on the same layouts) and KSPSetOperators on the device (split, scan, blocking).  gfx950, wave64, HBM-bound.
#include "spk_device.hpp"

namespace spk {
namespace k {

// ---------------------------------------------------------------------------
// CSR stream SpMV (A block).  One workgroup = one row tile whose non-zeros
// (<= 4096) are streamed with 16-byte loads, multiplied by gathered x and
// staged in LDS; then one thread per row adds its products in CSR order --
// the same order and roundings as a sequential CSR loop.
// Arrays are padded by >= 8 entries so whole quads can be loaded unguarded.
// ---------------------------------------------------------------------------
// Stored non-zeros per tile of the CSR stream kernel.  Measured at M = 1024 (same run):
// 4096 -> 99.9 us, 2048 -> 87.1 us, 1024 -> 87.3 us, 512 (one wave per tile) -> 86.4 us;
// non-temporal loads on the matrix stream: 102 us (slower; not used for CSR).
constexpr int kCsrTile = 2048;

void build_tiles(const int32_t *rowptr, int32_t nrows, std::vector<int32_t> &tile_row)
{
    const int kTileNnz = kCsrTile, kTileRows = kThreads;
    tile_row.clear();
    tile_row.push_back(0);
    int32_t r = 0;
    while (r < nrows) {
        const int32_t r0 = r;
        const int64_t a0 = (int64_t)rowptr[r0] & ~(int64_t)3;
        while (r < nrows && (r - r0) < kTileRows && ((int64_t)rowptr[r + 1] - a0) <= kTileNnz) ++r;
        if (r == r0) ++r;  // one row longer than a tile: long-row path
        tile_row.push_back(r);
    }
}

template <bool NT, int TILE, int T, bool RIDE>
__global__ __launch_bounds__(T) void spmv_stream_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
    const double *__restrict__ val, const int32_t *__restrict__ tile_row, int ntiles,
    int tiles_per_xcd, const double *__restrict__ x, double *__restrict__ y,
    const int32_t *__restrict__ bt_rowptr, const int32_t *__restrict__ bt_colidx,
    const double *__restrict__ bt_val, const double *__restrict__ lam, int accumulate, OffDiag od,
    const int32_t *__restrict__ done, GivensRider gr)
{
    if (done && *done) return;
    __shared__ double prod[TILE + 8];
    if (RIDE && blockIdx.x == 0) {  // the rider: a pending Givens step beside the tiles (its LDS: the product buffer)
        givens_rider(gr, prod);
        return;
    }
    // workgroups b, b+8, ... share an XCD (round-robin dispatch): give each XCD
    // a contiguous run of row tiles so the x window stays in ITS L2.
    const int bx = (int)blockIdx.x - (RIDE ? 1 : 0);
    const int t = (bx & 7) * tiles_per_xcd + (bx >> 3);
    if ((bx >> 3) >= tiles_per_xcd || t >= ntiles) return;

    const int r0 = tile_row[t], r1 = tile_row[t + 1];
    const int nz0 = rowptr[r0], nz1 = rowptr[r1];
    const int a0 = nz0 & ~3;
    const int cnt = nz1 - a0;

    if (cnt > TILE) {
        // a single row longer than a tile: strided partial sums + block reduce
        double acc[1] = {0.0};
        for (int k = nz0 + threadIdx.x; k < nz1; k += T) acc[0] += val[k] * x[colidx[k]];
        double out1;
        __shared__ double red[T / 64];
        const double s = wave_sum(acc[0]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            out1 = 0.0;
            for (int j = 0; j < T / 64; ++j) out1 += red[j];
            if (od.rowptr)
                for (int k = od.rowptr[r0]; k < od.rowptr[r0 + 1]; ++k) out1 += od.val[k] * od.xg[od.colidx[k]];
            if (bt_rowptr)
                for (int k = bt_rowptr[r0]; k < bt_rowptr[r0 + 1]; ++k) out1 += bt_val[k] * lam[bt_colidx[k]];
            if (accumulate) out1 += y[r0];
            y[r0] = out1;
        }
        return;
    }

    // the row phase's own loads first (see spmv_bcsr_kernel)
    const int r = r0 + threadIdx.x;
    int k0 = 0, k1 = 0;
    double yacc = 0.0;
    if (r < r1) {
        k0 = rowptr[r] - a0;
        k1 = rowptr[r + 1] - a0;
        if (accumulate) yacc = y[r];
    }
    // phase 1: issue every load of the tile first, then gather x, then stage.
    constexpr int kSteps = TILE / (T * 4);
    int4 c[kSteps];
    double2 v0[kSteps], v1[kSteps];
#pragma unroll
    for (int i = 0; i < kSteps; ++i) {
        const int q = (i * T + threadIdx.x) * 4;
        if (q < cnt) {
            c[i] = ld4i<NT>(colidx + a0 + q);
            v0[i] = ld2s<NT>(val + a0 + q, 0);
            v1[i] = ld2s<NT>(val + a0 + q + 2, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < kSteps; ++i) {
        const int q = (i * T + threadIdx.x) * 4;
        if (q < cnt) {
            double2 p0, p1;
            p0.x = v0[i].x * x[c[i].x];
            p0.y = v0[i].y * x[c[i].y];
            p1.x = v1[i].x * x[c[i].z];
            p1.y = v1[i].y * x[c[i].w];
            *reinterpret_cast<double2 *>(prod + q) = p0;
            *reinterpret_cast<double2 *>(prod + q + 2) = p1;
        }
    }
    __syncthreads();

    // phase 2: one thread per row, CSR order
    if (r < r1) {
        double s = 0.0;
        for (int k = k0; k < k1; ++k) s += prod[k];
        if (od.rowptr)  // off-rank columns of this row (ghost values already exchanged)
            for (int k = od.rowptr[r]; k < od.rowptr[r + 1]; ++k) s += od.val[k] * od.xg[od.colidx[k]];
        if (bt_rowptr)
            for (int k = bt_rowptr[r]; k < bt_rowptr[r + 1]; ++k) s += bt_val[k] * lam[bt_colidx[k]];
        if (accumulate) s += yacc;  // y pre-loaded with B^T lambda by the fused PC kernel
        y[r] = s;
    }
}

void spmv(const CsrDev &A, const double *x, double *y, const CsrDev *bt, const double *lam,
          const int32_t *done, hipStream_t s, bool accumulate, const OffDiag *od, const GivensRider *rider)
{
    if (A.nrows == 0) {
        if (rider) givens_rider_alone(*rider, done, s);
        return;
    }
    const int tpx = (A.ntiles + 7) / 8;
    const OffDiag o = od ? *od : OffDiag{nullptr, nullptr, nullptr, nullptr};
    const GivensRider gr = rider ? *rider : no_rider();
    if (rider)
        SPK_LAUNCH_PRODUCT((spmv_stream_kernel<false, kCsrTile, kThreads, true>), dim3(tpx * 8 + 1), dim3(kThreads), 0, s,
                           A.rowptr.p, A.colidx.p, A.val.p, A.tile_row.p, A.ntiles, tpx, x, y, bt ? bt->rowptr.p : nullptr,
                           bt ? bt->colidx.p : nullptr, bt ? bt->val.p : nullptr, lam, accumulate ? 1 : 0, o, done, gr);
    else
        SPK_LAUNCH_PRODUCT((spmv_stream_kernel<false, kCsrTile, kThreads, false>), dim3(tpx * 8), dim3(kThreads), 0, s,
                           A.rowptr.p, A.colidx.p, A.val.p, A.tile_row.p, A.ntiles, tpx, x, y, bt ? bt->rowptr.p : nullptr,
                           bt ? bt->colidx.p : nullptr, bt ? bt->val.p : nullptr, lam, accumulate ? 1 : 0, o, done, gr);
}

// ---------------------------------------------------------------------------
// 2x2-blocked stream SpMV: same structure as spmv_stream_kernel, one block (4 values,
// one block column) per thread-step, x gathered 16 bytes at a time.  The products of a
// block land in LDS as (a00 x0, a01 x1, a10 x0, a11 x1); row 2k adds its pairs in block
// order = CSR order, so the result is bit-identical to the CSR kernel and the oracle.
// ---------------------------------------------------------------------------
// blocks per tile (kBTile = 512 in spk_internal.hpp: 2048 stored non-zeros); measured 256: 63.6 us, 512: 61.3 us, 1024: 76.2 us

void build_btiles(const int32_t *browptr, int32_t nbrows, std::vector<int32_t> &tile_brow)
{
    tile_brow.clear();
    tile_brow.push_back(0);
    int32_t r = 0;
    while (r < nbrows) {
        const int32_t r0 = r;
        while (r < nbrows && (r - r0) < 128 && (browptr[r + 1] - browptr[r0]) <= kBTile) ++r;
        if (r == r0) ++r;  // block row longer than a tile: handled by the strided path
        tile_brow.push_back(r);
    }
}

// ACC: y += A x (the fused Schur path pre-loads y with B^T lambda); a separate instantiation so
// that profiles list the plain product (the one bench.py times for the roofline) on its own line
// BT: the launch carries B^T lambda rows (MatMult on the nest operator); a template flag because their early fetch costs
// registers the product without them must not pay (32 against 56 allocated VGPRs: 8 % of its time, section 5 of DESIGN.md)
template <bool NT, bool ACC, bool RIDE, bool BT>
__global__ __launch_bounds__(kThreads) void spmv_bcsr_kernel(
    const int32_t *__restrict__ browptr, const int32_t *__restrict__ bcol,
    const double *__restrict__ vtop, const double *__restrict__ vbot,
    const int32_t *__restrict__ tile_brow, int ntiles, int tiles_per_xcd,
    const double *__restrict__ x, double *__restrict__ y, const int32_t *__restrict__ bt_rowptr,
    const int32_t *__restrict__ bt_colidx, const double *__restrict__ bt_val,
    const double *__restrict__ lam, OffDiag od, const int32_t *__restrict__ done, GivensRider gr)
{
    if (done && *done) return;
    __shared__ double prod[kBTile * 4];
    if (RIDE && blockIdx.x == 0) {  // the rider: a pending Givens step beside the tiles (its LDS: the product buffer)
        givens_rider(gr, prod);
        return;
    }
    const int bx = (int)blockIdx.x - (RIDE ? 1 : 0);
    const int t = (bx & 7) * tiles_per_xcd + (bx >> 3);
    if (t >= ntiles) return;
    const int br0 = tile_brow[t], br1 = tile_brow[t + 1];
    const int b0 = browptr[br0], b1 = browptr[br1];
    const int cnt = b1 - b0;

    if (cnt > kBTile) {
        // one very long block row: strided partial sums, tree order
        double a0 = 0.0, a1 = 0.0;
        for (int q = b0 + threadIdx.x; q < b1; q += kThreads) {
            const double2 xv = reinterpret_cast<const double2 *>(x)[bcol[q]];
            const double2 tp = reinterpret_cast<const double2 *>(vtop)[q], bo = reinterpret_cast<const double2 *>(vbot)[q];
            a0 += tp.x * xv.x + tp.y * xv.y;
            a1 += bo.x * xv.x + bo.y * xv.y;
        }
        __shared__ double red[8];
        const double s0 = wave_sum(a0), s1 = wave_sum(a1);
        if ((threadIdx.x & 63) == 0) {
            red[threadIdx.x >> 6] = s0;
            red[4 + (threadIdx.x >> 6)] = s1;
        }
        __syncthreads();
        if (threadIdx.x < 2) {
            const int r = 2 * br0 + threadIdx.x;
            double o = ((red[4 * threadIdx.x] + red[4 * threadIdx.x + 1]) + red[4 * threadIdx.x + 2]) + red[4 * threadIdx.x + 3];
            if (od.rowptr)
                for (int k = od.rowptr[r]; k < od.rowptr[r + 1]; ++k) o += od.val[k] * od.xg[od.colidx[k]];
            if (bt_rowptr)
                for (int k = bt_rowptr[r]; k < bt_rowptr[r + 1]; ++k) o += bt_val[k] * lam[bt_colidx[k]];
            if (ACC) o += y[r];
            y[r] = o;
        }
        return;
    }

    // what the row phase needs from memory (its block range, the value y is accumulated onto) is requested FIRST: at
    // the end of the kernel these were dependent loads with nothing left to hide them (y += A x cost 5 us more than
    // y = A x at 1024^2 for 16.8 MB, three times what the bytes take)
    const int lr = threadIdx.x;  // local row
    const bool rowok = lr < 2 * (br1 - br0);
    int k0 = 0, k1 = 0;
    double yacc = 0.0;
    if (rowok) {
        const int br = br0 + (lr >> 1);
        k0 = browptr[br] - b0;
        k1 = browptr[br + 1] - b0;
        if (ACC) yacc = y[2 * br0 + lr];
    }
    constexpr int kSteps = kBTile / kThreads;
    int c[kSteps];
    double2 tp[kSteps], bo[kSteps];
#pragma unroll
    for (int i = 0; i < kSteps; ++i) {
        const int q = i * kThreads + threadIdx.x;
        if (q < cnt) {
            c[i] = NT ? __builtin_nontemporal_load(bcol + b0 + q) : bcol[b0 + q];
            tp[i] = ld2s<NT>(vtop, b0 + q);
            bo[i] = ld2s<NT>(vbot, b0 + q);
        }
    }
    // B^T lambda of this row (MatMult on the nest operator: the restart's true residual, spk_mult): its first entries
    // are fetched now, behind the matrix stream, not at the end of the kernel (1024^2: 80.2 us against 68.5 for the
    // product without them); a longer row takes the rest in the row phase
    constexpr int kBtPre = 4;
    int kb0 = 0, kb1 = 0;
    double btv[kBtPre], btl[kBtPre];
    if (BT && rowok) {
        kb0 = bt_rowptr[2 * br0 + lr];
        kb1 = bt_rowptr[2 * br0 + lr + 1];
#pragma unroll
        for (int j = 0; j < kBtPre; ++j) {
            const bool in = kb0 + j < kb1;
            btv[j] = in ? bt_val[kb0 + j] : 0.0;
            btl[j] = in ? lam[bt_colidx[kb0 + j]] : 0.0;
        }
    }
#pragma unroll
    for (int i = 0; i < kSteps; ++i) {
        const int q = i * kThreads + threadIdx.x;
        if (q < cnt) {
            const double2 xv = reinterpret_cast<const double2 *>(x)[c[i]];
            double2 p0, p1;
            p0.x = tp[i].x * xv.x;
            p0.y = tp[i].y * xv.y;
            p1.x = bo[i].x * xv.x;
            p1.y = bo[i].y * xv.y;
            *reinterpret_cast<double2 *>(prod + 4 * q) = p0;
            *reinterpret_cast<double2 *>(prod + 4 * q + 2) = p1;
        }
    }
    __syncthreads();

    if (rowok) {
        const int half = lr & 1;
        double s = 0.0;
        for (int k = k0; k < k1; ++k) {
            const double2 p = *reinterpret_cast<const double2 *>(prod + 4 * k + 2 * half);
            s += p.x;
            s += p.y;
        }
        const int r = 2 * br0 + lr;
        if (od.rowptr)  // off-rank columns of this row (ghost values already exchanged)
            for (int k = od.rowptr[r]; k < od.rowptr[r + 1]; ++k) s += od.val[k] * od.xg[od.colidx[k]];
        if (BT) {
#pragma unroll
            for (int j = 0; j < kBtPre; ++j)
                if (kb0 + j < kb1) s += btv[j] * btl[j];
            for (int k = kb0 + kBtPre; k < kb1; ++k) s += bt_val[k] * lam[bt_colidx[k]];
        }
        if (ACC) s += yacc;
        y[r] = s;
    }
}

void spmv_bcsr(const BcsrDev &A, const double *x, double *y, const CsrDev *bt, const double *lam,
               const int32_t *done, hipStream_t s, bool accumulate, const OffDiag *odp, const GivensRider *rider)
{
    if (A.nbrows == 0) {
        if (rider) givens_rider_alone(*rider, done, s);
        return;
    }
    const int tpx = (A.ntiles + 7) / 8;
    const OffDiag od = odp ? *odp : OffDiag{nullptr, nullptr, nullptr, nullptr};
    const GivensRider gr = rider ? *rider : no_rider();
    const int nride = rider ? 1 : 0;
    // non-temporal loads on the matrix planes (read once per SpMV): 70.7 -> 61.3 us in the same run
    // (the rider is a template flag: the plain product keeps its registers and its 16 KB of LDS)
#define SPK_LAUNCH_BCSR(ACC, RIDE, BTF)                                                                                         \
    SPK_LAUNCH_PRODUCT((spmv_bcsr_kernel<true, ACC, RIDE, BTF>), dim3(tpx * 8 + nride), dim3(kThreads), 0, s, A.browptr.p, A.bcol.p, \
                       A.vtop.p, A.vbot.p, A.tile_brow.p, A.ntiles, tpx, x, y, bt ? bt->rowptr.p : nullptr,                \
                       bt ? bt->colidx.p : nullptr, bt ? bt->val.p : nullptr, lam, od, done, gr)
    if (bt) {   // (MatMult on the nest operator: never the iteration's launch, no rider)
        if (rider) fail(SPK_ERR_ARG, "spmv_bcsr: B^T rows and a rider in one launch");
        if (accumulate) SPK_LAUNCH_BCSR(true, false, true);
        else SPK_LAUNCH_BCSR(false, false, true);
    } else if (accumulate) {
        if (rider) SPK_LAUNCH_BCSR(true, true, false);
        else SPK_LAUNCH_BCSR(true, false, false);
    } else {
        if (rider) SPK_LAUNCH_BCSR(false, true, false);
        else SPK_LAUNCH_BCSR(false, false, false);
    }
#undef SPK_LAUNCH_BCSR
}

// ---------------------------------------------------------------------------
// 3x3-blocked stream SpMV (dof-3 grids: BASELINE config 5's 3-D hexahedra, 81 stored entries per row).
// One block column index per NINE values (8.44 B per stored non-zero against 12 in CSR); the values sit in nine
// planes, plane k = entry (k / 3, k % 3) of every block, so consecutive lanes read consecutive doubles of a plane.
// One block per thread: its nine products land in LDS as (a00 x0, a01 x1, a02 x2, a10 x0, ...); row 3 br + r then
// adds its triples in block order = CSR order: bit-identical to the CSR kernel and the oracle.
// ---------------------------------------------------------------------------
void build_b3tiles(const int32_t *browptr, int32_t nbrows, std::vector<int32_t> &tile_brow)
{
    tile_brow.clear();
    tile_brow.push_back(0);
    int32_t r = 0;
    while (r < nbrows) {
        const int32_t r0 = r;
        while (r < nbrows && (r - r0) < kThreads / 3 && (browptr[r + 1] - browptr[r0]) <= kB3Tile) ++r;
        if (r == r0) ++r;  // block row longer than a tile: strided path
        tile_brow.push_back(r);
    }
}

template <bool ACC, bool RIDE>
__global__ __launch_bounds__(kThreads) void spmv_bcsr3_kernel(
    const int32_t *__restrict__ browptr, const int32_t *__restrict__ bcol, const double *__restrict__ v, int64_t ldp,
    const int32_t *__restrict__ tile_brow, int ntiles, int tiles_per_xcd, const double *__restrict__ x,
    double *__restrict__ y, const int32_t *__restrict__ bt_rowptr, const int32_t *__restrict__ bt_colidx,
    const double *__restrict__ bt_val, const double *__restrict__ lam, OffDiag od, const int32_t *__restrict__ done,
    GivensRider gr)
{
    if (done && *done) return;
    __shared__ double prod[kB3Tile * 9];
    if (RIDE && blockIdx.x == 0) {  // the rider: a pending Givens step beside the tiles (its LDS: the product buffer)
        givens_rider(gr, prod);
        return;
    }
    const int bx = (int)blockIdx.x - (RIDE ? 1 : 0);
    const int t = (bx & 7) * tiles_per_xcd + (bx >> 3);
    if (t >= ntiles) return;
    const int br0 = tile_brow[t], br1 = tile_brow[t + 1];
    const int b0 = browptr[br0], b1 = browptr[br1];
    const int cnt = b1 - b0;

    if (cnt > kB3Tile) {
        // one very long block row: strided partial sums, tree order
        double a[3] = {0.0, 0.0, 0.0};
        for (int q = b0 + threadIdx.x; q < b1; q += kThreads) {
            const int c = bcol[q];
            const double x0 = x[3 * (int64_t)c], x1 = x[3 * (int64_t)c + 1], x2 = x[3 * (int64_t)c + 2];
#pragma unroll
            for (int r = 0; r < 3; ++r)
                a[r] += v[(3 * r) * ldp + q] * x0 + v[(3 * r + 1) * ldp + q] * x1 + v[(3 * r + 2) * ldp + q] * x2;
        }
        __shared__ double red[12];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double sw = wave_sum(a[r]);
            if ((threadIdx.x & 63) == 0) red[4 * r + (threadIdx.x >> 6)] = sw;
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            const int r = 3 * br0 + threadIdx.x;
            double o = ((red[4 * threadIdx.x] + red[4 * threadIdx.x + 1]) + red[4 * threadIdx.x + 2]) + red[4 * threadIdx.x + 3];
            if (od.rowptr)
                for (int k = od.rowptr[r]; k < od.rowptr[r + 1]; ++k) o += od.val[k] * od.xg[od.colidx[k]];
            if (bt_rowptr)
                for (int k = bt_rowptr[r]; k < bt_rowptr[r + 1]; ++k) o += bt_val[k] * lam[bt_colidx[k]];
            if (ACC) o += y[r];
            y[r] = o;
        }
        return;
    }

    // the row phase's own loads first (see spmv_bcsr_kernel)
    const int lr = threadIdx.x;  // local row
    const bool rowok = lr < 3 * (br1 - br0);
    int k0 = 0, k1 = 0;
    double yacc = 0.0;
    if (rowok) {
        const int br = br0 + lr / 3;
        k0 = browptr[br] - b0;
        k1 = browptr[br + 1] - b0;
        if (ACC) yacc = y[3 * br0 + lr];
    }
    const int q = threadIdx.x;
    if (q < cnt) {
        const int c = __builtin_nontemporal_load(bcol + b0 + q);
        double a[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) a[k] = __builtin_nontemporal_load(v + k * ldp + b0 + q);
        const double x0 = x[3 * (int64_t)c], x1 = x[3 * (int64_t)c + 1], x2 = x[3 * (int64_t)c + 2];
        double *p = prod + 9 * q;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            p[3 * r] = a[3 * r] * x0;
            p[3 * r + 1] = a[3 * r + 1] * x1;
            p[3 * r + 2] = a[3 * r + 2] * x2;
        }
    }
    __syncthreads();

    if (rowok) {
        const int rr = lr % 3;
        double s = 0.0;
        for (int k = k0; k < k1; ++k) {
            const double *p = prod + 9 * k + 3 * rr;
            s += p[0];
            s += p[1];
            s += p[2];
        }
        const int r = 3 * br0 + lr;
        if (od.rowptr)  // off-rank columns of this row (ghost values already exchanged)
            for (int k = od.rowptr[r]; k < od.rowptr[r + 1]; ++k) s += od.val[k] * od.xg[od.colidx[k]];
        if (bt_rowptr)
            for (int k = bt_rowptr[r]; k < bt_rowptr[r + 1]; ++k) s += bt_val[k] * lam[bt_colidx[k]];
        if (ACC) s += yacc;
        y[r] = s;
    }
}

void spmv_bcsr3(const Bcsr3Dev &A, const double *x, double *y, const CsrDev *bt, const double *lam,
                const int32_t *done, hipStream_t s, bool accumulate, const OffDiag *odp, const GivensRider *rider)
{
    if (A.nbrows == 0) {
        if (rider) givens_rider_alone(*rider, done, s);
        return;
    }
    const int tpx = (A.ntiles + 7) / 8;
    const OffDiag od = odp ? *odp : OffDiag{nullptr, nullptr, nullptr, nullptr};
    const GivensRider gr = rider ? *rider : no_rider();
    const int nride = rider ? 1 : 0;
#define SPK_LAUNCH_B3(ACC, RIDE)                                                                                          \
    SPK_LAUNCH_PRODUCT((spmv_bcsr3_kernel<ACC, RIDE>), dim3(tpx * 8 + nride), dim3(kThreads), 0, s, A.browptr.p, A.bcol.p, \
                       A.v.p, A.ldp, A.tile_brow.p, A.ntiles, tpx, x, y, bt ? bt->rowptr.p : nullptr,                    \
                       bt ? bt->colidx.p : nullptr, bt ? bt->val.p : nullptr, lam, od, done, gr)
    if (accumulate) {
        if (rider) SPK_LAUNCH_B3(true, true);
        else SPK_LAUNCH_B3(true, false);
    } else {
        if (rider) SPK_LAUNCH_B3(false, true);
        else SPK_LAUNCH_B3(false, false);
    }
#undef SPK_LAUNCH_B3
}

// ---------------------------------------------------------------------------
// KSPSetOperators on the device (set-up must not dwarf the solve).  The caller's CSR slab is
// uploaded once as it is; what MatMPIAIJ does at assembly time -- the split into a diagonal block with local
// column numbers and an off-rank block -- and the 2x2 blocking run here, one thread per (block) row, entry
// order kept (the SpMV sums stay in CSR order: bitwise parity with the oracle).
// ---------------------------------------------------------------------------
// cnt[r] = entries of row r with a column outside [lo, hi); *bad = a column outside [0, ncols)
__global__ __launch_bounds__(kThreads) void csr_count_off_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                                                 int nrows, int64_t lo, int64_t hi, int64_t ncols,
                                                                 int32_t *__restrict__ cnt, int32_t *__restrict__ bad)
{
    const int r = blockIdx.x * kThreads + threadIdx.x;
    if (r >= nrows) return;
    int32_t no = 0;
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
        const int64_t c = colidx[k];
        if (c < 0 || c >= ncols) {
            bad[0] = 1;
            bad[1] = (int32_t)c;
        }
        no += (c < lo || c >= hi);
    }
    cnt[r] = no;
}
void csr_count_off(const int32_t *rowptr, const int32_t *colidx, int nrows, int64_t lo, int64_t hi, int64_t ncols, int32_t *cnt,
                   int32_t *bad, hipStream_t s)
{
    if (nrows == 0) return;
    hipLaunchKernelGGL(csr_count_off_kernel, dim3((nrows + kThreads - 1) / kThreads), dim3(kThreads), 0, s, rowptr, colidx, nrows,
                       lo, hi, ncols, cnt, bad);
}

// exclusive prefix sum of n int32 counts into out[0..n] (out[n] = total), three small kernels
constexpr int kScanItems = 8;
__global__ __launch_bounds__(kThreads) void scan_block_kernel(const int32_t *__restrict__ in, int64_t n, int32_t *__restrict__ out,
                                                              int32_t *__restrict__ block_sum)
{
    __shared__ int32_t lds[kThreads];
    const int64_t base = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * kScanItems;
    int32_t v[kScanItems], tot = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        v[i] = base + i < n ? in[base + i] : 0;
        tot += v[i];
    }
    lds[threadIdx.x] = tot;
    __syncthreads();
    for (int off = 1; off < kThreads; off <<= 1) {  // Hillis-Steele over the thread totals
        const int32_t add = (int)threadIdx.x >= off ? lds[threadIdx.x - off] : 0;
        __syncthreads();
        lds[threadIdx.x] += add;
        __syncthreads();
    }
    int32_t run = lds[threadIdx.x] - tot;  // exclusive prefix of this thread inside the block
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
    if (threadIdx.x == kThreads - 1) block_sum[blockIdx.x] = lds[threadIdx.x];
}
__global__ __launch_bounds__(kThreads) void scan_sums_kernel(int32_t *__restrict__ block_sum, int nblocks, int32_t *__restrict__ total)
{
    __shared__ int32_t lds[kThreads];
    __shared__ int32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblocks; b0 += kThreads) {
        const int i = b0 + threadIdx.x;
        const int32_t v = i < nblocks ? block_sum[i] : 0;
        lds[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < kThreads; off <<= 1) {
            const int32_t add = (int)threadIdx.x >= off ? lds[threadIdx.x - off] : 0;
            __syncthreads();
            lds[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < nblocks) block_sum[i] = carry + lds[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == kThreads - 1) carry += lds[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ __launch_bounds__(kThreads) void scan_add_kernel(int32_t *__restrict__ out, int64_t n, const int32_t *__restrict__ block_sum,
                                                            const int32_t *__restrict__ total)
{
    const int64_t base = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * kScanItems;
    const int32_t add = block_sum[blockIdx.x];
#pragma unroll
    for (int i = 0; i < kScanItems; ++i)
        if (base + i < n) out[base + i] += add;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = *total;
}
void exclusive_scan_i32(const int32_t *in, int64_t n, int32_t *out, int32_t *scratch, hipStream_t s)
{
    // scratch: ceil(n / 2048) + 1 ints
    const int nb = (int)((n + (int64_t)kThreads * kScanItems - 1) / ((int64_t)kThreads * kScanItems));
    if (nb == 0) {
        (void)hipMemsetAsync(out, 0, sizeof(int32_t), s);
        return;
    }
    hipLaunchKernelGGL(scan_block_kernel, dim3(nb), dim3(kThreads), 0, s, in, n, out, scratch);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kThreads), 0, s, scratch, nb, scratch + nb);
    hipLaunchKernelGGL(scan_add_kernel, dim3(nb), dim3(kThreads), 0, s, out, n, scratch, scratch + nb);
}

// the split itself: row r's diagonal entries (column - lo) to d_* at rowptr[r] - orp[r], its off-rank entries
// (GLOBAL column, renumbered by the host afterwards) to o_* at orp[r]; order inside a row kept
__global__ __launch_bounds__(kThreads) void csr_split_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                                             const double *__restrict__ val, int nrows, int64_t lo, int64_t hi,
                                                             const int32_t *__restrict__ orp, int32_t *__restrict__ d_rowptr,
                                                             int32_t *__restrict__ d_col, double *__restrict__ d_val,
                                                             int32_t *__restrict__ o_col, double *__restrict__ o_val)
{
    const int r = blockIdx.x * kThreads + threadIdx.x;
    if (r > nrows) return;
    if (r == nrows) {
        d_rowptr[r] = rowptr[r] - orp[r];
        return;
    }
    int kd = rowptr[r] - orp[r], ko = orp[r];
    d_rowptr[r] = kd;
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
        const int64_t c = colidx[k];
        if (c >= lo && c < hi) {
            d_col[kd] = (int32_t)(c - lo);
            d_val[kd++] = val[k];
        } else {
            o_col[ko] = (int32_t)c;
            o_val[ko++] = val[k];
        }
    }
}
void csr_split(const int32_t *rowptr, const int32_t *colidx, const double *val, int nrows, int64_t lo, int64_t hi,
               const int32_t *orp, int32_t *d_rowptr, int32_t *d_col, double *d_val, int32_t *o_col, double *o_val, hipStream_t s)
{
    hipLaunchKernelGGL(csr_split_kernel, dim3((nrows + 1 + kThreads - 1) / kThreads), dim3(kThreads), 0, s, rowptr, colidx, val, nrows,
                       lo, hi, orp, d_rowptr, d_col, d_val, o_col, o_val);
}

// 2x2 blocking: block row br = rows 2 br, 2 br + 1, which must share their column pattern with the columns in
// pairs (2c, 2c+1); then block q of the row starts at rowptr[2 br] / 4.  *fail is raised when the structure does
// not hold anywhere (the CSR stream kernel is used then).
__global__ __launch_bounds__(kThreads) void bcsr_fill_kernel(const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                             const double *__restrict__ va, int nbr, int32_t *__restrict__ browptr,
                                                             int32_t *__restrict__ bcol, double *__restrict__ vtop,
                                                             double *__restrict__ vbot, int32_t *__restrict__ fail)
{
    const int br = blockIdx.x * kThreads + threadIdx.x;
    if (br > nbr) return;
    if (br == nbr) {
        browptr[br] = rp[2 * nbr] / 4;
        return;
    }
    const int r = 2 * br;
    const int k0 = rp[r], k1 = rp[r + 1], l0 = k1, l1 = rp[r + 2];
    if ((k1 - k0) != (l1 - l0) || ((k1 - k0) & 1) || (k0 & 3)) {
        *fail = 1;
        return;
    }
    browptr[br] = k0 / 4;
    int64_t q = k0 / 4;
    for (int k = 0; k < k1 - k0; k += 2, ++q) {
        const int c0 = ci[k0 + k], c1 = ci[k0 + k + 1];
        if ((c0 & 1) || c1 != c0 + 1 || ci[l0 + k] != c0 || ci[l0 + k + 1] != c1) {
            *fail = 1;
            return;
        }
        bcol[q] = c0 >> 1;
        vtop[2 * q] = va[k0 + k];
        vtop[2 * q + 1] = va[k0 + k + 1];
        vbot[2 * q] = va[l0 + k];
        vbot[2 * q + 1] = va[l0 + k + 1];
    }
}
void bcsr_fill(const int32_t *rp, const int32_t *ci, const double *va, int nbr, int32_t *browptr, int32_t *bcol, double *vtop,
               double *vbot, int32_t *fail, hipStream_t s)
{
    hipLaunchKernelGGL(bcsr_fill_kernel, dim3((nbr + 1 + kThreads - 1) / kThreads), dim3(kThreads), 0, s, rp, ci, va, nbr, browptr,
                       bcol, vtop, vbot, fail);
}

// Same for 3 x 3 blocks: rows 3 br .. 3 br + 2 hold the same number of entries, a multiple of three, in column triples
// (3c, 3c+1, 3c+2) that agree between the three rows; block q of the row starts at rowptr[3 br] / 9.  Values go to nine
// planes of stride ldp (plane k = entry (k / 3, k % 3)).
__global__ __launch_bounds__(kThreads) void bcsr3_fill_kernel(const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                              const double *__restrict__ va, int nbr, int32_t *__restrict__ browptr,
                                                              int32_t *__restrict__ bcol, double *__restrict__ v, int64_t ldp,
                                                              int32_t *__restrict__ fail)
{
    const int br = blockIdx.x * kThreads + threadIdx.x;
    if (br > nbr) return;
    if (br == nbr) {
        browptr[br] = rp[3 * nbr] / 9;
        return;
    }
    const int r = 3 * br;
    const int k0 = rp[r], k1 = rp[r + 1], k2 = rp[r + 2], k3 = rp[r + 3];
    const int len = k1 - k0;
    if ((k2 - k1) != len || (k3 - k2) != len || (len % 3) || (k0 % 9)) {
        *fail = 1;
        return;
    }
    browptr[br] = k0 / 9;
    int64_t q = k0 / 9;
    for (int k = 0; k < len; k += 3, ++q) {
        const int c0 = ci[k0 + k];
        bool ok = (c0 % 3) == 0;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            const int base = k0 + rr * len + k;
            ok = ok && ci[base] == c0 && ci[base + 1] == c0 + 1 && ci[base + 2] == c0 + 2;
        }
        if (!ok) {
            *fail = 1;
            return;
        }
        bcol[q] = c0 / 3;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int j = 0; j < 3; ++j) v[(3 * rr + j) * ldp + q] = va[k0 + rr * len + k + j];
    }
}
void bcsr3_fill(const int32_t *rp, const int32_t *ci, const double *va, int nbr, int32_t *browptr, int32_t *bcol, double *v,
                int64_t ldp, int32_t *fail, hipStream_t s)
{
    hipLaunchKernelGGL(bcsr3_fill_kernel, dim3((nbr + 1 + kThreads - 1) / kThreads), dim3(kThreads), 0, s, rp, ci, va, nbr, browptr,
                       bcol, v, ldp, fail);
}

// ---------------------------------------------------------------------------
// FP32 inner solve: damped-Jacobi Richardson sweeps y <- y + omega D^-1 (x - A y) on the
// diagonal block, single precision throughout (BASELINE config 5).  The sweep reuses the
// CSR stream structure (tiles, int32 columns) with a float copy of the values: 8 B per stored
// non-zero.  Products are rounded once and summed in CSR order, the update is written with
// FMA contraction switched off -> bit-identical to the oracle's float loop.  (HIP's __fmul_rn /
// __fadd_rn helpers are inlined header functions that carry their own contract flag and DO fuse.)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void cvt_scale_f32_kernel(const double *__restrict__ x,
                                                                 const float *__restrict__ d32, float omega,
                                                                 float *__restrict__ x32, float *__restrict__ y32,
                                                                 int64_t n, const int32_t *__restrict__ done)
{
#pragma clang fp contract(off)  // every product and sum below is rounded on its own (the oracle's float loop)
    if (done && *done) return;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const float xv = (float)x[i];
        x32[i] = xv;
        y32[i] = ((omega * d32[i]) * xv);
    }
}
void cvt_scale_f32(const double *x, const float *d32, float omega, float *x32, float *y32, int64_t n,
                   const int32_t *done, hipStream_t s)
{
    if (n == 0) return;
    const int grid = (int)std::min<int64_t>((n + kThreads - 1) / kThreads, kMaxBlocks * 4);
    hipLaunchKernelGGL(cvt_scale_f32_kernel, dim3(grid), dim3(kThreads), 0, s, x, d32, omega, x32, y32, n, done);
}

__global__ __launch_bounds__(kThreads) void jacobi_sweep_f32_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const float *__restrict__ val32,
    const int32_t *__restrict__ tile_row, int ntiles, int tiles_per_xcd, const float *__restrict__ d32,
    float omega, const float *__restrict__ x32, const float *__restrict__ yin, float *__restrict__ yout,
    const int32_t *__restrict__ done)
{
#pragma clang fp contract(off)  // every product and sum below is rounded on its own (the oracle's float loop)
    if (done && *done) return;
    const int t = (blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
    if (t >= ntiles) return;
    __shared__ float prod[kCsrTile + 8];
    const int r0 = tile_row[t], r1 = tile_row[t + 1];
    const int nz0 = rowptr[r0], nz1 = rowptr[r1];
    const int a0 = nz0 & ~3;
    const int cnt = nz1 - a0;
    if (cnt > kCsrTile) {  // one row longer than a tile
        float acc = 0.0f;
        for (int k = nz0 + threadIdx.x; k < nz1; k += kThreads) acc += val32[k] * yin[colidx[k]];
        __shared__ float red[kThreads];
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int st = kThreads / 2; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0)
            yout[r0] = yin[r0] + ((omega * d32[r0]) * (x32[r0] - red[0]));
        return;
    }
    constexpr int kSteps = kCsrTile / (kThreads * 4);
#pragma unroll
    for (int i = 0; i < kSteps; ++i) {
        const int q = (i * kThreads + threadIdx.x) * 4;
        if (q < cnt) {
            const int4 c = *reinterpret_cast<const int4 *>(colidx + a0 + q);
            const float4 v = *reinterpret_cast<const float4 *>(val32 + a0 + q);
            float4 p;
            p.x = (v.x * yin[c.x]);
            p.y = (v.y * yin[c.y]);
            p.z = (v.z * yin[c.z]);
            p.w = (v.w * yin[c.w]);
            *reinterpret_cast<float4 *>(prod + q) = p;
        }
    }
    __syncthreads();
    const int r = r0 + threadIdx.x;
    if (r < r1) {
        const int k0 = rowptr[r] - a0, k1 = rowptr[r + 1] - a0;
        float s = 0.0f;
        for (int k = k0; k < k1; ++k) s = (s + prod[k]);
        yout[r] = yin[r] + ((omega * d32[r]) * (x32[r] - s));
    }
}
void jacobi_sweep_f32(const CsrDev &A, const float *val32, const float *d32, float omega, const float *x32,
                      const float *yin, float *yout, const int32_t *done, hipStream_t s)
{
    if (A.nrows == 0) return;
    const int tpx = (A.ntiles + 7) / 8;
    hipLaunchKernelGGL(jacobi_sweep_f32_kernel, dim3(tpx * 8), dim3(kThreads), 0, s, A.rowptr.p, A.colidx.p, val32,
                       A.tile_row.p, A.ntiles, tpx, d32, omega, x32, yin, yout, done);
}

// The same sweep from the 2x2-blocked copy with single-precision value planes (5 B per stored non-zero against 8):
// products rounded once each, summed per row in block order = CSR order -- the same bits as the CSR sweep and the oracle.
__global__ __launch_bounds__(kThreads) void jacobi_sweep_f32_b2_kernel(
    const int32_t *__restrict__ browptr, const int32_t *__restrict__ bcol, const float *__restrict__ vtop32,
    const float *__restrict__ vbot32, const int32_t *__restrict__ tile_brow, int ntiles, int tiles_per_xcd,
    const float *__restrict__ d32, float omega, const float *__restrict__ x32, const float *__restrict__ yin,
    float *__restrict__ yout, const int32_t *__restrict__ done)
{
#pragma clang fp contract(off)  // every product and sum below is rounded on its own (the oracle's float loop)
    if (done && *done) return;
    const int t = (blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
    if (t >= ntiles) return;
    __shared__ float prod[kBTile * 4];
    const int br0 = tile_brow[t], br1 = tile_brow[t + 1];
    const int b0 = browptr[br0], b1 = browptr[br1];
    const int cnt = b1 - b0;
    if (cnt > kBTile) {  // one block row longer than a tile: its two rows by two threads, CSR order
        if (threadIdx.x < 2) {
            const int rr = threadIdx.x, r = 2 * br0 + rr;
            const float *vv = rr ? vbot32 : vtop32;
            float s = 0.0f;
            for (int q = b0; q < b1; ++q) {
                const int c = bcol[q];
                s = (s + (vv[2 * (int64_t)q] * yin[2 * (int64_t)c]));
                s = (s + (vv[2 * (int64_t)q + 1] * yin[2 * (int64_t)c + 1]));
            }
            yout[r] = yin[r] + ((omega * d32[r]) * (x32[r] - s));
        }
        return;
    }
    constexpr int kSteps = kBTile / kThreads;
#pragma unroll
    for (int i = 0; i < kSteps; ++i) {
        const int q = i * kThreads + threadIdx.x;
        if (q < cnt) {
            const int c = __builtin_nontemporal_load(bcol + b0 + q);
            const float2 tp = reinterpret_cast<const float2 *>(vtop32)[b0 + q];
            const float2 bo = reinterpret_cast<const float2 *>(vbot32)[b0 + q];
            const float2 yv = reinterpret_cast<const float2 *>(yin)[c];
            float4 p;
            p.x = (tp.x * yv.x);
            p.y = (tp.y * yv.y);
            p.z = (bo.x * yv.x);
            p.w = (bo.y * yv.y);
            *reinterpret_cast<float4 *>(prod + 4 * q) = p;
        }
    }
    __syncthreads();
    const int lr = threadIdx.x;
    if (lr < 2 * (br1 - br0)) {
        const int br = br0 + (lr >> 1), half = lr & 1;
        const int k0 = browptr[br] - b0, k1 = browptr[br + 1] - b0;
        float s = 0.0f;
        for (int k = k0; k < k1; ++k) {
            const float2 p = *reinterpret_cast<const float2 *>(prod + 4 * k + 2 * half);
            s = (s + p.x);
            s = (s + p.y);
        }
        const int r = 2 * br0 + lr;
        yout[r] = yin[r] + ((omega * d32[r]) * (x32[r] - s));
    }
}
void jacobi_sweep_f32_b2(const BcsrDev &A, const float *d32, float omega, const float *x32, const float *yin, float *yout,
                         const int32_t *done, hipStream_t s)
{
    if (A.nbrows == 0) return;
    const int tpx = (A.ntiles + 7) / 8;
    hipLaunchKernelGGL(jacobi_sweep_f32_b2_kernel, dim3(tpx * 8), dim3(kThreads), 0, s, A.browptr.p, A.bcol.p, A.vtop32.p,
                       A.vbot32.p, A.tile_brow.p, A.ntiles, tpx, d32, omega, x32, yin, yout, done);
}

// The same sweep from the 3x3-blocked copy with single-precision planes (4.44 B per stored non-zero against 8):
// products rounded once each, summed per row in block order = CSR order -- the same bits as the CSR sweep and the oracle.
__global__ __launch_bounds__(kThreads) void jacobi_sweep_f32_b3_kernel(
    const int32_t *__restrict__ browptr, const int32_t *__restrict__ bcol, const float *__restrict__ v32, int64_t ldp,
    const int32_t *__restrict__ tile_brow, int ntiles, int tiles_per_xcd, const float *__restrict__ d32, float omega,
    const float *__restrict__ x32, const float *__restrict__ yin, float *__restrict__ yout, const int32_t *__restrict__ done)
{
#pragma clang fp contract(off)  // every product and sum below is rounded on its own (the oracle's float loop)
    if (done && *done) return;
    const int t = (blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
    if (t >= ntiles) return;
    __shared__ float prod[kB3Tile * 9];
    const int br0 = tile_brow[t], br1 = tile_brow[t + 1];
    const int b0 = browptr[br0], b1 = browptr[br1];
    const int cnt = b1 - b0;
    if (cnt > kB3Tile) {  // one block row longer than a tile: its three rows by three threads, CSR order
        if (threadIdx.x < 3) {
            const int rr = threadIdx.x, r = 3 * br0 + rr;
            float s = 0.0f;
            for (int q = b0; q < b1; ++q) {
                const int c = bcol[q];
                s = (s + (v32[(3 * rr) * ldp + q] * yin[3 * (int64_t)c]));
                s = (s + (v32[(3 * rr + 1) * ldp + q] * yin[3 * (int64_t)c + 1]));
                s = (s + (v32[(3 * rr + 2) * ldp + q] * yin[3 * (int64_t)c + 2]));
            }
            yout[r] = yin[r] + ((omega * d32[r]) * (x32[r] - s));
        }
        return;
    }
    const int q = threadIdx.x;
    if (q < cnt) {
        const int c = __builtin_nontemporal_load(bcol + b0 + q);
        float a[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) a[k] = __builtin_nontemporal_load(v32 + k * ldp + b0 + q);
        const float y0 = yin[3 * (int64_t)c], y1 = yin[3 * (int64_t)c + 1], y2 = yin[3 * (int64_t)c + 2];
        float *p = prod + 9 * q;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            p[3 * r] = (a[3 * r] * y0);
            p[3 * r + 1] = (a[3 * r + 1] * y1);
            p[3 * r + 2] = (a[3 * r + 2] * y2);
        }
    }
    __syncthreads();
    const int lr = threadIdx.x;
    if (lr < 3 * (br1 - br0)) {
        const int br = br0 + lr / 3, rr = lr % 3;
        const int k0 = browptr[br] - b0, k1 = browptr[br + 1] - b0;
        float s = 0.0f;
        for (int k = k0; k < k1; ++k) {
            const float *p = prod + 9 * k + 3 * rr;
            s = (s + p[0]);
            s = (s + p[1]);
            s = (s + p[2]);
        }
        const int r = 3 * br0 + lr;
        yout[r] = yin[r] + ((omega * d32[r]) * (x32[r] - s));
    }
}
void jacobi_sweep_f32_b3(const Bcsr3Dev &A, const float *d32, float omega, const float *x32, const float *yin, float *yout,
                         const int32_t *done, hipStream_t s)
{
    if (A.nbrows == 0) return;
    const int tpx = (A.ntiles + 7) / 8;
    hipLaunchKernelGGL(jacobi_sweep_f32_b3_kernel, dim3(tpx * 8), dim3(kThreads), 0, s, A.browptr.p, A.bcol.p, A.v32.p, A.ldp,
                       A.tile_brow.p, A.ntiles, tpx, d32, omega, x32, yin, yout, done);
}

// rows with off-rank columns: y[row] -= omega d (Ao_row . ghost values of the previous iterate)
__global__ __launch_bounds__(kThreads) void sweep_offdiag_f32_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const double *__restrict__ val,
    const int32_t *__restrict__ rows, int nrows, const float *__restrict__ d32, float omega,
    const double *__restrict__ xg, float *__restrict__ y, const int32_t *__restrict__ done)
{
#pragma clang fp contract(off)  // every product and sum below is rounded on its own (the oracle's float loop)
    if (done && *done) return;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= nrows) return;
    float sacc = 0.0f;
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) sacc = sacc + ((float)val[k] * (float)xg[colidx[k]]);
    const int r = rows[i];
    y[r] = y[r] - ((omega * d32[r]) * sacc);
}
void sweep_offdiag_f32(const CsrDev &Ao, const int32_t *rows, const float *d32, float omega, const double *xg,
                       float *y, const int32_t *done, hipStream_t s)
{
    if (Ao.nrows == 0) return;
    hipLaunchKernelGGL(sweep_offdiag_f32_kernel, dim3((Ao.nrows + kThreads - 1) / kThreads), dim3(kThreads), 0, s,
                       Ao.rowptr.p, Ao.colidx.p, Ao.val.p, rows, Ao.nrows, d32, omega, xg, y, done);
}

__global__ __launch_bounds__(kThreads) void gather_f32_kernel(const float *__restrict__ x, const int32_t *__restrict__ idx,
                                                              int64_t n, double *__restrict__ out,
                                                              const int32_t *__restrict__ done)
{
    if (done && *done) return;
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i < n) out[i] = (double)x[idx[i]];
}
void gather_f32(const float *x, const int32_t *idx, int64_t n, double *out, const int32_t *done, hipStream_t s)
{
    if (n == 0) return;
    hipLaunchKernelGGL(gather_f32_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, x, idx, n, out, done);
}

__global__ __launch_bounds__(kThreads) void cvt_f32_out_kernel(const float *__restrict__ y32, double *__restrict__ y,
                                                               int mode, int64_t n, const int32_t *__restrict__ done)
{
    if (done && *done) return;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads)
        y[i] = mode == 0 ? (double)y32[i] : y[i] - (double)y32[i];
}
void cvt_f32_out(const float *y32, double *y, int mode, int64_t n, const int32_t *done, hipStream_t s)
{
    if (n == 0) return;
    const int grid = (int)std::min<int64_t>((n + kThreads - 1) / kThreads, kMaxBlocks * 4);
    hipLaunchKernelGGL(cvt_f32_out_kernel, dim3(grid), dim3(kThreads), 0, s, y32, y, mode, n, done);
}

__global__ __launch_bounds__(kThreads) void cvt_vals_f32_kernel(const double *__restrict__ v, float *__restrict__ v32, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) v32[i] = (float)v[i];
}
void cvt_vals_f32(const double *v, float *v32, int64_t n, hipStream_t s)
{
    if (n == 0) return;
    const int grid = (int)std::min<int64_t>((n + kThreads - 1) / kThreads, kMaxBlocks * 4);
    hipLaunchKernelGGL(cvt_vals_f32_kernel, dim3(grid), dim3(kThreads), 0, s, v, v32, n);
}

}  // namespace k
}  // namespace spk
