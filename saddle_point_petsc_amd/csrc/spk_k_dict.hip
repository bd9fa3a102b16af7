// spk_k_dict.hip -- MatMult on the A block from ROW TYPES + DEVIATION CODES (DictDev, spk_internal.hpp), the FP32 Richardson
// sweep on the same layout, and the set-up passes that find the layout.  gfx950, wave64.
//
// Why: MatMult_SeqAIJ streams 12 B per stored non-zero (494 MB per product at 1024^2), the 2x2-blocked copy 9 B.  The
// reference assembles the SAME element matrix for every element of a uniform grid
// (/root/reference/src/Discretization.c:25, :293-332) -- up to rounding: the Jacobian is formed from node coordinates
// i*h (:96-128), so the entries of A scatter by a few hundred ulps around a handful of ideal values (measured at 1024^2:
// 34 185 distinct doubles, 7 ideal values; > 58 000 distinct 2x2 blocks at 256^2 already -- a dictionary of exact blocks
// does not exist).  What repeats is the block up to that noise (a few dozen CLASSES) and the sequence of (column offset,
// class) along a block row (a few dozen ROW TYPES).  Stored: one 16-bit type per block row, the small tables (in LDS),
// and per stored value an integer k with   value = base[class][entry] + k * 2^g[class][entry]   EXACTLY, as a
// two's-complement bit field exactly as wide as its class entry needs (5 .. 19 bits at 1024^2), the fields of a block
// packed into one 64-bit word (2x2) / two (3x3): 1.8 B per stored non-zero instead of 9 / 12.  Decoding is a bit-field
// extract, one integer conversion and one FMA whose result is exact, so the products and their order -- hence every bit
// of the sums -- are those of the CSR loop.
// Nothing is assumed about the grid: classes and types are FOUND in the caller's CSR by hashing, the codes are verified
// by decoding every value and comparing its bits, and a matrix that does not fit (too many classes / types, deviations
// that are not small multiples of one power of two) keeps the blocked / CSR kernels.
//
// Parity: a row type lists its blocks in the block row's CSR order; every product is rounded on its own (contraction off)
// and added in that order, starting from +0: bit-identical to spmv_bcsr_kernel, the CSR kernel and the oracle
// (tests/test_gpu_parity.py::test_spmv_A_block_bitwise, tests/test_gpu_dict.py).
#include "spk_dict.hpp"

#include <climits>

namespace spk {
namespace k {

// U3 (3x3 blocks only): 0 = per-class fields, 1..3 = one field layout for all classes (dict_field3u)
template <int BS, bool ACC, bool RIDE, bool BT, int U3>
__global__ __launch_bounds__(kThreads) void spmv_dict_kernel(DictArgs d, const double *__restrict__ x, double *__restrict__ y,
                                                             const int32_t *__restrict__ bt_rowptr,
                                                             const int32_t *__restrict__ bt_colidx,
                                                             const double *__restrict__ bt_val, const double *__restrict__ lam,
                                                             OffDiag od, const int32_t *__restrict__ done, GivensRider gr)
{
#pragma clang fp contract(off)  // every product rounded on its own, as in the CSR loop the sums are compared with
    if (done && *done) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (RIDE && blockIdx.x == 0) {  // the rider: a pending Givens step beside the row chunks (its LDS: the table space)
        givens_rider(gr, reinterpret_cast<double *>(smem));
        return;
    }
    const int bx = (int)blockIdx.x - (RIDE ? 1 : 0);
    // workgroups b, b+8, ... share an XCD: each XCD gets a contiguous run of chunks, so the x window stays in ITS L2
    const int c0 = ((bx & 7) * d.chunks_per_xcd + (bx >> 3) * d.chunks_per_wg);
    const int c1 = min(min(c0 + d.chunks_per_wg, ((bx & 7) + 1) * d.chunks_per_xcd), d.nchunks);
    if (c0 >= c1) return;
    // what does not depend on the tables is requested before they are copied
    int br = c0 * kDictChunk + (int)threadIdx.x;
    int tidn = br < d.nbrows ? (int)d.tid[br] : -1;
    dict_load_lds(d, (d.nclass + 1) * BS * BS, smem);
    const int32_t *tlen = reinterpret_cast<const int32_t *>(smem);
    const int2 *tent = reinterpret_cast<const int2 *>(smem + 4 * ((d.ntype + 1) & ~1));
    const double2 *cv = reinterpret_cast<const double2 *>(smem + d.cls_off);
    const int32_t *fl = reinterpret_cast<const int32_t *>(smem + d.fld_off);

    for (int ch = c0; ch < c1; ++ch) {
        const int tc = tidn;
        const int brc = br;
        br += kDictChunk;
        tidn = (ch + 1 < c1 && br < d.nbrows) ? (int)d.tid[br] : -1;   // next chunk's row type: in flight during this one
        if (tc < 0) continue;
        double s[BS];
#pragma unroll
        for (int r = 0; r < BS; ++r) s[r] = 0.0;
        double yacc[BS];
        if (ACC) {
            if (BS == 2) {
                const double2 t = reinterpret_cast<const double2 *>(y)[brc];
                yacc[0] = t.x;
                yacc[1] = t.y;
            } else {
#pragma unroll
                for (int r = 0; r < BS; ++r) yacc[r] = y[(int64_t)BS * brc + r];
            }
        }
        // B^T lambda of these rows (MatMult on the nest operator): first entries requested early (see spmv_bcsr_kernel)
        constexpr int kBtPre = 4;
        int kb0[BS], kb1[BS];
        double btv[BS][kBtPre], btl[BS][kBtPre];
        if (BT) {
#pragma unroll
            for (int r = 0; r < BS; ++r) {
                kb0[r] = bt_rowptr[(int64_t)BS * brc + r];
                kb1[r] = bt_rowptr[(int64_t)BS * brc + r + 1];
#pragma unroll
                for (int j = 0; j < kBtPre; ++j) {
                    const bool in = kb0[r] + j < kb1[r];
                    btv[r][j] = in ? bt_val[kb0[r] + j] : 0.0;
                    btl[r][j] = in ? lam[bt_colidx[kb0[r] + j]] : 0.0;
                }
            }
        }
        const int len = tlen[tc];
        const int2 *te = tent + (size_t)tc * d.kmax;
        // blocks whose loads are in flight together: a 2-D interior row whole (five 16-byte code loads + nine gathers of x);
        // 3x3: nine (a third of an interior row; measured on the 256 x 256 x 32 slab: 264 us with three, 244-259 with five,
        // 224-229 with nine)
        constexpr int G = BS == 2 ? 10 : kDictG3;
        for (int k0 = 0; k0 < len; k0 += G) {
            int2 e[G];
            u64 w0[G], w1[BS == 2 ? 1 : G];
            double xv[G][BS];
            if (BS == 2) {
#pragma unroll
                for (int h = 0; h < G / 2; ++h) {
                    w0[2 * h] = w0[2 * h + 1] = 0ull;
                    if (k0 + 2 * h < len) dict_issue_pair2(d, (k0 >> 1) + h, brc, w0[2 * h], w0[2 * h + 1]);
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const bool in = k0 + g < len;
                e[g] = in ? te[k0 + g] : make_int2(0, 0);
                if (in) {
                    if (BS == 3) dict_issue3(d, k0 + g, brc, w0[g], w1[BS == 2 ? 0 : g]);
                    const int64_t c = (int64_t)brc + e[g].x;
                    if (BS == 2) {
                        const double2 t = reinterpret_cast<const double2 *>(x)[c];
                        xv[g][0] = t.x;
                        xv[g][1] = t.y;
                    } else {
#pragma unroll
                        for (int j = 0; j < BS; ++j) xv[g][j] = x[BS * c + j];
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (k0 + g < len) {
                    const double2 *cb = cv + (size_t)e[g].y * (BS * BS);
                    const int32_t *fb = fl + (size_t)e[g].y * (BS * BS);
#pragma unroll
                    for (int r = 0; r < BS; ++r)
#pragma unroll
                        for (int j = 0; j < BS; ++j)
                            s[r] += dict_decode(BS == 2 ? dict_field2(w0[g], fb[r * BS + j], d.strad != 0)
                                                       : U3 ? dict_field3u<(U3 ? U3 : 1)>(w0[g], w1[BS == 2 ? 0 : g], d, r * BS + j)
                                                            : dict_field(w0[g], w1[BS == 2 ? 0 : g], fb[r * BS + j]),
                                                cb[r * BS + j]) * xv[g][j];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < BS; ++r) {
            const int64_t row = (int64_t)BS * brc + r;
            // (off-rank and B^T terms: fused multiply-adds, as the compiler contracts them in the blocked and CSR kernels --
            // every layout gives the same bits on these rows too)
            if (od.rowptr)  // off-rank columns of this row (ghost values already exchanged)
                for (int k = od.rowptr[row]; k < od.rowptr[row + 1]; ++k) s[r] = __builtin_fma(od.val[k], od.xg[od.colidx[k]], s[r]);
            if (BT) {
#pragma unroll
                for (int j = 0; j < kBtPre; ++j)
                    if (kb0[r] + j < kb1[r]) s[r] = __builtin_fma(btv[r][j], btl[r][j], s[r]);
                for (int k = kb0[r] + kBtPre; k < kb1[r]; ++k) s[r] = __builtin_fma(bt_val[k], lam[bt_colidx[k]], s[r]);
            }
            if (ACC) s[r] += yacc[r];
        }
        if (BS == 2) {
            double2 o;
            o.x = s[0];
            o.y = s[1];
            reinterpret_cast<double2 *>(y)[brc] = o;
        } else {
#pragma unroll
            for (int r = 0; r < BS; ++r) y[(int64_t)BS * brc + r] = s[r];
        }
    }
}

// ---------------------------------------------------------------------------
// 2x2 blocks, row types of at most KM = 9 blocks (the 9-point stencil of the reference's Q1 grid, Discretization.c:25):
// the same product, software-pipelined over the chunks of a workgroup.  The kernel above issues a chunk's loads, waits,
// computes, and only then turns to the next chunk: with the four waves a SIMD holds, HBM idles while they decode and the
// ALUs idle while they wait (PMC at 1024^2: SQ_WAIT_ANY 59 % of the wave cycles, ~10 us each of VALU and LDS work per CU
// inside 33 us).  Here the loads of chunk c+1 -- code words, the gathers of x, y -- are in flight while chunk c is
// decoded: two register stages, and NO branch around a load, so that the compiler's wait counts stay exact (a conditional
// load makes every later wait a wait for everything).  A position beyond a row's length reads the zero pad with the NULL
// class, whose decoded value is +0: the sums are unchanged, bit for bit; a chunk beyond the workgroup's range re-reads
// the last block row and is not computed.
// ---------------------------------------------------------------------------
template <int KM, bool ACC>
struct Dict2Stage {
    u64 w[KM + 1];
    double2 xv[KM];
    int cls[KM];
    double2 yv;
};

// (xr: x as a raw buffer of 16 * nbrows bytes -- a position beyond a row's length carries the offset 2^31 in the LDS copy
// of its row type and the null class: the range check of the buffer load returns zeros, no select, no branch)
template <int KM, bool ACC>
__device__ __forceinline__ void dict2_issue(const DictArgs &d, __amdgpu_buffer_rsrc_t xr, const double *__restrict__ y,
                                            const int2 *tent, int tc, int brr, Dict2Stage<KM, ACC> &S)
{
    static_assert(KM % 2 == 1, "an odd last position: its plane holds single words");
#pragma unroll
    for (int h = 0; h < KM / 2; ++h) {
        const int4v r = __builtin_nontemporal_load(reinterpret_cast<const int4v *>(d.codes + d.plane_off[h]) + brr);
        S.w[2 * h] = (u64)(uint32_t)r.x | ((u64)(uint32_t)r.y << 32);
        S.w[2 * h + 1] = (u64)(uint32_t)r.z | ((u64)(uint32_t)r.w << 32);
    }
    {
        const int2v r = __builtin_nontemporal_load(reinterpret_cast<const int2v *>(d.codes + d.plane_off[KM / 2]) + brr);
        S.w[KM - 1] = (u64)(uint32_t)r.x | ((u64)(uint32_t)r.y << 32);
    }
    if (ACC) S.yv = reinterpret_cast<const double2 *>(y)[brr];
    const int2 *te = tent + (size_t)tc * KM;
    const uint32_t b16 = (uint32_t)brr * 16u;
#pragma unroll
    for (int g = 0; g < KM; ++g) {
        const int2 e = te[g];
        S.cls[g] = e.y;
        const int4v r = __builtin_bit_cast(int4v, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(b16 + (uint32_t)e.x), 0, 0));
        S.xv[g].x = __hiloint2double(r.y, r.x);
        S.xv[g].y = __hiloint2double(r.w, r.z);
    }
}

template <bool ACC, bool RIDE, bool BT, int KM, bool UNI>
__global__ __launch_bounds__(kThreads) void spmv_dict2_kernel(DictArgs d, const double *__restrict__ x, double *__restrict__ y,
                                                              const int32_t *__restrict__ bt_rowptr,
                                                              const int32_t *__restrict__ bt_colidx,
                                                              const double *__restrict__ bt_val, const double *__restrict__ lam,
                                                              OffDiag od, const int32_t *__restrict__ done, GivensRider gr)
{
#pragma clang fp contract(off)
    if (done && *done) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (RIDE && blockIdx.x == 0) {
        givens_rider(gr, reinterpret_cast<double *>(smem));
        return;
    }
    const int bx = (int)blockIdx.x - (RIDE ? 1 : 0);
    const int c0 = ((bx & 7) * d.chunks_per_xcd + (bx >> 3) * d.chunks_per_wg);
    const int c1 = min(min(c0 + d.chunks_per_wg, ((bx & 7) + 1) * d.chunks_per_xcd), d.nchunks);
    if (c0 >= c1) return;
    const int last = d.nbrows - 1;
    auto rowof = [&](int ch) { return min(ch * kDictChunk + (int)threadIdx.x, last); };
    // row types of the first two chunks: requested before the tables are copied
    int tA = (int)d.tid[rowof(c0)];
    int tB = (int)d.tid[rowof(c0 + 1)];
    dict_load_lds(d, (d.nclass + 1) * 4, smem);
    const int32_t *tlen = reinterpret_cast<const int32_t *>(smem);
    int2 *tent = reinterpret_cast<int2 *>(smem + 4 * ((d.ntype + 1) & ~1));
    const double2 *cv = reinterpret_cast<const double2 *>(smem + d.cls_off);
    const int32_t *fl = reinterpret_cast<const int32_t *>(smem + d.fld_off);
    // the LDS copy of the row types as the issue stage wants it: byte offsets of x; beyond a row's length an offset
    // outside the buffer and the null class
    for (int i = threadIdx.x; i < d.ntype * KM; i += kThreads) {
        const int t = i / KM, k = i - t * KM;
        int2 e = tent[i];
        if (k < tlen[t]) e.x *= 16;
        else e = make_int2((int)0x80000000u, d.nclass);
        tent[i] = e;
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(x), 0, 16 * d.nbrows, 0x00020000);

    Dict2Stage<KM, ACC> SA, SB;
    auto compute = [&](int ch, const Dict2Stage<KM, ACC> &S) {
        const int brc = ch * kDictChunk + (int)threadIdx.x;
        double s[2] = {0.0, 0.0};
#pragma unroll
        for (int g = 0; g < KM; ++g) {
            const double2 *cb = cv + (size_t)S.cls[g] * 4;
            if (UNI) {
                // one field layout for every class (DictArgs::uw): entries 0 / 2 at the bottom of the low / high half,
                // 1 / 3 at the top -- one instruction each, no table read
                const int lo = (int)(uint32_t)S.w[g], hi = (int)(uint32_t)(S.w[g] >> 32);
                s[0] += dict_decode(__builtin_amdgcn_sbfe(lo, 0u, (unsigned)d.uw[0]), cb[0]) * S.xv[g].x;
                s[0] += dict_decode(lo >> (32 - d.uw[1]), cb[1]) * S.xv[g].y;
                s[1] += dict_decode(__builtin_amdgcn_sbfe(hi, 0u, (unsigned)d.uw[2]), cb[2]) * S.xv[g].x;
                s[1] += dict_decode(hi >> (32 - d.uw[3]), cb[3]) * S.xv[g].y;
            } else {
                const int32_t *fb = fl + (size_t)S.cls[g] * 4;
                s[0] += dict_decode(dict_field2(S.w[g], fb[0]), cb[0]) * S.xv[g].x;
                s[0] += dict_decode(dict_field2(S.w[g], fb[1]), cb[1]) * S.xv[g].y;
                s[1] += dict_decode(dict_field2(S.w[g], fb[2]), cb[2]) * S.xv[g].x;
                s[1] += dict_decode(dict_field2(S.w[g], fb[3]), cb[3]) * S.xv[g].y;
            }
        }
        if (brc > last) return;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int64_t row = 2 * (int64_t)brc + r;
            if (od.rowptr)
                for (int k = od.rowptr[row]; k < od.rowptr[row + 1]; ++k) s[r] = __builtin_fma(od.val[k], od.xg[od.colidx[k]], s[r]);
            if (BT)
                for (int k = bt_rowptr[row]; k < bt_rowptr[row + 1]; ++k) s[r] = __builtin_fma(bt_val[k], lam[bt_colidx[k]], s[r]);
            if (ACC) s[r] += r == 0 ? S.yv.x : S.yv.y;
        }
        double2 o;
        o.x = s[0];
        o.y = s[1];
        reinterpret_cast<double2 *>(y)[brc] = o;
    };

    dict2_issue<KM, ACC>(d, xr, y, tent, tA, rowof(c0), SA);
    tA = (int)d.tid[rowof(c0 + 2)];
    for (int ch = c0;; ch += 2) {
        // loads of the next chunk (and the row type of the one after it), then this chunk's arithmetic
        dict2_issue<KM, ACC>(d, xr, y, tent, tB, rowof(ch + 1), SB);
        tB = (int)d.tid[rowof(ch + 3)];
        compute(ch, SA);
        if (ch + 1 >= c1) break;
        dict2_issue<KM, ACC>(d, xr, y, tent, tA, rowof(ch + 2), SA);
        tA = (int)d.tid[rowof(ch + 4)];
        compute(ch + 1, SB);
        if (ch + 2 >= c1) break;
    }
}

void spmv_dict(const DictDev &A, const double *x, double *y, const CsrDev *bt, const double *lam, const int32_t *done,
               hipStream_t s, bool accumulate, const OffDiag *odp, const GivensRider *rider)
{
    if (A.nbrows == 0) {
        if (rider) givens_rider_alone(*rider, done, s);
        return;
    }
    if (spmv_dict3(A, x, y, bt, lam, done, s, accumulate, odp, rider)) return;   // 27-point row types, one field layout: pipelined
    int grid = 0;
    const DictArgs d = dict_args(A, &grid);
    const OffDiag od = odp ? *odp : OffDiag{nullptr, nullptr, nullptr, nullptr};
    const GivensRider gr = rider ? *rider : no_rider();
    const int nride = rider ? 1 : 0;
    // LDS: the tables; a launch with a rider needs the rider's scratch in block 0
    size_t lds = (size_t)A.lds_bytes;
    if (rider) lds = std::max(lds, sizeof(double) * (size_t)(kThreads + 4 * (kMaxNv + 2) + 4));
#define SPK_LAUNCH_DICT_U(BS, ACC, RIDE, BTF, U3)                                                                               \
    SPK_LAUNCH_PRODUCT((spmv_dict_kernel<BS, ACC, RIDE, BTF, U3>), dim3(grid + nride), dim3(kThreads), lds, s, d, x, y,        \
                       bt ? bt->rowptr.p : nullptr, bt ? bt->colidx.p : nullptr, bt ? bt->val.p : nullptr, lam, od, done, gr)
#define SPK_LAUNCH_DICT(BS, ACC, RIDE, BTF)                                                                                     \
    do {                                                                                                                        \
        if (BS == 2 || A.uniform3 == 0) SPK_LAUNCH_DICT_U(BS, ACC, RIDE, BTF, 0);                                               \
        else if (A.uniform3 == 1) SPK_LAUNCH_DICT_U(3, ACC, RIDE, BTF, 1);                                                      \
        else if (A.uniform3 == 2) SPK_LAUNCH_DICT_U(3, ACC, RIDE, BTF, 2);                                                      \
        else SPK_LAUNCH_DICT_U(3, ACC, RIDE, BTF, 3);                                                                           \
    } while (0)
#define SPK_DISPATCH_DICT(BS)                                                                                                   \
    if (bt) {                                                                                                                   \
        if (rider) fail(SPK_ERR_ARG, "spmv_dict: B^T rows and a rider in one launch");                                         \
        if (accumulate) SPK_LAUNCH_DICT(BS, true, false, true);                                                                 \
        else SPK_LAUNCH_DICT(BS, false, false, true);                                                                           \
    } else if (accumulate) {                                                                                                    \
        if (rider) SPK_LAUNCH_DICT(BS, true, true, false);                                                                      \
        else SPK_LAUNCH_DICT(BS, true, false, false);                                                                           \
    } else {                                                                                                                    \
        if (rider) SPK_LAUNCH_DICT(BS, false, true, false);                                                                     \
        else SPK_LAUNCH_DICT(BS, false, false, false);                                                                          \
    }
    if (A.bs == 2 && A.kmax == 9 && A.nbrows < (1 << 27) && !A.straddle) {
        // row types of the 9-point stencil: the pipelined kernel
#pragma push_macro("SPK_LAUNCH_DICT")
#undef SPK_LAUNCH_DICT
#define SPK_LAUNCH_DICT(BS_, ACC, RIDE, BTF)                                                                                    \
    do {                                                                                                                        \
        if (A.uniform)                                                                                                          \
            SPK_LAUNCH_PRODUCT((spmv_dict2_kernel<ACC, RIDE, BTF, 9, true>), dim3(grid + nride), dim3(kThreads), lds, s, d, x, \
                               y, bt ? bt->rowptr.p : nullptr, bt ? bt->colidx.p : nullptr, bt ? bt->val.p : nullptr, lam, od,  \
                               done, gr);                                                                                       \
        else                                                                                                                    \
            SPK_LAUNCH_PRODUCT((spmv_dict2_kernel<ACC, RIDE, BTF, 9, false>), dim3(grid + nride), dim3(kThreads), lds, s, d, x,\
                               y, bt ? bt->rowptr.p : nullptr, bt ? bt->colidx.p : nullptr, bt ? bt->val.p : nullptr, lam, od,  \
                               done, gr);                                                                                       \
    } while (0)
        SPK_DISPATCH_DICT(2)
#pragma pop_macro("SPK_LAUNCH_DICT")
    } else if (A.bs == 2) { SPK_DISPATCH_DICT(2) }
    else { SPK_DISPATCH_DICT(3) }
#undef SPK_DISPATCH_DICT
#undef SPK_LAUNCH_DICT
#undef SPK_LAUNCH_DICT_U
}

// ---------------------------------------------------------------------------
// FP32 damped-Jacobi Richardson sweep on the same layout (see jacobi_sweep_f32_kernel in spk_k_spmv.hip): the value is
// decoded exactly, rounded to single precision (= the float copy the other sweep kernels read), products rounded once
// each, summed per row in block order = CSR order -- the same bits as the CSR sweep and the oracle's float loop.
// ---------------------------------------------------------------------------
template <int BS, int U3>
__global__ __launch_bounds__(kThreads) void jacobi_sweep_f32_dict_kernel(DictArgs d, const float *__restrict__ d32, float omega,
                                                                         const float *__restrict__ x32,
                                                                         const float *__restrict__ yin, float *__restrict__ yout,
                                                                         const int32_t *__restrict__ done)
{
#pragma clang fp contract(off)  // every product and sum below is rounded on its own (the oracle's float loop)
    if (done && *done) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bx = (int)blockIdx.x;
    const int c0 = ((bx & 7) * d.chunks_per_xcd + (bx >> 3) * d.chunks_per_wg);
    const int c1 = min(min(c0 + d.chunks_per_wg, ((bx & 7) + 1) * d.chunks_per_xcd), d.nchunks);
    if (c0 >= c1) return;
    dict_load_lds(d, (d.nclass + 1) * BS * BS, smem);
    const int32_t *tlen = reinterpret_cast<const int32_t *>(smem);
    const int2 *tent = reinterpret_cast<const int2 *>(smem + 4 * ((d.ntype + 1) & ~1));
    const double2 *cv = reinterpret_cast<const double2 *>(smem + d.cls_off);
    const int32_t *fl = reinterpret_cast<const int32_t *>(smem + d.fld_off);
    for (int ch = c0; ch < c1; ++ch) {
        const int br = ch * kDictChunk + (int)threadIdx.x;
        if (br >= d.nbrows) continue;
        const int tc = (int)d.tid[br];
        float s[BS];
#pragma unroll
        for (int r = 0; r < BS; ++r) s[r] = 0.0f;
        const int len = tlen[tc];
        const int2 *te = tent + (size_t)tc * d.kmax;
        constexpr int G = BS == 2 ? 10 : 3;
        for (int k0 = 0; k0 < len; k0 += G) {
            int2 e[G];
            u64 w0[G], w1[BS == 2 ? 1 : G];
            float yv[G][BS];
            if (BS == 2) {
#pragma unroll
                for (int h = 0; h < G / 2; ++h) {
                    w0[2 * h] = w0[2 * h + 1] = 0ull;
                    if (k0 + 2 * h < len) dict_issue_pair2(d, (k0 >> 1) + h, br, w0[2 * h], w0[2 * h + 1]);
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const bool in = k0 + g < len;
                e[g] = in ? te[k0 + g] : make_int2(0, 0);
                if (in) {
                    if (BS == 3) dict_issue3(d, k0 + g, br, w0[g], w1[BS == 2 ? 0 : g]);
                    const int64_t c = (int64_t)br + e[g].x;
#pragma unroll
                    for (int j = 0; j < BS; ++j) yv[g][j] = yin[BS * c + j];
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (k0 + g < len) {
                    const double2 *cb = cv + (size_t)e[g].y * (BS * BS);
                    const int32_t *fb = fl + (size_t)e[g].y * (BS * BS);
#pragma unroll
                    for (int r = 0; r < BS; ++r)
#pragma unroll
                        for (int j = 0; j < BS; ++j)
                            s[r] = (s[r] + ((float)dict_decode(BS == 2 ? dict_field2(w0[g], fb[r * BS + j], d.strad != 0)
                                                                      : U3 ? dict_field3u<(U3 ? U3 : 1)>(w0[g], w1[BS == 2 ? 0 : g], d, r * BS + j)
                                                                           : dict_field(w0[g], w1[BS == 2 ? 0 : g], fb[r * BS + j]),
                                                               cb[r * BS + j]) * yv[g][j]));
                }
            }
        }
#pragma unroll
        for (int r = 0; r < BS; ++r) {
            const int64_t row = (int64_t)BS * br + r;
            yout[row] = yin[row] + ((omega * d32[row]) * (x32[row] - s[r]));
        }
    }
}

void jacobi_sweep_f32_dict(const DictDev &A, const float *d32, float omega, const float *x32, const float *yin, float *yout,
                           const int32_t *done, hipStream_t s)
{
    if (A.nbrows == 0) return;
    if (jacobi_sweep_f32_dict3(A, d32, omega, x32, yin, yout, done, s)) return;
    int grid = 0;
    const DictArgs d = dict_args(A, &grid);
#define SPK_LAUNCH_SWEEP(BS, U3)                                                                                                \
    hipLaunchKernelGGL((jacobi_sweep_f32_dict_kernel<BS, U3>), dim3(grid), dim3(kThreads), (size_t)A.lds_bytes, s, d, d32, omega, x32, \
                       yin, yout, done)
    if (A.bs == 2) SPK_LAUNCH_SWEEP(2, 0);
    else if (A.uniform3 == 0) SPK_LAUNCH_SWEEP(3, 0);
    else if (A.uniform3 == 1) SPK_LAUNCH_SWEEP(3, 1);
    else if (A.uniform3 == 2) SPK_LAUNCH_SWEEP(3, 2);
    else SPK_LAUNCH_SWEEP(3, 3);
#undef SPK_LAUNCH_SWEEP
}

// ---------------------------------------------------------------------------
// set-up: block classes (blocks equal up to rounding noise), then row types (the sequence of column offset and class
// along a block row), proposed by hashing into a small open-addressing table; then the per-entry granule and range of
// every class, the codes, and a verification pass that decodes EVERY value and compares its bits with the original.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long h, unsigned long long w)
{
    h ^= w + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 31;
    return h;
}
// slot of key h (claimed if new); -1: table full / too many keys (ctl[1] raised)
__device__ __forceinline__ int dict_insert(unsigned long long h, int32_t item, unsigned long long *keys, int32_t *rep,
                                           int32_t *ctl, int maxkeys)
{
    if (h == 0ull) h = 1ull;
    int idx = (int)(h & (unsigned long long)(kDictSlots - 1));
    for (int probe = 0; probe < kDictSlots; ++probe) {
        if (__hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return -1;
        unsigned long long prev = __hip_atomic_load(keys + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == 0ull) {
            prev = atomicCAS(keys + idx, 0ull, h);
            if (prev == 0ull && atomicAdd(ctl, 1) + 1 > maxkeys) {
                __hip_atomic_store(ctl + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return -1;
            }
        }
        if (prev == 0ull || prev == h) {
            // (millions of items fall into a few dozen classes: the atomic only where it would change something)
            if (item < __hip_atomic_load(rep + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(rep + idx, item);
            return idx;
        }
        idx = (idx + 1) & (kDictSlots - 1);
    }
    __hip_atomic_store(ctl + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return -1;
}

// value (r, j) of block q: bs = 2 -- v0 = (a00, a01) pairs, v1 = (a10, a11) pairs; bs = 3 -- nine planes of stride ldp in v0
template <int BS>
__device__ __forceinline__ double blk_val(const double *v0, const double *v1, int64_t ldp, int64_t q, int r, int j)
{
    if (BS == 2) return (r == 0 ? v0 : v1)[2 * q + j];
    return v0[(int64_t)(3 * r + j) * ldp + q];
}
// what two values of one class agree in: the value to about 1e-6 absolute (far above the assembly's rounding noise, far
// below the distance of the ideal entries); huge or non-finite values only match themselves
__device__ __forceinline__ unsigned long long coarse_key(double v)
{
    if (!(fabs(v) < 1e12)) return (unsigned long long)__double_as_longlong(v);
    return (unsigned long long)(long long)rint(v * 1048576.0);
}

template <int BS>
__global__ __launch_bounds__(kThreads) void dict_hash_blocks_kernel(const double *__restrict__ v0, const double *__restrict__ v1,
                                                                    int64_t ldp, int64_t nblocks, unsigned long long *keys,
                                                                    int32_t *rep, int32_t *slot, int32_t *ctl, int maxkeys)
{
    const int64_t q = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (q >= nblocks) return;
    unsigned long long h = 0x243F6A8885A308D3ull;
#pragma unroll
    for (int r = 0; r < BS; ++r)
#pragma unroll
        for (int j = 0; j < BS; ++j) h = mix64(h, coarse_key(blk_val<BS>(v0, v1, ldp, q, r, j)));
    slot[q] = dict_insert(h, (int32_t)q, keys, rep, ctl, maxkeys);
}
void dict_hash_blocks(int bs, const double *v0, const double *v1, int64_t ldp, int64_t nblocks, unsigned long long *keys,
                      int32_t *rep, int32_t *slot, int32_t *ctl, int maxkeys, hipStream_t s)
{
    if (nblocks == 0) return;
    const unsigned grid = (unsigned)((nblocks + kThreads - 1) / kThreads);
    if (bs == 2) hipLaunchKernelGGL((dict_hash_blocks_kernel<2>), dim3(grid), dim3(kThreads), 0, s, v0, v1, ldp, nblocks, keys, rep, slot, ctl, maxkeys);
    else hipLaunchKernelGGL((dict_hash_blocks_kernel<3>), dim3(grid), dim3(kThreads), 0, s, v0, v1, ldp, nblocks, keys, rep, slot, ctl, maxkeys);
}

// cls[(id, e)].base = entry e of the class's first block
template <int BS>
__global__ __launch_bounds__(kThreads) void dict_cls_base_kernel(const double *__restrict__ v0, const double *__restrict__ v1,
                                                                 int64_t ldp, const int32_t *__restrict__ rep_of_id, int nid,
                                                                 double *__restrict__ cls)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= nid * BS * BS) return;
    const int id = i / (BS * BS), e = i % (BS * BS);
    cls[2 * i] = blk_val<BS>(v0, v1, ldp, rep_of_id[id], e / BS, e % BS);
    cls[2 * i + 1] = 1.0;
}
// slot[q] -> class number; per (class, entry): the finest bit any deviation from the base uses (gexp, atomicMin) and the
// largest deviation (bits of |d|, atomicMax).  *bad: a deviation that is not exactly representable
template <int BS>
__global__ __launch_bounds__(kThreads) void dict_cls_stats_kernel(const double *__restrict__ v0, const double *__restrict__ v1,
                                                                  int64_t ldp, int64_t nblocks, const int32_t *__restrict__ slot2id,
                                                                  int32_t *__restrict__ slot, const double *__restrict__ cls,
                                                                  int32_t *__restrict__ gexp, unsigned long long *__restrict__ dmax,
                                                                  int32_t *__restrict__ bad)
{
    const int64_t q = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (q >= nblocks) return;
    const int sl = slot[q];
    const int id = sl >= 0 ? slot2id[sl] : -1;
    slot[q] = id;
    if (id < 0) {
        *bad = 1;
        return;
    }
#pragma unroll
    for (int r = 0; r < BS; ++r)
#pragma unroll
        for (int j = 0; j < BS; ++j) {
            const int e = r * BS + j;
            const double v = blk_val<BS>(v0, v1, ldp, q, r, j), base = cls[2 * (id * BS * BS + e)];
            const double dv = v - base;
            if (__double_as_longlong(base + dv) != __double_as_longlong(v) || !(fabs(dv) < 1e300)) {
                *bad = 1;
                continue;
            }
            if (dv == 0.0) continue;
            const unsigned long long bits = (unsigned long long)__double_as_longlong(fabs(dv));
            const int ef = (int)(bits >> 52);
            const unsigned long long mant = (bits & 0xFFFFFFFFFFFFFull) | (ef ? 0x10000000000000ull : 0ull);
            const int low = (ef ? ef - 1075 : -1074) + (int)__builtin_ctzll(mant);   // exponent of the lowest set bit
            if (low < __hip_atomic_load(gexp + id * BS * BS + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(gexp + id * BS * BS + e, low);
            if (bits > __hip_atomic_load(dmax + id * BS * BS + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dmax + id * BS * BS + e, bits);
        }
}
void dict_class_stats(int bs, const double *v0, const double *v1, int64_t ldp, int64_t nblocks, const int32_t *rep_of_id, int nid,
                      const int32_t *slot2id, int32_t *slot, double *cls, int32_t *gexp, unsigned long long *dmax, int32_t *bad,
                      hipStream_t s)
{
    if (nblocks == 0 || nid == 0) return;
    const unsigned g1 = (unsigned)((nid * bs * bs + kThreads - 1) / kThreads), g2 = (unsigned)((nblocks + kThreads - 1) / kThreads);
    if (bs == 2) {
        hipLaunchKernelGGL((dict_cls_base_kernel<2>), dim3(g1), dim3(kThreads), 0, s, v0, v1, ldp, rep_of_id, nid, cls);
        hipLaunchKernelGGL((dict_cls_stats_kernel<2>), dim3(g2), dim3(kThreads), 0, s, v0, v1, ldp, nblocks, slot2id, slot, cls, gexp, dmax, bad);
    } else {
        hipLaunchKernelGGL((dict_cls_base_kernel<3>), dim3(g1), dim3(kThreads), 0, s, v0, v1, ldp, rep_of_id, nid, cls);
        hipLaunchKernelGGL((dict_cls_stats_kernel<3>), dim3(g2), dim3(kThreads), 0, s, v0, v1, ldp, nblocks, slot2id, slot, cls, gexp, dmax, bad);
    }
}

// block rows: key = (length, then per block its column offset from the block row and its class)
__global__ __launch_bounds__(kThreads) void dict_hash_rows_kernel(const int32_t *__restrict__ browptr, const int32_t *__restrict__ bcol,
                                                                  const int32_t *__restrict__ blkid, int32_t nbrows,
                                                                  unsigned long long *keys, int32_t *rep, int32_t *slot, int32_t *ctl,
                                                                  int maxkeys, int kmax)
{
    const int br = blockIdx.x * kThreads + threadIdx.x;
    if (br >= nbrows) return;
    const int q0 = browptr[br], q1 = browptr[br + 1];
    if (q1 - q0 > kmax) {   // a row longer than a row type may be: no such layout
        __hip_atomic_store(ctl + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        slot[br] = -1;
        return;
    }
    unsigned long long h = mix64(0x13198A2E03707344ull, (unsigned long long)(q1 - q0));
    for (int q = q0; q < q1; ++q)
        h = mix64(h, ((unsigned long long)(uint32_t)(bcol[q] - br) << 32) | (unsigned long long)(uint32_t)blkid[q]);
    slot[br] = dict_insert(h, br, keys, rep, ctl, maxkeys);
}
void dict_hash_rows(const int32_t *browptr, const int32_t *bcol, const int32_t *blkid, int32_t nbrows, unsigned long long *keys,
                    int32_t *rep, int32_t *slot, int32_t *ctl, int maxkeys, int kmax, hipStream_t s)
{
    if (nbrows == 0) return;
    hipLaunchKernelGGL(dict_hash_rows_kernel, dim3((nbrows + kThreads - 1) / kThreads), dim3(kThreads), 0, s, browptr, bcol, blkid,
                       nbrows, keys, rep, slot, ctl, maxkeys, kmax);
}

// tab = [ntype lengths (padded to an even count)] [ntype x kmax x {offset, class}]
__global__ __launch_bounds__(kThreads) void dict_row_table_kernel(const int32_t *__restrict__ browptr, const int32_t *__restrict__ bcol,
                                                                  const int32_t *__restrict__ blkid, const int32_t *__restrict__ rep_of_id,
                                                                  int nid, int kmax, int32_t *__restrict__ tab)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= nid * kmax) return;
    const int id = i / kmax, k = i % kmax;
    const int br = rep_of_id[id];
    const int q0 = browptr[br], len = browptr[br + 1] - q0;
    if (k == 0) tab[id] = len;
    int32_t *ent = tab + ((nid + 1) & ~1) + 2 * (size_t)i;
    ent[0] = k < len ? bcol[q0 + k] - br : 0;
    ent[1] = k < len ? blkid[q0 + k] : 0;
}
__global__ __launch_bounds__(kThreads) void dict_row_verify_kernel(const int32_t *__restrict__ browptr, const int32_t *__restrict__ bcol,
                                                                   const int32_t *__restrict__ blkid, int32_t nbrows, int nid, int kmax,
                                                                   const int32_t *__restrict__ slot2id, const int32_t *__restrict__ slot,
                                                                   const int32_t *__restrict__ tab, uint16_t *__restrict__ tid,
                                                                   int32_t *__restrict__ bad)
{
    const int br = blockIdx.x * kThreads + threadIdx.x;
    if (br >= nbrows) return;
    const int sl = slot[br];
    const int id = sl >= 0 ? slot2id[sl] : -1;
    bool same = id >= 0;
    if (same) {
        const int q0 = browptr[br], len = browptr[br + 1] - q0;
        same = len == tab[id];
        const int32_t *ent = tab + ((nid + 1) & ~1) + 2 * (size_t)id * kmax;
        for (int k = 0; same && k < len; ++k) same = ent[2 * k] == bcol[q0 + k] - br && ent[2 * k + 1] == blkid[q0 + k];
    }
    if (!same) *bad = 1;
    tid[br] = (uint16_t)(same ? id : 0);
}
void dict_fill_rows(const int32_t *browptr, const int32_t *bcol, const int32_t *blkid, int32_t nbrows, const int32_t *rep_of_id,
                    int nid, int kmax, const int32_t *slot2id, const int32_t *slot, int32_t *tab, uint16_t *tid, int32_t *bad,
                    hipStream_t s)
{
    if (nbrows == 0 || nid == 0) return;
    hipLaunchKernelGGL(dict_row_table_kernel, dim3((nid * kmax + kThreads - 1) / kThreads), dim3(kThreads), 0, s, browptr, bcol, blkid,
                       rep_of_id, nid, kmax, tab);
    hipLaunchKernelGGL(dict_row_verify_kernel, dim3((nbrows + kThreads - 1) / kThreads), dim3(kThreads), 0, s, browptr, bcol, blkid,
                       nbrows, nid, kmax, slot2id, slot, tab, tid, bad);
}

// codes: k = (value - base) / 2^g as a bit field of its class entry's width in the block's word(s); then every value decoded
// as the product kernels decode it and compared bit by bit (*bad)
template <int BS>
__global__ __launch_bounds__(kThreads) void dict_encode_kernel(DictArgs d, const int32_t *__restrict__ browptr,
                                                               const int32_t *__restrict__ blkid, const double *__restrict__ v0,
                                                               const double *__restrict__ v1, int64_t ldp, unsigned char *__restrict__ codes,
                                                               int32_t *__restrict__ bad)
{
    const int64_t br = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (br >= d.nbrows) return;
    const int q0 = browptr[br], len = browptr[br + 1] - q0;
    for (int k = 0; k < len; ++k) {
        const int id = blkid[q0 + k];
        u64 w[2] = {0ull, 0ull};
#pragma unroll
        for (int r = 0; r < BS; ++r)
#pragma unroll
            for (int j = 0; j < BS; ++j) {
                const int e = r * BS + j;
                const double base = d.cls[2 * (id * BS * BS + e)], sc = d.cls[2 * (id * BS * BS + e) + 1];
                const int fd = d.fld[id * BS * BS + e];
                // (2x2: offset inside a 32-bit half, bit 31 = the high half; 3x3: shift inside a 64-bit word, word number)
                const int wd = BS == 2 ? (fd >> 8) & 31 : (fd >> 8) & 255;
                const int sh = BS == 2 ? ((fd & kDictAcross) ? (fd & 63) : (fd & 31) + ((fd >> 31) & 1) * 32) : (fd & 255);
                const int wi = BS == 2 ? 0 : (fd >> 16);
                const double kq = (blk_val<BS>(v0, v1, ldp, q0 + k, r, j) - base) / sc;
                const long long kk = (long long)rint(kq);
                const long long lim = 1ll << (wd - 1);
                if ((double)kk != kq || kk < -lim || kk > lim - 1) *bad = 1;
                w[wi] |= ((u64)kk & ((wd >= 64) ? ~0ull : ((1ull << wd) - 1ull))) << sh;
            }
        u64 *p = dict_word_ptr<BS>(d, codes, k, br);
        p[0] = w[0];
        if (BS == 3) p[1] = w[1];
    }
}
template <int BS>
__global__ __launch_bounds__(kThreads) void dict_verify_kernel(DictArgs d, const int32_t *__restrict__ browptr,
                                                               const int32_t *__restrict__ bcol, const double *__restrict__ v0,
                                                               const double *__restrict__ v1, int64_t ldp, int32_t *__restrict__ bad)
{
    const int64_t br = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (br >= d.nbrows) return;
    const int t = d.tid[br];
    const int q0 = browptr[br], len = browptr[br + 1] - q0;
    const int32_t *ent = d.tab + ((d.ntype + 1) & ~1) + 2 * (size_t)t * d.kmax;
    bool ok = t < d.ntype && d.tab[t] == len;
    for (int k = 0; ok && k < len; ++k) {
        u64 w0 = 0ull, w1 = 0ull;
        if (BS == 2) {
            u64 a0, a1;
            dict_issue_pair2(d, k >> 1, br, a0, a1);
            w0 = (k & 1) ? a1 : a0;
        } else {
            dict_issue3(d, k, br, w0, w1);
        }
        ok = ok && bcol[q0 + k] == (int)br + ent[2 * k];
        const int cls_id = ent[2 * k + 1];
        const double2 *cb = reinterpret_cast<const double2 *>(d.cls) + (size_t)cls_id * (BS * BS);
        const int32_t *fb = d.fld + (size_t)cls_id * (BS * BS);
#pragma unroll
        for (int r = 0; r < BS; ++r)
#pragma unroll
            for (int j = 0; j < BS; ++j)
                ok = ok && __double_as_longlong(dict_decode(BS == 2 ? dict_field2(w0, fb[r * BS + j], d.strad != 0) : dict_field(w0, w1, fb[r * BS + j]), cb[r * BS + j])) ==
                               __double_as_longlong(blk_val<BS>(v0, v1, ldp, q0 + k, r, j));
    }
    if (!ok) *bad = 1;
}
void dict_encode_verify(const DictDev &A, const int32_t *browptr, const int32_t *bcol, const int32_t *blkid, const double *v0,
                        const double *v1, int64_t ldp, int32_t *bad, hipStream_t s)
{
    if (A.nbrows == 0) return;
    int grid = 0;
    const DictArgs d = dict_args(A, &grid);
    const unsigned g = (unsigned)((A.nbrows + kThreads - 1) / kThreads);
    if (A.bs == 2) {
        hipLaunchKernelGGL((dict_encode_kernel<2>), dim3(g), dim3(kThreads), 0, s, d, browptr, blkid, v0, v1, ldp, A.codes.p, bad);
        hipLaunchKernelGGL((dict_verify_kernel<2>), dim3(g), dim3(kThreads), 0, s, d, browptr, bcol, v0, v1, ldp, bad);
    } else {
        hipLaunchKernelGGL((dict_encode_kernel<3>), dim3(g), dim3(kThreads), 0, s, d, browptr, blkid, v0, v1, ldp, A.codes.p, bad);
        hipLaunchKernelGGL((dict_verify_kernel<3>), dim3(g), dim3(kThreads), 0, s, d, browptr, bcol, v0, v1, ldp, bad);
    }
}

}  // namespace k
}  // namespace spk
