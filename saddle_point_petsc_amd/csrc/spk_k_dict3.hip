// spk_k_dict3.hip -- the product and the FP32 Richardson sweep on ROW TYPES + DEVIATION CODES for 3x3 blocks whose row
// types hold at most 27 blocks (the 27-point stencil of a Q1 hexahedral grid: BASELINE config 5) and whose classes share
// one field layout (DictDev::uniform3), software-pipelined.  gfx950, wave64.
//
// The plain kernels (spk_k_dict.hip) issue the loads of a few blocks, wait, decode, and turn to the next few: a chain of
// memory round trips per block row (nine with three blocks in flight: 264 us per product on the 256 x 256 x 32 slab,
// three with nine: 232 us -- 0.97 GB, so 0.46 / 0.52 of the 8 TB/s peak).  Here a block row is three GROUPS of nine
// blocks and the groups of a workgroup's chunks form one sequence: the loads of group q+1 -- nine 16-byte code words
// per lane, the gathers of x, the row's y -- are in flight while group q is decoded, in two register stages.  As in
// spmv_dict2_kernel there is no branch around a load (the compiler's wait counts stay exact): a position beyond a row's
// length reads through the range check of a buffer descriptor (zeros) with the NULL class (decodes to +0), a group beyond
// the workgroup's range re-reads the last block row and is not computed.
//
// Parity: blocks in the block row's CSR order, every product rounded on its own (contraction off), added in that order
// from +0 -- the sums of the plain kernel, the blocked kernel, the CSR kernel and the oracle, bit for bit
// (tests/test_gpu_dict.py::test_dictionary_3d_grid, tests/test_gpu_configs.py).
#include "spk_dict.hpp"

#include <cstring>

namespace spk {
namespace k {

constexpr int kD3G = 9;        // blocks per group
constexpr int kD3Groups = 3;   // groups per block row (27 positions)

// the LDS copy of the row types as the issue stage wants it: byte offsets (stride bytes per block column) instead of block
// columns; beyond a row's length an offset outside the buffer and the null class
__device__ __forceinline__ void dict3_prepare_types(const DictArgs &d, const int32_t *tlen, int2 *tent, int stride)
{
    for (int i = threadIdx.x; i < d.ntype * 27; i += kThreads) {
        const int t = i / 27, k = i - t * 27;
        int2 e = tent[i];
        if (k < tlen[t]) e.x *= stride;
        else e = make_int2((int)0x80000000u, d.nclass);
        tent[i] = e;
    }
    __syncthreads();
}

template <bool ACC>
struct Dict3Stage {
    u64 w0[kD3G], w1[kD3G];
    double x0[kD3G], x1[kD3G], x2[kD3G];
    int cls[kD3G];
    double y0, y1, y2;
};

// loads of group q (chunk q / 3, positions 9 (q % 3) ...): all of them unconditional
template <bool ACC>
__device__ __forceinline__ void dict3_issue(const DictArgs &d, __amdgpu_buffer_rsrc_t xr, const double *__restrict__ y,
                                            const int2 *tent, int tc, int brr, int g, Dict3Stage<ACC> &S)
{
    const int k0 = kD3G * g;
#pragma unroll
    for (int j = 0; j < kD3G; ++j) {
        const int4v r = __builtin_nontemporal_load(reinterpret_cast<const int4v *>(d.codes + d.plane_off[k0 + j]) + brr);
        S.w0[j] = (u64)(uint32_t)r.x | ((u64)(uint32_t)r.y << 32);
        S.w1[j] = (u64)(uint32_t)r.z | ((u64)(uint32_t)r.w << 32);
    }
    if (ACC) {
        S.y0 = y[3 * (int64_t)brr];
        S.y1 = y[3 * (int64_t)brr + 1];
        S.y2 = y[3 * (int64_t)brr + 2];
    }
    const int2 *te = tent + (size_t)tc * 27 + k0;
    const uint32_t b24 = (uint32_t)brr * 24u;
#pragma unroll
    for (int j = 0; j < kD3G; ++j) {
        const int2 e = te[j];
        S.cls[j] = e.y;
        const uint32_t off = b24 + (uint32_t)e.x;
        const int4v a = __builtin_bit_cast(int4v, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)off, 0, 0));
        const int2v b = __builtin_bit_cast(int2v, __builtin_amdgcn_raw_buffer_load_b64(xr, (int)(off + 16u), 0, 0));
        S.x0[j] = __hiloint2double(a.y, a.x);
        S.x1[j] = __hiloint2double(a.w, a.z);
        S.x2[j] = __hiloint2double(b.y, b.x);
    }
}

template <bool ACC, bool RIDE, bool BT, int U3>
__global__ __launch_bounds__(kThreads) void spmv_dict3_kernel(DictArgs d, const double *__restrict__ x, double *__restrict__ y,
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
    // the row type a group needs is requested one group ahead (two registers, alternating with the stages)
    int tA = (int)d.tid[rowof(c0)];
    int tB = tA;
    dict_load_lds(d, (d.nclass + 1) * 9, smem);
    const int32_t *tlen = reinterpret_cast<const int32_t *>(smem);
    int2 *tent = reinterpret_cast<int2 *>(smem + 4 * ((d.ntype + 1) & ~1));
    const double2 *cv = reinterpret_cast<const double2 *>(smem + d.cls_off);
    dict3_prepare_types(d, tlen, tent, 24);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(x), 0, 24 * d.nbrows, 0x00020000);

    Dict3Stage<ACC> SA, SB;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    const int nq = (c1 - c0) * kD3Groups;
    auto compute = [&](int q, const Dict3Stage<ACC> &S) {
        const int g = q % kD3Groups;
        if (g == 0) s0 = s1 = s2 = 0.0;
#pragma unroll
        for (int j = 0; j < kD3G; ++j) {
            const double2 *cb = cv + (size_t)S.cls[j] * 9;
            s0 += dict_decode(dict_field3u<U3>(S.w0[j], S.w1[j], d, 0), cb[0]) * S.x0[j];
            s0 += dict_decode(dict_field3u<U3>(S.w0[j], S.w1[j], d, 1), cb[1]) * S.x1[j];
            s0 += dict_decode(dict_field3u<U3>(S.w0[j], S.w1[j], d, 2), cb[2]) * S.x2[j];
            s1 += dict_decode(dict_field3u<U3>(S.w0[j], S.w1[j], d, 3), cb[3]) * S.x0[j];
            s1 += dict_decode(dict_field3u<U3>(S.w0[j], S.w1[j], d, 4), cb[4]) * S.x1[j];
            s1 += dict_decode(dict_field3u<U3>(S.w0[j], S.w1[j], d, 5), cb[5]) * S.x2[j];
            s2 += dict_decode(dict_field3u<U3>(S.w0[j], S.w1[j], d, 6), cb[6]) * S.x0[j];
            s2 += dict_decode(dict_field3u<U3>(S.w0[j], S.w1[j], d, 7), cb[7]) * S.x1[j];
            s2 += dict_decode(dict_field3u<U3>(S.w0[j], S.w1[j], d, 8), cb[8]) * S.x2[j];
        }
        if (g != kD3Groups - 1) return;
        const int brc = (c0 + q / kD3Groups) * kDictChunk + (int)threadIdx.x;
        if (brc > last) return;
        double s[3] = {s0, s1, s2};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int64_t row = 3 * (int64_t)brc + r;
            if (od.rowptr)
                for (int k = od.rowptr[row]; k < od.rowptr[row + 1]; ++k) s[r] = __builtin_fma(od.val[k], od.xg[od.colidx[k]], s[r]);
            if (BT)
                for (int k = bt_rowptr[row]; k < bt_rowptr[row + 1]; ++k) s[r] = __builtin_fma(bt_val[k], lam[bt_colidx[k]], s[r]);
            if (ACC) s[r] += r == 0 ? S.y0 : r == 1 ? S.y1 : S.y2;
            y[row] = s[r];
        }
    };
    // group q: chunk c0 + q / 3, group q % 3
    auto chunk_of = [&](int q) { return c0 + q / kD3Groups; };

    dict3_issue<ACC>(d, xr, y, tent, tA, rowof(c0), 0, SA);
    for (int q = 0;; q += 2) {
        tA = (int)d.tid[rowof(chunk_of(q + 2))];
        dict3_issue<ACC>(d, xr, y, tent, tB, rowof(chunk_of(q + 1)), (q + 1) % kD3Groups, SB);
        compute(q, SA);
        if (q + 1 >= nq) break;
        tB = (int)d.tid[rowof(chunk_of(q + 3))];
        dict3_issue<ACC>(d, xr, y, tent, tA, rowof(chunk_of(q + 2)), (q + 2) % kD3Groups, SA);
        compute(q + 1, SB);
        if (q + 2 >= nq) break;
    }
}

static bool dict3_applies(const DictDev &A)
{
    // SPK_DICT3_PIPELINE=0 / 1: never / at any size (tests run both forms on small grids); read at every call
    const char *env = getenv("SPK_DICT3_PIPELINE");
    const bool off = env && !strcmp(env, "0"), force = env && !strcmp(env, "1");
    // (from a million block rows: eight chunks and more per workgroup to pipeline over; measured on 96^3 = 0.88 M block
    // rows: 108 us plain with nine blocks in flight, 125 us pipelined; on the 256 x 256 x 32 slab 232 against 222)
    return !off && A.bs == 3 && A.uniform3 >= 1 && A.uniform3 <= 3 && A.kmax == 27 && (force || A.nbrows >= (1 << 20)) &&
           (int64_t)A.nbrows * 24 < (1ll << 31);
}

static DictArgs dict3_args(const DictDev &A, int *grid)
{
    DictArgs d = dict_args(A, grid);
    // workgroups of a large launch: two per CU, all co-resident, each pipelines its chunks
    static const int wgs = [] { const char *e = getenv("SPK_DICT3_WGS"); return e && atoi(e) > 0 ? atoi(e) : 512; }();   // (developer knob)
    d.chunks_per_wg = std::max(1, d.nchunks / wgs);
    const int cpx = (d.nchunks + 7) / 8;
    d.chunks_per_xcd = (cpx + d.chunks_per_wg - 1) / d.chunks_per_wg * d.chunks_per_wg;
    *grid = 8 * (d.chunks_per_xcd / d.chunks_per_wg);
    return d;
}

bool spmv_dict3(const DictDev &A, const double *x, double *y, const CsrDev *bt, const double *lam, const int32_t *done,
                hipStream_t s, bool accumulate, const OffDiag *odp, const GivensRider *rider)
{
    if (!dict3_applies(A) || A.nbrows == 0) return false;
    int grid = 0;
    const DictArgs d = dict3_args(A, &grid);
    const OffDiag od = odp ? *odp : OffDiag{nullptr, nullptr, nullptr, nullptr};
    const GivensRider gr = rider ? *rider : no_rider();
    const int nride = rider ? 1 : 0;
    size_t lds = (size_t)A.lds_bytes;
    if (rider) lds = std::max(lds, sizeof(double) * (size_t)(kThreads + 4 * (kMaxNv + 2) + 4));
#define SPK_L3U(ACC, RIDE, BTF, U3)                                                                                             \
    SPK_LAUNCH_PRODUCT((spmv_dict3_kernel<ACC, RIDE, BTF, U3>), dim3(grid + nride), dim3(kThreads), lds, s, d, x, y,           \
                       bt ? bt->rowptr.p : nullptr, bt ? bt->colidx.p : nullptr, bt ? bt->val.p : nullptr, lam, od, done, gr)
#define SPK_L3(ACC, RIDE, BTF)                                                                                                  \
    do {                                                                                                                        \
        if (A.uniform3 == 1) SPK_L3U(ACC, RIDE, BTF, 1);                                                                        \
        else if (A.uniform3 == 2) SPK_L3U(ACC, RIDE, BTF, 2);                                                                   \
        else SPK_L3U(ACC, RIDE, BTF, 3);                                                                                        \
    } while (0)
    if (bt) {
        if (rider) fail(SPK_ERR_ARG, "spmv_dict3: B^T rows and a rider in one launch");
        if (accumulate) SPK_L3(true, false, true);
        else SPK_L3(false, false, true);
    } else if (accumulate) {
        if (rider) SPK_L3(true, true, false);
        else SPK_L3(true, false, false);
    } else {
        if (rider) SPK_L3(false, true, false);
        else SPK_L3(false, false, false);
    }
#undef SPK_L3
#undef SPK_L3U
    return true;
}

// ---------------------------------------------------------------------------
// FP32 damped-Jacobi Richardson sweep (jacobi_sweep_f32_dict_kernel, spk_k_dict.hip), pipelined the same way: the value is
// decoded exactly, rounded to single precision, products rounded once each, summed per row in block order.
// ---------------------------------------------------------------------------
struct Dict3StageF {
    u64 w0[kD3G], w1[kD3G];
    float x0[kD3G], x1[kD3G], x2[kD3G];
    int cls[kD3G];
};

__device__ __forceinline__ void dict3_issue_f(const DictArgs &d, __amdgpu_buffer_rsrc_t yr, const int2 *tent, int tc, int brr, int g,
                                              Dict3StageF &S)
{
    const int k0 = kD3G * g;
#pragma unroll
    for (int j = 0; j < kD3G; ++j) {
        const int4v r = __builtin_nontemporal_load(reinterpret_cast<const int4v *>(d.codes + d.plane_off[k0 + j]) + brr);
        S.w0[j] = (u64)(uint32_t)r.x | ((u64)(uint32_t)r.y << 32);
        S.w1[j] = (u64)(uint32_t)r.z | ((u64)(uint32_t)r.w << 32);
    }
    const int2 *te = tent + (size_t)tc * 27 + k0;
    const uint32_t b12 = (uint32_t)brr * 12u;
#pragma unroll
    for (int j = 0; j < kD3G; ++j) {
        const int2 e = te[j];
        S.cls[j] = e.y;
        const uint32_t off = b12 + (uint32_t)e.x;
        const int2v a = __builtin_bit_cast(int2v, __builtin_amdgcn_raw_buffer_load_b64(yr, (int)off, 0, 0));
        const int b = __builtin_bit_cast(int, __builtin_amdgcn_raw_buffer_load_b32(yr, (int)(off + 8u), 0, 0));
        S.x0[j] = __int_as_float(a.x);
        S.x1[j] = __int_as_float(a.y);
        S.x2[j] = __int_as_float(b);
    }
}

template <int U3>
__global__ __launch_bounds__(kThreads) void jacobi_sweep_f32_dict3_kernel(DictArgs d, const float *__restrict__ d32, float omega,
                                                                          const float *__restrict__ x32,
                                                                          const float *__restrict__ yin, float *__restrict__ yout,
                                                                          const int32_t *__restrict__ done)
{
#pragma clang fp contract(off)
    if (done && *done) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bx = (int)blockIdx.x;
    const int c0 = ((bx & 7) * d.chunks_per_xcd + (bx >> 3) * d.chunks_per_wg);
    const int c1 = min(min(c0 + d.chunks_per_wg, ((bx & 7) + 1) * d.chunks_per_xcd), d.nchunks);
    if (c0 >= c1) return;
    const int last = d.nbrows - 1;
    auto rowof = [&](int ch) { return min(ch * kDictChunk + (int)threadIdx.x, last); };
    int tA = (int)d.tid[rowof(c0)];
    int tB = tA;
    dict_load_lds(d, (d.nclass + 1) * 9, smem);
    const int32_t *tlen = reinterpret_cast<const int32_t *>(smem);
    int2 *tent = reinterpret_cast<int2 *>(smem + 4 * ((d.ntype + 1) & ~1));
    const double2 *cv = reinterpret_cast<const double2 *>(smem + d.cls_off);
    dict3_prepare_types(d, tlen, tent, 12);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(yin), 0, 12 * d.nbrows, 0x00020000);

    Dict3StageF SA, SB;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
    const int nq = (c1 - c0) * kD3Groups;
    auto compute = [&](int q, const Dict3StageF &S) {
        const int g = q % kD3Groups;
        if (g == 0) s0 = s1 = s2 = 0.0f;
#pragma unroll
        for (int j = 0; j < kD3G; ++j) {
            const double2 *cb = cv + (size_t)S.cls[j] * 9;
            s0 = s0 + (float)dict_decode(dict_field3u<U3>(S.w0[j], S.w1[j], d, 0), cb[0]) * S.x0[j];
            s0 = s0 + (float)dict_decode(dict_field3u<U3>(S.w0[j], S.w1[j], d, 1), cb[1]) * S.x1[j];
            s0 = s0 + (float)dict_decode(dict_field3u<U3>(S.w0[j], S.w1[j], d, 2), cb[2]) * S.x2[j];
            s1 = s1 + (float)dict_decode(dict_field3u<U3>(S.w0[j], S.w1[j], d, 3), cb[3]) * S.x0[j];
            s1 = s1 + (float)dict_decode(dict_field3u<U3>(S.w0[j], S.w1[j], d, 4), cb[4]) * S.x1[j];
            s1 = s1 + (float)dict_decode(dict_field3u<U3>(S.w0[j], S.w1[j], d, 5), cb[5]) * S.x2[j];
            s2 = s2 + (float)dict_decode(dict_field3u<U3>(S.w0[j], S.w1[j], d, 6), cb[6]) * S.x0[j];
            s2 = s2 + (float)dict_decode(dict_field3u<U3>(S.w0[j], S.w1[j], d, 7), cb[7]) * S.x1[j];
            s2 = s2 + (float)dict_decode(dict_field3u<U3>(S.w0[j], S.w1[j], d, 8), cb[8]) * S.x2[j];
        }
        if (g != kD3Groups - 1) return;
        const int brc = (c0 + q / kD3Groups) * kDictChunk + (int)threadIdx.x;
        if (brc > last) return;
        const float s[3] = {s0, s1, s2};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int64_t row = 3 * (int64_t)brc + r;
            yout[row] = yin[row] + ((omega * d32[row]) * (x32[row] - s[r]));
        }
    };
    auto chunk_of = [&](int q) { return c0 + q / kD3Groups; };

    dict3_issue_f(d, yr, tent, tA, rowof(c0), 0, SA);
    for (int q = 0;; q += 2) {
        tA = (int)d.tid[rowof(chunk_of(q + 2))];
        dict3_issue_f(d, yr, tent, tB, rowof(chunk_of(q + 1)), (q + 1) % kD3Groups, SB);
        compute(q, SA);
        if (q + 1 >= nq) break;
        tB = (int)d.tid[rowof(chunk_of(q + 3))];
        dict3_issue_f(d, yr, tent, tA, rowof(chunk_of(q + 2)), (q + 2) % kD3Groups, SA);
        compute(q + 1, SB);
        if (q + 2 >= nq) break;
    }
}

bool jacobi_sweep_f32_dict3(const DictDev &A, const float *d32, float omega, const float *x32, const float *yin, float *yout,
                            const int32_t *done, hipStream_t s)
{
    if (!dict3_applies(A) || A.nbrows == 0 || (int64_t)A.nbrows * 12 >= (1ll << 31)) return false;
    int grid = 0;
    const DictArgs d = dict3_args(A, &grid);
    if (A.uniform3 == 1)
        hipLaunchKernelGGL((jacobi_sweep_f32_dict3_kernel<1>), dim3(grid), dim3(kThreads), (size_t)A.lds_bytes, s, d, d32, omega, x32, yin, yout, done);
    else if (A.uniform3 == 2)
        hipLaunchKernelGGL((jacobi_sweep_f32_dict3_kernel<2>), dim3(grid), dim3(kThreads), (size_t)A.lds_bytes, s, d, d32, omega, x32, yin, yout, done);
    else
        hipLaunchKernelGGL((jacobi_sweep_f32_dict3_kernel<3>), dim3(grid), dim3(kThreads), (size_t)A.lds_bytes, s, d, d32, omega, x32, yin, yout, done);
    return true;
}

}  // namespace k
}  // namespace spk
