// spk_dict.hpp -- device-side pieces of the "row types + deviation codes" layout (DictDev, spk_internal.hpp) shared by the
// product kernels (spk_k_dict.hip) and the resident restart-cycle kernel (spk_k_resident.hip).  Device code only.
#pragma once
#include "spk_device.hpp"

namespace spk {
namespace k {

constexpr int kDictChunk = kThreads;   // block rows per chunk: one per thread
typedef int int2v __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------
// the product
// ---------------------------------------------------------------------------
struct DictArgs {
    const uint16_t *tid;
    const int32_t *tab;
    const double *cls;             // nclass x bs*bs x {base, scale}
    const unsigned char *codes;
    int32_t nbrows, ntype, nclass, kmax;
    int64_t nbrows_pad;            // rows of a code plane (nbrows rounded up to 16); a wide position has two planes
    int32_t nchunks, chunks_per_xcd, chunks_per_wg;
    int32_t tab_ints, cls_off;     // ints of the type tables; byte offset of the class table in LDS
    uint32_t wide_mask;            // bit k: position k of a block row holds 32-bit codes
    int64_t plane_off[kDictMaxK];  // byte offset of the code plane of position k
};

// bytes of one block's 16-bit halves in a plane: bs = 2 -- four in 8 bytes; bs = 3 -- three rows of three, each row in 8
__host__ __device__ constexpr int dict_block_bytes(int bs) { return bs == 2 ? 8 : 24; }

// tables -> LDS (every workgroup; a few KB out of L2)
__device__ __forceinline__ void dict_load_lds(const DictArgs &d, int nclsvals, char *smem)
{
    int32_t *ti = reinterpret_cast<int32_t *>(smem);
    for (int i = threadIdx.x; i < d.tab_ints; i += kThreads) ti[i] = d.tab[i];
    double2 *cv = reinterpret_cast<double2 *>(smem + d.cls_off);
    const double2 *src = reinterpret_cast<const double2 *>(d.cls);
    for (int i = threadIdx.x; i < nclsvals; i += kThreads) cv[i] = src[i];
    __syncthreads();
}

// The codes of block position k of block row br, as stored: every position has a plane of 16-bit LOW halves (bs = 2:
// four in 8 bytes; bs = 3: three rows of three, each row in 8 bytes); a WIDE position has a second plane of the same
// shape with the high halves, adjusted so that code = sext(low) + (high << 16) needs no case distinction.  Narrow
// positions leave `hi` at zero.  The loads are issued here, unconditionally but for the one uniform test, and unpacked
// later (dict_unpack): nothing below waits for memory, so a row's loads are all in flight together.
template <int BS>
struct DictRaw {
    int2v lo[BS == 2 ? 1 : 3], hi[BS == 2 ? 1 : 3];
};
template <int BS>
__device__ __forceinline__ void dict_issue(const DictArgs &d, int k, int64_t br, DictRaw<BS> &w)
{
    constexpr int R = BS == 2 ? 1 : 3;
    const bool wide = (d.wide_mask >> k) & 1u;
    const int2v *p = reinterpret_cast<const int2v *>(d.codes + d.plane_off[k]) + R * br;
#pragma unroll
    for (int r = 0; r < R; ++r) w.lo[r] = __builtin_nontemporal_load(p + r);
#pragma unroll
    for (int r = 0; r < R; ++r) w.hi[r] = int2v{0, 0};
    if (wide) {
        const int2v *q = p + (int64_t)R * d.nbrows_pad;
#pragma unroll
        for (int r = 0; r < R; ++r) w.hi[r] = __builtin_nontemporal_load(q + r);
    }
}
template <int BS>
__device__ __forceinline__ void dict_unpack(const DictRaw<BS> &w, int (&c)[BS * BS])
{
    constexpr int R = BS == 2 ? 1 : 3;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int l0 = (int)(short)(w.lo[r].x & 0xffff), l1 = w.lo[r].x >> 16, l2 = (int)(short)(w.lo[r].y & 0xffff), l3 = w.lo[r].y >> 16;
        const int h0 = (int)(short)(w.hi[r].x & 0xffff), h1 = w.hi[r].x >> 16, h2 = (int)(short)(w.hi[r].y & 0xffff), h3 = w.hi[r].y >> 16;
        if (BS == 2) {
            c[0] = l0 + h0 * 65536;
            c[1] = l1 + h1 * 65536;
            c[2] = l2 + h2 * 65536;
            c[3] = l3 + h3 * 65536;
        } else {
            c[3 * r] = l0 + h0 * 65536;
            c[3 * r + 1] = l1 + h1 * 65536;
            c[3 * r + 2] = l2 + h2 * 65536;
        }
    }
}
// value = base + k * 2^g: both terms exact, the sum representable (it is the stored value): exact under any rounding
__device__ __forceinline__ double dict_decode(int code, double2 bs)
{
    return __builtin_fma((double)code, bs.y, bs.x);
}

inline DictArgs dict_args(const DictDev &A, int *grid)
{
    DictArgs d{};
    d.tid = A.tid.p;
    d.tab = A.tab.p;
    d.cls = A.cls.p;
    d.codes = A.codes.p;
    d.nbrows = A.nbrows;
    d.ntype = A.ntype;
    d.nclass = A.nclass;
    d.kmax = A.kmax;
    d.nbrows_pad = ((int64_t)A.nbrows + 15) & ~(int64_t)15;
    d.nchunks = (A.nbrows + kDictChunk - 1) / kDictChunk;
    // large systems: a few chunks per workgroup (the table copy is paid once per workgroup), about 2048 workgroups
    d.chunks_per_wg = std::max(1, d.nchunks / 2048);
    const int cpx = (d.nchunks + 7) / 8;
    d.chunks_per_xcd = (cpx + d.chunks_per_wg - 1) / d.chunks_per_wg * d.chunks_per_wg;
    d.tab_ints = ((A.ntype + 1) & ~1) + 2 * A.ntype * A.kmax;
    d.cls_off = (4 * d.tab_ints + 15) & ~15;
    d.wide_mask = A.wide_mask;
    for (int k = 0; k < kDictMaxK; ++k) d.plane_off[k] = A.plane_off[k];
    *grid = 8 * (d.chunks_per_xcd / d.chunks_per_wg);
    return d;
}

}  // namespace k
}  // namespace spk
