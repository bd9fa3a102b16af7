// spk_dict.hpp -- device-side pieces of the "row types + deviation codes" layout (DictDev, spk_internal.hpp) shared by the
// product kernels (spk_k_dict.hip) and the resident restart-cycle kernel (spk_k_resident.hip).  Device code only.
#pragma once
#include "spk_device.hpp"

namespace spk {
namespace k {

constexpr int kDictChunk = kThreads;   // block rows per chunk: one per thread
constexpr int kDictG3 = 9;             // 3x3 blocks whose loads are in flight together in the product kernel
constexpr int kDict2Wgs = 512;         // pipelined 2x2 product: workgroups of a large launch (two per CU, all co-resident; each pipelines its chunks)
typedef int int2v __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

// ---------------------------------------------------------------------------
// the product
// ---------------------------------------------------------------------------
struct DictArgs {
    const uint16_t *tid;
    const int32_t *tab;
    const double *cls;             // (nclass + 1) x bs*bs x {base, 2^g}; the last one the NULL class: base 0, scale 0
    const int32_t *fld;            // nclass x bs*bs bit fields: shift | width << 8 | word << 16
    const unsigned char *codes;
    const double *zpad;            // 32 bytes of zeros: what a position beyond a row's length gathers (with the null class)
    int32_t nbrows, ntype, nclass, kmax;
    int32_t nchunks, chunks_per_xcd, chunks_per_wg;
    int32_t tab_ints, cls_off, fld_off;   // ints of the type tables; byte offsets of the class and field tables in LDS
    // code planes.  bs = 2: one 64-bit word per block, the words of positions 2p and 2p+1 of a block row side by side in
    // plane p (16 bytes per block row: one full-width load; an odd last position: a plane of 8-byte words).
    // bs = 3: two words per block, plane k = position k (16 bytes per block row)
    int64_t plane_off[kDictMaxK];
    int32_t strad;                 // 2x2: some field lies across the halves of its word (dict_field2)
    int32_t u3l[9], u3r[9];        // 3x3, one field layout for all classes (DictDev::uniform3): shifts of dict_field3u
    int32_t uw[4];                 // 2x2, uniform field layout (DictDev::uniform): the widths of the four entries; uw[0] = 0: not uniform
};

// tables -> LDS (every workgroup; a few KB out of L2)
__device__ __forceinline__ void dict_load_lds(const DictArgs &d, int nclsvals, char *smem)
{
    int32_t *ti = reinterpret_cast<int32_t *>(smem);
    for (int i = threadIdx.x; i < d.tab_ints; i += blockDim.x) ti[i] = d.tab[i];
    double2 *cv = reinterpret_cast<double2 *>(smem + d.cls_off);
    const double2 *src = reinterpret_cast<const double2 *>(d.cls);
    for (int i = threadIdx.x; i < nclsvals; i += blockDim.x) cv[i] = src[i];
    int32_t *fl = reinterpret_cast<int32_t *>(smem + d.fld_off);
    for (int i = threadIdx.x; i < nclsvals; i += blockDim.x) fl[i] = d.fld[i];
    __syncthreads();
}

// The stored form of a block: its bs*bs integer deviations k (value = base + k 2^g), each as a two's-complement BIT FIELD
// of the width its class entry needs (1 .. 31 bits; the widths of a class fit 64 bits for 2x2 blocks, twice 64 for 3x3
// ones), packed into one / two 64-bit words.  Loads are issued first and unpacked later: a row's loads are all in flight
// together, all of them 16 bytes per lane but an odd last position of a dof-2 row.
__device__ __forceinline__ void dict_issue_pair2(const DictArgs &d, int p, int64_t br, u64 &w0, u64 &w1)
{
    const unsigned char *base = d.codes + d.plane_off[p];
    if (2 * p + 1 < d.kmax) {
        const int4v r = __builtin_nontemporal_load(reinterpret_cast<const int4v *>(base) + br);
        w0 = (u64)(uint32_t)r.x | ((u64)(uint32_t)r.y << 32);
        w1 = (u64)(uint32_t)r.z | ((u64)(uint32_t)r.w << 32);
    } else {
        const int2v r = __builtin_nontemporal_load(reinterpret_cast<const int2v *>(base) + br);
        w0 = (u64)(uint32_t)r.x | ((u64)(uint32_t)r.y << 32);
        w1 = 0ull;
    }
}
__device__ __forceinline__ void dict_issue3(const DictArgs &d, int k, int64_t br, u64 &w0, u64 &w1)
{
    const int4v r = __builtin_nontemporal_load(reinterpret_cast<const int4v *>(d.codes + d.plane_off[k]) + br);
    w0 = (u64)(uint32_t)r.x | ((u64)(uint32_t)r.y << 32);
    w1 = (u64)(uint32_t)r.z | ((u64)(uint32_t)r.w << 32);
}
// 3x3 blocks: field fd = shift | width << 8 | word << 16 of the 64-bit words (w0, w1), sign-extended
__device__ __forceinline__ int dict_field(u64 w0, u64 w1, int fd)
{
    const u64 w = (fd >> 16) ? w1 : w0;
    const int sh = fd & 255, wd = (fd >> 8) & 255;
    return (int)((long long)(w << (64 - sh - wd)) >> (64 - wd));
}
// 3x3 blocks, ONE field layout for every class (DictDev::uniform3 = SPLIT: which entries sit in the second word -- 1: 5..8,
// 2: 4..8, 3: 4 and 6..8, whichever lets the widest need of any class per entry fit twice 64 bits; the 256^3 slabs: 3).
// The word is a compile-time choice, the shifts are kernel arguments: field to the top of the word, then a sign-extending
// shift of the high half -- two instructions where the per-class form takes a dozen.
template <int SPLIT>
__device__ __forceinline__ constexpr bool dict3_in_w1(int e)
{
    return SPLIT == 1 ? e >= 5 : SPLIT == 2 ? e >= 4 : (e == 4 || e >= 6);
}
template <int SPLIT>
__device__ __forceinline__ int dict_field3u(u64 w0, u64 w1, const DictArgs &d, int e)
{
    const u64 w = dict3_in_w1<SPLIT>(e) ? w1 : w0;
    return (int)((long long)(w << d.u3l[e]) >> 32) >> d.u3r[e];
}
// 2x2 blocks: the fields never straddle the two 32-bit halves of the block's word (set-up packs them so), and the
// descriptor is laid out for the hardware's bit-field extract: fd = offset (bits 0-4) | width << 8 | (high half ? 1 << 31 : 0).
// Select the half, extract: 4 integer instructions per value where two 64-bit shifts and their bookkeeping took 12
// (the kernel spent 15.6 us of VALU issue per SIMD at 1024^2: SQ_ACTIVE_INST_VALU).
__device__ __forceinline__ int dict_field2(u64 w, int fd)
{
    const int lo = (int)(uint32_t)w, hi = (int)(uint32_t)(w >> 32);
    const int mask = fd >> 31;                       // all ones: the high half
    const int src = (hi & mask) | (lo & ~mask);      // (one v_bfi_b32)
    return __builtin_amdgcn_sbfe(src, (unsigned)fd, (unsigned)(fd >> 8));   // v_bfe_i32 reads 5 bits of offset and of width
}
// A class whose four widths allow no packing inside the halves (2048^2: 20 + 14 + 14 + 13 bits) gets its fields back to
// back in the 64 bits; one that lies ACROSS the halves carries kDictAcross and its shift from bit 0 of the word in bits
// 0-5.  strad (uniform): the layout holds such fields at all -- the plain kernels honour it, the pipelined product and the
// resident cycle are not launched on such a layout.
constexpr int kDictAcross = 1 << 30;
__device__ __forceinline__ int dict_field2(u64 w, int fd, bool strad)
{
    if (strad && (fd & kDictAcross)) {
        const int sh = fd & 63, wd = (fd >> 8) & 31;
        return (int)((long long)(w << (64 - sh - wd)) >> (64 - wd));
    }
    return dict_field2(w, fd);
}
// value = base + k * 2^g: both terms exact, the sum representable (it is the stored value): exact under any rounding
__device__ __forceinline__ double dict_decode(int code, double2 bs)
{
    return __builtin_fma((double)code, bs.y, bs.x);
}
// address of the word(s) of position k of block row br (set-up kernels, the resident kernel's LDS copy)
template <int BS>
__device__ __forceinline__ u64 *dict_word_ptr(const DictArgs &d, unsigned char *codes, int k, int64_t br)
{
    if (BS == 3) return reinterpret_cast<u64 *>(codes + d.plane_off[k]) + 2 * br;
    const int p = k >> 1;
    if (2 * p + 1 < d.kmax) return reinterpret_cast<u64 *>(codes + d.plane_off[p]) + 2 * br + (k & 1);
    return reinterpret_cast<u64 *>(codes + d.plane_off[p]) + br;
}

inline DictArgs dict_args(const DictDev &A, int *grid)
{
    DictArgs d{};
    d.tid = A.tid.p;
    d.tab = A.tab.p;
    d.cls = A.cls.p;
    d.fld = A.fld.p;
    d.codes = A.codes.p;
    d.zpad = A.zpad.p;
    d.strad = A.straddle ? 1 : 0;
    for (int e = 0; e < 9; ++e) {
        d.u3l[e] = A.u3l[e];
        d.u3r[e] = A.u3r[e];
    }
    for (int e = 0; e < 4; ++e) d.uw[e] = A.uniform ? A.uw[e] : 0;
    d.nbrows = A.nbrows;
    d.ntype = A.ntype;
    d.nclass = A.nclass;
    d.kmax = A.kmax;
    d.nchunks = (A.nbrows + kDictChunk - 1) / kDictChunk;
    // large systems: a few chunks per workgroup (the table copy is paid once per workgroup), about 2048 workgroups
    static const int wgs2 = [] { const char *e = getenv("SPK_DICT2_WGS"); return e && atoi(e) > 0 ? atoi(e) : kDict2Wgs; }();   // (developer knob)
    d.chunks_per_wg = std::max(1, d.nchunks / (A.bs == 2 && A.kmax == 9 ? wgs2 : 2048));
    const int cpx = (d.nchunks + 7) / 8;
    d.chunks_per_xcd = (cpx + d.chunks_per_wg - 1) / d.chunks_per_wg * d.chunks_per_wg;
    d.tab_ints = ((A.ntype + 1) & ~1) + 2 * A.ntype * A.kmax;
    d.cls_off = (4 * d.tab_ints + 15) & ~15;
    d.fld_off = d.cls_off + 16 * (A.nclass + 1) * A.bs * A.bs;
    for (int k = 0; k < kDictMaxK; ++k) d.plane_off[k] = A.plane_off[k];
    *grid = 8 * (d.chunks_per_xcd / d.chunks_per_wg);
    return d;
}

}  // namespace k
}  // namespace spk
