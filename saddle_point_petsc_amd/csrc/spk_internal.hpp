// spk_internal.hpp -- private declarations shared by the libspk.so sources.
//
// Layering (MI355X-first, not PETSc's):
//   spk_api.cpp      C ABI entry points, argument checks, error strings
//   spk_solver.cpp   device-resident FGMRES: the host only ENQUEUES a restart
//                    cycle; Hessenberg/Givens/convergence live on the device
//   spk_k_*.hip      hand-written gfx950 kernels (HBM-bound, FP64, no MFMA): spmv (+ set-up, FP32 sweeps), vec (MDot,
//                    MAXPY, PC pieces), krylov (scalar work, head kernels), iter (fused iteration), comm (peer-store
//                    launches); spk_device.hpp = the device helpers they share
//   spk_comm.cpp     collectives: peer-store windows over xGMI on top of RCCL (one process per
//                    GPU); host-callback and in-process backends for 1-GPU rehearsals
//   spk_partition.cpp host-only row-slab split + halo plan
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include <hip/hip_ext.h>

#include "../../include/spk.h"

namespace spk {

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
struct Error {
    int code;
    std::string msg;
};
[[noreturn]] void fail(int code, const char *fmt, ...);

#define SPK_HIP(call)                                                                  \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess)                                                          \
            ::spk::fail(SPK_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                           \
    } while (0)

// ---------------------------------------------------------------------------
// device memory
// ---------------------------------------------------------------------------
template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    // allocates `count` (+pad) elements, zero-filled
    void alloc(size_t count, size_t pad = 0)
    {
        release();
        n = count;
        size_t bytes = (count + pad) * sizeof(T);
        if (bytes == 0) bytes = sizeof(T);
        SPK_HIP(hipMalloc((void **)&p, bytes));
        SPK_HIP(hipMemset(p, 0, bytes));
    }
    // allocates without clearing (the caller writes every element it reads; only the pad is zeroed)
    void alloc_raw(size_t count, size_t pad = 0)
    {
        release();
        n = count;
        size_t bytes = (count + pad) * sizeof(T);
        if (bytes == 0) bytes = sizeof(T);
        SPK_HIP(hipMalloc((void **)&p, bytes));
        if (pad) SPK_HIP(hipMemset(p + count, 0, pad * sizeof(T)));
    }
    void upload(const T *h, size_t count, size_t pad = 0)
    {
        alloc(count, pad);
        if (count) SPK_HIP(hipMemcpy(p, h, count * sizeof(T), hipMemcpyHostToDevice));
    }
};

// CSR block on the device, prepared for the row-tiled stream kernel.
struct CsrDev {
    int32_t nrows = 0, ncols = 0;
    int64_t nnz = 0;
    DevBuf<int32_t> rowptr, colidx;
    DevBuf<double> val;
    // stream tiling: tile t covers rows [tile_row[t], tile_row[t+1])
    DevBuf<int32_t> tile_row;
    int32_t ntiles = 0;
};

// 2x2-blocked copy of the diagonal part of A (dof-2 DMDA matrices: rows 2k, 2k+1 share
// their column pattern, columns come in pairs 2c, 2c+1 -- PETSc exploits the same fact
// with BAIJ / inodes).  One int32 block column + 4 values per block: 36 B per 4 stored
// non-zeros instead of 48.  Values are kept as two planes so every load is 16 B/lane.
struct BcsrDev {
    int32_t nbrows = 0;
    int64_t nblocks = 0;
    DevBuf<int32_t> browptr, bcol;
    DevBuf<double> vtop, vbot;  // (a00,a01) and (a10,a11) per block
    DevBuf<float> vtop32, vbot32;  // the same in single precision, for the FP32 inner sweeps (spk_pc_setup with sweeps)
    DevBuf<int32_t> tile_brow;
    DevBuf<int32_t> tile_desc;  // per tile {first block row, end block row, first block, end block} (one load instead of a chain)
    int32_t ntiles = 0;
    bool ok = false;
    bool long_rows = false;  // a block row longer than one tile exists (the two-launch iteration does not take those)
    // "BA" iteration kernel (MAXPY + next SpMV in one launch, neighbour flags): workgroup rho (row order) owns
    // ba_tb consecutive tiles; ba_nbr[2 rho], [2 rho + 1] = first / last workgroup whose rows its columns touch
    bool ba_ok = false;
    int32_t ba_slots = 0, ba_tb = 0, ba_chunk = 0;
    DevBuf<int32_t> ba_nbr, ba_wt;   // ba_wt[2 rho], [2 rho + 1] = its tiles [t0, t1)
};

// 3x3-blocked copy (dof-3 grids): one block column index per nine values; plane k (stride ldp) = entry (k / 3, k % 3) of
// every block.  v32: the same planes in single precision for the FP32 inner sweeps (spk_pc_setup with inner sweeps).
struct Bcsr3Dev {
    int32_t nbrows = 0;
    int64_t nblocks = 0, ldp = 0;
    DevBuf<int32_t> browptr, bcol, tile_brow;
    DevBuf<double> v;
    DevBuf<float> v32;
    int32_t ntiles = 0;
    bool ok = false;
};

// Row types + deviation codes of a blocked matrix (bs = 2 or 3; spk_k_dict.hip).  The reference assembles the same
// element matrix for every element of a uniform grid (/root/reference/src/Discretization.c:25, :293-332) -- up to the
// rounding of a Jacobian formed from node coordinates (:96-128): the entries of A scatter by a few hundred ulps around a
// handful of ideal values.  So A has a few dozen block CLASSES (blocks equal up to that noise) and a few dozen ROW TYPES
// (sequences of (column offset, class) along a block row).  Stored per block row: one 16-bit type; per stored value: a
// two's-complement bit field k of the width its class entry needs with  value = base[class][entry] + k 2^g[class][entry]
// exactly, a block's fields packed into one (2x2) / two (3x3) 64-bit words.
// The small tables sit in LDS.  A product streams x, y and ~1.8 B per stored non-zero instead of 9 (blocked) / 12
// (CSR) and forms the same products in the same order: bit-identical sums.  Found in the caller's CSR at KSPSetOperators
// (hashing on the device, then EVERY value decoded and compared bit by bit); a matrix that does not fit (too many classes
// or types, a row beyond kDictMaxK blocks, deviations that are not small multiples of one power of two or do not fit the
// words, tables beyond the LDS budget) keeps the plain blocked layout.
constexpr int kDictMaxK = 32;      // blocks per block row
namespace k { constexpr int kDictAcrossHost = 1 << 30; }   // = kDictAcross (spk_dict.hpp, device side)
struct DictDev {
    int bs = 0;
    int32_t nbrows = 0, ntype = 0, nclass = 0, kmax = 0;
    int64_t nblocks = 0;
    DevBuf<uint16_t> tid;          // type of every block row
    DevBuf<int32_t> tab;           // [ntype] lengths (padded to an even count), then ntype x kmax x {column offset, class}
    DevBuf<double> cls;            // (nclass + 1) x bs*bs x {base, 2^g}: the last one is the NULL class (base 0, scale 0: decodes to +0)
    DevBuf<int32_t> fld;           // (nclass + 1) x bs*bs bit fields of the codes (spk_dict.hpp: dict_field / dict_field2)
    DevBuf<double> zpad;           // zeros: the x a position beyond a row's length gathers in the pipelined product
    DevBuf<unsigned char> codes;   // bs = 2: one 64-bit word per block, positions 2p / 2p+1 side by side in plane p (16 B per block
                                   // row; an odd last position: 8 B); bs = 3: two words per block, plane k = position k
    int64_t plane_off[kDictMaxK] = {};
    int64_t code_bytes = 0;        // bytes of codes a product reads (the planes without their padding rows)
    int32_t lds_bytes = 0;         // tables as laid out in LDS
    // 2x2: every class has the SAME field layout (the widest need of any class per entry still fits: entries 0, 1 in the
    // low half of the word, 2, 3 in the high half): the product kernel extracts without reading the field table
    bool uniform = false;
    int32_t uw[4] = {1, 1, 1, 1};
    bool straddle = false;         // 2x2: some class has a field across the halves of its word (spk_dict.hpp)
    int uniform3 = 0;              // 3x3: one field layout for all classes; 1..3 = which entries sit in the second word
    int32_t u3l[9] = {}, u3r[9] = {};   // its shifts (dict_field3u)
    bool ok = false;
};

// Short-and-wide block (B: m rows x n_local cols) cut into column windows so
// that x is streamed once for all m rows.
struct WideDev {
    int32_t m = 0, ncols = 0, nwin = 0, win = 0;
    int64_t nnz = 0;
    DevBuf<int32_t> colidx;  // nnz, rows concatenated, ascending in a row
    DevBuf<double> val;
    DevBuf<int32_t> winptr;  // (nwin+1) x m : start of window w in row r
};

// ---------------------------------------------------------------------------
// peer-store windows (one-shot collectives over xGMI, spk_comm.cpp / spk_device.hpp, spk_k_comm.hip)
// ---------------------------------------------------------------------------
namespace k {
struct SendRanges;
constexpr int kPeerMax = 8;        // ranks of one node
constexpr int kArSlots = 4;        // all-reduce slots in flight (a rank is never more than one ahead)
constexpr int kArGranules = 128;   // 8-byte {seq, half} granules per rank and slot = 64 doubles
// One-shot all-reduce: every rank stores its values as tagged granules into EVERY rank's window
// (win[p] = rank p's window as mapped here) and sums what arrived in its own, in rank order.
// Device-side diagnostics of the peer-store backend (spk_comm_get_info): accumulated by ONE lane per
// collective / waiting workgroup.  stats[2k] = 100 MHz ticks between "my stores are issued" and "every
// rank's contribution has arrived", stats[2k+1] = number of such waits; k = kStatArDots (all-reduce in
// the finish of MDot), kStatArNorm (the one after MAXPY), kStatArOther (stand-alone launches),
// kStatHalo (ghost rows).
enum { kStatArDots = 0, kStatArNorm = 1, kStatArOther = 2, kStatHalo = 3, kStatCount = 8 };
struct PeerAR {
    int P, me;                     // P == 0: off
    uint32_t seq, timeout_ms;
    unsigned long long *win[kPeerMax];
    int32_t *err;                  // set to 1 by a poll that timed out
    unsigned long long *stats;     // nullptr: no accounting
    int kind;                      // kStatAr*
};
// Halo exchange in the same style: my segment for peer i goes to remote[i] (the place in that
// rank's staging window where it expects my rows), what I expect arrives in `mine`.
struct PeerHalo {
    int npeers;                    // 0: off
    uint32_t seq, timeout_ms;
    unsigned long long *remote[4];
    const unsigned long long *mine;
    int64_t send_off[5], recv_off[5];
    int32_t *err;
    unsigned long long *stats;
};
// The same exchange for segments that are bandwidth-bound (a node PLANE of a 3-D slab): the payload
// goes as plain doubles in chunks of kBulkChunk, each followed -- after a system-scope release -- by
// ONE flag store; the receiver waits on the flag of a chunk, then copies it out of its staging.
constexpr int kBulkChunk = 2048;
struct PeerBulk {
    int npeers;
    uint32_t seq, timeout_ms;
    double *rdata[4];                 // where my segment for peer i starts in its staging
    unsigned long long *rflag[4];     // that staging's flag of element 0 of my segment (one flag per chunk, at its first element)
    const double *mdata;              // my staging (this parity)
    const unsigned long long *mflag;
    int64_t send_off[5], recv_off[5];
    int32_t send_chunk0[5], recv_chunk0[5];  // prefix sums of the chunk counts
    int32_t *err;
    unsigned long long *stats;
};
}  // namespace k

// ---------------------------------------------------------------------------
// collectives
// ---------------------------------------------------------------------------
struct HaloPlan;
class Comm {
public:
    virtual ~Comm() {}
    virtual int rank() const { return 0; }
    virtual int size() const { return 1; }
    // called once the halo plan of the (0,0) block is known (collective)
    virtual void setup_halo(int32_t n_ghost, const std::vector<int> &peers, const std::vector<int64_t> &send_off,
                            const std::vector<int64_t> &recv_off)
    { (void)n_ghost; (void)peers; (void)send_off; (void)recv_off; }
    // all-reduce performed INSIDE the reducing kernel's finish (peer-store backend): returns the
    // window set for the next all-reduce of `count` values, or P == 0 when the backend cannot
    // (the caller then calls allreduce_sum after the kernel)
    virtual k::PeerAR fused_allreduce(int count, int kind = k::kStatArOther) { (void)count; (void)kind; return k::PeerAR{}; }
    // what the first multi-GPU run needs to be diagnosable from its own output (spk_comm_get_info): a plain backend
    // reports how many all-reduces / halo exchanges it carried; the peer-store wrapper overrides with its own split
    virtual void info(spk_comm_info *out)
    {
        out->n_allreduce_inner = n_ar_calls_;
        out->n_halo_inner = n_ex_calls_;
    }
    // halo exchange performed INSIDE the kernel that produces the vector (contiguous send ranges
    // only): fills the peer fields of sr for the next exchange and returns true, or returns false
    // (the caller then calls exchange())
    virtual bool fused_halo(k::SendRanges &sr, double *xghost) { (void)sr; (void)xghost; return false; }
    // The resident restart-cycle kernel runs a whole cycle's collectives inside ONE launch: n_ar all-reduces and n_halo
    // halo exchanges with consecutive sequence numbers.  Fills the window set of the first all-reduce and the send ranges
    // of the first two exchanges (the two staging parities; sr0 / sr1 come in as copies of the context's ranges) and
    // reserves the numbers; false when this backend cannot (the caller keeps the launch-by-launch form)
    virtual bool resident_plan(int n_ar, int n_halo, k::PeerAR &ar, k::SendRanges &sr0, k::SendRanges &sr1, double *xghost)
    { (void)n_ar; (void)n_halo; (void)ar; (void)sr0; (void)sr1; (void)xghost; return false; }
    // throws SPK_ERR_COMM when a device-side wait of this backend has timed out (call after a sync)
    virtual void check(hipStream_t s) { (void)s; }
    virtual const char *name() const { return "self"; }
    // true when the Krylov all-reduces ride in the finish of the reducing kernels (fused_allreduce returns windows)
    virtual bool fuses() const { return false; }
    // device address of the backend's sticky error word (nullptr: none): the solver reads it with its own per-cycle
    // state copy and calls check() only when it is set
    virtual const int32_t *error_dev() const { return nullptr; }
    // in-place sum over ranks of `count` doubles in device memory, stream-ordered
    virtual void allreduce_sum(double *dev, int count, hipStream_t s) { (void)dev; (void)count; (void)s; }
    // exchange of packed halo segments: sendbuf[send_off[p]..] -> peer p,
    // peer p's segment lands in recvbuf[recv_off[p]..]
    virtual void exchange(const double *sendbuf, const std::vector<int> &peers,
                          const std::vector<int64_t> &send_off, double *recvbuf,
                          const std::vector<int64_t> &recv_off, hipStream_t s)
    { (void)sendbuf; (void)peers; (void)send_off; (void)recvbuf; (void)recv_off; (void)s; }
    // host-side exchange of small setup data (sizes, index lists)
    virtual void host_allgather(const void *in, void *out, size_t bytes_each) { memcpy_self(in, out, bytes_each); }
    virtual void host_allgatherv(const void *in, size_t bytes_in, std::vector<std::vector<char>> &out)
    {
        out.assign(1, std::vector<char>((const char *)in, (const char *)in + bytes_in));
    }
protected:
    static void memcpy_self(const void *in, void *out, size_t b);
    int64_t n_ar_calls_ = 0, n_ex_calls_ = 0;
};
Comm *make_self_comm();
Comm *make_rccl_comm(int rank, int nranks, const void *id128, int device);
Comm *make_local_comm(spk_local_group *grp, int rank);
Comm *make_host_comm(int rank, int nranks, const spk_host_comm &cb);
// wraps `inner` (kept for set-up traffic and as the fallback) with the peer-store backend when every
// rank can map every other rank's window; returns `inner` itself otherwise (*why says why)
Comm *make_peer_comm(Comm *inner, int device, std::string *why);

// ---------------------------------------------------------------------------
// host-side partition results
// ---------------------------------------------------------------------------
// uninitialised host array (std::vector would zero hundreds of MB on one thread first)
template <class T>
struct HostBuf {
    std::unique_ptr<T[]> p;
    size_t n = 0;
    void alloc(size_t count)
    {
        p.reset(new T[count ? count : 1]);
        n = count;
    }
    T *data() { return p.get(); }
    const T *data() const { return p.get(); }
    size_t size() const { return n; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
};
// fn(begin, end, thread) over [0, n) on up to `hardware threads` host threads
void parallel_for(int64_t n, const std::function<void(int64_t, int64_t, int)> &fn, int max_threads = 0);

struct SplitCsr {
    HostBuf<int32_t> d_rowptr, d_colidx, o_rowptr, o_colidx;
    HostBuf<double> d_val, o_val;
    std::vector<int32_t> garray;
    bool bad_column = false;  // a column outside [0, ncols_global)
    int32_t bad_value = 0;
};
void split_csr(int64_t row_begin, int32_t nrows_local, const int32_t *rowptr,
               const int32_t *colidx, const double *val, SplitCsr &out, int64_t ncols_global = -1);

// ---------------------------------------------------------------------------
// Krylov state that lives in device memory (one per context)
// ---------------------------------------------------------------------------
struct KrylovState {
    int32_t its, reason, done, loc_done;
    int32_t max_it, restart, hapend, skip_refine;
    // what the kernels of an ITERATION are gated by: set together with `done`, and alone when the
    // rest of the restart cycle is to be skipped (single-reduction mode: a convergence seen by the
    // recurrence is only tentative and is confirmed on the true residual of the restart)
    int32_t skip_iter, guess_nonzero;
    double rnorm, rnorm0, ttol, abstol, dtol, bnorm;
    double rtol;    // -ksp_rtol: ttol is fixed at iteration 0 (KSPConvergedDefault)
    double cnorm0;  // the norm the relative and the divergence tests refer to (||b|| or the initial residual)
    double inv_tt;  // 1/||w|| of the last orthogonalised vector (or 1/||r||)
    double tt;
};

// kernel launch wrappers (spk_k_*.hip)
namespace k {
constexpr int kMaxNv = 64;       // max vectors in one mdot/maxpy launch; restart <= 62 takes the fused kernels
constexpr int kBigNv = 1024;     // restart lengths up to kBigNv - 2 run Gram-Schmidt in chunks of <= 40 vectors
constexpr int kPartialLd = 64;   // leading dimension of block partials
constexpr int kMaxBlocks = 2048; // cap for grid-stride vector kernels
constexpr int kBTile = 512;      // 2x2 blocks per tile of the blocked SpMV kernels

void build_tiles(const int32_t *rowptr, int32_t nrows, std::vector<int32_t> &tile_row);
// arms the block-partials buffer of the cross-workgroup finish (spk_k_vec.hip)
void arm_partials(double *p, size_t n, hipStream_t s);
// test hook: a reduction with a partial that never arrives (see spk_debug_finish_timeout)
struct Finish;
void finish_probe(const Finish &f, hipStream_t s);

// where a reducing kernel leaves its result: block partials (armed with the sentinel of the
// "last block reduces" protocol, spk_device.hpp) and the output slot
struct Finish {
    double *partials;
    double *out;
    PeerAR ar;  // P != 0 (mdot, maxpy only): the finishing workgroup also sums over the ranks
    // sticky execution-error word of the context (bit 0: a partial never arrived within fin_ticks);
    // fgmres / the API calls turn it into SPK_ERR_HIP and re-arm the partials
    int32_t *err = nullptr;
    uint32_t fin_ticks = 400000000u;  // bound of the reducer's wait, 100 MHz ticks (4 s)
};

// off-rank part folded into the SpMV epilogue (rowptr over ALL local rows; nullptr: none)
struct OffDiag {
    const int32_t *rowptr, *colidx;
    const double *val, *xg;
};
// packed halo values written by the producer of a vector instead of a gather launch: up to four
// contiguous row ranges (slab partitions send whole node lines / planes)
struct SendRanges {
    int n;  // 0: none
    int32_t r0[4], len[4], off[4];
    double *buf;
    // peer-store backend (Comm::fused_halo): the rows go as granules straight into the neighbours'
    // staging instead of buf, and extra workgroups at the end of the grid wait for MY ghost rows and
    // unpack them into xghost -- the halo exchange costs no launch of its own
    int peer;  // 0: off
    uint32_t seq, timeout_ms;
    unsigned long long *remote[4];
    const unsigned long long *mine;
    int32_t nrecv;
    double *xghost;
    int32_t *err;
    unsigned long long *stats;
};

// y = A x  (+ Bt-rows * lam when bt != nullptr; y += when accumulate); CSR stream kernel.
// rider: one extra workgroup of the launch runs a pending Givens step beside the row tiles (GivensRider below)
struct GivensRider;
void spmv(const CsrDev &A, const double *x, double *y, const CsrDev *bt, const double *lam,
          const int32_t *done, hipStream_t s, bool accumulate = false, const OffDiag *od = nullptr,
          const GivensRider *rider = nullptr);
// same product from the 2x2-blocked copy (bitwise the same sums: CSR order is kept)
void spmv_bcsr(const BcsrDev &A, const double *x, double *y, const CsrDev *bt, const double *lam,
               const int32_t *done, hipStream_t s, bool accumulate = false, const OffDiag *od = nullptr,
               const GivensRider *rider = nullptr);
void build_btiles(const int32_t *browptr, int32_t nbrows, std::vector<int32_t> &tile_brow);
// ... and from the 3x3-blocked copy (same sums again)
constexpr int kB3Tile = 256;  // blocks per tile: one per thread
void spmv_bcsr3(const Bcsr3Dev &A, const double *x, double *y, const CsrDev *bt, const double *lam,
                const int32_t *done, hipStream_t s, bool accumulate = false, const OffDiag *od = nullptr,
                const GivensRider *rider = nullptr);
void build_b3tiles(const int32_t *browptr, int32_t nbrows, std::vector<int32_t> &tile_brow);
// ... and from row types + deviation codes (spk_k_dict.hip; same sums once more)
constexpr int kDictMaxPat = 1024, kDictMaxBlk = 1024, kDictSlots = 8192, kDictLdsMax = 48 * 1024;
void spmv_dict(const DictDev &A, const double *x, double *y, const CsrDev *bt, const double *lam, const int32_t *done,
               hipStream_t s, bool accumulate = false, const OffDiag *od = nullptr, const GivensRider *rider = nullptr);
void jacobi_sweep_f32_dict(const DictDev &A, const float *d32, float omega, const float *x32, const float *yin, float *yout,
                           const int32_t *done, hipStream_t s);
// set-up passes: classes by hashing (keys/rep: kDictSlots entries, zero / INT32_MAX before the call; slot: the table
// slot of every item; ctl[0] = number of distinct keys, ctl[1] = 1 when there are too many)
void dict_hash_blocks(int bs, const double *v0, const double *v1, int64_t ldp, int64_t nblocks, unsigned long long *keys,
                      int32_t *rep, int32_t *slot, int32_t *ctl, int maxkeys, hipStream_t s);
// cls[(id, e)] = {entry e of block rep_of_id[id], 1}; slot[q] -> class number; gexp / dmax: per (class, entry) the finest
// bit and the largest magnitude of value - base (INT32_MAX / 0 before the call); *bad: a deviation that is not exact
// Measurement (spk_debug_time_products): a product launch takes the kernel's OWN start / stop time stamps into a pair of events
// (hipExtLaunchKernelGGL) when the calling thread has set them -- a_mult around the iteration's product launch
struct LaunchTimer {
    hipEvent_t start = nullptr, stop = nullptr;
};
LaunchTimer &launch_timer();
#define SPK_LAUNCH_PRODUCT(kern, grid, block, lds, stream, ...)                                                                 \
    do {                                                                                                                        \
        const spk::k::LaunchTimer &lt_ = spk::k::launch_timer();                                                                \
        if (lt_.start) hipExtLaunchKernelGGL(kern, grid, block, (uint32_t)(lds), stream, lt_.start, lt_.stop, 0, __VA_ARGS__);  \
        else hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                                                   \
    } while (0)
// spk_k_dict3.hip: the pipelined forms for 3x3 blocks (27-point row types, DictDev::uniform3); false: not applicable
bool spmv_dict3(const DictDev &A, const double *x, double *y, const CsrDev *bt, const double *lam, const int32_t *done, hipStream_t s,
                bool accumulate, const OffDiag *od, const GivensRider *rider);
bool jacobi_sweep_f32_dict3(const DictDev &A, const float *d32, float omega, const float *x32, const float *yin, float *yout,
                            const int32_t *done, hipStream_t s);
void dict_class_stats(int bs, const double *v0, const double *v1, int64_t ldp, int64_t nblocks, const int32_t *rep_of_id, int nid,
                      const int32_t *slot2id, int32_t *slot, double *cls, int32_t *gexp, unsigned long long *dmax, int32_t *bad,
                      hipStream_t s);
void dict_hash_rows(const int32_t *browptr, const int32_t *bcol, const int32_t *blkid, int32_t nbrows, unsigned long long *keys,
                    int32_t *rep, int32_t *slot, int32_t *ctl, int maxkeys, int kmax, hipStream_t s);
void dict_fill_rows(const int32_t *browptr, const int32_t *bcol, const int32_t *blkid, int32_t nbrows, const int32_t *rep_of_id,
                    int nid, int kmax, const int32_t *slot2id, const int32_t *slot, int32_t *tab, uint16_t *tid, int32_t *bad,
                    hipStream_t s);
// writes the code planes of A (tables, plane offsets and widths set), then decodes every value and compares its bits
void dict_encode_verify(const DictDev &A, const int32_t *browptr, const int32_t *bcol, const int32_t *blkid, const double *v0,
                        const double *v1, int64_t ldp, int32_t *bad, hipStream_t s);
void bcsr3_fill(const int32_t *rp, const int32_t *ci, const double *va, int nbr, int32_t *browptr, int32_t *bcol, double *v,
                int64_t ldp, int32_t *fail, hipStream_t s);
void jacobi_sweep_f32_b2(const BcsrDev &A, const float *d32, float omega, const float *x32, const float *yin, float *yout,
                         const int32_t *done, hipStream_t s);
void jacobi_sweep_f32_b3(const Bcsr3Dev &A, const float *d32, float omega, const float *x32, const float *yin, float *yout,
                         const int32_t *done, hipStream_t s);
// KSPSetOperators on the device: count off-rank entries per row, exclusive scan, split, 2x2 blocking
void csr_count_off(const int32_t *rowptr, const int32_t *colidx, int nrows, int64_t lo, int64_t hi, int64_t ncols, int32_t *cnt,
                   int32_t *bad, hipStream_t s);
void exclusive_scan_i32(const int32_t *in, int64_t n, int32_t *out, int32_t *scratch, hipStream_t s);  // out[0..n]; scratch: n/2048 + 2 ints
void csr_split(const int32_t *rowptr, const int32_t *colidx, const double *val, int nrows, int64_t lo, int64_t hi,
               const int32_t *orp, int32_t *d_rowptr, int32_t *d_col, double *d_val, int32_t *o_col, double *o_val, hipStream_t s);
void bcsr_fill(const int32_t *rp, const int32_t *ci, const double *va, int nbr, int32_t *browptr, int32_t *bcol, double *vtop,
               double *vbot, int32_t *fail, hipStream_t s);
// f.out[r] = B_r . x, r < m
void wide_dot(const WideDev &B, const double *x, const Finish &f, const int32_t *done, hipStream_t s,
              const int32_t *rowmap = nullptr);
// shat[r] = sum_k B_rk^2 dinv[col_k], one wave per CSR row
void schur_diag_rows(const CsrDev &B, const double *dinv, double *shat, hipStream_t s);
// same with x replaced by x .* dinv (the B D x0 step of the Schur PC, no stored D x0)
void wide_dot_jacobi(const WideDev &B, const double *x, const double *dinv, const Finish &f,
                     const int32_t *done, hipStream_t s);
// dense[col] = val*dinv[col] over entries [k0,k1) of B (dinv == nullptr: dense[col] = 0)
void scatter_row(const int32_t *colidx, const double *val, int k0, int k1, const double *dinv,
                 double *dense, hipStream_t s);
// out[i] = sum_r slots[r*ld + i] in rank order (local-group all-reduce)
void sum_slots(const double *slots, int nslots, int ld, int count, double *out, hipStream_t s);
// f.out[i] = V_i . w for i < nv, then V2_i . w for i < nv2 (second slab, same stride), then w.w
void mdot(const double *V, int64_t ldv, int nv, const double *w, int64_t n, int64_t n_dot,
          const Finish &f, const int32_t *done, hipStream_t s, const double *V2 = nullptr, int nv2 = 0,
          int v2_split = 0);
// single-reduction Gram-Schmidt: scalars derived by workgroup 0 of the MAXPY kernel
struct PythArgs {
    int m;               // < 0: off
    const double *dots;  // [h_0..h_{nv-1}, q_0..q_{m-1}, w.w]  (reduced)
    double *tb;          // (restart+2) x 8: B D v_i per basis vector
    double *nrm_out;     // [||w'||^2, B D w' (m)]
};
// w += sign * sum_i a[i] * V_i ; f.out[0] = ||w_new||^2 over the first n_dot entries
// (f.out == nullptr: no norm)
// with bd != nullptr also f.out[1+r] = sum_i bd[i*MP + r] * w_new[i], i < n_bd (fused Schur path)
void maxpy(const double *V, int64_t ldv, int nv, const int32_t *nv_dev, const double *a,
           double coef_sign, double *w, int64_t n, int64_t n_dot, const Finish &f,
           const int32_t *done, hipStream_t s, const double *bd = nullptr, int64_t ldb = 0, int64_t n_bd = 0, int m = 0,
           double *w1side = nullptr, const PythArgs *pyth = nullptr, int bd_packed = 0);
void build_bd(const CsrDev &Bt, const double *dinv, int m, int64_t ldb, double *bd, hipStream_t s);
// rows 2q, 2q+1 of B D interleaved by parity into one plane each (bd_packed = 1 in the kernels that
// stream them); *bad != 0: the rows do not have that structure
void pack_bd(const double *bd, int64_t ldb, int64_t n, int m, double *bdp, int32_t *bad, hipStream_t s);
// (sa != nullptr: x = sa - sb is formed and stored in the same pass)
void sqnorm_bd(double *x, int64_t n, int64_t n_dot, const double *bd, int64_t ldb, int64_t n_bd, int m,
               double *w1side, const Finish &f, const int32_t *done, hipStream_t s, const double *sa = nullptr,
               const double *sb = nullptr);
// x *= *alpha_dev
void scale_dev(double *x, int64_t n, const double *alpha_dev, const int32_t *done, hipStream_t s);
// y = a*x + b*y with host scalars (b = 0: y = a*x without reading y)
void axpby(double a, const double *x, double b, double *y, int64_t n, const int32_t *done, hipStream_t s);
// f.out[0] = x.x over the first n_dot entries
void sqnorm(const double *x, int64_t n_dot, const Finish &f, const int32_t *done, hipStream_t s);
// x[0..n) = sa - sb and f.out[0] = x.x over the first n_dot entries, one pass
void sqnorm_sub(const double *sa, const double *sb, double *x, int64_t n, int64_t n_dot, const Finish &f, const int32_t *done,
                hipStream_t s);
// gather x[idx[i]] -> out[i]
void gather(const double *x, const int32_t *idx, int64_t n, double *out, const int32_t *done, hipStream_t s);
// peer-store collectives (stand-alone launches; the fused forms live in the reducing kernels)
void peer_allreduce(const PeerAR &a, double *buf, int count, hipStream_t s);
void peer_allreduce_loopback(const PeerAR &a, double *buf, int count, hipStream_t s);  // test hook: workgroup r = rank r
void peer_exchange(const PeerHalo &h, const double *sendbuf, double *recvbuf, hipStream_t s);
void peer_exchange_bulk(const PeerBulk &h, const double *sendbuf, double *recvbuf, hipStream_t s);
// Jacobi / Schur pieces
void jacobi(const double *dinv, const double *x, double *y, int64_t n, const int32_t *done, hipStream_t s);
void extract_diag_inv(const CsrDev &A, double *dinv, hipStream_t s);
// y0 = dinv .* (x0 - Bt y1)            (mode 0, UPPER)
// y0 = dinv .* x0 - dinv .* (Bt y1)    (mode 1, FULL third step, recomputing D x0)
// scratch != nullptr and Bt tiled (general blocks): Bt y1 by the stream kernel into scratch, then the combination
void bt_update(int mode, const CsrDev &Bt, const double *dinv, const double *x0, const double *y1,
               double *y0, const int32_t *done, hipStream_t s, double *scratch = nullptr);
// FP32 inner solve (damped-Jacobi Richardson sweeps on the diagonal block of A)
void cvt_scale_f32(const double *x, const float *d32, float omega, float *x32, float *y32, int64_t n,
                   const int32_t *done, hipStream_t s);                       // x32 = (float)x ; y32 = omega d32 x32
void jacobi_sweep_f32(const CsrDev &A, const float *val32, const float *d32, float omega, const float *x32,
                      const float *yin, float *yout, const int32_t *done, hipStream_t s);
void sweep_offdiag_f32(const CsrDev &Ao, const int32_t *rows, const float *d32, float omega, const double *xg,
                       float *y, const int32_t *done, hipStream_t s);        // y[row] -= omega d (Ao_row . xg)
void gather_f32(const float *x, const int32_t *idx, int64_t n, double *out, const int32_t *done, hipStream_t s);
// y = (double)y32  (mode 0)  |  y -= (double)y32  (mode 1)
void cvt_f32_out(const float *y32, double *y, int mode, int64_t n, const int32_t *done, hipStream_t s);
void cvt_vals_f32(const double *v, float *v32, int64_t n, hipStream_t s);
// c = x0 - Bt y1 (mode 2) | c = Bt y1 (mode 3): the vector handed to the inner solve
// small scalar kernels
void schur_y1(int fact, int m, const double *x1, const double *t, const double *shat, double *y1,
              const int32_t *done, hipStream_t s);
void copy_small(const double *src, double *dst, int n, const int32_t *done, hipStream_t s);

// Krylov scalar kernels (single wave)
// where krylov_cycle_begin reports the solve's state to the host (pinned, device-visible memory)
struct StateReport {
    KrylovState *host_state;
    int32_t *host_words;            // [0] reduction error word, [1] communicator error word
    const int32_t *errw, *commerr;  // device words (nullptr: none)
};
struct KrylovArrays {
    KrylovState *st;
    double *H, *cc, *ss, *rs, *nrs, *hcol, *hist, *tb;
    int32_t hist_cap, ldh;
    int32_t tentative;  // 1: the recurrence may end a cycle, only a true residual may end the solve
};
// where a reducer reports a partial that never arrived (execution failure, not a numerical one): the
// context's sticky error word; the bound of the wait in 100 MHz ticks
struct FinErr {
    int32_t *err;
    uint32_t ticks;
};
// The Givens step of iteration `loc` (Hessenberg column h[0..loc], ||w'||^2 in *nrm2), carried by workgroup 0 of a
// product launch: the serial chain runs beside the row tiles instead of at the tail of the MAXPY launch's reducer.
// The tiles of that launch read the gate words BEFORE the step may set them (they compute a product nobody uses when
// the step converges: no reduction in the launch depends on them).
struct GivensRider {
    KrylovArrays ka;
    int32_t loc;     // < 0: no rider
    const double *h;
    double *nrm2;
    double *sc;      // un-normalised basis: sc[loc + 1] = 1 / sqrt(*nrm2) is set first (nullptr: not)
    // fin_n > 0: *nrm2 is first REDUCED here from the fin_n partial rows the MAXPY launch published (+ the row behind
    // them: the multiplier entries' share), then all-reduced across ranks when ar.P != 0
    double *fin_partials;
    int32_t fin_n;
    FinErr fe;
    PeerAR ar;
};
// ---- two-launch iteration (spk_k_iter.hip, "Two-launch iteration") ----
// kernel A: w = s (A z~ + c~), v and z normalised on the way, h = V^T w and q = B D w from the tile epilogues
struct IterA {
    // 2x2-blocked matrix and its tiling
    const int32_t *browptr, *bcol;
    const double *vtop, *vbot;
    const int32_t *tile_brow;
    const int4 *tdesc;
    int ntiles, tiles_per_xcd, slots;  // slots: workgroups per XCD (iter_slots)
    OffDiag od;
    // vectors
    const double *zsrc;  // gathered: z~ (un-normalised, from kernel B) or Z_0 (first iteration of a cycle)
    double *zdst;        // Z_loc = s z~ (nrm2 != nullptr)
    double *vcur;        // V_loc: normalised in place (nrm2 != nullptr)
    double *w;           // V_{loc+1}: in c~ when acc, out w
    int acc;
    const double *nrm2;  // ||w'||^2 of the previous iteration: s = 1/sqrt(nrm2[0]); nullptr: s = 1, operands normalised
    const double *V;     // basis V_0 .. V_{nv-1} (V_{nv-1} = vcur)
    int64_t ldv;
    int nv;
    const double *bd;    // B D: m dense rows or m/2 parity-interleaved planes (nullptr with m = 0)
    int64_t ldb;
    int m, packed;
    int64_t nl;
    int lam_in_dot;      // this rank counts the m multiplier entries in inner products (rank 0)
    double *tb;          // B D v_i per basis vector, (restart+2) x 8
    double *wl_out;      // side copy of the multiplier entries of w for kernel B
    // finish: [h_0..h_{nv-1}, q_0..q_{m-1}]
    double *partials, *out;
    PeerAR ar;
    int32_t *err;
    uint32_t fin_ticks;
    // Givens step of the previous iteration (loc_prev < 0: none)
    KrylovArrays ka;
    int loc_prev;
    const double *dots_prev, *nrm_prev;
    const int32_t *done;
};
// kernel B: w' = w - V h, ||w'||^2, and the next iteration's PC / B^T product on the un-normalised w'
struct IterB {
    const double *V;
    int64_t ldv;
    int nv;
    const double *dots;  // reduced [h, q]
    double *tb;
    double *w;           // V_{loc+1}: in w, out w'
    const double *dinv, *bd;
    int64_t ldb;
    const double *shat, *gram;
    int fact;
    int64_t nl;
    int m, packed;
    double *zun;         // out: z~
    double *c;           // out: c~ = pre-load of the next product (V_{loc+2}); nullptr with m = 0
    const double *wl_in;
    int lam_in_dot;
    double *partials, *out;  // out[0] = ||w'||^2
    PeerAR ar;
    int32_t *err;
    uint32_t fin_ticks;
    SendRanges sr;
    int gmain;           // set by the launcher
    const int32_t *done;
    // un-normalised basis (opts.iteration_form = 5): sc != nullptr -- dots are RAW inner products of V~_i with w~, the
    // scale factors sc[i] = 1 / ||w'_i|| are applied to the scalars here (h_i = sc_i s_w dots_i, MAXPY coefficient
    // h_i sc_i, w = s_w w~); zun is Z_{loc+1} itself; the reducer stores the scaled Hessenberg column (hbuf).  sc[nv] and
    // the Givens step of THIS iteration follow in the rider of the next product launch (GivensRider).  No vector is ever
    // normalised, no pass exists for it.
    double *sc, *hbuf, *wl_out;
    KrylovArrays ka;
    int loc;
    int defer_fin;       // publish the partials of ||w'||^2 and leave: the rider of the next product launch reduces them
};
// dots = false: the SpMV / normalisation part alone (three-launch form; one tile per workgroup: slots = tiles_per_xcd)
void iter_spmv_mdot(const IterA &a, hipStream_t s, bool dots = true);
// BA: w' = s_w w~ - V~ (h .* sc) with ||w'||^2, z~ = M^-1 w', then -- behind neighbour flags instead of a kernel
// boundary -- the NEXT product w~ = A z~ + B^T y~ of the same row tiles; Givens of this iteration in the reducer.
// The basis stays UN-normalised (V~_i, Z~_i) with one scale factor per vector (sc[i] = 1 / ||w'_i||).
struct IterBA {
    const int32_t *browptr, *bcol;
    const double *vtop, *vbot;
    const int32_t *tile_brow;
    const int32_t *tdesc;  // per tile {first block row, end block row, first block, end block}
    int ntiles, tiles_per_xcd, slots, tb;
    int chunk;             // double2 entries per workgroup in phase B (<= 256)
    const int32_t *nbr, *wt;   // wt: per workgroup {t0, t1, first owner, last owner to wait for}
    uint32_t *flags;
    uint32_t seq;
    const double *V;       // basis, un-normalised from vector 1 on
    int64_t ldv;
    int nv;                // loc + 1
    const double *dots;    // reduced RAW [V~_i . w~ (nv), B D w~ (m)]
    double *sc;            // scale factors, (restart + 2)
    double *tb_;           // B D V~_i per basis vector, (restart + 2) x 8
    double *w;             // V_{loc+1}: in w~, out w'
    const double *dinv, *bd;
    int64_t ldb;
    const double *shat, *gram;
    int fact;
    int64_t nl;
    int m, packed;
    double *zout;          // Z_{loc+1} = z~ (gathered by the SpMV phase of this launch)
    double *wnext;         // V_{loc+2} = w~ of the next iteration
    const double *wl_in;
    double *wl_out;
    double *hbuf;          // scaled Hessenberg column of this iteration (nv values)
    int lam_in_dot, last;  // last: no SpMV phase (last iteration of a restart cycle)
    double *partials, *nrm_out;
    PeerAR ar;
    int32_t *err;
    uint32_t fin_ticks;
    KrylovArrays ka;
    int loc;
    const int32_t *done;
};
void iter_ba(const IterBA &p, hipStream_t s);
// Resident restart cycle (spk_k_resident.hip): ONE launch runs iterations 0 .. mk-1 of a cycle with the basis in registers
// (single rank, row-type layout with 2x2 blocks, <= 512 block rows per CU, restart <= 30, <= 4 planes of B D).  On entry
// V0 = v_0 (normalised), V1 = K z_0; Z_1.. are written; the Krylov scalars end up where krylov_cycle_end expects them.
struct ResidentArgs {
    int mk, m, packed, fact, lam_in_dot;
    int64_t nl, ld;
    const double *V0, *V1;
    double *Z;
    const double *dinv, *bd;
    int64_t ldb;
    const double *shat, *gram;
    double *P;             // resident_scratch_doubles() doubles
    KrylovArrays ka;
    double *sc_out;
    int32_t *err;
    uint32_t ticks;
    PeerAR ar;             // several ranks: Comm::resident_plan (P <= 1: single rank)
    SendRanges sr0, sr1;
    OffDiag od;
};
bool cycle_resident(const DictDev &A, int num_cus, ResidentArgs r, const int32_t *done, hipStream_t s);   // false: shape does not fit
int64_t resident_scratch_doubles(int num_cus, int mk);
bool resident_fits(const DictDev &A, int num_cus, int mk, int planes);   // planes: dense planes of B D the iteration streams
// block-column range of every tile of the blocked matrix (set-up of the BA kernel's neighbour lists)
void tile_col_range(const int32_t *browptr, const int32_t *bcol, const int32_t *tile_brow, int ntiles, int32_t *out, hipStream_t s);
int iter_maxpy_uhead(IterB b, hipStream_t s);   // returns the number of partial rows (GivensRider::fin_n)
int iter_slots(int tiles_per_xcd, int wg_per_cu);
void krylov_init(const KrylovArrays &ka, const spk_opts &o, const double *bnorm2, hipStream_t s);
void krylov_cycle_begin(const KrylovArrays &ka, const double *nrm2, hipStream_t s, double *tb = nullptr, int m = 0,
                        double *sc = nullptr, const StateReport *report = nullptr);
void krylov_givens(const KrylovArrays &ka, int loc, const double *dots, const double *nrm2, hipStream_t s);
// head of a fused Schur iteration: VecScale + PCApply + B^T part of MatMult in one pass, plus the
// previous iteration's Givens step in workgroup 0 (loc_prev < 0: none)
void fused_head(double *v, const double *nrm, const double *w1raw, const double *dinv, const double *bd, int64_t ldb,
                const double *shat, const double *gram, int fact, int64_t nl, int m, double *z, double *c,
                const KrylovArrays &ka, int loc_prev, const double *dots_prev, const int32_t *done, hipStream_t s,
                const SendRanges *sr = nullptr, int bd_packed = 0, double *wl_out = nullptr);
// single-reduction mode: MAXPY of iteration loc + head of iteration loc+1 + Givens of iteration loc
// in one pass (dots = reduced [h, B D w, w.w]; tb = B D v_i per basis vector, (restart+2) x 8)
void maxpy_head(const double *V, int64_t ldv, int nv, const double *dots, double *tb, double *nrm_out, double *w,
                const double *dinv, const double *bd, int64_t ldb, const double *shat, const double *gram, int fact,
                int64_t nl, int m, double *z, double *c, double *w1side, const double *wl_in, double *wl_out,
                const KrylovArrays &ka, int loc, const int32_t *done, hipStream_t s, const SendRanges *sr = nullptr,
                int bd_packed = 0);
// sc: y_i *= sc[i] (un-normalised Z); restart > kMaxNv - 2: the triangle stays in global memory
// pending: the Givens step of the cycle's last iteration, run first in the same launch (ka, loc, h, nrm2 are read)
void krylov_cycle_end(const KrylovArrays &ka, hipStream_t s, const double *sc = nullptr, int restart = 0,
                      const GivensRider *pending = nullptr);
// CGS refinement: decide (device side) whether the second pass runs, then fold its results
// (h2 into h, norm/traw of the refined vector over the first pass's)
void krylov_refine_decide(const KrylovArrays &ka, int loc, int mode, const double *dots, const double *nrm2,
                          double *dots2, hipStream_t s);
void krylov_refine_merge(const KrylovArrays &ka, int loc, double *dots, const double *dots2, double *nrm,
                         const double *nrm_b, int nn, hipStream_t s);
}  // namespace k

}  // namespace spk

// ---------------------------------------------------------------------------
// the context
// ---------------------------------------------------------------------------
struct spk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    mutable std::string err;
    std::string peer_why;  // why the peer-store backend is off (spk_comm_get_info)
    std::unique_ptr<spk::Comm> comm;

    // sizes
    int64_t n_global = 0, row_begin = 0;
    int32_t n_local = 0, m = 0, n_ghost = 0;
    int64_t ld = 0;  // padded vector length (n_local + m rounded up)

    // (0,0) block: diagonal part, compressed off-rank part
    spk::CsrDev Ad, Ao;
    spk::BcsrDev Ab;               // 2x2-blocked copy of Ad when the structure allows
    int spmv_format = 0;           // 0 = CSR stream kernel, 1 = 2x2 blocks (Ab), 2 = 3x3 blocks (Ab3)
    spk::Bcsr3Dev Ab3;             // 3x3-blocked copy (dof-3 grids)
    spk::DictDev Adict;            // row types + deviation codes over the blocked copy (uniform grids); Adict.ok: the products use it
    spk::DevBuf<int32_t> ao_rows;  // local row of each compressed Ao row
    spk::DevBuf<int32_t> ao_rowptr_full;  // Ao row pointers over all local rows (SpMV epilogue form)
    spk::k::SendRanges send_ranges{};     // halo rows as contiguous ranges, when they are
    spk::k::OffDiag offdiag() const { return spk::k::OffDiag{ao_rowptr_full.p, Ao.colidx.p, Ao.val.p, xghost.p}; }
    bool have_A = false, have_B = false;

    // constraint block: B^T (n_local x m) by rows, always; B itself
    //   m <= 8        : every row in column windows (WideDev) -- the long-row kernel, and the fused dense-plane path
    //   m  > 8        : a general sparse block: CSR by rows (Bc, the stream kernel) with up to 8 LONG rows (more than
    //                   kWideRowNnz local entries) taken out into the windowed form (B, rows listed in wide_rows)
    spk::WideDev B;
    spk::CsrDev Bt, Bc;
    bool b_general = false;
    int m_wide = 0;
    spk::DevBuf<int32_t> wide_rows;
    std::vector<int32_t> wide_rows_h;
    spk::DevBuf<double> tmpb;  // scratch vector of the general block's D x0 / B^T y1 products
    const double *bt_cached = nullptr;   // the multiplier vector whose B^T product tmpb holds (op_pc_apply -> op_mult)

    // halo plan
    std::vector<int> peers;
    std::vector<int64_t> send_off, recv_off;  // per peer offsets (+ total at end)
    spk::DevBuf<int32_t> send_idx;
    spk::DevBuf<double> send_buf, xghost;

    // preconditioner
    int pc_type = SPK_PC_NONE, schur_fact = SPK_SCHUR_FULL;
    // agreed over the ranks at KSPSetUp: every rank's local size is even / non-zero.  The head-kernel paths need an even
    // local size, and all ranks must take the SAME path (their collective sequences differ): decided collectively, not
    // from rank-local facts
    bool even_all = false, nonempty_all = false;
    bool res_fit_all = false;   // every rank's slab fits the resident restart-cycle kernel
    bool pc_ready = false;
    spk::DevBuf<double> dinv, shat, gram;  // n_local, m, m*m
    spk::DevBuf<double> bd;                // the m rows of B D as dense vectors of stride ld (fused Schur path)
    spk::DevBuf<double> bdpk;              // the same as m/2 parity-interleaved planes, when the rows allow
    bool bd_packed = false;
    // FP32 inner solve (0 sweeps = plain diag(A)^-1)
    int inner_sweeps = 0;
    double inner_omega = 1.0;
    spk::DevBuf<float> a32, d32, x32, y32a, y32b;
    spk::DevBuf<double> bigdots;   // Gram-Schmidt coefficients of restart lengths beyond the fused kernels' 62

    // scratch
    spk::DevBuf<double> partials;  // kMaxBlocks * kPartialLd
    spk::DevBuf<double> small;     // reduced scalars (256 doubles)
    spk::DevBuf<int32_t> errw;     // sticky device-side execution-error word (Finish::err)
    uint32_t fin_ticks = 400000000u;
    // measurement (spk_debug_time_products): HIP events on the solver's stream around the product launches of the
    // iterations (the launch with the Givens rider), read by spk_get_product_timing
    bool time_products = false;
    std::vector<hipEvent_t> tp_ev;   // pairs
    size_t tp_used = 0;
    spk::k::Finish fin(double *out) { return spk::k::Finish{partials.p, out, spk::k::PeerAR{}, errw.p, fin_ticks}; }
    spk::k::Finish fin(double *out, const spk::k::PeerAR &ar) { return spk::k::Finish{partials.p, out, ar, errw.p, fin_ticks}; }
    // throws SPK_ERR_HIP when a device-side wait of a cross-workgroup reduction has timed out (call after a
    // sync); re-arms the partials so that the context stays usable
    void check_device_error();
    spk::DevBuf<double> y1tmp, ttmp;

    // how the last spk_fgmres launched its iterations (spk_get_iteration_form): SPK_ITER_* actually run, -1 for the
    // step-by-step path (PCApply and MatMult as launches of their own); single-reduction mode beside it
    int last_form = -1, last_single = 0;

    // Krylov workspace (sized by restart)
    int ws_restart = -1;
    spk::DevBuf<double> V, Z, xsol, rhs, tmp;
    spk::DevBuf<double> zun;    // z~ of the two-launch iteration (un-normalised M^-1 w')
    spk::DevBuf<uint32_t> ba_flags;  // BA kernel: one 128-byte line per workgroup
    spk::DevBuf<double> ba_sc;       // scale factors of the un-normalised basis
    spk::DevBuf<double> res_P;       // resident cycle kernel: all-to-all buffer of the inner products
    int num_cus = 0;                 // compute units of the device (grid of the resident cycle kernel)
    uint32_t ba_seq = 0;
    spk::DevBuf<double> kry_d;  // H, cc, ss, rs, nrs, hcol, hist
    spk::DevBuf<spk::KrylovState> kst;
    spk::k::KrylovArrays ka{};

    // staging for host-pointer entry points
    spk::DevBuf<double> stage_x, stage_y;

    void ensure_scratch();
    void ensure_vectors();
    // pageable host memory -> device through two pinned staging buffers (KSPSetOperators)
    void upload_staged(void *dst, const void *src, size_t bytes);
    void *pin_state = nullptr;     // pinned landing place of the per-cycle state read-back (KrylovState + error words)
    hipEvent_t state_ev = nullptr; // ... and the event behind its copies
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
};

namespace spk {
// solver pieces used by the API layer (spk_solver.cpp)
// y (+)= Ad x in the layout the context holds: row types + codes, 2x2 / 3x3 blocks or CSR (same sums in all of them)
void a_mult(spk_ctx *c, const double *x, double *y, const CsrDev *bt, const double *lam, const int32_t *done, bool accumulate,
            const k::OffDiag *od, const k::GivensRider *rider = nullptr);
void op_mult(spk_ctx *c, const double *x, double *y, const int32_t *done, bool halo_done = false, bool reuse_bt = false);
void op_pc_apply(spk_ctx *c, const double *x, double *y, const int32_t *done);
void pc_setup(spk_ctx *c, int pc_type, int schur_fact);
void fgmres(spk_ctx *c, const double *b_dev, double *x_dev, const spk_opts &o, spk_result *res,
            double *history, int32_t history_cap);
void set_block(spk_ctx *c, int which, int64_t row_begin, int32_t nrows_local, int64_t ncols_global,
               const int32_t *rowptr, const int32_t *colidx, const double *val);
}  // namespace spk
