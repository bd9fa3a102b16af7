// spk_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the
// KSPSolve hot path.  Every kernel here is HBM-bandwidth bound (FP64 streams,
// ~0.17 flop/B): no MFMA, the levers are 16-byte coalesced loads, LDS-staged
// partial sums, wave-64 shuffle reductions, XCD-contiguous tile placement and
// fixed-order (deterministic) cross-workgroup reductions.
//
// What each kernel replaces inside PETSc (reached from
// /root/reference/src/SaddlePointProblem.c:70):
//   spmv_bcsr_kernel,    MatMult_SeqAIJ on the A block (+ MatMultTransposeAdd of B), the off-process
//   spmv_stream_kernel   part of MatMult_MPIAIJ in the epilogue of the same kernel
//   wide_dot_kernel      MatMult_SeqAIJ on the 4 long rows of B
//   mdot_kernel, _ws     VecMDot (+ the squared norm of w in the same pass)
//   maxpy_kernel         VecMAXPY (+ VecNorm of the result in the same pass)
//   scale/axpby/jacobi   VecScale, VecAXPY/WAXPY, PCApply_Jacobi
//   bt_update/schur_y1   the block steps of PCApply_FieldSplit_Schur
//   krylov_*             KSPFGMRESCycle's scalar work: Hessenberg column,
//                        Givens rotations, convergence test, back substitution
//   peer_* / granules    MPI_Allreduce and VecScatter across ranks (stores into the peers' HBM)
#include "spk_internal.hpp"

#include <cmath>
#include <cstdlib>

namespace spk {
namespace k {

constexpr int kThreads = 256;
constexpr int kWave = 64;
// Reducing vector kernels on big vectors run fat workgroups on a grid of <= 256 (one per
// CU): the reducer reads one partial row per workgroup, so few fat workgroups beat many thin ones.
constexpr int kVT = 1024;
constexpr int kVWaves = kVT / kWave;
constexpr int kVecUnroll = 4;    // double2 per thread per vector tile
constexpr int kVecMaxBlocks = 256;

// ---------------------------------------------------------------------------
// reductions inside a workgroup
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

// streamed-once operands: non-temporal 16-byte loads (global_load_dwordx4 ... nt);
// measured +9 % on the Krylov basis streams (5.25 -> 5.7 TB/s)
typedef double dbl2v __attribute__((ext_vector_type(2)));
typedef int int4v __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ double2 ld2s(const double *p, int64_t i2)
{
    if (NT) {
        const dbl2v v = __builtin_nontemporal_load(reinterpret_cast<const dbl2v *>(p) + i2);
        double2 r;
        r.x = v.x;
        r.y = v.y;
        return r;
    }
    return reinterpret_cast<const double2 *>(p)[i2];
}
template <bool NT>
__device__ __forceinline__ int4 ld4i(const int32_t *p)
{
    if (NT) {
        const int4v v = __builtin_nontemporal_load(reinterpret_cast<const int4v *>(p));
        int4 r;
        r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
        return r;
    }
    return *reinterpret_cast<const int4 *>(p);
}

// ---------------------------------------------------------------------------
// Cross-workgroup finish without a second launch, without fences and without
// counters.  Every slot of the partials buffer rests at a SENTINEL (a NaN bit
// pattern no arithmetic produces).  Every workgroup PUBLISHES its k partial
// sums with sc1 (write-through) 8-byte stores and is done -- no drain, no
// arrival.  The workgroup with the highest block index (dispatched last) is the
// reducer: it reads all partials with sc1 loads, spinning on any slot that still
// holds the sentinel, puts the sentinel back, and sums in a FIXED order (bitwise
// reproducible, no float atomics).  A value is its own arrival flag, so the chain
// after the last producer is one store flight + one load round trip, where
// "drain -> atomic arrival -> re-read" (cdna_hip_programming.md Guideline 16, R1)
// was three to four dependent round trips: measured on a 262 k-row vector, MAXPY +
// norm 7.3 -> 5.8 us, MDOT 7.8 -> 6.4 us (3.4 us for the MAXPY stream without any reduction).
// The reducer asks for kFinBatch partials per thread at a time and simply asks again while any
// of them is still armed; a per-slot re-poll, or 32 at a time, doubled the VGPRs of the WHOLE
// kernel (75 -> 149..256) and cost more occupancy in the streaming part than the finish gained.
// The sentinel is restored inside the kernel that consumed it, so the next launch
// on the stream (ordered by the kernel boundary) finds every slot armed.
// Every spin is bounded; a slot that never arrives reads as NaN
// (-> KSP_DIVERGED_NANORINF), it cannot hang the kernel.
// ---------------------------------------------------------------------------
constexpr unsigned long long kSentinelBits = 0xFFF8DEADBEEF5A5Aull;
__device__ __forceinline__ void publish(double *p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double peek(const double *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool is_sentinel(double v)
{
    return (unsigned long long)__double_as_longlong(v) == kSentinelBits;
}

// true in every thread of the reducing workgroup (the last block of the grid)
__device__ __forceinline__ bool arrive_last(unsigned nblocks)
{
    if (blockIdx.x != nblocks - 1) return false;
    __syncthreads();  // the caller's LDS staging is reused as scratch below
    return true;
}

// reducer (blockDim.x = T threads, power of two): scratch[i] = sum_b partials[b*ld + i], i < k <= 64.
// Strided slices (a thread's loads are all requested before its first add: one memory round trip
// when everything has arrived), then a fixed binary tree; the result is valid in LDS scratch[0..k)
// after return.  scratch: T doubles.
constexpr int kFinBatch = 16;  // partials a reducer thread requests together (registers of the WHOLE kernel: 32 cost 2x the VGPRs)
// where a reducer reports a partial that never arrived (execution failure, not a numerical one): the
// context's sticky error word; the bound of the wait in 100 MHz ticks
struct FinErr {
    int32_t *err;
    uint32_t ticks;
};
__device__ __forceinline__ void final_reduce(double *partials, int nb, int ld, int k, double *scratch, FinErr fe)
{
    const int T = blockDim.x;
    int kk = 1;
    while (kk < k) kk <<= 1;
    const int i = threadIdx.x & (kk - 1), sl = threadIdx.x / kk, nsl = T / kk;
    const double armed = __longlong_as_double((long long)kSentinelBits);
    double acc = 0.0;
    if (i < k) {
        for (int b0 = sl; b0 < nb; b0 += nsl * kFinBatch) {
            double v[kFinBatch];
            // the whole batch is requested at once (one round trip) and simply requested again while
            // any of its slots is still armed, i.e. its workgroup has not published yet
            const unsigned long long t0 = wall_clock64();
            bool armed_seen;
            do {
                armed_seen = false;
#pragma unroll
                for (int u = 0; u < kFinBatch; ++u) {
                    const int b = b0 + u * nsl;
                    v[u] = b < nb ? peek(partials + (size_t)b * ld + i) : 0.0;
                }
#pragma unroll
                for (int u = 0; u < kFinBatch; ++u) armed_seen = armed_seen || is_sentinel(v[u]);
                if (armed_seen) __builtin_amdgcn_s_sleep(1);
            } while (armed_seen && wall_clock64() - t0 < (unsigned long long)fe.ticks);  // default 4 s at 100 MHz
            // a slot still armed after the bound: its workgroup never published (never dispatched, or the
            // launch was rejected half way).  Not a numerical event: raise the context's sticky error word --
            // the host turns it into SPK_ERR_HIP and re-arms the whole buffer before the next use
            if (armed_seen && fe.err) __hip_atomic_store(fe.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < kFinBatch; ++u) {
                const int b = b0 + u * nsl;
                if (b < nb) publish(partials + (size_t)b * ld + i, armed);  // re-arm for the next launch
                acc += v[u];
            }
        }
    }
    scratch[sl * kk + i] = acc;
    __syncthreads();
    for (int st = nsl >> 1; st > 0; st >>= 1) {
        if (sl < st) scratch[sl * kk + i] += scratch[(sl + st) * kk + i];
        __syncthreads();
    }
}

// fills the partials buffer with the sentinel (once, at allocation)
__global__ void arm_partials_kernel(double *p, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = __longlong_as_double((long long)kSentinelBits);
}
void arm_partials(double *p, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL(arm_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n);
}

// Test hook (spk_debug_finish_timeout): a four-workgroup reduction whose last partial is never
// published -- the reducer must give up after fe.ticks, raise the error word and leave the kernel.
__global__ __launch_bounds__(256) void finish_probe_kernel(double *partials, double *out, FinErr fe)
{
    __shared__ double scratch[256];
    if (blockIdx.x + 1 < gridDim.x) {
        if (threadIdx.x == 0) publish(partials + (size_t)blockIdx.x * kPartialLd, 1.0);
        return;
    }
    __syncthreads();
    final_reduce(partials, gridDim.x, kPartialLd, 1, scratch, fe);  // slot gridDim.x - 1 stays armed
    if (threadIdx.x == 0) out[0] = scratch[0];
}
void finish_probe(const Finish &f, hipStream_t s)
{
    hipLaunchKernelGGL(finish_probe_kernel, dim3(4), dim3(256), 0, s, f.partials, f.out, FinErr{f.err, f.fin_ticks});
}

// ---------------------------------------------------------------------------
// Peer-store collectives over xGMI (replace MPI_Allreduce / VecScatter inside
// KSPSolve; SURVEY 8(e): the payloads are <= 64 doubles and one node line, so
// latency is everything).  Data travels as 8-byte GRANULES {sequence number,
// 32 payload bits} written by ONE system-scope store each into the receiver's
// window (uncached device memory mapped into every peer): a granule is its own
// arrival flag, so there is no fence and no second round trip -- the receiver
// spins on the tag of each granule it needs.  Every poll is bounded (the peer
// may have died): on time-out the error word is raised and the kernel ends.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void st_sys(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// waits until the granule at p carries `seq`; lo = its payload.  false: timed out, or an earlier
// wait of this context did (the error word is sticky: once a peer is lost every later wait gives up
// at once, so a whole enqueued restart cycle drains in one time-out, not one per collective).
__device__ __forceinline__ bool granule_wait(const unsigned long long *p, uint32_t seq, uint32_t timeout_ms, uint32_t &lo,
                                             const int32_t *err, const int32_t *done = nullptr)
{
    unsigned long long g = ld_sys(p);
    if ((uint32_t)(g >> 32) != seq) {
        const unsigned long long t0 = wall_clock64();  // 100 MHz
        for (;;) {
            __builtin_amdgcn_s_sleep(2);
            g = ld_sys(p);
            if ((uint32_t)(g >> 32) == seq) break;
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                wall_clock64() - t0 > (unsigned long long)timeout_ms * 100000ull) {
                lo = 0;
                return false;
            }
            // the solve converged while this kernel was in flight: the peers stop sending, nobody
            // reads what is missing (every consumer starts with "if (*done) return")
            if (done && __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                lo = 0;
                return true;
            }
        }
    }
    lo = (uint32_t)g;
    return true;
}
// Raises the sticky error word of the peer-store backend and notes WHICH wait gave up (first one wins):
// err[1] = where (1..3: all-reduce after MDot / after MAXPY / stand-alone; 16: halo rows in a head kernel,
// 17: in the MAXPY-head kernel, 18: in kernel B of the two-launch iteration, 19: granule exchange kernel,
// 20: bulk exchange kernel), err[2] = sequence number waited for.
__device__ __forceinline__ void raise_comm_error(int32_t *err, int where, uint32_t seq)
{
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        err[1] = where;
        err[2] = (int32_t)seq;
    }
    __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double join_halves(uint32_t lo, uint32_t hi)
{
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Threads 0 .. 2*count-1 of the calling workgroup (count <= 64; a double's two halves sit in
// adjacent lanes) sum vals[0..count) over the ranks into out[0..count): every rank adds the
// P contributions in rank order, its own included, so all ranks hold the same bits.
// No barrier inside; vals may be LDS or global, out may alias vals.
// The two halves of peer_allreduce_block, for a sum whose consumer sits in a LATER launch: `post` (threads
// 0 .. 2*count-1) stores this rank's contribution into every rank's window and returns; `wait` (same threads, any
// later launch of the stream) collects the P contributions.  What runs between the two overlaps the link latency.
__device__ __forceinline__ void peer_allreduce_post(const PeerAR &a, const double *vals, int count)
{
    const int t = threadIdx.x;
    if (t >= 2 * count) return;
    const int slot = (int)(a.seq & (kArSlots - 1));
    const uint32_t half = reinterpret_cast<const uint32_t *>(vals)[t];
    const unsigned long long g = ((unsigned long long)a.seq << 32) | half;
    const size_t mine = ((size_t)slot * a.P + a.me) * kArGranules + t;
    for (int p = 0; p < a.P; ++p) st_sys(a.win[p] + mine, g);
}
__device__ __forceinline__ void peer_allreduce_wait(const PeerAR &a, int count, double *out)
{
    const int t = threadIdx.x;
    if (t >= 2 * count) return;
    const int slot = (int)(a.seq & (kArSlots - 1));
    const unsigned long long *own = a.win[a.me] + (size_t)slot * a.P * kArGranules + t;
    const unsigned long long tw0 = (a.stats && t == 0) ? wall_clock64() : 0ull;
    double sum = 0.0;
    bool ok = true;
    for (int p = 0; p < a.P; ++p) {
        uint32_t lo;
        ok = granule_wait(own + (size_t)p * kArGranules, a.seq, a.timeout_ms, lo, a.err) && ok;
        const uint32_t other = __shfl_xor(lo, 1, kWave);
        sum += join_halves(lo, other);  // meaningful in even lanes
    }
    if (a.stats && t == 0) {
        atomicAdd(a.stats + 2 * a.kind, wall_clock64() - tw0);
        atomicAdd(a.stats + 2 * a.kind + 1, 1ull);
    }
    if (!(t & 1)) out[t >> 1] = sum;
    if (!ok) raise_comm_error(a.err, 1 + a.kind, a.seq);
}
__device__ __forceinline__ void peer_allreduce_block(const PeerAR &a, const double *vals, int count, double *out)
{
    const int t = threadIdx.x;
    if (t >= 2 * count) return;
    const int slot = (int)(a.seq & (kArSlots - 1));
    const uint32_t half = reinterpret_cast<const uint32_t *>(vals)[t];
    const unsigned long long g = ((unsigned long long)a.seq << 32) | half;
    const size_t mine = ((size_t)slot * a.P + a.me) * kArGranules + t;
    for (int p = 0; p < a.P; ++p) st_sys(a.win[p] + mine, g);
    const unsigned long long *own = a.win[a.me] + (size_t)slot * a.P * kArGranules + t;
    const unsigned long long tw0 = (a.stats && t == 0) ? wall_clock64() : 0ull;
    double sum = 0.0;
    bool ok = true;
    for (int p = 0; p < a.P; ++p) {
        uint32_t lo;
        ok = granule_wait(own + (size_t)p * kArGranules, a.seq, a.timeout_ms, lo, a.err) && ok;
        const uint32_t other = __shfl_xor(lo, 1, kWave);
        sum += join_halves(lo, other);  // meaningful in even lanes
    }
    if (a.stats && t == 0) {  // one lane accounts for the collective: stores issued -> every rank's lane arrived
        atomicAdd(a.stats + 2 * a.kind, wall_clock64() - tw0);
        atomicAdd(a.stats + 2 * a.kind + 1, 1ull);
    }
    if (!(t & 1)) out[t >> 1] = sum;
    if (!ok) raise_comm_error(a.err, 1 + a.kind, a.seq);
}

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

__device__ void givens_block(const KrylovArrays &ka, int loc, const double *dots, const double *nrm2, int *gate = nullptr);  // below
__device__ __forceinline__ double inv_norm(double nrm2)  // the VecScale guard of the head kernels
{
    const double tt = sqrt(nrm2);
    return tt > 1e-300 ? 1.0 / tt : 1.0;
}
__device__ __forceinline__ void givens_rider(const GivensRider &gr)
{
    // peer-store: the MAXPY launch only POSTED its ||w'||^2; the contributions are collected here, beside the row
    // tiles -- nothing in a product on an un-normalised basis needs the norm, so this all-reduce costs no time
    if (gr.ar.P) {
        peer_allreduce_wait(gr.ar, 1, gr.nrm2);
        __syncthreads();
    }
    // un-normalised basis: the scale factor of the vector the MAXPY launch just wrote (its norm is all-reduced by now)
    // (nothing compounds: V~_j = w' of the product of the NORMALISED v_{j-1}, so ||V~_j|| = h_{j,j-1} <= ||K M^-1||)
    if (gr.sc && threadIdx.x == 0) gr.sc[gr.loc + 1] = inv_norm(*gr.nrm2);
    givens_block(gr.ka, gr.loc, gr.h, gr.nrm2);
}

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
    if (RIDE && blockIdx.x == 0) {  // the rider: a pending Givens step beside the tiles
        givens_rider(gr);
        return;
    }
    // workgroups b, b+8, ... share an XCD (round-robin dispatch): give each XCD
    // a contiguous run of row tiles so the x window stays in ITS L2.
    const int bx = (int)blockIdx.x - (RIDE ? 1 : 0);
    const int t = (bx & 7) * tiles_per_xcd + (bx >> 3);
    if ((bx >> 3) >= tiles_per_xcd || t >= ntiles) return;

    __shared__ double prod[TILE + 8];
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
    const int r = r0 + threadIdx.x;
    if (r < r1) {
        const int k0 = rowptr[r] - a0, k1 = rowptr[r + 1] - a0;
        double s = 0.0;
        for (int k = k0; k < k1; ++k) s += prod[k];
        if (od.rowptr)  // off-rank columns of this row (ghost values already exchanged)
            for (int k = od.rowptr[r]; k < od.rowptr[r + 1]; ++k) s += od.val[k] * od.xg[od.colidx[k]];
        if (bt_rowptr)
            for (int k = bt_rowptr[r]; k < bt_rowptr[r + 1]; ++k) s += bt_val[k] * lam[bt_colidx[k]];
        if (accumulate) s += y[r];  // y pre-loaded with B^T lambda by the fused PC kernel
        y[r] = s;
    }
}

void givens_rider_alone(const GivensRider &gr, const int32_t *done, hipStream_t s);  // below
static GivensRider no_rider()
{
    GivensRider g{};
    g.loc = -1;
    return g;
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
        hipLaunchKernelGGL((spmv_stream_kernel<false, kCsrTile, kThreads, true>), dim3(tpx * 8 + 1), dim3(kThreads), 0, s,
                           A.rowptr.p, A.colidx.p, A.val.p, A.tile_row.p, A.ntiles, tpx, x, y, bt ? bt->rowptr.p : nullptr,
                           bt ? bt->colidx.p : nullptr, bt ? bt->val.p : nullptr, lam, accumulate ? 1 : 0, o, done, gr);
    else
        hipLaunchKernelGGL((spmv_stream_kernel<false, kCsrTile, kThreads, false>), dim3(tpx * 8), dim3(kThreads), 0, s,
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
template <bool NT, bool ACC, bool RIDE>
__global__ __launch_bounds__(kThreads) void spmv_bcsr_kernel(
    const int32_t *__restrict__ browptr, const int32_t *__restrict__ bcol,
    const double *__restrict__ vtop, const double *__restrict__ vbot,
    const int32_t *__restrict__ tile_brow, int ntiles, int tiles_per_xcd,
    const double *__restrict__ x, double *__restrict__ y, const int32_t *__restrict__ bt_rowptr,
    const int32_t *__restrict__ bt_colidx, const double *__restrict__ bt_val,
    const double *__restrict__ lam, OffDiag od, const int32_t *__restrict__ done, GivensRider gr)
{
    if (done && *done) return;
    if (RIDE && blockIdx.x == 0) {  // the rider: a pending Givens step beside the tiles
        givens_rider(gr);
        return;
    }
    const int bx = (int)blockIdx.x - (RIDE ? 1 : 0);
    const int t = (bx & 7) * tiles_per_xcd + (bx >> 3);
    if (t >= ntiles) return;
    __shared__ double prod[kBTile * 4];
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

    const int lr = threadIdx.x;  // local row
    if (lr < 2 * (br1 - br0)) {
        const int br = br0 + (lr >> 1), half = lr & 1;
        const int k0 = browptr[br] - b0, k1 = browptr[br + 1] - b0;
        double s = 0.0;
        for (int k = k0; k < k1; ++k) {
            const double2 p = *reinterpret_cast<const double2 *>(prod + 4 * k + 2 * half);
            s += p.x;
            s += p.y;
        }
        const int r = 2 * br0 + lr;
        if (od.rowptr)  // off-rank columns of this row (ghost values already exchanged)
            for (int k = od.rowptr[r]; k < od.rowptr[r + 1]; ++k) s += od.val[k] * od.xg[od.colidx[k]];
        if (bt_rowptr)
            for (int k = bt_rowptr[r]; k < bt_rowptr[r + 1]; ++k) s += bt_val[k] * lam[bt_colidx[k]];
        if (ACC) s += y[r];
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
#define SPK_LAUNCH_BCSR(ACC, RIDE)                                                                                         \
    hipLaunchKernelGGL((spmv_bcsr_kernel<true, ACC, RIDE>), dim3(tpx * 8 + nride), dim3(kThreads), 0, s, A.browptr.p, A.bcol.p, \
                       A.vtop.p, A.vbot.p, A.tile_brow.p, A.ntiles, tpx, x, y, bt ? bt->rowptr.p : nullptr,                \
                       bt ? bt->colidx.p : nullptr, bt ? bt->val.p : nullptr, lam, od, done, gr)
    if (accumulate) {
        if (rider) SPK_LAUNCH_BCSR(true, true);
        else SPK_LAUNCH_BCSR(true, false);
    } else {
        if (rider) SPK_LAUNCH_BCSR(false, true);
        else SPK_LAUNCH_BCSR(false, false);
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
    if (RIDE && blockIdx.x == 0) {  // the rider: a pending Givens step beside the tiles
        givens_rider(gr);
        return;
    }
    const int bx = (int)blockIdx.x - (RIDE ? 1 : 0);
    const int t = (bx & 7) * tiles_per_xcd + (bx >> 3);
    if (t >= ntiles) return;
    __shared__ double prod[kB3Tile * 9];
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

    const int lr = threadIdx.x;  // local row
    if (lr < 3 * (br1 - br0)) {
        const int br = br0 + lr / 3, rr = lr % 3;
        const int k0 = browptr[br] - b0, k1 = browptr[br + 1] - b0;
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
        if (ACC) s += y[r];
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
    hipLaunchKernelGGL((spmv_bcsr3_kernel<ACC, RIDE>), dim3(tpx * 8 + nride), dim3(kThreads), 0, s, A.browptr.p, A.bcol.p, \
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
// B x for the short-and-wide constraint block (4 rows of ~n/2 entries): one
// workgroup per (column window, row), 16-byte loads of the row's entries in the
// window, x (optionally x .* scale) gathered; the last block of the grid sums the
// window partials of each row in window order.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kVT) void wide_dot_kernel(
    const int32_t *__restrict__ colidx, const double *__restrict__ val,
    const int32_t *__restrict__ winptr, int m, int nwin, const double *__restrict__ x,
    const double *__restrict__ scale, double *__restrict__ partials,
    double *__restrict__ out, const int32_t *__restrict__ rowmap, FinErr fe, const int32_t *__restrict__ done)
{
    // rowmap != nullptr: the m rows here are the LONG rows of a larger constraint block (the short ones go
    // through the CSR stream kernel); row r of this launch is row rowmap[r] of the block
    if (done && *done) return;
    __shared__ double scratch[kVT];
    const int w = blockIdx.x;
    for (int r = 0; r < m; ++r) {
        const int k0 = winptr[w * m + r], k1 = winptr[(w + 1) * m + r];
        const int a0 = k0 & ~3;
        double acc = 0.0;
        for (int q = a0 + (int)threadIdx.x * 4; q < k1; q += kVT * 4) {
            const int4 c = *reinterpret_cast<const int4 *>(colidx + q);
            const double2 v0 = *reinterpret_cast<const double2 *>(val + q);
            const double2 v1 = *reinterpret_cast<const double2 *>(val + q + 2);
            double x0 = x[c.x], x1 = x[c.y], x2 = x[c.z], x3 = x[c.w];
            if (scale) {
                x0 *= scale[c.x]; x1 *= scale[c.y]; x2 *= scale[c.z]; x3 *= scale[c.w];
            }
            if (q >= k0 && q < k1) acc += v0.x * x0;
            if (q + 1 >= k0 && q + 1 < k1) acc += v0.y * x1;
            if (q + 2 >= k0 && q + 2 < k1) acc += v1.x * x2;
            if (q + 3 >= k0 && q + 3 < k1) acc += v1.y * x3;
        }
        const double sw = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = sw;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < kVWaves; ++j) t += scratch[j];
            publish(partials + (size_t)w * kPartialLd + r, t);
        }
        __syncthreads();
    }
    if (!arrive_last(gridDim.x)) return;
    final_reduce(partials, nwin, kPartialLd, m, scratch, fe);
    if ((int)threadIdx.x < m) out[rowmap ? rowmap[threadIdx.x] : (int)threadIdx.x] = scratch[threadIdx.x];
}

static void wide_dot_scaled(const WideDev &B, const double *x, const double *scale, const Finish &f,
                            const int32_t *done, hipStream_t s, const int32_t *rowmap)
{
    if (B.nwin == 0) return;
    hipLaunchKernelGGL(wide_dot_kernel, dim3(B.nwin), dim3(kVT), 0, s, B.colidx.p, B.val.p,
                       B.winptr.p, B.m, B.nwin, x, scale, f.partials, f.out, rowmap, FinErr{f.err, f.fin_ticks}, done);
}
void wide_dot(const WideDev &B, const double *x, const Finish &f, const int32_t *done, hipStream_t s, const int32_t *rowmap)
{
    wide_dot_scaled(B, x, nullptr, f, done, s, rowmap);
}
void wide_dot_jacobi(const WideDev &B, const double *x, const double *dinv, const Finish &f,
                     const int32_t *done, hipStream_t s)
{
    wide_dot_scaled(B, x, dinv, f, done, s, nullptr);
}

// S^_r = sum_k B_rk^2 dinv[col_k] for the rows of a CSR block, one wave per row (lanes stride the row, fixed
// shuffle tree: reproducible).  PCFIELDSPLIT's selfp Schur complement restricted to its diagonal, for constraint
// blocks with many short rows (the few long rows of such a block go through scatter_row + wide_dot).
__global__ __launch_bounds__(kThreads) void schur_diag_rows_kernel(const int32_t *__restrict__ rowptr,
                                                                   const int32_t *__restrict__ colidx,
                                                                   const double *__restrict__ val, int nrows,
                                                                   const double *__restrict__ dinv,
                                                                   double *__restrict__ shat)
{
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (kThreads / kWave) + (threadIdx.x >> 6);
    if (r >= nrows) return;
    double acc = 0.0;
    for (int k = rowptr[r] + lane; k < rowptr[r + 1]; k += kWave) acc += val[k] * val[k] * dinv[colidx[k]];
    const double sr = wave_sum(acc);
    if (lane == 0) shat[r] = sr;
}
void schur_diag_rows(const CsrDev &B, const double *dinv, double *shat, hipStream_t s)
{
    if (B.nrows == 0) return;
    const int wpb = kThreads / kWave;
    hipLaunchKernelGGL(schur_diag_rows_kernel, dim3((B.nrows + wpb - 1) / wpb), dim3(kThreads), 0, s, B.rowptr.p,
                       B.colidx.p, B.val.p, B.nrows, dinv, shat);
}

// ---------------------------------------------------------------------------
// VecMDot: all nv dot products V_i . w in ONE pass over w (kept in registers),
// plus w.w in slot nv.  Template NG = groups of 8 vectors (static accumulators).
// ---------------------------------------------------------------------------
__device__ __forceinline__ double2 ld2(const double *p, int64_t i2)
{
    return reinterpret_cast<const double2 *>(p)[i2];
}


// T threads per workgroup, G vectors loaded together (their 4*G 16-byte loads per
// thread are all issued before the first FMA: the bytes in flight, not the
// arithmetic, set the rate of this kernel).
template <int NG, int T, int G, bool NT, int U>
__global__ __launch_bounds__(T) void mdot_kernel(const double *__restrict__ V, int64_t ldv, int nv,
                                                 const double *__restrict__ V2, int nv1,
                                                 const double *__restrict__ w, int64_t n2,
                                                 int64_t n_dot, double *__restrict__ partials,
                                                 int with_ww,
                                                 double *__restrict__ out, PeerAR ar, int split, FinErr fe,
                                                 const int32_t *__restrict__ done)
{
    if (done && *done) return;
    constexpr int NA = NG * 8 + 1, W = T / kWave, TILE2 = T * U;
    __shared__ double lds[(W * NA > T) ? W * NA : T];
    double acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = 0.0;

    for (int64_t tile = blockIdx.x; tile * TILE2 < n2; tile += gridDim.x) {
        double2 wv[U];
        int64_t idx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            idx[u] = tile * TILE2 + u * T + threadIdx.x;
            if (idx[u] < n2) {
                wv[u] = ld2(w, idx[u]);
                if (2 * idx[u] >= n_dot) wv[u].x = 0.0;
                if (2 * idx[u] + 1 >= n_dot) wv[u].y = 0.0;
            } else {
                wv[u].x = wv[u].y = 0.0;
                idx[u] = 0;  // safe address, zero weight
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc[NA - 1] += wv[u].x * wv[u].x + wv[u].y * wv[u].y;
#pragma unroll
        for (int g0 = 0; g0 < NG * 8; g0 += G) {
            if (g0 < nv) {  // wave-uniform
                double2 a[G][U];
#pragma unroll
                for (int v = 0; v < G; ++v) {
                    // a slot past nv loads ONE broadcast address (w[0..1], weight 0) instead of a
                    // vector tile: the group stays branch-free and costs no bandwidth
                    const bool live = g0 + v < nv;
                    const int ic = live ? g0 + v : 0;
                    // vectors nv1.. come from a second slab (the rows of B D in the single-reduction mode);
                    // split: that slab holds parity-interleaved planes, "vector" j is half j & 1 of plane j / 2
                    const int j2 = ic - nv1;
                    const double *Vi = !live ? w : (ic < nv1 ? V + (size_t)ic * ldv : V2 + (size_t)(split ? j2 >> 1 : j2) * ldv);
#pragma unroll
                    for (int u = 0; u < U; ++u) a[v][u] = ld2s<NT>(Vi, live ? idx[u] : 0);
                }
#pragma unroll
                for (int v = 0; v < G; ++v) {
                    const double mk = (g0 + v < nv) ? 1.0 : 0.0;
                    double d = 0.0;
                    if (split && g0 + v >= nv1 && g0 + v < nv) {  // wave-uniform
                        if ((g0 + v - nv1) & 1) {
#pragma unroll
                            for (int u = 0; u < U; ++u) d += a[v][u].y * wv[u].y;
                        } else {
#pragma unroll
                            for (int u = 0; u < U; ++u) d += a[v][u].x * wv[u].x;
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < U; ++u) d += a[v][u].x * wv[u].x + a[v][u].y * wv[u].y;
                    }
                    acc[g0 + v] += mk * d;
                }
            }
        }
    }
    // workgroup sums -> partials[block][i]; w.w goes to slot nv
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const double s = wave_sum(acc[i]);
        if (lane == 0) lds[wave * NA + i] = s;
    }
    __syncthreads();
    if (threadIdx.x < NA) {
        const int i = threadIdx.x;
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < W; ++j) s += lds[j * NA + i];
        double *row = partials + (size_t)blockIdx.x * kPartialLd;
        if (i < nv) publish(row + i, s);
        else if (i == NA - 1 && with_ww) publish(row + nv, s);
    }
    if (!arrive_last(gridDim.x)) return;
    const int k = nv + (with_ww ? 1 : 0);
    final_reduce(partials, gridDim.x, kPartialLd, k, lds, fe);
    // across ranks: the workgroup that finished this rank's sums also exchanges them (no launch of its own)
    if (ar.P) peer_allreduce_block(ar, lds, k, out);
    else if ((int)threadIdx.x < k) out[threadIdx.x] = lds[threadIdx.x];
}

// ---------------------------------------------------------------------------
// Small vectors -- a rank's slab of a strong-scaling run (262 k rows at 1024^2 / 8), the 256^2 and
// 512^2 grids.  Such a vector gives every wave of the chip ONE tile: the kernels above then walk
// their j+1 basis vectors group after group, a chain of dependent memory round trips with nothing
// else resident to hide them (measured on the 1/8 slab: 13.7 / 25.6 us for 15 / 30 vectors,
// 2.5 TB/s out of the Infinity Cache).  The "wave-split" MDOT below turns the work by 90 degrees:
// the four waves of a workgroup share one LONG tile (64 lanes x U double2 = up to 8 KB per
// vector) and split the VECTORS between them, so a wave's chain is a quarter as long, its
// accumulators and shuffles a quarter as many, and every stream is read in 8 KB runs.
// Wave q owns vectors [q*per, (q+1)*per): its sums go straight to the partials.  Measured on the
// 1/8 slab: 11.0 / 17.6 us for 15 / 30 vectors (slope 0.33 us = 6.3 TB/s per vector).  The same turn
// applied to MAXPY (contributions combined through LDS) gained nothing; MAXPY instead runs thin
// workgroups with 8 vectors in flight there (vec_shape).
// Sums are formed in a fixed order: reproducible, not bit-equal to the streaming form.
// ---------------------------------------------------------------------------
template <int VW, int U, int G, bool NT>
__global__ __launch_bounds__(256) void mdot_ws_kernel(const double *__restrict__ V, int64_t ldv, int nv,
                                                      const double *__restrict__ V2, int nv1,
                                                      const double *__restrict__ w, int64_t n2, int64_t n_dot,
                                                      double *__restrict__ partials, int with_ww, double *__restrict__ out,
                                                      PeerAR ar, int split, FinErr fe, const int32_t *__restrict__ done)
{
    if (done && *done) return;
    __shared__ double lds[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = (nv + 3) >> 2;
    const int v0 = wave * per;
    const int cnt = (nv - v0) < per ? (nv - v0) : per;  // may be <= 0: a wave without vectors
    double acc[VW];
#pragma unroll
    for (int i = 0; i < VW; ++i) acc[i] = 0.0;
    double ww = 0.0;
    for (int64_t tile = blockIdx.x; tile * (64 * U) < n2; tile += gridDim.x) {
        double2 wv[U];
        int64_t idx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            idx[u] = tile * (64 * U) + u * 64 + lane;
            if (idx[u] < n2) {
                wv[u] = ld2(w, idx[u]);
                if (2 * idx[u] >= n_dot) wv[u].x = 0.0;
                if (2 * idx[u] + 1 >= n_dot) wv[u].y = 0.0;
            } else {
                wv[u].x = wv[u].y = 0.0;
                idx[u] = 0;
            }
        }
        if (wave == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) ww += wv[u].x * wv[u].x + wv[u].y * wv[u].y;
        }
#pragma unroll
        for (int g0 = 0; g0 < VW; g0 += G) {
            if (g0 < cnt) {  // wave-uniform
                double2 a[G][U];
#pragma unroll
                for (int v = 0; v < G; ++v) {
                    const bool live = g0 + v < cnt;
                    const int ic = v0 + (live ? g0 + v : 0);
                    const int j2 = ic - nv1;  // split: "vector" j2 of the second slab is half j2 & 1 of plane j2 / 2
                    const double *Vi = !live ? w : (ic < nv1 ? V + (size_t)ic * ldv : V2 + (size_t)(split ? j2 >> 1 : j2) * ldv);
#pragma unroll
                    for (int u = 0; u < U; ++u) a[v][u] = ld2s<NT>(Vi, live ? idx[u] : 0);
                }
#pragma unroll
                for (int v = 0; v < G; ++v) {
                    const double mk = (g0 + v < cnt) ? 1.0 : 0.0;
                    const int ic = v0 + g0 + v;
                    double d = 0.0;
                    if (split && ic >= nv1 && g0 + v < cnt) {  // wave-uniform
                        if ((ic - nv1) & 1) {
#pragma unroll
                            for (int u = 0; u < U; ++u) d += a[v][u].y * wv[u].y;
                        } else {
#pragma unroll
                            for (int u = 0; u < U; ++u) d += a[v][u].x * wv[u].x;
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < U; ++u) d += a[v][u].x * wv[u].x + a[v][u].y * wv[u].y;
                    }
                    acc[g0 + v] += mk * d;
                }
            }
        }
    }
    double *row = partials + (size_t)blockIdx.x * kPartialLd;
#pragma unroll
    for (int i = 0; i < VW; ++i) {
        if (i < cnt) {  // wave-uniform
            const double s = wave_sum(acc[i]);
            if (lane == 0) publish(row + v0 + i, s);
        }
    }
    if (wave == 0 && with_ww) {
        const double s = wave_sum(ww);
        if (lane == 0) publish(row + nv, s);
    }
    if (!arrive_last(gridDim.x)) return;
    const int k = nv + (with_ww ? 1 : 0);
    final_reduce(partials, gridDim.x, kPartialLd, k, lds, fe);
    if (ar.P) peer_allreduce_block(ar, lds, k, out);
    else if ((int)threadIdx.x < k) out[threadIdx.x] = lds[threadIdx.x];
}

// Second form of the same turn (default; SPK_VEC_WS16=0 falls back to the one above): SIXTEEN waves per
// workgroup, so a wave owns at most VW = 2..4 vectors and ALL its loads -- its tile of w and of each of
// its vectors -- are issued before the first FMA: one memory round trip per tile where the four-wave
// form walks its 8 vectors in 4 dependent rounds of two (measured on the 1/8 slab: 18.7 us for 30
// vectors with four waves, the kernel is a chain of latencies, not of bytes).
template <int VW, int U, bool NT>
__global__ __launch_bounds__(1024) void mdot_ws16_kernel(const double *__restrict__ V, int64_t ldv, int nv,
                                                         const double *__restrict__ V2, int nv1,
                                                         const double *__restrict__ w, int64_t n2, int64_t n_dot,
                                                         double *__restrict__ partials, int with_ww, double *__restrict__ out,
                                                         PeerAR ar, int split, FinErr fe, const int32_t *__restrict__ done)
{
    // the gate word is REQUESTED first and looked at behind the first tile's loads: a launch of these small forms is
    // a chain of a few memory round trips (~1.3 us each under load), and "read done, then start" was one of them
    const int32_t dn = done ? __builtin_nontemporal_load(done) : 0;
    __shared__ double lds[1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = (nv + 15) >> 4;
    const int v0 = wave * per;
    const int cnt = (nv - v0) < per ? (nv - v0) : per;  // may be <= 0: a wave without vectors
    double acc[VW];
#pragma unroll
    for (int i = 0; i < VW; ++i) acc[i] = 0.0;
    double ww = 0.0;
    if (cnt > 0 || (wave == 0 && with_ww)) {
        for (int64_t tile = blockIdx.x; tile * (64 * U) < n2; tile += gridDim.x) {
            double2 wv[U], a[VW][U];
            int64_t idx[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                idx[u] = tile * (64 * U) + u * 64 + lane;
                if (idx[u] >= n2) idx[u] = -1;
                wv[u] = ld2(w, idx[u] < 0 ? 0 : idx[u]);
            }
#pragma unroll
            for (int v = 0; v < VW; ++v) {
                const bool live = v < cnt;
                const int ic = v0 + (live ? v : 0);
                const int j2 = ic - nv1;  // split: "vector" j2 of the second slab is half j2 & 1 of plane j2 / 2
                const double *Vi = !live ? w : (ic < nv1 ? V + (size_t)ic * ldv : V2 + (size_t)(split ? j2 >> 1 : j2) * ldv);
#pragma unroll
                for (int u = 0; u < U; ++u) a[v][u] = ld2s<NT>(Vi, (live && idx[u] >= 0) ? idx[u] : 0);
            }
            if (dn) return;  // (uniform; every workgroup of the launch sees the same word)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (idx[u] < 0 || 2 * idx[u] >= n_dot) wv[u].x = 0.0;
                if (idx[u] < 0 || 2 * idx[u] + 1 >= n_dot) wv[u].y = 0.0;
            }
            if (wave == 0) {
#pragma unroll
                for (int u = 0; u < U; ++u) ww += wv[u].x * wv[u].x + wv[u].y * wv[u].y;
            }
#pragma unroll
            for (int v = 0; v < VW; ++v) {
                const int ic = v0 + v;
                double d = 0.0;
                if (split && ic >= nv1 && v < cnt) {  // wave-uniform
                    if ((ic - nv1) & 1) {
#pragma unroll
                        for (int u = 0; u < U; ++u) d += a[v][u].y * wv[u].y;
                    } else {
#pragma unroll
                        for (int u = 0; u < U; ++u) d += a[v][u].x * wv[u].x;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) d += a[v][u].x * wv[u].x + a[v][u].y * wv[u].y;
                }
                acc[v] += (v < cnt) ? d : 0.0;
            }
        }
    }
    if (dn) return;
    double *row = partials + (size_t)blockIdx.x * kPartialLd;
#pragma unroll
    for (int i = 0; i < VW; ++i) {
        if (i < cnt) {  // wave-uniform
            const double s = wave_sum(acc[i]);
            if (lane == 0) publish(row + v0 + i, s);
        }
    }
    if (wave == 0 && with_ww) {
        const double s = wave_sum(ww);
        if (lane == 0) publish(row + nv, s);
    }
    if (!arrive_last(gridDim.x)) return;
    const int k = nv + (with_ww ? 1 : 0);
    final_reduce(partials, gridDim.x, kPartialLd, k, lds, fe);
    if (ar.P) peer_allreduce_block(ar, lds, k, out);
    else if ((int)threadIdx.x < k) out[threadIdx.x] = lds[threadIdx.x];
}

// tile length (double2 per lane) and grid of the wave-split forms: about one tile per workgroup
struct WsShape {
    int U, grid;
    bool on;
};
static WsShape ws_shape(int64_t n2)
{
    WsShape v;
    static const int knob = [] { const char *e = getenv("SPK_VEC_WS"); return e ? atoi(e) : -1; }();  // 0: off, 2/4/8: force U
    v.on = n2 < (int64_t)kVecMaxBlocks * 2048 && knob != 0;
    v.U = n2 >= (int64_t)kVecMaxBlocks * 64 * 8 ? 8 : (n2 >= (int64_t)kVecMaxBlocks * 64 * 4 ? 4 : 2);
    if (knob == 2 || knob == 4 || knob == 8) v.U = knob;
    int64_t tiles = (n2 + 64 * v.U - 1) / (64 * v.U);
    if (tiles < 1) tiles = 1;
    v.grid = (int)(tiles < kVecMaxBlocks ? tiles : kVecMaxBlocks);
    return v;
}

// Workgroup shape of the reducing vector kernels: big vectors get 512 threads x 4 double2
// (few fat workgroups: cheap finish), small ones get thinner tiles so that ~256 workgroups
// still exist (a 131 k-row slab on 32 workgroups left 7/8 of the chip idle: 20 us instead of 5).
struct VecShape {
    int T, U, grid, G;
};
static VecShape vec_shape(int64_t n2, bool maxpy = false)
{
    VecShape v;
    v.G = 4;
    int cap = kVecMaxBlocks;
    if (n2 >= (int64_t)kVecMaxBlocks * 2048) { v.T = 512; v.U = 4; }
    else if (n2 >= (int64_t)kVecMaxBlocks * 1024) { v.T = 256; v.U = 4; }
    else if (n2 >= (int64_t)kVecMaxBlocks * 512) { v.T = 256; v.U = 2; }
    else { v.T = 256; v.U = 1; }
    // MAXPY on small vectors: thin workgroups, 8 vectors in flight (1/8 slab, 30 vectors: 17.1 -> 12.7 us,
    // 1/4 slab: 84 -> 79 us per iteration)
    if (maxpy && n2 < (int64_t)kVecMaxBlocks * 1024) { v.T = 256; v.U = 1; v.G = 8; cap = 1024; }
    else if (maxpy && n2 < (int64_t)kVecMaxBlocks * 2048) { v.T = 256; v.U = 2; v.G = 8; cap = 1024; }
    int64_t tiles = (n2 + (int64_t)v.T * v.U - 1) / ((int64_t)v.T * v.U);
    if (tiles < 1) tiles = 1;
    v.grid = (int)(tiles < cap ? tiles : cap);
    return v;
}
static int vec_grid(int64_t n2, int T = kVT)
{
    int64_t tiles = (n2 + (int64_t)T * kVecUnroll - 1) / ((int64_t)T * kVecUnroll);
    if (tiles < 1) tiles = 1;
    const int cap = T >= 512 ? kVecMaxBlocks : 2 * kVecMaxBlocks;
    return (int)(tiles < cap ? tiles : cap);
}

template <int T, int U, int G>
static void mdot_launch(int ng, int grid, hipStream_t s, const double *Vp, int64_t ldv, int cnt, const double *V2, int nv1,
                        const double *w, int64_t n2, int64_t n_dot, double *pp, int last, double *oo,
                        const PeerAR &ar, int split, FinErr fe, const int32_t *done)
{
#define SPK_MDOT(NGG) hipLaunchKernelGGL((mdot_kernel<NGG, T, G, true, U>), dim3(grid), dim3(T), 0, s, Vp, ldv, cnt, V2, nv1, w, \
                                         n2, n_dot, pp, last, oo, ar, split, fe, done)
    switch (ng) {
    case 1: SPK_MDOT(1); break;
    case 2: SPK_MDOT(2); break;
    case 3: SPK_MDOT(3); break;
    case 4: SPK_MDOT(4); break;
    default: SPK_MDOT(5); break;
    }
#undef SPK_MDOT
}

void mdot(const double *V, int64_t ldv, int nv, const double *w, int64_t n, int64_t n_dot,
          const Finish &f, const int32_t *done, hipStream_t s, const double *V2, int nv2, int split)
{
    // split: V2 holds nv2 / 2 parity-interleaved planes (pack_bd); result nv + j is half j & 1 of plane j / 2
    if (split && nv + nv2 > 40) fail(SPK_ERR_ARG, "mdot: split planes need one launch (<= 40 vectors)");
    // nv vectors from V, then nv2 from V2 (same stride); results in that order, w.w last
    const int ntot = nv + nv2;
    if (ntot > kMaxNv - 1) fail(SPK_ERR_ARG, "mdot: %d vectors exceed %d", ntot, kMaxNv - 1);
    if (f.ar.P && ntot > 40) fail(SPK_ERR_ARG, "mdot: the all-reduce rides in one launch only (<= 40 vectors)");
    const int64_t n2 = (n + 1) / 2;
    const VecShape vs = vec_shape(n2);
    // up to 40 vectors per launch; w.w is produced by the last launch
    int v0 = 0;
    do {
        const int cnt = (ntot - v0) < 40 ? (ntot - v0) : 40;
        const int last = (v0 + 40 >= ntot);
        // vector i of this launch is V[v0+i] while v0+i < nv, else V2[v0+i-nv]
        const double *Vp = V + (size_t)v0 * ldv;
        const int nv1 = nv - v0 > 0 ? nv - v0 : 0;
        const double *V2p = nv1 > 0 ? V2 : V2 + (size_t)(v0 - nv) * ldv;
        double *pp = f.partials + v0;
        double *oo = f.out + v0;
        const int ng = (cnt + 7) / 8 > 0 ? (cnt + 7) / 8 : 1;
        const WsShape ws = ws_shape(n2);
        static const int ws16 = [] { const char *e = getenv("SPK_VEC_WS16"); return e ? atoi(e) : 1; }();
        if (ws.on && ws16) {
            // sixteen waves, <= 3 vectors each (40 per launch), every load of a wave in flight at once
#define SPK_MDOT_W16(VW, UU) hipLaunchKernelGGL((mdot_ws16_kernel<VW, UU, true>), dim3(ws.grid), dim3(1024), 0, s, Vp, ldv, cnt, V2p, \
                                                nv1, w, n2, n_dot, pp, last, oo, f.ar, split, FinErr{f.err, f.fin_ticks}, done)
            const int per = (cnt + 15) / 16;
            if (ws.U == 8) { if (per <= 1) SPK_MDOT_W16(1, 8); else if (per <= 2) SPK_MDOT_W16(2, 8); else SPK_MDOT_W16(3, 4); }
            else if (ws.U == 4) { if (per <= 1) SPK_MDOT_W16(1, 4); else if (per <= 2) SPK_MDOT_W16(2, 4); else SPK_MDOT_W16(3, 4); }
            else { if (per <= 1) SPK_MDOT_W16(1, 2); else if (per <= 2) SPK_MDOT_W16(2, 2); else SPK_MDOT_W16(3, 2); }
#undef SPK_MDOT_W16
            v0 += 40;
            continue;
        }
        if (ws.on) {
#define SPK_MDOT_WS(VW, UU, GG) hipLaunchKernelGGL((mdot_ws_kernel<VW, UU, GG, true>), dim3(ws.grid), dim3(256), 0, s, Vp, ldv, cnt, \
                                                   V2p, nv1, w, n2, n_dot, pp, last, oo, f.ar, split, FinErr{f.err, f.fin_ticks}, done)
#define SPK_MDOT_WS_U(VW) do { if (ws.U == 8) SPK_MDOT_WS(VW, 8, 2); else if (ws.U == 4) SPK_MDOT_WS(VW, 4, 4); else SPK_MDOT_WS(VW, 2, 4); } while (0)
            const int per = (cnt + 3) / 4;
            if (per <= 4) SPK_MDOT_WS_U(4);
            else if (per <= 8) SPK_MDOT_WS_U(8);
            else SPK_MDOT_WS_U(12);
#undef SPK_MDOT_WS_U
#undef SPK_MDOT_WS
            v0 += 40;
            continue;
        }
#define SPK_MDOT_ARGS ng, vs.grid, s, Vp, ldv, cnt, V2p, nv1, w, n2, n_dot, pp, last, oo, f.ar, split, FinErr{f.err, f.fin_ticks}, done
        if (vs.T == 512) mdot_launch<512, 4, 4>(SPK_MDOT_ARGS);
        else if (vs.U == 4) mdot_launch<256, 4, 4>(SPK_MDOT_ARGS);
        else if (vs.U == 2 && vs.G == 8) mdot_launch<256, 2, 8>(SPK_MDOT_ARGS);
        else if (vs.U == 2) mdot_launch<256, 2, 4>(SPK_MDOT_ARGS);
        else if (vs.G == 8) mdot_launch<256, 1, 8>(SPK_MDOT_ARGS);
        else mdot_launch<256, 1, 4>(SPK_MDOT_ARGS);
#undef SPK_MDOT_ARGS
        v0 += 40;
    } while (v0 < ntot);
}

// ---------------------------------------------------------------------------
// VecMAXPY:  w += sign * sum_i a[i] V_i, coefficients read from device memory;
// the squared norm of the updated w (first n_dot entries) is produced in the
// same pass -> VecNorm costs no extra sweep.
// ---------------------------------------------------------------------------
template <int T, int G, bool NT, int MP, int U>
__global__ __launch_bounds__(T) void maxpy_kernel(const double *__restrict__ V, int64_t ldv,
                                                         int nv, const int32_t *__restrict__ nv_dev,
                                                         const double *__restrict__ a, double sign,
                                                         double *__restrict__ w, int64_t n2,
                                                         int64_t n_dot, double *__restrict__ partials,
                                                         double *__restrict__ out,
                                                         const double *__restrict__ bd, int64_t ldb,
                                                         int64_t n_bd, int m, double *__restrict__ w1side,
                                                         PythArgs py, PeerAR ar, int packed, FinErr fe,
                                                         const int32_t *__restrict__ done)
{
    if (done && *done) return;
    if (nv_dev) nv = *nv_dev;
    constexpr int NR = MP + 1, W = T / kWave;
    if (py.m >= 0 && blockIdx.x == 0 && threadIdx.x < kWave) {
        // single-reduction mode: the norm and B D w' of the vector this kernel is about to
        // build follow from the ONE reduced set {h = V^T w, q = B D w, w.w}:
        //   ||w'||^2 = w.w - sum h_i^2          (w' = w - V h, V orthonormal)
        //   B D w'   = q - sum h_i (B D v_i)    (tb[i] = B D v_i, kept per basis vector)
        // first wave of workgroup 0, lane i owns basis vector i (nv <= 63)
        const int i = threadIdx.x;
        const double hi = i < nv ? py.dots[i] : 0.0;
        const double hh = wave_sum(hi * hi);
        double tsum[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) tsum[r] = r < py.m ? wave_sum(i < nv ? hi * py.tb[i * 8 + r] : 0.0) : 0.0;
        if (i == 0) {
            // below ~64 eps w.w the difference is rounding noise (it can even come out negative): keep
            // the floor instead -- an over-estimated ||w'|| over-estimates the residual norm, so the
            // recurrence can never report a convergence that the true residual of the next restart
            // would not confirm (a zero here would read as a happy breakdown)
            const double ww = py.dots[nv + py.m];
            double tt2 = ww - hh;
            if (!(tt2 > 1.5e-14 * ww)) tt2 = 1.5e-14 * ww;
            py.nrm_out[0] = tt2;
            const double inv = tt2 > 0.0 ? 1.0 / sqrt(tt2) : 0.0;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (r < py.m) {
                    const double t = py.dots[nv + r] - tsum[r];
                    py.nrm_out[1 + r] = t;
                    py.tb[nv * 8 + r] = t * inv;
                }
            }
        }
    }
    __shared__ double red[(W * NR > T) ? W * NR : T];
    double nrm = 0.0;
    double tacc[MP > 0 ? MP : 1];
#pragma unroll
    for (int r = 0; r < (MP > 0 ? MP : 1); ++r) tacc[r] = 0.0;
    for (int64_t tile = blockIdx.x; tile * (T * U) < n2; tile += gridDim.x) {
        double2 wv[U];
        int64_t idx[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            idx[u] = tile * (T * U) + u * T + threadIdx.x;
            ok[u] = idx[u] < n2;
            if (!ok[u]) idx[u] = 0;
            wv[u] = ld2(w, idx[u]);
        }
        // G vectors per group: their 4*G loads are all in flight before the first FMA
        for (int g0 = 0; g0 < nv; g0 += G) {
            double2 t[G][U];
            double ai[G];
#pragma unroll
            for (int v = 0; v < G; ++v) {
                const bool live = g0 + v < nv;  // dead slots: one broadcast address, coefficient 0
                const int ic = live ? g0 + v : 0;
                ai[v] = live ? sign * a[ic] : 0.0;
                const double *Vi = V + (size_t)ic * ldv;
#pragma unroll
                for (int u = 0; u < U; ++u) t[v][u] = ld2s<NT>(Vi, live ? idx[u] : 0);
            }
#pragma unroll
            for (int v = 0; v < G; ++v) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    wv[u].x += ai[v] * t[v][u].x;
                    wv[u].y += ai[v] * t[v][u].y;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ok[u]) {
                reinterpret_cast<double2 *>(w)[idx[u]] = wv[u];
                if (2 * idx[u] < n_dot) nrm += wv[u].x * wv[u].x;
                if (2 * idx[u] + 1 < n_dot) nrm += wv[u].y * wv[u].y;
                if (MP > 0 && w1side) {  // lambda part of the un-normalised vector, for the next head kernel
                    const int64_t e0 = 2 * idx[u] - n_bd;
                    if (e0 >= 0 && e0 < m) w1side[e0] = wv[u].x;
                    if (e0 + 1 >= 0 && e0 + 1 < m) w1side[e0 + 1] = wv[u].y;
                }
            }
        }
        if (MP > 0 && bd) {
            // traw[r] += (B D)_r . w_new over the u rows; B D is stored PLANAR (row r = one dense
            // vector of stride ldb), so these are m more perfectly coalesced streams
            if (packed) {  // m/2 parity-interleaved planes: .x belongs to row 2q, .y to row 2q+1
#pragma unroll
                for (int q = 0; q < MP / 2; ++q) {
                    if (2 * q < m) {
                        double2 e[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) e[u] = ld2s<NT>(bd + (size_t)q * ldb, idx[u]);
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            if (ok[u]) {
                                if (2 * idx[u] < n_bd) tacc[2 * q] += e[u].x * wv[u].x;
                                if (2 * idx[u] + 1 < n_bd) tacc[2 * q + 1] += e[u].y * wv[u].y;
                            }
                        }
                    }
                }
            } else {
#pragma unroll
            for (int r = 0; r < MP; ++r) {
                if (r < m) {
                    double2 e[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) e[u] = ld2s<NT>(bd + (size_t)r * ldb, idx[u]);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (ok[u]) {
                            if (2 * idx[u] < n_bd) tacc[r] += e[u].x * wv[u].x;
                            if (2 * idx[u] + 1 < n_bd) tacc[r] += e[u].y * wv[u].y;
                        }
                    }
                }
            }
            }
        }
    }
    if (!out) return;  // caller does not want the norm
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        const double s = wave_sum(nrm);
        if (lane == 0) red[wave * NR] = s;
    }
#pragma unroll
    for (int r = 0; r < MP; ++r) {
        const double s = wave_sum(tacc[r]);
        if (lane == 0) red[wave * NR + 1 + r] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < NR) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < W; ++j) t += red[j * NR + threadIdx.x];
        if ((int)threadIdx.x <= m) publish(partials + (size_t)blockIdx.x * kPartialLd + threadIdx.x, t);
    }
    if (!arrive_last(gridDim.x)) return;
    const int k = 1 + (MP > 0 ? m : 0);
    final_reduce(partials, gridDim.x, kPartialLd, k, red, fe);
    if (ar.P) peer_allreduce_block(ar, red, k, out);
    else if ((int)threadIdx.x < k) out[threadIdx.x] = red[threadIdx.x];
}

template <int T, int U, int G>
static void maxpy_launch(int mp, int grid, hipStream_t s, const double *V, int64_t ldv, int nv, const int32_t *nv_dev,
                         const double *a, double sign, double *w, int64_t n2, int64_t n_dot, const Finish &f,
                         const double *bd, int64_t ldb, int64_t n_bd, int m, double *w1side, const PythArgs &py,
                         int packed, const int32_t *done)
{
#define SPK_MAXPY(MPP) hipLaunchKernelGGL((maxpy_kernel<T, G, true, MPP, U>), dim3(grid), dim3(T), 0, s, V, ldv, nv, nv_dev, a, \
                                          sign, w, n2, n_dot, f.partials, f.out, bd, ldb, n_bd, m, w1side, py, f.ar, packed, \
                                          FinErr{f.err, f.fin_ticks}, done)
    if (mp == 4) SPK_MAXPY(4);
    else if (mp == 8) SPK_MAXPY(8);
    else SPK_MAXPY(0);
#undef SPK_MAXPY
}

void maxpy(const double *V, int64_t ldv, int nv, const int32_t *nv_dev, const double *a,
           double coef_sign, double *w, int64_t n, int64_t n_dot, const Finish &f,
           const int32_t *done, hipStream_t s, const double *bd, int64_t ldb, int64_t n_bd, int m, double *w1side,
           const PythArgs *pyth, int packed)
{
    const int64_t n2 = (n + 1) / 2;
    const VecShape vs = vec_shape(n2, true);
    static const int deep = [] { const char *e = getenv("SPK_VEC_DEEP"); return e ? atoi(e) : 0; }();
    PythArgs py{};
    py.m = -1;
    if (pyth) py = *pyth;
    // MP > 0 also switches on the lambda side copy; in single-reduction mode bd is not read
    const int mp = ((bd || pyth) && m > 0) ? (m <= 4 ? 4 : 8) : 0;
#define SPK_MAXPY_ARGS mp, vs.grid, s, V, ldv, nv, nv_dev, a, coef_sign, w, n2, n_dot, f, bd, ldb, n_bd, m, w1side, py, packed, done
    if (vs.T == 512) maxpy_launch<512, 4, 4>(SPK_MAXPY_ARGS);
    else if (vs.U == 4) maxpy_launch<256, 4, 4>(SPK_MAXPY_ARGS);
    // thin forms (small vectors), SPK_VEC_DEEP=1 only: the whole basis in ONE group of loads.  Measured SLOWER
    // on the 1/8 slab (30 vectors: 13.2 us against 11.6 with groups of 8; 214 VGPRs leave two waves per SIMD):
    // kept as a knob, off
    else if (vs.U == 2 && vs.G == 8 && nv > 8 && deep) maxpy_launch<256, 2, 16>(SPK_MAXPY_ARGS);
    else if (vs.U == 2 && vs.G == 8) maxpy_launch<256, 2, 8>(SPK_MAXPY_ARGS);
    else if (vs.U == 2) maxpy_launch<256, 2, 4>(SPK_MAXPY_ARGS);
    else if (vs.G == 8 && nv > 16 && deep) maxpy_launch<256, 1, 32>(SPK_MAXPY_ARGS);
    else if (vs.G == 8 && nv > 8 && deep) maxpy_launch<256, 1, 16>(SPK_MAXPY_ARGS);
    else if (vs.G == 8) maxpy_launch<256, 1, 8>(SPK_MAXPY_ARGS);
    else maxpy_launch<256, 1, 4>(SPK_MAXPY_ARGS);
#undef SPK_MAXPY_ARGS
}

// ---------------------------------------------------------------------------
// level-1 streams
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void scale_dev_kernel(double *__restrict__ x, int64_t n2,
                                                             const double *__restrict__ alpha,
                                                             const int32_t *__restrict__ done)
{
    if (done && *done) return;
    const double a = *alpha;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kThreads) {
        double2 v = reinterpret_cast<double2 *>(x)[i];
        v.x *= a;
        v.y *= a;
        reinterpret_cast<double2 *>(x)[i] = v;
    }
}
void scale_dev(double *x, int64_t n, const double *alpha_dev, const int32_t *done, hipStream_t s)
{
    const int64_t n2 = (n + 1) / 2;
    const int grid = (int)std::min<int64_t>((n2 + kThreads - 1) / kThreads, kMaxBlocks * 4);
    hipLaunchKernelGGL(scale_dev_kernel, dim3(grid > 0 ? grid : 1), dim3(kThreads), 0, s, x, n2, alpha_dev, done);
}

__global__ __launch_bounds__(kThreads) void axpby_kernel(double a, const double *__restrict__ x,
                                                         double b, double *__restrict__ y, int64_t n2,
                                                         const int32_t *__restrict__ done)
{
    if (done && *done) return;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kThreads) {
        const double2 xv = reinterpret_cast<const double2 *>(x)[i];
        double2 yv;
        if (b == 0.0) {
            yv.x = a * xv.x;
            yv.y = a * xv.y;
        } else {
            yv = reinterpret_cast<double2 *>(y)[i];
            yv.x = a * xv.x + b * yv.x;
            yv.y = a * xv.y + b * yv.y;
        }
        reinterpret_cast<double2 *>(y)[i] = yv;
    }
}
void axpby(double a, const double *x, double b, double *y, int64_t n, const int32_t *done, hipStream_t s)
{
    const int64_t n2 = (n + 1) / 2;
    const int grid = (int)std::min<int64_t>((n2 + kThreads - 1) / kThreads, kMaxBlocks * 4);
    hipLaunchKernelGGL(axpby_kernel, dim3(grid > 0 ? grid : 1), dim3(kThreads), 0, s, a, x, b, y, n2, done);
}

__global__ __launch_bounds__(kVT) void sqnorm_kernel(const double *__restrict__ x, int64_t n2,
                                                          int64_t n_dot, double *__restrict__ partials,
                                                          double *__restrict__ out, FinErr fe,
                                                          const int32_t *__restrict__ done)
{
    if (done && *done) return;
    __shared__ double red[kVT];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kVT + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kVT) {
        const double2 v = reinterpret_cast<const double2 *>(x)[i];
        if (2 * i < n_dot) acc += v.x * v.x;
        if (2 * i + 1 < n_dot) acc += v.y * v.y;
    }
    const double s = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < kVWaves; ++j) t += red[j];
        publish(partials + (size_t)blockIdx.x * kPartialLd, t);
    }
    if (!arrive_last(gridDim.x)) return;
    final_reduce(partials, gridDim.x, kPartialLd, 1, red, fe);
    if (threadIdx.x == 0) out[0] = red[0];
}
void sqnorm(const double *x, int64_t n_dot, const Finish &f, const int32_t *done, hipStream_t s)
{
    const int64_t n2 = (n_dot + 1) / 2;
    const int grid = vec_grid(n2);
    hipLaunchKernelGGL(sqnorm_kernel, dim3(grid), dim3(kVT), 0, s, x, n2, n_dot, f.partials, f.out, FinErr{f.err, f.fin_ticks}, done);
}

__global__ __launch_bounds__(kThreads) void gather_kernel(const double *__restrict__ x,
                                                          const int32_t *__restrict__ idx, int64_t n,
                                                          double *__restrict__ out,
                                                          const int32_t *__restrict__ done)
{
    if (done && *done) return;
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i < n) out[i] = x[idx[i]];
}
void gather(const double *x, const int32_t *idx, int64_t n, double *out, const int32_t *done, hipStream_t s)
{
    if (n == 0) return;
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, x, idx, n, out, done);
}

// ---------------------------------------------------------------------------
// peer-store collectives: stand-alone launches (primitives: top of this file)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(2 * 64) void peer_allreduce_kernel(PeerAR a, double *buf, int count)
{
    peer_allreduce_block(a, buf, count, buf);
}
void peer_allreduce(const PeerAR &a, double *buf, int count, hipStream_t s)
{
    if (count < 1 || 2 * count > kArGranules) fail(SPK_ERR_COMM, "peer all-reduce: %d values (1..%d)", count, kArGranules / 2);
    hipLaunchKernelGGL(peer_allreduce_kernel, dim3(1), dim3(128), 0, s, a, buf, count);
}

// Loop-back form for tests: workgroup r plays rank r of a P-rank all-reduce through P windows that all live
// in this process (spk_debug_peer_allreduce_loopback): every lane of the window layout is exercised on one
// device, including lanes 4..7 that a box with at most six GPU processes cannot reach otherwise.
__global__ __launch_bounds__(2 * 64) void peer_allreduce_loopback_kernel(PeerAR a, double *buf, int count)
{
    a.me = (int)blockIdx.x;
    peer_allreduce_block(a, buf + (size_t)a.me * 64, count, buf + (size_t)a.me * 64);
}
void peer_allreduce_loopback(const PeerAR &a, double *buf, int count, hipStream_t s)
{
    if (count < 1 || 2 * count > kArGranules || a.P < 1 || a.P > kPeerMax) fail(SPK_ERR_ARG, "loop-back all-reduce: bad shape");
    hipLaunchKernelGGL(peer_allreduce_loopback_kernel, dim3(a.P), dim3(128), 0, s, a, buf, count);
}

// one thread per granule: send first, then wait for the granule with the same index of my own staging
__global__ __launch_bounds__(kThreads) void peer_exchange_kernel(PeerHalo h, const double *__restrict__ sendbuf,
                                                                 double *__restrict__ recvbuf)
{
    const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t ns2 = 2 * h.send_off[h.npeers], nr2 = 2 * h.recv_off[h.npeers];
    if (g < ns2) {
        const int64_t e = g >> 1;
        int p = 0;
        while (p + 1 < h.npeers && e >= h.send_off[p + 1]) ++p;
        const uint32_t half = reinterpret_cast<const uint32_t *>(sendbuf)[g];
        st_sys(h.remote[p] + (g - 2 * h.send_off[p]), ((unsigned long long)h.seq << 32) | half);
    }
    if (g < nr2) {
        uint32_t lo;
        const unsigned long long tw0 = (h.stats && threadIdx.x == 0) ? wall_clock64() : 0ull;
        const bool ok = granule_wait(h.mine + g, h.seq, h.timeout_ms, lo, h.err);
        if (h.stats && threadIdx.x == 0) {
            atomicAdd(h.stats + 2 * kStatHalo, wall_clock64() - tw0);
            atomicAdd(h.stats + 2 * kStatHalo + 1, 1ull);
        }
        const uint32_t other = __shfl_xor(lo, 1, kWave);
        if (!(g & 1)) recvbuf[g >> 1] = join_halves(lo, other);
        if (!ok) raise_comm_error(h.err, 19, h.seq);
    }
}
void peer_exchange(const PeerHalo &h, const double *sendbuf, double *recvbuf, hipStream_t s)
{
    const int64_t g = 2 * std::max(h.send_off[h.npeers], h.recv_off[h.npeers]);
    if (g == 0) return;
    hipLaunchKernelGGL(peer_exchange_kernel, dim3((unsigned)((g + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, h,
                       sendbuf, recvbuf);
}

// Bulk form (PeerBulk): one workgroup per chunk of kBulkChunk doubles.  Send workgroups come first in
// the grid: copy the chunk into the peer's staging, release at system scope, then ONE flag store.
// Receive workgroups wait for their chunk's flag, acquire, copy the chunk out.
__global__ __launch_bounds__(kThreads) void peer_exchange_bulk_kernel(PeerBulk h, const double *__restrict__ sendbuf,
                                                                      double *__restrict__ recvbuf)
{
    __shared__ int okf;
    const int nsend = h.send_chunk0[h.npeers];
    int b = blockIdx.x;
    if (b < nsend) {
        int i = 0;
        while (i + 1 < h.npeers && b >= h.send_chunk0[i + 1]) ++i;
        const int64_t e0 = (int64_t)(b - h.send_chunk0[i]) * kBulkChunk;
        const int64_t len = h.send_off[i + 1] - h.send_off[i];
        const int64_t n = len - e0 < kBulkChunk ? len - e0 : kBulkChunk;
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(sendbuf + h.send_off[i] + e0);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(h.rdata[i] + e0);
        for (int64_t j = threadIdx.x; j < n; j += kThreads) st_sys(dst + j, src[j]);
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) st_sys(h.rflag[i] + e0, (unsigned long long)h.seq);
        return;
    }
    b -= nsend;
    int i = 0;
    while (i + 1 < h.npeers && b >= h.recv_chunk0[i + 1]) ++i;
    const int64_t e0 = h.recv_off[i] + (int64_t)(b - h.recv_chunk0[i]) * kBulkChunk;
    const int64_t rem = h.recv_off[i + 1] - e0;
    const int64_t n = rem < kBulkChunk ? rem : kBulkChunk;
    if (threadIdx.x == 0) {
        bool ok = true;
        const unsigned long long tw0 = h.stats ? wall_clock64() : 0ull;
        if (ld_sys(h.mflag + e0) != (unsigned long long)h.seq) {
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                __builtin_amdgcn_s_sleep(4);
                if (ld_sys(h.mflag + e0) == (unsigned long long)h.seq) break;
                if (__hip_atomic_load(h.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                    wall_clock64() - t0 > (unsigned long long)h.timeout_ms * 100000ull) {
                    ok = false;
                    break;
                }
            }
        }
        if (h.stats) {
            atomicAdd(h.stats + 2 * kStatHalo, wall_clock64() - tw0);
            atomicAdd(h.stats + 2 * kStatHalo + 1, 1ull);
        }
        if (!ok) raise_comm_error(h.err, 20, h.seq);
        okf = ok;
    }
    __syncthreads();
    if (!okf) return;
    __threadfence_system();
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(h.mdata + e0);
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(recvbuf + e0);
    for (int64_t j = threadIdx.x; j < n; j += kThreads) dst[j] = ld_sys(src + j);
}
void peer_exchange_bulk(const PeerBulk &h, const double *sendbuf, double *recvbuf, hipStream_t s)
{
    const int grid = h.send_chunk0[h.npeers] + h.recv_chunk0[h.npeers];
    if (grid == 0) return;
    hipLaunchKernelGGL(peer_exchange_bulk_kernel, dim3((unsigned)grid), dim3(kThreads), 0, s, h, sendbuf, recvbuf);
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

// ---------------------------------------------------------------------------
// preconditioner pieces
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void jacobi_kernel(const double *__restrict__ dinv,
                                                          const double *__restrict__ x,
                                                          double *__restrict__ y, int64_t n,
                                                          const int32_t *__restrict__ done)
{
    if (done && *done) return;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads)
        y[i] = x[i] * dinv[i];
}
void jacobi(const double *dinv, const double *x, double *y, int64_t n, const int32_t *done, hipStream_t s)
{
    if (n == 0) return;
    const int grid = (int)std::min<int64_t>((n + kThreads - 1) / kThreads, kMaxBlocks * 4);
    hipLaunchKernelGGL(jacobi_kernel, dim3(grid), dim3(kThreads), 0, s, dinv, x, y, n, done);
}

// PCJACOBI set-up: inverse diagonal, zero -> 1
__global__ __launch_bounds__(kThreads) void extract_diag_inv_kernel(const int32_t *__restrict__ rowptr,
                                                                    const int32_t *__restrict__ colidx,
                                                                    const double *__restrict__ val,
                                                                    int nrows, double *__restrict__ dinv)
{
    const int r = blockIdx.x * kThreads + threadIdx.x;
    if (r >= nrows) return;
    double d = 0.0;
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k)
        if (colidx[k] == r) d = val[k];
    dinv[r] = (d == 0.0) ? 1.0 : 1.0 / d;
}
void extract_diag_inv(const CsrDev &A, double *dinv, hipStream_t s)
{
    if (A.nrows == 0) return;
    hipLaunchKernelGGL(extract_diag_inv_kernel, dim3((A.nrows + kThreads - 1) / kThreads), dim3(kThreads),
                       0, s, A.rowptr.p, A.colidx.p, A.val.p, A.nrows, dinv);
}

// mode 0:  y0 = dinv .* (x0 - Bt y1)          (UPPER)
// mode 1:  y0 = dinv .* x0 - dinv .* (Bt y1)  (FULL, third step)
__global__ __launch_bounds__(kThreads) void bt_update_kernel(int mode,
                                                             const int32_t *__restrict__ rowptr,
                                                             const int32_t *__restrict__ colidx,
                                                             const double *__restrict__ val, int nrows,
                                                             const double *__restrict__ dinv,
                                                             const double *__restrict__ x0,
                                                             const double *__restrict__ y1,
                                                             double *__restrict__ y0,
                                                             const int32_t *__restrict__ done)
{
    if (done && *done) return;
    for (int r = blockIdx.x * kThreads + threadIdx.x; r < nrows; r += gridDim.x * kThreads) {
        double c = 0.0;
        for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) c += val[k] * y1[colidx[k]];
        if (mode >= 2) {  // vectors handed to the inner solve: x0 - Bt y1 (2), Bt y1 (3)
            y0[r] = mode == 2 ? x0[r] - c : c;
            continue;
        }
        const double d = dinv[r], xv = x0[r];
        y0[r] = (mode == 0) ? (xv - c) * d : xv * d - c * d;
    }
}
void bt_update(int mode, const CsrDev &Bt, const double *dinv, const double *x0, const double *y1,
               double *y0, const int32_t *done, hipStream_t s)
{
    if (Bt.nrows == 0) return;
    const int grid = std::min((Bt.nrows + kThreads - 1) / kThreads, kMaxBlocks * 4);
    hipLaunchKernelGGL(bt_update_kernel, dim3(grid), dim3(kThreads), 0, s, mode, Bt.rowptr.p, Bt.colidx.p,
                       Bt.val.p, Bt.nrows, dinv, x0, y1, y0, done);
}

// the m-vector step of PCApply_FieldSplit_Schur with S~ = -S^:
//   DIAG : y1 =  x1 / S^        LOWER/FULL: y1 = -(x1 - t) / S^     UPPER: y1 = -x1 / S^
__global__ void schur_y1_kernel(int fact, int m, const double *__restrict__ x1,
                                const double *__restrict__ t, const double *__restrict__ shat,
                                double *__restrict__ y1, const int32_t *__restrict__ done)
{
    if (done && *done) return;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    double v;
    if (fact == SPK_SCHUR_DIAG) v = x1[r] / shat[r];
    else if (fact == SPK_SCHUR_UPPER) v = -x1[r] / shat[r];
    else v = -(x1[r] - t[r]) / shat[r];
    y1[r] = v;
}
void schur_y1(int fact, int m, const double *x1, const double *t, const double *shat, double *y1,
              const int32_t *done, hipStream_t s)
{
    if (m == 0) return;
    hipLaunchKernelGGL(schur_y1_kernel, dim3((m + 63) / 64), dim3(64), 0, s, fact, m, x1, t, shat, y1, done);
}

__global__ void copy_small_kernel(const double *__restrict__ src, double *__restrict__ dst, int n,
                                  const int32_t *__restrict__ done)
{
    if (done && *done) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}
void copy_small(const double *src, double *dst, int n, const int32_t *done, hipStream_t s)
{
    if (n == 0) return;
    hipLaunchKernelGGL(copy_small_kernel, dim3(n > 4096 ? 16 : 1), dim3(n > 64 ? 256 : 64), 0, s, src, dst, n, done);
}

// dense scatter of one B row scaled by dinv (set-up of S^ and G only)
__global__ __launch_bounds__(kThreads) void scatter_row_kernel(const int32_t *__restrict__ colidx,
                                                               const double *__restrict__ val, int k0,
                                                               int k1, const double *__restrict__ dinv,
                                                               double *__restrict__ dense)
{
    const int k = k0 + blockIdx.x * kThreads + threadIdx.x;
    if (k < k1) dense[colidx[k]] = dinv ? val[k] * dinv[colidx[k]] : 0.0;
}
void scatter_row(const int32_t *colidx, const double *val, int k0, int k1, const double *dinv,
                 double *dense, hipStream_t s)
{
    if (k1 <= k0) return;
    hipLaunchKernelGGL(scatter_row_kernel, dim3((k1 - k0 + kThreads - 1) / kThreads), dim3(kThreads), 0, s,
                       colidx, val, k0, k1, dinv, dense);
}

// bd[r*ldb + i] = dinv_i * B_ri : the m rows of B D as dense vectors (planar; zero where B has no entry)
__global__ __launch_bounds__(kThreads) void build_bd_kernel(const int32_t *__restrict__ rowptr,
                                                            const int32_t *__restrict__ colidx,
                                                            const double *__restrict__ val, int nrows,
                                                            const double *__restrict__ dinv, int m, int64_t ldb,
                                                            double *__restrict__ bd)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= nrows) return;
    for (int r = 0; r < m; ++r) bd[(size_t)r * ldb + i] = 0.0;
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) bd[(size_t)colidx[k] * ldb + i] += val[k] * dinv[i];
}
void build_bd(const CsrDev &Bt, const double *dinv, int m, int64_t ldb, double *bd, hipStream_t s)
{
    if (Bt.nrows == 0) return;
    hipLaunchKernelGGL(build_bd_kernel, dim3((Bt.nrows + kThreads - 1) / kThreads), dim3(kThreads), 0, s,
                       Bt.rowptr.p, Bt.colidx.p, Bt.val.p, Bt.nrows, dinv, m, ldb, bd);
}

// Rows 2q and 2q+1 of B D often have DISJOINT support by parity -- row 2q lives on even vector entries
// (the x degrees of freedom of a dof-2 grid), row 2q+1 on odd ones (y): half of each dense row is
// zeros.  Then the two rows share one plane, bdp[q][i] = i even ? bd[2q][i] : bd[2q+1][i]: a double2
// load delivers (row 2q, row 2q+1) and the kernels stream m/2 planes instead of m.  Adding the
// products of the stored zeros changed nothing, so every sum keeps its bits.  *bad is raised when
// the structure does not hold (the dense rows are used then).
__global__ __launch_bounds__(kThreads) void pack_bd_kernel(const double *__restrict__ bd, int64_t ldb, int64_t n, int m,
                                                           double *__restrict__ bdp, int32_t *__restrict__ bad)
{
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    for (int q = 0; 2 * q + 1 < m; ++q) {
        const double a = bd[(size_t)(2 * q) * ldb + i], b = bd[(size_t)(2 * q + 1) * ldb + i];
        if ((i & 1) ? a != 0.0 : b != 0.0) *bad = 1;
        bdp[(size_t)q * ldb + i] = (i & 1) ? b : a;
    }
}
void pack_bd(const double *bd, int64_t ldb, int64_t n, int m, double *bdp, int32_t *bad, hipStream_t s)
{
    if (n == 0) return;
    hipLaunchKernelGGL(pack_bd_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, bd, ldb, n, m, bdp, bad);
}

// out[0] = r.r (first n_dot entries), out[1+q] = sum_i (B D)[i][q] r_i : cycle start of the fused path
template <int MP>
__global__ __launch_bounds__(512) void sqnorm_bd_kernel(const double *__restrict__ x, int64_t n2, int64_t n_dot,
                                                        const double *__restrict__ bd, int64_t ldb, int64_t n_bd,
                                                        int m, double *__restrict__ w1side,
                                                        double *__restrict__ partials, double *__restrict__ out,
                                                        FinErr fe, const int32_t *__restrict__ done)
{
    if (done && *done) return;
    constexpr int T = 512, NR = MP + 1, W = T / kWave;
    __shared__ double red[(W * NR > T) ? W * NR : T];
    double acc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < n2; i += (int64_t)gridDim.x * T) {
        const double2 v = reinterpret_cast<const double2 *>(x)[i];
        if (2 * i < n_dot) acc[0] += v.x * v.x;
        if (2 * i + 1 < n_dot) acc[0] += v.y * v.y;
        {
            const int64_t e0 = 2 * i - n_bd;
            if (e0 >= 0 && e0 < m) w1side[e0] = v.x;
            if (e0 + 1 >= 0 && e0 + 1 < m) w1side[e0 + 1] = v.y;
        }
#pragma unroll
        for (int r = 0; r < MP; ++r) {
            if (r < m) {
                const double2 e = reinterpret_cast<const double2 *>(bd + (size_t)r * ldb)[i];
                if (2 * i < n_bd) acc[1 + r] += e.x * v.x;
                if (2 * i + 1 < n_bd) acc[1 + r] += e.y * v.y;
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const double s = wave_sum(acc[r]);
        if (lane == 0) red[wave * NR + r] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < NR) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < W; ++j) t += red[j * NR + threadIdx.x];
        if ((int)threadIdx.x <= m) publish(partials + (size_t)blockIdx.x * kPartialLd + threadIdx.x, t);
    }
    if (!arrive_last(gridDim.x)) return;
    final_reduce(partials, gridDim.x, kPartialLd, 1 + m, red, fe);
    if ((int)threadIdx.x < 1 + m) out[threadIdx.x] = red[threadIdx.x];
}
void sqnorm_bd(const double *x, int64_t n, int64_t n_dot, const double *bd, int64_t ldb, int64_t n_bd, int m,
               double *w1side, const Finish &f, const int32_t *done, hipStream_t s)
{
    const int64_t n2 = (n + 1) / 2;
    const int grid = vec_grid(n2, 512);
    if (m <= 4)
        hipLaunchKernelGGL(sqnorm_bd_kernel<4>, dim3(grid), dim3(512), 0, s, x, n2, n_dot, bd, ldb, n_bd, m, w1side, f.partials, f.out, FinErr{f.err, f.fin_ticks}, done);
    else
        hipLaunchKernelGGL(sqnorm_bd_kernel<8>, dim3(grid), dim3(512), 0, s, x, n2, n_dot, bd, ldb, n_bd, m, w1side, f.partials, f.out, FinErr{f.err, f.fin_ticks}, done);
}

__global__ void sum_slots_kernel(const double *__restrict__ slots, int nslots, int ld, int count,
                                 double *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double s = 0.0;
    for (int r = 0; r < nslots; ++r) s += slots[(size_t)r * ld + i];
    out[i] = s;
}
void sum_slots(const double *slots, int nslots, int ld, int count, double *out, hipStream_t s)
{
    if (count == 0) return;
    hipLaunchKernelGGL(sum_slots_kernel, dim3((count + 63) / 64), dim3(64), 0, s, slots, nslots, ld, count, out);
}

// ---------------------------------------------------------------------------
// Krylov scalar work on the device (one thread): the host never waits for a
// Hessenberg entry, it only enqueues.  Semantics: PETSc KSPFGMRESCycle /
// KSPFGMRESUpdateHessenberg / KSPFGMRESBuildSoln / KSPConvergedDefault.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int converged_default(double rnorm, const KrylovState *st)
{
    if (isnan(rnorm) || isinf(rnorm)) return SPK_DIVERGED_NANORINF;
    if (rnorm <= st->ttol) return (rnorm < st->abstol) ? SPK_CONVERGED_ATOL : SPK_CONVERGED_RTOL;
    if (rnorm >= st->dtol * st->cnorm0) return SPK_DIVERGED_DTOL;
    return 0;
}

__global__ void krylov_init_kernel(KrylovArrays ka, spk_opts o, const double *bnorm2)
{
    if (threadIdx.x != 0) return;
    KrylovState *st = ka.st;
    st->its = 0;
    st->reason = 0;
    st->done = 0;
    st->loc_done = 0;
    st->max_it = o.max_it;
    st->restart = o.restart;
    st->hapend = 0;
    st->skip_refine = 1;
    st->skip_iter = 0;
    st->bnorm = sqrt(*bnorm2);
    st->abstol = o.abstol;
    st->dtol = o.dtol;
    st->rtol = o.rtol;
    st->guess_nonzero = o.guess_nonzero;
    st->ttol = fmax(o.rtol * st->bnorm, o.abstol);  // fixed at iteration 0 (krylov_cycle_begin)
    st->cnorm0 = st->bnorm;
    st->rnorm = 0.0;
    st->rnorm0 = 0.0;
    st->inv_tt = 1.0;
    st->tt = 0.0;
}
void krylov_init(const KrylovArrays &ka, const spk_opts &o, const double *bnorm2, hipStream_t s)
{
    hipLaunchKernelGGL(krylov_init_kernel, dim3(1), dim3(64), 0, s, ka, o, bnorm2);
}

__global__ void krylov_cycle_begin_kernel(KrylovArrays ka, const double *nrm2, double *tb, int m, double *sc)
{
    if (threadIdx.x != 0) return;
    KrylovState *st = ka.st;
    if (sc) sc[0] = 1.0;  // v_0 is normalised; later basis vectors carry their own scale (BA iteration)
    st->loc_done = 0;
    st->skip_iter = st->done;  // a cycle ended early by the recurrence starts afresh here
    if (st->done) return;
    const double rnorm = sqrt(*nrm2);
    st->rnorm = rnorm;
    if (st->its == 0) {
        // KSPConvergedDefault at iteration 0 (PETSc iterativ.c, as published): zero initial guess -> the
        // reference norm is the initial residual; -ksp_initial_guess_nonzero -> ||b||, or the initial
        // residual when b = 0.  ttol and the divergence test both refer to it.
        double snorm = rnorm;
        if (st->guess_nonzero) {
            snorm = st->bnorm;
            if (snorm == 0.0) snorm = rnorm;
        }
        st->rnorm0 = rnorm;
        st->cnorm0 = snorm;
        st->ttol = fmax(st->rtol * snorm, st->abstol);
        if (ka.hist_cap > 0) ka.hist[0] = rnorm;
    }
    int reason = converged_default(rnorm, st);
    if (!reason && st->its >= st->max_it) reason = SPK_DIVERGED_ITS;
    st->reason = reason;
    st->hapend = 0;
    if (reason) {
        st->done = 1;
        st->skip_iter = 1;
        return;
    }
    ka.rs[0] = rnorm;
    st->inv_tt = 1.0 / rnorm;
    if (tb)  // B D v_0 for the single-reduction recurrence
        for (int r = 0; r < m; ++r) tb[r] = nrm2[1 + r] / rnorm;
}
void krylov_cycle_begin(const KrylovArrays &ka, const double *nrm2, hipStream_t s, double *tb, int m, double *sc)
{
    hipLaunchKernelGGL(krylov_cycle_begin_kernel, dim3(1), dim3(64), 0, s, ka, nrm2, tb, m, sc);
}

// One Arnoldi step's scalar work (KSPFGMRESUpdateHessenberg + KSPConvergedDefault), run by a
// whole workgroup: the lanes stage the column and the stored rotations in LDS (parallel
// loads), lane 0 runs the dependent chain out of LDS and writes the column back once.
// Called from the stand-alone kernel (generic path) and from workgroup 0 of the fused
// iteration-head kernel, where it overlaps with that kernel's streaming.
// gate != nullptr (two words in LDS, zeroed by the caller): the words that GATE the kernels of an
// iteration (done, skip_iter) are not stored here but handed back as gate[0], gate[1]; the caller
// stores them once no workgroup of ITS launch can still be about to read them (kernel A of the
// two-launch iteration: its workgroups must all take the same branch, they feed one reduction).
// CAP: capacity of the LDS staging (restart + 2); the fused kernels carry the small instance, restart lengths beyond
// kMaxNv - 2 take the stand-alone kernel with the large one (krylov_givens)
template <int CAP>
__device__ void givens_block_t(const KrylovArrays &ka, int loc, const double *dots, const double *nrm2, int *gate)
{
    __shared__ double Hc[CAP], Hr[CAP], ccs[CAP], sss[CAP], sc[4];
    KrylovState *st = ka.st;
    if (st->done || st->skip_iter) return;  // uniform: read before anyone writes it
    const int ldh = ka.ldh;
    double *Hg = ka.H + (size_t)ldh * loc;  // column loc
    // every global value the serial chain needs is fetched here, in parallel, once
    for (int j = threadIdx.x; j <= loc; j += blockDim.x) {
        Hc[j] = dots[j];
        ccs[j] = ka.cc[j];
        sss[j] = ka.ss[j];
    }
    if (threadIdx.x == 32) sc[0] = *nrm2;
    if (threadIdx.x == 33) sc[1] = ka.rs[loc];
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double rs_loc = sc[1];
    const double tt = sqrt(sc[0]);
    if (isnan(tt) || isinf(tt)) {  // KSPCheckNorm: KSP_DIVERGED_NANORINF
        st->rnorm = tt;
        st->reason = SPK_DIVERGED_NANORINF;
        if (gate) gate[0] = gate[1] = 1;
        else st->done = 1, st->skip_iter = 1;
        return;
    }
    // happy breakdown test
    double hapbnd = fabs(tt / rs_loc);
    if (hapbnd > 1e-30) hapbnd = 1e-30;
    const int hapend = !(tt > hapbnd);
    st->tt = tt;
    st->inv_tt = hapend ? 1.0 : 1.0 / tt;
    // previous rotations on the new column.  The running entry stays in a register and the rotated
    // entries go to an array of their own, so the loop's LDS loads do not wait for its stores: the
    // serial chain is two FMAs per step (LDS round trips per step cost ~3 us at loc = 30, on the
    // critical path of the head kernel this step rides in)
    double run = Hc[0];
    for (int j = 1; j <= loc; ++j) {
        const double h1 = Hc[j], cj = ccs[j - 1], sj = sss[j - 1];
        Hr[j - 1] = cj * run + sj * h1;
        run = cj * h1 - sj * run;
    }
    Hr[loc] = run;
    Hr[loc + 1] = tt;
    double rnorm;
    int reason = 0;
    if (!hapend) {
        const double h0 = run, h1 = tt;
        const double d = sqrt(h0 * h0 + h1 * h1);
        if (d == 0.0) {
            st->reason = SPK_DIVERGED_NULL;
            if (gate) gate[0] = gate[1] = 1;
            else st->done = 1, st->skip_iter = 1;
            return;
        }
        const double c = h0 / d, sn = h1 / d;
        ka.cc[loc] = c;
        ka.ss[loc] = sn;
        ka.rs[loc + 1] = -sn * rs_loc;
        ka.rs[loc] = c * rs_loc;
        Hr[loc] = c * h0 + sn * h1;
        rnorm = fabs(sn * rs_loc);
    } else {
        rnorm = 0.0;
    }
    for (int j = 0; j <= loc + 1; ++j) Hg[j] = Hr[j];
    st->its += 1;
    st->loc_done = loc + 1;
    st->rnorm = rnorm;
    st->hapend = hapend;
    if (st->its < ka.hist_cap) ka.hist[st->its] = rnorm;
    reason = converged_default(rnorm, st);
    if (hapend && !reason) reason = SPK_DIVERGED_BREAKDOWN;
    if (!reason && st->its >= st->max_it) reason = SPK_DIVERGED_ITS;
    if (reason > 0 && ka.tentative) {
        // single-reduction mode: ||w'|| came out of a difference that can sit in rounding noise, so the
        // recurrence is trusted to END THE CYCLE only; the restart's true residual decides (krylov_cycle_begin)
        if (gate) gate[1] = 1;
        else st->skip_iter = 1;
        return;
    }
    st->reason = reason;
    if (reason) {
        if (gate) gate[0] = gate[1] = 1;
        else st->done = 1, st->skip_iter = 1;
    }
}

__device__ void givens_block(const KrylovArrays &ka, int loc, const double *dots, const double *nrm2, int *gate)
{
    givens_block_t<kMaxNv + 2>(ka, loc, dots, nrm2, gate);
}

__global__ void krylov_givens_kernel(KrylovArrays ka, int loc, const double *dots, const double *nrm2)
{
    givens_block(ka, loc, dots, nrm2);
}
__global__ __launch_bounds__(256) void krylov_givens_big_kernel(KrylovArrays ka, int loc, const double *dots, const double *nrm2)
{
    givens_block_t<kBigNv + 2>(ka, loc, dots, nrm2, nullptr);
}
__global__ __launch_bounds__(kThreads) void givens_rider_kernel(GivensRider gr, const int32_t *done)
{
    if (done && *done) return;
    givens_rider(gr);
}
void givens_rider_alone(const GivensRider &gr, const int32_t *done, hipStream_t s)  // a rank without rows: the rider without tiles
{
    hipLaunchKernelGGL(givens_rider_kernel, dim3(1), dim3(kThreads), 0, s, gr, done);
}
void krylov_givens(const KrylovArrays &ka, int loc, const double *dots, const double *nrm2, hipStream_t s)
{
    if (loc + 2 > kMaxNv + 2) hipLaunchKernelGGL(krylov_givens_big_kernel, dim3(1), dim3(256), 0, s, ka, loc, dots, nrm2);
    else hipLaunchKernelGGL(krylov_givens_kernel, dim3(1), dim3(64), 0, s, ka, loc, dots, nrm2);
}

// Head of a fused Schur iteration (one pass over the new basis vector):
//   v  = w' / ||w'||                          VecScale, in place
//   z0 = D v - (B D)^T y1  (FULL) | D v (LOWER), z1 = y1        -> Z_j    (PCApply_FieldSplit_Schur)
//   c  = B^T y1 = ((B D)^T y1) ./ dinv  -> pre-load of the SpMV output;  c1 = B z0 = t - G y1
// Every workgroup derives the m-vector data itself from the reduced scalars of the previous
// MAXPY pass (nrm[0] = ||w'||^2, nrm[1..m] = B D w'; w1raw = lambda part of w'):
//   x1 = w1raw/||w'||, t = B D v, y1 = -(x1 - t)/S^.
// Workgroup 0 additionally runs the Givens step of the PREVIOUS iteration (loc_prev >= 0),
// which therefore costs no launch and overlaps with the streaming of the other workgroups.
// Its `done` word may thus rise while this iteration's kernels are in flight: they then
// only write vectors nobody reads again, the iterate is frozen by loc_done.
template <int MP>
__global__ __launch_bounds__(kThreads) void fused_head_kernel(
    double *__restrict__ v, const double *__restrict__ nrm, const double *__restrict__ w1raw,
    const double *__restrict__ dinv, const double *__restrict__ bd, int64_t ldb,
    const double *__restrict__ shat, const double *__restrict__ gram, int fact, int64_t nl, int m,
    double *__restrict__ z, double *__restrict__ c, KrylovArrays ka, int loc_prev,
    const double *__restrict__ dots_prev, SendRanges sr, int packed, const int32_t *__restrict__ done)
{
    // packed: bd holds m/2 parity-interleaved planes (pack_bd_kernel) instead of m dense rows
    // c == nullptr: Jacobi head (K = A, m = 0): v = w'/||w'||, z = D v, nothing pre-loaded
    if (*done) return;
    __shared__ double ys[MP], xs[MP], ts[MP];
    const double tt = sqrt(nrm[0]);
    const double inv_tt = tt > 1e-300 ? 1.0 / tt : 1.0;
    if ((int)threadIdx.x < MP) {
        const int r = threadIdx.x;
        double x1 = 0.0, t = 0.0, y = 0.0;
        if (r < m) {
            x1 = w1raw[r] * inv_tt;
            t = nrm[1 + r] * inv_tt;
            y = -(x1 - t) / shat[r];
        }
        xs[r] = x1;
        ts[r] = t;
        ys[r] = y;
    }
    __syncthreads();
    double yv[MP];
#pragma unroll
    for (int r = 0; r < MP; ++r) yv[r] = ys[r];

    // workgroup 0 streams nothing: it writes the m multiplier entries and runs the Givens step of
    // the previous iteration -- a serial chain of a few microseconds that must not sit in front of
    // rows somebody waits for (the first tile carries the halo rows of the lower neighbour)
    if (blockIdx.x == 0) {
        if ((int)threadIdx.x < m) {
            const int r = threadIdx.x;
            double w1 = ts[r];
            if (fact == SPK_SCHUR_FULL)
                for (int q = 0; q < m; ++q) w1 -= gram[r * m + q] * ys[q];
            v[nl + r] = xs[r];
            z[nl + r] = ys[r];
            c[nl + r] = w1;
        }
        if (loc_prev >= 0) givens_block(ka, loc_prev, dots_prev, nrm);
        return;
    }
    const int bid = (int)blockIdx.x - 1;

    const int64_t n2 = nl / 2;  // nl is even on this path (checked by the host)
    // peer-store halo: workgroups past the main grid wait for this rank's ghost rows (sent by the
    // neighbours' head kernels) and unpack them for the SpMV that follows
    const int gmain = (int)gridDim.x - 1 - (sr.peer ? (2 * sr.nrecv + kThreads - 1) / kThreads : 0);
    if (bid >= gmain) {
        const int64_t g = (int64_t)(bid - gmain) * kThreads + threadIdx.x;
        if (g < 2 * (int64_t)sr.nrecv) {
            uint32_t lo;
            const unsigned long long tw0 = (sr.stats && threadIdx.x == 0) ? wall_clock64() : 0ull;
            const bool ok = granule_wait(sr.mine + g, sr.seq, sr.timeout_ms, lo, sr.err, done);
            if (sr.stats && threadIdx.x == 0) {  // one lane per waiting workgroup
                atomicAdd(sr.stats + 2 * kStatHalo, wall_clock64() - tw0);
                atomicAdd(sr.stats + 2 * kStatHalo + 1, 1ull);
            }
            const uint32_t other = __shfl_xor(lo, 1, kWave);
            if (!(g & 1)) sr.xghost[g >> 1] = join_halves(lo, other);
            if (!ok) raise_comm_error(sr.err, 16, sr.seq);
        }
        return;
    }
    // with a halo to send the grid is walked from both ends inwards, so that the rows the two slab
    // neighbours wait for leave first
    const int bx = sr.peer ? ((bid & 1) ? gmain - 1 - (bid >> 1) : (bid >> 1)) : bid;
    for (int64_t i = (int64_t)bx * kThreads + threadIdx.x; i < n2; i += (int64_t)gmain * kThreads) {
        double2 w = reinterpret_cast<double2 *>(v)[i];
        const double2 d = reinterpret_cast<const double2 *>(dinv)[i];
        w.x *= inv_tt;
        w.y *= inv_tt;
        double s0 = 0.0, s1 = 0.0;
        if (packed) {
#pragma unroll
            for (int q = 0; q < MP / 2; ++q) {
                if (2 * q < m) {
                    const double2 e = ld2s<true>(bd + (size_t)q * ldb, i);
                    s0 += e.x * yv[2 * q];
                    s1 += e.y * yv[2 * q + 1];
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < MP; ++r) {
                if (r < m) {
                    const double2 e = ld2s<true>(bd + (size_t)r * ldb, i);
                    s0 += e.x * yv[r];
                    s1 += e.y * yv[r];
                }
            }
        }
        double2 zz, cc;
        zz.x = w.x * d.x;
        zz.y = w.y * d.y;
        if (fact == SPK_SCHUR_FULL) {
            zz.x -= s0;
            zz.y -= s1;
        }
        reinterpret_cast<double2 *>(v)[i] = w;
        reinterpret_cast<double2 *>(z)[i] = zz;
        if (c) {
            cc.x = s0 / d.x;
            cc.y = s1 / d.y;
            reinterpret_cast<double2 *>(c)[i] = cc;
        }
        // rows a neighbour needs go straight into the packed halo buffer (no gather launch), or,
        // with the peer-store backend, as granules into the neighbour's own memory
        for (int q = 0; q < sr.n; ++q) {
            const int64_t e = 2 * i - sr.r0[q];
            if (sr.peer) {
                const unsigned long long tag = (unsigned long long)sr.seq << 32;
                if (e >= 0 && e < sr.len[q]) {
                    const unsigned long long b = (unsigned long long)__double_as_longlong(zz.x);
                    st_sys(sr.remote[q] + 2 * e, tag | (b & 0xffffffffull));
                    st_sys(sr.remote[q] + 2 * e + 1, tag | (b >> 32));
                }
                if (e + 1 >= 0 && e + 1 < sr.len[q]) {
                    const unsigned long long b = (unsigned long long)__double_as_longlong(zz.y);
                    st_sys(sr.remote[q] + 2 * e + 2, tag | (b & 0xffffffffull));
                    st_sys(sr.remote[q] + 2 * e + 3, tag | (b >> 32));
                }
            } else {
                if (e >= 0 && e < sr.len[q]) sr.buf[sr.off[q] + e] = zz.x;
                if (e + 1 >= 0 && e + 1 < sr.len[q]) sr.buf[sr.off[q] + e + 1] = zz.y;
            }
        }
    }
}
void fused_head(double *v, const double *nrm, const double *w1raw, const double *dinv, const double *bd, int64_t ldb,
                const double *shat, const double *gram, int fact, int64_t nl, int m, double *z, double *c,
                const KrylovArrays &ka, int loc_prev, const double *dots_prev, const int32_t *done, hipStream_t s,
                const SendRanges *srp, int packed)
{
    const int64_t n2 = nl / 2;
    int grid = (int)std::min<int64_t>((n2 + kThreads - 1) / kThreads, kMaxBlocks * 2);
    if (grid < 1) grid = 1;
    SendRanges sr{};
    if (srp) sr = *srp;
    grid += 1;                                                      // workgroup 0: scalars + Givens only
    if (sr.peer) grid += (2 * sr.nrecv + kThreads - 1) / kThreads;  // the waiting workgroups come last
    if (m <= 4)
        hipLaunchKernelGGL(fused_head_kernel<4>, dim3(grid > 0 ? grid : 1), dim3(kThreads), 0, s, v, nrm, w1raw, dinv, bd, ldb,
                           shat, gram, fact, nl, m, z, c, ka, loc_prev, dots_prev, sr, packed, done);
    else
        hipLaunchKernelGGL(fused_head_kernel<8>, dim3(grid > 0 ? grid : 1), dim3(kThreads), 0, s, v, nrm, w1raw, dinv, bd, ldb,
                           shat, gram, fact, nl, m, z, c, ka, loc_prev, dots_prev, sr, packed, done);
}

// ---------------------------------------------------------------------------
// Single-reduction iteration, second half and first half of the next one in ONE pass
// (opts.single_reduce = 1, fused Schur path): with h = V^T w, q = B D w and w.w already
// reduced over the ranks, everything the head of iteration loc+1 needs is known before
// the update of iteration loc starts:
//   ||w'||^2 = w.w - |h|^2,  B D w' = q - sum h_i (B D v_i),  lambda part of w' (m entries,
//   recomputed by every workgroup from the m entries of the basis vectors)
// so MAXPY (w' = w - V h), VecScale (v = w'/||w'||), PCApply_FieldSplit_Schur (z) and the
// B^T part of the next operator product (c) stream the vector once, and the Givens step of
// iteration loc runs in workgroup 0 of the same launch.  An iteration is then three launches
// (this, SpMV, MDot) with one reduction.  Same arithmetic per entry as maxpy_kernel followed by
// fused_head_kernel.
// ---------------------------------------------------------------------------
template <int T, int G, int U, int MP>
__global__ __launch_bounds__(T) void maxpy_head_kernel(
    const double *__restrict__ V, int64_t ldv, int nv, const double *__restrict__ dots, double *__restrict__ tb,
    double *__restrict__ nrm_out, double *__restrict__ w, const double *__restrict__ dinv,
    const double *__restrict__ bd, int64_t ldb, const double *__restrict__ shat, const double *__restrict__ gram,
    int fact, int64_t nl, int m, double *__restrict__ z, double *__restrict__ c, double *__restrict__ w1side,
    const double *__restrict__ wl_in, double *__restrict__ wl_out, KrylovArrays ka, int loc, SendRanges sr,
    int packed, const int32_t *__restrict__ done)
{
    // wl_in: the m lambda entries of w (= what the previous head wrote into c[nl..]; the SpMV does not
    // touch them) as a side copy -- workgroup 0 overwrites w[nl..] with the normalised entries while the
    // other workgroups still need the raw ones; wl_out: the same for the next iteration
    if (*done) return;
    __shared__ double hs[kMaxNv], lam[kMaxNv * 8], ys[8], xs[8], ts[8], sc[2 + 8];
    // ---- scalars, derived by every workgroup ----
    if (threadIdx.x < kWave) {  // lane i owns basis vector i (nv <= 63)
        const int i = threadIdx.x;
        const double hi = i < nv ? dots[i] : 0.0;
        if (i < nv) hs[i] = hi;
        const double hh = wave_sum(hi * hi);
        double tsum[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) tsum[r] = r < m ? wave_sum(i < nv ? hi * tb[i * 8 + r] : 0.0) : 0.0;
        if (i == 0) {
            const double ww = dots[nv + m];
            double tt2 = ww - hh;
            if (!(tt2 > 1.5e-14 * ww)) tt2 = 1.5e-14 * ww;  // noise floor of the difference, see maxpy_kernel
            sc[0] = tt2;
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (r < m) sc[2 + r] = dots[nv + r] - tsum[r];  // B D w'
        }
    }
    for (int t = threadIdx.x; t < nv * m; t += T) lam[t] = V[(size_t)(t / m) * ldv + nl + (t % m)];
    __syncthreads();
    const double tt2 = sc[0];
    const double tt = sqrt(tt2);
    const double inv_tt = tt > 1e-300 ? 1.0 / tt : 1.0;
    if ((int)threadIdx.x < MP) {
        const int r = threadIdx.x;
        double x1 = 0.0, t = 0.0, y = 0.0, wraw = 0.0;
        if (r < m) {
            wraw = wl_in[r];
            for (int i = 0; i < nv; ++i) wraw += -hs[i] * lam[i * m + r];  // the MAXPY of the lambda entries
            x1 = wraw * inv_tt;
            t = sc[2 + r] * inv_tt;
            y = -(x1 - t) / shat[r];
        }
        xs[r] = x1;
        ts[r] = t;
        ys[r] = y;
        if (blockIdx.x == 0 && r < m) w1side[r] = wraw;
    }
    __syncthreads();
    double yv[MP];
#pragma unroll
    for (int r = 0; r < MP; ++r) yv[r] = ys[r];

    // workgroup 0: the m multiplier entries, the recurrence data of the next iteration, Givens
    if (blockIdx.x == 0) {
        if ((int)threadIdx.x < m) {
            const int r = threadIdx.x;
            double w1 = ts[r];
            if (fact == SPK_SCHUR_FULL)
                for (int q = 0; q < m; ++q) w1 -= gram[r * m + q] * ys[q];
            w[nl + r] = xs[r];
            z[nl + r] = ys[r];
            c[nl + r] = w1;
            wl_out[r] = w1;
            nrm_out[1 + r] = sc[2 + r];
            tb[nv * 8 + r] = tt2 > 0.0 ? sc[2 + r] * (1.0 / tt) : 0.0;
        }
        if (threadIdx.x == 0) nrm_out[0] = tt2;
        __syncthreads();
        givens_block(ka, loc, dots, nrm_out);
        return;
    }
    const int bid = (int)blockIdx.x - 1;
    const int64_t n2 = nl / 2;
    const int gmain = (int)gridDim.x - 1 - (sr.peer ? (2 * sr.nrecv + T - 1) / T : 0);
    if (bid >= gmain) {  // peer-store halo: unpack this rank's ghost rows (see fused_head_kernel)
        const int64_t g = (int64_t)(bid - gmain) * T + threadIdx.x;
        if (g < 2 * (int64_t)sr.nrecv) {
            uint32_t lo;
            const unsigned long long tw0 = (sr.stats && threadIdx.x == 0) ? wall_clock64() : 0ull;
            const bool ok = granule_wait(sr.mine + g, sr.seq, sr.timeout_ms, lo, sr.err, done);
            if (sr.stats && threadIdx.x == 0) {  // one lane per waiting workgroup
                atomicAdd(sr.stats + 2 * kStatHalo, wall_clock64() - tw0);
                atomicAdd(sr.stats + 2 * kStatHalo + 1, 1ull);
            }
            const uint32_t other = __shfl_xor(lo, 1, kWave);
            if (!(g & 1)) sr.xghost[g >> 1] = join_halves(lo, other);
            if (!ok) raise_comm_error(sr.err, 17, sr.seq);
        }
        return;
    }
    const int bx = sr.peer ? ((bid & 1) ? gmain - 1 - (bid >> 1) : (bid >> 1)) : bid;
    for (int64_t tile = bx; tile * (T * U) < n2; tile += gmain) {
        double2 wv[U], dv[U], sv[U];
        int64_t idx[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            idx[u] = tile * (T * U) + u * T + threadIdx.x;
            ok[u] = idx[u] < n2;
            if (!ok[u]) idx[u] = 0;
            wv[u] = ld2(w, idx[u]);
            dv[u] = ld2(dinv, idx[u]);
            sv[u].x = sv[u].y = 0.0;
        }
        if (packed) {
#pragma unroll
            for (int q = 0; q < MP / 2; ++q) {
                if (2 * q < m) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const double2 e = ld2s<true>(bd + (size_t)q * ldb, idx[u]);
                        sv[u].x += e.x * yv[2 * q];
                        sv[u].y += e.y * yv[2 * q + 1];
                    }
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < MP; ++r) {
                if (r < m) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const double2 e = ld2s<true>(bd + (size_t)r * ldb, idx[u]);
                        sv[u].x += e.x * yv[r];
                        sv[u].y += e.y * yv[r];
                    }
                }
            }
        }
        for (int g0 = 0; g0 < nv; g0 += G) {
            double2 t[G][U];
            double ai[G];
#pragma unroll
            for (int v = 0; v < G; ++v) {
                const bool live = g0 + v < nv;
                const int ic = live ? g0 + v : 0;
                ai[v] = live ? -hs[ic] : 0.0;
                const double *Vi = V + (size_t)ic * ldv;
#pragma unroll
                for (int u = 0; u < U; ++u) t[v][u] = ld2s<true>(Vi, live ? idx[u] : 0);
            }
#pragma unroll
            for (int v = 0; v < G; ++v) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    wv[u].x += ai[v] * t[v][u].x;
                    wv[u].y += ai[v] * t[v][u].y;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ok[u]) {
                const int64_t i = idx[u];
                double2 vn, zz, cc;
                vn.x = wv[u].x * inv_tt;
                vn.y = wv[u].y * inv_tt;
                zz.x = vn.x * dv[u].x;
                zz.y = vn.y * dv[u].y;
                if (fact == SPK_SCHUR_FULL) {
                    zz.x -= sv[u].x;
                    zz.y -= sv[u].y;
                }
                reinterpret_cast<double2 *>(w)[i] = vn;
                reinterpret_cast<double2 *>(z)[i] = zz;
                if (c) {  // nullptr: Jacobi head (K = A, m = 0), the next product is not pre-loaded
                    cc.x = sv[u].x / dv[u].x;
                    cc.y = sv[u].y / dv[u].y;
                    reinterpret_cast<double2 *>(c)[i] = cc;
                }
                for (int q = 0; q < sr.n; ++q) {
                    const int64_t e = 2 * i - sr.r0[q];
                    if (sr.peer) {
                        const unsigned long long tag = (unsigned long long)sr.seq << 32;
                        if (e >= 0 && e < sr.len[q]) {
                            const unsigned long long b = (unsigned long long)__double_as_longlong(zz.x);
                            st_sys(sr.remote[q] + 2 * e, tag | (b & 0xffffffffull));
                            st_sys(sr.remote[q] + 2 * e + 1, tag | (b >> 32));
                        }
                        if (e + 1 >= 0 && e + 1 < sr.len[q]) {
                            const unsigned long long b = (unsigned long long)__double_as_longlong(zz.y);
                            st_sys(sr.remote[q] + 2 * e + 2, tag | (b & 0xffffffffull));
                            st_sys(sr.remote[q] + 2 * e + 3, tag | (b >> 32));
                        }
                    } else {
                        if (e >= 0 && e < sr.len[q]) sr.buf[sr.off[q] + e] = zz.x;
                        if (e + 1 >= 0 && e + 1 < sr.len[q]) sr.buf[sr.off[q] + e + 1] = zz.y;
                    }
                }
            }
        }
    }
}
void maxpy_head(const double *V, int64_t ldv, int nv, const double *dots, double *tb, double *nrm_out, double *w,
                const double *dinv, const double *bd, int64_t ldb, const double *shat, const double *gram, int fact,
                int64_t nl, int m, double *z, double *c, double *w1side, const double *wl_in, double *wl_out,
                const KrylovArrays &ka, int loc, const int32_t *done, hipStream_t s, const SendRanges *srp, int packed)
{
    const int64_t n2 = nl / 2;
    SendRanges sr{};
    if (srp) sr = *srp;
    // thin workgroups below 0.5 M entries (as MAXPY), fat ones above
    const bool thin = n2 < (int64_t)kVecMaxBlocks * 2048;
    const int T = thin ? 256 : 512, U = thin ? (n2 < (int64_t)kVecMaxBlocks * 1024 ? 1 : 2) : 4;
    int64_t tiles = (n2 + (int64_t)T * U - 1) / ((int64_t)T * U);
    if (tiles < 1) tiles = 1;
    int grid = (int)std::min<int64_t>(tiles, thin ? 1024 : kVecMaxBlocks);
    grid += 1;
    if (sr.peer) grid += (2 * sr.nrecv + T - 1) / T;
#define SPK_MH(TT, GG, UU, MPP) hipLaunchKernelGGL((maxpy_head_kernel<TT, GG, UU, MPP>), dim3(grid), dim3(TT), 0, s, V, ldv, nv, dots, tb, \
                                                   nrm_out, w, dinv, bd, ldb, shat, gram, fact, nl, m, z, c, w1side, wl_in, wl_out, ka, loc, sr, packed, done)
    if (m <= 4) {
        if (!thin) SPK_MH(512, 4, 4, 4);
        else if (U == 2) SPK_MH(256, 8, 2, 4);
        else SPK_MH(256, 8, 1, 4);
    } else {
        if (!thin) SPK_MH(512, 4, 4, 8);
        else if (U == 2) SPK_MH(256, 8, 2, 8);
        else SPK_MH(256, 8, 1, 8);
    }
#undef SPK_MH
}

// ---------------------------------------------------------------------------
// Two-launch iteration (opts.iteration_form; the default below ~1 M rows, i.e. for a rank's slab of a
// strong-scaling run and the 256^2 / 512^2 grids).  On such vectors every kernel of the four-launch
// iteration costs ~5 us beyond its bytes (launch boundary, first-load latency, publish -> re-read of
// the reduction): 45-50 us against a 27 us byte floor on the 1/8 slab of the 1024^2 grid.  Same
// algorithm -- classical Gram-Schmidt with the norm taken directly from w', two reductions -- in TWO
// launches:
//
//   A  iter_spmv_mdot_kernel   w = s (A z~ + c~) row tile by row tile, and in the tile's epilogue, while
//                              its w values sit in LDS, their share of h = V^T w and q = B D w (VecMDot
//                              costs no launch and no second pass over w).  s = 1/||w'|| of the previous
//                              iteration: the normalisation (VecScale) of v and z rides here too.
//   B  iter_maxpy_uhead_kernel w' = w - V h (VecMAXPY) with ||w'||^2 from the same pass (VecNorm), and the
//                              next iteration's preconditioner and B^T product applied to the
//                              UN-normalised w' (both are linear): z~ = D w' - (B D)^T y~, c~ = B^T y~.
//                              y~ needs t~ = B D w', a reduction over the very vector this pass builds;
//                              it follows from q by linearity, t~ = q - sum_i h_i (B D v_i) -- an
//                              identity between sums of the same magnitude (no squares: none of the
//                              Pythagorean norm's cancellation), kept per basis vector in tb[].
//
// Workgroups keep their row tiles through a whole launch; the LAST workgroup streams nothing: it holds
// the m multiplier entries, runs the Givens step of the previous iteration while the others stream,
// and is the reducer (and the all-reducer across ranks) of the launch.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double ld1nt(const double *p) { return __builtin_nontemporal_load(p); }
template <int VW, int MINW>
__global__ __launch_bounds__(kThreads, MINW) void iter_spmv_mdot_kernel(IterA a)
{
    const int32_t dn = __builtin_nontemporal_load(a.done);  // looked at behind the first loads (see mdot_ws16_kernel)
    __shared__ double prod[kBTile * 4];
    __shared__ double wt[kThreads], vt[kThreads];
    const double scale = a.nrm2 ? inv_norm(a.nrm2[0]) : 1.0;
    const int nmain = 8 * a.slots;
    const int nv = a.nv, m = a.m;
    // the scalar workgroup: LAST where it is also the reducer (it has to wait for the others anyway), FIRST in the
    // three-launch form -- its Givens chain then runs beside the streaming instead of behind it
    const int scalar_wg = VW > 0 ? nmain : 0;
    const int bid = VW > 0 ? (int)blockIdx.x : (int)blockIdx.x - 1;
    if ((int)blockIdx.x == scalar_wg) {
        // ---- the scalar / reducing workgroup
        if (dn) return;
        double *lamw = wt;
        if ((int)threadIdx.x < m) {
            const int r = threadIdx.x;
            double wl;
            if (a.nrm2) {  // normalise what kernel B left un-normalised
                a.vcur[a.nl + r] *= scale;
                a.zdst[a.nl + r] = a.zsrc[a.nl + r] * scale;
                wl = a.w[a.nl + r] * scale;
                a.w[a.nl + r] = wl;
                a.tb[(size_t)(nv - 1) * 8 + r] *= scale;
            } else {
                wl = a.w[a.nl + r];
            }
            a.wl_out[r] = wl;
            lamw[r] = wl;
        }
        __shared__ int gate[2];
        if (threadIdx.x < 2) gate[threadIdx.x] = 0;
        __syncthreads();
        // Givens step of the previous iteration, while the others stream.  Its verdict (done / skip_iter) is
        // stored only at the very end: every workgroup of this launch has long passed its own look at `done`
        // by then -- they feed one reduction and must all take the same branch (under load the XCDs start
        // their workgroups at different times).
        if (a.loc_prev >= 0) givens_block(a.ka, a.loc_prev, a.dots_prev, a.nrm_prev, gate);
        __syncthreads();
        if (VW > 0) {  // VW == 0: the three-launch form, VecMDot is a launch of its own
            const int k = nv + m;
            final_reduce(a.partials, nmain, kPartialLd, k, prod, FinErr{a.err, a.fin_ticks});
            if (a.lam_in_dot && (int)threadIdx.x < nv) {  // multiplier entries of the inner products (rank 0 only)
                double sl = 0.0;
                for (int r = 0; r < m; ++r) sl += a.V[(size_t)threadIdx.x * a.ldv + a.nl + r] * lamw[r];
                prod[threadIdx.x] += sl;
            }
            __syncthreads();
            if (a.ar.P) peer_allreduce_block(a.ar, prod, k, a.out);
            else if ((int)threadIdx.x < k) a.out[threadIdx.x] = prod[threadIdx.x];
        }
        if (threadIdx.x == 0) {  // every partial has arrived: no workgroup of this launch reads the gate any more
            if (gate[0]) a.ka.st->done = 1;
            if (gate[1]) a.ka.st->skip_iter = 1;
        }
        return;
    }

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int np = a.packed ? m / 2 : m;  // streams of B D: dense rows, or parity-interleaved planes
    const int nvt = nv + np;
    const int per = (nvt + 3) >> 2;
    const int v0 = wave * per;
    const int cnt = (nvt - v0) < per ? (nvt - v0) : per;  // <= 0: a wave without vectors
    double acc[VW > 0 ? VW : 1];
#pragma unroll
    for (int i = 0; i < (VW > 0 ? VW : 1); ++i) acc[i] = 0.0;

    const int xcd = bid & 7, slot = bid >> 3;
    for (int tl = slot; tl < a.tiles_per_xcd; tl += a.slots) {
        const int t = xcd * a.tiles_per_xcd + tl;
        if (t >= a.ntiles) break;
        // one descriptor per tile {first block row, end block row, first block, end block}: one round trip where
        // tile_brow -> browptr was two
        const int4 td = a.tdesc[t];
        const int br0 = td.x, br1 = td.y;
        const int b0 = td.z, b1 = td.w;
        const int cntb = b1 - b0;
        const int nr = 2 * (br1 - br0), r0 = 2 * br0;
        const int lr = threadIdx.x;
        // ---- every load that depends on nothing computed in this tile is issued FIRST: the matrix stream, the
        // row thread's own operands, and the rows of the basis vectors the dot phase will need (they do not
        // depend on w).  The tile then costs two memory round trips (these, and the gather of x behind the
        // block columns) instead of one per phase -- with <= 4 workgroups per CU the phases of a tile are a
        // chain of latencies, not of bytes (first version: 27-38 us on the 1/8 slab against 10 + 12 for
        // SpMV and MDot as launches of their own).
        constexpr int kSteps = kBTile / kThreads;
        int c[kSteps];
        double2 tp[kSteps], bo[kSteps];
#pragma unroll
        for (int i = 0; i < kSteps; ++i) {
            const int q = i * kThreads + threadIdx.x;
            if (q < cntb) {
                c[i] = __builtin_nontemporal_load(a.bcol + b0 + q);
                tp[i] = ld2s<true>(a.vtop, b0 + q);
                bo[i] = ld2s<true>(a.vbot, b0 + q);
            }
        }
        double wpre = 0.0, vrow = 0.0, zrow = 0.0;
        int k0 = 0, k1 = 0, o0 = 0, o1 = 0;
        if (lr < nr) {
            const int br = br0 + (lr >> 1), r = r0 + lr;
            k0 = a.browptr[br] - b0;
            k1 = a.browptr[br + 1] - b0;
            if (a.acc) wpre = a.w[r];
            if (a.nrm2) {
                vrow = a.vcur[r];
                zrow = a.zsrc[r];
            }
            if (a.od.rowptr) {
                o0 = a.od.rowptr[r];
                o1 = a.od.rowptr[r + 1];
            }
        }
        constexpr int RL = 2;  // rows per lane whose basis entries are fetched ahead (tiles of <= 128 rows: all of them)
        double av[VW > 0 ? VW : 1][RL];
        if (VW > 0) {
#pragma unroll
            for (int v = 0; v < VW; ++v) {
                const int i = v0 + v;
                const bool live = v < cnt && !(a.nrm2 && i == nv - 1);  // the vector normalised here comes from LDS
                const double *src = i < nv ? a.V + (size_t)i * a.ldv : a.bd + (size_t)(i - nv) * a.ldb;
#pragma unroll
                for (int j = 0; j < RL; ++j) {
                    const int k = lane + 64 * j;
                    av[v][j] = (live && k < nr) ? ld1nt(src + r0 + k) : 0.0;
                }
            }
        }
        if (dn) return;  // (uniform over the launch)
        // phase 1: gather x 16 bytes at a time behind the block columns, products to LDS
#pragma unroll
        for (int i = 0; i < kSteps; ++i) {
            const int q = i * kThreads + threadIdx.x;
            if (q < cntb) {
                const double2 xv = reinterpret_cast<const double2 *>(a.zsrc)[c[i]];
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
        // phase 2: one thread per row, CSR order; scaling and normalisation of the row's entries
        if (lr < nr) {
            const int half = lr & 1;
            double sr = 0.0;
            for (int k = k0; k < k1; ++k) {
                const double2 p = *reinterpret_cast<const double2 *>(prod + 4 * k + 2 * half);
                sr += p.x;
                sr += p.y;
            }
            const int r = r0 + lr;
            for (int k = o0; k < o1; ++k) sr += a.od.val[k] * a.od.xg[a.od.colidx[k]];
            if (a.acc) sr += wpre;
            const double wv = sr * scale;
            a.w[r] = wv;
            wt[lr] = wv;
            if (a.nrm2) {
                const double vn = vrow * scale;
                a.vcur[r] = vn;
                vt[lr] = vn;
                a.zdst[r] = zrow * scale;
            }
        }
        __syncthreads();  // phase 2 has read prod; wt / vt are complete
        if (VW == 0) continue;  // three-launch form: VecMDot is a launch of its own
        // phase 3: this tile's share of V^T w and B D w; the waves split the VECTORS, lane k owns rows k, k+64, ..
        if (cnt > 0) {
            double wr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wr[j] = (lane + 64 * j) < nr ? wt[lane + 64 * j] : 0.0;
#pragma unroll
            for (int v = 0; v < VW; ++v) {
                const int i = v0 + v;
                if (v < cnt) {  // wave-uniform
                    if (a.nrm2 && i == nv - 1) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (64 * j < nr && lane + 64 * j < nr) acc[v] += vt[lane + 64 * j] * wr[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < RL; ++j) acc[v] += av[v][j] * wr[j];
                        if (nr > 64 * RL) {  // tiles beyond 128 rows (not the grids' 112): the remaining rows, fetched late
                            const double *src = i < nv ? a.V + (size_t)i * a.ldv : a.bd + (size_t)(i - nv) * a.ldb;
#pragma unroll
                            for (int j = RL; j < 4; ++j)
                                if (lane + 64 * j < nr) acc[v] += ld1nt(src + r0 + lane + 64 * j) * wr[j];
                        }
                    }
                }
            }
        }
        __syncthreads();  // prod / wt / vt are reused by the next tile
    }
    if (VW == 0 || dn) return;
    // this workgroup's partial sums: value i of [h_0..h_{nv-1}, q_0..q_{m-1}] comes from exactly one wave
    double *row = a.partials + (size_t)bid * kPartialLd;
#pragma unroll
    for (int il = 0; il < VW; ++il) {
        if (il < cnt) {  // wave-uniform
            const int i = v0 + il;
            if (i < nv || !a.packed) {
                const double sdot = wave_sum(acc[il]);
                if (lane == 0) publish(row + i, sdot);
            } else {  // a parity-interleaved plane: even rows belong to constraint row 2 pl, odd ones to 2 pl + 1
                const int pl = i - nv;
                const double se = wave_sum((lane & 1) ? 0.0 : acc[il]);
                const double so = wave_sum((lane & 1) ? acc[il] : 0.0);
                if (lane == 0) {
                    publish(row + nv + 2 * pl, se);
                    publish(row + nv + 2 * pl + 1, so);
                }
            }
        }
    }
}

// Workgroups a launch of kernel A runs per XCD: every slot gets the same number of tiles (+-1), all of
// them co-resident (wg_per_cu workgroups of 256 threads per CU: what the instantiation's registers allow)
int iter_slots(int tiles_per_xcd, int wg_per_cu)
{
    const int smax = 32 * wg_per_cu;  // CUs per XCD x workgroups per CU
    if (tiles_per_xcd <= smax) return tiles_per_xcd > 0 ? tiles_per_xcd : 1;
    const int tpw = (tiles_per_xcd + smax - 1) / smax;
    return (tiles_per_xcd + tpw - 1) / tpw;
}

void iter_spmv_mdot(const IterA &a0, hipStream_t s, bool dots)
{
    IterA a = a0;
    const int np = a.packed ? a.m / 2 : a.m;
    const int per = (a.nv + np + 3) / 4;
    if (a.nv + a.m > kMaxNv - 1) fail(SPK_ERR_ARG, "iter_spmv_mdot: %d values exceed one reduction", a.nv + a.m);
    static const int occ8 = [] { const char *e = getenv("SPK_ITERA_OCC"); return e ? atoi(e) : 4; }();
    // workgroups per XCD: one tile each without the dot phase; with it, as many as are co-resident (the
    // accumulators live across a workgroup's tiles), every slot the same number of tiles (+-1)
    const int occ = !dots ? 0 : (per <= 4 ? 4 : (per <= 8 ? (occ8 == 4 ? 4 : 3) : 2));
    a.slots = !dots ? (a.tiles_per_xcd > 0 ? a.tiles_per_xcd : 1) : iter_slots(a.tiles_per_xcd, occ);
    const dim3 grid(8 * a.slots + 1), block(kThreads);
    if (!dots) hipLaunchKernelGGL((iter_spmv_mdot_kernel<0, 4>), grid, block, 0, s, a);
    else if (per <= 4) hipLaunchKernelGGL((iter_spmv_mdot_kernel<4, 4>), grid, block, 0, s, a);
    else if (per <= 8 && occ8 == 4) hipLaunchKernelGGL((iter_spmv_mdot_kernel<8, 4>), grid, block, 0, s, a);
    else if (per <= 8) hipLaunchKernelGGL((iter_spmv_mdot_kernel<8, 3>), grid, block, 0, s, a);
    else if (per <= 12) hipLaunchKernelGGL((iter_spmv_mdot_kernel<12, 2>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((iter_spmv_mdot_kernel<16, 2>), grid, block, 0, s, a);
}

template <int T, int G, int U, int MP>
__global__ __launch_bounds__(T) void iter_maxpy_uhead_kernel(IterB b)
{
    const int32_t dn = __builtin_nontemporal_load(b.done);  // looked at behind the first loads (see mdot_ws16_kernel)
    __shared__ double hs[kMaxNv], lam[kMaxNv * 8], ys[8], wraws[8], tus[8];
    __shared__ double red[T];
    const int nv = b.nv, m = b.m;
    constexpr int NP = MP > 0 ? MP : 1;
    const int gmain = b.gmain;
    const int nhalo = b.sr.peer ? (2 * b.sr.nrecv + T - 1) / T : 0;
    const bool is_main = (int)blockIdx.x < gmain;
    const int64_t n2 = b.nl / 2;
    const int bid = blockIdx.x;
    // with a halo to send the grid is walked from both ends inwards (the rows the neighbours wait for leave first)
    const int bx = b.sr.peer ? ((bid & 1) ? gmain - 1 - (bid >> 1) : (bid >> 1)) : bid;

    // ---- a streaming workgroup puts the loads of its first tile in flight BEFORE the scalar prologue: w, D, the
    // planes of B D and the first group of basis vectors depend on none of it, and the prologue is a chain of two
    // memory round trips of its own (kernel of 16 us on the 1/8 slab, 8 us of them not bytes)
    constexpr bool PRE = U * (MP > 0 ? MP : 1) <= 16;  // planes of B D fetched ahead too, where the registers allow (not 512 x 4 x 8 rows)
    double2 wv[U], dv[U], pe[PRE ? NP : 1][U], t0[G][U];
    int64_t idx[U];
    bool ok[U];
    int64_t tile = bx;
    auto load_planes = [&]() {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const bool live = b.packed ? 2 * q < m : q < m;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                pe[q][u].x = pe[q][u].y = 0.0;
                if (live) pe[q][u] = ld2s<true>(b.bd + (size_t)q * b.ldb, idx[u]);
            }
        }
    };
    auto load_tile = [&](int64_t tl) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            idx[u] = tl * (T * U) + u * T + threadIdx.x;
            ok[u] = idx[u] < n2;
            if (!ok[u]) idx[u] = 0;
            wv[u] = ld2(b.w, idx[u]);
            dv[u] = ld2(b.dinv, idx[u]);
        }
        if (MP > 0 && PRE) load_planes();
#pragma unroll
        for (int v = 0; v < G; ++v) {
            const bool live = v < nv;
            const double *Vi = b.V + (size_t)(live ? v : 0) * b.ldv;
#pragma unroll
            for (int u = 0; u < U; ++u) t0[v][u] = ld2s<true>(Vi, live ? idx[u] : 0);
        }
    };
    bool have = is_main && tile * (T * U) < n2;
    if (have) load_tile(tile);

    // ---- scalars, derived by every workgroup from the reduced [h, q]; all their loads first
    double lamv = 0.0;
    const bool lam_mine = (int)threadIdx.x < nv * m;
    if (lam_mine) lamv = b.V[(size_t)(threadIdx.x / m) * b.ldv + b.nl + (threadIdx.x % m)];
    double wl = 0.0, sh = 1.0;
    if ((int)threadIdx.x < m) {
        wl = b.wl_in[threadIdx.x];
        sh = b.shat[threadIdx.x];
    }
    double hi_pre = 0.0, qv_pre = 0.0, tbv_pre[8], sci = 1.0;
    const double s_w = b.sc ? b.sc[nv - 1] : 1.0;   // w = s_w w~ (un-normalised basis); 1 otherwise
    if (threadIdx.x < kWave) {
        const int i = threadIdx.x;
        hi_pre = i < nv ? b.dots[i] : 0.0;
        if (b.sc) sci = i < nv ? b.sc[i] : 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) tbv_pre[r] = (r < m && i < nv) ? b.tb[i * 8 + r] : 0.0;
        qv_pre = (i < m) ? b.dots[nv + i] : 0.0;
    }
    if (dn) return;
    if (threadIdx.x < kWave) {  // lane i owns basis vector i (nv <= 63)
        const int i = threadIdx.x;
        // un-normalised basis: h_i = sc_i s_w (V~_i . w~); the MAXPY coefficient of V~_i and the weight of B D V~_i is h_i sc_i
        const double hi = b.sc ? sci * s_w * hi_pre : hi_pre;
        const double ci = b.sc ? hi * sci : hi;
        double tbv[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) tbv[r] = tbv_pre[r] * (b.sc ? sci : 1.0);
        const double qv = qv_pre * s_w;
        if (i < nv) hs[i] = ci;
        if (b.sc && (int)blockIdx.x == gmain + nhalo && i < nv) b.hbuf[i] = hi;  // the Hessenberg column (reducer only)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r < m) {  // uniform
                const double tsum = wave_sum(hi * tbv[r]);
                const double qr = __shfl(qv, r, kWave);
                if (i == 0) tus[r] = qr - tsum;  // B D w' = B D w - sum h_i (B D v_i)
            }
        }
    }
    if (lam_mine) lam[threadIdx.x] = lamv;
    for (int t = threadIdx.x + T; t < nv * m; t += T) lam[t] = b.V[(size_t)(t / m) * b.ldv + b.nl + (t % m)];
    __syncthreads();
    if ((int)threadIdx.x < NP && MP > 0) {
        const int r = threadIdx.x;
        double y = 0.0, wraw = 0.0;
        if (r < m) {
            wraw = s_w * wl;
            for (int i = 0; i < nv; ++i) wraw += -hs[i] * lam[i * m + r];  // the MAXPY of the multiplier entries
            y = -(wraw - tus[r]) / sh;
        }
        wraws[r] = wraw;
        ys[r] = y;
    }
    __syncthreads();
    double yv[NP];
#pragma unroll
    for (int r = 0; r < NP; ++r) yv[r] = MP > 0 ? ys[r] : 0.0;
    if (b.sc && have) {  // w = s_w w~
#pragma unroll
        for (int u = 0; u < U; ++u) {
            wv[u].x *= s_w;
            wv[u].y *= s_w;
        }
    }

    if ((int)blockIdx.x == gmain + nhalo) {
        // ---- the scalar / reducing workgroup: multiplier entries of w', z~, c~; B D w' for the recurrence
        if ((int)threadIdx.x < m) {
            const int r = threadIdx.x;
            double w1 = tus[r];
            if (b.fact == SPK_SCHUR_FULL)
                for (int q = 0; q < m; ++q) w1 -= b.gram[r * m + q] * ys[q];
            b.w[b.nl + r] = wraws[r];
            b.zun[b.nl + r] = ys[r];
            b.c[b.nl + r] = w1;
            b.tb[(size_t)nv * 8 + r] = tus[r];  // un-normalised; kernel A of the next iteration scales it (or nobody: b.sc)
            if (b.sc) b.wl_out[r] = w1;
        }
        __syncthreads();
        final_reduce(b.partials, gmain, kPartialLd, 1, red, FinErr{b.err, b.fin_ticks});
        if (threadIdx.x == 0) {
            double tot = red[0];
            for (int r = 0; r < m; ++r) tot += b.lam_in_dot ? wraws[r] * wraws[r] : 0.0;
            red[0] = tot;
        }
        __syncthreads();
        // (un-normalised basis: the new vector's scale factor and the Givens step of this iteration ride in the product
        // launch that follows (GivensRider); with ar_post_only that rider also collects the all-reduce posted here)
        if (b.ar.P && b.ar_post_only) peer_allreduce_post(b.ar, red, 1);
        else if (b.ar.P) peer_allreduce_block(b.ar, red, 1, b.out);
        else if (threadIdx.x == 0) b.out[0] = red[0];
        return;
    }
    if (!is_main) {  // peer-store halo: unpack this rank's ghost rows (see fused_head_kernel)
        const int64_t g = (int64_t)((int)blockIdx.x - gmain) * T + threadIdx.x;
        if (g < 2 * (int64_t)b.sr.nrecv) {
            uint32_t lo;
            const unsigned long long tw0 = (b.sr.stats && threadIdx.x == 0) ? wall_clock64() : 0ull;
            const bool okw = granule_wait(b.sr.mine + g, b.sr.seq, b.sr.timeout_ms, lo, b.sr.err, b.done);
            if (b.sr.stats && threadIdx.x == 0) {
                atomicAdd(b.sr.stats + 2 * kStatHalo, wall_clock64() - tw0);
                atomicAdd(b.sr.stats + 2 * kStatHalo + 1, 1ull);
            }
            const uint32_t other = __shfl_xor(lo, 1, kWave);
            if (!(g & 1)) b.sr.xghost[g >> 1] = join_halves(lo, other);
            if (!okw) raise_comm_error(b.sr.err, 18, b.sr.seq);
        }
        return;
    }
    double nrm = 0.0;
    while (have) {
        double2 sv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) sv[u].x = sv[u].y = 0.0;
        if (MP > 0 && !PRE) {  // fat workgroups with many rows: one plane at a time
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const bool live = b.packed ? 2 * q < m : q < m;
                if (live) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const double2 e = ld2s<true>(b.bd + (size_t)q * b.ldb, idx[u]);
                        if (b.packed) {
                            sv[u].x += e.x * yv[2 * q];
                            sv[u].y += e.y * yv[2 * q + 1];
                        } else {
                            sv[u].x += e.x * yv[q];
                            sv[u].y += e.y * yv[q];
                        }
                    }
                }
            }
        } else if (MP > 0) {
            if (b.packed) {
#pragma unroll
                for (int q = 0; q < NP / 2; ++q) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        sv[u].x += pe[q][u].x * yv[2 * q];
                        sv[u].y += pe[q][u].y * yv[2 * q + 1];
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < NP; ++r) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        sv[u].x += pe[r][u].x * yv[r];
                        sv[u].y += pe[r][u].y * yv[r];
                    }
                }
            }
        }
        // first group of basis vectors: already here
#pragma unroll
        for (int v = 0; v < G; ++v) {
            const double ai = v < nv ? -hs[v] : 0.0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                wv[u].x += ai * t0[v][u].x;
                wv[u].y += ai * t0[v][u].y;
            }
        }
        for (int g0 = G; g0 < nv; g0 += G) {
            double2 t[G][U];
            double ai[G];
#pragma unroll
            for (int v = 0; v < G; ++v) {
                const bool live = g0 + v < nv;
                const int ic = live ? g0 + v : 0;
                ai[v] = live ? -hs[ic] : 0.0;
                const double *Vi = b.V + (size_t)ic * b.ldv;
#pragma unroll
                for (int u = 0; u < U; ++u) t[v][u] = ld2s<true>(Vi, live ? idx[u] : 0);
            }
#pragma unroll
            for (int v = 0; v < G; ++v) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    wv[u].x += ai[v] * t[v][u].x;
                    wv[u].y += ai[v] * t[v][u].y;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ok[u]) {
                const int64_t i = idx[u];
                double2 zz, cc;
                nrm += wv[u].x * wv[u].x;
                nrm += wv[u].y * wv[u].y;
                zz.x = wv[u].x * dv[u].x;
                zz.y = wv[u].y * dv[u].y;
                if (MP > 0 && b.fact == SPK_SCHUR_FULL) {
                    zz.x -= sv[u].x;
                    zz.y -= sv[u].y;
                }
                reinterpret_cast<double2 *>(b.w)[i] = wv[u];
                reinterpret_cast<double2 *>(b.zun)[i] = zz;
                if (MP > 0) {
                    cc.x = sv[u].x / dv[u].x;
                    cc.y = sv[u].y / dv[u].y;
                    reinterpret_cast<double2 *>(b.c)[i] = cc;
                }
                for (int q = 0; q < b.sr.n; ++q) {
                    const int64_t e = 2 * i - b.sr.r0[q];
                    if (b.sr.peer) {
                        const unsigned long long tag = (unsigned long long)b.sr.seq << 32;
                        if (e >= 0 && e < b.sr.len[q]) {
                            const unsigned long long bits = (unsigned long long)__double_as_longlong(zz.x);
                            st_sys(b.sr.remote[q] + 2 * e, tag | (bits & 0xffffffffull));
                            st_sys(b.sr.remote[q] + 2 * e + 1, tag | (bits >> 32));
                        }
                        if (e + 1 >= 0 && e + 1 < b.sr.len[q]) {
                            const unsigned long long bits = (unsigned long long)__double_as_longlong(zz.y);
                            st_sys(b.sr.remote[q] + 2 * e + 2, tag | (bits & 0xffffffffull));
                            st_sys(b.sr.remote[q] + 2 * e + 3, tag | (bits >> 32));
                        }
                    } else {
                        if (e >= 0 && e < b.sr.len[q]) b.sr.buf[b.sr.off[q] + e] = zz.x;
                        if (e + 1 >= 0 && e + 1 < b.sr.len[q]) b.sr.buf[b.sr.off[q] + e + 1] = zz.y;
                    }
                }
            }
        }
        tile += gmain;
        have = tile * (T * U) < n2;
        if (have) {
            load_tile(tile);
            if (b.sc) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    wv[u].x *= s_w;
                    wv[u].y *= s_w;
                }
            }
        }
    }
    // ||w'||^2 of this workgroup's entries.  The partial goes to slot bx -- the FIRST TILE this workgroup
    // streamed -- so that the reducer adds the partials in tile order whichever way the grid was walked
    // (bit-identical norms with and without the peer-store halo in the same launch)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double sw = wave_sum(nrm);
    if (lane == 0) red[wave] = sw;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tsum = 0.0;
#pragma unroll
        for (int j = 0; j < T / kWave; ++j) tsum += red[j];
        publish(b.partials + (size_t)bx * kPartialLd, tsum);
    }
}

void iter_maxpy_uhead(IterB b, hipStream_t s)
{
    const int64_t n2 = b.nl / 2;
    // thin workgroups below 0.5 M entries (as MAXPY), fat ones above
    const bool thin = n2 < (int64_t)kVecMaxBlocks * 2048;
    static const int t128 = [] { const char *e = getenv("SPK_B_T128"); return e ? atoi(e) : 0; }();
    const int U = thin ? (n2 < (int64_t)kVecMaxBlocks * 1024 ? 1 : 2) : 4;
    // the smallest vectors (<= 1024 tiles of 128 double2): two-wave workgroups, twice the waves in flight per CU
    const bool tiny = thin && U == 1 && t128 && n2 <= (int64_t)1024 * 128;
    const int T = tiny ? 128 : (thin ? 256 : 512);
    int64_t tiles = (n2 + (int64_t)T * U - 1) / ((int64_t)T * U);
    if (tiles < 1) tiles = 1;
    b.gmain = (int)std::min<int64_t>(tiles, thin ? 1024 : kVecMaxBlocks);
    int grid = b.gmain + 1;
    if (b.sr.peer) grid += (2 * b.sr.nrecv + T - 1) / T;
    if (b.nv + b.m > kMaxNv - 1) fail(SPK_ERR_ARG, "iter_maxpy_uhead: %d values exceed one reduction", b.nv + b.m);
#define SPK_IB(TT, GG, UU, MPP) hipLaunchKernelGGL((iter_maxpy_uhead_kernel<TT, GG, UU, MPP>), dim3(grid), dim3(TT), 0, s, b)
    const int mp = b.m == 0 ? 0 : (b.m <= 4 ? 4 : 8);
    static const int deep = [] { const char *e = getenv("SPK_VEC_DEEP"); return e ? atoi(e) : 0; }();
    // thin forms: the whole basis in one group of loads where registers allow (see maxpy)
    const int g1 = !deep || b.nv <= 8 ? 8 : (b.nv <= 16 ? 16 : 32), g2 = !deep || b.nv <= 8 ? 8 : 16;
#define SPK_IB_MP(MPP)                                                           \
    do {                                                                         \
        if (!thin) SPK_IB(512, 4, 4, MPP);                                       \
        else if (tiny) SPK_IB(128, 8, 1, MPP);                                   \
        else if (U == 2) { if (g2 == 16) SPK_IB(256, 16, 2, MPP); else SPK_IB(256, 8, 2, MPP); } \
        else if (g1 == 32) SPK_IB(256, 32, 1, MPP);                              \
        else if (g1 == 16) SPK_IB(256, 16, 1, MPP);                              \
        else SPK_IB(256, 8, 1, MPP);                                             \
    } while (0)
    if (mp == 0) SPK_IB_MP(0);
    else if (mp == 4) SPK_IB_MP(4);
    else SPK_IB_MP(8);
#undef SPK_IB_MP
#undef SPK_IB
}

// ---------------------------------------------------------------------------
// "BA": VecMAXPY (+ VecNorm, + the next PCApply) and the NEXT MatMult in ONE launch (opts.iteration_form = 4; single
// rank, small vectors).  In the three-launch form kernel B ends, a boundary passes, and kernel A' starts streaming
// the matrix: two ramps, two tails and a gap around a dependency that is LOCAL -- the SpMV of a row tile needs z~
// only on the rows its columns touch (the adjacent grid lines).  Here a workgroup owns a fixed run of row tiles:
//   phase B   w' = s_w w~ - sum_i (h_i sc_i) V~_i over its rows, ||w'||^2 partial, z~ = D w' - (B D)^T y~ stored
//             write-through (sc1), c~ kept in LDS; then it raises ITS flag (sequence number, one line per workgroup)
//   phase A   the matrix stream of its tiles is requested, then it waits for the flags of the workgroups that own the
//             rows its columns touch (a handful), gathers z~ with sc1 loads and forms w~_next = A z~ + c~
// while the last workgroup reduces the norm (beside phase A of the others) and runs the Givens step of THIS iteration.
// No vector is normalised: the basis stays V~_i = w'_i with one scale factor sc_i = 1/||w'_i|| per vector, applied to the
// reduced scalars (h_i = sc_i s_w (V~_i . w~), ...) -- mathematically the same Arnoldi relation, no VecScale pass at all.
// Every wait is bounded; a flag that never rises raises the context's execution-error word.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void tile_col_range_kernel(const int32_t *__restrict__ browptr, const int32_t *__restrict__ bcol,
                                                               const int32_t *__restrict__ tile_brow, int ntiles, int32_t *__restrict__ out)
{
    const int t = blockIdx.x;
    if (t >= ntiles) return;
    int lo = INT32_MAX, hi = -1;
    for (int q = browptr[tile_brow[t]] + (int)threadIdx.x; q < browptr[tile_brow[t + 1]]; q += kWave) {
        const int c = bcol[q];
        lo = c < lo ? c : lo;
        hi = c > hi ? c : hi;
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        const int l2 = __shfl_down(lo, off, kWave), h2 = __shfl_down(hi, off, kWave);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if (threadIdx.x == 0) {
        out[2 * t] = lo;
        out[2 * t + 1] = hi;
    }
}
void tile_col_range(const int32_t *browptr, const int32_t *bcol, const int32_t *tile_brow, int ntiles, int32_t *out, hipStream_t s)
{
    if (ntiles == 0) return;
    hipLaunchKernelGGL(tile_col_range_kernel, dim3(ntiles), dim3(kWave), 0, s, browptr, bcol, tile_brow, ntiles, out);
}

__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int G, int MP>
__global__ __launch_bounds__(kThreads) void iter_ba_kernel(IterBA p)
{
    constexpr int T = kThreads;
    constexpr int NP = MP > 0 ? MP : 1;
    constexpr int kSteps = kBTile / T, kTB = 4;
    // everything a workgroup needs to START is requested before anything is looked at (a launch of this size is a
    // chain of a few memory round trips of ~1.3 us: measured 5.5 us from entry to the end of the scalar prologue when
    // "done", the workgroup's tile list and its rows were read one after the other)
    const int32_t dn = __builtin_nontemporal_load(p.done);
    __shared__ double prod[kBTile * 4];
    __shared__ double hs[kMaxNv], lam[kMaxNv * 8], ys[8], wraws[8], tus[8], red[T];
    __shared__ int okw;
    const int nv = p.nv, m = p.m;
    const int nwg = 8 * p.slots;
    const bool scalar_wg = (int)blockIdx.x == nwg;
    // row-order index: phase B owns the double2 entries [rho chunk, (rho + 1) chunk) -- no table look-up in front
    // of its loads; phase A owns the tiles wg[rho] = {t0, t1, first / last owner to wait for}
    const int xcd = blockIdx.x & 7, kslot = blockIdx.x >> 3;
    const int rho = scalar_wg ? 0 : xcd * p.slots + kslot;
    const int64_t n2 = p.nl / 2;
    const int64_t i2 = (int64_t)rho * p.chunk + threadIdx.x;
    const bool active = !scalar_wg && (int)threadIdx.x < p.chunk && i2 < n2;
    const bool dbg = p.dbg && (int)blockIdx.x == p.dbg_wg && threadIdx.x == 0;
    if (dbg) p.dbg[0] = wall_clock64();

    // ---- phase B loads first (they depend on nothing computed here)
    double2 wv, dv, pe[NP], t0v[G];
    wv.x = wv.y = dv.x = dv.y = 0.0;
    if (active) {
        wv = ld2(p.w, i2);
        dv = ld2(p.dinv, i2);
    }
    if (MP > 0) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const bool live = active && (p.packed ? 2 * q < m : q < m);
            pe[q].x = pe[q].y = 0.0;
            if (live) pe[q] = ld2s<true>(p.bd + (size_t)q * p.ldb, i2);
        }
    }
#pragma unroll
    for (int v = 0; v < G; ++v) {
        const bool live = active && v < nv;
        t0v[v].x = t0v[v].y = 0.0;
        if (live) t0v[v] = ld2s<true>(p.V + (size_t)v * p.ldv, i2);
    }
    // phase A's descriptors ride along (uniform loads)
    const int4 wgd = scalar_wg ? make_int4(0, 0, 0, -1) : reinterpret_cast<const int4 *>(p.wt)[rho];

    // ---- scalars, by every workgroup: h_i = sc_i s_w (V~_i . w~), a_i = -h_i sc_i, B D w' by linearity
    double lamv = 0.0;
    const bool lam_mine = (int)threadIdx.x < nv * m;
    if (lam_mine) lamv = p.V[(size_t)(threadIdx.x / m) * p.ldv + p.nl + (threadIdx.x % m)];
    double wl = 0.0, sh = 1.0;
    if ((int)threadIdx.x < m) {
        wl = p.wl_in[threadIdx.x];
        sh = p.shat[threadIdx.x];
    }
    const double s_w = p.sc[nv - 1];
    double sci = 0.0, draw = 0.0, qraw = 0.0, tbv[8];
    if (threadIdx.x < kWave) {  // lane i owns basis vector i (nv <= 63)
        const int i = threadIdx.x;
        sci = i < nv ? p.sc[i] : 0.0;
        draw = i < nv ? p.dots[i] : 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) tbv[r] = (r < m && i < nv) ? p.tb_[i * 8 + r] : 0.0;
        qraw = (i < m) ? p.dots[nv + i] : 0.0;
    }
    if (dn) return;  // (uniform over the launch: the gate word cannot change while its workgroups start)
    if (threadIdx.x < kWave) {
        const int i = threadIdx.x;
        const double hi = sci * s_w * draw;
        const double qv = s_w * qraw;
        if (i < nv) hs[i] = hi * sci;  // MAXPY coefficient of V~_i (sign applied at use)
        if (scalar_wg && i < nv) p.hbuf[i] = hi;  // the Hessenberg column of this iteration
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r < m) {  // uniform
                const double tsum = wave_sum(hi * sci * tbv[r]);
                const double qr = __shfl(qv, r, kWave);
                if (i == 0) tus[r] = qr - tsum;  // B D w' = B D w - sum h_i (B D v_i)
            }
        }
    }
    if (lam_mine) lam[threadIdx.x] = lamv;
    for (int t = threadIdx.x + T; t < nv * m; t += T) lam[t] = p.V[(size_t)(t / m) * p.ldv + p.nl + (t % m)];
    __syncthreads();
    if ((int)threadIdx.x < NP && MP > 0) {
        const int r = threadIdx.x;
        double y = 0.0, wraw = 0.0;
        if (r < m) {
            wraw = s_w * wl;
            for (int i = 0; i < nv; ++i) wraw += -hs[i] * lam[i * m + r];  // the MAXPY of the multiplier entries
            y = -(wraw - tus[r]) / sh;
        }
        wraws[r] = wraw;
        ys[r] = y;
    }
    __syncthreads();
    double yv[NP];
#pragma unroll
    for (int r = 0; r < NP; ++r) yv[r] = MP > 0 ? ys[r] : 0.0;
    if (dbg) p.dbg[1] = wall_clock64();  // prologue done

    if (scalar_wg) {
        // ---- the scalar / reducing workgroup
        if ((int)threadIdx.x < m) {
            const int r = threadIdx.x;
            double w1 = tus[r];
            if (p.fact == SPK_SCHUR_FULL)
                for (int q = 0; q < m; ++q) w1 -= p.gram[r * m + q] * ys[q];
            p.w[p.nl + r] = wraws[r];
            p.tb_[(size_t)nv * 8 + r] = tus[r];
            if (!p.last) {
                p.zout[p.nl + r] = ys[r];
                p.wnext[p.nl + r] = w1;
                p.wl_out[r] = w1;
            }
        }
        __syncthreads();
        final_reduce(p.partials, nwg, kPartialLd, 1, red, FinErr{p.err, p.fin_ticks});
        if (threadIdx.x == 0) {
            double tot = red[0];
            for (int r = 0; r < m; ++r) tot += p.lam_in_dot ? wraws[r] * wraws[r] : 0.0;
            red[0] = tot;
        }
        __syncthreads();
        if (p.ar.P) peer_allreduce_block(p.ar, red, 1, p.nrm_out);
        else if (threadIdx.x == 0) p.nrm_out[0] = red[0];
        __syncthreads();
        if (threadIdx.x == 0) p.sc[nv] = inv_norm(p.ar.P ? p.nrm_out[0] : red[0]);
        __syncthreads();
        // Givens step of THIS iteration: every workgroup of the launch passed its look at `done` long ago, and
        // none of them feeds another reduction of this launch
        givens_block(p.ka, p.loc, p.hbuf, p.nrm_out);
        return;
    }

    // phase A's matrix descriptors: requested now, consumed after the flag
    int4 td[kTB];
#pragma unroll
    for (int j = 0; j < kTB; ++j)
        td[j] = (wgd.x + j < wgd.y) ? reinterpret_cast<const int4 *>(p.tdesc)[wgd.x + j] : make_int4(0, 0, 0, 0);

    // ---- phase B arithmetic
    double2 sv;
    sv.x = sv.y = 0.0;
    if (MP > 0) {
        if (p.packed) {
#pragma unroll
            for (int q = 0; q < NP / 2; ++q) {
                sv.x += pe[q].x * yv[2 * q];
                sv.y += pe[q].y * yv[2 * q + 1];
            }
        } else {
#pragma unroll
            for (int r = 0; r < NP; ++r) {
                sv.x += pe[r].x * yv[r];
                sv.y += pe[r].y * yv[r];
            }
        }
    }
    wv.x *= s_w;
    wv.y *= s_w;
#pragma unroll
    for (int v = 0; v < G; ++v) {
        const double ai = v < nv ? -hs[v] : 0.0;
        wv.x += ai * t0v[v].x;
        wv.y += ai * t0v[v].y;
    }
    for (int g0 = G; g0 < nv; g0 += G) {
        double2 tt[G];
        double ai[G];
#pragma unroll
        for (int v = 0; v < G; ++v) {
            const bool live = active && g0 + v < nv;
            ai[v] = g0 + v < nv ? -hs[g0 + v] : 0.0;
            tt[v].x = tt[v].y = 0.0;
            if (live) tt[v] = ld2s<true>(p.V + (size_t)(g0 + v) * p.ldv, i2);
        }
#pragma unroll
        for (int v = 0; v < G; ++v) {
            wv.x += ai[v] * tt[v].x;
            wv.y += ai[v] * tt[v].y;
        }
    }
    if (dbg) p.dbg[2] = wall_clock64() + (unsigned long long)(wv.x == 1.2345e300);  // MAXPY arithmetic done (loads arrived)
    double nrm = 0.0;
    if (active) {
        double2 zz;
        nrm = wv.x * wv.x + wv.y * wv.y;
        zz.x = wv.x * dv.x;
        zz.y = wv.y * dv.y;
        if (MP > 0 && p.fact == SPK_SCHUR_FULL) {
            zz.x -= sv.x;
            zz.y -= sv.y;
        }
        reinterpret_cast<double2 *>(p.w)[i2] = wv;
        if (!p.last) {
            // write-through: read by OTHER workgroups of this launch (z~ gathered, c~ by the owner of the row's tile)
            st_agent(p.zout + 2 * i2, zz.x);
            st_agent(p.zout + 2 * i2 + 1, zz.y);
            if (MP > 0) {
                st_agent(p.wnext + 2 * i2, sv.x / dv.x);
                st_agent(p.wnext + 2 * i2 + 1, sv.y / dv.y);
            }
        }
    }
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const double sw = wave_sum(nrm);
        if (lane == 0) red[wave] = sw;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have left before the flag may rise
    __syncthreads();
    if (threadIdx.x == 0) {
        double tsum = 0.0;
#pragma unroll
        for (int j = 0; j < T / kWave; ++j) tsum += red[j];
        publish(p.partials + (size_t)blockIdx.x * kPartialLd, tsum);
        if (!p.last) __hip_atomic_store(p.flags + (size_t)rho * 32, p.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (dbg) p.dbg[3] = wall_clock64();  // flag raised
    const int t0 = wgd.x, t1 = wgd.y;
    if (p.last || t0 >= t1) return;

    // ---- phase A: the next product of this workgroup's tiles.  The matrix stream of ALL its tiles is requested at
    // once (descriptors are here already), then the wait, then ONE agent-scope acquire per workgroup (this CU's L1
    // forgets what it may hold; the producers stored write-through, and no line of z~ can sit in this XCD's L2 yet: it
    // was never read in this launch), then ONE gather round trip for all tiles on plain cached loads (every z~ entry is
    // used 18 times; sc1 gathers, tried first, sent each use over the fabric as an 8-byte request).
    int c[kTB][kSteps];
    double2 tp[kTB][kSteps], bo[kTB][kSteps], xv[kTB][kSteps];
#pragma unroll
    for (int j = 0; j < kTB; ++j) {
        if (t0 + j < t1) {  // uniform
            const int b0 = td[j].z, cntb = td[j].w - td[j].z;
#pragma unroll
            for (int i = 0; i < kSteps; ++i) {
                const int q = i * T + threadIdx.x;
                c[j][i] = 0;
                if (q < cntb) {
                    c[j][i] = __builtin_nontemporal_load(p.bcol + b0 + q);
                    tp[j][i] = ld2s<true>(p.vtop, b0 + q);
                    bo[j][i] = ld2s<true>(p.vbot, b0 + q);
                }
            }
        }
    }
    {   // wait for the owners of the rows my columns (and my own rows' c~) live in (bounded)
        const int wlo = wgd.z, whi = wgd.w;
        if (threadIdx.x == 0) okw = 1;
        __syncthreads();
        for (int w = wlo + (int)threadIdx.x; w <= whi; w += T) {
            const uint32_t *f = p.flags + (size_t)w * 32;
            if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.seq) {
                const unsigned long long tw0 = wall_clock64();
                for (;;) {
                    __builtin_amdgcn_s_sleep(1);
                    if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p.seq) break;
                    if (wall_clock64() - tw0 > (unsigned long long)p.fin_ticks) {
                        okw = 0;
                        break;
                    }
                }
            }
        }
        __syncthreads();
        if (!okw) {  // an owner never finished its phase B: execution failure (reported by the host), no product
            if (threadIdx.x == 0 && p.err) __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (dbg) p.dbg[4] = wall_clock64();  // neighbours' flags seen
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (dbg) p.dbg[5] = wall_clock64();  // acquire done
    }
#pragma unroll
    for (int j = 0; j < kTB; ++j) {
        if (t0 + j < t1) {
            const int cntb = td[j].w - td[j].z;
#pragma unroll
            for (int i = 0; i < kSteps; ++i) {
                const int q = i * T + threadIdx.x;
                if (q < cntb) xv[j][i] = reinterpret_cast<const double2 *>(p.zout)[c[j][i]];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < kTB; ++j) {
        if (t0 + j < t1) {  // uniform
            const int tb0 = td[j].x, tb1 = td[j].y;
            const int b0 = td[j].z, cntb = td[j].w - td[j].z;
            const int nr = 2 * (tb1 - tb0), r0 = 2 * tb0;
            const int lr = threadIdx.x;
            // row thread's own operands (row extent, c~) requested before the products
            int k0 = 0, k1 = 0;
            double cpre = 0.0;
            if (lr < nr) {
                const int br = tb0 + (lr >> 1);
                k0 = p.browptr[br] - b0;
                k1 = p.browptr[br + 1] - b0;
                if (MP > 0) cpre = p.wnext[r0 + lr];
            }
#pragma unroll
            for (int i = 0; i < kSteps; ++i) {
                const int q = i * T + threadIdx.x;
                if (q < cntb) {
                    double2 p0, p1;
                    p0.x = tp[j][i].x * xv[j][i].x;
                    p0.y = tp[j][i].y * xv[j][i].y;
                    p1.x = bo[j][i].x * xv[j][i].x;
                    p1.y = bo[j][i].y * xv[j][i].y;
                    *reinterpret_cast<double2 *>(prod + 4 * q) = p0;
                    *reinterpret_cast<double2 *>(prod + 4 * q + 2) = p1;
                }
            }
            __syncthreads();
            if (lr < nr) {
                const int half = lr & 1;
                double sr = 0.0;
                for (int k = k0; k < k1; ++k) {
                    const double2 pp = *reinterpret_cast<const double2 *>(prod + 4 * k + 2 * half);
                    sr += pp.x;
                    sr += pp.y;
                }
                sr += cpre;
                p.wnext[r0 + lr] = sr;
            }
            __syncthreads();  // prod is rewritten by the next tile
            if (dbg) p.dbg[6 + j] = wall_clock64();  // tile done
        }
    }
}

void iter_ba(const IterBA &p, hipStream_t s)
{
    const dim3 grid(8 * p.slots + 1), block(kThreads);
    if (p.nv + p.m > kMaxNv - 1) fail(SPK_ERR_ARG, "iter_ba: %d values exceed one reduction", p.nv + p.m);
    if (8 * p.slots > kMaxBlocks) fail(SPK_ERR_ARG, "iter_ba: %d workgroups exceed the partials buffer", 8 * p.slots);
    if (p.m == 0) hipLaunchKernelGGL((iter_ba_kernel<8, 0>), grid, block, 0, s, p);
    else if (p.m <= 4) hipLaunchKernelGGL((iter_ba_kernel<8, 4>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((iter_ba_kernel<8, 8>), grid, block, 0, s, p);
}

// -ksp_gmres_cgs_refinement_type: mode 2 (always) refines unless done; mode 1 (ifneeded)
// refines when ||w'|| < ||h|| (PETSc's test); the second-pass kernels take skip_refine as
// their "done" word.  dots2 is zeroed so that a skipped pass merges as a no-op.
__global__ void krylov_refine_decide_kernel(KrylovArrays ka, int loc, int mode, const double *dots,
                                            const double *nrm2, double *dots2)
{
    KrylovState *st = ka.st;
    if ((int)threadIdx.x <= loc) dots2[threadIdx.x] = 0.0;
    if (threadIdx.x != 0) return;
    int skip = st->done ? 1 : 0;
    if (!skip && mode == SPK_REFINE_IFNEEDED) {
        double hn = 0.0;
        for (int j = 0; j <= loc; ++j) hn += dots[j] * dots[j];
        skip = !(sqrt(*nrm2) < sqrt(hn));
    }
    st->skip_refine = skip;
}
void krylov_refine_decide(const KrylovArrays &ka, int loc, int mode, const double *dots, const double *nrm2,
                          double *dots2, hipStream_t s)
{
    hipLaunchKernelGGL(krylov_refine_decide_kernel, dim3(1), dim3(64), 0, s, ka, loc, mode, dots, nrm2, dots2);
}
__global__ void krylov_refine_merge_kernel(KrylovArrays ka, int loc, double *dots, const double *dots2,
                                           double *nrm, const double *nrm_b, int nn)
{
    if (ka.st->skip_refine) return;
    if ((int)threadIdx.x <= loc) dots[threadIdx.x] += dots2[threadIdx.x];
    if ((int)threadIdx.x < nn) nrm[threadIdx.x] = nrm_b[threadIdx.x];
}
void krylov_refine_merge(const KrylovArrays &ka, int loc, double *dots, const double *dots2, double *nrm,
                         const double *nrm_b, int nn, hipStream_t s)
{
    hipLaunchKernelGGL(krylov_refine_merge_kernel, dim3(1), dim3(64), 0, s, ka, loc, dots, dots2, nrm, nrm_b, nn);
}

// back substitution for the loc_done columns built in this cycle; the triangle is staged
// in LDS by the whole workgroup first (450 dependent global loads took 47 us)
__global__ __launch_bounds__(256) void krylov_cycle_end_kernel(KrylovArrays ka, const double *sc)
{
    __shared__ double Hs[(kMaxNv) * (kMaxNv + 1)];
    __shared__ double rss[kMaxNv + 2], ys[kMaxNv + 2];
    KrylovState *st = ka.st;
    const int n = st->loc_done, ldh = ka.ldh;
    for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
        const int k = e % n, j = e / n;  // row k, column j
        Hs[j * n + k] = ka.H[(size_t)ldh * j + k];
    }
    for (int k = threadIdx.x; k < n; k += blockDim.x) rss[k] = ka.rs[k];
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int k = n - 1; k >= 0; --k) {
        double t = rss[k];
        for (int j = k + 1; j < n; ++j) t -= Hs[j * n + k] * ys[j];
        const double piv = Hs[k * n + k];
        if (piv == 0.0) {
            if (st->reason >= 0) st->reason = SPK_DIVERGED_BREAKDOWN;
            st->done = 1;
            st->loc_done = 0;
            return;
        }
        ys[k] = t / piv;
    }
    // (un-normalised Z~_k of the BA iteration: x += sum y_k sc_k Z~_k)
    for (int k = 0; k < n; ++k) ka.nrs[k] = sc ? ys[k] * sc[k] : ys[k];
}
// The same back substitution for restart lengths whose triangle does not fit LDS (-ksp_gmres_restart > 62): one thread,
// same order of operations, H read from global memory eight entries at a time (the loads do not depend on the chain)
__global__ __launch_bounds__(64) void krylov_cycle_end_big_kernel(KrylovArrays ka)
{
    __shared__ double ys[kBigNv + 2];
    if (threadIdx.x != 0) return;
    KrylovState *st = ka.st;
    const int n = st->loc_done, ldh = ka.ldh;
    for (int k = n - 1; k >= 0; --k) {
        double t = ka.rs[k];
        int j = k + 1;
        for (; j + 8 <= n; j += 8) {
            double h[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) h[u] = ka.H[(size_t)ldh * (j + u) + k];
#pragma unroll
            for (int u = 0; u < 8; ++u) t -= h[u] * ys[j + u];
        }
        for (; j < n; ++j) t -= ka.H[(size_t)ldh * j + k] * ys[j];
        const double piv = ka.H[(size_t)ldh * k + k];
        if (piv == 0.0) {
            if (st->reason >= 0) st->reason = SPK_DIVERGED_BREAKDOWN;
            st->done = 1;
            st->loc_done = 0;
            return;
        }
        ys[k] = t / piv;
    }
    for (int k = 0; k < n; ++k) ka.nrs[k] = ys[k];
}
void krylov_cycle_end(const KrylovArrays &ka, hipStream_t s, const double *sc, int restart)
{
    if (restart > kMaxNv - 2) hipLaunchKernelGGL(krylov_cycle_end_big_kernel, dim3(1), dim3(64), 0, s, ka);
    else hipLaunchKernelGGL(krylov_cycle_end_kernel, dim3(1), dim3(256), 0, s, ka, sc);
}

}  // namespace k
}  // namespace spk
