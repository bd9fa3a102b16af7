// spk_device.hpp -- device-side helpers shared by the kernel files (spk_k_*.hip): workgroup reductions, the
// sentinel-based cross-workgroup finish, the peer-store granule protocol, the Givens step and its rider, launch shapes.
// Device code only: include from .hip files.
#pragma once
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
__device__ __forceinline__ void st2nt(double *p, int64_t i2, double2 v)   // non-temporal 16-byte store
{
    dbl2v t;
    t.x = v.x;
    t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<dbl2v *>(p) + i2);
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
// (FinErr -- where a reducer reports a partial that never arrived -- is declared in spk_internal.hpp)
// FB: partials a reducer thread requests together (kFinBatch; 2 where the caller's kernel must stay small in registers)
template <int FB = kFinBatch>
__device__ __forceinline__ void final_reduce(double *partials, int nb, int ld, int k, double *scratch, FinErr fe)
{
    const int T = blockDim.x;
    int kk = 1;
    while (kk < k) kk <<= 1;
    const int i = threadIdx.x & (kk - 1), sl = threadIdx.x / kk, nsl = T / kk;
    const double armed = __longlong_as_double((long long)kSentinelBits);
    double acc = 0.0;
    if (i < k) {
        for (int b0 = sl; b0 < nb; b0 += nsl * FB) {
            double v[FB];
            // the whole batch is requested at once (one round trip) and simply requested again while
            // any of its slots is still armed, i.e. its workgroup has not published yet
            const unsigned long long t0 = wall_clock64();
            bool armed_seen;
            do {
                armed_seen = false;
#pragma unroll
                for (int u = 0; u < FB; ++u) {
                    const int b = b0 + u * nsl;
                    v[u] = b < nb ? peek(partials + (size_t)b * ld + i) : 0.0;
                }
#pragma unroll
                for (int u = 0; u < FB; ++u) armed_seen = armed_seen || is_sentinel(v[u]);
                if (armed_seen) __builtin_amdgcn_s_sleep(1);
            } while (armed_seen && wall_clock64() - t0 < (unsigned long long)fe.ticks);  // default 4 s at 100 MHz
            // a slot still armed after the bound: its workgroup never published (never dispatched, or the
            // launch was rejected half way).  Not a numerical event: raise the context's sticky error word --
            // the host turns it into SPK_ERR_HIP and re-arms the whole buffer before the next use
            if (armed_seen && fe.err) __hip_atomic_store(fe.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < FB; ++u) {
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

__device__ __forceinline__ double2 ld2(const double *p, int64_t i2)
{
    return reinterpret_cast<const double2 *>(p)[i2];
}
__device__ __forceinline__ double ld1nt(const double *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int converged_default(double rnorm, const KrylovState *st)
{
    if (isnan(rnorm) || isinf(rnorm)) return SPK_DIVERGED_NANORINF;
    if (rnorm <= st->ttol) return (rnorm < st->abstol) ? SPK_CONVERGED_ATOL : SPK_CONVERGED_RTOL;
    if (rnorm >= st->dtol * st->cnorm0) return SPK_DIVERGED_DTOL;
    return 0;
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
// lds: 4 * cap + 4 doubles of LDS scratch (the caller's own staging where it has any: a rider workgroup of a product
// launch uses the tile's product buffer, so the launch needs no LDS beyond what its row tiles need)
// LEAN: loops kept rolled -- the step then fits 32 VGPRs, which is what a rider workgroup of the SpMV kernels may use
// without raising the register allocation of every row-tile wave of the launch (measured: 64 instead of 32 allocated
// VGPRs cost the 1024^2 product 5 us of 62); the serial chain takes ~1 us longer, beside the row tiles
// What the step reads that does NOT depend on the norm (the Hessenberg column, the stored rotations, rs[loc], the gate
// words), requested by the lanes that will stage it: a caller with work of its own between the request and the use (the
// rider's reduction) overlaps the two round trips.  loc <= blockDim.x - 1 (the small instance: restart <= 62).
struct GivensPre {
    double h, c, s, rs;
    int go;   // 0: the solve / cycle is over (uniform)
};
__device__ __forceinline__ GivensPre givens_prefetch(const KrylovArrays &ka, int loc, const double *dots)
{
    GivensPre p{0.0, 0.0, 0.0, 0.0, 0};
    const KrylovState *st = ka.st;
    p.go = !(st->done || st->skip_iter);
    if ((int)threadIdx.x <= loc) {
        p.h = dots[threadIdx.x];
        p.c = ka.cc[threadIdx.x];
        p.s = ka.ss[threadIdx.x];
    }
    if (threadIdx.x == 33) p.rs = ka.rs[loc];
    return p;
}
template <bool LEAN = false>
__device__ inline void givens_block_lds(const KrylovArrays &ka, int loc, const double *dots, const double *nrm2, int *gate,
                                        double *lds, int cap, const GivensPre *pre = nullptr)
{
    double *Hc = lds, *Hr = lds + cap, *ccs = lds + 2 * cap, *sss = lds + 3 * cap, *sc = lds + 4 * cap;
    KrylovState *st = ka.st;
    if (pre ? !pre->go : (st->done || st->skip_iter)) return;  // uniform: read before anyone writes it
    const int ldh = ka.ldh;
    double *Hg = ka.H + (size_t)ldh * loc;  // column loc
    // every global value the serial chain needs is fetched here, in parallel, once
    if (pre) {
        if ((int)threadIdx.x <= loc) {
            Hc[threadIdx.x] = pre->h;
            ccs[threadIdx.x] = pre->c;
            sss[threadIdx.x] = pre->s;
        }
        if (threadIdx.x == 33) sc[1] = pre->rs;
    } else {
        for (int j = threadIdx.x; j <= loc; j += blockDim.x) {
            Hc[j] = dots[j];
            ccs[j] = ka.cc[j];
            sss[j] = ka.ss[j];
        }
        if (threadIdx.x == 33) sc[1] = ka.rs[loc];
    }
    if (threadIdx.x == 32) sc[0] = *nrm2;
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
    if constexpr (LEAN) {
#pragma unroll 2
        for (int j = 1; j <= loc; ++j) {
            const double h1 = Hc[j], cj = ccs[j - 1], sj = sss[j - 1];
            Hr[j - 1] = cj * run + sj * h1;
            run = cj * h1 - sj * run;
        }
    } else {
        for (int j = 1; j <= loc; ++j) {
            const double h1 = Hc[j], cj = ccs[j - 1], sj = sss[j - 1];
            Hr[j - 1] = cj * run + sj * h1;
            run = cj * h1 - sj * run;
        }
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
    if constexpr (LEAN) {
#pragma unroll 1
        for (int j = 0; j <= loc + 1; ++j) Hg[j] = Hr[j];
    } else {
        for (int j = 0; j <= loc + 1; ++j) Hg[j] = Hr[j];
    }
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

template <int CAP>
__device__ void givens_block_t(const KrylovArrays &ka, int loc, const double *dots, const double *nrm2, int *gate)
{
    __shared__ double lds[4 * CAP + 4];
    givens_block_lds<false>(ka, loc, dots, nrm2, gate, lds, CAP);
}
__device__ inline void givens_block(const KrylovArrays &ka, int loc, const double *dots, const double *nrm2, int *gate = nullptr)
{
    givens_block_t<kMaxNv + 2>(ka, loc, dots, nrm2, gate);
}

__device__ __forceinline__ double inv_norm(double nrm2)  // the VecScale guard of the head kernels
{
    const double tt = sqrt(nrm2);
    return tt > 1e-300 ? 1.0 / tt : 1.0;
}
// lds: >= kThreads + 4 * (kMaxNv + 2) + 4 doubles of the calling workgroup's LDS
__device__ __forceinline__ void givens_rider(const GivensRider &gr, double *lds)
{
    // The MAXPY launch left ||w'||^2 as one partial per workgroup (IterB::defer_fin): reduced here, in the fixed order,
    // and all-reduced across ranks (peer-store) -- beside the row tiles, since nothing in a product on an un-normalised
    // basis needs the norm: neither the reduction tail nor the link latency is on the critical path
    // (what the Givens step reads besides the norm is requested first: one round trip beside the reduction's)
    const GivensPre pre = givens_prefetch(gr.ka, gr.loc, gr.h);
    if (gr.fin_n > 0) {
        double *red = lds;
        double *slot = gr.fin_partials + (size_t)gr.fin_n * kPartialLd;  // the multiplier entries' share
        double lam2 = 0.0;
        if (threadIdx.x == 0) lam2 = peek(slot);
        final_reduce<2>(gr.fin_partials, gr.fin_n, kPartialLd, 1, red, gr.fe);
        if (threadIdx.x == 0) {
            publish(slot, __longlong_as_double((long long)kSentinelBits));   // re-armed for the next user of the row
            red[0] = red[0] + lam2;
        }
        __syncthreads();
        if (gr.ar.P) peer_allreduce_block(gr.ar, red, 1, gr.nrm2);
        else if (threadIdx.x == 0) gr.nrm2[0] = red[0];
        __syncthreads();
    }
    // un-normalised basis: the scale factor of the vector the MAXPY launch just wrote (its norm is all-reduced by now)
    // (nothing compounds: V~_j = w' of the product of the NORMALISED v_{j-1}, so ||V~_j|| = h_{j,j-1} <= ||K M^-1||)
    if (gr.sc && threadIdx.x == 0) gr.sc[gr.loc + 1] = inv_norm(*gr.nrm2);
    givens_block_lds<true>(gr.ka, gr.loc, gr.h, gr.nrm2, nullptr, lds + kThreads, kMaxNv + 2, &pre);
}

void givens_rider_alone(const GivensRider &gr, const int32_t *done, hipStream_t s);  // spk_k_krylov.hip
inline GivensRider no_rider()
{
    GivensRider g{};
    g.loc = -1;
    return g;
}

// tile length (double2 per lane) and grid of the wave-split forms: about one tile per workgroup
struct WsShape {
    int U, grid;
    bool on;
};
inline WsShape ws_shape(int64_t n2)
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
inline VecShape vec_shape(int64_t n2, bool maxpy = false)
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
inline int vec_grid(int64_t n2, int T = kVT)
{
    // one double2 per thread and pass: small vectors still get a workgroup per CU (a 1/8 slab of the 1024^2 grid ran its
    // restart norms on 64 workgroups)
    int64_t tiles = (n2 + (int64_t)T - 1) / (int64_t)T;
    if (tiles < 1) tiles = 1;
    const int cap = T >= 512 ? kVecMaxBlocks : 2 * kVecMaxBlocks;
    return (int)(tiles < cap ? tiles : cap);
}

}  // namespace k
}  // namespace spk
