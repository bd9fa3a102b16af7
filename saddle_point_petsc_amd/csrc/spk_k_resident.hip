// spk_k_resident.hip -- one launch per RESTART CYCLE with the Krylov basis resident on chip (opts.iteration_form =
// SPK_ITER_RESIDENT; small local systems: a rank's slab of a strong-scaling run, the 256^2 grid).  gfx950, wave64.
//
// What PETSc's KSPFGMRESCycle does per iteration -- PCApply, MatMult, VecMDot, VecMAXPY, VecNorm, the Hessenberg update
// (KSPSolve at /root/reference/src/SaddlePointProblem.c:70) -- is three launches in the default form (MDot, MAXPY + next
// PCApply, MatMult).  On a slab of 262 k rows those launches move 141 MB per iteration out of the Infinity Cache and pay
// ~5 us each beyond their bytes: 35-39 us per iteration against 200 us for the eight times larger grid, i.e. at most 65 %
// strong-scaling efficiency at N = 8 before any link latency.  Here the whole cycle is ONE launch of one workgroup per
// CU, and nothing the orthogonalisation touches leaves the chip:
//   * every thread owns ONE block row (two vector entries) and keeps its entries of the un-normalised basis V~_0 .. V~_j
//     in REGISTERS (31 x double2 = 124 VGPRs; 512-thread workgroups, two waves per SIMD) -- VecMDot is a register dot
//     product and a reduction, VecMAXPY touches no memory at all;
//   * the matrix is the "row types + deviation codes" layout (spk_dict.hpp): ~100 B per block row out of the L2;
//   * per iteration the workgroups meet twice: (a) the inner products -- every workgroup publishes its partial sums
//     (values are their own arrival flags, the buffer is armed with a sentinel once per cycle); value v is added up, in
//     one fixed order, by one workgroup of every group of 32 (an XCD's) and its total published to that group; every
//     workgroup reads the <= 40 totals of its group, so all workgroups hold the same bits and run the same scalar work
//     (Hessenberg column, Givens rotation, KSPConvergedDefault) redundantly.  (First form of this round: every workgroup
//     read all 256 x 40 partials itself -- one hop, but 16 MB through the fabric per iteration: 5.8 us.)  (b) the product: a workgroup stores its rows of z~ = M^-1 w' write-through into Z_{j+1} (armed with the
//     sentinel), its neighbours gather them with polling loads.  ||w'||^2 rides in the NEXT iteration's all-to-all.
// Same algorithm as the default form (classical Gram-Schmidt, two reductions per iteration, un-normalised basis with one
// scale factor per vector: include/spk.h, SPK_ITER_UNNORM); products of the A block bit-identical to the CSR loop.
// Every wait is bounded (FinErr ticks): a workgroup that is not resident, or a lost partial, raises the context's sticky
// execution-error word (SPK_ERR_HIP), it cannot hang the device.
#include "spk_dict.hpp"

namespace spk {
namespace k {

constexpr int kResMaxV = 31;   // basis vectors a thread holds (restart <= 30)
constexpr int kResLd = 64;     // values per iteration in the all-to-all buffer (<= kResMaxVals used)
constexpr int kResG = 256;     // workgroups (one per CU); the buffer is [iteration][value][workgroup]: a reader's 8 partials are contiguous

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() is a release fence first: it also waits for every
// outstanding GLOBAL access of the wave (vmcnt(0)) -- here the write-through stores of partial sums and of z~, a trip to
// memory each (measured: ~1.5 us per barrier behind such a store).  What other workgroups read is handed over by polling,
// so inside the iteration loop only the LDS counter has to drain.
__device__ __forceinline__ void bar_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ bool res_timed_out(unsigned long long t0, uint32_t ticks) { return wall_clock64() - t0 > (unsigned long long)ticks; }

// One Arnoldi step's scalar work on the workgroup's OWN copy of the Krylov scalars (every workgroup runs it on the same
// inputs).  Semantics of givens_block_lds (spk_device.hpp).  Hr: the rotated column (loc + 2 entries), kept for the flush.
__device__ __forceinline__ void res_givens(KrylovState &L, int loc, const double *__restrict__ hcol, double nrm2,
                                           double *__restrict__ cc, double *__restrict__ ss, double *__restrict__ rs,
                                           double *__restrict__ Hr)
{
    const double rs_loc = rs[loc];
    const double tt = sqrt(nrm2);
    L.hapend = 0;
    if (isnan(tt) || isinf(tt)) {
        L.rnorm = tt;
        L.reason = SPK_DIVERGED_NANORINF;
        L.done = L.skip_iter = 1;
        L.hapend = -1;   // (flush: nothing but the state)
        return;
    }
    double hapbnd = fabs(tt / rs_loc);
    if (hapbnd > 1e-30) hapbnd = 1e-30;
    const int hapend = !(tt > hapbnd);
    L.tt = tt;
    L.inv_tt = hapend ? 1.0 : 1.0 / tt;
    double run = hcol[0];
#pragma unroll 4
    for (int j = 1; j <= loc; ++j) {
        const double h1 = hcol[j], cj = cc[j - 1], sj = ss[j - 1];
        Hr[j - 1] = cj * run + sj * h1;
        run = cj * h1 - sj * run;
    }
    Hr[loc] = run;
    Hr[loc + 1] = tt;
    double rnorm = 0.0;
    if (!hapend) {
        const double h0 = run, h1 = tt;
        const double d = sqrt(h0 * h0 + h1 * h1);
        if (d == 0.0) {
            L.reason = SPK_DIVERGED_NULL;
            L.done = L.skip_iter = 1;
            L.hapend = -1;
            return;
        }
        const double c = h0 / d, sn = h1 / d;
        cc[loc] = c;
        ss[loc] = sn;
        rs[loc + 1] = -sn * rs_loc;
        rs[loc] = c * rs_loc;
        Hr[loc] = c * h0 + sn * h1;
        rnorm = fabs(sn * rs_loc);
    }
    L.its += 1;
    L.loc_done = loc + 1;
    L.rnorm = rnorm;
    L.hapend = hapend;
    int reason = converged_default(rnorm, &L);
    if (hapend && !reason) reason = SPK_DIVERGED_BREAKDOWN;
    if (!reason && L.its >= L.max_it) reason = SPK_DIVERGED_ITS;
    L.reason = reason;
    if (reason) L.done = L.skip_iter = 1;
}
// what krylov_cycle_end and the host read of that step (master workgroup, off the critical path)
__device__ __forceinline__ void res_givens_flush(const KrylovState &L, const KrylovArrays &ka, int loc, const double *Hr,
                                                 const double *cc, const double *ss, const double *rs)
{
    if (L.hapend >= 0) {
        double *Hg = ka.H + (size_t)ka.ldh * loc;
        for (int j = 0; j <= loc + 1; ++j) Hg[j] = Hr[j];
        if (!L.hapend) {
            ka.cc[loc] = cc[loc];
            ka.ss[loc] = ss[loc];
            ka.rs[loc + 1] = rs[loc + 1];
            ka.rs[loc] = rs[loc];
        }
        if (L.its < ka.hist_cap) ka.hist[L.its] = L.rnorm;
    }
    KrylovState o = L;
    if (o.hapend < 0) o.hapend = 0;
    *ka.st = o;
}

struct ResArgs {
    DictArgs d;
    int G, rpw;            // workgroups; block rows per workgroup (<= T)
    int mk, m, np, packed, fact, lam_in_dot;   // np: planes of B D actually present (<= NP of the instantiation)
    int64_t nl, ld;
    const double *V0, *V1; // v_0 (normalised) and w~ = K z_0 (first product of the cycle, made by the launches before)
    double *Z;             // Z_j = Z + j ld: Z_1 .. written here (rows armed with the sentinel)
    const double *dinv, *bd;
    int64_t ldb;
    const double *shat, *gram;
    double *P;             // partial sums (mk + 1) x kResLd x kResG, then the groups' totals (mk + 1) x 8 x kResLd; armed
    KrylovArrays ka;
    double *sc_out;        // scale factors of the un-normalised basis (krylov_cycle_end)
    int32_t *err;
    uint32_t ticks;
    // several ranks (peer-store backend): the ranks' sums of every exchange go through the all-reduce windows -- the master
    // workgroup stores this rank's, EVERY workgroup reads all ranks' out of the rank's own window and adds them in rank
    // order; the halo rows of z~ go as granules into the neighbours' staging, the rows with off-rank columns wait for theirs
    PeerAR ar;             // P <= 1: single rank; seq = the first all-reduce of this launch
    SendRanges sr0, sr1;   // the two staging parities; sr0.seq = the first exchange of this launch (n == 0: no neighbours)
    OffDiag od;            // off-rank columns by local row (colidx = ghost number)
    int tab_bytes;         // LDS bytes of the matrix tables (16-byte multiple)
#ifdef SPK_RES_STAMPS
    unsigned long long *stamps;   // developer build: 100 MHz time stamps of the phases of iterations 10 and 25, per workgroup
#endif
};
#ifdef SPK_RES_STAMPS
#define RES_STAMP(k_) do { if (t == 0 && (loc == 10 || loc == 25)) a.stamps[((size_t)wg * 2 + (loc == 25)) * 16 + (k_)] = wall_clock64(); } while (0)
#else
#define RES_STAMP(k_) do { } while (0)
#endif

constexpr int kResPass = 20;      // inner products reduced per pass through the LDS staging
static_assert(kResLd >= 40, "an exchange carries nv + m + 1 <= 30 + 8 + 1 values");

// NP: planes of B D a thread holds (0: K = A; packed: m = 2 NP, dense: m = NP)
template <int T, int NP>
__global__ __launch_bounds__(T) void cycle_resident_kernel(ResArgs a)
{
    constexpr int LDP = T + 8;                                   // row stride of the product staging (bank shift per row)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // ---- LDS: matrix tables | code planes of this workgroup's rows | product staging | scalar work space
    u64 *cwd = reinterpret_cast<u64 *>(smem + a.tab_bytes);        // kmax x T: the code word of every block of this workgroup's rows
    double *prod = reinterpret_cast<double *>(cwd + (size_t)a.d.kmax * T);   // kResPass x LDP; also the all-to-all partials
    double *ws = prod + kResPass * LDP;
    double *dots = ws;                      // 64
    double *dotsg = dots + 64;              // 64: the same summed over the ranks
    double *hs = dotsg + 64;                // 32: MAXPY coefficients h_i sc_i
    double *hcolb = hs + 32;                // 2 x 34: Hessenberg column (scaled), by iteration parity
    double *scl = hcolb + 68;               // 34: scale factors
    double *gcc = scl + 34, *gss = gcc + 34, *grs = gss + 34, *gHr = grs + 34;   // rotations, rhs, rotated column
    double *lamV = gHr + 34;                // 32 x 8: multiplier entries of the basis vectors
    double *tbl = lamV + 32 * 8;            // 32 x 8: B D V~_i
    double *ys = tbl + 32 * 8, *wraws = ys + 8, *tus = wraws + 8, *w1s = tus + 8, *wl = w1s + 8, *shs = wl + 8;   // 8 each
    double *grm = shs + 8;                  // 64
    KrylovState *L = reinterpret_cast<KrylovState *>(grm + 64);
    int *flag = reinterpret_cast<int *>(L + 1);   // [0] stop (done / timed out), [1] Givens step waiting for its flush

    const int wg = blockIdx.x, t = threadIdx.x;
    const int m = a.m, mk = a.mk;
    const bool master = wg == 0;
    const int64_t br = (int64_t)wg * a.rpw + t;
    const bool active = t < a.rpw && br < a.d.nbrows;
    constexpr int TG = T - kWave;   // the thread that runs the Givens steps (lane 0 of the last wave: beside wave 0's scalar work)

    // ---- prologue: state, tables, this thread's entries, this workgroup's matrix codes
    for (int i = t; i < (int)(reinterpret_cast<double *>(L) - ws); i += T) ws[i] = 0.0;
    if (t == 0) {
        *L = *a.ka.st;
        flag[0] = 0;
        flag[1] = -1;
    }
    double2 V[kResMaxV];
#pragma unroll
    for (int i = 0; i < kResMaxV; ++i) V[i].x = V[i].y = 0.0;
    double2 w, dv, pe[NP > 0 ? NP : 1];
    w.x = w.y = 0.0;
    dv.x = dv.y = 1.0;
    int tid = 0;
    if (active) {
        V[0] = ld2(a.V0, br);
        w = ld2(a.V1, br);
        dv = ld2(a.dinv, br);
        tid = (int)a.d.tid[br];
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        pe[q].x = pe[q].y = 0.0;
        if (active && q < a.np) pe[q] = ld2(a.bd + (size_t)q * a.ldb, br);
    }
    // the codes of this thread's block row: read once per cycle, decoded out of LDS in every product
    for (int k = 0; k < a.d.kmax; ++k)
        cwd[k * T + t] = active ? *dict_word_ptr<2>(a.d, const_cast<unsigned char *>(a.d.codes), k, br) : 0ull;
    dict_load_lds(a.d, (a.d.nclass + 1) * 4, smem);   // (ends with a barrier)
    if (L->done || L->skip_iter) return;         // uniform: set by krylov_cycle_begin before this launch
    if (t < m) {
        lamV[t] = a.V0[a.nl + t];
        wl[t] = a.V1[a.nl + t];
        tbl[t] = a.ka.tb[t];
        shs[t] = a.shat[t];
    }
    if (t < m * m) grm[t] = a.gram[t];
    if (t == 0) {
        scl[0] = 1.0;
        grs[0] = a.ka.rs[0];
    }
    const int32_t *tlen = reinterpret_cast<const int32_t *>(smem);
    const int2 *tent = reinterpret_cast<const int2 *>(smem + 4 * ((a.d.ntype + 1) & ~1));
    const double2 *cv = reinterpret_cast<const double2 *>(smem + a.d.cls_off);
    const int32_t *fl = reinterpret_cast<const int32_t *>(smem + a.d.fld_off);
    const int len = active ? tlen[tid] : 0;
    const int2 *te = tent + (size_t)tid * a.d.kmax;
    __syncthreads();

    double nrmp = 0.0;   // this thread's share of ||w'||^2 of the previous iteration
    for (int loc = 0;; ++loc) {
        const int nv = loc + 1;
        const int nvt = (loc < mk ? nv + m : 0) + 1;   // values of this exchange: V~_i . w~ (raw), B D w~, and LAST the pending norm
        RES_STAMP(0);
        // ---- (a) this workgroup's partial sums: every thread stages its products in LDS, a few threads per value add
        // them in a fixed order (no 64-lane shuffle chains: 36 values x 6 steps cost 5-9 us here)
        double *Pl = a.P + (size_t)loc * kResLd * kResG + wg;   // value sl of this workgroup: Pl[sl * kResG]
        for (int s0 = 0; s0 < nvt; s0 += kResPass) {
            if (loc < mk) {
                // (no test per vector: the entries of V~_i beyond the current basis are zero, their slots are either not
                // read or overwritten just below -- thirty tests and branches cost more than the products)
                if (s0 == 0) {
#pragma unroll
                    for (int i = 0; i < kResPass; ++i) prod[i * LDP + t] = V[i].x * w.x + V[i].y * w.y;
                } else {
#pragma unroll
                    for (int i = kResPass; i < kResMaxV - 1; ++i) prod[(i - kResPass) * LDP + t] = V[i].x * w.x + V[i].y * w.y;
                }
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    if (a.packed) {
                        const int sl = nv + 2 * q;
                        if (2 * q < m && sl >= s0 && sl < s0 + kResPass) prod[(sl - s0) * LDP + t] = pe[q].x * w.x;
                        if (2 * q + 1 < m && sl + 1 >= s0 && sl + 1 < s0 + kResPass) prod[(sl + 1 - s0) * LDP + t] = pe[q].y * w.y;
                    } else {
                        const int sl = nv + q;
                        if (q < m && sl >= s0 && sl < s0 + kResPass) prod[(sl - s0) * LDP + t] = pe[q].x * w.x + pe[q].y * w.y;
                    }
                }
            }
            if (nvt - 1 >= s0 && nvt - 1 < s0 + kResPass) prod[(nvt - 1 - s0) * LDP + t] = nrmp;
            if (s0 == 0) RES_STAMP(8);
            bar_lds();
            if (s0 == 0) RES_STAMP(9);
            const int nvp = min(kResPass, nvt - s0);
            constexpr int TPV = T / 32;   // threads per value (16 | 8), 32 staged products each
            if (t < TPV * nvp) {
                const int sv = t / TPV, j = t % TPV;
                const double *pr = prod + sv * LDP + j;
                double a4[4] = {0.0, 0.0, 0.0, 0.0};   // four chains (a fixed order all the same): 8 dependent additions, not 32
#pragma unroll
                for (int k = 0; k < 32; ++k) a4[k & 3] += pr[TPV * k];
                double acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
                if (TPV == 16) acc += __shfl_xor(acc, 8, kWave);
                acc += __shfl_xor(acc, 4, kWave);
                acc += __shfl_xor(acc, 2, kWave);
                acc += __shfl_xor(acc, 1, kWave);
                if (j == 0) {
                    const int sl = s0 + sv;
                    if (master && a.lam_in_dot) {   // the multiplier entries live in workgroup 0 (rank 0 counts them)
                        if (sl < nvt - 1 && sl < nv)
                            for (int r = 0; r < m; ++r) acc += lamV[sl * 8 + r] * wl[r];
                        else if (sl == nvt - 1 && loc > 0)
                            for (int r = 0; r < m; ++r) acc += wraws[r] * wraws[r];
                    }
                    publish(Pl + (size_t)sl * kResG, acc);
                }
            }
            if (s0 == 0) RES_STAMP(10);
            bar_lds();
            if (s0 == 0) RES_STAMP(11);
        }
        RES_STAMP(1);
        // ---- (b) the sums over the workgroups, in two hops.  (Every workgroup reading ALL partials -- 256 x nvt x 256
        // doubles = 16 MB per iteration through the fabric, the partials being written under system scope -- cost 5.8 us.)
        //   b1: value v belongs to member v mod S of every GROUP of workgroups (wg mod 8: the workgroups of one XCD where the
        //       launch covers the chip).  That member reads the G partials of v, one per thread, adds them in ONE fixed
        //       order and publishes the total for its group -- eight workgroups form every total, all with the same bits;
        //   b2: every workgroup reads the nvt totals of its own group.
        {
            const double *Pb = a.P + (size_t)loc * kResLd * kResG;
            const int NG = a.G >= 64 ? 8 : 1, grp = wg % NG, mem = wg / NG, S = a.G / NG;
            double *Tb = a.P + (size_t)(mk + 1) * kResLd * kResG + ((size_t)loc * 8 + grp) * kResLd;
            if (mem < S) {
                for (int v = mem; v < nvt; v += S) {
                    double pv = 0.0;
                    if (t < a.G) {
                        const unsigned long long t0 = wall_clock64();
                        for (;;) {
                            pv = peek(Pb + (size_t)v * kResG + t);
                            if (!is_sentinel(pv)) break;
                            if (res_timed_out(t0, a.ticks) || flag[0]) {
                                __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                flag[0] = 1;
                                pv = 0.0;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    if (t < kResG) prod[t] = pv;
                    bar_lds();
                    if (t < 32) {
                        double acc = 0.0;
                        for (int j = t; j < a.G; j += 32) acc += prod[j];
                        acc += __shfl_xor(acc, 16, kWave);
                        acc += __shfl_xor(acc, 8, kWave);
                        acc += __shfl_xor(acc, 4, kWave);
                        acc += __shfl_xor(acc, 2, kWave);
                        acc += __shfl_xor(acc, 1, kWave);
                        if (t == 0) publish(Tb + v, acc);
                    }
                    bar_lds();
                }
            }
            if (t < nvt) {
                const unsigned long long t0 = wall_clock64();
                double tv;
                for (;;) {
                    tv = peek(Tb + t);
                    if (!is_sentinel(tv)) break;
                    if (res_timed_out(t0, a.ticks) || flag[0]) {
                        __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        flag[0] = 1;
                        tv = 0.0;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                dots[t] = tv;
            }
        }
        bar_lds();
        if (a.ar.P > 1) {
            // ---- (b') over the ranks: this rank's sums leave ONCE (master workgroup, 8-byte tagged granules into every
            // rank's window); every workgroup of every rank then finds all P contributions in its rank's OWN window and adds
            // them in rank order -- the same bits on every workgroup of every rank
            if (t < 2 * nvt) {
                const uint32_t seq = a.ar.seq + (uint32_t)loc;
                const int slot = (int)(seq & (kArSlots - 1));
                const uint32_t half = reinterpret_cast<const uint32_t *>(dots)[t];
                if (master) {
                    const unsigned long long g = ((unsigned long long)seq << 32) | half;
                    const size_t mine = ((size_t)slot * a.ar.P + a.ar.me) * kArGranules + t;
                    for (int p = 0; p < a.ar.P; ++p) st_sys(a.ar.win[p] + mine, g);
                }
                const unsigned long long *own = a.ar.win[a.ar.me] + (size_t)slot * a.ar.P * kArGranules + t;
                const unsigned long long tw0 = (a.ar.stats && master && t == 0) ? wall_clock64() : 0ull;
                double sum = 0.0;
                bool ok = true;
                for (int p = 0; p < a.ar.P; ++p) {
                    uint32_t lo;
                    ok = granule_wait(own + (size_t)p * kArGranules, seq, a.ar.timeout_ms, lo, a.ar.err) && ok;
                    const uint32_t other = __shfl_xor(lo, 1, kWave);
                    sum += join_halves(lo, other);   // meaningful in even lanes
                }
                if (a.ar.stats && master && t == 0) {
                    atomicAdd(a.ar.stats + 2 * kStatArDots, wall_clock64() - tw0);
                    atomicAdd(a.ar.stats + 2 * kStatArDots + 1, 1ull);
                }
                if (!(t & 1)) dotsg[t >> 1] = sum;
                if (!ok) {
                    raise_comm_error(a.ar.err, 1 + kStatArDots, seq);
                    flag[0] = 1;
                }
            }
            bar_lds();
        }
        const double *dt = a.ar.P > 1 ? dotsg : dots;
        RES_STAMP(2);
        // ---- (c) scalar work, every workgroup for itself.  Pending of iteration loc - 1: its norm has just arrived, so
        // its Givens step runs now -- in the LAST wave, beside wave 0's work for this iteration
        const double nrm2 = dt[nvt - 1];
        if (t == TG && !flag[0] && loc > 0) {
            res_givens(*L, loc - 1, hcolb + 34 * ((loc - 1) & 1), nrm2, gcc, gss, grs, gHr);
            flag[1] = loc - 1;
            if (L->done || L->skip_iter) flag[0] = 1;
        }
        const double s_w = loc > 0 ? inv_norm(nrm2) : 1.0;   // scale factor of V~_loc (v_0 is normalised)
        if (t < kWave && loc < mk) {   // lane i owns basis vector i; the m-vectors are finished inside this wave
            const int i = t;
            double *hcol = hcolb + 34 * (loc & 1);
            const double sci = i < nv ? (i == loc ? s_w : scl[i]) : 0.0;
            const double hi = i < nv ? sci * s_w * dt[i] : 0.0;
            const double ci = hi * sci;
            if (i == 0) scl[loc] = s_w;
            if (i < nv) {
                hs[i] = ci;
                hcol[i] = hi;
            }
            double tu = 0.0, wraw = 0.0;   // lane r < m: B D w' and the multiplier entry of w'
            constexpr int MM = 2 * NP;     // rows this instantiation may carry (m <= MM)
            if (MM > 0) {
                double tsv[MM > 0 ? MM : 1], lsv[MM > 0 ? MM : 1];
#pragma unroll
                for (int r = 0; r < MM; ++r) {
                    tsv[r] = (i < nv && r < m) ? hi * (tbl[i * 8 + r] * sci) : 0.0;   // sum h_i (B D v_i)
                    lsv[r] = (i < nv && r < m) ? ci * lamV[i * 8 + r] : 0.0;           // the MAXPY of the multiplier entries
                }
                // the 2 m wave sums side by side (one after the other they were 2 us of dependent shuffles)
#pragma unroll
                for (int off = kWave / 2; off > 0; off >>= 1) {
#pragma unroll
                    for (int r = 0; r < MM; ++r) {
                        tsv[r] += __shfl_down(tsv[r], off, kWave);
                        lsv[r] += __shfl_down(lsv[r], off, kWave);
                    }
                }
#pragma unroll
                for (int r = 0; r < MM; ++r) {
                    const double tsum0 = __shfl(tsv[r], 0, kWave), lsum0 = __shfl(lsv[r], 0, kWave);
                    if (i == r && r < m) {
                        tu = dt[nv + r] * s_w - tsum0;   // B D w' = B D w - sum h_i (B D v_i)
                        wraw = s_w * wl[r] - lsum0;
                    }
                }
            }
            const double y = i < m ? -(wraw - tu) / shs[i] : 0.0;
            double w1 = tu;
            if (a.fact == SPK_SCHUR_FULL) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q < m) {   // uniform
                        const double yq = __shfl(y, q, kWave);
                        if (i < m) w1 -= grm[i * m + q] * yq;
                    }
            }
            if (i < m) {
                ys[i] = y;
                wraws[i] = wraw;
                w1s[i] = w1;
                lamV[nv * 8 + i] = wraw;
                tbl[nv * 8 + i] = tu;
                if (master && loc + 1 < mk) a.Z[(size_t)(loc + 1) * a.ld + a.nl + i] = y;
            }
        }
        bar_lds();
        if (flag[0] || loc >= mk) break;
        RES_STAMP(3);
        // ---- (d) VecMAXPY in registers, the norm's share, the next PCApply (+ B^T part of the next product)
        w.x *= s_w;
        w.y *= s_w;
#pragma unroll
        for (int i = 0; i < kResMaxV - 1; ++i) {
            if (i < nv) {   // uniform
                const double ai = -hs[i];
                w.x += ai * V[i].x;
                w.y += ai * V[i].y;
            }
        }
#pragma unroll
        for (int i = 1; i < kResMaxV; ++i)
            if (i == nv) V[i] = w;   // uniform: V~_{loc+1} = w'
        nrmp = 0.0;
        if (active) {
            nrmp += w.x * w.x;
            nrmp += w.y * w.y;
        }
        if (loc + 1 >= mk) {   // last iteration of the cycle: only its norm is still wanted
            bar_lds();
            if (t < m) wl[t] = w1s[t];
            if (master && t == TG && flag[1] >= 0) {
                res_givens_flush(*L, a.ka, flag[1], gHr, gcc, gss, grs);
                flag[1] = -1;
            }
            continue;
        }
        double2 sv, zz, cc;
        sv.x = sv.y = 0.0;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (a.packed) {
                sv.x += pe[q].x * ys[2 * q];
                sv.y += pe[q].y * ys[2 * q + 1];
            } else {
                sv.x += pe[q].x * ys[q];
                sv.y += pe[q].y * ys[q];
            }
        }
        zz.x = w.x * dv.x;
        zz.y = w.y * dv.y;
        if (NP > 0 && a.fact == SPK_SCHUR_FULL) {
            zz.x -= sv.x;
            zz.y -= sv.y;
        }
        cc.x = cc.y = 0.0;
        if (NP > 0) {
            cc.x = sv.x / dv.x;
            cc.y = sv.y / dv.y;
        }
        double *Zn = a.Z + (size_t)(loc + 1) * a.ld;
        if (active) {   // write-through: the neighbours gather these rows below
            st_agent(Zn + 2 * br, zz.x);
            st_agent(Zn + 2 * br + 1, zz.y);
            if (a.sr0.n > 0) {   // rows a neighbouring RANK needs: as granules straight into its staging (this exchange's parity)
                const uint32_t hseq = a.sr0.seq + (uint32_t)loc;
                const SendRanges &sr = (loc & 1) ? a.sr1 : a.sr0;
                const unsigned long long tag = (unsigned long long)hseq << 32;
                for (int q = 0; q < sr.n; ++q) {
                    const int64_t e0 = 2 * br - sr.r0[q];
                    if (e0 >= 0 && e0 < sr.len[q]) {
                        const unsigned long long bits = (unsigned long long)__double_as_longlong(zz.x);
                        st_sys(sr.remote[q] + 2 * e0, tag | (bits & 0xffffffffull));
                        st_sys(sr.remote[q] + 2 * e0 + 1, tag | (bits >> 32));
                    }
                    if (e0 + 1 >= 0 && e0 + 1 < sr.len[q]) {
                        const unsigned long long bits = (unsigned long long)__double_as_longlong(zz.y);
                        st_sys(sr.remote[q] + 2 * e0 + 2, tag | (bits & 0xffffffffull));
                        st_sys(sr.remote[q] + 2 * e0 + 3, tag | (bits >> 32));
                    }
                }
            }
        }
        bar_lds();   // (ys, w1s read above by everybody; the multiplier entries of the next w~)
        if (t < m) wl[t] = w1s[t];
        RES_STAMP(4);
        // ---- (e) MatMult: w~ = A z~ (+ c~), rows of z~ gathered as their owners publish them
        if (active) {
#pragma clang fp contract(off)   // every product rounded on its own, added in CSR order (spk_k_dict.hip)
            double s0 = 0.0, s1 = 0.0;
            constexpr int G9 = 9;
            for (int k0 = 0; k0 < len; k0 += G9) {
                int2 e[G9];
                double xv[G9][2];
#pragma unroll
                for (int g = 0; g < G9; ++g) {
                    const bool in = k0 + g < len;
                    e[g] = in ? te[k0 + g] : make_int2(0, 0);
                    xv[g][0] = xv[g][1] = 0.0;
                    if (in) {
                        const int64_t c = br + e[g].x;
                        xv[g][0] = ld_agent(Zn + 2 * c);
                        xv[g][1] = ld_agent(Zn + 2 * c + 1);
                    }
                }
                const unsigned long long t0 = wall_clock64();
                for (;;) {
                    bool miss = false;
#pragma unroll
                    for (int g = 0; g < G9; ++g) {
                        if (k0 + g < len && (is_sentinel(xv[g][0]) || is_sentinel(xv[g][1]))) {
                            const int64_t c = br + e[g].x;
                            xv[g][0] = ld_agent(Zn + 2 * c);
                            xv[g][1] = ld_agent(Zn + 2 * c + 1);
                            miss = true;
                        }
                    }
                    if (!miss) break;
                    if (res_timed_out(t0, a.ticks)) {
                        __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;   // (the missing rows stay at the sentinel: NaN products, and the error word says why)
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int g = 0; g < G9; ++g) {
                    if (k0 + g < len) {
                        const int k = k0 + g;
                        const double2 *cb = cv + (size_t)e[g].y * 4;
                        const u64 wd = cwd[k * T + t];
                        if (a.d.uw[0] > 0) {   // (uniform) one field layout for all classes: spk_k_dict.hip, spmv_dict2_kernel
                            const int lo = (int)(uint32_t)wd, hi = (int)(uint32_t)(wd >> 32);
                            s0 += dict_decode(__builtin_amdgcn_sbfe(lo, 0u, (unsigned)a.d.uw[0]), cb[0]) * xv[g][0];
                            s0 += dict_decode(lo >> (32 - a.d.uw[1]), cb[1]) * xv[g][1];
                            s1 += dict_decode(__builtin_amdgcn_sbfe(hi, 0u, (unsigned)a.d.uw[2]), cb[2]) * xv[g][0];
                            s1 += dict_decode(hi >> (32 - a.d.uw[3]), cb[3]) * xv[g][1];
                        } else {
                            const int32_t *fb = fl + (size_t)e[g].y * 4;
                            s0 += dict_decode(dict_field2(wd, fb[0]), cb[0]) * xv[g][0];
                            s0 += dict_decode(dict_field2(wd, fb[1]), cb[1]) * xv[g][1];
                            s1 += dict_decode(dict_field2(wd, fb[2]), cb[2]) * xv[g][0];
                            s1 += dict_decode(dict_field2(wd, fb[3]), cb[3]) * xv[g][1];
                        }
                    }
                }
            }
            if (a.od.rowptr) {   // off-rank columns of these two rows: the neighbour ranks' granules (fused multiply-adds, as
                                 // in the product kernels' epilogue)
                const uint32_t hseq = a.sr0.seq + (uint32_t)loc;
                const unsigned long long *mine = ((loc & 1) ? a.sr1 : a.sr0).mine;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    double acc = r ? s1 : s0;
                    for (int k = a.od.rowptr[2 * br + r]; k < a.od.rowptr[2 * br + r + 1]; ++k) {
                        const int g = a.od.colidx[k];
                        uint32_t lo = 0, hi = 0;
                        const bool ok = granule_wait(mine + 2 * (size_t)g, hseq, a.sr0.timeout_ms, lo, a.sr0.err) &&
                                        granule_wait(mine + 2 * (size_t)g + 1, hseq, a.sr0.timeout_ms, hi, a.sr0.err);
                        if (!ok) raise_comm_error(a.sr0.err, 21, hseq);
                        acc = __builtin_fma(a.od.val[k], join_halves(lo, hi), acc);
                    }
                    if (r) s1 = acc;
                    else s0 = acc;
                }
            }
            if (NP > 0) {
                s0 += cc.x;
                s1 += cc.y;
            }
            w.x = s0;
            w.y = s1;
        }
        // (what krylov_cycle_end and the host read of the Givens step taken above: written here, off the critical path)
        if (master && t == TG && flag[1] >= 0) {
            res_givens_flush(*L, a.ka, flag[1], gHr, gcc, gss, grs);
            flag[1] = -1;
        }
        RES_STAMP(5);
        bar_lds();
        RES_STAMP(6);
    }
    if (master && t == TG) {
        if (flag[1] >= 0) res_givens_flush(*L, a.ka, flag[1], gHr, gcc, gss, grs);
    }
    // the scale factors for krylov_cycle_end (x += sum y_i sc_i Z~_i)
    if (master && t <= mk) a.sc_out[t] = t <= L->loc_done ? scl[t] : 1.0;
}

// sentinel into the all-to-all buffer and into rows [0, nl) of Z_1 .. Z_{nvec}
__global__ __launch_bounds__(kThreads) void res_arm_kernel(double *P, int64_t nP, double *Z, int64_t ld, int64_t nl, int nvec,
                                                           const int32_t *done)
{
    if (done && *done) return;
    const double armed = __longlong_as_double((long long)kSentinelBits);
    const int64_t stride = (int64_t)gridDim.x * kThreads, i0 = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    for (int64_t i = i0; i < nP; i += stride) P[i] = armed;
    const int64_t nz = nl * nvec;
    for (int64_t i = i0; i < nz; i += stride) Z[(i / nl + 1) * ld + i % nl] = armed;
}

static int res_threads(int rpw) { return rpw <= 256 ? 256 : 512; }
// workgroups of the launch: one per compute unit.  SPK_RES_WGS caps it (test rigs where several processes share ONE
// device: their launches must all be resident together, or each waits for workgroups of its own that cannot start)
static int res_grid_cap(int num_cus)
{
    static const int cap = [] { const char *e = getenv("SPK_RES_WGS"); return e ? atoi(e) : 0; }();
    int g = std::min(num_cus, kResG);
    if (cap > 0) g = std::min(g, cap);
    return g;
}
size_t resident_lds_bytes(const DictDev &A, int T)
{
    const size_t tab = ((size_t)A.lds_bytes + 15) & ~(size_t)15;
    const size_t codes = (size_t)A.kmax * T * 8;
    const size_t dbl = (size_t)kResPass * (T + 8) + 128 + 32 + 68 + 34 * 5 + 32 * 8 * 2 + 8 * 6 + 64;
    return tab + codes + dbl * sizeof(double) + sizeof(KrylovState) + 64;
}
// does the resident cycle kernel take this system?  (np: planes of B D)
bool resident_fits(const DictDev &A, int num_cus, int mk, int np)
{
    if (!A.ok || A.bs != 2 || A.straddle || num_cus < 1 || A.nbrows < 1) return false;
    const int G = (int)std::min<int64_t>(res_grid_cap(num_cus), ((int64_t)A.nbrows + 63) / 64);
    const int rpw = (A.nbrows + G - 1) / G;
    if (rpw > 512 || mk > kResMaxV - 1 || mk < 2 || np > 4) return false;
    return resident_lds_bytes(A, res_threads(rpw)) <= 160 * 1024;
}

// host side: arguments checked by the caller (spk_solver.cpp: resident_fits)
bool cycle_resident(const DictDev &A, int num_cus, ResidentArgs r, const int32_t *done, hipStream_t s)
{
    const int np = r.m == 0 ? 0 : (r.packed ? r.m / 2 : r.m);
    if (!resident_fits(A, num_cus, r.mk, np)) return false;
    const int G = (int)std::min<int64_t>(res_grid_cap(num_cus), ((int64_t)A.nbrows + 63) / 64);
    const int rpw = (A.nbrows + G - 1) / G;
    const int T = res_threads(rpw);
    int dummy = 0;
    ResArgs a{};
    a.d = dict_args(A, &dummy);
    a.G = G;
    a.rpw = rpw;
    a.mk = r.mk; a.m = r.m; a.np = np; a.packed = r.packed; a.fact = r.fact; a.lam_in_dot = r.lam_in_dot;
    a.nl = r.nl; a.ld = r.ld;
    a.V0 = r.V0; a.V1 = r.V1; a.Z = r.Z; a.dinv = r.dinv; a.bd = r.bd; a.ldb = r.ldb; a.shat = r.shat; a.gram = r.gram;
    a.P = r.P; a.ka = r.ka; a.sc_out = r.sc_out; a.err = r.err; a.ticks = r.ticks;
    a.ar = r.ar; a.sr0 = r.sr0; a.sr1 = r.sr1; a.od = r.od;
    a.tab_bytes = (int)(((size_t)A.lds_bytes + 15) & ~(size_t)15);
#ifdef SPK_RES_STAMPS
    static unsigned long long *stamp_buf = nullptr;
    if (!stamp_buf) (void)hipMalloc((void **)&stamp_buf, 8 * 32 * 1024);
    a.stamps = stamp_buf;
#endif
    const size_t lds = resident_lds_bytes(A, T);
    // arm: the all-to-all buffer of this cycle and the rows of Z the product gathers
    {
        const int64_t nP = (int64_t)(r.mk + 1) * (kResG + 8) * kResLd;   // the partials and the groups' totals
        const int64_t tot = std::max<int64_t>(nP, r.nl * (int64_t)(r.mk - 1));
        const int grid = (int)std::min<int64_t>((tot + kThreads - 1) / kThreads, 4096);
        hipLaunchKernelGGL(res_arm_kernel, dim3(std::max(grid, 1)), dim3(kThreads), 0, s, r.P, nP, r.Z, r.ld, r.nl, r.mk - 1, done);
    }
    static bool attr_set[6] = {false, false, false, false, false, false};
#define SPK_RES(TT, NPP, IDX)                                                                                                       \
    do {                                                                                                                            \
        if (!attr_set[IDX]) {   /* more than 64 KB of dynamic LDS has to be asked for */                                            \
            SPK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&cycle_resident_kernel<TT, NPP>),                            \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                                   \
            attr_set[IDX] = true;                                                                                                   \
        }                                                                                                                           \
        hipLaunchKernelGGL((cycle_resident_kernel<TT, NPP>), dim3(G), dim3(TT), lds, s, a);                                         \
    } while (0)
    const int npt = np == 0 ? 0 : (np <= 2 ? 2 : 4);
    if (T == 256) {
        if (npt == 0) SPK_RES(256, 0, 0);
        else if (npt == 2) SPK_RES(256, 2, 1);
        else SPK_RES(256, 4, 2);
    } else {
        if (npt == 0) SPK_RES(512, 0, 3);
        else if (npt == 2) SPK_RES(512, 2, 4);
        else SPK_RES(512, 4, 5);
    }
#undef SPK_RES
#ifdef SPK_RES_STAMPS
    {
        static int shown = 0;
        if (shown++ == 3) {
            (void)hipStreamSynchronize(s);
            std::vector<unsigned long long> h((size_t)G * 32);
            (void)hipMemcpy(h.data(), a.stamps, h.size() * 8, hipMemcpyDeviceToHost);
            const char *nm[6] = {"dots+publish", "all-to-all", "scalars", "maxpy+z", "spmv(own)", "barrier"};
            for (int which = 0; which < 2; ++which) {
                fprintf(stderr, "[resident stamps, loc %d, G %d, T %d] us:", which ? 25 : 10, G, T);
                for (int ph = 0; ph < 6; ++ph) {
                    double sum = 0, mx = 0;
                    for (int g = 0; g < G; ++g) {
                        const double dt = (double)(h[((size_t)g * 2 + which) * 16 + ph + 1] - h[((size_t)g * 2 + which) * 16 + ph]) / 100.0;
                        sum += dt; mx = dt > mx ? dt : mx;
                    }
                    fprintf(stderr, "  %s %.2f (max %.2f)", nm[ph], sum / G, mx);
                }
                {
                    double d[4] = {0, 0, 0, 0};
                    for (int g = 0; g < G; ++g) {
                        const unsigned long long *q = &h[((size_t)g * 2 + which) * 16];
                        d[0] += (double)(q[8] - q[0]) / 100.0; d[1] += (double)(q[9] - q[8]) / 100.0;
                        d[2] += (double)(q[10] - q[9]) / 100.0; d[3] += (double)(q[11] - q[10]) / 100.0;
                    }
                    fprintf(stderr, "  || pass 0: writes %.2f barrier %.2f reduce+publish %.2f barrier %.2f", d[0] / G, d[1] / G, d[2] / G, d[3] / G);
                }
                unsigned long long lo = ~0ull, hi = 0;
                for (int g = 0; g < G; ++g) { lo = std::min(lo, h[((size_t)g * 2 + which) * 16]); hi = std::max(hi, h[((size_t)g * 2 + which) * 16]); }
                fprintf(stderr, "  | entry skew %.2f\n", (double)(hi - lo) / 100.0);
            }
        }
    }
#endif
    return true;
}

int64_t resident_scratch_doubles(int num_cus, int mk) { (void)num_cus; return (int64_t)(mk + 1) * (kResG + 8) * kResLd; }

}  // namespace k
}  // namespace spk
