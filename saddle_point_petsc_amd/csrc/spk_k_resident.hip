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
//   * per iteration the workgroups meet twice: (a) the inner products -- every workgroup publishes its partial sums and
//     every workgroup reads and adds ALL of them in one fixed order (an all-to-all of <= 40 doubles x 256; values are their
//     own arrival flags, the buffer is armed with a sentinel once per cycle), so all workgroups hold the same bits and run
//     the same scalar work (Hessenberg column, Givens rotation, KSPConvergedDefault) redundantly -- no broadcast, no
//     second hop; (b) the product: a workgroup stores its rows of z~ = M^-1 w' write-through into Z_{j+1} (armed with the
//     sentinel), its neighbours gather them with polling loads.  ||w'||^2 rides in the NEXT iteration's all-to-all.
// Same algorithm as the default form (classical Gram-Schmidt, two reductions per iteration, un-normalised basis with one
// scale factor per vector: include/spk.h, SPK_ITER_UNNORM); products of the A block bit-identical to the CSR loop.
// Every wait is bounded (FinErr ticks): a workgroup that is not resident, or a lost partial, raises the context's sticky
// execution-error word (SPK_ERR_HIP), it cannot hang the device.
#include "spk_dict.hpp"

namespace spk {
namespace k {

constexpr int kResMaxV = 31;   // basis vectors a thread holds (restart <= 30)
constexpr int kResLd = 64;     // doubles per workgroup and iteration in the all-to-all buffer; [63] = ||w'||^2 partial

__device__ __forceinline__ bool res_timed_out(unsigned long long t0, uint32_t ticks) { return wall_clock64() - t0 > (unsigned long long)ticks; }

// One Arnoldi step's scalar work on the workgroup's OWN copy of the Krylov scalars (every workgroup runs it on the same
// inputs); the master also writes what krylov_cycle_end and the host read.  Semantics of givens_block_lds (spk_device.hpp).
__device__ __forceinline__ void res_givens(KrylovState &L, const KrylovArrays &ka, int loc, const double *hcol, double nrm2,
                                           double *cc, double *ss, double *rs, double *Hr, bool master)
{
    const double rs_loc = rs[loc];
    const double tt = sqrt(nrm2);
    if (isnan(tt) || isinf(tt)) {
        L.rnorm = tt;
        L.reason = SPK_DIVERGED_NANORINF;
        L.done = L.skip_iter = 1;
        if (master) *ka.st = L;
        return;
    }
    double hapbnd = fabs(tt / rs_loc);
    if (hapbnd > 1e-30) hapbnd = 1e-30;
    const int hapend = !(tt > hapbnd);
    L.tt = tt;
    L.inv_tt = hapend ? 1.0 : 1.0 / tt;
    double run = hcol[0];
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
            if (master) *ka.st = L;
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
    if (master) {
        double *Hg = ka.H + (size_t)ka.ldh * loc;
        for (int j = 0; j <= loc + 1; ++j) Hg[j] = Hr[j];
        if (!hapend) {
            ka.cc[loc] = cc[loc];
            ka.ss[loc] = ss[loc];
            ka.rs[loc + 1] = rs[loc + 1];
            ka.rs[loc] = rs[loc];
        }
        if (L.its < ka.hist_cap) ka.hist[L.its] = rnorm;
        *ka.st = L;
    }
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
    double *P;             // all-to-all buffer, (mk + 1) x G x kResLd, armed
    KrylovArrays ka;
    double *sc_out;        // scale factors of the un-normalised basis (krylov_cycle_end)
    int32_t *err;
    uint32_t ticks;
    int tab_bytes;         // LDS bytes of the matrix tables (16-byte multiple)
};

// NP: planes of B D a thread holds (0: K = A; packed: m = 2 NP, dense: m = NP)
template <int T, int NP>
__global__ __launch_bounds__(T) void cycle_resident_kernel(ResArgs a)
{
    constexpr int W = T / kWave;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // ---- LDS: matrix tables, then the scalar work space (all doubles)
    double *ws = reinterpret_cast<double *>(smem + a.tab_bytes);
    double *red = ws;                       // W x 64
    double *dots = red + W * kResLd;        // 64
    double *hs = dots + 64;                 // 32: MAXPY coefficients h_i sc_i
    double *hcol = hs + 32;                 // 34: Hessenberg column of the iteration (scaled)
    double *scl = hcol + 34;                // 34: scale factors
    double *gcc = scl + 34, *gss = gcc + 34, *grs = gss + 34, *gHr = grs + 34;   // rotations, rhs, column scratch
    double *lamV = gHr + 34;                // 32 x 8: multiplier entries of the basis vectors
    double *tbl = lamV + 32 * 8;            // 32 x 8: B D V~_i
    double *ys = tbl + 32 * 8, *wraws = ys + 8, *tus = wraws + 8, *w1s = tus + 8, *wl = w1s + 8, *shs = wl + 8;   // 8 each
    double *grm = shs + 8;                  // 64
    KrylovState *L = reinterpret_cast<KrylovState *>(grm + 64);
    int *flag = reinterpret_cast<int *>(L + 1);   // [0] stop (done / timed out)

    const int wg = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int m = a.m, mk = a.mk;
    const bool master = wg == 0;
    const int64_t br = (int64_t)wg * a.rpw + t;
    const bool active = t < a.rpw && br < a.d.nbrows;
    const double armed = __longlong_as_double((long long)kSentinelBits);

    // ---- prologue: state, tables, this thread's entries
    for (int i = t; i < (int)(reinterpret_cast<double *>(L) - ws); i += T) ws[i] = 0.0;
    if (t == 0) {
        *L = *a.ka.st;
        flag[0] = 0;
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
    dict_load_lds(a.d, a.d.nclass * 4, smem);   // (ends with a barrier)
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
    const int len = active ? tlen[tid] : 0;
    const int2 *te = tent + (size_t)tid * a.d.kmax;
    __syncthreads();

    double nrmp = 0.0;   // this thread's share of ||w'||^2 of the previous iteration
    for (int loc = 0;; ++loc) {
        const int nv = loc + 1;
        const int nvals = loc < mk ? nv + m : 0;
        // ---- (a) partial inner products V~_i . w~ (raw: scaled where consumed), B D w~, and the pending norm
        double *Pl = a.P + ((size_t)loc * a.G + wg) * kResLd;
        if (loc < mk) {
#pragma unroll
            for (int i = 0; i < kResMaxV - 1; ++i) {
                if (i < nv) {   // uniform
                    const double s = wave_sum(V[i].x * w.x + V[i].y * w.y);
                    if (lane == 0) red[wave * kResLd + i] = s;
                }
            }
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                if (a.packed) {
                    const double s0 = wave_sum(pe[q].x * w.x), s1 = wave_sum(pe[q].y * w.y);
                    if (lane == 0) {
                        red[wave * kResLd + nv + 2 * q] = s0;
                        red[wave * kResLd + nv + 2 * q + 1] = s1;
                    }
                } else {
                    const double s0 = wave_sum(pe[q].x * w.x + pe[q].y * w.y);
                    if (lane == 0) red[wave * kResLd + nv + q] = s0;
                }
            }
        }
        {
            const double s = wave_sum(nrmp);
            if (lane == 0) red[wave * kResLd + 63] = s;
        }
        __syncthreads();
        if (t < nvals || t == 63) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < W; ++j) s += red[j * kResLd + t];
            if (master && a.lam_in_dot) {   // the multiplier entries live in workgroup 0 (rank 0 counts them)
                if (t < nv)
                    for (int r = 0; r < m; ++r) s += lamV[t * 8 + r] * wl[r];
                else if (t == 63 && loc > 0)
                    for (int r = 0; r < m; ++r) s += wraws[r] * wraws[r];
            }
            publish(Pl + t, s);
        }
        __syncthreads();
        // ---- every workgroup reads ALL partials and adds them in workgroup order: the same bits everywhere
        {
            const int i = lane;
            const bool live = i < nvals || i == 63;
            double acc = 0.0;
            const double *Pi = a.P + (size_t)loc * a.G * kResLd + i;
            constexpr int B = 8;
            for (int g0 = wave; g0 < a.G; g0 += W * B) {
                double v[B];
                const unsigned long long t0 = wall_clock64();
                bool miss;
                do {
                    miss = false;
#pragma unroll
                    for (int u = 0; u < B; ++u) {
                        const int g = g0 + u * W;
                        v[u] = (live && g < a.G) ? peek(Pi + (size_t)g * kResLd) : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < B; ++u) miss = miss || is_sentinel(v[u]);
                    if (miss) {
                        if (res_timed_out(t0, a.ticks) || flag[0]) {
                            __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            flag[0] = 1;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                } while (miss);
#pragma unroll
                for (int u = 0; u < B; ++u) acc += v[u];
            }
            red[wave * kResLd + i] = acc;
        }
        __syncthreads();
        if (t < 64) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < W; ++j) s += red[j * kResLd + t];
            dots[t] = s;
        }
        __syncthreads();
        // ---- scalar work, every workgroup for itself: first what was pending of iteration loc - 1
        if (t == 0 && !flag[0] && loc > 0) {
            const double nrm2 = dots[63];
            scl[loc] = inv_norm(nrm2);
            res_givens(*L, a.ka, loc - 1, hcol, nrm2, gcc, gss, grs, gHr, master);
            if (L->done || L->skip_iter) flag[0] = 1;
        }
        __syncthreads();
        if (flag[0] || loc >= mk) break;
        const double s_w = scl[loc];
        if (t < kWave) {   // lane i owns basis vector i
            const int i = t;
            const double sci = i < nv ? scl[i] : 0.0;
            const double hi = i < nv ? sci * s_w * dots[i] : 0.0;
            if (i < nv) {
                hs[i] = hi * sci;
                hcol[i] = hi;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (r < m) {   // uniform
                    const double tsum = wave_sum(hi * (i < nv ? tbl[i * 8 + r] * sci : 0.0));
                    if (i == 0) tus[r] = dots[nv + r] * s_w - tsum;   // B D w' = B D w - sum h_i (B D v_i)
                }
            }
        }
        __syncthreads();
        if (t < m) {
            const int r = t;
            double wraw = s_w * wl[r];
            for (int i = 0; i < nv; ++i) wraw += -hs[i] * lamV[i * 8 + r];   // the MAXPY of the multiplier entries
            const double y = -(wraw - tus[r]) / shs[r];
            wraws[r] = wraw;
            ys[r] = y;
        }
        __syncthreads();
        if (t < m) {
            const int r = t;
            double w1 = tus[r];
            if (a.fact == SPK_SCHUR_FULL)
                for (int q = 0; q < m; ++q) w1 -= grm[r * m + q] * ys[q];
            w1s[r] = w1;
            lamV[nv * 8 + r] = wraws[r];
            tbl[nv * 8 + r] = tus[r];
            if (master && loc + 1 < mk) a.Z[(size_t)(loc + 1) * a.ld + a.nl + r] = ys[r];
        }
        // ---- VecMAXPY in registers, the norm's share, the next PCApply (+ B^T part of the next product)
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
            __syncthreads();
            if (t < m) wl[t] = w1s[t];
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
        }
        __syncthreads();   // (ys, w1s read above by everybody; the multiplier entries of the next w~)
        if (t < m) wl[t] = w1s[t];
        // ---- MatMult: w~ = A z~ (+ c~), rows of z~ gathered as their owners publish them
        if (active) {
#pragma clang fp contract(off)   // every product rounded on its own, added in CSR order (spk_k_dict.hip)
            double s0 = 0.0, s1 = 0.0;
            constexpr int G9 = 9;
            for (int k0 = 0; k0 < len; k0 += G9) {
                int2 e[G9];
                DictRaw<2> raw[G9];
                double xv[G9][2];
#pragma unroll
                for (int g = 0; g < G9; ++g) {
                    const bool in = k0 + g < len;
                    e[g] = in ? te[k0 + g] : make_int2(0, 0);
                    xv[g][0] = xv[g][1] = 0.0;
                    if (in) {
                        dict_issue<2>(a.d, k0 + g, br, raw[g]);
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
                        const double2 *cb = cv + (size_t)e[g].y * 4;
                        int code[4];
                        dict_unpack<2>(raw[g], code);
                        s0 += dict_decode(code[0], cb[0]) * xv[g][0];
                        s0 += dict_decode(code[1], cb[1]) * xv[g][1];
                        s1 += dict_decode(code[2], cb[2]) * xv[g][0];
                        s1 += dict_decode(code[3], cb[3]) * xv[g][1];
                    }
                }
            }
            if (NP > 0) {
                s0 += cc.x;
                s1 += cc.y;
            }
            w.x = s0;
            w.y = s1;
        }
        __syncthreads();
    }
    // the scale factors for krylov_cycle_end (x += sum y_i sc_i Z~_i)
    if (master && t <= mk) a.sc_out[t] = t <= L->loc_done ? scl[t] : 1.0;
    (void)armed;
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

size_t resident_lds_bytes(const DictDev &A, int T)
{
    const size_t tab = ((size_t)A.lds_bytes + 15) & ~(size_t)15;
    const size_t dbl = (size_t)(T / kWave) * kResLd + 64 + 32 + 34 * 6 + 32 * 8 * 2 + 8 * 6 + 64;
    return tab + dbl * sizeof(double) + sizeof(KrylovState) + 64;
}

// host side: arguments checked by the caller (spk_solver.cpp); returns false when the launch shape does not fit
bool cycle_resident(const DictDev &A, int num_cus, ResidentArgs r, const int32_t *done, hipStream_t s)
{
    if (A.bs != 2 || !A.ok) return false;
    const int G = std::min<int64_t>(num_cus, ((int64_t)A.nbrows + 63) / 64);
    const int rpw = (A.nbrows + G - 1) / G;
    if (rpw > 512 || r.mk > kResMaxV - 1) return false;
    const int T = rpw <= 256 ? 256 : 512;
    const int np = r.m == 0 ? 0 : (r.packed ? r.m / 2 : r.m);
    if (np > 4) return false;
    int dummy = 0;
    ResArgs a{};
    a.d = dict_args(A, &dummy);
    a.G = G;
    a.rpw = rpw;
    a.mk = r.mk; a.m = r.m; a.np = np; a.packed = r.packed; a.fact = r.fact; a.lam_in_dot = r.lam_in_dot;
    a.nl = r.nl; a.ld = r.ld;
    a.V0 = r.V0; a.V1 = r.V1; a.Z = r.Z; a.dinv = r.dinv; a.bd = r.bd; a.ldb = r.ldb; a.shat = r.shat; a.gram = r.gram;
    a.P = r.P; a.ka = r.ka; a.sc_out = r.sc_out; a.err = r.err; a.ticks = r.ticks;
    a.tab_bytes = (int)(((size_t)A.lds_bytes + 15) & ~(size_t)15);
    const size_t lds = resident_lds_bytes(A, T);
    // arm: the all-to-all buffer of this cycle and the rows of Z the product gathers
    {
        const int64_t nP = (int64_t)(r.mk + 1) * G * kResLd;
        const int64_t tot = std::max<int64_t>(nP, r.nl * (int64_t)(r.mk - 1));
        const int grid = (int)std::min<int64_t>((tot + kThreads - 1) / kThreads, 4096);
        hipLaunchKernelGGL(res_arm_kernel, dim3(std::max(grid, 1)), dim3(kThreads), 0, s, r.P, nP, r.Z, r.ld, r.nl, r.mk - 1, done);
    }
#define SPK_RES(TT, NPP) hipLaunchKernelGGL((cycle_resident_kernel<TT, NPP>), dim3(G), dim3(TT), lds, s, a)
    const int npt = np == 0 ? 0 : (np <= 2 ? 2 : 4);
    if (T == 256) {
        if (npt == 0) SPK_RES(256, 0);
        else if (npt == 2) SPK_RES(256, 2);
        else SPK_RES(256, 4);
    } else {
        if (npt == 0) SPK_RES(512, 0);
        else if (npt == 2) SPK_RES(512, 2);
        else SPK_RES(512, 4);
    }
#undef SPK_RES
    return true;
}

int64_t resident_scratch_doubles(int num_cus, int mk) { return (int64_t)(mk + 1) * num_cus * kResLd; }

}  // namespace k
}  // namespace spk
