// spk_k_krylov.hip -- KSPFGMRESCycle's scalar work on the device (init, cycle begin / end, Givens, refinement) and the
// head kernels of the normalised iteration forms (fused_head, maxpy_head).
#include "spk_device.hpp"

namespace spk {
namespace k {

// ---------------------------------------------------------------------------
// Krylov scalar work on the device (one thread): the host never waits for a
// Hessenberg entry, it only enqueues.  Semantics: PETSc KSPFGMRESCycle /
// KSPFGMRESUpdateHessenberg / KSPFGMRESBuildSoln / KSPConvergedDefault.
// ---------------------------------------------------------------------------
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

__device__ __forceinline__ void cycle_begin_body(const KrylovArrays &ka, const double *nrm2, double *tb, int m, double *sc)
{
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
// report: the verdict so far (the state as this cycle finds / leaves it, the reduction and communicator error words)
// is written straight into pinned host memory -- the host reads it behind an event while the cycle is already running
__global__ void krylov_cycle_begin_kernel(KrylovArrays ka, const double *nrm2, double *tb, int m, double *sc, StateReport rp)
{
    if (threadIdx.x != 0) return;
    cycle_begin_body(ka, nrm2, tb, m, sc);
    if (rp.host_state) {
        *rp.host_state = *ka.st;
        rp.host_words[0] = rp.errw ? *rp.errw : 0;
        rp.host_words[1] = rp.commerr ? *rp.commerr : 0;
    }
}
void krylov_cycle_begin(const KrylovArrays &ka, const double *nrm2, hipStream_t s, double *tb, int m, double *sc,
                        const StateReport *rp)
{
    hipLaunchKernelGGL(krylov_cycle_begin_kernel, dim3(1), dim3(64), 0, s, ka, nrm2, tb, m, sc,
                       rp ? *rp : StateReport{nullptr, nullptr, nullptr, nullptr});
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
    __shared__ double lds[kThreads + 4 * (kMaxNv + 2) + 4];
    givens_rider(gr, lds);
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
    const double *__restrict__ dots_prev, SendRanges sr, int packed, const int32_t *__restrict__ done,
    double *__restrict__ wl_out)
{
    // packed: bd holds m/2 parity-interleaved planes (pack_bd_kernel) instead of m dense rows
    // wl_out != nullptr: a side copy of the multiplier entries of c (the un-normalised iteration keeps them beside the basis)
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
            if (wl_out) wl_out[r] = w1;
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
                const SendRanges *srp, int packed, double *wl_out)
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
                           shat, gram, fact, nl, m, z, c, ka, loc_prev, dots_prev, sr, packed, done, wl_out);
    else
        hipLaunchKernelGGL(fused_head_kernel<8>, dim3(grid > 0 ? grid : 1), dim3(kThreads), 0, s, v, nrm, w1raw, dinv, bd, ldb,
                           shat, gram, fact, nl, m, z, c, ka, loc_prev, dots_prev, sr, packed, done, wl_out);
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

// -ksp_gmres_cgs_refinement_type: mode 2 (always) refines unless done; mode 1 (ifneeded)
// refines when ||w'|| < ||h|| (PETSc's test); the second-pass kernels take skip_refine as
// their "done" word.  dots2 is zeroed so that a skipped pass merges as a no-op.
__global__ void krylov_refine_decide_kernel(KrylovArrays ka, int loc, int mode, const double *dots,
                                            const double *nrm2, double *dots2)
{
    KrylovState *st = ka.st;
    for (int j = threadIdx.x; j <= loc; j += blockDim.x) dots2[j] = 0.0;   // (any restart length: the long-restart path too)
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
    for (int j = threadIdx.x; j <= loc; j += blockDim.x) dots[j] += dots2[j];
    if ((int)threadIdx.x < nn) nrm[threadIdx.x] = nrm_b[threadIdx.x];
}
void krylov_refine_merge(const KrylovArrays &ka, int loc, double *dots, const double *dots2, double *nrm,
                         const double *nrm_b, int nn, hipStream_t s)
{
    hipLaunchKernelGGL(krylov_refine_merge_kernel, dim3(1), dim3(64), 0, s, ka, loc, dots, dots2, nrm, nrm_b, nn);
}

// Back substitution for the loc_done columns built in this cycle (KSPFGMRESBuildSoln).  The triangle is staged in LDS by
// the whole workgroup (450 dependent global loads took 47 us); then ONE WAVE solves it column by column: lane k keeps
// the running right-hand side t_k of row k in a register, y_j = t_j / H_jj is broadcast through v_readlane and every
// row above takes its update t_k -= H_kj y_j at once.  The critical path is one division and one multiply-subtract per
// unknown (measured: 20.7 us -> ~5 us at restart 30 against one lane walking the triangle row by row, which pays an LDS
// round trip or a readlane per ENTRY).  Each row's sum runs over j in descending instead of ascending order: the same
// terms, one rounding each, no tree -- deterministic and identical on every rank; it differs from the serial loop of
// the oracle by rounding only.
// pend.loc >= 0: the Givens step of the cycle's last iteration runs first, in the same launch.
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__global__ __launch_bounds__(256) void krylov_cycle_end_kernel(KrylovArrays ka, const double *sc, GivensRider pend)
{
    __shared__ double Hs[(kMaxNv) * (kMaxNv + 1)];
    __shared__ double rss[kMaxNv + 2];
    if (pend.loc >= 0) {
        givens_block(pend.ka, pend.loc, pend.h, pend.nrm2);
        __syncthreads();
    }
    KrylovState *st = ka.st;
    const int n = st->loc_done, ldh = ka.ldh;
    {
        // four threads per column, every load of a thread requested before its first LDS store (the strided loop with
        // a division per entry made four dependent round trips of it: ~10 us of this kernel's 20)
        const int j = threadIdx.x >> 2, k0 = threadIdx.x & 3;
        constexpr int kPer = (kMaxNv + 3) / 4;
        double hv[kPer];
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int k = k0 + 4 * u;
            hv[u] = (j < n && k <= j) ? ka.H[(size_t)ldh * j + k] : 0.0;   // upper triangle only
        }
        const double rv = (int)threadIdx.x < n ? ka.rs[threadIdx.x] : 0.0;
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int k = k0 + 4 * u;
            if (j < n && k <= j) Hs[j * n + k] = hv[u];
        }
        if ((int)threadIdx.x < n) rss[threadIdx.x] = rv;
    }
    __syncthreads();
    if (threadIdx.x >= kWave) return;
    const int lane = threadIdx.x;  // n <= kMaxNv - 2 < 64: lane k owns row k
    const double scl = (sc && lane < n) ? sc[lane] : 1.0;
    double t = lane < n ? rss[lane] : 0.0, y = 0.0;
    double hcol = (n > 0 && lane < n - 1) ? Hs[(n - 1) * n + lane] : 0.0;   // column n-1 above the diagonal
    for (int j = n - 1; j >= 0; --j) {
        const double piv = Hs[j * n + j];
        const double hnext = (j > 0 && lane < j - 1) ? Hs[(j - 1) * n + lane] : 0.0;   // next column, requested early
        if (piv == 0.0) {  // (uniform)
            if (lane == 0) {
                if (st->reason >= 0) st->reason = SPK_DIVERGED_BREAKDOWN;
                st->done = 1;
                st->loc_done = 0;
            }
            return;
        }
        const double yj = readlane_f64(t, j) / piv;
        if (lane == j) y = yj;
        if (lane < j) t -= hcol * yj;
        hcol = hnext;
    }
    // (un-normalised Z~_k: x += sum y_k sc_k Z~_k)
    if (lane < n) ka.nrs[lane] = sc ? y * scl : y;
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
void krylov_cycle_end(const KrylovArrays &ka, hipStream_t s, const double *sc, int restart, const GivensRider *pending)
{
    if (restart > kMaxNv - 2) {
        if (pending) krylov_givens(pending->ka, pending->loc, pending->h, pending->nrm2, s);
        hipLaunchKernelGGL(krylov_cycle_end_big_kernel, dim3(1), dim3(64), 0, s, ka);
    } else {
        hipLaunchKernelGGL(krylov_cycle_end_kernel, dim3(1), dim3(256), 0, s, ka, sc, pending ? *pending : no_rider());
    }
}


}  // namespace k
}  // namespace spk
