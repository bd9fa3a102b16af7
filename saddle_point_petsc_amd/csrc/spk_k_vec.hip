// spk_k_vec.hip -- VecMDot / VecMAXPY (+ norms), level-1 streams, the short-and-wide constraint block, the block steps
// of PCApply_FieldSplit_Schur / PCApply_Jacobi.  gfx950, wave64, HBM-bound; fixed-order cross-workgroup reductions.
#include "spk_device.hpp"

namespace spk {
namespace k {

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

__global__ __launch_bounds__(kVT) void sqnorm_kernel(double *__restrict__ x, int64_t n2,
                                                          int64_t n_dot, double *__restrict__ partials,
                                                          double *__restrict__ out, FinErr fe,
                                                          const int32_t *__restrict__ done,
                                                          const double *__restrict__ sa, const double *__restrict__ sb)
{
    // sa != nullptr: x = sa - sb formed and stored on the way (see sqnorm_bd_kernel); n2 then covers the whole vector
    if (done && *done) return;
    __shared__ double red[kVT];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kVT + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kVT) {
        double2 v;
        if (sa) {
            const double2 av = reinterpret_cast<const double2 *>(sa)[i], bv = reinterpret_cast<const double2 *>(sb)[i];
            v.x = av.x - bv.x;
            v.y = av.y - bv.y;
            reinterpret_cast<double2 *>(x)[i] = v;
        } else {
            v = reinterpret_cast<const double2 *>(x)[i];
        }
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
    hipLaunchKernelGGL(sqnorm_kernel, dim3(grid), dim3(kVT), 0, s, const_cast<double *>(x), n2, n_dot, f.partials, f.out,
                       FinErr{f.err, f.fin_ticks}, done, (const double *)nullptr, (const double *)nullptr);
}
// x[0..n) = sa - sb and ||x[0..n_dot)||^2 in one pass (the restart's true residual and the next cycle's starting norm)
void sqnorm_sub(const double *sa, const double *sb, double *x, int64_t n, int64_t n_dot, const Finish &f, const int32_t *done,
                hipStream_t s)
{
    const int64_t n2 = (n + 1) / 2;
    const int grid = vec_grid(n2);
    hipLaunchKernelGGL(sqnorm_kernel, dim3(grid), dim3(kVT), 0, s, x, n2, n_dot, f.partials, f.out, FinErr{f.err, f.fin_ticks},
                       done, sa, sb);
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
// the same with t = Bt y1 already formed (general constraint blocks: the product by the tiled stream kernel)
__global__ __launch_bounds__(kThreads) void bt_combine_kernel(int mode, int nrows, const double *__restrict__ dinv,
                                                              const double *__restrict__ x0, const double *__restrict__ t,
                                                              double *__restrict__ y0, const int32_t *__restrict__ done)
{
    if (done && *done) return;
    for (int r = blockIdx.x * kThreads + threadIdx.x; r < nrows; r += gridDim.x * kThreads) {
        const double c = t[r];
        if (mode >= 2) {
            y0[r] = mode == 2 ? x0[r] - c : c;
            continue;
        }
        const double d = dinv[r], xv = x0[r];
        y0[r] = (mode == 0) ? (xv - c) * d : xv * d - c * d;
    }
}
void bt_update(int mode, const CsrDev &Bt, const double *dinv, const double *x0, const double *y1,
               double *y0, const int32_t *done, hipStream_t s, double *scratch)
{
    if (Bt.nrows == 0) return;
    if (scratch && Bt.ntiles > 0) {
        // a block of many short rows (B^T: a handful of entries per row, scattered over the multipliers): one thread per
        // row reads its entries uncoalesced (measured 344 us on the 96^3 divergence block) -- the tiled stream kernel
        // streams them (52 us), the combination is one more pass over the vector
        spmv(Bt, y1, scratch, nullptr, nullptr, done, s);
        const int grid = std::min((Bt.nrows + kThreads - 1) / kThreads, kMaxBlocks * 4);
        hipLaunchKernelGGL(bt_combine_kernel, dim3(grid), dim3(kThreads), 0, s, mode, Bt.nrows, dinv, x0, scratch, y0, done);
        return;
    }
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
__global__ __launch_bounds__(512) void sqnorm_bd_kernel(double *__restrict__ x, int64_t n2, int64_t n_dot,
                                                        const double *__restrict__ bd, int64_t ldb, int64_t n_bd,
                                                        int m, double *__restrict__ w1side,
                                                        double *__restrict__ partials, double *__restrict__ out,
                                                        FinErr fe, const int32_t *__restrict__ done,
                                                        const double *__restrict__ sa, const double *__restrict__ sb)
{
    // sa != nullptr: x = sa - sb is formed (and stored) on the way -- the restart's true residual b - K x and the
    // norms the next cycle starts from in ONE pass
    if (done && *done) return;
    constexpr int T = 512, NR = MP + 1, W = T / kWave;
    __shared__ double red[(W * NR > T) ? W * NR : T];
    double acc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < n2; i += (int64_t)gridDim.x * T) {
        double2 v;
        if (sa) {
            const double2 av = reinterpret_cast<const double2 *>(sa)[i], bv = reinterpret_cast<const double2 *>(sb)[i];
            v.x = av.x - bv.x;
            v.y = av.y - bv.y;
            reinterpret_cast<double2 *>(x)[i] = v;
        } else {
            v = reinterpret_cast<const double2 *>(x)[i];
        }
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
void sqnorm_bd(double *x, int64_t n, int64_t n_dot, const double *bd, int64_t ldb, int64_t n_bd, int m,
               double *w1side, const Finish &f, const int32_t *done, hipStream_t s, const double *sa, const double *sb)
{
    const int64_t n2 = (n + 1) / 2;
    const int grid = vec_grid(n2, 512);
    if (m <= 4)
        hipLaunchKernelGGL(sqnorm_bd_kernel<4>, dim3(grid), dim3(512), 0, s, x, n2, n_dot, bd, ldb, n_bd, m, w1side, f.partials, f.out, FinErr{f.err, f.fin_ticks}, done, sa, sb);
    else
        hipLaunchKernelGGL(sqnorm_bd_kernel<8>, dim3(grid), dim3(512), 0, s, x, n2, n_dot, bd, ldb, n_bd, m, w1side, f.partials, f.out, FinErr{f.err, f.fin_ticks}, done, sa, sb);
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


}  // namespace k
}  // namespace spk
