// spk_k_iter.hip -- the fused iteration kernels: A / A' (SpMV + MDot / normalisation), B (MAXPY + norm + next PCApply;
// un-normalised basis mode = the default iteration), BA (MAXPY and the next SpMV behind neighbour flags).
#include "spk_device.hpp"

namespace spk {
namespace k {

// ---------------------------------------------------------------------------
// Two- / three-launch iteration on a NORMALISED basis (opts.iteration_form = 2 / 3: opt-in since the un-normalised
// form -- kernel B in `sc` mode behind a plain MDot, the plain SpMV carrying the Givens step -- became the default;
// kept for comparison and as the origin of kernel B).  On a rank's slab of a strong-scaling run and the
// 256^2 / 512^2 grids every kernel of the four-launch
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
        double lam2 = 0.0;  // the multiplier entries' share of ||w'||^2 (rank 0 only)
        if (threadIdx.x == 0)
            for (int r = 0; r < m; ++r) lam2 += b.lam_in_dot ? wraws[r] * wraws[r] : 0.0;
        if (b.defer_fin) {
            // un-normalised basis: nothing in the product launch that follows needs ||w'||^2 except its rider workgroup
            // (scale factor, Givens step), so the rider also REDUCES the partials (and all-reduces the sum) beside the
            // row tiles: this launch ends with its last streaming workgroup -- no publish -> re-read tail
            if (threadIdx.x == 0) publish(b.partials + (size_t)gmain * kPartialLd, lam2);
            return;
        }
        final_reduce(b.partials, gmain, kPartialLd, 1, red, FinErr{b.err, b.fin_ticks});
        if (threadIdx.x == 0) red[0] = red[0] + lam2;
        __syncthreads();
        if (b.ar.P) peer_allreduce_block(b.ar, red, 1, b.out);
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
                // streamed out past the L2 (non-temporal stores): nobody on this XCD reads them again before the kernel
                // boundary writes them back anyway (same box, alternating: 512^2 60.6 -> 59.1 us per iteration, 1024^2 203.3 -> 202.1, 1/8 slab 39.3 -> 39.0)
                st2nt(b.w, i, wv[u]);
                st2nt(b.zun, i, zz);
                if (MP > 0) {
                    cc.x = sv[u].x / dv[u].x;
                    cc.y = sv[u].y / dv[u].y;
                    st2nt(b.c, i, cc);
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

int iter_maxpy_uhead(IterB b, hipStream_t s)   // returns the number of partial rows its norm is spread over (defer_fin)
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
    return b.gmain;
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
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
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


}  // namespace k
}  // namespace spk
