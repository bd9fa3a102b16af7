// spk_solver.cpp -- operator, preconditioner and the device-resident FGMRES.
//
// Replaces what executes below KSPSolve(ksp, f, *u) at
// /root/reference/src/SaddlePointProblem.c:70 when the reference is run with
// -ksp_type fgmres -pc_type fieldsplit -pc_fieldsplit_type schur ... (options
// read at :67): PETSc's KSPSolve_FGMRES drives the loop from the host and waits
// on every dot product; here the host only ENQUEUES a whole restart cycle on
// one HIP stream.  The Hessenberg column, the Givens rotations, the convergence
// test and the back substitution run in single-wave kernels on the device, and
// every kernel of an iteration starts with "if (*done) return", so the iterate
// is exactly the one a stop-at-convergence loop would produce while the host
// looks at the state only once per cycle (or every opts.check_every
// iterations).  Inner products across ranks go through Comm::allreduce_sum on
// the same stream (RCCL), never through the host.
#include <algorithm>
#include <chrono>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "spk_internal.hpp"

using namespace spk;

void spk_ctx::ensure_scratch()
{
    if (!partials.p) {
        partials.alloc((size_t)k::kMaxBlocks * k::kPartialLd);
        k::arm_partials(partials.p, partials.n, stream);
        SPK_HIP(hipStreamSynchronize(stream));
    }
    if (!errw.p) errw.alloc(4);
    if (!small.p) small.alloc(512);
    if (!y1tmp.p) y1tmp.alloc(64);
    if (!ttmp.p) ttmp.alloc(64);
}

void spk_ctx::check_device_error()
{
    if (!errw.p) return;
    int32_t e = 0;
    SPK_HIP(hipMemcpy(&e, errw.p, sizeof e, hipMemcpyDeviceToHost));
    if (!e) return;
    // the word is sticky on the device; clear it and put every slot of the partials buffer back to the
    // sentinel (a workgroup that publishes late would otherwise leave a stale "arrived" slot behind)
    SPK_HIP(hipStreamSynchronize(stream));
    SPK_HIP(hipMemset(errw.p, 0, sizeof(int32_t)));
    k::arm_partials(partials.p, partials.n, stream);
    SPK_HIP(hipStreamSynchronize(stream));
    fail(SPK_ERR_HIP, "a cross-workgroup reduction timed out on the device (a workgroup never published its partial "
                      "sums within %.1f s): execution failure, the result of this call is not valid", fin_ticks / 1e8);
}

void spk_ctx::ensure_vectors()
{
    const int64_t want = ((int64_t)n_local + m + 255) / 256 * 256;
    if (want != ld) {
        ld = want;
        ws_restart = -1;
        tmp.release();
        zun.release();   // every vector sized by ld goes with it (a second KSPSetOperators may bring a larger system)
        tmpb.release();
        bt_cached = nullptr;
        stage_x.release();
        stage_y.release();
        xsol.release();
        rhs.release();
    }
    if (!tmp.p) tmp.alloc((size_t)ld);
    if (!xsol.p) xsol.alloc((size_t)ld);
    if (!rhs.p) rhs.alloc((size_t)ld);
    if (!stage_x.p) stage_x.alloc((size_t)ld);
    if (!stage_y.p) stage_y.alloc((size_t)ld);
}

// Pageable host array -> device through two pinned staging buffers: host threads fill one while the other is on
// the wire (a plain hipMemcpy of pageable memory runs far below the link: 0.35-0.55 s for the 461 MB of the
// 1024^2 matrix in round 1, against ~10 ms of PCIe time).
void spk_ctx::upload_staged(void *dst, const void *src, size_t bytes)
{
    constexpr size_t kChunk = (size_t)32 << 20;
    if (bytes == 0) return;
    if (bytes < ((size_t)1 << 20)) {  // small arrays: not worth the pipeline
        SPK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
        SPK_HIP(hipStreamSynchronize(stream));
        return;
    }
    for (int i = 0; i < 2; ++i) {
        if (!pin[i]) SPK_HIP(hipHostMalloc(&pin[i], kChunk, hipHostMallocDefault));
        if (!pin_ev[i]) SPK_HIP(hipEventCreateWithFlags(&pin_ev[i], hipEventDisableTiming));
    }
    int which = 0;
    for (size_t off = 0; off < bytes; off += kChunk, which ^= 1) {
        const size_t len = std::min(kChunk, bytes - off);
        SPK_HIP(hipEventSynchronize(pin_ev[which]));  // the copy that last used this buffer has left it
        char *stage = (char *)pin[which];
        const char *from = (const char *)src + off;
        parallel_for((int64_t)len, [&](int64_t b0, int64_t b1, int) { std::memcpy(stage + b0, from + b0, (size_t)(b1 - b0)); }, 16);
        SPK_HIP(hipMemcpyAsync((char *)dst + off, stage, len, hipMemcpyHostToDevice, stream));
        SPK_HIP(hipEventRecord(pin_ev[which], stream));
    }
}

namespace spk {

// ---------------------------------------------------------------------------
// KSPSetOperators: upload one block (SaddlePointProblem.c:66; the nest at :45-60)
// ---------------------------------------------------------------------------
// host array -> fresh device buffer through the pinned staging pipeline (large arrays), pad zeroed
template <class T>
static void up(spk_ctx *c, DevBuf<T> &d, const T *h, size_t count, size_t pad)
{
    d.alloc_raw(count, pad);
    c->upload_staged(d.p, h, count * sizeof(T));
}

template <class VR, class VI, class VD>
static void upload_csr(spk_ctx *c, CsrDev &D, int32_t nrows, int32_t ncols, const VR &rowptr, const VI &colidx, const VD &val,
                       bool tiles)
{
    D.nrows = nrows;
    D.ncols = ncols;
    D.nnz = (int64_t)colidx.size();
    up(c, D.rowptr, rowptr.data(), rowptr.size(), 8);
    up(c, D.colidx, colidx.data(), colidx.size(), 16);
    up(c, D.val, val.data(), val.size(), 16);
    if (tiles) {
        std::vector<int32_t> tr;
        k::build_tiles(rowptr.data(), nrows, tr);
        D.ntiles = (int32_t)tr.size() - 1;
        D.tile_row.upload(tr.data(), tr.size(), 8);
    }
}

// Collective agreement on a set-up step (KSPSetOperators is collective, as in PETSc): every rank
// reports whether its LOCAL part succeeded; when any rank failed, ALL ranks throw -- the failing one its
// own message, the others a note naming it -- so nobody is left waiting inside the next collective.
static void agree_or_fail(spk_ctx *c, const Error *mine, const char *step)
{
    const int P = c->comm->size();
    if (P > 1) {
        std::vector<int32_t> all((size_t)P, 0);
        const int32_t ok = mine ? 0 : 1;
        c->comm->host_allgather(&ok, all.data(), sizeof ok);
        if (!mine)
            for (int r = 0; r < P; ++r)
                if (!all[(size_t)r])
                    fail(SPK_ERR_COMM, "%s: rank %d failed its local part; the collective set-up is abandoned on every rank", step, r);
    }
    if (mine) throw *mine;
}

// Row types + deviation codes over the blocked copy just built (DictDev, spk_internal.hpp): block classes, then row
// types, proposed by hashing on the device; granule and range of every class entry; codes; every value decoded and compared
// bit by bit.  Leaves Adict.ok = false (the blocked kernels stay) for matrices that do not fit.  brp: the block row
// pointers on the host.
static void build_dict(spk_ctx *c, int bs, const int32_t *brp)
{
    DictDev &D = c->Adict;
    D.ok = false;
    D.tid.release(); D.tab.release(); D.cls.release(); D.fld.release(); D.codes.release(); D.zpad.release();
    const char *fmt = getenv("SPK_SPMV_FORMAT");
    if (fmt && (!strcmp(fmt, "csr") || !strcmp(fmt, "bcsr"))) return;
    static const bool verbose = getenv("SPK_DICT_VERBOSE") != nullptr;
    auto refuse = [&](const char *why, long a = 0, long b = 0) {
        D.tid.release(); D.tab.release(); D.cls.release(); D.fld.release(); D.codes.release(); D.zpad.release();
        if (verbose) fprintf(stderr, "[spk] row types + codes refused: %s (%ld, %ld)\n", why, a, b);
    };
    hipStream_t s = c->stream;
    const int32_t nbr = bs == 2 ? c->Ab.nbrows : c->Ab3.nbrows;
    const int64_t nb = bs == 2 ? c->Ab.nblocks : c->Ab3.nblocks;
    const int32_t *browptr = bs == 2 ? c->Ab.browptr.p : c->Ab3.browptr.p, *bcol = bs == 2 ? c->Ab.bcol.p : c->Ab3.bcol.p;
    const double *v0 = bs == 2 ? c->Ab.vtop.p : c->Ab3.v.p, *v1 = bs == 2 ? c->Ab.vbot.p : nullptr;
    const int64_t ldp = bs == 2 ? 0 : c->Ab3.ldp;
    const int bb = bs * bs;
    if (nbr == 0 || nb == 0 || nb > INT32_MAX) return refuse("empty or too many blocks", nbr, (long)nb);
    DevBuf<unsigned long long> keys, dmax;
    DevBuf<int32_t> rep, slot, ctl, slot2id_d, rep_d, bad, gexp;
    keys.alloc_raw(k::kDictSlots);
    rep.alloc_raw(k::kDictSlots);
    ctl.alloc_raw(4);
    bad.alloc(4);
    slot2id_d.alloc_raw(k::kDictSlots);
    rep_d.alloc_raw(std::max(k::kDictMaxPat, k::kDictMaxBlk));
    std::vector<unsigned long long> hk(k::kDictSlots);
    std::vector<int32_t> hr(k::kDictSlots), s2i(k::kDictSlots), reps;
    int32_t hctl[4];
    // one round of "hash, read the table back, number the classes by their first member": returns the class count or -1
    auto classes = [&]() -> int {
        SPK_HIP(hipMemcpyAsync(hctl, ctl.p, sizeof hctl, hipMemcpyDeviceToHost, s));
        SPK_HIP(hipMemcpyAsync(hk.data(), keys.p, sizeof(unsigned long long) * k::kDictSlots, hipMemcpyDeviceToHost, s));
        SPK_HIP(hipMemcpyAsync(hr.data(), rep.p, sizeof(int32_t) * k::kDictSlots, hipMemcpyDeviceToHost, s));
        SPK_HIP(hipStreamSynchronize(s));
        if (hctl[1]) return -1;
        std::vector<std::pair<int32_t, int32_t>> used;   // (first member, slot)
        for (int i = 0; i < k::kDictSlots; ++i)
            if (hk[(size_t)i]) used.push_back({hr[(size_t)i], i});
        std::sort(used.begin(), used.end());
        std::fill(s2i.begin(), s2i.end(), -1);
        reps.clear();
        for (size_t i = 0; i < used.size(); ++i) {
            s2i[(size_t)used[i].second] = (int32_t)i;
            reps.push_back(used[i].first);
        }
        SPK_HIP(hipMemcpyAsync(slot2id_d.p, s2i.data(), sizeof(int32_t) * k::kDictSlots, hipMemcpyHostToDevice, s));
        SPK_HIP(hipMemcpyAsync(rep_d.p, reps.data(), sizeof(int32_t) * reps.size(), hipMemcpyHostToDevice, s));
        SPK_HIP(hipStreamSynchronize(s));
        return (int)reps.size();
    };
    auto reset = [&]() {
        SPK_HIP(hipMemsetAsync(keys.p, 0, sizeof(unsigned long long) * k::kDictSlots, s));
        SPK_HIP(hipMemsetAsync(rep.p, 0x7f, sizeof(int32_t) * k::kDictSlots, s));
        SPK_HIP(hipMemsetAsync(ctl.p, 0, sizeof(int32_t) * 4, s));
    };
    // ---- block classes: blocks equal up to ~1e-6 absolute; base = the first member; granule and range per entry
    slot.alloc_raw((size_t)nb, 8);
    reset();
    k::dict_hash_blocks(bs, v0, v1, ldp, nb, keys.p, rep.p, slot.p, ctl.p, k::kDictMaxBlk, s);
    const int ncls = classes();
    if (ncls <= 0) return refuse("block classes beyond the table", hctl[0], hctl[1]);
    D.cls.alloc_raw((size_t)(ncls + 1) * bb * 2, 8);
    gexp.alloc_raw((size_t)ncls * bb, 8);
    dmax.alloc((size_t)ncls * bb, 8);
    SPK_HIP(hipMemsetAsync(gexp.p, 0x7f, sizeof(int32_t) * (size_t)ncls * bb, s));
    k::dict_class_stats(bs, v0, v1, ldp, nb, rep_d.p, ncls, slot2id_d.p, slot.p, D.cls.p, gexp.p, dmax.p, bad.p, s);   // slot[q] := class
    std::vector<int32_t> hg((size_t)ncls * bb);
    std::vector<unsigned long long> hm((size_t)ncls * bb);
    std::vector<double> hcls((size_t)(ncls + 1) * bb * 2, 0.0);   // (the null class behind the found ones: zeros)
    int32_t hbad = 1;
    SPK_HIP(hipMemcpyAsync(hg.data(), gexp.p, sizeof(int32_t) * hg.size(), hipMemcpyDeviceToHost, s));
    SPK_HIP(hipMemcpyAsync(hm.data(), dmax.p, sizeof(unsigned long long) * hm.size(), hipMemcpyDeviceToHost, s));
    SPK_HIP(hipMemcpyAsync(hcls.data(), D.cls.p, sizeof(double) * (size_t)ncls * bb * 2, hipMemcpyDeviceToHost, s));
    SPK_HIP(hipMemcpyAsync(&hbad, bad.p, sizeof hbad, hipMemcpyDeviceToHost, s));
    SPK_HIP(hipStreamSynchronize(s));
    if (hbad) return refuse("a deviation from its class base is not exactly representable");
    // bit fields: entry (class, e) gets the width its largest deviation needs (two's complement), the fields of a block are
    // packed into one 64-bit word (2x2 blocks) or two (3x3: a field never straddles the words)
    std::vector<int32_t> hfld((size_t)(ncls + 1) * bb, 1 << 8);   // (null class: one bit at offset 0)
    std::vector<int> hwid((size_t)ncls * bb, 1);
    std::vector<double> hscale((size_t)ncls * bb, 1.0);
    for (int i = 0; i < ncls * bb; ++i) {
        if (hg[(size_t)i] < 0x7f000000) {   // some member deviates: granule = the finest bit in use
            if (hg[(size_t)i] < -1000 || hg[(size_t)i] > 1000) return refuse("deviation granule out of range", i, hg[(size_t)i]);
            const double scale = std::ldexp(1.0, hg[(size_t)i]);
            double mag;
            std::memcpy(&mag, &hm[(size_t)i], sizeof mag);
            const double kabs = mag / scale;
            if (!(kabs <= 1.0e9)) return refuse("a class entry scatters beyond 31-bit codes", i, (long)hg[(size_t)i]);
            int width = 2;
            while ((double)((1ll << (width - 1)) - 1) < kabs) ++width;
            hwid[(size_t)i] = width;
            hscale[(size_t)i] = scale;
        }
    }
    // 2x2: one layout for all classes where the widest need per entry allows it (1024^2: 19 + 13 | 13 + 19 bits)
    D.uniform = false;
    if (bs == 2 && !getenv("SPK_DICT_NOUNIFORM")) {
        int uw[4] = {1, 1, 1, 1};
        for (int cl = 0; cl < ncls; ++cl)
            for (int e = 0; e < 4; ++e) uw[e] = std::max(uw[e], hwid[(size_t)cl * 4 + e]);
        if (uw[0] + uw[1] <= 32 && uw[2] + uw[3] <= 32) {
            D.uniform = true;
            for (int e = 0; e < 4; ++e) D.uw[e] = uw[e];
            const uint32_t f[4] = {0u | ((uint32_t)uw[0] << 8), (uint32_t)(32 - uw[1]) | ((uint32_t)uw[1] << 8),
                                   0x80000000u | ((uint32_t)uw[2] << 8), 0x80000000u | (uint32_t)(32 - uw[3]) | ((uint32_t)uw[3] << 8)};
            for (int cl = 0; cl <= ncls; ++cl)   // (the null class too: any field of a zero word decodes to 0)
                for (int e = 0; e < 4; ++e) hfld[(size_t)cl * 4 + e] = (int32_t)f[e];
        }
    }
    // 3x3: the same idea over the two words of a block; the entries of the second word are one of three fixed sets
    D.uniform3 = 0;
    if (bs == 3 && !getenv("SPK_DICT_NOUNIFORM")) {
        int uw[9];
        for (int e = 0; e < 9; ++e) {
            uw[e] = 1;
            for (int cl = 0; cl < ncls; ++cl) uw[e] = std::max(uw[e], hwid[(size_t)cl * 9 + e]);
        }
        for (int split = 1; split <= 3 && !D.uniform3; ++split) {
            auto in_w1 = [&](int e) { return split == 1 ? e >= 5 : split == 2 ? e >= 4 : (e == 4 || e >= 6); };
            int used[2] = {0, 0};
            for (int e = 0; e < 9; ++e) used[in_w1(e) ? 1 : 0] += uw[e];
            if (used[0] > 64 || used[1] > 64) continue;
            D.uniform3 = split;
            int sh[2] = {0, 0};
            for (int e = 0; e < 9; ++e) {
                const int wd = in_w1(e) ? 1 : 0;
                for (int cl = 0; cl <= ncls; ++cl) hfld[(size_t)cl * 9 + e] = sh[wd] | (uw[e] << 8) | (wd << 16);
                D.u3l[e] = 64 - sh[wd] - uw[e];
                D.u3r[e] = 32 - uw[e];
                sh[wd] += uw[e];
            }
        }
    }
    D.straddle = false;
    for (int cl = 0; cl < ncls; ++cl) {
        int used[2] = {0, 0}, word = 0;
        if (bs == 2 && !D.uniform) {
            // no packing of this class's fields inside the halves, but 64 bits suffice: back to back, a field across
            // the halves flagged (the plain kernels extract it with 64-bit shifts)
            const int *w = &hwid[(size_t)cl * 4];
            const bool fits_halves = w[0] + w[1] <= 32 && w[2] + w[3] <= 32;
            if (!fits_halves && w[0] + w[1] + w[2] + w[3] <= 64) {
                int sh = 0;
                for (int e = 0; e < 4; ++e) {
                    const int i = cl * 4 + e;
                    const bool across = sh < 32 && sh + w[e] > 32;
                    hfld[(size_t)i] = across ? (int32_t)((uint32_t)sh | ((uint32_t)w[e] << 8) | (uint32_t)k::kDictAcrossHost)
                                             : (int32_t)((uint32_t)(sh & 31) | ((uint32_t)w[e] << 8) | (sh >= 32 ? 0x80000000u : 0u));
                    D.straddle = D.straddle || across;
                    sh += w[e];
                    hcls[(size_t)2 * i + 1] = hscale[(size_t)i];
                }
                continue;
            }
        }
        for (int e = 0; e < bb; ++e) {
            const int i = cl * bb + e;
            const double scale = hscale[(size_t)i];
            const int width = hwid[(size_t)i];
            if (D.uniform || D.uniform3) {
                hcls[(size_t)2 * i + 1] = scale;
                continue;
            }
            // 2x2: two 32-bit halves of one word, a field inside one half (hardware bit-field extract); 3x3: two 64-bit words
            const int cap = bs == 2 ? 32 : 64;
            if (used[word] + width > cap) ++word;
            if (word > 1) return refuse("the codes of a block class do not fit its word(s)", cl, used[0] + used[1] + width);
            hfld[(size_t)i] = bs == 2 ? (int32_t)((uint32_t)used[word] | ((uint32_t)width << 8) | (word ? 0x80000000u : 0u))
                                      : (used[word] | (width << 8) | (word << 16));
            used[word] += width;
            hcls[(size_t)2 * i + 1] = scale;
        }
    }
    SPK_HIP(hipMemcpyAsync(D.cls.p, hcls.data(), sizeof(double) * hcls.size(), hipMemcpyHostToDevice, s));
    D.fld.alloc_raw((size_t)(ncls + 1) * bb, 8);
    D.zpad.alloc(8, 8);
    SPK_HIP(hipMemcpyAsync(D.fld.p, hfld.data(), sizeof(int32_t) * hfld.size(), hipMemcpyHostToDevice, s));
    // ---- row types
    DevBuf<int32_t> rslot;
    rslot.alloc_raw((size_t)nbr, 8);
    reset();
    k::dict_hash_rows(browptr, bcol, slot.p, nbr, keys.p, rep.p, rslot.p, ctl.p, k::kDictMaxPat, kDictMaxK, s);
    const int ntype = classes();
    if (ntype <= 0) return refuse("row types beyond the table, or a row beyond kDictMaxK blocks", hctl[0], hctl[1]);
    int kmax = 1;
    for (int32_t r : reps) kmax = std::max(kmax, brp[(size_t)r + 1] - brp[(size_t)r]);
    const int tab_ints = ((ntype + 1) & ~1) + 2 * ntype * kmax;
    const int lds_bytes = ((((4 * tab_ints + 15) & ~15) + 16 * (ncls + 1) * bb + 4 * (ncls + 1) * bb) + 15) & ~15;
    if (lds_bytes > k::kDictLdsMax) return refuse("tables beyond the LDS budget", lds_bytes, ntype);
    D.tab.alloc((size_t)tab_ints, 8);
    D.tid.alloc_raw((size_t)nbr, 64);
    k::dict_fill_rows(browptr, bcol, slot.p, nbr, rep_d.p, ntype, kmax, slot2id_d.p, rslot.p, D.tab.p, D.tid.p, bad.p, s);
    std::vector<int32_t> htab((size_t)tab_ints);
    SPK_HIP(hipMemcpyAsync(htab.data(), D.tab.p, sizeof(int32_t) * htab.size(), hipMemcpyDeviceToHost, s));
    SPK_HIP(hipMemcpyAsync(&hbad, bad.p, sizeof hbad, hipMemcpyDeviceToHost, s));
    SPK_HIP(hipStreamSynchronize(s));
    if (hbad) return refuse("row type verification failed (hash collision)", ntype, ncls);
    // ---- code planes (DictArgs::plane_off): 2x2 blocks -- positions 2p, 2p+1 side by side in plane p; 3x3 -- plane k
    const int64_t nbr_pad = ((int64_t)nbr + 15) & ~(int64_t)15;
    int64_t off = 0;
    // (the planes are read side by side, row r of each at the same time: a skew of 17 x 256 B per plane keeps planes whose
    // size is a power of two -- 16 MiB each at 1024^2 -- from landing on one memory channel together)
    static const int64_t skew = [] { const char *e = getenv("SPK_DICT_SKEW"); return e ? (int64_t)atoll(e) : (int64_t)(17 * 256); }();
    for (int kk = 0; kk < kDictMaxK; ++kk) {
        off += kk ? skew : 0;
        D.plane_off[kk] = off;
        if (bs == 2) {
            if (2 * kk + 1 < kmax) off += 16 * nbr_pad;
            else if (2 * kk < kmax) off += 8 * nbr_pad;
        } else if (kk < kmax) {
            off += 16 * nbr_pad;
        }
    }
    D.codes.alloc((size_t)off, 64);
    D.bs = bs;
    D.nbrows = nbr;
    D.nblocks = nb;
    D.ntype = ntype;
    D.nclass = ncls;
    D.kmax = kmax;
    D.lds_bytes = lds_bytes;
    D.code_bytes = (int64_t)(bs == 2 ? 8 : 16) * nb;   // bytes of codes one product reads: every stored block once
    k::dict_encode_verify(D, browptr, bcol, slot.p, v0, v1, ldp, bad.p, s);
    SPK_HIP(hipMemcpyAsync(&hbad, bad.p, sizeof hbad, hipMemcpyDeviceToHost, s));
    SPK_HIP(hipStreamSynchronize(s));
    if (hbad) return refuse("a decoded value differs from the stored one", ntype, ncls);
    D.ok = true;
    if (verbose) {
        int wmax = 0;
        for (int cl = 0; cl < ncls; ++cl) {
            int tot = 0;
            for (int e = 0; e < bb; ++e) tot += (hfld[(size_t)cl * bb + e] >> 8) & 255;
            wmax = std::max(wmax, tot);
        }
        fprintf(stderr, "[spk] row types + codes: %d block rows, %d types (<= %d blocks), %d classes of %d x %d (<= %d bits of codes per block), "
                        "%d B of LDS, %.1f B of codes per block row%s\n", nbr, ntype, kmax, ncls, bs, bs, wmax, lds_bytes, (double)D.code_bytes / nbr,
                D.uniform || D.uniform3 ? ", one field layout for all classes" : "");
        fprintf(stderr, "[spk]   widest need per block entry:");
        for (int e = 0; e < bb; ++e) {
            int w = 1;
            for (int cl = 0; cl < ncls; ++cl) w = std::max(w, hwid[(size_t)cl * bb + e]);
            fprintf(stderr, " %d", w);
        }
        fprintf(stderr, "\n");
    }
}

namespace k {
LaunchTimer &launch_timer()
{
    static thread_local LaunchTimer t;
    return t;
}
}  // namespace k

void a_mult(spk_ctx *c, const double *x, double *y, const CsrDev *bt, const double *lam, const int32_t *done, bool accumulate,
            const k::OffDiag *od, const k::GivensRider *rider)
{
    hipStream_t s = c->stream;
    // (bench.py's roofline: the iteration's product launches timed where they run, inside a solve)
    // the kernel's own start / stop time stamps go into a pair of events (SPK_LAUNCH_PRODUCT)
    const bool timed = c->time_products && rider && c->tp_used + 2 <= c->tp_ev.size();
    struct Timed {
        spk_ctx *c; bool on;
        Timed(spk_ctx *c_, bool on_) : c(c_), on(on_) { if (on) k::launch_timer() = k::LaunchTimer{c->tp_ev[c->tp_used], c->tp_ev[c->tp_used + 1]}; }
        ~Timed() { if (on) { k::launch_timer() = k::LaunchTimer{}; c->tp_used += 2; } }
    } timer(c, timed);
    if (c->spmv_format != 0 && c->Adict.ok) k::spmv_dict(c->Adict, x, y, bt, lam, done, s, accumulate, od, rider);
    else if (c->spmv_format == 2) k::spmv_bcsr3(c->Ab3, x, y, bt, lam, done, s, accumulate, od, rider);
    else if (c->spmv_format == 1) k::spmv_bcsr(c->Ab, x, y, bt, lam, done, s, accumulate, od, rider);
    else k::spmv(c->Ad, x, y, bt, lam, done, s, accumulate, od, rider);
}

static void set_block_A(spk_ctx *c, int64_t row_begin, int32_t nrows_local, int64_t ncols_global,
                        const int32_t *rowptr, const int32_t *colidx, const double *val)
{
    std::vector<int32_t> garray;   // sorted global numbers of the off-rank columns (MatMPIAIJ's garray)
    Error local{0, ""};
    try {  // ---- local part: validation, upload, split and blocking on the device (no collective inside)
    if (rowptr[0] != 0) fail(SPK_ERR_ARG, "A00: rowptr[0] must be 0");
    if (row_begin < 0 || row_begin + nrows_local > ncols_global)
        fail(SPK_ERR_ARG, "A00: rows [%lld,%lld) outside the %lld x %lld block", (long long)row_begin,
             (long long)(row_begin + nrows_local), (long long)ncols_global, (long long)ncols_global);
    for (int32_t r = 0; r < nrows_local; ++r)
        if (rowptr[r + 1] < rowptr[r]) fail(SPK_ERR_ARG, "A00: rowptr not monotone at row %d", r);

    hipStream_t s = c->stream;
    const int32_t n = nrows_local;
    const int64_t nnz = rowptr[n];
    const int64_t lo = row_begin, hi = row_begin + n;
    // the caller's slab as it is, once
    DevBuf<int32_t> rp_in, ci_in, cnt, orp, scratch, flags;
    DevBuf<double> va_in;
    rp_in.alloc_raw((size_t)n + 1, 8);
    ci_in.alloc_raw((size_t)nnz, 16);
    va_in.alloc_raw((size_t)nnz, 16);
    c->upload_staged(rp_in.p, rowptr, sizeof(int32_t) * ((size_t)n + 1));
    c->upload_staged(ci_in.p, colidx, sizeof(int32_t) * (size_t)nnz);
    c->upload_staged(va_in.p, val, sizeof(double) * (size_t)nnz);
    // off-rank entries per row (and the column range check), exclusive scan
    cnt.alloc_raw((size_t)n, 8);
    orp.alloc_raw((size_t)n + 1, 8);
    scratch.alloc_raw((size_t)n / 2048 + 8);
    flags.alloc(4);
    k::csr_count_off(rp_in.p, ci_in.p, n, lo, hi, ncols_global, cnt.p, flags.p, s);
    k::exclusive_scan_i32(cnt.p, n, orp.p, scratch.p, s);
    int32_t hflags[4] = {0, 0, 0, 0}, noff = 0;
    SPK_HIP(hipMemcpyAsync(hflags, flags.p, sizeof hflags, hipMemcpyDeviceToHost, s));
    SPK_HIP(hipMemcpyAsync(&noff, orp.p + n, sizeof noff, hipMemcpyDeviceToHost, s));
    SPK_HIP(hipStreamSynchronize(s));
    if (hflags[0]) fail(SPK_ERR_ARG, "A00: column %d out of range [0,%lld)", hflags[1], (long long)ncols_global);
    // validation passed: from here on the previous operator is being replaced (a refused block, above,
    // leaves it in place and usable)
    c->have_A = false;
    c->pc_ready = false;
    if (c->n_global != ncols_global || c->row_begin != row_begin || c->n_local != nrows_local) {
        // another row range: a constraint block set before belongs to the old one (its column slice and every
        // array sized by it); KSPSetOperators has to bring the new A10 as well
        c->have_B = false;
        c->m = 0;
        c->b_general = false;
        c->m_wide = 0;
        c->bd.release();
        c->bdpk.release();
    }
    c->n_global = ncols_global;
    c->row_begin = row_begin;
    c->n_local = nrows_local;
    const int64_t nnzd = nnz - noff;
    CsrDev &Ad = c->Ad;
    Ad.nrows = n;
    Ad.ncols = n;
    Ad.nnz = nnzd;
    Ad.rowptr.alloc_raw((size_t)n + 1, 8);
    Ad.colidx.alloc_raw((size_t)nnzd, 16);
    Ad.val.alloc_raw((size_t)nnzd, 16);
    c->Ao.nrows = 0;
    c->Ao.ncols = 0;
    c->Ao.nnz = noff;
    c->Ao.colidx.alloc_raw((size_t)noff, 16);
    c->Ao.val.alloc_raw((size_t)noff, 16);
    k::csr_split(rp_in.p, ci_in.p, va_in.p, n, lo, hi, orp.p, Ad.rowptr.p, Ad.colidx.p, Ad.val.p, c->Ao.colidx.p, c->Ao.val.p, s);
    // what the host still needs: the diagonal block's row pointers (tile tables are a greedy scan of them) and the
    // few off-rank entries (ghost numbering, halo plan)
    HostBuf<int32_t> drp;
    drp.alloc((size_t)n + 1);
    std::vector<int32_t> orp_h, ocol_h((size_t)noff);
    if (noff > 0) {
        orp_h.resize((size_t)n + 1);
        SPK_HIP(hipMemcpyAsync(orp_h.data(), orp.p, sizeof(int32_t) * ((size_t)n + 1), hipMemcpyDeviceToHost, s));
        SPK_HIP(hipMemcpyAsync(ocol_h.data(), c->Ao.colidx.p, sizeof(int32_t) * (size_t)noff, hipMemcpyDeviceToHost, s));
        SPK_HIP(hipStreamSynchronize(s));
        parallel_for((int64_t)n + 1, [&](int64_t r0, int64_t r1, int) {
            for (int64_t r = r0; r < r1; ++r) drp[(size_t)r] = rowptr[r] - orp_h[(size_t)r];
        });
    } else {
        parallel_for((int64_t)n + 1, [&](int64_t r0, int64_t r1, int) {
            for (int64_t r = r0; r < r1; ++r) drp[(size_t)r] = rowptr[r];
        });
    }
    {
        std::vector<int32_t> tr;
        k::build_tiles(drp.data(), n, tr);
        Ad.ntiles = (int32_t)tr.size() - 1;
        Ad.tile_row.upload(tr.data(), tr.size(), 8);
    }
    // ghost numbering: sorted unique global columns, off-rank column indices rewritten to ghost numbers
    garray.assign(ocol_h.begin(), ocol_h.end());
    std::sort(garray.begin(), garray.end());
    garray.erase(std::unique(garray.begin(), garray.end()), garray.end());
    c->n_ghost = (int32_t)garray.size();
    if (noff > 0) {
        for (auto &g : ocol_h) g = (int32_t)(std::lower_bound(garray.begin(), garray.end(), g) - garray.begin());
        SPK_HIP(hipMemcpyAsync(c->Ao.colidx.p, ocol_h.data(), sizeof(int32_t) * (size_t)noff, hipMemcpyHostToDevice, s));
        SPK_HIP(hipStreamSynchronize(s));
    }
    // off-rank block: compressed to the rows that have entries (FP32 sweeps), and over all rows (SpMV epilogue)
    {
        std::vector<int32_t> rows, corp(1, 0);
        if (noff > 0)
            for (int32_t r = 0; r < n; ++r)
                if (orp_h[(size_t)r + 1] > orp_h[(size_t)r]) {
                    rows.push_back(r);
                    corp.push_back(orp_h[(size_t)r + 1]);
                }
        c->Ao.nrows = (int32_t)rows.size();
        c->Ao.ncols = c->n_ghost;
        c->Ao.rowptr.upload(corp.data(), corp.size(), 8);
        c->ao_rows.upload(rows.data(), rows.size(), 8);
        c->ao_rowptr_full.release();
        if (c->n_ghost > 0) std::swap(c->ao_rowptr_full.p, orp.p), std::swap(c->ao_rowptr_full.n, orp.n);
    }

    // 2x2-blocked copy when every row pair shares its pattern and columns pair up (dof-2 grids): verified
    // and filled by one kernel, block row br starting at block rowptr[2 br] / 4
    {
        BcsrDev &Ab = c->Ab;
        c->Adict.ok = false;
        Ab.ok = false;
        Ab.nbrows = 0;
        Ab.ntiles = 0;
        Ab.long_rows = false;
        if (n % 2 == 0 && n > 0 && nnzd % 4 == 0) {
            const int32_t nbr = n / 2;
            Ab.browptr.alloc_raw((size_t)nbr + 1, 8);
            Ab.bcol.alloc_raw((size_t)(nnzd / 4), 16);
            Ab.vtop.alloc_raw((size_t)(nnzd / 2), 32);
            Ab.vbot.alloc_raw((size_t)(nnzd / 2), 32);
            SPK_HIP(hipMemsetAsync(flags.p, 0, sizeof(int32_t) * 4, s));
            k::bcsr_fill(Ad.rowptr.p, Ad.colidx.p, Ad.val.p, nbr, Ab.browptr.p, Ab.bcol.p, Ab.vtop.p, Ab.vbot.p, flags.p, s);
            SPK_HIP(hipMemcpyAsync(hflags, flags.p, sizeof hflags, hipMemcpyDeviceToHost, s));
            SPK_HIP(hipStreamSynchronize(s));
            if (!hflags[0]) {
                HostBuf<int32_t> brp;
                brp.alloc((size_t)nbr + 1);
                parallel_for((int64_t)nbr + 1, [&](int64_t b0, int64_t b1, int) {
                    for (int64_t br = b0; br < b1; ++br) brp[(size_t)br] = drp[(size_t)(2 * br)] / 4;
                });
                Ab.nbrows = nbr;
                Ab.nblocks = nnzd / 4;
                std::vector<int32_t> tb;
                k::build_btiles(brp.data(), Ab.nbrows, tb);
                for (int32_t br = 0; br < Ab.nbrows && !Ab.long_rows; ++br) Ab.long_rows = brp[(size_t)br + 1] - brp[(size_t)br] > k::kBTile;
                Ab.ntiles = (int32_t)tb.size() - 1;
                Ab.tile_brow.upload(tb.data(), tb.size(), 8);
                {
                    std::vector<int32_t> td((size_t)4 * Ab.ntiles);
                    for (int32_t t = 0; t < Ab.ntiles; ++t) {
                        td[(size_t)4 * t] = tb[(size_t)t];
                        td[(size_t)4 * t + 1] = tb[(size_t)t + 1];
                        td[(size_t)4 * t + 2] = brp[(size_t)tb[(size_t)t]];
                        td[(size_t)4 * t + 3] = brp[(size_t)tb[(size_t)t + 1]];
                    }
                    Ab.tile_desc.upload(td.data(), td.size(), 8);
                }
                Ab.ok = true;
                build_dict(c, 2, brp.data());
                // BA iteration kernel: workgroup rho (row order) owns ba_tb consecutive tiles of its XCD's range; it
                // may start its SpMV phase once the owners of the rows its block columns touch have stored their z~
                Ab.ba_ok = false;
                if (!Ab.long_rows && Ab.ntiles > 0) {
                    const int TB = 4, tpx = (Ab.ntiles + 7) / 8;
                    DevBuf<int32_t> rng;
                    rng.alloc_raw((size_t)2 * Ab.ntiles, 8);
                    k::tile_col_range(Ab.browptr.p, Ab.bcol.p, Ab.tile_brow.p, Ab.ntiles, rng.p, s);
                    std::vector<int32_t> rh((size_t)2 * Ab.ntiles);
                    SPK_HIP(hipMemcpyAsync(rh.data(), rng.p, rh.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
                    SPK_HIP(hipStreamSynchronize(s));
                    // groups of <= TB consecutive tiles with <= 256 block rows (one double2 per thread in phase B),
                    // XCD by XCD (each XCD keeps its contiguous run of tiles, as in the SpMV kernels)
                    std::vector<std::vector<std::pair<int32_t, int32_t>>> groups(8);
                    bool fits = true;
                    int S = 1;
                    for (int x = 0; x < 8; ++x) {
                        const int tbeg = std::min(x * tpx, (int)Ab.ntiles), tend = std::min((x + 1) * tpx, (int)Ab.ntiles);
                        int t = tbeg;
                        while (t < tend) {
                            int t1 = t;
                            while (t1 < tend && t1 - t < TB && tb[(size_t)t1 + 1] - tb[(size_t)t] <= 256) ++t1;
                            if (t1 == t) { fits = false; break; }   // one tile beyond 256 block rows
                            groups[(size_t)x].push_back({t, t1});
                            t = t1;
                        }
                        S = std::max(S, (int)groups[(size_t)x].size());
                    }
                    const int nwg = 8 * S;
                    if (nwg > 1024) fits = false;
                    // phase B deals the double2 entries (= block rows) out evenly, in the same row order: workgroup rho owns
                    // [rho chunk, (rho + 1) chunk) -- no table in front of its loads; one entry per thread
                    const int chunk = (Ab.nbrows + nwg - 1) / nwg;
                    if (chunk > 256 || chunk < 1) fits = false;
                    // per workgroup rho = x S + k: {t0, t1, first owner, last owner it waits for}: the owners (phase B) of the
                    // block rows its columns touch and of its own rows (their c~)
                    std::vector<int32_t> wt((size_t)4 * nwg, 0);
                    for (int rho = 0; rho < nwg && fits; ++rho) {
                        const int x = rho / S, kk = rho % S;
                        const auto &gx = groups[(size_t)x];
                        int32_t *w4 = wt.data() + (size_t)4 * rho;
                        if (kk < (int)gx.size()) {
                            const int t0 = gx[(size_t)kk].first, t1 = gx[(size_t)kk].second;
                            int32_t lo = tb[(size_t)t0], hi = tb[(size_t)t1] - 1;
                            for (int t = t0; t < t1; ++t) {
                                lo = std::min(lo, rh[(size_t)2 * t]);
                                hi = std::max(hi, rh[(size_t)2 * t + 1]);
                            }
                            w4[0] = t0;
                            w4[1] = t1;
                            w4[2] = std::max(0, lo / chunk);
                            w4[3] = std::min(nwg - 1, hi / chunk);
                        } else {  // no tiles: waits for nobody
                            w4[0] = w4[1] = 0;
                            w4[2] = 0;
                            w4[3] = -1;
                        }
                    }
                    if (fits) {
                        Ab.ba_slots = S;
                        Ab.ba_tb = TB;
                        Ab.ba_chunk = chunk;
                        Ab.ba_wt.upload(wt.data(), wt.size(), 8);
                        Ab.ba_ok = true;
                    }
                }
            } else {
                Ab.browptr.release(); Ab.bcol.release(); Ab.vtop.release(); Ab.vbot.release();
            }
        }
        // 3x3-blocked copy for dof-3 grids (the 3-D generator: 81 entries per row in 27 blocks), same verification
        Bcsr3Dev &A3 = c->Ab3;
        A3.ok = false;
        A3.nbrows = 0;
        A3.ntiles = 0;
        A3.v32.release();
        if (!Ab.ok && n % 3 == 0 && n > 0 && nnzd % 9 == 0) {
            const int32_t nbr = n / 3;
            const int64_t nb = nnzd / 9;
            A3.ldp = (nb + 1 + 7) & ~(int64_t)7;
            A3.browptr.alloc_raw((size_t)nbr + 1, 8);
            A3.bcol.alloc_raw((size_t)nb, 16);
            A3.v.alloc_raw((size_t)(9 * A3.ldp), 32);
            SPK_HIP(hipMemsetAsync(flags.p, 0, sizeof(int32_t) * 4, s));
            k::bcsr3_fill(Ad.rowptr.p, Ad.colidx.p, Ad.val.p, nbr, A3.browptr.p, A3.bcol.p, A3.v.p, A3.ldp, flags.p, s);
            SPK_HIP(hipMemcpyAsync(hflags, flags.p, sizeof hflags, hipMemcpyDeviceToHost, s));
            SPK_HIP(hipStreamSynchronize(s));
            if (!hflags[0]) {
                HostBuf<int32_t> brp;
                brp.alloc((size_t)nbr + 1);
                parallel_for((int64_t)nbr + 1, [&](int64_t b0, int64_t b1, int) {
                    for (int64_t br = b0; br < b1; ++br) brp[(size_t)br] = drp[(size_t)(3 * br)] / 9;
                });
                std::vector<int32_t> tb;
                k::build_b3tiles(brp.data(), nbr, tb);
                A3.nbrows = nbr;
                A3.nblocks = nb;
                A3.ntiles = (int32_t)tb.size() - 1;
                A3.tile_brow.upload(tb.data(), tb.size(), 8);
                A3.ok = true;
                build_dict(c, 3, brp.data());
            } else {
                A3.browptr.release(); A3.bcol.release(); A3.v.release();
            }
        } else {
            A3.browptr.release(); A3.bcol.release(); A3.v.release();
        }
        const char *fmt = getenv("SPK_SPMV_FORMAT");
        const bool csr_forced = fmt && !strcmp(fmt, "csr");
        c->spmv_format = csr_forced ? 0 : (Ab.ok ? 1 : (A3.ok ? 2 : 0));
    }
    } catch (const Error &e) {
        local = e;
    } catch (const std::exception &e) {
        local = Error{SPK_ERR_NOMEM, e.what()};
    }
    agree_or_fail(c, local.code ? &local : nullptr, "A00");

    // ---- halo plan (VecScatter of MatMult_MPIAIJ) ----
    const int P = c->comm->size(), me = c->comm->rank();
    c->peers.clear();
    c->send_off.assign(1, 0);
    c->recv_off.assign(1, 0);
    std::vector<int32_t> send_idx;
    if (P > 1) {
        std::vector<int64_t> mine = {row_begin, row_begin + nrows_local}, all((size_t)2 * P);
        c->comm->host_allgather(mine.data(), all.data(), 2 * sizeof(int64_t));
        for (int r = 1; r < P; ++r)
            if (all[2 * r] != all[2 * r - 1]) fail(SPK_ERR_ARG, "A00: row slabs must tile [0,n) in rank order");
        std::vector<std::vector<char>> ghosts;
        c->comm->host_allgatherv(garray.data(), garray.size() * sizeof(int32_t), ghosts);
        for (int p = 0; p < P; ++p) {
            if (p == me) continue;
            // what I receive from p: my ghosts inside p's range (contiguous in sorted garray)
            const int64_t plo = all[2 * p], phi = all[2 * p + 1];
            int64_t nrecv = 0;
            for (int32_t g : garray) nrecv += (g >= plo && g < phi);
            // what I send to p: p's ghosts inside my range, in p's order
            const int32_t *pg = (const int32_t *)ghosts[(size_t)p].data();
            const size_t npg = ghosts[(size_t)p].size() / sizeof(int32_t);
            int64_t nsend = 0;
            for (size_t i = 0; i < npg; ++i)
                if (pg[i] >= row_begin && pg[i] < row_begin + nrows_local) {
                    send_idx.push_back((int32_t)(pg[i] - row_begin));
                    ++nsend;
                }
            if (nsend == 0 && nrecv == 0) continue;
            c->peers.push_back(p);
            c->send_off.push_back(c->send_off.back() + nsend);
            c->recv_off.push_back(c->recv_off.back() + nrecv);
        }
    } else if (c->n_ghost != 0) {
        fail(SPK_ERR_ARG, "A00: %d columns fall outside the local rows but there is only one rank", c->n_ghost);
    }
    local = Error{0, ""};
    try {  // ---- local again: the rest of the plan and its uploads; agreed on before the collective setup_halo
    if (P > 1 && c->recv_off.back() != c->n_ghost) fail(SPK_ERR_ARG, "A00: ghost columns not owned by any rank");
    c->send_idx.upload(send_idx.data(), send_idx.size(), 8);
    c->send_buf.alloc(send_idx.size(), 8);
    // (the off-rank part in "SpMV epilogue" form -- row pointers over all local rows -- is the scan result kept above)
    // halo rows as contiguous ranges (slab partitions): lets the producer of z fill send_buf itself
    c->send_ranges = k::SendRanges{};
    {
        bool ok = !c->peers.empty() && c->peers.size() <= 4;
        for (size_t p = 0; ok && p < c->peers.size(); ++p) {
            const int64_t a = c->send_off[p], b = c->send_off[p + 1];
            for (int64_t i = a + 1; ok && i < b; ++i) ok = send_idx[(size_t)i] == send_idx[(size_t)i - 1] + 1;
            if (ok) {
                c->send_ranges.r0[p] = b > a ? send_idx[(size_t)a] : 0;
                c->send_ranges.len[p] = (int32_t)(b - a);
                c->send_ranges.off[p] = (int32_t)a;
            }
        }
        if (ok) {
            c->send_ranges.n = (int)c->peers.size();
            c->send_ranges.buf = c->send_buf.p;
        }
    }
    c->xghost.alloc((size_t)c->n_ghost, 8);
    } catch (const Error &e) {
        local = e;
    } catch (const std::exception &e) {
        local = Error{SPK_ERR_NOMEM, e.what()};
    }
    agree_or_fail(c, local.code ? &local : nullptr, "A00 (halo plan)");
    c->comm->setup_halo(c->n_ghost, c->peers, c->send_off, c->recv_off);  // collective
    c->have_A = true;
    c->pc_ready = false;
    c->ensure_vectors();
}

static void set_block_B(spk_ctx *c, int32_t m, int64_t ncols_global, const int32_t *rowptr,
                        const int32_t *colidx, const double *val)
{
    if (!c->have_A) fail(SPK_ERR_STATE, "A10: set SPK_BLOCK_A00 first");
    c->bt_cached = nullptr;
    if (ncols_global != c->n_global) fail(SPK_ERR_ARG, "A10: %lld columns, A00 has %lld", (long long)ncols_global, (long long)c->n_global);
    if (m < 0) fail(SPK_ERR_ARG, "A10: negative row count");
    if ((int64_t)c->n_local + m > INT32_MAX - 1024) fail(SPK_ERR_UNSUPPORTED, "A10: n_local + m exceeds 32-bit vector indices");
    for (int32_t r = 0; r < m; ++r)
        if (rowptr[r + 1] < rowptr[r]) fail(SPK_ERR_ARG, "A10: rowptr not monotone at row %d", r);
    const int32_t nl = c->n_local;
    const int64_t lo = c->row_begin, hi = lo + nl;
    // local column numbers, ascending inside each row
    // (threaded over the entries: the reference's 4 rows hold ~n/2 entries each -- 4 M at 1024^2)
    HostBuf<int32_t> col;
    HostBuf<double> v;
    col.alloc((size_t)rowptr[m]);
    v.alloc((size_t)rowptr[m]);
    {
        // a flag of its own per thread: any int32 -- -1 included -- can be the offending column number
        std::vector<char> bad(64, 0);
        std::vector<int32_t> badcol(64, 0);
        parallel_for(rowptr[m], [&](int64_t a, int64_t b, int t) {
            for (int64_t k = a; k < b; ++k) {
                const int32_t g = colidx[k];
                if (g < lo || g >= hi) bad[(size_t)t] = 1, badcol[(size_t)t] = g;
                col[(size_t)k] = (int32_t)(g - lo);
                v[(size_t)k] = val[k];
            }
        });
        for (size_t t = 0; t < bad.size(); ++t)
            if (bad[t])
                fail(SPK_ERR_ARG, "A10: column %d not owned by this rank [%lld,%lld)", badcol[t], (long long)lo, (long long)hi);
        for (int32_t r = 0; r < m; ++r) {   // PETSc rows come sorted: nothing to do then
            const int32_t k0 = rowptr[r], k1 = rowptr[r + 1];
            if (std::is_sorted(col.data() + k0, col.data() + k1)) continue;
            std::vector<int32_t> perm((size_t)(k1 - k0));
            std::iota(perm.begin(), perm.end(), 0);
            std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) { return col[(size_t)(k0 + a)] < col[(size_t)(k0 + b)]; });
            std::vector<int32_t> c2((size_t)(k1 - k0));
            std::vector<double> v2((size_t)(k1 - k0));
            for (int32_t i = 0; i < k1 - k0; ++i) {
                c2[(size_t)i] = col[(size_t)(k0 + perm[(size_t)i])];
                v2[(size_t)i] = v[(size_t)(k0 + perm[(size_t)i])];
            }
            std::copy(c2.begin(), c2.end(), col.data() + k0);
            std::copy(v2.begin(), v2.end(), v.data() + k0);
        }
    }
    // Which rows go through the column-window (long-row) kernel: all of them for m <= 8 (the reference's 4
    // rows, the fused dense-plane path); for a general block only its LONG rows (local entries beyond
    // kWideRowNnz; at most 8, the longest first) -- the rest is a CSR by rows for the stream kernel.
    constexpr int32_t kWideRowNnz = 8192;
    c->b_general = m > 8;
    std::vector<int32_t> wide;
    if (!c->b_general) {
        for (int32_t r = 0; r < m; ++r) wide.push_back(r);
    } else {
        std::vector<int32_t> cand;
        for (int32_t r = 0; r < m; ++r)
            if (rowptr[r + 1] - rowptr[r] > kWideRowNnz) cand.push_back(r);
        std::sort(cand.begin(), cand.end(), [&](int32_t a, int32_t b) {
            const int32_t la = rowptr[a + 1] - rowptr[a], lb = rowptr[b + 1] - rowptr[b];
            return la != lb ? la > lb : a < b;
        });
        if (cand.size() > 8) cand.resize(8);
        std::sort(cand.begin(), cand.end());
        wide = cand;
    }
    const int32_t mw = (int32_t)wide.size();
    c->m_wide = mw;
    c->wide_rows_h = wide;
    c->wide_rows.upload(wide.data(), wide.size(), 8);
    // column windows over the wide rows (concatenated in `wide` order)
    WideDev &B = c->B;
    B.m = mw;
    B.ncols = nl;
    std::vector<int32_t> wcol_own, wrp(1, 0);
    std::vector<double> wv_own;
    const int32_t *wcol = col.data();
    const double *wv = v.data();
    if (!c->b_general) {  // every row, in order: the arrays as they are
        for (int32_t r = 0; r < m; ++r) wrp.push_back(rowptr[r + 1]);
    } else {
        for (int32_t r : wide) {
            wcol_own.insert(wcol_own.end(), col.data() + rowptr[r], col.data() + rowptr[r + 1]);
            wv_own.insert(wv_own.end(), v.data() + rowptr[r], v.data() + rowptr[r + 1]);
            wrp.push_back((int32_t)wcol_own.size());
        }
        wcol = wcol_own.data();
        wv = wv_own.data();
    }
    B.nnz = wrp.back();
    int32_t win = 8192;
    while ((int64_t)(nl + win - 1) / win > k::kMaxBlocks) win *= 2;
    // small local sizes: narrower windows, so that the launch still has ~128 workgroups (a rank's 1/8 slab of the
    // 1024^2 grid got 32 workgroups on 256 CUs: 12.6 us for 6 MB)
    while (win > 1024 && (int64_t)(nl + win - 1) / win < 128) win /= 2;
    B.win = win;
    B.nwin = mw > 0 ? (nl + win - 1) / win : 0;
    std::vector<int32_t> winptr((size_t)(B.nwin + 1) * (size_t)std::max(mw, 1));
    for (int32_t r = 0; r < mw; ++r) {
        const int32_t *b = wcol + wrp[(size_t)r], *e = wcol + wrp[(size_t)r + 1];
        for (int32_t w = 0; w <= B.nwin; ++w) {
            const int64_t c0 = (int64_t)w * win;
            winptr[(size_t)w * mw + r] = wrp[(size_t)r] + (int32_t)(std::lower_bound(b, e, (int32_t)std::min<int64_t>(c0, nl)) - b);
        }
    }
    up(c, B.colidx, wcol, (size_t)B.nnz, 16);
    up(c, B.val, wv, (size_t)B.nnz, 16);
    B.winptr.upload(winptr.data(), winptr.size(), 8);
    // the general block by rows (its long rows left empty: the window kernel fills their results in)
    c->Bc.rowptr.release(); c->Bc.colidx.release(); c->Bc.val.release(); c->Bc.tile_row.release();
    c->Bc.nrows = c->Bc.ncols = c->Bc.ntiles = 0;
    c->Bc.nnz = 0;
    if (c->b_general) {
        std::vector<char> is_wide((size_t)m, 0);
        for (int32_t r : wide) is_wide[(size_t)r] = 1;
        std::vector<int32_t> crp((size_t)m + 1, 0), cci;
        std::vector<double> cv;
        for (int32_t r = 0; r < m; ++r) {
            if (!is_wide[(size_t)r]) {
                cci.insert(cci.end(), col.data() + rowptr[r], col.data() + rowptr[r + 1]);
                cv.insert(cv.end(), v.data() + rowptr[r], v.data() + rowptr[r + 1]);
            }
            crp[(size_t)r + 1] = (int32_t)cci.size();
        }
        upload_csr(c, c->Bc, m, nl, crp, cci, cv, true);
    }
    if ((size_t)m + 64 > c->y1tmp.n) c->y1tmp.alloc((size_t)m + 64);
    if ((size_t)m + 64 > c->ttmp.n) c->ttmp.alloc((size_t)m + 64);

    // B^T by rows (n_local x m), entries of a row ordered by constraint index
    // (threads own disjoint column ranges and walk the sorted rows' entries inside them: counts, then fill in
    // row order -- the same arrays as a sequential counting sort)
    std::vector<int32_t> trp((size_t)nl + 1, 0), tci((size_t)rowptr[m]);
    std::vector<double> tv((size_t)rowptr[m]);
    auto row_range = [&](int32_t r, int64_t c0, int64_t c1, int32_t &b, int32_t &e) {
        const int32_t *rb = col.data() + rowptr[r], *re = col.data() + rowptr[r + 1];
        b = rowptr[r] + (int32_t)(std::lower_bound(rb, re, (int32_t)c0) - rb);
        e = rowptr[r] + (int32_t)(std::lower_bound(rb, re, (int32_t)c1) - rb);
    };
    parallel_for(nl, [&](int64_t c0, int64_t c1, int) {
        for (int32_t r = 0; r < m; ++r) {
            int32_t b, e;
            row_range(r, c0, c1, b, e);
            for (int32_t k = b; k < e; ++k) trp[(size_t)col[(size_t)k] + 1]++;
        }
    });
    for (int32_t i = 0; i < nl; ++i) trp[(size_t)i + 1] += trp[(size_t)i];
    {
        HostBuf<int32_t> fill;
        fill.alloc((size_t)nl + 1);
        parallel_for(nl, [&](int64_t c0, int64_t c1, int) {
            for (int64_t i = c0; i < c1; ++i) fill[(size_t)i] = trp[(size_t)i];
            for (int32_t r = 0; r < m; ++r) {
                int32_t b, e;
                row_range(r, c0, c1, b, e);
                for (int32_t k = b; k < e; ++k) {
                    const int32_t p = fill[(size_t)col[(size_t)k]]++;
                    tci[(size_t)p] = r;
                    tv[(size_t)p] = v[(size_t)k];
                }
            }
        });
    }
    upload_csr(c, c->Bt, nl, m, trp, tci, tv, c->b_general);   // (a general block: tiles for the stream kernel)
    c->m = m;
    c->have_B = m > 0;
    c->pc_ready = false;
    c->ensure_vectors();
    if (c->b_general && c->tmpb.n < (size_t)c->ld) c->tmpb.alloc((size_t)c->ld);   // scratch of the B^T products (bt_update)
}

void set_block(spk_ctx *c, int which, int64_t row_begin, int32_t nrows_local, int64_t ncols_global,
               const int32_t *rowptr, const int32_t *colidx, const double *val)
{
    if (!rowptr || (!colidx && rowptr[nrows_local] > 0) || (!val && rowptr[nrows_local] > 0))
        fail(SPK_ERR_ARG, "set_block: null array");
    if (nrows_local < 0) fail(SPK_ERR_ARG, "set_block: negative row count");
    c->ensure_scratch();
    if (which == SPK_BLOCK_A00) set_block_A(c, row_begin, nrows_local, ncols_global, rowptr, colidx, val);
    else if (which == SPK_BLOCK_A10) {
        // no collective inside, but every rank sets its column slice: agree on the outcome so that a rank
        // whose slice was refused does not leave the others to run into the next collective alone
        Error local{0, ""};
        try {
            set_block_B(c, nrows_local, ncols_global, rowptr, colidx, val);
        } catch (const Error &e) {
            local = e;
        } catch (const std::exception &e) {
            local = Error{SPK_ERR_NOMEM, e.what()};
        }
        agree_or_fail(c, local.code ? &local : nullptr, "A10");
    } else fail(SPK_ERR_ARG, "set_block: unknown block %d", which);
}

// out[r] = B_r . (x .* scale)  over this rank's columns (scale == nullptr: B_r . x); MatMult on the (1,0) block.
// m <= 8: the column-window kernel for every row.  General block: short rows row by row through the CSR
// stream kernel, its few long rows through the window kernel (results scattered to their row numbers).
static void apply_B(spk_ctx *c, const double *x, const double *scale, double *out, const int32_t *done)
{
    hipStream_t s = c->stream;
    if (!c->b_general) {
        if (scale) k::wide_dot_jacobi(c->B, x, scale, c->fin(out), done, s);
        else k::wide_dot(c->B, x, c->fin(out), done, s);
        return;
    }
    const double *xs = x;
    if (scale) {  // D x0 as a vector of its own (the fused forms that avoid it are for the reference's 4 rows)
        if (c->tmpb.n < (size_t)c->ld) c->tmpb.alloc((size_t)c->ld);
        c->bt_cached = nullptr;
        k::jacobi(scale, x, c->tmpb.p, c->n_local, done, s);
        xs = c->tmpb.p;
    }
    k::spmv(c->Bc, xs, out, nullptr, nullptr, done, s);
    if (c->m_wide > 0) k::wide_dot(c->B, xs, c->fin(out), done, s, c->wide_rows.p);
}

// ---------------------------------------------------------------------------
// y = K x   (MatMult_Nest over MatMult_MPIAIJ blocks)
// ---------------------------------------------------------------------------
void op_mult(spk_ctx *c, const double *x, double *y, const int32_t *done, bool halo_done, bool reuse_bt)
{
    hipStream_t s = c->stream;
    const int32_t nl = c->n_local, m = c->m;
    if (!c->peers.empty() && !halo_done) {  // a rank may have rows to send without needing any itself
        k::gather(x, c->send_idx.p, c->send_off.back(), c->send_buf.p, done, s);
        c->comm->exchange(c->send_buf.p, c->peers, c->send_off, c->xghost.p, c->recv_off, s);
    }
    const k::OffDiag od = c->offdiag();
    const k::OffDiag *odp = c->n_ghost > 0 ? &od : nullptr;   // off-rank columns in the same kernel
    if (m > 0 && c->b_general) {
        // B^T lambda of a general block first (tiled stream kernel), the A block accumulates onto it: the row-by-row
        // epilogue of the product kernels reads such rows uncoalesced (96^3 divergence block: 474 us against 148 + 52)
        // (right after PCApply on the same multipliers -- the step path of the solve -- that product is still in the
        // scratch vector: the same kernel on the same input, so the same bits; 58 us at 96^3 become a 7 us copy)
        if (reuse_bt && c->bt_cached == x + nl && c->Bt.ntiles > 0)
            SPK_HIP(hipMemcpyAsync(y, c->tmpb.p, sizeof(double) * (size_t)nl, hipMemcpyDeviceToDevice, s));
        else k::spmv(c->Bt, x + nl, y, nullptr, nullptr, done, s);
        a_mult(c, x, y, nullptr, nullptr, done, true, odp);
    } else {
        a_mult(c, x, y, m > 0 ? &c->Bt : nullptr, x + nl, done, false, odp);
    }
    if (m > 0) {
        apply_B(c, x, nullptr, y + nl, done);
        c->comm->allreduce_sum(y + nl, m, s);
    }
}

// ---------------------------------------------------------------------------
// KSPSetUp / PCSetUp: diag(A)^-1, S^ = diag(B diag(A)^-1 B^T)
// ---------------------------------------------------------------------------
void pc_setup(spk_ctx *c, int pc_type, int schur_fact)
{
    if (!c->have_A) fail(SPK_ERR_STATE, "pc_setup: no A00 block");
    if (pc_type < SPK_PC_NONE || pc_type > SPK_PC_SCHUR) fail(SPK_ERR_ARG, "pc_setup: unknown pc_type %d", pc_type);
    if (pc_type == SPK_PC_SCHUR && !c->have_B) fail(SPK_ERR_STATE, "pc_setup: Schur fieldsplit needs the A10 block");
    if (schur_fact < SPK_SCHUR_DIAG || schur_fact > SPK_SCHUR_FULL) fail(SPK_ERR_ARG, "pc_setup: unknown schur_fact %d", schur_fact);
    hipStream_t s = c->stream;
    c->ensure_scratch();
    c->ensure_vectors();
    {   // which iteration path the solve takes must not depend on one rank's slab (KSPSetUp is collective)
        const int P = c->comm->size();
        const int32_t mine = (c->n_local % 2 == 0 ? 1 : 0) | (c->n_local > 0 ? 2 : 0);
        std::vector<int32_t> all((size_t)P, mine);
        if (P > 1) c->comm->host_allgather(&mine, all.data(), sizeof mine);
        c->even_all = c->nonempty_all = true;
        for (int32_t v : all) {
            c->even_all = c->even_all && (v & 1);
            c->nonempty_all = c->nonempty_all && (v & 2);
        }
    }
    c->dinv.alloc((size_t)c->n_local, 8);
    k::extract_diag_inv(c->Ad, c->dinv.p, s);
    const int m = c->m;
    if (m > 0 && c->b_general) {
        // S^ = diag(B D B^T), row by row: short rows one wave each; the long rows as below (scatter + window kernel)
        c->gram.release();
        c->shat.alloc((size_t)m, 8);
        k::schur_diag_rows(c->Bc, c->dinv.p, c->shat.p, s);
        if (c->m_wide > 0) {
            const int mw = c->m_wide;
            DevBuf<double> grow;
            grow.alloc((size_t)mw);
            SPK_HIP(hipMemsetAsync(c->tmp.p, 0, sizeof(double) * (size_t)c->ld, s));
            std::vector<int32_t> wp((size_t)(c->B.nwin + 1) * mw);
            SPK_HIP(hipMemcpy(wp.data(), c->B.winptr.p, wp.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
            for (int r = 0; r < mw; ++r) {
                const int k0 = wp[(size_t)r], k1 = wp[(size_t)c->B.nwin * mw + r];
                k::scatter_row(c->B.colidx.p, c->B.val.p, k0, k1, c->dinv.p, c->tmp.p, s);
                k::wide_dot(c->B, c->tmp.p, c->fin(grow.p), nullptr, s);          // row r of the long rows' Gram matrix
                SPK_HIP(hipMemcpyAsync(c->shat.p + c->wide_rows_h[(size_t)r], grow.p + r, sizeof(double), hipMemcpyDeviceToDevice, s));
                k::scatter_row(c->B.colidx.p, c->B.val.p, k0, k1, nullptr, c->tmp.p, s);
            }
            SPK_HIP(hipStreamSynchronize(s));
        }
        c->comm->allreduce_sum(c->shat.p, m, s);
        SPK_HIP(hipStreamSynchronize(s));
    } else if (m > 0) {
        c->gram.alloc((size_t)m * m);
        c->shat.alloc((size_t)m);
        // row r of B .* dinv scattered densely, then B * that = G[r, :]
        SPK_HIP(hipMemsetAsync(c->tmp.p, 0, sizeof(double) * (size_t)c->ld, s));
        std::vector<int32_t> wp((size_t)(c->B.nwin + 1) * m);
        SPK_HIP(hipMemcpy(wp.data(), c->B.winptr.p, wp.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (int r = 0; r < m; ++r) {
            const int k0 = wp[(size_t)r], k1 = wp[(size_t)c->B.nwin * m + r];
            k::scatter_row(c->B.colidx.p, c->B.val.p, k0, k1, c->dinv.p, c->tmp.p, s);
            k::wide_dot(c->B, c->tmp.p, c->fin(c->gram.p + (size_t)r * m), nullptr, s);
            k::scatter_row(c->B.colidx.p, c->B.val.p, k0, k1, nullptr, c->tmp.p, s);
        }
        c->comm->allreduce_sum(c->gram.p, m * m, s);
        SPK_HIP(hipStreamSynchronize(s));
        std::vector<double> G((size_t)m * m), sh((size_t)m);
        SPK_HIP(hipMemcpy(G.data(), c->gram.p, G.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int r = 0; r < m; ++r) sh[(size_t)r] = G[(size_t)r * m + r];
        SPK_HIP(hipMemcpy(c->shat.p, sh.data(), sh.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    // FP32 copies for the inner solve
    c->a32.release(); c->d32.release(); c->x32.release(); c->y32a.release(); c->y32b.release();
    c->Ab3.v32.release();
    c->Ab.vtop32.release();
    c->Ab.vbot32.release();
    const bool dict = c->spmv_format != 0 && c->Adict.ok;   // (the sweeps then decode the codes and round to single precision)
    if (c->inner_sweeps > 0 && !dict && c->spmv_format == 1) {   // ... or the 2x2-blocked value planes
        c->Ab.vtop32.alloc_raw((size_t)(2 * c->Ab.nblocks + 8), 32);
        c->Ab.vbot32.alloc_raw((size_t)(2 * c->Ab.nblocks + 8), 32);
        k::cvt_vals_f32(c->Ab.vtop.p, c->Ab.vtop32.p, 2 * c->Ab.nblocks, s);
        k::cvt_vals_f32(c->Ab.vbot.p, c->Ab.vbot32.p, 2 * c->Ab.nblocks, s);
    }
    if (c->inner_sweeps > 0 && !dict && c->spmv_format == 2) {   // the sweeps read the 3x3-blocked planes in single precision
        c->Ab3.v32.alloc_raw((size_t)(9 * c->Ab3.ldp), 32);
        k::cvt_vals_f32(c->Ab3.v.p, c->Ab3.v32.p, 9 * c->Ab3.ldp, s);
    }
    if (c->inner_sweeps > 0 && c->spmv_format == 0) {
        c->a32.alloc((size_t)c->Ad.nnz, 32);
        k::cvt_vals_f32(c->Ad.val.p, c->a32.p, c->Ad.nnz, s);
    }
    if (c->inner_sweeps > 0) {
        c->d32.alloc((size_t)c->n_local, 8);
        c->x32.alloc((size_t)c->n_local, 8);
        c->y32a.alloc((size_t)c->n_local, 8);
        c->y32b.alloc((size_t)c->n_local, 8);
        k::cvt_vals_f32(c->dinv.p, c->d32.p, c->n_local, s);
    }
    // dense rows of B D for the fused path (Schur LOWER/FULL, even local size)
    c->bd.release();
    if (pc_type == SPK_PC_SCHUR && m > 0 && !c->b_general && (schur_fact == SPK_SCHUR_FULL || schur_fact == SPK_SCHUR_LOWER) &&
        c->even_all && c->inner_sweeps == 0) {
        c->bd.alloc((size_t)c->ld * m, 16);
        k::build_bd(c->Bt, c->dinv.p, m, c->ld, c->bd.p, s);
        // rows 2q / 2q+1 on even / odd entries (x / y degrees of freedom): m/2 planes instead of m rows
        c->bdpk.release();
        c->bd_packed = false;
        if (m % 2 == 0 && !getenv("SPK_BD_DENSE")) {
            c->bdpk.alloc((size_t)c->ld * (m / 2), 16);
            DevBuf<int32_t> bad;
            bad.alloc(1);
            k::pack_bd(c->bd.p, c->ld, c->n_local, m, c->bdpk.p, bad.p, s);
            int32_t hb = 1;
            SPK_HIP(hipMemcpyAsync(&hb, bad.p, sizeof hb, hipMemcpyDeviceToHost, s));
            SPK_HIP(hipStreamSynchronize(s));
            c->bd_packed = hb == 0;
            if (!c->bd_packed) c->bdpk.release();
        }
    }
    SPK_HIP(hipStreamSynchronize(s));
    c->pc_type = pc_type;
    c->schur_fact = schur_fact;
    {   // does EVERY rank's slab fit the resident cycle kernel (restart <= 30)?  agreed here, like the iteration path above
        const int planes = !c->bd.p ? 0 : (c->bd_packed ? m / 2 : m);
        const int32_t mine = (c->spmv_format == 1 && c->Adict.ok && c->Adict.bs == 2 && c->n_local % 2 == 0 &&
                              k::resident_fits(c->Adict, c->num_cus, 30, planes)) ? 1 : 0;
        const int P = c->comm->size();
        std::vector<int32_t> all((size_t)P, mine);
        if (P > 1) c->comm->host_allgather(&mine, all.data(), sizeof mine);
        c->res_fit_all = true;
        for (int32_t v : all) c->res_fit_all = c->res_fit_all && v != 0;
    }
    c->pc_ready = true;
}

// ---------------------------------------------------------------------------
// y = M^-1 x   (PCApply_Jacobi / PCApply_FieldSplit_Schur)
// ---------------------------------------------------------------------------
// y (op) A^ ^-1 x with the FP32 Richardson/Jacobi sweeps; mode 0: y = , mode 1: y -=
static void inner_apply(spk_ctx *c, const double *x, double *y, int mode, const int32_t *done)
{
    hipStream_t s = c->stream;
    const int32_t nl = c->n_local;
    const float om = (float)c->inner_omega;
    float *ya = c->y32a.p, *yb = c->y32b.p;
    k::cvt_scale_f32(x, c->d32.p, om, c->x32.p, ya, nl, done, s);
    for (int sw = 1; sw < c->inner_sweeps; ++sw) {
        if (!c->peers.empty()) {  // halo of the single-precision iterate, staged as doubles
            k::gather_f32(ya, c->send_idx.p, c->send_off.back(), c->send_buf.p, done, s);
            c->comm->exchange(c->send_buf.p, c->peers, c->send_off, c->xghost.p, c->recv_off, s);
        }
        if (c->spmv_format != 0 && c->Adict.ok) k::jacobi_sweep_f32_dict(c->Adict, c->d32.p, om, c->x32.p, ya, yb, done, s);
        else if (c->spmv_format == 2) k::jacobi_sweep_f32_b3(c->Ab3, c->d32.p, om, c->x32.p, ya, yb, done, s);
        else if (c->spmv_format == 1) k::jacobi_sweep_f32_b2(c->Ab, c->d32.p, om, c->x32.p, ya, yb, done, s);
        else k::jacobi_sweep_f32(c->Ad, c->a32.p, c->d32.p, om, c->x32.p, ya, yb, done, s);
        if (c->n_ghost > 0) k::sweep_offdiag_f32(c->Ao, c->ao_rows.p, c->d32.p, om, c->xghost.p, yb, done, s);
        std::swap(ya, yb);
    }
    k::cvt_f32_out(ya, y, mode, nl, done, s);
}

void op_pc_apply(spk_ctx *c, const double *x, double *y, const int32_t *done)
{
    hipStream_t s = c->stream;
    const int32_t nl = c->n_local, m = c->m;
    const double *x0 = x, *x1 = x + nl;
    double *y0 = y, *y1 = y + nl;
    // general block, UPPER / FULL: the last thing written to the scratch vector is B^T y1 (bt_update) -- op_mult on the
    // result may take it from there
    c->bt_cached = (c->b_general && c->Bt.ntiles > 0 && m > 0 && c->pc_type == SPK_PC_SCHUR &&
                    (c->schur_fact == SPK_SCHUR_UPPER || c->schur_fact == SPK_SCHUR_FULL)) ? y1 : nullptr;
    if (c->inner_sweeps > 0 && c->pc_type != SPK_PC_NONE) {
        // same block algebra with the inner solve standing for diag(A)^-1
        if (c->pc_type == SPK_PC_JACOBI) {
            inner_apply(c, x0, y0, 0, done);
            k::copy_small(x1, y1, m, done, s);
            return;
        }
        switch (c->schur_fact) {
        case SPK_SCHUR_DIAG:
            inner_apply(c, x0, y0, 0, done);
            k::schur_y1(SPK_SCHUR_DIAG, m, x1, nullptr, c->shat.p, y1, done, s);
            break;
        case SPK_SCHUR_UPPER:
            k::schur_y1(SPK_SCHUR_UPPER, m, x1, nullptr, c->shat.p, y1, done, s);
            k::bt_update(2, c->Bt, c->dinv.p, x0, y1, c->tmp.p, done, s, c->b_general ? c->tmpb.p : nullptr);   // x0 - B^T y1
            inner_apply(c, c->tmp.p, y0, 0, done);
            break;
        default:  // LOWER, FULL
            inner_apply(c, x0, y0, 0, done);
            apply_B(c, y0, nullptr, c->ttmp.p, done);
            c->comm->allreduce_sum(c->ttmp.p, m, s);
            k::schur_y1(c->schur_fact, m, x1, c->ttmp.p, c->shat.p, y1, done, s);
            if (c->schur_fact == SPK_SCHUR_FULL) {
                k::bt_update(3, c->Bt, c->dinv.p, x0, y1, c->tmp.p, done, s, c->b_general ? c->tmpb.p : nullptr);   // B^T y1
                inner_apply(c, c->tmp.p, y0, 1, done);                            // y0 -= A^ ^-1 B^T y1
            }
            break;
        }
        return;
    }
    if (c->pc_type == SPK_PC_NONE) {
        k::axpby(1.0, x, 0.0, y, (int64_t)nl + m, done, s);
        return;
    }
    if (c->pc_type == SPK_PC_JACOBI) {
        k::jacobi(c->dinv.p, x0, y0, nl, done, s);
        k::copy_small(x1, y1, m, done, s);  // zero diagonal of the (1,1) block -> 1
        return;
    }
    switch (c->schur_fact) {
    case SPK_SCHUR_DIAG:
        k::jacobi(c->dinv.p, x0, y0, nl, done, s);
        k::schur_y1(SPK_SCHUR_DIAG, m, x1, nullptr, c->shat.p, y1, done, s);
        break;
    case SPK_SCHUR_UPPER:
        k::schur_y1(SPK_SCHUR_UPPER, m, x1, nullptr, c->shat.p, y1, done, s);
        k::bt_update(0, c->Bt, c->dinv.p, x0, y1, y0, done, s, c->b_general ? c->tmpb.p : nullptr);
        break;
    case SPK_SCHUR_LOWER:
    default:  // FULL
        // t = B (D x0) (without storing D x0 for the reference's few long rows)
        apply_B(c, x0, c->dinv.p, c->ttmp.p, done);
        c->comm->allreduce_sum(c->ttmp.p, m, s);
        k::schur_y1(c->schur_fact, m, x1, c->ttmp.p, c->shat.p, y1, done, s);
        if (c->schur_fact == SPK_SCHUR_LOWER) k::jacobi(c->dinv.p, x0, y0, nl, done, s);
        else k::bt_update(1, c->Bt, c->dinv.p, x0, y1, y0, done, s, c->b_general ? c->tmpb.p : nullptr);
        break;
    }
}

// ---------------------------------------------------------------------------
// KSPSolve_FGMRES, device resident
// ---------------------------------------------------------------------------
static void ensure_krylov(spk_ctx *c, const spk_opts &o)
{
    c->ensure_scratch();
    c->ensure_vectors();
    const int mk = o.restart;
    if (mk < 1 || mk > k::kBigNv - 2) fail(SPK_ERR_ARG, "fgmres: restart %d outside [1,%d]", mk, k::kBigNv - 2);
    const int32_t hist_cap = (int32_t)std::min<int64_t>((int64_t)std::max(o.max_it, 0) + 2, 1 << 22);
    if (c->ws_restart != mk) {
        // one more of each than the cycle uses: the un-normalised three-launch form lets the last iteration of a cycle
        // write its (unused) next vectors too
        c->V.alloc((size_t)c->ld * (mk + 2));
        c->Z.alloc((size_t)c->ld * (mk + 1));
        c->ws_restart = mk;
    }
    const int ldh = mk + 2;
    const size_t nd = (size_t)ldh * (mk + 1) + 4 * (size_t)(mk + 2) + 8 * (size_t)(mk + 2) + (size_t)hist_cap + 64;
    if (c->kry_d.n < nd) c->kry_d.alloc(nd);
    if (!c->kst.p) c->kst.alloc(1);
    double *p = c->kry_d.p;
    c->ka.st = c->kst.p;
    c->ka.ldh = ldh;
    c->ka.H = p;      p += (size_t)ldh * (mk + 1);
    c->ka.cc = p;     p += mk + 2;
    c->ka.ss = p;     p += mk + 2;
    c->ka.rs = p;     p += mk + 2;
    c->ka.nrs = p;    p += mk + 2;
    c->ka.hcol = nullptr;
    c->ka.tb = p;     p += 8 * (size_t)(mk + 2);
    c->ka.hist = p;
    c->ka.hist_cap = hist_cap;
}

void fgmres(spk_ctx *c, const double *b, double *x, const spk_opts &o, spk_result *res, double *history,
            int32_t history_cap)
{
    if (!c->have_A) fail(SPK_ERR_STATE, "fgmres: no operator");
    if (!c->pc_ready) fail(SPK_ERR_STATE, "fgmres: call spk_pc_setup first (KSPSetUp)");
    if (o.orthog != SPK_ORTHOG_CGS && o.orthog != SPK_ORTHOG_MGS) fail(SPK_ERR_ARG, "fgmres: unknown orthogonalisation %d", o.orthog);
    if (o.cgs_refine < SPK_REFINE_NEVER || o.cgs_refine > SPK_REFINE_ALWAYS) fail(SPK_ERR_ARG, "fgmres: unknown cgs_refine %d", o.cgs_refine);
    ensure_krylov(c, o);
    hipStream_t s = c->stream;
    const int mk = o.restart;
    const int64_t N = (int64_t)c->n_local + c->m, ld = c->ld;
    const int64_t n_dot = (int64_t)c->n_local + (c->comm->rank() == 0 ? c->m : 0);
    const int32_t *done = &c->kst.p->done;
    const int32_t *loc_done = &c->kst.p->loc_done;
    const double *inv_tt = &c->kst.p->inv_tt;
    double *sm = c->small.p;  // [0..63] dots (+w.w), [64] norm^2, [128] ||b||^2
    double *V = c->V.p, *Z = c->Z.p;
    auto Vj = [&](int j) { return V + (size_t)ld * j; };
    auto Zj = [&](int j) { return Z + (size_t)ld * j; };

    SPK_HIP(hipStreamSynchronize(s));
    const auto t0 = std::chrono::steady_clock::now();

    // small[]: two parity sets so that a deferred Givens step (fused path) can still read iteration
    // j-1's scalars while iteration j produces its own: dots at p*128, norm (+ B D w') at p*128+64
    auto dotsbuf = [&](int p) { return sm + (p & 1) * 128; };
    auto nrmbuf = [&](int p) { return sm + (p & 1) * 128 + 64; };
    double *sm2 = sm + 256, *nrm2b = sm + 320, *bn2 = sm + 384, *w1side = c->y1tmp.p + 48;

    // ||b|| for KSPConvergedDefault
    k::sqnorm(b, n_dot, c->fin(bn2), nullptr, s);
    c->comm->allreduce_sum(bn2, 1, s);
    k::krylov_init(c->ka, o, bn2, s);

    // initial residual into V0
    if (!o.guess_nonzero) {
        SPK_HIP(hipMemsetAsync(x, 0, sizeof(double) * (size_t)N, s));
        SPK_HIP(hipMemcpyAsync(Vj(0), b, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice, s));
    } else {
        op_mult(c, x, c->tmp.p, nullptr);
        SPK_HIP(hipMemcpyAsync(Vj(0), b, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice, s));
        k::axpby(-1.0, c->tmp.p, 1.0, Vj(0), N, nullptr, s);
    }

    // fused Schur path: VecScale + PC + B^T part of the operator in one pass ("head"), B D w' in the
    // MAXPY pass, the Givens step of iteration j-1 inside the head kernel of iteration j
    // -ksp_gmres_restart beyond 62: the step-by-step path, Gram-Schmidt in chunks of <= 40 vectors (the fused kernels
    // keep one lane / one LDS slot per basis vector)
    const bool big = mk > k::kMaxNv - 2;
    if (big && c->bigdots.n < 2 * ((size_t)mk + 4)) c->bigdots.alloc(2 * ((size_t)mk + 4));   // first and refinement pass
    // (a long restart keeps the head kernel -- VecScale + PCApply + the B^T part of the product in one pass, the product
    // accumulating onto it, B D w' out of the last MAXPY chunk -- and only the Givens step is a launch of its own: its
    // column no longer fits the head kernel's workgroup 0)
    const bool fused = o.fused && c->bd.p && c->pc_type == SPK_PC_SCHUR;
    const int m = c->m;
    const int32_t nl = c->n_local;
    // the same head kernel without a constraint block: Jacobi on K = A (the reference as written,
    // SaddlePointProblem.c:66, and BASELINE config 2): VecScale + PCApply_Jacobi + deferred Givens
    const bool fusedj = o.fused && !fused && c->pc_type == SPK_PC_JACOBI && m == 0 && c->even_all && c->nonempty_all &&
                        c->inner_sweeps == 0;
    const bool head = fused || fusedj;
    const int nn = fused ? 1 + m : 1;  // norm (+ B D w') coming out of the last MAXPY of an iteration
    const int bpk = fused && c->bd_packed ? 1 : 0;   // B D as m/2 parity-interleaved planes
    const double *bdp = fused ? (bpk ? c->bdpk.p : c->bd.p) : nullptr;
    // single-reduction Gram-Schmidt (fused CGS without refinement): h = V^T w, q = B D w and w.w
    // come out of ONE pass and ONE all-reduce; ||w'||^2 = w.w - |h|^2 and B D w' = q - sum h_i B D v_i
    // follow without touching w' -- one collective per iteration instead of two.
    // Opt-in (opts.single_reduce = 1): the subtraction cancels, see include/spk.h.
    // The Jacobi head path (K = A) takes the same route with m = 0: ||w'||^2 = w.w - |h|^2 only.
    // (MDot then carries restart + m vectors: both must fit one reduction.)
    const bool single = head && o.orthog == SPK_ORTHOG_CGS && o.cgs_refine == SPK_REFINE_NEVER &&
                        o.single_reduce == 1 && mk + c->m <= k::kMaxNv - 1;

    // How one classical Gram-Schmidt iteration (two reductions) is launched on the head-kernel paths
    // (opts.iteration_form / SPK_ITER_FORM; include/spk.h lists the forms and their measured times).
    int form = o.iteration_form;
    if (const char *e = getenv("SPK_ITER_FORM")) {   // test-only knob: the whole GPU suite is run under each form with it
        char *end = nullptr;
        const long f = strtol(e, &end, 10);
        if (end == e || *end || f < SPK_ITER_AUTO || f > SPK_ITER_LAST)
            fail(SPK_ERR_ARG, "SPK_ITER_FORM=%s: not an iteration form (%d..%d)", e, (int)SPK_ITER_AUTO, (int)SPK_ITER_LAST);
        form = (int)f;
    }
    if (form < SPK_ITER_AUTO || form > SPK_ITER_LAST) fail(SPK_ERR_ARG, "fgmres: unknown iteration_form %d", form);
    // AUTO: three launches on an UN-normalised basis -- MDot (raw inner products), MAXPY + norm + next PCApply, plain SpMV
    // with the Givens step and the new scale factor in one extra workgroup (GivensRider).  V~_j = h_{j,j-1} v_j: nothing
    // compounds, no vector is ever scaled in memory.  Either matrix format, any number of ranks, any transport.
    const bool un3 = head && !single && o.orthog == SPK_ORTHOG_CGS && o.cgs_refine == SPK_REFINE_NEVER &&
                     mk + c->m <= k::kMaxNv - 2 && (form == SPK_ITER_UNNORM || form == SPK_ITER_AUTO || form == SPK_ITER_RESIDENT);
    // RESIDENT: one launch per restart cycle, the basis in registers (spk_k_resident.hip): what AUTO takes where it fits
    static const bool res_env_off = [] { const char *e = getenv("SPK_RESIDENT"); return e && !strcmp(e, "0"); }();
    const int res_planes = !fused ? 0 : (bpk ? m / 2 : m);
    // several ranks: every rank must take it (agreed at KSPSetUp: res_fit_all), the collectives must be the peer-store
    // backend's (they run inside the launch), and the halo rows must be contiguous send ranges
    const bool res_multi = c->comm->size() > 1;
    const bool res_rank_ok = !res_multi ? (c->peers.empty() && c->n_ghost == 0)
                                        : (c->res_fit_all && c->comm->fuses() && (c->peers.empty() || c->send_ranges.n > 0));
    const bool resident = un3 && (form == SPK_ITER_RESIDENT || (form == SPK_ITER_AUTO && !res_env_off)) && res_rank_ok &&
                          c->spmv_format == 1 && c->Adict.ok && c->Adict.bs == 2 && nl % 2 == 0 && mk >= (res_multi ? 3 : 2) &&
                          k::resident_fits(c->Adict, c->num_cus, mk, res_planes);
    if (resident) {
        const size_t need = (size_t)k::resident_scratch_doubles(c->num_cus, mk);
        if (c->res_P.n < need) c->res_P.alloc(need);
    }
    const bool two_ok = head && !single && o.orthog == SPK_ORTHOG_CGS && o.cgs_refine == SPK_REFINE_NEVER &&
                        c->spmv_format == 1 && c->Ab.ok && !c->Ab.long_rows && mk + c->m <= k::kMaxNv - 2;
    const bool two = two_ok && !un3 && (form == SPK_ITER_TWO_LAUNCH || form == SPK_ITER_THREE_LAUNCH);
    // forms 2 / 3 (normalised basis, MDot inside / behind the SpMV launch): opt-in, kept for comparison
    const bool three = two && form != SPK_ITER_TWO_LAUNCH;
    // BA: MAXPY + the next SpMV in one launch behind neighbour flags, un-normalised basis (two launches per
    // iteration: MDot, BA).  Single rank; opt-in (opts.iteration_form = 4 / SPK_ITER_FORM=4)
    const bool ba = two_ok && form == SPK_ITER_BA && c->Ab.ba_ok && c->peers.empty() && c->comm->size() == 1 &&
                    (c->m <= 4 || 8 * c->Ab.ba_slots + 1 <= 512);   // every workgroup resident at once (registers: 3 / 2 per CU)
    if (ba) {
        const size_t nfl = (size_t)(8 * c->Ab.ba_slots + 1) * 32;
        if (c->ba_flags.n < nfl) {
            c->ba_flags.alloc(nfl);
            c->ba_seq = 0;
        }
    }
    if (ba || un3) {
        if (c->ba_sc.n < (size_t)mk + 2) c->ba_sc.alloc((size_t)mk + 2);
    }
    if (two && c->zun.n < (size_t)ld) c->zun.alloc((size_t)ld);
    const int lam_in_dot = c->comm->rank() == 0 ? 1 : 0;
    c->last_form = resident ? SPK_ITER_RESIDENT : ba ? SPK_ITER_BA : un3 ? SPK_ITER_UNNORM : two ? (three ? SPK_ITER_THREE_LAUNCH : SPK_ITER_TWO_LAUNCH)
                   : head ? SPK_ITER_FOUR_LAUNCH : -1;
    c->last_single = single ? 1 : 0;

    c->ka.tentative = single ? 1 : 0;
    KrylovState st{};
    int32_t errword = 0;
    int cycles = 0;
    // The solve's state reaches the host through pinned memory written by krylov_cycle_begin (no copy on the stream); the
    // host waits for the event behind that launch only after it has enqueued the start of the cycle (kAhead iterations),
    // whose kernels are gated off on the device if the solve is over: the stream never drains at a restart.  (Pageable
    // read-back copies and a stream synchronisation per cycle cost 35-85 us of idle GPU: 256^2 28.1 -> 23.7 us per iteration.)
    if (!c->pin_state) SPK_HIP(hipHostMalloc(&c->pin_state, 512, hipHostMallocDefault));
    if (!c->state_ev) SPK_HIP(hipEventCreateWithFlags(&c->state_ev, hipEventDisableTiming));
    KrylovState *ps = (KrylovState *)c->pin_state;
    int32_t *pe = (int32_t *)((char *)c->pin_state + 384);
    pe[0] = pe[1] = 0;
    bool pend_check = false;   // the previous cycle's state is on its way to the pinned slot (state_ev)
    bool finished = false;
    const int kAhead = 2;      // iterations of a cycle enqueued before the host looks at the previous cycle's verdict
    auto read_state = [&]() {
        SPK_HIP(hipEventSynchronize(c->state_ev));
        st = *ps;
        errword = pe[0];
        pend_check = false;
        if (pe[1]) c->comm->check(s);         // a peer that never arrived: SPK_ERR_COMM instead of a wrong answer
        if (errword) c->check_device_error();  // a reduction that timed out: SPK_ERR_HIP, not KSP_DIVERGED_NANORINF
        return st.done != 0;
    };
    for (;;) {
        // ---- cycle start: ||r|| (parity slot 1 = "iteration -1"), convergence test, v0 = r/||r|| ----
        // (from the second cycle on the previous cycle's end has formed r = b - K x and these sums in one pass)
        if (cycles == 0) {
            if (fused) k::sqnorm_bd(Vj(0), N, n_dot, c->bd.p, ld, nl, m, w1side, c->fin(nrmbuf(1)), done, s);
            else k::sqnorm(Vj(0), n_dot, c->fin(nrmbuf(1)), done, s);
        }
        c->comm->allreduce_sum(nrmbuf(1), nn, s);
        // (the kernel also reports the state it finds / leaves into pinned memory: the verdict on the PREVIOUS cycle and,
        // through its own convergence test on the true residual, on the solve -- read by the host at loc == kAhead)
        const k::StateReport report{ps, pe, c->errw.p, c->comm->error_dev()};
        k::krylov_cycle_begin(c->ka, nrmbuf(1), s, (single || two || ba || un3) ? c->ka.tb : nullptr, m,
                              (ba || un3) ? c->ba_sc.p : nullptr, &report);
        SPK_HIP(hipEventRecord(c->state_ev, s));
        pend_check = true;
        if (!head) k::scale_dev(Vj(0), N, inv_tt, done, s);

        bool stop = false;
        int last = -1;  // last iteration of this cycle whose Givens step is still pending (fused path)
        int64_t its_cap = INT64_MAX;   // iterations this cycle may still run before -ksp_max_it (known once the report is read)
        int pend_loc = -1;   // ... handed to the cycle-end launch (its Hessenberg column and norm)
        const double *pend_h = nullptr;
        double *pend_n = nullptr;
        // single-reduction mode: the MAXPY of iteration loc also runs the head of iteration loc+1 and the
        // Givens step of iteration loc (k::maxpy_head): three launches and one reduction per iteration
        bool head_done = false, prev_inhead = false;
        auto wl = [&](int p) { return sm + 400 + (p & 1) * 8; };  // lambda entries of w, side copies
        for (int loc = 0; loc < mk && !stop; ++loc) {
            if (pend_check && loc == kAhead) {
                if (read_state()) {   // the previous cycle ended the solve: what was enqueued of this one is gated off
                    finished = true;  // on the device
                    break;
                }
                its_cap = (int64_t)o.max_it - st.its;   // (the report is this cycle's cycle_begin: its = iterations before it)
            }
            // -ksp_max_it ends the solve inside this cycle: the device stops there by itself, the host need not enqueue
            // (gated) launches beyond it
            if (loc >= its_cap) break;
            const int32_t *done = &c->kst.p->skip_iter;  // the gate of everything inside an iteration
            double *w = Vj(loc + 1);
            double *db = big ? c->bigdots.p : dotsbuf(loc), *nb = nrmbuf(loc);
            if (ba || un3) {
                // the product K z~ of a vector (halo, then diagonal and off-rank columns in one kernel), either format
                auto product = [&](const double *zvec, double *wvec, bool halo_done, const k::GivensRider *rider = nullptr) {
                    k::SendRanges srp = c->send_ranges;
                    if (!c->peers.empty() && !halo_done) {
                        if (srp.n == 0) k::gather(zvec, c->send_idx.p, c->send_off.back(), c->send_buf.p, done, s);
                        c->comm->exchange(c->send_buf.p, c->peers, c->send_off, c->xghost.p, c->recv_off, s);
                    }
                    const k::OffDiag od = c->offdiag();
                    const k::OffDiag *odp = c->n_ghost > 0 ? &od : nullptr;
                    a_mult(c, zvec, wvec, nullptr, nullptr, done, fused, odp, rider);
                };
                if (loc == 0) {
                    // first iteration of a cycle: the classic head on the normalised r, then the plain product
                    k::SendRanges sr0 = c->send_ranges;
                    const bool packed = sr0.n > 0;
                    const bool inhead = packed && c->comm->fused_halo(sr0, c->xghost.p);
                    if (fused)
                        k::fused_head(Vj(0), nrmbuf(1), w1side, c->dinv.p, bdp, ld, c->shat.p, c->gram.p, c->schur_fact, nl, m,
                                      Zj(0), w, c->ka, -1, dotsbuf(1), done, s, packed ? &sr0 : nullptr, bpk, wl(0));
                    else
                        k::fused_head(Vj(0), nrmbuf(1), nullptr, c->dinv.p, nullptr, ld, nullptr, nullptr, SPK_SCHUR_LOWER, nl, 0,
                                      Zj(0), nullptr, c->ka, -1, dotsbuf(1), done, s, packed ? &sr0 : nullptr);
                    product(Zj(0), w, inhead);
                }
                if (resident) {
                    // the rest of the cycle is ONE launch: iterations 0 .. mk-1 with the basis in registers
                    k::ResidentArgs r{};
                    r.mk = mk; r.m = fused ? m : 0; r.packed = bpk; r.fact = fused ? c->schur_fact : SPK_SCHUR_LOWER;
                    r.lam_in_dot = lam_in_dot; r.nl = nl; r.ld = ld;
                    r.V0 = Vj(0); r.V1 = Vj(1); r.Z = Z; r.dinv = c->dinv.p; r.bd = bdp; r.ldb = ld;
                    r.shat = c->shat.p; r.gram = c->gram.p; r.P = c->res_P.p; r.ka = c->ka; r.sc_out = c->ba_sc.p;
                    r.err = c->errw.p; r.ticks = c->fin_ticks;
                    r.sr0 = k::SendRanges{};
                    r.sr1 = k::SendRanges{};
                    if (res_multi) {
                        // the launch's mk + 1 all-reduces and mk - 1 halo exchanges: consecutive sequence numbers, reserved now
                        r.sr0 = c->send_ranges;
                        r.sr1 = c->send_ranges;
                        if (!c->comm->resident_plan(mk + 1, c->peers.empty() ? 0 : mk - 1, r.ar, r.sr0, r.sr1, c->xghost.p))
                            fail(SPK_ERR_COMM, "fgmres: the communicator cannot carry a resident cycle (agreed at set-up, refused now)");
                        if (c->peers.empty()) r.sr0.n = r.sr1.n = 0;
                        if (c->n_ghost > 0) r.od = c->offdiag();
                    }
                    if (!k::cycle_resident(c->Adict, c->num_cus, r, done, s)) fail(SPK_ERR_STATE, "fgmres: resident cycle kernel refused its shape");
                    break;
                }
                // raw inner products of the un-normalised basis with w~ (and B D w~); scaled where they are consumed
                {
                    const bool one = loc + 1 + m <= 40;
                    const bool spl = bpk && one;
                    const k::PeerAR ar = one ? c->comm->fused_allreduce(loc + 2 + (fused ? m : 0), k::kStatArDots) : k::PeerAR{};
                    k::mdot(V, ld, loc + 1, w, N, n_dot, c->fin(db, ar), done, s, fused ? (spl ? c->bdpk.p : c->bd.p) : nullptr,
                            fused ? m : 0, spl ? 1 : 0);
                    if (!ar.P) c->comm->allreduce_sum(db, loc + 2 + (fused ? m : 0), s);
                }
                if (un3) {
                    const k::PeerAR ar2 = c->comm->fused_allreduce(1, k::kStatArNorm);
                    k::IterB b{};
                    b.V = V; b.ldv = ld; b.nv = loc + 1; b.dots = db; b.tb = c->ka.tb;
                    b.w = w; b.dinv = c->dinv.p; b.bd = bdp; b.ldb = ld; b.shat = c->shat.p; b.gram = c->gram.p;
                    b.fact = fused ? c->schur_fact : SPK_SCHUR_LOWER;
                    b.nl = nl; b.m = m; b.packed = bpk;
                    b.zun = Zj(loc + 1); b.c = fused ? Vj(loc + 2) : nullptr; b.wl_in = wl(loc); b.wl_out = wl(loc + 1);
                    b.lam_in_dot = lam_in_dot;
                    b.partials = c->partials.p; b.out = nb; b.ar = ar2; b.err = c->errw.p; b.fin_ticks = c->fin_ticks;
                    b.sc = c->ba_sc.p; b.hbuf = sm2; b.ka = c->ka; b.loc = loc;
                    // ||w'||^2 is left as one partial per workgroup: the rider of the product launch reduces it (and all-reduces
                    // it, peer-store) beside the row tiles -- the product of an un-normalised vector does not need the norm.
                    // (Not with an all-reduce that is a launch of its own, nor behind the last iteration of a cycle.)
                    const bool defer = loc + 1 < mk && (c->comm->size() == 1 || ar2.P);
                    b.defer_fin = defer ? 1 : 0;
                    k::SendRanges sr = c->send_ranges;
                    const bool inb = sr.n > 0 && c->comm->fused_halo(sr, c->xghost.p);
                    if (sr.n > 0) b.sr = sr;
                    b.done = done;
                    const int fin_n = k::iter_maxpy_uhead(b, s);
                    if (!ar2.P && !defer) c->comm->allreduce_sum(nb, 1, s);
                    // the Givens step of this iteration (and the new vector's scale factor) ride in the next product launch
                    k::GivensRider gr{c->ka, loc, sm2, nb, c->ba_sc.p, defer ? c->partials.p : nullptr, defer ? fin_n : 0,
                                      k::FinErr{c->errw.p, c->fin_ticks}, defer ? ar2 : k::PeerAR{}};
                    if (loc + 1 < mk) product(Zj(loc + 1), Vj(loc + 2), inb, &gr);
                    else pend_h = sm2, pend_n = nb, pend_loc = loc;   // no product behind it: the step runs in the cycle-end launch
                    last = -1;
                }
                if (ba) {
                k::IterBA p{};
                p.browptr = c->Ab.browptr.p; p.bcol = c->Ab.bcol.p; p.vtop = c->Ab.vtop.p; p.vbot = c->Ab.vbot.p;
                p.tile_brow = c->Ab.tile_brow.p; p.ntiles = c->Ab.ntiles; p.tiles_per_xcd = (c->Ab.ntiles + 7) / 8;
                p.slots = c->Ab.ba_slots; p.tb = c->Ab.ba_tb; p.chunk = c->Ab.ba_chunk; p.nbr = nullptr; p.wt = c->Ab.ba_wt.p;
                p.tdesc = c->Ab.tile_desc.p;
                p.flags = c->ba_flags.p; p.seq = ++c->ba_seq;
                p.V = V; p.ldv = ld; p.nv = loc + 1; p.dots = db; p.sc = c->ba_sc.p; p.tb_ = c->ka.tb;
                p.w = w; p.dinv = c->dinv.p; p.bd = bdp; p.ldb = ld; p.shat = c->shat.p; p.gram = c->gram.p;
                p.fact = fused ? c->schur_fact : SPK_SCHUR_LOWER;
                p.nl = nl; p.m = m; p.packed = bpk;
                p.last = loc + 1 >= mk ? 1 : 0;
                p.zout = p.last ? nullptr : Zj(loc + 1);
                p.wnext = p.last ? nullptr : Vj(loc + 2);
                p.wl_in = wl(loc); p.wl_out = wl(loc + 1);
                p.hbuf = sm2;
                p.lam_in_dot = lam_in_dot;
                p.partials = c->partials.p; p.nrm_out = nb;
                p.err = c->errw.p; p.fin_ticks = c->fin_ticks;
                p.ka = c->ka; p.loc = loc; p.done = done;
                k::iter_ba(p, s);
                last = -1;  // the Givens step of this iteration ran inside the launch
                }
            } else if (two) {
                k::SendRanges sr0 = c->send_ranges;
                const bool packed = sr0.n > 0;
                if (loc == 0) {
                    // first iteration of a cycle: the classic head on the normalised r (no kernel B behind it)
                    bool inhead = packed && c->comm->fused_halo(sr0, c->xghost.p);
                    if (fused)
                        k::fused_head(Vj(0), nrmbuf(1), w1side, c->dinv.p, bdp, ld, c->shat.p, c->gram.p, c->schur_fact, nl, m,
                                      Zj(0), w, c->ka, -1, dotsbuf(1), done, s, packed ? &sr0 : nullptr, bpk);
                    else
                        k::fused_head(Vj(0), nrmbuf(1), nullptr, c->dinv.p, nullptr, ld, nullptr, nullptr, SPK_SCHUR_LOWER, nl, 0,
                                      Zj(0), nullptr, c->ka, -1, dotsbuf(1), done, s, packed ? &sr0 : nullptr);
                    prev_inhead = inhead;
                }
                // halo of the vector kernel A gathers (z~ of kernel B, or Z_0): the producer sent it (peer-store),
                // or packed it (exchange here), or it is gathered first
                const double *zsrc = loc == 0 ? Zj(0) : c->zun.p;
                if (!c->peers.empty() && !prev_inhead) {
                    if (!packed) k::gather(zsrc, c->send_idx.p, c->send_off.back(), c->send_buf.p, done, s);
                    c->comm->exchange(c->send_buf.p, c->peers, c->send_off, c->xghost.p, c->recv_off, s);
                }
                k::IterA a{};
                a.browptr = c->Ab.browptr.p; a.bcol = c->Ab.bcol.p; a.vtop = c->Ab.vtop.p; a.vbot = c->Ab.vbot.p;
                a.tile_brow = c->Ab.tile_brow.p; a.ntiles = c->Ab.ntiles; a.tiles_per_xcd = (c->Ab.ntiles + 7) / 8;
                a.tdesc = reinterpret_cast<const int4 *>(c->Ab.tile_desc.p);
                a.slots = 0;  // set by the launcher
                a.od = c->n_ghost > 0 ? c->offdiag() : k::OffDiag{nullptr, nullptr, nullptr, nullptr};
                a.zsrc = zsrc;
                a.zdst = loc == 0 ? nullptr : Zj(loc);
                a.vcur = Vj(loc);
                a.w = w;
                a.acc = fused ? 1 : 0;
                a.nrm2 = loc == 0 ? nullptr : nrmbuf(loc - 1);
                a.V = V; a.ldv = ld; a.nv = loc + 1;
                a.bd = bdp; a.ldb = ld; a.m = m; a.packed = bpk;
                a.nl = nl; a.lam_in_dot = lam_in_dot;
                a.tb = c->ka.tb; a.wl_out = wl(0);
                a.partials = c->partials.p; a.out = db;
                if (!three) a.ar = c->comm->fused_allreduce(loc + 1 + m, k::kStatArDots);
                a.err = c->errw.p; a.fin_ticks = c->fin_ticks;
                a.ka = c->ka; a.loc_prev = loc - 1; a.dots_prev = dotsbuf(loc - 1); a.nrm_prev = nrmbuf(loc - 1);
                a.done = done;
                k::iter_spmv_mdot(a, s, !three);
                if (three) {
                    // h = V^T w and q = B D w (dense rows, or two halves per parity-interleaved plane) in one pass
                    const bool one = loc + 1 + m <= 40;  // beyond 40 vectors MDot is two launches: dense rows, no ride-along
                    const k::PeerAR ar = one ? c->comm->fused_allreduce(loc + 2 + m, k::kStatArDots) : k::PeerAR{};
                    const bool spl = bpk && one;
                    k::mdot(V, ld, loc + 1, w, N, n_dot, c->fin(db, ar), done, s, spl ? c->bdpk.p : c->bd.p, m, spl ? 1 : 0);
                    if (!ar.P) c->comm->allreduce_sum(db, loc + 2 + m, s);
                } else if (!a.ar.P) c->comm->allreduce_sum(db, loc + 1 + m, s);
                const k::PeerAR ar2 = c->comm->fused_allreduce(1, k::kStatArNorm);
                if (loc + 1 < mk) {
                    k::IterB b{};
                    b.V = V; b.ldv = ld; b.nv = loc + 1; b.dots = db; b.tb = c->ka.tb;
                    b.w = w; b.dinv = c->dinv.p; b.bd = bdp; b.ldb = ld; b.shat = c->shat.p; b.gram = c->gram.p;
                    b.fact = fused ? c->schur_fact : SPK_SCHUR_LOWER;
                    b.nl = nl; b.m = m; b.packed = bpk;
                    b.zun = c->zun.p; b.c = fused ? Vj(loc + 2) : nullptr; b.wl_in = wl(0);
                    b.lam_in_dot = lam_in_dot;
                    b.partials = c->partials.p; b.out = nb; b.ar = ar2; b.err = c->errw.p; b.fin_ticks = c->fin_ticks;
                    k::SendRanges sr = c->send_ranges;
                    prev_inhead = sr.n > 0 && c->comm->fused_halo(sr, c->xghost.p);
                    if (sr.n > 0) b.sr = sr;
                    b.done = done;
                    k::iter_maxpy_uhead(b, s);
                } else {
                    // last iteration of the cycle: nothing follows the update but its norm
                    k::maxpy(V, ld, loc + 1, nullptr, db, -1.0, w, N, n_dot, c->fin(nb, ar2), done, s);
                }
                if (!ar2.P) c->comm->allreduce_sum(nb, 1, s);
                last = loc;  // its Givens step rides in kernel A of the next iteration (or runs alone below)
            } else if (fused) {
                // v_j = w'/||w'|| (in place), z_j = M^-1 v_j, w = B^T z1 (u part) | B z0 (lambda part);
                // workgroup 0 also runs the Givens step of iteration loc-1
                k::SendRanges sr = c->send_ranges;
                const bool packed = sr.n > 0;   // head fills the halo buffer itself ...
                bool inhead = prev_inhead;
                if (!head_done) {
                    inhead = packed && c->comm->fused_halo(sr, c->xghost.p);   // ... or does the whole exchange
                    k::fused_head(Vj(loc), nrmbuf(loc + 1), w1side, c->dinv.p, bdp, ld, c->shat.p, c->gram.p,
                                  c->schur_fact, nl, m, Zj(loc), w, c->ka, big ? -1 : loc - 1, dotsbuf(loc + 1), done, s,
                                  packed ? &sr : nullptr, bpk);
                    last = big ? -1 : loc;
                    if (single) k::copy_small(w + nl, wl(loc), m, done, s);
                }
                // w += A z0 (halo exchange, then diagonal and off-rank columns in ONE kernel)
                const k::OffDiag od = c->offdiag();
                if (!c->peers.empty() && !inhead) {
                    if (!packed) k::gather(Zj(loc), c->send_idx.p, c->send_off.back(), c->send_buf.p, done, s);
                    c->comm->exchange(c->send_buf.p, c->peers, c->send_off, c->xghost.p, c->recv_off, s);
                }
                a_mult(c, Zj(loc), w, nullptr, nullptr, done, true, c->n_ghost > 0 ? &od : nullptr);
            } else if (fusedj) {
                bool inhead = prev_inhead;
                if (!head_done) {
                    k::SendRanges sr = c->send_ranges;
                    inhead = sr.n > 0 && c->comm->fused_halo(sr, c->xghost.p);
                    k::fused_head(Vj(loc), nrmbuf(loc + 1), nullptr, c->dinv.p, nullptr, ld, nullptr, nullptr, SPK_SCHUR_LOWER,
                                  nl, 0, Zj(loc), nullptr, c->ka, big ? -1 : loc - 1, dotsbuf(loc + 1), done, s, inhead ? &sr : nullptr);
                    last = big ? -1 : loc;
                }
                op_mult(c, Zj(loc), w, done, inhead);    // w = A z_j (halo inside, unless the head kernel did it)
            } else {
                op_pc_apply(c, Vj(loc), Zj(loc), done);  // z_j = M^-1 v_j
                op_mult(c, Zj(loc), w, done, false, true);   // w = K z_j (B^T z_1 of a general block: as PCApply left it)
            }
            if (two || ba || un3) {
                // (orthogonalisation done above, inside the launches)
            } else if (o.orthog == SPK_ORTHOG_MGS) {
                // KSPGMRESModifiedGramSchmidtOrthogonalization: one dot + one axpy per basis vector
                for (int j = 0; j <= loc; ++j) {
                    k::mdot(Vj(j), ld, 1, w, N, n_dot, c->fin(db + j), done, s);
                    c->comm->allreduce_sum(db + j, 1, s);
                    const bool lastv = j == loc;
                    k::maxpy(Vj(j), ld, 1, nullptr, db + j, -1.0, w, N, n_dot, c->fin(lastv ? nb : nullptr), done, s,
                             lastv ? bdp : nullptr, ld, nl, m, lastv && fused ? w1side : nullptr, nullptr, bpk);
                }
                c->comm->allreduce_sum(nb, nn, s);
            } else if (single) {
                const k::PeerAR ar = loc + 1 + m <= 40 ? c->comm->fused_allreduce(loc + 2 + m, k::kStatArDots) : k::PeerAR{};
                // B D w from the same pass: the dense rows, or two halves per parity-interleaved plane
                const bool spl = bpk && loc + 1 + m <= 40;
                k::mdot(V, ld, loc + 1, w, N, n_dot, c->fin(db, ar), done, s, spl ? c->bdpk.p : c->bd.p, m, spl ? 1 : 0);
                if (!ar.P) c->comm->allreduce_sum(db, loc + 2 + m, s);
                if (loc + 1 < mk) {
                    k::SendRanges sr = c->send_ranges;
                    prev_inhead = sr.n > 0 && c->comm->fused_halo(sr, c->xghost.p);
                    // Schur: the packed halo buffer is filled even without the peer backend; Jacobi: op_mult gathers
                    const k::SendRanges *srp = sr.n > 0 && (fused || prev_inhead) ? &sr : nullptr;
                    k::maxpy_head(V, ld, loc + 1, db, c->ka.tb, nb, w, c->dinv.p, bdp, ld, c->shat.p, c->gram.p,
                                  fused ? c->schur_fact : SPK_SCHUR_LOWER, nl, m, Zj(loc + 1), fused ? Vj(loc + 2) : nullptr,
                                  w1side, wl(loc), wl(loc + 1), c->ka, loc, done, s, srp, bpk);
                    head_done = true;
                    last = -1;  // its Givens step is done
                } else {
                    k::PythArgs py{m, db, c->ka.tb, nb};
                    k::maxpy(V, ld, loc + 1, nullptr, db, -1.0, w, N, n_dot, c->fin(nullptr), done, s, nullptr, ld, nl, m,
                             w1side, &py);
                    head_done = false;
                    last = loc;
                }
            } else if (big) {
                // classical Gram-Schmidt in chunks: every inner product is taken with the SAME w before any update
                for (int v0 = 0; v0 <= loc; v0 += 40) {
                    const int cnt = std::min(40, loc + 1 - v0);
                    k::mdot(Vj(v0), ld, cnt, w, N, n_dot, c->fin(db + v0), done, s);
                }
                c->comm->allreduce_sum(db, loc + 2, s);
                for (int v0 = 0; v0 <= loc; v0 += 40) {
                    const int cnt = std::min(40, loc + 1 - v0);
                    const bool lastc = v0 + 40 > loc;
                    k::maxpy(Vj(v0), ld, cnt, nullptr, db + v0, -1.0, w, N, n_dot, c->fin(lastc ? nb : nullptr), done, s,
                             lastc ? bdp : nullptr, ld, nl, m, lastc && fused ? w1side : nullptr, nullptr, bpk);
                }
                c->comm->allreduce_sum(nb, nn, s);
                if (o.cgs_refine != SPK_REFINE_NEVER) {
                    // -ksp_gmres_cgs_refinement_type on a long restart: the second pass in the same chunks, on the device's
                    // own decision (PETSc's ||w'|| < ||h|| test for ifneeded)
                    const int32_t *skip = &c->kst.p->skip_refine;
                    double *db2 = c->bigdots.p + (size_t)mk + 4;
                    k::krylov_refine_decide(c->ka, loc, o.cgs_refine, db, nb, db2, s);
                    for (int v0 = 0; v0 <= loc; v0 += 40) {
                        const int cnt = std::min(40, loc + 1 - v0);
                        k::mdot(Vj(v0), ld, cnt, w, N, n_dot, c->fin(db2 + v0), skip, s);
                    }
                    c->comm->allreduce_sum(db2, loc + 2, s);
                    for (int v0 = 0; v0 <= loc; v0 += 40) {
                        const int cnt = std::min(40, loc + 1 - v0);
                        const bool lastc = v0 + 40 > loc;
                        k::maxpy(Vj(v0), ld, cnt, nullptr, db2 + v0, -1.0, w, N, n_dot, c->fin(lastc ? nrm2b : nullptr), skip, s,
                                 lastc ? bdp : nullptr, ld, nl, m, lastc && fused ? w1side : nullptr, nullptr, bpk);
                    }
                    c->comm->allreduce_sum(nrm2b, nn, s);
                    k::krylov_refine_merge(c->ka, loc, db, db2, nb, nrm2b, nn, s);
                }
            } else {
                // classical Gram-Schmidt: h = V^T w (one pass), w -= V h (+ ||w||^2 [+ B D w'] in the same pass)
                // across ranks the all-reduces ride in the finish of the two kernels (peer-store backend)
                const k::PeerAR ar1 = loc + 1 <= 40 ? c->comm->fused_allreduce(loc + 2, k::kStatArDots) : k::PeerAR{};
                k::mdot(V, ld, loc + 1, w, N, n_dot, c->fin(db, ar1), done, s);
                if (!ar1.P) c->comm->allreduce_sum(db, loc + 2, s);
                const k::PeerAR ar2 = c->comm->fused_allreduce(nn, k::kStatArNorm);
                k::maxpy(V, ld, loc + 1, nullptr, db, -1.0, w, N, n_dot, c->fin(nb, ar2), done, s, bdp, ld, nl, m,
                         fused ? w1side : nullptr, nullptr, bpk);
                if (!ar2.P) c->comm->allreduce_sum(nb, nn, s);
                if (o.cgs_refine != SPK_REFINE_NEVER) {
                    // second pass on the device's own decision (-ksp_gmres_cgs_refinement_type)
                    const int32_t *skip = &c->kst.p->skip_refine;
                    k::krylov_refine_decide(c->ka, loc, o.cgs_refine, db, nb, sm2, s);
                    k::mdot(V, ld, loc + 1, w, N, n_dot, c->fin(sm2), skip, s);
                    c->comm->allreduce_sum(sm2, loc + 2, s);
                    k::maxpy(V, ld, loc + 1, nullptr, sm2, -1.0, w, N, n_dot, c->fin(nrm2b), skip, s, bdp, ld, nl, m,
                             fused ? w1side : nullptr, nullptr, bpk);
                    c->comm->allreduce_sum(nrm2b, nn, s);
                    k::krylov_refine_merge(c->ka, loc, db, sm2, nb, nrm2b, nn, s);
                }
            }
            if (!head || big) {
                // Hessenberg column, Givens, convergence -- on the device; then v_{j+1} = w / ||w|| (the head kernel of the
                // next iteration does that scaling where there is one)
                k::krylov_givens(c->ka, loc, db, nb, s);
                if (!head) k::scale_dev(w, N, inv_tt, done, s);
            }
            if (o.check_every > 0 && (loc + 1) % o.check_every == 0 && loc + 1 < mk) {
                SPK_HIP(hipMemcpyAsync(&st, c->kst.p, sizeof st, hipMemcpyDeviceToHost, s));
                SPK_HIP(hipStreamSynchronize(s));
                stop = st.done != 0 || st.skip_iter != 0;  // fused path: lags by one iteration, the iterate does not care
            }
        }
        if (finished || (pend_check && read_state())) break;   // (second form: cycles shorter than kAhead iterations)
        // fused path: the Givens step of the cycle's last iteration has no head kernel to ride on
        if (head && last >= 0) pend_h = dotsbuf(last), pend_n = nrmbuf(last), pend_loc = last;
        // ---- x += Z y (KSPFGMRESBuildSoln); always runs, count comes from the device ----
        {
            k::GivensRider pend{c->ka, pend_loc, pend_h, pend_n, nullptr, nullptr, 0, k::FinErr{nullptr, 0}, k::PeerAR{}};
            k::krylov_cycle_end(c->ka, s, (ba || un3) ? c->ba_sc.p : nullptr, mk, pend_loc >= 0 ? &pend : nullptr);
        }
        k::maxpy(Z, ld, mk, loc_done, c->ka.nrs, 1.0, x, N, 0, c->fin(nullptr), nullptr, s);
        // ---- true residual for the next cycle (KSPFGMRESResidual); skipped once done ----
        op_mult(c, x, c->tmp.p, done);
        // r = b - K x into V0 together with ||r||^2 (and B D r) for the start of the next cycle
        if (fused) k::sqnorm_bd(Vj(0), N, n_dot, c->bd.p, ld, nl, m, w1side, c->fin(nrmbuf(1)), done, s, b, c->tmp.p);
        else k::sqnorm_sub(b, c->tmp.p, Vj(0), N, n_dot, c->fin(nrmbuf(1)), done, s);
        ++cycles;
        SPK_HIP(hipGetLastError());  // a rejected launch inside the cycle surfaces here, not as a wrong answer
    }
    SPK_HIP(hipStreamSynchronize(s));  // (a speculative start of a cycle that will not run drains as no-ops)
    c->comm->check(s);          // what was raised after the last report (once per solve: blocking reads)
    c->check_device_error();
    const auto t1 = std::chrono::steady_clock::now();

    res->its = st.its;
    res->reason = st.reason;
    res->rnorm = st.rnorm;
    res->rnorm0 = st.rnorm0;
    res->cycles = cycles;
    res->solve_seconds = std::chrono::duration<double>(t1 - t0).count();
    int32_t nh = std::min<int32_t>(st.its + 1, c->ka.hist_cap);
    if (!history) nh = 0;
    nh = std::min(nh, history_cap);
    if (nh > 0) SPK_HIP(hipMemcpy(history, c->ka.hist, sizeof(double) * (size_t)nh, hipMemcpyDeviceToHost));
    res->hist_len = nh;
}

}  // namespace spk
