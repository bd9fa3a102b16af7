// spk_api.cpp -- the extern "C" surface declared in include/spk.h.
// Argument checks, error capture (exceptions never cross the ABI), staging of
// host vectors.  The work is in spk_solver.cpp / spk_k_*.hip.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>

#include "spk_internal.hpp"

namespace spk {
void rccl_unique_id(void *id128);
}

static thread_local std::string g_create_error;

#define SPK_TRY(ctx)                                                     \
    if (!(ctx)) return SPK_ERR_ARG;                                      \
    try {                                                                \
        if (hipSetDevice((ctx)->device) != hipSuccess)                   \
            spk::fail(SPK_ERR_HIP, "hipSetDevice(%d) failed", (ctx)->device);

#define SPK_CATCH(ctx)                                                   \
    }                                                                    \
    catch (const spk::Error &e)                                          \
    {                                                                    \
        (ctx)->err = e.msg;                                              \
        return e.code;                                                   \
    }                                                                    \
    catch (const std::exception &e)                                      \
    {                                                                    \
        (ctx)->err = e.what();                                           \
        return SPK_ERR_NOMEM;                                            \
    }                                                                    \
    return SPK_OK;

extern "C" {

int spk_version(void) { return SPK_VERSION; }

void spk_default_opts(spk_opts *o)
{
    if (!o) return;
    std::memset(o, 0, sizeof *o);
    o->restart = 30;
    o->max_it = 10000;
    o->rtol = 1e-5;
    o->abstol = 1e-50;
    o->dtol = 1e4;
    o->guess_nonzero = 0;
    o->orthog = SPK_ORTHOG_CGS;
    o->check_every = 0;
    o->fused = 1;
}

int spk_create(spk_ctx **out, int device)
{
    if (!out) return SPK_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_error = std::string("spk_create: no HIP device available (") + hipGetErrorString(e) +
                         "); libspk has no CPU fallback";
        return SPK_ERR_HIP;
    }
    if (device < 0 || device >= ndev) {
        g_create_error = "spk_create: device index out of range";
        return SPK_ERR_ARG;
    }
    spk_ctx *c = nullptr;
    try {
        c = new spk_ctx();
        c->device = device;
        SPK_HIP(hipSetDevice(device));
        SPK_HIP(hipStreamCreate(&c->stream));
        SPK_HIP(hipDeviceGetAttribute(&c->num_cus, hipDeviceAttributeMultiprocessorCount, device));
        // the solver's state report (pinned block + event): here, not inside the first solve
        SPK_HIP(hipHostMalloc(&c->pin_state, 512, hipHostMallocDefault));
        SPK_HIP(hipEventCreateWithFlags(&c->state_ev, hipEventDisableTiming));
        c->comm.reset(spk::make_self_comm());
        c->ensure_scratch();
    } catch (const spk::Error &er) {
        g_create_error = er.msg;
        delete c;
        return er.code;
    } catch (const std::exception &er) {
        g_create_error = er.what();
        delete c;
        return SPK_ERR_NOMEM;
    }
    *out = c;
    return SPK_OK;
}

int spk_destroy(spk_ctx *c)
{
    if (!c) return SPK_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) {
        (void)hipStreamSynchronize(c->stream);
    }
    c->comm.reset();
    hipStream_t s = c->stream;
    for (int i = 0; i < 2; ++i) {
        if (c->pin[i]) (void)hipHostFree(c->pin[i]);
        if (i == 0 && c->pin_state) (void)hipHostFree(c->pin_state);
        if (i == 0 && c->state_ev) (void)hipEventDestroy(c->state_ev);
        if (c->pin_ev[i]) (void)hipEventDestroy(c->pin_ev[i]);
    }
    for (hipEvent_t e : c->tp_ev) (void)hipEventDestroy(e);
    delete c;  // DevBuf destructors free device memory
    if (s) (void)hipStreamDestroy(s);
    return SPK_OK;
}

const char *spk_last_error(const spk_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int spk_comm_unique_id(void *id128)
{
    if (!id128) return SPK_ERR_ARG;
    try {
        spk::rccl_unique_id(id128);
    } catch (const spk::Error &e) {
        g_create_error = e.msg;
        return e.code;
    }
    return SPK_OK;
}

int spk_comm_init_rccl(spk_ctx *c, int rank, int nranks, const void *id128)
{
    SPK_TRY(c)
    if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) spk::fail(SPK_ERR_ARG, "comm_init_rccl: bad rank/nranks");
    if (c->have_A) spk::fail(SPK_ERR_STATE, "comm_init: must precede spk_set_block");
    c->comm.reset(spk::make_rccl_comm(rank, nranks, id128, c->device));
    SPK_CATCH(c)
}

int spk_comm_init_host(spk_ctx *c, int rank, int nranks, const spk_host_comm *cb)
{
    SPK_TRY(c)
    if (!cb || !cb->allreduce || !cb->exchange || !cb->allgather || nranks < 1 || rank < 0 || rank >= nranks)
        spk::fail(SPK_ERR_ARG, "comm_init_host: bad arguments");
    if (c->have_A) spk::fail(SPK_ERR_STATE, "comm_init: must precede spk_set_block");
    c->comm.reset(spk::make_host_comm(rank, nranks, *cb));
    SPK_CATCH(c)
}

int spk_comm_init_local(spk_ctx *c, spk_local_group *grp, int rank)
{
    SPK_TRY(c)
    if (!grp) spk::fail(SPK_ERR_ARG, "comm_init_local: null group");
    if (c->have_A) spk::fail(SPK_ERR_STATE, "comm_init: must precede spk_set_block");
    c->comm.reset(spk::make_local_comm(grp, rank));
    SPK_CATCH(c)
}

int spk_comm_enable_peer(spk_ctx *c, int32_t *enabled)
{
    SPK_TRY(c)
    if (enabled) *enabled = 0;
    if (c->have_A) spk::fail(SPK_ERR_STATE, "comm_enable_peer: must precede spk_set_block");
    const char *off = getenv("SPK_COMM_PEER");
    if (!(off && !strcmp(off, "0")) && c->comm->size() > 1 && strcmp(c->comm->name(), "peer-store") != 0) {
        std::string why;
        spk::Comm *inner = c->comm.release();
        c->comm.reset(spk::make_peer_comm(inner, c->device, &why));  // never throws: returns `inner` when it cannot
        c->err = why;  // informational when the backend stayed off
        c->peer_why = why;
    } else if (off && !strcmp(off, "0")) {
        c->peer_why = "switched off (SPK_COMM_PEER=0)";
    }
    if (enabled) *enabled = strcmp(c->comm->name(), "peer-store") == 0;
    SPK_CATCH(c)
}

const char *spk_comm_backend(const spk_ctx *c) { return c && c->comm ? c->comm->name() : "self"; }

int spk_comm_get_info(spk_ctx *c, spk_comm_info *o)
{
    SPK_TRY(c)
    if (!o) spk::fail(SPK_ERR_ARG, "spk_comm_get_info: null output");
    std::memset(o, 0, sizeof *o);
    o->rank = c->comm->rank();
    o->nranks = c->comm->size();
    o->device = c->device;
    o->window_tier = -1;
    o->halo_mode = c->peers.empty() ? 0 : 3;
    std::snprintf(o->backend, sizeof o->backend, "%s", c->comm->name());
    std::snprintf(o->inner_backend, sizeof o->inner_backend, "%s", c->comm->name());
    std::snprintf(o->why, sizeof o->why, "%s", c->peer_why.c_str());
    c->comm->info(o);  // the peer-store backend fills in the rest
    SPK_CATCH(c)
}

/* Test hook: a P-rank peer-store all-reduce played by P workgroups of ONE launch through P windows of this
 * process (no IPC): checks the window layout for every lane up to kPeerMax = 8 on one device.
 * vals: P x count inputs (row r = rank r's contribution); out: P x count results (every row must hold the
 * same rank-ordered sums). */
int spk_debug_peer_allreduce_loopback(spk_ctx *c, int nranks, int count, int rounds, const double *vals, double *out)
{
    SPK_TRY(c)
    if (!vals || !out || nranks < 1 || nranks > spk::k::kPeerMax || count < 1 || count > 64 || rounds < 1)
        spk::fail(SPK_ERR_ARG, "spk_debug_peer_allreduce_loopback: bad arguments");
    const size_t wbytes = sizeof(unsigned long long) * (size_t)spk::k::kArSlots * nranks * spk::k::kArGranules;
    std::vector<unsigned long long *> win((size_t)nranks, nullptr);
    spk::DevBuf<double> buf;
    spk::DevBuf<int32_t> err;
    buf.alloc((size_t)nranks * 64);
    err.alloc(4);
    auto cleanup = [&]() { for (auto *w : win) if (w) (void)hipFree(w); };
    try {
        for (int r = 0; r < nranks; ++r) {
            if (hipExtMallocWithFlags((void **)&win[(size_t)r], wbytes, hipDeviceMallocUncached) != hipSuccess) {
                (void)hipGetLastError();
                SPK_HIP(hipMalloc((void **)&win[(size_t)r], wbytes));
            }
            SPK_HIP(hipMemset(win[(size_t)r], 0, wbytes));
        }
        std::vector<double> h((size_t)nranks * 64, 0.0);
        for (int round = 0; round < rounds; ++round) {   // walks through every slot, and the wrap
            for (int r = 0; r < nranks; ++r)
                for (int i = 0; i < count; ++i) h[(size_t)r * 64 + i] = vals[(size_t)r * count + i] * (1.0 + round);
            SPK_HIP(hipMemcpy(buf.p, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
            spk::k::PeerAR a{};
            a.P = nranks;
            a.seq = (uint32_t)(round + 1);
            a.timeout_ms = 3000;
            for (int r = 0; r < nranks; ++r) a.win[r] = win[(size_t)r];
            a.err = err.p;
            spk::k::peer_allreduce_loopback(a, buf.p, count, c->stream);
            SPK_HIP(hipStreamSynchronize(c->stream));
            int32_t e = 0;
            SPK_HIP(hipMemcpy(&e, err.p, sizeof e, hipMemcpyDeviceToHost));
            if (e) spk::fail(SPK_ERR_COMM, "loop-back all-reduce timed out in round %d", round);
            SPK_HIP(hipMemcpy(h.data(), buf.p, h.size() * sizeof(double), hipMemcpyDeviceToHost));
            for (int r = 0; r < nranks; ++r)
                for (int i = 0; i < count; ++i) {
                    const double v = h[(size_t)r * 64 + i] / (1.0 + round);
                    if (round == 0) out[(size_t)r * count + i] = v;
                    else if ((round == 1 && out[(size_t)r * count + i] != v) ||   // x2 is exact: the same bits
                             std::fabs(out[(size_t)r * count + i] - v) > 1e-13 * std::fabs(v) + 1e-300)
                        spk::fail(SPK_ERR_COMM, "loop-back all-reduce: round %d differs from round 0", round);
                }
        }
    } catch (...) {
        cleanup();
        throw;
    }
    cleanup();
    SPK_CATCH(c)
}

int spk_set_block(spk_ctx *c, int which, int64_t row_begin, int32_t nrows_local, int64_t ncols_global,
                  const int32_t *rowptr, const int32_t *colidx, const double *val)
{
    SPK_TRY(c)
    spk::set_block(c, which, row_begin, nrows_local, ncols_global, rowptr, colidx, val);
    SPK_CATCH(c)
}

int spk_pc_setup(spk_ctx *c, int pc_type, int schur_fact)
{
    SPK_TRY(c)
    spk::pc_setup(c, pc_type, schur_fact);
    SPK_CATCH(c)
}

int spk_pc_set_inner(spk_ctx *c, int sweeps, double omega)
{
    SPK_TRY(c)
    if (sweeps < 0 || sweeps > 64 || !(omega > 0.0) || !(omega < 2.0)) spk::fail(SPK_ERR_ARG, "pc_set_inner: sweeps in [0,64], omega in (0,2)");
    c->inner_sweeps = sweeps;
    c->inner_omega = omega;
    c->pc_ready = false;
    SPK_CATCH(c)
}

int spk_get_schur_diag(spk_ctx *c, double *shat)
{
    SPK_TRY(c)
    if (!shat) spk::fail(SPK_ERR_ARG, "null output");
    if (!c->pc_ready || c->m == 0 || !c->shat.p) spk::fail(SPK_ERR_STATE, "no Schur data: call spk_pc_setup with the A10 block set");
    SPK_HIP(hipMemcpy(shat, c->shat.p, sizeof(double) * (size_t)c->m, hipMemcpyDeviceToHost));
    SPK_CATCH(c)
}

int spk_get_jacobi_diag(spk_ctx *c, double *dinv)
{
    SPK_TRY(c)
    if (!dinv) spk::fail(SPK_ERR_ARG, "null output");
    if (!c->pc_ready) spk::fail(SPK_ERR_STATE, "call spk_pc_setup first");
    SPK_HIP(hipMemcpy(dinv, c->dinv.p, sizeof(double) * (size_t)c->n_local, hipMemcpyDeviceToHost));
    SPK_CATCH(c)
}

int spk_get_bd_planes(const spk_ctx *c, int32_t *planes)
{
    if (!c || !planes) return SPK_ERR_ARG;
    *planes = !c->bd.p ? 0 : (c->bd_packed ? c->m / 2 : c->m);
    return SPK_OK;
}

int spk_get_sizes(const spk_ctx *c, int64_t *n_global, int32_t *n_local, int32_t *m, int64_t *nnz_local,
                  int32_t *n_ghost)
{
    if (!c) return SPK_ERR_ARG;
    if (n_global) *n_global = c->n_global;
    if (n_local) *n_local = c->n_local;
    if (m) *m = c->m;
    if (nnz_local) *nnz_local = c->Ad.nnz + c->Ao.nnz;
    if (n_ghost) *n_ghost = c->n_ghost;
    return SPK_OK;
}

int spk_get_spmv_info(const spk_ctx *c, int32_t *format, int64_t *layout_bytes)
{
    if (!c) return SPK_ERR_ARG;
    const bool dict = c->spmv_format != 0 && c->Adict.ok;
    if (format) *format = dict ? 2 + c->spmv_format : c->spmv_format;
    if (layout_bytes) {
        const int64_t n = c->n_local;
        *layout_bytes = dict                  ? c->Adict.code_bytes + 2 * (int64_t)c->Adict.nbrows + c->Adict.lds_bytes + 16 * n
                        : c->spmv_format == 1 ? 36 * c->Ab.nblocks + 4 * ((int64_t)c->Ab.nbrows + 1) + 16 * n
                        : c->spmv_format == 2 ? 76 * c->Ab3.nblocks + 4 * ((int64_t)c->Ab3.nbrows + 1) + 16 * n
                                              : 12 * c->Ad.nnz + 4 * (n + 1) + 16 * n;
    }
    return SPK_OK;
}

int spk_get_iteration_form(const spk_ctx *c, int32_t *form, int32_t *single_reduce)
{
    if (!c) return SPK_ERR_ARG;
    if (form) *form = c->last_form;
    if (single_reduce) *single_reduce = c->last_single;
    return SPK_OK;
}

int spk_get_spmv_models(const spk_ctx *c, int64_t *csr_bytes, int64_t *blocked_bytes, int64_t *dict_bytes, int32_t *npat,
                        int32_t *nblk)
{
    if (!c) return SPK_ERR_ARG;
    const int64_t n = c->n_local;
    if (csr_bytes) *csr_bytes = 12 * c->Ad.nnz + 4 * (n + 1) + 16 * n;
    if (blocked_bytes)
        *blocked_bytes = c->Ab.ok ? 36 * c->Ab.nblocks + 4 * ((int64_t)c->Ab.nbrows + 1) + 16 * n
                         : c->Ab3.ok ? 76 * c->Ab3.nblocks + 4 * ((int64_t)c->Ab3.nbrows + 1) + 16 * n : 0;
    if (dict_bytes) *dict_bytes = c->Adict.ok ? c->Adict.code_bytes + 2 * (int64_t)c->Adict.nbrows + c->Adict.lds_bytes + 16 * n : 0;
    if (npat) *npat = c->Adict.ok ? c->Adict.ntype : 0;
    if (nblk) *nblk = c->Adict.ok ? c->Adict.nclass : 0;
    return SPK_OK;
}

// stage a host vector into a zero-padded device buffer / pass a device pointer through
static const double *stage_in(spk_ctx *c, const double *p, int mem, spk::DevBuf<double> &buf, int64_t n)
{
    if (mem == SPK_MEM_DEVICE) return p;
    SPK_HIP(hipMemcpyAsync(buf.p, p, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    return buf.p;
}

int spk_mult(spk_ctx *c, const double *x, double *y, int mem)
{
    SPK_TRY(c)
    if (!x || !y) spk::fail(SPK_ERR_ARG, "spk_mult: null vector");
    if (!c->have_A) spk::fail(SPK_ERR_STATE, "spk_mult: no operator");
    c->ensure_vectors();
    const int64_t N = (int64_t)c->n_local + c->m;
    const double *xd = stage_in(c, x, mem, c->stage_x, N);
    double *yd = mem == SPK_MEM_DEVICE ? y : c->stage_y.p;
    spk::op_mult(c, xd, yd, nullptr);
    SPK_HIP(hipGetLastError());
    if (mem != SPK_MEM_DEVICE) SPK_HIP(hipMemcpyAsync(y, yd, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost, c->stream));
    SPK_HIP(hipStreamSynchronize(c->stream));
    c->comm->check(c->stream);
    c->check_device_error();
    SPK_CATCH(c)
}

int spk_pc_apply(spk_ctx *c, const double *x, double *y, int mem)
{
    SPK_TRY(c)
    if (!x || !y) spk::fail(SPK_ERR_ARG, "spk_pc_apply: null vector");
    if (!c->pc_ready) spk::fail(SPK_ERR_STATE, "spk_pc_apply: call spk_pc_setup first");
    const int64_t N = (int64_t)c->n_local + c->m;
    const double *xd = stage_in(c, x, mem, c->stage_x, N);
    double *yd = mem == SPK_MEM_DEVICE ? y : c->stage_y.p;
    spk::op_pc_apply(c, xd, yd, nullptr);
    SPK_HIP(hipGetLastError());
    if (mem != SPK_MEM_DEVICE) SPK_HIP(hipMemcpyAsync(y, yd, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost, c->stream));
    SPK_HIP(hipStreamSynchronize(c->stream));
    c->comm->check(c->stream);
    c->check_device_error();
    SPK_CATCH(c)
}

int spk_fgmres(spk_ctx *c, const double *b, double *x, int mem, const spk_opts *opts, spk_result *result,
               double *history, int32_t history_cap)
{
    SPK_TRY(c)
    if (!b || !x || !opts || !result) spk::fail(SPK_ERR_ARG, "spk_fgmres: null argument");
    if (!c->have_A) spk::fail(SPK_ERR_STATE, "spk_fgmres: no operator");
    if (!(opts->rtol >= 0) || !(opts->abstol >= 0) || !(opts->dtol > 0) || opts->max_it < 0)
        spk::fail(SPK_ERR_ARG, "spk_fgmres: tolerances must be non-negative, max_it >= 0");
    c->ensure_vectors();
    const int64_t N = (int64_t)c->n_local + c->m;
    std::memset(result, 0, sizeof *result);
    if (mem == SPK_MEM_DEVICE) {
        spk::fgmres(c, b, x, *opts, result, history, history_cap);
    } else {
        spk_opts o = *opts;
        double *xs = c->xsol.p, *rh = c->rhs.p;
        SPK_HIP(hipMemcpy(rh, b, sizeof(double) * (size_t)N, hipMemcpyHostToDevice));
        if (o.guess_nonzero) SPK_HIP(hipMemcpy(xs, x, sizeof(double) * (size_t)N, hipMemcpyHostToDevice));
        spk::fgmres(c, rh, xs, o, result, history, history_cap);
        SPK_HIP(hipMemcpy(x, xs, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost));
    }
    SPK_CATCH(c)
}

int spk_debug_finish_timeout(spk_ctx *c, int timeout_ms)
{
    SPK_TRY(c)
    if (timeout_ms < 1 || timeout_ms > 4000) spk::fail(SPK_ERR_ARG, "spk_debug_finish_timeout: 1..4000 ms");
    c->ensure_scratch();
    const uint32_t keep = c->fin_ticks;
    c->fin_ticks = (uint32_t)timeout_ms * 100000u;
    spk::k::finish_probe(c->fin(c->small.p + 500), c->stream);
    c->fin_ticks = keep;
    SPK_HIP(hipStreamSynchronize(c->stream));
    c->check_device_error();  // throws SPK_ERR_HIP: that is the expected outcome
    SPK_CATCH(c)
}

int spk_debug_set_wait_bound(spk_ctx *c, uint32_t ticks)
{
    if (!c) return SPK_ERR_ARG;
    c->fin_ticks = ticks ? ticks : 400000000u;
    return SPK_OK;
}

int spk_debug_time_products(spk_ctx *c, int32_t max_launches)
{
    SPK_TRY(c)
    SPK_HIP(hipStreamSynchronize(c->stream));
    c->tp_used = 0;
    c->time_products = max_launches > 0;
    const size_t want = max_launches > 0 ? 2 * (size_t)max_launches : 0;
    while (c->tp_ev.size() < want) {
        hipEvent_t e;
        SPK_HIP(hipEventCreate(&e));
        c->tp_ev.push_back(e);
    }
    SPK_CATCH(c)
}

int spk_get_product_timing(spk_ctx *c, int32_t *launches, double *mean_ms, double *median_ms, double *min_ms, double *max_ms,
                           int32_t *gated, double *gated_mean_ms)
{
    SPK_TRY(c)
    SPK_HIP(hipStreamSynchronize(c->stream));
    std::vector<float> t;
    for (size_t i = 0; i + 1 < c->tp_used; i += 2) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, c->tp_ev[i], c->tp_ev[i + 1]) != hipSuccess) {   // (a launch that took no stamps: an
            (void)hipGetLastError();                                                   // empty rank runs the rider alone)
            continue;
        }
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    // (launches the device had gated off -- the iterations the host enqueues ahead of a solve's end return at once -- are
    // not the kernel at work: anything shorter than half the median is left out of the mean and the count)
    const double med = t.empty() ? 0.0 : t[t.size() / 2];
    double sum = 0.0;
    size_t used = 0;
    double lo = 0.0;
    for (float v : t)
        if (v >= 0.5 * med) {
            if (!used) lo = v;
            sum += v;
            ++used;
        }
    double gsum = 0.0;
    size_t ng = 0;
    for (float v : t)
        if (v < 0.5 * med) {
            gsum += v;
            ++ng;
        }
    if (gated) *gated = (int32_t)ng;
    if (gated_mean_ms) *gated_mean_ms = ng ? gsum / (double)ng : 0.0;
    if (launches) *launches = (int32_t)used;
    if (mean_ms) *mean_ms = used ? sum / (double)used : 0.0;
    if (median_ms) *median_ms = med;
    if (min_ms) *min_ms = lo;
    if (max_ms) *max_ms = t.empty() ? 0.0 : t.back();
    SPK_CATCH(c)
}

int spk_vec_create(spk_ctx *c, int64_t n, double **dev)
{
    SPK_TRY(c)
    if (!dev || n < 0) spk::fail(SPK_ERR_ARG, "spk_vec_create: bad arguments");
    const size_t len = (size_t)((n + 255) / 256 * 256 + 256);
    SPK_HIP(hipMalloc((void **)dev, len * sizeof(double)));
    SPK_HIP(hipMemset(*dev, 0, len * sizeof(double)));
    SPK_CATCH(c)
}

int spk_vec_destroy(spk_ctx *c, double *dev)
{
    SPK_TRY(c)
    if (dev) SPK_HIP(hipFree(dev));
    SPK_CATCH(c)
}

int spk_vec_set(spk_ctx *c, double *dev, const double *host, int64_t n)
{
    SPK_TRY(c)
    if (!dev || !host || n < 0) spk::fail(SPK_ERR_ARG, "spk_vec_set: bad arguments");
    SPK_HIP(hipMemcpy(dev, host, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    SPK_CATCH(c)
}

int spk_vec_get(spk_ctx *c, const double *dev, double *host, int64_t n)
{
    SPK_TRY(c)
    if (!dev || !host || n < 0) spk::fail(SPK_ERR_ARG, "spk_vec_get: bad arguments");
    SPK_HIP(hipStreamSynchronize(c->stream));
    SPK_HIP(hipMemcpy(host, dev, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    SPK_CATCH(c)
}

// ---- single kernels through the ABI -----------------------------------------
int spk_kernel_mdot(spk_ctx *c, int64_t n, int32_t nv, const double *V, int64_t ldv, const double *w, double *h)
{
    SPK_TRY(c)
    if (!V || !w || !h || n <= 0 || nv < 0 || nv > spk::k::kMaxNv - 1 || ldv < n)
        spk::fail(SPK_ERR_ARG, "spk_kernel_mdot: bad arguments");
    c->ensure_scratch();
    const int64_t ld = (n + 255) / 256 * 256;
    spk::DevBuf<double> dV, dw;
    dV.alloc((size_t)ld * (size_t)std::max(nv, 1));
    dw.alloc((size_t)ld);
    for (int i = 0; i < nv; ++i)
        SPK_HIP(hipMemcpy(dV.p + (size_t)ld * i, V + (size_t)ldv * i, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    SPK_HIP(hipMemcpy(dw.p, w, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    spk::k::mdot(dV.p, ld, nv, dw.p, n, n, c->fin(c->small.p), nullptr, c->stream);
    SPK_HIP(hipStreamSynchronize(c->stream));
    SPK_HIP(hipMemcpy(h, c->small.p, sizeof(double) * (size_t)(nv + 1), hipMemcpyDeviceToHost));
    SPK_CATCH(c)
}

int spk_kernel_maxpy(spk_ctx *c, int64_t n, int32_t nv, const double *a, const double *V, int64_t ldv,
                     double *w, double *nrm2)
{
    SPK_TRY(c)
    if (!V || !w || !a || n <= 0 || nv < 0 || nv > spk::k::kMaxNv - 1 || ldv < n)
        spk::fail(SPK_ERR_ARG, "spk_kernel_maxpy: bad arguments");
    c->ensure_scratch();
    const int64_t ld = (n + 255) / 256 * 256;
    spk::DevBuf<double> dV, dw, da;
    dV.alloc((size_t)ld * (size_t)std::max(nv, 1));
    dw.alloc((size_t)ld);
    da.upload(a, (size_t)nv, 8);
    for (int i = 0; i < nv; ++i)
        SPK_HIP(hipMemcpy(dV.p + (size_t)ld * i, V + (size_t)ldv * i, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    SPK_HIP(hipMemcpy(dw.p, w, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    spk::k::maxpy(dV.p, ld, nv, nullptr, da.p, 1.0, dw.p, n, n, c->fin(c->small.p), nullptr, c->stream);
    SPK_HIP(hipStreamSynchronize(c->stream));
    SPK_HIP(hipMemcpy(w, dw.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    if (nrm2) SPK_HIP(hipMemcpy(nrm2, c->small.p, sizeof(double), hipMemcpyDeviceToHost));
    SPK_CATCH(c)
}

int spk_time_spmv(spk_ctx *c, int warmup, int reps, double *ms_per_launch)
{
    SPK_TRY(c)
    if (!ms_per_launch || reps < 1 || warmup < 0) spk::fail(SPK_ERR_ARG, "spk_time_spmv: bad arguments");
    if (!c->have_A) spk::fail(SPK_ERR_STATE, "spk_time_spmv: no operator");
    c->ensure_vectors();
    const int64_t N = (int64_t)c->n_local + c->m;
    std::vector<double> hx((size_t)N);
    for (int64_t i = 0; i < N; ++i) hx[(size_t)i] = std::sin(0.37 * (double)(c->row_begin + i));
    SPK_HIP(hipMemcpy(c->stage_x.p, hx.data(), sizeof(double) * (size_t)N, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    SPK_HIP(hipEventCreate(&e0));
    SPK_HIP(hipEventCreate(&e1));
    auto one = [&]() {  // the kernel the solver launches for the A block
        spk::a_mult(c, c->stage_x.p, c->stage_y.p, nullptr, nullptr, nullptr, false, nullptr);
    };
    for (int i = 0; i < warmup; ++i) one();
    SPK_HIP(hipEventRecord(e0, c->stream));
    for (int i = 0; i < reps; ++i) one();
    SPK_HIP(hipEventRecord(e1, c->stream));
    SPK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    SPK_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_launch = (double)ms / reps;
    SPK_CATCH(c)
}

int spk_time_kernel(spk_ctx *c, const char *which, int nv, int warmup, int reps, double *ms_per_launch)
{
    SPK_TRY(c)
    if (!which || !ms_per_launch || reps < 1 || warmup < 0) spk::fail(SPK_ERR_ARG, "spk_time_kernel: bad arguments");
    if (!c->have_A) spk::fail(SPK_ERR_STATE, "spk_time_kernel: no operator");
    if (nv < 0 || nv > spk::k::kMaxNv - 2) spk::fail(SPK_ERR_ARG, "spk_time_kernel: nv out of range");
    c->ensure_scratch();
    c->ensure_vectors();
    const std::string w(which);
    const int64_t N = (int64_t)c->n_local + c->m, ld = c->ld;
    spk::DevBuf<double> V, coef;
    V.alloc((size_t)ld * (size_t)(nv + 2));
    {
        std::vector<double> h((size_t)ld, 0.0);
        for (int j = 0; j < nv + 2; ++j) {
            for (int64_t i = 0; i < N; ++i) h[(size_t)i] = std::sin(0.37 * (double)i + (double)j) * 1e-3;
            SPK_HIP(hipMemcpy(V.p + (size_t)ld * j, h.data(), sizeof(double) * (size_t)ld, hipMemcpyHostToDevice));
        }
        std::vector<double> a((size_t)nv + 8, 1e-6);
        coef.upload(a.data(), a.size(), 8);
    }
    double *x = V.p + (size_t)ld * nv, *y = V.p + (size_t)ld * (nv + 1);
    hipStream_t s = c->stream;
    auto run = [&]() {
        if (w == "spmv") spk::k::spmv(c->Ad, x, y, nullptr, nullptr, nullptr, s);
        else if (w == "spmv_bcsr3") { if (!c->Ab3.ok) spk::fail(SPK_ERR_STATE, "no 3x3-blocked copy"); spk::k::spmv_bcsr3(c->Ab3, x, y, nullptr, nullptr, nullptr, s); }
        else if (w == "spmv_dict") { if (!c->Adict.ok) spk::fail(SPK_ERR_STATE, "no row-type layout"); spk::k::spmv_dict(c->Adict, x, y, nullptr, nullptr, nullptr, s); }
        else if (w == "spmv_bcsr") { if (!c->Ab.ok) spk::fail(SPK_ERR_STATE, "no 2x2-blocked copy"); spk::k::spmv_bcsr(c->Ab, x, y, nullptr, nullptr, nullptr, s); }
        else if (w == "spmv_gated") {
            // the iteration's product launch as the device gates it off (`done` set: every workgroup returns at once)
            spk::k::GivensRider gr{c->ka, 0, c->small.p, c->small.p + 64, nullptr, nullptr, 0, spk::k::FinErr{nullptr, 0}, spk::k::PeerAR{}};
            if (!c->kst.p) spk::fail(SPK_ERR_STATE, "spk_time_kernel: spmv_gated needs the state of a solve");
            spk::a_mult(c, x, y, nullptr, nullptr, &c->kst.p->done, true, nullptr, &gr);
        }
        else if (w == "spmv_acc" || w == "spmv_ride") {
            // y += A x (spmv_ride: y = A x) in the active format, as the default iteration launches it: with the Givens
            // rider in workgroup 0 once a solve has left its state behind (the rider finds `done` set and leaves)
            spk::k::GivensRider gr{c->ka, 0, c->small.p, c->small.p + 64, nullptr, nullptr, 0, spk::k::FinErr{nullptr, 0}, spk::k::PeerAR{}};
            const spk::k::GivensRider *rp = c->kst.p ? &gr : nullptr;
            const bool acc = w == "spmv_acc";
            spk::a_mult(c, x, y, nullptr, nullptr, nullptr, acc, nullptr, rp);
        }
        else if (w == "mult") spk::op_mult(c, x, y, nullptr);
        else if (w == "pc") { if (!c->pc_ready) spk::fail(SPK_ERR_STATE, "pc not set up"); spk::op_pc_apply(c, x, y, nullptr); }
        else if (w == "mdot") spk::k::mdot(V.p, ld, nv, x, N, N, c->fin(c->small.p), nullptr, s);
        else if (w == "maxpy") spk::k::maxpy(V.p, ld, nv, nullptr, coef.p, -1.0, x, N, N, c->fin(c->small.p + 64), nullptr, s);
        else if (w == "maxpy_nonorm") spk::k::maxpy(V.p, ld, nv, nullptr, coef.p, -1.0, x, N, N, c->fin(nullptr), nullptr, s);
        else if (w == "scale") spk::k::scale_dev(x, N, coef.p, nullptr, s);
        else if (w == "wide_dot") { if (!c->have_B) spk::fail(SPK_ERR_STATE, "no B"); spk::k::wide_dot(c->B, x, c->fin(c->small.p), nullptr, s); }
        else if (w == "bt_update") { if (!c->have_B || !c->pc_ready) spk::fail(SPK_ERR_STATE, "no B / pc"); spk::k::bt_update(1, c->Bt, c->dinv.p, x, c->small.p + 200, y, nullptr, s); }
        else spk::fail(SPK_ERR_ARG, "spk_time_kernel: unknown kernel '%s'", which);
    };
    if ((w == "spmv_acc" || w == "spmv_ride" || w == "spmv_gated") && c->kst.p) {
        const int32_t one = 1;   // (krylov_init of the next solve resets it)
        SPK_HIP(hipMemcpyAsync(&c->kst.p->done, &one, sizeof one, hipMemcpyHostToDevice, s));
    }
    hipEvent_t e0, e1;
    SPK_HIP(hipEventCreate(&e0));
    SPK_HIP(hipEventCreate(&e1));
    for (int i = 0; i < warmup; ++i) run();
    SPK_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) run();
    SPK_HIP(hipEventRecord(e1, s));
    SPK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    SPK_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_launch = (double)ms / reps;
    SPK_CATCH(c)
}

}  // extern "C"
