// spk_comm.cpp -- the two collectives KSPSolve needs across ranks, replacing
// what PETSc does with MPI inside the call at
// /root/reference/src/SaddlePointProblem.c:70:
//   * MPI_Allreduce of <= restart+2 doubles per Krylov reduction (VecMDot, VecNorm)
//   * VecScatter of the ghost entries MatMult_MPIAIJ needs (the two slab neighbours)
// Backends:
//   PeerComm   production: windows of the peers' HBM mapped through HIP IPC, the solver's own
//              kernels store tagged 8-byte granules into them over xGMI (spk_device.hpp);
//              wraps one of the backends below for set-up traffic and as the fallback
//   RcclComm   one process per GPU: ncclAllReduce / grouped ncclSend+ncclRecv on the solver's
//              stream, ncclAllGather for set-up (RCCL is dlopen'ed so that the library loads,
//              and the CPU tests run, on a box without it and so that the process shares
//              whichever librccl is already mapped)
//   HostCbComm every operation staged through host callbacks (gloo/MPI): lets N processes
//              share ONE GPU, which RCCL refuses -- rehearsals on a 1-GPU box
//   LocalComm  several logical ranks inside one process on ONE device, each
//              driven by its own host thread -- lets a 1-GPU box run the
//              partitioned algorithm (parity tests); host barriers, slow.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>

#include "spk_internal.hpp"

// ---------------------------------------------------------------------------
// local group (in-process logical ranks)
// ---------------------------------------------------------------------------
struct spk_local_group {
    int nranks = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long generation = 0;
    // all-reduce slots: nranks x 256 doubles, device memory (allocated lazily by rank 0)
    double *slots = nullptr;
    // exchange bookkeeping published by every rank
    std::vector<const double *> sendbuf;
    std::vector<std::vector<int>> peers;
    std::vector<std::vector<int64_t>> send_off;
    // host staging for allgather
    std::vector<std::vector<char>> stage;

    // bounded: a rank that failed (exception) must not leave the others waiting forever
    void barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) spk::fail(SPK_ERR_COMM, "local group: a rank has failed");
        const long gen = generation;
        if (++arrived == nranks) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else if (!cv.wait_for(lk, std::chrono::seconds(120), [&] { return generation != gen || broken; }) || broken) {
            broken = true;
            cv.notify_all();
            spk::fail(SPK_ERR_COMM, "local group: barrier timed out or a rank failed");
        }
    }
    bool broken = false;
};

namespace spk {
namespace {

// ---------------------------------------------------------------------------
// RCCL, resolved at run time
// ---------------------------------------------------------------------------
struct RcclApi {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi &rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // share an already-mapped librccl (e.g. torch's) before opening ROCm's
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            api.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
            if (api.h) break;
        }
        for (const char *n : names) {
            if (api.h) break;
            api.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!api.h) return;
        auto sym = [&](const char *s) { return dlsym(api.h, s); };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.Send = (decltype(api.Send))sym("ncclSend");
        api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    });
    if (!api.h || !api.CommInitRank || !api.AllReduce || !api.Send || !api.Recv) {
        const char *de = dlerror();  // one call: the second would return NULL (the message is consumed)
        fail(SPK_ERR_COMM, "RCCL (librccl.so.1) could not be loaded: %s", de ? de : "missing symbols");
    }
    return api;
}

#define SPK_NCCL(call)                                                                          \
    do {                                                                                        \
        ncclResult_t r_ = (call);                                                               \
        if (r_ != ncclSuccess)                                                                  \
            fail(SPK_ERR_COMM, "%s failed: %s", #call,                                          \
                 rccl().GetErrorString ? rccl().GetErrorString(r_) : "rccl error");             \
    } while (0)

class SelfComm : public Comm {};

class RcclComm : public Comm {
public:
    RcclComm(int rank, int nranks, const void *id128) : rank_(rank), n_(nranks)
    {
        ncclUniqueId id;
        static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
        std::memcpy(&id, id128, sizeof id);
        SPK_NCCL(rccl().CommInitRank(&comm_, nranks, id, rank));
    }
    ~RcclComm() override
    {
        if (comm_) (void)rccl().CommDestroy(comm_);
    }
    const char *name() const override { return "rccl"; }
    int rank() const override { return rank_; }
    int size() const override { return n_; }
    void allreduce_sum(double *dev, int count, hipStream_t s) override
    {
        ++n_ar_calls_;   // (spk_comm_get_info: collectives that went through this backend)
        if (count <= 0) return;
        SPK_NCCL(rccl().AllReduce(dev, dev, (size_t)count, ncclDouble, ncclSum, comm_, s));
    }
    void exchange(const double *sendbuf, const std::vector<int> &peers, const std::vector<int64_t> &send_off,
                  double *recvbuf, const std::vector<int64_t> &recv_off, hipStream_t s) override
    {
        ++n_ex_calls_;   // (spk_comm_get_info: collectives that went through this backend)
        if (peers.empty()) return;
        SPK_NCCL(rccl().GroupStart());
        for (size_t i = 0; i < peers.size(); ++i) {
            const int64_t ns = send_off[i + 1] - send_off[i], nr = recv_off[i + 1] - recv_off[i];
            if (ns > 0) SPK_NCCL(rccl().Send(sendbuf + send_off[i], (size_t)ns, ncclDouble, peers[i], comm_, s));
            if (nr > 0) SPK_NCCL(rccl().Recv(recvbuf + recv_off[i], (size_t)nr, ncclDouble, peers[i], comm_, s));
        }
        SPK_NCCL(rccl().GroupEnd());
    }
    void host_allgather(const void *in, void *out, size_t bytes_each) override
    {
        DevBuf<char> din, dout;
        din.upload((const char *)in, bytes_each);
        dout.alloc(bytes_each * (size_t)n_);
        SPK_NCCL(rccl().AllGather(din.p, dout.p, bytes_each, ncclChar, comm_, nullptr));
        SPK_HIP(hipStreamSynchronize(nullptr));
        SPK_HIP(hipMemcpy(out, dout.p, bytes_each * (size_t)n_, hipMemcpyDeviceToHost));
    }
    void host_allgatherv(const void *in, size_t bytes_in, std::vector<std::vector<char>> &out) override
    {
        std::vector<int64_t> sizes((size_t)n_);
        const int64_t mine = (int64_t)bytes_in;
        host_allgather(&mine, sizes.data(), sizeof(int64_t));
        int64_t mx = 16;
        for (auto v : sizes) mx = std::max(mx, v);
        std::vector<char> padded((size_t)mx, 0), all((size_t)mx * (size_t)n_);
        if (bytes_in) std::memcpy(padded.data(), in, bytes_in);
        host_allgather(padded.data(), all.data(), (size_t)mx);
        out.resize((size_t)n_);
        for (int r = 0; r < n_; ++r)
            out[(size_t)r].assign(all.begin() + (size_t)mx * r, all.begin() + (size_t)mx * r + (size_t)sizes[(size_t)r]);
    }

private:
    int rank_, n_;
    ncclComm_t comm_ = nullptr;
};

class LocalComm : public Comm {
public:
    LocalComm(spk_local_group *g, int rank) : g_(g), rank_(rank)
    {
        if (rank == 0 && !g->slots) {
            SPK_HIP(hipMalloc((void **)&g->slots, sizeof(double) * 256 * (size_t)g->nranks));
            SPK_HIP(hipMemset(g->slots, 0, sizeof(double) * 256 * (size_t)g->nranks));
        }
        g->barrier();
    }
    const char *name() const override { return "local"; }
    int rank() const override { return rank_; }
    int size() const override { return g_->nranks; }
    void allreduce_sum(double *dev, int count, hipStream_t s) override
    {
        ++n_ar_calls_;   // (spk_comm_get_info: collectives that went through this backend)
        if (count <= 0) return;
        for (int off = 0; off < count; off += 256) {  // slots of 256 values per rank: longer payloads in pieces
            const int cnt = std::min(256, count - off);
            SPK_HIP(hipMemcpyAsync(g_->slots + 256 * (size_t)rank_, dev + off, sizeof(double) * (size_t)cnt,
                                   hipMemcpyDeviceToDevice, s));
            SPK_HIP(hipStreamSynchronize(s));
            g_->barrier();
            k::sum_slots(g_->slots, g_->nranks, 256, cnt, dev + off, s);
            SPK_HIP(hipStreamSynchronize(s));
            g_->barrier();
        }
    }
    void exchange(const double *sendbuf, const std::vector<int> &peers, const std::vector<int64_t> &send_off,
                  double *recvbuf, const std::vector<int64_t> &recv_off, hipStream_t s) override
    {
        ++n_ex_calls_;   // (spk_comm_get_info: collectives that went through this backend)
        g_->sendbuf[(size_t)rank_] = sendbuf;
        g_->peers[(size_t)rank_] = peers;
        g_->send_off[(size_t)rank_] = send_off;
        SPK_HIP(hipStreamSynchronize(s));
        g_->barrier();
        for (size_t i = 0; i < peers.size(); ++i) {
            const int p = peers[i];
            const int64_t nr = recv_off[i + 1] - recv_off[i];
            if (nr <= 0) continue;
            const auto &pp = g_->peers[(size_t)p];
            size_t j = 0;
            while (j < pp.size() && pp[j] != rank_) ++j;
            if (j == pp.size()) fail(SPK_ERR_COMM, "local exchange: rank %d not a peer of %d", rank_, p);
            const auto &po = g_->send_off[(size_t)p];
            if (po[j + 1] - po[j] != nr) fail(SPK_ERR_COMM, "local exchange: size mismatch %d<->%d", rank_, p);
            SPK_HIP(hipMemcpyAsync(recvbuf + recv_off[i], g_->sendbuf[(size_t)p] + po[j], sizeof(double) * (size_t)nr,
                                   hipMemcpyDeviceToDevice, s));
        }
        SPK_HIP(hipStreamSynchronize(s));
        g_->barrier();
    }
    void host_allgather(const void *in, void *out, size_t bytes_each) override
    {
        g_->stage[(size_t)rank_].assign((const char *)in, (const char *)in + bytes_each);
        g_->barrier();
        for (int r = 0; r < g_->nranks; ++r) std::memcpy((char *)out + bytes_each * (size_t)r, g_->stage[(size_t)r].data(), bytes_each);
        g_->barrier();
    }
    void host_allgatherv(const void *in, size_t bytes_in, std::vector<std::vector<char>> &out) override
    {
        g_->stage[(size_t)rank_].assign((const char *)in, (const char *)in + bytes_in);
        g_->barrier();
        out = g_->stage;
        g_->barrier();
    }

private:
    spk_local_group *g_;
    int rank_;
};

// Host-callback transport: every operation goes device -> host -> callback -> device.
class HostCbComm : public Comm {
public:
    HostCbComm(int rank, int nranks, const spk_host_comm &cb) : rank_(rank), n_(nranks), cb_(cb) {}
    const char *name() const override { return "host-callback"; }
    int rank() const override { return rank_; }
    int size() const override { return n_; }
    void allreduce_sum(double *dev, int count, hipStream_t s) override
    {
        ++n_ar_calls_;   // (spk_comm_get_info: collectives that went through this backend)
        if (count <= 0) return;
        std::vector<double> h((size_t)count);
        SPK_HIP(hipMemcpyAsync(h.data(), dev, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, s));
        SPK_HIP(hipStreamSynchronize(s));
        if (cb_.allreduce(cb_.user, h.data(), count)) fail(SPK_ERR_COMM, "host all-reduce callback failed");
        SPK_HIP(hipMemcpyAsync(dev, h.data(), sizeof(double) * (size_t)count, hipMemcpyHostToDevice, s));
        SPK_HIP(hipStreamSynchronize(s));
    }
    void exchange(const double *sendbuf, const std::vector<int> &peers, const std::vector<int64_t> &send_off,
                  double *recvbuf, const std::vector<int64_t> &recv_off, hipStream_t s) override
    {
        ++n_ex_calls_;   // (spk_comm_get_info: collectives that went through this backend)
        if (peers.empty()) return;
        const int64_t ns = send_off.back(), nr = recv_off.back();
        std::vector<double> hs((size_t)std::max<int64_t>(ns, 1)), hr((size_t)std::max<int64_t>(nr, 1));
        if (ns) SPK_HIP(hipMemcpyAsync(hs.data(), sendbuf, sizeof(double) * (size_t)ns, hipMemcpyDeviceToHost, s));
        SPK_HIP(hipStreamSynchronize(s));
        for (size_t i = 0; i < peers.size(); ++i)
            if (cb_.exchange(cb_.user, peers[i], hs.data() + send_off[i], send_off[i + 1] - send_off[i],
                             hr.data() + recv_off[i], recv_off[i + 1] - recv_off[i]))
                fail(SPK_ERR_COMM, "host exchange callback failed");
        if (nr) SPK_HIP(hipMemcpyAsync(recvbuf, hr.data(), sizeof(double) * (size_t)nr, hipMemcpyHostToDevice, s));
        SPK_HIP(hipStreamSynchronize(s));
    }
    void host_allgather(const void *in, void *out, size_t bytes_each) override
    {
        if (cb_.allgather(cb_.user, in, out, (int64_t)bytes_each)) fail(SPK_ERR_COMM, "host all-gather callback failed");
    }
    void host_allgatherv(const void *in, size_t bytes_in, std::vector<std::vector<char>> &out) override
    {
        std::vector<int64_t> sizes((size_t)n_);
        const int64_t mine = (int64_t)bytes_in;
        host_allgather(&mine, sizes.data(), sizeof(int64_t));
        int64_t mx = 16;
        for (auto v : sizes) mx = std::max(mx, v);
        std::vector<char> padded((size_t)mx, 0), all((size_t)mx * (size_t)n_);
        if (bytes_in) std::memcpy(padded.data(), in, bytes_in);
        host_allgather(padded.data(), all.data(), (size_t)mx);
        out.resize((size_t)n_);
        for (int r = 0; r < n_; ++r)
            out[(size_t)r].assign(all.begin() + (size_t)mx * r, all.begin() + (size_t)mx * r + (size_t)sizes[(size_t)r]);
    }

private:
    int rank_, n_;
    spk_host_comm cb_;
};

// ---------------------------------------------------------------------------
// Peer-store backend: one-shot all-reduce and halo exchange written straight into
// the peers' memory over xGMI by the solver's own kernels (spk_device.hpp, "granules").
// Each rank owns two windows of UNCACHED device memory, mapped into every peer
// through HIP IPC (ranks in other processes) or used directly (logical ranks of one
// process): the all-reduce window (fixed size) and the halo staging (sized by the
// halo plan).  `inner` -- RCCL or the host transport -- carries the set-up traffic
// and stays the fallback: the backend is only switched on when every rank mapped
// every window AND a self-test all-reduce returned the right sums on every rank.
// ---------------------------------------------------------------------------
struct WinInfo {
    int64_t pid;
    uint64_t ptr;
    hipIpcMemHandle_t handle;
    int32_t ok, device;
    int64_t n_ghost;
    int64_t max_seg;                    // my longest halo segment (doubles)
    int64_t recv_off_for[k::kPeerMax];  // where rank p's rows land in my ghost array (-1: not a peer)
};

class PeerComm : public Comm {
public:
    // `inner` is adopted only once construction has succeeded (adopt()): a constructor that throws must
    // not take the caller's communicator down with it
    PeerComm(int P, int me, int device) : device_(device), P_(P), me_(me)
    {
        const char *t = getenv("SPK_PEER_TIMEOUT_MS");
        timeout_ms_ = t ? (uint32_t)std::max(1, atoi(t)) : 30000u;
        const char *fz = getenv("SPK_PEER_FUSE");  // 0: all-reduces as launches of their own (A/B runs)
        fuse_ = !(fz && !strcmp(fz, "0"));
        // Granules are the latency tool: 8-byte tagged stores, twice the bytes.  When any rank has a halo
        // segment beyond this many doubles (a node PLANE of a 3-D slab: 1.57 MB at 256^3 against 16 KiB
        // for a node line at 1024^2) the exchange is bandwidth-bound and takes the bulk form: plain
        // doubles in chunks, one flag per chunk (k::PeerBulk).
        const char *hm = getenv("SPK_PEER_HALO_MAX");
        halo_max_ = hm ? std::max(0, atoi(hm)) : 8192;
        err_.alloc(4);
        stats_.alloc(2 * k::kStatCount);
        std::memset(ar_map_, 0, sizeof ar_map_);
        std::memset(halo_map_, 0, sizeof halo_map_);
    }
    void adopt(Comm *inner) { inner_.reset(inner); }
    ~PeerComm() override
    {
        close_maps(ar_map_, ar_own_);
        close_maps(halo_map_, halo_own_);
        if (ar_own_) (void)hipFree(ar_own_);
        if (halo_own_) (void)hipFree(halo_own_);
    }
    Comm *release_inner() { return inner_.release(); }
    const char *name() const override { return "peer-store"; }
    bool fuses() const override { return fuse_; }
    const int32_t *error_dev() const override { return err_.p; }
    int rank() const override { return me_; }
    int size() const override { return P_; }

    // maps the all-reduce windows and runs the self-test; false (with *why) when any rank failed
    // maps the all-reduce windows and runs the self-test; false (with *why) when any rank failed.
    // Three kinds of window memory are tried in turn -- uncached, fine-grained, plain -- and a kind only
    // counts when EVERY rank mapped every window and the self-test returned the right sums everywhere
    // (a kind whose polls cannot see a peer's stores fails it by time-out).
    bool enable(std::string *why)
    {
        const size_t bytes = sizeof(unsigned long long) * (size_t)k::kArSlots * P_ * k::kArGranules;
        std::string mywhy;
        bool ok = false;
        for (tier_ = 0; tier_ < 3 && !ok; ++tier_) {
            // every local step is fenced: whatever fails on this rank (an allocation, a HIP call inside the
            // self-test) must still reach the agree() the other ranks are waiting in
            try {
                ok = alloc_window(&ar_own_, bytes, &mywhy);
            } catch (const Error &e) {
                ok = false;
                mywhy = e.msg;
            }
            ok = share_window(ar_own_, ok, 0, nullptr, ar_map_, nullptr, &mywhy) && ok;  // collective inside
            ok = agree(ok);
            if (ok) {
                bool st = false;
                try {
                    st = self_test(&mywhy);
                } catch (const Error &e) {
                    mywhy = e.msg;
                    (void)hipGetLastError();
                }
                ok = agree(st);
            }
            if (!ok) {
                close_maps(ar_map_, ar_own_);
                if (ar_own_) (void)hipFree(ar_own_);
                ar_own_ = nullptr;
                (void)hipGetLastError();
                SPK_HIP(hipMemset(err_.p, 0, sizeof(int32_t)));
            }
        }
        --tier_;  // the kind that worked (also used for the halo staging)
        if (!ok) why_ = mywhy.empty() ? "another rank could not map the windows or failed its self-test" : mywhy;
        if (!ok && why) *why = why_;
        self_test_ok_ = ok;
        return ok;
    }
    // six all-reduces (every slot, and the wrap) of values that differ per rank and per position, on a
    // stream of its own
    bool self_test(std::string *why)
    {
        bool ok = true;
        DevBuf<double> buf;
        buf.alloc(64);
        hipStream_t ts = nullptr;
        SPK_HIP(hipStreamCreateWithFlags(&ts, hipStreamNonBlocking));
        for (int round = 0; round < 6 && ok; ++round) {
            std::vector<double> h(64), r(64);
            for (int i = 0; i < 64; ++i) h[(size_t)i] = (double)(me_ + 1) * (i + 1 + round) + 0.25 * round;
            SPK_HIP(hipMemcpy(buf.p, h.data(), 64 * sizeof(double), hipMemcpyHostToDevice));
            k::PeerAR a = next_ar();
            a.stats = nullptr;  // the self-test is not part of the account
            a.timeout_ms = 5000;
            k::peer_allreduce(a, buf.p, 64, ts);
            SPK_HIP(hipStreamSynchronize(ts));
            SPK_HIP(hipMemcpy(r.data(), buf.p, 64 * sizeof(double), hipMemcpyDeviceToHost));
            for (int i = 0; i < 64 && ok; ++i) {
                const double want = 0.5 * P_ * (P_ + 1) * (i + 1 + round) + 0.25 * round * P_;
                if (r[(size_t)i] != want) {
                    ok = false;
                    *why = "self-test all-reduce returned a wrong sum";
                }
            }
            if (error_word()) {
                ok = false;
                *why = "self-test all-reduce timed out";
            }
        }
        (void)hipStreamDestroy(ts);
        return ok;
    }

    void allreduce_sum(double *dev, int count, hipStream_t s) override
    {
        if (count <= 0) return;
        if (2 * count > k::kArGranules) {
            ++n_ar_inner_;
            inner_->allreduce_sum(dev, count, s);
            return;
        }
        ++n_ar_kernel_;
        k::peer_allreduce(next_ar(k::kStatArOther), dev, count, s);
    }
    k::PeerAR fused_allreduce(int count, int kind) override
    {
        if (!fuse_ || count <= 0 || 2 * count > k::kArGranules) return k::PeerAR{};
        ++n_ar_fused_;
        return next_ar(kind);
    }
    void info(spk_comm_info *o) override
    {
        o->peer_enabled = 1;
        o->window_tier = tier_;
        o->self_test_ok = self_test_ok_ ? 1 : 0;
        o->halo_mode = halo_peers_.empty() && !halo_ok_ ? 0 : (!halo_ok_ ? 3 : (bulk_ ? 2 : 1));
        o->halo_fused = (halo_ok_ && !bulk_ && fuse_) ? 1 : 0;
        o->n_allreduce_fused = n_ar_fused_;
        o->n_allreduce_kernel = n_ar_kernel_;
        o->n_allreduce_inner = n_ar_inner_;
        o->n_halo_fused = n_halo_fused_;
        o->n_halo_kernel = n_halo_kernel_;
        o->n_halo_inner = n_halo_inner_;
        unsigned long long st[2 * k::kStatCount] = {0};
        if (hipMemcpy(st, stats_.p, sizeof st, hipMemcpyDeviceToHost) != hipSuccess) (void)hipGetLastError();
        for (int i = 0; i < 4; ++i) {
            o->wait_ticks[i] = st[2 * i];
            o->wait_count[i] = st[2 * i + 1];
        }
        std::snprintf(o->inner_backend, sizeof o->inner_backend, "%s", inner_->name());
        std::snprintf(o->why, sizeof o->why, "%s", why_.c_str());
    }
    void setup_halo(int32_t n_ghost, const std::vector<int> &peers, const std::vector<int64_t> &send_off,
                    const std::vector<int64_t> &recv_off) override
    {
        inner_->setup_halo(n_ghost, peers, send_off, recv_off);
        close_maps(halo_map_, halo_own_);
        if (halo_own_) (void)hipFree(halo_own_);
        halo_own_ = nullptr;
        halo_ok_ = false;
        n_ghost_ = n_ghost;
        std::string why;
        bool ok = peers.size() <= 4;
        if (!ok) why = "more than four halo neighbours";
        int64_t max_seg = 0;
        for (size_t i = 0; i < peers.size(); ++i)
            max_seg = std::max(max_seg, std::max(send_off[i + 1] - send_off[i], recv_off[i + 1] - recv_off[i]));
        // staging: two parities x n_ghost doubles x two granules (the bulk form needs less: data + flags)
        const size_t bytes = sizeof(unsigned long long) * 4 * (size_t)std::max<int32_t>(n_ghost, 8);
        try {  // local failures are fenced: the collectives below are reached on every rank
            ok = alloc_window(&halo_own_, bytes, &why) && ok;
        } catch (const Error &e) {
            ok = false;
            why = e.msg;
        }
        int64_t roff[k::kPeerMax];
        for (int p = 0; p < k::kPeerMax; ++p) roff[p] = -1;
        for (size_t i = 0; i < peers.size(); ++i)
            if (peers[i] < k::kPeerMax) roff[peers[i]] = recv_off[i];
        std::vector<WinInfo> all;
        ok = share_window(halo_own_, ok, n_ghost, roff, halo_map_, &all, &why, max_seg) && ok;
        bulk_ = false;
        for (const WinInfo &w : all) bulk_ = bulk_ || w.max_seg > halo_max_;  // the same decision on every rank
        if (ok) {
            halo_peers_ = peers;
            halo_send_off_ = send_off;
            halo_recv_off_ = recv_off;
            for (size_t i = 0; i < peers.size() && ok; ++i) {
                const WinInfo &w = all[(size_t)peers[i]];
                halo_remote_ng_[i] = w.n_ghost;
                halo_remote_off_[i] = w.recv_off_for[me_];
                const int64_t ns = send_off[i + 1] - send_off[i];
                if (ns > 0 && (w.recv_off_for[me_] < 0 || w.recv_off_for[me_] + ns > w.n_ghost)) {
                    ok = false;
                    why = "halo plans of two ranks do not match";
                }
            }
        }
        ok = agree(ok);
        if (ok) {
            // self-test through both staging parities: element j of the segment rank p sends me must
            // arrive as p * 2^20 + j (+ 0.5 in the second round)
            halo_ok_ = true;
            const int64_t ns = send_off.back(), nr = recv_off.back();
            DevBuf<double> sb, rb;
            sb.alloc((size_t)std::max<int64_t>(ns, 1));
            rb.alloc((size_t)std::max<int64_t>(nr, 1));
            hipStream_t ts = nullptr;
            const uint32_t keep = timeout_ms_;
            timeout_ms_ = 5000;
            try {
            SPK_HIP(hipStreamCreateWithFlags(&ts, hipStreamNonBlocking));
            for (int round = 0; round < 2 && ok; ++round) {
                std::vector<double> hs((size_t)std::max<int64_t>(ns, 1)), hr((size_t)std::max<int64_t>(nr, 1), -1.0);
                for (size_t i = 0; i < peers.size(); ++i)
                    for (int64_t j = send_off[i]; j < send_off[i + 1]; ++j)
                        hs[(size_t)j] = (double)me_ * 1048576.0 + (double)(j - send_off[i]) + 0.5 * round;
                SPK_HIP(hipMemcpy(sb.p, hs.data(), hs.size() * sizeof(double), hipMemcpyHostToDevice));
                SPK_HIP(hipMemcpy(rb.p, hr.data(), hr.size() * sizeof(double), hipMemcpyHostToDevice));
                exchange(sb.p, peers, send_off, rb.p, recv_off, ts);
                SPK_HIP(hipStreamSynchronize(ts));
                SPK_HIP(hipMemcpy(hr.data(), rb.p, hr.size() * sizeof(double), hipMemcpyDeviceToHost));
                for (size_t i = 0; i < peers.size() && ok; ++i)
                    for (int64_t j = recv_off[i]; j < recv_off[i + 1] && ok; ++j)
                        ok = hr[(size_t)j] == (double)peers[i] * 1048576.0 + (double)(j - recv_off[i]) + 0.5 * round;
                if (error_word()) {
                    ok = false;
                    why = "halo self-test timed out";
                } else if (!ok) {
                    why = "halo self-test delivered wrong values";
                }
            }
            } catch (const Error &e) {  // a HIP failure inside the self-test: still reach agree()
                ok = false;
                why = e.msg;
                (void)hipGetLastError();
            }
            timeout_ms_ = keep;
            if (ts) (void)hipStreamDestroy(ts);
            ok = agree(ok);
            if (!ok && hipMemset(err_.p, 0, sizeof(int32_t)) != hipSuccess) (void)hipGetLastError();  // the fallback starts clean
        }
        halo_ok_ = ok;
        if (!ok && !peers.empty()) why_ = "halo exchange on the inner backend: " + (why.empty() ? std::string("another rank failed") : why);
    }
    void exchange(const double *sendbuf, const std::vector<int> &peers, const std::vector<int64_t> &send_off,
                  double *recvbuf, const std::vector<int64_t> &recv_off, hipStream_t s) override
    {
        if (peers.empty()) return;
        if (!halo_ok_ || peers != halo_peers_ || send_off != halo_send_off_ || recv_off != halo_recv_off_) {
            ++n_halo_inner_;
            inner_->exchange(sendbuf, peers, send_off, recvbuf, recv_off, s);
            return;
        }
        ++n_halo_kernel_;
        if (bulk_) {
            k::PeerBulk h{};
            h.npeers = (int)peers.size();
            h.seq = ++halo_seq_;
            h.timeout_ms = timeout_ms_;
            const size_t par = h.seq & 1u;
            for (size_t i = 0; i < peers.size(); ++i) {
                // staging of a rank, in 8-byte units: parity p at p * 2 ng: ng doubles, then ng flag slots
                unsigned long long *base = halo_map_[peers[i]] + par * 2 * (size_t)halo_remote_ng_[i];
                h.rdata[i] = reinterpret_cast<double *>(base) + halo_remote_off_[i];
                h.rflag[i] = base + (size_t)halo_remote_ng_[i] + (size_t)halo_remote_off_[i];
                h.send_off[i] = send_off[i];
                h.recv_off[i] = recv_off[i];
                h.send_chunk0[i + 1] = h.send_chunk0[i] + (int32_t)((send_off[i + 1] - send_off[i] + k::kBulkChunk - 1) / k::kBulkChunk);
                h.recv_chunk0[i + 1] = h.recv_chunk0[i] + (int32_t)((recv_off[i + 1] - recv_off[i] + k::kBulkChunk - 1) / k::kBulkChunk);
            }
            h.send_off[peers.size()] = send_off.back();
            h.recv_off[peers.size()] = recv_off.back();
            const unsigned long long *mine = halo_own_ + par * 2 * (size_t)n_ghost_;
            h.mdata = reinterpret_cast<const double *>(mine);
            h.mflag = mine + (size_t)n_ghost_;
            h.err = err_.p;
            h.stats = stats_.p;
            k::peer_exchange_bulk(h, sendbuf, recvbuf, s);
            return;
        }
        k::PeerHalo h{};
        h.npeers = (int)peers.size();
        h.seq = ++halo_seq_;
        h.timeout_ms = timeout_ms_;
        const int par = (int)(h.seq & 1u);
        for (size_t i = 0; i < peers.size(); ++i) {
            h.remote[i] = halo_map_[peers[i]] + 2 * ((size_t)par * (size_t)halo_remote_ng_[i] + (size_t)halo_remote_off_[i]);
            h.send_off[i] = send_off[i];
            h.recv_off[i] = recv_off[i];
        }
        h.send_off[peers.size()] = send_off.back();
        h.recv_off[peers.size()] = recv_off.back();
        h.mine = halo_own_ + 2 * (size_t)par * (size_t)n_ghost_;
        h.err = err_.p;
        h.stats = stats_.p;
        k::peer_exchange(h, sendbuf, recvbuf, s);
    }
    bool fused_halo(k::SendRanges &sr, double *xghost) override
    {
        if (!fuse_ || !halo_ok_ || bulk_ || sr.n != (int)halo_peers_.size() || sr.n < 1) return false;
        for (int i = 0; i < sr.n; ++i)
            if (sr.len[i] != halo_send_off_[(size_t)i + 1] - halo_send_off_[(size_t)i]) return false;
        sr.peer = 1;
        sr.seq = ++halo_seq_;
        sr.timeout_ms = timeout_ms_;
        const int par = (int)(sr.seq & 1u);
        for (int i = 0; i < sr.n; ++i)
            sr.remote[i] = halo_map_[halo_peers_[(size_t)i]] +
                           2 * ((size_t)par * (size_t)halo_remote_ng_[i] + (size_t)halo_remote_off_[i]);
        sr.mine = halo_own_ + 2 * (size_t)par * (size_t)n_ghost_;
        sr.nrecv = n_ghost_;
        sr.xghost = xghost;
        sr.err = err_.p;
        sr.stats = stats_.p;
        ++n_halo_fused_;
        return true;
    }
    bool resident_plan(int n_ar, int n_halo, k::PeerAR &ar, k::SendRanges &sr0, k::SendRanges &sr1, double *xghost) override
    {
        if (!fuse_ || n_ar < 1) return false;
        const bool have_halo = !halo_peers_.empty();
        if (have_halo && (!halo_ok_ || bulk_ || n_halo < 0)) return false;
        if (have_halo && n_halo > 0) {
            if (!fused_halo(sr0, xghost)) return false;
            if (n_halo > 1) {
                if (!fused_halo(sr1, xghost)) return false;
                halo_seq_ += (uint32_t)(n_halo - 2);
                n_halo_fused_ += n_halo - 2;
            }
        }
        ar = next_ar(k::kStatArDots);
        ar_seq_ += (uint32_t)(n_ar - 1);
        n_ar_fused_ += n_ar;
        return true;
    }
    void check(hipStream_t s) override
    {
        (void)s;
        if (error_word()) {
            int32_t w[4] = {0, 0, 0, 0};
            if (hipMemcpy(w, err_.p, sizeof w, hipMemcpyDeviceToHost) != hipSuccess) (void)hipGetLastError();
            const char *where = w[1] == 1 ? "all-reduce after MDot" : w[1] == 2 ? "all-reduce after MAXPY" : w[1] == 3 ? "stand-alone all-reduce"
                              : w[1] == 16 ? "halo rows (head kernel)" : w[1] == 17 ? "halo rows (MAXPY-head kernel)"
                              : w[1] == 18 ? "halo rows (two-launch kernel B)" : w[1] == 19 ? "halo exchange kernel"
                              : w[1] == 20 ? "bulk halo exchange kernel" : "unknown wait";
            fail(SPK_ERR_COMM, "peer-store collective timed out after %u ms waiting for another rank (rank %d, first in: %s, sequence %d; "
                               "all-reduces issued %u, halo exchanges issued %u)", timeout_ms_, me_, where, w[2], ar_seq_, halo_seq_);
        }
    }
    void host_allgather(const void *in, void *out, size_t bytes_each) override { inner_->host_allgather(in, out, bytes_each); }
    void host_allgatherv(const void *in, size_t bytes_in, std::vector<std::vector<char>> &out) override
    {
        inner_->host_allgatherv(in, bytes_in, out);
    }
    void set_fuse(bool f) { fuse_ = f; }

private:
    k::PeerAR next_ar(int kind = k::kStatArOther)
    {
        k::PeerAR a{};
        a.stats = stats_.p;
        a.kind = kind;
        a.P = P_;
        a.me = me_;
        a.seq = ++ar_seq_;
        a.timeout_ms = timeout_ms_;
        for (int p = 0; p < P_; ++p) a.win[p] = ar_map_[p];
        a.err = err_.p;
        return a;
    }
    int32_t error_word()
    {
        int32_t e = 0;
        SPK_HIP(hipMemcpy(&e, err_.p, sizeof e, hipMemcpyDeviceToHost));
        return e;
    }
    bool agree(bool ok)
    {
        std::vector<int32_t> all((size_t)P_);
        const int32_t mine = ok ? 1 : 0;
        inner_->host_allgather(&mine, all.data(), sizeof mine);
        for (int32_t v : all) ok = ok && v != 0;
        return ok;
    }
    bool alloc_window(unsigned long long **p, size_t bytes, std::string *why)
    {
        *p = nullptr;
        // uncached (MTYPE UC) first: a peer's stores must be seen by my polls without any cache
        // maintenance (what RCCL allocates for its own flags); then fine-grained, then plain
        hipError_t e = tier_ == 0   ? hipExtMallocWithFlags((void **)p, bytes, hipDeviceMallocUncached)
                       : tier_ == 1 ? hipExtMallocWithFlags((void **)p, bytes, hipDeviceMallocFinegrained)
                                    : hipMalloc((void **)p, bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            *p = nullptr;
            *why = "window memory of this kind could not be allocated";
            return false;
        }
        if (hipMemset(*p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            (void)hipGetLastError();
            *why = "window could not be cleared";
            return false;
        }
        return true;
    }
    // publishes my window, maps everybody else's; collective (one host all-gather)
    bool share_window(unsigned long long *own, bool ok, int64_t n_ghost, const int64_t *roff, unsigned long long **map,
                      std::vector<WinInfo> *all_out, std::string *why, int64_t max_seg = 0)
    {
        WinInfo mine{};
        mine.max_seg = max_seg;
        mine.pid = (int64_t)getpid();
        mine.ptr = (uint64_t)(uintptr_t)own;
        mine.device = device_;
        mine.n_ghost = n_ghost;
        for (int p = 0; p < k::kPeerMax; ++p) mine.recv_off_for[p] = roff ? roff[p] : -1;
        if (ok && own && hipIpcGetMemHandle(&mine.handle, own) != hipSuccess) {
            (void)hipGetLastError();
            ok = false;
            *why = "hipIpcGetMemHandle failed on the window";
        }
        mine.ok = ok ? 1 : 0;
        std::vector<WinInfo> all((size_t)P_);
        inner_->host_allgather(&mine, all.data(), sizeof mine);
        bool good = true;
        for (int p = 0; p < P_; ++p) {
            map[p] = nullptr;
            const WinInfo &w = all[(size_t)p];
            if (p == me_) {
                map[p] = own;
                good = good && own != nullptr && w.ok != 0;
            } else if (w.ok == 0) {
                good = false;
            } else if (w.pid == mine.pid) {
                // logical ranks of ONE process: their streams are coupled by the null stream and by
                // hipFree's device-wide wait, so one rank's host thread can block behind another
                // rank's kernel that is waiting for it -- a deadlock until the time-out
                good = false;
                *why = "ranks share a process (peer-store needs one process per rank)";
            } else if (w.ok != 1) {
                good = false;
            } else {
                // direct access to the peer's device, as RCCL's P2P transport sets it up (harmless when the
                // peer is not visible under this index or access is already on)
                int ndev = 0, can = 0;
                if (w.device != device_ && hipGetDeviceCount(&ndev) == hipSuccess && w.device < ndev &&
                    hipDeviceCanAccessPeer(&can, device_, w.device) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(w.device, 0);
                (void)hipGetLastError();
                void *q = nullptr;
                if (hipIpcOpenMemHandle(&q, w.handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
                    (void)hipGetLastError();
                    good = false;
                    *why = "hipIpcOpenMemHandle failed for rank " + std::to_string(p);
                } else {
                    map[p] = (unsigned long long *)q;
                    opened_.push_back(q);
                }
            }
        }
        if (all_out) *all_out = all;
        return good;
    }
    void close_maps(unsigned long long **map, unsigned long long *own)
    {
        for (int p = 0; p < k::kPeerMax; ++p) {
            void *q = map[p];
            if (q && q != own) {
                auto it = std::find(opened_.begin(), opened_.end(), q);
                if (it != opened_.end()) {
                    (void)hipIpcCloseMemHandle(q);
                    opened_.erase(it);
                }
            }
            map[p] = nullptr;
        }
        (void)hipGetLastError();  // a failed close must not surface later as somebody else's launch error
    }

    std::unique_ptr<Comm> inner_;
    int device_, P_, me_;
    uint32_t timeout_ms_ = 30000, ar_seq_ = 0, halo_seq_ = 0;
    int halo_max_ = 8192;
    int tier_ = 0;  // kind of window memory: 0 uncached, 1 fine-grained, 2 plain
    bool fuse_ = true, halo_ok_ = false, bulk_ = false;
    DevBuf<int32_t> err_;
    DevBuf<unsigned long long> stats_;
    int64_t n_ar_fused_ = 0, n_ar_kernel_ = 0, n_ar_inner_ = 0, n_halo_fused_ = 0, n_halo_kernel_ = 0, n_halo_inner_ = 0;
    bool self_test_ok_ = false;
    std::string why_;
    unsigned long long *ar_own_ = nullptr, *halo_own_ = nullptr;
    unsigned long long *ar_map_[k::kPeerMax], *halo_map_[k::kPeerMax];
    std::vector<void *> opened_;
    int32_t n_ghost_ = 0;
    std::vector<int> halo_peers_;
    std::vector<int64_t> halo_send_off_, halo_recv_off_;
    int64_t halo_remote_ng_[4] = {0, 0, 0, 0}, halo_remote_off_[4] = {0, 0, 0, 0};
};

}  // namespace

Comm *make_peer_comm(Comm *inner, int device, std::string *why)
{
    if (inner->size() < 2 || inner->size() > k::kPeerMax) {
        if (why) *why = "peer-store backend needs 2.." + std::to_string(k::kPeerMax) + " ranks";
        return inner;
    }
    // `inner` stays the caller's until the peer communicator is fully up: whatever throws below, the caller
    // gets `inner` back and keeps a working communicator
    PeerComm *pc = nullptr;
    bool ok = false;
    try {
        pc = new PeerComm(inner->size(), inner->rank(), device);
        pc->adopt(inner);
        ok = pc->enable(why);
    } catch (const Error &e) {
        // (enable() fences its local steps; what arrives here is a failure of the inner communicator itself)
        if (why) *why = e.msg;
        ok = false;
    } catch (const std::exception &e) {
        if (why) *why = e.what();
        ok = false;
    }
    if (!ok) {
        if (pc) {
            (void)pc->release_inner();
            delete pc;
        }
        return inner;
    }
    return pc;
}

Comm *make_host_comm(int rank, int nranks, const spk_host_comm &cb) { return new HostCbComm(rank, nranks, cb); }
Comm *make_self_comm() { return new SelfComm(); }
Comm *make_rccl_comm(int rank, int nranks, const void *id128, int device)
{
    (void)device;
    return new RcclComm(rank, nranks, id128);
}
Comm *make_local_comm(spk_local_group *grp, int rank) { return new LocalComm(grp, rank); }

void rccl_unique_id(void *id128)
{
    ncclUniqueId id;
    SPK_NCCL(rccl().GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof id);
}

}  // namespace spk

extern "C" int spk_local_group_create(spk_local_group **grp, int nranks)
{
    if (!grp || nranks < 1 || nranks > 64) return SPK_ERR_ARG;
    auto *g = new spk_local_group();
    g->nranks = nranks;
    g->sendbuf.assign((size_t)nranks, nullptr);
    g->peers.resize((size_t)nranks);
    g->send_off.resize((size_t)nranks);
    g->stage.resize((size_t)nranks);
    *grp = g;
    return SPK_OK;
}

extern "C" int spk_local_group_destroy(spk_local_group *grp)
{
    if (!grp) return SPK_OK;
    if (grp->slots) (void)hipFree(grp->slots);
    delete grp;
    return SPK_OK;
}
